// pa_device.h -- glue between the host plotfile containers and the C ABI (device levels / multifabs)
#pragma once
#include <chrono>
#include <future>
#include <memory>

#include "pa_plotfile.h"

namespace pa {

struct Ctx {
  pa_ctx* h = nullptr;
  explicit Ctx(int device = 0) {
    h = pa_ctx_create(device, nullptr);
    if (!h) Abort("no MI355X / HIP device available (this build has no CPU fallback)");
  }
  ~Ctx() { pa_ctx_destroy(h); }
  void check(int rc) const {
    if (rc != 0) Abort(pa_last_error(h));
  }
};

// HIP context brought up on a second thread (~0.25 s of runtime start-up) while the caller reads the plotfile
struct AsyncCtx {
  std::future<std::unique_ptr<Ctx>> fut;
  std::unique_ptr<Ctx> ctx;
  AsyncCtx() : fut(std::async(std::launch::async, [] { return std::unique_ptr<Ctx>(new Ctx()); })) {}
  Ctx& get() {
    if (!ctx) ctx = fut.get();
    return *ctx;
  }
};

// bench_json=1 on any tool's command line: one machine-readable line at exit with the wall time of each phase (the
// reference only has isosurface's wall-clock prints; SURVEY 5, metrics row).  {"tool": .., "cells": .., "phases_s": {..}}
struct PhaseTimer {
  std::string tool;
  bool on = false;
  long long cells = 0;
  std::vector<std::pair<std::string, double>> ph;
  double t0, tl;
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  PhaseTimer(const ParmParse& pp, const std::string& name) : tool(name), t0(now()), tl(t0) {
    int v = 0;
    pp.query("bench_json", v);
    on = v != 0;
  }
  void mark(const std::string& phase) {  // everything since the previous mark belongs to `phase`
    const double t = now();
    for (auto& p : ph)
      if (p.first == phase) { p.second += t - tl; tl = t; return; }
    ph.emplace_back(phase, t - tl);
    tl = t;
  }
  ~PhaseTimer() { report(); }
  void report() {  // once: at the end of main (before pa::Finish, which does not unwind) or from the destructor
    if (!on) return;
    on = false;
    std::cout << "{\"tool\": \"" << tool << "\", \"cells\": " << cells << ", \"total_s\": " << now() - t0 << ", \"phases_s\": {";
    for (size_t i = 0; i < ph.size(); ++i) std::cout << (i ? ", " : "") << "\"" << ph[i].first << "\": " << ph[i].second;
    std::cout << "}}" << std::endl;
  }
};

struct DevLevel {
  pa_level* h = nullptr;
  // owner (one rank per box) + rank + nranks: this context holds only the boxes with owner == rank (pa_level_create_sharded)
  DevLevel(const Ctx& c, const std::vector<Box3>& boxes, const Box3& dom, const int is_per[3], const double plo[3], const double phi[3],
           const std::vector<int32_t>* owner = nullptr, int rank = 0, int nranks = 1) {
    std::vector<int32_t> b6(6 * boxes.size());
    for (size_t i = 0; i < boxes.size(); ++i)
      for (int d = 0; d < 3; ++d) { b6[6 * i + d] = boxes[i].lo[d]; b6[6 * i + 3 + d] = boxes[i].hi[d]; }
    int32_t per[3] = {is_per[0], is_per[1], is_per[2]};
    if (owner && nranks > 1) h = pa_level_create_sharded(c.h, (int)boxes.size(), b6.data(), owner->data(), rank, nranks, dom.lo, dom.hi, per, plo, phi);
    else h = pa_level_create(c.h, (int)boxes.size(), b6.data(), dom.lo, dom.hi, per, plo, phi);
    if (!h) Abort(pa_last_error(c.h));
  }
  ~DevLevel() { pa_level_destroy(h); }
  DevLevel(const DevLevel&) = delete;
};

struct DevMF {
  pa_mf* h = nullptr;
  DevMF(const Ctx& c, const DevLevel& L, int ncomp, int ng) {
    h = pa_mf_create(c.h, L.h, ncomp, ng, nullptr);
    if (!h) Abort(pa_last_error(c.h));
  }
  ~DevMF() { pa_mf_destroy(h); }
  DevMF(const DevMF&) = delete;
};

inline void bc_from_flags(const std::vector<int>& is_per, const std::vector<int>& sym_dir, int32_t bc[3]) {
  // grad.cpp:180-193 / curvature.cpp:428-441
  for (int d = 0; d < 3; ++d) bc[d] = is_per[d] == 1 ? PA_BC_PERIODIC : (sym_dir[d] == 1 ? PA_BC_REFLECT_ODD : PA_BC_NEUMANN);
}

}  // namespace pa
