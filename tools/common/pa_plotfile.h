// pa_plotfile.h -- AMReX plotfile (HyperCLaw-V1.1) reader / writer and MEF writer for the tool
// drivers (Docs/source/data.rst:19-32, isosurface.cpp:2097-2134 of the
// reference).  Host-only C++17, no AMReX.
//
// HostMF keeps a level's data in the flat layout of pa_mf_layout (include/peleanalysis_amd.h), so
// a whole level goes to / from HBM with one pa_mf_upload / pa_mf_download.
#pragma once
#include <fcntl.h>
#include <ftw.h>
#include <limits.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <memory>
#include <regex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/peleanalysis_amd.h"
#include "pa_parmparse.h"

namespace pa {

// The plotfile side of a tool is host work over independent FABs (reads, the per-FAB min / max, writes): spread over the
// CPUs the process may use (affinity mask capped by the cgroup quota, at most 16)
inline int io_threads() {
  static const int n = [] {
    int v = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) v = std::min(v, CPU_COUNT(&set));
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32];
      long long per = 0;
      if (std::fscanf(f, "%31s %lld", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) v = std::min<long long>(v, std::max<long long>(1, (std::atoll(q) + per / 2) / per));
      std::fclose(f);
    }
    if (const char* e = std::getenv("PA_IO_THREADS")) v = std::atoi(e);
    return std::max(1, std::min(v, 16));
  }();
  return n;
}
// min / max of a run of doubles folded into (mn, mx): eight independent comparison chains instead of one (the one-chain loop
// `mn = std::min(mn, p[i])` is bound by the 4-cycle latency of minsd -- 0.75 GB/s per thread, 60 % of the plotfile writer's time).
// Same selections as std::min / std::max (a NaN is never selected), so the FAB minima / maxima in Cell_H are unchanged.
inline void minmax_run(const double* p, long long n, double& mn, double& mx) {
  double a[4] = {mn, mn, mn, mn}, b[4] = {mx, mx, mx, mx};
  long long i = 0;
  for (; i + 4 <= n; i += 4)
    for (int q = 0; q < 4; ++q) {
      const double v = p[i + q];
      a[q] = v < a[q] ? v : a[q];
      b[q] = b[q] < v ? v : b[q];
    }
  for (; i < n; ++i) {
    const double v = p[i];
    a[0] = v < a[0] ? v : a[0];
    b[0] = b[0] < v ? v : b[0];
  }
  for (int q = 0; q < 4; ++q) { mn = a[q] < mn ? a[q] : mn; mx = mx < b[q] ? b[q] : mx; }
}

template <typename F>
inline void parallel_for(size_t n, F fn) {
  const size_t nt = std::min<size_t>((size_t)io_threads(), n);
  if (nt <= 1) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; ++t)
    th.emplace_back([=] { for (size_t i = t; i < n; i += nt) fn(i); });
  for (auto& x : th) x.join();
}

struct Box3 {
  int lo[3], hi[3];
  long long numPts() const { return (long long)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1); }
};

struct LevelMeta {
  Box3 domain;
  std::vector<Box3> boxes;
  int level_step = 0;
  // where each FAB lives on disk
  std::vector<std::string> fab_file;
  std::vector<long long> fab_off;
};

struct PlotfileHeader {
  std::vector<std::string> names;
  double time = 0.0;
  int nlev = 0, dim = 3;
  double prob_lo[3], prob_hi[3];
  std::vector<int> ref_ratio;
  std::vector<LevelMeta> lev;
  std::string path;
  int comp(const std::string& n) const {
    for (size_t i = 0; i < names.size(); ++i)
      if (names[i] == n) return (int)i;
    return -1;
  }
};

// level data in pa_mf_layout order
// std::vector that leaves new doubles uninitialised: the multi-GB host multifabs are then first touched (page-faulted)
// by the filling threads instead of by one thread inside resize()
// Blocks of >= 4 MiB are 2-MiB aligned and marked MADV_HUGEPAGE (PA_HOST_THP=0: off): a level's data are tens of GB that are
// touched once by the reader, pinned once by the copy to the device and torn down at exit -- per 4-KiB page each time otherwise
template <typename T>
struct DefaultInitAlloc : std::allocator<T> {
  template <typename U> struct rebind { using other = DefaultInitAlloc<U>; };
  static bool thp() { static const bool on = [] { const char* e = getenv("PA_HOST_THP"); return !(e && !atoi(e)); }(); return on; }
  T* allocate(size_t n) {
    const size_t bytes = n * sizeof(T), huge = (size_t)2 << 20;
    if (bytes < 2 * huge || !thp()) return std::allocator<T>::allocate(n);
    void* p = nullptr;
    if (posix_memalign(&p, huge, (bytes + huge - 1) / huge * huge) != 0 || !p) throw std::bad_alloc();
    madvise(p, (bytes + huge - 1) / huge * huge, MADV_HUGEPAGE);
    return static_cast<T*>(p);
  }
  void deallocate(T* p, size_t n) noexcept {
    if (n * sizeof(T) < ((size_t)4 << 20) || !thp()) std::allocator<T>::deallocate(p, n);
    else free(p);
  }
  template <typename U> void construct(U* p) noexcept { ::new ((void*)p) U; }
  template <typename U, typename... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};

struct HostMF {
  std::vector<Box3> boxes;
  int ncomp = 0, ng = 0;
  std::vector<int64_t> off, cs;
  std::vector<double, DefaultInitAlloc<double>> data;
  void define(const std::vector<Box3>& b, int nc, int g, double fill = 0.0) {
    boxes = b; ncomp = nc; ng = g;
    std::vector<int32_t> b6(6 * b.size());
    for (size_t i = 0; i < b.size(); ++i)
      for (int d = 0; d < 3; ++d) { b6[6 * i + d] = b[i].lo[d]; b6[6 * i + 3 + d] = b[i].hi[d]; }
    off.resize(b.size()); cs.resize(b.size());
    const int64_t tot = pa_mf_layout((int)b.size(), b6.data(), nc, g, off.data(), cs.data());
    data.clear();
    data.resize((size_t)tot);  // no value-initialisation (DefaultInitAlloc)
    const size_t chunk = (size_t)1 << 22, nch = ((size_t)tot + chunk - 1) / chunk;
    double* d = data.data();
    parallel_for(nch, [=](size_t q) { std::fill(d + q * chunk, d + std::min((size_t)tot, (q + 1) * chunk), fill); });
  }
  double* ptr(int b, int c, int i, int j, int k) {
    const Box3& B = boxes[b];
    const long long nx = B.hi[0] - B.lo[0] + 1 + 2 * ng, ny = B.hi[1] - B.lo[1] + 1 + 2 * ng;
    return data.data() + off[b] + (long long)c * cs[b] + ((long long)(k - B.lo[2] + ng) * ny + (j - B.lo[1] + ng)) * nx + (i - B.lo[0] + ng);
  }
  std::vector<int32_t> boxes6() const {
    std::vector<int32_t> b6(6 * boxes.size());
    for (size_t i = 0; i < boxes.size(); ++i)
      for (int d = 0; d < 3; ++d) { b6[6 * i + d] = boxes[i].lo[d]; b6[6 * i + 3 + d] = boxes[i].hi[d]; }
    return b6;
  }
};

inline bool parse_box(const std::string& s, size_t& pos, Box3& b, int dim = 3) {
  static const std::regex re3(R"(\(\((-?\d+),(-?\d+),(-?\d+)\)\s*\((-?\d+),(-?\d+),(-?\d+)\)\s*\((-?\d+),(-?\d+),(-?\d+)\)\))");
  static const std::regex re2(R"(\(\((-?\d+),(-?\d+)\)\s*\((-?\d+),(-?\d+)\)\s*\((-?\d+),(-?\d+)\)\))");
  std::smatch m;
  std::string::const_iterator st = s.begin() + pos;
  if (!std::regex_search(st, s.end(), m, dim == 2 ? re2 : re3)) return false;
  if (dim == 2) {  // a 2-D box is the plane k = 0
    for (int d = 0; d < 2; ++d) { b.lo[d] = std::stoi(m[1 + d]); b.hi[d] = std::stoi(m[3 + d]); }
    b.lo[2] = b.hi[2] = 0;
  } else {
    for (int d = 0; d < 3; ++d) { b.lo[d] = std::stoi(m[1 + d]); b.hi[d] = std::stoi(m[4 + d]); }
  }
  pos += m.position(0) + m.length(0);
  return true;
}

// dim_wanted = 3: the 3-D tools (a 2-D plotfile aborts); 2: the 2-D build of isosurface (boxes become the plane k = 0)
inline PlotfileHeader read_header(const std::string& path, int dim_wanted = 3, bool any_ratio = false) {
  PlotfileHeader H;
  H.path = path;
  std::ifstream f(path + "/Header");
  if (!f) Abort("Unable to open plotfile Header: " + path + "/Header");
  std::string line;
  std::getline(f, line);  // version
  int ncomp;
  f >> ncomp;
  std::getline(f, line);
  H.names.resize(ncomp);
  for (int i = 0; i < ncomp; ++i) {
    std::getline(f, H.names[i]);
    while (!H.names[i].empty() && (H.names[i].back() == ' ' || H.names[i].back() == '\r')) H.names[i].pop_back();
  }
  int dim, finest;
  f >> dim >> H.time >> finest;
  if (dim != dim_wanted) Abort(dim_wanted == 3 ? "only 3-D plotfiles are supported by this build" : "this is the 2-D build: the plotfile is not 2-D");
  H.dim = dim;
  H.nlev = finest + 1;
  H.prob_lo[2] = 0.0;
  H.prob_hi[2] = 1.0;
  for (int d = 0; d < dim; ++d) f >> H.prob_lo[d];
  for (int d = 0; d < dim; ++d) f >> H.prob_hi[d];
  std::getline(f, line);
  std::getline(f, line);  // ref ratios (possibly empty)
  {
    std::stringstream ss(line);
    int r;
    while (ss >> r) H.ref_ratio.push_back(r);
  }
  std::getline(f, line);  // domains
  H.lev.resize(H.nlev);
  {
    size_t pos = 0;
    for (int l = 0; l < H.nlev; ++l)
      if (!parse_box(line, pos, H.lev[l].domain, dim)) Abort("bad domain line in plotfile Header");
  }
  for (int l = 0; l < H.nlev; ++l) f >> H.lev[l].level_step;
  std::getline(f, line);
  for (int l = 0; l < H.nlev; ++l) std::getline(f, line);  // dx (recomputed like amrex::Geometry)
  std::getline(f, line);                                   // coord sys
  std::getline(f, line);                                   // boundary width
  for (int l = 0; l < H.nlev; ++l) {
    int lev, ngrids, step;
    double t;
    f >> lev >> ngrids >> t >> step;
    double a, b;
    for (int g = 0; g < dim * ngrids; ++g) f >> a >> b;
    std::string rel;
    f >> rel;  // Level_n/Cell
    std::ifstream h(path + "/" + rel + "_H");
    if (!h) Abort("Unable to open " + path + "/" + rel + "_H");
    std::stringstream ss;
    ss << h.rdbuf();
    const std::string txt = ss.str();
    const size_t fod = txt.find("FabOnDisk");
    const std::string blk = txt.substr(0, fod == std::string::npos ? txt.size() : fod);
    size_t pos = blk.find('(');
    Box3 bx;
    while (parse_box(blk, pos, bx, dim)) H.lev[l].boxes.push_back(bx);
    if ((int)H.lev[l].boxes.size() != ngrids) Abort("Cell_H box count does not match the Header");
    const std::string dir = rel.substr(0, rel.rfind('/') + 1);
    size_t p = fod;
    while (p != std::string::npos) {
      std::stringstream ls(txt.substr(p + 10, 256));
      std::string fn;
      long long off;
      ls >> fn >> off;
      H.lev[l].fab_file.push_back(path + "/" + dir + fn);
      H.lev[l].fab_off.push_back(off);
      p = txt.find("FabOnDisk", p + 9);
    }
    if ((int)H.lev[l].fab_file.size() != ngrids) Abort("Cell_H FabOnDisk count does not match the Header");
  }
  // grad / curvature hard-code ratio 2 in the reference too (curvature.cpp:445, grad.cpp:255); isosurface and filterPlt take
  // the file's ratio (isosurface.cpp:1472,1518,1543; filterPlt.cpp:133,200) and pass any_ratio
  for (int r : H.ref_ratio) {
    if (r != 2 && H.nlev > 1 && !any_ratio) Abort("only refinement ratio 2 is supported (the reference tools write ratio 2 as well)");
    if (r < 2 && H.nlev > 1) Abort("bad refinement ratio in the plotfile Header");
  }
  return H;
}

// read component `comp` of the plotfile on level lev into component `dcomp` of dst (valid cells of
// every dst box that lies inside a file box: dst boxes are the file's or a re-chop of them)
inline void read_comp(const PlotfileHeader& H, int lev, int comp, HostMF& dst, int dcomp) {
  const LevelMeta& L = H.lev[lev];
  parallel_for(L.boxes.size(), [&](size_t fb) {  // file FABs are independent; destinations are disjoint valid regions
    // does any dst box intersect this file box?
    std::vector<int> hits;
    for (size_t b = 0; b < dst.boxes.size(); ++b) {
      bool in = true;
      for (int d = 0; d < 3; ++d) in = in && dst.boxes[b].lo[d] <= L.boxes[fb].hi[d] && dst.boxes[b].hi[d] >= L.boxes[fb].lo[d];
      if (in) hits.push_back((int)b);
    }
    if (hits.empty()) return;
    std::ifstream f(L.fab_file[fb], std::ios::binary);
    if (!f) Abort("Unable to open " + L.fab_file[fb]);
    f.seekg(L.fab_off[fb]);
    std::string hdr;
    std::getline(f, hdr);
    size_t pos = hdr.find(")))");  // end of the real descriptor
    Box3 fbx;
    if (pos == std::string::npos || !parse_box(hdr, pos, fbx, H.dim)) Abort("bad FAB header in " + L.fab_file[fb]);
    const long long n = fbx.numPts();
    // RealDescriptor of the FAB: "((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))" = little-endian IEEE doubles,
    // "((8, (32 8 23 0 1 9 0 127)),(4, (4 3 2 1)))" = little-endian IEEE floats (FABio::FAB_NATIVE_32, what AmrLevel-based
    // codes write by default); AmrData converts to Real on read, and so does this
    int nbytes = 0;
    {
      const size_t q = hdr.find("),(");
      if (q != std::string::npos) nbytes = std::atoi(hdr.c_str() + q + 3);
      const bool le8 = hdr.find("(8, (8 7 6 5 4 3 2 1))") != std::string::npos, le4 = hdr.find("(4, (4 3 2 1))") != std::string::npos;
      if (!((nbytes == 8 && le8) || (nbytes == 4 && le4))) Abort("unsupported FAB RealDescriptor in " + L.fab_file[fb] + ": " + hdr.substr(0, 80));
    }
    std::vector<double, DefaultInitAlloc<double>> buf((size_t)n);  // (not zero-filled; huge pages)
    f.seekg((long long)f.tellg() + (long long)comp * n * nbytes);
    if (nbytes == 8) {
      f.read((char*)buf.data(), n * 8);
    } else {
      std::vector<float, DefaultInitAlloc<float>> b4((size_t)n);
      f.read((char*)b4.data(), n * 4);
      for (long long i = 0; i < n; ++i) buf[(size_t)i] = (double)b4[(size_t)i];
    }
    if (!f) Abort("short read in " + L.fab_file[fb]);
    const long long fx = fbx.hi[0] - fbx.lo[0] + 1, fy = fbx.hi[1] - fbx.lo[1] + 1;
    for (int b : hits) {
      const Box3& B = dst.boxes[b];
      for (int k = std::max(B.lo[2], L.boxes[fb].lo[2]); k <= std::min(B.hi[2], L.boxes[fb].hi[2]); ++k)
        for (int j = std::max(B.lo[1], L.boxes[fb].lo[1]); j <= std::min(B.hi[1], L.boxes[fb].hi[1]); ++j) {
          const int i0 = std::max(B.lo[0], L.boxes[fb].lo[0]), i1 = std::min(B.hi[0], L.boxes[fb].hi[0]);
          std::memcpy(dst.ptr(b, dcomp, i0, j, k), &buf[((long long)(k - fbx.lo[2]) * fy + (j - fbx.lo[1])) * fx + (i0 - fbx.lo[0])],
                      sizeof(double) * (size_t)(i1 - i0 + 1));
        }
    }
  });
}

inline std::string box_str(const Box3& b) {
  char s[160];
  std::snprintf(s, sizeof s, "((%d,%d,%d) (%d,%d,%d) (0,0,0))", b.lo[0], b.lo[1], b.lo[2], b.hi[0], b.hi[1], b.hi[2]);
  return s;
}
inline std::string g17(double v) {
  char s[64];
  std::snprintf(s, sizeof s, "%.17g", v);
  return s;
}

// WriteMultiLevelPlotfile restated: valid cells of comps [0, names.size()) of each level's HostMF
inline void write_plotfile(const std::string& path, const std::vector<std::string>& names, const std::vector<Box3>& domains,
                           const double prob_lo[3], const double prob_hi[3], std::vector<HostMF>& mf, double time,
                           const std::vector<int>& level_steps, int ref_ratio = 2, int dim = 3, const std::vector<int>* comps = nullptr,
                           const std::vector<std::vector<Box3>>* file_boxes = nullptr, const std::function<void(int)>* wait_level = nullptr) {
  // dim = 2: the levels are one plane of cells (k = 0) and the file is what a 2-D AMReX code writes; comps: the HostMF
  // component behind each name (default: 0, 1, 2, ...); file_boxes: the BoxArray the file gets on each level when the
  // multifabs live on ANOTHER tiling of the same cells (retile_levels below: the tools compute on merged boxes and write
  // the input's BoxArray, as grad.cpp:256 does) -- every FAB is then gathered from the multifab boxes it intersects
  const int nlev = (int)mf.size(), ncomp = (int)names.size();
  auto wboxes = [&](int l) -> const std::vector<Box3>& { return file_boxes ? (*file_boxes)[l] : mf[l].boxes; };
  auto bstr = [dim](const Box3& b) {
    if (dim == 3) return box_str(b);
    char s[160];
    std::snprintf(s, sizeof s, "((%d,%d) (%d,%d) (0,0))", b.lo[0], b.lo[1], b.hi[0], b.hi[1]);
    return std::string(s);
  };
  auto src = [comps](int c) { return comps ? (*comps)[c] : c; };
  ::mkdir(path.c_str(), 0755);
  {
    std::ofstream f(path + "/Header");
    if (!f) Abort("Unable to create " + path + "/Header");
    f << "HyperCLaw-V1.1\n" << ncomp << "\n";
    for (auto& n : names) f << n << "\n";
    f << dim << "\n" << g17(time) << "\n" << nlev - 1 << "\n";
    for (int d = 0; d < dim; ++d) f << g17(prob_lo[d]) << ' ';
    f << "\n";
    for (int d = 0; d < dim; ++d) f << g17(prob_hi[d]) << ' ';
    f << "\n";
    for (int l = 0; l < nlev - 1; ++l) f << ref_ratio << ' ';
    f << "\n";
    for (int l = 0; l < nlev; ++l) f << bstr(domains[l]) << ' ';
    f << "\n";
    for (int l = 0; l < nlev; ++l) f << level_steps[l] << ' ';
    f << "\n";
    for (int l = 0; l < nlev; ++l) {
      for (int d = 0; d < dim; ++d) f << g17((prob_hi[d] - prob_lo[d]) / (double)(domains[l].hi[d] - domains[l].lo[d] + 1)) << ' ';
      f << "\n";
    }
    f << "0\n0\n";
    for (int l = 0; l < nlev; ++l) {
      f << l << ' ' << wboxes(l).size() << ' ' << g17(time) << "\n" << level_steps[l] << "\n";
      for (auto& B : wboxes(l))
        for (int d = 0; d < dim; ++d) {
          const double dx = (prob_hi[d] - prob_lo[d]) / (double)(domains[l].hi[d] - domains[l].lo[d] + 1);
          f << g17(prob_lo[d] + B.lo[d] * dx) << ' ' << g17(prob_lo[d] + (B.hi[d] + 1) * dx) << "\n";
        }
      f << "Level_" << l << "/Cell\n";
    }
  }
  for (int l = 0; l < nlev; ++l) {
    const std::string dir = path + "/Level_" + std::to_string(l);
    ::mkdir(dir.c_str(), 0755);
    if (wait_level) (*wait_level)(l);  // the level's data may still be on its way from the device (LevelGate below)
    HostMF& M = mf[l];
    const std::vector<Box3>& WB = wboxes(l);
    const size_t nb = WB.size();
    std::vector<long long> offs(nb);
    std::vector<std::vector<double>> mins(nb), maxs(nb);
    {
      // FAB b = header line + ncomp * npts doubles at a known offset: the FABs are gathered, reduced (min / max) and
      // written by several threads (pwrite)
      std::vector<std::string> hdr(nb);
      long long pos = 0;
      for (size_t b = 0; b < nb; ++b) {
        hdr[b] = "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))" + bstr(WB[b]) + ' ' + std::to_string(ncomp) + "\n";
        offs[b] = pos;
        pos += (long long)hdr[b].size() + (long long)ncomp * WB[b].numPts() * 8;
      }
      const std::string fname = dir + "/Cell_D_00000";
      const int fd = ::open(fname.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
      if (fd < 0) Abort("Unable to create " + fname);
      std::vector<int> bad(nb, 0);
      auto put = [&](size_t b, const char* p, size_t n, long long at) {
        size_t done = 0;
        while (done < n) {
          const ssize_t r = ::pwrite(fd, p + done, n - done, (off_t)(at + (long long)done));
          if (r <= 0) { bad[b] = 1; return; }
          done += (size_t)r;
        }
      };
      parallel_for(nb, [&](size_t b) {
        const Box3& B = WB[b];
        const int nx = B.hi[0] - B.lo[0] + 1;
        const long long npts = B.numPts();
        if (file_boxes) {  // the FAB's cells lie in one or more boxes of the multifab's tiling: gathered run by run
          std::vector<char, DefaultInitAlloc<char>> buf(hdr[b].size() + (size_t)ncomp * (size_t)npts * 8);  // (not zero-filled; huge pages)
          std::memcpy(buf.data(), hdr[b].data(), hdr[b].size());
          char* base = buf.data() + hdr[b].size();
          const long long ny = B.hi[1] - B.lo[1] + 1;
          long long got = 0;
          for (size_t t = 0; t < M.boxes.size(); ++t) {
            const Box3& T = M.boxes[t];
            int lo[3], hi[3];
            bool in = true;
            for (int d = 0; d < 3; ++d) { lo[d] = std::max(B.lo[d], T.lo[d]); hi[d] = std::min(B.hi[d], T.hi[d]); in = in && lo[d] <= hi[d]; }
            if (!in) continue;
            const size_t run = sizeof(double) * (size_t)(hi[0] - lo[0] + 1);
            for (int c = 0; c < ncomp; ++c)
              for (int k = lo[2]; k <= hi[2]; ++k)
                for (int j = lo[1]; j <= hi[1]; ++j)
                  std::memcpy(base + 8 * ((long long)c * npts + ((long long)(k - B.lo[2]) * ny + (j - B.lo[1])) * nx + (lo[0] - B.lo[0])), M.ptr((int)t, src(c), lo[0], j, k), run);
            got += (long long)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
          }
          if (got != npts) { bad[b] = 1; return; }  // the two tilings do not hold the same cells
          std::vector<double> mn(ncomp, 1e300), mx(ncomp, -1e300);
          std::vector<double> tmp;
          for (int c = 0; c < ncomp; ++c) {
            const char* p = base + 8 * (long long)c * npts;
            if (((uintptr_t)p & 7u) == 0) minmax_run((const double*)p, npts, mn[c], mx[c]);
            else {  // the header's length leaves the data unaligned: through an aligned copy
              tmp.resize((size_t)npts);
              std::memcpy(tmp.data(), p, (size_t)npts * 8);
              minmax_run(tmp.data(), npts, mn[c], mx[c]);
            }
          }
          mins[b] = mn; maxs[b] = mx;
          put(b, buf.data(), buf.size(), offs[b]);
          return;
        }
        if (M.ng == 0) {  // a component of a ghost-free FAB is one contiguous run: written straight from the multifab
          std::vector<double> mn(ncomp, 1e300), mx(ncomp, -1e300);
          put(b, hdr[b].data(), hdr[b].size(), offs[b]);
          for (int c = 0; c < ncomp; ++c) {
            const double* p = M.ptr((int)b, src(c), B.lo[0], B.lo[1], B.lo[2]);
            minmax_run(p, npts, mn[c], mx[c]);
            put(b, (const char*)p, (size_t)npts * 8, offs[b] + (long long)hdr[b].size() + (long long)c * npts * 8);
          }
          mins[b] = mn; maxs[b] = mx;
          return;
        }
        std::vector<char, DefaultInitAlloc<char>> buf(hdr[b].size() + (size_t)ncomp * (size_t)npts * 8);  // (not zero-filled; huge pages)
        std::memcpy(buf.data(), hdr[b].data(), hdr[b].size());
        double* out = (double*)(buf.data() + hdr[b].size());  // may be unaligned: filled with memcpy
        char* w = (char*)out;
        std::vector<double> mn(ncomp, 1e300), mx(ncomp, -1e300);
        for (int c = 0; c < ncomp; ++c)
          for (int k = B.lo[2]; k <= B.hi[2]; ++k)
            for (int j = B.lo[1]; j <= B.hi[1]; ++j) {
              const double* p = M.ptr((int)b, src(c), B.lo[0], j, k);
              std::memcpy(w, p, sizeof(double) * (size_t)nx);
              w += sizeof(double) * (size_t)nx;
              minmax_run(p, nx, mn[c], mx[c]);
            }
        mins[b] = mn; maxs[b] = mx;
        put(b, buf.data(), buf.size(), offs[b]);
      });
      ::close(fd);
      for (int x : bad) if (x) Abort("short write to " + fname + (file_boxes ? " (or the output BoxArray does not match the computed tiling)" : ""));
    }
    std::ofstream h(dir + "/Cell_H");
    h << "1\n1\n" << ncomp << "\n0\n(" << nb << " 0\n";
    for (auto& B : WB) h << bstr(B) << "\n";
    h << ")\n" << nb << "\n";
    for (size_t b = 0; b < nb; ++b) h << "FabOnDisk: Cell_D_00000 " << offs[b] << "\n";
    h << "\n" << nb << "," << ncomp << "\n";
    for (auto& m : mins) { for (double v : m) h << g17(v) << ","; h << "\n"; }
    h << "\n" << nb << "," << ncomp << "\n";
    for (auto& m : maxs) { for (double v : m) h << g17(v) << ","; h << "\n"; }
  }
}

// MEF surface file (isosurface.cpp:2097-2134): label, names, "nElts nodesPerElt", FAB of Box
// (0..N-1,0,0) x ncomp written node-major, then raw 1-based int32 connectivity, no trailing newline
inline void write_mef(const std::string& file, double time, const std::vector<std::string>& names, const std::vector<double>& nodes /* [N][ncomp] */,
                      const std::vector<int32_t>& elts0 /* [M][3], 0-based */) {
  const int ncomp = (int)names.size();
  const long long N = ncomp ? (long long)nodes.size() / ncomp : 0, M = (long long)elts0.size() / 3;
  std::ofstream f(file, std::ios::binary);
  if (!f) Abort("Unable to create " + file);
  char lab[64];
  std::snprintf(lab, sizeof lab, "%g", time);
  f << lab << "\n";
  for (int c = 0; c < ncomp; ++c) f << names[c] << (c + 1 < ncomp ? " " : "");
  f << "\n" << M << " 3\n";
  Box3 b{{0, 0, 0}, {(int)N - 1, 0, 0}};
  f << "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))" << box_str(b) << ' ' << ncomp << "\n";
  f.write((const char*)nodes.data(), sizeof(double) * nodes.size());
  std::vector<int32_t> e1(elts0.size());
  for (size_t q = 0; q < elts0.size(); ++q) e1[q] = elts0[q] + 1;
  f.write((const char*)e1.data(), sizeof(int32_t) * e1.size());
}

// reader of the same file, as the MEF consumers parse it (surfMEFtoDAT.cpp:46-72, checkIso.cpp:84-120)
struct MefSurface {
  std::string title;
  std::vector<std::string> names;
  long long nElts = 0, nodesPerElt = 0, nNodes = 0;
  std::vector<double> nodes;   // [nNodes][names.size()], node-major as stored
  std::vector<int32_t> conn;   // [nElts][nodesPerElt], 1-based as stored
};
inline MefSurface read_mef(const std::string& file) {
  MefSurface S;
  std::ifstream is(file, std::ios::in | std::ios::binary);
  if (!is) Abort("Unable to open file : " + file);
  std::string line;
  std::getline(is, S.title);  // parseTitle
  std::getline(is, line);     // parseVarNames: tokens separated by ", "
  {
    std::string t;
    for (char ch : line) {
      if (ch == ' ' || ch == ',') { if (!t.empty()) S.names.push_back(t); t.clear(); }
      else t.push_back(ch);
    }
    if (!t.empty()) S.names.push_back(t);
  }
  is >> S.nElts >> S.nodesPerElt;
  std::getline(is, line);
  std::getline(is, line);  // FAB header: "... ((0,0,0) (N-1,0,0) (0,0,0)) ncomp"
  const size_t p0 = line.rfind("((0,0,0) (");
  if (p0 == std::string::npos || S.nElts < 0 || S.nodesPerElt < 0) Abort("cannot parse the node FAB header of " + file);
  S.nNodes = std::atoll(line.c_str() + p0 + 10) + 1;
  S.nodes.resize((size_t)S.nNodes * S.names.size());
  is.read((char*)S.nodes.data(), sizeof(double) * S.nodes.size());
  S.conn.assign((size_t)S.nElts * S.nodesPerElt, 0);
  is.read((char*)S.conn.data(), sizeof(int32_t) * S.conn.size());
  if (!is) Abort("truncated MEF file " + file);
  return S;
}

// XDMF surface (isosurface.cpp:2135-2229): <base>.xmf (XML, precision 8) + <base>.mesh = raw int32 0-based
// connectivity, then xyz per node, then each mapped component as one array of doubles
inline void write_xdmf(const std::string& base, double time, const std::string& isoCompName, double isoVal, const std::vector<std::string>& varnames,
                       const std::vector<double>& nodes /* [N][dim+nvar] */, const std::vector<int32_t>& elts0 /* [M][dim], 0-based */, int dim = 3) {
  // dim = 2: the 2-D build -- Polyline topology with two nodes per element, XY geometry (isosurface.cpp:2168-2183)
  const int nvar = (int)varnames.size(), nc = dim + nvar;
  const size_t N = nodes.size() / (size_t)nc, M = elts0.size() / (size_t)dim;
  const std::string mesh = base + ".mesh";
  std::ofstream x(base + ".xmf");
  if (!x) Abort("Unable to create " + base + ".xmf");
  x.precision(8);
  size_t seek = sizeof(int32_t) * elts0.size();
  x << "<?xml version=\"1.0\"?>\n<Xdmf Version=\"3.0\" xmlns:xi=\"http://www.w3.org/2001/XInclude\">\n   <Domain>\n      <Grid Name=\"isoSurface\">\n";
  x << "      <Information Name=\"Variable\" Value=\"" << isoCompName << "\"/>\n";
  x << "      <Information Name=\"IsoValue\" Value=\"" << isoVal << "\"/>\n";
  x << "      <Time Value=\"" << time << "\"/>\n";
  if (dim == 2) x << "         <Topology TopologyType=\"Polyline\" NodesPerElement=\"2\" NumberOfElements=\"" << M << "\">\n";
  else x << "         <Topology TopologyType=\"Triangle\" NumberOfElements=\"" << M << "\">\n";
  x << "            <DataItem Name=\"Conn\" Format=\"Binary\" DataType=\"Int\" Dimensions=\"" << dim * M << "\">\n               " << mesh << "\n            </DataItem>\n";
  x << "         </Topology>\n         <Geometry GeometryType=\"" << (dim == 2 ? "XY" : "XYZ") << "\">\n";
  x << "            <DataItem Name=\"Coord\" Format=\"Binary\" Precision=\"8\" DataType=\"Float\" Seek=\"" << seek << "\" Dimensions=\"" << dim * N << "\">\n               " << mesh
    << "\n            </DataItem>\n         </Geometry>\n";
  seek += dim * N * sizeof(double);
  for (int c = 0; c < nvar; ++c) {
    x << "         <Attribute Name=\"" << varnames[c] << "\" AttributeType=\"Scalar\" Center=\"Node\">\n";
    x << "            <DataItem Format=\"Binary\" Precision=\"8\" DataType=\"Float\" Seek=\"" << seek << "\" Dimensions=\"" << N << "\">\n               " << mesh
      << "\n            </DataItem>\n         </Attribute>\n";
    seek += N * sizeof(double);
  }
  x << "      </Grid>\n   </Domain>\n</Xdmf>\n";
  std::ofstream f(mesh, std::ios::binary | std::ios::trunc);
  if (!f) Abort("Unable to create " + mesh);
  f.write((const char*)elts0.data(), sizeof(int32_t) * elts0.size());
  for (size_t q = 0; q < N; ++q) f.write((const char*)&nodes[q * nc], sizeof(double) * dim);
  for (int c = 0; c < nvar; ++c)
    for (size_t q = 0; q < N; ++q) f.write((const char*)&nodes[q * nc + dim + c], sizeof(double));
}

// BoxArray::maxSize: chop every box into pieces <= n per direction (even split)
inline std::vector<Box3> max_size(const std::vector<Box3>& in, int n) {
  std::vector<Box3> out;
  for (const Box3& B : in) {
    std::vector<std::pair<int, int>> cut[3];
    for (int d = 0; d < 3; ++d) {
      const int len = B.hi[d] - B.lo[d] + 1, parts = (len + n - 1) / n, base = len / parts, rem = len % parts;
      int s = B.lo[d];
      for (int p = 0; p < parts; ++p) {
        const int sz = base + (p < rem ? 1 : 0);
        cut[d].push_back({s, s + sz - 1});
        s += sz;
      }
    }
    for (auto& z : cut[2])
      for (auto& y : cut[1])
        for (auto& x : cut[0]) out.push_back(Box3{{x.first, y.first, z.first}, {x.second, y.second, z.second}});
  }
  return out;
}

// An output plotfile that already exists (a tool run again on the same input): AMReX's UtilCreateCleanDirectory -- called by
// WriteMultiLevelPlotfile at WRITE time (grad.cpp:256, curvature.cpp:843, filterPlt.cpp:52) -- renames the old directory to
// <path>.old.<unique> and KEEPS it.  Same here: move_away() is called once the header, the variables and the parameters are
// validated, immediately before the first byte is written; the old directory is renamed (writing over the old files made the
// kernel drop their cached pages inside the timed write: 16 GB at 3.7 GB/s against ~12 GB/s into a fresh directory,
// profiles/r05_tool_e2e_512.txt) and kept.  remove_old_output=1 removes it -- in finish(), i.e. only after the new plotfile is
// completely written, so a run that aborts in between leaves the old output intact under its .old name.
// Guards (round-5 advisor): only a PLOTFILE directory (Header + Level_0) is ever renamed -- any other existing directory is
// written into as it is -- and an output path that is the input or contains it aborts before anything is touched.
struct OldOutput {
  std::string old;
  bool remove_after = false;
  static int rm_entry(const char* p, const struct stat*, int, struct FTW*) { return ::remove(p); }
  static bool is_plotfile_dir(const std::string& path) {
    struct stat st;
    if (::stat((path + "/Header").c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return false;
    return ::stat((path + "/Level_0").c_str(), &st) == 0 && S_ISDIR(st.st_mode);
  }
  // input: the plotfile the tool has read its header from
  void move_away(const std::string& path, const std::string& input, const ParmParse& pp) {
    int rm = 0;
    pp.query("remove_old_output", rm);
    remove_after = rm != 0;
    struct stat st;
    if (::stat(path.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return;
    char rp[PATH_MAX], ri[PATH_MAX];
    if (!input.empty() && ::realpath(path.c_str(), rp) && ::realpath(input.c_str(), ri)) {
      const std::string P(rp), I(ri);
      if (I == P) return;  // outfile=<infile>: written over in place, the data are in memory by then
      if (I.size() > P.size() && I.compare(0, P.size(), P) == 0 && I[P.size()] == '/')
        Abort("the output path " + path + " contains the input plotfile " + input);
    }
    if (!is_plotfile_dir(path)) return;  // not something this tool (or the reference) wrote: left alone
    old = path + ".old." + std::to_string((long)::getpid());
    if (::rename(path.c_str(), old.c_str()) != 0) old.clear();  // left in place: the writer truncates the files
  }
  // the new plotfile is complete
  void finish() {
    if (remove_after && !old.empty()) ::nftw(old.c_str(), rm_entry, 64, FTW_DEPTH | FTW_PHYS);
    old.clear();
  }
};

// Levels handed from the thread that downloads them to the thread that writes them: the writer starts on level l as soon as it is
// on the host while the levels after it are still coming down (the plotfile write is 60-70 % of a tool's wall time and the
// downloads 5-15 %: tools/tool_e2e.py).  done(l): level l (and every level before it) is complete; wait(l) blocks until then.
struct LevelGate {
  std::mutex m;
  std::condition_variable cv;
  int ready = 0;  // levels 0 .. ready-1 are complete
  void done(int l) {
    { std::lock_guard<std::mutex> g(m); ready = std::max(ready, l + 1); }
    cv.notify_all();
  }
  void wait(int l) {
    std::unique_lock<std::mutex> g(m);
    cv.wait(g, [&] { return ready > l; });
  }
};

// Internal re-tiling (pa_level_retile, include/peleanalysis_amd.h): the boxes the tools COMPUTE on for each level -- the file's
// cells merged into the largest rectangles -- while their output keeps the file's BoxArray (write_plotfile's file_boxes).
// retile=0 on the command line: the file's boxes as they are.  read_comp fills such boxes from every file FAB they intersect.
// max_cells > 0: that limit per direction instead of pa_hierarchy_retile_limits' choice (filterPlt: 128 -- k_filter_sep stages whole
// rows of a box in LDS and deals boxes, not tiles, to the XCDs: one 256^3 box per level ran 1.25 ms against 0.25 for eight of 128^3)
inline std::vector<std::vector<Box3>> retile_levels(const std::vector<std::vector<Box3>>& file_boxes, const ParmParse& pp, int max_cells = 0) {
  int on = 1;
  pp.query("retile", on);
  if (!on) return file_boxes;
  const int nlev = (int)file_boxes.size();
  std::vector<std::vector<int32_t>> b6(nlev);
  std::vector<int32_t> nb(nlev);
  std::vector<const int32_t*> ptr(nlev);
  for (int l = 0; l < nlev; ++l) {
    nb[l] = (int32_t)file_boxes[l].size();
    for (const Box3& B : file_boxes[l]) {
      for (int d = 0; d < 3; ++d) b6[l].push_back(B.lo[d]);
      for (int d = 0; d < 3; ++d) b6[l].push_back(B.hi[d]);
    }
    ptr[l] = b6[l].data();
  }
  int32_t mx[3] = {max_cells, max_cells, max_cells};
  int ngpus = 1;
  pp.query("ngpus", ngpus);  // a hierarchy dealt to several GPUs keeps at least four boxes per level and rank
  if (max_cells <= 0 && pa_hierarchy_retile_limits_ranks(nlev, nb.data(), ptr.data(), 3, std::max(ngpus, 1), mx) != 0) return file_boxes;
  std::vector<std::vector<Box3>> out(nlev);
  for (int l = 0; l < nlev; ++l) {
    const int cap = 4 * nb[l] + 16;
    std::vector<int32_t> o6((size_t)cap * 6);
    const int n = nb[l] > 0 ? pa_level_retile(nb[l], b6[l].data(), mx, 3, o6.data(), cap) : 0;
    if (n <= 0) { out[l] = file_boxes[l]; continue; }
    for (int b = 0; b < n; ++b) out[l].push_back(Box3{{o6[6 * b], o6[6 * b + 1], o6[6 * b + 2]}, {o6[6 * b + 3], o6[6 * b + 4], o6[6 * b + 5]}});
  }
  return out;
}
// write_plotfile's file_boxes argument: null when the computed tiling IS the file's (the writer then streams ghost-free FABs
// straight from the multifab)
inline const std::vector<std::vector<Box3>>* boxes_if_retiled(const std::vector<std::vector<Box3>>& file_boxes, const std::vector<std::vector<Box3>>& tile) {
  bool same = file_boxes.size() == tile.size();
  for (size_t l = 0; same && l < tile.size(); ++l) {
    same = file_boxes[l].size() == tile[l].size();
    for (size_t b = 0; same && b < tile[l].size(); ++b) same = std::memcmp(&file_boxes[l][b], &tile[l][b], sizeof(Box3)) == 0;
  }
  return same ? nullptr : &file_boxes;
}
inline std::vector<std::vector<Box3>> level_boxes(const PlotfileHeader& H, int nlev) {
  std::vector<std::vector<Box3>> b;
  for (int l = 0; l < nlev; ++l) b.push_back(H.lev[l].boxes);
  return b;
}

}  // namespace pa
