// pa_stream.hip -- streamline tracer of partStream.cpp:121-207 / StreamPC.cpp on gfx950:
// RK4 through a piecewise-trilinear vector field on the AMR hierarchy, two lines per seed (forward and
// backward), one thread per line.  Same arithmetic, operation order and quirks as the reference (vnrml's
// 1e12 test, the step cut of StreamPC.cpp:247 as written, the +-1e-10 clamp), and the same FAB choice:
// a line keeps interpolating from the (level, grid) it was last assigned to, ghost cells included, and
// when ANY live line has left its grid grown by nGrow-1 cells ALL lines are re-assigned to the finest level
// containing them (SetParticleLocation :88-141 -> Redistribute).  Two launches per step, no host round
// trip: k_stream_check raises the step's flag, k_stream_step re-assigns if it is set and advances.
// Gather-bound (8 x 3 scattered loads per stage, 4 stages per step), thousands of lines: latency-bound.
#include "pa_internal.h"
#include <cmath>
#include <vector>

#define PA_STREAM_MAXLEV 8
struct StreamLevels {
  int nlev, ng, vcomp;
  DLevelView L[PA_STREAM_MAXLEV];
  DMFView V[PA_STREAM_MAXLEV];
  double dx[PA_STREAM_MAXLEV][3], plo[3], phi[3];
};

__device__ __forceinline__ void s_vnrml(double vec[3], int dir) {  // StreamPC.cpp:143-157
  double sum = vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2];
  if (sum < 1.e12) {
    sum = 1. / sqrt(sum);
#pragma unroll
    for (int i = 0; i < 3; ++i) vec[i] *= dir * sum;
  } else {
    vec[0] = vec[1] = vec[2] = 0.0;
  }
}

// StreamPC.cpp:159-206 inside FAB (lev, b)
__device__ __forceinline__ bool s_ntrpv(const StreamLevels& S, int lev, int b, const DBox& B, const double x[3], double u[3]) {
  int bi[3];
  double n[3];
  const int ng = S.ng;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    bi[d] = (int)floor((x[d] - S.plo[d]) / S.dx[lev][d] - 0.5);
    n[d] = (x[d] - ((bi[d] + 0.5) * S.dx[lev][d] + S.plo[d])) / S.dx[lev][d];
    n[d] = (n[d] < 1.) ? n[d] : 1.;
    n[d] = (0. < n[d]) ? n[d] : 0.;
    if (bi[d] < B.lo[d] - ng || bi[d] > B.hi[d] + ng - 1) return false;
  }
  const DMFView& V = S.V[lev];
  const long long nx = B.hi[0] - B.lo[0] + 1 + 2 * ng, ny = B.hi[1] - B.lo[1] + 1 + 2 * ng, nz = B.hi[2] - B.lo[2] + 1 + 2 * ng;
  const long long cs = pa_cstride(nx * ny * nz, V.ncomp), sy = nx, sz = nx * ny;
  const double* g0 = V.data + V.off[b] + ((long long)(bi[2] - B.lo[2] + ng) * ny + (bi[1] - B.lo[1] + ng)) * nx + (bi[0] - B.lo[0] + ng);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double* g = g0 + (long long)(S.vcomp + i) * cs;
    u[i] = +n[0] * n[1] * n[2] * g[1 + sy + sz]
           + n[0] * (1 - n[1]) * n[2] * g[1 + sz]
           + n[0] * n[1] * (1 - n[2]) * g[1 + sy]
           + n[0] * (1 - n[1]) * (1 - n[2]) * g[1]
           + (1 - n[0]) * n[1] * n[2] * g[sy + sz]
           + (1 - n[0]) * (1 - n[1]) * n[2] * g[sz]
           + (1 - n[0]) * n[1] * (1 - n[2]) * g[sy]
           + (1 - n[0]) * (1 - n[1]) * (1 - n[2]) * g[0];
  }
  return true;
}

__device__ __forceinline__ bool s_rk4(const StreamLevels& S, int lev, int b, const DBox& B, double x[3], double dt, int dir) {  // :208-260
  double vec[3], k1[3], k2[3], k3[3], k4[3], xx[3] = {x[0], x[1], x[2]};
  if (!s_ntrpv(S, lev, b, B, xx, vec)) return false;
  s_vnrml(vec, dir);
  for (int d = 0; d < 3; ++d) { k1[d] = vec[d] * dt; xx[d] = x[d] + k1[d] * 0.5; }
  if (!s_ntrpv(S, lev, b, B, xx, vec)) return false;
  s_vnrml(vec, dir);
  for (int d = 0; d < 3; ++d) { k2[d] = vec[d] * dt; xx[d] = x[d] + k2[d] * 0.5; }
  if (!s_ntrpv(S, lev, b, B, xx, vec)) return false;
  s_vnrml(vec, dir);
  for (int d = 0; d < 3; ++d) { k3[d] = vec[d] * dt; xx[d] = x[d] + k3[d]; }
  if (!s_ntrpv(S, lev, b, B, xx, vec)) return false;
  s_vnrml(vec, dir);
  const double third = 1. / 3., sixth = 1. / 6.;
  double delta[3];
  for (int d = 0; d < 3; ++d) {
    k4[d] = vec[d] * dt;
    delta[d] = (k1[d] + k4[d]) * sixth + (k2[d] + k3[d]) * third;
  }
  double scale = 1;
  for (int d = 0; d < 3; ++d) {
    if (x[d] + delta[d] < S.plo[d]) { const double s = fabs((x[d] - S.plo[d]) / delta[d]); scale = (s < scale) ? s : scale; }
    if (x[d] + delta[d] > S.plo[d]) { const double s = fabs((S.phi[d] - x[d]) / delta[d]); scale = (s < scale) ? s : scale; }  // :247 as written
  }
  for (int d = 0; d < 3; ++d) {
    x[d] += scale * delta[d];
    const double lo = S.plo[d] + 1.e-10, hi = S.phi[d] - 1.e-10;
    const double m = (lo < x[d]) ? x[d] : lo;
    x[d] = (m < hi) ? m : hi;
  }
  return true;
}

__device__ __forceinline__ void s_where(const StreamLevels& S, const double x[3], int& lev, int& grid) {  // Redistribute -> Where()
  for (int l = S.nlev - 1; l >= 0; --l) {
    int p[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) p[d] = (int)floor((x[d] - S.plo[d]) / S.dx[l][d]);
    const int b = owner_of(S.L[l], p);
    if (b >= 0) { lev = l; grid = b; return; }
  }
  lev = -1; grid = -1;
}

__global__ __launch_bounds__(256) void k_stream_init(StreamLevels S, long long np, const double* seeds, int nsteps, double* pos, int* lev, int* grd) {
  const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (p >= np) return;
  double x[3];
  for (int d = 0; d < 3; ++d) { x[d] = seeds[(p / 2) * 3 + d]; pos[(p * nsteps) * 3 + d] = x[d]; }
  s_where(S, x, lev[p], grd[p]);
}

// SetParticleLocation(step, nGrow): has any live line left its grid grown by nGrow-1?
__global__ __launch_bounds__(256) void k_stream_check(StreamLevels S, long long np, int nsteps, int step, const double* pos, const int* lev, const int* grd, int* flags) {
  const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  bool out = false;
  if (p < np && lev[p] >= 0) {
    const int l = lev[p];
    const DBox B = S.L[l].boxes[grd[p]];
    for (int d = 0; d < 3; ++d) {
      const double blo = S.plo[d] + (B.lo[d] - (S.ng - 1)) * S.dx[l][d], bhi = S.plo[d] + (B.hi[d] + (S.ng - 1) + 1) * S.dx[l][d];
      const double x = pos[(p * nsteps + step) * 3 + d];
      out = out || (x < blo || x > bhi);
    }
  }
  if (__builtin_amdgcn_ballot_w64(out) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&flags[step], 1);
}

// [Redistribute if flagged] + ComputeNextLocation(step)
__global__ __launch_bounds__(256) void k_stream_step(StreamLevels S, long long np, int nsteps, int step, double dt, double* pos, int* lev, int* grd, const int* flags,
                                                     int* bad) {
  const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (p >= np) return;
  double x[3] = {pos[(p * nsteps + step) * 3], pos[(p * nsteps + step) * 3 + 1], pos[(p * nsteps + step) * 3 + 2]};
  int l = lev[p], b = grd[p];
  if (flags[step] && l >= 0) {
    s_where(S, x, l, b);
    lev[p] = l; grd[p] = b;
  }
  if (l >= 0) {
    const DBox B = S.L[l].boxes[b];
    if (!s_rk4(S, l, b, B, x, dt, (p & 1) ? -1 : +1)) atomicMin(bad, (int)(p + 1));  // the reference aborts ("bad RK")
  }
  for (int d = 0; d < 3; ++d) pos[(p * nsteps + step + 1) * 3 + d] = x[d];
}

extern "C" int pa_stream_trace(pa_ctx* ctx, int nlev, pa_mf* const* vfield, int vcomp, int64_t nseed, const double* seeds, int nsteps, double dt,
                               double* dev_pos, int32_t* nredist) {
  return pa_stream_trace_ranks(ctx, nlev, vfield, vcomp, nseed, seeds, nsteps, dt, dev_pos, nredist, 0);
}

// share_flags: the lines are dealt to the ranks of the context's transport (every rank holds the WHOLE vector field and
// traces its own seeds); "some line has left its grid" is then a property of all ranks' lines, as it is of all MPI ranks'
// particles in the reference (StreamPC.cpp:88-141 Redistribute), so the step's flag is max-reduced over the ranks before
// the step uses it -- every rank makes nsteps - 1 reductions, also one without seeds.  Results equal the one-rank run.
extern "C" int pa_stream_trace_ranks(pa_ctx* ctx, int nlev, pa_mf* const* vfield, int vcomp, int64_t nseed, const double* seeds, int nsteps, double dt,
                                     double* dev_pos, int32_t* nredist, int share_flags) {
  PaBind bind_(ctx);
  if (!ctx || !vfield || nlev <= 0 || nlev > PA_STREAM_MAXLEV || (nseed > 0 && (!seeds || !dev_pos))) return pa_fail(ctx, "pa_stream_trace: bad argument");
  if (nsteps < 1) return pa_fail(ctx, "pa_stream_trace: Nsteps must be at least 1");
  if (nredist) *nredist = 0;
  const bool share = share_flags && ctx->comm.nranks > 1;
  // Every rank checks the SAME things about the field before the first collective (a rank without seeds too), so that an
  // argument error makes all ranks return together; failures that only one rank can have (allocation, copies, a line leaving
  // its ghost cells) travel with the per-step reduction as a second, max-reduced value: every rank sees it at the same step
  // and all of them stop there -- no rank is left waiting in a collective (advisor finding, round 3).
  StreamLevels S;
  S.nlev = nlev; S.vcomp = vcomp; S.ng = vfield[0] ? vfield[0]->ng : 0;
  for (int l = 0; l < nlev; ++l) {
    const pa_mf* m = vfield[l];
    if (!m) return pa_fail(ctx, "pa_stream_trace: null multifab");
    if (m->ng != S.ng || m->ng < 1) return pa_fail(ctx, "pa_stream_trace: the vector field needs the same nGrow >= 1 on every level");
    if (vcomp < 0 || vcomp + 3 > m->ncomp) return pa_fail(ctx, "pa_stream_trace: component range");
    if (m->lev->nremote > 0) return pa_fail(ctx, "pa_stream_trace: levels sharded across ranks are not supported");
    S.L[l] = m->lev->view;
    S.V[l] = m->view;
    for (int d = 0; d < 3; ++d) S.dx[l][d] = (m->lev->prob_hi[d] - m->lev->prob_lo[d]) / (double)(m->lev->domhi[d] - m->lev->domlo[d] + 1);
  }
  for (int d = 0; d < 3; ++d) { S.plo[d] = vfield[0]->lev->prob_lo[d]; S.phi[d] = vfield[0]->lev->prob_hi[d]; }
  // one reduction of (flag, error) per step; returns false when the transport itself failed
  auto reduce2 = [&](int& flag, bool& err) {
    double fe[2] = {flag ? 1.0 : 0.0, err ? 1.0 : 0.0};
    if (pa_allreduce(ctx, fe, 2, 1) != 0) return false;
    flag = fe[0] != 0.0;
    err = fe[1] != 0.0;
    return true;
  };
  const long long np = 2 * nseed;
  double* dseeds = nullptr;
  int *dlev = nullptr, *dflags = nullptr;
  bool lerr = false;       // this rank cannot go on
  std::string lmsg;
  if (nseed > 0) {
    if (hipMalloc(&dseeds, sizeof(double) * 3 * (size_t)nseed) != hipSuccess || hipMalloc(&dlev, sizeof(int) * 2 * (size_t)np) != hipSuccess ||
        hipMalloc(&dflags, sizeof(int) * ((size_t)nsteps + 1)) != hipSuccess) {
      lerr = true; lmsg = "pa_stream_trace: device allocation failed";
    }
  }
  int* dgrd = dlev ? dlev + np : nullptr;
  int* dbad = dflags ? dflags + nsteps : nullptr;
  const int big = 0x7fffffff;
  int rc = 0, nr_shared = 0;
  const unsigned g = (unsigned)((np + 255) / 256);
  if (nseed > 0 && !lerr) {
    if (hipMemcpyAsync(dseeds, seeds, sizeof(double) * 3 * (size_t)nseed, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemsetAsync(dflags, 0, sizeof(int) * (size_t)nsteps, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(dbad, &big, sizeof(int), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { lerr = true; lmsg = "pa_stream_trace: copy failed"; }
    else hipLaunchKernelGGL(k_stream_init, dim3(g), dim3(256), 0, ctx->stream, S, np, dseeds, nsteps, dev_pos, dlev, dgrd);
  }
  bool gerr = false;  // some rank cannot go on (after a reduction: the same on every rank)
  for (int step = 0; step + 1 < nsteps; ++step) {
    const bool work = nseed > 0 && !lerr;
    if (work) hipLaunchKernelGGL(k_stream_check, dim3(g), dim3(256), 0, ctx->stream, S, np, nsteps, step, dev_pos, dlev, dgrd, dflags);
    if (share) {  // the flag of ALL ranks' lines (every rank takes part in every step's reduction, whatever happened to it)
      int hf1 = 0;
      if (work && (hipMemcpyAsync(&hf1, dflags + step, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        lerr = true; lmsg = "pa_stream_trace: reading the redistribution flag failed";
      }
      gerr = lerr;
      if (!reduce2(hf1, gerr)) { rc = pa_fail(ctx, "pa_stream_trace: sharing the redistribution flag between the ranks failed"); gerr = true; break; }
      if (gerr) break;  // every rank leaves at this step
      nr_shared += hf1;
      if (work && (hipMemcpyAsync(dflags + step, &hf1, sizeof(int), hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        lerr = true; lmsg = "pa_stream_trace: writing the redistribution flag failed";  // reported with the next step's reduction
      }
    } else if (lerr) {
      break;
    }
    if (nseed > 0 && !lerr) hipLaunchKernelGGL(k_stream_step, dim3(g), dim3(256), 0, ctx->stream, S, np, nsteps, step, dt, dev_pos, dlev, dgrd, dflags, dbad);
  }
  if (rc == 0 && (lerr || gerr)) rc = pa_fail(ctx, lerr ? lmsg : std::string("pa_stream_trace: another rank failed"));
  if (rc == 0 && nseed > 0) {
    do {
      if (hipGetLastError() != hipSuccess) { rc = pa_fail(ctx, "pa_stream_trace: launch failed"); break; }
      std::vector<int> hf((size_t)nsteps + 1);
      if (hipMemcpyAsync(hf.data(), dflags, sizeof(int) * hf.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = pa_fail(ctx, "pa_stream_trace: synchronisation failed"); break; }
      int nr = 0;
      for (int s2 = 0; s2 < nsteps; ++s2) nr += hf[(size_t)s2] != 0;
      if (nredist) *nredist = nr;
      if (hf[(size_t)nsteps] != big) rc = pa_fail(ctx, "pa_stream_trace: bad RK (line " + std::to_string(hf[(size_t)nsteps]) + " left the ghost cells of its grid; increase nGrow or lower hRK)");
    } while (0);
  } else if (rc == 0 && nredist) {
    *nredist = nr_shared;  // a rank without seeds: the shared flags it saw
  }
  if (dseeds) (void)hipFree(dseeds);
  if (dlev) (void)hipFree(dlev);
  if (dflags) (void)hipFree(dflags);
  // One last reduction when the ranks share flags: an error found only after the step loop (a "bad RK" line, the write-back of the
  // last step's flag, the final read-back) must come back from EVERY rank, or the caller's next collective pairs a rank that
  // returned an error with ranks that went on (advisor finding, round 4).  Skipped when the transport itself failed / a shared
  // error already ended the loop on every rank together (gerr).
  if (share && !gerr) {
    double e = rc ? 1.0 : 0.0;
    if (pa_allreduce(ctx, &e, 1, 1) != 0) return rc ? rc : pa_fail(ctx, "pa_stream_trace: sharing the final status between the ranks failed");
    if (e != 0.0 && rc == 0) rc = pa_fail(ctx, "pa_stream_trace: another rank failed (bad RK or a device error after its last step)");
  }
  return rc;
}
