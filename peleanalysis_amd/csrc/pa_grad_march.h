// pa_grad_march.h -- gradient (grad.cpp:211-236), k-marching kernel.
//
// The sweep of pa_fused_march3.h reduced to what the gradient needs (read that header first; the step schedule,
// the roles and the addressing are the ones measured there):
//   * workgroup = 64 columns x TY rows marching a z-segment; one wavefront per row (TY output rows, the two
//     y-halo rows, one edge wavefront for the two x-halo columns of every row);
//   * the x/y neighbours of a plane come from a 3-slot LDS ring (every value of phi is requested from memory
//     once per tile, not five times through L1), the z-column stays in registers and the z-face flux is
//     carried from plane to plane;
//   * three planes in flight per wavefront, requested right AFTER the plane's barrier; the four results of a
//     plane leave in one burst at the top of the next step;
//   * wave-uniform base address (SGPR pair) + one loop-invariant lane offset.
// Arithmetic is cdiff() of pa_internal.h term by term, so results are bit-identical to k_grad and the oracle.
#pragma once
#include "pa_fused_march3.h"

template <int TY>
struct GradLds {
  double p[3][TY + 2][PA_MLW];  // x index 0 = left edge column, 1..64 = lanes, llast+2 = right edge column
};

struct GradMarchArgs {
  int comp, ocomp, kseg, nboxes, tiles_max;
  const int* wgtab = nullptr;  // boxes of different sizes: workgroup i = tile wgtab[2 i + 1] of box wgtab[2 i] (< 0: none); see MarchArgs::wgtab
};

template <typename BP, int TY>
__device__ __forceinline__ void grad_march_body(const BP& bp, const GradMarchArgs& A, const unsigned bid_in) {
  FabView P, O;
  DBox V;
  double dxinv[3];
  constexpr int ROWS = TY + 2;
  static_assert(2 * ROWS <= 64, "the edge wavefront serves two columns of every row");
  // XCD-aware order (MarchArgs order 2): block 8*T*g + 8*t + q works on tile t of box 8*g + q
  unsigned bid = bid_in;
  int box;
  if (A.wgtab) {
    box = A.wgtab[2 * bid];
    if (box < 0) return;
    bid = (unsigned)A.wgtab[2 * bid + 1];
  } else {
    const unsigned per8 = 8u * (unsigned)A.tiles_max, grp = bid / per8, rem = bid % per8;
    box = (int)(grp * 8u + (rem & 7u));
    bid = rem >> 3;
    if (box >= A.nboxes) return;
  }
  if (!bp.get(box, P, O, V, dxinv)) return;
  const int comp = A.comp, kseg = A.kseg;
  const int nx = V.hi[0] - V.lo[0] + 1, ny = V.hi[1] - V.lo[1] + 1, nz = V.hi[2] - V.lo[2] + 1;
  const int tx = (nx + 63) / 64, ty = (ny + TY - 1) / TY, tz = (nz + kseg - 1) / kseg;
  if (bid >= (unsigned)tx * ty * tz) return;  // uniform for the whole workgroup
  const int bx = bid % tx, by = (bid / tx) % ty, bz = bid / (tx * ty);
  const int i0 = V.lo[0] + bx * 64, j0 = V.lo[1] + by * TY;
  const int k0 = V.lo[2] + bz * kseg, k1 = min(k0 + kseg - 1, V.hi[2]);
  const int iR = min(i0 + 64, V.hi[0] + 1);  // column right of the tile's last valid column
  const int llast = iR - 1 - i0;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rtop = min(ROWS - 1, V.hi[1] + 1 - j0 + 1);  // row slot of the last live row
  const int kf = k1 + 1;                                  // last plane of phi the segment reads

  __shared__ GradLds<TY> S;
  const long long pps = (long long)P.nx * P.ny * 8;  // plane stride of phi, bytes
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  // step(slot, p): plane p is in ring slot `slot` and in the registers; plane p+1 arrives
#define PA_GRUN3(step)                                                                                                 \
  {                                                                                                                    \
    int p = k0;                                                                                                        \
    _Pragma("unroll 1") for (; p + 2 <= k1; p += 3) {                                                                  \
      step(I0{}, p);                                                                                                   \
      step(I1{}, p + 1);                                                                                               \
      step(I2{}, p + 2);                                                                                               \
    }                                                                                                                  \
    if (p <= k1) {                                                                                                     \
      step(I0{}, p);                                                                                                   \
      if (p + 1 <= k1) step(I1{}, p + 1);                                                                              \
    }                                                                                                                  \
  }

  if (w < ROWS && w > rtop) {  // dead row of a partial tile
    for (int it = 0; it <= k1 - k0 + 1; ++it) __syncthreads();
    return;
  }

  if (w < ROWS) {
    // ------------------------------------------------------------------------- row waves
    const int rr = w;
    const int j = j0 + rr - 1;
    const int le = min(lane, llast);  // lanes past the box edge mirror the last valid lane
    const int xs = le + 1;
    unsigned lo8 = (unsigned)le * 8u;
    const char* gp = (const char*)(P.p + P.idx(i0, j, k0 - 1, comp));  // wave-uniform
    const double pm = PA_LDG(gp, lo8);
    double pc = PA_LDG(gp + pps, lo8);
    double f[3];
    f[0] = PA_LDG(gp + 2 * pps, lo8);
    gp += 2 * pps;
    gp += (k0 + 2 <= kf) ? pps : 0;
    f[1] = PA_LDG(gp, lo8);
    gp += (k0 + 3 <= kf) ? pps : 0;
    f[2] = PA_LDG(gp, lo8);  // gp -> plane min(k0+3, k1+1), the youngest plane requested
    asm volatile("" ::"v"(f[0]), "v"(f[1]), "v"(f[2]));  // enter the loop with nothing in flight (pa_fused_march3.h)
    S.p[0][rr][xs] = pc;
    __syncthreads();
    if (rr == 0 || rr == rtop) {  // y-halo rows: supply neighbours only
      auto step = [&](auto spc, int p) __attribute__((always_inline)) {
        constexpr int SP = decltype(spc)::value, SP1 = (SP + 1) % 3;
        double x;
        PA_TAKE(x, f[SP]);
        __builtin_amdgcn_sched_barrier(0);
        gp += (p + 4 <= kf) ? pps : 0;
        S.p[SP1][rr][xs] = x;
        __syncthreads();
        PA_OPAQUE(lo8);
        f[SP] = PA_LDG(gp, lo8);
      };
      PA_GRUN3(step)
      return;
    }
    double fz = zflux(dxinv[2], pm, pc);                         // low z-face flux at plane k0
    char* ob = (char*)(O.p + O.idx(i0, j, k0, A.ocomp));         // wave-uniform
    const long long ops = (long long)O.nx * O.ny * 8, osc = O.sc * 8;
    // the 4 results of a plane are stored at the top of the next step; the first step has nothing valid yet
    // and writes to plane k0, which the same thread overwrites in program order
    double o0 = 0, o1 = 0, o2 = 0, o3 = 0;
    auto step = [&](auto spc, int p) __attribute__((always_inline)) {
      constexpr int SP = decltype(spc)::value, SP1 = (SP + 1) % 3;
      double x;  // phi(p+1), requested three steps ago
      PA_TAKE(x, f[SP]);
      PA_OPAQUE(lo8);
      __builtin_amdgcn_sched_barrier(0);
      gp += (p + 4 <= kf) ? pps : 0;
      PA_STG(ob, lo8, o0); PA_STG(ob + osc, lo8, o1); PA_STG(ob + 2 * osc, lo8, o2); PA_STG(ob + 3 * osc, lo8, o3);
      __builtin_amdgcn_sched_barrier(0);  // the burst stays ahead of the plane's arithmetic
      const double pl = S.p[SP][rr][xs - 1], pr = S.p[SP][rr][xs + 1];
      const double ps = S.p[SP][rr - 1][xs], pn = S.p[SP][rr + 1][xs];
      S.p[SP1][rr][xs] = x;
      const double gx = cdiff(dxinv[0], pl, pc, pr);
      const double gy = cdiff(dxinv[1], ps, pc, pn);
      const double fzh = zflux(dxinv[2], pc, x);
      const double gz = favg(fz, fzh);
      const double gm = sqrt(gx * gx + gy * gy + gz * gz);
      __syncthreads();
      PA_OPAQUE(lo8);
      f[SP] = PA_LDG(gp, lo8);  // request for plane p+4, after the barrier
      ob += (p >= k0 + 1) ? ops : 0;
      o0 = gx; o1 = gy; o2 = gz; o3 = gm;
      fz = fzh;
      pc = x;
    };
    PA_GRUN3(step)
    PA_STG(ob, lo8, o0);  // results of the last plane (k1)
    PA_STG(ob + osc, lo8, o1);
    PA_STG(ob + 2 * osc, lo8, o2);
    PA_STG(ob + 3 * osc, lo8, o3);
    return;
  }

  // ----------------------------------------------------------------------------- edge wave
  {
    const int l2 = lane % (2 * ROWS);  // idle lanes mirror the active ones
    const int rr = min(l2 >> 1, rtop);
    const int side = l2 & 1;
    const int j = j0 + rr - 1;
    const int i = side ? iR : i0 - 1;
    const int xs = side ? llast + 2 : 0;
    const char* gp = (const char*)(P.p + P.idx(P.lo[0], P.lo[1], k0, comp));  // uniform base + in-plane lane offset (< 4 GiB)
    unsigned og = (unsigned)((j - P.lo[1]) * P.nx + (i - P.lo[0])) * 8u;
    const double pc = PA_LDG(gp, og);
    double f[3];
    f[0] = PA_LDG(gp + pps, og);
    gp += pps;
    gp += (k0 + 2 <= kf) ? pps : 0;
    f[1] = PA_LDG(gp, og);
    gp += (k0 + 3 <= kf) ? pps : 0;
    f[2] = PA_LDG(gp, og);
    S.p[0][rr][xs] = pc;
    __syncthreads();
    auto step = [&](auto spc, int p) __attribute__((always_inline)) {
      constexpr int SP = decltype(spc)::value, SP1 = (SP + 1) % 3;
      double x;
      PA_TAKE(x, f[SP]);
      __builtin_amdgcn_sched_barrier(0);
      gp += (p + 4 <= kf) ? pps : 0;
      S.p[SP1][rr][xs] = x;
      __syncthreads();
      PA_OPAQUE(og);
      f[SP] = PA_LDG(gp, og);
    };
    PA_GRUN3(step)
  }
#undef PA_GRUN3
}
template <typename BP, int TY>
__global__ __launch_bounds__(64 * (TY + 3)) void k_grad_march(BP bp, GradMarchArgs A) {
  grad_march_body<BP, TY>(bp, A, blockIdx.x);
}
// the gradient sweeps of several levels in ONE launch (round 5; as k_gradcurv_march3_levels: the levels of grad.cpp:215-236 do not
// depend on each other once their ghost cells are filled, and a level of a few boxes is 1-2 rounds of workgroups whose launch ends in
// an idle tail): level l owns workgroups wg0[l] .. wg0[l+1]-1, each range a multiple of 8 (XCD numbering)
struct GradBatch {
  int n;
  unsigned wg0[PA_MAXB + 1];
  LevelBP2 bp[PA_MAXB];
  GradMarchArgs A[PA_MAXB];
};
template <int TY>
__global__ __launch_bounds__(64 * (TY + 3)) void k_grad_march_levels(GradBatch S) {
  int l = 0;
  while (l + 1 < S.n && blockIdx.x >= S.wg0[l + 1]) ++l;
  grad_march_body<LevelBP2, TY>(S.bp[l], S.A[l], blockIdx.x - S.wg0[l]);
}

// Boxes at most 32 cells wide (AMReX's default max_grid_size in 3-D): the same sweep with TWO rows of 32 columns per
// wavefront, so that a wave's loads and stores stay 512 contiguous bytes (two adjacent 256-byte rows of a 32-wide output
// FAB) -- as pa_fused_march3n.h does for the fused sweep.  Waves 0..NRW-1: output rows 2w, 2w+1 (lanes 0-31 / 32-63); wave
// NRW: the two halo rows; wave NRW+1: the edge columns.  Rows past a partial tile repeat the last valid row (same LDS slot,
// same addresses, same values).  Same arithmetic as k_grad_march: bit-identical.
template <int NRW>
struct GradLdsN {
  double p[3][2 * NRW + 2][34];  // x index 0 = left edge column, 1..32 = columns, llast+2 = right edge column
};

template <typename BP, int NRW>
__global__ __launch_bounds__(64 * (NRW + 2)) void k_grad_marchn(BP bp, GradMarchArgs A) {
  FabView P, O;
  DBox V;
  double dxinv[3];
  constexpr int MTY2 = 2 * NRW, ROWS = MTY2 + 2;
  static_assert(2 * ROWS <= 64, "the edge wavefront serves two columns of every row");
  unsigned bid = blockIdx.x;
  int box;
  if (A.wgtab) {
    box = A.wgtab[2 * bid];
    if (box < 0) return;
    bid = (unsigned)A.wgtab[2 * bid + 1];
  } else {
    const unsigned per8 = 8u * (unsigned)A.tiles_max, grp = bid / per8, rem = bid % per8;
    box = (int)(grp * 8u + (rem & 7u));
    bid = rem >> 3;
    if (box >= A.nboxes) return;
  }
  if (!bp.get(box, P, O, V, dxinv)) return;
  const int comp = A.comp, kseg = A.kseg;
  const int nx = V.hi[0] - V.lo[0] + 1, ny = V.hi[1] - V.lo[1] + 1, nz = V.hi[2] - V.lo[2] + 1;
  const int tx = (nx + 31) / 32, ty = (ny + MTY2 - 1) / MTY2, tz = (nz + kseg - 1) / kseg;
  if (bid >= (unsigned)tx * ty * tz) return;
  const int bx = bid % tx, by = (bid / tx) % ty, bz = bid / (tx * ty);
  const int i0 = V.lo[0] + bx * 32, j0 = V.lo[1] + by * MTY2;
  const int k0 = V.lo[2] + bz * kseg, k1 = min(k0 + kseg - 1, V.hi[2]);
  const int iR = min(i0 + 32, V.hi[0] + 1);
  const int llast = iR - 1 - i0;
  const int nrows = min(MTY2, V.hi[1] - j0 + 1);
  const int rtop = nrows + 1;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rsub = lane >> 5, col = lane & 31;
  const int kf = k1 + 1;

  __shared__ GradLdsN<NRW> S;
  const long long pps = (long long)P.nx * P.ny * 8;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
#define PA_GRUN3(step)                                                                                                 \
  {                                                                                                                    \
    int p = k0;                                                                                                        \
    _Pragma("unroll 1") for (; p + 2 <= k1; p += 3) {                                                                  \
      step(I0{}, p);                                                                                                   \
      step(I1{}, p + 1);                                                                                               \
      step(I2{}, p + 2);                                                                                               \
    }                                                                                                                  \
    if (p <= k1) {                                                                                                     \
      step(I0{}, p);                                                                                                   \
      if (p + 1 <= k1) step(I1{}, p + 1);                                                                              \
    }                                                                                                                  \
  }

  if (w < NRW) {
    // ------------------------------------------------------------------------- output rows
    const int rr = min(1 + 2 * w + rsub, nrows);
    const int le = min(col, llast);
    const int xs = le + 1;
    unsigned lo8 = (unsigned)((rr - 1) * P.nx + le) * 8u;
    unsigned so8 = (unsigned)((rr - 1) * O.nx + le) * 8u;
    const char* gp = (const char*)(P.p + P.idx(i0, j0, k0 - 1, comp));  // wave-uniform: the tile's first row
    const double pm = PA_LDG(gp, lo8);
    double pc = PA_LDG(gp + pps, lo8);
    double f[3];
    f[0] = PA_LDG(gp + 2 * pps, lo8);
    gp += 2 * pps;
    gp += (k0 + 2 <= kf) ? pps : 0;
    f[1] = PA_LDG(gp, lo8);
    gp += (k0 + 3 <= kf) ? pps : 0;
    f[2] = PA_LDG(gp, lo8);
    asm volatile("" ::"v"(f[0]), "v"(f[1]), "v"(f[2]));  // enter the loop with nothing in flight (pa_fused_march3.h)
    S.p[0][rr][xs] = pc;
    __syncthreads();
    double fz = zflux(dxinv[2], pm, pc);
    char* ob = (char*)(O.p + O.idx(i0, j0, k0, A.ocomp));
    const long long ops = (long long)O.nx * O.ny * 8, osc = O.sc * 8;
    double o0 = 0, o1 = 0, o2 = 0, o3 = 0;
    auto step = [&](auto spc, int p) __attribute__((always_inline)) {
      constexpr int SP = decltype(spc)::value, SP1 = (SP + 1) % 3;
      double x;
      PA_TAKE(x, f[SP]);
      PA_OPAQUE(so8);
      __builtin_amdgcn_sched_barrier(0);
      gp += (p + 4 <= kf) ? pps : 0;
      PA_STG(ob, so8, o0); PA_STG(ob + osc, so8, o1); PA_STG(ob + 2 * osc, so8, o2); PA_STG(ob + 3 * osc, so8, o3);
      __builtin_amdgcn_sched_barrier(0);
      const double pl = S.p[SP][rr][xs - 1], pr = S.p[SP][rr][xs + 1];
      const double ps = S.p[SP][rr - 1][xs], pn = S.p[SP][rr + 1][xs];
      S.p[SP1][rr][xs] = x;
      const double gx = cdiff(dxinv[0], pl, pc, pr);
      const double gy = cdiff(dxinv[1], ps, pc, pn);
      const double fzh = zflux(dxinv[2], pc, x);
      const double gz = favg(fz, fzh);
      const double gm = sqrt(gx * gx + gy * gy + gz * gz);
      __syncthreads();
      PA_OPAQUE(lo8);
      f[SP] = PA_LDG(gp, lo8);
      ob += (p >= k0 + 1) ? ops : 0;
      o0 = gx; o1 = gy; o2 = gz; o3 = gm;
      fz = fzh;
      pc = x;
    };
    PA_GRUN3(step)
    PA_STG(ob, so8, o0);
    PA_STG(ob + osc, so8, o1);
    PA_STG(ob + 2 * osc, so8, o2);
    PA_STG(ob + 3 * osc, so8, o3);
    return;
  }

  if (w == NRW) {
    // ------------------------------------------------------------------------- the two halo rows (neighbours only)
    const int rr = rsub ? rtop : 0;
    const int j = rsub ? j0 + nrows : j0 - 1;
    const int le = min(col, llast);
    const int xs = le + 1;
    const char* gp = (const char*)(P.p + P.idx(i0, P.lo[1], k0, comp));  // wave-uniform: first row of the FAB, plane k0
    unsigned lo8 = (unsigned)((j - P.lo[1]) * P.nx + le) * 8u;
    const double pc = PA_LDG(gp, lo8);
    double f[3];
    f[0] = PA_LDG(gp + pps, lo8);
    gp += pps;
    gp += (k0 + 2 <= kf) ? pps : 0;
    f[1] = PA_LDG(gp, lo8);
    gp += (k0 + 3 <= kf) ? pps : 0;
    f[2] = PA_LDG(gp, lo8);
    S.p[0][rr][xs] = pc;
    __syncthreads();
    auto step = [&](auto spc, int p) __attribute__((always_inline)) {
      constexpr int SP = decltype(spc)::value, SP1 = (SP + 1) % 3;
      double x;
      PA_TAKE(x, f[SP]);
      __builtin_amdgcn_sched_barrier(0);
      gp += (p + 4 <= kf) ? pps : 0;
      S.p[SP1][rr][xs] = x;
      __syncthreads();
      PA_OPAQUE(lo8);
      f[SP] = PA_LDG(gp, lo8);
    };
    PA_GRUN3(step)
    return;
  }

  // ----------------------------------------------------------------------------- edge wave
  {
    const int l2 = lane % (2 * ROWS);
    const int rr = min(l2 >> 1, rtop);
    const int side = l2 & 1;
    const int j = j0 + rr - 1;
    const int i = side ? iR : i0 - 1;
    const int xs = side ? llast + 2 : 0;
    const char* gp = (const char*)(P.p + P.idx(P.lo[0], P.lo[1], k0, comp));
    unsigned og = (unsigned)((j - P.lo[1]) * P.nx + (i - P.lo[0])) * 8u;
    const double pc = PA_LDG(gp, og);
    double f[3];
    f[0] = PA_LDG(gp + pps, og);
    gp += pps;
    gp += (k0 + 2 <= kf) ? pps : 0;
    f[1] = PA_LDG(gp, og);
    gp += (k0 + 3 <= kf) ? pps : 0;
    f[2] = PA_LDG(gp, og);
    S.p[0][rr][xs] = pc;
    __syncthreads();
    auto step = [&](auto spc, int p) __attribute__((always_inline)) {
      constexpr int SP = decltype(spc)::value, SP1 = (SP + 1) % 3;
      double x;
      PA_TAKE(x, f[SP]);
      __builtin_amdgcn_sched_barrier(0);
      gp += (p + 4 <= kf) ? pps : 0;
      S.p[SP1][rr][xs] = x;
      __syncthreads();
      PA_OPAQUE(og);
      f[SP] = PA_LDG(gp, og);
    };
    PA_GRUN3(step)
  }
#undef PA_GRUN3
}

