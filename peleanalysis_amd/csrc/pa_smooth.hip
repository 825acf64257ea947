// pa_smooth.hip -- do_smooth of curvature.cpp:328-406 on gfx950: one implicit diffusion step of the
// progress variable, (I - dt Lap) x = c, as a COMPOSITE solve over the AMR hierarchy (MLABecLaplacian
// alpha = 1, A = 1, beta = smoothing_time, B = 1; periodic / homogeneous Neumann domain boundaries;
// fine ghosts from MLCellLinOp::applyBC; coarse flux at a coarse-fine face = average of the fine fluxes;
// covered coarse cells = child averages; tol_rel = tol_abs = 1e-12).
// The reference solves with AMReX's MLMG (absent); any solver that reaches the
// tolerance gives the same field to ~tol*cond, so this is BiCGStab on the composite operator -- the same
// algorithm as oracle/pa_oracle_smooth.c, compared with it to a tolerance.  Every building block is a
// bandwidth-bound stencil / vector kernel over the level's boxes; the 7-point apply reads x once and
// writes y once (16 B/cell), reductions are two-stage with a fixed summation order per launch shape.
#include "pa_internal.h"
#include "pa_fabview.h"
#include "pa_dpp.h"
#include "pa_dist.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <vector>

int pa_ensure_red(pa_ctx* ctx, size_t n);
int pa_fill_boundary_impl(pa_ctx* ctx, pa_mf* M, int comp, int ncomp, int ng, int no_exchange);
int pa_apply_bc_impl(pa_ctx* ctx, pa_mf* F, int comp, const pa_mf* C, int ccomp, const int32_t bc[3], int ratio, int only_dir, int edges, const double* crse_xform);

struct BoxIt {  // thread -> cell of box blockIdx.y (grid-stride over the box's valid cells)
  DBox B;
  int nx, ny, nz;
  long long n;
  __device__ __forceinline__ BoxIt(const DLevelView& L, int b) {
    B = L.boxes[b];
    nx = B.hi[0] - B.lo[0] + 1; ny = B.hi[1] - B.lo[1] + 1; nz = B.hi[2] - B.lo[2] + 1;
    n = (long long)nx * ny * nz;
  }
  __device__ __forceinline__ void cell(long long t, int& i, int& j, int& k) const {
    const unsigned u = (unsigned)t, r = u / (unsigned)nx;
    i = B.lo[0] + (int)(u - r * (unsigned)nx);
    j = B.lo[1] + (int)(r % (unsigned)ny);
    k = B.lo[2] + (int)(r / (unsigned)ny);
  }
};
// refinement ratio of direction d: a 2-D hierarchy is stored as one plane of cells per level (k = 0), not refined in z
__device__ __forceinline__ int rdir(const DLevelView& LF, int d, int ratio) { return (d == 2 && LF.domlo[2] == LF.domhi[2]) ? 1 : ratio; }
#define PA_BOX_LOOP(L)                                                                                                \
  const int b = blockIdx.y;                                                                                            \
  const BoxIt it(L, b);                                                                                                \
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < it.n; t += (long long)gridDim.x * blockDim.x)

// mask = 1 on valid cells not covered by the next finer level, else 0
__global__ __launch_bounds__(256) void k_smooth_mask(DLevelView L, DMFView M, DLevelView LF, int has_fine, int ratio) {
  PA_BOX_LOOP(L) {
    int i, j, k;
    it.cell(t, i, j, k);
    double m = 1.0;
    if (has_fine) {
      int p[3] = {i * ratio, j * ratio, k * rdir(LF, 2, ratio)};
      if (owner_of(LF, p) != -1) m = 0.0;
    }
    M.data[M.off[b] + fab_index(it.B, M.ng, M.ncomp, 0, i, j, k)] = m;
  }
}

// average_down: thread per coarse cell under fine box b (sum of the children in k,j,i order, * 1/r^3)
__global__ __launch_bounds__(256) void k_smooth_avgdown(DLevelView LF, DMFView F, DLevelView LC, DMFView Cm, int ratio) {
  const int b = blockIdx.y;
  const DBox B = LF.boxes[b];
  const int rz = rdir(LF, 2, ratio);
  const int cx = (B.hi[0] - B.lo[0] + 1) / ratio, cy = (B.hi[1] - B.lo[1] + 1) / ratio, cz = (B.hi[2] - B.lo[2] + 1) / rz;
  const long long n = (long long)cx * cy * cz;
  const double fac = 1.0 / (double)(ratio * ratio * rz);
  const double* f = F.data + F.off[b];
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const unsigned u = (unsigned)t, r = u / (unsigned)cx;
    const int ic = coarsen_idx(B.lo[0], ratio) + (int)(u - r * cx), jc = coarsen_idx(B.lo[1], ratio) + (int)(r % cy), kc = coarsen_idx(B.lo[2], rz) + (int)(r / cy);
    double c = 0.0;
    for (int kk = 0; kk < rz; ++kk)
      for (int jj = 0; jj < ratio; ++jj)
        for (int ii = 0; ii < ratio; ++ii) c += f[fab_index(B, F.ng, F.ncomp, 0, ic * ratio + ii, jc * ratio + jj, kc * rz + kk)];
    c *= fac;
    const int p[3] = {ic, jc, kc};
    const int cb = owner_of(LC, p);
    if (cb >= 0) Cm.data[Cm.off[cb] + fab_index(LC.boxes[cb], Cm.ng, Cm.ncomp, 0, ic, jc, kc)] = c;
  }
}

// y = x - dt * div(grad x), flux form (same association as the oracle); x has resolved ring-1 face ghosts
__global__ __launch_bounds__(256) void k_smooth_apply(DLevelView L, DMFView X, DMFView Y, double dt) {
  PA_BOX_LOOP(L) {
    int i, j, k;
    it.cell(t, i, j, k);
    const double* x = X.data + X.off[b];
    const long long nxg = it.nx + 2 * X.ng, nyg = it.ny + 2 * X.ng;
    const long long q = fab_index(it.B, X.ng, X.ncomp, 0, i, j, k);
    const double c = x[q];
    const double d0 = L.dxinv[0], d1 = L.dxinv[1], d2 = L.dxinv[2];
    double div = 0.0;
    div += d0 * (d0 * (x[q + 1] - c) - d0 * (c - x[q - 1]));
    div += d1 * (d1 * (x[q + nxg] - c) - d1 * (c - x[q - nxg]));
    div += d2 * (d2 * (x[q + nxg * nyg] - c) - d2 * (c - x[q - nxg * nyg]));
    Y.data[Y.off[b] + fab_index(it.B, Y.ng, Y.ncomp, 0, i, j, k)] = c - dt * div;
  }
}

// multigrid smoother / residual in one pass over a level (x with resolved ring-1 face ghosts):
//   MODE 0: y = x + om (r - A x)   (a damped-Jacobi step into a SECOND buffer)      MODE 1: y = r - A x
template <int MODE>
__global__ __launch_bounds__(256) void k_smooth_jacobi(DLevelView L, DMFView X, DMFView R, DMFView Y, double dt, double om) {
  PA_BOX_LOOP(L) {
    int i, j, k;
    it.cell(t, i, j, k);
    const double* x = X.data + X.off[b];
    const long long nxg = it.nx + 2 * X.ng, nyg = it.ny + 2 * X.ng;
    const long long q = fab_index(it.B, X.ng, X.ncomp, 0, i, j, k);
    const double c = x[q];
    const double d0 = L.dxinv[0], d1 = L.dxinv[1], d2 = L.dxinv[2];
    double div = 0.0;
    div += d0 * (d0 * (x[q + 1] - c) - d0 * (c - x[q - 1]));
    div += d1 * (d1 * (x[q + nxg] - c) - d1 * (c - x[q - nxg]));
    div += d2 * (d2 * (x[q + nxg * nyg] - c) - d2 * (c - x[q - nxg * nyg]));
    const double res = R.data[R.off[b] + fab_index(it.B, R.ng, R.ncomp, 0, i, j, k)] - (c - dt * div);
    Y.data[Y.off[b] + fab_index(it.B, Y.ng, Y.ncomp, 0, i, j, k)] = MODE == 0 ? c + om * res : res;
  }
}

// The same three passes as z-MARCHING kernels (round 5, third session): a wavefront holds a row of 64 cells (boxes at most 32
// wide: two rows of 32), TY wavefronts a tile of TY rows, and the tile marches kz planes -- the z-neighbours ride in registers (one
// new plane per step), the x-neighbours come from the neighbouring LANES (pa_dpp.h; the first / last lane of a row loads a
// neighbouring tile's or a ghost cell), the y-neighbours of a plane are requested one step early together with its centre value
// (the rows next door ask for the same lines as THEIR centre values in that step).  3 load instructions per cell instead of 7, no
// integer division per cell; same operations in the same order as k_smooth_apply / k_smooth_jacobi, so the same bits.
//   MODE 0: y = A x    1: y = x + om (r - A x)    2: y = r - A x;   MASKED: y = 0 where M == 0 (the cells under the finer level)
// Tiles of one (x, z) column are consecutive workgroups of ONE XCD (workgroup i runs on XCD i mod 8); gridDim.x is a multiple of 8.
struct SmArgs {
  DLevelView L;
  DMFView X, R, Y, M;
  double dt, om;
  int kz;
};
template <int MODE, bool MASKED, int TY>
__global__ __launch_bounds__(64 * TY) void k_smooth_march(SmArgs A) {
  const int b = blockIdx.y;
  const DBox V = A.L.boxes[b];
  const int nx = V.hi[0] - V.lo[0] + 1, ny = V.hi[1] - V.lo[1] + 1, nz = V.hi[2] - V.lo[2] + 1;
  const bool narrow = nx <= 32;
  const int TX = narrow ? 32 : 64, rows = narrow ? 2 * TY : TY;
  const int tx = (nx + TX - 1) / TX, tz = (nz + A.kz - 1) / A.kz;
  const unsigned ty = (unsigned)((ny + rows - 1) / rows);
  const int lw = (int)threadIdx.x & (TX - 1), jr = narrow ? (int)(threadIdx.x >> 5) : (int)(threadIdx.x >> 6);
  const FabView X = mf_view(A.X, V, b), Y = mf_view(A.Y, V, b), R = mf_view(MODE ? A.R : A.X, V, b), M = mf_view(MASKED ? A.M : A.X, V, b);
  const long long xs = (long long)X.nx * X.ny, ys = (long long)Y.nx * Y.ny, rs = (long long)R.nx * R.ny, ms = (long long)M.nx * M.ny;
  const double d0 = A.L.dxinv[0], d1 = A.L.dxinv[1], d2 = A.L.dxinv[2], dt = A.dt, om = A.om;
  for (unsigned w = blockIdx.x;; w += gridDim.x) {
    const unsigned q = w & 7u, m = w >> 3;
    const int col = (int)(q + 8u * (m / ty)), by = (int)(m % ty);
    if (col >= tx * tz) break;  // (col grows with w: gridDim.x is a multiple of 8)
    const int bx = col % tx, bz = col / tx;
    const int i = V.lo[0] + bx * TX + lw, j = V.lo[1] + by * rows + jr, k0 = V.lo[2] + bz * A.kz, k1 = min(k0 + A.kz - 1, V.hi[2]);
    if (i > V.hi[0] || j > V.hi[1]) continue;
    const bool ledge = lw == 0, redge = lw == TX - 1 || i == V.hi[0];  // this lane's x-neighbour is not held by the lane next to it
    const double* xp = X.p + X.idx(i, j, k0, 0);
    double* yp = Y.p + Y.idx(i, j, k0, 0);
    const double* rp = R.p + R.idx(i, j, k0, 0);
    const double* mp = M.p + M.idx(i, j, k0, 0);
    double zm = xp[-xs], c = xp[0], yl = xp[-X.nx], yr = xp[X.nx];
    for (int k = k0; k <= k1; ++k) {
      const double nxt = xp[xs], nyl = xp[xs - X.nx], nyr = xp[xs + X.nx];  // plane k + 1 (k1 + 1: a ghost plane at most)
      double xl = lane_from_left(c), xr = lane_from_right(c);
      if (ledge) xl = xp[-1];
      if (redge) xr = xp[1];
      double div = 0.0;
      div += d0 * (d0 * (xr - c) - d0 * (c - xl));
      div += d1 * (d1 * (yr - c) - d1 * (c - yl));
      div += d2 * (d2 * (nxt - c) - d2 * (c - zm));
      double out = c - dt * div;
      if (MODE) {
        const double res = rp[0] - out;
        out = MODE == 1 ? c + om * res : res;
        rp += rs;
      }
      if (MASKED) {
        if (mp[0] == 0.0) out = 0.0;
        mp += ms;
      }
      yp[0] = out;
      zm = c; c = nxt; yl = nyl; yr = nyr;
      xp += xs;
      yp += ys;
    }
  }
}

// reflux from the fine side: thread per coarse face of a special fine face (coarse-fine cells only)
__global__ __launch_bounds__(256) void k_smooth_reflux(DLevelView LF, DMFView XF, DLevelView LC, DMFView XC, DMFView YC, double dt, int ratio) {
  const int e = LF.sfaces[blockIdx.y];
  const int b = e / 6, dir = (e % 6) >> 1, side = e & 1;
  const DBox B = LF.boxes[b];
  const int t0 = (dir == 0) ? 1 : 0, t1 = (dir == 2) ? 1 : 2;
  const int n0 = B.hi[t0] - B.lo[t0] + 1, n1 = B.hi[t1] - B.lo[t1] + 1;
  const int r0 = rdir(LF, t0, ratio), r1 = rdir(LF, t1, ratio), rn = rdir(LF, dir, ratio);
  const unsigned c0 = n0 / r0, c1 = n1 / r1;
  const long long tt = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (tt >= (long long)c0 * c1) return;
  const unsigned u = (unsigned)tt, r = u / c0;
  const int a0 = (int)(u - r * c0), b1 = (int)r;  // coarse offsets inside the face
  // class of the first child ghost cell: 1 = coarse-fine (a coarse cell is covered entirely or not at all)
  if ((LF.sfcode[LF.sfoff[blockIdx.y] + (long long)(a0 * r0) + (long long)n0 * (b1 * r1)] & 3u) != 1u) return;
  const int gq = side ? B.hi[dir] + 1 : B.lo[dir] - 1, inq = side ? B.hi[dir] : B.lo[dir];
  int oc[3], ic[3];
  oc[dir] = coarsen_idx(gq, rn); ic[dir] = coarsen_idx(inq, rn);
  oc[t0] = ic[t0] = coarsen_idx(B.lo[t0], r0) + a0;
  oc[t1] = ic[t1] = coarsen_idx(B.lo[t1], r1) + b1;
  int ow[3] = {oc[0], oc[1], oc[2]}, iw[3] = {ic[0], ic[1], ic[2]};
  if (!wrap_cell(LC, ow) || !wrap_cell(LC, iw)) return;
  const int ob = owner_of(LC, ow), ib = owner_of(LC, iw);
  if (ob < 0 || ib < 0) return;
  const double* xf = XF.data + XF.off[b];
  const double dxf = LF.dxinv[dir], dxc = LC.dxinv[dir];
  double favg = 0.0;
  for (int v = 0; v < r1; ++v)
    for (int uu = 0; uu < r0; ++uu) {
      int g[3], in[3];
      g[dir] = gq; in[dir] = inq;
      g[t0] = in[t0] = B.lo[t0] + a0 * r0 + uu;
      g[t1] = in[t1] = B.lo[t1] + b1 * r1 + v;
      const double xg = xf[fab_index(B, XF.ng, XF.ncomp, 0, g[0], g[1], g[2])], xi = xf[fab_index(B, XF.ng, XF.ncomp, 0, in[0], in[1], in[2])];
      favg += side ? dxf * (xg - xi) : dxf * (xi - xg);
    }
  favg *= 1.0 / (double)(r0 * r1);
  const double xo = XC.data[XC.off[ob] + fab_index(LC.boxes[ob], XC.ng, XC.ncomp, 0, ow[0], ow[1], ow[2])];
  const double xin = XC.data[XC.off[ib] + fab_index(LC.boxes[ib], XC.ng, XC.ncomp, 0, iw[0], iw[1], iw[2])];
  const double fc = side ? dxc * (xo - xin) : dxc * (xin - xo);
  const double corr = dt * (dxc * (favg - fc));
  atomicAdd(&YC.data[YC.off[ob] + fab_index(LC.boxes[ob], YC.ng, YC.ncomp, 0, ow[0], ow[1], ow[2])], side ? corr : -corr);
}

__global__ __launch_bounds__(256) void k_smooth_zero_covered(DLevelView L, DMFView Y, DMFView M) {
  PA_BOX_LOOP(L) {
    int i, j, k;
    it.cell(t, i, j, k);
    const long long q = fab_index(it.B, Y.ng, Y.ncomp, 0, i, j, k);
    if (M.data[M.off[b] + q] == 0.0) Y.data[Y.off[b] + q] = 0.0;
  }
}

// ---- the same operator on a hierarchy SHARDED over ranks: the fine rank restricts its own boxes into the coarsened level
// (pa_dist.h: RsPlan), an exchange moves the result to the coarse owners
__global__ __launch_bounds__(256) void k_smooth_avgdown_cf(DLevelView LF, DMFView F, DLevelView LCF, DMFView CF, int ratio) {
  const int b = blockIdx.y;
  const DBox B = LF.boxes[b];
  const int rz = rdir(LF, 2, ratio);
  const int cx = (B.hi[0] - B.lo[0] + 1) / ratio, cy = (B.hi[1] - B.lo[1] + 1) / ratio, cz = (B.hi[2] - B.lo[2] + 1) / rz;
  const long long n = (long long)cx * cy * cz;
  const double fac = 1.0 / (double)(ratio * ratio * rz);
  const double* f = F.data + F.off[b];
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const unsigned u = (unsigned)t, r = u / (unsigned)cx;
    const int ic = coarsen_idx(B.lo[0], ratio) + (int)(u - r * cx), jc = coarsen_idx(B.lo[1], ratio) + (int)(r % cy), kc = coarsen_idx(B.lo[2], rz) + (int)(r / cy);
    double c = 0.0;
    for (int kk = 0; kk < rz; ++kk)
      for (int jj = 0; jj < ratio; ++jj)
        for (int ii = 0; ii < ratio; ++ii) c += f[fab_index(B, F.ng, F.ncomp, 0, ic * ratio + ii, jc * ratio + jj, kc * rz + kk)];
    CF.data[CF.off[b] + fab_index(LCF.boxes[b], CF.ng, CF.ncomp, 0, ic, jc, kc)] = c * fac;
  }
}

// flux register, fine side: thread per coarse face of a special fine face; the average fine flux across it (0 where the face
// is not coarse-fine) goes to the ghost cell of the coarsened box behind that face.  mask_mode: 1.0 on coarse-fine faces instead.
__global__ __launch_bounds__(256) void k_smooth_fluxreg(DLevelView LF, DMFView XF, DLevelView LCF, DMFView CF, int ratio, int mask_mode) {
  const int e = LF.sfaces[blockIdx.y];
  const int b = e / 6, dir = (e % 6) >> 1, side = e & 1;
  const DBox B = LF.boxes[b];
  const int t0 = (dir == 0) ? 1 : 0, t1 = (dir == 2) ? 1 : 2;
  const int n0 = B.hi[t0] - B.lo[t0] + 1, n1 = B.hi[t1] - B.lo[t1] + 1;
  const int r0 = rdir(LF, t0, ratio), r1 = rdir(LF, t1, ratio), rn = rdir(LF, dir, ratio);
  const unsigned c0 = n0 / r0, c1 = n1 / r1;
  const long long tt = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (tt >= (long long)c0 * c1) return;
  const unsigned u = (unsigned)tt, r = u / c0;
  const int a0 = (int)(u - r * c0), b1 = (int)r;
  const bool cf = (LF.sfcode[LF.sfoff[blockIdx.y] + (long long)(a0 * r0) + (long long)n0 * (b1 * r1)] & 3u) == 1u;
  const int gq = side ? B.hi[dir] + 1 : B.lo[dir] - 1, inq = side ? B.hi[dir] : B.lo[dir];
  int oc[3];
  oc[dir] = coarsen_idx(gq, rn);
  oc[t0] = coarsen_idx(B.lo[t0], r0) + a0;
  oc[t1] = coarsen_idx(B.lo[t1], r1) + b1;
  double favg = 0.0;
  if (cf && mask_mode) favg = 1.0;
  if (cf && !mask_mode) {
    const double* xf = XF.data + XF.off[b];
    const double dxf = LF.dxinv[dir];
    for (int v = 0; v < r1; ++v)
      for (int uu = 0; uu < r0; ++uu) {
        int g[3], in[3];
        g[dir] = gq; in[dir] = inq;
        g[t0] = in[t0] = B.lo[t0] + a0 * r0 + uu;
        g[t1] = in[t1] = B.lo[t1] + b1 * r1 + v;
        const double xg = xf[fab_index(B, XF.ng, XF.ncomp, 0, g[0], g[1], g[2])], xi = xf[fab_index(B, XF.ng, XF.ncomp, 0, in[0], in[1], in[2])];
        favg += side ? dxf * (xg - xi) : dxf * (xi - xg);
      }
    favg *= 1.0 / (double)(r0 * r1);
  }
  CF.data[CF.off[b] + fab_index(LCF.boxes[b], CF.ng, CF.ncomp, 0, oc[0], oc[1], oc[2])] = favg;
}

// the six 0/1 components of the received mask register -> one bit set per coarse cell, kept in component 0
__global__ __launch_bounds__(256) void k_smooth_fluxmask(DLevelView L, DMFView FM) {
  PA_BOX_LOOP(L) {
    int i, j, k;
    it.cell(t, i, j, k);
    int m = 0;
    for (int o = 0; o < 6; ++o)
      if (FM.data[FM.off[b] + fab_index(it.B, FM.ng, FM.ncomp, o, i, j, k)] != 0.0) m |= 1 << o;
    FM.data[FM.off[b] + fab_index(it.B, FM.ng, FM.ncomp, 0, i, j, k)] = (double)m;
  }
}

// flux register, coarse side: the coarse flux the 7-point apply used across a coarse-fine face is replaced by the average fine
// flux (component 2 * dir + side of FR: the fine box lies on the low side of the cell for side = 1); fixed order, no atomics
__global__ __launch_bounds__(256) void k_smooth_reflux_apply(DLevelView L, DMFView X, DMFView Y, DMFView FR, DMFView FM, double dt) {
  PA_BOX_LOOP(L) {
    int i, j, k;
    it.cell(t, i, j, k);
    const int m = (int)FM.data[FM.off[b] + fab_index(it.B, FM.ng, FM.ncomp, 0, i, j, k)];
    if (!m) continue;
    const double* x = X.data + X.off[b];
    const long long nxg = it.nx + 2 * X.ng, nyg = it.ny + 2 * X.ng;
    const long long q = fab_index(it.B, X.ng, X.ncomp, 0, i, j, k);
    const long long st[3] = {1, nxg, nxg * nyg};
    const double xo = x[q];
    double y = Y.data[Y.off[b] + fab_index(it.B, Y.ng, Y.ncomp, 0, i, j, k)];
    for (int o = 0; o < 6; ++o) {
      if (!((m >> o) & 1)) continue;
      const int dir = o >> 1, side = o & 1;
      const double favg = FR.data[FR.off[b] + fab_index(it.B, FR.ng, FR.ncomp, o, i, j, k)];
      const double xin = x[q + (side ? -st[dir] : st[dir])];
      const double dxc = L.dxinv[dir];
      const double fc = side ? dxc * (xo - xin) : dxc * (xin - xo);
      const double corr = dt * (dxc * (favg - fc));
      y += side ? corr : -corr;
    }
    Y.data[Y.off[b] + fab_index(it.B, Y.ng, Y.ncomp, 0, i, j, k)] = y;
  }
}

// z = a x + bc y + c z on valid cells (x, y may be null)
__global__ __launch_bounds__(256) void k_smooth_axpbypcz(DLevelView L, double a, DMFView X, int hx, double bc, DMFView Y, int hy, double c, DMFView Z) {
  PA_BOX_LOOP(L) {
    int i, j, k;
    it.cell(t, i, j, k);
    const long long q = fab_index(it.B, Z.ng, Z.ncomp, 0, i, j, k);
    double v = c * Z.data[Z.off[b] + q];
    if (hx) v += a * X.data[X.off[b] + q];
    if (hy) v += bc * Y.data[Y.off[b] + q];
    Z.data[Z.off[b] + q] = v;
  }
}

// copy comp sc of S (any ng) into the 1-comp vector D / back
__global__ __launch_bounds__(256) void k_smooth_copy(DLevelView L, DMFView S, int sc, DMFView D, int dc) {
  PA_BOX_LOOP(L) {
    int i, j, k;
    it.cell(t, i, j, k);
    D.data[D.off[b] + fab_index(it.B, D.ng, D.ncomp, dc, i, j, k)] = S.data[S.off[b] + fab_index(it.B, S.ng, S.ncomp, sc, i, j, k)];
  }
}

// multigrid preconditioner: e_fine += e_coarse(parent cell) (piecewise-constant prolongation, ratio 2)
__global__ __launch_bounds__(256) void k_smooth_prolong_add(DLevelView LF, DMFView EF, DLevelView LC, DMFView EC, int ratio) {
  PA_BOX_LOOP(LF) {
    int i, j, k;
    it.cell(t, i, j, k);
    const int p[3] = {coarsen_idx(i, ratio), coarsen_idx(j, ratio), coarsen_idx(k, rdir(LF, 2, ratio))};
    const int cb = owner_of(LC, p);
    if (cb < 0) continue;
    EF.data[EF.off[b] + fab_index(it.B, EF.ng, EF.ncomp, 0, i, j, k)] += EC.data[EC.off[cb] + fab_index(LC.boxes[cb], EC.ng, EC.ncomp, 0, p[0], p[1], p[2])];
  }
}

// per-block partial sums of a*b and max|a| over uncovered cells
__global__ __launch_bounds__(256) void k_smooth_dot(DLevelView L, DMFView A, DMFView Bv, DMFView M, double* part) {
  double s = 0.0, m = 0.0;
  {
    PA_BOX_LOOP(L) {
      int i, j, k;
      it.cell(t, i, j, k);
      const long long q = fab_index(it.B, A.ng, A.ncomp, 0, i, j, k);
      if (M.data[M.off[b] + q] != 0.0) {
        const double va = A.data[A.off[b] + q];
        s += va * Bv.data[Bv.off[b] + q];
        m = fabs(va) > m ? fabs(va) : m;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    const double m2 = __shfl_xor(m, o);
    m = m2 > m ? m2 : m;
  }
  __shared__ double ss[4], sm[4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { ss[w] = s; sm[w] = m; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 1; q < 4; ++q) { s += ss[q]; m = sm[q] > m ? sm[q] : m; }
    const long long slot = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    part[2 * slot] = s;
    part[2 * slot + 1] = m;
  }
}

// Round 5: the Krylov vector passes fused with their reductions.  Every vector of the iteration is exactly 0.0 on the coarse cells
// a finer level covers (the right-hand side is zeroed there once, k_smooth_zero_covered ends every operator application, and linear
// combinations of zeros are zeros), so a product or a maximum over ALL valid cells equals the one over the uncovered cells bit for
// bit (x + 0.0 = x) and the mask multifab need not be read.  Block partials: {sum0, sum1, max}.
__device__ __forceinline__ void smooth_block_reduce(double s0, double s1, double m, double* part) {
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o);
    s1 += __shfl_xor(s1, o);
    const double m2 = __shfl_xor(m, o);
    m = m2 > m ? m2 : m;
  }
  __shared__ double ss0[4], ss1[4], sm[4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { ss0[w] = s0; ss1[w] = s1; sm[w] = m; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 1; q < 4; ++q) { s0 += ss0[q]; s1 += ss1[q]; m = sm[q] > m ? sm[q] : m; }
    const long long slot = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    part[3 * slot] = s0;
    part[3 * slot + 1] = s1;
    part[3 * slot + 2] = m;
  }
}
// sum0 = a . b, sum1 = c . d (hc = 0: not formed), max = max |a|
__global__ __launch_bounds__(256) void k_smooth_dot2(DLevelView L, DMFView A, DMFView Bv, DMFView Cv, DMFView Dv, int hc, double* part) {
  double s0 = 0.0, s1 = 0.0, m = 0.0;
  {
    PA_BOX_LOOP(L) {
      int i, j, k;
      it.cell(t, i, j, k);
      const long long q = fab_index(it.B, A.ng, A.ncomp, 0, i, j, k);
      const double va = A.data[A.off[b] + q];
      s0 += va * Bv.data[Bv.off[b] + q];
      if (hc) s1 += Cv.data[Cv.off[b] + q] * Dv.data[Dv.off[b] + q];
      m = fabs(va) > m ? fabs(va) : m;
    }
  }
  smooth_block_reduce(s0, s1, m, part);
}
// z = a x + bc y + c z (hz = 0: z is only written) and, of the NEW z: sum0 = z . w (hw = 0: not formed), max = max |z|
__global__ __launch_bounds__(256) void k_smooth_lincomb_red(DLevelView L, double a, DMFView X, double bc, DMFView Y, double c, DMFView Z, int hz, DMFView W, int hw, double* part) {
  double s0 = 0.0, m = 0.0;
  {
    PA_BOX_LOOP(L) {
      int i, j, k;
      it.cell(t, i, j, k);
      const long long q = fab_index(it.B, Z.ng, Z.ncomp, 0, i, j, k);
      double v = hz ? c * Z.data[Z.off[b] + q] : 0.0;
      v += a * X.data[X.off[b] + q];
      v += bc * Y.data[Y.off[b] + q];
      Z.data[Z.off[b] + q] = v;
      if (hw) s0 += v * W.data[W.off[b] + q];
      m = fabs(v) > m ? fabs(v) : m;
    }
  }
  smooth_block_reduce(s0, 0.0, m, part);
}

namespace {
struct Vecs {  // one 1-comp ng-1 vector per level; owned, or work multifabs kept with the levels (pa_level_scratch)
  std::vector<pa_mf*> v;
  bool owned = true;
  ~Vecs() { if (owned) for (pa_mf* m : v) pa_mf_destroy(m); }
};
dim3 box_grid(const pa_level* L, unsigned gx = 0) {
  if (!gx) gx = (unsigned)std::min<long long>(((long long)L->maxn[0] * L->maxn[1] * L->maxn[2] + 255) / 256, 1024);
  return dim3(gx, (unsigned)L->boxes.size());
}
}  // namespace

struct SmoothSolver {
  pa_ctx* ctx;
  int nlev, ratio;
  double dt;
  int32_t bc[3];
  std::vector<const pa_level*> lev;
  Vecs r, rh, p, v, s, t, mask, ph, sh;
  // sharded hierarchy (dist): rs[l] restricts level l onto level l - 1; cfw[l] lives on rs[l]->cf (valid cells: child averages,
  // face ghosts: fine fluxes); fr[l] / fm[l] on level l: the received fluxes of level l + 1 (6 components) and where they apply
  bool dist = false;
  std::vector<RsPlan*> rs;
  Vecs cfw, fr, fm;
  long long nallreduce = 0, nexchange = 0;
  // A failure on THIS rank inside an operator application or a local dot product (a launch error, a failed copy) does not return
  // at once on a sharded hierarchy: the rank keeps making the same exchanges as its peers and the flag travels with the next
  // reduction, where every rank sees it and leaves together -- a rank that returned alone would leave the others blocked in
  // pa_xexchange / pa_allreduce.
  bool lerr = false;
  int fail_local() { lerr = true; return dist ? 0 : 1; }

  template <class K, class... A>
  void on_boxes(K kern, const pa_level* L, dim3 g, A... a) {  // a rank may own no box of a level
    if (g.y) hipLaunchKernelGGL(kern, g, dim3(256), 0, ctx->stream, a...);
  }
  static long long face_cells(const pa_level* LF, int ratio) {
    return std::max((long long)LF->maxn[1] * LF->maxn[2], std::max((long long)LF->maxn[0] * LF->maxn[2], (long long)LF->maxn[0] * LF->maxn[1])) / ratio;
  }
  int setup_dist() {
    rs.assign((size_t)nlev, nullptr);
    cfw.v.assign((size_t)nlev, nullptr); fr.v.assign((size_t)nlev, nullptr); fm.v.assign((size_t)nlev, nullptr);
    for (int l = 1; l < nlev; ++l) {
      rs[l] = pa_rs_plan(ctx, lev[l], lev[l - 1], ratio);
      if (!rs[l]) return 1;
      cfw.v[l] = pa_mf_create(ctx, rs[l]->cf, 1, 1, nullptr);
      fr.v[l - 1] = pa_mf_create(ctx, lev[l - 1], 6, 0, nullptr);
      fm.v[l - 1] = pa_mf_create(ctx, lev[l - 1], 6, 0, nullptr);
      if (!cfw.v[l] || !fr.v[l - 1] || !fm.v[l - 1]) return 1;
    }
    // the plans (and their buffers) every operator application will ask for: built here, before the first collective, so that a
    // rank that cannot allocate them fails BEFORE its peers enter an exchange (advisor finding, round 4)
    for (int l = 0; l < nlev; ++l)
      if (!pa_fb_plan(ctx, lev[l], 1)) return 1;
    for (int l = 1; l < nlev; ++l) {
      CsPlan* cs = pa_cs_plan(ctx, lev[l], lev[l - 1], 0, 0, 0);
      if (!cs || (cs->cs && !cs->mf(ctx, 1))) return 1;
    }
    return 0;
  }
  int setup_dist_mask() {
    // where the fluxes of the finer level apply: the same exchange once with 1.0 on every coarse-fine face
    if (flux_exchange(nullptr, fm, 1)) return 1;
    for (int l = 0; l + 1 < nlev; ++l) on_boxes(k_smooth_fluxmask, lev[l], box_grid(lev[l]), lev[l]->view, fm.v[l]->view);
    PA_HIP(hipGetLastError());
    return 0;
  }
  // fine fluxes (or the mask) of every level into the registers of the level below: one grouped exchange for the hierarchy
  int flux_exchange(Vecs* X, Vecs& R, int mask_mode) {
    std::vector<XJob> jobs;
    for (int l = nlev - 1; l > 0; --l) {
      const pa_level* LF = lev[l];
      if (!LF->sfaces.empty())
        hipLaunchKernelGGL(k_smooth_fluxreg, dim3((unsigned)((face_cells(LF, ratio) + 255) / 256), (unsigned)LF->sfaces.size()), dim3(256), 0, ctx->stream, LF->view,
                           X ? X->v[l]->view : cfw.v[l]->view, rs[l]->cf->view, cfw.v[l]->view, ratio, mask_mode);
      for (int o = 0; o < 6; ++o) jobs.push_back({&rs[l]->flux[o], cfw.v[l], 0, R.v[l - 1], o, 1});
    }
    PA_HIP(hipGetLastError());
    ++nexchange;
    return pa_xexchange(ctx, (int)jobs.size(), jobs.data());
  }

  // One rank: the vectors are work multifabs kept with the levels (role > 0; zeroed here, as a fresh allocation would be) -- 30-50 GB
  // of hipMalloc + hipFree per solve at the headline size took 0.8-1.2 s, more than the preconditioned iteration itself.
  // Sharded hierarchy: allocated and freed per solve as before.
  int alloc(Vecs& V, int role = 0) {
    for (int l = 0; l < nlev; ++l) {
      pa_mf* m = (role > 0 && !dist) ? pa_level_scratch(ctx, lev[l], 1, 1, role) : pa_mf_create(ctx, lev[l], 1, 1, nullptr);
      if (!m) return 1;
      V.v.push_back(m);
      if (role > 0 && !dist) {
        V.owned = false;
        if (m->total > 0 && hipMemsetAsync(m->data, 0, sizeof(double) * (size_t)m->total, ctx->stream) != hipSuccess) return 1;
      }
    }
    return 0;
  }
  // one stencil pass over a level as a z-marching launch (k_smooth_march): mode 0 y = A x, 1 a damped-Jacobi step, 2 the residual;
  // M non-null: y = 0 on the cells the finer level covers.  PA_SMOOTH_MARCH=0 (read per solve): the cell-per-thread kernels
  bool use_march = true;
  int march_kz = 64;
  std::map<const pa_level*, unsigned> march_gx;
  void stencil(int mode, const pa_level* L, pa_mf* X, pa_mf* R, pa_mf* Y, pa_mf* M, double om) {
    if (L->boxes.empty()) return;
    if (!use_march) {
      if (mode == 0) hipLaunchKernelGGL(k_smooth_apply, box_grid(L), dim3(256), 0, ctx->stream, L->view, X->view, Y->view, dt);
      else if (mode == 1) hipLaunchKernelGGL(k_smooth_jacobi<0>, box_grid(L), dim3(256), 0, ctx->stream, L->view, X->view, R->view, Y->view, dt, om);
      else hipLaunchKernelGGL(k_smooth_jacobi<1>, box_grid(L), dim3(256), 0, ctx->stream, L->view, X->view, R->view, Y->view, dt, 0.0);
      if (M) hipLaunchKernelGGL(k_smooth_zero_covered, box_grid(L), dim3(256), 0, ctx->stream, L->view, Y->view, M->view);
      return;
    }
    constexpr int TY = 4;
    unsigned& gx = march_gx[L];
    if (!gx) {
      gx = 8;
      for (const DBox& B : L->boxes) {
        const int nx = B.hi[0] - B.lo[0] + 1, ny = B.hi[1] - B.lo[1] + 1, nz = B.hi[2] - B.lo[2] + 1;
        const int TX = nx <= 32 ? 32 : 64, rows = nx <= 32 ? 2 * TY : TY;
        const unsigned ncol = (unsigned)(((nx + TX - 1) / TX) * ((nz + march_kz - 1) / march_kz)), ty = (unsigned)((ny + rows - 1) / rows);
        gx = std::max(gx, 8u * ((ncol + 7u) / 8u) * ty);
      }
      gx = std::min(gx, 4096u);  // a multiple of 8 either way: the kernel strides over the rest
    }
    const dim3 g(gx, (unsigned)L->boxes.size());
    SmArgs A{L->view, X->view, R ? R->view : X->view, Y->view, M ? M->view : X->view, dt, om, march_kz};
#define PA_SM(MD, MK) hipLaunchKernelGGL((k_smooth_march<MD, MK, TY>), g, dim3(64 * TY), 0, ctx->stream, A)
    if (mode == 0) { if (M) PA_SM(0, true); else PA_SM(0, false); }
    else if (mode == 1) { if (M) PA_SM(1, true); else PA_SM(1, false); }
    else { if (M) PA_SM(2, true); else PA_SM(2, false); }
#undef PA_SM
  }
  int apply(Vecs& X, Vecs& Y) {  // y = A x (x: covered cells and ghosts are overwritten)
    for (int l = nlev - 1; l > 0; --l) {
      if (!dist) {
        hipLaunchKernelGGL(k_smooth_avgdown, box_grid(lev[l]), dim3(256), 0, ctx->stream, lev[l]->view, X.v[l]->view, lev[l - 1]->view, X.v[l - 1]->view, ratio);
        continue;
      }
      on_boxes(k_smooth_avgdown_cf, lev[l], box_grid(lev[l]), lev[l]->view, X.v[l]->view, rs[l]->cf->view, cfw.v[l]->view, ratio);
      const XJob J = {&rs[l]->down, cfw.v[l], 0, X.v[l - 1], 0, 1};  // level l - 1 must be complete before it is restricted in turn
      ++nexchange;
      if (pa_xexchange(ctx, 1, &J) && fail_local()) return 1;
    }
    std::vector<const pa_mf*> crse((size_t)nlev, nullptr);
    if (dist) {
      // every ghost fill of the operator reads VALID cells only (same-level neighbours; the coarse cells under the coarse-fine
      // faces), and those are final once the restriction is done: the cross-rank half of all of them is ONE grouped exchange
      std::vector<XJob> jobs;
      for (int l = 0; l < nlev; ++l) {
        XPlan* P = pa_fb_plan(ctx, lev[l], 1);  // built in setup_dist
        if (!P) return 1;
        jobs.push_back({P, X.v[l], 0, X.v[l], 0, 1});
      }
      for (int l = 1; l < nlev; ++l) {
        CsPlan* cs = pa_cs_plan(ctx, lev[l], lev[l - 1], 0, 0, 0);
        if (!cs) return 1;
        pa_mf* m = cs->mf(ctx, 1);
        if (cs->cs && !m) return 1;
        crse[(size_t)l] = m;  // null: no coarse-fine ghost cell on this rank
        jobs.push_back({&cs->x, X.v[l - 1], 0, m, 0, 1});
      }
      ++nexchange;
      if (pa_xexchange(ctx, (int)jobs.size(), jobs.data()) && fail_local()) return 1;
    }
    for (int l = 0; l < nlev; ++l) {
      if (dist) {
        if (pa_fill_boundary_impl(ctx, X.v[l], 0, 1, 1, 1) && fail_local()) return 1;
        if (pa_apply_bc_impl(ctx, X.v[l], 0, crse[(size_t)l], 0, bc, ratio, -1, 0, nullptr) && fail_local()) return 1;
      } else {
        if (pa_fill_boundary(ctx, X.v[l], 0, 1, 1)) return 1;
        if (pa_apply_bc(ctx, X.v[l], 0, l ? X.v[l - 1] : nullptr, 0, bc, ratio, -1)) return 1;
      }
      // (the cells under level l + 1 are zeroed in the same pass: reflux touches uncovered cells only)
      stencil(0, lev[l], X.v[l], nullptr, Y.v[l], (use_march && l + 1 < nlev) ? mask.v[l] : nullptr, 0.0);
    }
    if (dist) {
      if (nlev > 1) {
        if (flux_exchange(&X, fr, 0) && fail_local()) return 1;
        for (int l = 0; l + 1 < nlev; ++l)
          on_boxes(k_smooth_reflux_apply, lev[l], box_grid(lev[l]), lev[l]->view, X.v[l]->view, Y.v[l]->view, fr.v[l]->view, fm.v[l]->view, dt);
      }
    } else {
      for (int l = nlev - 1; l > 0; --l) {
        const pa_level* LF = lev[l];
        if (LF->sfaces.empty()) continue;
        const long long nf = face_cells(LF, ratio);  // >= coarse faces per fine face (2-D planes: not refined in z)
        hipLaunchKernelGGL(k_smooth_reflux, dim3((unsigned)((nf + 255) / 256), (unsigned)LF->sfaces.size()), dim3(256), 0, ctx->stream, LF->view, X.v[l]->view,
                           lev[l - 1]->view, X.v[l - 1]->view, Y.v[l - 1]->view, dt, ratio);
      }
    }
    if (!use_march)
      for (int l = 0; l + 1 < nlev; ++l) on_boxes(k_smooth_zero_covered, lev[l], box_grid(lev[l]), lev[l]->view, Y.v[l]->view, mask.v[l]->view);
    if (hipGetLastError() != hipSuccess) { pa_fail(ctx, "pa_smooth_solve: a kernel launch of the operator failed"); if (fail_local()) return 1; }
    return 0;
  }
  // ---- fused passes (round 5): one launch per level writes its block partials behind the previous level's, ONE read-back and one
  // host sum in a fixed order for the hierarchy; red[0..1] sums, red[2] maximum of this rank
  dim3 red_grid(const pa_level* L) const {
    const long long cells = (long long)L->maxn[0] * L->maxn[1] * L->maxn[2];
    return dim3((unsigned)std::min<long long>(std::max<long long>(cells / (256 * 32), 64), 1024), (unsigned)L->boxes.size());
  }
  template <class F>
  int reduce_levels(F launch, double red[3]) {
    size_t tot = 0;
    std::vector<size_t> off((size_t)nlev, 0);
    for (int l = 0; l < nlev; ++l) {
      const dim3 g = red_grid(lev[l]);
      off[(size_t)l] = tot;
      tot += 3 * (size_t)g.x * g.y;
    }
    red[0] = red[1] = red[2] = 0.0;
    if (!tot) return 0;
    if (pa_ensure_red(ctx, tot)) return 1;
    for (int l = 0; l < nlev; ++l) {
      const dim3 g = red_grid(lev[l]);
      if (g.y) launch(l, g, ctx->d_red + off[(size_t)l]);
    }
    PA_HIP(hipGetLastError());
    hred.resize(tot);
    PA_HIP(hipMemcpyAsync(hred.data(), ctx->d_red, tot * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PA_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t q = 0; q + 2 < tot + 1 && q < tot; q += 3) { red[0] += hred[q]; red[1] += hred[q + 1]; red[2] = std::max(red[2], hred[q + 2]); }  // fixed order
    return 0;
  }
  std::vector<double> hred;
  // the ranks' parts of a reduction: nsum sums (red[0 .. nsum-1]) and, with_max, the maximum red[2]; the local error flag rides along
  int share(double red[3], int nsum, bool with_max) {
    if (!dist) return 0;
    double e = 0.0;
    if (nsum > 0) {
      double v[3] = {lerr ? 0.0 : red[0], lerr ? 0.0 : red[1], lerr ? 1.0 : 0.0};
      if (nsum == 1) v[1] = v[2];
      ++nallreduce;
      if (pa_allreduce(ctx, v, nsum + 1, 2)) return 1;
      red[0] = v[0];
      if (nsum == 2) red[1] = v[1];
      e += v[nsum];
    }
    if (with_max) {
      double v[2] = {lerr ? 0.0 : red[2], lerr ? 1.0 : 0.0};
      ++nallreduce;
      if (pa_allreduce(ctx, v, 2, 1)) return 1;
      red[2] = v[0];
      e += v[1];
    }
    if (e != 0.0) return lerr ? 1 : pa_fail(ctx, "pa_smooth_solve: another rank failed");
    return 0;
  }
  // a . b (+ c . d) and max |a| in one pass
  int fdot(Vecs& A, Vecs& B, Vecs* Cc, Vecs* Dd, double red[3], int nsum, bool with_max) {
    if (reduce_levels([&](int l, dim3 g, double* part) {
          hipLaunchKernelGGL(k_smooth_dot2, g, dim3(256), 0, ctx->stream, lev[l]->view, A.v[l]->view, B.v[l]->view, (Cc ? Cc : &A)->v[l]->view, (Dd ? Dd : &B)->v[l]->view, Cc ? 1 : 0, part);
        }, red) && fail_local()) return 1;
    return share(red, nsum, with_max);
  }
  // z = a x + b y + c z (c == 0: z is only written) with z . w (W non-null) and max |z| of the new z in the same pass
  int flin(double a, Vecs& X, double b, Vecs& Y, double c, Vecs& Z, Vecs* W, double red[3], bool with_max) {
    if (reduce_levels([&](int l, dim3 g, double* part) {
          hipLaunchKernelGGL(k_smooth_lincomb_red, g, dim3(256), 0, ctx->stream, lev[l]->view, a, X.v[l]->view, b, Y.v[l]->view, c, Z.v[l]->view, c != 0.0 ? 1 : 0, (W ? W : &Z)->v[l]->view, W ? 1 : 0, part);
        }, red) && fail_local()) return 1;
    return share(red, W ? 1 : 0, with_max);
  }
  void zero_covered(Vecs& Y) {
    for (int l = 0; l + 1 < nlev; ++l) on_boxes(k_smooth_zero_covered, lev[l], box_grid(lev[l]), lev[l]->view, Y.v[l]->view, mask.v[l]->view);
  }
  // this rank's part of a . b and of max |a| over the uncovered cells
  int ldot(Vecs& A, Vecs& B, double* d, double* amax) {
    double sd = 0.0, sm = 0.0;
    for (int l = 0; l < nlev; ++l) {
      const dim3 g = box_grid(lev[l], 64);
      const size_t np = (size_t)g.x * g.y;
      if (!np) continue;
      if (pa_ensure_red(ctx, 2 * np)) return 1;
      hipLaunchKernelGGL(k_smooth_dot, g, dim3(256), 0, ctx->stream, lev[l]->view, A.v[l]->view, B.v[l]->view, mask.v[l]->view, ctx->d_red);
      PA_HIP(hipGetLastError());
      std::vector<double> h(2 * np);
      PA_HIP(hipMemcpyAsync(h.data(), ctx->d_red, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      PA_HIP(hipStreamSynchronize(ctx->stream));
      for (size_t q = 0; q < np; ++q) { sd += h[2 * q]; sm = std::max(sm, h[2 * q + 1]); }  // fixed order
    }
    *d = sd;
    *amax = sm;
    return 0;
  }
  // what: 1 = the dot product, 2 = max |a|, 3 = both (each is one reduction over the ranks of a sharded hierarchy; every rank
  // gets the same bits back, so every rank takes the same branches of the iteration)
  // (the local error flag rides along as a second / third element: a sum or a maximum of non-negative flags is non-zero iff some
  // rank failed, and every rank receives the same value)
  int dot(Vecs& A, Vecs& B, double* d, double* amax, int what = 3) {
    if (ldot(A, B, d, amax) && fail_local()) return 1;
    if (dist) {
      double e = 0.0;
      if (what & 1) {
        double v[2] = {lerr ? 0.0 : *d, lerr ? 1.0 : 0.0};
        ++nallreduce;
        if (pa_allreduce(ctx, v, 2, 2)) return 1;
        *d = v[0]; e += v[1];
      }
      if (what & 2) {
        double v[2] = {lerr ? 0.0 : *amax, lerr ? 1.0 : 0.0};
        ++nallreduce;
        if (pa_allreduce(ctx, v, 2, 1)) return 1;
        *amax = v[0]; e += v[1];
      }
      if (e != 0.0) return lerr ? 1 : pa_fail(ctx, "pa_smooth_solve: another rank failed");
    }
    return 0;
  }
  int dot2(Vecs& A, Vecs& B, Vecs& C, Vecs& D, double* ab, double* cd) {  // two dot products, one reduction
    double v[3] = {0.0, 0.0, 0.0}, dummy;
    if ((ldot(A, B, &v[0], &dummy) || ldot(C, D, &v[1], &dummy)) && fail_local()) return 1;
    if (dist) {
      if (lerr) { v[0] = v[1] = 0.0; v[2] = 1.0; }
      ++nallreduce;
      if (pa_allreduce(ctx, v, 3, 2)) return 1;
      if (v[2] != 0.0) return lerr ? 1 : pa_fail(ctx, "pa_smooth_solve: another rank failed");
    }
    *ab = v[0];
    *cd = v[1];
    return 0;
  }
  // ---- multigrid preconditioner (one rank, 3-D): z = M^-1 r by one V(2,4) cycle of damped Jacobi over the AMR levels and, below
  // level 0, coarsened copies of it (until dt / dx^2 is small or the boxes stop halving).  curvature.cpp:381-399 solves with MLMG;
  // unpreconditioned BiCGStab needs ~sqrt(cond) iterations (165 at dt / dx^2 = 42 on the finest level, hundreds for a plotfile in
  // physical units), a V-cycle per application keeps the count at a handful whatever dt.  Level-local problems: the finer level's
  // residual is averaged down onto the cells it covers, a level's coarse-fine ghost cells come from the coarser level's correction by
  // the operator's own applyBC (zero on the way down), the flux mismatch at coarse-fine faces is left to the Krylov iteration.  M is a
  // fixed linear operator, so BiCGStab's recurrences hold; the solution it converges to is the unpreconditioned one.
  // owned: a coarsened copy of level 0 (its vectors too).  Sharded hierarchy: rsd restricts this level onto the one below (cfd lives on
  // rsd->cf: the fine rank's child averages), csp brings the parents of this rank's cells of this level from their owners (pmf on csp->cs)
  struct MgLev {
    const pa_level* L = nullptr; pa_level* owned = nullptr; pa_mf *e = nullptr, *r = nullptr, *w = nullptr;
    RsPlan* rsd = nullptr; pa_mf* cfd = nullptr; CsPlan* csp = nullptr; pa_mf* pmf = nullptr; bool own_cfd = false, own_vec = false;
  };
  std::vector<MgLev> mg;
  int mg_sub = 0;  // coarsened copies of level 0: mg[0 .. mg_sub - 1]; AMR level l = mg[mg_sub + l]
  ~SmoothSolver() {
    for (MgLev& g : mg) {
      if (g.own_cfd) pa_mf_destroy(g.cfd);
      if (g.own_vec) { pa_mf_destroy(g.e); pa_mf_destroy(g.r); pa_mf_destroy(g.w); }  // (one rank: the AMR levels' vectors are kept with the levels)
    }
    for (MgLev& g : mg)
      if (g.owned) pa_level_destroy(g.owned);  // after every multifab on it (and on the plans it owns) is gone
  }
  int mg_setup() {
    const pa_level* Lc = lev[0];  // (every coarsened level goes into `mg` at once: the destructor frees it whatever happens later)
    // coarsen level 0 while every box halves evenly, stays >= 4 cells thick and dt / dx^2 of the CURRENT coarsest level is not small yet
    const int nd = lev[0]->domlo[2] == lev[0]->domhi[2] ? 2 : 3;  // a 2-D hierarchy is one plane of cells per level: not coarsened in z
    for (int n = 0; n < 8; ++n) {
      double q = 0.0;
      for (int d = 0; d < nd; ++d) q = std::max(q, dt * Lc->dxinv[d] * Lc->dxinv[d]);
      if (q < 0.25) break;
      bool ok = true;
      std::vector<int32_t> b6;
      const std::vector<DBox>& all = dist ? Lc->gboxes : Lc->boxes;  // the WHOLE BoxArray: every rank of a sharded hierarchy decides alike
      for (const DBox& B : all)
        for (int d = 0; d < nd; ++d) ok = ok && !(B.lo[d] & 1) && ((B.hi[d] - B.lo[d] + 1) % 2 == 0) && (B.hi[d] - B.lo[d] + 1) >= 8;
      for (int d = 0; d < nd; ++d) ok = ok && !(Lc->domlo[d] & 1) && ((Lc->domhi[d] - Lc->domlo[d] + 1) % 2 == 0);
      if (!ok) break;
      for (const DBox& B : all) {
        for (int d = 0; d < 3; ++d) b6.push_back(d < nd ? B.lo[d] / 2 : B.lo[d]);
        for (int d = 0; d < 3; ++d) b6.push_back(d < nd ? (B.hi[d] + 1) / 2 - 1 : B.hi[d]);
      }
      int32_t dlo[3], dhi[3], per[3];
      for (int d = 0; d < 3; ++d) { dlo[d] = d < nd ? Lc->domlo[d] / 2 : Lc->domlo[d]; dhi[d] = d < nd ? (Lc->domhi[d] + 1) / 2 - 1 : Lc->domhi[d]; per[d] = Lc->is_per[d]; }
      pa_level* Ln = dist ? pa_level_create_sharded(ctx, (int)all.size(), b6.data(), Lc->gowner.data(), Lc->rank, Lc->nranks, dlo, dhi, per, Lc->prob_lo, Lc->prob_hi)
                          : pa_level_create(ctx, (int)all.size(), b6.data(), dlo, dhi, per, Lc->prob_lo, Lc->prob_hi);
      if (!Ln) return 1;
      MgLev g;
      g.L = g.owned = Ln;
      mg.push_back(g);
      Lc = Ln;
    }
    mg_sub = (int)mg.size();
    std::reverse(mg.begin(), mg.end());  // coarsest first
    {  // Jacobi steps on the coarsest level: 8 where the coarsening went on until dt / dx^2 < 0.25; where it had to stop earlier (boxes that
       // do not halve: odd corners, fewer than 8 cells) the coarsest problem is still stiff and gets ~3 sqrt(cond) steps (at most 64)
      double q = 0.0;
      for (int d = 0; d < nd; ++d) q = std::max(q, dt * Lc->dxinv[d] * Lc->dxinv[d]);
      nub = std::max(8, std::min(64, (int)(3.0 * std::sqrt(1.0 + 12.0 * q))));
    }
    for (int l = 0; l < nlev; ++l) { MgLev g; g.L = lev[l]; mg.push_back(g); }
    for (MgLev& g : mg) {
      g.own_vec = g.owned || dist;
      g.e = g.own_vec ? pa_mf_create(ctx, g.L, 1, 1, nullptr) : pa_level_scratch(ctx, g.L, 1, 1, 20);
      g.r = g.own_vec ? pa_mf_create(ctx, g.L, 1, 1, nullptr) : pa_level_scratch(ctx, g.L, 1, 1, 21);
      g.w = g.own_vec ? pa_mf_create(ctx, g.L, 1, 1, nullptr) : pa_level_scratch(ctx, g.L, 1, 1, 22);
      if (!g.e || !g.r || !g.w) return 1;
    }
    if (dist) {  // every plan and buffer the V-cycle will ask for, before the first collective (see setup_dist)
      for (size_t g = 0; g < mg.size(); ++g) {
        MgLev& X = mg[g];
        if (!pa_fb_plan(ctx, X.L, 1)) return 1;
        if (g == 0) continue;
        const int l = (int)g - mg_sub;  // AMR level (> 0: its restriction plan and buffer exist already)
        if (l >= 1) { X.rsd = rs[(size_t)l]; X.cfd = cfw.v[(size_t)l]; }
        else {
          X.rsd = pa_rs_plan(ctx, X.L, mg[g - 1].L, ratio);
          if (!X.rsd) return 1;
          X.cfd = pa_mf_create(ctx, X.rsd->cf, 1, 1, nullptr);
          X.own_cfd = true;
          if (!X.cfd) return 1;
        }
        X.csp = pa_cs_plan(ctx, X.L, mg[g - 1].L, 2, 0, 0, ratio);  // the coarse cells under this rank's boxes of level g: the parents
        if (!X.csp) return 1;
        X.pmf = X.csp->mf(ctx, 1, 2);
        if (X.csp->cs && !X.pmf) return 1;
        if ((int)g > mg_sub) {  // coarse-fine ghost cells of the correction: the operator's coarse-source plan (and its buffer)
          CsPlan* c0 = pa_cs_plan(ctx, X.L, mg[g - 1].L, 0, 0, 0);
          if (!c0 || (c0->cs && !c0->mf(ctx, 1))) return 1;
        }
      }
    }
    return 0;
  }
  // ghost cells of the CURRENT correction of MG level g (level-local: same-level neighbours, walls, and at coarse-fine faces the coarser
  // level's current correction through the operator's own applyBC)
  // (sharded: pa_fill_boundary / pa_apply_bc exchange inside; every rank makes the same calls)
  int mg_ghosts(int g) {
    MgLev& X = mg[(size_t)g];
    if (pa_fill_boundary(ctx, X.e, 0, 1, 1) && fail_local()) return 1;
    if (pa_apply_bc(ctx, X.e, 0, g > mg_sub ? mg[(size_t)g - 1].e : nullptr, 0, bc, ratio, -1) && fail_local()) return 1;
    return 0;
  }
  // nu damped-Jacobi steps on A_g e = r_g, each ONE pass (k_smooth_jacobi<0>: reads e and r, writes the new e into the level's second
  // buffer; the two are swapped: X.e is always the current one); zero_start: e is 0 (the first step needs no operator application)
  int mg_smooth(int g, int nu, bool zero_start) {
    MgLev& X = mg[(size_t)g];
    double D = 1.0;
    for (int d = 0; d < 3; ++d)
      if (!(d == 2 && X.L->domlo[2] == X.L->domhi[2])) D += 2.0 * dt * X.L->dxinv[d] * X.L->dxinv[d];
    const double om = jac_omega / D;
    for (int it = 0; it < nu; ++it) {
      if (it == 0 && zero_start) {
        on_boxes(k_smooth_axpbypcz, X.L, box_grid(X.L), X.L->view, om, X.r->view, 1, 0.0, X.r->view, 0, 0.0, X.e->view);  // e = om r
        continue;
      }
      if (mg_ghosts(g)) return 1;
      stencil(1, X.L, X.e, X.r, X.w, nullptr, om);
      std::swap(X.e, X.w);
    }
    return 0;
  }
  // pre- / post-smoothing steps, steps on the coarsest level (PA_MG_NU="nu1 nu2 nub omega", read per solve).  V(2,4): a post-smoothing step
  // is one pass, a pre-smoothing step brings a residual pass with it -- at the headline size 7 iterations in 0.44 / 0.48 s for
  // dt / dx^2 = 42 / 250 against 9 / 11 in 0.51 / 0.59 for V(2,2), 8 / 8 in 0.46 / 0.49 for V(2,3), 6 / 7 in 0.44 / 0.52 for V(3,4)
  int nu1 = 2, nu2 = 4, nub = 8;
  double jac_omega = 0.85;
  int vcycle(Vecs& R, Vecs& Z) {
    const int G = (int)mg.size();
    // the right-hand side of the FINEST level is only read (nothing is averaged down onto it): R itself, no copy; its correction starts
    // as om r on every valid cell (no memset); the coarser levels' corrections must read 0 as the finer levels' coarse-fine ghost source
    struct Borrow { pa_mf*& slot; pa_mf* mine; ~Borrow() { slot = mine; } } borrow{mg[(size_t)G - 1].r, mg[(size_t)G - 1].r};
    mg[(size_t)G - 1].r = R.v[(size_t)nlev - 1];
    for (int l = 0; l + 1 < nlev; ++l) on_boxes(k_smooth_copy, lev[l], box_grid(lev[l]), lev[l]->view, R.v[l]->view, 0, mg[(size_t)(mg_sub + l)].r->view, 0);
    for (int g = 0; g + 1 < G; ++g) PA_HIP(hipMemsetAsync(mg[(size_t)g].e->data, 0, sizeof(double) * (size_t)mg[(size_t)g].e->total, ctx->stream));
    for (int g = G - 1; g > 0; --g) {
      if (mg_smooth(g, nu1, true)) return 1;
      if (mg_ghosts(g)) return 1;
      MgLev& X = mg[(size_t)g];
      stencil(2, X.L, X.e, X.r, X.w, nullptr, 0.0);  // w = r - A e
      MgLev& C = mg[(size_t)g - 1];
      if (!dist) {
        hipLaunchKernelGGL(k_smooth_avgdown, box_grid(X.L), dim3(256), 0, ctx->stream, X.L->view, X.w->view, C.L->view, C.r->view, ratio);
      } else {  // the fine rank averages its own boxes, one exchange hands the averages to the coarse owners
        on_boxes(k_smooth_avgdown_cf, X.L, box_grid(X.L), X.L->view, X.w->view, X.rsd->cf->view, X.cfd->view, ratio);
        const XJob J = {&X.rsd->down, X.cfd, 0, C.r, 0, 1};
        ++nexchange;
        if (pa_xexchange(ctx, 1, &J) && fail_local()) return 1;
      }
    }
    if (mg_smooth(0, G > 1 ? nub : std::max(nu1 + nu2, 2), true)) return 1;
    for (int g = 1; g < G; ++g) {
      MgLev& X = mg[(size_t)g];
      MgLev& C = mg[(size_t)g - 1];
      if (!dist) {
        on_boxes(k_smooth_prolong_add, X.L, box_grid(X.L), X.L->view, X.e->view, C.L->view, C.e->view, ratio);
      } else {  // the parents of this rank's cells come from their owners (coarse-source plan of the coarsened boxes)
        const XJob J = {&X.csp->x, C.e, 0, X.pmf, 0, 1};
        ++nexchange;
        if (pa_xexchange(ctx, 1, &J) && fail_local()) return 1;
        if (X.csp->cs && X.pmf) on_boxes(k_smooth_prolong_add, X.L, box_grid(X.L), X.L->view, X.e->view, X.csp->cs->view, X.pmf->view, ratio);
      }
      if (mg_smooth(g, nu2, false)) return 1;
    }
    // z = the corrections: the vectors change hands (same level, same shape) instead of being copied.  z keeps the coarse levels'
    // corrections on the cells a finer level covers: its only readers are the operator (which averages the finer level down onto
    // them first) and the update of x (whose covered cells are averaged down at the end of the solve)
    for (int l = 0; l < nlev; ++l) std::swap(Z.v[(size_t)l], mg[(size_t)(mg_sub + l)].e);
    PA_HIP(hipGetLastError());
    return 0;
  }
  void axpbypcz(double a, Vecs* X, double b, Vecs* Y, double c, Vecs& Z) {
    for (int l = 0; l < nlev; ++l)
      on_boxes(k_smooth_axpbypcz, lev[l], box_grid(lev[l]), lev[l]->view, a, X ? X->v[l]->view : Z.v[l]->view, X ? 1 : 0, b, Y ? Y->v[l]->view : Z.v[l]->view,
               Y ? 1 : 0, c, Z.v[l]->view);
  }
  void copy(Vecs& S, Vecs& D) {
    for (int l = 0; l < nlev; ++l) on_boxes(k_smooth_copy, lev[l], box_grid(lev[l]), lev[l]->view, S.v[l]->view, 0, D.v[l]->view, 0);
  }
};

// A hierarchy sharded over ranks, REPLICATED form (PA_SMOOTH_REPLICATED=1): every rank gathers the right-hand side of the whole
// hierarchy (one grouped exchange, RepPlan in pa_dist.hip), runs the same composite solve on its own GPU -- same input, same
// kernels, same fixed summation order, so every rank gets the same field and the result equals the one-rank run BIT FOR BIT --
// and keeps the boxes it owns.  No speed-up with the number of GPUs and (nranks - 1) x this rank's cells of send buffer: the
// default is the distributed solve below (the reference's MLMG distributes this solve too); this form stays for runs that
// must reproduce the one-GPU bits.
static int smooth_solve_replicated(pa_ctx* ctx, int nlev, pa_mf* const* rhs, int rcomp, pa_mf* const* sol, int scomp, double dt, const int32_t bc[3], double tol,
                                   int maxiter, int* iters, double* res) {
  std::vector<RepPlan*> P((size_t)nlev, nullptr);
  Vecs R, X;
  std::vector<XJob> gj, bj;
  for (int l = 0; l < nlev; ++l) {
    if (!rhs[l] || !sol[l] || rhs[l]->lev != sol[l]->lev) return pa_fail(ctx, "pa_smooth_solve: rhs/sol on different levels");
    if (rcomp < 0 || rcomp >= rhs[l]->ncomp || scomp < 0 || scomp >= sol[l]->ncomp) return pa_fail(ctx, "pa_smooth_solve: component range");
    if (rhs[l]->lev->nranks != ctx->comm.nranks) return pa_fail(ctx, "pa_smooth_solve: the context's transport has a different number of ranks than the levels");
    P[(size_t)l] = pa_rep_plan(ctx, rhs[l]->lev);
    if (!P[(size_t)l]) return 1;
    pa_mf* r = pa_mf_create(ctx, P[(size_t)l]->rep, 1, 0, nullptr);
    pa_mf* x = pa_mf_create(ctx, P[(size_t)l]->rep, 1, 0, nullptr);
    if (r) R.v.push_back(r);
    if (x) X.v.push_back(x);
    if (!r || !x) return 1;
    gj.push_back({&P[(size_t)l]->gather, rhs[l], rcomp, r, 0, 1});
    bj.push_back({&P[(size_t)l]->back, x, 0, sol[l], scomp, 1});
  }
  if (pa_xexchange(ctx, nlev, gj.data())) return 1;
  // a solve that stopped short of `tol` still hands its last iterate back (the caller decides: pa_pipeline.hip accepts 1e-12)
  const int rc = pa_smooth_solve(ctx, nlev, R.v.data(), 0, X.v.data(), 0, dt, bc, tol, maxiter, iters, res);
  const std::string why = ctx->err;
  if (pa_xexchange(ctx, nlev, bj.data())) return 1;  // local copies only: no transport call
  PA_HIP(hipStreamSynchronize(ctx->stream));   // R, X are freed on return
  if (rc) ctx->err = why;
  return rc;
}

extern "C" int pa_smooth_solve(pa_ctx* ctx, int nlev, pa_mf* const* rhs, int rcomp, pa_mf* const* sol, int scomp, double dt, const int32_t bc[3], double tol,
                               int maxiter, int* iters, double* res) {
  PaBind bind_(ctx);
  if (!ctx || nlev <= 0 || !rhs || !sol || !bc) return pa_fail(ctx, "pa_smooth_solve: null argument");
  // A hierarchy sharded over ranks: DISTRIBUTED solve.  Every rank keeps the vectors of its own boxes; the operator's
  // average_down and reflux go through the restriction plans (pa_dist.h: RsPlan -- the fine rank restricts / sums the fluxes of
  // its own boxes, one exchange hands them to the coarse owners), its ghost fills are the sharded pa_fill_boundary /
  // pa_apply_bc, and each dot product / norm is one pa_allreduce.  The iteration is the one-rank iteration with the sums taken
  // in another order: the field agrees with the one-rank (and the oracle's) to the solve's tolerance, not bit for bit.
  const bool sharded = rhs[0] && rhs[0]->lev->nranks > 1;
  if (sharded && pa_opt().smooth_replicated) return smooth_solve_replicated(ctx, nlev, rhs, rcomp, sol, scomp, dt, bc, tol, maxiter, iters, res);
  const bool timing = pa_opt().smooth_timing != 0;  // diagnostic: setup / iteration / total wall time on stderr
  const auto tm0 = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
  SmoothSolver S;
  S.ctx = ctx; S.nlev = nlev; S.ratio = 2; S.dt = dt; S.dist = sharded;
  for (int d = 0; d < 3; ++d) S.bc[d] = bc[d];
  for (int l = 0; l < nlev; ++l) {
    if (!rhs[l] || !sol[l] || rhs[l]->lev != sol[l]->lev) return pa_fail(ctx, "pa_smooth_solve: rhs/sol on different levels");
    if (rcomp < 0 || rcomp >= rhs[l]->ncomp || scomp < 0 || scomp >= sol[l]->ncomp) return pa_fail(ctx, "pa_smooth_solve: component range");
    if ((rhs[l]->lev->nranks > 1) != sharded) return pa_fail(ctx, "pa_smooth_solve: sharded and unsharded levels in one hierarchy");
    if (sharded && rhs[l]->lev->nranks != ctx->comm.nranks) return pa_fail(ctx, "pa_smooth_solve: the context's transport has a different number of ranks than the levels");
    S.lev.push_back(rhs[l]->lev);
    if (l > 0)
      for (const DBox& B : S.lev[l]->boxes)
        for (int d = 0; d < 3; ++d) {
          if (d == 2 && S.lev[l]->domlo[2] == S.lev[l]->domhi[2]) continue;  // 2-D hierarchy: one plane per level, not refined in z
          if ((B.lo[d] & 1) || !((B.hi[d] - B.lo[d]) & 1)) return pa_fail(ctx, "pa_smooth_solve: fine boxes must be aligned to the refinement ratio 2");
        }
  }
  Vecs x;  // the solution as a 1-comp vector (sol may have other components / ghost widths)
  {
    // every allocation of the solve -- Krylov vectors, restriction / flux-register plans, ghost-fill plans and their buffers --
    // happens here, before the first collective; the ranks of a sharded hierarchy then agree on success with ONE reduction
    int bad = (S.alloc(S.r, 10) || S.alloc(S.rh, 11) || S.alloc(S.p, 12) || S.alloc(S.v, 13) || S.alloc(S.s, 14) || S.alloc(S.t, 15) || S.alloc(S.mask, 16) || S.alloc(x, 17)) ? 1 : 0;
    if (!bad && S.dist && S.setup_dist()) bad = 1;
    if (S.dist) {
      double e = bad ? 1.0 : 0.0;
      if (pa_allreduce(ctx, &e, 1, 1)) return 1;
      if (e != 0.0) return bad ? 1 : pa_fail(ctx, "pa_smooth_solve: another rank could not set the solve up");
    } else if (bad) {
      return 1;
    }
    if (S.dist && S.setup_dist_mask()) return 1;
  }
  if (!pa_opt().smooth_march) S.use_march = false;  // PA_SMOOTH_MARCH=0: the cell-per-thread stencil kernels
  // PA_SMOOTH_MG: 1 / 0 = the multigrid preconditioner on / off; default: on where the finest level's dt / dx^2
  // exceeds 8 (below that the unpreconditioned iteration needs < ~45 iterations and two V-cycles per iteration cost more than they
  // save); 3-D and 2-D hierarchies, one rank or sharded
  bool use_mg = false;
  {
    const pa_level* Lf = S.lev[(size_t)nlev - 1];
    double q = 0.0;
    for (int d = 0; d < (Lf->domlo[2] == Lf->domhi[2] ? 2 : 3); ++d) q = std::max(q, dt * Lf->dxinv[d] * Lf->dxinv[d]);
    use_mg = pa_opt().smooth_mg >= 0 ? pa_opt().smooth_mg != 0 : q > 8.0;
    if (use_mg) {
      int bad = (S.mg_setup() || S.alloc(S.ph, 18) || S.alloc(S.sh, 19)) ? 1 : 0;
      if (S.dist) {  // the ranks agree on the preconditioner's setup before its first exchange
        double e = bad ? 1.0 : 0.0;
        if (pa_allreduce(ctx, &e, 1, 1)) return 1;
        if (e != 0.0) return bad ? 1 : pa_fail(ctx, "pa_smooth_solve: another rank could not set the multigrid preconditioner up");
      } else if (bad) {
        return 1;
      }
    }
  }
  for (int l = 0; l < nlev; ++l) {
    const pa_level* L = S.lev[l];
    S.on_boxes(k_smooth_mask, L, box_grid(L), L->view, S.mask.v[l]->view, l + 1 < nlev ? S.lev[l + 1]->view : L->view, l + 1 < nlev ? 1 : 0, 2);
    S.on_boxes(k_smooth_copy, L, box_grid(L), L->view, rhs[l]->view, rcomp, S.r.v[l]->view, 0);
    S.on_boxes(k_smooth_copy, L, box_grid(L), L->view, rhs[l]->view, rcomp, S.rh.v[l]->view, 0);
  }
  PA_HIP(hipGetLastError());
  // r and r^ are zeroed on the covered coarse cells once: every vector of the iteration then stays exactly zero there (see the
  // fused kernels above) and the reductions run over all valid cells without reading the mask
  S.zero_covered(S.r);
  S.zero_covered(S.rh);
  PA_HIP(hipGetLastError());
  double red[3], rho = 1.0, alpha = 1.0, omega = 1.0;
  if (S.fdot(S.r, S.rh, nullptr, nullptr, red, 1, true)) return 1;  // rho_1 = r^ . r and ||b||_inf in one pass
  const double setup_ms = since(tm0);
  const auto tm1 = std::chrono::steady_clock::now();
  const double bnorm = red[2];
  double rho1 = red[0];
  int it = 0, status = -1;
  double rnorm = bnorm;
  if (bnorm == 0.0) status = 0;
  // (round 4's form of the iteration -- one pass per vector operation and per reduction: 42.5 against 19-25 ms per iteration on the
  // headline hierarchy -- is gone; DESIGN_HISTORY.md R5)
  constexpr bool fused = true;
  double dummy;
  while (status != 0 && it < maxiter) {
    ++it;
    if (!fused && S.dot(S.rh, S.r, &rho1, &dummy, 1)) return 1;
    if (rho1 == 0.0) { status = -2; break; }
    const double beta = (rho1 / rho) * (alpha / omega);
    if (fused) {
      S.axpbypcz(1.0, &S.r, -omega * beta, &S.v, beta, S.p);  // p = r + beta (p - omega v), one pass
    } else {
      S.axpbypcz(-omega * beta, &S.v, 0.0, nullptr, beta, S.p);
      S.axpbypcz(1.0, &S.r, 0.0, nullptr, 1.0, S.p);
    }
    if (use_mg) {  // right preconditioning: v = A M^-1 p
      if (S.vcycle(S.p, S.ph) || S.apply(S.ph, S.v)) return 1;
    } else if (S.apply(S.p, S.v)) return 1;
    double rhv;
    if (fused) {
      if (S.fdot(S.rh, S.v, nullptr, nullptr, red, 1, false)) return 1;
      rhv = red[0];
    } else if (S.dot(S.rh, S.v, &rhv, &dummy, 1)) return 1;
    if (rhv == 0.0) { status = -3; break; }
    alpha = rho1 / rhv;
    double snorm;
    if (fused) {
      if (S.flin(1.0, S.r, -alpha, S.v, 0.0, S.s, nullptr, red, true)) return 1;  // s = r - alpha v and ||s||_inf
      snorm = red[2];
    } else {
      S.copy(S.r, S.s);
      S.axpbypcz(-alpha, &S.v, 0.0, nullptr, 1.0, S.s);
      if (S.dot(S.s, S.s, &dummy, &snorm, 2)) return 1;
    }
    if (snorm <= tol * bnorm) {
      S.axpbypcz(alpha, use_mg ? &S.ph : &S.p, 0.0, nullptr, 1.0, x);
      rnorm = snorm;
      status = 0;
      break;
    }
    if (use_mg) {
      if (S.vcycle(S.s, S.sh) || S.apply(S.sh, S.t)) return 1;
    } else if (S.apply(S.s, S.t)) return 1;
    double ts, tt;
    if (fused) {
      if (S.fdot(S.t, S.s, &S.t, &S.t, red, 2, false)) return 1;  // t . s and t . t in one pass
      ts = red[0]; tt = red[1];
    } else if (S.dot2(S.t, S.s, S.t, S.t, &ts, &tt)) return 1;
    if (tt == 0.0) { status = -4; break; }
    omega = ts / tt;
    S.axpbypcz(alpha, use_mg ? &S.ph : &S.p, omega, use_mg ? &S.sh : &S.s, 1.0, x);  // x += alpha p + omega s (preconditioned: M^-1 p, M^-1 s)
    if (fused) {
      if (S.flin(1.0, S.s, -omega, S.t, 0.0, S.r, &S.rh, red, true)) return 1;  // r = s - omega t, ||r||_inf and the NEXT rho_1 = r^ . r
      rnorm = red[2];
      rho = rho1;
      rho1 = red[0];
    } else {
      S.copy(S.s, S.r);
      S.axpbypcz(-omega, &S.t, 0.0, nullptr, 1.0, S.r);
      if (S.dot(S.r, S.r, &dummy, &rnorm, 2)) return 1;
      rho = rho1;
    }
    if (rnorm <= tol * bnorm) { status = 0; break; }
    if (omega == 0.0) { status = -5; break; }
  }
  for (int l = nlev - 1; l > 0; --l) {
    if (!S.dist) {
      hipLaunchKernelGGL(k_smooth_avgdown, box_grid(S.lev[l]), dim3(256), 0, ctx->stream, S.lev[l]->view, x.v[l]->view, S.lev[l - 1]->view, x.v[l - 1]->view, 2);
      continue;
    }
    S.on_boxes(k_smooth_avgdown_cf, S.lev[l], box_grid(S.lev[l]), S.lev[l]->view, x.v[l]->view, S.rs[l]->cf->view, S.cfw.v[l]->view, 2);
    const XJob J = {&S.rs[l]->down, S.cfw.v[l], 0, x.v[l - 1], 0, 1};
    if (pa_xexchange(ctx, 1, &J)) return 1;
  }
  for (int l = 0; l < nlev; ++l) S.on_boxes(k_smooth_copy, S.lev[l], box_grid(S.lev[l]), S.lev[l]->view, x.v[l]->view, 0, sol[l]->view, scomp);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(ctx->stream));  // the work vectors are freed on return
  if (timing) fprintf(stderr, "pa_smooth_solve: setup %.1f ms, %d iterations %.1f ms (%s)\n", setup_ms, it, since(tm1), use_mg ? "multigrid-preconditioned" : "plain");
  if (iters) *iters = it;
  if (res) *res = bnorm > 0.0 ? rnorm / bnorm : 0.0;
  if (status != 0) return pa_fail(ctx, "pa_smooth_solve: BiCGStab did not reach the tolerance (status " + std::to_string(status) + " after " + std::to_string(it) + " iterations)");
  return 0;
}
