// pa_curvopts.hip -- optional outputs of curvature.cpp on gfx950:
//   Gaussian curvature (curvature.cpp:575-677), strain rate / rate-of-strain tensor (:679-757),
//   flame-normal velocity (:765-787).
// The reference builds three more MLPoisson operators per option just to differentiate three fields;
// here each option is ONE sweep that differentiates from ghost-resolved inputs (FillBoundary +
// applyBC done by the caller, pa_pipeline.hip) in the reference's operation order (cdiff).
#include "pa_internal.h"
#include "pa_fabview.h"
#include "pa_dpp.h"

struct BP3 {
  DLevelView L;
  DMFView A, B, C;
  __device__ __forceinline__ bool get(int b, FabView& a, FabView& o, FabView& c, DBox& V, double dxinv[3]) const {
    if (b >= L.nboxes) return false;
    const DBox X = L.boxes[b];
    a = mf_view(A, X, b);
    o = mf_view(B, X, b);
    c = mf_view(C, X, b);
    V = X;
    for (int d = 0; d < 3; ++d) dxinv[d] = L.dxinv[d];
    return true;
  }
};

// Hessian rows = grad(G_d) (:582-613), adjugate (:630-638), Kg = G^T adj(H) G / normgrad^4 (:659-668)
__global__ __launch_bounds__(256) void k_gauss_curv(BP3 bp, int gcomp, DMFView MNG, int ngcomp, int ccomp, double thr, int kcomp) {
  FabView G, K, C;
  DBox V;
  double dxinv[3];
  if (!bp.get(blockIdx.y, G, K, C, V, dxinv)) return;
  int i, j, k0, k1;
  if (!tile_cell(V, i, j, k0, k1)) return;
  const FabView NG = mf_view(MNG, V, blockIdx.y);
  for (int k = k0; k <= k1; ++k) {
    double H[3][3], g[3];
    for (int d = 0; d < 3; ++d) {
      const double c0 = G(i, j, k, gcomp + d);
      g[d] = c0;
      H[d][0] = cdiff(dxinv[0], G(i - 1, j, k, gcomp + d), c0, G(i + 1, j, k, gcomp + d));
      H[d][1] = cdiff(dxinv[1], G(i, j - 1, k, gcomp + d), c0, G(i, j + 1, k, gcomp + d));
      H[d][2] = cdiff(dxinv[2], G(i, j, k - 1, gcomp + d), c0, G(i, j, k + 1, gcomp + d));
    }
#define HX(n) H[0][n]
#define HY(n) H[1][n]
#define HZ(n) H[2][n]
    const double ax0 = HY(1) * HZ(2) - HZ(1) * HY(2);
    const double ay0 = HY(2) * HZ(0) - HZ(2) * HY(0);
    const double az0 = HY(0) * HZ(1) - HZ(0) * HY(1);
    const double ax1 = HX(2) * HZ(1) - HZ(2) * HX(1);
    const double ay1 = HX(0) * HZ(2) - HZ(0) * HX(2);
    const double az1 = HX(1) * HZ(0) - HZ(1) * HX(0);
    const double ax2 = HX(1) * HY(2) - HY(1) * HX(2);
    const double ay2 = HX(2) * HY(0) - HY(2) * HX(0);
    const double az2 = HX(0) * HY(1) - HY(0) * HX(1);
#undef HX
#undef HY
#undef HZ
    const double cx = g[0], cy = g[1], cz = g[2];
    const double gn = NG(i, j, k, ngcomp);
    // pow(x,4.0) == (x*x)*(x*x) (quirk Q12)
    double kg = (cx * (ax0 * cx + ax1 * cy + ax2 * cz) + cy * (ay0 * cx + ay1 * cy + ay2 * cz) + cz * (az0 * cx + az1 * cy + az2 * cz)) /
                ((gn * gn) * (gn * gn));
    if (thr >= 0.0) {
      const double p = C(i, j, k, ccomp);
      if (p < thr || p > 1.0 - thr) kg = 0.0;
    }
    K(i, j, k, kcomp) = kg;
  }
}

// grad u (9 components, row = velocity component) and the "strain rate" the reference stores:
// its first assignment (-nn:grad u) is overwritten by div u (curvature.cpp:736-747, quirk Q3)
__global__ __launch_bounds__(256) void k_strain(LevelBP2 bp, int ucomp, int srcomp, int rostcomp) {
  FabView U, O;
  DBox V;
  double dxinv[3];
  if (!bp.get(blockIdx.y, U, O, V, dxinv)) return;
  int i, j, k0, k1;
  if (!tile_cell(V, i, j, k0, k1)) return;
  for (int k = k0; k <= k1; ++k) {
    double gu[9];
    for (int d = 0; d < 3; ++d) {
      const double c0 = U(i, j, k, ucomp + d);
      gu[3 * d + 0] = cdiff(dxinv[0], U(i - 1, j, k, ucomp + d), c0, U(i + 1, j, k, ucomp + d));
      gu[3 * d + 1] = cdiff(dxinv[1], U(i, j - 1, k, ucomp + d), c0, U(i, j + 1, k, ucomp + d));
      gu[3 * d + 2] = cdiff(dxinv[2], U(i, j, k - 1, ucomp + d), c0, U(i, j, k + 1, ucomp + d));
    }
    O(i, j, k, srcomp) = +gu[0] + gu[4] + gu[8];
    if (rostcomp >= 0)
      for (int q = 0; q < 9; ++q) O(i, j, k, rostcomp + q) = gu[q];
  }
}

// u . n with the (already thresholded) flame normal, clipped like the curvature (curvature.cpp:776-785)
__global__ __launch_bounds__(256) void k_velnormal(BP3 bp, int ucomp, int ncomp0, int ocomp, DMFView MC, int ccomp, double thr) {
  FabView U, O, N;
  DBox V;
  double dxinv[3];
  if (!bp.get(blockIdx.y, U, O, N, V, dxinv)) return;
  int i, j, k0, k1;
  if (!tile_cell(V, i, j, k0, k1)) return;
  const FabView C = mf_view(MC, V, blockIdx.y);
  for (int k = k0; k <= k1; ++k) {
    double v = +U(i, j, k, ucomp) * N(i, j, k, ncomp0) + U(i, j, k, ucomp + 1) * N(i, j, k, ncomp0 + 1) + U(i, j, k, ucomp + 2) * N(i, j, k, ncomp0 + 2);
    if (thr >= 0.0) {
      const double p = C(i, j, k, ccomp);
      if (p < thr || p > 1.0 - thr) v = 0.0;
    }
    O(i, j, k, ocomp) = v;
  }
}

// The three options in ONE z-marching pass (pa_curvature_run's fast path): per cell the Hessian of c from G (ghost cells
// resolved by the caller), grad u from the velocity (likewise), u . n from the stored normal; z-neighbours of the six
// differentiated components ride in registers (one new plane per step), x / y neighbours come from the cache.  Same
// operations in the same order as k_gauss_curv / k_strain / k_velnormal above (normgrad is formed again from G's centre value
// with the expression of pa_normal_level: its sign drops out of normgrad^4).  c for the threshold = the Progress component of out.
// x-neighbours from the neighbouring LANES (the centre values of a plane are in registers anyway): DPP wavefront shifts instead
// of two more loads per component -- the three stencil kernels above run at 3-4 TB/s of real traffic, bound by their 21 load
// instructions per cell, not by HBM.  Only the first / last lane of a row still loads (a neighbouring tile's or a ghost cell).
struct OptArgs {
  DLevelView L;
  DMFView G, U, O;
  int ucomp, pc, nc, kgc, src, vnc, rostc;
  double thr;
};
// Tiles: TY wavefronts per workgroup, one row of 64 cells each (boxes at most 32 wide: two rows of 32), marching kz planes; the
// tiles of one (x, z) column are consecutive workgroups of ONE XCD (workgroup i runs on XCD i mod 8).  Default 4 rows x 64 planes.
template <bool GAUSS, bool STRAIN, bool VELN, int TY, bool YQ = true>
__global__ __launch_bounds__(64 * TY, TY >= 8 ? 4 : 1) void k_curvopts(OptArgs A, int kz) {
  const int b = blockIdx.y;
  if (b >= A.L.nboxes) return;
  const DBox V = A.L.boxes[b];
  int i, j, k0, k1;
  bool ledge, redge;  // this lane's x-neighbour is not held by the lane next to it
  {
    const int nx = V.hi[0] - V.lo[0] + 1, ny = V.hi[1] - V.lo[1] + 1, nz = V.hi[2] - V.lo[2] + 1;
    const bool narrow = nx <= 32;
    const int TX = narrow ? 32 : 64, rows = narrow ? 2 * TY : TY;
    const int tx = (nx + TX - 1) / TX, ty = (ny + rows - 1) / rows, tz = (nz + kz - 1) / kz;
    const unsigned q = blockIdx.x & 7u, m = blockIdx.x >> 3;
    const int col = (int)(q + 8u * (m / (unsigned)ty)), by = (int)(m % (unsigned)ty);
    if (col >= tx * tz) return;
    const int bx = col % tx, bz = col / tx;
    i = V.lo[0] + bx * TX + (narrow ? (threadIdx.x & 31) : (threadIdx.x & 63));
    j = V.lo[1] + by * rows + (narrow ? (threadIdx.x >> 5) : (threadIdx.x >> 6));
    k0 = V.lo[2] + bz * kz;
    k1 = min(k0 + kz - 1, V.hi[2]);
    if (i > V.hi[0] || j > V.hi[1]) return;
    const int lw = (int)threadIdx.x & (TX - 1);
    ledge = lw == 0;
    redge = lw == TX - 1 || i == V.hi[0];
  }
  // (Measured and not kept, profiles/r05_curvopts.txt: rows of 8 / 16 per workgroup, and the y-neighbours handed from wave to wave
  // through LDS with one barrier per plane instead of loaded again -- FETCH_SIZE is 1.5-1.9 x the bytes needed because a row's
  // values of plane k, loaded one step earlier as centre values, have left the L2 by the time the rows next to it want them --
  // both correct, both slower: 18.8 against 17.6 ms per headline pass for the LDS form.)
  const FabView G = mf_view(A.G, V, b), U = mf_view(A.U, V, b), O = mf_view(A.O, V, b);
  const double dx0 = A.L.dxinv[0], dx1 = A.L.dxinv[1], dx2 = A.L.dxinv[2];
  const double thr = A.thr;
  double gm[3] = {0, 0, 0}, gc[3] = {0, 0, 0}, um[3] = {0, 0, 0}, uc[3] = {0, 0, 0};
  // y-neighbours of the CURRENT plane, requested one step earlier together with the centre value of that plane (YQ): the rows next
  // door ask for the same lines as THEIR centre values in the same step, so the request meets them in the cache -- asked for a step
  // later, as the neighbours of the plane being computed, they had left the L2 (FETCH_SIZE 1.5-1.9 x the bytes needed)
  double gyl[3] = {0, 0, 0}, gyr[3] = {0, 0, 0}, uyl[3] = {0, 0, 0}, uyr[3] = {0, 0, 0};
  const long long gps = (long long)G.nx * G.ny, ups = (long long)U.nx * U.ny, ops = (long long)O.nx * O.ny;
  const double* gp = GAUSS ? G.p + G.idx(i, j, k0, 0) : nullptr;
  const double* up = (STRAIN || VELN) ? U.p + U.idx(i, j, k0, A.ucomp) : nullptr;
  double* op = O.p + O.idx(i, j, k0, 0);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (GAUSS) { gm[d] = gp[d * G.sc - gps]; gc[d] = gp[d * G.sc]; }
    if (GAUSS && YQ) { gyl[d] = gp[d * G.sc - G.nx]; gyr[d] = gp[d * G.sc + G.nx]; }
    if (STRAIN) um[d] = up[d * U.sc - ups];
    if (STRAIN && YQ) { uyl[d] = up[d * U.sc - U.nx]; uyr[d] = up[d * U.sc + U.nx]; }
    if (STRAIN || VELN) uc[d] = up[d * U.sc];
  }
  for (int k = k0; k <= k1; ++k) {
    bool clip = false;
    if (thr >= 0.0 && (GAUSS || VELN)) {
      const double p = op[A.pc * O.sc];
      clip = p < thr || p > 1.0 - thr;
    }
    if (GAUSS) {
      double H[3][3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double* q = gp + d * G.sc;
        const double nxt = q[gps];
        double xl = lane_from_left(gc[d]), xr = lane_from_right(gc[d]);
        if (ledge) xl = q[-1];
        if (redge) xr = q[1];
        H[d][0] = cdiff(dx0, xl, gc[d], xr);
        if (YQ) {
          H[d][1] = cdiff(dx1, gyl[d], gc[d], gyr[d]);
          gyl[d] = q[gps - G.nx];
          gyr[d] = q[gps + G.nx];
        } else {
          H[d][1] = cdiff(dx1, q[-G.nx], gc[d], q[G.nx]);
        }
        H[d][2] = cdiff(dx2, gm[d], gc[d], nxt);
        gm[d] = nxt;  // holds plane k + 1 until the swap below
      }
#define HX(n) H[0][n]
#define HY(n) H[1][n]
#define HZ(n) H[2][n]
      const double ax0 = HY(1) * HZ(2) - HZ(1) * HY(2);
      const double ay0 = HY(2) * HZ(0) - HZ(2) * HY(0);
      const double az0 = HY(0) * HZ(1) - HZ(0) * HY(1);
      const double ax1 = HX(2) * HZ(1) - HZ(2) * HX(1);
      const double ay1 = HX(0) * HZ(2) - HZ(0) * HX(2);
      const double az1 = HX(1) * HZ(0) - HZ(1) * HX(0);
      const double ax2 = HX(1) * HY(2) - HY(1) * HX(2);
      const double ay2 = HX(2) * HY(0) - HY(2) * HX(0);
      const double az2 = HX(0) * HY(1) - HY(0) * HX(1);
#undef HX
#undef HY
#undef HZ
      const double cx = gc[0], cy = gc[1], cz = gc[2];
      const double sn = sqrt(cx * cx + cy * cy + cz * cz);
      const double gn = (1e-14 < sn) ? sn : 1e-14;
      double kg = (cx * (ax0 * cx + ax1 * cy + ax2 * cz) + cy * (ay0 * cx + ay1 * cy + ay2 * cz) + cz * (az0 * cx + az1 * cy + az2 * cz)) /
                  ((gn * gn) * (gn * gn));
      if (clip) kg = 0.0;
      op[A.kgc * O.sc] = kg;
#pragma unroll
      for (int d = 0; d < 3; ++d) { const double t = gm[d]; gm[d] = gc[d]; gc[d] = t; }
      gp += gps;
    }
    if (VELN) {
      const double* n = op + A.nc * O.sc;
      double v = +uc[0] * n[0] + uc[1] * n[O.sc] + uc[2] * n[2 * O.sc];
      if (clip) v = 0.0;
      op[A.vnc * O.sc] = v;
    }
    if (STRAIN) {
      double gu[9];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double* q = up + d * U.sc;
        const double nxt = q[ups];
        double xl = lane_from_left(uc[d]), xr = lane_from_right(uc[d]);
        if (ledge) xl = q[-1];
        if (redge) xr = q[1];
        gu[3 * d + 0] = cdiff(dx0, xl, uc[d], xr);
        if (YQ) {
          gu[3 * d + 1] = cdiff(dx1, uyl[d], uc[d], uyr[d]);
          uyl[d] = q[ups - U.nx];
          uyr[d] = q[ups + U.nx];
        } else {
          gu[3 * d + 1] = cdiff(dx1, q[-U.nx], uc[d], q[U.nx]);
        }
        gu[3 * d + 2] = cdiff(dx2, um[d], uc[d], nxt);
        um[d] = uc[d];
        uc[d] = nxt;
      }
      op[A.src * O.sc] = +gu[0] + gu[4] + gu[8];
      if (A.rostc >= 0)
#pragma unroll
        for (int q = 0; q < 9; ++q) op[(A.rostc + q) * O.sc] = gu[q];
      up += ups;
    } else if (VELN) {
      up += ups;
#pragma unroll
      for (int d = 0; d < 3; ++d) uc[d] = (k < k1) ? up[d * U.sc] : 0.0;
    }
    op += ops;
  }
}

// which: bit 0 Gaussian curvature (G: 3 components from 0, >= 1 resolved ghost layer), bit 1 strain (u: >= 1 resolved ghost layer),
// bit 2 normal velocity; out components pc (Progress, read), nc (normal, read), kgc / src / vnc / rostc (written; rostc < 0: no tensor)
int pa_curvopts_level(pa_ctx* ctx, int which, const pa_mf* G, const pa_mf* u, int ucomp, pa_mf* out, int pc, int nc, int kgc, int src, int vnc, int rostc, double thr) {
  if (!ctx || !out || ((which & 1) && !G) || ((which & 6) && !u)) return pa_fail(ctx, "pa_curvopts_level: null argument");
  const pa_level* L = out->lev;
  if (((which & 1) && (G->lev != L || G->ng < 1 || G->ncomp < 3)) || ((which & 6) && u->lev != L) || ((which & 2) && u->ng < 1))
    return pa_fail(ctx, "pa_curvopts_level: G and the velocity need >= 1 ghost layer on the level of out");
  if ((which & 6) && (ucomp < 0 || ucomp + 3 > u->ncomp)) return pa_fail(ctx, "pa_curvopts_level: velocity component range");
  const int hi = std::max(std::max(pc, nc + 2), std::max(std::max((which & 1) ? kgc : 0, (which & 2) ? (rostc >= 0 ? rostc + 8 : src) : 0), (which & 4) ? vnc : 0));
  if (pc < 0 || nc < 0 || hi >= out->ncomp) return pa_fail(ctx, "pa_curvopts_level: out component range");
  if (L->boxes.empty() || !(which & 7)) return 0;
  OptArgs A{L->view, (which & 1) ? G->view : out->view, (which & 6) ? u->view : out->view, out->view, ucomp, pc, nc, kgc, src, vnc, (which & 2) ? rostc : -1, thr};
  constexpr int TY = 4, kz = 64;  // rows of 64 cells per workgroup x planes per segment; measured: 4 x 64 17.65 ms per headline pass, 8 x 64 18.0 (spills under 128 VGPRs), 16 x 64 25.3
  unsigned gx = 8;
  for (const DBox& B : L->boxes) {
    const int nx = B.hi[0] - B.lo[0] + 1, ny = B.hi[1] - B.lo[1] + 1, nz = B.hi[2] - B.lo[2] + 1;
    const int TX = nx <= 32 ? 32 : 64, rows = nx <= 32 ? 2 * TY : TY;
    const unsigned ncol = (unsigned)(((nx + TX - 1) / TX) * ((nz + kz - 1) / kz)), ty = (unsigned)((ny + rows - 1) / rows);
    gx = std::max(gx, 8u * ((ncol + 7u) / 8u) * ty);
  }
  const dim3 g(gx, (unsigned)L->boxes.size());
  // (the y-neighbours of a plane are requested one step early, with that plane's centre value: profiles/r05_curvopts.txt)
#define PA_OPT(W, T, Y) hipLaunchKernelGGL((k_curvopts<(W & 1) != 0, (W & 2) != 0, (W & 4) != 0, T, Y>), g, dim3(64 * T), 0, ctx->stream, A, kz)
#define PA_OPTW(W) case W: PA_OPT(W, 4, true); break;
  switch (which & 7) {
    PA_OPTW(1) PA_OPTW(2) PA_OPTW(3) PA_OPTW(4) PA_OPTW(5) PA_OPTW(6) PA_OPTW(7)
    default: break;
  }
#undef PA_OPTW
#undef PA_OPT
  PA_HIP(hipGetLastError());
  return 0;
}

int pa_gauss_curv_level(pa_ctx* ctx, const pa_mf* G, int gcomp, const pa_mf* normgrad, int ngcomp, const pa_mf* c, int ccomp, double thr, pa_mf* out,
                        int kcomp) {
  if (!ctx || !G || !normgrad || !out || (thr >= 0.0 && !c)) return pa_fail(ctx, "pa_gauss_curv_level: null argument");
  if (G->ng < 1) return pa_fail(ctx, "pa_gauss_curv_level: G needs >= 1 ghost layer");
  if (G->lev != out->lev || G->lev != normgrad->lev) return pa_fail(ctx, "pa_gauss_curv_level: different levels");
  if (G->lev->boxes.empty()) return 0;  // a rank that owns no box of this level
  BP3 bp{G->lev->view, G->view, out->view, c ? c->view : G->view};
  hipLaunchKernelGGL(k_gauss_curv, tile_grid(G->lev), dim3(256), 0, ctx->stream, bp, gcomp, normgrad->view, ngcomp, ccomp, thr, kcomp);
  PA_HIP(hipGetLastError());
  return 0;
}

int pa_strain_level(pa_ctx* ctx, const pa_mf* u, int ucomp, pa_mf* out, int srcomp, int rostcomp) {
  if (!ctx || !u || !out) return pa_fail(ctx, "pa_strain_level: null argument");
  if (u->ng < 1 || u->lev != out->lev) return pa_fail(ctx, "pa_strain_level: velocity needs >= 1 ghost layer on the same level");
  if (ucomp < 0 || ucomp + 3 > u->ncomp || srcomp >= out->ncomp || (rostcomp >= 0 && rostcomp + 9 > out->ncomp)) return pa_fail(ctx, "pa_strain_level: component range");
  if (u->lev->boxes.empty()) return 0;  // a rank that owns no box of this level
  LevelBP2 bp{u->lev->view, u->view, out->view};
  hipLaunchKernelGGL(k_strain, tile_grid(u->lev), dim3(256), 0, ctx->stream, bp, ucomp, srcomp, rostcomp);
  PA_HIP(hipGetLastError());
  return 0;
}

int pa_velnormal_level(pa_ctx* ctx, const pa_mf* u, int ucomp, const pa_mf* n, int ncomp0, const pa_mf* c, int ccomp, double thr, pa_mf* out, int ocomp) {
  if (!ctx || !u || !n || !out || (thr >= 0.0 && !c)) return pa_fail(ctx, "pa_velnormal_level: null argument");
  if (u->lev != out->lev || n->lev != out->lev) return pa_fail(ctx, "pa_velnormal_level: different levels");
  if (ucomp < 0 || ucomp + 3 > u->ncomp || ncomp0 + 3 > n->ncomp || ocomp >= out->ncomp) return pa_fail(ctx, "pa_velnormal_level: component range");
  if (u->lev->boxes.empty()) return 0;  // a rank that owns no box of this level
  BP3 bp{u->lev->view, u->view, out->view, n->view};
  hipLaunchKernelGGL(k_velnormal, tile_grid(u->lev), dim3(256), 0, ctx->stream, bp, ucomp, ncomp0, ocomp, c ? c->view : u->view, ccomp, thr);
  PA_HIP(hipGetLastError());
  return 0;
}
