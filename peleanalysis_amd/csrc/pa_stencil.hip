// pa_stencil.hip -- gradient / curvature kernels (gfx950).
// Pass-by-pass kernels mirror the reference call sites one to one; the fused
// grad->curvature kernel lives in pa_fused.hip.
#include "pa_internal.h"
#include "pa_fabview.h"
#include "pa_grad_march.h"
#include <algorithm>
#include <cfloat>
#include <cstdlib>

template <typename BP>
static void grad_launch(hipStream_t st, const BP& bp, int nx, int ny, int nz, unsigned nboxes, int comp, int ocomp, const pa_level* L = nullptr) {
  // the k-marching kernels of pa_grad_march.h.  512^3 level of 128^3 boxes: 13 rows x 8/16/32/64 planes 1.037/1.039/1.081/1.088 ms,
  // 8 rows 1.13-1.17 ms (the tiled cell-per-thread kernel it replaced: 1.45 ms; DESIGN_HISTORY.md)
  constexpr int kseg_env = 16, mty_env = 0;
  // boxes wider than 32 cells: a row per wavefront; at most 32: two rows per wavefront (k_grad_marchn).  A level that holds both
  // kinds, or boxes of different sizes, takes a workgroup table per kind (pa_sweep_wgtab): no box is swept by the wrong kernel
  // and no workgroup is launched for a tile its box does not have.
  const bool tables = L != nullptr;
  auto table = [&](int cls, int tw, int mty, int kseg, bool force) -> const WgTab* { return tables ? pa_sweep_wgtab(L, cls, tw, mty, kseg, force) : nullptr; };
  auto wide = [&](int wx, int wy, int wz, int cls, bool force) {
    GradMarchArgs A{comp, ocomp, std::max(1, std::min(kseg_env, wz)), (int)nboxes, 0};
    const int mty = mty_env ? mty_env : (wy >= 52 ? 13 : (wy >= 16 ? 8 : 4));
    auto go = [&](auto tyc) {
      constexpr int M = decltype(tyc)::value;
      A.tiles_max = ((wx + 63) / 64) * ((wy + M - 1) / M) * ((wz + A.kseg - 1) / A.kseg);
      const WgTab* wt = table(cls, 64, M, A.kseg, force);
      if (wt) A.wgtab = wt->d;
      const dim3 g(wt ? wt->n : (unsigned)A.tiles_max * 8u * ((nboxes + 7u) / 8u));
      hipLaunchKernelGGL((k_grad_march<BP, M>), g, dim3(64 * (M + 3)), 0, st, bp, A);
    };
    switch (mty) {
      case 13: go(std::integral_constant<int, 13>{}); break;
      case 8: go(std::integral_constant<int, 8>{}); break;
      case 5: go(std::integral_constant<int, 5>{}); break;
      default: go(std::integral_constant<int, 4>{}); break;
    }
  };
  auto narrow = [&](int wx, int wy, int wz, int cls, bool force) {
    constexpr int NRW = 8;
    GradMarchArgs A{comp, ocomp, std::max(1, std::min(kseg_env, wz)), (int)nboxes, 0};
    A.tiles_max = ((wx + 31) / 32) * ((wy + 2 * NRW - 1) / (2 * NRW)) * ((wz + A.kseg - 1) / A.kseg);
    const WgTab* wt = table(cls, 32, 2 * NRW, A.kseg, force);
    if (wt) A.wgtab = wt->d;
    const dim3 g(wt ? wt->n : (unsigned)A.tiles_max * 8u * ((nboxes + 7u) / 8u));
    hipLaunchKernelGGL((k_grad_marchn<BP, NRW>), g, dim3(64 * (NRW + 2)), 0, st, bp, A);
  };
  if (tables && L->nwide && L->nnarrow) {  // both kinds: two launches, each over its own boxes (the tables carry the
    wide(L->wmax[0], L->wmax[1], L->wmax[2], 0, true);               // box lists; without one a launch covers every box, which either kernel can)
    narrow(L->nmax[0], L->nmax[1], L->nmax[2], 1, true);
    return;
  }
  if (nx > 32) wide(nx, ny, nz, 2, false);
  else narrow(nx, ny, nz, 2, false);
}

extern "C" int pa_grad_level(pa_ctx* ctx, const pa_mf* phi, int comp, pa_mf* out, int ocomp) {
  PaBind bind_(ctx);
  if (!ctx || !phi || !out) return pa_fail(ctx, "pa_grad_level: null argument");
  if (phi->lev != out->lev) return pa_fail(ctx, "pa_grad_level: phi and out live on different levels");
  if (phi->ng < 1) return pa_fail(ctx, "pa_grad_level: phi needs >= 1 ghost layer");
  if (comp < 0 || comp >= phi->ncomp || ocomp < 0 || ocomp + 4 > out->ncomp) return pa_fail(ctx, "pa_grad_level: component range");
  LevelBP2 bp{phi->lev->view, phi->view, out->view};
  const pa_level* L = phi->lev;
  if (phi->lev->boxes.empty()) return 0;  // a rank that owns no box of this level
  ProfScope prof(ctx, PA_TAG_GRAD);
  grad_launch(ctx->stream, bp, L->maxn[0], L->maxn[1], L->maxn[2], (unsigned)L->boxes.size(), comp, ocomp, L);
  PA_HIP(hipGetLastError());
  return 0;
}

// grad.cpp:215-236 for several levels in one launch: 0 = launched, -1 = not applicable (the caller goes level by level), 1 = error.
// Applicable: every level's boxes are wider than 32 cells and at least 52 rows tall (the 13-row tiles).
int pa_grad_levels(pa_ctx* ctx, int nlev, pa_mf* const* phi, int comp, pa_mf* const* out, int ocomp) {
  if (nlev < 2 || pa_opt().force_fallbacks) return -1;
  for (int l = 0; l < nlev; ++l) {  // applicable to every level, or to none (nothing is launched before this is known)
    const pa_level* L = phi[l]->lev;
    if (L->boxes.empty()) continue;
    if (L->nnarrow > 0 || L->maxn[1] < 52) return -1;
    for (const DBox& B : L->boxes)
      if (B.hi[0] - B.lo[0] + 1 <= 32) return -1;
  }
  for (int l0 = 0; l0 < nlev; l0 += PA_MAXB) {  // up to PA_MAXB levels per launch
  GradBatch S;
  S.n = 0;
  S.wg0[0] = 0;
  for (int l = l0; l < nlev && l < l0 + PA_MAXB; ++l) {
    const pa_level* L = phi[l]->lev;
    if (L->boxes.empty()) continue;
    GradMarchArgs A{comp, ocomp, std::max(1, std::min(16, L->maxn[2])), (int)L->boxes.size(), 0};
    A.tiles_max = ((L->maxn[0] + 63) / 64) * ((L->maxn[1] + 12) / 13) * ((L->maxn[2] + A.kseg - 1) / A.kseg);
    const WgTab* wt = pa_sweep_wgtab(L, 2, 64, 13, A.kseg, false);
    if (wt) A.wgtab = wt->d;
    S.bp[S.n] = LevelBP2{L->view, phi[l]->view, out[l]->view};
    S.A[S.n] = A;
    S.wg0[S.n + 1] = S.wg0[S.n] + (wt ? wt->n : (unsigned)A.tiles_max * 8u * (((unsigned)L->boxes.size() + 7u) / 8u));
    ++S.n;
  }
  if (!S.n) continue;
  ProfScope prof(ctx, PA_TAG_GRAD);
  hipLaunchKernelGGL(k_grad_march_levels<13>, dim3(S.wg0[S.n]), dim3(64 * 16), 0, ctx->stream, S);
  PA_HIP(hipGetLastError());
  }
  return 0;
}

extern "C" int pa_grad_fab(pa_ctx* ctx, pa_box valid, const pa_fab* phi, int comp, const double dxinv[3], pa_fab* out, int ocomp) {
  PaBind bind_(ctx);
  if (!ctx || !phi || !out || !dxinv) return pa_fail(ctx, "pa_grad_fab: null argument");
  std::string why;
  if (!fab_covers(*phi, valid, 1, comp, 1, why) || !fab_covers(*out, valid, 0, ocomp, 4, why)) return pa_fail(ctx, "pa_grad_fab: " + why);
  FabBP2 bp{fab_view(*phi), fab_view(*out), to_dbox(valid), {dxinv[0], dxinv[1], dxinv[2]}};
  grad_launch(ctx->stream, bp, valid.hi[0] - valid.lo[0] + 1, valid.hi[1] - valid.lo[1] + 1, valid.hi[2] - valid.lo[2] + 1, 1, comp, ocomp);
  PA_HIP(hipGetLastError());
  return 0;
}

// ======================================================= curvature.cpp:139-149
__global__ __launch_bounds__(256) void k_minmax(DLevelView L, DMFView M, int comp, double* part) {
  const int b = blockIdx.y;
  const DBox B = L.boxes[b];
  const int nx = B.hi[0] - B.lo[0] + 1, ny = B.hi[1] - B.lo[1] + 1, nz = B.hi[2] - B.lo[2] + 1;
  const long long n = (long long)nx * ny * nz;
  double lo = DBL_MAX, hi = -DBL_MAX;
  const double* f = M.data + M.off[b];
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t % nx), j = (int)((t / nx) % ny), k = (int)(t / ((long long)nx * ny));
    const double v = f[fab_index(B, M.ng, M.ncomp, comp, B.lo[0] + i, B.lo[1] + j, B.lo[2] + k)];
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  __shared__ double slo[4], shi[4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { slo[w] = lo; shi[w] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 1; q < 4; ++q) { lo = slo[q] < lo ? slo[q] : lo; hi = shi[q] > hi ? shi[q] : hi; }
    const long long slot = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    part[2 * slot] = lo;
    part[2 * slot + 1] = hi;
  }
}

int pa_ensure_red(pa_ctx* ctx, size_t n);

extern "C" int pa_minmax_level(pa_ctx* ctx, const pa_mf* s, int comp, double* mn, double* mx) {
  PaBind bind_(ctx);
  if (!ctx || !s || !mn || !mx) return pa_fail(ctx, "pa_minmax_level: null argument");
  if (comp < 0 || comp >= s->ncomp) return pa_fail(ctx, "pa_minmax_level: component range");
  const unsigned nb = (unsigned)s->lev->boxes.size();
  const unsigned gx = 32;
  if (nb == 0) { *mn = DBL_MAX; *mx = -DBL_MAX; return 0; }  // a rank that owns no box of this level (the caller reduces over the ranks)
  if (pa_ensure_red(ctx, 2 * (size_t)gx * nb)) return 1;
  hipLaunchKernelGGL(k_minmax, dim3(gx, nb), dim3(256), 0, ctx->stream, s->lev->view, s->view, comp, ctx->d_red);
  PA_HIP(hipGetLastError());
  std::vector<double> h(2 * (size_t)gx * nb);
  PA_HIP(hipMemcpyAsync(h.data(), ctx->d_red, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PA_HIP(hipStreamSynchronize(ctx->stream));
  double lo = DBL_MAX, hi = -DBL_MAX;
  for (size_t q = 0; q < h.size(); q += 2) { lo = std::min(lo, h[q]); hi = std::max(hi, h[q + 1]); }
  *mn = lo;
  *mx = hi;
  return 0;
}

// ======================================================= curvature.cpp:310-321
template <typename BP>
__global__ __launch_bounds__(256) void k_progress(BP bp, int comp, int ccomp, double pmin, double invdenom) {
  FabView S, C;
  DBox V;
  double dxinv[3];
  if (!bp.get(blockIdx.y, S, C, V, dxinv)) return;
  int i, j, k0, k1;
  if (!tile_cell(V, i, j, k0, k1)) return;
  for (int k = k0; k <= k1; ++k) C(i, j, k, ccomp) = (S(i, j, k, comp) - pmin) * invdenom;
}

extern "C" int pa_progress_level(pa_ctx* ctx, const pa_mf* s, int comp, double pmin, double pmax, pa_mf* c, int ccomp, int ng) {
  PaBind bind_(ctx);
  if (!ctx || !s || !c) return pa_fail(ctx, "pa_progress_level: null argument");
  if (s->lev != c->lev) return pa_fail(ctx, "pa_progress_level: different levels");
  if (ng > s->ng || ng > c->ng || comp >= s->ncomp || ccomp >= c->ncomp) return pa_fail(ctx, "pa_progress_level: ng/component range");
  const double invdenom = 1.0 / (pmax - pmin);  // curvature.cpp:315 (quirk Q13: multiply, not divide)
  if (s->lev->boxes.empty()) return 0;  // a rank that owns no box of this level
  LevelBP2 bp{s->lev->view, s->view, c->view, ng};
  ProfScope prof(ctx, PA_TAG_PROGRESS);
  hipLaunchKernelGGL(k_progress<LevelBP2>, tile_grid(s->lev, ng), dim3(256), 0, ctx->stream, bp, comp, ccomp, pmin, invdenom);
  PA_HIP(hipGetLastError());
  return 0;
}

extern "C" int pa_progress_fab(pa_ctx* ctx, pa_box bx, const pa_fab* s, int comp, double pmin, double pmax, pa_fab* c, int ccomp) {
  PaBind bind_(ctx);
  if (!ctx || !s || !c) return pa_fail(ctx, "pa_progress_fab: null argument");
  std::string why;
  if (!fab_covers(*s, bx, 0, comp, 1, why) || !fab_covers(*c, bx, 0, ccomp, 1, why)) return pa_fail(ctx, "pa_progress_fab: " + why);
  const double invdenom = 1.0 / (pmax - pmin);
  FabBP2 bp{fab_view(*s), fab_view(*c), to_dbox(bx), {1, 1, 1}};
  hipLaunchKernelGGL(k_progress<FabBP2>, tile_grid(bx), dim3(256), 0, ctx->stream, bp, comp, ccomp, pmin, invdenom);
  PA_HIP(hipGetLastError());
  return 0;
}

// ======================================================= curvature.cpp:451-502
// G = grad c ; normgrad = -max(1e-14, sqrt(Gx^2+Gy^2+Gz^2)) ; n = G / normgrad
template <typename BP>
__global__ __launch_bounds__(256) void k_normal(BP bp, int comp, int gcomp, int ngcomp, int ncomp0) {
  FabView C, G, NG, N;
  DBox V;
  double dxinv[3];
  if (!bp.get(blockIdx.y, C, G, NG, N, V, dxinv)) return;
  int i, j, k0, k1;
  if (!tile_cell(V, i, j, k0, k1)) return;
  double zm = C(i, j, k0 - 1, comp), zc = C(i, j, k0, comp);
  for (int k = k0; k <= k1; ++k) {
    const double zp = C(i, j, k + 1, comp);
    const double gx = cdiff(dxinv[0], C(i - 1, j, k, comp), zc, C(i + 1, j, k, comp));
    const double gy = cdiff(dxinv[1], C(i, j - 1, k, comp), zc, C(i, j + 1, k, comp));
    const double gz = cdiff(dxinv[2], zm, zc, zp);
    const double sn = sqrt(gx * gx + gy * gy + gz * gz);  // pow(x,2.0) == x*x (quirk Q12)
    const double ng = -((1e-14 < sn) ? sn : 1e-14);
    if (G.p) { G(i, j, k, gcomp) = gx; G(i, j, k, gcomp + 1) = gy; G(i, j, k, gcomp + 2) = gz; }
    if (NG.p) NG(i, j, k, ngcomp) = ng;
    N(i, j, k, ncomp0) = gx / ng;
    N(i, j, k, ncomp0 + 1) = gy / ng;
    N(i, j, k, ncomp0 + 2) = gz / ng;
    zm = zc;
    zc = zp;
  }
}

extern "C" int pa_normal_level(pa_ctx* ctx, const pa_mf* c, int comp, pa_mf* G, int gcomp, pa_mf* normgrad, int ngcomp,
                               pa_mf* n, int ncomp0) {
  PaBind bind_(ctx);
  if (!ctx || !c || !n) return pa_fail(ctx, "pa_normal_level: null argument");
  if (c->ng < 1) return pa_fail(ctx, "pa_normal_level: c needs >= 1 ghost layer");
  if (c->lev != n->lev || (G && G->lev != c->lev) || (normgrad && normgrad->lev != c->lev)) return pa_fail(ctx, "pa_normal_level: different levels");
  if (comp >= c->ncomp || ncomp0 + 3 > n->ncomp || (G && gcomp + 3 > G->ncomp) || (normgrad && ngcomp >= normgrad->ncomp))
    return pa_fail(ctx, "pa_normal_level: component range");
  DMFView none{nullptr, nullptr, 0, 0};
  if (c->lev->boxes.empty()) return 0;  // a rank that owns no box of this level
  LevelBP4 bp{c->lev->view, c->view, G ? G->view : none, normgrad ? normgrad->view : none, n->view};
  hipLaunchKernelGGL(k_normal<LevelBP4>, tile_grid(c->lev), dim3(256), 0, ctx->stream, bp, comp, gcomp, ngcomp, ncomp0);
  PA_HIP(hipGetLastError());
  return 0;
}

extern "C" int pa_normal_fab(pa_ctx* ctx, pa_box valid, const pa_fab* c, int comp, const double dxinv[3], pa_fab* G, int gcomp,
                             pa_fab* normgrad, int ngcomp, pa_fab* n, int ncomp0) {
  PaBind bind_(ctx);
  if (!ctx || !c || !n || !dxinv) return pa_fail(ctx, "pa_normal_fab: null argument");
  std::string why;
  if (!fab_covers(*c, valid, 1, comp, 1, why) || !fab_covers(*n, valid, 0, ncomp0, 3, why) ||
      (G && !fab_covers(*G, valid, 0, gcomp, 3, why)) || (normgrad && !fab_covers(*normgrad, valid, 0, ngcomp, 1, why)))
    return pa_fail(ctx, "pa_normal_fab: " + why);
  FabView none{};
  FabBP4 bp{fab_view(*c), G ? fab_view(*G) : none, normgrad ? fab_view(*normgrad) : none, fab_view(*n), to_dbox(valid),
            {dxinv[0], dxinv[1], dxinv[2]}};
  hipLaunchKernelGGL(k_normal<FabBP4>, tile_grid(valid), dim3(256), 0, ctx->stream, bp, comp, gcomp, ngcomp, ncomp0);
  PA_HIP(hipGetLastError());
  return 0;
}

// ======================================================= curvature.cpp:505-546
// Curv = 0 ; Curv += dn_x/dx ; += dn_y/dy ; += dn_z/dz ; Curv *= scale
template <typename BP>
__global__ __launch_bounds__(256) void k_div(BP bp, int ncomp0, int kcomp, double scale) {
  FabView N, K;
  DBox V;
  double dxinv[3];
  if (!bp.get(blockIdx.y, N, K, V, dxinv)) return;
  int i, j, k0, k1;
  if (!tile_cell(V, i, j, k0, k1)) return;
  for (int k = k0; k <= k1; ++k) {
    double curv = 0.0;
    curv += cdiff(dxinv[0], N(i - 1, j, k, ncomp0), N(i, j, k, ncomp0), N(i + 1, j, k, ncomp0));
    curv += cdiff(dxinv[1], N(i, j - 1, k, ncomp0 + 1), N(i, j, k, ncomp0 + 1), N(i, j + 1, k, ncomp0 + 1));
    curv += cdiff(dxinv[2], N(i, j, k - 1, ncomp0 + 2), N(i, j, k, ncomp0 + 2), N(i, j, k + 1, ncomp0 + 2));
    K(i, j, k, kcomp) = curv * scale;
  }
}

// curvature.cpp:549-567
__global__ __launch_bounds__(256) void k_threshold(LevelBP4 bp, int ccomp, double thr, int kcomp, int ncomp0) {
  FabView C, K, N, unused;
  DBox V;
  double dxinv[3];
  if (!bp.get(blockIdx.y, C, K, N, unused, V, dxinv)) return;
  int i, j, k0, k1;
  if (!tile_cell(V, i, j, k0, k1)) return;
  for (int k = k0; k <= k1; ++k) {
    const double p = C(i, j, k, ccomp);
    if (p < thr || p > 1.0 - thr) {
      K(i, j, k, kcomp) = 0.0;
      N(i, j, k, ncomp0) = 0.0;
      N(i, j, k, ncomp0 + 1) = 0.0;
      N(i, j, k, ncomp0 + 2) = 0.0;
    }
  }
}

extern "C" int pa_div_level(pa_ctx* ctx, pa_mf* n, int ncomp0, double scale, const pa_mf* c, int ccomp, double thr, pa_mf* K, int kcomp) {
  PaBind bind_(ctx);
  if (!ctx || !n || !K) return pa_fail(ctx, "pa_div_level: null argument");
  if (n->ng < 1) return pa_fail(ctx, "pa_div_level: n needs >= 1 ghost layer");
  if (n->lev != K->lev || (c && c->lev != n->lev)) return pa_fail(ctx, "pa_div_level: different levels");
  if (ncomp0 + 3 > n->ncomp || kcomp >= K->ncomp) return pa_fail(ctx, "pa_div_level: component range");
  if (n->lev->boxes.empty()) return 0;  // a rank that owns no box of this level
  LevelBP2 bp{n->lev->view, n->view, K->view};
  hipLaunchKernelGGL(k_div<LevelBP2>, tile_grid(n->lev), dim3(256), 0, ctx->stream, bp, ncomp0, kcomp, scale);
  PA_HIP(hipGetLastError());
  if (thr >= 0.0) {
    if (!c) return pa_fail(ctx, "pa_div_level: threshold needs the progress variable");
    LevelBP4 bp4{n->lev->view, c->view, K->view, n->view, n->view};
    hipLaunchKernelGGL(k_threshold, tile_grid(n->lev), dim3(256), 0, ctx->stream, bp4, ccomp, thr, kcomp, ncomp0);
    PA_HIP(hipGetLastError());
  }
  return 0;
}

extern "C" int pa_div_fab(pa_ctx* ctx, pa_box valid, const pa_fab* n, int ncomp0, const double dxinv[3], double scale, pa_fab* K, int kcomp) {
  PaBind bind_(ctx);
  if (!ctx || !n || !K || !dxinv) return pa_fail(ctx, "pa_div_fab: null argument");
  std::string why;
  if (!fab_covers(*n, valid, 1, ncomp0, 3, why) || !fab_covers(*K, valid, 0, kcomp, 1, why)) return pa_fail(ctx, "pa_div_fab: " + why);
  FabBP2 bp{fab_view(*n), fab_view(*K), to_dbox(valid), {dxinv[0], dxinv[1], dxinv[2]}};
  hipLaunchKernelGGL(k_div<FabBP2>, tile_grid(valid), dim3(256), 0, ctx->stream, bp, ncomp0, kcomp, scale);
  PA_HIP(hipGetLastError());
  return 0;
}
