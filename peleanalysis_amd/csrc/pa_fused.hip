// pa_fused.hip -- fused grad -> curvature kernels (headline path), gfx950.
//
// One sweep reads phi (resolved ring-1 face ghosts) and the progress variable c (ring-2
// same-level ghosts + resolved ring-1 faces/edges) and writes gx,gy,gz,|g|,Nx,Ny,Nz,K.
// The flame normal at the six neighbours is recomputed from c instead of being stored,
// ghost-exchanged and re-read (curvature.cpp:451-546 does ~10 passes over memory).
// Arithmetic order is that of the reference call sites (see pa_internal.h cdiff).
//
// Cells adjacent to a coarse-fine or physical face need the ghost NORMAL, which the
// reference obtains from applyBC on n itself: k_gradcurv_faces recomputes
// K for those cells only.
#include "pa_internal.h"
#include "pa_fabview.h"
#include "pa_fused_march.h"
#include <cstdlib>

struct Vec3 { double x, y, z; };

// flame normal n = G/normgrad at cell (i,j,k), from c
__device__ __forceinline__ Vec3 normal_at(const FabView& C, int cc, int i, int j, int k, const double dxinv[3]) {
  const double c0 = C(i, j, k, cc);
  const double gx = cdiff(dxinv[0], C(i - 1, j, k, cc), c0, C(i + 1, j, k, cc));
  const double gy = cdiff(dxinv[1], C(i, j - 1, k, cc), c0, C(i, j + 1, k, cc));
  const double gz = cdiff(dxinv[2], C(i, j, k - 1, cc), c0, C(i, j, k + 1, cc));
  const double sn = sqrt(gx * gx + gy * gy + gz * gz);
  const double ng = -((1e-14 < sn) ? sn : 1e-14);
  Vec3 n;
  n.x = gx / ng;
  n.y = gy / ng;
  n.z = gz / ng;
  return n;
}

// ---------------------------------------------------------------- v1: direct loads
template <typename BP>
__global__ __launch_bounds__(256) void k_gradcurv_naive(BP bp, int pcomp, int ccomp, int ocomp, double thr) {
  FabView P, C, O, unused;
  DBox V;
  double dxinv[3];
  if (!bp.get(blockIdx.y, P, C, O, unused, V, dxinv)) return;
  int i, j, k0, k1;
  if (!tile_cell(V, i, j, k0, k1)) return;
  for (int k = k0; k <= k1; ++k) {
    const double p0 = P(i, j, k, pcomp);
    const double gx = cdiff(dxinv[0], P(i - 1, j, k, pcomp), p0, P(i + 1, j, k, pcomp));
    const double gy = cdiff(dxinv[1], P(i, j - 1, k, pcomp), p0, P(i, j + 1, k, pcomp));
    const double gz = cdiff(dxinv[2], P(i, j, k - 1, pcomp), p0, P(i, j, k + 1, pcomp));
    O(i, j, k, ocomp) = gx;
    O(i, j, k, ocomp + 1) = gy;
    O(i, j, k, ocomp + 2) = gz;
    O(i, j, k, ocomp + 3) = sqrt(gx * gx + gy * gy + gz * gz);
    const Vec3 n0 = normal_at(C, ccomp, i, j, k, dxinv);
    double curv = 0.0;
    curv += cdiff(dxinv[0], normal_at(C, ccomp, i - 1, j, k, dxinv).x, n0.x, normal_at(C, ccomp, i + 1, j, k, dxinv).x);
    curv += cdiff(dxinv[1], normal_at(C, ccomp, i, j - 1, k, dxinv).y, n0.y, normal_at(C, ccomp, i, j + 1, k, dxinv).y);
    curv += cdiff(dxinv[2], normal_at(C, ccomp, i, j, k - 1, dxinv).z, n0.z, normal_at(C, ccomp, i, j, k + 1, dxinv).z);
    curv = curv * 0.5;
    Vec3 n = n0;
    if (thr >= 0.0) {
      const double c0 = C(i, j, k, ccomp);
      if (c0 < thr || c0 > 1.0 - thr) { curv = 0.0; n.x = 0.0; n.y = 0.0; n.z = 0.0; }
    }
    O(i, j, k, ocomp + 4) = n.x;
    O(i, j, k, ocomp + 5) = n.y;
    O(i, j, k, ocomp + 6) = n.z;
    O(i, j, k, ocomp + 7) = curv;
  }
}

// ---------------------------------------------------------------- face fix-up
// Thread per (box face cell X).  If the ghost cell beyond the face is not a valid cell of this
// level, K(X) is recomputed with the ghost normal given by MLMG applyBC on n_d
// (curvature.cpp:510-531): wall -> +-n_d(X); coarse-fine -> cubic through the coarse
// boundary value (InterpBndryData of the coarse, already thresholded, normal: quirk Q2) and
// n_d at X, X-+1, X-+2.  All other normals are recomputed from c.
struct FaceArgs {
  int bc[3];
  int ratio;
  int has_crse;
  double thr;
};

__device__ __forceinline__ double comp_of(const Vec3& v, int d) { return d == 0 ? v.x : (d == 1 ? v.y : v.z); }

__global__ __launch_bounds__(256) void k_gradcurv_faces(DLevelView L, DMFView MC_, int ccomp, DLevelView LCr, DMFView MN, int cncomp0,
                                                        DMFView MO, int kcomp, FaceArgs A, int* nbad) {
  const int b = blockIdx.y;
  if (b >= L.nboxes) return;
  const DBox B = L.boxes[b];
  const int n[3] = {B.hi[0] - B.lo[0] + 1, B.hi[1] - B.lo[1] + 1, B.hi[2] - B.lo[2] + 1};
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  int fdir = -1, side = 0, a0 = 0, b1 = 0;
  for (int d = 0; d < 3; ++d) {
    const int t0 = (d == 0) ? 1 : 0, t1 = (d == 2) ? 1 : 2;
    const long long fs = (long long)n[t0] * n[t1];
    if (t < 2 * fs) {
      fdir = d;
      side = t >= fs;
      if (side) t -= fs;
      a0 = (int)(t % n[t0]);
      b1 = (int)(t / n[t0]);
      break;
    }
    t -= 2 * fs;
  }
  if (fdir < 0) return;
  int X[3];
  {
    const int t0 = (fdir == 0) ? 1 : 0, t1 = (fdir == 2) ? 1 : 2;
    X[fdir] = side ? B.hi[fdir] : B.lo[fdir];
    X[t0] = B.lo[t0] + a0;
    X[t1] = B.lo[t1] + b1;
  }
  {
    int q[3] = {X[0], X[1], X[2]};
    q[fdir] += side ? 1 : -1;
    if (classify(L, q[0], q[1], q[2]) == 0) return;  // ordinary same-level face: the fused kernel was exact
  }
  const FabView C = mf_view(MC_, B, b);
  const double dxinv[3] = {L.dxinv[0], L.dxinv[1], L.dxinv[2]};
  const Vec3 n0 = normal_at(C, ccomp, X[0], X[1], X[2], dxinv);
  double curv = 0.0;
  bool ok = true;
  for (int d = 0; d < 3; ++d) {
    double nb[2];
    for (int s2 = 0; s2 < 2; ++s2) {
      const int sg = s2 ? 1 : -1;
      int q[3] = {X[0], X[1], X[2]};
      q[d] += sg;
      const int cls = classify(L, q[0], q[1], q[2]);
      if (cls == 0) {
        nb[s2] = comp_of(normal_at(C, ccomp, q[0], q[1], q[2], dxinv), d);
      } else if (cls == 2) {
        const double v = comp_of(n0, d);
        nb[s2] = (A.bc[d] == PA_BC_REFLECT_ODD) ? -v : v;
      } else {
        if (!A.has_crse) { ok = false; nb[s2] = 0.0; continue; }
        double coef[4];
        const int NX = cf_normal_coef(n[d], A.ratio, coef);
        const double bv = cf_bndry_value(L, LCr, MN, cncomp0 + d, q, d, A.ratio, ok);
        double tmp = 0.0;
        for (int m = 1; m < NX; ++m) {
          int pc[3] = {q[0], q[1], q[2]};
          pc[d] -= sg * m;  // into the box
          const double v = (m == 1) ? comp_of(n0, d) : comp_of(normal_at(C, ccomp, pc[0], pc[1], pc[2], dxinv), d);
          tmp += v * coef[m];
        }
        double g = tmp;
        g += bv * coef[0];
        nb[s2] = g;
      }
    }
    curv += cdiff(dxinv[d], nb[0], comp_of(n0, d), nb[1]);
  }
  curv = curv * 0.5;
  if (A.thr >= 0.0) {
    const double c0 = C(X[0], X[1], X[2], ccomp);
    if (c0 < A.thr || c0 > 1.0 - A.thr) curv = 0.0;
  }
  if (!ok) atomicAdd(nbad, 1);
  MO.data[MO.off[b] + fab_index(B, MO.ng, MO.ncomp, kcomp, X[0], X[1], X[2])] = curv;
}

int pa_gradcurv_launch(pa_ctx* ctx, const pa_mf* phi, int pcomp, const pa_mf* c, int ccomp, double thr, pa_mf* out, int ocomp);

extern "C" int pa_gradcurv_level(pa_ctx* ctx, const pa_mf* phi, int pcomp, const pa_mf* c, int ccomp, double thr, pa_mf* out, int ocomp) {
  if (!ctx || !phi || !c || !out) return pa_fail(ctx, "pa_gradcurv_level: null argument");
  if (phi->lev != c->lev || phi->lev != out->lev) return pa_fail(ctx, "pa_gradcurv_level: different levels");
  if (phi->ng < 1 || c->ng < 2) return pa_fail(ctx, "pa_gradcurv_level: phi needs >= 1 and c >= 2 ghost layers");
  if (pcomp < 0 || pcomp >= phi->ncomp || ccomp < 0 || ccomp >= c->ncomp || ocomp < 0 || ocomp + 8 > out->ncomp)
    return pa_fail(ctx, "pa_gradcurv_level: component range");
  return pa_gradcurv_launch(ctx, phi, pcomp, c, ccomp, thr, out, ocomp);
}

extern "C" int pa_gradcurv_faces_level(pa_ctx* ctx, const pa_mf* c, int ccomp, const pa_mf* crse_n, int cncomp0, const int32_t bc[3],
                                       int ratio, double thr, pa_mf* out, int kcomp) {
  if (!ctx || !c || !out) return pa_fail(ctx, "pa_gradcurv_faces_level: null argument");
  if (c->lev != out->lev) return pa_fail(ctx, "pa_gradcurv_faces_level: different levels");
  if (c->ng < 2) return pa_fail(ctx, "pa_gradcurv_faces_level: c needs >= 2 ghost layers");
  if (ccomp >= c->ncomp || kcomp >= out->ncomp || (crse_n && cncomp0 + 3 > crse_n->ncomp)) return pa_fail(ctx, "pa_gradcurv_faces_level: component range");
  if (crse_n && ratio != 2) return pa_fail(ctx, "pa_gradcurv_faces_level: only refinement ratio 2 is supported");
  const pa_level* L = c->lev;
  for (const DBox& B : L->boxes)
    for (int d = 0; d < 3; ++d)
      if (crse_n && B.hi[d] - B.lo[d] + 1 < 3) return pa_fail(ctx, "pa_gradcurv_faces_level: boxes thinner than 3 cells need the pass-by-pass path");
  FaceArgs A;
  for (int d = 0; d < 3; ++d) A.bc[d] = bc[d];
  A.ratio = ratio; A.has_crse = crse_n ? 1 : 0; A.thr = thr;
  const long long n0 = L->maxn[0], n1 = L->maxn[1], n2 = L->maxn[2];
  const long long nt = 2 * (n1 * n2 + n0 * n2 + n0 * n1);
  dim3 grid((unsigned)((nt + 255) / 256), (unsigned)L->boxes.size());
  ProfScope prof(ctx, PA_TAG_GRADCURV_FACES);
  hipLaunchKernelGGL(k_gradcurv_faces, grid, dim3(256), 0, ctx->stream, L->view, c->view, ccomp, crse_n ? crse_n->lev->view : L->view,
                     crse_n ? crse_n->view : c->view, cncomp0, out->view, kcomp, A, ctx->d_flags);
  PA_HIP(hipGetLastError());
  return 0;
}

// tuning knobs (environment, read once): PA_FUSED_VARIANT=naive|march, PA_KSEG=<planes per workgroup>
static int fused_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PA_FUSED_VARIANT");
    v = (e && std::string(e) == "naive") ? 0 : 1;
  }
  return v;
}
static int fused_kseg() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PA_KSEG");
    v = e ? atoi(e) : 64;
    if (v < 4) v = 4;
  }
  return v;
}
static int fused_mty() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PA_MTY");
    v = e ? atoi(e) : 81;  // MTY*10 + min waves per SIMD
  }
  return v;
}
static dim3 march_grid(int nx, int ny, int nz, int kseg, int mty, unsigned nboxes) {
  const unsigned tx = (nx + 63) / 64, ty = (ny + mty - 1) / mty, tz = (nz + kseg - 1) / kseg;
  return dim3(tx * ty * tz, nboxes);
}
template <typename BP>
static void march_launch(hipStream_t st, const BP& bp, int nx, int ny, int nz, unsigned nboxes, int pcomp, int ccomp, int ocomp, double thr) {
  const int kseg = fused_kseg();
  switch (fused_mty()) {
#define PA_CASE(M, W)                                                                                                         \
  case M * 10 + W:                                                                                                            \
    hipLaunchKernelGGL((k_gradcurv_march<BP, M, W>), march_grid(nx, ny, nz, kseg, M, nboxes), dim3(64 * (M + 3)), 0, st, bp, \
                       pcomp, ccomp, ocomp, thr, kseg);                                                                       \
    break;
    PA_CASE(4, 1) PA_CASE(4, 6) PA_CASE(4, 7) PA_CASE(8, 1) PA_CASE(8, 6) PA_CASE(12, 1) PA_CASE(13, 1)
#undef PA_CASE
    default:
      hipLaunchKernelGGL((k_gradcurv_march<BP, 8, 1>), march_grid(nx, ny, nz, kseg, 8, nboxes), dim3(64 * 11), 0, st, bp, pcomp, ccomp,
                         ocomp, thr, kseg);
  }
}

int pa_gradcurv_launch(pa_ctx* ctx, const pa_mf* phi, int pcomp, const pa_mf* c, int ccomp, double thr, pa_mf* out, int ocomp) {
  LevelBP4 bp{phi->lev->view, phi->view, c->view, out->view, out->view};
  ProfScope prof(ctx, PA_TAG_GRADCURV);
  const pa_level* L = phi->lev;
  if (fused_variant() == 0) {
    hipLaunchKernelGGL(k_gradcurv_naive<LevelBP4>, tile_grid(L), dim3(256), 0, ctx->stream, bp, pcomp, ccomp, ocomp, thr);
  } else {
    march_launch(ctx->stream, bp, L->maxn[0], L->maxn[1], L->maxn[2], (unsigned)L->boxes.size(), pcomp, ccomp, ocomp, thr);
  }
  PA_HIP(hipGetLastError());
  return 0;
}

extern "C" int pa_gradcurv_fab(pa_ctx* ctx, pa_box valid, const pa_fab* phi, int pcomp, const pa_fab* c, int ccomp, const double dxinv[3],
                               double thr, pa_fab* out, int ocomp) {
  if (!ctx || !phi || !c || !out || !dxinv) return pa_fail(ctx, "pa_gradcurv_fab: null argument");
  std::string why;
  if (!fab_covers(*phi, valid, 1, pcomp, 1, why) || !fab_covers(*c, valid, 2, ccomp, 1, why) || !fab_covers(*out, valid, 0, ocomp, 8, why))
    return pa_fail(ctx, "pa_gradcurv_fab: " + why);
  FabBP4 bp{fab_view(*phi), fab_view(*c), fab_view(*out), fab_view(*out), to_dbox(valid), {dxinv[0], dxinv[1], dxinv[2]}};
  if (fused_variant() == 0) {
    hipLaunchKernelGGL(k_gradcurv_naive<FabBP4>, tile_grid(valid), dim3(256), 0, ctx->stream, bp, pcomp, ccomp, ocomp, thr);
  } else {
    march_launch(ctx->stream, bp, valid.hi[0] - valid.lo[0] + 1, valid.hi[1] - valid.lo[1] + 1, valid.hi[2] - valid.lo[2] + 1, 1, pcomp,
                 ccomp, ocomp, thr);
  }
  PA_HIP(hipGetLastError());
  return 0;
}
