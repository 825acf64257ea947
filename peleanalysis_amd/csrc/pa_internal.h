// pa_internal.h -- shared host/device definitions of libpeleanalysis_amd (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/peleanalysis_amd.h"

struct DBox { int lo[3]; int hi[3]; };

// Device view of one AMR level: the BoxArray plus an "owner map" -- a coarse
// 3-D table at granularity g (gcd of all box origins/extents) that answers
// "which box owns cell (i,j,k)" with one load.  Replaces AMReX's BoxArray hash.
struct DLevelView {
  int nboxes;
  const DBox* boxes;
  int domlo[3], domhi[3], is_per[3];
  int g, mlo[3], mn[3];
  const int* owner;
  double dxinv[3];
};

struct DMFView {
  double* data;
  const long long* off;  // per box, in doubles
  int ncomp, ng;
  // optional affine view used when this multifab is read as COARSE data: value = (v - xa) * xb.
  // Lets applyBC on the progress variable interpolate from the coarse phi without a stored coarse c.
  int xform;
  double xa, xb;
};

struct pa_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  double* d_red = nullptr;  // reduction scratch
  size_t red_cap = 0;
  int* d_flags = nullptr;   // [0] = coarse-fine ghost cells whose coarse data was missing
  void* d_scr = nullptr;    // grow-only scratch (marching cubes)
  size_t scr_cap = 0;
  // optional per-launch timing of tagged kernels with HIP events on ctx->stream (bench.py roofline)
  bool profile = false;
  struct Ev { hipEvent_t a, b; int tag; };
  std::vector<Ev> evs;
};

// RAII: records an event pair around a tagged kernel launch when profiling is enabled
struct ProfScope {
  pa_ctx* ctx;
  pa_ctx::Ev e;
  bool on;
  ProfScope(pa_ctx* c, int tag) : ctx(c), on(c->profile) {
    if (!on) return;
    e.tag = tag;
    if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(e.a, ctx->stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e.b, ctx->stream);
    ctx->evs.push_back(e);
  }
};
enum { PA_TAG_GRADCURV = 1, PA_TAG_GRADCURV_FACES = 2, PA_TAG_FILL = 3, PA_TAG_BC = 4, PA_TAG_GRAD = 5, PA_TAG_PROGRESS = 6,
       PA_TAG_FILTER = 7, PA_TAG_MC = 8 };

struct pa_level {
  pa_ctx* ctx = nullptr;
  std::vector<DBox> boxes;
  int domlo[3], domhi[3], is_per[3];
  double prob_lo[3], prob_hi[3], dx[3], dxinv[3];
  int g = 1, mlo[3], mn[3];
  std::vector<int> owner;  // host copy
  DBox* d_boxes = nullptr;
  int* d_owner = nullptr;
  int maxn[3] = {0, 0, 0};  // max box extent per dim
  long long ncells = 0;
  bool fusable = true;      // no concave coarse-fine corner (see pa_level_create)
  DLevelView view;
};

struct pa_mf {
  const pa_level* lev = nullptr;
  int ncomp = 0, ng = 0;
  double* data = nullptr;
  bool owned = false;
  std::vector<long long> off;
  long long* d_off = nullptr;
  long long total = 0;
  DMFView view;
};

#define PA_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

int pa_fail(pa_ctx* ctx, const std::string& msg);

// ---------------------------------------------------------------- device helpers
// Component stride of a FAB inside a pa_mf (in doubles): the cell count rounded up to 512 B and
// kept off multiples of 16 KiB.  With the plain AMReX stride (nx*ny*nz) a 128^3 box puts all
// components of one cell 16 MiB apart = on the same HBM channel, which costs ~20 % of the write
// bandwidth of an 8-output kernel (tools/bench/membench2.hip); amrex::Array4 carries an explicit
// nstride too, so this stays within the reference's data model.
__host__ __device__ __forceinline__ long long pa_cstride(long long ncells, int ncomp) {
  long long cs = (ncells + 63) / 64 * 64;
  if (ncomp > 1 && (cs % 2048) == 0) cs += 64;
  return cs;
}

__device__ __forceinline__ long long fab_index(const DBox& B, int ng, int ncomp, int c, int i, int j, int k) {
  const long long nx = B.hi[0] - B.lo[0] + 1 + 2 * ng, ny = B.hi[1] - B.lo[1] + 1 + 2 * ng,
                  nz = B.hi[2] - B.lo[2] + 1 + 2 * ng;
  return (long long)c * pa_cstride(nx * ny * nz, ncomp) + ((long long)(k - B.lo[2] + ng) * ny + (j - B.lo[1] + ng)) * nx + (i - B.lo[0] + ng);
}

// wrap into the domain along periodic directions; false if outside a wall
__device__ __forceinline__ bool wrap_cell(const DLevelView& L, int p[3]) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int len = L.domhi[d] - L.domlo[d] + 1;
    if (p[d] < L.domlo[d] || p[d] > L.domhi[d]) {
      if (!L.is_per[d]) return false;
      while (p[d] < L.domlo[d]) p[d] += len;
      while (p[d] > L.domhi[d]) p[d] -= len;
    }
  }
  return true;
}

__device__ __forceinline__ int owner_of(const DLevelView& L, const int p[3]) {
  int m[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int r = p[d] - L.mlo[d];
    if (r < 0) return -1;
    m[d] = r / L.g;
    if (m[d] >= L.mn[d]) return -1;
  }
  return L.owner[((long long)m[2] * L.mn[1] + m[1]) * L.mn[0] + m[0]];
}

// 0 covered (valid cell, maybe through a periodic image; box id + wrapped cell
// returned), 1 not covered (inside domain) = coarse-fine, 2 outside a wall
__device__ __forceinline__ int classify(const DLevelView& L, int i, int j, int k, int& box, int p[3]) {
  p[0] = i; p[1] = j; p[2] = k;
  if (!wrap_cell(L, p)) return 2;
  box = owner_of(L, p);  // >= 0: box of this rank; -2: valid cell of a box owned by another rank; -1: none
  return (box >= 0 || box == -2) ? 0 : 1;
}
__device__ __forceinline__ int classify(const DLevelView& L, int i, int j, int k) {
  int b, p[3];
  return classify(L, i, j, k, b, p);
}

__device__ __forceinline__ int coarsen_idx(int i, int r) { return (i < 0) ? -((-i + r - 1) / r) : i / r; }

// amrex::poly_interp_coeff restated (Lagrange weights evaluated in fp64)
__device__ __forceinline__ void poly_interp_coeff(double xInt, const double* x, int N, double* c) {
  for (int j = 0; j < N; ++j) {
    double num = 1.0, den = 1.0;
    for (int i = 0; i < N; ++i) {
      if (i == j) continue;
      num *= xInt - x[i];
      den *= x[j] - x[i];
    }
    c[j] = num / den;
  }
}

// central difference in the reference's operation order:
// flux = dxinv*(a-b) [mlpoisson_flux], *(-1) [1/bscalar], cc = 0.5*(f_lo+f_hi)
// [average_face_to_cellcenter], *(-1) [mult(-1)].  Keeps roundings and zero signs.
__device__ __forceinline__ double cdiff(double dxinv, double m, double c, double p) {
  const double fl = -(dxinv * (c - m));
  const double fh = -(dxinv * (p - c));
  return -(0.5 * (fl + fh));
}

// coarse value with periodic wrap; ok cleared if the cell has no owner
__device__ __forceinline__ double crse_val(const DLevelView& LC, const DMFView& MC, int comp, int ic, int jc,
                                           int kc, bool& ok) {
  int p[3] = {ic, jc, kc};
  if (!wrap_cell(LC, p)) { ok = false; return 0.0; }
  const int b = owner_of(LC, p);
  if (b < 0) { ok = false; return 0.0; }
  const double v = MC.data[MC.off[b] + fab_index(LC.boxes[b], MC.ng, MC.ncomp, comp, p[0], p[1], p[2])];
  return MC.xform ? (v - MC.xa) * MC.xb : v;
}

// InterpBndryData (order 3) restated -- see oracle/pa_oracle.c cf_bndry_value
__device__ inline double cf_bndry_value(const DLevelView& LF, const DLevelView& LC, const DMFView& MC, int ccomp,
                                        const int q[3], int dir, int r, bool& ok) {
  const int qc[3] = {coarsen_idx(q[0], r), coarsen_idx(q[1], r), coarsen_idx(q[2], r)};
  const int tdir[2] = {dir == 0 ? 1 : 0, dir == 2 ? 1 : 2};
  double b = 0.0;
  double xi[2];
  for (int t = 0; t < 2; ++t) {
    const int td = tdir[t];
    int m1[3] = {q[0], q[1], q[2]}, p1[3] = {q[0], q[1], q[2]}, m2[3] = {q[0], q[1], q[2]}, p2[3] = {q[0], q[1], q[2]};
    m1[td] -= r; p1[td] += r; m2[td] -= 2 * r; p2[td] += 2 * r;
    const bool okm1 = classify(LF, m1[0], m1[1], m1[2]) == 1;
    const bool okp1 = classify(LF, p1[0], p1[1], p1[2]) == 1;
    int lo = okm1 ? -1 : 0, hi = okp1 ? 1 : 0;
    if (lo == -1 && hi == 0 && classify(LF, m2[0], m2[1], m2[2]) == 1) lo = -2;
    else if (hi == 1 && lo == 0 && classify(LF, p2[0], p2[1], p2[2]) == 1) hi = 2;
    const int N = hi - lo + 1;
    double x[3], c[3];
    for (int m = 0; m < N; ++m) x[m] = (double)(lo + m);
    const double xInt = -0.5 + ((double)(q[td] - qc[td] * r) + 0.5) / (double)r;
    xi[t] = xInt;
    poly_interp_coeff(xInt, x, N, c);
    for (int m = 0; m < N; ++m) {
      int cc[3] = {qc[0], qc[1], qc[2]};
      cc[td] += lo + m;
      b += c[m] * crse_val(LC, MC, ccomp, cc[0], cc[1], cc[2], ok);
    }
  }
  b -= crse_val(LC, MC, ccomp, qc[0], qc[1], qc[2], ok);
  {
    const int t0 = tdir[0], t1 = tdir[1];
    bool all = true;
    for (int s1 = -1; s1 <= 1 && all; s1 += 2)
      for (int s0 = -1; s0 <= 1; s0 += 2) {
        int p[3] = {q[0], q[1], q[2]};
        p[t0] += s0 * r; p[t1] += s1 * r;
        if (classify(LF, p[0], p[1], p[2]) != 1) { all = false; break; }
      }
    if (all) {
      int cpp[3] = {qc[0], qc[1], qc[2]}, cmp[3] = {qc[0], qc[1], qc[2]}, cmm[3] = {qc[0], qc[1], qc[2]},
          cpm[3] = {qc[0], qc[1], qc[2]};
      cpp[t0] += 1; cpp[t1] += 1;
      cmp[t0] -= 1; cmp[t1] += 1;
      cmm[t0] -= 1; cmm[t1] -= 1;
      cpm[t0] += 1; cpm[t1] -= 1;
      const double vpp = crse_val(LC, MC, ccomp, cpp[0], cpp[1], cpp[2], ok);
      const double vmp = crse_val(LC, MC, ccomp, cmp[0], cmp[1], cmp[2], ok);
      const double vmm = crse_val(LC, MC, ccomp, cmm[0], cmm[1], cmm[2], ok);
      const double vpm = crse_val(LC, MC, ccomp, cpm[0], cpm[1], cpm[2], ok);
      b += ((xi[0] * xi[1]) * 0.25) * (((vpp - vmp) + vmm) - vpm);
    }
  }
  return b;
}

// normal-direction Lagrange weights of MLMG applyBC at a coarse-fine face:
// points {-ratio/2 (bc), 0.5, 1.5, 2.5}, evaluated at -0.5, NX = min(len+1, 4)
__device__ __forceinline__ int cf_normal_coef(int blen, int ratio, double coef[4]) {
  const int NX = (blen + 1 < 4) ? blen + 1 : 4;
  const double x[4] = {-0.5 * (double)ratio, 0.5, 1.5, 2.5};
  poly_interp_coeff(-0.5, x, NX, coef);
  return NX;
}
