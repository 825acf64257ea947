// pa_fabview.h -- Array4-like device view of one FAB, the "box providers" that let the same
// kernel run over every box of a level (blockIdx.y = box) or over one caller-supplied FAB,
// and the cell-tile mapping shared by the stencil kernels.
#pragma once
#include "pa_internal.h"

struct FabView {
  double* p = nullptr;
  int lo[3] = {0, 0, 0};  // lower corner incl. ghosts
  int nx = 0, ny = 0;     // extents incl. ghosts
  long long sc = 0;       // component stride
  __host__ __device__ __forceinline__ long long idx(int i, int j, int k, int c) const {
    return (long long)c * sc + ((long long)(k - lo[2]) * ny + (j - lo[1])) * nx + (i - lo[0]);
  }
  __device__ __forceinline__ double& operator()(int i, int j, int k, int c) const { return p[idx(i, j, k, c)]; }
};

__device__ __forceinline__ FabView mf_view(const DMFView& M, const DBox& B, int b) {
  FabView f;
  if (!M.data) return f;
  f.p = M.data + M.off[b];
  const int nz = B.hi[2] - B.lo[2] + 1 + 2 * M.ng;
  f.nx = B.hi[0] - B.lo[0] + 1 + 2 * M.ng;
  f.ny = B.hi[1] - B.lo[1] + 1 + 2 * M.ng;
  f.sc = pa_cstride((long long)f.nx * f.ny * nz, M.ncomp);
  for (int d = 0; d < 3; ++d) f.lo[d] = B.lo[d] - M.ng;
  return f;
}

inline FabView fab_view(const pa_fab& f) {
  FabView v;
  v.p = f.p;
  v.nx = f.hi[0] - f.lo[0] + 1;
  v.ny = f.hi[1] - f.lo[1] + 1;
  v.sc = f.nstride > 0 ? (long long)f.nstride : (long long)v.nx * v.ny * (f.hi[2] - f.lo[2] + 1);
  for (int d = 0; d < 3; ++d) v.lo[d] = f.lo[d];
  return v;
}

inline DBox to_dbox(const pa_box& b) {
  DBox r;
  for (int d = 0; d < 3; ++d) { r.lo[d] = b.lo[d]; r.hi[d] = b.hi[d]; }
  return r;
}

// host-side shape check before a hand-written kernel touches caller memory
inline bool fab_covers(const pa_fab& f, const pa_box& valid, int grow, int comp, int ncomp, std::string& why) {
  if (!f.p) { why = "null fab pointer"; return false; }
  for (int d = 0; d < 3; ++d) {
    if (valid.hi[d] < valid.lo[d]) { why = "empty box"; return false; }
    if (f.lo[d] > valid.lo[d] - grow || f.hi[d] < valid.hi[d] + grow) {
      why = "fab does not cover the box grown by " + std::to_string(grow);
      return false;
    }
  }
  if (comp < 0 || comp + ncomp > f.ncomp) { why = "component range"; return false; }
  return true;
}

// ------------------------------------------------------------------ box providers
struct LevelBP2 {
  DLevelView L;
  DMFView A, B;
  int grow = 0;
  __device__ __forceinline__ bool get(int b, FabView& a, FabView& o, DBox& V, double dxinv[3]) const {
    if (b >= L.nboxes) return false;
    const DBox X = L.boxes[b];
    a = mf_view(A, X, b);
    o = mf_view(B, X, b);
    V = X;
    for (int d = 0; d < 3; ++d) { V.lo[d] -= grow; V.hi[d] += grow; dxinv[d] = L.dxinv[d]; }
    return true;
  }
  // compact resolved-ghost array of face f = dir * 2 + side of box b (DLevelView::cg), null if the face is ordinary
  __device__ __forceinline__ const double* cg_face(int b, int f) const {
    if (!L.cg) return nullptr;
    const int e = L.sfindex[b * 6 + f];
    return e < 0 ? nullptr : L.cg + L.cgoff[e];
  }
  __device__ __forceinline__ const double* cg_base() const { return L.cg; }
};
struct LevelBP4 {
  DLevelView L;
  DMFView A, B, C, D;
  __device__ __forceinline__ bool get(int b, FabView& a, FabView& o, FabView& c, FabView& e, DBox& V, double dxinv[3]) const {
    if (b >= L.nboxes) return false;
    const DBox X = L.boxes[b];
    a = mf_view(A, X, b);
    o = mf_view(B, X, b);
    c = mf_view(C, X, b);
    e = mf_view(D, X, b);
    V = X;
    for (int d = 0; d < 3; ++d) dxinv[d] = L.dxinv[d];
    return true;
  }
};
struct FabBP2 {
  FabView A, B;
  DBox V;
  double dxinv[3];
  __device__ __forceinline__ bool get(int, FabView& a, FabView& o, DBox& v, double dx[3]) const {
    a = A; o = B; v = V;
    for (int d = 0; d < 3; ++d) dx[d] = dxinv[d];
    return true;
  }
  __device__ __forceinline__ const double* cg_face(int, int) const { return nullptr; }
  __device__ __forceinline__ const double* cg_base() const { return nullptr; }
};
struct FabBP4 {
  FabView A, B, C, D;
  DBox V;
  double dxinv[3];
  __device__ __forceinline__ bool get(int, FabView& a, FabView& o, FabView& c, FabView& e, DBox& v, double dx[3]) const {
    a = A; o = B; c = C; e = D; v = V;
    for (int d = 0; d < 3; ++d) dx[d] = dxinv[d];
    return true;
  }
};

// ------------------------------------------------------------------ cell tiles
// 256 threads = 64 (x, one wavefront per row: 512 contiguous bytes) x 4 (y); each thread
// marches PA_TZ cells in z.  blockIdx.x enumerates tiles of the box, blockIdx.y the box.
#define PA_TX 64
#define PA_TY 4
#define PA_TZ 16

// Boxes at most 32 cells wide (AMReX's default max_grid_size): 32 (x) x 8 (y) tiles, two rows per wavefront, so that no
// lane idles and a wave still touches 512 contiguous bytes of a 32-wide FAB row pair (the pass-by-pass kernels ran such
// boxes with half of every wavefront masked off).  The rule is per BOX; the launch grid is sized for the larger count.
__device__ __forceinline__ bool tile_cell(const DBox& V, int& i, int& j, int& k0, int& k1) {
  const int nx = V.hi[0] - V.lo[0] + 1, ny = V.hi[1] - V.lo[1] + 1, nz = V.hi[2] - V.lo[2] + 1;
  const bool narrow = nx <= 32;
  const int TX = narrow ? 32 : PA_TX, TY = narrow ? 8 : PA_TY;
  const int tx = (nx + TX - 1) / TX, ty = (ny + TY - 1) / TY, tz = (nz + PA_TZ - 1) / PA_TZ;
  const unsigned bid = blockIdx.x;
  if (bid >= (unsigned)tx * ty * tz) return false;
  const int bx = bid % tx, by = (bid / tx) % ty, bz = bid / (tx * ty);
  i = V.lo[0] + bx * TX + (narrow ? (threadIdx.x & 31) : (threadIdx.x & 63));
  j = V.lo[1] + by * TY + (narrow ? (threadIdx.x >> 5) : (threadIdx.x >> 6));
  k0 = V.lo[2] + bz * PA_TZ;
  k1 = min(k0 + PA_TZ - 1, V.hi[2]);
  return i <= V.hi[0] && j <= V.hi[1];
}

inline dim3 tile_grid_dims(int nx, int ny, int nz, unsigned nboxes) {
  // the level's widest box decides: a wide tiling (64 x 4) never has fewer tiles than the narrow one (32 x 8) of a narrower box
  const bool narrow = nx <= 32;
  const unsigned tx = narrow ? 1u : (nx + PA_TX - 1) / PA_TX, ty = narrow ? (ny + 7) / 8 : (ny + PA_TY - 1) / PA_TY, tz = (nz + PA_TZ - 1) / PA_TZ;
  return dim3(tx * ty * tz, nboxes);
}
inline dim3 tile_grid(const pa_level* L, int grow = 0) {
  return tile_grid_dims(L->maxn[0] + 2 * grow, L->maxn[1] + 2 * grow, L->maxn[2] + 2 * grow, (unsigned)L->boxes.size());
}
inline dim3 tile_grid(const pa_box& b) {
  return tile_grid_dims(b.hi[0] - b.lo[0] + 1, b.hi[1] - b.lo[1] + 1, b.hi[2] - b.lo[2] + 1, 1);
}
