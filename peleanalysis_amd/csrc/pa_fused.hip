// pa_fused.hip -- fused grad -> curvature (headline path), gfx950.
//
// k_gradcurv_march (pa_fused_march.h) sweeps every box once: reads phi, writes
// gx,gy,gz,|g|,Nx,Ny,Nz,K.  The flame normal at the neighbours is recomputed on chip instead of
// being stored, ghost-exchanged and re-read (curvature.cpp:451-546 makes ~10 passes over memory).
//
// k_gradcurv_faces then recomputes, for the cells within two layers of a coarse-fine or physical
// face of their box, what depends on boundary conditions the sweep cannot know:
//   * the ghost NORMAL beyond such a face comes from MLMG applyBC on n_d itself
//     (curvature.cpp:510-531), not from c;
//   * the ghost PROGRESS VARIABLE there comes from applyBC on c (curvature.cpp:443-457), which is
//     not (phi_ghost - pmin)*invdenom at coarse-fine and reflect_odd faces.
// It works from a resolved copy of c (FillBoundary(2) + applyBC on faces and edge ghosts) that only
// needs to be valid in a shell around the box faces.
#include "pa_internal.h"
#include "pa_dist.h"
#include "pa_fabview.h"
#include "pa_fused_march.h"
#include "pa_fused_march3.h"
#include "pa_fused_march3n.h"
#include <algorithm>
#include <cstdlib>

#ifndef PA_FC_WAVES
#define PA_FC_WAVES 3  /* measured: 1 -> 1.30, 2 -> 1.34, 3 -> 1.21, 4 (spills) -> 2.28 ms of face fix-up per step */
#endif
struct Vec3 { double x, y, z; };

// Where the face fix-up reads the progress variable from.
// ShellAcc: the stored copy with resolved ghost cells (`work`, filled in a shell around the special faces).
struct ShellAcc {
  FabView C;
  int cc;
  __device__ __forceinline__ double operator()(int i, int j, int k) const { return C(i, j, k, cc); }
};
// CgAcc (exact-normal pipeline, below): nothing is stored but the ghost values behind special faces.  A cell of the box
// or a ghost cell that is a valid cell of the level: (phi - pmin) * invdenom; the ghost cell behind a special face: that
// face's compact array; an edge ghost (outside in two directions): the ring of the special one of the two faces.
struct CgAcc {
  const DLevelView* L;
  FabView P;
  DBox B;
  int b, pcomp;
  double pmin, invd;
  const double* cg;  // the component slot's set of compact arrays (L->cg + slot * stride)
  __device__ __forceinline__ double operator()(int i, int j, int k) const {
    const int p[3] = {i, j, k};
    int nout = 0, fd[2] = {0, 0}, fs[2] = {0, 0};
    bool near = true;  // within one cell of the box in every direction
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int o = p[d] < B.lo[d] ? B.lo[d] - p[d] : (p[d] > B.hi[d] ? p[d] - B.hi[d] : 0);
      if (o) {
        if (nout < 2) { fd[nout] = d; fs[nout] = p[d] > B.hi[d]; }
        ++nout;
        near = near && o == 1;
      }
    }
    int e = -1, d = 0;
    if (near && nout == 1) { d = fd[0]; e = L->sfindex[b * 6 + d * 2 + fs[0]]; }
    if (near && nout == 2) {
      const int ea = L->sfindex[b * 6 + fd[0] * 2 + fs[0]], ec = L->sfindex[b * 6 + fd[1] * 2 + fs[1]];
      if (ea >= 0 && ec >= 0) return 0.0;  // needed by nobody (its face-ring neighbours are not valid cells either)
      if (ea >= 0) { e = ea; d = fd[0]; }
      if (ec >= 0) { e = ec; d = fd[1]; }
    }
    if (e < 0) return (P(i, j, k, pcomp) - pmin) * invd;
    const int t0 = (d == 0) ? 1 : 0, t1 = (d == 2) ? 1 : 2;
    return cg[L->cgoff[e] + (long long)(p[t1] - B.lo[t1] + 1) * (B.hi[t0] - B.lo[t0] + 3) + (p[t0] - B.lo[t0] + 1)];
  }
};

// flame normal n = G/normgrad at cell (i,j,k), from c
template <typename Acc>
__device__ __forceinline__ Vec3 normal_at(const Acc& C, int i, int j, int k, const double dxinv[3]) {
  Vec3 n;
  normal_from(C(i - 1, j, k), C(i + 1, j, k), C(i, j - 1, k), C(i, j + 1, k), C(i, j, k - 1), C(i, j, k), C(i, j, k + 1), dxinv, n.x, n.y, n.z);
  return n;
}

struct FaceArgs {
  int bc[3];
  int ratio;
  int has_crse;
  int layers;  // cells per face normal that are recomputed (2)
  double thr;
  int perim_only;  // k_faces_curv: only the cells on the perimeter of each face (k_faces_curv_fast does the interior)
  double pmin, invd;  // CgAcc: progress variable from phi
};

// one level's arguments of the curvature fix-up kernels (several levels per launch: LevBatch)
struct FixArgs {
  DLevelView L;
  DMFView MC_;
  int ccomp;
  DLevelView LCr;
  DMFView MN;
  int cncomp0;
  DMFView MO;
  int ncomp0, kcomp;
  FaceArgs A;
  int use_cp = 0;  // the level's coarse patches hold the coarse normal component of each face's direction (k_cpatch ran)
  long long cg_stride = 0, cp_stride = 0;  // component slots (blockIdx.z): doubles between the slots' sets of compact arrays / coarse patches
  const int2* wg = nullptr; int nwg = 0;   // the level's work table {special face, chunk of 256 face cells} (pa_level::d_sfwg)
  const int2* pwg = nullptr; int npwg = 0; // ... {special face, chunk of 256 perimeter cells} (pa_level::d_pfwg)
  // NCG: this pass's sweep mirrored the first-layer data of the special x faces of boxes at least ncg_minw wide (pa_fused_march.h)
  const double* ncg = nullptr; long long ncgs = 0; int ncg_minw = 0;
};
typedef double pa_fix_d2 __attribute__((ext_vector_type(2)));
// the x faces the wide CG sweep mirrors (the tile that holds the face must hold the first three columns behind it: pa_fused_march3.h ncgl / ncgh)
__device__ __forceinline__ bool ncg_face_ok(const DBox& B, int side, int minw) {
  const int nx = B.hi[0] - B.lo[0] + 1;
  return nx >= minw && (side ? ((nx - 1) & 63) + 1 : min(nx, 64)) >= 3;
}
// workgroup -> (batch level, special face, first face cell) through the levels' work tables
template <typename BT>
__device__ __forceinline__ bool wg_decode(const BT& Bt, int& blev, unsigned& fy, long long& t, unsigned w = blockIdx.x) {
  blev = 0;
  while (blev + 1 < Bt.n && w >= (unsigned)Bt.a[blev].nwg) { w -= (unsigned)Bt.a[blev].nwg; ++blev; }
  if (w >= (unsigned)Bt.a[blev].nwg) return false;
  const int2 c = Bt.a[blev].wg[w];
  fy = (unsigned)c.x;
  t = (long long)c.y * 256 + threadIdx.x;
  return true;
}
// Component slots: the boundary kernels of the exact-normal pipeline run for several components in ONE launch, blockIdx.z =
// slot z: phi component + z, output components + 8 z, coarse-normal components + cn_z z, the slot's own set of compact ghost
// arrays and coarse patches, its own progress-variable range prog[2 z], prog[2 z + 1] = (pmin, 1 / (pmax - pmin)) (null: the
// one in the level arguments).  One launch over 16 components costs little more than over one: at the size of a level's
// special faces these kernels are latency bound (pa_gradcurv_run_comps2).
struct SlotK { const double* prog = nullptr; int cn_z = 8; };

__device__ __forceinline__ double comp_of(const Vec3& v, int d) { return d == 0 ? v.x : (d == 1 ? v.y : v.z); }

__device__ __forceinline__ bool in_box(const DBox& B, const int q[3]) {
  return q[0] >= B.lo[0] && q[0] <= B.hi[0] && q[1] >= B.lo[1] && q[1] <= B.hi[1] && q[2] >= B.lo[2] && q[2] <= B.hi[2];
}

// Phase A: thread per layer-1 cell of a special face whose ghost cell is not a valid cell.  Its
// normal depends on the resolved ghost c (applyBC on c), which the sweep did not have: recompute.
__global__ __launch_bounds__(256) void k_faces_normal(DLevelView L, DMFView MC_, int ccomp, DMFView MO, int ncomp0, FaceArgs A) {
  int b, dir, side, layer, q[3];
  DBox B;
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (!sface_decode(L, blockIdx.y, t, 1, b, B, dir, side, q, layer)) return;
  if ((L.sfcode[L.sfoff[blockIdx.y] + t] & 3u) == 0) return;  // ordinary same-level ghost: the sweep was exact
  int X[3] = {q[0], q[1], q[2]};
  X[dir] += side ? -1 : 1;
  const ShellAcc C = {mf_view(MC_, B, b), ccomp};
  const double dxinv[3] = {L.dxinv[0], L.dxinv[1], L.dxinv[2]};
  Vec3 no = normal_at(C, X[0], X[1], X[2], dxinv);
  if (A.thr >= 0.0) {
    const double c0 = C(X[0], X[1], X[2]);
    if (c0 < A.thr || c0 > 1.0 - A.thr) { no.x = 0.0; no.y = 0.0; no.z = 0.0; }
  }
  double* o = MO.data + MO.off[b];
  o[fab_index(B, MO.ng, MO.ncomp, ncomp0, X[0], X[1], X[2])] = no.x;
  o[fab_index(B, MO.ng, MO.ncomp, ncomp0 + 1, X[0], X[1], X[2])] = no.y;
  o[fab_index(B, MO.ng, MO.ncomp, ncomp0 + 2, X[0], X[1], X[2])] = no.z;
}

// Phase B: thread per (cell, layer 1..2) behind such a ghost cell: K = 0.5 * div n with the ghost
// normals of MLMG applyBC on n_d (curvature.cpp:510-546).  Normals of cells of this box are read
// back from the output (exact after phase A) unless the threshold clip zeroed them there; normals
// of valid cells of neighbouring boxes are recomputed from the local ghost c.
// CG = false: c from the stored shell copy MC_[ccomp]; CG = true: MC_[ccomp] is PHI and c comes through CgAcc.
// cells the clip-aware fast path hands to the general one (exact-normal pipeline with the threshold clip): {batch row, face cell}
struct SlowList {
  int* count; int2* items; int cap;
  // cells with a VALID ghost cell (general BoxArrays) that need the neighbouring box's unclipped normal: a list for k_curv_general
  int* gcount = nullptr; int4* gitems = nullptr; int gcap = 0; unsigned glev = 0;
};
// CGCLIP: the threshold clip in the exact-normal pipeline (compiled in only where it is used: 116 against 168 VGPRs)
template <bool CG, bool PATCH, bool CGCLIP = false>
__device__ __forceinline__ void faces_curv_cell(const LevBatch<FixArgs>& Bt, unsigned y, long long t, int perim_only, int* nbad, const SlotK& sk, int z) {
  unsigned fy;
  const FixArgs& Fx = Bt.a[Bt.find(y, fy)];
  const DLevelView& L = Fx.L;
  const DMFView& MC_ = Fx.MC_;
  const DLevelView& LCr = Fx.LCr;
  const DMFView& MN = Fx.MN;
  const DMFView& MO = Fx.MO;
  FaceArgs A = Fx.A;
  if (sk.prog) { A.pmin = sk.prog[2 * z]; A.invd = sk.prog[2 * z + 1]; }
  const int ccomp = Fx.ccomp + z, cncomp0 = Fx.cncomp0 + sk.cn_z * z, ncomp0 = Fx.ncomp0 + 8 * z, kcomp = Fx.kcomp + 8 * z;
  const double* cgz = L.cg + z * Fx.cg_stride;
  const double* cpz = L.cp ? L.cp + z * Fx.cp_stride : nullptr;
  int b, fdir, side, layer, q0[3];
  DBox B;
  if (perim_only) {
    // compact enumeration of the perimeter cells of the face (the interior belongs to k_faces_curv_fast):
    // two full rows in t0, then the two end columns of the rows in between; layers slowest
    const int e = L.sfaces[fy];
    b = e / 6; fdir = (e % 6) >> 1; side = e & 1;
    B = L.boxes[b];
    const int t0 = (fdir == 0) ? 1 : 0, t1 = (fdir == 2) ? 1 : 2;
    const unsigned n0 = B.hi[t0] - B.lo[t0] + 1, n1 = B.hi[t1] - B.lo[t1] + 1;
    const unsigned P = (n1 >= 2) ? 2 * n0 + 2 * (n1 - 2) : n0;
    if (t >= (long long)P * A.layers) return;
    unsigned r = (unsigned)t;
    layer = 0;
    while (r >= P) { r -= P; ++layer; }
    unsigned a0, a1;
    if (r < n0) { a0 = r; a1 = 0; }
    else if (r < 2 * n0) { a0 = r - n0; a1 = n1 - 1; }
    else { r -= 2 * n0; a0 = (r & 1u) ? n0 - 1 : 0; a1 = 1 + (r >> 1); }
    if (n0 == 1 && (r & 1u) && t >= 2 * (long long)n0) return;  // a single column: do not visit it twice
    q0[fdir] = side ? B.hi[fdir] + 1 : B.lo[fdir] - 1;
    q0[t0] = B.lo[t0] + (int)a0;
    q0[t1] = B.lo[t1] + (int)a1;
    if (layer >= B.hi[fdir] - B.lo[fdir] + 1) return;
    if ((L.sfcode[L.sfoff[fy] + a0 + (long long)n0 * a1] & 3u) == 0) return;
  } else {
    if (!sface_decode(L, fy, t, A.layers, b, B, fdir, side, q0, layer)) return;
    const int t0 = (fdir == 0) ? 1 : 0, t1 = (fdir == 2) ? 1 : 2;
    if (layer >= B.hi[fdir] - B.lo[fdir] + 1) return;
    if ((L.sfcode[L.sfoff[fy] + (t - (long long)layer * (B.hi[t0] - B.lo[t0] + 1) * (B.hi[t1] - B.lo[t1] + 1))] & 3u) == 0) return;
  }
  const int n[3] = {B.hi[0] - B.lo[0] + 1, B.hi[1] - B.lo[1] + 1, B.hi[2] - B.lo[2] + 1};
  int X[3] = {q0[0], q0[1], q0[2]};
  X[fdir] += side ? -(1 + layer) : (1 + layer);
  const ShellAcc Cs = {mf_view(MC_, B, b), ccomp};
  const CgAcc Cg = {&L, mf_view(MC_, B, b), B, b, ccomp, A.pmin, A.invd, cgz};
  const double dxinv[3] = {L.dxinv[0], L.dxinv[1], L.dxinv[2]};
  const double* o = MO.data + MO.off[b];
  auto C = [&](int i, int j, int k) -> double { return CG ? Cg(i, j, k) : Cs(i, j, k); };
  // component d of the (unclipped) normal at a valid cell p of this box
  auto nrm = [&](const int p[3], int d) -> double {
    if ((!CG || CGCLIP) && A.thr >= 0.0) {  // the sweep zeroed clipped normals in the output: the divergence needs the unclipped ones
      const double cp = C(p[0], p[1], p[2]);
      if (cp < A.thr || cp > 1.0 - A.thr) return comp_of(normal_at(C, p[0], p[1], p[2], dxinv), d);
    }
    return o[fab_index(B, MO.ng, MO.ncomp, ncomp0 + d, p[0], p[1], p[2])];
  };
  double curv = 0.0;
  bool ok = true;
  for (int d = 0; d < 3; ++d) {
    const double n0d = nrm(X, d);
    double nb[2];
    for (int s2 = 0; s2 < 2; ++s2) {
      const int sg = s2 ? 1 : -1;
      int q[3] = {X[0], X[1], X[2]};
      q[d] += sg;
      if (in_box(B, q)) { nb[s2] = nrm(q, d); continue; }
      // q is a ghost cell of face (d, s2) of this box: its masks are stored unless the face is ordinary
      unsigned code = 0;
      const int e2 = L.sfindex[b * 6 + d * 2 + s2];
      {
        const int u0 = (d == 0) ? 1 : 0, u1 = (d == 2) ? 1 : 2;
        if (e2 >= 0) code = L.sfcode[L.sfoff[e2] + (q[u0] - B.lo[u0]) + (long long)n[u0] * (q[u1] - B.lo[u1])];
      }
      const int cls = (int)(code & 3u);
      if (cls == 0) {
        nb[s2] = comp_of(normal_at(C, q[0], q[1], q[2], dxinv), d);
      } else if (cls == 2) {
        nb[s2] = (A.bc[d] == PA_BC_REFLECT_ODD) ? -n0d : n0d;
      } else {
        if (!A.has_crse) { ok = false; nb[s2] = 0.0; continue; }
        double coef[4];
        const int NX = cf_normal_coef(n[d], A.ratio, coef);
        const int xf[1] = {0};
        double bv1[1];
        const long long cpo = (Fx.use_cp && L.cp) ? L.cpoff[e2] : -1;  // that face's coarse patch holds component cncomp0 + d
        if (PATCH || cpo >= 0) cf_interp_patch<1>(code, cpz + cpo, B, s2, MN, q, d, xf, ok, bv1);
        else cf_interp<1>(code, LCr, MN, cncomp0 + d, q, d, A.ratio, xf, ok, bv1);
        const double bv = bv1[0];
        double tmp = 0.0;
        for (int m = 1; m < NX; ++m) {
          int pc[3] = {q[0], q[1], q[2]};
          pc[d] -= sg * m;  // into the box
          const double v = (m == 1) ? n0d : nrm(pc, d);
          tmp += v * coef[m];
        }
        double g = tmp;
        g += bv * coef[0];
        nb[s2] = g;
      }
    }
    curv += cdiff(dxinv[d], nb[0], n0d, nb[1]);
  }
  curv = curv * 0.5;
  if ((!CG || CGCLIP) && A.thr >= 0.0) {
    const double c0 = C(X[0], X[1], X[2]);
    if (c0 < A.thr || c0 > 1.0 - A.thr) curv = 0.0;
  }
  if (!ok) atomicAdd(nbad, 1);
  MO.data[MO.off[b] + fab_index(B, MO.ng, MO.ncomp, kcomp, X[0], X[1], X[2])] = curv;
}
template <bool CG, bool PATCH = false, bool CGCLIP = false>
__global__ __launch_bounds__(256, PA_FC_WAVES) void k_faces_curv(LevBatch<FixArgs> Bt, int* nbad, SlotK sk = SlotK()) {
  unsigned fy;
  const int perim = Bt.a[Bt.find(blockIdx.y, fy)].A.perim_only;
  faces_curv_cell<CG, PATCH, CGCLIP>(Bt, blockIdx.y, blockIdx.x * (long long)blockDim.x + threadIdx.x, perim, nbad, sk, (int)blockIdx.z);
}
// The perimeter cells through the levels' perimeter work tables (round 5): the grid of k_faces_curv is (longest perimeter of the
// batch / 256) x faces -- on a hierarchy whose faces differ in size (a 256^2 wall face next to the 32^2 .. 128^2 faces of a flame
// sheet) most workgroups find nothing to do: 333 -> 595 us when level 0 was re-tiled to 256^3 boxes.  One layer only.
template <bool CG, bool PATCH, bool CGCLIP>
__device__ __forceinline__ void faces_tab_wg(const LevBatch<FixArgs>& Bt, int* nbad, const SlotK& sk, unsigned w) {
  int blev = 0;
  while (blev + 1 < Bt.n && w >= (unsigned)Bt.a[blev].npwg) { w -= (unsigned)Bt.a[blev].npwg; ++blev; }
  if (w >= (unsigned)Bt.a[blev].npwg) return;
  const int2 it = Bt.a[blev].pwg[w];
  faces_curv_cell<CG, PATCH, CGCLIP>(Bt, (unsigned)Bt.ycum[blev] + (unsigned)it.x, (long long)it.y * 256 + threadIdx.x, 1, nbad, sk, (int)blockIdx.z);
}
template <bool CG, bool PATCH = false, bool CGCLIP = false>
__global__ __launch_bounds__(256, PA_FC_WAVES) void k_faces_curv_tab(LevBatch<FixArgs> Bt, int* nbad, SlotK sk = SlotK()) {
  faces_tab_wg<CG, PATCH, CGCLIP>(Bt, nbad, sk, blockIdx.x);
}
// the cells of SlowList through the general path (any cell of a face, one layer)
template <bool PATCH>
__global__ __launch_bounds__(256, PA_FC_WAVES) void k_faces_curv_list(LevBatch<FixArgs> Bt, int* nbad, SlowList sl, SlotK sk = SlotK()) {
  const int n = min(*sl.count, sl.cap);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int2 it = sl.items[i];  // x = batch row | slot << 24
    faces_curv_cell<true, PATCH, true>(Bt, (unsigned)it.x & 0xffffffu, it.y, 0, nbad, sk, (int)((unsigned)it.x >> 24));
  }
}

// Phase B, fast path: the interior cells of a special face (all four tangential neighbours inside the
// box), no threshold clip, boxes >= 3 cells thick.  One thread per face cell does BOTH layers: the
// normals it needs are plain loads from the output (exact after phase A), shared between the two
// layers, and only the ghost normal beyond the face needs the boundary condition.  Same operations in
// the same order as k_faces_curv (d = 0,1,2; cdiff; *0.5), which still handles the perimeter cells.
// CLIP (exact-normal pipeline with the threshold clip, curvature.cpp:549-570): the sweep wrote N = 0, K = 0 where the
// progress variable is outside [thr, 1 - thr].  A clipped cell keeps its K = 0; an unclipped one needs the UNCLIPPED normals
// of its neighbours: a stored component that is exactly 0.0 may be a clipped one -- then (and only then) the neighbour's
// progress variable is formed from phi and, if it is clipped, the cell is appended to SlowList and k_faces_curv_list
// recomputes it through the general path (normals from c: CgAcc, the same operations as the sweep's); inlining that
// recomputation here cost 2.4 KB of scratch per lane.  The coarse normal under a coarse-fine face is taken as stored =
// clipped (quirk Q2).
template <int FD, int NL, bool PATCH, bool CLIP = false>
__device__ __forceinline__ void faces_curv_fast_body(const DLevelView& L, const DLevelView& LCr, const DMFView& MN, int cncomp0, const DMFView& MO,
                                                     int ncomp0, int kcomp, const FaceArgs& A, int* nbad, int b, const DBox& B, int side,
                                                     const int q0[3], unsigned code, const double* patch, const DMFView& MP = DMFView(), int pcomp = 0, SlowList sl = SlowList(),
                                                     unsigned row = 0, long long tcell = 0, const double* ncgp = nullptr, long long ncgs = 0) {
  constexpr int T0 = (FD == 0) ? 1 : 0, T1 = (FD == 2) ? 1 : 2;
  const int cls = (int)(code & 3u);
  // cls == 0: a VALID ghost cell behind a special face (a face that is coarse-fine elsewhere; general BoxArrays).  The sweep's
  // ghost normals behind a special face are unusable (pa_fused_march3.h re-aims the stream they need at the compact array), so
  // the cell's curvature is formed here too, with the neighbour's FINAL normal read from the box that owns the ghost cell.
  // Exact-normal pipeline only (NL == 1: the normals are final once the sweeps are done); a ghost cell owned by another rank's
  // box is on the level's irregular list instead (k_find_irregular, k_curv_general rebuilds its normal).
  int sbn = -1, qw[3] = {0, 0, 0};
  if (cls == 0) {
    if (NL > 1) return;
    if (classify(L, q0[0], q0[1], q0[2], sbn, qw) != 0 || sbn < 0) return;
  }
  const int n[3] = {B.hi[0] - B.lo[0] + 1, B.hi[1] - B.lo[1] + 1, B.hi[2] - B.lo[2] + 1};
  const int sg = side ? 1 : -1;  // the ghost cell sits at X1 + sg e_FD
  int X1[3] = {q0[0], q0[1], q0[2]};
  X1[FD] -= sg;
  const long long nxo = n[0] + 2 * MO.ng, nyo = n[1] + 2 * MO.ng, nzo = n[2] + 2 * MO.ng;
  const long long cso = pa_cstride(nxo * nyo * nzo, MO.ncomp);
  const long long st[3] = {1, nxo, nxo * nyo};
  const long long idx1 = ((long long)(X1[2] - B.lo[2] + MO.ng) * nyo + (X1[1] - B.lo[1] + MO.ng)) * nxo + (X1[0] - B.lo[0] + MO.ng);
  const long long in = -sg * st[FD];
  double* o = MO.data + MO.off[b];
  const double* nf = o + (long long)(ncomp0 + FD) * cso + idx1;
  const double* n0p = o + (long long)(ncomp0 + T0) * cso + idx1;
  const double* n1p = o + (long long)(ncomp0 + T1) * cso + idx1;
  // NCG (x faces, one layer, no clip): N_x of the first three cells and the two tangential terms of K as the sweep formed them, from the
  // level's face-major arrays -- five contiguous streams instead of 8 bytes of five different lines
  const bool pre = FD == 0 && NL == 1 && !CLIP && ncgp != nullptr;
  double nfd1, nfd2, nfd3, a0m = 0, a0c = 0, a0p = 0, a1m = 0, a1c = 0, a1p = 0, t01n = 0, t11n = 0;
  if (pre) {  // three arrays of pairs: (N_x of the first, second cell), (N_x of the third cell, y term of K), (z term of K, -)
    const pa_fix_d2* np = (const pa_fix_d2*)ncgp;
    const pa_fix_d2 v0 = np[0], v1 = np[ncgs], v2 = np[2 * ncgs];
    nfd1 = v0.x; nfd2 = v0.y; nfd3 = v1.x; t01n = v1.y; t11n = v2.x;
  } else {
    nfd1 = nf[0]; nfd2 = nf[in]; nfd3 = nf[2 * in];
    a0m = n0p[-st[T0]]; a0c = n0p[0]; a0p = n0p[st[T0]];
    a1m = n1p[-st[T1]]; a1c = n1p[0]; a1p = n1p[st[T1]];
  }
  if (CLIP) {
    static_assert(!CLIP || NL == 1, "the clip-aware fast path fixes one layer");
    const FabView P = mf_view(MP, B, b);
    auto clipped = [&](int i, int j, int k) {
      const double c = (P(i, j, k, pcomp) - A.pmin) * A.invd;
      return c < A.thr || c > 1.0 - A.thr;
    };
    if (clipped(X1[0], X1[1], X1[2])) return;  // K = 0 from the sweep
    const int iv = -sg;  // one cell into the box along FD
    bool slow = false;   // a needed neighbour component that the sweep clipped: this cell goes through the general path
    if (cls == 0) {  // the neighbouring box's normal: clipped there if its progress variable (this FAB's ghost cell holds its phi) is
      const double gn = MO.data[MO.off[sbn] + fab_index(L.boxes[sbn], MO.ng, MO.ncomp, ncomp0 + FD, qw[0], qw[1], qw[2])];
      slow = gn == 0.0 && clipped(q0[0], q0[1], q0[2]);
    }
    slow = slow || (nfd2 == 0.0 && clipped(X1[0] + (FD == 0 ? iv : 0), X1[1] + (FD == 1 ? iv : 0), X1[2] + (FD == 2 ? iv : 0)));
    slow = slow || (nfd3 == 0.0 && clipped(X1[0] + (FD == 0 ? 2 * iv : 0), X1[1] + (FD == 1 ? 2 * iv : 0), X1[2] + (FD == 2 ? 2 * iv : 0)));
    slow = slow || (a0m == 0.0 && clipped(X1[0] - (T0 == 0), X1[1] - (T0 == 1), X1[2]));
    slow = slow || (a0p == 0.0 && clipped(X1[0] + (T0 == 0), X1[1] + (T0 == 1), X1[2]));
    slow = slow || (a1m == 0.0 && clipped(X1[0], X1[1] - (T1 == 1), X1[2] - (T1 == 2)));
    slow = slow || (a1p == 0.0 && clipped(X1[0], X1[1] + (T1 == 1), X1[2] + (T1 == 2)));
    if (slow && cls == 0) {  // k_curv_general<true> over the context's dynamic list: {box | batch level << 24 | slot << 27, cell}
      const int i = atomicAdd(sl.gcount, 1);
      if (i < sl.gcap) sl.gitems[i] = make_int4(b | (int)(sl.glev << 24) | (int)((row >> 24) << 27), X1[0], X1[1], X1[2]);
      else atomicAdd(nbad, 1);
      return;
    }
    if (slow) {
      const int i = atomicAdd(sl.count, 1);
      if (i < sl.cap) sl.items[i] = make_int2((int)row, (int)tcell);
      else atomicAdd(nbad, 1);
      return;
    }
  }
  double b0m = 0, b0c = 0, b0p = 0, b1m = 0, b1c = 0, b1p = 0;
  if (NL > 1) {
    b0m = n0p[in - st[T0]]; b0c = n0p[in]; b0p = n0p[in + st[T0]];
    b1m = n1p[in - st[T1]]; b1c = n1p[in]; b1p = n1p[in + st[T1]];
  }
  // ghost normal: MLMG applyBC on n_FD (curvature.cpp:510-531)
  double g;
  bool ok = true;
  if (cls == 0) {
    g = MO.data[MO.off[sbn] + fab_index(L.boxes[sbn], MO.ng, MO.ncomp, ncomp0 + FD, qw[0], qw[1], qw[2])];
  } else if (cls == 2) {
    g = (A.bc[FD] == PA_BC_REFLECT_ODD) ? -nfd1 : nfd1;
  } else {
    if (!A.has_crse) { ok = false; g = 0.0; }
    else {
      double coef[4];
      const int NX = cf_normal_coef(n[FD], A.ratio, coef);
      const int xf[1] = {0};
      double bv1[1];
      if (PATCH || patch) cf_interp_patch<1>(code, patch, B, side, MN, q0, FD, xf, ok, bv1);  // the face's coarse patch holds component cncomp0 + FD
      else cf_interp<1>(code, LCr, MN, cncomp0 + FD, q0, FD, A.ratio, xf, ok, bv1);
      double tmp = 0.0;
      for (int m = 1; m < NX; ++m) {
        const double v = (m == 1) ? nfd1 : (m == 2 ? nfd2 : nfd3);
        tmp += v * coef[m];
      }
      g = tmp;
      g += bv1[0] * coef[0];
    }
  }
  const double dx0 = L.dxinv[0], dx1 = L.dxinv[1], dx2 = L.dxinv[2];
  // face-normal terms: (minus neighbour, centre, plus neighbour)
  const double f1 = side ? cdiff(L.dxinv[FD], nfd2, nfd1, g) : cdiff(L.dxinv[FD], g, nfd1, nfd2);
  const double f2 = side ? cdiff(L.dxinv[FD], nfd3, nfd2, nfd1) : cdiff(L.dxinv[FD], nfd1, nfd2, nfd3);
  const double t01 = pre ? t01n : cdiff(L.dxinv[T0], a0m, a0c, a0p), t11 = pre ? t11n : cdiff(L.dxinv[T1], a1m, a1c, a1p);
  const double t02 = cdiff(L.dxinv[T0], b0m, b0c, b0p), t12 = cdiff(L.dxinv[T1], b1m, b1c, b1p);
  (void)dx0; (void)dx1; (void)dx2;
  double k1 = 0.0, k2 = 0.0;
  // d = 0, 1, 2 in order: the term of direction d is the face-normal one when d == FD, else T0's or T1's
  k1 += (FD == 0) ? f1 : t01;
  k1 += (FD == 1) ? f1 : (FD == 0 ? t01 : t11);
  k1 += (FD == 2) ? f1 : t11;
  k2 += (FD == 0) ? f2 : t02;
  k2 += (FD == 1) ? f2 : (FD == 0 ? t02 : t12);
  k2 += (FD == 2) ? f2 : t12;
  k1 = k1 * 0.5;
  k2 = k2 * 0.5;
  if (!ok) atomicAdd(nbad, 1);
  double* ko = o + (long long)kcomp * cso + idx1;
  ko[0] = k1;
  if (NL > 1) ko[in] = k2;
}

// PATCH: every coarse-fine face of every level of the batch has its coarse patch (the owner-map interpolation is not compiled in)
template <int NL, bool PATCH, bool CLIP>
__device__ __forceinline__ void faces_fast_cell(const LevBatch<FixArgs>& Bt, int* nbad, SlowList sl, const SlotK& sk, int blev, unsigned fy, long long t) {
  const FixArgs& Fx = Bt.a[blev];
  sl.glev = (unsigned)blev;
  const DLevelView& L = Fx.L;
  const DLevelView& LCr = Fx.LCr;
  const DMFView& MN = Fx.MN;
  const DMFView& MO = Fx.MO;
  const int z = (int)blockIdx.z;
  FaceArgs A = Fx.A;
  if (sk.prog) { A.pmin = sk.prog[2 * z]; A.invd = sk.prog[2 * z + 1]; }
  const int cncomp0 = Fx.cncomp0 + sk.cn_z * z, ncomp0 = Fx.ncomp0 + 8 * z, kcomp = Fx.kcomp + 8 * z;
  int b, fdir, side, layer, q0[3];
  DBox B;
  if (!sface_decode(L, fy, t, 1, b, B, fdir, side, q0, layer)) return;
  const int t0 = (fdir == 0) ? 1 : 0, t1 = (fdir == 2) ? 1 : 2;
  if (!(q0[t0] > B.lo[t0] && q0[t0] < B.hi[t0] && q0[t1] > B.lo[t1] && q0[t1] < B.hi[t1])) return;  // perimeter: k_faces_curv
  const unsigned code = L.sfcode[L.sfoff[fy] + t];
  const long long cpo = (Fx.use_cp && L.cp) ? L.cpoff[fy] : -1;  // wave-uniform
  const double* patch = cpo >= 0 ? L.cp + z * Fx.cp_stride + cpo : nullptr;
  const unsigned row = ((unsigned)Bt.ycum[blev] + fy) | ((unsigned)z << 24);  // batch row of the face; SlowList entries carry the slot
  const double* ncgp = nullptr;
  if (NL == 1 && !CLIP && fdir == 0 && Fx.ncg && z == 0 && ncg_face_ok(B, side, Fx.ncg_minw))
    ncgp = Fx.ncg + 2 * (L.cgoff[fy] + (long long)(q0[2] - B.lo[2] + 1) * (B.hi[1] - B.lo[1] + 3) + (q0[1] - B.lo[1] + 1));
  switch (fdir) {  // uniform per workgroup
    case 0: faces_curv_fast_body<0, NL, PATCH, CLIP>(L, LCr, MN, cncomp0, MO, ncomp0, kcomp, A, nbad, b, B, side, q0, code, patch, Fx.MC_, Fx.ccomp + z, sl, row, t, ncgp, Fx.ncgs); break;
    case 1: faces_curv_fast_body<1, NL, PATCH, CLIP>(L, LCr, MN, cncomp0, MO, ncomp0, kcomp, A, nbad, b, B, side, q0, code, patch, Fx.MC_, Fx.ccomp + z, sl, row, t); break;
    default: faces_curv_fast_body<2, NL, PATCH, CLIP>(L, LCr, MN, cncomp0, MO, ncomp0, kcomp, A, nbad, b, B, side, q0, code, patch, Fx.MC_, Fx.ccomp + z, sl, row, t); break;
  }
}
template <int NL, bool PATCH, bool CLIP>
__device__ __forceinline__ void faces_fast_wg(const LevBatch<FixArgs>& Bt, int* nbad, SlowList sl, const SlotK& sk, unsigned w) {
  unsigned fy;
  int blev;
  long long t;
  if (!wg_decode(Bt, blev, fy, t, w)) return;
  faces_fast_cell<NL, PATCH, CLIP>(Bt, nbad, sl, sk, blev, fy, t);
}


// ---- round 6: the coarse-fine interpolation of the four ghost cells of a 2 x 2 block for ANY mix of codes, without branches.  The
// block's cells share the coarse parent qc; InterpBndryData reaches at most two coarse cells along each tangential axis and the
// four diagonal neighbours, so 13 values of the face's coarse patch are a superset of what the four cells read (always inside the
// patch: it covers coarsen(lo - 1) - 2 .. coarsen(hi + 1) + 2).  Per cell: the stencil extents come out of its code, the weights
// out of a copy of g_cf_coef.tan in LDS, the values out of the superset by selects, and a term outside the cell's stencil is
// SKIPPED by a select (not multiplied by zero) -- the additions that happen are cf_interp_core's, in its order.
struct CfBlock {
  double ax0[5], ax1[5], dg[4];  // coarse(qc + a e_t0), coarse(qc + a e_t1) for a = -2 .. 2; diagonals (+,+) (-,+) (-,-) (+,-)
  unsigned miss;                 // bit i: ax0[i]; bit 5 + i: ax1[i]; bit 10 + i: dg[i] has no coarse owner (zeroed here, as craw does)
};
__device__ __forceinline__ void cf_block_load(const double* cb, int pw, CfBlock& K) {
#pragma unroll
  for (int a = 0; a < 5; ++a) K.ax0[a] = cb[a - 2];
#pragma unroll
  for (int a = 0; a < 5; ++a) K.ax1[a] = (a == 2) ? 0.0 : cb[(long long)(a - 2) * pw];
  K.dg[0] = cb[pw + 1]; K.dg[1] = cb[pw - 1]; K.dg[2] = cb[-pw - 1]; K.dg[3] = cb[-pw + 1];
}
__device__ __forceinline__ void cf_block_finish(CfBlock& K) {  // after the loads have been issued with everything else
  K.ax1[2] = K.ax0[2];
  K.miss = 0;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    if (__double_as_longlong(K.ax0[a]) == PA_CP_MISSING) { K.miss |= 1u << a; K.ax0[a] = 0.0; }
    if (__double_as_longlong(K.ax1[a]) == PA_CP_MISSING) { K.miss |= 1u << (5 + a); K.ax1[a] = 0.0; }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
    if (__double_as_longlong(K.dg[a]) == PA_CP_MISSING) { K.miss |= 1u << (10 + a); K.dg[a] = 0.0; }
}
__device__ __forceinline__ void cf_tab_to_lds(double* tabs) {  // g_cf_coef.tan, flat: ((rem * 3 + lo + 2) * 3 + hi) * 3 + m
  if (threadIdx.x < 54) tabs[threadIdx.x] = (&g_cf_coef.tan[0][0][0][0])[threadIdx.x];
  __syncthreads();
}
// child (du, dv) of the parent, masks `code` (class 1); field 0 sees the raw coarse values, field 1 (NF == 2) (v - xa) * xb.
// Returns true when the cell's stencil touches a coarse cell without an owner.
// one tangential direction of cf_block_interp: the (up to three) stencil points lo .. hi out of the five axis values e0 .. e4 = coarse(qc + a e_t),
// a = -2 .. 2 (by value: a pointer chosen by the direction would keep the block in memory and turn the selects into an indexed load)
template <int NF>
__device__ __forceinline__ unsigned cf_block_axis(unsigned fld, int rem, const double* tabs, double e0, double e1, double e2, double e3, double e4, double xa, double xb, double b[NF]) {
  const int lo2 = (int)(fld & 3u), hi = (int)((fld >> 2) & 3u);  // lo2 = lo + 2
  const int N = hi - (lo2 - 2) + 1;
  const double* ct = tabs + ((rem * 3 + lo2) * 3 + hi) * 3;
  const double c0 = ct[0], c1 = ct[1], c2 = ct[2];
  const double v0 = lo2 == 0 ? e0 : (lo2 == 1 ? e1 : e2), v1 = lo2 == 0 ? e1 : (lo2 == 1 ? e2 : e3), v2 = lo2 == 0 ? e2 : (lo2 == 1 ? e3 : e4);
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const double w0 = f ? (v0 - xa) * xb : v0, w1 = f ? (v1 - xa) * xb : v1, w2 = f ? (v2 - xa) * xb : v2;
    b[f] += c0 * w0;
    const double s1 = b[f] + c1 * w1;
    b[f] = N > 1 ? s1 : b[f];
    const double s2 = b[f] + c2 * w2;
    b[f] = N > 2 ? s2 : b[f];
  }
  return ((1u << (hi + 3)) - 1u) & ~((1u << lo2) - 1u);  // the axis values the stencil uses
}
// child (du, dv) of the parent, masks `code` (class 1); field 0 sees the raw coarse values, field 1 (NF == 2) (v - xa) * xb.
// Returns true when the cell's stencil touches a coarse cell without an owner.
template <int NF>
__device__ __forceinline__ bool cf_block_interp(const CfBlock& K, unsigned code, int du, int dv, const double* tabs, double xa, double xb, double b[NF]) {
#pragma unroll
  for (int f = 0; f < NF; ++f) b[f] = 0.0;
  unsigned used = cf_block_axis<NF>((code >> 2) & 15u, du, tabs, K.ax0[0], K.ax0[1], K.ax0[2], K.ax0[3], K.ax0[4], xa, xb, b);
  used |= cf_block_axis<NF>((code >> 6) & 15u, dv, tabs, K.ax1[0], K.ax1[1], K.ax1[2], K.ax1[3], K.ax1[4], xa, xb, b) << 5;
  const bool cross = (code & (1u << 10)) != 0;
  used |= cross ? (0xFu << 10) : 0u;
  const double xi0 = du ? 0.25 : -0.25, xi1 = dv ? 0.25 : -0.25;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const double ce = f ? (K.ax0[2] - xa) * xb : K.ax0[2];
    b[f] -= ce;
    const double vpp = f ? (K.dg[0] - xa) * xb : K.dg[0], vmp = f ? (K.dg[1] - xa) * xb : K.dg[1];
    const double vmm = f ? (K.dg[2] - xa) * xb : K.dg[2], vpm = f ? (K.dg[3] - xa) * xb : K.dg[3];
    const double sc = b[f] + ((xi0 * xi1) * 0.25) * (((vpp - vmp) + vmm) - vpm);
    b[f] = cross ? sc : b[f];
  }
  return (K.miss & used) != 0;
}

// ---- round 6: the face interiors from the levels' chunk records (see k_prep_faces_chunks below for the scheme): a thread takes the
// 2 x 2 block of first-layer cells whose ghost cells share one coarse parent.  Uniform chunks (all coarse-fine with the full
// stencil / all behind a wall) run straight-line code -- the parent's 3 x 3 coarse normals loaded once for the four cells, the
// tangential neighbours of a row or column of the block shared, interpolation weights as literals; per cell the operations and their
// order are faces_curv_fast_body<FD, 1, PATCH>'s with code PA_CODE_FULL (same bits).  Mixed chunks: faces_fast_cell per cell.
template <int FD>
__device__ __forceinline__ void fix_chunk_uniform(const FixArgs& Fx, const SfChunk& D, int z, int* nbad) {
  constexpr int T0 = (FD == 0) ? 1 : 0, T1 = (FD == 2) ? 1 : 2;
  const int side = D.dir_side & 1;
  const DMFView& MO = Fx.MO;
  const DLevelView& L = Fx.L;
  const int n[3] = {D.hi[0] - D.lo[0] + 1, D.hi[1] - D.lo[1] + 1, D.hi[2] - D.lo[2] + 1};
  const int n0 = n[T0], n1 = n[T1];
  const int hw = D.cw >> 1, sh = 31 - __builtin_clz((unsigned)hw);
  const int u = D.u0 + 2 * ((int)threadIdx.x & (hw - 1)), v = D.v0 + 2 * ((int)threadIdx.x >> sh);
  if (u >= n0 || v >= n1) return;
  const int ng = MO.ng, ncomp0 = Fx.ncomp0 + 8 * z, kcomp = Fx.kcomp + 8 * z;
  const long long nxo = n[0] + 2 * ng, nyo = n[1] + 2 * ng, nzo = n[2] + 2 * ng;
  const long long cso = pa_cstride(nxo * nyo * nzo, MO.ncomp);
  const long long st[3] = {1, nxo, nxo * nyo};
  int X1[3];
  X1[FD] = side ? D.hi[FD] : D.lo[FD];
  X1[T0] = D.lo[T0] + u;
  X1[T1] = D.lo[T1] + v;
  const long long idx1 = ((long long)(X1[2] - D.lo[2] + ng) * nyo + (X1[1] - D.lo[1] + ng)) * nxo + (X1[0] - D.lo[0] + ng);
  const long long in = side ? -st[FD] : st[FD], s0 = st[T0], s1 = st[T1];
  double* const o = MO.data + MO.off[D.box];
  const double* const nf = o + (long long)(ncomp0 + FD) * cso + idx1;
  const double* const n0p = o + (long long)(ncomp0 + T0) * cso + idx1;
  const double* const n1p = o + (long long)(ncomp0 + T1) * cso + idx1;
  double* const ko = o + (long long)kcomp * cso + idx1;
  DBox B;
#pragma unroll
  for (int d = 0; d < 3; ++d) { B.lo[d] = D.lo[d]; B.hi[d] = D.hi[d]; }
  const bool wall = (D.flags & PA_SFC_WALL) != 0;
  // NCG: this pass's sweep mirrored the first layer behind this x face (N_x of the first three cells, the y and z terms of K)
  const bool pre = FD == 0 && Fx.ncg && z == 0 && ncg_face_ok(B, side, Fx.ncg_minw);
  double nfd[2][2][3], A0[2][4], A1[2][4], t01n[2][2], t11n[2][2];
  // tangential neighbours: positions u - 1 .. u + 2 of each of the block's two rows (rows: v - 1 .. v + 2 of its two columns); a
  // position outside the face belongs to a perimeter cell, which is not written here -- clamped onto the face
  long long k0[4], k1o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    k0[k] = (long long)(min(max(u + k - 1, 0), n0 - 1) - u) * s0;
    k1o[k] = (long long)(min(max(v + k - 1, 0), n1 - 1) - v) * s1;
  }
  if (pre) {
    const pa_fix_d2* np = (const pa_fix_d2*)(Fx.ncg + 2 * (D.cgoff + (long long)(v + 1) * (n0 + 2) + (u + 1)));
#pragma unroll
    for (int dv = 0; dv < 2; ++dv)
#pragma unroll
      for (int du = 0; du < 2; ++du) {
        const pa_fix_d2 v0 = np[dv * (n0 + 2) + du], v1 = np[Fx.ncgs + dv * (n0 + 2) + du], v2 = np[2 * Fx.ncgs + dv * (n0 + 2) + du];
        nfd[dv][du][0] = v0.x; nfd[dv][du][1] = v0.y; nfd[dv][du][2] = v1.x; t01n[dv][du] = v1.y; t11n[dv][du] = v2.x;
      }
  } else {
#pragma unroll
    for (int dv = 0; dv < 2; ++dv)
#pragma unroll
      for (int du = 0; du < 2; ++du)
#pragma unroll
        for (int m = 0; m < 3; ++m) nfd[dv][du][m] = nf[dv * s1 + du * s0 + m * in];
#pragma unroll
    for (int dv = 0; dv < 2; ++dv)
#pragma unroll
      for (int k = 0; k < 4; ++k) A0[dv][k] = n0p[dv * s1 + k0[k]];
#pragma unroll
    for (int du = 0; du < 2; ++du)
#pragma unroll
      for (int k = 0; k < 4; ++k) A1[du][k] = n1p[du * s0 + k1o[k]];
  }
  double r[3][3];
  if (!wall) {
    int plane, pu0, pv0, pw, ph;
    cpatch_geom(B, FD, side, plane, pu0, pv0, pw, ph);
    const double* const cb = L.cp + z * Fx.cp_stride + D.cpoff + (long long)((X1[T1] >> 1) - pv0) * pw + ((X1[T0] >> 1) - pu0);
    bool ok = true;
#pragma unroll
    for (int a1 = 0; a1 < 3; ++a1)
#pragma unroll
      for (int a0 = 0; a0 < 3; ++a0) r[a1][a0] = cb[(a1 - 1) * pw + (a0 - 1)];
#pragma unroll
    for (int a1 = 0; a1 < 3; ++a1)
#pragma unroll
      for (int a0 = 0; a0 < 3; ++a0)
        if (__double_as_longlong(r[a1][a0]) == PA_CP_MISSING) { ok = false; r[a1][a0] = 0.0; }
    if (!ok) atomicAdd(nbad, ((int)(u > 0) + (int)(u + 1 < n0 - 1)) * ((int)(v > 0) + (int)(v + 1 < n1 - 1)));  // counted per face-interior cell of the block
  }
  const bool odd = Fx.A.bc[FD] == PA_BC_REFLECT_ODD;
  constexpr double nc0 = k_cf_coef.nrm[4][0], nc1 = k_cf_coef.nrm[4][1], nc2 = k_cf_coef.nrm[4][2], nc3 = k_cf_coef.nrm[4][3];
#pragma unroll
  for (int dv = 0; dv < 2; ++dv)
#pragma unroll
    for (int du = 0; du < 2; ++du) {
      const double nfd1 = nfd[dv][du][0], nfd2 = nfd[dv][du][1], nfd3 = nfd[dv][du][2];
      double g;
      if (wall) {
        g = odd ? -nfd1 : nfd1;
      } else {
        const double c00 = du ? k_cf_coef.tan[1][1][1][0] : k_cf_coef.tan[0][1][1][0], c01 = du ? k_cf_coef.tan[1][1][1][1] : k_cf_coef.tan[0][1][1][1],
                     c02 = du ? k_cf_coef.tan[1][1][1][2] : k_cf_coef.tan[0][1][1][2];
        const double c10 = dv ? k_cf_coef.tan[1][1][1][0] : k_cf_coef.tan[0][1][1][0], c11 = dv ? k_cf_coef.tan[1][1][1][1] : k_cf_coef.tan[0][1][1][1],
                     c12 = dv ? k_cf_coef.tan[1][1][1][2] : k_cf_coef.tan[0][1][1][2];
        const double xi0 = du ? 0.25 : -0.25, xi1 = dv ? 0.25 : -0.25;
        double b0 = 0.0;
        b0 += c00 * r[1][0];
        b0 += c01 * r[1][1];
        b0 += c02 * r[1][2];
        b0 += c10 * r[0][1];
        b0 += c11 * r[1][1];
        b0 += c12 * r[2][1];
        b0 -= r[1][1];
        b0 += ((xi0 * xi1) * 0.25) * (((r[2][2] - r[2][0]) + r[0][0]) - r[0][2]);
        double tmp = 0.0;
        tmp += nfd1 * nc1;
        tmp += nfd2 * nc2;
        tmp += nfd3 * nc3;
        g = tmp;
        g += b0 * nc0;
      }
      const double f1 = side ? cdiff(L.dxinv[FD], nfd2, nfd1, g) : cdiff(L.dxinv[FD], g, nfd1, nfd2);
      const double t01 = pre ? t01n[dv][du] : cdiff(L.dxinv[T0], A0[dv][du], A0[dv][du + 1], A0[dv][du + 2]);
      const double t11 = pre ? t11n[dv][du] : cdiff(L.dxinv[T1], A1[du][dv], A1[du][dv + 1], A1[du][dv + 2]);
      double k1 = 0.0;
      k1 += (FD == 0) ? f1 : t01;
      k1 += (FD == 1) ? f1 : (FD == 0 ? t01 : t11);
      k1 += (FD == 2) ? f1 : t11;
      k1 = k1 * 0.5;
      const int uu = u + du, vv = v + dv;
      if (uu > 0 && uu < n0 - 1 && vv > 0 && vv < n1 - 1) ko[dv * s1 + du * s0] = k1;  // the perimeter is k_faces_curv_tab's
    }
}


// the face interiors of a chunk with any mix of cell kinds (see prep_chunk_mixed): the ghost normal of a first-layer cell is the
// boundary condition on n across a coarse-fine face, the mirror image behind a wall, or -- a valid ghost cell behind a partly
// covered face -- the neighbouring box's FINAL normal, read in the box that owns it (a branch, taken only by waves that have such
// cells; a ghost cell owned by another rank's box is on the level's irregular list: not written here)
template <int FD>
__device__ __forceinline__ void fix_chunk_mixed(const FixArgs& Fx, const SfChunk& D, int z, int* nbad, const double* tabs) {
  constexpr int T0 = (FD == 0) ? 1 : 0, T1 = (FD == 2) ? 1 : 2;
  const int side = D.dir_side & 1;
  const DMFView& MO = Fx.MO;
  const DLevelView& L = Fx.L;
  const int n[3] = {D.hi[0] - D.lo[0] + 1, D.hi[1] - D.lo[1] + 1, D.hi[2] - D.lo[2] + 1};
  const int n0 = n[T0], n1 = n[T1];
  const int hw = D.cw >> 1, sh = 31 - __builtin_clz((unsigned)hw);
  const int u = D.u0 + 2 * ((int)threadIdx.x & (hw - 1)), v = D.v0 + 2 * ((int)threadIdx.x >> sh);
  if (u >= n0 || v >= n1) return;
  const int ng = MO.ng, ncomp0 = Fx.ncomp0 + 8 * z, kcomp = Fx.kcomp + 8 * z;
  const long long nxo = n[0] + 2 * ng, nyo = n[1] + 2 * ng, nzo = n[2] + 2 * ng;
  const long long cso = pa_cstride(nxo * nyo * nzo, MO.ncomp);
  const long long st[3] = {1, nxo, nxo * nyo};
  int X1[3];
  X1[FD] = side ? D.hi[FD] : D.lo[FD];
  X1[T0] = D.lo[T0] + u;
  X1[T1] = D.lo[T1] + v;
  const long long idx1 = ((long long)(X1[2] - D.lo[2] + ng) * nyo + (X1[1] - D.lo[1] + ng)) * nxo + (X1[0] - D.lo[0] + ng);
  const long long in = side ? -st[FD] : st[FD], s0 = st[T0], s1 = st[T1];
  double* const o = MO.data + MO.off[D.box];
  const double* const nf = o + (long long)(ncomp0 + FD) * cso + idx1;
  const double* const n0p = o + (long long)(ncomp0 + T0) * cso + idx1;
  const double* const n1p = o + (long long)(ncomp0 + T1) * cso + idx1;
  double* const ko = o + (long long)kcomp * cso + idx1;
  DBox B;
#pragma unroll
  for (int d = 0; d < 3; ++d) { B.lo[d] = D.lo[d]; B.hi[d] = D.hi[d]; }
  const bool pre = FD == 0 && Fx.ncg && z == 0 && ncg_face_ok(B, side, Fx.ncg_minw);
  // only face-interior cells are written (the perimeter is k_faces_curv_tab's): every other cell of the block is folded onto the
  // nearest interior cell, so that all of its loads stay inside the FAB (faces at least three cells wide in both directions;
  // narrower ones have no interior cell)
  bool live[2][2];
  long long off[2][2], cgo[2][2];
  unsigned code[2][2];
  const int ulo = min(1, n0 - 1), uhi = max(n0 - 2, 0), vlo = min(1, n1 - 1), vhi = max(n1 - 2, 0);
#pragma unroll
  for (int dv = 0; dv < 2; ++dv)
#pragma unroll
    for (int du = 0; du < 2; ++du) {
      const int uu = u + du, vv = v + dv;
      live[dv][du] = uu > 0 && uu < n0 - 1 && vv > 0 && vv < n1 - 1;
      const int uc = min(max(uu, ulo), uhi), vc = min(max(vv, vlo), vhi);
      off[dv][du] = (long long)(vc - v) * s1 + (long long)(uc - u) * s0;
      cgo[dv][du] = (long long)(vc + 1) * (n0 + 2) + (uc + 1);
      code[dv][du] = L.sfcode[D.sfoff + (long long)vc * n0 + uc];
    }
  if (n0 < 3 || n1 < 3) return;
  const bool cf_here = (D.flags & PA_SFC_HAS_CF) != 0;
  CfBlock K;
  if (cf_here) {
    int plane, pu0, pv0, pw, ph;
    cpatch_geom(B, FD, side, plane, pu0, pv0, pw, ph);
    cf_block_load(L.cp + z * Fx.cp_stride + D.cpoff + (long long)((X1[T1] >> 1) - pv0) * pw + ((X1[T0] >> 1) - pu0), pw, K);
  }
  double nfd[2][2][3], a0[2][2][3], a1[2][2][3], t01n[2][2], t11n[2][2];
  if (pre) {
    const pa_fix_d2* np = (const pa_fix_d2*)(Fx.ncg + 2 * D.cgoff);
#pragma unroll
    for (int dv = 0; dv < 2; ++dv)
#pragma unroll
      for (int du = 0; du < 2; ++du) {
        const pa_fix_d2 v0 = np[cgo[dv][du]], v1 = np[Fx.ncgs + cgo[dv][du]], v2 = np[2 * Fx.ncgs + cgo[dv][du]];
        nfd[dv][du][0] = v0.x; nfd[dv][du][1] = v0.y; nfd[dv][du][2] = v1.x; t01n[dv][du] = v1.y; t11n[dv][du] = v2.x;
      }
  } else {
#pragma unroll
    for (int dv = 0; dv < 2; ++dv)
#pragma unroll
      for (int du = 0; du < 2; ++du) {
#pragma unroll
        for (int m = 0; m < 3; ++m) nfd[dv][du][m] = nf[off[dv][du] + m * in];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          a0[dv][du][k] = n0p[off[dv][du] + (k - 1) * s0];
          a1[dv][du][k] = n1p[off[dv][du] + (k - 1) * s1];
        }
      }
  }
  if (cf_here) cf_block_finish(K);
  const bool odd = Fx.A.bc[FD] == PA_BC_REFLECT_ODD;
  constexpr double nc0 = k_cf_coef.nrm[4][0], nc1 = k_cf_coef.nrm[4][1], nc2 = k_cf_coef.nrm[4][2], nc3 = k_cf_coef.nrm[4][3];
  int nbad_here = 0;
#pragma unroll
  for (int dv = 0; dv < 2; ++dv)
#pragma unroll
    for (int du = 0; du < 2; ++du) {
      const unsigned cd = live[dv][du] ? code[dv][du] : 2u;
      const int cls = (int)(cd & 3u);
      const double nfd1 = nfd[dv][du][0], nfd2 = nfd[dv][du][1], nfd3 = nfd[dv][du][2];
      double g = odd ? -nfd1 : nfd1;  // wall
      bool write = live[dv][du];
      if (cf_here) {
        double b[1];
        const bool bad = cf_block_interp<1>(K, cd, du, dv, tabs, 0.0, 1.0, b);
        double tmp = 0.0;
        tmp += nfd1 * nc1;
        tmp += nfd2 * nc2;
        tmp += nfd3 * nc3;
        double h = tmp;
        h += b[0] * nc0;
        g = cls == 1 ? h : g;
        nbad_here += (cls == 1 && bad) ? 1 : 0;
      }
      if (cls == 0) {  // (rare) the ghost cell is a valid cell of a neighbouring box
        int q0[3] = {X1[0], X1[1], X1[2]};
        q0[FD] += side ? 1 : -1;
        q0[T0] += du;
        q0[T1] += dv;
        int sbn = -1, qw[3];
        if (classify(L, q0[0], q0[1], q0[2], sbn, qw) == 0 && sbn >= 0) g = MO.data[MO.off[sbn] + fab_index(L.boxes[sbn], MO.ng, MO.ncomp, ncomp0 + FD, qw[0], qw[1], qw[2])];
        else write = false;
      }
      const double f1 = side ? cdiff(L.dxinv[FD], nfd2, nfd1, g) : cdiff(L.dxinv[FD], g, nfd1, nfd2);
      const double t01 = pre ? t01n[dv][du] : cdiff(L.dxinv[T0], a0[dv][du][0], a0[dv][du][1], a0[dv][du][2]);
      const double t11 = pre ? t11n[dv][du] : cdiff(L.dxinv[T1], a1[dv][du][0], a1[dv][du][1], a1[dv][du][2]);
      double k1 = 0.0;
      k1 += (FD == 0) ? f1 : t01;
      k1 += (FD == 1) ? f1 : (FD == 0 ? t01 : t11);
      k1 += (FD == 2) ? f1 : t11;
      k1 = k1 * 0.5;
      if (write) ko[off[dv][du]] = k1;
    }
  if (nbad_here) atomicAdd(nbad, nbad_here);
}

struct LevChunks { const SfChunk* ck[PA_MAXB]; unsigned w0[PA_MAXB + 1]; };  // level l of the batch owns workgroups w0[l] .. w0[l + 1] - 1
// nperim > 0: the first nperim workgroups take the face PERIMETERS' work tables (faces_tab_wg: other cells, the same final normals, long
// chains of dependent loads -- in front, so that they run under the interiors instead of after them)
template <bool PATCH>
__global__ __launch_bounds__(256) void k_faces_fix_chunks(LevBatch<FixArgs> Bt, LevChunks Ck, int* nbad, SlotK sk, unsigned nperim) {
  if (blockIdx.x < nperim) { faces_tab_wg<true, PATCH, false>(Bt, nbad, sk, blockIdx.x); return; }
  const unsigned wx = blockIdx.x - nperim;
  int blev = 0;
  while (blev + 1 < Bt.n && wx >= Ck.w0[blev + 1]) ++blev;
  const FixArgs& Fx = Bt.a[blev];
  const SfChunk D = Ck.ck[blev][wx - Ck.w0[blev]];
  const int dir = D.dir_side >> 1;
  const int e0 = D.hi[0] - D.lo[0] + 1, e1 = D.hi[1] - D.lo[1] + 1, e2 = D.hi[2] - D.lo[2] + 1;
  const int blen = dir == 0 ? e0 : (dir == 1 ? e1 : e2), n0 = dir == 0 ? e1 : e0, n1 = dir == 2 ? e1 : e2;
  const bool cfok = Fx.A.has_crse && Fx.A.ratio == 2 && Fx.use_cp && Fx.L.cp && D.cpoff >= 0;
  const bool straight = blen >= 3 && ((D.flags & PA_SFC_WALL) != 0 || ((D.flags & PA_SFC_FULL) != 0 && cfok));
  const bool mixed = !straight && blen >= 3 && (cfok || !(D.flags & PA_SFC_HAS_CF));
  if (straight) {
    switch (dir) {  // (uniform)
      case 0: fix_chunk_uniform<0>(Fx, D, (int)blockIdx.z, nbad); break;
      case 1: fix_chunk_uniform<1>(Fx, D, (int)blockIdx.z, nbad); break;
      default: fix_chunk_uniform<2>(Fx, D, (int)blockIdx.z, nbad); break;
    }
    return;
  }
  if (mixed) {
    __shared__ double tabs[54];
    cf_tab_to_lds(tabs);
    switch (dir) {
      case 0: fix_chunk_mixed<0>(Fx, D, (int)blockIdx.z, nbad, tabs); break;
      case 1: fix_chunk_mixed<1>(Fx, D, (int)blockIdx.z, nbad, tabs); break;
      default: fix_chunk_mixed<2>(Fx, D, (int)blockIdx.z, nbad, tabs); break;
    }
    return;
  }
  const int hw = D.cw >> 1, sh = 31 - __builtin_clz((unsigned)hw);
  const int u = D.u0 + 2 * ((int)threadIdx.x & (hw - 1)), v = D.v0 + 2 * ((int)threadIdx.x >> sh);
  for (int dv = 0; dv < 2; ++dv)
    for (int du = 0; du < 2; ++du)
      if (u + du >= 0 && v + dv >= 0 && u + du < n0 && v + dv < n1) faces_fast_cell<1, PATCH, false>(Bt, nbad, SlowList(), sk, blev, (unsigned)D.face, (long long)(v + dv) * n0 + (u + du));
}

template <int NL, bool PATCH = false, bool CLIP = false>
__global__ __launch_bounds__(256) void k_faces_curv_fast(LevBatch<FixArgs> Bt, int* nbad, SlowList sl = SlowList(), SlotK sk = SlotK()) {
  faces_fast_wg<NL, PATCH, CLIP>(Bt, nbad, sl, sk, blockIdx.x);
}
// planes per workgroup: 64 vs 128 measured 1.91 vs 1.94 ms with the burst schedule (within noise; more, shorter workgroups).  The tile
// heights (13 / 8 / 4 rows by the boxes' height), the XCD-aware workgroup order (2) and 8-byte stores are what the measurements of
// rounds 1-5 left standing; the 16-byte paired stores, the first marching kernel, tiles of 9-12 rows and the diagnostic builds of the
// sweep went with their switches in round 6 (DESIGN_HISTORY.md lists what each measured).
static int fused_kseg() { return 64; }
static dim3 march_grid(int nx, int ny, int nz, int kseg, int mty, unsigned nboxes) {
  const unsigned tx = (nx + 63) / 64, ty = (ny + mty - 1) / mty, tz = (nz + kseg - 1) / kseg;
  return dim3(tx * ty * tz, nboxes);
}
static int fused_order() { return 2; }
// kname: receives the variant that was launched (what bench.py matches the committed PMC traffic figure against)
template <typename BP>
static void march_launch(hipStream_t st, const BP& bp, int nx, int ny, int nz, unsigned nboxes, const MarchArgs& A0, bool /*pair_ok*/ = false, std::string* kname = nullptr) {
  MarchArgs A = A0;
  // Small levels: a 256^3 level of 64^3 boxes is 320 workgroups at 64 planes each -- 1.25 rounds on 256 CUs -- and
  // ran at 50 % of the HBM figure; shorter z segments give the chip enough workgroups to balance (64 / 32 /
  // 16 / 8 planes on that level: 0.305 / 0.280 / 0.267 / 0.269 ms per launch; on the 512^3 headline level 64 stays best).
  // Round 2: the segment length is chosen from a small model instead of halved -- one workgroup per CU (LDS), so a launch
  // takes about ceil(workgroups / 256) rounds of (planes per segment + ~4 planes of pipeline fill); measured on rank 0's
  // share of the headline (8 boxes of 128^3 per level = 160 tiles): 16 / 22 / 32 / 43 / 64 planes -> 0.266 / 0.272 / 0.280 /
  // 0.254 / 0.309 ms per launch (model: 100 / 104 / 108 / 94 / 136), 16 boxes: 32 planes best (model and measurement).
  {
    const long long per_seg = (long long)((nx + 63) / 64) * ((ny + 12) / 13) * nboxes;
    if (per_seg * ((nz + A.kseg - 1) / A.kseg) < 2048) {
      long long best = -1;
      int best_k = A.kseg;
      for (int tz = 1; tz <= std::max(1, nz / 8); ++tz) {
        const int k = (nz + tz - 1) / tz;
        const long long rounds = (per_seg * ((nz + k - 1) / k) + 255) / 256, cost = rounds * (k + 4);
        if (best < 0 || cost < best) { best = cost; best_k = k; }
      }
      A.kseg = std::max(best_k, 4);
    }
  }
  A.order = fused_order();
  A.nboxes = (int)nboxes;
  const bool clip = A.thr >= 0.0;
  if (nx <= 32) {  // boxes at most 32 cells wide: two rows per wavefront (pa_fused_march3n.h)
    constexpr int NRW = 8;
    const unsigned tiles = (unsigned)(((nx + 31) / 32) * ((ny + 2 * NRW - 1) / (2 * NRW)) * ((nz + A.kseg - 1) / A.kseg));
    A.tiles_max = (int)tiles;
    const dim3 g(tiles * 8u * ((nboxes + 7u) / 8u), 1);
    if (A.cg && clip) hipLaunchKernelGGL((k_gradcurv_march3n<BP, NRW, true, true>), g, dim3(64 * (NRW + 2)), 0, st, bp, A);
    else if (A.cg) hipLaunchKernelGGL((k_gradcurv_march3n<BP, NRW, false, true>), g, dim3(64 * (NRW + 2)), 0, st, bp, A);
    else if (clip) hipLaunchKernelGGL((k_gradcurv_march3n<BP, NRW, true>), g, dim3(64 * (NRW + 2)), 0, st, bp, A);
    else hipLaunchKernelGGL((k_gradcurv_march3n<BP, NRW, false>), g, dim3(64 * (NRW + 2)), 0, st, bp, A);
    if (kname) *kname = std::string("k_gradcurv_march3n<NRW=8,CLIP=") + (clip ? "1" : "0") + ",CG=" + (A.cg ? "1>" : "0>");
    return;
  }
  auto go = [&](auto mc) {  // short boxes do not fill a 13-row tile
    constexpr int M = decltype(mc)::value;
    dim3 g = march_grid(nx, ny, nz, A.kseg, M, nboxes);
    A.txy_max = ((nx + 63) / 64) * ((ny + M - 1) / M);
    A.tiles_max = (int)g.x;
    g = dim3(g.x * 8u * ((nboxes + 7u) / 8u), 1);
    if (kname) *kname = "k_gradcurv_march3<MTY=" + std::to_string(M) + ",CLIP=" + std::to_string((int)clip) + ",PAIR=0,CG=" + std::to_string((int)(A.cg != 0)) + ">";
    if (A.cg && !clip) hipLaunchKernelGGL((k_gradcurv_march3<BP, M, false, false, 0, true>), g, dim3(64 * (M + 3)), 0, st, bp, A);
    else if (A.cg) hipLaunchKernelGGL((k_gradcurv_march3<BP, M, true, false, 0, true>), g, dim3(64 * (M + 3)), 0, st, bp, A);
    else if (clip) hipLaunchKernelGGL((k_gradcurv_march3<BP, M, true>), g, dim3(64 * (M + 3)), 0, st, bp, A);
    else hipLaunchKernelGGL((k_gradcurv_march3<BP, M, false>), g, dim3(64 * (M + 3)), 0, st, bp, A);
  };
  if (ny >= 52) go(std::integral_constant<int, 13>{});
  else if (ny >= 16) go(std::integral_constant<int, 8>{});
  else go(std::integral_constant<int, 4>{});
}

// diagnostic: workgroups of a sweep kernel the runtime's occupancy query admits per CU (which = 0: the wide all-levels sweep, 13 rows;
// 1: the narrow all-levels sweep, 2 x 8 rows).  < 0: the query failed.  NOTE (profiles/r04_wg_residency.txt): the query says 2 for the
// narrow kernel (640 threads, 56 KB of LDS, 80 VGPRs) but per-workgroup clocks show exactly ONE resident per CU; 512-thread variants
// of the same kernel are admitted two per CU.
extern "C" int pa_sweep_occupancy(pa_ctx* ctx, int which) {
  PaBind bind_(ctx);
  if (!ctx) return -1;
  int nb = -1;
  hipError_t e = which == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_gradcurv_march3_levels<13, false>, 64 * 16, 0)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_gradcurv_march3n_levels<8, false>, 64 * 10, 0);
  if (e != hipSuccess) { (void)hipGetLastError(); return -1; }
  return nb;
}

extern "C" int pa_gradcurv_level(pa_ctx* ctx, const pa_mf* phi, int pcomp, double pmin, double pmax, double thr, pa_mf* out, int ocomp) {
  PaBind bind_(ctx);
  if (!ctx || !phi || !out) return pa_fail(ctx, "pa_gradcurv_level: null argument");
  if (phi->lev != out->lev) return pa_fail(ctx, "pa_gradcurv_level: different levels");
  if (phi->ng < 2) return pa_fail(ctx, "pa_gradcurv_level: phi needs >= 2 ghost layers");
  if (pcomp < 0 || pcomp >= phi->ncomp || ocomp < 0 || ocomp + 8 > out->ncomp) return pa_fail(ctx, "pa_gradcurv_level: component range");
  if (!(pmax > pmin)) return pa_fail(ctx, "pa_gradcurv_level: progress variable has no range");
  const pa_level* L = phi->lev;
  if (phi->lev->boxes.empty()) return 0;  // a rank that owns no box of this level
  LevelBP2 bp{L->view, phi->view, out->view};
  MarchArgs A{pcomp, ocomp, fused_kseg(), pmin, 1.0 / (pmax - pmin), thr, 0, 1, 1, 1};
  ProfScope prof(ctx, PA_TAG_GRADCURV);
  march_launch(ctx->stream, bp, L->maxn[0], L->maxn[1], L->maxn[2], (unsigned)L->boxes.size(), A, false, &ctx->sweep_kernel);
  PA_HIP(hipGetLastError());
  return 0;
}

// phase: 1 = recompute the layer-1 normals (k_faces_normal), 2 = curvature of layers 1-2 (needs phase 1 of this
// level AND of the coarser level), 3 = both
int pa_gradcurv_faces_phase(pa_ctx* ctx, const pa_mf* c, int ccomp, const pa_mf* crse_n, int cncomp0, const int32_t bc[3], int ratio, double thr,
                            pa_mf* out, int ncomp0, int kcomp, int phase);
extern "C" int pa_gradcurv_faces_level(pa_ctx* ctx, const pa_mf* c, int ccomp, const pa_mf* crse_n, int cncomp0, const int32_t bc[3],
                                       int ratio, double thr, pa_mf* out, int ncomp0, int kcomp) {
  PaBind bind_(ctx);
  return pa_gradcurv_faces_phase(ctx, c, ccomp, crse_n, cncomp0, bc, ratio, thr, out, ncomp0, kcomp, 3);
}
int pa_gradcurv_faces_phase(pa_ctx* ctx, const pa_mf* c, int ccomp, const pa_mf* crse_n, int cncomp0, const int32_t bc[3], int ratio, double thr,
                            pa_mf* out, int ncomp0, int kcomp, int phase) {
  if (!ctx || !c || !out) return pa_fail(ctx, "pa_gradcurv_faces_level: null argument");
  if (c->lev != out->lev) return pa_fail(ctx, "pa_gradcurv_faces_level: different levels");
  if (c->ng < 2) return pa_fail(ctx, "pa_gradcurv_faces_level: c needs >= 2 ghost layers");
  if (ccomp >= c->ncomp || kcomp >= out->ncomp || ncomp0 + 3 > out->ncomp || (crse_n && cncomp0 + 3 > crse_n->ncomp))
    return pa_fail(ctx, "pa_gradcurv_faces_level: component range");
  if (crse_n && ratio != 2) return pa_fail(ctx, "pa_gradcurv_faces_level: only refinement ratio 2 is supported");
  const pa_level* L = c->lev;
  for (const DBox& B : L->boxes)
    for (int d = 0; d < 3; ++d)
      if (crse_n && B.hi[d] - B.lo[d] + 1 < 3) return pa_fail(ctx, "pa_gradcurv_faces_level: boxes thinner than 3 cells need the pass-by-pass path");
  // sharded coarse level: the three components of the coarse normal from this rank's coarse-source copy
  if (crse_n && (phase & 2) && pa_coarse_source(ctx, L, crse_n, cncomp0, 3, 0, 0, 0, &crse_n, &cncomp0)) return 1;
  FaceArgs A;
  for (int d = 0; d < 3; ++d) A.bc[d] = bc[d];
  A.ratio = ratio; A.has_crse = crse_n ? 1 : 0; A.thr = thr; A.layers = 2; A.perim_only = 0; A.pmin = 0.0; A.invd = 1.0;
  // fast path for the interior of the faces: no threshold clip, every box >= 3 cells thick
  bool fast = !(thr >= 0.0);
  for (const DBox& B : c->lev->boxes)
    for (int d = 0; d < 3; ++d) fast = fast && (B.hi[d] - B.lo[d] + 1 >= 3);
  constexpr int fast_env = 1;
  fast = fast && fast_env;
  if (L->sfaces.empty()) return 0;
  const long long n0 = L->maxn[0], n1 = L->maxn[1], n2 = L->maxn[2];
  const long long nf = std::max(n1 * n2, std::max(n0 * n2, n0 * n1));
  const unsigned nsf = (unsigned)L->sfaces.size();
  ProfScope prof(ctx, PA_TAG_GRADCURV_FACES);
  if (phase & 1) hipLaunchKernelGGL(k_faces_normal, dim3((unsigned)((nf + 255) / 256), nsf), dim3(256), 0, ctx->stream, L->view, c->view, ccomp, out->view, ncomp0, A);
  if (!(phase & 2)) {
    PA_HIP(hipGetLastError());
    return 0;
  }
  LevBatch<FixArgs> Bt;
  Bt.n = 1;
  Bt.ycum[1] = (int)nsf;
  Bt.a[0] = FixArgs{L->view, c->view, ccomp, crse_n ? crse_n->lev->view : L->view, crse_n ? crse_n->view : c->view, cncomp0, out->view, ncomp0, kcomp, A};
  Bt.a[0].wg = (const int2*)L->d_sfwg;
  Bt.a[0].nwg = L->nsfwg;
  if (fast) {
    hipLaunchKernelGGL(k_faces_curv_fast<2>, dim3((unsigned)L->nsfwg), dim3(256), 0, ctx->stream, Bt, ctx->d_flags);
    Bt.a[0].A.perim_only = 1;
  }
  const long long ncell = fast ? 2 * (std::max(n0, std::max(n1, n2)) + std::max(n0, std::max(n1, n2))) : nf;  // perimeter <= 4 * longest edge
  hipLaunchKernelGGL(k_faces_curv<false>, dim3((unsigned)((ncell * A.layers + 255) / 256), nsf), dim3(256), 0, ctx->stream, Bt, ctx->d_flags);
  PA_HIP(hipGetLastError());
  return 0;
}

extern "C" int pa_gradcurv_fab(pa_ctx* ctx, pa_box valid, const pa_fab* phi, int pcomp, double pmin, double pmax, const double dxinv[3],
                               double thr, pa_fab* out, int ocomp) {
  PaBind bind_(ctx);
  if (!ctx || !phi || !out || !dxinv) return pa_fail(ctx, "pa_gradcurv_fab: null argument");
  std::string why;
  if (!fab_covers(*phi, valid, 2, pcomp, 1, why) || !fab_covers(*out, valid, 0, ocomp, 8, why)) return pa_fail(ctx, "pa_gradcurv_fab: " + why);
  if (!(pmax > pmin)) return pa_fail(ctx, "pa_gradcurv_fab: progress variable has no range");
  FabBP2 bp{fab_view(*phi), fab_view(*out), to_dbox(valid), {dxinv[0], dxinv[1], dxinv[2]}};
  MarchArgs A{pcomp, ocomp, fused_kseg(), pmin, 1.0 / (pmax - pmin), thr, 0, 1, 1, 1};
  march_launch(ctx->stream, bp, valid.hi[0] - valid.lo[0] + 1, valid.hi[1] - valid.lo[1] + 1, valid.hi[2] - valid.lo[2] + 1, 1, A);
  PA_HIP(hipGetLastError());
  return 0;
}

// ===================================================================================== exact-normal pipeline
// The sweep with CG (pa_fused_march3.h) takes the progress variable behind special faces from the level's compact
// face-major arrays, so N is final after the sweep and only the curvature of the FIRST layer behind a special face is
// left (its ghost normal is MLMG applyBC on n_d, curvature.cpp:510-531).  Per level: k_prep_faces (+ k_prep_ring) before
// the sweep, k_faces_curv_fast<1> + k_faces_curv<true> after it -- against progress shell + two applyBC launches before
// and three fix-up launches over two layers after it in the first pipeline (kept for the threshold clip, mixed faces and
// narrow boxes).
struct PrepArgs {
  int bc[3];
  int ratio, has_crse;
  double pmin, invd;
};

// Thread per ghost cell of a special face: the face ghost of phi (MLMG applyBC, as k_apply_bc_sfaces) and the resolved
// ghost value of c = the same boundary condition applied to c, whose interior values are (phi - pmin) * invd formed on the
// fly and whose coarse values are the affine view of the coarse phi -- the operations of k_apply_bc_sfaces<2> on a stored c.
struct PrepLev { DLevelView L; DMFView M; int comp; DLevelView LC; DMFView MC; int ccomp; PrepArgs A; int use_cp; long long cg_stride = 0, cp_stride = 0; const int2* wg = nullptr; int nwg = 0; const int* sfboxes = nullptr; };
// PHIONLY (the gradient tool's pass): only the face ghost of phi -- MLMG applyBC as k_apply_bc_sfaces does it, but on the per-face
// work tables and from the coarse PATCHES instead of owner-map lookups into the coarse FABs (the three applyBC launches of a
// 3-level hierarchy took 0.43 ms, this kernel 0.19 ms with the progress variable on top)
template <bool PATCH, bool PHIONLY>
__device__ __forceinline__ void prep_faces_cell(const PrepLev& Pl, int* nbad, const SlotK& sk, unsigned fy, long long t) {
  const DLevelView& L = Pl.L;
  const DMFView& M = Pl.M;
  const DLevelView& LC = Pl.LC;
  const int z = (int)blockIdx.z;  // component slot
  DMFView MC = Pl.MC;
  PrepArgs A = Pl.A;
  if (sk.prog) { A.pmin = MC.xa = sk.prog[2 * z]; A.invd = MC.xb = sk.prog[2 * z + 1]; }
  const int comp = Pl.comp + z, ccomp = Pl.ccomp + z;
  double* const cgz = L.cg + z * Pl.cg_stride;
  const double* const cpz = L.cp ? L.cp + z * Pl.cp_stride : nullptr;
  int b, dir, side, layer, q[3];
  DBox B;
  if (!sface_decode(L, fy, t, 1, b, B, dir, side, q, layer)) return;
  const unsigned code = L.sfcode[L.sfoff[fy] + t];
  const int cls = (int)(code & 3u);
  const int t0 = (dir == 0) ? 1 : 0, t1 = (dir == 2) ? 1 : 2;
  double* cgp = PHIONLY ? nullptr : cgz + L.cgoff[fy] + (long long)(q[t1] - B.lo[t1] + 1) * (B.hi[t0] - B.lo[t0] + 3) + (q[t0] - B.lo[t0] + 1);
  double* p = M.data + M.off[b];
  if (PHIONLY && cls == 0) return;  // a valid cell of the level: FillBoundary's
  if (cls == 0) {
    // a valid cell of the level (a face that is partly coarse-fine, partly covered by a neighbouring box): the progress variable
    // of the cell itself.  Read in the box that OWNS the cell when that box is local -- this kernel runs next to the local
    // FillBoundary, which is what fills the ghost cell -- and in the ghost cell when the owner is another rank's box (the
    // cross-rank exchange has completed on this stream before this launch)
    int sb, xw[3];
    double v;
    if (classify(L, q[0], q[1], q[2], sb, xw) == 0 && sb >= 0) v = M.data[M.off[sb] + fab_index(L.boxes[sb], M.ng, M.ncomp, comp, xw[0], xw[1], xw[2])];
    else v = p[fab_index(B, M.ng, M.ncomp, comp, q[0], q[1], q[2])];
    *cgp = (v - A.pmin) * A.invd;
    return;
  }
  const int s = side ? -1 : 1;
  if (cls == 2) {
    int in[3] = {q[0], q[1], q[2]};
    in[dir] += s;
    const double v = p[fab_index(B, M.ng, M.ncomp, comp, in[0], in[1], in[2])];
    const double vc = (v - A.pmin) * A.invd;
    const bool odd = A.bc[dir] == PA_BC_REFLECT_ODD;
    p[fab_index(B, M.ng, M.ncomp, comp, q[0], q[1], q[2])] = odd ? -v : v;
    if (!PHIONLY) *cgp = odd ? -vc : vc;
    return;
  }
  if (!A.has_crse) { atomicAdd(nbad, 1); return; }
  bool ok = true;
  double coef[4], bv[2];
  const int NX = cf_normal_coef(B.hi[dir] - B.lo[dir] + 1, A.ratio, coef);
  const int xf[2] = {0, 1};
  const long long cpo = (Pl.use_cp && L.cp) ? L.cpoff[fy] : -1;  // wave-uniform
  if (PATCH || cpo >= 0) cf_interp_patch<2>(code, cpz + cpo, B, side, MC, q, dir, xf, ok, bv);
  else cf_interp<2>(code, LC, MC, ccomp, q, dir, A.ratio, xf, ok, bv);
  if (!ok) atomicAdd(nbad, 1);
  double tp = 0.0, tc = 0.0;
  for (int m = 1; m < NX; ++m) {
    int pc[3] = {q[0], q[1], q[2]};
    pc[dir] += s * m;
    const double v = p[fab_index(B, M.ng, M.ncomp, comp, pc[0], pc[1], pc[2])];
    tp += v * coef[m];
    tc += ((v - A.pmin) * A.invd) * coef[m];
  }
  double gp = tp, gc = tc;
  gp += bv[0] * coef[0];
  gc += bv[1] * coef[0];
  p[fab_index(B, M.ng, M.ncomp, comp, q[0], q[1], q[2])] = gp;
  if (!PHIONLY) *cgp = gc;
}

// ---- round 6: the same work from the levels' CHUNK RECORDS (SfChunk, pa_internal.h).  A workgroup takes one record = a rectangle of
// 1024 ghost cells of one face, a thread the 2 x 2 block of ghost cells that share ONE coarse parent.  Chunks of one kind run
// straight-line code: no per-cell code, no code-dependent trip counts, the interpolation weights as literals -- so the 21 loads of
// a thread's four cells (9 coarse values of the parent's 3 x 3 neighbourhood, loaded once instead of four times, + 3 interior cells
// each) are issued together.  Before: one thread per cell behind the chain work table -> sfaces -> boxes -> offsets -> code ->
// weights + patch -> data (a wave lived 13 us, 3/4 of it waiting).  Per cell the operations and their order are those of
// prep_faces_cell / cf_interp_core with code PA_CODE_FULL, so the same bits; mixed chunks take prep_faces_cell itself.
template <int dir, bool PHIONLY>
__device__ __forceinline__ void prep_chunk_uniform(const PrepLev& Pl, const SfChunk& D, const PrepArgs& A, double xa, double xb, int comp, double* cgz, const double* cpz, int* nbad) {
  const int side = D.dir_side & 1;
  constexpr int t0 = dir == 0 ? 1 : 0, t1 = dir == 2 ? 1 : 2;
  const DMFView& M = Pl.M;
  const int n0 = D.hi[t0] - D.lo[t0] + 1, n1 = D.hi[t1] - D.lo[t1] + 1;
  const int hw = D.cw >> 1, sh = 31 - __builtin_clz((unsigned)hw);
  const int u = D.u0 + 2 * ((int)threadIdx.x & (hw - 1)), v = D.v0 + 2 * ((int)threadIdx.x >> sh);
  if (u >= n0 || v >= n1) return;  // (even extents: a block is inside the face or outside it)
  const int ng = M.ng;
  const long long nxg = D.hi[0] - D.lo[0] + 1 + 2 * ng, nyg = D.hi[1] - D.lo[1] + 1 + 2 * ng, nzg = D.hi[2] - D.lo[2] + 1 + 2 * ng;
  double* const p = M.data + M.off[D.box] + (long long)comp * pa_cstride(nxg * nyg * nzg, M.ncomp);
  const long long st[3] = {1, nxg, nxg * nyg};
  int q[3];
  q[dir] = side ? D.hi[dir] + 1 : D.lo[dir] - 1;
  q[t0] = D.lo[t0] + u;
  q[t1] = D.lo[t1] + v;
  const long long iq = ((long long)(q[2] - D.lo[2] + ng) * nyg + (q[1] - D.lo[1] + ng)) * nxg + (q[0] - D.lo[0] + ng);
  const long long sn = side ? -st[dir] : st[dir], s0 = st[t0], s1 = st[t1];
  double* const cgp = PHIONLY ? nullptr : cgz + D.cgoff + (long long)(v + 1) * (n0 + 2) + (u + 1);
  const bool odd = A.bc[dir] == PA_BC_REFLECT_ODD;
  if (D.flags & PA_SFC_WALL) {
    double w[2][2];
#pragma unroll
    for (int dv = 0; dv < 2; ++dv)
#pragma unroll
      for (int du = 0; du < 2; ++du) w[dv][du] = p[iq + dv * s1 + du * s0 + sn];
#pragma unroll
    for (int dv = 0; dv < 2; ++dv)
#pragma unroll
      for (int du = 0; du < 2; ++du) {
        const double x = w[dv][du], xc = (x - A.pmin) * A.invd;
        p[iq + dv * s1 + du * s0] = odd ? -x : x;
        if (!PHIONLY) cgp[dv * (n0 + 2) + du] = odd ? -xc : xc;
      }
    return;
  }
  // PA_SFC_FULL: the coarse parent of the block and its 3 x 3 neighbourhood in the face's coarse patch
  DBox B;
#pragma unroll
  for (int d = 0; d < 3; ++d) { B.lo[d] = D.lo[d]; B.hi[d] = D.hi[d]; }
  int plane, pu0, pv0, pw, ph;
  cpatch_geom(B, dir, side, plane, pu0, pv0, pw, ph);
  const double* const cb = cpz + D.cpoff + (long long)((q[t1] >> 1) - pv0) * pw + ((q[t0] >> 1) - pu0);
  double r[3][3];  // r[a1 + 1][a0 + 1] = coarse(qc + a0 e_t0 + a1 e_t1)
#pragma unroll
  for (int a1 = 0; a1 < 3; ++a1)
#pragma unroll
    for (int a0 = 0; a0 < 3; ++a0) r[a1][a0] = cb[(a1 - 1) * pw + (a0 - 1)];
  double f[2][2][3];  // the three cells behind the face of every ghost cell of the block
#pragma unroll
  for (int dv = 0; dv < 2; ++dv)
#pragma unroll
    for (int du = 0; du < 2; ++du)
#pragma unroll
      for (int m = 0; m < 3; ++m) f[dv][du][m] = p[iq + dv * s1 + du * s0 + (m + 1) * sn];
  bool ok = true;
#pragma unroll
  for (int a1 = 0; a1 < 3; ++a1)
#pragma unroll
    for (int a0 = 0; a0 < 3; ++a0)
      if (__double_as_longlong(r[a1][a0]) == PA_CP_MISSING) { ok = false; r[a1][a0] = 0.0; }
  if (!ok) atomicAdd(nbad, 4);  // counted per ghost cell: the four cells of the block use all nine values
  double rc[3][3];  // the progress variable as the affine view of the coarse phi
#pragma unroll
  for (int a1 = 0; a1 < 3; ++a1)
#pragma unroll
    for (int a0 = 0; a0 < 3; ++a0) rc[a1][a0] = (r[a1][a0] - xa) * xb;
  constexpr double nc0 = k_cf_coef.nrm[4][0], nc1 = k_cf_coef.nrm[4][1], nc2 = k_cf_coef.nrm[4][2], nc3 = k_cf_coef.nrm[4][3];
#pragma unroll
  for (int dv = 0; dv < 2; ++dv)
#pragma unroll
    for (int du = 0; du < 2; ++du) {
      // InterpBndryData, cf_interp_core with lo = -1, hi = 1 in both directions and the cross term, child (du, dv) of the parent
      const double c00 = du ? k_cf_coef.tan[1][1][1][0] : k_cf_coef.tan[0][1][1][0], c01 = du ? k_cf_coef.tan[1][1][1][1] : k_cf_coef.tan[0][1][1][1],
                   c02 = du ? k_cf_coef.tan[1][1][1][2] : k_cf_coef.tan[0][1][1][2];
      const double c10 = dv ? k_cf_coef.tan[1][1][1][0] : k_cf_coef.tan[0][1][1][0], c11 = dv ? k_cf_coef.tan[1][1][1][1] : k_cf_coef.tan[0][1][1][1],
                   c12 = dv ? k_cf_coef.tan[1][1][1][2] : k_cf_coef.tan[0][1][1][2];
      const double xi0 = du ? 0.25 : -0.25, xi1 = dv ? 0.25 : -0.25;
      double b0 = 0.0, b1 = 0.0;
      b0 += c00 * r[1][0];  b1 += c00 * rc[1][0];
      b0 += c01 * r[1][1];  b1 += c01 * rc[1][1];
      b0 += c02 * r[1][2];  b1 += c02 * rc[1][2];
      b0 += c10 * r[0][1];  b1 += c10 * rc[0][1];
      b0 += c11 * r[1][1];  b1 += c11 * rc[1][1];
      b0 += c12 * r[2][1];  b1 += c12 * rc[2][1];
      b0 -= r[1][1];        b1 -= rc[1][1];
      b0 += ((xi0 * xi1) * 0.25) * (((r[2][2] - r[2][0]) + r[0][0]) - r[0][2]);
      b1 += ((xi0 * xi1) * 0.25) * (((rc[2][2] - rc[2][0]) + rc[0][0]) - rc[0][2]);
      // MLMG applyBC across the face: points {-1 (the interpolated value), 0.5, 1.5, 2.5} seen from -0.5
      double tp = 0.0, tc = 0.0;
      tp += f[dv][du][0] * nc1;  tc += ((f[dv][du][0] - A.pmin) * A.invd) * nc1;
      tp += f[dv][du][1] * nc2;  tc += ((f[dv][du][1] - A.pmin) * A.invd) * nc2;
      tp += f[dv][du][2] * nc3;  tc += ((f[dv][du][2] - A.pmin) * A.invd) * nc3;
      double gp = tp, gc = tc;
      gp += b0 * nc0;
      gc += b1 * nc0;
      p[iq + dv * s1 + du * s0] = gp;
      if (!PHIONLY) cgp[dv * (n0 + 2) + du] = gc;
    }
}


// Any mix of cell kinds in the chunk (coarse-fine with any stencil, wall, valid cells of the level behind a partly covered face),
// block origins on even GLOBAL indices (u0 / v0 may be -1: cells outside the face are predicated off).  Everything a block may need
// is requested up front -- four codes, the 13 coarse values, three interior cells per ghost cell -- then each cell's value is chosen
// by selects; only a valid ghost cell (its value lives in the box that owns it: an owner-map lookup) takes a branch.
template <int dir, bool PHIONLY>
__device__ __forceinline__ void prep_chunk_mixed(const PrepLev& Pl, const SfChunk& D, const PrepArgs& A, double xa, double xb, int comp, double* cgz, const double* cpz,
                                                 int* nbad, const double* tabs) {
  const int side = D.dir_side & 1;
  constexpr int t0 = dir == 0 ? 1 : 0, t1 = dir == 2 ? 1 : 2;
  const DMFView& M = Pl.M;
  const DLevelView& L = Pl.L;
  const int n0 = D.hi[t0] - D.lo[t0] + 1, n1 = D.hi[t1] - D.lo[t1] + 1;
  const int hw = D.cw >> 1, sh = 31 - __builtin_clz((unsigned)hw);
  const int u = D.u0 + 2 * ((int)threadIdx.x & (hw - 1)), v = D.v0 + 2 * ((int)threadIdx.x >> sh);
  if (u >= n0 || v >= n1) return;
  const int ng = M.ng;
  const long long nxg = D.hi[0] - D.lo[0] + 1 + 2 * ng, nyg = D.hi[1] - D.lo[1] + 1 + 2 * ng, nzg = D.hi[2] - D.lo[2] + 1 + 2 * ng;
  double* const p = M.data + M.off[D.box] + (long long)comp * pa_cstride(nxg * nyg * nzg, M.ncomp);
  const long long st[3] = {1, nxg, nxg * nyg};
  int q[3];
  q[dir] = side ? D.hi[dir] + 1 : D.lo[dir] - 1;
  q[t0] = D.lo[t0] + u;
  q[t1] = D.lo[t1] + v;
  const long long iq = ((long long)(q[2] - D.lo[2] + ng) * nyg + (q[1] - D.lo[1] + ng)) * nxg + (q[0] - D.lo[0] + ng);
  const long long sn = side ? -st[dir] : st[dir], s0 = st[t0], s1 = st[t1];
  double* const cgp = PHIONLY ? nullptr : cgz + D.cgoff + (long long)(v + 1) * (n0 + 2) + (u + 1);
  const bool odd = A.bc[dir] == PA_BC_REFLECT_ODD;
  bool in[2][2];
  long long off[2][2];
  unsigned code[2][2];
#pragma unroll
  for (int dv = 0; dv < 2; ++dv)
#pragma unroll
    for (int du = 0; du < 2; ++du) {
      const int uu = u + du, vv = v + dv;
      in[dv][du] = uu >= 0 && uu < n0 && vv >= 0 && vv < n1;
      const int uc = min(max(uu, 0), n0 - 1), vc = min(max(vv, 0), n1 - 1);
      off[dv][du] = (long long)(vc - v) * s1 + (long long)(uc - u) * s0;
      code[dv][du] = L.sfcode[D.sfoff + (long long)vc * n0 + uc];
    }
  const bool cf_here = (D.flags & PA_SFC_HAS_CF) != 0;  // (uniform) a chunk without coarse-fine cells may belong to a face without a patch
  CfBlock K;
  if (cf_here) {
    DBox B;
#pragma unroll
    for (int d = 0; d < 3; ++d) { B.lo[d] = D.lo[d]; B.hi[d] = D.hi[d]; }
    int plane, pu0, pv0, pw, ph;
    cpatch_geom(B, dir, side, plane, pu0, pv0, pw, ph);
    cf_block_load(cpz + D.cpoff + (long long)((q[t1] >> 1) - pv0) * pw + ((q[t0] >> 1) - pu0), pw, K);
  }
  double f[2][2][3];
#pragma unroll
  for (int dv = 0; dv < 2; ++dv)
#pragma unroll
    for (int du = 0; du < 2; ++du)
#pragma unroll
      for (int m = 0; m < 3; ++m) f[dv][du][m] = p[iq + off[dv][du] + (m + 1) * sn];
  if (cf_here) cf_block_finish(K);
  constexpr double nc0 = k_cf_coef.nrm[4][0], nc1 = k_cf_coef.nrm[4][1], nc2 = k_cf_coef.nrm[4][2], nc3 = k_cf_coef.nrm[4][3];
  int nbad_here = 0;
  bool any_valid = false;
#pragma unroll
  for (int dv = 0; dv < 2; ++dv)
#pragma unroll
    for (int du = 0; du < 2; ++du) {
      const unsigned cd = in[dv][du] ? code[dv][du] : 0u;
      const int cls = (int)(cd & 3u);
      any_valid = any_valid || (in[dv][du] && cls == 0);
      double gp, gc;
      {  // wall: the mirror image of the first cell
        const double x = f[dv][du][0], xc = (x - A.pmin) * A.invd;
        gp = odd ? -x : x;
        gc = odd ? -xc : xc;
      }
      if (cf_here) {  // coarse-fine: InterpBndryData + MLMG applyBC across the face (both formed, chosen by the class)
        double b[2];
        const bool bad = cf_block_interp<2>(K, cd, du, dv, tabs, xa, xb, b);
        double tp = 0.0, tc = 0.0;
        tp += f[dv][du][0] * nc1;  tc += ((f[dv][du][0] - A.pmin) * A.invd) * nc1;
        tp += f[dv][du][1] * nc2;  tc += ((f[dv][du][1] - A.pmin) * A.invd) * nc2;
        tp += f[dv][du][2] * nc3;  tc += ((f[dv][du][2] - A.pmin) * A.invd) * nc3;
        double hp = tp, hc = tc;
        hp += b[0] * nc0;
        hc += b[1] * nc0;
        gp = cls == 1 ? hp : gp;
        gc = cls == 1 ? hc : gc;
        nbad_here += (cls == 1 && bad) ? 1 : 0;
      }
      if (in[dv][du] && cls != 0) {
        p[iq + off[dv][du]] = gp;
        if (!PHIONLY) cgp[dv * (n0 + 2) + du] = gc;
      }
    }
  if (nbad_here) atomicAdd(nbad, nbad_here);
  if (!PHIONLY && any_valid) {
    // a valid cell of the level behind a partly covered face: the progress variable of the cell itself, read in the box that OWNS it
    // when that box is local (this kernel runs next to the local FillBoundary), in the ghost cell when another rank owns it
#pragma unroll
    for (int dv = 0; dv < 2; ++dv)
#pragma unroll
      for (int du = 0; du < 2; ++du) {
        if (!(in[dv][du] && (code[dv][du] & 3u) == 0u)) continue;
        int qq[3] = {q[0], q[1], q[2]};
        qq[t0] += du;
        qq[t1] += dv;
        int sb, xw[3];
        double x;
        if (classify(L, qq[0], qq[1], qq[2], sb, xw) == 0 && sb >= 0) x = M.data[M.off[sb] + fab_index(L.boxes[sb], M.ng, M.ncomp, comp, xw[0], xw[1], xw[2])];
        else x = p[iq + off[dv][du]];
        cgp[dv * (n0 + 2) + du] = (x - A.pmin) * A.invd;
      }
  }
}

// The edge ghost cells of c (outside the box in two directions a < c) that are the boundary ghost of a valid cell of a
// NEIGHBOURING box (k_apply_bc_edges): stored in the ring of the special face they continue.  WHICH edge ghost cells those are, the
// face they continue and their interpolation masks depend on the level's geometry only: a list built once per level
// (k_find_ring; until round 6 every pass re-derived it from three owner-map classifications per edge ghost cell of every box with a
// special face, a 40-us chain of dependent lookups for a few thousand values).  The values read ghost cells of phi that are valid
// cells of the level: from this box's FAB once FillBoundary has filled them (DIRECT = false), or -- DIRECT, an unsharded level -- in
// the box that owns them, so that the work does not wait for FillBoundary and runs next to it.
struct RingItem { int b, q[3], w /* dir | side << 2 | class << 3 */, ef, code, pad; };
__global__ __launch_bounds__(256) void k_find_ring(DLevelView L, const int* sfboxes, int nsfboxes, RingItem* items, int* count, int cap) {
  const int b = sfboxes[blockIdx.y];
  const DBox B = L.boxes[b];
  const int n[3] = {B.hi[0] - B.lo[0] + 1, B.hi[1] - B.lo[1] + 1, B.hi[2] - B.lo[2] + 1};
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  int e = -1, which = 0, pos = 0;
  for (int d = 0; d < 3; ++d) {
    if (t < 4LL * n[d]) { e = d; which = (int)((unsigned)t / (unsigned)n[d]); pos = (int)((unsigned)t % (unsigned)n[d]); break; }
    t -= 4LL * n[d];
  }
  if (e < 0) return;
  const int a = (e == 0) ? 1 : 0, c = (e == 2) ? 1 : 2;
  const int sa = which & 1, sc = which >> 1;
  int q[3];
  q[e] = B.lo[e] + pos;
  q[a] = sa ? B.hi[a] + 1 : B.lo[a] - 1;
  q[c] = sc ? B.hi[c] + 1 : B.lo[c] - 1;
  const int cls = classify(L, q[0], q[1], q[2]);
  if (cls == 0) return;
  int qa[3] = {q[0], q[1], q[2]}, qc[3] = {q[0], q[1], q[2]};
  qa[a] += sa ? -1 : 1;
  qc[c] += sc ? -1 : 1;
  const bool va = classify(L, qa[0], qa[1], qa[2]) == 0;
  const bool vc = classify(L, qc[0], qc[1], qc[2]) == 0;
  if (va == vc) return;
  const int dir = va ? a : c, sd = va ? sa : sc;
  if (cls == 2 && !((q[dir] < L.domlo[dir] || q[dir] > L.domhi[dir]) && !L.is_per[dir])) return;
  const int ef = L.sfindex[b * 6 + dir * 2 + sd];
  if (ef < 0) return;
  const int i = atomicAdd(count, 1);
  if (items && i < cap) {
    RingItem R;
    R.b = b; R.q[0] = q[0]; R.q[1] = q[1]; R.q[2] = q[2]; R.w = dir | (sd << 2) | (cls << 3); R.ef = ef;
    R.code = cls == 1 ? (int)(cf_masks(L, q, dir, 2) | 1u) : 0;
    R.pad = 0;
    items[i] = R;
  }
}

template <bool PATCH, bool DIRECT>
__device__ __forceinline__ void prep_ring_item(const PrepLev& Pl, const RingItem& R, int* nbad, const SlotK& sk) {
  const DLevelView& L = Pl.L;
  const DMFView& M = Pl.M;
  const DLevelView& LC = Pl.LC;
  const int z = (int)blockIdx.z;  // component slot
  DMFView MC = Pl.MC;
  PrepArgs A = Pl.A;
  if (sk.prog) { A.pmin = MC.xa = sk.prog[2 * z]; A.invd = MC.xb = sk.prog[2 * z + 1]; }
  const int comp = Pl.comp + z, ccomp = Pl.ccomp + z;
  double* const cgz = L.cg + z * Pl.cg_stride;
  const double* const cpz = L.cp ? L.cp + z * Pl.cp_stride : nullptr;
  const int b = R.b, dir = R.w & 3, sd = (R.w >> 2) & 1, cls = R.w >> 3, ef = R.ef;
  const int q[3] = {R.q[0], R.q[1], R.q[2]};
  const DBox B = L.boxes[b];
  const int n[3] = {B.hi[0] - B.lo[0] + 1, B.hi[1] - B.lo[1] + 1, B.hi[2] - B.lo[2] + 1};
  const int s = sd ? -1 : 1;
  if (cls == 1 && !A.has_crse) { atomicAdd(nbad, 1); return; }
  const double* p = M.data + M.off[b];
  auto phi_at = [&](const int x[3]) -> double {  // a cell one row / plane outside this box that is a valid cell of the level
    if (DIRECT) {
      int sb, xw[3];
      if (classify(L, x[0], x[1], x[2], sb, xw) == 0 && sb >= 0) return M.data[M.off[sb] + fab_index(L.boxes[sb], M.ng, M.ncomp, comp, xw[0], xw[1], xw[2])];
    }
    return p[fab_index(B, M.ng, M.ncomp, comp, x[0], x[1], x[2])];
  };
  double g;
  if (cls == 2) {
    int in[3] = {q[0], q[1], q[2]};
    in[dir] += s;
    const double v = (phi_at(in) - A.pmin) * A.invd;
    g = (A.bc[dir] == PA_BC_REFLECT_ODD) ? -v : v;
  } else {
    bool ok = true;
    double coef[4];
    const int NX = cf_normal_coef(n[dir], A.ratio, coef);
    double bv;
    const long long cpo = (Pl.use_cp && L.cp) ? L.cpoff[ef] : -1;
    if (PATCH || cpo >= 0) {  // the coarse values from the face's patch (its ring of two coarse cells covers the edge ghosts)
      const int xf[1] = {MC.xform};
      double b1[1];
      cf_interp_patch<1>((unsigned)R.code, cpz + cpo, B, sd, MC, q, dir, xf, ok, b1);
      bv = b1[0];
    } else {
      const int xf[1] = {MC.xform};
      double b1[1];
      cf_interp<1>((unsigned)R.code, LC, MC, ccomp, q, dir, A.ratio, xf, ok, b1);  // MC carries the affine view
      bv = b1[0];
    }
    if (!ok) atomicAdd(nbad, 1);
    double tmp = 0.0;
    for (int m = 1; m < NX; ++m) {
      int pc[3] = {q[0], q[1], q[2]};
      pc[dir] += s * m;
      tmp += ((phi_at(pc) - A.pmin) * A.invd) * coef[m];
    }
    g = tmp;
    g += bv * coef[0];
  }
  const int t0 = (dir == 0) ? 1 : 0, t1 = (dir == 2) ? 1 : 2;
  cgz[L.cgoff[ef] + (long long)(q[t1] - B.lo[t1] + 1) * (B.hi[t0] - B.lo[t0] + 3) + (q[t0] - B.lo[t0] + 1)] = g;
}
struct LevRings { const RingItem* it[PA_MAXB]; unsigned n[PA_MAXB]; unsigned w0[PA_MAXB + 1]; };  // level l of the batch owns workgroups w0[l] .. w0[l + 1] - 1
template <bool PATCH, bool DIRECT>
__device__ __forceinline__ void prep_ring_wg(const LevBatch<PrepLev>& Bt, const LevRings& Rg, int* nbad, const SlotK& sk, unsigned w) {
  int blev = 0;
  while (blev + 1 < Bt.n && w >= Rg.w0[blev + 1]) ++blev;
  const unsigned i = (w - Rg.w0[blev]) * 256u + threadIdx.x;
  if (i >= Rg.n[blev]) return;
  prep_ring_item<PATCH, DIRECT>(Bt.a[blev], Rg.it[blev][i], nbad, sk);
}
template <bool PATCH, bool DIRECT = false>
__global__ __launch_bounds__(256) void k_prep_ring(LevBatch<PrepLev> Bt, LevRings Rg, int* nbad, SlotK sk = SlotK()) {
  prep_ring_wg<PATCH, DIRECT>(Bt, Rg, nbad, sk, blockIdx.x);
}

template <bool PATCH, bool PHIONLY = false>
__global__ __launch_bounds__(256) void k_prep_faces_chunks(LevBatch<PrepLev> Bt, LevChunks Ck, LevRings Rg, int* nbad, SlotK sk = SlotK()) {
  // the levels' ring items in front (Rg.w0[Bt.n] workgroups, none when the ring has its own launch): independent of the faces -- they
  // read valid cells in the boxes that own them -- and a longer chain of dependent loads, so they run under the faces' workgroups
  const unsigned nrw = Rg.w0[Bt.n];
  if (blockIdx.x < nrw) { if (!PHIONLY) prep_ring_wg<PATCH, true>(Bt, Rg, nbad, sk, blockIdx.x); return; }
  const unsigned wx = blockIdx.x - nrw;
  int blev = 0;
  while (blev + 1 < Bt.n && wx >= Ck.w0[blev + 1]) ++blev;
  const PrepLev& Pl = Bt.a[blev];
  const SfChunk D = Ck.ck[blev][wx - Ck.w0[blev]];
  const int dir = D.dir_side >> 1;
  // (selects, not D.lo[dir]: a run-time index would send the record through scratch)
  const int e0 = D.hi[0] - D.lo[0] + 1, e1 = D.hi[1] - D.lo[1] + 1, e2 = D.hi[2] - D.lo[2] + 1;
  const int blen = dir == 0 ? e0 : (dir == 1 ? e1 : e2), n0 = dir == 0 ? e1 : e0, n1 = dir == 2 ? e1 : e2;
  const bool cfok = Pl.A.has_crse && Pl.A.ratio == 2 && Pl.use_cp && Pl.L.cp && D.cpoff >= 0 && blen >= 3;  // coarse-fine cells: from the face's coarse patch, four points across the face
  const bool straight = (D.flags & PA_SFC_WALL) != 0 || ((D.flags & PA_SFC_FULL) != 0 && cfok);
  const bool mixed = !straight && blen >= 3 && (cfok || !(D.flags & PA_SFC_HAS_CF));
  if (straight || mixed) {
    __shared__ double tabs[54];
    if (mixed) cf_tab_to_lds(tabs);  // (uniform per workgroup; before any thread leaves)
    const int z = (int)blockIdx.z;  // component slot
    PrepArgs A = Pl.A;
    double xa = Pl.MC.xa, xb = Pl.MC.xb;
    if (sk.prog) { A.pmin = xa = sk.prog[2 * z]; A.invd = xb = sk.prog[2 * z + 1]; }
    double* const cgz = PHIONLY ? nullptr : Pl.L.cg + z * Pl.cg_stride;
    const double* const cpz = Pl.L.cp ? Pl.L.cp + z * Pl.cp_stride : nullptr;
    if (straight) {
      switch (dir) {  // (uniform) compile-time directions: every index into lo / hi / strides is a constant
        case 0: prep_chunk_uniform<0, PHIONLY>(Pl, D, A, xa, xb, Pl.comp + z, cgz, cpz, nbad); break;
        case 1: prep_chunk_uniform<1, PHIONLY>(Pl, D, A, xa, xb, Pl.comp + z, cgz, cpz, nbad); break;
        default: prep_chunk_uniform<2, PHIONLY>(Pl, D, A, xa, xb, Pl.comp + z, cgz, cpz, nbad); break;
      }
    } else {
      switch (dir) {
        case 0: prep_chunk_mixed<0, PHIONLY>(Pl, D, A, xa, xb, Pl.comp + z, cgz, cpz, nbad, tabs); break;
        case 1: prep_chunk_mixed<1, PHIONLY>(Pl, D, A, xa, xb, Pl.comp + z, cgz, cpz, nbad, tabs); break;
        default: prep_chunk_mixed<2, PHIONLY>(Pl, D, A, xa, xb, Pl.comp + z, cgz, cpz, nbad, tabs); break;
      }
    }
    return;
  }
  // what is left: a level that interpolates through the owner map (no coarse patches), boxes thinner than three cells, a level
  // without a coarser one that has coarse-fine cells (counted as errors): cell by cell
  const int hw = D.cw >> 1, sh = 31 - __builtin_clz((unsigned)hw);
  const int u = D.u0 + 2 * ((int)threadIdx.x & (hw - 1)), v = D.v0 + 2 * ((int)threadIdx.x >> sh);
  for (int dv = 0; dv < 2; ++dv)
    for (int du = 0; du < 2; ++du)
      if (u + du >= 0 && v + dv >= 0 && u + du < n0 && v + dv < n1) prep_faces_cell<PATCH, PHIONLY>(Pl, nbad, sk, (unsigned)D.face, (long long)(v + dv) * n0 + (u + du));
}

// the level's compact ghost arrays, allocated on first use (a cache of the level object)
static int level_cg(pa_ctx* ctx, const pa_level* Lc, int nsets = 1) {
  pa_level* L = const_cast<pa_level*>(Lc);
  if (L->d_cg && L->cg_sets >= nsets) return 0;
  if (L->d_cg) {  // more component slots than before: a larger buffer (every pass rewrites what it reads)
    PA_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->stream2) PA_HIP(hipStreamSynchronize(ctx->stream2));
    (void)hipFree(L->d_cg);
    L->d_cg = nullptr;
  }
  const size_t n = (size_t)std::max<long long>(L->cg_total, 8) * (size_t)nsets;
  PA_HIP(hipMalloc(&L->d_cg, sizeof(double) * n));
  PA_HIP(hipMemsetAsync(L->d_cg, 0, sizeof(double) * n, ctx->stream));
  L->cg_sets = nsets;
  L->view.cg = L->d_cg;
  return 0;
}
static long long cg_stride(const pa_level* L) { return std::max<long long>(L->cg_total, 8); }
// NCG arrays of a level (MarchArgs::ncg): three arrays of pairs, each shaped like one set of the compact ghost arrays, allocated on first use
static int level_ncg(pa_ctx* ctx, const pa_level* Lc) {
  pa_level* L = const_cast<pa_level*>(Lc);
  if (L->d_ncg) return 0;
  const size_t n = 6 * (size_t)std::max<long long>(L->cg_total, 8);  // three arrays of pairs
  if (hipMalloc(&L->d_ncg, sizeof(double) * n) != hipSuccess) { L->d_ncg = nullptr; return pa_fail(ctx, "compact first-layer arrays: device allocation failed"); }
  PA_HIP(hipMemsetAsync(L->d_ncg, 0, sizeof(double) * n, ctx->stream));
  return 0;
}
static long long cp_stride(const pa_level* L) { return std::max<long long>(L->cp_total, 8); }
// the level's ring items (k_find_ring), built on first use: count, then fill; sorted by (face, position) so that neighbouring threads
// touch neighbouring cells
static int level_ring(pa_ctx* ctx, const pa_level* Lc) {
  pa_level* L = const_cast<pa_level*>(Lc);
  if (L->nring >= 0) return 0;
  if (L->nsfboxes == 0 || L->sfaces.empty()) { L->nring = 0; return 0; }
  int* d_count = nullptr;
  auto fail = [&](const char* what) {
    if (d_count) (void)hipFree(d_count);
    if (L->d_ring) { (void)hipFree(L->d_ring); L->d_ring = nullptr; }
    (void)hipGetLastError();
    return pa_fail(ctx, std::string("ring list: ") + what);
  };
  if (hipMalloc(&d_count, sizeof(int)) != hipSuccess) return fail("device allocation failed");
  const long long nt = 4LL * (L->maxn[0] + L->maxn[1] + L->maxn[2]);
  int n = 0;
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      if (n == 0) break;
      if (hipMalloc(&L->d_ring, sizeof(RingItem) * (size_t)n) != hipSuccess) { L->d_ring = nullptr; return fail("device allocation failed"); }
    }
    if (hipMemsetAsync(d_count, 0, sizeof(int), ctx->stream) != hipSuccess) return fail("memset failed");
    for (int y0 = 0; y0 < L->nsfboxes; y0 += 65535)
      hipLaunchKernelGGL(k_find_ring, dim3((unsigned)((nt + 255) / 256), (unsigned)std::min(65535, L->nsfboxes - y0)), dim3(256), 0, ctx->stream, L->view, L->d_sfboxes + y0,
                         L->nsfboxes - y0, pass ? (RingItem*)L->d_ring : nullptr, d_count, n);
    int m = 0;
    if (hipMemcpyAsync(&m, d_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return fail("reading the count failed");
    if (pass == 1 && m != n) return fail("the two passes disagree");
    n = m;
  }
  (void)hipFree(d_count);
  d_count = nullptr;
  if (n > 1) {
    std::vector<RingItem> h((size_t)n);
    if (hipMemcpy(h.data(), L->d_ring, sizeof(RingItem) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return fail("download failed");
    std::sort(h.begin(), h.end(), [](const RingItem& a, const RingItem& b) {
      if (a.ef != b.ef) return a.ef < b.ef;
      if (a.q[2] != b.q[2]) return a.q[2] < b.q[2];
      if (a.q[1] != b.q[1]) return a.q[1] < b.q[1];
      return a.q[0] < b.q[0];
    });
    if (hipMemcpy(L->d_ring, h.data(), sizeof(RingItem) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) return fail("upload failed");
  }
  L->nring = n;
  return 0;
}


// ---- coarse patches (DLevelView::cp, pa_internal.h): one thread per patch cell fetches the coarse value through the owner
// map of the coarse level (or of this rank's coarse-source copy) -- 1/4 of the fine face cells, once, instead of every
// fine ghost cell walking owner map -> box -> offset before its 5-11 coarse loads.  by_dir: the patch of a face of
// direction d holds component ccomp + d (the coarse normal the fix-up needs), else component ccomp for every face.
struct CpLev { DLevelView L; DLevelView LC; DMFView MC; int ccomp, by_dir; long long cp_stride = 0; int zstride = 1; };  // slot z: component ccomp + zstride z, patches + cp_stride z
__global__ __launch_bounds__(256) void k_cpatch(LevBatch<CpLev> Bt) {
  unsigned fy;
  const CpLev& P = Bt.a[Bt.find(blockIdx.y, fy)];
  const DLevelView& L = P.L;
  const long long off = L.cpoff[fy];
  if (off < 0) return;  // a wall face
  const int e = L.sfaces[fy], dir = (e % 6) >> 1, side = e & 1;
  const DBox B = L.boxes[e / 6];
  int plane, u0, v0, pw, ph;
  cpatch_geom(B, dir, side, plane, u0, v0, pw, ph);
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  if (t >= (unsigned)(pw * ph)) return;
  const int t0 = dir == 0 ? 1 : 0, t1 = dir == 2 ? 1 : 2;
  const unsigned r = t / (unsigned)pw;
  int p[3];
  p[dir] = plane; p[t0] = u0 + (int)(t - r * (unsigned)pw); p[t1] = v0 + (int)r;
  double v = __longlong_as_double(PA_CP_MISSING);
  const int z = (int)blockIdx.z;
  if (wrap_cell(P.LC, p)) {
    const int cb = owner_of(P.LC, p);
    if (cb >= 0) v = P.MC.data[P.MC.off[cb] + fab_index(P.LC.boxes[cb], P.MC.ng, P.MC.ncomp, P.ccomp + P.zstride * z + (P.by_dir ? dir : 0), p[0], p[1], p[2])];
  }
  L.cp[z * P.cp_stride + off + t] = v;
}
// the same gather from the plan of copy regions (pa_dist.h: CpPlan): no owner-map lookups at all
struct CprLev { DLevelView L; DLevelView LC; DMFView MC; const int* regs; const int* wgs; int nwg, ccomp, by_dir; long long cp_stride = 0; int zstride = 1; };
__global__ __launch_bounds__(256) void k_cpatch_regions(LevBatch<CprLev> Bt) {
  const CprLev& P = Bt.a[blockIdx.y];
  if ((int)blockIdx.x >= P.nwg) return;
  const int* R = P.regs + 12 * P.wgs[2 * blockIdx.x];
  const unsigned t = (unsigned)P.wgs[2 * blockIdx.x + 1] * 256u + threadIdx.x, nu = (unsigned)R[7];
  if (t >= nu * (unsigned)R[8]) return;
  const unsigned b = nu == 1 ? t : __umulhi(t, (unsigned)R[10]), a = t - b * nu;
  const int dir = R[9], t0 = dir == 0 ? 1 : 0, t1 = dir == 2 ? 1 : 2;
  const DLevelView& L = P.L;
  const int e = L.sfaces[R[0]], side = e & 1;
  int plane, u0, v0, pw, ph;
  cpatch_geom(L.boxes[e / 6], dir, side, plane, u0, v0, pw, ph);
  int p[3] = {R[4], R[5], R[6]};
  p[t0] += (int)a; p[t1] += (int)b;
  const int z = (int)blockIdx.z;
  L.cp[z * P.cp_stride + L.cpoff[R[0]] + (long long)(R[3] + (int)b) * pw + (R[2] + (int)a)] =
      P.MC.data[P.MC.off[R[1]] + fab_index(P.LC.boxes[R[1]], P.MC.ng, P.MC.ncomp, P.ccomp + P.zstride * z + (P.by_dir ? dir : 0), p[0], p[1], p[2])];
}
__global__ __launch_bounds__(256) void k_cpatch_clear(double* cp, long long n) {
  const long long t = blockIdx.x * 256LL + threadIdx.x;
  if (t < n) cp[t] = __longlong_as_double(PA_CP_MISSING);
}
static int level_cp(pa_ctx* ctx, const pa_level* Lc, int nsets = 1) {
  pa_level* L = const_cast<pa_level*>(Lc);
  if (L->d_cp && L->cp_sets >= nsets) return 0;
  if (L->d_cp) {
    PA_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->stream2) PA_HIP(hipStreamSynchronize(ctx->stream2));
    (void)hipFree(L->d_cp);
    L->d_cp = nullptr;
  }
  const long long n = cp_stride(L) * nsets;
  PA_HIP(hipMalloc(&L->d_cp, sizeof(double) * (size_t)n));
  hipLaunchKernelGGL(k_cpatch_clear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, L->d_cp, n);  // cells without a coarse owner stay "missing"
  L->cp_sets = nsets;
  L->view.cp = L->d_cp;
  return 0;
}
static bool cpatch_on() { return true; }  // (the face kernels' owner-map interpolation is what levels without patches -- none today -- would take)
// gather the patches of levels [l0, l1) (those that have a coarse source): one launch
static int cpatch_launch(pa_ctx* ctx, int l0, int l1, pa_mf* const* fine, const pa_mf* const* crse, int ccomp, int by_dir, int nslots = 1, int zstride = 1) {
  {  // copy regions when every level of the batch has a plan
    LevBatch<CprLev> Br;
    bool regions = true;
    int mw = 0;
    for (int l = l0; l < l1 && regions; ++l) {
      const pa_level* L = fine[l]->lev;
      if (!crse[l] || L->boxes.empty() || L->sfaces.empty() || L->cp_total == 0) continue;
      if (level_cp(ctx, L, nslots)) return 1;
      const CpPlan* P = pa_cp_plan(ctx, L, crse[l]->lev);
      regions = P && P->ok;
      if (!regions || P->nwg == 0) continue;
      Br.a[Br.n] = CprLev{L->view, crse[l]->lev->view, crse[l]->view, P->d_regs, P->d_wgs, P->nwg, ccomp, by_dir, cp_stride(L), zstride};
      ++Br.n;
      mw = std::max(mw, P->nwg);
    }
    if (regions) {
      if (Br.n) hipLaunchKernelGGL(k_cpatch_regions, dim3((unsigned)mw, (unsigned)Br.n, (unsigned)nslots), dim3(256), 0, ctx->stream, Br);
      return 0;
    }
  }
  LevBatch<CpLev> Bt;
  long long mp = 0;
  for (int l = l0; l < l1; ++l) {
    const pa_level* L = fine[l]->lev;
    if (!crse[l] || L->boxes.empty() || L->sfaces.empty() || L->cp_total == 0) continue;
    if (level_cp(ctx, L, nslots)) return 1;
    Bt.a[Bt.n] = CpLev{L->view, crse[l]->lev->view, crse[l]->view, ccomp, by_dir, cp_stride(L), zstride};
    Bt.ycum[Bt.n + 1] = Bt.ycum[Bt.n] + (int)L->sfaces.size();
    ++Bt.n;
    const long long n0 = L->maxn[0] / 2 + 8, n1 = L->maxn[1] / 2 + 8, n2 = L->maxn[2] / 2 + 8;
    mp = std::max(mp, std::max(n1 * n2, std::max(n0 * n2, n0 * n1)));
  }
  if (Bt.n) hipLaunchKernelGGL(k_cpatch, dim3((unsigned)((mp + 255) / 256), (unsigned)Bt.ycum[Bt.n], (unsigned)nslots), dim3(256), 0, ctx->stream, Bt);
  return 0;
}

// the context's lists of cells for k_faces_curv_list and (general BoxArrays) k_curv_general (grow-never: 1 M entries each);
// layout: [count, gcount, pad, pad][int2 x CAP][int4 x CAP]
static int pa_slow_list(pa_ctx* ctx, SlowList* sl) {
  constexpr int CAP = 1 << 20;
  if (!ctx->d_slow) {
    PA_HIP(hipMalloc(&ctx->d_slow, (sizeof(int2) + sizeof(int4)) * (size_t)CAP + 16));
  }
  sl->count = (int*)ctx->d_slow;
  sl->items = (int2*)((char*)ctx->d_slow + 16);
  sl->cap = CAP;
  sl->gcount = sl->count + 1;
  sl->gitems = (int4*)((char*)ctx->d_slow + 16 + sizeof(int2) * (size_t)CAP);
  sl->gcap = CAP;
  return 0;
}

// diagnostic (tests): how many cells the clip-aware fix-up of the LAST pass handed to its general path; -1: never used
extern "C" int pa_last_slow_cells(pa_ctx* ctx) {
  PaBind bind_(ctx);
  if (!ctx) return -1;
  if (!ctx->d_slow) return -1;
  int n = -1;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(&n, ctx->d_slow, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return n;
}

// ===================================================================================== irregular cells (general BoxArrays)
// The exact-normal pipeline leaves the flame normal exact in EVERY valid cell whatever the BoxArray looks like: a cell's normal
// needs the progress variable in the six face neighbours only, and a face ghost cell holds one well-defined value (a valid
// cell's, or the reference's boundary condition of the face's direction).  The CURVATURE of a boundary cell X of box B needs
// the normal of the ghost cell Y behind the face; where Y is a valid cell of a neighbouring box N, the sweep forms that
// normal from the progress variable around Y as B's FAB holds it, and that is N's view of it only if every tangential
// neighbour Z of Y is a valid cell too -- or the ghost cell of exactly one reader.  On general BoxArrays it is not:
//   * a face that is partly covered by a neighbour and partly coarse-fine: at the line where it changes, Z is a face ghost of
//     B (boundary condition normal to B's face) AND N's ghost cell in the tangential direction (another boundary value);
//   * a concave coarse-fine corner: the edge ghost Z has two valid neighbours in two boxes, each with its own boundary value.
// k_find_irregular lists those cells X (a property of the BoxArray, found once per level): a boundary cell with a valid ghost
// neighbour Y = X + s e_d is REGULAR if, for both tangential directions t and both signs, Z = Y +- e_t is
//   - a valid cell, reached through a FAB slot the sweep / the fix-up read as such: Z inside the face's extent, or an edge
//     ghost whose two faces (d and t) are both ordinary -- the ring of a special face holds no valid cells' values;
//   - or an edge ghost that is not a valid cell while W = X +- e_t is not one either and face d is ordinary: the convex corner
//     the face fix-up handles (X lies in the first layer behind the special face t, the ring of that face's compact array holds
//     N's boundary value: k_prep_ring);
// and the second ghost layer Y + s e_d is a valid cell (N at least two cells thick).  Everything else is irregular.  The
// rule is deliberately conservative: an irregular cell costs a few hundred dependent loads once per pass, a missed one a
// wrong curvature.  k_curv_general then recomputes K of the listed cells from the FINAL normals of the box and from normals
// of ghost cells rebuilt as their owner sees them (gen_c: valid cell -> phi of B's FAB, which FillBoundary(2) filled
// everywhere; otherwise the boundary condition of the direction Z - Y, MLMG applyBC / InterpBndryData as k_prep_faces) --
// nothing in it depends on compact arrays, rings or stored masks.
__global__ __launch_bounds__(256) void k_find_irregular(DLevelView L, int y0, int4* items, int* count, int cap) {
  const int row = (int)blockIdx.y + y0, b = row / 6, f = row % 6;
  if (b >= L.nboxes) return;
  const DBox B = L.boxes[b];
  const int fdir = f >> 1, fside = f & 1;
  const int t0 = fdir == 0 ? 1 : 0, t1 = fdir == 2 ? 1 : 2;
  const unsigned n0 = B.hi[t0] - B.lo[t0] + 1, n1 = B.hi[t1] - B.lo[t1] + 1;
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  if (t >= n0 * n1) return;
  int X[3];
  X[fdir] = fside ? B.hi[fdir] : B.lo[fdir];
  X[t0] = B.lo[t0] + (int)(t % n0);
  X[t1] = B.lo[t1] + (int)(t / n0);
  for (int g = 0; g < f; ++g)  // a cell on an edge / corner of the box belongs to the lowest-numbered face it touches
    if (X[g >> 1] == ((g & 1) ? B.hi[g >> 1] : B.lo[g >> 1])) return;
  bool irr = false;
  for (int g = 0; g < 6 && !irr; ++g) {
    const int d = g >> 1, sg = (g & 1) ? 1 : -1;
    if (X[d] != ((g & 1) ? B.hi[d] : B.lo[d])) continue;
    int Y[3] = {X[0], X[1], X[2]};
    Y[d] += sg;
    if (classify(L, Y[0], Y[1], Y[2]) != 0) continue;  // boundary condition on n itself: exact from the box's own normals
    const bool fd_special = L.sfindex[b * 6 + g] >= 0;
    // a valid ghost cell behind a SPECIAL face (one that is coarse-fine elsewhere): the sweep's ghost row / column / plane of
    // such a face comes from the compact array and its second stream -- the row beyond, which the ghost normal needs -- is
    // re-aimed at that array (pa_fused_march3.h), so no ghost normal behind a special face is usable.  In the interior of
    // the face k_faces_curv_fast forms the curvature with the neighbouring box's final normal (when that box is local);
    // cells on the face's perimeter and ghost cells owned by another rank's box are listed
    if (fd_special) {
      bool other = false;
      for (int tt = 0; tt < 3; ++tt) other = other || (tt != d && (X[tt] == B.lo[tt] || X[tt] == B.hi[tt]));
      int sb, yw[3];
      (void)classify(L, Y[0], Y[1], Y[2], sb, yw);
      if (other || sb < 0) irr = true;
      continue;
    }
    {
      int Y2[3] = {Y[0], Y[1], Y[2]};
      Y2[d] += sg;
      if (classify(L, Y2[0], Y2[1], Y2[2]) != 0) irr = true;
    }
    for (int tt = 0; tt < 3 && !irr; ++tt) {
      if (tt == d) continue;
      for (int s2 = 0; s2 < 2 && !irr; ++s2) {
        const int sg2 = s2 ? 1 : -1;
        int Z[3] = {Y[0], Y[1], Y[2]};
        Z[tt] += sg2;
        const bool zvalid = classify(L, Z[0], Z[1], Z[2]) == 0;
        if (Z[tt] >= B.lo[tt] && Z[tt] <= B.hi[tt]) { irr = !zvalid; continue; }
        int W[3] = {X[0], X[1], X[2]};
        W[tt] += sg2;
        const bool ft_special = L.sfindex[b * 6 + tt * 2 + s2] >= 0;
        if (!zvalid) irr = !(classify(L, W[0], W[1], W[2]) != 0 && !fd_special);
        else irr = fd_special || ft_special;
      }
    }
  }
  if (!irr) return;
  const int i = atomicAdd(count, 1);
  if (items && i < cap) items[i] = make_int4(b, X[0], X[1], X[2]);
}

// the level's list of irregular cells, built on first use (a cache of the level object, like level_cg)
static int level_irregular(pa_ctx* ctx, const pa_level* Lc) {
  pa_level* L = const_cast<pa_level*>(Lc);
  if (L->nirr >= 0) return 0;
  const int nb = (int)L->boxes.size();
  if (nb == 0) { L->nirr = 0; return 0; }
  const long long n0 = L->maxn[0], n1 = L->maxn[1], n2 = L->maxn[2];
  const long long nf = std::max(n1 * n2, std::max(n0 * n2, n0 * n1));
  int* d_count = nullptr;
  // every error path releases the counter and the half-built list: nirr stays -1, so the next call starts over (advisor, round 4)
  auto fail = [&](const char* what) {
    if (d_count) (void)hipFree(d_count);
    if (L->d_irr) { (void)hipFree(L->d_irr); L->d_irr = nullptr; }
    (void)hipGetLastError();
    return pa_fail(ctx, std::string("irregular-cell list: ") + what);
  };
  if (L->d_irr) { (void)hipFree(L->d_irr); L->d_irr = nullptr; }  // left by a call that failed before this guard existed
  if (hipMalloc(&d_count, sizeof(int)) != hipSuccess) return fail("device allocation failed");
  int n = 0;
  for (int pass = 0; pass < 2; ++pass) {  // count, then fill
    if (pass == 1) {
      if (n == 0) break;
      if (hipMalloc(&L->d_irr, sizeof(int4) * (size_t)n) != hipSuccess) { L->d_irr = nullptr; return fail("device allocation failed"); }
    }
    if (hipMemsetAsync(d_count, 0, sizeof(int), ctx->stream) != hipSuccess) return fail("memset failed");
    for (int y0 = 0; y0 < nb * 6; y0 += 65535 / 6 * 6)
      hipLaunchKernelGGL(k_find_irregular, dim3((unsigned)((nf + 255) / 256), (unsigned)std::min(65535 / 6 * 6, nb * 6 - y0)), dim3(256), 0, ctx->stream, L->view, y0,
                         pass ? (int4*)L->d_irr : nullptr, d_count, n);
    int m = 0;
    if (hipMemcpyAsync(&m, d_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return fail("reading the count failed");
    if (pass == 1 && m != n) return fail("the two passes disagree");
    n = m;
  }
  (void)hipFree(d_count);
  d_count = nullptr;
  if (n > 1) {  // the kernel appended the cells in any order: sort by (box, k, j, i) so that neighbouring threads touch neighbouring cells
    std::vector<int4> h((size_t)n);
    if (hipMemcpy(h.data(), L->d_irr, sizeof(int4) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return fail("download failed");
    std::sort(h.begin(), h.end(), [](const int4& a, const int4& b) {
      if (a.x != b.x) return a.x < b.x;
      if (a.w != b.w) return a.w < b.w;
      if (a.z != b.z) return a.z < b.z;
      return a.y < b.y;
    });
    if (hipMemcpy(L->d_irr, h.data(), sizeof(int4) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) return fail("upload failed");
  }
  L->nirr = n;
  return 0;
}
extern "C" int64_t pa_level_irregular_cells(pa_ctx* ctx, const pa_level* L) {
  PaBind bind_(ctx);
  if (!ctx || !L) return -1;
  if (level_irregular(ctx, L)) return -1;
  return L->nirr;
}

struct GenLev {
  DLevelView L; DMFView MP; int pcomp;          // the level and its phi (2 ghost layers, FillBoundary done)
  DLevelView LCp; DMFView MCp; int cpcomp;      // coarse phi (or this rank's coarse-source copy of it)
  DLevelView LCn; DMFView MCn; int cncomp0;     // coarse flame normal (likewise)
  DMFView MO; int ncomp0, kcomp;
  FaceArgs A;
  const int4* items; int n;
  // dynamic list (filled by k_faces_curv_fast<CLIP> in this pass): the count lives on the device, an item's first word carries
  // box | batch level << 24 | slot << 27 and only the items of batch level `lev` are this launch's
  const int* ncount = nullptr; int lev = -1;
  int use_cp = 0; long long cp_stride = 0;  // the level's coarse patches hold the coarse normal component of each face's direction
};
struct GenBox {
  const DLevelView* L; const DLevelView* LCp; DMFView MCp; int cpcomp;
  DBox B; FabView P; int pcomp, has_crse, bc[3]; double pmin, invd;
  __device__ __forceinline__ double c_in(const int p[3]) const { return (P(p[0], p[1], p[2], pcomp) - pmin) * invd; }
};
// the progress variable at Z = Y + sg e_dir as the owner of the valid cell Y sees it (Y: a cell of the box or in the first ghost layer)
__device__ __noinline__ double gen_c(const GenBox& g, const int Y[3], int dir, int sg, bool& ok) {
  int Z[3] = {Y[0], Y[1], Y[2]};
  Z[dir] += sg;
  if (in_box(g.B, Z)) return g.c_in(Z);
  const int cls = classify(*g.L, Z[0], Z[1], Z[2]);
  if (cls == 0) return g.c_in(Z);
  if (cls == 2) {
    const double v = g.c_in(Y);
    return (g.bc[dir] == PA_BC_REFLECT_ODD) ? -v : v;
  }
  if (!g.has_crse) { ok = false; return 0.0; }
  int blen = 3;  // thickness of Y's box along dir (levels with boxes thinner than 3 cells do not take this pipeline)
  if (in_box(g.B, Y)) blen = g.B.hi[dir] - g.B.lo[dir] + 1;
  else {
    int sb, yw[3];
    if (classify(*g.L, Y[0], Y[1], Y[2], sb, yw) == 0 && sb >= 0) blen = g.L->boxes[sb].hi[dir] - g.L->boxes[sb].lo[dir] + 1;
  }
  double coef[4];
  const int NX = cf_normal_coef(blen, 2, coef);
  const double bv = cf_bndry_value(*g.L, *g.LCp, g.MCp, g.cpcomp, Z, dir, 2, ok);  // MCp carries the affine view of the coarse phi
  double tmp = 0.0;
  for (int m = 1; m < NX; ++m) {
    int pc[3] = {Z[0], Z[1], Z[2]};
    pc[dir] -= sg * m;
    tmp += g.c_in(pc) * coef[m];
  }
  double r = tmp;
  r += bv * coef[0];
  return r;
}
__device__ __noinline__ Vec3 gen_normal(const GenBox& g, const int Y[3], const double dxinv[3], bool& ok) {
  Vec3 n;
  const double cxm = gen_c(g, Y, 0, -1, ok), cxp = gen_c(g, Y, 0, 1, ok), cym = gen_c(g, Y, 1, -1, ok), cyp = gen_c(g, Y, 1, 1, ok);
  const double czm = gen_c(g, Y, 2, -1, ok), czp = gen_c(g, Y, 2, 1, ok);
  normal_from(cxm, cxp, cym, cyp, czm, g.c_in(Y), czp, dxinv, n.x, n.y, n.z);
  return n;
}

// the coarse-normal boundary value of the cell's own coarse-fine face (out of line: the common irregular cell has none)
__device__ __noinline__ double gen_crse_normal(const DLevelView& L, const DLevelView& LCn, const DMFView& MCn, int comp, const int q[3], int d, int ratio, bool& ok) {
  return cf_bndry_value(L, LCn, MCn, comp, q, d, ratio, ok);
}
// GEN: the code that rebuilds a ghost normal from the progress variable is compiled in (needed where the ghost cell's owner is
// another rank's box, or -- CLIP -- where the sweep zeroed a stored normal); PATCH: the coarse-fine boundary value of the
// cell's own face comes from the face's coarse patch (the owner-map interpolation is not compiled in).  The common case --
// one rank, no clip, patches on -- is <false, false, true>: a handful of loads per cell, no calls.
template <bool CLIP, bool GEN, bool PATCH>
__device__ __forceinline__ void curv_general_body(const GenLev& G, int* nbad, const SlotK& sk, const long long i0, const long long stride) {
  const DLevelView& L = G.L;
  const double dxinv[3] = {L.dxinv[0], L.dxinv[1], L.dxinv[2]};
  const long long ntot = G.ncount ? min(*G.ncount, G.n) : G.n;
  for (long long i = i0; i < ntot; i += stride) {  // (64-bit: i0 + stride must not wrap)
    const int4 it = G.items[i];
    int z = (int)blockIdx.z, b = it.x;
    if (G.lev >= 0) {
      if (((it.x >> 24) & 7) != G.lev) continue;
      z = (int)((unsigned)it.x >> 27);
      b = it.x & 0xffffff;
    }
    FaceArgs A = G.A;
    if (sk.prog) { A.pmin = sk.prog[2 * z]; A.invd = sk.prog[2 * z + 1]; }
    const int cncomp0 = G.cncomp0 + sk.cn_z * z, ncomp0 = G.ncomp0 + 8 * z, kcomp = G.kcomp + 8 * z;
    const int X[3] = {it.y, it.z, it.w};
    GenBox g;
    g.L = &G.L; g.LCp = &G.LCp; g.MCp = G.MCp; g.cpcomp = G.cpcomp + z;
    g.MCp.xform = 1; g.MCp.xa = A.pmin; g.MCp.xb = A.invd;
    g.B = L.boxes[b];
    g.P = mf_view(G.MP, g.B, b);
    g.pcomp = G.pcomp + z; g.has_crse = A.has_crse; g.pmin = A.pmin; g.invd = A.invd;
    for (int d = 0; d < 3; ++d) g.bc[d] = A.bc[d];
    const DBox& B = g.B;
    const int n[3] = {B.hi[0] - B.lo[0] + 1, B.hi[1] - B.lo[1] + 1, B.hi[2] - B.lo[2] + 1};
    double* o = G.MO.data + G.MO.off[b];
    bool ok = true;
    auto clipped = [&](const int p[3]) { const double c = g.c_in(p); return c < A.thr || c > 1.0 - A.thr; };
    double* kout = o + fab_index(B, G.MO.ng, G.MO.ncomp, kcomp, X[0], X[1], X[2]);
    if (CLIP && clipped(X)) { *kout = 0.0; continue; }
    // component d of the UNCLIPPED normal of cell p of this box (the sweep zeroed the clipped ones in the output)
    auto nrm = [&](const int p[3], int d) -> double {
      if (CLIP && GEN && clipped(p)) return comp_of(gen_normal(g, p, dxinv, ok), d);
      return o[fab_index(B, G.MO.ng, G.MO.ncomp, ncomp0 + d, p[0], p[1], p[2])];
    };
    double curv = 0.0;
    for (int d = 0; d < 3; ++d) {
      const double n0d = nrm(X, d);
      double nb[2];
      for (int s2 = 0; s2 < 2; ++s2) {
        const int sg = s2 ? 1 : -1;
        int q[3] = {X[0], X[1], X[2]};
        q[d] += sg;
        if (in_box(B, q)) { nb[s2] = nrm(q, d); continue; }
        int sb, qw[3];
        const int cls = classify(L, q[0], q[1], q[2], sb, qw);
        if (cls == 0) {
          // a valid cell of a neighbouring box: its FINAL normal from that box's output when the box is local (the sweeps of
          // the level are done), rebuilt from the progress variable as its owner sees it when it is another rank's
          if (sb >= 0) {
            const double v = G.MO.data[G.MO.off[sb] + fab_index(L.boxes[sb], G.MO.ng, G.MO.ncomp, ncomp0 + d, qw[0], qw[1], qw[2])];
            nb[s2] = (CLIP && GEN && v == 0.0 && clipped(q)) ? comp_of(gen_normal(g, q, dxinv, ok), d) : v;
          } else if (GEN) {
            nb[s2] = comp_of(gen_normal(g, q, dxinv, ok), d);
          } else {  // (cannot happen: the host compiles GEN in for sharded levels)
            ok = false; nb[s2] = 0.0;
          }
        } else if (cls == 2) {
          nb[s2] = (A.bc[d] == PA_BC_REFLECT_ODD) ? -n0d : n0d;
        } else {
          if (!A.has_crse) { ok = false; nb[s2] = 0.0; continue; }
          double coef[4];
          const int NX = cf_normal_coef(n[d], A.ratio, coef);
          // q is a ghost cell of face (d, s2) of this box, which is special: its masks are stored, its coarse patch holds component d
          double bv;
          const int e2 = L.sfindex[b * 6 + d * 2 + s2];
          const long long cpo = (G.use_cp && L.cp && e2 >= 0) ? L.cpoff[e2] : -1;
          if (e2 < 0 || ((PATCH || cpo >= 0) && (!L.cp || L.cpoff[e2] < 0))) {  // (cannot happen: a coarse-fine ghost cell makes its face special, with a patch)
            ok = false; bv = 0.0;
          } else if (PATCH || cpo >= 0) {
            const int u0 = (d == 0) ? 1 : 0, u1 = (d == 2) ? 1 : 2;
            const unsigned code = L.sfcode[L.sfoff[e2] + (q[u0] - B.lo[u0]) + (long long)n[u0] * (q[u1] - B.lo[u1])];
            const int xf[1] = {0};
            double bv1[1];
            cf_interp_patch<1>(code, L.cp + z * G.cp_stride + L.cpoff[e2], B, s2, G.MCn, q, d, xf, ok, bv1);
            bv = bv1[0];
          } else {
            bv = gen_crse_normal(L, G.LCn, G.MCn, cncomp0 + d, q, d, A.ratio, ok);
          }
          double tmp = 0.0;
          for (int m = 1; m < NX; ++m) {
            int pc[3] = {q[0], q[1], q[2]};
            pc[d] -= sg * m;  // into the box
            const double v = (m == 1) ? n0d : nrm(pc, d);
            tmp += v * coef[m];
          }
          double gv = tmp;
          gv += bv * coef[0];
          nb[s2] = gv;
        }
      }
      curv += cdiff(dxinv[d], nb[0], n0d, nb[1]);
    }
    curv = curv * 0.5;
    if (!ok) atomicAdd(nbad, 1);
    *kout = curv;
  }
}
template <bool CLIP, bool GEN = true, bool PATCH = false>
__global__ __launch_bounds__(256) void k_curv_general(GenLev G, int* nbad, SlotK sk) {
  curv_general_body<CLIP, GEN, PATCH>(G, nbad, sk, blockIdx.x * 256LL + threadIdx.x, gridDim.x * 256LL);
}
// the static lists of several levels in one launch: level l owns workgroups wg0[l] .. wg0[l+1]-1, 256 list items each
struct GenBatch { int n; unsigned wg0[PA_MAXB + 1]; GenLev a[PA_MAXB]; };
static_assert(sizeof(GenBatch) + sizeof(SlotK) + 16 <= 4000, "kernel arguments of k_curv_general_levels");
template <bool CLIP, bool GEN, bool PATCH>
__global__ __launch_bounds__(256) void k_curv_general_levels(GenBatch Bt, int* nbad, SlotK sk) {
  int l = 0;
  while (l + 1 < Bt.n && blockIdx.x >= Bt.wg0[l + 1]) ++l;
  const GenLev G = Bt.a[l];
  curv_general_body<CLIP, GEN, PATCH>(G, nbad, sk, (long long)(blockIdx.x - Bt.wg0[l]) * 256 + threadIdx.x, 1LL << 40);
}

// can the exact-normal pipeline run on this level (same answer on every rank of a sharded level)?
bool pa_fused2_level_ok(const pa_level* L) {
  // ANY BoxArray whose boxes are at least three cells thick: general BoxArrays (concave coarse-fine corners, faces partly covered by a
  // neighbour) have their irregular cells listed per level and recomputed after the fix-up (k_curv_general); boxes at most 32 cells
  // wide run k_gradcurv_march3n's CG variant
  const std::vector<DBox>& all = L->gboxes.empty() ? L->boxes : L->gboxes;
  for (const DBox& B : all)
    for (int d = 0; d < 3; ++d)
      if (B.hi[d] - B.lo[d] + 1 < 3) return false;
  return true;
}

// ---- round 6: the Gaussian curvature of the cells the KG sweep cannot get right (pa_fused_march3.h, GOUT == 2): the first layer behind
// every special face -- its ghost G is the boundary condition on G (the caller has run FillBoundary + applyBC on the stored G), not what
// the sweep forms from ghost c -- and the level's irregular cells.  k_gauss_curv's operations (pa_curvopts.hip) on the stored G, a thread
// per 2 x 2 block of a chunk record / per listed cell; every other cell keeps the sweep's value, which is the same arithmetic on the
// same values.
struct GaussFixLev { DLevelView L; DMFView G, O; int pc, kgc; double thr; const SfChunk* ck; unsigned nck; const int4* irr; unsigned nirr; unsigned w0; };
__device__ __forceinline__ void gauss_cell(const GaussFixLev& A, int b, int i, int j, int k) {
  const DBox V = A.L.boxes[b];
  const FabView G = mf_view(A.G, V, b), O = mf_view(A.O, V, b);
  double H[3][3], g[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const double c0 = G(i, j, k, d);
    g[d] = c0;
    H[d][0] = cdiff(A.L.dxinv[0], G(i - 1, j, k, d), c0, G(i + 1, j, k, d));
    H[d][1] = cdiff(A.L.dxinv[1], G(i, j - 1, k, d), c0, G(i, j + 1, k, d));
    H[d][2] = cdiff(A.L.dxinv[2], G(i, j, k - 1, d), c0, G(i, j, k + 1, d));
  }
  const double ax0 = H[1][1] * H[2][2] - H[2][1] * H[1][2];
  const double ay0 = H[1][2] * H[2][0] - H[2][2] * H[1][0];
  const double az0 = H[1][0] * H[2][1] - H[2][0] * H[1][1];
  const double ax1 = H[0][2] * H[2][1] - H[2][2] * H[0][1];
  const double ay1 = H[0][0] * H[2][2] - H[2][0] * H[0][2];
  const double az1 = H[0][1] * H[2][0] - H[2][1] * H[0][0];
  const double ax2 = H[0][1] * H[1][2] - H[1][1] * H[0][2];
  const double ay2 = H[0][2] * H[1][0] - H[1][2] * H[0][0];
  const double az2 = H[0][0] * H[1][1] - H[1][0] * H[0][1];
  const double cx = g[0], cy = g[1], cz = g[2];
  const double sn = sqrt(cx * cx + cy * cy + cz * cz);
  const double gn = (1e-14 < sn) ? sn : 1e-14;
  double kg = (cx * (ax0 * cx + ax1 * cy + ax2 * cz) + cy * (ay0 * cx + ay1 * cy + ay2 * cz) + cz * (az0 * cx + az1 * cy + az2 * cz)) / ((gn * gn) * (gn * gn));
  if (A.thr >= 0.0) {
    const double p = O(i, j, k, A.pc);
    if (p < A.thr || p > 1.0 - A.thr) kg = 0.0;
  }
  O(i, j, k, A.kgc) = kg;
}
__global__ __launch_bounds__(256) void k_gauss_cells(LevBatch<GaussFixLev> Bt) {
  int l = 0;
  while (l + 1 < Bt.n && blockIdx.x >= Bt.a[l + 1].w0) ++l;
  const GaussFixLev& A = Bt.a[l];
  const unsigned w = blockIdx.x - A.w0;
  if (w < A.nck) {  // a chunk of a special face: the first-layer cells behind its ghost cells
    const SfChunk D = A.ck[w];
    const int dir = D.dir_side >> 1, side = D.dir_side & 1;
    const int e0 = D.hi[0] - D.lo[0] + 1, e1 = D.hi[1] - D.lo[1] + 1, e2 = D.hi[2] - D.lo[2] + 1;
    const int n0 = dir == 0 ? e1 : e0, n1 = dir == 2 ? e1 : e2;
    // the chunk's cw x ch = 1024 cells, a cell per thread and pass with u fastest (consecutive lanes = consecutive cells along the row
    // for y and z faces; the 2 x 2 blocks of the boundary-condition kernels would read every other cell per pass)
    const int sh = 31 - __builtin_clz((unsigned)D.cw);
#pragma unroll 2
    for (int r = 0; r < 4; ++r) {
      const int c = (int)threadIdx.x + 256 * r;
      const int uu = D.u0 + (c & (D.cw - 1)), vv = D.v0 + (c >> sh);
      if (uu < 0 || vv < 0 || uu >= n0 || vv >= n1) continue;
      int i, j, k;
      if (dir == 0) { i = side ? D.hi[0] : D.lo[0]; j = D.lo[1] + uu; k = D.lo[2] + vv; }
      else if (dir == 1) { i = D.lo[0] + uu; j = side ? D.hi[1] : D.lo[1]; k = D.lo[2] + vv; }
      else { i = D.lo[0] + uu; j = D.lo[1] + vv; k = side ? D.hi[2] : D.lo[2]; }
      gauss_cell(A, D.box, i, j, k);
    }
    return;
  }
  const unsigned q = (w - A.nck) * 256u + threadIdx.x;
  if (q < A.nirr) {
    const int4 it = A.irr[q];
    gauss_cell(A, it.x, it.y, it.z, it.w);
  }
}
// G[l]: components 0 .. 2, >= 1 ghost layer, FillBoundary + applyBC done; out[l]: Progress at pc (read for the clip), Kg written at kgc
int pa_gauss_cells_levels(pa_ctx* ctx, int nlev, pa_mf* const* G, pa_mf* const* out, int pc, int kgc, double thr) {
  for (int l0 = 0; l0 < nlev; l0 += PA_MAXB) {
    LevBatch<GaussFixLev> Bt;
    unsigned w = 0;
    for (int l = l0; l < nlev && l < l0 + PA_MAXB; ++l) {
      const pa_level* L = out[l]->lev;
      if (L->boxes.empty()) continue;
      if (G[l]->lev != L || G[l]->ng < 1 || G[l]->ncomp < 3) return pa_fail(ctx, "pa_gauss_cells_levels: G needs 3 components and a ghost layer on the level of out");
      if (L->nirr < 0 && level_irregular(ctx, L)) return 1;
      GaussFixLev A{L->view, G[l]->view, out[l]->view, pc, kgc, thr, L->d_sfchunk, (unsigned)L->nsfchunk, (const int4*)L->d_irr, (unsigned)std::max(L->nirr, 0), w};
      w += A.nck + (A.nirr + 255u) / 256u;
      Bt.a[Bt.n++] = A;
    }
    if (!Bt.n || w == 0) continue;
    hipLaunchKernelGGL(k_gauss_cells, dim3(w), dim3(256), 0, ctx->stream, Bt);
  }
  PA_HIP(hipGetLastError());
  return 0;
}

// before the sweeps: face ghosts of phi + resolved ghost c (faces and ring) of several levels, one launch pair for up to
// PA_MAXB levels.  crse[l]: the coarser level's phi (component ccomp) or this rank's coarse-source copy of it, null on
// level 0 / where this rank has no coarse-fine face; the local half of FillBoundary(2) must have run.
// phase: 1 = the faces (k_prep_faces: reads valid cells and coarse data only, so it may run NEXT TO FillBoundary), 2 = the
// ring (k_prep_ring: reads ghost cells FillBoundary fills), 3 = both
// nslots > 1: components comp .. comp + nslots - 1 (coarse components ccomp ..) in one launch each, slot z with the progress
// range prog[2 z], prog[2 z + 1] (device) and its own set of compact arrays / coarse patches (SlotK)
// phase & 4: the ring reads its neighbours' cells in the boxes that own them (unsharded levels only), not the ghost cells
// phase & 8 (with 1): only the face ghosts of phi (k_prep_faces<.., PHIONLY>): pa_grad_run's applyBC of every level in one launch
int pa_gradcurv_prep_levels(pa_ctx* ctx, int nlev, pa_mf* const* phi, int comp, const pa_mf* const* crse, int ccomp, const int32_t bc[3], double pmin, double pmax, int phase,
                            int nslots, const double* prog) {
  bool direct = (phase & 4) != 0;
  for (int l = 0; l < nlev; ++l) direct = direct && phi[l]->lev->nranks == 1;
  SlotK sk;
  sk.prog = prog;
  const bool use_cp = cpatch_on();  // the patches are gathered with the faces (phase 1) and still hold the coarse phi when the ring runs (phase 2)
  for (int l0 = 0; l0 < nlev; l0 += PA_MAXB) {
    if (use_cp && (phase & 1) && cpatch_launch(ctx, l0, std::min(nlev, l0 + PA_MAXB), phi, crse, ccomp, 0, nslots, 1)) return 1;  // before P.L = L->view picks up cp
    LevBatch<PrepLev> Bf;
    const pa_level* batch_lev[PA_MAXB] = {};
    long long ntf = 0;
    for (int l = l0; l < nlev && l < l0 + PA_MAXB; ++l) {
      const pa_level* L = phi[l]->lev;
      if (L->boxes.empty()) continue;
      if (!(phase & 8) && level_cg(ctx, L, nslots)) return 1;  // phase & 8: phi only (the gradient tool), no compact arrays
      if (L->sfaces.empty()) continue;
      PrepLev P;
      P.cg_stride = cg_stride(L); P.cp_stride = cp_stride(L);
      for (int d = 0; d < 3; ++d) P.A.bc[d] = bc[d];
      P.A.ratio = 2; P.A.has_crse = crse[l] ? 1 : 0; P.A.pmin = pmin; P.A.invd = 1.0 / (pmax - pmin);
      P.L = L->view; P.M = phi[l]->view; P.comp = comp;
      P.LC = crse[l] ? crse[l]->lev->view : L->view;
      P.MC = crse[l] ? crse[l]->view : phi[l]->view;
      P.MC.xform = 1; P.MC.xa = pmin; P.MC.xb = P.A.invd;
      P.ccomp = ccomp;
      P.use_cp = (use_cp && crse[l] && L->cp_total > 0) ? 1 : 0;
      P.wg = (const int2*)L->d_sfwg; P.nwg = L->nsfwg; P.sfboxes = L->d_sfboxes;
      const long long n0 = L->maxn[0], n1 = L->maxn[1], n2 = L->maxn[2];
      ntf = std::max(ntf, std::max(n1 * n2, std::max(n0 * n2, n0 * n1)));
      batch_lev[Bf.n] = L;
      Bf.a[Bf.n] = P; Bf.ycum[Bf.n + 1] = Bf.ycum[Bf.n] + (int)L->sfaces.size(); ++Bf.n;
    }
    if (!Bf.n) continue;
    ProfScope prof(ctx, PA_TAG_BC);
    bool all_patch = true;  // every level of the batch that has a coarser level interpolates from patches: the owner-map path is not compiled in
    for (int q = 0; q < Bf.n; ++q) all_patch = all_patch && (Bf.a[q].use_cp || !Bf.a[q].A.has_crse);
    // the faces from the levels' chunk records (k_prep_faces_chunks; every level with special faces has them)
    LevChunks Ck;
    Ck.w0[0] = 0;
    for (int q = 0; q < Bf.n; ++q) {
      const pa_level* Lq = batch_lev[q];
      if (!Lq->d_sfchunk || Lq->nsfchunk <= 0) return pa_fail(ctx, "pa_gradcurv_prep_levels: a level without chunk records");
      Ck.ck[q] = Lq->d_sfchunk;
      Ck.w0[q + 1] = Ck.w0[q] + (unsigned)Lq->nsfchunk;
    }
    // the ring items of the batch's levels (level_ring): with the faces' launch when both are asked for and the ring reads its
    // neighbours in place (one rank), else a launch of their own
    LevRings Rg, Rnone;
    Rg.w0[0] = 0;
    for (int q = 0; q <= PA_MAXB; ++q) Rnone.w0[q] = 0;
    for (int q = 0; q < PA_MAXB; ++q) { Rnone.it[q] = nullptr; Rnone.n[q] = 0; }
    if (phase & 2)
      for (int q = 0; q < Bf.n; ++q) {
        const pa_level* Lq = batch_lev[q];
        if (level_ring(ctx, Lq)) return 1;
        Rg.it[q] = (const RingItem*)Lq->d_ring;
        Rg.n[q] = (unsigned)Lq->nring;
        Rg.w0[q + 1] = Rg.w0[q] + (unsigned)((Lq->nring + 255) / 256);
      }
    // (few items -- large faces -- hide under the faces: headline 5.913 -> 5.895 ms per pass; the long lists of a BoxArray of many small boxes
    // are latency-bound work that wants its own launch at its own occupancy: irregular hierarchy 6.39 against 6.48 ms merged)
    const bool ring_with_faces = (phase & 1) && (phase & 2) && !(phase & 8) && direct && Rg.w0[Bf.n] * 8u <= Ck.w0[Bf.n];
    if (phase & 1) {
      const LevRings& Rk = ring_with_faces ? Rg : Rnone;
      const dim3 gc(Rk.w0[Bf.n] + Ck.w0[Bf.n], 1, (unsigned)nslots);
      if (phase & 8) {
        if (all_patch) hipLaunchKernelGGL((k_prep_faces_chunks<true, true>), gc, dim3(256), 0, ctx->stream, Bf, Ck, Rk, ctx->d_flags, sk);
        else hipLaunchKernelGGL((k_prep_faces_chunks<false, true>), gc, dim3(256), 0, ctx->stream, Bf, Ck, Rk, ctx->d_flags, sk);
      } else {
        if (all_patch) hipLaunchKernelGGL((k_prep_faces_chunks<true, false>), gc, dim3(256), 0, ctx->stream, Bf, Ck, Rk, ctx->d_flags, sk);
        else hipLaunchKernelGGL((k_prep_faces_chunks<false, false>), gc, dim3(256), 0, ctx->stream, Bf, Ck, Rk, ctx->d_flags, sk);
      }
    }
    if ((phase & 2) && !ring_with_faces && Rg.w0[Bf.n] > 0) {
      const dim3 gr(Rg.w0[Bf.n], 1, (unsigned)nslots);
      if (all_patch && direct) hipLaunchKernelGGL((k_prep_ring<true, true>), gr, dim3(256), 0, ctx->stream, Bf, Rg, ctx->d_flags, sk);
      else if (direct) hipLaunchKernelGGL((k_prep_ring<false, true>), gr, dim3(256), 0, ctx->stream, Bf, Rg, ctx->d_flags, sk);
      else if (all_patch) hipLaunchKernelGGL(k_prep_ring<true>, gr, dim3(256), 0, ctx->stream, Bf, Rg, ctx->d_flags, sk);
      else hipLaunchKernelGGL(k_prep_ring<false>, gr, dim3(256), 0, ctx->stream, Bf, Rg, ctx->d_flags, sk);
    }
  }
  PA_HIP(hipGetLastError());
  return 0;
}

// A sweep group: the boxes of one level that one kernel variant covers (all of them, or -- a level with boxes both wider and
// not wider than 32 cells -- its wide or its narrow ones through an index list)
struct SweepGroup { int lev; const int* list; int n; int dims[3]; };
static void sweep_groups(int l, const pa_level* L, std::vector<SweepGroup>& out) {
  if (L->boxes.empty()) return;
  if (!L->d_blist) { out.push_back({l, nullptr, (int)L->boxes.size(), {L->maxn[0], L->maxn[1], L->maxn[2]}}); return; }
  out.push_back({l, L->d_blist, L->nwide, {L->wmax[0], L->wmax[1], L->wmax[2]}});
  out.push_back({l, L->d_blist + L->nwide, L->nnarrow, {L->nmax[0], L->nmax[1], L->nmax[2]}});
}

static const WgTab* sweep_wgtab(const pa_level* L, const SweepGroup& g, int tw, int mty, int kseg, int part = 0, bool force = false) {
  if (!part && g.n <= 0) return nullptr;
  return pa_sweep_wgtab(L, !g.list ? 2 : (g.list == L->d_blist ? 0 : 1), tw, mty, kseg, force, part);
}

// the sweep with exact normals (the level's compact ghost arrays must be current: pa_gradcurv_prep_level)
static int sweep_group_cg(pa_ctx* ctx, const SweepGroup& g, const pa_mf* phi, int pcomp, double pmin, double pmax, pa_mf* out, int ocomp, double thr, int slot) {
  const pa_level* L = phi->lev;
  if (level_cg(ctx, L, slot + 1)) return 1;
  LevelBP2 bp{L->view, phi->view, out->view};
  bp.L.cg += slot * cg_stride(L);  // the component slot's set of compact arrays
  MarchArgs A{pcomp, ocomp, fused_kseg(), pmin, 1.0 / (pmax - pmin), thr >= 0.0 ? thr : -1.0, 0, 1, 1, 1};
  A.cg = 1;
  A.boxlist = g.list;
  ProfScope prof(ctx, PA_TAG_GRADCURV);
  march_launch(ctx->stream, bp, g.dims[0], g.dims[1], g.dims[2], (unsigned)g.n, A, false, &ctx->sweep_kernel);
  PA_HIP(hipGetLastError());
  return 0;
}
int pa_gradcurv_level_cg(pa_ctx* ctx, const pa_mf* phi, int pcomp, double pmin, double pmax, pa_mf* out, int ocomp, double thr, int slot) {
  std::vector<SweepGroup> gs;
  sweep_groups(0, phi->lev, gs);
  for (const SweepGroup& g : gs)
    if (sweep_group_cg(ctx, g, phi, pcomp, pmin, pmax, out, ocomp, thr, slot)) return 1;
  return 0;
}

// the CG sweeps of all levels: the groups of boxes wider than 32 cells in one launch (k_gradcurv_march3_levels) when they agree
// on the tile variant, else group by group (PA_FORCE_FALLBACKS=1: always); narrow groups one launch each.
// nslots > 1 (slot must be 0): components pcomp .. pcomp + nslots - 1 in ONE launch per kernel variant (blockIdx.y = slot: outputs at
// ocomp + 8 z, compact arrays of slot z, progress range prog[2 z], prog[2 z + 1] on the device); groups that do not take a batched
// launch run slot by slot with the host's ranges pmins / pmaxs
// gout (pa_curvature_run with options): the GOUT variants of the sweeps -- Progress, K, N at out components ocomp .. ocomp + 4, the
// cell-centred gradient of c at components 0 .. 2 of gout[l] (any ghost width).  Only as all-levels launches:
// pa_gradcurv_gout_ok says whether this hierarchy takes them (else the caller runs pass by pass).
// the sweeps of this hierarchy can be split into early and late tiles: every group of wide boxes goes through the all-levels launches
bool pa_gradcurv_gout_ok(int nlev, pa_mf* const* phi);
bool pa_gradcurv_parts_ok(int nlev, pa_mf* const* phi) { return pa_gradcurv_gout_ok(nlev, phi); }
// the G-output sweeps of this hierarchy can form the Gaussian curvature themselves (GOUT == 2, pa_fused_march3.h): every box wider
// than 32 cells (the narrow sweep has no such variant) and at least 16 rows tall (tiles of 13 or 8 rows)
bool pa_gradcurv_kg_ok(int nlev, pa_mf* const* phi) {
  if (!pa_gradcurv_gout_ok(nlev, phi)) return false;
  for (int l = 0; l < nlev; ++l) {
    const pa_level* L = phi[l]->lev;
    if (L->boxes.empty()) continue;
    if (L->nnarrow > 0 || L->maxn[1] < 16) return false;
    for (const DBox& B : L->boxes)
      if (B.hi[0] - B.lo[0] + 1 <= 32) return false;
  }
  return true;
}
bool pa_gradcurv_gout_ok(int nlev, pa_mf* const* phi) {
  if (pa_opt().force_fallbacks) return false;
  std::vector<SweepGroup> all;
  for (int l = 0; l < nlev; ++l) sweep_groups(l, phi[l]->lev, all);
  return all.size() <= 8 * PA_MAXB;  // (the launches come in chunks of PA_MAXB groups)
}

// part (a sharded hierarchy's pass, pa_pipeline.hip): 1 = only the EARLY tiles of the wide boxes (pa_sweep_wgtab: their input is
// complete after the local FillBoundary), 2 = the other tiles and every narrow box; 0 = everything
int pa_gradcurv_levels_cg(pa_ctx* ctx, int nlev, pa_mf* const* phi, int pcomp, double pmin, double pmax, pa_mf* const* out, int ocomp, double thr, int slot,
                          int nslots, const double* prog, const double* pmins, const double* pmaxs, pa_mf* const* gout, int part, int kg) {
  const bool clip = thr >= 0.0;
  if (kg && (!gout || !pa_gradcurv_kg_ok(nlev, phi))) return pa_fail(ctx, "pa_gradcurv_levels_cg: the Gaussian curvature inside the sweep needs the G-output sweeps of wide boxes");
  if (part && !pa_gradcurv_parts_ok(nlev, phi)) return pa_fail(ctx, "pa_gradcurv_levels_cg: this hierarchy's sweeps cannot be split into early and late tiles");
  if (nslots > 1 && (slot != 0 || !prog || !pmins || !pmaxs)) return pa_fail(ctx, "pa_gradcurv_levels_cg: component slots need slot 0 and the progress ranges");
  if (gout && (nslots != 1 || slot != 0 || !pa_gradcurv_gout_ok(nlev, phi))) return pa_fail(ctx, "pa_gradcurv_levels_cg: the G-output sweeps take one component of a hierarchy pa_gradcurv_gout_ok accepts");
  std::vector<SweepGroup> all, lv, rest;
  for (int l = 0; l < nlev; ++l) sweep_groups(l, phi[l]->lev, all);
  for (int l = 0; l < nlev; ++l) const_cast<pa_level*>(phi[l]->lev)->ncg_live = false;  // (set again below by the launches that mirror the x faces' first layer)
  bool any_ncg = false;
  int mty = 0;
  bool same = true;
  for (const SweepGroup& g : all) {
    if (g.dims[0] <= 32) { rest.push_back(g); continue; }
    const int m = g.dims[1] >= 52 ? 13 : (g.dims[1] >= 16 ? 8 : 4);  // as march_launch (tools/ab_driver.py, 13 against 12 / 11 / 10 / 9 rows: +0.058 / +0.041 / +0.31 / +0.97 ms per pass)
    same = same && (mty == 0 || m == mty);
    mty = m;
    lv.push_back(g);
  }
  // (Round 5, measured and not kept: tiles of 11 or 12 rows on BoxArrays whose boxes are 32 or 96 rows tall -- 13 + 13 + 6 rows
  // leave a fifth of the row slots idle -- took 7.52 against 7.41 ms per pass on the re-tiled irregular hierarchy: idle ROWS cost
  // nothing, a partly filled 64-cell tile in x does; profiles/r05_retile.txt.)
  // groups of different tile heights (a level of flat boxes next to one of tall ones): one launch per tile height (until round 5's
  // second session such a hierarchy went group by group, a launch each)
  (void)same;
  const bool ok = !lv.empty() && lv.size() <= 8u * PA_MAXB && !pa_opt().force_fallbacks;  // more than PA_MAXB groups (a hierarchy of 5+ levels): several launches
  if (!ok) {
    rest.insert(rest.begin(), lv.begin(), lv.end());
    lv.clear();
  }
  const std::vector<SweepGroup> lv_all = lv;
  // Round 6: a hierarchy with wide AND narrow boxes -- the narrow-box launch goes to the side stream and runs BESIDE the wide one (both
  // are sweeps: its workgroups fill the CUs the wide launch's tail leaves idle; 6.213 -> 6.167 ms per pass on the irregular hierarchy,
  // tools/ab_driver.py, profiles/r06_narrow_side_ab.txt).  One rank only: a sharded pass uses the side stream for its exchanges.
  bool nar_side = false;
  for (const SweepGroup& g : rest) nar_side = nar_side || g.dims[0] <= 32;
  nar_side = nar_side && !lv_all.empty() && part == 0 && !pa_opt().force_fallbacks && phi[0]->lev->nranks == 1;
  if (nar_side) {
    if (!ctx->stream2) PA_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    while (ctx->sync_evs.size() < 5) {
      hipEvent_t e;
      PA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ctx->sync_evs.push_back(e);
    }
    PA_HIP(hipEventRecord(ctx->sync_evs[3], ctx->stream));  // the ghost cells and compact arrays the sweeps read are complete
    PA_HIP(hipStreamWaitEvent(ctx->stream2, ctx->sync_evs[3], 0));
  }
  for (int pass_i = 0; pass_i < 3 * 8; ++pass_i) {  // (tile height) x (chunk of PA_MAXB groups)
    const int mty_pass = pass_i / 8 == 0 ? 13 : (pass_i / 8 == 1 ? 8 : 4), chunk = pass_i % 8;
    std::vector<SweepGroup> lvm, lv;
    for (const SweepGroup& g : lv_all)
      if ((g.dims[1] >= 52 ? 13 : (g.dims[1] >= 16 ? 8 : 4)) == mty_pass) lvm.push_back(g);
    for (size_t q = (size_t)chunk * PA_MAXB; q < lvm.size() && q < (size_t)(chunk + 1) * PA_MAXB; ++q) lv.push_back(lvm[q]);
    if (lv.empty()) continue;
    const int mty = mty_pass;
    SweepBatch S;
    S.n = (int)lv.size();
    S.wg0[0] = 0;
    // planes per workgroup: the model of march_launch on the whole launch (workgroups of all levels share the rounds)
    std::vector<long long> per_seg(lv.size());
    long long wgs = 0;
    const int kdef = fused_kseg();
    for (size_t q = 0; q < lv.size(); ++q) {
      per_seg[q] = (long long)((lv[q].dims[0] + 63) / 64) * ((lv[q].dims[1] + mty - 1) / mty) * (long long)lv[q].n;
      wgs += per_seg[q] * ((lv[q].dims[2] + kdef - 1) / kdef);
    }
    int tz_best = 0;
    if (wgs < 2048) {
      long long best = -1;
      int nzmax = 0;
      for (const SweepGroup& g : lv) nzmax = std::max(nzmax, g.dims[2]);
      for (int tz = 1; tz <= std::max(1, nzmax / 8); ++tz) {
        long long w = 0;
        int kmax = 0;
        for (size_t q = 0; q < lv.size(); ++q) {
          const int nz = lv[q].dims[2], k = std::max(4, (nz + tz - 1) / tz);
          w += per_seg[q] * ((nz + k - 1) / k);
          kmax = std::max(kmax, k);
        }
        const long long cost = ((w + 255) / 256) * (kmax + 4);
        if (best < 0 || cost < best) { best = cost; tz_best = tz; }
      }
    }
    for (size_t q = 0; q < lv.size(); ++q) {
      const int l = lv[q].lev;
      const pa_level* L = phi[l]->lev;
      if (level_cg(ctx, L, slot + nslots)) return 1;
      S.bp[q] = LevelBP2{L->view, phi[l]->view, out[l]->view};
      S.bp[q].L.cg += slot * cg_stride(L);
      S.cgs[q] = cg_stride(L);
      MarchArgs A{pcomp, ocomp, kdef, pmin, 1.0 / (pmax - pmin), clip ? thr : -1.0, 2, 1, 1, 1};
      A.cg = 1;
      {  // NCG: the first-layer data of the special x faces for the fix-up (PA_NCG=0: off)
        if (pa_opt().ncg && !clip && nslots == 1 && slot == 0 && !L->sfaces.empty()) {
          if (level_ncg(ctx, L)) return 1;
          pa_level* Lm = const_cast<pa_level*>(L);
          A.ncg = L->d_ncg; A.ncgs = cg_stride(L);
          Lm->ncg_live = true;
          any_ncg = true;
          Lm->ncg_minw = lv[q].list ? 33 : 1;  // a list: the level's boxes wider than 32 cells; none: every box of the level is in this group
        }
      }
      A.boxlist = lv[q].list;
      if (tz_best) A.kseg = std::max(4, (lv[q].dims[2] + tz_best - 1) / tz_best);
      const unsigned nb = (unsigned)lv[q].n;
      const dim3 g = march_grid(lv[q].dims[0], lv[q].dims[1], lv[q].dims[2], A.kseg, mty, nb);
      A.nboxes = (int)nb;
      A.txy_max = ((lv[q].dims[0] + 63) / 64) * ((lv[q].dims[1] + mty - 1) / mty);
      A.tiles_max = (int)g.x;
      const WgTab* wt = sweep_wgtab(L, lv[q], 64, mty, A.kseg, part, kg == 2);
      if (wt) A.wgtab = wt->d;
      if (gout) { A.gdata = gout[l]->data; A.goff = gout[l]->d_off; A.gng = gout[l]->ng; }
      if (kg == 2 && wt && wt->d) {  // G only where something reads it (null: everywhere)
        const pa_level* F = l + 1 < nlev ? phi[l + 1]->lev : nullptr;
        const CpPlan* cp = (F && !F->boxes.empty()) ? pa_cp_plan(ctx, F, L) : nullptr;
        const bool no_patches = F && !F->boxes.empty() && F->cp_total > 0 && !(cp && cp->ok);  // the finer level reads the coarse G through the owner map: anywhere
        if (!no_patches) A.gneed = pa_sweep_gneed(L, wt, 64, mty, A.kseg, F ? F->serial : 0, cp && cp->ok ? &cp->hregs : nullptr);
      }
      S.A[q] = A;
      S.wg0[q + 1] = S.wg0[q] + (wt ? wt->n : (part ? 0u : g.x * 8u * ((nb + 7u) / 8u)));  // (a part's table may be empty)
    }
    if (S.wg0[S.n] == 0) continue;
    ProfScope prof(ctx, PA_TAG_GRADCURV);
    S.prog = nslots > 1 ? prog : nullptr;
    const dim3 grid(S.wg0[S.n], (unsigned)nslots);
    if (gout && kg) {  // GOUT == 2: + the Gaussian curvature at out component ocomp + 5
      if (clip) {
        if (mty == 13) hipLaunchKernelGGL((k_gradcurv_march3_levels<13, true, 2>), grid, dim3(64 * 16), 0, ctx->stream, S);
        else hipLaunchKernelGGL((k_gradcurv_march3_levels<8, true, 2>), grid, dim3(64 * 11), 0, ctx->stream, S);
      } else {
        if (mty == 13) hipLaunchKernelGGL((k_gradcurv_march3_levels<13, false, 2>), grid, dim3(64 * 16), 0, ctx->stream, S);
        else hipLaunchKernelGGL((k_gradcurv_march3_levels<8, false, 2>), grid, dim3(64 * 11), 0, ctx->stream, S);
      }
    } else if (gout) {
      if (clip) {
        if (mty == 13) hipLaunchKernelGGL((k_gradcurv_march3_levels<13, true, true>), grid, dim3(64 * 16), 0, ctx->stream, S);
        else if (mty == 8) hipLaunchKernelGGL((k_gradcurv_march3_levels<8, true, true>), grid, dim3(64 * 11), 0, ctx->stream, S);
        else hipLaunchKernelGGL((k_gradcurv_march3_levels<4, true, true>), grid, dim3(64 * 7), 0, ctx->stream, S);
      } else {
        if (mty == 13) hipLaunchKernelGGL((k_gradcurv_march3_levels<13, false, true>), grid, dim3(64 * 16), 0, ctx->stream, S);
        else if (mty == 8) hipLaunchKernelGGL((k_gradcurv_march3_levels<8, false, true>), grid, dim3(64 * 11), 0, ctx->stream, S);
        else hipLaunchKernelGGL((k_gradcurv_march3_levels<4, false, true>), grid, dim3(64 * 7), 0, ctx->stream, S);
      }
    } else if (clip) {
      if (mty == 13) hipLaunchKernelGGL((k_gradcurv_march3_levels<13, true>), grid, dim3(64 * 16), 0, ctx->stream, S);
      else if (mty == 8) hipLaunchKernelGGL((k_gradcurv_march3_levels<8, true>), grid, dim3(64 * 11), 0, ctx->stream, S);
      else hipLaunchKernelGGL((k_gradcurv_march3_levels<4, true>), grid, dim3(64 * 7), 0, ctx->stream, S);
    } else {
      if (mty == 13) hipLaunchKernelGGL(k_gradcurv_march3_levels<13>, grid, dim3(64 * 16), 0, ctx->stream, S);
      else if (mty == 8) hipLaunchKernelGGL(k_gradcurv_march3_levels<8>, grid, dim3(64 * 11), 0, ctx->stream, S);
      else hipLaunchKernelGGL(k_gradcurv_march3_levels<4>, grid, dim3(64 * 7), 0, ctx->stream, S);
    }
    PA_HIP(hipGetLastError());
  }
  if (part == 1) return 0;  // the narrow boxes and the groups outside the all-levels launches go with the late tiles
  // the narrow groups (boxes at most 32 cells wide) of all levels in one launch too
  std::vector<SweepGroup> nar;
  {
    std::vector<SweepGroup> keep;
    for (const SweepGroup& g : rest) (g.dims[0] <= 32 ? nar : keep).push_back(g);
    if (!nar.empty() && !pa_opt().force_fallbacks) rest.swap(keep);
    else nar.clear();
  }
  const std::vector<SweepGroup> nar_all = nar;
  for (size_t n0 = 0; n0 < nar_all.size(); n0 += PA_MAXB) {  // chunks of PA_MAXB groups
    const std::vector<SweepGroup> nar(nar_all.begin() + (long)n0, nar_all.begin() + (long)std::min(nar_all.size(), n0 + PA_MAXB));
    constexpr int NRW = 8;
    SweepBatch S;
    S.n = (int)nar.size();
    S.wg0[0] = 0;
    for (size_t q = 0; q < nar.size(); ++q) {
      const int l = nar[q].lev;
      const pa_level* L = phi[l]->lev;
      if (level_cg(ctx, L, slot + nslots)) return 1;
      S.bp[q] = LevelBP2{L->view, phi[l]->view, out[l]->view};
      S.bp[q].L.cg += slot * cg_stride(L);
      S.cgs[q] = cg_stride(L);
      MarchArgs A{pcomp, ocomp, fused_kseg(), pmin, 1.0 / (pmax - pmin), clip ? thr : -1.0, 2, 1, 1, 1};
      A.cg = 1;
      A.boxlist = nar[q].list;
      A.nboxes = nar[q].n;
      // planes per workgroup as march_launch chooses them for one level
      const int nx = nar[q].dims[0], ny = nar[q].dims[1], nz = nar[q].dims[2];
      {
        const long long per_seg = (long long)((nx + 63) / 64) * ((ny + 12) / 13) * nar[q].n;
        if (per_seg * ((nz + A.kseg - 1) / A.kseg) < 2048) {
          long long best = -1;
          int best_k = A.kseg;
          for (int tz = 1; tz <= std::max(1, nz / 8); ++tz) {
            const int k = (nz + tz - 1) / tz;
            const long long rounds = (per_seg * ((nz + k - 1) / k) + 255) / 256, cost = rounds * (k + 4);
            if (best < 0 || cost < best) { best = cost; best_k = k; }
          }
          A.kseg = std::max(best_k, 4);
        }
      }
      const unsigned tiles = (unsigned)(((nx + 31) / 32) * ((ny + 2 * NRW - 1) / (2 * NRW)) * ((nz + A.kseg - 1) / A.kseg));
      A.tiles_max = (int)tiles;
      const WgTab* wt = sweep_wgtab(L, nar[q], 32, 2 * NRW, A.kseg);
      if (wt) A.wgtab = wt->d;
      if (gout) { A.gdata = gout[l]->data; A.goff = gout[l]->d_off; A.gng = gout[l]->ng; }
      S.A[q] = A;
      S.wg0[q + 1] = S.wg0[q] + (wt ? wt->n : tiles * 8u * (((unsigned)nar[q].n + 7u) / 8u));
    }
    ProfScope prof(ctx, PA_TAG_GRADCURV);
    S.prog = nslots > 1 ? prog : nullptr;
    hipStream_t ns = nar_side ? ctx->stream2 : ctx->stream;
    if (gout && clip) hipLaunchKernelGGL((k_gradcurv_march3n_levels<NRW, true, true>), dim3(S.wg0[S.n], 1u), dim3(64 * (NRW + 2)), 0, ns, S);
    else if (gout) hipLaunchKernelGGL((k_gradcurv_march3n_levels<NRW, false, true>), dim3(S.wg0[S.n], 1u), dim3(64 * (NRW + 2)), 0, ns, S);
    else if (clip) hipLaunchKernelGGL((k_gradcurv_march3n_levels<NRW, true>), dim3(S.wg0[S.n], (unsigned)nslots), dim3(64 * (NRW + 2)), 0, ns, S);
    else hipLaunchKernelGGL((k_gradcurv_march3n_levels<NRW, false>), dim3(S.wg0[S.n], (unsigned)nslots), dim3(64 * (NRW + 2)), 0, ns, S);
    PA_HIP(hipGetLastError());
    if (lv.empty()) ctx->sweep_kernel = "k_gradcurv_march3n_levels<NRW=8" + std::string(clip ? ",CLIP" : "") + ">[" + std::to_string(S.n) + " levels per launch]";
  }
  if (nar_side) {  // later work on the caller's stream sees the narrow boxes swept
    PA_HIP(hipEventRecord(ctx->sync_evs[4], ctx->stream2));
    PA_HIP(hipStreamWaitEvent(ctx->stream, ctx->sync_evs[4], 0));
  }
  if (gout && !rest.empty()) return pa_fail(ctx, "pa_gradcurv_levels_cg: a sweep group outside the all-levels launches (G-output variant)");
  for (const SweepGroup& g : rest)
    for (int z = 0; z < nslots; ++z)
      if (sweep_group_cg(ctx, g, phi[g.lev], pcomp + z, nslots > 1 ? pmins[z] : pmin, nslots > 1 ? pmaxs[z] : pmax, out[g.lev], ocomp + 8 * z, thr, slot + z)) return 1;
  if (!lv.empty())
    ctx->sweep_kernel = "k_gradcurv_march3_levels<MTY=" + std::to_string(mty) + (clip ? ",CLIP" : "") + ">[" + std::to_string(lv.size()) + " levels per launch" + (any_ncg ? "; x faces mirrored" : "") + "]" +
                        ((rest.empty() && nar.empty()) ? "" : " + narrow-box launch(es)");
  return 0;
}

// after the sweeps of ALL levels: curvature of the first layer behind every special face, several levels per launch pair.
// crse_n[l]: the coarser level's output (normal components from cncomp0) or this rank's coarse-source copy of them.
// nslots > 1: components pcomp .. pcomp + nslots - 1, slot z with outputs at ncomp0 + 8 z / kcomp + 8 z, coarse normals at
// cncomp0 + cn_z z, progress range prog[2 z], prog[2 z + 1] (device): one launch each (SlotK)
// crse_phi[l] (component cpcomp + slot): the coarse progress source behind level l's coarse-fine faces, as handed to
// pa_gradcurv_prep_levels -- the irregular cells of general BoxArrays rebuild ghost normals from it (k_curv_general)
int pa_gradcurv_fix_levels(pa_ctx* ctx, int nlev, pa_mf* const* phi, int pcomp, const pa_mf* const* crse_n, int cncomp0, const int32_t bc[3], double pmin, double pmax,
                           pa_mf* const* out, int ncomp0, int kcomp, double thr, int nslots, const double* prog, int cn_z, const pa_mf* const* crse_phi, int cpcomp) {
  SlotK sk;
  sk.prog = prog;
  sk.cn_z = cn_z;
  const bool use_cp = cpatch_on();
  const bool clip = thr >= 0.0;
  auto gen_lev = [&](int l) {  // level l's arguments of k_curv_general
    const pa_level* L = phi[l]->lev;
    GenLev G;
    G.L = L->view; G.MP = phi[l]->view; G.pcomp = pcomp;
    const pa_mf* cp = crse_phi ? crse_phi[l] : nullptr;
    G.LCp = cp ? cp->lev->view : L->view; G.MCp = cp ? cp->view : phi[l]->view; G.cpcomp = cpcomp;
    G.LCn = crse_n[l] ? crse_n[l]->lev->view : L->view; G.MCn = crse_n[l] ? crse_n[l]->view : phi[l]->view; G.cncomp0 = cncomp0;
    G.MO = out[l]->view; G.ncomp0 = ncomp0; G.kcomp = kcomp;
    for (int d = 0; d < 3; ++d) G.A.bc[d] = bc[d];
    G.A.ratio = 2; G.A.has_crse = (crse_n[l] && cp) ? 1 : 0; G.A.thr = clip ? thr : -1.0; G.A.layers = 1; G.A.perim_only = 0; G.A.pmin = pmin; G.A.invd = 1.0 / (pmax - pmin);
    G.items = nullptr; G.n = 0;
    G.use_cp = (use_cp && crse_n[l] && L->cp_total > 0) ? 1 : 0;
    G.cp_stride = cp_stride(L);
    return G;
  };
  for (int l0 = 0; l0 < nlev; l0 += PA_MAXB) {
    if (use_cp && cpatch_launch(ctx, l0, std::min(nlev, l0 + PA_MAXB), phi, crse_n, cncomp0, 1, nslots, cn_z)) return 1;
    LevBatch<FixArgs> Bt;
    int blev[PA_MAXB];  // hierarchy level of every batch row
    for (int l = l0; l < nlev && l < l0 + PA_MAXB; ++l) {
      const pa_level* L = phi[l]->lev;
      if (L->boxes.empty() || L->sfaces.empty()) continue;
      blev[Bt.n] = l;
      FaceArgs A;
      for (int d = 0; d < 3; ++d) A.bc[d] = bc[d];
      A.ratio = 2; A.has_crse = crse_n[l] ? 1 : 0; A.thr = clip ? thr : -1.0; A.layers = 1; A.perim_only = 1; A.pmin = pmin; A.invd = 1.0 / (pmax - pmin);
      Bt.a[Bt.n] = FixArgs{L->view, phi[l]->view, pcomp, crse_n[l] ? crse_n[l]->lev->view : L->view, crse_n[l] ? crse_n[l]->view : phi[l]->view, cncomp0,
                           out[l]->view, ncomp0, kcomp, A, (use_cp && crse_n[l] && L->cp_total > 0) ? 1 : 0, cg_stride(L), cp_stride(L), (const int2*)L->d_sfwg, L->nsfwg, (const int2*)L->d_pfwg, L->npfwg};
      if (L->ncg_live) {  // this pass's sweep mirrored the first layer behind the special x faces
        if (!clip && nslots == 1) { Bt.a[Bt.n].ncg = L->d_ncg; Bt.a[Bt.n].ncgs = cg_stride(L); Bt.a[Bt.n].ncg_minw = L->ncg_minw; }
        const_cast<pa_level*>(L)->ncg_live = false;
      }
      Bt.ycum[Bt.n + 1] = Bt.ycum[Bt.n] + (int)L->sfaces.size();
      ++Bt.n;
    }
    if (!Bt.n) continue;
    ProfScope prof(ctx, PA_TAG_GRADCURV_FACES);
    bool all_patch = true;  // every level of the batch that interpolates from a coarser level does so from patches
    for (int q = 0; q < Bt.n; ++q) all_patch = all_patch && (Bt.a[q].use_cp || !Bt.a[q].A.has_crse);
    if (Bt.ycum[Bt.n] >= (1 << 24)) return pa_fail(ctx, "pa_gradcurv_fix_levels: too many special faces in one batch");
    unsigned nwgf = 0;
    for (int q = 0; q < Bt.n; ++q) nwgf += (unsigned)Bt.a[q].nwg;
    const dim3 gfast(nwgf, 1, (unsigned)nslots);
    LevChunks Ck;  // the levels' chunk records (the face interiors without the clip: k_faces_fix_chunks)
    Ck.w0[0] = 0;
    unsigned npt = 0;  // the perimeters' work tables
    for (int q = 0; q < Bt.n; ++q) {
      const pa_level* Lq = phi[blev[q]]->lev;
      if (!Lq->d_sfchunk || Lq->nsfchunk <= 0 || !Bt.a[q].pwg || Bt.a[q].npwg <= 0) return pa_fail(ctx, "pa_gradcurv_fix_levels: a level without chunk records / perimeter tables");
      Ck.ck[q] = Lq->d_sfchunk;
      Ck.w0[q + 1] = Ck.w0[q] + (unsigned)Lq->nsfchunk;
      npt += (unsigned)Bt.a[q].npwg;
    }
    const dim3 gtab(npt, 1, (unsigned)nslots);
    if (clip) {
      SlowList sl;
      if (pa_slow_list(ctx, &sl)) return 1;
      PA_HIP(hipMemsetAsync(sl.count, 0, 2 * sizeof(int), ctx->stream));
      if (all_patch) hipLaunchKernelGGL((k_faces_curv_fast<1, true, true>), gfast, dim3(256), 0, ctx->stream, Bt, ctx->d_flags, sl, sk);
      else hipLaunchKernelGGL((k_faces_curv_fast<1, false, true>), gfast, dim3(256), 0, ctx->stream, Bt, ctx->d_flags, sl, sk);
      // the hand-over list is a few thousand cells through a long chain of dependent loads (~0.12 ms whatever its length): the
      // perimeter kernel (other cells, as latency bound) runs next to it on the side stream
      hipStream_t pst = ctx->stream;
      if (ctx->stream2 != ctx->stream) {
        if (!ctx->stream2) PA_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
        for (int e = 0; e < 2; ++e)
          if (!ctx->fix_evs[e]) PA_HIP(hipEventCreateWithFlags(&ctx->fix_evs[e], hipEventDisableTiming));
        pst = ctx->stream2;
        PA_HIP(hipEventRecord(ctx->fix_evs[0], ctx->stream));
        PA_HIP(hipStreamWaitEvent(pst, ctx->fix_evs[0], 0));
      }
      if (all_patch) hipLaunchKernelGGL((k_faces_curv_list<true>), dim3(1024), dim3(256), 0, ctx->stream, Bt, ctx->d_flags, sl, sk);
      else hipLaunchKernelGGL((k_faces_curv_list<false>), dim3(1024), dim3(256), 0, ctx->stream, Bt, ctx->d_flags, sl, sk);
      // general BoxArrays: cells with a valid ghost cell behind a special face whose neighbour's normal the sweep clipped
      for (int q = 0; q < Bt.n; ++q) {
        if (phi[blev[q]]->lev->pure_faces) continue;  // no such cells on this level
        GenLev G = gen_lev(blev[q]);
        G.items = sl.gitems; G.n = sl.gcap; G.ncount = sl.gcount; G.lev = q;
        hipLaunchKernelGGL(k_curv_general<true>, dim3(256), dim3(256), 0, ctx->stream, G, ctx->d_flags, sk);
      }
      if (all_patch) hipLaunchKernelGGL((k_faces_curv_tab<true, true, true>), gtab, dim3(256), 0, pst, Bt, ctx->d_flags, sk);
      else hipLaunchKernelGGL((k_faces_curv_tab<true, false, true>), gtab, dim3(256), 0, pst, Bt, ctx->d_flags, sk);
      if (pst != ctx->stream) {
        PA_HIP(hipEventRecord(ctx->fix_evs[1], pst));
        PA_HIP(hipStreamWaitEvent(ctx->stream, ctx->fix_evs[1], 0));
      }
    } else {
      // The face interiors from the chunk records, the perimeters' work tables IN FRONT of them in the same launch when they are few next
      // to the interiors (large faces: headline 5.906 -> 5.870 ms per pass) or on a rank's share of a sharded hierarchy (two short
      // chains, always together); the many short perimeters of small faces run better as their own launch with their own register
      // budget (irregular hierarchy: 6.50 against 6.60 ms merged).  (The perimeter kernel on a side stream next to the interiors was
      // measured twice and bought nothing: DESIGN_HISTORY.md R2, R4.)
      const bool sharded = nlev > 0 && phi[0]->lev->nranks > 1;
      const bool with_perim = sharded || npt * 8u <= Ck.w0[Bt.n];
      const unsigned np = with_perim ? npt : 0u;
      const dim3 gc(np + Ck.w0[Bt.n], 1, (unsigned)nslots);
      if (all_patch) hipLaunchKernelGGL((k_faces_fix_chunks<true>), gc, dim3(256), 0, ctx->stream, Bt, Ck, ctx->d_flags, sk, np);
      else hipLaunchKernelGGL((k_faces_fix_chunks<false>), gc, dim3(256), 0, ctx->stream, Bt, Ck, ctx->d_flags, sk, np);
      if (!with_perim) {
        if (all_patch) hipLaunchKernelGGL((k_faces_curv_tab<true, true>), gtab, dim3(256), 0, ctx->stream, Bt, ctx->d_flags, sk);
        else hipLaunchKernelGGL((k_faces_curv_tab<true, false>), gtab, dim3(256), 0, ctx->stream, Bt, ctx->d_flags, sk);
      }
    }
  }
  // general BoxArrays: the listed irregular cells, after (and over) whatever the kernels above wrote there; the lists of up to
  // PA_MAXB levels in one launch
  {
    const int phi_nranks = nlev > 0 ? phi[0]->lev->nranks : 1;
    GenBatch Gb;
    Gb.n = 0;
    Gb.wg0[0] = 0;
    auto flush = [&]() {
      if (!Gb.n) return;
      ProfScope prof(ctx, PA_TAG_GRADCURV_FACES);
      const dim3 gg(Gb.wg0[Gb.n], 1, (unsigned)nslots);
      bool light = !clip, patch = true;  // one rank, no clip, every coarse-fine face has its patch: the small variant
      for (int q = 0; q < Gb.n; ++q) {
        light = light && Gb.a[q].L.nboxes > 0 && phi_nranks == 1;
        patch = patch && (Gb.a[q].use_cp || !Gb.a[q].A.has_crse);
      }
      if (light && patch) hipLaunchKernelGGL((k_curv_general_levels<false, false, true>), gg, dim3(256), 0, ctx->stream, Gb, ctx->d_flags, sk);
      else if (clip) hipLaunchKernelGGL((k_curv_general_levels<true, true, false>), gg, dim3(256), 0, ctx->stream, Gb, ctx->d_flags, sk);
      else hipLaunchKernelGGL((k_curv_general_levels<false, true, false>), gg, dim3(256), 0, ctx->stream, Gb, ctx->d_flags, sk);
      Gb.n = 0;
    };
    for (int l = 0; l < nlev; ++l) {
      const pa_level* L = phi[l]->lev;
      if (L->boxes.empty()) continue;
      if (level_irregular(ctx, L)) return 1;
      if (L->nirr == 0) continue;
      GenLev G = gen_lev(l);
      G.items = (const int4*)L->d_irr; G.n = L->nirr;
      Gb.a[Gb.n] = G;
      Gb.wg0[Gb.n + 1] = Gb.wg0[Gb.n] + (unsigned)((L->nirr + 255) / 256);
      if (++Gb.n == PA_MAXB) flush();
    }
    flush();
  }
  PA_HIP(hipGetLastError());
  return 0;
}
