// pa_dist.hip -- one hierarchy sharded over ranks (one rank per GPU), gfx950.
//
// The reference distributes the boxes of every level with DistributionMapping(ba) (grad.cpp:162,
// curvature.cpp:289, filterPlt.cpp:142, isosurface.cpp:1441) and lets AMReX move ghost data between MPI
// ranks: FabArray::FillBoundary for same-level ghost cells, and a ParallelCopy of coarse data under the
// fine level's boundary for every coarse->fine fill (MLMG boundary registers behind grad.cpp:212-213 and
// curvature.cpp:443-445,514-518; FillPatchTwoLevels at filterPlt.cpp:193).  This file is the MI355X form
// of both:
//   * pa_distribution_map: Morton order + equal-volume cuts (what DistributionMapping's SFC strategy does);
//   * FillBoundary plan: for every (destination box, source box, periodic shift) whose two boxes live on
//     different ranks, both sides derive the same region, in the same order;
//   * coarse-source plan: each rank keeps, per fine level, a small BoxArray of disjoint pieces of the coarse
//     level -- exactly the coarse cells the stencils of its fine boxes touch -- and a multifab on it that is
//     refilled from the coarse level's owners; the coarse-fine kernels read that multifab through the same
//     owner-map lookup they use for a local coarse level (AMReX keeps a coarse BndryRegister the same way);
//   * one exchange = one pack launch, ONE grouped point-to-point call for all plans and peers in it, one
//     unpack launch per plan; nothing synchronises the host;
//   * transports: RCCL over xGMI (grouped ncclSend/ncclRecv on the context's stream, loaded with dlopen so
//     that the library also loads where RCCL is absent), or a caller-supplied pa_comm.
#include "pa_internal.h"
#include "pa_dist.h"
#include <algorithm>
#include <array>
#include <cstring>
#include <dlfcn.h>
#include <numeric>

#define PA_TRY(x)           \
  do {                      \
    if ((x) != 0) return 1; \
  } while (0)

// ------------------------------------------------------------------------------------ host geometry
static inline bool bx_isect(const DBox& a, const DBox& b, DBox& r) {
  for (int d = 0; d < 3; ++d) {
    r.lo[d] = std::max(a.lo[d], b.lo[d]);
    r.hi[d] = std::min(a.hi[d], b.hi[d]);
    if (r.lo[d] > r.hi[d]) return false;
  }
  return true;
}
static inline long long bx_cells(const DBox& b) { return (long long)(b.hi[0] - b.lo[0] + 1) * (b.hi[1] - b.lo[1] + 1) * (b.hi[2] - b.lo[2] + 1); }
static inline DBox bx_grow(DBox b, int n) {
  for (int d = 0; d < 3; ++d) { b.lo[d] -= n; b.hi[d] += n; }
  return b;
}
static inline DBox bx_shift(DBox b, const int s[3]) {
  for (int d = 0; d < 3; ++d) { b.lo[d] += s[d]; b.hi[d] += s[d]; }
  return b;
}
static inline int fl2(int i) { return i >> 1; }  // floor(i / 2)
// a \ b appended to out as up to six disjoint boxes (z slabs, then y slabs of the middle, then x pieces)
static void bx_diff(const DBox& a, const DBox& b, std::vector<DBox>& out) {
  DBox I;
  if (!bx_isect(a, b, I)) { out.push_back(a); return; }
  DBox rest = a;
  for (int d = 2; d >= 0; --d) {
    if (rest.lo[d] < I.lo[d]) { DBox p = rest; p.hi[d] = I.lo[d] - 1; out.push_back(p); rest.lo[d] = I.lo[d]; }
    if (rest.hi[d] > I.hi[d]) { DBox p = rest; p.lo[d] = I.hi[d] + 1; out.push_back(p); rest.hi[d] = I.hi[d]; }
  }
}

// periodic shifts (in cells) of a domain, in one fixed order on every rank
static std::vector<std::array<int, 3>> domain_shifts(const int domlo[3], const int domhi[3], const int is_per[3]) {
  std::vector<std::array<int, 3>> sh;
  const int n[3] = {is_per[0] ? 1 : 0, is_per[1] ? 1 : 0, is_per[2] ? 1 : 0};
  for (int a = -n[0]; a <= n[0]; ++a)
    for (int b = -n[1]; b <= n[1]; ++b)
      for (int c = -n[2]; c <= n[2]; ++c)
        sh.push_back({a * (domhi[0] - domlo[0] + 1), b * (domhi[1] - domlo[1] + 1), c * (domhi[2] - domlo[2] + 1)});
  return sh;
}

// "is cell (i,j,k) a valid cell of the level" for a whole BoxArray (host; the owner map of pa_level_create restated
// without a device): 0 valid (maybe through a periodic image), 1 inside the domain and not covered, 2 outside a wall
struct HostGeom {
  int domlo[3], domhi[3], is_per[3];
  int g = 1, mlo[3], mn[3];
  std::vector<int> owner;  // global box index or -1
  HostGeom(const std::vector<DBox>& boxes, const int dl[3], const int dh[3], const int per[3]) {
    int mhi[3];
    for (int d = 0; d < 3; ++d) { domlo[d] = dl[d]; domhi[d] = dh[d]; is_per[d] = per[d] ? 1 : 0; mlo[d] = INT32_MAX; mhi[d] = INT32_MIN; }
    for (const DBox& B : boxes)
      for (int d = 0; d < 3; ++d) { mlo[d] = std::min(mlo[d], B.lo[d]); mhi[d] = std::max(mhi[d], B.hi[d]); }
    int gg = 0;
    for (const DBox& B : boxes)
      for (int d = 0; d < 3; ++d) { gg = std::gcd(gg, B.lo[d] - mlo[d]); gg = std::gcd(gg, B.hi[d] - B.lo[d] + 1); }
    g = gg > 0 ? gg : 1;
    size_t msz = 1;
    for (int d = 0; d < 3; ++d) { mn[d] = (mhi[d] - mlo[d] + 1) / g; msz *= (size_t)mn[d]; }
    owner.assign(msz, -1);
    for (size_t b = 0; b < boxes.size(); ++b) {
      const DBox& B = boxes[b];
      for (int kz = (B.lo[2] - mlo[2]) / g; kz <= (B.hi[2] - mlo[2]) / g; ++kz)
        for (int ky = (B.lo[1] - mlo[1]) / g; ky <= (B.hi[1] - mlo[1]) / g; ++ky)
          for (int kx = (B.lo[0] - mlo[0]) / g; kx <= (B.hi[0] - mlo[0]) / g; ++kx) owner[((size_t)kz * mn[1] + ky) * mn[0] + kx] = (int)b;
    }
  }
  int classify(int i, int j, int k) const {
    int p[3] = {i, j, k};
    for (int d = 0; d < 3; ++d) {
      const int len = domhi[d] - domlo[d] + 1;
      if (p[d] < domlo[d] || p[d] > domhi[d]) {
        if (!is_per[d]) return 2;
        while (p[d] < domlo[d]) p[d] += len;
        while (p[d] > domhi[d]) p[d] -= len;
      }
    }
    int m[3];
    for (int d = 0; d < 3; ++d) {
      const int r = p[d] - mlo[d];
      if (r < 0) return 1;
      m[d] = r / g;
      if (m[d] >= mn[d]) return 1;
    }
    return owner[((size_t)m[2] * mn[1] + m[1]) * mn[0] + m[0]] >= 0 ? 0 : 1;
  }
  bool face_is_special(const DBox& B, int d, int side) const {
    const int t0 = (d == 0) ? 1 : 0, t1 = (d == 2) ? 1 : 2;
    int q[3];
    q[d] = side ? B.hi[d] + 1 : B.lo[d] - 1;
    for (int v = B.lo[t1]; v <= B.hi[t1]; v += g)
      for (int u = B.lo[t0]; u <= B.hi[t0]; u += g) {
        q[t0] = u; q[t1] = v;
        if (classify(q[0], q[1], q[2]) != 0) return true;
      }
    return false;
  }
};

// --------------------------------------------------------------------------------- distribution map
static inline unsigned long long spread3(unsigned long long v) {  // 21 bits -> every third bit
  v &= 0x1fffffull;
  v = (v | v << 32) & 0x1f00000000ffffull;
  v = (v | v << 16) & 0x1f0000ff0000ffull;
  v = (v | v << 8) & 0x100f00f00f00f00full;
  v = (v | v << 4) & 0x10c30c30c30c30c3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}

extern "C" int pa_distribution_map(int nboxes, const int32_t* b6, int nranks, int32_t* owner) {
  if (nboxes <= 0 || !b6 || !owner || nranks < 1) return 1;
  int mlo[3] = {INT32_MAX, INT32_MAX, INT32_MAX};
  for (int b = 0; b < nboxes; ++b)
    for (int d = 0; d < 3; ++d) mlo[d] = std::min(mlo[d], b6[6 * b + d]);
  std::vector<unsigned long long> key(nboxes);
  std::vector<long long> vol(nboxes);
  long long total = 0;
  for (int b = 0; b < nboxes; ++b) {
    key[b] = spread3((unsigned)(b6[6 * b] - mlo[0])) | spread3((unsigned)(b6[6 * b + 1] - mlo[1])) << 1 | spread3((unsigned)(b6[6 * b + 2] - mlo[2])) << 2;
    vol[b] = 1;
    for (int d = 0; d < 3; ++d) vol[b] *= b6[6 * b + 3 + d] - b6[6 * b + d] + 1;
    total += vol[b];
  }
  std::vector<int> ord(nboxes);
  std::iota(ord.begin(), ord.end(), 0);
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key[a] < key[b]; });
  // a box goes to the rank whose share of the curve holds the box's midpoint (in cells along the curve): contiguous
  // pieces of the curve, cell counts equal to within one box
  long long cum = 0;
  for (int i = 0; i < nboxes; ++i) {
    const int b = ord[i];
    const __int128 mid2 = (__int128)2 * cum + vol[b];  // twice the midpoint
    int r = (int)((mid2 * nranks) / ((__int128)2 * total));
    owner[b] = std::min(std::max(r, 0), nranks - 1);
    cum += vol[b];
  }
  return 0;
}

// --------------------------------------------------------------------------------------- host plans
struct HRegion { int peer; int gbox; DBox r; };  // gbox: global box whose index space the region lives in

// FillBoundary: what `rank` sends and receives for ghost width ng.  Enumeration order (dst box, src box, shift) on all ranks.
static void plan_fill_boundary(const std::vector<DBox>& boxes, const std::vector<int>& owner, int rank, const int domlo[3], const int domhi[3],
                               const int is_per[3], int ng, std::vector<HRegion>& send, std::vector<HRegion>& recv) {
  const auto shifts = domain_shifts(domlo, domhi, is_per);
  const int n = (int)boxes.size();
  for (int d = 0; d < n; ++d) {
    const DBox G = bx_grow(boxes[d], ng);
    for (int s = 0; s < n; ++s) {
      if (owner[d] == owner[s] || (owner[d] != rank && owner[s] != rank)) continue;
      // quick reject: the source box must come within ng cells of the destination box, modulo a period
      for (const auto& sh : shifts) {
        DBox I;
        if (!bx_isect(G, bx_shift(boxes[s], sh.data()), I)) continue;
        if (owner[d] == rank) {
          recv.push_back({owner[s], d, I});
        } else {
          const int neg[3] = {-sh[0], -sh[1], -sh[2]};
          send.push_back({owner[d], s, bx_shift(I, neg)});
        }
      }
    }
  }
}

struct CsPiece { int cbox; DBox box; };

// Can box B have IRREGULAR cells (pa_fused.hip: k_find_irregular)?  A test at the granularity of the owner grid, true for every
// box with such a cell (and for a few without): a face that is partly covered by a neighbouring box and partly not, or an
// edge ghost line that is a concave corner of the level (not a valid cell while both its face-ring neighbours are) or a
// valid cell next to a special face.  Pure geometry of the whole BoxArray: every rank gives the same answer for every box.
static bool box_suspect(const HostGeom& FG, const DBox& B) {
  bool special[6];
  for (int d = 0; d < 3; ++d)
    for (int side = 0; side < 2; ++side) {
      const int t0 = (d == 0) ? 1 : 0, t1 = (d == 2) ? 1 : 2;
      int q[3], nval = 0, nnot = 0;
      q[d] = side ? B.hi[d] + 1 : B.lo[d] - 1;
      for (int v = B.lo[t1]; v <= B.hi[t1]; v += FG.g)
        for (int u = B.lo[t0]; u <= B.hi[t0]; u += FG.g) {
          q[t0] = u; q[t1] = v;
          (FG.classify(q[0], q[1], q[2]) == 0 ? nval : nnot)++;
        }
      if (nval && nnot) return true;
      special[d * 2 + side] = nnot > 0;
    }
  for (int a = 0; a < 3; ++a)
    for (int c = a + 1; c < 3; ++c) {
      const int e = 3 - a - c;
      for (int sa = 0; sa < 2; ++sa)
        for (int sc = 0; sc < 2; ++sc)
          for (int t = B.lo[e]; t <= B.hi[e]; t += FG.g) {
            int q[3];
            q[e] = t;
            q[a] = sa ? B.hi[a] + 1 : B.lo[a] - 1;
            q[c] = sc ? B.hi[c] + 1 : B.lo[c] - 1;
            if (FG.classify(q[0], q[1], q[2]) == 0) {
              if (special[a * 2 + sa] || special[c * 2 + sc]) return true;
              continue;
            }
            int qa[3] = {q[0], q[1], q[2]}, qc[3] = {q[0], q[1], q[2]};
            qa[a] += sa ? -1 : 1;
            qc[c] += sc ? -1 : 1;
            if (FG.classify(qa[0], qa[1], qa[2]) == 0 && FG.classify(qc[0], qc[1], qc[2]) == 0) return true;
          }
    }
  return false;
}

// coarse cells the coarse-fine stencils of fine box B touch (coarse index space, may reach outside the domain)
static void cs_rects_of_box(const HostGeom& FG, const DBox& B, int mode, int ng, int halo, std::vector<DBox>& rects, int ratio = 2) {
  if (mode == 0) {
    // A box that may hold irregular cells rebuilds ghost normals as the NEIGHBOURING box sees them (k_curv_general: boundary
    // values of the tangential directions at cells one layer outside the box): the coarse parents of the box grown by one
    // cell, +-2 coarse cells in every direction, minus the coarse cells deep under the box
    if (box_suspect(FG, B)) {
      DBox outer, inner;
      bool has_inner = true;
      for (int t = 0; t < 3; ++t) {
        outer.lo[t] = fl2(B.lo[t] - 1) - 2;
        outer.hi[t] = fl2(B.hi[t] + 1) + 2;
        inner.lo[t] = fl2(B.lo[t] + 1) + 3;
        inner.hi[t] = fl2(B.hi[t] + 1) - 1 - 3;
        has_inner = has_inner && inner.lo[t] <= inner.hi[t];
      }
      if (has_inner) bx_diff(outer, inner, rects);
      else rects.push_back(outer);
      return;
    }
    // pa_apply_bc family (InterpBndryData order 3 behind MLMG applyBC): for a ghost cell q behind face
    // (d, side) the stencil sits in the coarse plane coarsen(q[d]) and spans +-2 coarse cells tangentially around
    // coarsen(q); the edge ghost cells the fused path resolves lie one fine cell beyond the face's tangential extent.
    for (int d = 0; d < 3; ++d)
      for (int side = 0; side < 2; ++side) {
        if (!FG.face_is_special(B, d, side)) continue;
        DBox r;
        for (int t = 0; t < 3; ++t) {
          if (t == d) { r.lo[t] = r.hi[t] = fl2(side ? B.hi[d] + 1 : B.lo[d] - 1); }
          else { r.lo[t] = fl2(B.lo[t] - 1) - 2; r.hi[t] = fl2(B.hi[t] + 1) + 2; }
        }
        rects.push_back(r);
      }
  } else if (mode == 2) {
    // the parents of the box's valid cells, all of them (prolongation of the smoothing solve's multigrid preconditioner)
    DBox r;
    for (int t = 0; t < 3; ++t) { r.lo[t] = coarsen_idx(B.lo[t], ratio); r.hi[t] = coarsen_idx(B.hi[t], ratio); }
    rects.push_back(r);
  } else {
    // pa_fillpatch_two_levels: parents of the ng ghost layers, grown by `halo` coarse cells (slopes / min-max of
    // mf_cell_cons_interp: 1), minus the coarse cells well inside the box
    DBox outer, inner;
    for (int t = 0; t < 3; ++t) {
      outer.lo[t] = coarsen_idx(B.lo[t] - ng, ratio) - halo;
      outer.hi[t] = coarsen_idx(B.hi[t] + ng, ratio) + halo;
      inner.lo[t] = coarsen_idx(B.lo[t] + ratio - 1, ratio) + halo;   // first coarse cell with all children inside, moved in by halo
      inner.hi[t] = coarsen_idx(B.hi[t] + 1, ratio) - 1 - halo;
    }
    bool has_inner = true;
    for (int t = 0; t < 3; ++t) has_inner = has_inner && inner.lo[t] <= inner.hi[t];
    if (has_inner) bx_diff(outer, inner, rects);
    else rects.push_back(outer);
  }
}

// the disjoint pieces of the coarse level that rank r keeps a copy of, in the order of r's coarse-source BoxArray
static std::vector<CsPiece> cs_pieces(const HostGeom& FG, const std::vector<DBox>& fboxes, const std::vector<int>& fowner, const std::vector<DBox>& cboxes,
                                      const int cdomlo[3], const int cdomhi[3], const int is_per[3], int r, int mode, int ng, int halo, int ratio = 2) {
  std::vector<DBox> rects;
  for (size_t b = 0; b < fboxes.size(); ++b)
    if (fowner[b] == r) cs_rects_of_box(FG, fboxes[b], mode, ng, halo, rects, ratio);
  std::vector<CsPiece> out;
  if (rects.empty()) return out;
  DBox hull = rects[0];
  for (const DBox& q : rects)
    for (int d = 0; d < 3; ++d) { hull.lo[d] = std::min(hull.lo[d], q.lo[d]); hull.hi[d] = std::max(hull.hi[d], q.hi[d]); }
  const auto shifts = domain_shifts(cdomlo, cdomhi, is_per);
  std::vector<DBox> mine, tmp, tmp2;
  for (size_t j = 0; j < cboxes.size(); ++j) {
    mine.clear();
    for (const auto& sh : shifts) {
      const DBox Cs = bx_shift(cboxes[j], sh.data());
      DBox I;
      if (!bx_isect(hull, Cs, I)) continue;
      const int neg[3] = {-sh[0], -sh[1], -sh[2]};
      for (const DBox& q : rects) {
        if (!bx_isect(q, Cs, I)) continue;
        tmp.assign(1, bx_shift(I, neg));  // in the coarse box's own coordinates
        for (const DBox& have : mine) {
          tmp2.clear();
          for (const DBox& t : tmp) bx_diff(t, have, tmp2);
          tmp.swap(tmp2);
          if (tmp.empty()) break;
        }
        for (const DBox& t : tmp) mine.push_back(t);
      }
    }
    for (const DBox& t : mine) out.push_back({(int)j, t});
  }
  return out;
}

static std::vector<DBox> boxes_from6(int n, const int32_t* b6) {
  std::vector<DBox> v(n);
  for (int b = 0; b < n; ++b)
    for (int d = 0; d < 3; ++d) { v[b].lo[d] = b6[6 * b + d]; v[b].hi[d] = b6[6 * b + 3 + d]; }
  return v;
}
static inline void put_row(int32_t* rows, int64_t cap, int64_t& n, int kind, int peer, int box, const DBox& r) {
  if (n < cap && rows) {
    int32_t* p = rows + 9 * n;
    p[0] = kind; p[1] = peer; p[2] = box;
    for (int d = 0; d < 3; ++d) { p[3 + d] = r.lo[d]; p[6 + d] = r.hi[d]; }
  }
  ++n;
}

extern "C" int64_t pa_plan_fill_boundary(int nboxes, const int32_t* b6, const int32_t* owner, int rank, const int32_t domlo[3], const int32_t domhi[3],
                                         const int32_t is_per[3], int ng, int32_t* rows9, int64_t cap) {
  if (nboxes <= 0 || !b6 || !owner) return -1;
  std::vector<HRegion> send, recv;
  plan_fill_boundary(boxes_from6(nboxes, b6), std::vector<int>(owner, owner + nboxes), rank, domlo, domhi, is_per, ng, send, recv);
  int64_t n = 0;
  for (const HRegion& h : send) put_row(rows9, cap, n, 0, h.peer, h.gbox, h.r);
  for (const HRegion& h : recv) put_row(rows9, cap, n, 1, h.peer, h.gbox, h.r);
  return n;
}

extern "C" int64_t pa_plan_coarse_source(int nfine, const int32_t* fb6, const int32_t* fowner, const int32_t fdomlo[3], const int32_t fdomhi[3], int ncrse,
                                         const int32_t* cb6, const int32_t* cowner, const int32_t cdomlo[3], const int32_t cdomhi[3],
                                         const int32_t is_per[3], int rank, int mode, int ng, int halo, int32_t* rows9, int64_t cap) {
  if (nfine <= 0 || ncrse <= 0 || !fb6 || !fowner || !cb6 || !cowner) return -1;
  const std::vector<DBox> fb = boxes_from6(nfine, fb6), cb = boxes_from6(ncrse, cb6);
  const std::vector<int> fo(fowner, fowner + nfine);
  const HostGeom FG(fb, fdomlo, fdomhi, is_per);
  int nranks = 0;
  for (int v : fo) nranks = std::max(nranks, v + 1);
  for (int j = 0; j < ncrse; ++j) nranks = std::max(nranks, cowner[j] + 1);
  int64_t n = 0;
  for (const CsPiece& p : cs_pieces(FG, fb, fo, cb, cdomlo, cdomhi, is_per, rank, mode, ng, halo)) put_row(rows9, cap, n, 2, cowner[p.cbox], p.cbox, p.box);
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) continue;
    for (const CsPiece& p : cs_pieces(FG, fb, fo, cb, cdomlo, cdomhi, is_per, r, mode, ng, halo))
      if (cowner[p.cbox] == rank) put_row(rows9, cap, n, 0, r, p.cbox, p.box);
  }
  return n;
}

// ------------------------------------------------------------------------------------- device plans
XSide::~XSide() {
  if (d_regs) (void)hipFree(d_regs);
  if (d_coff) (void)hipFree(d_coff);
}
XPlan::~XPlan() {
  if (sbuf) (void)hipFree(sbuf);
  if (rbuf) (void)hipFree(rbuf);
  if (d_lsrc) (void)hipFree(d_lsrc);
  if (d_ldst) (void)hipFree(d_ldst);
}
CsPlan::~CsPlan() {
  for (auto& kv : mfs) pa_mf_destroy(kv.second);
  if (cs) pa_level_destroy(cs);
}
pa_level::~pa_level() {
  for (auto& kv : scratch) pa_mf_destroy(kv.second);
}
extern "C" int64_t pa_level_free_scratch(pa_level* L) {
  if (!L) return 0;
  PaBind bind_(L->ctx);
  if (L->ctx && L->ctx->stream) (void)hipStreamSynchronize(L->ctx->stream);  // a kernel still in flight may be using them
  int64_t bytes = 0;
  for (auto& kv : L->scratch) {
    bytes += kv.second ? 8 * (int64_t)kv.second->total : 0;
    pa_mf_destroy(kv.second);
  }
  L->scratch.clear();
  return bytes;
}
pa_mf* pa_level_scratch(pa_ctx* ctx, const pa_level* L, int ncomp, int ng, int role) {
  const std::array<int, 3> key{ncomp, ng, role};
  auto it = L->scratch.find(key);
  pa_mf* m = it != L->scratch.end() ? it->second : pa_mf_create(ctx, L, ncomp, ng, nullptr);
  if (m && it == L->scratch.end()) L->scratch[key] = m;
  // the contents are undefined between calls (what the previous call left): PA_SCRATCH_POISON=1 makes that visible -- all bits set = NaN
  if (m && pa_opt().scratch_poison && m->total > 0 && hipMemsetAsync(m->data, 0xFF, sizeof(double) * (size_t)m->total, ctx->stream) != hipSuccess) (void)hipGetLastError();
  return m;
}

// regions (already sorted by peer, stable) -> host + device tables
static int side_finish(pa_ctx* ctx, XSide& S, std::vector<std::pair<int, std::array<int32_t, 7>>>& regs) {
  std::stable_sort(regs.begin(), regs.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  S.regs7.clear(); S.coff.assign(1, 0); S.peers.clear(); S.first.clear();
  for (size_t i = 0; i < regs.size(); ++i) {
    if (S.peers.empty() || S.peers.back() != regs[i].first) { S.peers.push_back(regs[i].first); S.first.push_back((int)i); }
    const auto& R = regs[i].second;
    S.regs7.insert(S.regs7.end(), R.begin(), R.end());
    const long long n = (long long)(R[4] - R[1] + 1) * (R[5] - R[2] + 1) * (R[6] - R[3] + 1);
    S.maxcells = std::max(S.maxcells, n);
    S.coff.push_back(S.coff.back() + n);
  }
  S.first.push_back((int)regs.size());
  if (!regs.empty()) {
    PA_HIP(hipMalloc(&S.d_regs, sizeof(int) * S.regs7.size()));
    PA_HIP(hipMalloc(&S.d_coff, sizeof(long long) * S.coff.size()));
    PA_HIP(hipMemcpy(S.d_regs, S.regs7.data(), sizeof(int) * S.regs7.size(), hipMemcpyHostToDevice));
    PA_HIP(hipMemcpy(S.d_coff, S.coff.data(), sizeof(long long) * S.coff.size(), hipMemcpyHostToDevice));
  }
  return 0;
}
static std::array<int32_t, 7> reg7(int box, const DBox& r) { return {box, r.lo[0], r.lo[1], r.lo[2], r.hi[0], r.hi[1], r.hi[2]}; }

FbLocal::~FbLocal() {
  if (d_regs) (void)hipFree(d_regs);
  if (d_wgs) (void)hipFree(d_wgs);
}
FbLocal* pa_fb_local_plan(pa_ctx* ctx, const pa_level* L, int ng) {
  auto it = L->fb_local.find(ng);
  if (it != L->fb_local.end()) return it->second.get();
  std::unique_ptr<FbLocal> P(new FbLocal());
  FbLocal* raw = P.get();
  L->fb_local[ng] = std::move(P);
  const int n = (int)L->boxes.size();
  const int on = !pa_opt().force_fallbacks;
  if (!on || n == 0) return raw;
  const auto shifts = domain_shifts(L->domlo, L->domhi, L->is_per);
  std::vector<int> regs, wgs, cand;
  // source boxes that can reach (G - shift): from the level's owner grid (cells of g^3), or all boxes when that grid is
  // too fine to walk (ragged BoxArrays) and the level is small enough for the O(n^2) search
  const DBox M = {{L->mlo[0], L->mlo[1], L->mlo[2]}, {L->mlo[0] + L->mn[0] * L->g - 1, L->mlo[1] + L->mn[1] * L->g - 1, L->mlo[2] + L->mn[2] * L->g - 1}};
  auto candidates = [&](const DBox& Q) -> bool {
    cand.clear();
    DBox I;
    if (!bx_isect(Q, M, I)) return true;
    int c0[3], c1[3];
    long long cells = 1;
    for (int d = 0; d < 3; ++d) { c0[d] = (I.lo[d] - L->mlo[d]) / L->g; c1[d] = (I.hi[d] - L->mlo[d]) / L->g; cells *= c1[d] - c0[d] + 1; }
    if (cells > 4096) return false;
    for (int kz = c0[2]; kz <= c1[2]; ++kz)
      for (int ky = c0[1]; ky <= c1[1]; ++ky)
        for (int kx = c0[0]; kx <= c1[0]; ++kx) {
          const int o = L->owner[((size_t)kz * L->mn[1] + ky) * L->mn[0] + kx];
          if (o >= 0 && std::find(cand.begin(), cand.end(), o) == cand.end()) cand.push_back(o);
        }
    std::sort(cand.begin(), cand.end());  // (dst box, src box, shift) order, whatever the search
    return true;
  };
  for (int d = 0; d < n; ++d) {
    const DBox G = bx_grow(L->boxes[d], ng);
    for (const auto& sh : shifts) {
      const int neg[3] = {-sh[0], -sh[1], -sh[2]};
      if (!candidates(bx_shift(G, neg))) {
        if (n > 1024) return raw;
        cand.resize(n);
        std::iota(cand.begin(), cand.end(), 0);
      }
      for (int s : cand) {
        if (s == d && sh[0] == 0 && sh[1] == 0 && sh[2] == 0) continue;
        DBox I;
        if (!bx_isect(G, bx_shift(L->boxes[s], sh.data()), I)) continue;
        const long long n0 = I.hi[0] - I.lo[0] + 1, n1 = I.hi[1] - I.lo[1] + 1, n2 = I.hi[2] - I.lo[2] + 1, nc = n0 * n1 * n2;
        if (nc >= (1LL << 21) || n0 >= 2048 || n1 >= 2048) return raw;  // q = umulhi(t, ceil(2^32 / d)) is exact for t d < 2^32
        const int r = (int)(regs.size() / 16);
        const unsigned m0 = (unsigned)(((1ULL << 32) + n0 - 1) / n0), m1 = (unsigned)(((1ULL << 32) + n1 - 1) / n1);
        const int row[16] = {d, s, I.lo[0], I.lo[1], I.lo[2], (int)n0, (int)n1, (int)n2, sh[0], sh[1], sh[2], (int)nc, (int)m0, (int)m1, 0, 0};
        regs.insert(regs.end(), row, row + 16);
        for (int c = 0; c < (int)((nc + 255) / 256); ++c) { wgs.push_back(r); wgs.push_back(c); }
      }
    }
  }
  raw->nreg = (int)(regs.size() / 16);
  raw->nwg = (int)(wgs.size() / 2);
  if (raw->nreg > 0) {
    if (hipMalloc(&raw->d_regs, sizeof(int) * regs.size()) != hipSuccess || hipMalloc(&raw->d_wgs, sizeof(int) * wgs.size()) != hipSuccess) return raw;
    if (hipMemcpy(raw->d_regs, regs.data(), sizeof(int) * regs.size(), hipMemcpyHostToDevice) != hipSuccess) return raw;
    if (hipMemcpy(raw->d_wgs, wgs.data(), sizeof(int) * wgs.size(), hipMemcpyHostToDevice) != hipSuccess) return raw;
  }
  (void)ctx;
  raw->ok = true;
  return raw;
}

CpPlan::~CpPlan() {
  if (d_regs) (void)hipFree(d_regs);
  if (d_wgs) (void)hipFree(d_wgs);
}
CpPlan* pa_cp_plan(pa_ctx* ctx, const pa_level* F, const pa_level* C) {
  auto it = F->cp_plans.find(C->serial);
  if (it != F->cp_plans.end()) return it->second.get();
  std::unique_ptr<CpPlan> P(new CpPlan());
  CpPlan* raw = P.get();
  F->cp_plans[C->serial] = std::move(P);
  const int on = !pa_opt().force_fallbacks;
  const int nc = (int)C->boxes.size();
  if (!on || nc == 0 || F->sfaces.empty()) return raw;
  const auto shifts = domain_shifts(C->domlo, C->domhi, C->is_per);
  const DBox M = {{C->mlo[0], C->mlo[1], C->mlo[2]}, {C->mlo[0] + C->mn[0] * C->g - 1, C->mlo[1] + C->mn[1] * C->g - 1, C->mlo[2] + C->mn[2] * C->g - 1}};
  std::vector<int> regs, wgs, cand;
  for (size_t e = 0; e < F->sfaces.size(); ++e) {
    const int f = F->sfaces[e], dir = (f % 6) >> 1, side = f & 1;
    const DBox& B = F->boxes[f / 6];
    const int qd = side ? B.hi[dir] + 1 : B.lo[dir] - 1;
    if (!(F->is_per[dir] || (qd >= F->domlo[dir] && qd <= F->domhi[dir]))) continue;  // a wall face: no patch (pa_level_create)
    const int t0 = dir == 0 ? 1 : 0, t1 = dir == 2 ? 1 : 2;
    int plane, u0, v0, pw, ph;
    cpatch_geom(B, dir, side, plane, u0, v0, pw, ph);
    DBox R;  // the patch in the coarse index space (unwrapped)
    R.lo[dir] = R.hi[dir] = plane;
    R.lo[t0] = u0; R.hi[t0] = u0 + pw - 1;
    R.lo[t1] = v0; R.hi[t1] = v0 + ph - 1;
    for (const auto& sh : shifts) {  // coarse box s shifted by sh covers part of R  <=>  s covers part of R - sh
      const int neg[3] = {-sh[0], -sh[1], -sh[2]};
      const DBox Q = bx_shift(R, neg);
      DBox I;
      cand.clear();
      if (bx_isect(Q, M, I)) {
        long long cells = 1;
        int c0[3], c1[3];
        for (int d = 0; d < 3; ++d) { c0[d] = (I.lo[d] - C->mlo[d]) / C->g; c1[d] = (I.hi[d] - C->mlo[d]) / C->g; cells *= c1[d] - c0[d] + 1; }
        if (cells > 65536) return raw;
        for (int kz = c0[2]; kz <= c1[2]; ++kz)
          for (int ky = c0[1]; ky <= c1[1]; ++ky)
            for (int kx = c0[0]; kx <= c1[0]; ++kx) {
              const int o = C->owner[((size_t)kz * C->mn[1] + ky) * C->mn[0] + kx];
              if (o >= 0 && std::find(cand.begin(), cand.end(), o) == cand.end()) cand.push_back(o);
            }
      }
      for (int sbox : cand) {
        if (!bx_isect(Q, C->boxes[sbox], I)) continue;  // I: coarse cells of box sbox (its own coordinates)
        const long long nu = I.hi[t0] - I.lo[t0] + 1, nv = I.hi[t1] - I.lo[t1] + 1;
        if (nu * nv >= (1LL << 21) || nu >= 2048) return raw;
        const int r = (int)(regs.size() / 12);
        const unsigned m = (unsigned)(((1ULL << 32) + nu - 1) / nu);
        const int row[12] = {(int)e, sbox, I.lo[t0] + sh[t0] - u0, I.lo[t1] + sh[t1] - v0, I.lo[0], I.lo[1], I.lo[2], (int)nu, (int)nv, dir, (int)m, 0};
        regs.insert(regs.end(), row, row + 12);
        for (int c = 0; c < (int)((nu * nv + 255) / 256); ++c) { wgs.push_back(r); wgs.push_back(c); }
      }
    }
  }
  raw->nreg = (int)(regs.size() / 12);
  raw->nwg = (int)(wgs.size() / 2);
  if (raw->nreg > 0) {
    if (hipMalloc(&raw->d_regs, sizeof(int) * regs.size()) != hipSuccess || hipMalloc(&raw->d_wgs, sizeof(int) * wgs.size()) != hipSuccess) return raw;
    if (hipMemcpy(raw->d_regs, regs.data(), sizeof(int) * regs.size(), hipMemcpyHostToDevice) != hipSuccess) return raw;
    if (hipMemcpy(raw->d_wgs, wgs.data(), sizeof(int) * wgs.size(), hipMemcpyHostToDevice) != hipSuccess) return raw;
  }
  (void)ctx;
  raw->hregs = std::move(regs);
  raw->ok = true;
  return raw;
}

XPlan* pa_fb_plan(pa_ctx* ctx, const pa_level* L, int ng) {
  auto it = L->fb_plans.find(ng);
  if (it != L->fb_plans.end()) return it->second.get();
  std::vector<HRegion> send, recv;
  plan_fill_boundary(L->gboxes, L->gowner, L->rank, L->domlo, L->domhi, L->is_per, ng, send, recv);
  std::unique_ptr<XPlan> P(new XPlan());
  std::vector<std::pair<int, std::array<int32_t, 7>>> s, r;
  for (const HRegion& h : send) s.push_back({h.peer, reg7(L->glocal[h.gbox], h.r)});
  for (const HRegion& h : recv) r.push_back({h.peer, reg7(L->glocal[h.gbox], h.r)});
  if (side_finish(ctx, P->send, s) || side_finish(ctx, P->recv, r)) return nullptr;
  XPlan* raw = P.get();
  L->fb_plans[ng] = std::move(P);
  return raw;
}

RepPlan::~RepPlan() {
  if (rep) pa_level_destroy(rep);
}
RepPlan* pa_rep_plan(pa_ctx* ctx, const pa_level* L) {
  if (L->rep_plan) return L->rep_plan.get();
  if (L->nranks <= 1 || L->gboxes.empty()) { pa_fail(ctx, "pa_rep_plan: the level is not sharded"); return nullptr; }
  std::unique_ptr<RepPlan> P(new RepPlan());
  LevelSpec S;
  S.local = L->gboxes;  // the whole BoxArray, unsharded: box index = global index
  P->rep = pa_level_create_spec(ctx, S, L->domlo, L->domhi, L->is_per, L->prob_lo, L->prob_hi);
  if (!P->rep) return nullptr;
  std::vector<std::pair<int, std::array<int32_t, 7>>> s, r, none;
  std::vector<int32_t> own_s, own_g;  // this rank's boxes: (local index, box) and (global index, box)
  for (size_t b = 0; b < L->boxes.size(); ++b) {
    for (int q = 0; q < L->nranks; ++q)
      if (q != L->rank) s.push_back({q, reg7((int)b, L->boxes[b])});  // side_finish groups by peer and keeps the box order
    const auto a = reg7((int)b, L->boxes[b]), g = reg7(L->gid[b], L->boxes[b]);
    own_s.insert(own_s.end(), a.begin(), a.end());
    own_g.insert(own_g.end(), g.begin(), g.end());
    P->gather.lmax = std::max(P->gather.lmax, bx_cells(L->boxes[b]));
  }
  for (size_t g = 0; g < L->gboxes.size(); ++g)
    if (L->gowner[g] != L->rank) r.push_back({L->gowner[g], reg7((int)g, L->gboxes[g])});  // ascending global index = the sender's box order
  if (side_finish(ctx, P->gather.send, s) || side_finish(ctx, P->gather.recv, r)) return nullptr;
  if (side_finish(ctx, P->back.send, none) || side_finish(ctx, P->back.recv, none)) return nullptr;
  P->back.lmax = P->gather.lmax;
  P->gather.nlocal = P->back.nlocal = (int)L->boxes.size();
  if (!L->boxes.empty()) {
    const size_t bytes = sizeof(int) * own_s.size();
    if (hipMalloc(&P->gather.d_lsrc, bytes) != hipSuccess || hipMalloc(&P->gather.d_ldst, bytes) != hipSuccess || hipMalloc(&P->back.d_lsrc, bytes) != hipSuccess ||
        hipMalloc(&P->back.d_ldst, bytes) != hipSuccess || hipMemcpy(P->gather.d_lsrc, own_s.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(P->gather.d_ldst, own_g.data(), bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(P->back.d_lsrc, own_g.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(P->back.d_ldst, own_s.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
      pa_fail(ctx, "pa_rep_plan: device allocation failed");
      return nullptr;
    }
  }
  L->rep_plan = std::move(P);
  return L->rep_plan.get();
}

// ------------------------------------------------------------------ restriction onto a sharded coarse level
RsPlan::~RsPlan() {
  if (cf) pa_level_destroy(cf);
}
// global boxes of `G`'s BoxArray that may intersect Q, ascending (from the owner grid; all boxes when Q spans too much of it)
static void boxes_near(const HostGeom& G, int nboxes, const DBox& Q, std::vector<int>& out) {
  out.clear();
  int c0[3], c1[3];
  long long cells = 1;
  for (int d = 0; d < 3; ++d) {
    const int lo = std::max(Q.lo[d], G.mlo[d]), hi = std::min(Q.hi[d], G.mlo[d] + G.mn[d] * G.g - 1);
    if (lo > hi) return;
    c0[d] = (lo - G.mlo[d]) / G.g; c1[d] = (hi - G.mlo[d]) / G.g;
    cells *= c1[d] - c0[d] + 1;
  }
  if (cells > 4096) {
    for (int b = 0; b < nboxes; ++b) out.push_back(b);
    return;
  }
  for (int kz = c0[2]; kz <= c1[2]; ++kz)
    for (int ky = c0[1]; ky <= c1[1]; ++ky)
      for (int kx = c0[0]; kx <= c1[0]; ++kx) {
        const int o = G.owner[((size_t)kz * G.mn[1] + ky) * G.mn[0] + kx];
        if (o >= 0) out.push_back(o);
      }
  std::sort(out.begin(), out.end());
  out.erase(std::unique(out.begin(), out.end()), out.end());
}
// The region lists of one restriction plan as host arithmetic (which = 0: `down`, 1 + 2 * dir + side: that flux plan), for rank
// `me`: what it sends (fine global box, region in the coarsened fine level's index space), what it receives (coarse global box,
// region in the coarse level's index space = the sender's region minus the periodic shift) and its same-rank pairs.  Every rank
// walks (fine box, shift, coarse box) in the same order, so a sender's list to a peer and that peer's list from it pair up.
struct RsRegions {
  std::vector<std::pair<int, std::array<int32_t, 7>>> send, recv;  // (peer, {global box, lo, hi})
  std::vector<std::array<int32_t, 7>> lsrc, ldst;
};
static std::vector<DBox> rs_coarsen(const std::vector<DBox>& fb, const int fdomlo[3], const int fdomhi[3], int ratio) {
  const int r3[3] = {ratio, ratio, fdomlo[2] == fdomhi[2] ? 1 : ratio};  // a 2-D hierarchy is one plane of cells per level
  auto fdiv = [](int a, int r) { return a >= 0 ? a / r : -((-a + r - 1) / r); };
  std::vector<DBox> out;
  for (const DBox& B : fb) {
    DBox c;
    for (int d = 0; d < 3; ++d) { c.lo[d] = fdiv(B.lo[d], r3[d]); c.hi[d] = fdiv(B.hi[d], r3[d]); }
    out.push_back(c);
  }
  return out;
}
static RsRegions rs_regions(const HostGeom& FG, const std::vector<DBox>& fb, const std::vector<int>& fowner, const std::vector<DBox>& cfb, const HostGeom& CG,
                            const std::vector<DBox>& cb, const std::vector<int>& cowner, int me, int which) {
  RsRegions R;
  const int nF = (int)fb.size(), nC = (int)cb.size();
  std::vector<int> cand;
  auto emit = [&](int gf, const DBox& Q, const std::array<int, 3>& sh) {
    const int neg[3] = {-sh[0], -sh[1], -sh[2]}, pos[3] = {sh[0], sh[1], sh[2]};
    const DBox Qc = bx_shift(Q, neg);  // in the coarse level's index space
    boxes_near(CG, nC, Qc, cand);
    for (int gc : cand) {
      DBox I;
      if (!bx_isect(Qc, cb[gc], I)) continue;
      const DBox Is = bx_shift(I, pos);
      const int fo = fowner[gf], co = cowner[gc];
      if (fo == me && co == me) { R.lsrc.push_back(reg7(gf, Is)); R.ldst.push_back(reg7(gc, I)); }
      else if (fo == me) R.send.push_back({co, reg7(gf, Is)});
      else if (co == me) R.recv.push_back({fo, reg7(gc, I)});
    }
  };
  if (which == 0) {
    const std::array<int, 3> zero = {0, 0, 0};
    for (int gf = 0; gf < nF; ++gf) emit(gf, cfb[gf], zero);  // fine boxes lie inside the domain: no periodic image
  } else {
    const int dir = (which - 1) >> 1, side = (which - 1) & 1;
    const auto shifts = domain_shifts(CG.domlo, CG.domhi, CG.is_per);
    for (int gf = 0; gf < nF; ++gf) {
      if (!FG.face_is_special(fb[gf], dir, side)) continue;
      DBox Q = cfb[gf];
      Q.lo[dir] = Q.hi[dir] = side ? Q.hi[dir] + 1 : Q.lo[dir] - 1;
      for (const auto& sh : shifts) emit(gf, Q, sh);
    }
  }
  return R;
}
extern "C" int64_t pa_plan_restriction(int nfine, const int32_t* fb6, const int32_t* fowner, const int32_t fdomlo[3], const int32_t fdomhi[3], int ncrse,
                                       const int32_t* cb6, const int32_t* cowner, const int32_t cdomlo[3], const int32_t cdomhi[3], const int32_t is_per[3],
                                       int rank, int ratio, int which, int32_t* rows9, int64_t cap) {
  if (nfine <= 0 || ncrse <= 0 || !fb6 || !fowner || !cb6 || !cowner || which < 0 || which > 6 || ratio < 2) return -1;
  const std::vector<DBox> fb = boxes_from6(nfine, fb6), cb = boxes_from6(ncrse, cb6);
  const HostGeom FG(fb, fdomlo, fdomhi, is_per), CG(cb, cdomlo, cdomhi, is_per);
  const RsRegions R = rs_regions(FG, fb, std::vector<int>(fowner, fowner + nfine), rs_coarsen(fb, fdomlo, fdomhi, ratio), CG, cb, std::vector<int>(cowner, cowner + ncrse), rank, which);
  auto box_of = [](const std::array<int32_t, 7>& a) { DBox b; for (int d = 0; d < 3; ++d) { b.lo[d] = a[1 + d]; b.hi[d] = a[4 + d]; } return b; };
  int64_t n = 0;
  for (const auto& s : R.send) put_row(rows9, cap, n, 0, s.first, s.second[0], box_of(s.second));
  for (const auto& r : R.recv) put_row(rows9, cap, n, 1, r.first, r.second[0], box_of(r.second));
  for (size_t i = 0; i < R.lsrc.size(); ++i) {
    put_row(rows9, cap, n, 3, rank, R.lsrc[i][0], box_of(R.lsrc[i]));
    put_row(rows9, cap, n, 4, rank, R.ldst[i][0], box_of(R.ldst[i]));
  }
  return n;
}

static int xplan_locals(pa_ctx* ctx, XPlan& X, const std::vector<int32_t>& lsrc, const std::vector<int32_t>& ldst) {
  X.nlocal = (int)(lsrc.size() / 7);
  if (!X.nlocal) return 0;
  PA_HIP(hipMalloc(&X.d_lsrc, sizeof(int) * lsrc.size()));
  PA_HIP(hipMalloc(&X.d_ldst, sizeof(int) * ldst.size()));
  PA_HIP(hipMemcpy(X.d_lsrc, lsrc.data(), sizeof(int) * lsrc.size(), hipMemcpyHostToDevice));
  PA_HIP(hipMemcpy(X.d_ldst, ldst.data(), sizeof(int) * ldst.size(), hipMemcpyHostToDevice));
  return 0;
}
RsPlan* pa_rs_plan(pa_ctx* ctx, const pa_level* F, const pa_level* C, int ratio) {
  auto it = F->rs_plans.find(C->serial);
  if (it != F->rs_plans.end()) return it->second.get();
  if (F->nranks <= 1 || F->gboxes.empty() || C->gboxes.empty()) { pa_fail(ctx, "pa_rs_plan: the levels are not sharded"); return nullptr; }
  if (F->nranks != C->nranks || F->rank != C->rank) { pa_fail(ctx, "coarse and fine level are sharded over different rank sets"); return nullptr; }
  std::unique_ptr<RsPlan> P(new RsPlan());
  const std::vector<DBox> cfb = rs_coarsen(F->gboxes, F->domlo, F->domhi, ratio);
  LevelSpec S;
  S.source_only = true;
  S.rank = F->rank; S.nranks = F->nranks;
  S.gid = F->gid; S.gowner = F->gowner;
  S.gboxes = cfb;
  for (int g : F->gid) S.local.push_back(cfb[g]);
  P->cf = pa_level_create_spec(ctx, S, C->domlo, C->domhi, C->is_per, C->prob_lo, C->prob_hi);
  if (!P->cf) return nullptr;
  const HostGeom FG(F->gboxes, F->domlo, F->domhi, F->is_per), CG(C->gboxes, C->domlo, C->domhi, C->is_per);
  for (int which = 0; which <= 6; ++which) {
    XPlan& X = which ? P->flux[which - 1] : P->down;
    RsRegions R = rs_regions(FG, F->gboxes, F->gowner, cfb, CG, C->gboxes, C->gowner, F->rank, which);
    std::vector<int32_t> lsrc, ldst;
    for (auto& s : R.send) s.second[0] = F->glocal[s.second[0]];  // global -> this rank's box indices
    for (auto& r : R.recv) r.second[0] = C->glocal[r.second[0]];
    for (size_t i = 0; i < R.lsrc.size(); ++i) {
      R.lsrc[i][0] = F->glocal[R.lsrc[i][0]];
      R.ldst[i][0] = C->glocal[R.ldst[i][0]];
      lsrc.insert(lsrc.end(), R.lsrc[i].begin(), R.lsrc[i].end());
      ldst.insert(ldst.end(), R.ldst[i].begin(), R.ldst[i].end());
      X.lmax = std::max(X.lmax, (long long)(R.ldst[i][4] - R.ldst[i][1] + 1) * (R.ldst[i][5] - R.ldst[i][2] + 1) * (R.ldst[i][6] - R.ldst[i][3] + 1));
    }
    if (side_finish(ctx, X.send, R.send) || side_finish(ctx, X.recv, R.recv) || xplan_locals(ctx, X, lsrc, ldst)) return nullptr;
  }
  RsPlan* raw = P.get();
  F->rs_plans[C->serial] = std::move(P);
  return raw;
}

CsPlan* pa_cs_plan(pa_ctx* ctx, const pa_level* F, const pa_level* C, int mode, int ng, int halo, int ratio) {
  if (mode == 0 && ratio != 2) { pa_fail(ctx, "coarse-source plan of the MLMG boundary: refinement ratio 2 only"); return nullptr; }
  const auto key = std::make_pair(C->serial, mode == 0 ? 0 : 1 + ng * 16 + halo + 4096 * ratio + (mode == 2 ? (1 << 20) : 0));
  auto it = F->cs_plans.find(key);
  if (it != F->cs_plans.end()) return it->second.get();
  if (F->nranks != C->nranks || F->rank != C->rank) { pa_fail(ctx, "coarse and fine level are sharded over different rank sets"); return nullptr; }
  const HostGeom FG(F->gboxes, F->domlo, F->domhi, F->is_per);
  std::unique_ptr<CsPlan> P(new CsPlan());
  const std::vector<CsPiece> mine = cs_pieces(FG, F->gboxes, F->gowner, C->gboxes, C->domlo, C->domhi, C->is_per, F->rank, mode, ng, halo, ratio);
  LevelSpec S;
  S.source_only = true;
  std::vector<std::pair<int, std::array<int32_t, 7>>> s, r;
  std::vector<int32_t> lsrc, ldst;
  for (size_t i = 0; i < mine.size(); ++i) {
    S.local.push_back(mine[i].box);
    const int o = C->gowner[mine[i].cbox];
    if (o == C->rank) {
      const auto a = reg7(C->glocal[mine[i].cbox], mine[i].box), b = reg7((int)i, mine[i].box);
      lsrc.insert(lsrc.end(), a.begin(), a.end());
      ldst.insert(ldst.end(), b.begin(), b.end());
      P->x.lmax = std::max(P->x.lmax, bx_cells(mine[i].box));
    } else {
      r.push_back({o, reg7((int)i, mine[i].box)});
    }
  }
  for (int q = 0; q < F->nranks; ++q) {
    if (q == F->rank) continue;
    for (const CsPiece& p : cs_pieces(FG, F->gboxes, F->gowner, C->gboxes, C->domlo, C->domhi, C->is_per, q, mode, ng, halo, ratio))
      if (C->gowner[p.cbox] == C->rank) s.push_back({q, reg7(C->glocal[p.cbox], p.box)});
  }
  if (side_finish(ctx, P->x.send, s) || side_finish(ctx, P->x.recv, r)) return nullptr;
  P->x.nlocal = (int)(lsrc.size() / 7);
  if (P->x.nlocal) {
    if (hipMalloc(&P->x.d_lsrc, sizeof(int) * lsrc.size()) != hipSuccess || hipMalloc(&P->x.d_ldst, sizeof(int) * ldst.size()) != hipSuccess ||
        hipMemcpy(P->x.d_lsrc, lsrc.data(), sizeof(int) * lsrc.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(P->x.d_ldst, ldst.data(), sizeof(int) * ldst.size(), hipMemcpyHostToDevice) != hipSuccess) {
      pa_fail(ctx, "coarse-source plan: device allocation failed");
      return nullptr;
    }
  }
  if (!S.local.empty()) {
    P->cs = pa_level_create_spec(ctx, S, C->domlo, C->domhi, C->is_per, C->prob_lo, C->prob_hi);
    if (!P->cs) return nullptr;
  }
  CsPlan* raw = P.get();
  F->cs_plans[key] = std::move(P);
  return raw;
}

pa_mf* CsPlan::mf(pa_ctx* ctx, int ncomp, int slot) {
  if (!cs) return nullptr;
  const int key = ncomp * 4 + slot;
  auto it = mfs.find(key);
  if (it != mfs.end()) return it->second;
  pa_mf* m = pa_mf_create(ctx, cs, ncomp, 0, nullptr);
  if (m) mfs[key] = m;
  return m;
}

// ---------------------------------------------------------------------------------- pack / unpack
// Thread t of block row r handles cell t of region r for all components.  Regions are thin (1-2 cells along one
// direction); 32-bit index arithmetic.  One launch covers the region lists of several plans (LevBatch rows = regions).
#define PA_XB 8
struct XRegArgs { DLevelView L; DMFView M; int comp, ncomp; const int* regs; const long long* coff; double* buf; int group = 0, gstride = 0; };
#define PA_YMAX 65535
__device__ __forceinline__ void xregions_row(const XRegArgs& X, const unsigned ry, const int unpack) {
  const DLevelView& L = X.L;
  const DMFView& M = X.M;
  const int comp = X.comp, ncomp = X.ncomp;
  const int* R = X.regs + 7 * ry;
  const int b = R[0];
  const unsigned nx = R[4] - R[1] + 1, ny = R[5] - R[2] + 1, nz = R[6] - R[3] + 1, n = nx * ny * nz;
  const DBox B = L.boxes[b];
  const long long gnx = B.hi[0] - B.lo[0] + 1 + 2 * M.ng, gny = B.hi[1] - B.lo[1] + 1 + 2 * M.ng, gnz = B.hi[2] - B.lo[2] + 1 + 2 * M.ng;
  const long long cs = pa_cstride(gnx * gny * gnz, M.ncomp);
  double* f = M.data + M.off[b] + (long long)comp * cs;
  double* q = X.buf + X.coff[ry] * ncomp;
  const int oi = R[1] - B.lo[0] + M.ng, oj = R[2] - B.lo[1] + M.ng, ok = R[3] - B.lo[2] + M.ng;
  for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const unsigned r = t / nx, i = t - r * nx, k = r / ny, j = r - k * ny;
    const long long idx = ((long long)(k + ok) * gny + (j + oj)) * gnx + (i + oi);
    for (int c = 0; c < ncomp; ++c) {
      const int fc = X.group ? (c / X.group) * X.gstride + c % X.group : c;  // component groups (XJob)
      if (unpack) f[fc * cs + idx] = q[(long long)c * n + t];
      else q[(long long)c * n + t] = f[fc * cs + idx];
    }
  }
}
__global__ __launch_bounds__(256) void k_xregions(LevBatch<XRegArgs, PA_XB> Bt, int unpack, int y0) {
  unsigned ry;
  const XRegArgs& X = Bt.a[Bt.find(blockIdx.y + (unsigned)y0, ry)];
  xregions_row(X, ry, unpack);
}

// same-rank part of a coarse-source refill: region pairs of equal shape, coarse level -> coarse-source level
struct XCopyArgs { DLevelView LS; DMFView MS; int scomp; DLevelView LD; DMFView MD; int dcomp, ncomp; const int* sregs; const int* dregs; int group = 0, sgstride = 0, dgstride = 0; };
__device__ __forceinline__ void xcopy_row(const XCopyArgs& X, const unsigned ry) {
  const int* R = X.sregs + 7 * ry;
  const int* D = X.dregs + 7 * ry;
  const unsigned nx = R[4] - R[1] + 1, ny = R[5] - R[2] + 1, nz = R[6] - R[3] + 1, n = nx * ny * nz;
  const DBox BS = X.LS.boxes[R[0]], BD = X.LD.boxes[D[0]];
  for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const unsigned r = t / nx, i = t - r * nx, k = r / ny, j = r - k * ny;
    for (int c = 0; c < X.ncomp; ++c) {
      const int sc = X.group ? (c / X.group) * X.sgstride + c % X.group : c, dc = X.group ? (c / X.group) * X.dgstride + c % X.group : c;
      X.MD.data[X.MD.off[D[0]] + fab_index(BD, X.MD.ng, X.MD.ncomp, X.dcomp + dc, D[1] + (int)i, D[2] + (int)j, D[3] + (int)k)] =
          X.MS.data[X.MS.off[R[0]] + fab_index(BS, X.MS.ng, X.MS.ncomp, X.scomp + sc, R[1] + (int)i, R[2] + (int)j, R[3] + (int)k)];
    }
  }
}
__global__ __launch_bounds__(256) void k_xcopy(LevBatch<XCopyArgs, PA_XB> Bt, int y0) {
  unsigned ry;
  const XCopyArgs& X = Bt.a[Bt.find(blockIdx.y + (unsigned)y0, ry)];
  xcopy_row(X, ry);
}
// The producer side of an exchange in ONE launch (round 5): the pack of every send list AND the same-rank pieces of the
// coarse-source refills -- both read valid cells of the source multifabs, one writes the send buffers, the other the
// coarse-source multifab, nothing depends on anything.  Rows [0, npack) pack, the rest copy.  On a rank's share of an 8-way shard
// the two launches were 27 + 24 us of a ~1 ms pass, twice per pass.
#define PA_XC 3
struct XProd { LevBatch<XRegArgs, PA_XB> pk; LevBatch<XCopyArgs, PA_XC> cp; };
static_assert(sizeof(XProd) <= 4000, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(256) void k_xproduce(XProd P) {
  const unsigned npack = (unsigned)P.pk.ycum[P.pk.n];
  unsigned ry;
  if (blockIdx.y < npack) {
    const XRegArgs& X = P.pk.a[P.pk.find(blockIdx.y, ry)];
    xregions_row(X, ry, 0);
  } else {
    const XCopyArgs& X = P.cp.a[P.cp.find(blockIdx.y - npack, ry)];
    xcopy_row(X, ry);
  }
}

static int ensure_buf(pa_ctx* ctx, double*& buf, long long& cap, long long need) {
  if (need <= cap) return 0;
  if (buf) {
    PA_HIP(hipStreamSynchronize(ctx->stream));  // an earlier exchange may still read it
    (void)hipFree(buf);
    buf = nullptr; cap = 0;
  }
  PA_HIP(hipMalloc(&buf, sizeof(double) * (size_t)need));
  cap = need;
  return 0;
}

// pack (unpack = 0: the send lists) or unpack (1: the receive lists) of all jobs, PA_XB plans per launch
static void launch_regions(pa_ctx* ctx, int njobs, const XJob* jobs, int unpack) {
  for (int q0 = 0; q0 < njobs; q0 += PA_XB) {
    LevBatch<XRegArgs, PA_XB> Bt;
    long long maxcells = 0;
    for (int q = q0; q < njobs && q < q0 + PA_XB; ++q) {
      const XJob& J = jobs[q];
      const XSide& S = unpack ? J.plan->recv : J.plan->send;
      const int nreg = (int)(S.regs7.size() / 7);
      if (!nreg) continue;
      const pa_mf* M = unpack ? J.dst : J.src;
      Bt.a[Bt.n] = XRegArgs{M->lev->view, M->view, unpack ? J.dcomp : J.scomp, J.ncomp, S.d_regs, S.d_coff, unpack ? J.plan->rbuf : J.plan->sbuf, J.group, unpack ? J.dgstride : J.sgstride};
      Bt.ycum[Bt.n + 1] = Bt.ycum[Bt.n] + nreg;
      ++Bt.n;
      maxcells = std::max(maxcells, S.maxcells);
    }
    if (!Bt.n) continue;
    const unsigned gx = (unsigned)std::min<long long>((maxcells + 255) / 256, 64);
    for (int y0 = 0; y0 < Bt.ycum[Bt.n]; y0 += PA_YMAX)  // gridDim.y is limited to 65535 rows
      hipLaunchKernelGGL(k_xregions, dim3(gx, (unsigned)std::min(PA_YMAX, Bt.ycum[Bt.n] - y0)), dim3(256), 0, ctx->stream, Bt, unpack, y0);
  }
}

int pa_xexchange(pa_ctx* ctx, int njobs, const XJob* jobs) {
  std::vector<pa_xfer> xf;
  for (int q = 0; q < njobs; ++q) {
    const XJob& J = jobs[q];
    XPlan& P = *J.plan;
    // a plan owns ONE pair of packed buffers: two jobs of one call on the same plan would pack over each other
    for (int p = 0; p < q; ++p)
      if (jobs[p].plan == J.plan) return pa_fail(ctx, "ghost exchange: the same plan appears twice in one exchange (one packed buffer per plan)");
    if (J.ncomp < 1 || J.scomp < 0 || !J.src || J.scomp + J.ncomp > J.src->ncomp || J.dcomp < 0 || (J.dst && J.dcomp + J.ncomp > J.dst->ncomp))
      return pa_fail(ctx, "ghost exchange: component range");
    PA_TRY(ensure_buf(ctx, P.sbuf, P.scap, P.send.coff.back() * J.ncomp));
    PA_TRY(ensure_buf(ctx, P.rbuf, P.rcap, P.recv.coff.back() * J.ncomp));
    // one entry per peer of this plan, peers ascending: both sides walk jobs and peers in the same order
    size_t a = 0, b = 0;
    while (a < P.send.peers.size() || b < P.recv.peers.size()) {
      const int ps = a < P.send.peers.size() ? P.send.peers[a] : INT32_MAX, pr = b < P.recv.peers.size() ? P.recv.peers[b] : INT32_MAX;
      const int p = std::min(ps, pr);
      pa_xfer x = {p, nullptr, 0, nullptr, 0};
      if (ps == p) {
        x.sendbuf = P.sbuf + P.send.coff[P.send.first[a]] * J.ncomp;
        x.nsend = (P.send.coff[P.send.first[a + 1]] - P.send.coff[P.send.first[a]]) * J.ncomp;
        ++a;
      }
      if (pr == p) {
        x.recvbuf = P.rbuf + P.recv.coff[P.recv.first[b]] * J.ncomp;
        x.nrecv = (P.recv.coff[P.recv.first[b + 1]] - P.recv.coff[P.recv.first[b]]) * J.ncomp;
        ++b;
      }
      xf.push_back(x);
    }
  }
  bool produced = false;
  {  // pack + same-rank copies in one launch when everything fits one batch
    XProd P;
    long long maxcells = 0;
    int ncopyjobs = 0, npackjobs = 0;
    bool fits = njobs <= PA_XB;
    for (int q = 0; q < njobs && fits; ++q) {
      const XJob& J = jobs[q];
      const XSide& S = J.plan->send;
      const int nreg = (int)(S.regs7.size() / 7);
      if (nreg) {
        P.pk.a[P.pk.n] = XRegArgs{J.src->lev->view, J.src->view, J.scomp, J.ncomp, S.d_regs, S.d_coff, J.plan->sbuf, J.group, J.sgstride};
        P.pk.ycum[P.pk.n + 1] = P.pk.ycum[P.pk.n] + nreg;
        ++P.pk.n;
        ++npackjobs;
        maxcells = std::max(maxcells, S.maxcells);
      }
      if (J.plan->nlocal) {
        if (P.cp.n >= PA_XC) { fits = false; break; }
        P.cp.a[P.cp.n] = XCopyArgs{J.src->lev->view, J.src->view, J.scomp, J.dst->lev->view, J.dst->view, J.dcomp, J.ncomp, J.plan->d_lsrc, J.plan->d_ldst, J.group, J.sgstride, J.dgstride};
        P.cp.ycum[P.cp.n + 1] = P.cp.ycum[P.cp.n] + J.plan->nlocal;
        ++P.cp.n;
        ++ncopyjobs;
        maxcells = std::max(maxcells, J.plan->lmax);
      }
    }
    const long long rows = fits ? (long long)P.pk.ycum[P.pk.n] + P.cp.ycum[P.cp.n] : 0;
    if (fits && npackjobs && ncopyjobs && rows <= PA_YMAX) {
      hipLaunchKernelGGL(k_xproduce, dim3((unsigned)std::min<long long>((maxcells + 255) / 256, 64), (unsigned)rows), dim3(256), 0, ctx->stream, P);
      produced = true;
    }
  }
  if (!produced) launch_regions(ctx, njobs, jobs, 0);
  for (int q0 = 0; q0 < njobs && !produced; q0 += PA_XB) {  // same-rank pieces of coarse-source refills
    LevBatch<XCopyArgs, PA_XB> Bt;
    long long lmax = 0;
    for (int q = q0; q < njobs && q < q0 + PA_XB; ++q) {
      const XJob& J = jobs[q];
      if (!J.plan->nlocal) continue;
      Bt.a[Bt.n] = XCopyArgs{J.src->lev->view, J.src->view, J.scomp, J.dst->lev->view, J.dst->view, J.dcomp, J.ncomp, J.plan->d_lsrc, J.plan->d_ldst, J.group, J.sgstride, J.dgstride};
      Bt.ycum[Bt.n + 1] = Bt.ycum[Bt.n] + J.plan->nlocal;
      ++Bt.n;
      lmax = std::max(lmax, J.plan->lmax);
    }
    if (Bt.n)
      for (int y0 = 0; y0 < Bt.ycum[Bt.n]; y0 += PA_YMAX)
        hipLaunchKernelGGL(k_xcopy, dim3((unsigned)std::min<long long>((lmax + 255) / 256, 64), (unsigned)std::min(PA_YMAX, Bt.ycum[Bt.n] - y0)), dim3(256), 0, ctx->stream, Bt, y0);
  }
  PA_HIP(hipGetLastError());
  if (!xf.empty()) {
    if (!ctx->comm.exchange) return pa_fail(ctx, "sharded level without a transport: call pa_ctx_init_rccl or pa_ctx_set_comm first");
    if (ctx->comm.exchange(ctx->comm.user, (void*)ctx->stream, (int32_t)xf.size(), xf.data()) != 0)
      return pa_fail(ctx, ctx->rccl ? "RCCL point-to-point exchange failed: " + ctx->err : std::string("the transport's exchange failed"));
  }
  launch_regions(ctx, njobs, jobs, 1);
  PA_HIP(hipGetLastError());
  return 0;
}

// The multifab to read coarse data from when filling ghost cells of `fine` from crse[ccomp .. ccomp+ncomp): crse itself
// on one rank, else this rank's freshly refilled coarse-source copy (component 0 = ccomp).  *src = nullptr when this
// rank has nothing to fill (no box, or no coarse-fine face) -- the exchange still runs: other ranks may need our data.
int pa_coarse_source(pa_ctx* ctx, const pa_level* fine, const pa_mf* crse, int ccomp, int ncomp, int mode, int ng, int halo, const pa_mf** src, int* scomp, int ratio) {
  *src = crse;
  *scomp = ccomp;
  if (!crse || crse->lev->nranks <= 1) return 0;
  if (fine->nranks != crse->lev->nranks) return pa_fail(ctx, "coarse and fine level are sharded over different rank sets");
  CsPlan* P = pa_cs_plan(ctx, fine, crse->lev, mode, ng, halo, ratio);
  if (!P) return 1;
  pa_mf* m = P->mf(ctx, ncomp);
  if (P->cs && !m) return 1;
  XJob J = {&P->x, crse, ccomp, m, 0, ncomp};
  PA_TRY(pa_xexchange(ctx, 1, &J));
  *src = m;
  *scomp = 0;
  return 0;
}

// --------------------------------------------------------------------------------------- transports
extern "C" int pa_ctx_set_comm(pa_ctx* ctx, const pa_comm* c) {
  if (!ctx) return 1;
  if (!c) { ctx->comm = pa_comm{nullptr, 0, 1, nullptr, nullptr}; return 0; }
  if (c->nranks < 1 || c->rank < 0 || c->rank >= c->nranks) return pa_fail(ctx, "pa_ctx_set_comm: bad rank / nranks");
  if (c->nranks > 1 && (!c->exchange || !c->allreduce)) return pa_fail(ctx, "pa_ctx_set_comm: exchange and allreduce are required");
  ctx->comm = *c;
  return 0;
}
extern "C" int pa_ctx_nranks(const pa_ctx* ctx) { return ctx ? ctx->comm.nranks : 0; }

extern "C" int pa_allreduce(pa_ctx* ctx, double* vals, int n, int op) {
  if (!ctx || !vals || n < 0 || op < 0 || op > 2) return pa_fail(ctx, "pa_allreduce: bad argument");
  if (ctx->comm.nranks <= 1 || n == 0) return 0;
  if (!ctx->comm.allreduce) return pa_fail(ctx, "pa_allreduce: no transport");
  if (ctx->comm.allreduce(ctx->comm.user, vals, n, op) != 0) return pa_fail(ctx, "the transport's allreduce failed");
  return 0;
}

// RCCL, bound at run time: the library must load (and every single-rank path must work) where librccl is absent,
// and inside a process that already carries a RCCL (PyTorch bundles one under the same soname) we use that copy.
typedef struct ncclComm* pa_ncclComm_t;
struct pa_ncclUniqueId { char internal[128]; };
enum { PA_NCCL_FLOAT64 = 8, PA_NCCL_SUM = 0, PA_NCCL_MAX = 2, PA_NCCL_MIN = 3 };
struct RcclApi {
  void* h = nullptr;
  int (*GetUniqueId)(pa_ncclUniqueId*) = nullptr;
  int (*CommInitRank)(pa_ncclComm_t*, int, pa_ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(pa_ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, pa_ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, pa_ncclComm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, pa_ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi* rccl_api(std::string& why) {
  static RcclApi api;
  static bool tried = false, ok = false;
  static std::string err;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (api.h) break;
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (api.h) break;
      api.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!api.h) {
      err = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?");
    } else {
#define PA_SYM(f, n) *(void**)(&api.f) = dlsym(api.h, n)
      PA_SYM(GetUniqueId, "ncclGetUniqueId"); PA_SYM(CommInitRank, "ncclCommInitRank"); PA_SYM(CommDestroy, "ncclCommDestroy");
      PA_SYM(GroupStart, "ncclGroupStart"); PA_SYM(GroupEnd, "ncclGroupEnd"); PA_SYM(Send, "ncclSend"); PA_SYM(Recv, "ncclRecv");
      PA_SYM(AllReduce, "ncclAllReduce"); PA_SYM(GetErrorString, "ncclGetErrorString");
#undef PA_SYM
      ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Send && api.Recv && api.AllReduce;
      if (!ok) err = "librccl lacks a required symbol";
    }
  }
  why = err;
  return ok ? &api : nullptr;
}

struct RcclState {
  RcclApi* api = nullptr;
  pa_ncclComm_t comm = nullptr;
  pa_ctx* ctx = nullptr;
  double* d_red = nullptr;
};
static int rccl_fail(RcclState* S, const char* what, int rc) {
  S->ctx->err = std::string(what) + ": " + (S->api->GetErrorString ? S->api->GetErrorString(rc) : "error " + std::to_string(rc));
  return 1;
}
static int rccl_exchange(void* user, void* stream, int32_t n, const pa_xfer* x) {
  RcclState* S = (RcclState*)user;
  int rc = S->api->GroupStart();
  if (rc) return rccl_fail(S, "ncclGroupStart", rc);
  for (int i = 0; i < n; ++i) {
    if (x[i].nsend > 0 && (rc = S->api->Send(x[i].sendbuf, (size_t)x[i].nsend, PA_NCCL_FLOAT64, x[i].peer, S->comm, (hipStream_t)stream))) break;
    if (x[i].nrecv > 0 && (rc = S->api->Recv(x[i].recvbuf, (size_t)x[i].nrecv, PA_NCCL_FLOAT64, x[i].peer, S->comm, (hipStream_t)stream))) break;
  }
  const int rc2 = S->api->GroupEnd();
  if (rc) return rccl_fail(S, "ncclSend/ncclRecv", rc);
  if (rc2) return rccl_fail(S, "ncclGroupEnd", rc2);
  return 0;
}
static int rccl_allreduce(void* user, double* vals, int32_t n, int32_t op) {
  RcclState* S = (RcclState*)user;
  pa_ctx* ctx = S->ctx;
  if (n > 64) return 1;
  PA_HIP(hipMemcpyAsync(S->d_red, vals, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  const int rc = S->api->AllReduce(S->d_red, S->d_red, (size_t)n, PA_NCCL_FLOAT64, op == 0 ? PA_NCCL_MIN : (op == 1 ? PA_NCCL_MAX : PA_NCCL_SUM), S->comm, ctx->stream);
  if (rc) return rccl_fail(S, "ncclAllReduce", rc);
  PA_HIP(hipMemcpyAsync(vals, S->d_red, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  PA_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int pa_rccl_unique_id(pa_ctx* ctx, void* id128) {
  PaBind bind_(ctx);
  if (!ctx || !id128) return pa_fail(ctx, "pa_rccl_unique_id: null argument");
  std::string why;
  RcclApi* api = rccl_api(why);
  if (!api) return pa_fail(ctx, "pa_rccl_unique_id: " + why);
  pa_ncclUniqueId id;
  const int rc = api->GetUniqueId(&id);
  if (rc) return pa_fail(ctx, std::string("ncclGetUniqueId: ") + (api->GetErrorString ? api->GetErrorString(rc) : "error"));
  memcpy(id128, &id, 128);
  return 0;
}

void pa_rccl_destroy(pa_ctx* ctx) {
  if (!ctx || !ctx->rccl) return;
  if (ctx->rccl->comm) (void)ctx->rccl->api->CommDestroy(ctx->rccl->comm);
  if (ctx->rccl->d_red) (void)hipFree(ctx->rccl->d_red);
  delete ctx->rccl;
  ctx->rccl = nullptr;
}

extern "C" int pa_ctx_init_rccl(pa_ctx* ctx, int nranks, int rank, const void* id128) {
  PaBind bind_(ctx);
  if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return pa_fail(ctx, "pa_ctx_init_rccl: bad argument");
  std::string why;
  RcclApi* api = rccl_api(why);
  if (!api) return pa_fail(ctx, "pa_ctx_init_rccl: " + why);
  pa_rccl_destroy(ctx);
  RcclState* S = new RcclState();
  S->api = api; S->ctx = ctx;
  pa_ncclUniqueId id;
  memcpy(&id, id128, 128);
  const int rc = api->CommInitRank(&S->comm, nranks, id, rank);
  if (rc) {
    pa_fail(ctx, std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(rc) : "error"));
    delete S;
    return 1;
  }
  if (hipMalloc(&S->d_red, 64 * sizeof(double)) != hipSuccess) { (void)api->CommDestroy(S->comm); delete S; return pa_fail(ctx, "pa_ctx_init_rccl: device allocation failed"); }
  ctx->rccl = S;
  ctx->comm = pa_comm{S, rank, nranks, rccl_exchange, rccl_allreduce};
  return 0;
}

// ---- a transport that moves NOTHING and costs what a link would: for projections of an N-rank run on one GPU (bench.py
// --sim-of N --xdelay ...).  Every exchange enqueues, on the stream it is issued on, a kernel that spins for
//   fixed_us + (largest number of bytes this rank sends to or receives from ONE peer) / link_GBs
// -- xGMI is point to point, every peer has its own link, so a grouped exchange is bound by the busiest link --
// so the schedule's overlaps hide or expose that time exactly as they would a real exchange.  Results are wrong in ghost cells.
struct DelayState { double fixed_us, link_GBs; double total_us = 0; long long calls = 0; };
__global__ void k_spin_us(long long ticks) {  // 100 MHz constant clock; bounded: exits after `ticks` (host caps them at 20 ms)
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static int delay_exchange(void* user, void* stream, int32_t n, const pa_xfer* x) {
  DelayState* S = (DelayState*)user;
  std::map<int, long long> bs, br;
  for (int i = 0; i < n; ++i) { bs[x[i].peer] += 8 * x[i].nsend; br[x[i].peer] += 8 * x[i].nrecv; }
  long long mx = 0;
  for (auto& kv : bs) mx = std::max(mx, kv.second);
  for (auto& kv : br) mx = std::max(mx, kv.second);
  double us = S->fixed_us + (S->link_GBs > 0 ? (double)mx / (S->link_GBs * 1e3) : 0.0);
  us = std::min(us, 20000.0);
  S->total_us += us; ++S->calls;
  if (us > 0) hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long)(us * 100.0));
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
static int delay_allreduce(void*, double*, int32_t, int32_t) { return 0; }
extern "C" int pa_ctx_set_delay_comm(pa_ctx* ctx, int nranks, int rank, double fixed_us, double link_GBs) {
  if (!ctx || nranks < 1 || rank < 0 || rank >= nranks || fixed_us < 0) return pa_fail(ctx, "pa_ctx_set_delay_comm: bad argument");
  DelayState* S = new DelayState{fixed_us, link_GBs};  // lives as long as the process (a diagnostic transport)
  ctx->comm = pa_comm{S, rank, nranks, delay_exchange, delay_allreduce};
  return 0;
}
// (calls, modelled microseconds) of the delay transport since it was set
extern "C" int pa_delay_comm_stats(const pa_ctx* ctx, int64_t* calls, double* total_us) {
  if (!ctx || ctx->comm.exchange != delay_exchange || !calls || !total_us) return 1;
  const DelayState* S = (const DelayState*)ctx->comm.user;
  *calls = S->calls; *total_us = S->total_us;
  return 0;
}

extern "C" int pa_level_global_ids(const pa_level* L, int32_t* gids) {
  if (!L || !gids) return 1;
  for (size_t b = 0; b < L->boxes.size(); ++b) gids[b] = L->gid.empty() ? (int32_t)b : L->gid[b];
  return 0;
}

// Transport check (what bench.py and the tools run once before trusting a transport): every rank sends n doubles of a
// rank-specific pattern to rank+1 and receives from rank-1 (itself on one rank) through the context's transport, and a
// max-reduction over the ranks is compared with its known answer.  Synchronous.  0 = the transport works.
__global__ void k_selftest_fill(double* p, long long n, int rank) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 1000.0 * rank + (double)(i % 977);
}
extern "C" int pa_comm_selftest(pa_ctx* ctx, int64_t n) {
  PaBind bind_(ctx);
  if (!ctx || n < 1) return pa_fail(ctx, "pa_comm_selftest: bad argument");
  if (!ctx->comm.exchange || !ctx->comm.allreduce) return pa_fail(ctx, "pa_comm_selftest: the context has no transport");
  const int N = ctx->comm.nranks, r = ctx->comm.rank, to = (r + 1) % N, from = (r + N - 1) % N;
  double *s = nullptr, *d = nullptr;
  PA_HIP(hipMalloc(&s, sizeof(double) * n));
  PA_HIP(hipMalloc(&d, sizeof(double) * n));
  hipLaunchKernelGGL(k_selftest_fill, dim3(64), dim3(256), 0, ctx->stream, s, (long long)n, r);
  PA_HIP(hipMemsetAsync(d, 0, sizeof(double) * n, ctx->stream));
  int rc = 0;
  if (to == from) {
    pa_xfer x = {to, s, n, d, n};
    rc = ctx->comm.exchange(ctx->comm.user, (void*)ctx->stream, 1, &x);
  } else {
    pa_xfer x[2] = {{std::min(to, from), nullptr, 0, nullptr, 0}, {std::max(to, from), nullptr, 0, nullptr, 0}};
    for (auto& e : x) {
      if (e.peer == to) { e.sendbuf = s; e.nsend = n; }
      if (e.peer == from) { e.recvbuf = d; e.nrecv = n; }
    }
    rc = ctx->comm.exchange(ctx->comm.user, (void*)ctx->stream, 2, x);
  }
  std::vector<double> h((size_t)n);
  if (!rc && (hipMemcpyAsync(h.data(), d, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) rc = 2;
  (void)hipFree(s);
  (void)hipFree(d);
  if (rc) return pa_fail(ctx, "pa_comm_selftest: exchange failed" + (ctx->err.empty() ? std::string() : " (" + ctx->err + ")"));
  for (int64_t i = 0; i < n; ++i)
    if (h[(size_t)i] != 1000.0 * from + (double)(i % 977)) return pa_fail(ctx, "pa_comm_selftest: received data differ from what rank " + std::to_string(from) + " sent");
  double v[2] = {(double)r, -(double)r};
  if (ctx->comm.allreduce(ctx->comm.user, v, 2, 1) != 0) return pa_fail(ctx, "pa_comm_selftest: allreduce failed");
  if (v[0] != (double)(N - 1) || v[1] != 0.0) return pa_fail(ctx, "pa_comm_selftest: allreduce(max) gave a wrong answer");
  return 0;
}
