// pa_internal.h -- shared host/device definitions of libpeleanalysis_amd (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <array>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include "../../include/peleanalysis_amd.h"

struct DBox { int lo[3]; int hi[3]; };

// ---- pa_options: EVERY environment switch of the library (round 6: 76 ad-hoc getenv sites before).  Read once -- when the first
// context is created -- and again only when the caller asks (pa_options_reload: tests and bench.py --ab flip a switch between two
// passes of one process).  Each one selects between code paths that both exist for a reason (a class of inputs needs the other
// path, or a result is only verifiable one way); the A/B switches of measured-and-lost experiments are gone with their losing sides
// (DESIGN_HISTORY.md lists them).  tests/test_options.py flips every one.  The tools' own switches: tools/common/pa_parmparse.h.
struct pa_options {
  int filter_exact = 0;               // PA_FILTER_EXACT=1: Filter::apply_filter in the reference's tap order (bit-exact) instead of the separable form (<= 1e-12 Linf)
  int allow_unverified_gaussian = 0;  // PA_ALLOW_UNVERIFIED_GAUSSIAN=1: filter_type 2 with the textbook weights (unverified against PelePhysics)
  int retile_max[3] = {0, 0, 0};      // PA_RETILE_MAX="x y z": limits of the internal tiling instead of pa_hierarchy_retile_limits' choice
  int fused2 = 1;                     // PA_FUSED2=0: the first fused pipeline (two-layer fix-up from a resolved shell of c) wherever it is legal
  int ncg = 1;                        // PA_NCG=0: the sweep does not mirror the first layer behind special x faces for the fix-up
  int dist_early = 0;                 // PA_DIST_EARLY=1: sharded pass, early tiles of the sweep on the side stream under exchange A
  int smooth_replicated = 0;          // PA_SMOOTH_REPLICATED=1: do_smooth on a sharded hierarchy as N replicated solves (the one-rank bits)
  int smooth_mg = -1;                 // PA_SMOOTH_MG=1 / 0: the multigrid preconditioner always / never (default: where dt / dx^2 > 8)
  int smooth_march = 1;               // PA_SMOOTH_MARCH=0: the cell-per-thread stencil kernels of the solve
  int smooth_timing = 0;              // PA_SMOOTH_TIMING=1: setup / iteration times of a solve on stderr
  int scratch_poison = 0;             // PA_SCRATCH_POISON=1 (tests): a level's work multifabs (pa_level_scratch: kept between calls, contents undefined) are
                                      // filled with NaN at every acquisition -- a kernel that reads a cell no step of THIS call wrote shows up in the result
  int force_fallbacks = 0;            // PA_FORCE_FALLBACKS=1 (tests): every path that exists for inputs the tuned one does not take -- FillBoundary /
                                      // patch gather per ghost cell (regions that do not fit a plan), the sweeps group by group (more groups than a
                                      // launch holds), the first form of the marching-cubes cell pass (FABs wider than 819 cells) and its
                                      // level-by-level loop, FillPatchTwoLevels per ghost cell (ratio != 2), ghost fills level by level -- on ANY input
};
const pa_options& pa_opt();

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
  int gshift;        // log2(g) when g is a power of two (owner_of shifts instead of dividing), else -1
  int nsf;           // number of "special" box faces: faces with at least one ghost cell that is not a valid cell
  const int* sfaces; // nsf entries box*6 + dir*2 + side; the boundary kernels launch over these only
  const int* sfindex;           // [nboxes*6] entry of a box face in sfaces, or -1 (ordinary face: every ghost cell is a valid cell)
  const long long* sfoff;       // [nsf] start of the face's ghost-cell codes in sfcode
  const unsigned short* sfcode; // cf_masks of every ghost cell of every special face, (t0 fastest, t1) per face
  // Resolved ghost values of the progress variable behind every special face, compact and FACE-MAJOR (pa_fused2.hip):
  // face e holds (n0+2) x (n1+2) doubles at cg + cgoff[e], element (a0, a1) = a1 * (n0+2) + a0 with a0 / a1 = tangential
  // coordinate - box lo + 1 (a ring of one cell for the edge ghosts), + 2 rows of slack.  Null: not allocated.
  const long long* cgoff;
  double* cg;
  // Coarse patches (pa_fused.hip, k_cpatch): for every special coarse-fine face the coarse values its boundary
  // interpolation can touch -- the coarse plane behind the face, tangentially coarsen(lo - 1) - 2 .. coarsen(hi + 1) + 2 --
  // as one dense 2-D array at cp + cpoff[e] (cpoff < 0: a wall face).  Filled by a gather pass so that the face kernels
  // read coarse data without owner-map lookups.  Null: not allocated.
  const long long* cpoff;
  double* cp;
};

// One work item of the chunked special-face kernels (round 6, pa_fused.hip: k_prep_faces_chunks / k_faces_fix_chunks): a rectangle
// (u0 .. u0 + cw - 1) x (v0 .. v0 + ch - 1) of the ghost cells of ONE special face in the face's tangential coordinates (t0 fastest),
// cw * ch = 1024, one thread per 2 x 2 block; blocks start on even GLOBAL indices (the four ghost cells of a block share one coarse
// parent), so u0 / v0 are -1 where the face starts on an odd index: cells outside the face are predicated off.  Everything the kernels need to know about the face sits in the record -- one wide
// scalar load instead of the chain work table -> sfaces -> boxes -> sfoff / cgoff / cpoff -- and `flags` says whether the whole
// chunk is of ONE kind, so that the uniform kinds run straight-line code without reading a per-cell code first.
enum { PA_SFC_FULL = 1,   // every cell is coarse-fine with the full centred stencil (code PA_CODE_FULL), the face starts on even tangential
                          // indices and has even extents: a thread's 2 x 2 block shares ONE coarse parent
       PA_SFC_WALL = 2,   // every cell lies outside a wall
       PA_SFC_HAS_CF = 4, // some cell is coarse-fine
       PA_SFC_HAS_VALID = 8 };  // some cell is a valid cell of the level (a face that is partly covered by a neighbouring box)
struct alignas(16) SfChunk {
  int face;      // entry of the face in DLevelView::sfaces
  int box;
  int dir_side;  // dir * 2 + side
  int flags;
  int lo[3], hi[3];
  int u0, v0, cw, ch;
  long long sfoff, cgoff, cpoff;
};
// cf_masks of a coarse-fine ghost cell whose 3 x 3 coarse neighbourhood in the ghost plane is coarse-fine throughout: class 1,
// tangential stencils -1 .. 1 in both directions, cross term on
#define PA_CODE_FULL 0x555u

struct DMFView {
  double* data;
  const long long* off;  // per box, in doubles
  int ncomp, ng;
  // optional affine view used when this multifab is read as COARSE data: value = (v - xa) * xb.
  // Lets applyBC on the progress variable interpolate from the coarse phi without a stored coarse c.
  int xform;
  double xa, xb;
};

// Several levels in ONE launch (the boundary kernels of small or sharded levels are launch / latency bound: a launch per
// level and kernel costs more than the work): blockIdx.y runs over the concatenated rows (boxes / special faces) of up to
// PA_MAXB levels; level l owns rows ycum[l] .. ycum[l+1]-1.  Passed by value (kernel arguments, < 4 KB).
#define PA_MAXB 4
#define PA_MAXSLOTS 16  // component slots per batch of the boundary kernels (pa_fused.hip: SlotK)
template <typename A, int CAP = PA_MAXB>
struct LevBatch {
  int n = 0;
  int ycum[CAP + 1] = {};
  A a[CAP];
  __device__ __forceinline__ int find(unsigned y, unsigned& local) const {
    int l = 0;
    while (l + 1 < n && y >= (unsigned)ycum[l + 1]) ++l;
    local = y - (unsigned)ycum[l];
    return l;
  }
};

struct pa_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  double* d_red = nullptr;  // reduction scratch
  size_t red_cap = 0;
  int* d_flags = nullptr;   // [0] = coarse-fine ghost cells whose coarse data was missing
  void* d_slow = nullptr;   // cells the clip-aware curvature fix-up hands to its general path (pa_fused.hip: SlowList)
  hipEvent_t fix_evs[2] = {nullptr, nullptr};  // fix-up: perimeter kernel on the side stream (pa_fused.hip)
  double* d_prog = nullptr; // (pmin, 1 / (pmax - pmin)) of the component slots of a batch (pa_gradcurv_run_comps2)
  void* d_scr = nullptr;    // grow-only scratch (marching cubes)
  size_t scr_cap = 0;
  void* h_pin = nullptr;    // grow-only pinned host buffer (count read-backs of marching cubes)
  size_t h_pin_cap = 0;
  void* d_mcz = nullptr;    // per-cell code bytes of the level-batched marching cubes: all zeros between calls (pa_mc.hip)
  size_t mcz_cap = 0;
  bool mcz_dirty = true;
  // surface blocks handed out by pa_mc_level* / taken back by pa_device_free: a freed block is kept (up to 4 of them) for
  // the next level's surface instead of a hipFree + hipMalloc pair per level (~0.1 ms, as long as the whole GPU pass)
  std::map<void*, size_t> surf_live;
  std::vector<std::pair<void*, size_t>> surf_cache;
  // optional per-launch timing of tagged kernels with HIP events on ctx->stream (bench.py roofline)
  unsigned profile = 0;  // bit t set: launches under tag t are timed (pa_profile_enable)
  struct Ev { hipEvent_t a, b; int tag; };
  std::vector<Ev> evs;
  // second stream + ordering events of the fused pipeline (boundary kernels of one level next to the sweep of
  // another; pa_pipeline.hip); created on first use
  hipStream_t stream2 = nullptr;
  std::vector<hipStream_t> lev_streams;  // one per level: level-concurrent boundary kernels (pa_pipeline.hip)
  std::vector<hipEvent_t> sync_evs;
  // transport between the ranks that share a sharded hierarchy (pa_dist.hip): caller-supplied (pa_ctx_set_comm)
  // or the built-in RCCL one (pa_ctx_init_rccl)
  std::string sweep_kernel;  // variant of the fused sweep launched last (pa_sweep_kernel_name)
  int smooth_iters = -1;     // the last do_smooth solve of pa_curvature_run (pa_smooth_last): iterations, relative residual
  double smooth_res = 0.0;
  int curv_path = -1;        // implementation of the last pa_curvature_run (pa_curvature_last_path)
  pa_comm comm = {nullptr, 0, 1, nullptr, nullptr};
  struct RcclState* rccl = nullptr;
};

// RAII: records an event pair around a tagged kernel launch when profiling is enabled
struct ProfScope {
  pa_ctx* ctx;
  pa_ctx::Ev e;
  bool on;
  ProfScope(pa_ctx* c, int tag) : ctx(c), on(((c->profile >> tag) & 1u) != 0) {
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
       PA_TAG_FILTER = 7, PA_TAG_MC = 8, PA_TAG_XCHG = 9 };

// Workgroup table of a sweep over boxes of DIFFERENT sizes (pa_fused.hip: sweep_wgtab): entry i = {box, tile of that box} or
// {-1, 0}; groups of 8 boxes with similar tile counts are interleaved so that entry i and i + 8 (same XCD) belong to one box
struct WgTab {
  int* d = nullptr;
  unsigned n = 0;
  std::vector<int> h;  // host copy of the table
  // GOUT == 2 sweeps that store G only where something reads it (pa_sweep_gneed): one byte per table entry, by the serial of the finer level (0: none)
  mutable std::map<long long, unsigned char*> gneed;
  ~WgTab() {
    if (d) (void)hipFree(d);
    for (auto& kv : gneed)
      if (kv.second) (void)hipFree(kv.second);
  }
};

struct pa_level {
  pa_ctx* ctx = nullptr;
  std::vector<DBox> boxes;
  int domlo[3], domhi[3], is_per[3];
  double prob_lo[3], prob_hi[3], dx[3], dxinv[3];
  int g = 1, mlo[3], mn[3];
  std::vector<int> owner;  // host copy
  DBox* d_boxes = nullptr;
  int* d_owner = nullptr;
  std::vector<int> sfaces;  // special faces (see DLevelView)
  int* d_sfaces = nullptr;
  int* d_sfindex = nullptr;
  long long* d_sfoff = nullptr;
  unsigned short* d_sfcode = nullptr;
  int maxn[3] = {0, 0, 0};  // max box extent per dim
  long long ncells = 0;
  bool fusable = true;      // no concave coarse-fine corner (see pa_level_create)
  bool pure_faces = true;   // every special face of the WHOLE BoxArray has no ghost cell that is a valid cell (pa_fused2.hip)
  long long* d_cpoff = nullptr;
  double* d_cp = nullptr;   // allocated on first use (level_cp)
  int cp_sets = 0, cg_sets = 0;  // component slots the buffers hold (pa_fused.hip: SlotK)
  long long cp_total = 0;
  long long* d_cgoff = nullptr;
  double* d_cg = nullptr;   // allocated on first use (pa_level_cg)
  long long cg_total = 0;
  double* d_ncg = nullptr;  // NCG (pa_fused_march.h MarchArgs::ncg): 5 arrays of cg_stride doubles, allocated on first use (level_ncg)
  bool ncg_live = false;    // this pass's sweep wrote them (set by pa_gradcurv_levels_cg, consumed by pa_gradcurv_fix_levels)
  int ncg_minw = 0;         // ... for the boxes at least this wide
  std::vector<long long> cgoff;
  // Irregular cells (pa_fused.hip: k_find_irregular / k_curv_general): boundary cells of local boxes next to a concave
  // coarse-fine corner or to the line where a box face changes from covered to coarse-fine, whose curvature neither the
  // sweep nor the face fix-up gets right; recomputed one by one through a geometry-independent path.  Built on first use.
  // Sweep groups (pa_fused.hip): boxes wider than 32 cells take the wide sweep kernel (64-column tiles), the others the narrow
  // one (two 32-column rows per wavefront); a level that has both kinds keeps two index lists (wide first) for the two launches
  int* d_blist = nullptr;
  int nwide = 0, nnarrow = 0;
  int wmax[3] = {0, 0, 0}, nmax[3] = {0, 0, 0};  // largest extents among the wide / the narrow boxes
  int* d_sfboxes = nullptr; // local boxes with at least one special face (k_prep_ring runs over these only)
  int nsfboxes = 0;
  void* d_sfwg = nullptr;   // int2 {special face, chunk of 256 of its ghost cells}: the work table of the per-face-cell kernels
  int nsfwg = 0;
  void* d_pfwg = nullptr;   // int2 {special face, chunk of 256 of its PERIMETER cells}: the work table of k_faces_curv_tab (round 5)
  int npfwg = 0;
  SfChunk* d_sfchunk = nullptr;  // the chunk records of the level's special faces (round 6)
  int nsfchunk = 0;
  void* d_ring = nullptr;   // RingItem (pa_fused.hip): the edge ghost cells whose resolved progress variable goes into a face's ring, built on first use
  int nring = -1;           // -1: not built yet
  std::vector<unsigned char> xneed;  // per box: bits 1 / 2 as pa_sweep_gneed (built on first use)
  void* d_irr = nullptr;    // int4 {box, i, j, k}
  int nirr = -1;            // -1: not built yet
  int nremote = 0;          // boxes of this level owned by other ranks (pa_level_create_sharded)
  DLevelView view;
  // ---- sharding (pa_level_create_sharded): the level's whole BoxArray and its DistributionMapping; `boxes` are the
  // ones this rank owns, in global order.  Owner-map entries: >= 0 local box, -1 none, -2 - g = global box g of another rank.
  int rank = 0, nranks = 1;
  long long serial = 0;            // unique per level object (keys of the plan caches)
  std::vector<DBox> gboxes;        // global BoxArray (empty for an unsharded level)
  std::vector<int> gowner;         // owner rank of every global box
  std::vector<int> gid;            // global index of local box b
  std::vector<int> glocal;         // local index of global box g, or -1
  bool source_only = false;        // coarse-source level (pa_dist.hip): no special faces, never swept
  mutable std::map<int, std::unique_ptr<struct XPlan>> fb_plans;                       // FillBoundary plans by ghost width
  mutable std::map<long long, std::unique_ptr<struct CpPlan>> cp_plans;               // coarse-patch gather as copy regions, by coarse level serial (pa_dist.hip)
  mutable std::map<int, std::unique_ptr<struct FbLocal>> fb_local;                     // local FillBoundary as copy regions, by ghost width (pa_dist.hip)
  mutable std::map<std::pair<long long, int>, std::unique_ptr<struct CsPlan>> cs_plans; // coarse-source plans by (coarse level serial, mode)
  mutable std::unique_ptr<struct RepPlan> rep_plan;                                     // the level replicated on every rank (pa_dist.hip)
  mutable std::map<long long, std::unique_ptr<WgTab>> wgtabs;                           // sweep workgroup tables by (group, tile shape) (pa_fused.hip)
  mutable std::map<long long, std::unique_ptr<struct RsPlan>> rs_plans;                 // restriction onto a sharded coarse level, by coarse level serial (pa_dist.hip)
  mutable std::map<std::pair<long long, int>, std::unique_ptr<struct FpPlan>> fp_plans; // FillPatchTwoLevels parent lists by (coarse level serial, ghost width) (pa_filter.hip)
  mutable std::map<std::array<int, 3>, struct pa_mf*> scratch;                          // work multifabs by (components, ghost width, role), kept for the level's lifetime (pa_level_scratch)
  ~pa_level();
};

// coarse parents of a fine level's ghost shell that have coarse-fine children (pa_filter.hip: k_fp_find / k_fp_do)
struct FpPlan {
  int n = 0;
  void* d_items = nullptr;  // int4 {box | children mask << 24, parent cell}
  ~FpPlan();
};

struct LevelSpec {
  std::vector<DBox> local;   // boxes this rank owns
  std::vector<int> gid;      // their global indices (sharded levels)
  std::vector<DBox> gboxes;  // the whole BoxArray (sharded levels; empty otherwise)
  std::vector<int> gowner;
  int rank = 0, nranks = 1;
  bool source_only = false;
};
pa_level* pa_level_create_spec(pa_ctx* ctx, const LevelSpec& S, const int32_t domlo[3], const int32_t domhi[3], const int32_t is_per[3],
                               const double prob_lo[3], const double prob_hi[3]);
bool pa_face_is_special(const pa_level* L, const DBox& B, int d, int side);
// a work multifab of the level that lives as long as the level does (contents undefined between calls): a 10-GB hipMalloc +
// hipFree per call of pa_curvature_run cost more than the kernels it served
// role: two work multifabs of the same shape that are alive at the same time take different roles
struct pa_mf* pa_level_scratch(pa_ctx* ctx, const pa_level* L, int ncomp, int ng, int role = 0);
const WgTab* pa_sweep_wgtab(const pa_level* L, int cls, int tw, int mty, int kseg, bool force, int part = 0);
// Where a GOUT == 2 sweep (pa_fused_march3.h) has to store G, one byte per entry of table T (tiles of tw x mty x kseg cells): bit 0 =
// every cell of the tile (a coarse patch of the finer level reads some of them: cpregs = the host copy of that level's CpPlan regions,
// null when there is no finer level), bits 1 / 2 = the three columns behind the low / high x face of the tile's box.  Null: no memory.
const unsigned char* pa_sweep_gneed(const pa_level* L, const WgTab* T, int tw, int mty, int kseg, long long fine_serial, const std::vector<int>* cpregs);
int pa_host_classify(const pa_level* L, int i, int j, int k);

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

// The current HIP device is per host thread, and a pa_ctx may be used from a thread other than the one that created it
// (tools bring the context up on a worker thread) or next to contexts of other devices (ngpus > 1 in one process):
// every entry point that allocates or launches binds the context's device first.
struct PaBind {
  explicit PaBind(const pa_ctx* c) { if (c) (void)hipSetDevice(c->device); }
};

// ---------------------------------------------------------------- device helpers
// Component stride of a FAB inside a pa_mf (in doubles): the cell count rounded up to 512 B and placed at
// 2 KiB past a multiple of 16 KiB (boxes of >= 32^3 cells; smaller ones are only kept off multiples of 16 KiB).
// With the plain AMReX stride (nx*ny*nz) a 128^3 box puts all components of one cell 16 MiB apart = on the same
// HBM channel, which costs ~20 % of the write bandwidth of an 8-output kernel (tools/bench/membench2.hip).  The
// address interleave repeats every 16 KiB (tools/bench/membench5.hip "stride", profiles/r03_membench5_stride.txt:
// 8 output streams + 1 input stream, 1-D streaming, stride = 16 MiB + pad: pad 0 2.49 ms, 4 KiB 2.11, 512 B 1.81,
// 1 KiB 1.69, 2 KiB 1.65 = the rate of streams 1 GiB apart): 2 KiB spreads the 8 results of the fused sweep evenly
// over the period.  amrex::Array4 carries an explicit nstride too, so this stays within the reference's data model.
__host__ __device__ __forceinline__ long long pa_cstride(long long ncells, int ncomp) {
  long long cs = (ncells + 63) / 64 * 64;
  if (ncomp > 1) {
    if (ncells >= 32768) cs += (256 + 2048 - cs % 2048) % 2048;
    else if ((cs % 2048) == 0) cs += 64;
  }
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
    m[d] = L.gshift >= 0 ? (r >> L.gshift) : (r / L.g);
    if (m[d] >= L.mn[d]) return -1;
  }
  return L.owner[((long long)m[2] * L.mn[1] + m[1]) * L.mn[0] + m[0]];
}

// 0 covered (valid cell, maybe through a periodic image; box id + wrapped cell
// returned), 1 not covered (inside domain) = coarse-fine, 2 outside a wall
__device__ __forceinline__ int classify(const DLevelView& L, int i, int j, int k, int& box, int p[3]) {
  p[0] = i; p[1] = j; p[2] = k;
  if (!wrap_cell(L, p)) return 2;
  box = owner_of(L, p);  // >= 0: box of this rank; <= -2: valid cell of a box owned by another rank; -1: none
  return (box != -1) ? 0 : 1;
}
__device__ __forceinline__ int classify(const DLevelView& L, int i, int j, int k) {
  int b, p[3];
  return classify(L, i, j, k, b, p);
}

__host__ __device__ __forceinline__ int coarsen_idx(int i, int r) { return r == 2 ? (i >> 1) : ((i < 0) ? -((-i + r - 1) / r) : i / r); }  // floor

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

// raw coarse value with periodic wrap; ok cleared if the cell has no owner
__device__ __forceinline__ double crse_raw(const DLevelView& LC, const DMFView& MC, int comp, int ic, int jc, int kc, bool& ok) {
  int p[3] = {ic, jc, kc};
  if (!wrap_cell(LC, p)) { ok = false; return 0.0; }
  const int b = owner_of(LC, p);
  if (b < 0) { ok = false; return 0.0; }
  return MC.data[MC.off[b] + fab_index(LC.boxes[b], MC.ng, MC.ncomp, comp, p[0], p[1], p[2])];
}
__device__ __forceinline__ double crse_val(const DLevelView& LC, const DMFView& MC, int comp, int ic, int jc,
                                           int kc, bool& ok) {
  const double v = crse_raw(LC, MC, comp, ic, jc, kc, ok);
  return MC.xform ? (v - MC.xa) * MC.xb : v;
}

// amrex::poly_interp_coeff at compile time (same IEEE operations, correctly rounded) for the
// stencils of refinement ratio 2: tangential points lo..hi (lo in {-2,-1,0}, hi in {0,1,2}) seen
// from xInt = -0.25 (even fine cell) or +0.25 (odd), and the normal-direction points
// {-1, 0.5, 1.5, 2.5} seen from -0.5.
struct CfCoefTab {
  double tan[2][3][3][3];  // [odd][lo+2][hi][m]
  double nrm[5][4];        // [NX][m]
};
constexpr CfCoefTab make_cf_coef_tab() {
  CfCoefTab T{};
  for (int odd = 0; odd < 2; ++odd)
    for (int lo = -2; lo <= 0; ++lo)
      for (int hi = 0; hi <= 2; ++hi) {
        const int N = hi - lo + 1;
        if (N > 3) continue;
        const double xInt = -0.5 + ((double)odd + 0.5) / 2.0;
        double x[3] = {0.0, 0.0, 0.0};
        for (int m = 0; m < N; ++m) x[m] = (double)(lo + m);
        for (int jj = 0; jj < N; ++jj) {
          double num = 1.0, den = 1.0;
          for (int ii = 0; ii < N; ++ii) {
            if (ii == jj) continue;
            num *= xInt - x[ii];
            den *= x[jj] - x[ii];
          }
          T.tan[odd][lo + 2][hi][jj] = num / den;
        }
      }
  for (int NX = 1; NX <= 4; ++NX) {
    const double x[4] = {-1.0, 0.5, 1.5, 2.5};
    for (int jj = 0; jj < NX; ++jj) {
      double num = 1.0, den = 1.0;
      for (int ii = 0; ii < NX; ++ii) {
        if (ii == jj) continue;
        num *= -0.5 - x[ii];
        den *= x[jj] - x[ii];
      }
      T.nrm[NX][jj] = num / den;
    }
  }
  return T;
}
static __device__ __constant__ const CfCoefTab g_cf_coef = make_cf_coef_tab();
// the same table for uses with compile-time indices: literals in the instruction stream instead of loads (the straight-line paths of
// the chunked face kernels); same constexpr evaluation, so the same bits as the table above
static constexpr CfCoefTab k_cf_coef = make_cf_coef_tab();

// Masks of InterpBndryData for ghost cell q of a face normal to `dir` (ratio r), packed:
//   bits 0-1 class of q (0 valid cell, 1 coarse-fine, 2 outside a wall); for class 1 also
//   bits 2-3 lo0+2, 4-5 hi0, 6-7 lo1+2, 8-9 hi1 (tangential stencil extents in coarse cells),
//   bit 10: all four diagonal neighbours are coarse-fine cells too (cross term on).
// Depends on the fine level only; stored per special-face ghost cell at level creation (sfcode).
__device__ inline unsigned cf_masks(const DLevelView& LF, const int q[3], int dir, int r) {
  const unsigned cls = (unsigned)classify(LF, q[0], q[1], q[2]);
  if (cls != 1u) return cls;
  const int t0 = dir == 0 ? 1 : 0, t1 = dir == 2 ? 1 : 2;
  const int tdir[2] = {t0, t1};
  unsigned code = cls;
  for (int t = 0; t < 2; ++t) {
    const int td = tdir[t];
    int m1[3] = {q[0], q[1], q[2]}, p1[3] = {q[0], q[1], q[2]}, m2[3] = {q[0], q[1], q[2]}, p2[3] = {q[0], q[1], q[2]};
    m1[td] -= r; p1[td] += r; m2[td] -= 2 * r; p2[td] += 2 * r;
    const bool okm1 = classify(LF, m1[0], m1[1], m1[2]) == 1;
    const bool okp1 = classify(LF, p1[0], p1[1], p1[2]) == 1;
    int lo = okm1 ? -1 : 0, hi = okp1 ? 1 : 0;
    if (lo == -1 && hi == 0 && classify(LF, m2[0], m2[1], m2[2]) == 1) lo = -2;
    else if (hi == 1 && lo == 0 && classify(LF, p2[0], p2[1], p2[2]) == 1) hi = 2;
    code |= (unsigned)(lo + 2) << (2 + 4 * t);
    code |= (unsigned)hi << (4 + 4 * t);
  }
  bool all = true;
  for (int s1 = -1; s1 <= 1 && all; s1 += 2)
    for (int s0 = -1; s0 <= 1; s0 += 2) {
      int p[3] = {q[0], q[1], q[2]};
      p[t0] += s0 * r; p[t1] += s1 * r;
      if (classify(LF, p[0], p[1], p[2]) != 1) { all = false; break; }
    }
  if (all) code |= 1u << 10;
  return code;
}

// InterpBndryData (order 3) restated -- see oracle/pa_oracle.c cf_bndry_value -- for a ghost cell
// whose masks are `code` (cf_masks).  NF fields are interpolated from the SAME coarse component with
// shared weights and loads: field f sees the coarse value v when xf[f] == 0 and
// (v - MC.xa) * MC.xb when xf[f] != 0 (the progress variable as an affine view of the coarse phi).
// Per field the operation order is the reference's.  Coarse neighbours inside the FAB that holds
// the coarse cell of q are addressed relative to it (one owner-map lookup instead of eleven).
// the arithmetic of InterpBndryData on raw coarse values craw(a0, a1) = coarse(qc + a0 e_t0 + a1 e_t1): shared by the two
// ways of fetching them (owner map / coarse patch), so that both are the same operations in the same order
template <int NF, typename CRAW>
__device__ __forceinline__ void cf_interp_core(unsigned code, CRAW craw, const DMFView& MC, const int q[3], const int qc[3], int dir, int r,
                                               const int xf[NF], double b[NF]) {
  const int t0 = dir == 0 ? 1 : 0, t1 = dir == 2 ? 1 : 2;
#pragma unroll
  for (int f = 0; f < NF; ++f) b[f] = 0.0;
  double xi[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int td = t == 0 ? t0 : t1;
    const int lo = (int)((code >> (2 + 4 * t)) & 3u) - 2, hi = (int)((code >> (4 + 4 * t)) & 3u);
    const int N = hi - lo + 1;
    const int rem = q[td] - qc[td] * r;
    double c[3];
    if (r == 2) {
      xi[t] = rem ? 0.25 : -0.25;
      for (int m = 0; m < N; ++m) c[m] = g_cf_coef.tan[rem][lo + 2][hi][m];
    } else {
      double x[3];
      for (int m = 0; m < N; ++m) x[m] = (double)(lo + m);
      const double xInt = -0.5 + ((double)rem + 0.5) / (double)r;
      xi[t] = xInt;
      poly_interp_coeff(xInt, x, N, c);
    }
    for (int m = 0; m < N; ++m) {
      const double v = t == 0 ? craw(lo + m, 0) : craw(0, lo + m);
#pragma unroll
      for (int f = 0; f < NF; ++f) b[f] += c[m] * (xf[f] ? (v - MC.xa) * MC.xb : v);
    }
  }
  {
    const double v = craw(0, 0);
#pragma unroll
    for (int f = 0; f < NF; ++f) b[f] -= xf[f] ? (v - MC.xa) * MC.xb : v;
  }
  if (code & (1u << 10)) {
    const double rpp = craw(1, 1), rmp = craw(-1, 1), rmm = craw(-1, -1), rpm = craw(1, -1);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const double vpp = xf[f] ? (rpp - MC.xa) * MC.xb : rpp, vmp = xf[f] ? (rmp - MC.xa) * MC.xb : rmp;
      const double vmm = xf[f] ? (rmm - MC.xa) * MC.xb : rmm, vpm = xf[f] ? (rpm - MC.xa) * MC.xb : rpm;
      b[f] += ((xi[0] * xi[1]) * 0.25) * (((vpp - vmp) + vmm) - vpm);
    }
  }
}

template <int NF>
__device__ inline void cf_interp(unsigned code, const DLevelView& LC, const DMFView& MC, int ccomp, const int q[3], int dir, int r,
                                 const int xf[NF], bool& ok, double b[NF]) {
  const int qc[3] = {coarsen_idx(q[0], r), coarsen_idx(q[1], r), coarsen_idx(q[2], r)};
  const int t0 = dir == 0 ? 1 : 0, t1 = dir == 2 ? 1 : 2;
  int pq[3] = {qc[0], qc[1], qc[2]};
  const int cb = wrap_cell(LC, pq) ? owner_of(LC, pq) : -1;
  DBox Bc = {{0, 0, 0}, {-1, -1, -1}};
  const double* base = MC.data;
  long long st0 = 0, st1 = 0;
  if (cb >= 0) {
    Bc = LC.boxes[cb];
    const long long nxg = Bc.hi[0] - Bc.lo[0] + 1 + 2 * MC.ng, nyg = Bc.hi[1] - Bc.lo[1] + 1 + 2 * MC.ng;
    base = MC.data + MC.off[cb] + fab_index(Bc, MC.ng, MC.ncomp, ccomp, pq[0], pq[1], pq[2]);
    st0 = t0 == 0 ? 1 : nxg;          // t0 is x or y
    st1 = t1 == 1 ? nxg : nxg * nyg;  // t1 is y or z
  }
  // raw coarse value at qc + a0 e_t0 + a1 e_t1
  auto craw = [&](int a0, int a1) -> double {
    const int u = pq[t0] + a0, v = pq[t1] + a1;
    if (u >= Bc.lo[t0] && u <= Bc.hi[t0] && v >= Bc.lo[t1] && v <= Bc.hi[t1]) return base[a0 * st0 + a1 * st1];
    int cc[3] = {qc[0], qc[1], qc[2]};
    cc[t0] += a0; cc[t1] += a1;
    return crse_raw(LC, MC, ccomp, cc[0], cc[1], cc[2], ok);
  };
  cf_interp_core<NF>(code, craw, MC, q, qc, dir, r, xf, b);
}

// ---- coarse patches (DLevelView::cp): geometry of the patch of a special face of box B, and the same interpolation from it
#define PA_CP_MISSING 0x7FF8C0A45EC0FFEEll /* bit pattern of a patch cell that has no coarse owner (a NaN payload no data carries) */
__host__ __device__ __forceinline__ void cpatch_geom(const DBox& B, int dir, int side, int& plane, int& u0, int& v0, int& pw, int& ph) {
  const int t0 = dir == 0 ? 1 : 0, t1 = dir == 2 ? 1 : 2;
  plane = coarsen_idx(side ? B.hi[dir] + 1 : B.lo[dir] - 1, 2);
  u0 = coarsen_idx(B.lo[t0] - 1, 2) - 2;
  v0 = coarsen_idx(B.lo[t1] - 1, 2) - 2;
  pw = coarsen_idx(B.hi[t0] + 1, 2) + 2 - u0 + 1;
  ph = coarsen_idx(B.hi[t1] + 1, 2) + 2 - v0 + 1;
}
template <int NF>
__device__ inline void cf_interp_patch(unsigned code, const double* patch, const DBox& B, int side, const DMFView& MC, const int q[3], int dir,
                                       const int xf[NF], bool& ok, double b[NF]) {
  const int qc[3] = {coarsen_idx(q[0], 2), coarsen_idx(q[1], 2), coarsen_idx(q[2], 2)};
  const int t0 = dir == 0 ? 1 : 0, t1 = dir == 2 ? 1 : 2;
  int plane, u0, v0, pw, ph;
  cpatch_geom(B, dir, side, plane, u0, v0, pw, ph);
  const double* base = patch + (long long)(qc[t1] - v0) * pw + (qc[t0] - u0);
  auto craw = [&](int a0, int a1) -> double {
    const double v = base[a1 * pw + a0];
    if (__double_as_longlong(v) == PA_CP_MISSING) { ok = false; return 0.0; }
    return v;
  };
  cf_interp_core<NF>(code, craw, MC, q, qc, dir, 2, xf, b);
}

template <int NF>
__device__ inline void cf_bndry_values(const DLevelView& LF, const DLevelView& LC, const DMFView& MC, int ccomp,
                                       const int q[3], int dir, int r, const int xf[NF], bool& ok, double b[NF]) {
  cf_interp<NF>(cf_masks(LF, q, dir, r) | 1u, LC, MC, ccomp, q, dir, r, xf, ok, b);
}
__device__ inline double cf_bndry_value(const DLevelView& LF, const DLevelView& LC, const DMFView& MC, int ccomp,
                                        const int q[3], int dir, int r, bool& ok) {
  const int xf[1] = {MC.xform};
  double b[1];
  cf_bndry_values<1>(LF, LC, MC, ccomp, q, dir, r, xf, ok, b);
  return b[0];
}

// normal-direction Lagrange weights of MLMG applyBC at a coarse-fine face:
// points {-ratio/2 (bc), 0.5, 1.5, 2.5}, evaluated at -0.5, NX = min(len+1, 4)
__device__ __forceinline__ int cf_normal_coef(int blen, int ratio, double coef[4]) {
  const int NX = (blen + 1 < 4) ? blen + 1 : 4;
  if (ratio == 2) {
    for (int m = 0; m < NX; ++m) coef[m] = g_cf_coef.nrm[NX][m];
    return NX;
  }
  const double x[4] = {-0.5 * (double)ratio, 0.5, 1.5, 2.5};
  poly_interp_coeff(-0.5, x, NX, coef);
  return NX;
}

// Decode thread t of special face `entry` (DLevelView::sfaces): box, direction, side and the ghost
// cell q adjacent to the face; nlayer > 1 enumerates layers slowest (waves do not mix layers).
__device__ __forceinline__ bool sface_decode(const DLevelView& L, int entry, long long tt, int nlayer, int& b, DBox& B, int& dir, int& side,
                                             int q[3], int& layer) {
  const int e = L.sfaces[entry];
  b = e / 6;
  dir = (e % 6) >> 1;
  side = e & 1;
  B = L.boxes[b];
  const int t0 = (dir == 0) ? 1 : 0, t1 = (dir == 2) ? 1 : 2;
  const unsigned n0 = B.hi[t0] - B.lo[t0] + 1, n1 = B.hi[t1] - B.lo[t1] + 1;
  const unsigned fs = n0 * n1;
  if (tt >= (long long)fs * nlayer) return false;
  unsigned t = (unsigned)tt;  // 32-bit index arithmetic (a 64-bit division costs hundreds of instructions)
  layer = 0;
  while (t >= fs) { t -= fs; ++layer; }
  const unsigned r = t / n0;
  q[dir] = side ? B.hi[dir] + 1 : B.lo[dir] - 1;
  q[t0] = B.lo[t0] + (int)(t - r * n0);
  q[t1] = B.lo[t1] + (int)r;
  return true;
}
