/*
 * peleanalysis_amd.h -- C ABI of the MI355X-native PeleAnalysis stencil path.
 *
 * The reference (AMReX-Combustion/PeleAnalysis) has no plugin/FFI interface:
 * its boundary is (i) each tool's key=value command line + plotfile formats and
 * (ii) the per-FArrayBox call shape inside the MFIter loops of the tool mains.
 * This header is the C ABI at (ii); tools/ keeps (i).  Every entry point cites
 * the reference call site it replaces.  Plain pointers and sizes only; all
 * `double*` named dev* / stored in pa_mf are DEVICE pointers (HBM).
 *
 * Conventions: return 0 = OK, non-zero = error (text via pa_last_error); the
 * library never aborts.  All work is enqueued on the pa_ctx's HIP stream and is
 * asynchronous unless stated; pa_sync() waits.  A pa_ctx is not thread-safe;
 * use one per host thread (the reference's per-FAB loops are serial per rank).
 *
 * Array layout: AMReX FArrayBox layout -- [comp][k][j][i], i fastest, over the
 * valid box grown by ng ghost cells.  Boxes are inclusive cell-index boxes
 * {lo0,lo1,lo2,hi0,hi1,hi2}.
 */
#ifndef PELEANALYSIS_AMD_H
#define PELEANALYSIS_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pa_ctx pa_ctx;     /* device, stream, error text, scratch */
typedef struct pa_level pa_level; /* BoxArray + Geometry of one AMR level (host + device copies) */
typedef struct pa_mf pa_mf;       /* MultiFab: ncomp x (boxes grown by ng) doubles in HBM */

/* mirrors amrex::Array4 / FArrayBox: p = device pointer, lo/hi incl. ghosts, nstride = component
 * stride in doubles (Array4::nstride; 0 = contiguous nx*ny*nz like a plain FArrayBox) */
typedef struct { double* p; int32_t lo[3]; int32_t hi[3]; int32_t ncomp; int64_t nstride; } pa_fab;
typedef struct { int32_t lo[3]; int32_t hi[3]; } pa_box;

/* LinOpBCType subset used by grad.cpp:180-193 / curvature.cpp:428-441 */
enum { PA_BC_PERIODIC = 0, PA_BC_NEUMANN = 1, PA_BC_REFLECT_ODD = 2 };

/* ------------------------------------------------------------------ context */
int         pa_version(void);
pa_ctx*     pa_ctx_create(int device, void* hip_stream /* NULL: library-owned stream */);
void        pa_ctx_destroy(pa_ctx*);
const char* pa_last_error(const pa_ctx*);
int         pa_sync(pa_ctx*);
void*       pa_ctx_stream(pa_ctx*);
/* The library's environment switches (the PA_* table in peleanalysis_amd/csrc/pa_internal.h: pa_options) are read ONCE, when the
 * first context is created.  A caller that changes one afterwards -- a test, an A/B measurement inside one process -- asks for a
 * re-read with this call.  No reference counterpart (AMReX's ParmParse is read at amrex::Initialize, grad.cpp:31). */
void        pa_options_reload(void);

/* raw HBM buffers for callers without their own device allocator (vertex / triangle buffers of
 * the marching-cubes entry points, caller-owned FABs).  Copies are synchronous. */
void* pa_device_malloc(pa_ctx*, int64_t bytes);
void  pa_device_free(pa_ctx*, void* devptr);
int   pa_memcpy_h2d(pa_ctx*, void* devdst, const void* hostsrc, int64_t bytes);
int   pa_memcpy_d2h(pa_ctx*, void* hostdst, const void* devsrc, int64_t bytes);
/* device to device, also between the devices of two contexts of one process (peer copy over xGMI); synchronous */
int   pa_memcpy_d2d(pa_ctx*, void* devdst, const void* devsrc, int64_t bytes);
/* number of HIP devices visible to the process (0: none / no runtime) */
int   pa_device_count(void);

/* Per-launch timing of the library's own kernels with HIP events recorded on the
 * context's stream (what bench.py's roofline object reports).  Tags: 1 fused
 * grad->curvature, 2 its face fix-up, 3 FillBoundary, 4 applyBC, 5 grad,
 * 6 progress, 7 box filter, 8 marching cubes.  on: 0 off, 1 every tag, otherwise a bit mask (1 << tag) of the
 * tags to time.  pa_profile_read is synchronous. */
int pa_profile_enable(pa_ctx*, int on);
/* which variant of the fused grad->curvature sweep the last pa_gradcurv_* call launched, e.g.
 * "k_gradcurv_march3<MTY=13,CLIP=0,PAIR=0,CG=1>" (bench.py keys the committed PMC traffic figure on it) */
const char* pa_sweep_kernel_name(const pa_ctx*);
/* diagnostic: workgroups of the all-levels sweep kernel the runtime keeps resident per CU (which = 0: boxes wider than 32 cells,
 * 1: the narrow-box kernel); < 0: the query failed */
int pa_sweep_occupancy(pa_ctx*, int which);
/* diagnostic: cells that the clip-aware curvature fix-up of the last fused pass (threshold_prog, curvature.cpp:549-570) had to
 * recompute through its general path because a neighbour's normal was clipped; -1 if that path has never run.  Synchronous. */
int pa_last_slow_cells(pa_ctx*);
/* diagnostic: number of IRREGULAR cells of a level -- boundary cells of its local boxes next to a concave coarse-fine corner or to
 * the line where a box face changes from covered-by-a-neighbour to coarse-fine (general BoxArrays; the reference's MLPoisson /
 * FillBoundary code has no such distinction, curvature.cpp:426-546) -- whose curvature the fused grad->curvature pass recomputes
 * cell by cell after its sweep.  0 for the nested, convex hierarchies of the SURVEY's configs.  Builds the list on first use;
 * synchronous; -1 on error. */
int64_t pa_level_irregular_cells(pa_ctx*, const pa_level*);
int pa_profile_read(pa_ctx*, int tag, int64_t* nlaunch, double* total_ms, int reset);

/* ---------------------------------------------------- level = BoxArray+Geometry
 * replaces: amrData.boxArray(lev) / Geometry(ProbDomain, rb, coord, is_per)
 * (grad.cpp:160-163, curvature.cpp:287-291).  dx = (prob_hi-prob_lo)/ncells. */
pa_level* pa_level_create(pa_ctx*, int nboxes, const int32_t* boxes6, const int32_t domlo[3],
                          const int32_t domhi[3], const int32_t is_per[3], const double prob_lo[3],
                          const double prob_hi[3]);
/* ------------------------------------------------------- multi-GPU: one rank per GPU
 * The reference distributes the boxes of every level over MPI ranks with DistributionMapping(ba)
 * (grad.cpp:162, curvature.cpp:289, filterPlt.cpp:142, isosurface.cpp:1441) and moves ghost data with
 * point-to-point messages inside FillBoundary / FillPatch / the MLMG boundary registers.  Here a rank is one
 * pa_ctx (one process per GPU, or one host thread per GPU inside a tool).  A level created with
 * pa_level_create_sharded knows the whole BoxArray and its owner map; every entry point that fills ghost cells
 * (pa_fill_boundary, pa_apply_bc, pa_fillpatch_two_levels, the pa_*_run pipelines) then also performs the
 * cross-rank half -- pack kernels -> one grouped point-to-point exchange -> unpack kernels -- through the
 * context's transport, and pa_*_run reduces the progress-variable range over the ranks (curvature.cpp:147-148).
 * All ranks must make the same sequence of calls (as MPI ranks of the reference do). */
typedef struct { int32_t peer; double* sendbuf; int64_t nsend; double* recvbuf; int64_t nrecv; } pa_xfer;
typedef struct {
  void*   user;
  int32_t rank, nranks;
  /* One grouped exchange: for every i send nsend doubles from sendbuf to rank peer and receive nrecv doubles from it
   * into recvbuf (device pointers; either count may be 0).  Several entries may name the same peer: they are matched
   * in list order on both sides.  Stream-ordered: the call consumes data produced by work already enqueued on
   * hip_stream, and work enqueued on it afterwards sees the received data.
   * POINT-TO-POINT contract (like MPI_Isend/Irecv inside FillBoundary): the library calls exchange on a rank only when
   * that rank has at least one entry, so a rank without a box of the level, or without a neighbour on another rank,
   * does not call it -- an implementation must pair sends with receives per (sender, receiver) in FIFO order and must
   * never wait for ranks that are not named in x. */
  int (*exchange)(void* user, void* hip_stream, int32_t n, const pa_xfer* x);
  /* vals[i] = reduction over the ranks of vals[i] (host memory); op: 0 min, 1 max, 2 sum */
  int (*allreduce)(void* user, double* vals, int32_t n, int32_t op);
} pa_comm;
/* caller-supplied transport (tests: gloo through host memory; tools: peer copies between host threads) */
int pa_ctx_set_comm(pa_ctx*, const pa_comm*);
/* built-in transport: RCCL over xGMI (grouped ncclSend / ncclRecv on the context's stream, ncclAllReduce).
 * pa_rccl_unique_id fills 128 bytes on one rank; the caller broadcasts them; every rank then calls pa_ctx_init_rccl. */
/* diagnostic transport for PROJECTIONS of an N-rank run on one GPU (bench.py --sim-of N --xdelay-us / --xlink-GBs): moves no
 * data; every exchange enqueues on its stream a kernel that spins for fixed_us + (most bytes to or from one peer) / link_GBs,
 * the cost model of a grouped point-to-point exchange over per-peer xGMI links -- the schedule hides or exposes that time as it
 * would a real exchange.  Ghost-cell results are wrong by construction.  pa_delay_comm_stats: exchanges issued / modelled time. */
int pa_ctx_set_delay_comm(pa_ctx*, int nranks, int rank, double fixed_us, double link_GBs);
int pa_delay_comm_stats(const pa_ctx*, int64_t* calls, double* total_us);
int pa_rccl_unique_id(pa_ctx*, void* id128);
int pa_ctx_init_rccl(pa_ctx*, int nranks, int rank, const void* id128);
int pa_ctx_nranks(const pa_ctx*);
/* transport check: a ring exchange of n doubles with known contents + a max-reduction with a known answer; synchronous */
int pa_comm_selftest(pa_ctx*, int64_t n);
/* the transport's reduction over the ranks (no-op on one rank): op 0 min, 1 max, 2 sum */
int pa_allreduce(pa_ctx*, double* vals, int n, int op);
/* DistributionMapping(ba) restated: boxes in Morton order of their low corners, cut into nranks contiguous pieces of
 * (nearly) equal cell count.  Host arithmetic only.  owner[b] in [0, nranks). */
int pa_distribution_map(int nboxes, const int32_t* boxes6, int nranks, int32_t* owner);
/* This rank's share of a level: the whole BoxArray + the owner rank of every box.  The level's boxes are the ones with
 * owner == rank, in BoxArray order (pa_level_global_ids gives their indices in the BoxArray). */
pa_level* pa_level_create_sharded(pa_ctx*, int nboxes, const int32_t* boxes6, const int32_t* owner, int rank, int nranks,
                                  const int32_t domlo[3], const int32_t domhi[3], const int32_t is_per[3],
                                  const double prob_lo[3], const double prob_hi[3]);
int pa_level_global_ids(const pa_level*, int32_t* gids /* [pa_level_nboxes] */);
/* The region lists behind those exchanges, as plain host arithmetic (no GPU; what the CPU-tier tests check against the
 * undistributed answer).  Rows of 9 int32: {kind, peer, box, lo0, lo1, lo2, hi0, hi1, hi2}; returns the number of rows
 * (writes at most cap).  kind 0 = send (box = global index of a box this rank owns, region in its index space),
 * kind 1 = receive (FillBoundary: box = global index of the destination box, ghost region).
 * pa_plan_coarse_source: the coarse data rank `rank` needs under the coarse-fine faces of its fine boxes (mode 0: the
 * stencils of pa_apply_bc) or under the ng ghost layers of its fine boxes grown by `halo` coarse cells (mode 1:
 * pa_fillpatch_two_levels); kind 2 = piece (box = global coarse box it is cut from, peer = its owner; pieces are
 * disjoint and listed in the order of the rank's coarse-source BoxArray), kind 0 = what this rank sends to `peer`. */
int64_t pa_plan_fill_boundary(int nboxes, const int32_t* boxes6, const int32_t* owner, int rank, const int32_t domlo[3],
                              const int32_t domhi[3], const int32_t is_per[3], int ng, int32_t* rows9, int64_t cap);
int64_t pa_plan_coarse_source(int nfine, const int32_t* fboxes6, const int32_t* fowner, const int32_t fdomlo[3],
                              const int32_t fdomhi[3], int ncrse, const int32_t* cboxes6, const int32_t* cowner,
                              const int32_t cdomlo[3], const int32_t cdomhi[3], const int32_t is_per[3], int rank,
                              int mode, int ng, int halo, int32_t* rows9, int64_t cap);
/* pa_plan_restriction: what moves between ranks in the distributed smoothing solve (pa_smooth_solve on sharded levels; what
 * amrex::average_down and MLMG's flux register move, curvature.cpp:328-406).  which = 0: the child averages -- kind 0 = send
 * (box = global FINE box this rank owns, region = part of that box coarsened by `ratio`), kind 1 = receive (box = global COARSE
 * box, the same region: its valid cells), kinds 3 / 4 = source / destination of a same-rank copy, in pairs.  which = 1 + 2 * dir
 * + side: the flux register of that face orientation -- the source regions lie in the one-cell ghost slab behind face (dir, side)
 * of the coarsened fine box (special faces only), the destination regions are those cells folded into the coarse domain through
 * its periodic images.  Both sides of a pair of ranks list their regions in the same order. */
int64_t pa_plan_restriction(int nfine, const int32_t* fboxes6, const int32_t* fowner, const int32_t fdomlo[3],
                            const int32_t fdomhi[3], int ncrse, const int32_t* cboxes6, const int32_t* cowner,
                            const int32_t cdomlo[3], const int32_t cdomhi[3], const int32_t is_per[3], int rank, int ratio,
                            int which, int32_t* rows9, int64_t cap);
/* Internal re-tiling (host arithmetic only): another BoxArray with exactly the cells of boxes6 -- boxes at least min_thick cells
 * thick in every direction merged into the largest rectangles the cell set allows (maximal x-runs, stacked in y, then z; at
 * most max_size[d] cells per direction, cut as evenly as the input edges allow; an input box larger than that is not split),
 * thinner boxes passed through unchanged at the end of the list.  grad.cpp:158-236, curvature.cpp:283-570 and
 * Filter::apply_filter (filterPlt.cpp:206-219) are point-wise functions of the level's cell set except for the coarse-fine
 * interpolant of a box fewer than 3 cells thick (MLCellLinOp::applyBC, order min(n + 1, 4)), so with min_thick >= 3 a level
 * created on the returned boxes gives the same value in every cell -- bit for bit -- as one created on boxes6
 * (tests/test_retile.py; filterPlt.cpp:141 re-chops the file's BoxArray itself: the tiling is not part of the tools'
 * contract).  Marching cubes numbers nodes in box order and must keep the file's boxes.  The caller moves FAB data between
 * the two tilings by box intersection (tools/common/pa_plotfile.h read_comp / write_plotfile, hierarchy.regrid_copy).
 * out_boxes6: room for cap >= nboxes boxes (an L-shaped union of two boxes may come back as three rectangles); returns the
 * number of boxes written (the input itself when nothing can be merged, a piece would come out thinner than min_thick or cap
 * is too small for the result), -1 on bad arguments.  max_size NULL: 128 per direction. */
int pa_level_retile(int nboxes, const int32_t* boxes6, const int32_t max_size[3], int min_thick, int32_t* out_boxes6, int cap);
/* The max_size the tools and bench.py pass to pa_level_retile for the levels of one hierarchy (boxes6[l]: the nboxes[l] boxes of
 * level l): 512 x 256 x 256 cells where every level then consists of blocks at least 128 cells thick, else 128^3 (measured:
 * profiles/r05_retile.txt); PA_RETILE_MAX="x y z" in the environment overrides.  Host arithmetic only; 0 = OK. */
int pa_hierarchy_retile_limits(int nlev, const int32_t* nboxes, const int32_t* const* boxes6, int min_thick, int32_t max_size[3]);
/* the same for a hierarchy sharded over nranks ranks: the largest of 512 x 256 x 256 / 256^3 / 256 x 256 x 128 / 128^3 that leaves every level at least
 * 4 nranks boxes (nranks <= 1: pa_hierarchy_retile_limits) */
int pa_hierarchy_retile_limits_ranks(int nlev, const int32_t* nboxes, const int32_t* const* boxes6, int min_thick, int nranks, int32_t max_size[3]);
void      pa_level_destroy(pa_level*);
int       pa_level_nboxes(const pa_level*);

/* -------------------------------------------------------------- MultiFab
 * pa_mf_layout is pure host arithmetic (no GPU): offset (in doubles) of each box in the flat
 * buffer and its component stride (>= nx*ny*nz incl. ghosts: rounded up to 512 B and kept off
 * multiples of 16 KiB so that the components of one cell do not share an HBM channel); returns
 * the total size in doubles.  Inside a component the layout is the FArrayBox one ([k][j][i]).
 * replaces: MultiFab(ba, dm, ncomp, ngrow) (grad.cpp:164, curvature.cpp:294). */
int64_t pa_mf_layout(int nboxes, const int32_t* boxes6, int ncomp, int ng, int64_t* off, int64_t* cstride);
pa_mf*  pa_mf_create(pa_ctx*, const pa_level*, int ncomp, int ng, double* devptr /* NULL: library allocates */);
void    pa_mf_destroy(pa_mf*);
double* pa_mf_data(pa_mf*);
int64_t pa_mf_size(const pa_mf*);
int     pa_mf_upload(pa_ctx*, pa_mf*, const double* host);  /* synchronous */
int     pa_mf_download(pa_ctx*, const pa_mf*, double* host); /* synchronous */
/* components comp .. comp + ncomp - 1 only; host is laid out like the whole multifab (pa_mf_layout), its other components are not
 * read / written (a tool's multifab that holds inputs AND outputs, grad.cpp:164: only the inputs go up, only the outputs come back) */
int     pa_mf_upload_comps(pa_ctx*, pa_mf*, const double* host, int comp, int ncomp);   /* synchronous */
int     pa_mf_download_comps(pa_ctx*, const pa_mf*, double* host, int comp, int ncomp); /* synchronous */
int     pa_mf_setval(pa_ctx*, pa_mf*, int comp, int ncomp, double v); /* whole fabs incl. ghosts */
/* MultiFab::Copy(dst, src, scomp, dcomp, ncomp, ng) -- same BoxArray */
int     pa_mf_copy(pa_ctx*, const pa_mf* src, int scomp, pa_mf* dst, int dcomp, int ncomp, int ng);

/* ------------------------------------------------------------ ghost cells
 * FabArray::FillBoundary(comp, ncomp, periodicity): grad.cpp:169,
 * curvature.cpp:322,484,488,502; isosurface.cpp:1468. */
int pa_fill_boundary(pa_ctx*, pa_mf*, int comp, int ncomp, int ng);
/* MLCellLinOp::applyBC on the ring-1 face ghosts (inside MLMG::getFluxes,
 * grad.cpp:212-213, curvature.cpp:445-457,518-531): covered cells untouched,
 * physical walls Neumann / reflect_odd, coarse-fine cells = InterpBndryData
 * (order 3, tangential) + cubic in the normal direction (setMaxOrder(4)).
 * crse may be NULL on level 0.  only_dir = -1 for all three directions. */
int pa_apply_bc(pa_ctx*, pa_mf* fine, int comp, const pa_mf* crse, int ccomp, const int32_t bc[3],
                int ratio, int only_dir);
/* coarse-fine ghost cells (since the last call) whose coarse data was missing: improper
 * nesting.  Synchronous; resets the counter.  0 = all ghost fills were well defined. */
int pa_bc_errors(pa_ctx*);

/* --------------------------------------------------- level-batched kernels
 * One launch covers every box of the level (many small FABs per launch). */
/* grad.cpp:211-236: gx,gy,gz,|g| of phi[comp] into out[ocomp..ocomp+3] */
int pa_grad_level(pa_ctx*, const pa_mf* phi, int comp, pa_mf* out, int ocomp);
/* curvature.cpp:139-149 (AmrData::MinMax of one level; valid cells) */
int pa_minmax_level(pa_ctx*, const pa_mf* s, int comp, double* mn, double* mx); /* synchronous */
/* curvature.cpp:310-321: c = (s - pmin) * (1/(pmax-pmin)), ng ghost layers too */
int pa_progress_level(pa_ctx*, const pa_mf* s, int comp, double pmin, double pmax, pa_mf* c, int ccomp, int ng);
/* curvature.cpp:451-502: G = grad c, normgrad = -max(1e-14,|G|), n = G/normgrad
 * (valid cells).  G, normgrad may be NULL (not stored). */
int pa_normal_level(pa_ctx*, const pa_mf* c, int comp, pa_mf* G, int gcomp, pa_mf* normgrad, int ngcomp,
                    pa_mf* n, int ncomp0);
/* curvature.cpp:508-546: K = scale * sum_d d n_d/dx_d  (n: 3 comps with
 * resolved face ghosts), + threshold clip :549-567 if thr >= 0 (c needed) */
int pa_div_level(pa_ctx*, pa_mf* n, int ncomp0, double scale, const pa_mf* c, int ccomp, double thr,
                 pa_mf* K, int kcomp);
/* same, only where the fused path reads a stored progress variable: for every box face that has
 * a ghost cell which is not a valid cell of the level (coarse-fine or wall), the `depth` valid
 * layers + ng ghost layers behind/beyond that face over the grown tangential extent.  Other cells
 * of c are left untouched. */
int pa_progress_shell_level(pa_ctx*, const pa_mf* s, int comp, double pmin, double pmax, pa_mf* c, int ccomp, int ng,
                            int depth);
/* fused grad->curvature (headline kernel; grad.cpp:211-236 + curvature.cpp:316-320,451-567 in
 * one sweep).  Reads phi only (ng>=2: FillBoundary(2) + applyBC'd face ghosts; 8 B/cell) and
 * writes out[ocomp..+3] = gx,gy,gz,|g|, out[ocomp+4..+6] = FlameNormal, out[ocomp+7] =
 * MeanCurvature (64 B/cell); the progress variable (phi-pmin)/(pmax-pmin) is formed on chip.
 * Cells within two layers of a coarse-fine or physical face of their box then get N and K from
 * pa_gradcurv_faces_level, which applies the reference's boundary conditions on c and n there
 * (c: ng>=2, FillBoundary(2) + applyBC on face and edge ghosts, valid within 4 cells of the faces). */
int pa_gradcurv_level(pa_ctx*, const pa_mf* phi, int pcomp, double prog_min, double prog_max, double thr,
                      pa_mf* out, int ocomp);
int pa_gradcurv_faces_level(pa_ctx*, const pa_mf* c, int ccomp, const pa_mf* crse_n /* NULL on level 0 */,
                            int cncomp0, const int32_t bc[3], int ratio, double thr, pa_mf* out, int ncomp0,
                            int kcomp);

/* -------------------------------------------------------- per-FAB entry points
 * (the body of one MFIter iteration; device pointers in pa_fab) */
/* grad.cpp:211-236 for one FAB; phi has >=1 resolved ghost layer */
int pa_grad_fab(pa_ctx*, pa_box valid, const pa_fab* phi, int comp, const double dxinv[3], pa_fab* out, int ocomp);
/* curvature.cpp:316-320 */
int pa_progress_fab(pa_ctx*, pa_box bx, const pa_fab* s, int comp, double pmin, double pmax, pa_fab* c, int ccomp);
/* curvature.cpp:451-502 */
int pa_normal_fab(pa_ctx*, pa_box valid, const pa_fab* c, int comp, const double dxinv[3], pa_fab* G, int gcomp,
                  pa_fab* normgrad, int ngcomp, pa_fab* n, int ncomp0);
/* curvature.cpp:508-546 */
int pa_div_fab(pa_ctx*, pa_box valid, const pa_fab* n, int ncomp0, const double dxinv[3], double scale,
               pa_fab* K, int kcomp);
/* fused sweep for one FAB (phi with 2 ghost layers; exact where they hold same-level data) */
int pa_gradcurv_fab(pa_ctx*, pa_box valid, const pa_fab* phi, int pcomp, double prog_min, double prog_max,
                    const double dxinv[3], double thr, pa_fab* out, int ocomp);
/* filterPlt.cpp:217 Filter::apply_filter(box, in, out) */
int pa_boxfilter_fab(pa_ctx*, pa_box valid, const pa_fab* in, pa_fab* out, int scomp, int ncomp, int ng,
                     const double* w /* host, 2ng+1 */);

/* ----------------------------------------------------------------- filterPlt
 * Filter(type=box, fgr) weights (filterPlt.cpp:136-137); returns ngrow */
int pa_box_filter_weights(int fgr, double* w);
/* filterPlt.cpp:80 filter_type -> the PelePhysics Filter weights for the types restated here: 0 none, 1 box, 3 / 7 the
 * 3-point and 4 / 8 the 5-point approximations of the box / Gaussian filter (w: at least max(fgr + 2, 5) doubles).
 * Returns ngrow, or -1 for a type that is not available (5 6 9 10 "optimized"; 2 Gaussian unless PA_ALLOW_UNVERIFIED_GAUSSIAN=1 is
 * in the environment: its weights -- the textbook kernel sampled at cell centres, cut at 4 standard deviations -- could not be
 * checked against PelePhysics, whose source is not in the reference tree) or fgr < 1.  Host only.
 * [weights re-derived from the moment conditions; PelePhysics is not part of the reference tree: parity unpinned] */
int pa_filter_weights(int type, int fgr, double* w);
/* filterPlt.cpp:206-219, all boxes of a level */
int pa_boxfilter_level(pa_ctx*, const pa_mf* in, pa_mf* out, int scomp, int ncomp, int ng, const double* w);
/* Filter::apply_filter on every level of a hierarchy in one call (the level loop of filterPlt.cpp:206-219): in[l] / out[l] / ngs[l] /
 * ws[l] as pa_boxfilter_level's arguments for level l (the levels one after the other on the context's stream). */
int pa_boxfilter_hierarchy(pa_ctx*, int nlev, const pa_mf* const* in, pa_mf* const* out, int scomp, int ncomp, const int32_t* ngs,
                           const double* const* ws);
/* the AMREX_SPACEDIM == 2 build of the same call on a level stored as one plane of cells (k = 0):
 * out(i,j,c) = sum_m sum_l (w_l w_m) in(i+l, j+m, c) */
int pa_boxfilter_level2d(pa_ctx*, const pa_mf* in, pa_mf* out, int scomp, int ncomp, int ng, const double* w);
/* filterPlt.cpp:174-203 ghost fill pieces */
int pa_foextrap(pa_ctx*, pa_mf*, int comp, int ncomp, int ng);
int pa_fillpatch_two_levels(pa_ctx*, pa_mf* fine, const pa_mf* crse, int comp, int ncomp, int ng, int ratio,
                            int interp_type);

/* The ghost fill of a whole hierarchy in three launches: pa_fill_boundary of every level, pa_fillpatch_two_levels of every level
 * pair, and (foextrap != 0) pa_foextrap of every level -- filterPlt.cpp:159-203 with foextrap, isosurface.cpp:1468-1524 (PCInterp,
 * interp_type 0) without.  mfs[l] on level l (coarse first), ngs[l] ghost layers on level l.  Results identical to the per-level
 * calls (no step of a level reads what another level's step writes); sharded levels take the per-level calls. */
int pa_fill_ghosts_hierarchy(pa_ctx*, int nlev, pa_mf* const* mfs, int comp, int ncomp, const int32_t* ngs, int ratio, int interp_type, int foextrap);

/* ---------------------------------------------------------------- isosurface
 * isosurface.cpp:1531-1592 for one FAB: state = 3 coordinate comps + fields,
 * mask (<0 = covered by a finer level), loop = box of cube base points.
 * pa_mc_count_fab classifies and counts (wave ballot + popcount); pa_mc_emit_fab
 * writes vertices in the reference's vertCache (std::map<Edge>) order and
 * triangles in cube traversal order; both deterministic.  Synchronous. */
int pa_mc_count_fab(pa_ctx*, pa_box loop, const pa_fab* state, const pa_fab* mask, int isocomp, double isoval,
                    int64_t* nvert, int64_t* ntri);
int pa_mc_emit_fab(pa_ctx*, pa_box loop, const pa_fab* state, const pa_fab* mask, int isocomp, double isoval,
                   double* dev_verts /* [nvert][ncomp] */, int32_t* dev_vkeys /* [nvert][6] */,
                   int32_t* dev_tris /* [ntri][3] */, int64_t nvert, int64_t ntri);
/* The same for every FAB of a level in one pass (the whole MFIter loop of isosurface.cpp:1531-1592).
 * pa_iso_mask_level: mask = 1, and -1 on every cell (ghost cells included) covered by the next finer level
 * (isosurface.cpp:1540-1563; fine = NULL: nothing covered).  pa_mc_level: state and mask are multifabs of one
 * level with the same ghost width; loops[b] = cube base points of FAB b (lo > hi: FAB skipped).  Per-FAB results
 * are identical to pa_mc_count_fab / pa_mc_emit_fab and are concatenated in box order: FAB b owns vertices
 * [sum_{b'<b} nvert[b'], +nvert[b]) and its triangles hold FAB-local vertex ids.  The three output arrays are
 * parts of ONE device allocation made by the library whose base is *dev_verts (all NULL when the level has no
 * surface): release it with pa_device_free(*dev_verts) only.  Synchronous. */
int pa_iso_mask_level(pa_ctx*, pa_mf* mask, int comp, const pa_level* fine, int ratio);
/* isosurface.cpp:1458-1465: comps comp0..comp0+2 of every cell of every grown FAB = its cell-centre coordinates,
 * (index + 0.5) * dx + prob_lo in that operation order (what FillBoundary / FillPatch then overwrite in ghost cells) */
int pa_iso_coords_level(pa_ctx*, pa_mf* state, int comp0);
int pa_mc_level(pa_ctx*, const pa_mf* state, const pa_mf* mask, int mcomp, const pa_box* loops /* host [nboxes] */,
                int isocomp, double isoval, int64_t* nvert /* host [nboxes] */, int64_t* ntri /* host [nboxes] */,
                double** dev_verts /* [sum nvert][ncomp] */, int32_t** dev_vkeys /* [sum nvert][6] */,
                int32_t** dev_tris /* [sum ntri][3] */);
/* AMREX_SPACEDIM == 2 builds of the reference (Segmentise, isosurface.cpp:303-406; the loop :1574-1582): a 2-D level
 * is stored as ONE plane of cells, k = 0 (boxes lo[2] = hi[2] = 0; ghost planes in z are ignored); state = 2
 * coordinate components + fields; loops[b] = square base points, lo[2] = hi[2] = 0.  Vertices in vertCache order
 * ((j, i) of the edge's lower endpoint, x before y), segments in traversal order, as rows of THREE int32
 * (id0, id1, -1) with FAB-local vertex ids; everything else as pa_mc_level (one allocation, base *dev_verts). */
int pa_msq_level(pa_ctx*, const pa_mf* state, const pa_mf* mask, int mcomp, const pa_box* loops /* host [nboxes] */,
                 int isocomp, double isoval, int64_t* nvert /* host [nboxes] */, int64_t* nseg /* host [nboxes] */,
                 double** dev_verts /* [sum nvert][ncomp] */, int32_t** dev_vkeys /* [sum nvert][6] */,
                 int32_t** dev_segs /* [sum nseg][3] */);
/* The same two calls with the mask of isosurface.cpp:1540-1563 evaluated inside the cell pass instead of read from a
 * multifab: a cell is masked iff its refined image has an owner on `fine` (periodic images included; fine = NULL:
 * nothing is masked, as when the distance function is built).  Halves the bytes the pass reads. */
int pa_mc_level_fine(pa_ctx*, const pa_mf* state, const pa_level* fine, int ratio, const pa_box* loops, int isocomp,
                     double isoval, int64_t* nvert, int64_t* ntri, double** dev_verts, int32_t** dev_vkeys,
                     int32_t** dev_tris);
int pa_msq_level_fine(pa_ctx*, const pa_mf* state, const pa_level* fine, int ratio, const pa_box* loops, int isocomp,
                      double isoval, int64_t* nvert, int64_t* nseg, double** dev_verts, int32_t** dev_vkeys,
                      int32_t** dev_segs);
/* The level loop of isosurface.cpp:1434-1728 as ONE call: pa_mc_level_fine on every level of the hierarchy, with the cell
 * passes and counts of all levels enqueued back to back, ONE read-back of the per-FAB counts, ONE pooled allocation for the
 * surfaces of all levels and ONE final synchronisation (per level the reference pays an MFIter loop; here a level costs no
 * host round trip of its own).  states[l] lives on level l; fine_mask[l] != 0: cells covered by level l + 1 are masked
 * (isosurface.cpp:1540-1563; null: nothing is masked); loops[l] / nvert[l] / ntri[l]: host arrays over the FABs of level l.
 * dev_verts[l] / dev_vkeys[l] / dev_tris[l] (host arrays of nlev pointers) point into *block (null for a level without
 * surface); free *block with pa_device_free.  Per-level results are identical to pa_mc_level_fine's. */
int pa_mc_hierarchy_fine(pa_ctx*, int nlev, const pa_mf* const* states, const int32_t* fine_mask, int ratio,
                         const pa_box* const* loops, int isocomp, double isoval, int64_t* const* nvert, int64_t* const* ntri,
                         double** dev_verts, int32_t** dev_vkeys, int32_t** dev_tris, void** block);
/* pa_mc_hierarchy_fine WITHOUT the three coordinate components: fields[l] holds only the plotfile components (the iso field at
 * isocomp + the mapped ones, ghost cells filled by pa_fill_ghosts_hierarchy with PCInterp).  The coordinates isosurface.cpp:1458-1465
 * stores per cell and :1468-1478 FillBoundaries / FillPatches in ghost cells are a function of the cell index and of which level
 * covers the cell -- (i + 0.5) dx + plo where the level itself covers it (through periodic images too), the coarse parent's centre
 * in every other ghost cell -- so the vertex kernels form them in registers, with the operations of pa_iso_coords_level: vertices
 * ([nvert][3 + ncomp]), keys and triangles are bit-identical to pa_mc_hierarchy_fine's on the reference-shaped state, and 24 B
 * per cell (+ their ghost fills) are never written.  3-D levels; level l must be the `ratio` refinement of level l - 1. */
int pa_mc_hierarchy_xyz(pa_ctx*, int nlev, const pa_mf* const* fields, const int32_t* fine_mask, int ratio, const pa_box* const* loops,
                        int isocomp, double isoval, int64_t* const* nvert, int64_t* const* ntri, double** dev_verts, int32_t** dev_vkeys,
                        int32_t** dev_tris, void** block);
/* isosurface.cpp:1687-1726 + 1751-1812 on the device: the global node / element sets from the per-FAB fragments, in
 * insertion order (level by level, FAB by FAB: exactly the fragments whose ntri > 0, as the reference skips the others).
 * A vertex within 1e-15 (Euclidean) of an earlier node IS that node (Node::operator<, :834-873), otherwise a new node
 * numbered by insertion rank; elements = node-id triples rotated so the smallest id leads, degenerate ones dropped,
 * unique, sorted (std::set<Element>, :877-927).  verts / tris are DEVICE pointers ([nvert][ncomp] with the position in
 * the first three components; [ntri][3] fragment-local vertex ids) -- e.g. slices of what pa_mc_level* returned.
 * Outputs: two device allocations ([nnodes][ncomp], [nelts][3]; release each with pa_device_free; NULL when empty).
 * Returns 0, or 1 on error, or 2 when some cluster of nearby vertices is not transitive under the tolerance (a ~ b,
 * b ~ c, a !~ c): the reference's answer then depends on the insertion chain, nothing is returned, and the caller runs
 * the sequential merge (tools/common/pa_isomerge.h).  Synchronous. */
typedef struct { const double* verts; int64_t nvert; const int32_t* tris; int64_t ntri; } pa_iso_frag;
int pa_iso_merge(pa_ctx*, int nfrag, const pa_iso_frag* frags /* host array */, int ncomp, int64_t* nnodes,
                 double** dev_nodes, int64_t* nelts, int32_t** dev_elts);
const uint16_t* pa_mc_edge_table(void); /* [256] host */
const int8_t*   pa_mc_tri_table(void);  /* [256][16] host */

/* ----------------------------------------------- isosurface: distance function
 * isosurface.cpp:1595-1655 (build_distance_function).  One pa_sdf_grid = one call of
 * make_level_set3(faceList, vertList, local_origin, dx, ni, nj, nk, phi_grid) (isosurface.cpp:1625,
 * Tools/SDFGen/makelevelset3.cpp:118-185): unsigned distance, in the reference's float arithmetic and
 * visiting order, from grid point (i,j,k) = origin + (i,j,k)*dx to the triangle mesh; phi is
 * [k][j][i] with i fastest (Array3f order), ni*nj*nk floats.  All pointers are DEVICE pointers; tri
 * holds 3 vertex indices per triangle, x 3 floats per vertex.  A batch of grids runs concurrently
 * (one workgroup per grid for the sweeps).  Asynchronous on the context's stream. */
typedef struct {
  int64_t ntri;  const uint32_t* tri;
  int64_t nvert; const float* x;
  float origin[3]; float dx;
  int32_t n[3];
  float* phi;
} pa_sdf_grid;
int pa_sdf_level_set3(pa_ctx*, int ngrids, const pa_sdf_grid* grids /* host array */, int exact_band /* reference default 1 */);
/* isosurface.cpp:1637-1650: dist(i,j,k,dcomp) = sgn * min(dmax, phi(i-lo,j-lo,k-lo)) over vbox (the
 * distance FAB's box incl. ghosts), sgn = state(i,j,k,isocomp) < isoval ? -1 : +1 */
int pa_sdf_signed_fab(pa_ctx*, pa_box vbox, const float* dev_phi, const pa_fab* state, int isocomp, double isoval,
                      double dmax, pa_fab* dist, int dcomp);

/* ------------------------------------------------------------- streamlines
 * partStream.cpp:121-207 / StreamPC.cpp: two lines per seed (line 2s forward, 2s+1 backward) of nsteps
 * points each, RK4 with step dt (= hRK * finest dx in the tool) through the trilinear interpolant of
 * vfield[lev] comps vcomp..vcomp+2, whose nGrow ghost layers the caller has filled (FillPatch with
 * piecewise-constant interpolation + FillBoundary, partStream.cpp:160-177).  A line interpolates from the
 * grid it was last assigned to; all lines are re-assigned to the finest level containing them whenever one
 * leaves its grid grown by nGrow-1 (StreamPC.cpp:88-141).  seeds: host [nseed][3]; dev_pos: device
 * [2*nseed][nsteps][3].  Returns non-zero where the reference aborts with "bad RK".  Synchronous. */
int pa_stream_trace(pa_ctx*, int nlev, pa_mf* const* vfield, int vcomp, int64_t nseed, const double* seeds, int nsteps,
                    double dt, double* dev_pos, int32_t* nredist /* may be NULL */);
/* The same with the lines dealt to the ranks of the context's transport (partStream.cpp under MPI: the particles live on
 * the ranks, StreamPC.cpp:88-141 Redistribute is collective): every rank holds the WHOLE vector field (levels created
 * unsharded) and traces its own seeds; with share_flags != 0 the per-step "a line has left its grid" flag is max-reduced
 * over the ranks, so every line equals the one-rank run's.  Every rank must call it, also with nseed = 0. */
int pa_stream_trace_ranks(pa_ctx*, int nlev, pa_mf* const* vfield, int vcomp, int64_t nseed, const double* seeds, int nsteps,
                          double dt, double* dev_pos, int32_t* nredist /* may be NULL */, int share_flags);

/* ------------------------------------------------------------ tool pipelines
 * The level loops of the tool mains, operating on device-resident MultiFabs.
 * levels/state/out are arrays of nlev pointers, coarse first. */
/* grad.cpp:158-236.  state[lev]: comp = gradVar, ng >= 1.  out[lev][ocomp..+3]. */
int pa_grad_run(pa_ctx*, int nlev, pa_mf* const* state, int comp, const int32_t bc[3], pa_mf* const* out, int ocomp);

typedef struct {
  double  prog_min, prog_max; /* if prog_min > prog_max: use file min/max over the levels */
  int32_t do_threshold;       /* threshold_prog */
  double  threshold;          /* threshold_value */
  int32_t fused;              /* 1: fused grad->curvature kernels, 0: pass-by-pass */
  /* options of curvature.cpp:575-789 (pa_curvature_run only) */
  int32_t do_gauss_curv;      /* do_gaussCurv  (:575-677) */
  int32_t do_strain;          /* do_strain     (:679-749), keeps the reference's result = div u (quirk Q3) */
  int32_t get_strain_tensor;  /* getStrainTensor (:755-757): the 9 components of grad u */
  int32_t do_velnormal;       /* do_velnormal  (:765-787) */
  int32_t vel_comp;           /* first of the 3 consecutive velocity components in state */
  /* do_smooth (:328-406, pa_curvature_run only): one implicit diffusion step of the progress variable,
   * composite over the levels; everything downstream (curvature, threshold) then uses the smoothed field */
  int32_t do_smooth;
  double  smoothing_time;     /* smoothing_time (reference default 1e-7) */
  /* 0 or 3: the 3-D build.  2: the AMREX_SPACEDIM == 2 build on a level stored as ONE plane of cells (boxes with
   * lo[2] = hi[2] = 0, z a non-periodic direction: the homogeneous-Neumann wall makes every z difference an exact
   * zero, and adding exact zeros changes no sum of the path): MeanCurvature = sum_d d(n_d)/dx_d WITHOUT the 0.5 of
   * the 3-D build (curvature.cpp:542-546); pass-by-pass kernels (the fused sweep has the 0.5 built in); do_strain /
   * do_velnormal need a zero third velocity component at vel_comp + 2; do_gauss_curv and do_smooth are refused. */
  int32_t spacedim;
} pa_curv_params;
/* curvature.cpp:283-326 + 408-570 (core) + 575-789 (options).  state[lev][comp] = progress source
 * (ng>=2; with do_strain the velocity components get their ghost cells filled in place).
 * out[lev] comps: ocomp+0 Progress, +1 MeanCurvature, +2..4 FlameNormal, and when requested
 * +5 GaussianCurvature, +6 StrainRate, +7 VelFlameNormal, +8..16 ROST_dU?d? (row-major grad u), and with
 * do_smooth +17 SmoothedProgress.
 * Two implementations with identical results (bit for bit; both tested against the oracle): fused != 0, 3-D, boxes >= 3 cells thick
 * (one rank or a sharded hierarchy; with do_smooth the smoothed field is the pipeline's progress source with range [0, 1]): Progress / MeanCurvature / FlameNormal from the exact-normal pipeline of pa_gradcurv_run, whose
 * sweeps leave the cell-centred gradient of c (curvature.cpp:457-490, what do_gaussCurv differentiates again at :582-613) in a work
 * multifab of the level (3 components + 1 ghost layer, kept for the level's lifetime) instead of grad phi, then one pass per level
 * for the options; otherwise (or with params.fused = 0) one pass per AMReX call of the reference. */
int pa_curvature_run(pa_ctx*, int nlev, pa_mf* const* state, int comp, const int32_t bc[3],
                     const pa_curv_params*, pa_mf* const* out, int ocomp);
/* curvature.cpp:328-406: (I - dt Lap) sol = rhs[rcomp] as a composite solve over the levels (periodic /
 * homogeneous Neumann walls from bc, fine ghosts by applyBC, refluxed coarse-fine fluxes, covered coarse
 * cells = child averages), BiCGStab to ||b - A x||_inf <= tol ||b||_inf (the reference: 1e-12).  sol[lev]
 * comp scomp receives the solution on valid cells.  Refinement ratio 2.  Synchronous.
 * dt / dx^2 > 8 on the finest level (PA_SMOOTH_MG=1 / 0 in the environment: always / never; one rank or sharded): right-preconditioned with
 * one multigrid V(2,4) cycle per application (damped Jacobi on the AMR levels and on coarsened copies of level 0) -- the reference
 * solves with MLMG, whose iteration count does not grow with dt either; same solution to the tolerance.  The work vectors stay with
 * the levels (pa_level_destroy frees them).
 * Levels from pa_level_create_sharded: the solve is DISTRIBUTED over the context's ranks -- every rank iterates on its own boxes,
 * the restriction / ghost fills / flux register of one operator application cross ranks in nlev + 1 grouped exchanges
 * (pa_plan_restriction lists two of them), every dot product is one allreduce of the transport -- and returns the one-rank field
 * to the tolerance, not bit for bit; all ranks must call it together.  PA_SMOOTH_REPLICATED=1 in the environment: every rank
 * gathers the right-hand side of the whole hierarchy and runs the one-rank solve (its bits; no speed-up). */
int pa_smooth_solve(pa_ctx*, int nlev, pa_mf* const* rhs, int rcomp, pa_mf* const* sol, int scomp, double dt,
                    const int32_t bc[3], double tol, int maxiter, int* iters, double* rel_residual);
/* what the do_smooth solve of the last pa_curvature_run took: iterations and ||b - A x||_inf / ||b||_inf (the reference runs MLMG
 * with setVerbose(1), curvature.cpp:396-399, which prints its iteration count and residuals; the tool prints these).  Non-zero
 * when no such solve has run on the context. */
int pa_smooth_last(const pa_ctx*, int* iters, double* rel_residual);
/* which implementation the last pa_curvature_run on the context took: 1 = the exact-normal pipeline's G-output sweeps + one options
 * pass per level, 2 = the same with the Gaussian curvature formed inside the sweeps (one rank, every box wider than 32 cells and at
 * least 16 rows tall; only its first layer behind special faces is recomputed from the stored G), 0 = one kernel per AMReX call (fused = 0, 2-D levels, boxes thinner than 3 cells, or a hierarchy the
 * all-levels sweeps do not take; on a sharded hierarchy the ranks agree on one answer), -1 = none yet.  Diagnostic (tests, bench.py). */
int pa_curvature_last_path(const pa_ctx*);
/* Work multifabs the library keeps with a level between calls (pa_curvature_run: the gradient of c, 3 components, or the pass-by-pass
 * path's 8; pa_smooth_solve: its 8-13 vectors) -- kept because a multi-GB hipMalloc + hipFree per call cost more than the kernels they
 * served.  pa_level_destroy frees them; this frees them now (after a synchronisation of the context's stream) and returns the bytes
 * released.  The next call that needs them allocates again. */
int64_t pa_level_free_scratch(pa_level*);
/* fused grad+curvature of one variable: out[lev] comps ocomp+0..3 = gx,gy,gz,|g|,
 * +4..6 FlameNormal, +7 MeanCurvature.  work[lev]: scratch mf, 1 comp, ng=2. */
int pa_gradcurv_run(pa_ctx*, int nlev, pa_mf* const* state, int comp, const int32_t bc[3],
                    const pa_curv_params*, pa_mf* const* work, pa_mf* const* out, int ocomp);

/* Components comp0 .. comp0+ncomps-1 of state, one after the other into the SAME output buffers (what a tool does with the
 * variables of a plotfile).  done(user, comp) is called when a component's results are complete in `out` -- stream-ordered:
 * pa_sync before reading them on the host -- and before the next component overwrites them (NULL: no callback).  Ghost
 * fills that do not depend on results are done once for all components: FillBoundary of every component in one launch
 * and, on a sharded hierarchy, the first cross-rank exchange. */
int pa_gradcurv_run_comps(pa_ctx*, int nlev, pa_mf* const* state, int comp0, int ncomps, const int32_t bc[3],
                          const pa_curv_params*, pa_mf* const* work, pa_mf* const* out, int ocomp,
                          int (*done)(void* user, int comp), void* user);
/* The same in batches of nbatch components (1 .. 16): out[lev] holds nbatch SLOTS of 8 components from ocomp, component
 * comp0 + i of a batch goes to slot i % nbatch, and done(user, comp, ocomp_of_its_slot) is called for every component of a
 * batch once the batch is complete (stream-ordered, as above).  The boundary kernels of the exact-normal pipeline -- resolved
 * ghost values, coarse-patch gathers, curvature fix-up, and on a sharded hierarchy the exchange of the coarse normals --
 * then run once per BATCH instead of once per component; the sweeps stay component by component.  nbatch = 1 is
 * pa_gradcurv_run_comps.  (curvature.cpp:126-330 applied to each variable of a plotfile in turn.) */
int pa_gradcurv_run_comps2(pa_ctx*, int nlev, pa_mf* const* state, int comp0, int ncomps, const int32_t bc[3],
                           const pa_curv_params*, pa_mf* const* work, pa_mf* const* out, int ocomp, int nbatch,
                           int (*done)(void* user, int comp, int ocomp), void* user);

#ifdef __cplusplus
}
#endif
#endif
