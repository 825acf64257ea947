// pa_pipeline.hip -- the level loops of the tool mains (grad.cpp:158-236,
// curvature.cpp:283-326 + 408-570) on device-resident MultiFabs.  Host orchestration only;
// every numeric step is a kernel launch through the level-batched entry points.
#include "pa_internal.h"
#include "pa_dist.h"
#include <cstdlib>
#include <memory>

int pa_apply_bc_impl(pa_ctx* ctx, pa_mf* F, int comp, const pa_mf* C, int ccomp, const int32_t bc[3], int ratio, int only_dir,
                     int edges, const double* crse_xform);
int pa_apply_bc_dual(pa_ctx* ctx, pa_mf* F0, int comp0, pa_mf* F1, int comp1, const pa_mf* C, int ccomp, const int32_t bc[3], int ratio,
                     const double* xform);
int pa_fill_boundary_impl(pa_ctx* ctx, pa_mf* M, int comp, int ncomp, int ng, int no_exchange);
// exact-normal pipeline (pa_fused.hip)
bool pa_fused2_level_ok(const pa_level* L);
int pa_fill_boundary_local_batch(pa_ctx* ctx, int n, pa_mf* const* Ms, int comp, int ncomp, int ng);
int pa_grad_levels(pa_ctx* ctx, int nlev, pa_mf* const* phi, int comp, pa_mf* const* out, int ocomp);
int pa_gradcurv_prep_levels(pa_ctx* ctx, int nlev, pa_mf* const* phi, int comp, const pa_mf* const* crse, int ccomp, const int32_t bc[3], double pmin, double pmax, int phase = 3,
                            int nslots = 1, const double* prog = nullptr);
int pa_gradcurv_level_cg(pa_ctx* ctx, const pa_mf* phi, int pcomp, double pmin, double pmax, pa_mf* out, int ocomp, double thr = -1.0, int slot = 0);
bool pa_gradcurv_gout_ok(int nlev, pa_mf* const* phi);
int pa_gradcurv_levels_cg(pa_ctx* ctx, int nlev, pa_mf* const* phi, int pcomp, double pmin, double pmax, pa_mf* const* out, int ocomp, double thr = -1.0, int slot = 0,
                          int nslots = 1, const double* prog = nullptr, const double* pmins = nullptr, const double* pmaxs = nullptr, pa_mf* const* gout = nullptr, int part = 0, int kg = 0);
bool pa_gradcurv_parts_ok(int nlev, pa_mf* const* phi);
bool pa_gradcurv_kg_ok(int nlev, pa_mf* const* phi);
int pa_gauss_cells_levels(pa_ctx* ctx, int nlev, pa_mf* const* G, pa_mf* const* out, int pc, int kgc, double thr);
int pa_gradcurv_fix_levels(pa_ctx* ctx, int nlev, pa_mf* const* phi, int pcomp, const pa_mf* const* crse_n, int cncomp0, const int32_t bc[3], double pmin, double pmax,
                           pa_mf* const* out, int ncomp0, int kcomp, double thr = -1.0, int nslots = 1, const double* prog = nullptr, int cn_z = 8,
                           const pa_mf* const* crse_phi = nullptr, int cpcomp = 0);
int pa_gauss_curv_level(pa_ctx* ctx, const pa_mf* G, int gcomp, const pa_mf* normgrad, int ngcomp, const pa_mf* c, int ccomp, double thr, pa_mf* out,
                        int kcomp);
int pa_strain_level(pa_ctx* ctx, const pa_mf* u, int ucomp, pa_mf* out, int srcomp, int rostcomp);
int pa_velnormal_level(pa_ctx* ctx, const pa_mf* u, int ucomp, const pa_mf* n, int ncomp0, const pa_mf* c, int ccomp, double thr, pa_mf* out, int ocomp);
int pa_curvopts_level(pa_ctx* ctx, int which, const pa_mf* G, const pa_mf* u, int ucomp, pa_mf* out, int pc, int nc, int kgc, int src, int vnc, int rostc, double thr);

int pa_gradcurv_faces_phase(pa_ctx* ctx, const pa_mf* c, int ccomp, const pa_mf* crse_n, int cncomp0, const int32_t bc[3], int ratio, double thr,
                            pa_mf* out, int ncomp0, int kcomp, int phase);

#define PA_TRY(x)        \
  do {                   \
    if ((x) != 0) return 1; \
  } while (0)

static int check_levels(pa_ctx* ctx, int nlev, pa_mf* const* a, const char* who) {
  if (!ctx) return 1;
  if (nlev <= 0 || !a) return pa_fail(ctx, std::string(who) + ": no levels");
  for (int l = 0; l < nlev; ++l) {
    if (!a[l]) return pa_fail(ctx, std::string(who) + ": null multifab at level " + std::to_string(l));
    if (a[l]->lev->nranks != a[0]->lev->nranks) return pa_fail(ctx, std::string(who) + ": levels sharded over different numbers of ranks");
    if (a[l]->lev->nranks > 1 && a[l]->lev->nranks != ctx->comm.nranks) return pa_fail(ctx, std::string(who) + ": the context's transport has a different number of ranks than the levels");
    // these pipelines hard-code refinement ratio 2 like the reference (curvature.cpp:445, grad.cpp:255); a ratio-4 hierarchy (the
    // python reader and filterPlt / isosurface accept them) would get wrong coarse-fine ghost cells without any other sign
    if (l > 0)
      for (int d = 0; d < 3; ++d) {
        const long long nc = (long long)a[l - 1]->lev->domhi[d] - a[l - 1]->lev->domlo[d] + 1, nf = (long long)a[l]->lev->domhi[d] - a[l]->lev->domlo[d] + 1;
        const bool flat = nc == 1 && nf == 1;  // a 2-D hierarchy stored as the plane k = 0
        if (!flat && (nf != 2 * nc || a[l]->lev->domlo[d] != 2 * a[l - 1]->lev->domlo[d]))
          return pa_fail(ctx, std::string(who) + ": level " + std::to_string(l) + " is not a factor-2 refinement of level " + std::to_string(l - 1) + " (only refinement ratio 2 is supported here)");
      }
  }
  return 0;
}

struct StreamSwap {  // the level entry points launch on ctx->stream
  pa_ctx* c;
  hipStream_t keep;
  StreamSwap(pa_ctx* ctx, hipStream_t s) : c(ctx), keep(ctx->stream) { ctx->stream = s; }
  ~StreamSwap() { c->stream = keep; }
};

extern "C" int pa_grad_run(pa_ctx* ctx, int nlev, pa_mf* const* state, int comp, const int32_t bc[3], pa_mf* const* out, int ocomp) {
  PaBind bind_(ctx);
  bool bc_done = false;
  PA_TRY(check_levels(ctx, nlev, state, "pa_grad_run"));
  PA_TRY(check_levels(ctx, nlev, out, "pa_grad_run"));
  // grad.cpp:169 FillBoundary on every level, then MLMG getFluxes level by level (applyBC + flux)
  if (state[0]->lev->nranks > 1) {
    for (int l = 0; l < nlev; ++l) PA_TRY(pa_fill_boundary(ctx, state[l], comp, 1, 1));  // + the cross-rank half, level by level
  } else {
    for (int l = 0; l < nlev; ++l)
      if (state[l]->ng < 1 || comp < 0 || comp >= state[l]->ncomp) return pa_fail(ctx, "pa_grad_run: state needs >= 1 ghost layer and the component");
    // applyBC of every level on the side stream NEXT TO FillBoundary: it reads valid cells and coarse valid cells and writes
    // the ghost cells behind special faces that are NOT valid cells of the level, FillBoundary writes those that are -- disjoint
    // (as k_prep_faces in the fused pass)
    {
      if (!ctx->stream2) PA_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
      while (ctx->sync_evs.size() < 2) {
        hipEvent_t e;
        PA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->sync_evs.push_back(e);
      }
      PA_HIP(hipEventRecord(ctx->sync_evs[0], ctx->stream));  // after the inputs / the previous call
      PA_HIP(hipStreamWaitEvent(ctx->stream2, ctx->sync_evs[0], 0));
      {
        StreamSwap sw(ctx, ctx->stream2);
        // the face ghosts of every level in ONE launch on the per-face work tables, coarse values from the gathered coarse patches
        // (k_prep_faces_chunks<.., PHIONLY>, pa_fused.hip: 0.43 -> ~0.2 ms on the headline hierarchy);
        // 2-D hierarchies (one plane of cells: the patches are a 3-D layout): applyBC level by level
        bool patch = true;
        for (int l = 0; l < nlev; ++l) patch = patch && state[l]->lev->domhi[2] > state[l]->lev->domlo[2];
        if (patch) {
          std::vector<const pa_mf*> crse((size_t)nlev, nullptr);
          for (int l = 1; l < nlev; ++l) crse[(size_t)l] = state[l - 1];
          PA_TRY(pa_gradcurv_prep_levels(ctx, nlev, state, comp, crse.data(), comp, bc, 0.0, 1.0, 1 | 8, 1, nullptr));
        } else {
          for (int l = 0; l < nlev; ++l) PA_TRY(pa_apply_bc(ctx, state[l], comp, l > 0 ? state[l - 1] : nullptr, comp, bc, 2, -1));
        }
      }
      PA_HIP(hipEventRecord(ctx->sync_evs[1], ctx->stream2));
      bc_done = true;
    }
    {
      ProfScope prof(ctx, PA_TAG_FILL);
      PA_TRY(pa_fill_boundary_local_batch(ctx, nlev, state, comp, 1, 1));  // every level in one launch
    }
    if (bc_done) PA_HIP(hipStreamWaitEvent(ctx->stream, ctx->sync_evs[1], 0));
  }
  if (bc_done) {  // every level's ghost cells are final: the sweeps of all levels in one launch where the levels allow it
    for (int l = 0; l < nlev; ++l) {
      if (state[l]->lev != out[l]->lev) return pa_fail(ctx, "pa_grad_run: state and out live on different levels");
      if (ocomp < 0 || ocomp + 4 > out[l]->ncomp) return pa_fail(ctx, "pa_grad_run: component range");
    }
    const int rc = pa_grad_levels(ctx, nlev, state, comp, out, ocomp);
    if (rc >= 0) return rc;
  }
  for (int l = 0; l < nlev; ++l) {
    if (!bc_done) PA_TRY(pa_apply_bc(ctx, state[l], comp, l > 0 ? state[l - 1] : nullptr, comp, bc, 2, -1));
    PA_TRY(pa_grad_level(ctx, state[l], comp, out[l], ocomp));
  }
  return 0;
}

struct MFDel { void operator()(pa_mf* m) const { pa_mf_destroy(m); } };
using MFPtr = std::unique_ptr<pa_mf, MFDel>;

static int prog_minmax(pa_ctx* ctx, int nlev, pa_mf* const* state, int comp, const pa_curv_params* P, double& pmin, double& pmax) {
  pmin = P->prog_min;
  pmax = P->prog_max;
  if (pmin > pmax) {  // useFileMinMax (curvature.cpp:139-149): min/max over the levels used
    pmin = 1e300; pmax = -1e300;
    for (int l = 0; l < nlev; ++l) {
      double a, b;
      PA_TRY(pa_minmax_level(ctx, state[l], comp, &a, &b));
      pmin = a < pmin ? a : pmin;
      pmax = b > pmax ? b : pmax;
    }
    // ParallelDescriptor::ReduceRealMin / ReduceRealMax (curvature.cpp:147-148)
    PA_TRY(pa_allreduce(ctx, &pmin, 1, 0));
    PA_TRY(pa_allreduce(ctx, &pmax, 1, 1));
  }
  if (!(pmax > pmin)) return pa_fail(ctx, "progress variable has no range (progMax <= progMin)");
  return 0;
}

// pass-by-pass curvature core; out comps: pc (Progress, -1 = skip), kc (MeanCurvature), nc (FlameNormal x3)
// opt >= 0: also the options of curvature.cpp:575-789 selected in P, written at out comps opt+5.. (see the header)
static int curvature_passes(pa_ctx* ctx, int nlev, pa_mf* const* state, int comp, const int32_t bc[3], double pmin, double pmax,
                            double thr, pa_mf* const* out, int pc, int kc, int nc, const pa_curv_params* P = nullptr, int opt = -1,
                            double kscale = 0.5 /* curvature.cpp:542-546: 0.5 in the 3-D build, none (1.0, exact) in the 2-D build */) {
  // work multifabs of the level, kept for its lifetime (pa_level_scratch; until round 5 a hipMalloc + hipFree of up to 8 components
  // per level and call -- a call that did not get the previous call's blocks back took 0.8 s instead of 50 ms at the headline size);
  // every cell that is read is written first in each call
  std::vector<pa_mf*> cmf(nlev, nullptr), nmf(nlev, nullptr), gmf(nlev, nullptr), ngmf(nlev, nullptr);
  const bool gauss = opt >= 0 && P && P->do_gauss_curv;
  const bool strain = opt >= 0 && P && P->do_strain;
  const bool veln = opt >= 0 && P && P->do_velnormal;
  for (int l = 0; l < nlev && opt >= 0; ++l) {
    const int need = opt + (strain && P->get_strain_tensor ? 17 : (veln ? 8 : (strain ? 7 : (gauss ? 6 : 5))));
    if (out[l]->ncomp < need) return pa_fail(ctx, "pa_curvature_run: out needs " + std::to_string(need) + " components for the requested options");
    if ((strain || veln) && (P->vel_comp < 0 || P->vel_comp + 3 > state[l]->ncomp)) return pa_fail(ctx, "pa_curvature_run: vel_comp out of range");
  }
  for (int l = 0; l < nlev; ++l) {
    cmf[l] = pa_level_scratch(ctx, state[l]->lev, 1, 1);
    nmf[l] = pa_level_scratch(ctx, state[l]->lev, 3, 1, 1);
    if (!cmf[l] || !nmf[l]) return 1;
    PA_TRY(pa_progress_level(ctx, state[l], comp, pmin, pmax, cmf[l], 0, 0));  // curvature.cpp:316-320
    if (pc >= 0) PA_TRY(pa_mf_copy(ctx, cmf[l], 0, out[l], pc, 1, 0));         // Progress is the unsmoothed field
    PA_TRY(pa_fill_boundary(ctx, cmf[l], 0, 1, 1));                            // :322
  }
  if (opt >= 0 && P && P->do_smooth) {  // :328-406; idprogvar = idSmProg from here on (:408)
    std::vector<pa_mf*> cs(nlev);
    for (int l = 0; l < nlev; ++l) {
      cs[l] = cmf[l];
      if (out[l]->ncomp < opt + 18) return pa_fail(ctx, "pa_curvature_run: out needs " + std::to_string(opt + 18) + " components with do_smooth");
    }
    int iters = 0;
    double res = 0.0;
    // the smoothing operator is Periodic / Neumann only, whatever sym_dir says (curvature.cpp:348-357)
    const int32_t bc_s[3] = {bc[0] == PA_BC_PERIODIC ? PA_BC_PERIODIC : PA_BC_NEUMANN, bc[1] == PA_BC_PERIODIC ? PA_BC_PERIODIC : PA_BC_NEUMANN,
                             bc[2] == PA_BC_PERIODIC ? PA_BC_PERIODIC : PA_BC_NEUMANN};
    // The reference stops MLMG at a relative residual of 1e-12 (curvature.cpp:392-399), which fixes its smoothed field only
    // to ~cond(A) * 1e-12.  To stay within 1e-12 of ANY solution that meets that criterion the product iterates on to 1e-14
    // and accepts whatever it reached if that is at least the reference's 1e-12.  Iteration cap: the reference's setMaxIter(100)
    // counts multigrid V-cycles, whose convergence does not depend on the condition number; one unpreconditioned BiCGStab
    // iteration is worth far less -- measured at the headline size (bench.py secondary.f1_do_smooth_headline): 17 iterations
    // for the default smoothing_time 1e-7 (dt / dx^2 = 0.4 on the finest level), but 100 iterations reach only 2e-8 for 1e-5
    // (dt / dx^2 = 42) -- so the cap here is 2000 Krylov iterations; a solve that still stalls fails as MLMG's would.
    const int src = pa_smooth_solve(ctx, nlev, cs.data(), 0, cs.data(), 0, P->smoothing_time, bc_s, 1e-14, 2000, &iters, &res);
    ctx->smooth_iters = iters;
    ctx->smooth_res = res;
    if (src != 0 && !(iters > 0 && res <= 1e-12)) return 1;
    for (int l = 0; l < nlev; ++l) {
      PA_TRY(pa_mf_copy(ctx, cmf[l], 0, out[l], opt + 17, 1, 0));
      PA_TRY(pa_fill_boundary(ctx, cmf[l], 0, 1, 1));  // :403
    }
  }
  for (int l = 0; l < nlev; ++l) {
    pa_mf* c = cmf[l];
    pa_mf* n = nmf[l];
    PA_TRY(pa_apply_bc(ctx, c, 0, l > 0 ? cmf[l - 1] : nullptr, 0, bc, 2, -1));  // :426-457 (inside getFluxes)
    if (gauss) {
      gmf[l] = pa_level_scratch(ctx, state[l]->lev, 3, 1);  // the role the fast path's G plays
      ngmf[l] = pa_level_scratch(ctx, state[l]->lev, 1, 0);
      if (!gmf[l] || !ngmf[l]) return 1;
    }
    PA_TRY(pa_normal_level(ctx, c, 0, gmf[l], 0, ngmf[l], 0, n, 0));         // :457-500
    PA_TRY(pa_fill_boundary(ctx, n, 0, 3, 1));                                          // :502
    for (int d = 0; d < 3; ++d)                                                         // :508-531
      PA_TRY(pa_apply_bc(ctx, n, d, l > 0 ? out[l - 1] : nullptr, nc + d, bc, 2, d));
    PA_TRY(pa_div_level(ctx, n, 0, kscale, c, 0, thr, out[l], kc));                        // :533-567
    PA_TRY(pa_mf_copy(ctx, n, 0, out[l], nc, 3, 0));                                    // :569-570
    if (gauss) {  // :575-677: Hessian of c from cell_normal (= G), coarse-fine BC from cell_normal[lev-1]
      pa_mf* G = gmf[l];
      PA_TRY(pa_fill_boundary(ctx, G, 0, 3, 1));                                        // :488
      for (int d = 0; d < 3; ++d) PA_TRY(pa_apply_bc(ctx, G, d, l > 0 ? gmf[l - 1] : nullptr, d, bc, 2, -1));
      PA_TRY(pa_gauss_curv_level(ctx, G, 0, ngmf[l], 0, c, 0, thr, out[l], opt + 5));
    }
    if (strain) {  // :679-757: grad u through the same ghost-resolved central differences
      PA_TRY(pa_fill_boundary(ctx, state[l], P->vel_comp, 3, 1));
      for (int d = 0; d < 3; ++d) PA_TRY(pa_apply_bc(ctx, state[l], P->vel_comp + d, l > 0 ? state[l - 1] : nullptr, P->vel_comp + d, bc, 2, -1));
      PA_TRY(pa_strain_level(ctx, state[l], P->vel_comp, out[l], opt + 6, P->get_strain_tensor ? opt + 8 : -1));
    }
    if (veln) PA_TRY(pa_velnormal_level(ctx, state[l], P->vel_comp, n, 0, c, 0, thr, out[l], opt + 7));  // :765-787
  }
  return 0;
}

// cs: coarse data for the coarse-fine ghost cells of level l -- the coarser level's state, or (sharded hierarchy) this
// rank's coarse-source copy of its component `comp` with the cross-rank half of FillBoundary already done by the caller
static int fused_pre(pa_ctx* ctx, int l, pa_mf* const* state, int comp, const int32_t bc[3], double pmin, double pmax, pa_mf* const* work,
                     const pa_mf* cs_dist = nullptr, bool dist = false) {
  // The stored progress variable (work) is only needed near the special faces (face fix-up).  The
  // coarse-fine boundary values of c are interpolated from the coarse PHI through the affine view
  // (v - pmin) * invdenom, which is bit-identical to a stored coarse c; they read the coarse phi's
  // VALID cells only, so the order relative to applyBC on the coarse phi does not matter.
  const double xf[2] = {pmin, 1.0 / (pmax - pmin)};
  const pa_mf* cs = dist ? cs_dist : (l > 0 ? state[l - 1] : nullptr);
  const int ccomp = dist ? 0 : comp;
  PA_TRY(pa_fill_boundary_impl(ctx, state[l], comp, 1, 2, dist ? 1 : 0));
  if (state[l]->lev->boxes.empty()) return 0;
  PA_TRY(pa_progress_shell_level(ctx, state[l], comp, pmin, pmax, work[l], 0, 2, 4));
  PA_TRY(pa_apply_bc_dual(ctx, state[l], comp, work[l], 0, cs, ccomp, bc, 2, xf));  // face ghosts of phi and of c, one launch
  PA_TRY(pa_apply_bc_impl(ctx, work[l], 0, cs, ccomp, bc, 2, -1, 1, xf));           // edge ghosts of c
  return 0;
}

// The fused pipeline on a hierarchy sharded over ranks.  Everything a level needs from other ranks depends on VALID cells
// only, so the cross-rank traffic of a whole pass collapses into TWO grouped exchanges:
//   A  ghost cells of phi on every level (FillBoundary, 2 layers) + the coarse phi under every fine level's coarse-fine
//      faces (what setCoarseFineBC / the MLMG boundary registers copy, curvature.cpp:443-445, grad.cpp:212)
//   B  the coarse flame normal under those faces, once the normals of the coarser level are final (curvature.cpp:514-518)
// with the ghost preparation + sweeps (+ layer-1 normals in the first pipeline) between them and the face curvature after
// B.  In the exact-normal pipeline B is split per level and overlapped with the sweeps (nlev exchanges, two exposed).
// gout (exact only; pa_curvature_run's fast path): the G-output sweeps -- [Progress K Nx Ny Nz] at ocomp .. ocomp + 4, G in gout[l]
static int fused_passes_dist(pa_ctx* ctx, int nlev, pa_mf* const* state, int comp, const int32_t bc[3], double pmin, double pmax, double thr,
                             pa_mf* const* work, pa_mf* const* out, int ocomp, bool exact, pa_mf* const* gout = nullptr) {
  const int ncomp0 = gout ? ocomp + 2 : ocomp + 4, kcomp = gout ? ocomp + 1 : ocomp + 7;
  if (gout && !exact) return pa_fail(ctx, "fused_passes_dist: the G-output sweeps belong to the exact-normal pipeline");
  std::vector<XJob> jobs;
  std::vector<CsPlan*> cs(nlev, nullptr);
  std::vector<pa_mf*> csphi(nlev, nullptr), csn(nlev, nullptr);
  for (int l = 0; l < nlev; ++l) {
    XPlan* P = pa_fb_plan(ctx, state[l]->lev, 2);
    if (!P) return 1;
    jobs.push_back({P, state[l], comp, state[l], comp, 1});
  }
  for (int l = 1; l < nlev; ++l) {
    cs[l] = pa_cs_plan(ctx, state[l]->lev, state[l - 1]->lev, 0, 0, 0);
    if (!cs[l]) return 1;
    csphi[l] = cs[l]->mf(ctx, 1);
    csn[l] = cs[l]->mf(ctx, 3, 1);
    if (cs[l]->cs && (!csphi[l] || !csn[l])) return 1;
    jobs.push_back({&cs[l]->x, state[l - 1], comp, csphi[l], 0, 1});
  }
  if (exact) {
    // Exact-normal pipeline of a sharded hierarchy.  Two grouped exchanges: A (ghost cells of phi + coarse phi under coarse-fine faces)
    // and B (the coarse normals, after the sweeps).  The chain the sweep waits for -- exchange A, patch gather + faces, ring -- is on
    // the main stream, the LOCAL FillBoundary next to it on the side stream (the other way round cost a stream hand-over of ~26 us on
    // the critical path, round 3); the sweeps of all levels are ONE launch (per-XCD queues of tile chunks keep it balanced: two
    // launches pay two tails, round 5), exchange B after it.
    //   stream A: [exchange A][patches + faces]  wait(C) [ring][sweep: late tiles / all] wait(C) [exchange B][fix-up]
    //   stream C: [FillBoundary local][sweep: early tiles (PA_DIST_EARLY=1)]
    // The variants this replaced (exchange on the side stream, a launch per level with its exchange under the next sweep, the finest
    // level's sweep apart) and their measurements: DESIGN_HISTORY.md R2-R5, profiles/r04_sim8_delay.txt, r05_sim8_delay.txt.
    std::vector<const pa_mf*> crse(csphi.begin(), csphi.end()), crse_n(csn.begin(), csn.end());
    hipStream_t A = ctx->stream;
    if (!ctx->stream2) PA_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    hipStream_t C = ctx->stream2;
    while (ctx->sync_evs.size() < 3) {
      hipEvent_t e;
      PA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ctx->sync_evs.push_back(e);
    }
    PA_HIP(hipEventRecord(ctx->sync_evs[0], A));  // the side stream starts after everything already queued (inputs, the previous pass)
    PA_HIP(hipStreamWaitEvent(C, ctx->sync_evs[0], 0));
    {
      ProfScope prof(ctx, PA_TAG_XCHG);
      PA_TRY(pa_xexchange(ctx, (int)jobs.size(), jobs.data()));
    }
    // the faces' half of the ghost preparation (patch gather + faces) reads valid cells and the coarse data that just arrived, and
    // writes only ghost cells behind special faces: it runs next to the local FillBoundary
    PA_TRY(pa_gradcurv_prep_levels(ctx, nlev, state, comp, crse.data(), 0, bc, pmin, pmax, 1));
    {
      StreamSwap sw(ctx, C);
      ProfScope prof(ctx, PA_TAG_FILL);
      PA_TRY(pa_fill_boundary_local_batch(ctx, nlev, state, comp, 1, 2));
    }
    PA_HIP(hipEventRecord(ctx->sync_evs[1], C));
    // EARLY TILES (PA_DIST_EARLY=1; OFF by default).  A tile of the sweep whose read region touches only valid cells of this rank's
    // own boxes has its input complete once the LOCAL FillBoundary is done; those tiles (pa_sweep_wgtab part 1: 30 % of an 8-way share
    // of the headline) are swept on the side stream right behind it -- under exchange A, the patch gather, the face and ring
    // preparation -- and only the tiles next to another rank's box or to a special face wait for that chain.  Bit-identical
    // (tests/test_gpu_dist.py runs both), and measured with the delay-model transport on rank 0's share of 8
    // (profiles/r06_sim8_delay.txt): 1.108 -> 1.100 ms per pass with 68-us exchanges, 1.024 -> 1.033 with 18 us + bytes / 153 GB/s,
    // 0.955 -> 0.965 with no-op exchanges.  Why it hides so little: an eighth of the headline is 480 tiles on 256 CUs, one workgroup
    // per CU -- two rounds of ~350 us as ONE launch, but 150 early + 330 late tiles are one round + two rounds.  It pays only where
    // an exchange costs more than a round is long; on a hierarchy with more work per rank the rounds stop mattering.
    const bool early = pa_opt().dist_early && pa_gradcurv_parts_ok(nlev, state);
    if (early) {
      {
        StreamSwap sw(ctx, C);
        PA_TRY(pa_gradcurv_levels_cg(ctx, nlev, state, comp, pmin, pmax, out, ocomp, thr, 0, 1, nullptr, nullptr, nullptr, gout, 1));
      }
      PA_HIP(hipEventRecord(ctx->sync_evs[2], C));
    }
    PA_HIP(hipStreamWaitEvent(A, ctx->sync_evs[1], 0));
    PA_TRY(pa_gradcurv_prep_levels(ctx, nlev, state, comp, crse.data(), 0, bc, pmin, pmax, 2));
    PA_TRY(pa_gradcurv_levels_cg(ctx, nlev, state, comp, pmin, pmax, out, ocomp, thr, 0, 1, nullptr, nullptr, nullptr, gout, early ? 2 : 0));
    if (early) PA_HIP(hipStreamWaitEvent(A, ctx->sync_evs[2], 0));  // the early tiles' normals go into exchange B and the fix-up too
    {
      std::vector<XJob> nj;
      for (int l = 1; l < nlev; ++l) nj.push_back({&cs[l]->x, out[l - 1], ncomp0, csn[l], 0, 3});
      ProfScope prof(ctx, PA_TAG_XCHG);
      PA_TRY(pa_xexchange(ctx, (int)nj.size(), nj.data()));
    }
    PA_TRY(pa_gradcurv_fix_levels(ctx, nlev, state, comp, crse_n.data(), 0, bc, pmin, pmax, out, ncomp0, kcomp, thr, 1, nullptr, 8, crse.data(), 0));
    return 0;
  }
  {
    ProfScope prof(ctx, PA_TAG_XCHG);
    PA_TRY(pa_xexchange(ctx, (int)jobs.size(), jobs.data()));
  }
  for (int l = 0; l < nlev; ++l) PA_TRY(fused_pre(ctx, l, state, comp, bc, pmin, pmax, work, csphi[l], true));
  for (int l = 0; l < nlev; ++l) {
    if (state[l]->lev->boxes.empty()) continue;
    PA_TRY(pa_gradcurv_level(ctx, state[l], comp, pmin, pmax, thr, out[l], ocomp));
    PA_TRY(pa_gradcurv_faces_phase(ctx, work[l], 0, nullptr, 0, bc, 2, thr, out[l], ocomp + 4, ocomp + 7, 1));
  }
  jobs.clear();
  for (int l = 1; l < nlev; ++l) jobs.push_back({&cs[l]->x, out[l - 1], ocomp + 4, csn[l], 0, 3});
  {
    ProfScope prof(ctx, PA_TAG_XCHG);
    PA_TRY(pa_xexchange(ctx, (int)jobs.size(), jobs.data()));
  }
  for (int l = 0; l < nlev; ++l) {
    if (state[l]->lev->boxes.empty()) continue;
    PA_TRY(pa_gradcurv_faces_phase(ctx, work[l], 0, l > 0 ? csn[l] : nullptr, 0, bc, 2, thr, out[l], ocomp + 4, ocomp + 7, 2));
  }
  return 0;
}

// Two-stream schedule of the fused pipeline.  The sweep leaves the CU's vector units, a third of its registers
// and 69 KB of LDS idle (it is bound by the store path), and the boundary kernels are latency-
// bound on little data, so the ghost-cell preparation of the NEXT levels and the face fix-up of the PREVIOUS
// level run on a second stream next to the sweep:
//   stream A: prep(0) sweep(0)          sweep(1)            sweep(2)        [wait faces(2)]
//   stream B: prep(1) prep(2) .. faces(0)  ..     faces(1)   ..       faces(2)
// prep(l) writes only ghost cells of phi_l and the shell copy of c_l and reads VALID cells of phi_{l-1};
// sweep(l) needs prep(l); faces(l) needs sweep(l), the shell of c_l and the final normals of level l-1.
// MEASURED (MI355X, headline workload, twice): 8.53 / 8.61 ms per step with the overlap against 8.43 / 8.69 ms
// without -- the sweep slows from 2.23-2.31 to 2.57-2.60 ms per launch and gives back everything the second
// stream hides: both sides wait on the same memory path.  Kept for A/B (PA_OVERLAP=1), OFF by default.
static int overlap_on() {
  constexpr int v = 0;
  return v;
}

static int fused_faces(pa_ctx* ctx, int l, const int32_t bc[3], double thr, pa_mf* const* work, pa_mf* const* out, int ocomp, int phase = 3) {
  return pa_gradcurv_faces_phase(ctx, work[l], 0, l > 0 ? out[l - 1] : nullptr, ocomp + 4, bc, 2, thr, out[l], ocomp + 4, ocomp + 7, phase);
}

// Level-concurrent boundary schedule (PA_CONC=1 / 0; by default on for small levels only, see conc_on).  Between levels almost nothing depends on
// anything: prep(l) writes ghost cells of phi_l / the shell of c_l and reads VALID cells of phi_{l-1}; the layer-1
// normals of level l need only sweep(l) and that shell; the face curvature of level l needs the layer-1 normals of
// levels l and l-1.  So the three groups can run with one stream per level and only the sweeps -- which want the
// memory path for themselves (PA_OVERLAP above) -- one after the other with nothing beside:
//   [prep(0) | prep(1) | prep(2)]  sweep(0) sweep(1) sweep(2)  [normal(0) | normal(1) | normal(2)]  [curv(0) | curv(1) | curv(2)]
// MEASURED (MI355X, headline workload, twice): 7.87 / 7.92 ms per step against 7.81 / 7.88 ms sequential.  The
// kernel trace shows the kernels DO overlap (two hardware queues) and every kernel that runs beside another takes
// about twice as long (k_fill_boundary 188 + 192 us side by side against 89 us alone): the boundary kernels are
// not waiting on latency, they are bound by the memory system's rate for short scattered segments (2-cell-wide
// ghost strips), which two streams share.  Concurrency buys nothing; kept for A/B only.
// PA_CONC unset: on for hierarchies of SMALL levels only.  There the boundary launches are short enough not to saturate
// the memory system and running the levels side by side does pay: 4 levels of 256^3 in 64^3 boxes 16.6-17.0 -> 16.2 ms
// per 8-component step; on the 512^3 headline levels it costs 1 % (above).
static int conc_on(int nlev, pa_mf* const* state) {
  constexpr int v = -1;
  if (v >= 0) return v;
  long long big = 0;
  for (int l = 0; l < nlev; ++l) {
    long long n = 0;
    for (const DBox& B : state[l]->lev->boxes) n += (long long)(B.hi[0] - B.lo[0] + 1) * (B.hi[1] - B.lo[1] + 1) * (B.hi[2] - B.lo[2] + 1);
    big = std::max(big, n);
  }
  return big <= 40000000LL;
}
static int fused_passes_conc(pa_ctx* ctx, int nlev, pa_mf* const* state, int comp, const int32_t bc[3], double pmin, double pmax, double thr,
                             pa_mf* const* work, pa_mf* const* out, int ocomp) {
  while ((int)ctx->lev_streams.size() < nlev) {
    hipStream_t s;
    PA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    ctx->lev_streams.push_back(s);
  }
  const size_t nev = 2 + 3 * (size_t)nlev;
  while (ctx->sync_evs.size() < nev) {
    hipEvent_t e;
    PA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->sync_evs.push_back(e);
  }
  hipStream_t A = ctx->stream;
  hipEvent_t e_start = ctx->sync_evs[0], e_sweeps = ctx->sync_evs[1];
  auto e_prep = [&](int l) { return ctx->sync_evs[2 + l]; };
  auto e_norm = [&](int l) { return ctx->sync_evs[2 + nlev + l]; };
  auto e_curv = [&](int l) { return ctx->sync_evs[2 + 2 * nlev + l]; };
  PA_HIP(hipEventRecord(e_start, A));
  for (int l = 0; l < nlev; ++l) {  // ghost-cell preparation, one stream per level
    hipStream_t T = ctx->lev_streams[l];
    PA_HIP(hipStreamWaitEvent(T, e_start, 0));
    StreamSwap sw(ctx, T);
    PA_TRY(fused_pre(ctx, l, state, comp, bc, pmin, pmax, work));
    PA_HIP(hipEventRecord(e_prep(l), T));
  }
  for (int l = 0; l < nlev; ++l) PA_HIP(hipStreamWaitEvent(A, e_prep(l), 0));  // nothing runs beside a sweep
  for (int l = 0; l < nlev; ++l) PA_TRY(pa_gradcurv_level(ctx, state[l], comp, pmin, pmax, thr, out[l], ocomp));
  PA_HIP(hipEventRecord(e_sweeps, A));
  for (int l = 0; l < nlev; ++l) {  // layer-1 normals
    hipStream_t T = ctx->lev_streams[l];
    PA_HIP(hipStreamWaitEvent(T, e_sweeps, 0));
    StreamSwap sw(ctx, T);
    PA_TRY(fused_faces(ctx, l, bc, thr, work, out, ocomp, 1));
    PA_HIP(hipEventRecord(e_norm(l), T));
  }
  for (int l = 0; l < nlev; ++l) {  // face curvature: normals of this level (same stream) and of the coarser one
    hipStream_t T = ctx->lev_streams[l];
    if (l > 0) PA_HIP(hipStreamWaitEvent(T, e_norm(l - 1), 0));
    StreamSwap sw(ctx, T);
    PA_TRY(fused_faces(ctx, l, bc, thr, work, out, ocomp, 2));
    PA_HIP(hipEventRecord(e_curv(l), T));
  }
  for (int l = 0; l < nlev; ++l) PA_HIP(hipStreamWaitEvent(A, e_curv(l), 0));  // later work on the caller's stream sees the results
  return 0;
}

static bool exact_ok(int nlev, pa_mf* const* state, double thr);
// The exact-normal pipeline on one rank.  gout == null: [gx gy gz |g| Nx Ny Nz K] at out components ocomp .. ocomp + 7;
// gout != null (pa_curvature_run with options): [Progress K Nx Ny Nz] at ocomp .. ocomp + 4 and G, the cell-centred gradient of c,
// at components 0 .. 2 of gout[l] (the GOUT sweeps, pa_fused_march3.h)
static int exact_passes(pa_ctx* ctx, int nlev, pa_mf* const* state, int comp, const int32_t bc[3], double pmin, double pmax, double thr,
                        pa_mf* const* out, int ocomp, pa_mf* const* gout = nullptr, int kg = 0) {
  const int ncomp0 = gout ? ocomp + 2 : ocomp + 4, kcomp = gout ? ocomp + 1 : ocomp + 7;
  {
    // 3 + nlev launches per pass: ghost cells of every level (one launch), resolved ghost c (two), the sweeps, the curvature
    // of the first layer behind the special faces of every level (two).  (Running the boundary launches on a side stream
    // next to the sweeps was measured again with this lighter pipeline: 7.04-7.09 against 6.93-7.03 ms per step, the
    // sweeps slowing from 1.94 to 2.15 ms -- they share the memory path -- so everything stays on one stream.)
    std::vector<const pa_mf*> crse(nlev, nullptr), crse_n(nlev, nullptr);
    for (int l = 1; l < nlev; ++l) { crse[l] = state[l - 1]; crse_n[l] = out[l - 1]; }
    // k_prep_faces (latency bound: dependent lookups, 1.8 TB/s) runs on the side stream NEXT TO FillBoundary (bandwidth
    // bound): it reads valid cells and coarse data and writes the ghost cells of special faces + the compact arrays, FillBoundary
    // writes the ghost cells that are valid cells elsewhere -- disjoint on pure faces, which this pipeline requires.
    // Measured with tools/ab_driver.py (alternating blocks in one process): 6.588 against 6.642 ms per pass on one stream.  The same tool on the two other overlaps that look plausible: the perimeter fix-up kernel next to
    // the interior one +0.009 ms (nothing), the fix-up of level l under the sweep of level l + 1 -0.21 ms (the sweep pays
    // more than the fix-up hides) -- neither is kept.
    constexpr int pov = 1;
    constexpr bool ring_side = true;  // the ring with the faces: it reads its neighbours' valid cells in place
    // (Round 6, measured and not kept: level 0 swept on its own as soon as ITS ghost cells are ready, the ghost preparation of the finer
    // levels on the side stream under it -- 6.27 against 6.11 ms per pass on the irregular hierarchy, 5.845 against 5.827 on the
    // headline, tools/ab_driver.py: as in round 4, whatever runs beside a sweep costs the sweep more than it hides.)
    if (pov) {
      if (!ctx->stream2) PA_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
      while (ctx->sync_evs.size() < 2) {
        hipEvent_t e;
        PA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->sync_evs.push_back(e);
      }
      PA_HIP(hipEventRecord(ctx->sync_evs[0], ctx->stream));  // after everything already queued (the inputs, the previous pass)
      PA_HIP(hipStreamWaitEvent(ctx->stream2, ctx->sync_evs[0], 0));
      {
        StreamSwap sw(ctx, ctx->stream2);
        // faces AND ring: the ring reads the neighbouring boxes' valid cells in place (phase 4), so it does not wait for FillBoundary
        PA_TRY(pa_gradcurv_prep_levels(ctx, nlev, state, comp, crse.data(), comp, bc, pmin, pmax, ring_side ? 7 : 1));
      }
      PA_HIP(hipEventRecord(ctx->sync_evs[1], ctx->stream2));
    }
    {
      ProfScope prof(ctx, PA_TAG_FILL);
      PA_TRY(pa_fill_boundary_local_batch(ctx, nlev, state, comp, 1, 2));
    }
    if (pov) PA_HIP(hipStreamWaitEvent(ctx->stream, ctx->sync_evs[1], 0));
    if (!(pov && ring_side)) PA_TRY(pa_gradcurv_prep_levels(ctx, nlev, state, comp, crse.data(), comp, bc, pmin, pmax, pov ? 2 : 3));
    PA_TRY(pa_gradcurv_levels_cg(ctx, nlev, state, comp, pmin, pmax, out, ocomp, thr, 0, 1, nullptr, nullptr, nullptr, gout, 0, kg));
    PA_TRY(pa_gradcurv_fix_levels(ctx, nlev, state, comp, crse_n.data(), ncomp0, bc, pmin, pmax, out, ncomp0, kcomp, thr, 1, nullptr, 8, crse.data(), comp));
    return 0;
  }
}

static int fused_passes(pa_ctx* ctx, int nlev, pa_mf* const* state, int comp, const int32_t bc[3], double pmin, double pmax, double thr,
                        pa_mf* const* work, pa_mf* const* out, int ocomp) {
  for (int l = 0; l < nlev; ++l)
    if (state[l]->ng < 2 || work[l]->ng < 2) return pa_fail(ctx, "fused grad->curvature needs 2 ghost layers on state and work");
  // exact-normal pipeline (pa_fused.hip): pure special faces, boxes wider than 32 cells; with the threshold clip the sweep
  // zeroes N and K itself and the one-layer fix-up recomputes the (few) clipped normals it needs (PA_FUSED2_CLIP=0: first pipeline)
  const bool exact = exact_ok(nlev, state, thr);
  if (state[0]->lev->nranks > 1) return fused_passes_dist(ctx, nlev, state, comp, bc, pmin, pmax, thr, work, out, ocomp, exact);
  if (exact) return exact_passes(ctx, nlev, state, comp, bc, pmin, pmax, thr, out, ocomp);
  if (nlev >= 2 && !overlap_on() && conc_on(nlev, state)) return fused_passes_conc(ctx, nlev, state, comp, bc, pmin, pmax, thr, work, out, ocomp);
  if (!overlap_on() || nlev < 2) {
    for (int l = 0; l < nlev; ++l) PA_TRY(fused_pre(ctx, l, state, comp, bc, pmin, pmax, work));
    for (int l = 0; l < nlev; ++l) {
      PA_TRY(pa_gradcurv_level(ctx, state[l], comp, pmin, pmax, thr, out[l], ocomp));
      PA_TRY(fused_faces(ctx, l, bc, thr, work, out, ocomp));
    }
    return 0;
  }
  if (!ctx->stream2) PA_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  const size_t nev = 2 + 3 * (size_t)nlev;
  while (ctx->sync_evs.size() < nev) {
    hipEvent_t e;
    PA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->sync_evs.push_back(e);
  }
  hipStream_t A = ctx->stream, B = ctx->stream2;
  hipEvent_t e_start = ctx->sync_evs[0], e_end = ctx->sync_evs[1];
  auto e_prep = [&](int l) { return ctx->sync_evs[2 + l]; };
  auto e_sweep = [&](int l) { return ctx->sync_evs[2 + nlev + l]; };
  PA_HIP(hipEventRecord(e_start, A));  // B starts after everything already queued on A (inputs, the previous component)
  PA_HIP(hipStreamWaitEvent(B, e_start, 0));
  {
    StreamSwap sw(ctx, B);
    for (int l = 1; l < nlev; ++l) {
      PA_TRY(fused_pre(ctx, l, state, comp, bc, pmin, pmax, work));
      PA_HIP(hipEventRecord(e_prep(l), B));
    }
  }
  PA_TRY(fused_pre(ctx, 0, state, comp, bc, pmin, pmax, work));
  for (int l = 0; l < nlev; ++l) {
    if (l > 0) PA_HIP(hipStreamWaitEvent(A, e_prep(l), 0));
    PA_TRY(pa_gradcurv_level(ctx, state[l], comp, pmin, pmax, thr, out[l], ocomp));
    PA_HIP(hipEventRecord(e_sweep(l), A));
    PA_HIP(hipStreamWaitEvent(B, e_sweep(l), 0));
    StreamSwap sw(ctx, B);
    PA_TRY(fused_faces(ctx, l, bc, thr, work, out, ocomp));
  }
  PA_HIP(hipEventRecord(e_end, B));
  PA_HIP(hipStreamWaitEvent(A, e_end, 0));  // later work on the caller's stream sees the finished level data
  return 0;
}

// pa_curvature_run on the exact-normal pipeline (see there).  out components as curvature_passes writes them.
static int curvature_fast(pa_ctx* ctx, int nlev, pa_mf* const* state, int comp, const int32_t bc[3], double pmin, double pmax, double thr,
                          pa_mf* const* out, int opt, const pa_curv_params* P) {
  const bool gauss = P->do_gauss_curv, strain = P->do_strain, veln = P->do_velnormal, smooth = P->do_smooth != 0;
  for (int l = 0; l < nlev; ++l) {
    const int need = opt + (smooth ? 18 : (strain && P->get_strain_tensor ? 17 : (veln ? 8 : (strain ? 7 : (gauss ? 6 : 5)))));
    if (out[l]->ncomp < need) return pa_fail(ctx, "pa_curvature_run: out needs " + std::to_string(need) + " components for the requested options");
    if ((strain || veln) && (P->vel_comp < 0 || P->vel_comp + 3 > state[l]->ncomp)) return pa_fail(ctx, "pa_curvature_run: vel_comp out of range");
  }
  std::vector<pa_mf*> G(nlev), src(state, state + nlev);
  for (int l = 0; l < nlev; ++l) {
    G[l] = pa_level_scratch(ctx, state[l]->lev, 3, 1);
    if (!G[l]) return 1;
  }
  int scomp = comp;
  double smin = pmin, smax = pmax;
  if (smooth) {
    // :328-406: c~ solves (I - dt Lap) c~ = c; everything below works on c~ (idprogvar = idSmProg, :408).  The sweeps form the progress
    // variable as (x - pmin) * (1 / (pmax - pmin)): with x = c~, pmin = 0, pmax = 1 that is (c~ - 0.0) * 1.0 = c~ bit for bit, so the
    // smoothed field goes through the same pipeline as a source with range [0, 1] (2 ghost layers: a work multifab of the level).
    for (int l = 0; l < nlev; ++l) {
      src[l] = pa_level_scratch(ctx, state[l]->lev, 1, 2, 30);
      if (!src[l]) return 1;
      PA_TRY(pa_progress_level(ctx, state[l], comp, pmin, pmax, src[l], 0, 0));  // :316-320
    }
    int iters = 0;
    double res = 0.0;
    const int32_t bc_s[3] = {bc[0] == PA_BC_PERIODIC ? PA_BC_PERIODIC : PA_BC_NEUMANN, bc[1] == PA_BC_PERIODIC ? PA_BC_PERIODIC : PA_BC_NEUMANN,
                             bc[2] == PA_BC_PERIODIC ? PA_BC_PERIODIC : PA_BC_NEUMANN};  // :348-357, as in curvature_passes
    const int rc = pa_smooth_solve(ctx, nlev, src.data(), 0, src.data(), 0, P->smoothing_time, bc_s, 1e-14, 2000, &iters, &res);
    ctx->smooth_iters = iters;
    ctx->smooth_res = res;
    if (rc != 0 && !(iters > 0 && res <= 1e-12)) return 1;
    scomp = 0; smin = 0.0; smax = 1.0;
  }
  const bool dist = state[0]->lev->nranks > 1;  // sharded hierarchy: the same pipeline with its two exchanges, ghost fills below across ranks
  // round 6: the sweeps form the Gaussian curvature themselves where they can (one rank, every box wider than 32 cells and at least 16
  // rows tall: pa_fused_march3.h GOUT == 2) -- out component opt + 5, right everywhere but in the first layer behind special faces and in
  // the level's irregular cells, which k_gauss_cells recomputes below from the stored G once its ghost cells are resolved
  const bool kg = gauss && !dist && pa_gradcurv_kg_ok(nlev, src.data());
  if (kg) ctx->curv_path = 2;
  if (dist) PA_TRY(fused_passes_dist(ctx, nlev, src.data(), scomp, bc, smin, smax, thr, src.data(), out, opt, true, G.data()));
  else PA_TRY(exact_passes(ctx, nlev, src.data(), scomp, bc, smin, smax, thr, out, opt, G.data(), kg ? 2 : 0));  // :316-322, 426-570
  // :575-613 ghost cells of G = cell_normal before its normalisation, coarse-fine values from the coarser level's G; :679-757 the
  // velocity's likewise: FillBoundary of all levels in one launch each, applyBC of both fields on all levels in ONE launch
  if (dist) {  // every rank makes the same calls in the same order (pa_fill_boundary / pa_apply_bc exchange inside)
    for (int l = 0; l < nlev && gauss; ++l) PA_TRY(pa_fill_boundary(ctx, G[l], 0, 3, 1));
    for (int l = 0; l < nlev && strain; ++l) PA_TRY(pa_fill_boundary(ctx, state[l], P->vel_comp, 3, 1));
  } else {
    if (gauss) PA_TRY(pa_fill_boundary_local_batch(ctx, nlev, G.data(), 0, 3, 1));
    if (strain) PA_TRY(pa_fill_boundary_local_batch(ctx, nlev, state, P->vel_comp, 3, 1));
  }
  if (gauss || strain) {
    // round 6: one rank -- MLMG applyBC of the three components of G and of the velocity through the chunked special-face kernel
    // (k_prep_faces_chunks<.., PHIONLY>: the gradient tool's applyBC, component = slot, coarse values from the gathered patches):
    // k_apply_bc_multi, a thread per ghost cell and component through the owner map, took 1.2 ms of a 17-ms headline pass
    int rc = 2;
    if (!dist) {
      std::vector<const pa_mf*> crG((size_t)nlev, nullptr), crU((size_t)nlev, nullptr);
      for (int l = 1; l < nlev; ++l) { crG[(size_t)l] = G[(size_t)l - 1]; crU[(size_t)l] = state[l - 1]; }
      if (gauss) PA_TRY(pa_gradcurv_prep_levels(ctx, nlev, G.data(), 0, crG.data(), 0, bc, 0.0, 1.0, 1 | 8, 3, nullptr));
      if (strain) PA_TRY(pa_gradcurv_prep_levels(ctx, nlev, state, P->vel_comp, crU.data(), P->vel_comp, bc, 0.0, 1.0, 1 | 8, 3, nullptr));
      rc = 0;
    }
    if (rc == 2) {
      for (int l = 0; l < nlev && gauss; ++l)
        for (int d = 0; d < 3; ++d) PA_TRY(pa_apply_bc(ctx, G[l], d, l > 0 ? G[l - 1] : nullptr, d, bc, 2, -1));
      for (int l = 0; l < nlev && strain; ++l)
        for (int d = 0; d < 3; ++d) PA_TRY(pa_apply_bc(ctx, state[l], P->vel_comp + d, l > 0 ? state[l - 1] : nullptr, P->vel_comp + d, bc, 2, -1));
    }
  }
  if (kg) PA_TRY(pa_gauss_cells_levels(ctx, nlev, G.data(), out, opt, opt + 5, thr));
  const int which = ((gauss && !kg) ? 1 : 0) | (strain ? 2 : 0) | (veln ? 4 : 0);
  // the Gaussian curvature apart from strain + normal velocity: 80 and 118 VGPRs against 158 for all three in one kernel, c read twice;
  // 17.1 against 17.6 ms per headline pass in one kernel
  const bool split = (which & 1) && (which & 6);
  for (int l = 0; l < nlev && which; ++l) {
    if (split) {
      PA_TRY(pa_curvopts_level(ctx, 1, G[l], state[l], P->vel_comp, out[l], opt, opt + 2, opt + 5, opt + 6, opt + 7, -1, thr));
      PA_TRY(pa_curvopts_level(ctx, which & 6, G[l], state[l], P->vel_comp, out[l], opt, opt + 2, opt + 5, opt + 6, opt + 7, P->get_strain_tensor ? opt + 8 : -1, thr));
    } else {
      PA_TRY(pa_curvopts_level(ctx, which, G[l], state[l], P->vel_comp, out[l], opt, opt + 2, opt + 5, opt + 6, opt + 7, P->get_strain_tensor ? opt + 8 : -1, thr));
    }
  }
  if (smooth)  // the sweeps left c~ in the Progress slot (the options' threshold read it there): SmoothedProgress = c~, Progress = the unsmoothed field
    for (int l = 0; l < nlev; ++l) {
      PA_TRY(pa_mf_copy(ctx, out[l], opt, out[l], opt + 17, 1, 0));
      PA_TRY(pa_progress_level(ctx, state[l], comp, pmin, pmax, out[l], opt, 0));
    }
  return 0;
}

static bool all_fusable(int nlev, pa_mf* const* state) {
  for (int l = 0; l < nlev; ++l) {
    if (!state[l]->lev->fusable) return false;
    if (l > 0)  // the whole BoxArray of a sharded level: every rank must take the same path
      for (const DBox& B : (state[l]->lev->gboxes.empty() ? state[l]->lev->boxes : state[l]->lev->gboxes))
        for (int d = 0; d < 3; ++d)
          if (B.hi[d] - B.lo[d] + 1 < 3) return false;
  }
  return true;
}

// can the exact-normal pipeline take this hierarchy (the same answer on every rank)?
static bool exact_ok(int nlev, pa_mf* const* state, double thr) {
  (void)thr;  // (the threshold clip stays on this pipeline since round 3)
  bool exact = pa_opt().fused2 != 0;  // PA_FUSED2=0: the first fused pipeline wherever it is legal
  for (int l = 0; l < nlev; ++l) exact = exact && state[l]->ng >= 2 && pa_fused2_level_ok(state[l]->lev);
  return exact;
}

extern "C" int pa_curvature_last_path(const pa_ctx* ctx) { return ctx ? ctx->curv_path : -1; }

extern "C" int pa_smooth_last(const pa_ctx* ctx, int* iters, double* rel_residual) {
  if (!ctx || ctx->smooth_iters < 0) return 1;
  if (iters) *iters = ctx->smooth_iters;
  if (rel_residual) *rel_residual = ctx->smooth_res;
  return 0;
}

extern "C" int pa_curvature_run(pa_ctx* ctx, int nlev, pa_mf* const* state, int comp, const int32_t bc[3], const pa_curv_params* P,
                                pa_mf* const* out, int ocomp) {
  PaBind bind_(ctx);
  PA_TRY(check_levels(ctx, nlev, state, "pa_curvature_run"));
  PA_TRY(check_levels(ctx, nlev, out, "pa_curvature_run"));
  if (!P) return pa_fail(ctx, "pa_curvature_run: null params");
  double pmin, pmax;
  PA_TRY(prog_minmax(ctx, nlev, state, comp, P, pmin, pmax));
  const double thr = P->do_threshold ? P->threshold : -1.0;
  // 2-D levels: the strain / normal-velocity options work as they are when the caller supplies a zero third velocity
  // component (their z terms are then exact zeros); Gaussian curvature is 3-D only in the reference (curvature.cpp:208-216)
  if (P->spacedim == 2 && P->do_gauss_curv)
    return pa_fail(ctx, "pa_curvature_run: do_gauss_curv is not available for 2-D levels (spacedim = 2)");
  // Fast path (round 5): Progress, K and N from the exact-normal pipeline -- its G-output sweeps leave the cell-centred gradient
  // of c in a work multifab of the level instead of grad phi -- then ONE pass per level for the options.  One rank, 3-D, no
  // smoothing solve, a hierarchy the all-levels sweeps take; fused = 0: pass by pass.
  {
    const bool dist = state[0]->lev->nranks > 1;
    bool fast = P->fused && P->spacedim != 2;
    for (int l = 0; l < nlev && fast; ++l) fast = state[l]->ng >= 2 && (dist || !state[l]->lev->boxes.empty());
    fast = fast && exact_ok(nlev, state, thr) && pa_gradcurv_gout_ok(nlev, state);
    if (dist) {  // the answer depends on the boxes a rank owns: all ranks take the path every one of them can take
      double v = fast ? 1.0 : 0.0;
      PA_TRY(pa_allreduce(ctx, &v, 1, 0));
      fast = v > 0.5;
    }
    ctx->curv_path = fast ? 1 : 0;
    if (fast) return curvature_fast(ctx, nlev, state, comp, bc, pmin, pmax, thr, out, ocomp, P);
  }
  return curvature_passes(ctx, nlev, state, comp, bc, pmin, pmax, thr, out, ocomp, ocomp + 1, ocomp + 2, P, ocomp, P->spacedim == 2 ? 1.0 : 0.5);
}

extern "C" int pa_gradcurv_run(pa_ctx* ctx, int nlev, pa_mf* const* state, int comp, const int32_t bc[3], const pa_curv_params* P,
                               pa_mf* const* work, pa_mf* const* out, int ocomp) {
  PaBind bind_(ctx);
  PA_TRY(check_levels(ctx, nlev, state, "pa_gradcurv_run"));
  PA_TRY(check_levels(ctx, nlev, out, "pa_gradcurv_run"));
  if (!P) return pa_fail(ctx, "pa_gradcurv_run: null params");
  double pmin, pmax;
  PA_TRY(prog_minmax(ctx, nlev, state, comp, P, pmin, pmax));
  const double thr = P->do_threshold ? P->threshold : -1.0;
  // exact-normal pipeline: any BoxArray of boxes >= 3 cells thick (irregular cells are listed and recomputed, pa_fused.hip);
  // first pipeline (A/B switches only): hierarchies without concave coarse-fine corners
  if (P->fused && P->spacedim != 2 && (exact_ok(nlev, state, thr) || all_fusable(nlev, state))) {
    PA_TRY(check_levels(ctx, nlev, work, "pa_gradcurv_run"));
    return fused_passes(ctx, nlev, state, comp, bc, pmin, pmax, thr, work, out, ocomp);
  }
  // boxes thinner than 3 cells, 2-D levels or fused=0: pass by pass
  PA_TRY(pa_grad_run(ctx, nlev, state, comp, bc, out, ocomp));
  return curvature_passes(ctx, nlev, state, comp, bc, pmin, pmax, thr, out, -1, ocomp + 7, ocomp + 4, nullptr, -1, P->spacedim == 2 ? 1.0 : 0.5);
}

// Components comp0 .. comp0+ncomps-1 through the fused pipeline (the reference's tools push one variable through at a time
// too; SURVEY 8d memory budget).  What does not depend on results is done ONCE for all components: the local FillBoundary
// of every component is one launch, and on a sharded hierarchy exchange A (ghost cells of phi + coarse phi under the
// coarse-fine faces) carries all components.  The components go through in BATCHES of nbatch: `out` holds nbatch slots of 8
// components from ocomp, the boundary kernels (resolved ghost values before the sweeps, curvature fix-up after them, the
// coarse-patch gathers, on a sharded hierarchy exchange B of the coarse normals) run ONCE per batch with the slot as a
// grid dimension (pa_fused.hip: SlotK) -- at the size of a level's special faces they are latency bound, so a batch costs
// little more than one component did -- and the sweeps component by component in between.  Exact-normal pipeline only;
// anything else runs component by component through pa_gradcurv_run into slot 0.
extern "C" int pa_gradcurv_run_comps2(pa_ctx* ctx, int nlev, pa_mf* const* state, int comp0, int ncomps, const int32_t bc[3], const pa_curv_params* P,
                                      pa_mf* const* work, pa_mf* const* out, int ocomp, int nbatch, int (*done)(void* user, int comp, int ocomp), void* user) {
  PaBind bind_(ctx);
  PA_TRY(check_levels(ctx, nlev, state, "pa_gradcurv_run_comps"));
  PA_TRY(check_levels(ctx, nlev, out, "pa_gradcurv_run_comps"));
  if (!P) return pa_fail(ctx, "pa_gradcurv_run_comps: null params");
  if (ncomps < 1 || comp0 < 0) return pa_fail(ctx, "pa_gradcurv_run_comps: component range");
  for (int l = 0; l < nlev; ++l)
    if (comp0 + ncomps > state[l]->ncomp) return pa_fail(ctx, "pa_gradcurv_run_comps: component range");
  nbatch = std::max(1, std::min(std::min(nbatch, ncomps), PA_MAXSLOTS));
  for (int l = 0; l < nlev; ++l)
    if (ocomp < 0 || ocomp + 8 * nbatch > out[l]->ncomp) return pa_fail(ctx, "pa_gradcurv_run_comps: out needs 8 components per slot of the batch");
  const double thr = P->do_threshold ? P->threshold : -1.0;
  const bool exact = P->fused && P->spacedim != 2 && exact_ok(nlev, state, thr);
  if (!exact || ncomps == 1) {
    for (int c = comp0; c < comp0 + ncomps; ++c) {
      PA_TRY(pa_gradcurv_run(ctx, nlev, state, c, bc, P, work, out, ocomp));
      if (done && done(user, c, ocomp) != 0) return pa_fail(ctx, "pa_gradcurv_run_comps: the caller's callback failed");
    }
    return 0;
  }
  const bool dist = state[0]->lev->nranks > 1;
  std::vector<CsPlan*> cs(nlev, nullptr);
  std::vector<pa_mf*> csn(nlev, nullptr);
  std::vector<const pa_mf*> crse(nlev, nullptr), crse_n(nlev, nullptr);
  if (dist) {
    std::vector<XJob> jobs;
    for (int l = 0; l < nlev; ++l) {
      XPlan* Pl = pa_fb_plan(ctx, state[l]->lev, 2);
      if (!Pl) return 1;
      jobs.push_back({Pl, state[l], comp0, state[l], comp0, ncomps});
    }
    for (int l = 1; l < nlev; ++l) {
      cs[l] = pa_cs_plan(ctx, state[l]->lev, state[l - 1]->lev, 0, 0, 0);
      if (!cs[l]) return 1;
      pa_mf* m = cs[l]->mf(ctx, ncomps);
      csn[l] = cs[l]->mf(ctx, 3 * nbatch, 1);
      if (cs[l]->cs && (!m || !csn[l])) return 1;
      crse[l] = m;
      crse_n[l] = csn[l];
      jobs.push_back({&cs[l]->x, state[l - 1], comp0, m, 0, ncomps});
    }
    ProfScope prof(ctx, PA_TAG_XCHG);
    PA_TRY(pa_xexchange(ctx, (int)jobs.size(), jobs.data()));
  } else {
    for (int l = 1; l < nlev; ++l) { crse[l] = state[l - 1]; crse_n[l] = out[l - 1]; }
  }
  {
    ProfScope prof(ctx, PA_TAG_FILL);
    PA_TRY(pa_fill_boundary_local_batch(ctx, nlev, state, comp0, ncomps, 2));
  }
  if (!ctx->d_prog) PA_HIP(hipMalloc(&ctx->d_prog, sizeof(double) * 2 * PA_MAXSLOTS));
  std::vector<double> prog(2 * (size_t)nbatch), pmins(nbatch), pmaxs(nbatch);
  for (int g0 = comp0; g0 < comp0 + ncomps; g0 += nbatch) {
    const int ns = std::min(nbatch, comp0 + ncomps - g0);
    for (int z = 0; z < ns; ++z) {
      PA_TRY(prog_minmax(ctx, nlev, state, g0 + z, P, pmins[z], pmaxs[z]));
      prog[2 * z] = pmins[z];
      prog[2 * z + 1] = 1.0 / (pmaxs[z] - pmins[z]);
    }
    // (pageable source: the copy is staged before the call returns, so the host vector may be rewritten for the next batch)
    PA_HIP(hipMemcpyAsync(ctx->d_prog, prog.data(), sizeof(double) * 2 * ns, hipMemcpyHostToDevice, ctx->stream));
    PA_TRY(pa_gradcurv_prep_levels(ctx, nlev, state, g0, crse.data(), dist ? g0 - comp0 : g0, bc, pmins[0], pmaxs[0], 3, ns, ctx->d_prog));
    {
      // the sweeps of the batch's components: ONE launch with the slot as a grid dimension
      if (ns > 1) PA_TRY(pa_gradcurv_levels_cg(ctx, nlev, state, g0, pmins[0], pmaxs[0], out, ocomp, thr, 0, ns, ctx->d_prog, pmins.data(), pmaxs.data()));
      else for (int z = 0; z < ns; ++z) PA_TRY(pa_gradcurv_levels_cg(ctx, nlev, state, g0 + z, pmins[z], pmaxs[z], out, ocomp + 8 * z, thr, z));
    }
    if (dist) {
      std::vector<XJob> jobs;
      for (int l = 1; l < nlev; ++l) {
        XJob J = {&cs[l]->x, out[l - 1], ocomp + 4, csn[l], 0, 3 * ns};
        J.group = 3; J.sgstride = 8; J.dgstride = 3;
        jobs.push_back(J);
      }
      ProfScope prof(ctx, PA_TAG_XCHG);
      PA_TRY(pa_xexchange(ctx, (int)jobs.size(), jobs.data()));
    }
    PA_TRY(pa_gradcurv_fix_levels(ctx, nlev, state, g0, crse_n.data(), dist ? 0 : ocomp + 4, bc, pmins[0], pmaxs[0], out, ocomp + 4, ocomp + 7, thr, ns, ctx->d_prog,
                                  dist ? 3 : 8, crse.data(), dist ? g0 - comp0 : g0));
    for (int z = 0; z < ns; ++z)
      if (done && done(user, g0 + z, ocomp + 8 * z) != 0) return pa_fail(ctx, "pa_gradcurv_run_comps: the caller's callback failed");
  }
  return 0;
}

// one slot: every component into out components ocomp .. ocomp + 7 (the round-2 entry point)
namespace {
struct Done1 { int (*done)(void*, int); void* user; };
int done1_thunk(void* u, int comp, int) {
  const Done1* d = static_cast<const Done1*>(u);
  return d->done ? d->done(d->user, comp) : 0;
}
}  // namespace
extern "C" int pa_gradcurv_run_comps(pa_ctx* ctx, int nlev, pa_mf* const* state, int comp0, int ncomps, const int32_t bc[3], const pa_curv_params* P,
                                     pa_mf* const* work, pa_mf* const* out, int ocomp, int (*done)(void* user, int comp), void* user) {
  Done1 d{done, user};
  return pa_gradcurv_run_comps2(ctx, nlev, state, comp0, ncomps, bc, P, work, out, ocomp, 1, done1_thunk, &d);
}
