/*
 * pa_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the PeleAnalysis stencil hot path:
 *   grad.cpp:158-236, curvature.cpp:283-326 + 408-570 (+ options 575-789),
 *   filterPlt.cpp:120-219, isosurface.cpp:257-301 + 415-802 + 1434-1728.
 *
 * PARITY UNPINNED: the reference cannot be compiled here (AMReX / PelePhysics
 * submodules are absent) and it ships no tests, golden vectors
 * or fixtures for this path.  The arithmetic of the absent AMReX pieces
 * (MLPoisson flux, MLMG applyBC, InterpBndryData, FillPatch, Filter) is restated
 * from their published algorithm as summarised in SURVEY.md Appendix A.  What
 * pins the oracle instead: analytic known-answer tests, the marching-cubes table
 * digests computed from isosurface.cpp:451-741, and MEF layout cross-checks.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library, and only as the checker.
 *
 * Data model: a level is a list of boxes (inclusive cell-index boxes) inside a
 * domain box; a multifab is one flat double buffer, box b starting at off[b],
 * laid out [comp][k][j][i] over the box grown by ng (i fastest, like an AMReX
 * FArrayBox).
 */
#ifndef PA_ORACLE_H
#define PA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t nboxes;
  const int32_t* boxes; /* [nboxes][6] = lo0 lo1 lo2 hi0 hi1 hi2 */
  int32_t domlo[3], domhi[3];
  int32_t is_per[3];
  double prob_lo[3], prob_hi[3];
} orc_level;

typedef struct {
  const orc_level* lev;
  int32_t ncomp, ng;
  double* data;
  const int64_t* off; /* [nboxes] offset of box b in doubles */
  const int64_t* cstride; /* [nboxes] component stride in doubles (>= cells incl. ghosts) */
} orc_mf;

/* boundary-condition types per dimension (lo == hi in the reference tools) */
enum { ORC_BC_PERIODIC = 0, ORC_BC_NEUMANN = 1, ORC_BC_REFLECT_ODD = 2 };

/* geometry: dx = (prob_hi-prob_lo)/ncells, dxinv = 1/dx (AMReX Geometry) */
void orc_dxinv(const orc_level* L, double dxinv[3]);

/* FabArray::FillBoundary: every ghost cell (faces, edges, corners, ng_fill
 * deep) that overlaps a valid box or a periodic image is copied. */
void orc_fill_boundary(orc_mf* mf, int comp, int ncomp, int ng_fill);

/* MLCellLinOp::applyBC restatement for the ring-1 FACE ghost cells of comp:
 * covered -> untouched; outside a non-periodic domain -> Neumann/reflect_odd;
 * otherwise coarse-fine: tangential interpolation of the coarse data
 * (InterpBndryData, order 3) + normal cubic (maxorder 4).  crse may be NULL
 * on level 0.  ratio is the refinement ratio (2).  Returns the number of
 * coarse-fine ghost cells whose coarse data could not be found (0 = OK). */
int orc_apply_bc(orc_mf* fine, int comp, const orc_mf* crse, int ccomp,
                 const int32_t bc[3], int ratio, int only_dir /* -1 = all */);

/* grad.cpp:211-236, reference-shaped multi-pass form (face fluxes with the
 * MLPoisson sign, 1/bscalar = -1, average_face_to_cellcenter, mult(-1),
 * magnitude).  phi ghosts must be resolved.  out comps ocomp..ocomp+3 =
 * gx gy gz |g| on valid cells (out may have any ng). */
void orc_grad_multipass(const orc_mf* phi, int comp, orc_mf* out, int ocomp);
/* same arithmetic, one sweep (the "fused" CPU baseline variant) */
void orc_grad_fused(const orc_mf* phi, int comp, orc_mf* out, int ocomp, int with_mag);

/* curvature.cpp:139-149 : min / max over valid cells of all boxes */
void orc_minmax(const orc_mf* s, int comp, double* mn, double* mx);
/* curvature.cpp:310-321 : c = (s - pmin) * (1/(pmax-pmin)) on valid cells */
void orc_progress(const orc_mf* s, int comp, double pmin, double pmax, orc_mf* c, int ccomp);
/* curvature.cpp:467-502 : from G (3 comps, valid) -> normgrad = -max(1e-14,|G|),
 * n = G / normgrad on valid cells */
void orc_normal(const orc_mf* G, int gcomp, orc_mf* normgrad, int ngcomp, orc_mf* n, int ncomp0);
/* curvature.cpp:508-540 one direction: Curv += d n_d / d x_d (central, via
 * face fluxes).  nd has resolved ring-1 face ghosts in direction dir. */
void orc_div_accum(const orc_mf* nd, int comp, int dir, orc_mf* curv, int kcomp);
/* fused single-sweep CPU variant of grad.cpp:211-236 + curvature.cpp:457-546 (bench.py cpu_baseline "fused"; see pa_oracle.c) */
void orc_gradcurv_fused(const orc_mf* phi, int pcomp, const orc_mf* c, int ccomp, orc_mf* gout, int gcomp, orc_mf* nmf, int ncomp0, orc_mf* K, int kcomp);
void orc_curv_first_layer(const orc_mf* nmf, int ncomp0, orc_mf* K, int kcomp);
/* mf[comp] = v on valid cells / mf *= v on valid cells */
void orc_setval(orc_mf* mf, int comp, double v);
void orc_mult(orc_mf* mf, int comp, double v);
/* curvature.cpp:549-567 threshold clip (in place on K and n) */
void orc_threshold(const orc_mf* c, int ccomp, double thr, orc_mf* K, int kcomp, orc_mf* n, int ncomp0);
/* copy valid cells (MultiFab::Copy with ng=0); same BoxArray */
void orc_copy(const orc_mf* src, int scomp, orc_mf* dst, int dcomp, int ncomp, int ng);
/* curvature.cpp:618-674 Gaussian curvature from Hessian (9 comps, rows =
 * d(G_row)/dx_col), G (3) and normgrad; thr<0 disables the threshold */
void orc_gauss_curv(const orc_mf* H, const orc_mf* G, const orc_mf* normgrad, const orc_mf* c,
                    int ccomp, double thr, orc_mf* Kg, int kcomp);
/* curvature.cpp:722-749 (quirk Q3 kept: result is div u) and :765-787 */
void orc_strain_rate(const orc_mf* gradU, const orc_mf* n, orc_mf* sr, int comp);
void orc_vel_normal(const orc_mf* u, int ucomp, const orc_mf* n, const orc_mf* c, int ccomp,
                    double thr, orc_mf* out, int ocomp);

/* ---- filterPlt --------------------------------------------------------- */
/* PelePhysics Filter (box): weights for filter-to-grid ratio fgr; returns ngrow */
int orc_box_filter_weights(int fgr, double* w /* >= fgr+1 entries */);
/* Filter::apply_filter: out = sum_n sum_m sum_l w w w in(i+l,j+m,k+n) */
void orc_apply_filter(const orc_mf* in, orc_mf* out, int scomp, int ncomp, int ngf, const double* w);
/* first-order extrapolation at non-periodic walls (FillPatchSingleLevel +
 * foextrap), ng_fill deep, applied x then y then z on the grown box */
void orc_foextrap(orc_mf* mf, int comp, int ncomp, int ng_fill);
/* FillPatchTwoLevels for ghost cells: fine data where covered, else
 * interpolation from the coarse multifab (valid + its own filled ghosts);
 * interp_type 1 = cell-conservative linear, 0 = piecewise constant */
int orc_fillpatch_two_levels(orc_mf* fine, const orc_mf* crse, int comp, int ncomp, int ng_fill,
                             int ratio, int interp_type);

/* ---- curvature.cpp:328-406 (do_smooth), pa_oracle_smooth.c ------------- */
void orc_smooth_mask(const orc_mf* mask, const orc_level* fine, int ratio);
void orc_smooth_avgdown(const orc_mf* fine, int fcomp, orc_mf* crse, int ccomp, int ratio);
void orc_smooth_apply_level(const orc_mf* x, int xc, orc_mf* y, int yc, double dt);
void orc_smooth_reflux(const orc_mf* xf, int xfc, const orc_mf* xc, int xcc, orc_mf* yc, int ycc, double dt, int ratio);
void orc_smooth_dot(const orc_mf* a, const orc_mf* b, const orc_mf* mask, double* dot, double* amax);
/* y = A x, the composite operator (x: covered cells and ghost cells are overwritten) */
void orc_smooth_apply(int nlev, orc_mf* const* x, orc_mf* const* y, const orc_mf* const* mask, double dt, const int32_t bc[3], int ratio);
/* (I - dt Lap) sol = rhs, composite, BiCGStab to ||b - A x||_inf <= tol ||b||_inf; work = 7 vectors per
 * level (1 comp, ng 1), the last one receives the covered-cell mask.  Returns iterations (< 0: failed). */
int orc_smooth_solve(int nlev, orc_mf* const* rhs, int rcomp, orc_mf* const* sol, orc_mf* const* work, double dt,
                     const int32_t bc[3], int ratio, double tol, int maxiter, double* res);

/* ---- partStream.cpp / StreamPC.cpp, pa_oracle_stream.c ----------------- */
int orc_stream_trace(int nlev, const orc_mf* const* v, int vcomp, int64_t nseed, const double* seeds, int nsteps, double dt, double* pos,
                     int32_t* nredist);

/* ---- isosurface -------------------------------------------------------- */
const int32_t* orc_mc_edge_table(void); /* [256] */
const int32_t* orc_mc_tri_table(void);  /* [256][16] */
/* Marching cubes on one FAB (isosurface.cpp:1531-1592 + Polygonise).
 * state: [ncomp][nz][ny][nx] over box (slo..shi) with comps 0..2 = x,y,z and
 * field comps after; mask: same box, 1 comp (<0 = covered).  loop box =
 * base points.  Output (caller allocated, capacities given):
 *   verts[nv][ncomp]  vertex data in vertCache (std::map<Edge>) order,
 *   vkeys[nv][6]      sorted edge endpoints (l then r),
 *   tris[nt][3]       local vertex ids, in cube traversal order.
 * returns 0, or -1 if a capacity was exceeded; *nv,*nt always = needed. */
int orc_mc_fab(const double* state, const double* mask, const int32_t slo[3], const int32_t shi[3],
               int ncomp, int isocomp, double isoval, const int32_t llo[3], const int32_t lhi[3],
               double* verts, int32_t* vkeys, int64_t vcap, int32_t* tris, int64_t tcap,
               int64_t* nv, int64_t* nt);

#ifdef __cplusplus
}
#endif
#endif
