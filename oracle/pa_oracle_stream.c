/*
 * pa_oracle_stream.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Streamline tracer of partStream.cpp:121-207 / StreamPC.cpp: RK4 through a
 * piecewise-trilinear vector field on the AMR hierarchy, two lines per seed (forward / backward).
 *   vnrml  StreamPC.cpp:143-157  (normalise unless |v|^2 >= 1e12, times the direction)
 *   ntrpv  :159-206              (trilinear interpolation inside ONE FAB incl. its ghost cells)
 *   RK4    :208-260              (classic RK4 on the normalised field; the step is cut by the quirky
 *                                 test `x+delta > plo` at :247, kept; final clamp to the domain +-1e-10)
 *   SetParticleLocation :88-141 + ComputeNextLocation :262-306: a line keeps interpolating from the FAB
 *   (level, grid) it was last assigned to; when ANY live line has left its grid grown by nGrow-1 cells,
 *   ALL lines are re-assigned (AMReX Redistribute -> Where(): finest level whose grids contain the
 *   line's cell).  [Redistribute/Where are AMReX: recalled, PARITY UNPINNED.]
 * The vector field multifabs must already hold their ghost cells (FillPatch with piecewise-constant
 * interpolation + FillBoundary, partStream.cpp:160-177).
 */
#include "pa_oracle.h"
#include <math.h>
#include <stdlib.h>

typedef struct { int lo[3], hi[3], n[3]; } tbx_t;
static tbx_t tbox(const orc_level* L, int b) {
  tbx_t r;
  for (int d = 0; d < 3; ++d) {
    r.lo[d] = L->boxes[6 * b + d];
    r.hi[d] = L->boxes[6 * b + 3 + d];
    r.n[d] = r.hi[d] - r.lo[d] + 1;
  }
  return r;
}
static int tfind(const orc_level* L, const int p[3]) {
  for (int b = 0; b < L->nboxes; ++b) {
    const int32_t* q = L->boxes + 6 * b;
    if (p[0] >= q[0] && p[0] <= q[3] && p[1] >= q[1] && p[1] <= q[4] && p[2] >= q[2] && p[2] <= q[5]) return b;
  }
  return -1;
}

/* StreamPC.cpp:143-157 */
static void vnrml(double vec[3], int dir) {
  const double eps = 1.e12;
  double sum = vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2];
  if (sum < eps) {
    sum = 1. / sqrt(sum);
    for (int i = 0; i < 3; ++i) vec[i] *= dir * sum;
  } else {
    vec[0] = vec[1] = vec[2] = 0.0;
  }
}

/* StreamPC.cpp:159-206; returns 0 if the interpolation cell leaves the FAB */
static int ntrpv(const double x[3], const orc_mf* v, int vcomp, int b, const tbx_t* B, const double dx[3], const double plo[3], double u[3]) {
  int bi[3];
  double n[3];
  const int ng = v->ng;
  for (int d = 0; d < 3; ++d) {
    bi[d] = (int)floor((x[d] - plo[d]) / dx[d] - 0.5);
    n[d] = (x[d] - ((bi[d] + 0.5) * dx[d] + plo[d])) / dx[d];
    n[d] = (n[d] < 1.) ? n[d] : 1.;  /* std::min(1., n) */
    n[d] = (0. < n[d]) ? n[d] : 0.;  /* std::max(0., .) */
    if (bi[d] < B->lo[d] - ng || bi[d] > B->hi[d] + ng - 1) return 0;
  }
  const int64_t nx = B->n[0] + 2 * ng, ny = B->n[1] + 2 * ng;
  for (int i = 0; i < 3; ++i) {
    const double* g = v->data + v->off[b] + (int64_t)(vcomp + i) * v->cstride[b] +
                      ((int64_t)(bi[2] - B->lo[2] + ng) * ny + (bi[1] - B->lo[1] + ng)) * nx + (bi[0] - B->lo[0] + ng);
    const int64_t sy = nx, sz = nx * ny;
    u[i] = +n[0] * n[1] * n[2] * g[1 + sy + sz]
           + n[0] * (1 - n[1]) * n[2] * g[1 + sz]
           + n[0] * n[1] * (1 - n[2]) * g[1 + sy]
           + n[0] * (1 - n[1]) * (1 - n[2]) * g[1]
           + (1 - n[0]) * n[1] * n[2] * g[sy + sz]
           + (1 - n[0]) * (1 - n[1]) * n[2] * g[sz]
           + (1 - n[0]) * n[1] * (1 - n[2]) * g[sy]
           + (1 - n[0]) * (1 - n[1]) * (1 - n[2]) * g[0];
  }
  return 1;
}

/* StreamPC.cpp:208-260 */
static int rk4(double x[3], double dt, const orc_mf* v, int vcomp, int b, const tbx_t* B, const double dx[3], const double plo[3], const double phi[3],
               int dir) {
  double vec[3], k1[3], k2[3], k3[3], k4[3], xx[3] = {x[0], x[1], x[2]};
  if (!ntrpv(xx, v, vcomp, b, B, dx, plo, vec)) return 0;
  vnrml(vec, dir);
  for (int d = 0; d < 3; ++d) { k1[d] = vec[d] * dt; xx[d] = x[d] + k1[d] * 0.5; }
  if (!ntrpv(xx, v, vcomp, b, B, dx, plo, vec)) return 0;
  vnrml(vec, dir);
  for (int d = 0; d < 3; ++d) { k2[d] = vec[d] * dt; xx[d] = x[d] + k2[d] * 0.5; }
  if (!ntrpv(xx, v, vcomp, b, B, dx, plo, vec)) return 0;
  vnrml(vec, dir);
  for (int d = 0; d < 3; ++d) { k3[d] = vec[d] * dt; xx[d] = x[d] + k3[d]; }
  if (!ntrpv(xx, v, vcomp, b, B, dx, plo, vec)) return 0;
  vnrml(vec, dir);
  const double third = 1. / 3., sixth = 1. / 6.;
  double delta[3];
  for (int d = 0; d < 3; ++d) {
    k4[d] = vec[d] * dt;
    delta[d] = (k1[d] + k4[d]) * sixth + (k2[d] + k3[d]) * third;
  }
  double scale = 1;
  for (int d = 0; d < 3; ++d) {
    if (x[d] + delta[d] < plo[d]) { const double s = fabs((x[d] - plo[d]) / delta[d]); scale = (s < scale) ? s : scale; }
    if (x[d] + delta[d] > plo[d]) { const double s = fabs((phi[d] - x[d]) / delta[d]); scale = (s < scale) ? s : scale; }  /* :247 as written */
  }
  for (int d = 0; d < 3; ++d) {
    x[d] += scale * delta[d];
    const double lo = plo[d] + 1.e-10, hi = phi[d] - 1.e-10;
    const double m = (lo < x[d]) ? x[d] : lo;  /* std::max(plo+1e-10, x) */
    x[d] = (m < hi) ? m : hi;                  /* std::min(phi-1e-10, .) */
  }
  return 1;
}

/* Where(): finest level whose grids contain the cell of x */
static void where_is(int nlev, const orc_level* const* L, const double x[3], int* lev, int* grid) {
  for (int l = nlev - 1; l >= 0; --l) {
    int p[3];
    for (int d = 0; d < 3; ++d) {
      const double dx = (L[l]->prob_hi[d] - L[l]->prob_lo[d]) / (double)(L[l]->domhi[d] - L[l]->domlo[d] + 1);
      p[d] = (int)floor((x[d] - L[l]->prob_lo[d]) / dx);
    }
    const int b = tfind(L[l], p);
    if (b >= 0) { *lev = l; *grid = b; return; }
  }
  *lev = -1; *grid = -1;
}

/* pos: [2*nseed][nsteps][3]; line 2s runs forward (dir +1), 2s+1 backward; point 0 = the seed.
 * Returns 0, or the 1-based index of the first line whose interpolation left its FAB ("bad RK" abort). */
int orc_stream_trace(int nlev, const orc_mf* const* v, int vcomp, int64_t nseed, const double* seeds, int nsteps, double dt, double* pos,
                     int32_t* nredist /* out: how many redistributions happened */) {
  const int64_t np = 2 * nseed;
  const orc_level** L = (const orc_level**)malloc(sizeof(void*) * (size_t)nlev);
  int* lev = (int*)malloc(sizeof(int) * (size_t)(np > 0 ? np : 1));
  int* grd = (int*)malloc(sizeof(int) * (size_t)(np > 0 ? np : 1));
  for (int l = 0; l < nlev; ++l) L[l] = v[l]->lev;
  const int ng = v[0]->ng;
  int rc = 0, nred = 0;
  for (int64_t p = 0; p < np; ++p) {
    for (int d = 0; d < 3; ++d) pos[(p * nsteps) * 3 + d] = seeds[(p / 2) * 3 + d];
    where_is(nlev, L, &pos[(p * nsteps) * 3], &lev[p], &grd[p]);  /* InitParticles -> Redistribute() */
  }
  for (int step = 0; step + 1 < nsteps && !rc; ++step) {
    int redist = 0;  /* SetParticleLocation(step, nGrow) */
    for (int64_t p = 0; p < np; ++p) {
      if (lev[p] < 0) continue;
      const orc_level* Lp = L[lev[p]];
      const tbx_t B = tbox(Lp, grd[p]);
      for (int d = 0; d < 3; ++d) {
        const double dx = (Lp->prob_hi[d] - Lp->prob_lo[d]) / (double)(Lp->domhi[d] - Lp->domlo[d] + 1);
        const double blo = Lp->prob_lo[d] + (B.lo[d] - (ng - 1)) * dx, bhi = Lp->prob_lo[d] + (B.hi[d] + (ng - 1) + 1) * dx;
        const double x = pos[(p * nsteps + step) * 3 + d];
        redist |= (x < blo || x > bhi);
      }
    }
    if (redist) {
      ++nred;
      for (int64_t p = 0; p < np; ++p)
        if (lev[p] >= 0) where_is(nlev, L, &pos[(p * nsteps + step) * 3], &lev[p], &grd[p]);
    }
    for (int64_t p = 0; p < np && !rc; ++p) {  /* ComputeNextLocation */
      double x[3] = {pos[(p * nsteps + step) * 3], pos[(p * nsteps + step) * 3 + 1], pos[(p * nsteps + step) * 3 + 2]};
      if (lev[p] >= 0) {
        const orc_level* Lp = L[lev[p]];
        const tbx_t B = tbox(Lp, grd[p]);
        double dx[3];
        for (int d = 0; d < 3; ++d) dx[d] = (Lp->prob_hi[d] - Lp->prob_lo[d]) / (double)(Lp->domhi[d] - Lp->domlo[d] + 1);
        if (!rk4(x, dt, v[lev[p]], vcomp, grd[p], &B, dx, Lp->prob_lo, Lp->prob_hi, (p & 1) ? -1 : +1)) rc = (int)(p + 1);
      }
      for (int d = 0; d < 3; ++d) pos[(p * nsteps + step + 1) * 3 + d] = x[d];
    }
  }
  if (nredist) *nredist = nred;
  free(L); free(lev); free(grd);
  return rc;
}
