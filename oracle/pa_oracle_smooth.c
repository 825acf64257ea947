/*
 * pa_oracle_smooth.c -- CPU ORACLE (test infrastructure, NOT product code).
 * curvature.cpp:328-406 (do_smooth): one implicit diffusion step of the progress variable,
 *     (alpha A - beta div B grad) phi = rhs,  alpha = 1, A = 1, beta = smoothing_time, B = 1,
 * as a COMPOSITE solve over the AMR hierarchy (MLABecLaplacian + MLMG in the reference; both live in
 * AMReX, which is absent here: PARITY UNPINNED, semantics recalled -- SURVEY Appendix A):
 *   - domain boundaries periodic or homogeneous Neumann (setLevelBC(lev, nullptr), :358-370);
 *   - fine ghost cells at coarse-fine faces interpolated from the coarse solution (the same
 *     MLCellLinOp::applyBC as the gradient operators, setMaxOrder(4), :344);
 *   - the flux through a coarse-fine face seen from the coarse side is the average of the fine
 *     fluxes (reflux); coarse cells covered by a finer level carry the average of their children;
 *   - solved to ||b - A x||_inf <= tol * ||b||_inf (tol_rel = tol_abs = 1e-12, :346-347).
 * MLMG's multigrid cycle is not restated: any solver that reaches the tolerance gives the same
 * solution to ~tol * cond; BiCGStab on the composite operator is used here and in the HIP library,
 * and the two are compared to a tolerance (tests/test_gpu_smooth.py), not bit for bit.
 * What pins this file: discrete eigenfunctions on a periodic level (exact solution known),
 * conservation of the composite integral (telescoping of the refluxed fluxes) and the residual
 * itself (tests/test_oracle_smooth.py).
 */
#include "pa_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int lo[3], hi[3], n[3]; } sbx_t;
static sbx_t sbox(const orc_level* L, int b) {
  sbx_t r;
  for (int d = 0; d < 3; ++d) {
    r.lo[d] = L->boxes[6 * b + d];
    r.hi[d] = L->boxes[6 * b + 3 + d];
    r.n[d] = r.hi[d] - r.lo[d] + 1;
  }
  return r;
}
static int64_t sidx(const orc_mf* m, const sbx_t* B, int b, int c, int i, int j, int k) {
  const int ng = m->ng;
  const int64_t nx = B->n[0] + 2 * ng, ny = B->n[1] + 2 * ng;
  return m->off[b] + (int64_t)c * m->cstride[b] + ((int64_t)(k - B->lo[2] + ng) * ny + (j - B->lo[1] + ng)) * nx + (i - B->lo[0] + ng);
}
#define SAT(m, B, b, c, i, j, k) ((m)->data[sidx((m), (B), (b), (c), (i), (j), (k))])

static int swrap(const orc_level* L, int p[3]) {
  for (int d = 0; d < 3; ++d) {
    const int len = L->domhi[d] - L->domlo[d] + 1;
    if (p[d] < L->domlo[d] || p[d] > L->domhi[d]) {
      if (!L->is_per[d]) return 0;
      while (p[d] < L->domlo[d]) p[d] += len;
      while (p[d] > L->domhi[d]) p[d] -= len;
    }
  }
  return 1;
}
static int sfind(const orc_level* L, const int p[3], int hint) {
  if (hint >= 0) {
    const int32_t* q = L->boxes + 6 * hint;
    if (p[0] >= q[0] && p[0] <= q[3] && p[1] >= q[1] && p[1] <= q[4] && p[2] >= q[2] && p[2] <= q[5]) return hint;
  }
  for (int b = 0; b < L->nboxes; ++b) {
    const int32_t* q = L->boxes + 6 * b;
    if (p[0] >= q[0] && p[0] <= q[3] && p[1] >= q[1] && p[1] <= q[4] && p[2] >= q[2] && p[2] <= q[5]) return b;
  }
  return -1;
}
static int sfloor_div(int i, int r) { return (i < 0) ? -((-i + r - 1) / r) : i / r; }
/* refinement ratio of direction d: the AMREX_SPACEDIM == 2 build is a hierarchy of one-cell-thick planes (k = 0 on every
 * level), which is not refined in z */
static int rdir(const orc_level* fine, int d, int ratio) { return (d == 2 && fine->domlo[2] == fine->domhi[2]) ? 1 : ratio; }

/* mask[l] = 1 on valid cells not covered by level l+1, 0 on covered ones (same layout as the vectors) */
void orc_smooth_mask(const orc_mf* mask, const orc_level* fine /* NULL on the finest level */, int ratio) {
  const orc_level* L = mask->lev;
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    sbx_t B = sbox(L, b);
    int hint = -1;
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
          double m = 1.0;
          if (fine) {
            int p[3] = {i * ratio, j * ratio, k * rdir(fine, 2, ratio)};
            const int fb = sfind(fine, p, hint);
            if (fb >= 0) { hint = fb; m = 0.0; }
          }
          SAT(mask, &B, b, 0, i, j, k) = m;
        }
  }
}

/* average_down: coarse cells under a fine box = mean of their ratio^3 children (sum in k,j,i order) */
void orc_smooth_avgdown(const orc_mf* fine, int fcomp, orc_mf* crse, int ccomp, int ratio) {
  const orc_level *LF = fine->lev, *LC = crse->lev;
  const int rz = rdir(LF, 2, ratio);
  const double fac = 1.0 / (double)(ratio * ratio * rz);
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < LF->nboxes; ++b) {
    sbx_t B = sbox(LF, b);
    int hint = -1;
    for (int kc = sfloor_div(B.lo[2], rz); kc <= sfloor_div(B.hi[2], rz); ++kc)
      for (int jc = sfloor_div(B.lo[1], ratio); jc <= sfloor_div(B.hi[1], ratio); ++jc)
        for (int ic = sfloor_div(B.lo[0], ratio); ic <= sfloor_div(B.hi[0], ratio); ++ic) {
          double c = 0.0;
          for (int kk = 0; kk < rz; ++kk)
            for (int jj = 0; jj < ratio; ++jj)
              for (int ii = 0; ii < ratio; ++ii) c += SAT(fine, &B, b, fcomp, ic * ratio + ii, jc * ratio + jj, kc * rz + kk);
          c *= fac;
          int p[3] = {ic, jc, kc};
          const int cb = sfind(LC, p, hint);
          if (cb < 0) continue;
          hint = cb;
          sbx_t C = sbox(LC, cb);
          SAT(crse, &C, cb, ccomp, ic, jc, kc) = c;
        }
  }
}

/* y = x - dt * div(grad x) on the valid cells of one level; x has resolved ring-1 face ghosts */
void orc_smooth_apply_level(const orc_mf* x, int xc, orc_mf* y, int yc, double dt) {
  const orc_level* L = x->lev;
  double dxinv[3];
  orc_dxinv(L, dxinv);
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    sbx_t B = sbox(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
          const double c = SAT(x, &B, b, xc, i, j, k);
          double div = 0.0;
          div += dxinv[0] * (dxinv[0] * (SAT(x, &B, b, xc, i + 1, j, k) - c) - dxinv[0] * (c - SAT(x, &B, b, xc, i - 1, j, k)));
          div += dxinv[1] * (dxinv[1] * (SAT(x, &B, b, xc, i, j + 1, k) - c) - dxinv[1] * (c - SAT(x, &B, b, xc, i, j - 1, k)));
          div += dxinv[2] * (dxinv[2] * (SAT(x, &B, b, xc, i, j, k + 1) - c) - dxinv[2] * (c - SAT(x, &B, b, xc, i, j, k - 1)));
          SAT(y, &B, b, yc, i, j, k) = c - dt * div;
        }
  }
}

/* reflux: for every coarse face between an uncovered coarse cell and the fine region, replace the
 * coarse flux in y_crse by the average of the ratio^2 fine fluxes (fine ghosts resolved by applyBC) */
void orc_smooth_reflux(const orc_mf* xf, int xfc, const orc_mf* xc, int xcc, orc_mf* yc, int ycc, double dt, int ratio) {
  const orc_level *LF = xf->lev, *LC = xc->lev;
  double dxf[3], dxc[3];
  orc_dxinv(LF, dxf);
  orc_dxinv(LC, dxc);
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < LF->nboxes; ++b) {
    sbx_t B = sbox(LF, b);
    int fh = b, ch = -1;
    for (int dir = 0; dir < 3; ++dir) {
      const int t0 = (dir == 0) ? 1 : 0, t1 = (dir == 2) ? 1 : 2;
      const int r0 = rdir(LF, t0, ratio), r1 = rdir(LF, t1, ratio), rn = rdir(LF, dir, ratio);
      const double fac = 1.0 / (double)(r0 * r1);
      for (int side = 0; side < 2; ++side) {
        const int gq = side ? B.hi[dir] + 1 : B.lo[dir] - 1;  /* fine ghost plane */
        const int inq = side ? B.hi[dir] : B.lo[dir];        /* fine cells next to the face */
        for (int b1 = sfloor_div(B.lo[t1], r1); b1 <= sfloor_div(B.hi[t1], r1); ++b1)
          for (int a0 = sfloor_div(B.lo[t0], r0); a0 <= sfloor_div(B.hi[t0], r0); ++a0) {
            int q[3];
            q[dir] = gq; q[t0] = a0 * r0; q[t1] = b1 * r1;
            int p[3] = {q[0], q[1], q[2]};
            if (!swrap(LF, p)) continue;                 /* physical boundary: no coarse-fine face */
            const int nb = sfind(LF, p, fh);
            if (nb >= 0) continue;                       /* another fine box: interior face */
            /* outside coarse cell (uncovered) and inside coarse cell (covered, holds the average) */
            int oc[3], ic[3];
            oc[dir] = sfloor_div(gq, rn); oc[t0] = a0; oc[t1] = b1;
            ic[dir] = sfloor_div(inq, rn); ic[t0] = a0; ic[t1] = b1;
            int ow[3] = {oc[0], oc[1], oc[2]}, iw[3] = {ic[0], ic[1], ic[2]};
            if (!swrap(LC, ow) || !swrap(LC, iw)) continue;
            const int ob = sfind(LC, ow, ch);
            if (ob < 0) continue;
            ch = ob;
            const int ib = sfind(LC, iw, ch);
            if (ib < 0) continue;
            sbx_t OB = sbox(LC, ob), IB = sbox(LC, ib);
            double favg = 0.0;
            for (int v = 0; v < r1; ++v)
              for (int u = 0; u < r0; ++u) {
                int g[3], in[3];
                g[dir] = gq; g[t0] = a0 * r0 + u; g[t1] = b1 * r1 + v;
                in[dir] = inq; in[t0] = g[t0]; in[t1] = g[t1];
                const double xg = SAT(xf, &B, b, xfc, g[0], g[1], g[2]), xi = SAT(xf, &B, b, xfc, in[0], in[1], in[2]);
                favg += side ? dxf[dir] * (xg - xi) : dxf[dir] * (xi - xg);  /* flux in +dir */
              }
            favg *= fac;
            const double xo = SAT(xc, &OB, ob, xcc, ow[0], ow[1], ow[2]), xin = SAT(xc, &IB, ib, xcc, iw[0], iw[1], iw[2]);
            const double fc = side ? dxc[dir] * (xo - xin) : dxc[dir] * (xin - xo);
            /* low fine face = HIGH face of the outside coarse cell: div += dxinv*(favg - fc) */
            const double corr = dt * (dxc[dir] * (favg - fc));
            double* yo = &SAT(yc, &OB, ob, ycc, ow[0], ow[1], ow[2]);
#pragma omp atomic
            *yo += side ? corr : -corr;
          }
      }
    }
  }
}

/* sum over uncovered valid cells of a*b, and max |a| */
void orc_smooth_dot(const orc_mf* a, const orc_mf* bb, const orc_mf* mask, double* dot, double* amax) {
  const orc_level* L = a->lev;
  double s = 0.0, m = 0.0;
  for (int b = 0; b < L->nboxes; ++b) {
    sbx_t B = sbox(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
          if (SAT(mask, &B, b, 0, i, j, k) == 0.0) continue;
          const double va = SAT(a, &B, b, 0, i, j, k);
          s += va * SAT(bb, &B, b, 0, i, j, k);
          if (fabs(va) > m) m = fabs(va);
        }
  }
  *dot = s;
  *amax = m;
}

/* z = a*x + b*y + c*z on valid cells (covered ones too: they are overwritten by avgdown) */
static void axpbypcz(double a, const orc_mf* x, double bcoef, const orc_mf* y, double c, orc_mf* z) {
  const orc_level* L = z->lev;
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    sbx_t B = sbox(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
          double v = c * SAT(z, &B, b, 0, i, j, k);
          if (x) v += a * SAT(x, &B, b, 0, i, j, k);
          if (y) v += bcoef * SAT(y, &B, b, 0, i, j, k);
          SAT(z, &B, b, 0, i, j, k) = v;
        }
  }
}

/* composite operator y = A x (x is modified: covered cells <- averages, ghosts filled) */
void orc_smooth_apply(int nlev, orc_mf* const* x, orc_mf* const* y, const orc_mf* const* mask, double dt, const int32_t bc[3], int ratio) {
  for (int l = nlev - 1; l > 0; --l) orc_smooth_avgdown(x[l], 0, x[l - 1], 0, ratio);
  for (int l = 0; l < nlev; ++l) {
    orc_fill_boundary(x[l], 0, 1, 1);
    orc_apply_bc(x[l], 0, l ? x[l - 1] : NULL, 0, bc, ratio, -1);
    orc_smooth_apply_level(x[l], 0, y[l], 0, dt);
  }
  for (int l = nlev - 1; l > 0; --l) orc_smooth_reflux(x[l], 0, x[l - 1], 0, y[l - 1], 0, dt, ratio);
  for (int l = 0; l < nlev; ++l) {  /* covered cells carry no equation */
    const orc_level* L = y[l]->lev;
    for (int b = 0; b < L->nboxes; ++b) {
      sbx_t B = sbox(L, b);
      for (int k = B.lo[2]; k <= B.hi[2]; ++k)
        for (int j = B.lo[1]; j <= B.hi[1]; ++j)
          for (int i = B.lo[0]; i <= B.hi[0]; ++i)
            if (SAT(mask[l], &B, b, 0, i, j, k) == 0.0) SAT(y[l], &B, b, 0, i, j, k) = 0.0;
    }
  }
}

static void cdot(int nlev, orc_mf* const* a, orc_mf* const* b, const orc_mf* const* mask, double* dot, double* amax) {
  double s = 0.0, m = 0.0;
  for (int l = 0; l < nlev; ++l) {
    double sl, ml;
    orc_smooth_dot(a[l], b[l], mask[l], &sl, &ml);
    s += sl;
    if (ml > m) m = ml;
  }
  *dot = s;
  *amax = m;
}

/* BiCGStab on the composite operator.  sol (1 comp, ng >= 1) starts from 0 (solution.setVal(0.0), :392).
 * work: 7 vectors per level (r, rhat, p, v, s, t, mask), 1 comp, ng = 1, laid out like sol.
 * Returns the number of iterations (negative: breakdown / not converged); *res = final ||b - A x||_inf / ||b||_inf. */
int orc_smooth_solve(int nlev, orc_mf* const* rhs, int rcomp, orc_mf* const* sol, orc_mf* const* work /* [7][nlev] */, double dt,
                     const int32_t bc[3], int ratio, double tol, int maxiter, double* res) {
  orc_mf* const* r = work;
  orc_mf* const* rh = work + nlev;
  orc_mf* const* p = work + 2 * nlev;
  orc_mf* const* v = work + 3 * nlev;
  orc_mf* const* s = work + 4 * nlev;
  orc_mf* const* t = work + 5 * nlev;
  orc_mf* const* mask = work + 6 * nlev;
  for (int l = 0; l < nlev; ++l) {
    orc_smooth_mask(mask[l], l + 1 < nlev ? sol[l + 1]->lev : NULL, ratio);
    orc_setval(sol[l], 0, 0.0);
    orc_copy(rhs[l], rcomp, r[l], 0, 1, 0);
    orc_copy(rhs[l], rcomp, rh[l], 0, 1, 0);
    orc_setval(p[l], 0, 0.0);
    orc_setval(v[l], 0, 0.0);
  }
  double bnorm, dummy, rho = 1.0, alpha = 1.0, omega = 1.0;
  cdot(nlev, r, r, (const orc_mf* const*)mask, &dummy, &bnorm);
  if (bnorm == 0.0) { *res = 0.0; return 0; }
  int it = 0, status = -1;
  double rnorm = bnorm;
  while (it < maxiter) {
    ++it;
    double rho1;
    cdot(nlev, rh, r, (const orc_mf* const*)mask, &rho1, &dummy);
    if (rho1 == 0.0) { status = -2; break; }
    const double beta = (rho1 / rho) * (alpha / omega);
    for (int l = 0; l < nlev; ++l) {  /* p = r + beta (p - omega v) */
      axpbypcz(-omega * beta, v[l], 0.0, NULL, beta, p[l]);
      axpbypcz(1.0, r[l], 0.0, NULL, 1.0, p[l]);
    }
    orc_smooth_apply(nlev, p, v, (const orc_mf* const*)mask, dt, bc, ratio);
    double rhv;
    cdot(nlev, rh, v, (const orc_mf* const*)mask, &rhv, &dummy);
    if (rhv == 0.0) { status = -3; break; }
    alpha = rho1 / rhv;
    for (int l = 0; l < nlev; ++l) {  /* s = r - alpha v */
      orc_copy(r[l], 0, s[l], 0, 1, 0);
      axpbypcz(-alpha, v[l], 0.0, NULL, 1.0, s[l]);
    }
    double snorm;
    cdot(nlev, s, s, (const orc_mf* const*)mask, &dummy, &snorm);
    if (snorm <= tol * bnorm) {
      for (int l = 0; l < nlev; ++l) axpbypcz(alpha, p[l], 0.0, NULL, 1.0, sol[l]);
      rnorm = snorm;
      status = 0;
      break;
    }
    orc_smooth_apply(nlev, s, t, (const orc_mf* const*)mask, dt, bc, ratio);
    double ts, tt;
    cdot(nlev, t, s, (const orc_mf* const*)mask, &ts, &dummy);
    cdot(nlev, t, t, (const orc_mf* const*)mask, &tt, &dummy);
    if (tt == 0.0) { status = -4; break; }
    omega = ts / tt;
    for (int l = 0; l < nlev; ++l) {
      axpbypcz(alpha, p[l], omega, s[l], 1.0, sol[l]);  /* x += alpha p + omega s */
      orc_copy(s[l], 0, r[l], 0, 1, 0);                 /* r = s - omega t */
      axpbypcz(-omega, t[l], 0.0, NULL, 1.0, r[l]);
    }
    cdot(nlev, r, r, (const orc_mf* const*)mask, &dummy, &rnorm);
    rho = rho1;
    if (rnorm <= tol * bnorm) { status = 0; break; }
    if (omega == 0.0) { status = -5; break; }
  }
  for (int l = nlev - 1; l > 0; --l) orc_smooth_avgdown(sol[l], 0, sol[l - 1], 0, ratio);
  *res = rnorm / bnorm;
  return status == 0 ? it : (status == -1 ? -it : status * 1000 - it);
}
