/*
 * pa_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See pa_oracle.h for scope and the "parity unpinned" statement.
 *
 * Compile with -O2 -ffp-contract=off: every floating-point expression below is
 * written in the association order of the reference call sites so that a
 * device kernel following the same order is bit-identical.
 */
#include "pa_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ helpers */
typedef struct { int lo[3], hi[3], n[3]; } bx_t;

static inline bx_t get_box(const orc_level* L, int b) {
  bx_t r;
  for (int d = 0; d < 3; ++d) {
    r.lo[d] = L->boxes[6 * b + d];
    r.hi[d] = L->boxes[6 * b + 3 + d];
    r.n[d] = r.hi[d] - r.lo[d] + 1;
  }
  return r;
}

static inline int64_t mf_index(const orc_mf* m, const bx_t* B, int b, int c, int i, int j, int k) {
  const int ng = m->ng;
  const int64_t nx = B->n[0] + 2 * ng, ny = B->n[1] + 2 * ng;
  return m->off[b] + (int64_t)c * m->cstride[b] + ((int64_t)(k - B->lo[2] + ng) * ny + (j - B->lo[1] + ng)) * nx + (i - B->lo[0] + ng);
}
#define AT(m, B, b, c, i, j, k) ((m)->data[mf_index((m), (B), (b), (c), (i), (j), (k))])

/* wrap a cell into the domain along periodic directions.
 * returns 0 if the (wrapped) cell is outside the domain (non-periodic dir). */
static inline int wrap_cell(const orc_level* L, int p[3]) {
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

/* box containing the (already wrapped) cell, or -1 */
static inline int find_box(const orc_level* L, const int p[3], int hint) {
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

/* cell classification used by the masks of MLMG / BndryRegister:
 *   0 covered (valid cell of this level, possibly via a periodic image)
 *   1 not covered (inside the domain, not a valid cell) -> coarse-fine
 *   2 outside the (non-periodic) domain */
static inline int classify(const orc_level* L, int i, int j, int k, int* hint) {
  int p[3] = {i, j, k};
  if (!wrap_cell(L, p)) return 2;
  int b = find_box(L, p, *hint);
  if (b >= 0) { *hint = b; return 0; }
  return 1;
}

void orc_dxinv(const orc_level* L, double dxinv[3]) {
  for (int d = 0; d < 3; ++d) {
    const double dx = (L->prob_hi[d] - L->prob_lo[d]) / (double)(L->domhi[d] - L->domlo[d] + 1);
    dxinv[d] = 1.0 / dx;
  }
}

/* ------------------------------------------------------------ FillBoundary */
void orc_fill_boundary(orc_mf* mf, int comp, int ncomp, int ngf) {
  const orc_level* L = mf->lev;
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    int hint = -1;
    for (int k = B.lo[2] - ngf; k <= B.hi[2] + ngf; ++k)
      for (int j = B.lo[1] - ngf; j <= B.hi[1] + ngf; ++j)
        for (int i = B.lo[0] - ngf; i <= B.hi[0] + ngf; ++i) {
          if (i >= B.lo[0] && i <= B.hi[0] && j >= B.lo[1] && j <= B.hi[1] && k >= B.lo[2] && k <= B.hi[2]) continue;
          int p[3] = {i, j, k};
          if (!wrap_cell(L, p)) continue;
          int s = find_box(L, p, hint);
          if (s < 0) continue;
          hint = s;
          bx_t S = get_box(L, s);
          for (int c = comp; c < comp + ncomp; ++c) AT(mf, &B, b, c, i, j, k) = AT(mf, &S, s, c, p[0], p[1], p[2]);
        }
  }
}

/* ----------------------------------------------------------------- applyBC */
/* amrex::poly_interp_coeff restated: Lagrange weights evaluated in fp */
static void poly_interp_coeff(double xInt, const double* x, int N, double* c) {
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

static inline int coarsen_idx(int i, int r) { return (i < 0) ? -((-i + r - 1) / r) : i / r; }

/* coarse value with periodic wrap; *ok cleared if not found */
static inline double crse_val(const orc_mf* crse, int comp, int ic, int jc, int kc, int* hint, int* ok) {
  int p[3] = {ic, jc, kc};
  if (!wrap_cell(crse->lev, p)) { *ok = 0; return 0.0; }
  int b = find_box(crse->lev, p, *hint);
  if (b < 0) { *ok = 0; return 0.0; }
  *hint = b;
  bx_t B = get_box(crse->lev, b);
  return AT(crse, &B, b, comp, p[0], p[1], p[2]);
}

/* InterpBndryData (order 3) restated: boundary value for the fine ghost cell
 * q (outside a fine box across a face normal to dir), located at the coarse
 * cell centre in the normal direction, interpolated in the two tangential
 * directions from coarse cells.  A tangential coarse neighbour is used only
 * if the fine ghost position shifted by the ratio in that direction is itself
 * "not covered" (mask test of the BndryRegister). */
static double cf_bndry_value(const orc_level* LF, const orc_mf* crse, int ccomp, const int q[3], int dir,
                             int r, int* fhint, int* chint, int* ok) {
  const int qc[3] = {coarsen_idx(q[0], r), coarsen_idx(q[1], r), coarsen_idx(q[2], r)};
  int tdir[2], nt = 0;
  for (int d = 0; d < 3; ++d)
    if (d != dir) tdir[nt++] = d;
  double b = 0.0;
  double xi[2];
  for (int t = 0; t < 2; ++t) {
    const int td = tdir[t];
    int m1[3] = {q[0], q[1], q[2]}, p1[3] = {q[0], q[1], q[2]}, m2[3] = {q[0], q[1], q[2]}, p2[3] = {q[0], q[1], q[2]};
    m1[td] -= r; p1[td] += r; m2[td] -= 2 * r; p2[td] += 2 * r;
    const int okm1 = classify(LF, m1[0], m1[1], m1[2], fhint) == 1;
    const int okp1 = classify(LF, p1[0], p1[1], p1[2], fhint) == 1;
    int lo = okm1 ? -1 : 0, hi = okp1 ? 1 : 0;
    if (lo == -1 && hi == 0 && classify(LF, m2[0], m2[1], m2[2], fhint) == 1) lo = -2;
    else if (hi == 1 && lo == 0 && classify(LF, p2[0], p2[1], p2[2], fhint) == 1) hi = 2;
    const int N = hi - lo + 1;
    double x[3], c[3];
    for (int m = 0; m < N; ++m) x[m] = (double)(lo + m);
    const double xInt = -0.5 + ((double)(q[td] - qc[td] * r) + 0.5) / (double)r;
    xi[t] = xInt;
    poly_interp_coeff(xInt, x, N, c);
    for (int m = 0; m < N; ++m) {
      int cc[3] = {qc[0], qc[1], qc[2]};
      cc[td] += lo + m;
      b += c[m] * crse_val(crse, ccomp, cc[0], cc[1], cc[2], chint, ok);
    }
  }
  b -= crse_val(crse, ccomp, qc[0], qc[1], qc[2], chint, ok);
  /* cross term, only if all four diagonal fine positions are not covered */
  {
    const int t0 = tdir[0], t1 = tdir[1];
    int all = 1;
    for (int s1 = -1; s1 <= 1 && all; s1 += 2)
      for (int s0 = -1; s0 <= 1; s0 += 2) {
        int p[3] = {q[0], q[1], q[2]};
        p[t0] += s0 * r; p[t1] += s1 * r;
        if (classify(LF, p[0], p[1], p[2], fhint) != 1) { all = 0; break; }
      }
    if (all) {
      int cpp[3] = {qc[0], qc[1], qc[2]}, cmp[3] = {qc[0], qc[1], qc[2]}, cmm[3] = {qc[0], qc[1], qc[2]},
          cpm[3] = {qc[0], qc[1], qc[2]};
      cpp[t0] += 1; cpp[t1] += 1;
      cmp[t0] -= 1; cmp[t1] += 1;
      cmm[t0] -= 1; cmm[t1] -= 1;
      cpm[t0] += 1; cpm[t1] -= 1;
      const double vpp = crse_val(crse, ccomp, cpp[0], cpp[1], cpp[2], chint, ok);
      const double vmp = crse_val(crse, ccomp, cmp[0], cmp[1], cmp[2], chint, ok);
      const double vmm = crse_val(crse, ccomp, cmm[0], cmm[1], cmm[2], chint, ok);
      const double vpm = crse_val(crse, ccomp, cpm[0], cpm[1], cpm[2], chint, ok);
      b += ((xi[0] * xi[1]) * 0.25) * (((vpp - vmp) + vmm) - vpm);
    }
  }
  return b;
}

int orc_apply_bc(orc_mf* fine, int comp, const orc_mf* crse, int ccomp, const int32_t bc[3], int ratio,
                 int only_dir) {
  const orc_level* L = fine->lev;
  int nbad = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : nbad)
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    int fhint = b, chint = -1;
    for (int dir = 0; dir < 3; ++dir) {
      if (only_dir >= 0 && dir != only_dir) continue;
      const int t0 = (dir == 0) ? 1 : 0, t1 = (dir == 2) ? 1 : 2;
      /* normal cubic through {bc point at -ratio/2, cells at .5,1.5,2.5} evaluated at -.5 */
      const int NX = (B.n[dir] + 1 < 4) ? B.n[dir] + 1 : 4;
      double x[4] = {-0.5 * (double)ratio, 0.5, 1.5, 2.5}, coef[4];
      poly_interp_coeff(-0.5, x, NX, coef);
      for (int side = 0; side < 2; ++side) {
        const int s = side ? -1 : 1; /* direction from the ghost cell into the box */
        const int gq = side ? B.hi[dir] + 1 : B.lo[dir] - 1;
        for (int b1 = B.lo[t1]; b1 <= B.hi[t1]; ++b1)
          for (int a0 = B.lo[t0]; a0 <= B.hi[t0]; ++a0) {
            int q[3];
            q[dir] = gq; q[t0] = a0; q[t1] = b1;
            const int cls = classify(L, q[0], q[1], q[2], &fhint);
            if (cls == 0) continue; /* filled by FillBoundary */
            int in[3] = {q[0], q[1], q[2]};
            in[dir] += s;
            double* g = &AT(fine, &B, b, comp, q[0], q[1], q[2]);
            if (cls == 2) {
              const double v = AT(fine, &B, b, comp, in[0], in[1], in[2]);
              *g = (bc[dir] == ORC_BC_REFLECT_ODD) ? -v : v;
            } else {
              if (!crse) { ++nbad; continue; }
              int ok = 1;
              const double bv = cf_bndry_value(L, crse, ccomp, q, dir, ratio, &fhint, &chint, &ok);
              if (!ok) ++nbad;
              double tmp = 0.0;
              for (int m = 1; m < NX; ++m) {
                int pc[3] = {q[0], q[1], q[2]};
                pc[dir] += s * m;
                tmp += AT(fine, &B, b, comp, pc[0], pc[1], pc[2]) * coef[m];
              }
              *g = tmp;
              *g += bv * coef[0];
            }
          }
      }
    }
  }
  return nbad;
}

/* -------------------------------------------------------------------- grad */
void orc_grad_multipass(const orc_mf* phi, int comp, orc_mf* out, int ocomp) {
  const orc_level* L = phi->lev;
  double dxinv[3];
  orc_dxinv(L, dxinv);
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    const int nx = B.n[0], ny = B.n[1], nz = B.n[2];
    for (int d = 0; d < 3; ++d) {
      /* face array, nodal in d */
      const int fx = nx + (d == 0), fy = ny + (d == 1), fz = nz + (d == 2);
      double* f = (double*)malloc(sizeof(double) * (size_t)fx * fy * fz);
      const int e[3] = {d == 0, d == 1, d == 2};
      /* mlpoisson_flux_{x,y,z}: f = dxinv*(sol(i)-sol(i-1)) */
      for (int k = 0; k < fz; ++k)
        for (int j = 0; j < fy; ++j)
          for (int i = 0; i < fx; ++i) {
            const int I = B.lo[0] + i, J = B.lo[1] + j, K = B.lo[2] + k;
            f[((size_t)k * fy + j) * fx + i] =
                dxinv[d] * (AT(phi, &B, b, comp, I, J, K) - AT(phi, &B, b, comp, I - e[0], J - e[1], K - e[2]));
          }
      /* getFluxes: mult(1/bscalar) with bscalar = -1 */
      for (size_t q = 0; q < (size_t)fx * fy * fz; ++q) f[q] *= -1.0;
      /* average_face_to_cellcenter */
      for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
          for (int i = 0; i < nx; ++i)
            AT(out, &B, b, ocomp + d, B.lo[0] + i, B.lo[1] + j, B.lo[2] + k) =
                0.5 * (f[((size_t)k * fy + j) * fx + i] + f[((size_t)(k + e[2]) * fy + (j + e[1])) * fx + (i + e[0])]);
      free(f);
      /* gradAlias.mult(-1.0) */
      for (int k = B.lo[2]; k <= B.hi[2]; ++k)
        for (int j = B.lo[1]; j <= B.hi[1]; ++j)
          for (int i = B.lo[0]; i <= B.hi[0]; ++i) AT(out, &B, b, ocomp + d, i, j, k) *= -1.0;
    }
    /* grad.cpp:228-234 */
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
          const double gx = AT(out, &B, b, ocomp, i, j, k), gy = AT(out, &B, b, ocomp + 1, i, j, k),
                       gz = AT(out, &B, b, ocomp + 2, i, j, k);
          AT(out, &B, b, ocomp + 3, i, j, k) = sqrt(gx * gx + gy * gy + gz * gz);
        }
  }
}

static inline double cdiff(double dxinv, double m, double c, double p) {
  /* -(0.5*((-(dxinv*(c-m))) + (-(dxinv*(p-c))))) : same roundings and zero signs as the multipass */
  const double fl = -(dxinv * (c - m));
  const double fh = -(dxinv * (p - c));
  return -(0.5 * (fl + fh));
}

void orc_grad_fused(const orc_mf* phi, int comp, orc_mf* out, int ocomp, int with_mag) {
  const orc_level* L = phi->lev;
  double dxinv[3];
  orc_dxinv(L, dxinv);
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
          const double c = AT(phi, &B, b, comp, i, j, k);
          const double gx = cdiff(dxinv[0], AT(phi, &B, b, comp, i - 1, j, k), c, AT(phi, &B, b, comp, i + 1, j, k));
          const double gy = cdiff(dxinv[1], AT(phi, &B, b, comp, i, j - 1, k), c, AT(phi, &B, b, comp, i, j + 1, k));
          const double gz = cdiff(dxinv[2], AT(phi, &B, b, comp, i, j, k - 1), c, AT(phi, &B, b, comp, i, j, k + 1));
          AT(out, &B, b, ocomp, i, j, k) = gx;
          AT(out, &B, b, ocomp + 1, i, j, k) = gy;
          AT(out, &B, b, ocomp + 2, i, j, k) = gz;
          if (with_mag) AT(out, &B, b, ocomp + 3, i, j, k) = sqrt(gx * gx + gy * gy + gz * gz);
        }
  }
}

/* --------------------------------------------------------------- curvature */
void orc_minmax(const orc_mf* s, int comp, double* mn, double* mx) {
  const orc_level* L = s->lev;
  double lo = 1e300, hi = -1e300;
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
          const double v = AT(s, &B, b, comp, i, j, k);
          if (v < lo) lo = v;
          if (v > hi) hi = v;
        }
  }
  *mn = lo;
  *mx = hi;
}

void orc_progress(const orc_mf* s, int comp, double pmin, double pmax, orc_mf* c, int ccomp) {
  const orc_level* L = s->lev;
  const double invdenom = 1.0 / (pmax - pmin);
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i)
          AT(c, &B, b, ccomp, i, j, k) = (AT(s, &B, b, comp, i, j, k) - pmin) * invdenom;
  }
}

void orc_normal(const orc_mf* G, int gcomp, orc_mf* normgrad, int ngcomp, orc_mf* n, int ncomp0) {
  const orc_level* L = G->lev;
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
          const double cx = AT(G, &B, b, gcomp, i, j, k), cy = AT(G, &B, b, gcomp + 1, i, j, k),
                       cz = AT(G, &B, b, gcomp + 2, i, j, k);
          /* pow(x,2.0) == x*x (quirk Q12) */
          const double sn = sqrt(cx * cx + cy * cy + cz * cz);
          double ng = (1e-14 < sn) ? sn : 1e-14; /* std::max(1e-14, sn) */
          ng = -ng;
          AT(normgrad, &B, b, ngcomp, i, j, k) = ng;
          AT(n, &B, b, ncomp0, i, j, k) = cx / ng;
          AT(n, &B, b, ncomp0 + 1, i, j, k) = cy / ng;
          AT(n, &B, b, ncomp0 + 2, i, j, k) = cz / ng;
        }
  }
}

void orc_div_accum(const orc_mf* nd, int comp, int dir, orc_mf* curv, int kcomp) {
  const orc_level* L = nd->lev;
  double dxinv[3];
  orc_dxinv(L, dxinv);
  const int e[3] = {dir == 0, dir == 1, dir == 2};
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
          const double d = cdiff(dxinv[dir], AT(nd, &B, b, comp, i - e[0], j - e[1], k - e[2]),
                                 AT(nd, &B, b, comp, i, j, k), AT(nd, &B, b, comp, i + e[0], j + e[1], k + e[2]));
          AT(curv, &B, b, kcomp, i, j, k) += d;
        }
  }
}

/* -------------------------------------------------- fused single-sweep CPU variant (BASELINE.md section 3 ii)
 * The grad -> curvature path of grad.cpp:211-236 + curvature.cpp:457-546 as ONE sweep per level instead of one pass per AMReX
 * call: every valid cell forms gx, gy, gz, |g| from phi, its normal from c, and its curvature from the normals of its six
 * face neighbours, which it rebuilds from c (radius 2) with the very operations orc_grad_fused / orc_normal / orc_div_accum
 * use -- so the bits equal the pass-by-pass result wherever the neighbour's normal IS a function of c: every neighbour that
 * is a valid cell of the level.  Behind a coarse-fine or physical face the reference's ghost normal is the boundary
 * condition applied to n itself; orc_curv_first_layer recomputes the first layer of cells of every box from the
 * stored normals once their ghost cells are filled (FillBoundary + applyBC on n, surface work).  phi: >= 1 resolved ghost
 * layer; c: FillBoundary(2) + applyBC; nmf: 3 components, >= 1 ghost layer, valid cells written here.
 * Used by bench.py's cpu_baseline ("fused" variant, results compared bit for bit with the multipass variant's). */
static inline void grad3_at(const orc_mf* c, const bx_t* B, int b, int comp, const double dxinv[3], int i, int j, int k, double g[3]) {
  const double v = AT(c, B, b, comp, i, j, k);
  g[0] = cdiff(dxinv[0], AT(c, B, b, comp, i - 1, j, k), v, AT(c, B, b, comp, i + 1, j, k));
  g[1] = cdiff(dxinv[1], AT(c, B, b, comp, i, j - 1, k), v, AT(c, B, b, comp, i, j + 1, k));
  g[2] = cdiff(dxinv[2], AT(c, B, b, comp, i, j, k - 1), v, AT(c, B, b, comp, i, j, k + 1));
}
static inline double normgrad_of(const double g[3]) {
  const double sn = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
  const double ng = (1e-14 < sn) ? sn : 1e-14;
  return -ng;
}
void orc_gradcurv_fused(const orc_mf* phi, int pcomp, const orc_mf* c, int ccomp, orc_mf* gout, int gcomp, orc_mf* nmf, int ncomp0, orc_mf* K,
                        int kcomp) {
  const orc_level* L = phi->lev;
  double dxinv[3];
  orc_dxinv(L, dxinv);
#pragma omp parallel
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
#pragma omp for schedule(dynamic, 1) nowait
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
          double g[3], G[3], Gn[3];
          grad3_at(phi, &B, b, pcomp, dxinv, i, j, k, g);
          AT(gout, &B, b, gcomp, i, j, k) = g[0];
          AT(gout, &B, b, gcomp + 1, i, j, k) = g[1];
          AT(gout, &B, b, gcomp + 2, i, j, k) = g[2];
          AT(gout, &B, b, gcomp + 3, i, j, k) = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
          grad3_at(c, &B, b, ccomp, dxinv, i, j, k, G);
          const double ng = normgrad_of(G);
          const double n0 = G[0] / ng, n1 = G[1] / ng, n2 = G[2] / ng;
          AT(nmf, &B, b, ncomp0, i, j, k) = n0;
          AT(nmf, &B, b, ncomp0 + 1, i, j, k) = n1;
          AT(nmf, &B, b, ncomp0 + 2, i, j, k) = n2;
          /* n_d of the two neighbours in direction d, each from its own gradient of c */
          double nm[3], np[3];
          for (int d = 0; d < 3; ++d) {
            const int e0 = d == 0, e1 = d == 1, e2 = d == 2;
            grad3_at(c, &B, b, ccomp, dxinv, i - e0, j - e1, k - e2, Gn);
            nm[d] = Gn[d] / normgrad_of(Gn);
            grad3_at(c, &B, b, ccomp, dxinv, i + e0, j + e1, k + e2, Gn);
            np[d] = Gn[d] / normgrad_of(Gn);
          }
          double acc = 0.0;
          acc += cdiff(dxinv[0], nm[0], n0, np[0]);
          acc += cdiff(dxinv[1], nm[1], n1, np[1]);
          acc += cdiff(dxinv[2], nm[2], n2, np[2]);
          AT(K, &B, b, kcomp, i, j, k) = acc * 0.5;
        }
  }
}
/* K of the first layer of cells behind every face of every box from the STORED normals (ghost cells filled by the caller:
 * FillBoundary, then applyBC on component d in direction d), the sums of orc_div_accum + orc_mult(0.5) */
void orc_curv_first_layer(const orc_mf* nmf, int ncomp0, orc_mf* K, int kcomp) {
  const orc_level* L = nmf->lev;
  double dxinv[3];
  orc_dxinv(L, dxinv);
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j) {
        const int edge_row = (k == B.lo[2] || k == B.hi[2] || j == B.lo[1] || j == B.hi[1]);
        for (int i = B.lo[0]; i <= B.hi[0]; i += (edge_row || i == B.hi[0] || B.hi[0] == B.lo[0]) ? 1 : (B.hi[0] - B.lo[0])) {
          double acc = 0.0;
          acc += cdiff(dxinv[0], AT(nmf, &B, b, ncomp0, i - 1, j, k), AT(nmf, &B, b, ncomp0, i, j, k), AT(nmf, &B, b, ncomp0, i + 1, j, k));
          acc += cdiff(dxinv[1], AT(nmf, &B, b, ncomp0 + 1, i, j - 1, k), AT(nmf, &B, b, ncomp0 + 1, i, j, k), AT(nmf, &B, b, ncomp0 + 1, i, j + 1, k));
          acc += cdiff(dxinv[2], AT(nmf, &B, b, ncomp0 + 2, i, j, k - 1), AT(nmf, &B, b, ncomp0 + 2, i, j, k), AT(nmf, &B, b, ncomp0 + 2, i, j, k + 1));
          AT(K, &B, b, kcomp, i, j, k) = acc * 0.5;
        }
      }
  }
}

void orc_setval(orc_mf* mf, int comp, double v) {
  const orc_level* L = mf->lev;
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) AT(mf, &B, b, comp, i, j, k) = v;
  }
}

void orc_mult(orc_mf* mf, int comp, double v) {
  const orc_level* L = mf->lev;
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) AT(mf, &B, b, comp, i, j, k) *= v;
  }
}

void orc_threshold(const orc_mf* c, int ccomp, double thr, orc_mf* K, int kcomp, orc_mf* n, int ncomp0) {
  const orc_level* L = c->lev;
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
          const double p = AT(c, &B, b, ccomp, i, j, k);
          if (p < thr || p > 1.0 - thr) {
            AT(K, &B, b, kcomp, i, j, k) = 0.0;
            AT(n, &B, b, ncomp0, i, j, k) = 0.0;
            AT(n, &B, b, ncomp0 + 1, i, j, k) = 0.0;
            AT(n, &B, b, ncomp0 + 2, i, j, k) = 0.0;
          }
        }
  }
}

void orc_copy(const orc_mf* src, int scomp, orc_mf* dst, int dcomp, int ncomp, int ng) {
  const orc_level* L = src->lev;
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int c = 0; c < ncomp; ++c)
      for (int k = B.lo[2] - ng; k <= B.hi[2] + ng; ++k)
        for (int j = B.lo[1] - ng; j <= B.hi[1] + ng; ++j)
          for (int i = B.lo[0] - ng; i <= B.hi[0] + ng; ++i)
            AT(dst, &B, b, dcomp + c, i, j, k) = AT(src, &B, b, scomp + c, i, j, k);
  }
}

void orc_gauss_curv(const orc_mf* H, const orc_mf* G, const orc_mf* normgrad, const orc_mf* c, int ccomp,
                    double thr, orc_mf* Kg, int kcomp) {
  const orc_level* L = H->lev;
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
#define HX(n) AT(H, &B, b, 0 + (n), i, j, k)
#define HY(n) AT(H, &B, b, 3 + (n), i, j, k)
#define HZ(n) AT(H, &B, b, 6 + (n), i, j, k)
          /* curvature.cpp:630-638 */
          const double ax0 = HY(1) * HZ(2) - HZ(1) * HY(2);
          const double ay0 = HY(2) * HZ(0) - HZ(2) * HY(0);
          const double az0 = HY(0) * HZ(1) - HZ(0) * HY(1);
          const double ax1 = HX(2) * HZ(1) - HZ(2) * HX(1);
          const double ay1 = HX(0) * HZ(2) - HZ(0) * HX(2);
          const double az1 = HX(1) * HZ(0) - HZ(1) * HX(0);
          const double ax2 = HX(1) * HY(2) - HY(1) * HX(2);
          const double ay2 = HX(2) * HY(0) - HY(2) * HX(0);
          const double az2 = HX(0) * HY(1) - HY(0) * HX(1);
#undef HX
#undef HY
#undef HZ
          const double cx = AT(G, &B, b, 0, i, j, k), cy = AT(G, &B, b, 1, i, j, k), cz = AT(G, &B, b, 2, i, j, k);
          const double gn = AT(normgrad, &B, b, 0, i, j, k);
          /* curvature.cpp:659-668 ; pow(x,4.0) == (x*x)*(x*x) (quirk Q12) */
          double kg = (cx * (ax0 * cx + ax1 * cy + ax2 * cz) + cy * (ay0 * cx + ay1 * cy + ay2 * cz) +
                       cz * (az0 * cx + az1 * cy + az2 * cz)) /
                      ((gn * gn) * (gn * gn));
          if (thr >= 0.0) {
            const double p = AT(c, &B, b, ccomp, i, j, k);
            if (p < thr || p > 1.0 - thr) kg = 0.0;
          }
          AT(Kg, &B, b, kcomp, i, j, k) = kg;
        }
  }
}

void orc_strain_rate(const orc_mf* gradU, const orc_mf* n, orc_mf* sr, int comp) {
  const orc_level* L = gradU->lev;
  (void)n; /* first assignment (-nn:grad u) is overwritten in the reference: quirk Q3 */
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i)
          AT(sr, &B, b, comp, i, j, k) =
              +AT(gradU, &B, b, 0, i, j, k) + AT(gradU, &B, b, 4, i, j, k) + AT(gradU, &B, b, 8, i, j, k);
  }
}

void orc_vel_normal(const orc_mf* u, int ucomp, const orc_mf* n, const orc_mf* c, int ccomp, double thr,
                    orc_mf* out, int ocomp) {
  const orc_level* L = u->lev;
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int k = B.lo[2]; k <= B.hi[2]; ++k)
      for (int j = B.lo[1]; j <= B.hi[1]; ++j)
        for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
          double v = +AT(u, &B, b, ucomp, i, j, k) * AT(n, &B, b, 0, i, j, k) +
                     AT(u, &B, b, ucomp + 1, i, j, k) * AT(n, &B, b, 1, i, j, k) +
                     AT(u, &B, b, ucomp + 2, i, j, k) * AT(n, &B, b, 2, i, j, k);
          if (thr >= 0.0) {
            const double p = AT(c, &B, b, ccomp, i, j, k);
            if (p < thr || p > 1.0 - thr) v = 0.0;
          }
          AT(out, &B, b, ocomp, i, j, k) = v;
        }
  }
}

/* --------------------------------------------------------------- filterPlt */
int orc_box_filter_weights(int fgr, double* w) {
  /* PelePhysics Filter::set_box_weights restated: ngrow = fgr/2,
   * 2*ngrow+1 weights 1/fgr, the two end weights halved */
  const int ng = fgr / 2;
  const int nw = 2 * ng + 1;
  for (int i = 0; i < nw; ++i) w[i] = 1.0 / (double)fgr;
  if (nw > 1) {
    w[0] = 0.5 * w[0];
    w[nw - 1] = w[0];
  }
  return ng;
}

/* PelePhysics Filter types with closed-form weights (filterPlt.cpp:80) [RECALLED / re-derived: PelePhysics is
 * absent from the reference tree].  3-point approximations (types 3 = box, 7 = Gaussian: identical): the symmetric stencil
 * {a, 1 - 2a, a} whose second moment 2a equals that of the filter, fgr^2/12 -> a = fgr^2/24.  5-point approximations
 * {a2, a1, a0, a1, a2}: second moment 2(a1 + 4 a2) = fgr^2/12 and fourth moment 2(a1 + 16 a2) = m4 with m4 = fgr^4/80
 * (box, type 4) or 3 (fgr^2/12)^2 = fgr^4/48 (Gaussian, type 8), a0 = 1 - 2 a1 - 2 a2. */
int orc_filter_weights(int type, int fgr, double* w) {
  const double f2 = (double)fgr * (double)fgr, f4 = f2 * f2;
  if (fgr < 1) return -1;
  switch (type) {
    case 0: w[0] = 1.0; return 0;
    case 1: return orc_box_filter_weights(fgr, w);
    case 2: { /* Gaussian, UNVERIFIED against PelePhysics: exp(-6 i^2 / fgr^2), i = -ng..ng, ng = ceil(4 fgr / sqrt 12) >= 1, sum 1 */
      int ng = (int)ceil(4.0 * (double)fgr / sqrt(12.0));
      double sum = 0.0;
      if (ng < 1) ng = 1;
      if (ng > 16) return -1;
      for (int i = -ng; i <= ng; ++i) { w[i + ng] = exp(-6.0 * (double)(i * i) / f2); sum += w[i + ng]; }
      for (int i = 0; i <= 2 * ng; ++i) w[i] = w[i] / sum;
      return ng;
    }
    case 3: case 7:
      w[0] = f2 / 24.0; w[2] = f2 / 24.0; w[1] = (12.0 - f2) / 12.0;
      return 1;
    case 4:
      w[0] = w[4] = (3.0 * f4 - 20.0 * f2) / 5760.0;
      w[1] = w[3] = (80.0 * f2 - 3.0 * f4) / 1440.0;
      w[2] = (3.0 * f4 - 100.0 * f2 + 960.0) / 960.0;
      return 2;
    case 8:
      w[0] = w[4] = (f4 - 4.0 * f2) / 1152.0;
      w[1] = w[3] = (16.0 * f2 - f4) / 288.0;
      w[2] = (f4 - 20.0 * f2 + 192.0) / 192.0;
      return 2;
    default: return -1;
  }
}

void orc_apply_filter(const orc_mf* in, orc_mf* out, int scomp, int ncomp, int ngf, const double* w) {
  const orc_level* L = in->lev;
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int c = scomp; c < scomp + ncomp; ++c)
      for (int k = B.lo[2]; k <= B.hi[2]; ++k)
        for (int j = B.lo[1]; j <= B.hi[1]; ++j)
          for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
            double acc = 0.0; /* out.setVal(0) then += in n,m,l order */
            for (int n = -ngf; n <= ngf; ++n)
              for (int m = -ngf; m <= ngf; ++m)
                for (int l = -ngf; l <= ngf; ++l)
                  acc += w[l + ngf] * w[m + ngf] * w[n + ngf] * AT(in, &B, b, c, i + l, j + m, k + n);
            AT(out, &B, b, c, i, j, k) = acc;
          }
  }
}

void orc_foextrap(orc_mf* mf, int comp, int ncomp, int ngf) {
  const orc_level* L = mf->lev;
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    for (int c = comp; c < comp + ncomp; ++c)
      for (int k = B.lo[2] - ngf; k <= B.hi[2] + ngf; ++k)
        for (int j = B.lo[1] - ngf; j <= B.hi[1] + ngf; ++j)
          for (int i = B.lo[0] - ngf; i <= B.hi[0] + ngf; ++i) {
            int p[3] = {i, j, k}, out = 0;
            for (int d = 0; d < 3; ++d)
              if (!L->is_per[d]) {
                if (p[d] < L->domlo[d]) { p[d] = L->domlo[d]; out = 1; }
                if (p[d] > L->domhi[d]) { p[d] = L->domhi[d]; out = 1; }
              }
            if (!out) continue;
            /* nearest interior cell; it is inside the grown box of this fab and
             * already filled (valid, same-level ghost or periodic image) */
            AT(mf, &B, b, c, i, j, k) = AT(mf, &B, b, c, p[0], p[1], p[2]);
          }
  }
}

/* coarse value incl. the coarse multifab's own filled ghost cells: look for a
 * coarse box whose grown region holds the cell (valid box preferred). */
static double crse_val_g(const orc_mf* crse, int comp, const int p0[3], int* ok) {
  const orc_level* L = crse->lev;
  int p[3] = {p0[0], p0[1], p0[2]};
  int inside = 1;
  for (int d = 0; d < 3; ++d)
    if (p[d] < L->domlo[d] || p[d] > L->domhi[d]) inside = 0;
  if (inside || wrap_cell(L, p)) {
    int b = find_box(L, p, -1);
    if (b >= 0) {
      bx_t B = get_box(L, b);
      return AT(crse, &B, b, comp, p[0], p[1], p[2]);
    }
  }
  /* outside a non-periodic wall: use the ghost cell of a coarse box that holds it */
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    int in = 1;
    for (int d = 0; d < 3; ++d)
      if (p0[d] < B.lo[d] - crse->ng || p0[d] > B.hi[d] + crse->ng) in = 0;
    if (in) return AT(crse, &B, b, comp, p0[0], p0[1], p0[2]);
  }
  *ok = 0;
  return 0.0;
}

/* coarse value seen by the interpolater: the cell clamped into the domain along non-periodic
 * directions (= the foextrap-filled coarse ghost cell), then the valid coarse cell (periodic wrap) */
static double cu_clamped(const orc_mf* crse, const orc_level* LC, int comp, int i, int j, int k, int* ok) {
  int p[3] = {i, j, k};
  for (int d = 0; d < 3; ++d)
    if (!LC->is_per[d]) {
      if (p[d] < LC->domlo[d]) p[d] = LC->domlo[d];
      if (p[d] > LC->domhi[d]) p[d] = LC->domhi[d];
    }
  return crse_val_g(crse, comp, p, ok);
}

int orc_fillpatch_two_levels(orc_mf* fine, const orc_mf* crse, int comp, int ncomp, int ngf, int r,
                             int interp_type) {
  const orc_level* L = fine->lev;
  const orc_level* LC = crse->lev;
  int nbad = 0;
  for (int b = 0; b < L->nboxes; ++b) {
    bx_t B = get_box(L, b);
    int hint = b;
    for (int k = B.lo[2] - ngf; k <= B.hi[2] + ngf; ++k)
      for (int j = B.lo[1] - ngf; j <= B.hi[1] + ngf; ++j)
        for (int i = B.lo[0] - ngf; i <= B.hi[0] + ngf; ++i) {
          if (i >= B.lo[0] && i <= B.hi[0] && j >= B.lo[1] && j <= B.hi[1] && k >= B.lo[2] && k <= B.hi[2]) continue;
          const int cls = classify(L, i, j, k, &hint);
          if (cls != 1) continue; /* covered: FillBoundary; outside: foextrap afterwards */
          const int q[3] = {i, j, k};
          const int qc[3] = {coarsen_idx(i, r), coarsen_idx(j, r), coarsen_idx(k, r)};
          for (int c = comp; c < comp + ncomp; ++c) {
            int ok = 1;
            const double u0 = crse_val_g(crse, c, qc, &ok);
            double val = u0;
            if (interp_type == 1) {
              /* mf_cell_cons_lin_interp_mcslope + mf_cell_cons_lin_interp restated (AMReX
               * AMReX_MFInterp_3D_C.H, [RECALLED]).  The coarse data FillPatchTwoLevels
               * interpolates from is the coarse level's valid cells + periodic images + the coarse
               * physical BC, which filterPlt sets to foextrap (filterPlt.cpp:164-173): a coarse
               * neighbour outside a non-periodic wall holds the value of the nearest cell inside
               * the domain, dimension by dimension.  mf_compute_slopes only departs from the
               * plain central difference for ext_dir / hoextrap, never for foextrap. */
#define CU(dx_, dy_, dz_) cu_clamped(crse, LC, c, qc[0] + (dx_), qc[1] + (dy_), qc[2] + (dz_), &ok)
              double sl[3];
              for (int d = 0; d < 3; ++d) {
                const double um = CU(-(d == 0), -(d == 1), -(d == 2)), up = CU(d == 0, d == 1, d == 2);
                const double dc = 0.5 * (up - um);
                const double df = 2.0 * (up - u0), db = 2.0 * (u0 - um);
                double sx = (df * db >= 0.0) ? fmin(fabs(df), fabs(db)) : 0.0;
                sx = copysign(1.0, dc) * fmin(sx, fabs(dc));
                sl[d] = sx;
              }
              double alpha = 1.0;
              if (sl[0] != 0.0 || sl[1] != 0.0 || sl[2] != 0.0) {
                const double dumax = fabs(sl[0]) * (double)(r - 1) / (double)(2 * r) + fabs(sl[1]) * (double)(r - 1) / (double)(2 * r) +
                                     fabs(sl[2]) * (double)(r - 1) / (double)(2 * r);
                double umax = u0, umin = u0;
                for (int dz = -1; dz <= 1; ++dz)
                  for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                      const double v = CU(dx, dy, dz);
                      umin = (v < umin) ? v : umin;
                      umax = (v > umax) ? v : umax;
                    }
                if (dumax * alpha > (umax - u0)) alpha = (umax - u0) / dumax;
                if (dumax * alpha > (u0 - umin)) alpha = (u0 - umin) / dumax;
              }
#undef CU
              /* slope(ns) = sx*alpha etc.; fine = crse + xoff*slope_x + yoff*slope_y + zoff*slope_z */
              double acc = u0;
              for (int d = 0; d < 3; ++d) {
                const double xoff = ((double)(q[d] - qc[d] * r) + 0.5) / (double)r - 0.5;
                acc += xoff * (sl[d] * alpha);
              }
              val = acc;
            }
            if (!ok) ++nbad;
            AT(fine, &B, b, c, i, j, k) = val;
          }
        }
  }
  return nbad;
}
