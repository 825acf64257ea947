/*
 * pa_oracle_sdf.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Unsigned distance to a triangle mesh on a regular grid: restatement of
 * Tools/SDFGen/makelevelset3.cpp (called from isosurface.cpp:1625-1626), in the reference's float
 * arithmetic and visiting order:
 *   point_segment_distance   makelevelset3.cpp:4-18
 *   point_triangle_distance  :21-44
 *   check_neighbour / sweep  :46-85   (Gauss-Seidel, 8 directions x 2 passes, :169-178)
 *   make_level_set3          :118-185 (exact band :133-145; the intersection counts :146-165 only
 *                            feed the sign step, which the reference compiles out with #if 0
 *                            (:179-184), so they are not restated)
 * PINNED: this file is checked bit for bit against the reference's own code compiled from
 * /root/reference (oracle/_ref/libsdfgen_ref.so, tests/test_sdf_oracle.py) and against golden
 * vectors that build produced (tests/golden/sdf_*.npz).
 * Build without contraction (-ffp-contract=off): the reference is baseline x86-64 (no FMA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { float v[3]; } v3f;

static float sqrf(float x) { return x * x; }
static v3f sub3(v3f a, v3f b) { v3f r = {{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]}}; return r; }
static float dot3(v3f a, v3f b) {  /* vec.h:328-333 */
  float d = a.v[0] * b.v[0];
  d += a.v[1] * b.v[1];
  d += a.v[2] * b.v[2];
  return d;
}
static float mag2_3(v3f a) {  /* vec.h:199-204 */
  float l = sqrf(a.v[0]);
  l += sqrf(a.v[1]);
  l += sqrf(a.v[2]);
  return l;
}
static float dist3(v3f a, v3f b) {  /* vec.h:211-220 */
  float d = sqrf(a.v[0] - b.v[0]);
  d += sqrf(a.v[1] - b.v[1]);
  d += sqrf(a.v[2] - b.v[2]);
  return sqrtf(d);
}
static v3f scale3(float a, v3f w) { v3f r = {{w.v[0] * a, w.v[1] * a, w.v[2] * a}}; return r; }  /* vec.h:121-125,288-293: w*=a */
static v3f add3(v3f a, v3f b) { v3f r = {{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]}}; return r; }

/* makelevelset3.cpp:4-18 */
static float point_segment_distance(v3f x0, v3f x1, v3f x2) {
  const v3f dx = sub3(x2, x1);
  const double m2 = (double)mag2_3(dx);
  float s12 = (float)((double)dot3(sub3(x2, x0), dx) / m2);
  if (s12 < 0) s12 = 0;
  else if (s12 > 1) s12 = 1;
  return dist3(x0, add3(scale3(s12, x1), scale3(1 - s12, x2)));
}

/* makelevelset3.cpp:21-44 */
static float point_triangle_distance(v3f x0, v3f x1, v3f x2, v3f x3) {
  const v3f x13 = sub3(x1, x3), x23 = sub3(x2, x3), x03 = sub3(x0, x3);
  const float m13 = mag2_3(x13), m23 = mag2_3(x23), d = dot3(x13, x23);
  const float det = m13 * m23 - d * d;
  const float invdet = 1.f / (det < 1e-30f ? 1e-30f : det);  /* std::max(a,b): (a<b)?b:a */
  const float a = dot3(x13, x03), b = dot3(x23, x03);
  const float w23 = invdet * (m23 * a - d * b);
  const float w31 = invdet * (m13 * b - d * a);
  const float w12 = 1 - w23 - w31;
  if (w23 >= 0 && w31 >= 0 && w12 >= 0) {
    return dist3(x0, add3(add3(scale3(w23, x1), scale3(w31, x2)), scale3(w12, x3)));
  } else {
    float p, q;
    if (w23 > 0) { p = point_segment_distance(x0, x1, x2); q = point_segment_distance(x0, x1, x3); }
    else if (w31 > 0) { p = point_segment_distance(x0, x1, x2); q = point_segment_distance(x0, x2, x3); }
    else { p = point_segment_distance(x0, x1, x3); q = point_segment_distance(x0, x2, x3); }
    return (q < p) ? q : p;  /* std::min(p,q) */
  }
}

typedef struct {
  const uint32_t* tri;
  const v3f* x;
  float* phi;
  int32_t* ct;
  int ni, nj, nk;
} sdf_t;
#define IDX(S, i, j, k) ((((int64_t)(k)) * (S)->nj + (j)) * (S)->ni + (i))

/* makelevelset3.cpp:46-58 */
static void check_neighbour(sdf_t* S, v3f gx, int i0, int j0, int k0, int i1, int j1, int k1) {
  const int32_t t = S->ct[IDX(S, i1, j1, k1)];
  if (t >= 0) {
    const float d = point_triangle_distance(gx, S->x[S->tri[3 * t]], S->x[S->tri[3 * t + 1]], S->x[S->tri[3 * t + 2]]);
    if (d < S->phi[IDX(S, i0, j0, k0)]) {
      S->phi[IDX(S, i0, j0, k0)] = d;
      S->ct[IDX(S, i0, j0, k0)] = t;
    }
  }
}

/* makelevelset3.cpp:60-85 */
static void sweep(sdf_t* S, const float origin[3], float dx, int di, int dj, int dk) {
  int i0, i1, j0, j1, k0, k1;
  if (di > 0) { i0 = 1; i1 = S->ni; } else { i0 = S->ni - 2; i1 = -1; }
  if (dj > 0) { j0 = 1; j1 = S->nj; } else { j0 = S->nj - 2; j1 = -1; }
  if (dk > 0) { k0 = 1; k1 = S->nk; } else { k0 = S->nk - 2; k1 = -1; }
  /* a dimension of extent 1 gives i0 == i1 (or i0 = -1 == i1): no iterations, as in the reference */
  for (int k = k0; k != k1; k += dk)
    for (int j = j0; j != j1; j += dj)
      for (int i = i0; i != i1; i += di) {
        const v3f gx = {{i * dx + origin[0], j * dx + origin[1], k * dx + origin[2]}};
        check_neighbour(S, gx, i, j, k, i - di, j, k);
        check_neighbour(S, gx, i, j, k, i, j - dj, k);
        check_neighbour(S, gx, i, j, k, i - di, j - dj, k);
        check_neighbour(S, gx, i, j, k, i, j, k - dk);
        check_neighbour(S, gx, i, j, k, i - di, j, k - dk);
        check_neighbour(S, gx, i, j, k, i, j - dj, k - dk);
        check_neighbour(S, gx, i, j, k, i - di, j - dj, k - dk);
      }
}

static double min3d(double a, double b, double c) { double m = (b < a) ? b : a; return (c < m) ? c : m; }  /* util.h:31-33 */
static double max3d(double a, double b, double c) { double m = (a < b) ? b : a; return (m < c) ? c : m; }  /* util.h:47-49 */
static int clampi(int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); }                       /* util.h:170-175 */

/* makelevelset3.cpp:118-185.  phi: ni*nj*nk floats, i fastest.  closest (optional, may be NULL):
 * the closest-triangle index per grid point after the sweeps (not an output of the reference; kept
 * for debugging the HIP kernels). */
int orc_make_level_set3(int64_t ntri, const uint32_t* tri, int64_t nvert, const float* xv, const float origin[3], float dx, int ni, int nj,
                        int nk, float* phi, int exact_band, int32_t* closest) {
  (void)nvert;
  const int64_t n = (int64_t)ni * nj * nk;
  sdf_t S;
  S.tri = tri; S.x = (const v3f*)xv; S.phi = phi; S.ni = ni; S.nj = nj; S.nk = nk;
  S.ct = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  if (!S.ct) return 1;
  const float far = (ni + nj + nk) * dx;  /* :123 upper bound on distance */
  for (int64_t q = 0; q < n; ++q) { phi[q] = far; S.ct[q] = -1; }
  for (int64_t t = 0; t < ntri; ++t) {
    const uint32_t p = tri[3 * t], q = tri[3 * t + 1], r = tri[3 * t + 2];
    /* :131-133 coordinates in grid to high precision */
    const double fip = ((double)S.x[p].v[0] - origin[0]) / dx, fjp = ((double)S.x[p].v[1] - origin[1]) / dx, fkp = ((double)S.x[p].v[2] - origin[2]) / dx;
    const double fiq = ((double)S.x[q].v[0] - origin[0]) / dx, fjq = ((double)S.x[q].v[1] - origin[1]) / dx, fkq = ((double)S.x[q].v[2] - origin[2]) / dx;
    const double fir = ((double)S.x[r].v[0] - origin[0]) / dx, fjr = ((double)S.x[r].v[1] - origin[1]) / dx, fkr = ((double)S.x[r].v[2] - origin[2]) / dx;
    /* :135-137 */
    const int i0 = clampi((int)min3d(fip, fiq, fir) - exact_band, 0, ni - 1), i1 = clampi((int)max3d(fip, fiq, fir) + exact_band + 1, 0, ni - 1);
    const int j0 = clampi((int)min3d(fjp, fjq, fjr) - exact_band, 0, nj - 1), j1 = clampi((int)max3d(fjp, fjq, fjr) + exact_band + 1, 0, nj - 1);
    const int k0 = clampi((int)min3d(fkp, fkq, fkr) - exact_band, 0, nk - 1), k1 = clampi((int)max3d(fkp, fkq, fkr) + exact_band + 1, 0, nk - 1);
    for (int k = k0; k <= k1; ++k)
      for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
          const v3f gx = {{i * dx + origin[0], j * dx + origin[1], k * dx + origin[2]}};
          const float d = point_triangle_distance(gx, S.x[p], S.x[q], S.x[r]);
          if (d < phi[IDX(&S, i, j, k)]) {
            phi[IDX(&S, i, j, k)] = d;
            S.ct[IDX(&S, i, j, k)] = (int32_t)t;
          }
        }
  }
  for (int pass = 0; pass < 2; ++pass) {  /* :169-178 */
    sweep(&S, origin, dx, +1, +1, +1);
    sweep(&S, origin, dx, -1, -1, -1);
    sweep(&S, origin, dx, +1, +1, -1);
    sweep(&S, origin, dx, -1, -1, +1);
    sweep(&S, origin, dx, +1, -1, +1);
    sweep(&S, origin, dx, -1, +1, -1);
    sweep(&S, origin, dx, +1, -1, -1);
    sweep(&S, origin, dx, -1, +1, +1);
  }
  if (closest)
    for (int64_t q = 0; q < n; ++q) closest[q] = S.ct[q];
  free(S.ct);
  return 0;
}
