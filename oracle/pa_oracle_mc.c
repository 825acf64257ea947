/*
 * pa_oracle_mc.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Marching cubes on one FAB, restating isosurface.cpp:415-802 (Polygonise),
 * :257-301 (VI_doIt / VertexInterp), :50-82 (Edge ordering) and the per-FAB
 * loop :1566-1592.  Tables: public-domain data, digests pinned (mc_tables.h).
 */
#include "pa_oracle.h"
#include "mc_tables.h"
#include <math.h>
#include <stdlib.h>

static int32_t g_edge[256];
static int32_t g_tri[256 * 16];
static int g_init = 0;
static void init_tables(void) {
  if (g_init) return;
  for (int i = 0; i < 256; ++i) g_edge[i] = ORC_MC_EDGE_TABLE[i];
  for (int i = 0; i < 256; ++i)
    for (int j = 0; j < 16; ++j) g_tri[16 * i + j] = ORC_MC_TRI_TABLE[i][j];
  g_init = 1;
}
const int32_t* orc_mc_edge_table(void) { init_tables(); return g_edge; }
const int32_t* orc_mc_tri_table(void) { init_tables(); return g_tri; }

/* cube corners p0..p7 (isosurface.cpp:426-433) and edge endpoints in the
 * order the reference passes them to VertexInterp (:762-785) */
static const int CORNER[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
static const int EDGE_A[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3};
static const int EDGE_B[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};

#define EPS_DEF 1.e-15

typedef struct { int64_t key; int64_t slot; } ks_t;
static int ks_cmp(const void* a, const void* b) {
  const ks_t* x = (const ks_t*)a;
  const ks_t* y = (const ks_t*)b;
  return (x->key < y->key) ? -1 : (x->key > y->key);
}

int orc_mc_fab(const double* state, const double* mask, const int32_t slo[3], const int32_t shi[3], int ncomp,
               int isocomp, double isoval, const int32_t llo[3], const int32_t lhi[3], double* verts,
               int32_t* vkeys, int64_t vcap, int32_t* tris, int64_t tcap, int64_t* nv_out, int64_t* nt_out) {
  init_tables();
  const int64_t nx = shi[0] - slo[0] + 1, ny = shi[1] - slo[1] + 1, nz = shi[2] - slo[2] + 1;
  const int64_t ncell = nx * ny * nz;
#define LIN(i, j, k) ((((int64_t)(k) - slo[2]) * ny + ((j) - slo[1])) * nx + ((i) - slo[0]))
  /* edge -> insertion slot; edge id = 3*lin(lower endpoint) + dir  (== std::map<Edge> order) */
  int64_t* slot = (int64_t*)malloc(sizeof(int64_t) * 3 * (size_t)ncell);
  for (int64_t q = 0; q < 3 * ncell; ++q) slot[q] = -1;
  int64_t cap = 1024, nv = 0, nt = 0;
  double* vbuf = (double*)malloc(sizeof(double) * cap * ncomp);
  ks_t* ks = (ks_t*)malloc(sizeof(ks_t) * cap);
  int64_t tcap_l = 1024;
  int64_t* tbuf = (int64_t*)malloc(sizeof(int64_t) * 3 * tcap_l);

  for (int k = llo[2]; k <= lhi[2]; ++k)
    for (int j = llo[1]; j <= lhi[1]; ++j)
      for (int i = llo[0]; i <= lhi[0]; ++i) {
        int64_t cl[8];
        int masked = 0;
        for (int m = 0; m < 8; ++m) {
          cl[m] = LIN(i + CORNER[m][0], j + CORNER[m][1], k + CORNER[m][2]);
          if (mask[cl[m]] < 0) masked = 1;
        }
        if (masked) continue;
        int cubeindex = 0;
        for (int m = 0; m < 8; ++m)
          if (state[(int64_t)isocomp * ncell + cl[m]] < isoval) cubeindex |= (1 << m);
        const int em = g_edge[cubeindex];
        if (em == 0) continue;
        int64_t vl[12];
        for (int e = 0; e < 12; ++e) {
          if (!(em & (1 << e))) continue;
          const int a = EDGE_A[e], b = EDGE_B[e];
          const int64_t la = cl[a], lb = cl[b];
          const int64_t ll = la < lb ? la : lb; /* IntVect '<' (z major) == linear order */
          const int dir = (CORNER[a][0] != CORNER[b][0]) ? 0 : ((CORNER[a][1] != CORNER[b][1]) ? 1 : 2);
          const int64_t eid = 3 * ll + dir;
          if (slot[eid] < 0) {
            if (nv == cap) {
              cap *= 2;
              vbuf = (double*)realloc(vbuf, sizeof(double) * cap * ncomp);
              ks = (ks_t*)realloc(ks, sizeof(ks_t) * cap);
            }
            /* VI_doIt(isoVal, isoComp, p1 = corner a, p2 = corner b) */
            const double v1 = state[(int64_t)isocomp * ncell + la], v2 = state[(int64_t)isocomp * ncell + lb];
            double* r = vbuf + nv * ncomp;
            if (fabs(isoval - v1) < EPS_DEF) {
              for (int c = 0; c < ncomp; ++c) r[c] = state[(int64_t)c * ncell + la];
            } else if (fabs(isoval - v2) < EPS_DEF) {
              for (int c = 0; c < ncomp; ++c) r[c] = state[(int64_t)c * ncell + lb];
            } else if (fabs(v1 - v2) < EPS_DEF) {
              for (int c = 0; c < ncomp; ++c) r[c] = state[(int64_t)c * ncell + la];
            } else {
              const double mu = (isoval - v1) / (v2 - v1);
              for (int c = 0; c < ncomp; ++c) {
                const double a1 = state[(int64_t)c * ncell + la], a2 = state[(int64_t)c * ncell + lb];
                r[c] = a1 + mu * (a2 - a1);
              }
            }
            ks[nv].key = eid;
            ks[nv].slot = nv;
            slot[eid] = nv++;
          }
          vl[e] = slot[eid];
        }
        for (int t = 0; g_tri[16 * cubeindex + t] != -1; t += 3) {
          if (nt == tcap_l) {
            tcap_l *= 2;
            tbuf = (int64_t*)realloc(tbuf, sizeof(int64_t) * 3 * tcap_l);
          }
          tbuf[3 * nt + 0] = vl[g_tri[16 * cubeindex + t]];
          tbuf[3 * nt + 1] = vl[g_tri[16 * cubeindex + t + 1]];
          tbuf[3 * nt + 2] = vl[g_tri[16 * cubeindex + t + 2]];
          ++nt;
        }
      }
  *nv_out = nv;
  *nt_out = nt;
  int rc = 0;
  if (nv > vcap || nt > tcap) rc = -1;
  if (rc == 0) {
    /* vertCache iteration order = key order */
    qsort(ks, (size_t)nv, sizeof(ks_t), ks_cmp);
    int64_t* rank = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nv ? nv : 1));
    for (int64_t q = 0; q < nv; ++q) {
      rank[ks[q].slot] = q;
      for (int c = 0; c < ncomp; ++c) verts[q * ncomp + c] = vbuf[ks[q].slot * ncomp + c];
      const int64_t ll = ks[q].key / 3;
      const int dir = (int)(ks[q].key % 3);
      int l[3];
      l[0] = (int)(ll % nx) + slo[0];
      l[1] = (int)((ll / nx) % ny) + slo[1];
      l[2] = (int)(ll / (nx * ny)) + slo[2];
      for (int d = 0; d < 3; ++d) {
        vkeys[6 * q + d] = l[d];
        vkeys[6 * q + 3 + d] = l[d] + (d == dir);
      }
    }
    for (int64_t t = 0; t < 3 * nt; ++t) tris[t] = (int32_t)rank[tbuf[t]];
    free(rank);
  }
  free(slot);
  free(vbuf);
  free(ks);
  free(tbuf);
  return rc;
#undef LIN
}
