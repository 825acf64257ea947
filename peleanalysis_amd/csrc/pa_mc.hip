// pa_mc.hip -- marching cubes on one FAB for gfx950 (isosurface.cpp:1566-1592 + Polygonise
// :415-802 + VertexInterp/VI_doIt :257-301), with order-preserving compaction so that the output
// equals what the reference's serial loop + std::map<Edge,Point> produce:
//   vertices  in vertCache order  = (linear index of the edge's lower endpoint in the FAB, dir x<y<z)
//   triangles in cube traversal order (x fastest), triTable order inside a cube.
// Pipeline (all deterministic, no atomics on the output order):
//   k_mc_classify  cube index + "cube is live" (base in loop box, 8 corners unmasked) per cell
//   k_mc_count     per cell: which of its 3 edges carry a vertex, triangles of its cube;
//                  wavefront ballot + popcount prefix, per-block sums
//   k_mc_scan      exclusive scan of the block sums (one workgroup)
//   k_mc_verts     vertex offsets per cell (kept for the triangle pass) + interpolated vertices
//   k_mc_tris      triangle connectivity (local vertex ids)
// Each edge vertex is interpolated with the endpoint order of the FIRST live cube that touches the
// edge in traversal order, like the reference's vertCache does.
#include "pa_internal.h"
#include "pa_fabview.h"
#include "mc_tables.h"
#include <vector>
#define PA_TRY_RET(x) do { if ((x) != 0) return 1; } while (0)

__constant__ unsigned short c_edge[256];
__constant__ signed char c_tri[256][16];
__constant__ unsigned char c_ntri[256];
static bool g_tables_up = false;

extern "C" const uint16_t* pa_mc_edge_table(void) { return PA_MC_EDGE_TABLE; }
extern "C" const int8_t* pa_mc_tri_table(void) { return &PA_MC_TRI_TABLE[0][0]; }

static int upload_tables(pa_ctx* ctx) {
  if (g_tables_up) return 0;
  unsigned char nt[256];
  for (int c = 0; c < 256; ++c) {
    int n = 0;
    while (PA_MC_TRI_TABLE[c][3 * n] != -1) ++n;
    nt[c] = (unsigned char)n;
  }
  PA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_edge), PA_MC_EDGE_TABLE, sizeof(PA_MC_EDGE_TABLE)));
  PA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tri), PA_MC_TRI_TABLE, sizeof(PA_MC_TRI_TABLE)));
  PA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_ntri), nt, sizeof(nt)));
  g_tables_up = true;
  return 0;
}

struct McGeom {
  FabView S, M;      // state (3 coords + fields), mask
  int slo[3], n[3];  // state box
  int llo[3], lhi[3];
  int ncomp, isocomp;
  double iso;
  long long ncell;
};

// cube corners p0..p7 (isosurface.cpp:426-433); edge e: lower endpoint offset + direction; and the
// reference's endpoint order (a -> b) per cube edge (isosurface.cpp:762-785)
__device__ __constant__ int d_corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
__device__ __constant__ int d_elo[12][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 0}, {0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {0, 0, 1}, {0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}};
__device__ __constant__ int d_edir[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};

__device__ __forceinline__ void cell_of(const McGeom& G, long long lin, int& i, int& j, int& k) {
  i = (int)(lin % G.n[0]) + G.slo[0];
  j = (int)((lin / G.n[0]) % G.n[1]) + G.slo[1];
  k = (int)(lin / ((long long)G.n[0] * G.n[1])) + G.slo[2];
}
__device__ __forceinline__ long long lin_of(const McGeom& G, int i, int j, int k) {
  return ((long long)(k - G.slo[2]) * G.n[1] + (j - G.slo[1])) * G.n[0] + (i - G.slo[0]);
}

__global__ __launch_bounds__(256) void k_mc_classify(McGeom G, unsigned char* live, unsigned char* cidx) {
  const long long lin = blockIdx.x * 256LL + threadIdx.x;
  if (lin >= G.ncell) return;
  int i, j, k;
  cell_of(G, lin, i, j, k);
  bool ok = i >= G.llo[0] && i <= G.lhi[0] && j >= G.llo[1] && j <= G.lhi[1] && k >= G.llo[2] && k <= G.lhi[2];
  int ci = 0;
  if (ok) {
    for (int m = 0; m < 8; ++m) {
      const int ii = i + d_corner[m][0], jj = j + d_corner[m][1], kk = k + d_corner[m][2];
      if (G.M(ii, jj, kk, 0) < 0.0) ok = false;  // Polygonise bails if any corner is masked (:436-438)
      if (G.S(ii, jj, kk, G.isocomp) < G.iso) ci |= (1 << m);
    }
  }
  live[lin] = ok ? 1 : 0;
  cidx[lin] = ok ? (unsigned char)ci : 0;
}

// which of the (up to 4) cubes around the edge (cell l, direction dir) is the first live one in
// traversal order; returns -1 if none.  Order and orientation: SURVEY A.7.
__device__ __forceinline__ int first_toucher(const McGeom& G, const unsigned char* live, int i, int j, int k, int dir, bool& reversed) {
  // bases of the cubes sharing the edge, in traversal (z-major) order, and whether the cube's own
  // edge runs high -> low along dir
  int bi[4], bj[4], bk[4];
  bool rev[4];
  if (dir == 0) {
    bi[0] = i; bj[0] = j - 1; bk[0] = k - 1; rev[0] = true;   // cube edge 6: p6 -> p7
    bi[1] = i; bj[1] = j;     bk[1] = k - 1; rev[1] = false;  // edge 4: p4 -> p5
    bi[2] = i; bj[2] = j - 1; bk[2] = k;     rev[2] = true;   // edge 2: p2 -> p3
    bi[3] = i; bj[3] = j;     bk[3] = k;     rev[3] = false;  // edge 0: p0 -> p1
  } else if (dir == 1) {
    bi[0] = i - 1; bj[0] = j; bk[0] = k - 1; rev[0] = false;  // edge 5: p5 -> p6
    bi[1] = i;     bj[1] = j; bk[1] = k - 1; rev[1] = true;   // edge 7: p7 -> p4
    bi[2] = i - 1; bj[2] = j; bk[2] = k;     rev[2] = false;  // edge 1: p1 -> p2
    bi[3] = i;     bj[3] = j; bk[3] = k;     rev[3] = true;   // edge 3: p3 -> p0
  } else {
    bi[0] = i - 1; bj[0] = j - 1; bk[0] = k; rev[0] = false;  // edge 10
    bi[1] = i;     bj[1] = j - 1; bk[1] = k; rev[1] = false;  // edge 11
    bi[2] = i - 1; bj[2] = j;     bk[2] = k; rev[2] = false;  // edge 9
    bi[3] = i;     bj[3] = j;     bk[3] = k; rev[3] = false;  // edge 8
  }
  for (int q = 0; q < 4; ++q) {
    if (bi[q] < G.slo[0] || bj[q] < G.slo[1] || bk[q] < G.slo[2]) continue;
    if (live[lin_of(G, bi[q], bj[q], bk[q])]) { reversed = rev[q]; return q; }
  }
  return -1;
}

// per cell: bit d set = the edge (cell, d) carries a vertex
__device__ __forceinline__ int edge_bits(const McGeom& G, const unsigned char* live, int i, int j, int k) {
  int bits = 0;
  const bool in0 = G.S(i, j, k, G.isocomp) < G.iso;
  const int hi[3] = {G.slo[0] + G.n[0] - 1, G.slo[1] + G.n[1] - 1, G.slo[2] + G.n[2] - 1};
  const int p[3] = {i, j, k};
  for (int d = 0; d < 3; ++d) {
    if (p[d] + 1 > hi[d]) continue;
    const bool in1 = G.S(i + (d == 0), j + (d == 1), k + (d == 2), G.isocomp) < G.iso;
    if (in0 == in1) continue;  // edgeTable flags an edge iff its endpoints are on different sides
    bool rev;
    if (first_toucher(G, live, i, j, k, d, rev) >= 0) bits |= (1 << d);
  }
  return bits;
}

__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// exclusive prefix inside the workgroup of (nv, nt) using ballots of the count bits; returns block totals
__device__ __forceinline__ void block_prefix(int nv, int nt, int& pv, int& pt, int& tv, int& tt) {
  __shared__ int s_v[4], s_t[4];
  const unsigned long long lt = lanemask_lt();
  int wv = 0, wt = 0, av = 0, at = 0;
  for (int b = 0; b < 3; ++b) {
    const unsigned long long mv = __ballot((nv >> b) & 1), mt = __ballot((nt >> b) & 1);
    wv += __popcll(mv & lt) << b;
    wt += __popcll(mt & lt) << b;
    av += __popcll(mv) << b;
    at += __popcll(mt) << b;
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { s_v[w] = av; s_t[w] = at; }
  __syncthreads();
  pv = wv; pt = wt; tv = 0; tt = 0;
  for (int q = 0; q < 4; ++q) {
    if (q < w) { pv += s_v[q]; pt += s_t[q]; }
    tv += s_v[q]; tt += s_t[q];
  }
}

__global__ __launch_bounds__(256) void k_mc_count(McGeom G, const unsigned char* live, const unsigned char* cidx, unsigned char* vflag,
                                                  int* bsum /* [nblocks][2] */) {
  const long long lin = blockIdx.x * 256LL + threadIdx.x;
  int bits = 0, nt = 0;
  if (lin < G.ncell) {
    int i, j, k;
    cell_of(G, lin, i, j, k);
    bits = edge_bits(G, live, i, j, k);
    nt = live[lin] ? c_ntri[cidx[lin]] : 0;
    vflag[lin] = (unsigned char)bits;
  }
  int pv, pt, tv, tt;
  block_prefix(__popc(bits), nt, pv, pt, tv, tt);
  if (threadIdx.x == 0) { bsum[2 * blockIdx.x] = tv; bsum[2 * blockIdx.x + 1] = tt; }
}

// exclusive scan of the block sums in place; totals to tot[0..1]
__global__ __launch_bounds__(1024) void k_mc_scan(int* bsum, int nblocks, long long* tot) {
  __shared__ long long s_a[1024], s_b[1024];
  const int t = threadIdx.x;
  const int per = (nblocks + 1023) / 1024;
  const int lo = t * per, hi = min(lo + per, nblocks);
  long long a = 0, b = 0;
  for (int q = lo; q < hi; ++q) { a += bsum[2 * q]; b += bsum[2 * q + 1]; }
  s_a[t] = a; s_b[t] = b;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan of the per-thread sums
    const long long xa = t >= o ? s_a[t - o] : 0, xb = t >= o ? s_b[t - o] : 0;
    __syncthreads();
    s_a[t] += xa; s_b[t] += xb;
    __syncthreads();
  }
  long long ea = s_a[t] - a, eb = s_b[t] - b;
  for (int q = lo; q < hi; ++q) {
    const int va = bsum[2 * q], vb = bsum[2 * q + 1];
    bsum[2 * q] = (int)ea; bsum[2 * q + 1] = (int)eb;
    ea += va; eb += vb;
  }
  if (t == 1023) { tot[0] = s_a[1023]; tot[1] = s_b[1023]; }
}

#define PA_EPS_DEF 1.e-15

__global__ __launch_bounds__(256) void k_mc_verts(McGeom G, const unsigned char* live, const unsigned char* vflag, const int* bsum, int* voff,
                                                  double* verts, int* vkeys) {
  const long long lin = blockIdx.x * 256LL + threadIdx.x;
  const int bits = lin < G.ncell ? vflag[lin] : 0;
  int pv, pt, tv, tt;
  block_prefix(__popc(bits), 0, pv, pt, tv, tt);
  if (lin >= G.ncell) return;
  int vid = bsum[2 * blockIdx.x] + pv;
  voff[lin] = vid;
  if (!bits) return;
  int i, j, k;
  cell_of(G, lin, i, j, k);
  for (int d = 0; d < 3; ++d) {
    if (!(bits & (1 << d))) continue;
    bool rev = false;
    first_toucher(G, live, i, j, k, d, rev);
    const int hi_i = i + (d == 0), hi_j = j + (d == 1), hi_k = k + (d == 2);
    // VI_doIt(isoVal, isoComp, p1 = first endpoint of the first toucher's edge, p2 = the other)
    const int a[3] = {rev ? hi_i : i, rev ? hi_j : j, rev ? hi_k : k};
    const int b[3] = {rev ? i : hi_i, rev ? j : hi_j, rev ? k : hi_k};
    const double v1 = G.S(a[0], a[1], a[2], G.isocomp), v2 = G.S(b[0], b[1], b[2], G.isocomp);
    double* o = verts + (long long)vid * G.ncomp;
    int mode;  // 0 copy p1, 1 copy p2, 2 interpolate
    if (fabs(G.iso - v1) < PA_EPS_DEF) mode = 0;
    else if (fabs(G.iso - v2) < PA_EPS_DEF) mode = 1;
    else if (fabs(v1 - v2) < PA_EPS_DEF) mode = 0;
    else mode = 2;
    const double mu = mode == 2 ? (G.iso - v1) / (v2 - v1) : 0.0;
    for (int c = 0; c < G.ncomp; ++c) {
      const double a1 = G.S(a[0], a[1], a[2], c), a2 = G.S(b[0], b[1], b[2], c);
      o[c] = mode == 0 ? a1 : (mode == 1 ? a2 : a1 + mu * (a2 - a1));
    }
    int* key = vkeys + 6LL * vid;
    key[0] = i; key[1] = j; key[2] = k; key[3] = hi_i; key[4] = hi_j; key[5] = hi_k;
    ++vid;
  }
}

__global__ __launch_bounds__(256) void k_mc_tris(McGeom G, const unsigned char* live, const unsigned char* cidx, const unsigned char* vflag,
                                                 const int* bsum, const int* voff, int* tris) {
  const long long lin = blockIdx.x * 256LL + threadIdx.x;
  const int ci = (lin < G.ncell && live[lin]) ? cidx[lin] : 0;
  const int nt = c_ntri[ci];
  int pv, pt, tv, tt;
  block_prefix(0, nt, pv, pt, tv, tt);
  if (nt == 0) return;
  int i, j, k;
  cell_of(G, lin, i, j, k);
  int* o = tris + 3LL * (bsum[2 * blockIdx.x + 1] + pt);
  for (int q = 0; q < 3 * nt; ++q) {
    const int e = c_tri[ci][q];
    const long long le = lin_of(G, i + d_elo[e][0], j + d_elo[e][1], k + d_elo[e][2]);
    const int dir = d_edir[e];
    o[q] = voff[le] + __popc(vflag[le] & ((1 << dir) - 1));
  }
}

static int ensure_scr(pa_ctx* ctx, size_t bytes) {
  if (ctx->scr_cap >= bytes) return 0;
  if (ctx->d_scr) (void)hipFree(ctx->d_scr);
  ctx->d_scr = nullptr;
  ctx->scr_cap = 0;
  PA_HIP(hipMalloc(&ctx->d_scr, bytes));
  ctx->scr_cap = bytes;
  return 0;
}

struct McScratch {
  unsigned char *live, *cidx, *vflag;
  int *voff, *bsum;
  long long* tot;
  int nblocks;
};

static int mc_setup(pa_ctx* ctx, pa_box loop, const pa_fab* state, const pa_fab* mask, int isocomp, double isoval, McGeom& G, McScratch& W) {
  if (!ctx || !state || !mask) return pa_fail(ctx, "pa_mc: null argument");
  if (!state->p || !mask->p) return pa_fail(ctx, "pa_mc: null fab pointer");
  if (state->ncomp < 4) return pa_fail(ctx, "pa_mc: state needs 3 coordinate components + at least one field");
  if (isocomp < 0 || isocomp >= state->ncomp) return pa_fail(ctx, "pa_mc: isocomp out of range");
  for (int d = 0; d < 3; ++d) {
    if (state->lo[d] != mask->lo[d] || state->hi[d] != mask->hi[d]) return pa_fail(ctx, "pa_mc: state and mask must live on the same box");
    if (loop.lo[d] < state->lo[d] || loop.hi[d] + 1 > state->hi[d]) return pa_fail(ctx, "pa_mc: loop box + 1 must lie inside the state box");
  }
  PA_TRY_RET(upload_tables(ctx));
  G.S = fab_view(*state);
  G.M = fab_view(*mask);
  G.ncell = 1;
  for (int d = 0; d < 3; ++d) {
    G.slo[d] = state->lo[d];
    G.n[d] = state->hi[d] - state->lo[d] + 1;
    G.llo[d] = loop.lo[d];
    G.lhi[d] = loop.hi[d];
    G.ncell *= G.n[d];
  }
  G.ncomp = state->ncomp; G.isocomp = isocomp; G.iso = isoval;
  W.nblocks = (int)((G.ncell + 255) / 256);
  const size_t a = ((size_t)G.ncell + 255) / 256 * 256;
  const size_t bytes = 3 * a + 4 * a + 8 * (size_t)W.nblocks + 64;
  if (ensure_scr(ctx, bytes)) return 1;
  unsigned char* p = (unsigned char*)ctx->d_scr;
  W.tot = (long long*)p; p += 64;
  W.voff = (int*)p; p += 4 * a;
  W.bsum = (int*)p; p += 8 * (size_t)W.nblocks;
  W.live = p; p += a;
  W.cidx = p; p += a;
  W.vflag = p;
  return 0;
}

static int mc_count(pa_ctx* ctx, const McGeom& G, const McScratch& W, long long tot[2]) {
  hipLaunchKernelGGL(k_mc_classify, dim3(W.nblocks), dim3(256), 0, ctx->stream, G, W.live, W.cidx);
  hipLaunchKernelGGL(k_mc_count, dim3(W.nblocks), dim3(256), 0, ctx->stream, G, W.live, W.cidx, W.vflag, W.bsum);
  hipLaunchKernelGGL(k_mc_scan, dim3(1), dim3(1024), 0, ctx->stream, W.bsum, W.nblocks, W.tot);
  PA_HIP(hipGetLastError());
  PA_HIP(hipMemcpyAsync(tot, W.tot, 2 * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
  PA_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int pa_mc_count_fab(pa_ctx* ctx, pa_box loop, const pa_fab* state, const pa_fab* mask, int isocomp, double isoval, int64_t* nvert,
                               int64_t* ntri) {
  if (!nvert || !ntri) return pa_fail(ctx, "pa_mc_count_fab: null argument");
  McGeom G;
  McScratch W;
  if (mc_setup(ctx, loop, state, mask, isocomp, isoval, G, W)) return 1;
  long long tot[2];
  ProfScope prof(ctx, PA_TAG_MC);
  if (mc_count(ctx, G, W, tot)) return 1;
  *nvert = tot[0];
  *ntri = tot[1];
  return 0;
}

extern "C" int pa_mc_emit_fab(pa_ctx* ctx, pa_box loop, const pa_fab* state, const pa_fab* mask, int isocomp, double isoval, double* dev_verts,
                              int32_t* dev_vkeys, int32_t* dev_tris, int64_t nvert, int64_t ntri) {
  McGeom G;
  McScratch W;
  if (mc_setup(ctx, loop, state, mask, isocomp, isoval, G, W)) return 1;
  long long tot[2];
  ProfScope prof(ctx, PA_TAG_MC);
  if (mc_count(ctx, G, W, tot)) return 1;
  if (tot[0] != nvert || tot[1] != ntri)
    return pa_fail(ctx, "pa_mc_emit_fab: buffer sizes (" + std::to_string(nvert) + " vertices, " + std::to_string(ntri) +
                            " triangles) do not match the surface (" + std::to_string(tot[0]) + ", " + std::to_string(tot[1]) + ")");
  if (nvert > 0 && (!dev_verts || !dev_vkeys)) return pa_fail(ctx, "pa_mc_emit_fab: null vertex buffers");
  if (ntri > 0 && !dev_tris) return pa_fail(ctx, "pa_mc_emit_fab: null triangle buffer");
  if (nvert > 0x7fffffffLL || ntri > 0x7fffffffLL / 3) return pa_fail(ctx, "pa_mc_emit_fab: surface too large for 32-bit ids");
  hipLaunchKernelGGL(k_mc_verts, dim3(W.nblocks), dim3(256), 0, ctx->stream, G, W.live, W.vflag, W.bsum, W.voff, dev_verts, dev_vkeys);
  hipLaunchKernelGGL(k_mc_tris, dim3(W.nblocks), dim3(256), 0, ctx->stream, G, W.live, W.cidx, W.vflag, W.bsum, W.voff, dev_tris);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
