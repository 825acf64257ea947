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
#include <cstring>
#include "pa_internal.h"
#include "pa_fabview.h"
#include "mc_tables.h"
#include <cstdlib>
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
template <typename GEO>
__device__ __forceinline__ long long lin_of(const GEO& G, int i, int j, int k) {
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
template <typename GEO>
__device__ __forceinline__ int first_toucher(const GEO& G, const unsigned char* live, int i, int j, int k, int dir, bool& reversed) {
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
    if (live[lin_of(G, bi[q], bj[q], bk[q])] & 1) { reversed = rev[q]; return q; }  // bit 0 (k_mcl_cells packs more into the byte)
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
  PaBind bind_(ctx);
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
  PaBind bind_(ctx);
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


// =============================================================================================
// Level-batched marching cubes: every FAB of a level in one pass (isosurface.cpp:1531-1592 for the
// whole MFIter loop).  Same logic as the per-FAB kernels above, launched over (tiles of a FAB) x (FABs);
// the per-FAB outputs are identical to pa_mc_count_fab / pa_mc_emit_fab and are concatenated in box order.
// The state and the mask are level multifabs with the same ghost width, so a cell's index in its
// grown FAB addresses both, and the per-cell scratch (2 B from the cell pass, 1 B edge bits, 4 B vertex
// offset) is laid out FAB after FAB, each FAB padded to a whole 256-cell block.
//   k_mcl_cells     ONE streaming pass over the iso component and the mask (16 B per cell): per cell the byte
//                   lc = cube is live | own x/y/z edge crosses the iso value << 1, and the cube index; marks
//                   the 256-cell blocks that contain anything (most of a level does not touch the surface)
//   k_mcl_count     marked blocks only: edge bits (first live cube that touches the edge) + triangle counts
//   k_mcl_scan      one workgroup per FAB: exclusive scan of its block sums, FAB totals
//   k_mcl_lists     marked blocks only: vertex offsets + one work item per vertex / triangle
//   k_mcl_verts / k_mcl_tris  one thread per vertex / triangle; as k_mc_verts / k_mc_tris, behind the FAB's base
struct MclGeo {
  int slo[3], n[3];
  int llo[3], lhi[3];
  unsigned ncell;
};
struct MclArgs {
  DLevelView L;
  DMFView S, M;
  int mcomp, isocomp, ncomp, kseg;
  int rows, nslab, tiles, ldsw;  // k_mcl_cells4: rows of a slab (multiple of 4), slabs / tiles of the largest FAB, LDS dwords per parity
  const int* tiletab = nullptr;  // FABs of different sizes (mcl_tiletab): workgroup w = {FAB, first row, rows, first plane} = tiletab[4 w .. 4 w + 3]; `tiles` = entries
  unsigned ntab = 0;
  int nomask, has_fine, ratio;  // nomask: no mask multifab -- a cell is masked iff the next finer level LF covers it
  DLevelView LF;
  int dim2;                // marching squares on the plane k = loops[b].lo[2] (isosurface.cpp:303-406), see below
  double iso;
  const DBox* loops;       // [nboxes] cube base points, lo > hi: FAB skipped
  const long long* coff;   // [nboxes + 1] first scratch cell of a FAB (multiples of 256)
  unsigned char *lc, *cidx, *vflag, *bact;
  int* alist;              // marked (block, FAB) pairs in any order, *nact of them
  int* nact;
  int* voff;
  int* bsum;               // [coff[nboxes] / 256][2]
  long long* tot;          // [nboxes][2]
  const long long* base;   // [nboxes][2] first vertex / triangle of a FAB in the output
  // xyz != 0 (pa_mc_hierarchy_xyz): the state holds the FIELDS only and a vertex's first three components are formed from cell
  // indices -- the cell-centre coordinates isosurface.cpp:1458-1465 stores per cell and :1468-1478 overwrites in ghost cells
  // are a function of (i, j, k) and of which level covers the cell (mcl_xyz) -- instead of read from three stored components;
  // ncomp stays the number of values per vertex, 3 + the state's components
  int xyz = 0, cratio = 0;
  double gdx[3] = {0, 0, 0}, gplo[3] = {0, 0, 0}, gdxc[3] = {0, 0, 0};
};
// Component d of what the reference's state holds in its coordinate components at cell p of a grown FAB of level A.L:
//   a cell covered by the level itself (valid, or a ghost cell inside a neighbouring box / a periodic image of one: FillBoundary
//   copies the owner's value, isosurface.cpp:1468)          -> (i_w + 0.5) * dx + plo with i_w the index folded into the domain;
//   any other ghost cell inside the (periodically extended) domain: FillPatchTwoLevels with PCInterp hands it its coarse
//   parent's value (:1474-1478)                              -> (coarsen(i_w) + 0.5) * dx_coarse + plo.
// Same operations in the same order as k_iso_coords, so the doubles are the ones the stored components would hold.
__device__ __forceinline__ void mcl_xyz(const MclArgs& A, const int p[3], double x[3]) {
  int pw[3] = {p[0], p[1], p[2]};
  const bool inside = wrap_cell(A.L, pw);
  const bool covered = !inside || A.cratio <= 0 || owner_of(A.L, pw) != -1;  // (cells beyond a wall are never corners of a visited cube)
#pragma unroll
  for (int d = 0; d < 3; ++d)
    x[d] = covered ? (pw[d] + 0.5) * A.gdx[d] + A.gplo[d] : (coarsen_idx(pw[d], A.cratio) + 0.5) * A.gdxc[d] + A.gplo[d];
}

__device__ __forceinline__ bool mcl_geo(const MclArgs& A, int b, MclGeo& G) {
  const DBox B = A.L.boxes[b], Lp = A.loops[b];
  unsigned nc = 1;
  bool live = true;
  for (int d = 0; d < 3; ++d) {
    G.slo[d] = B.lo[d] - A.S.ng;
    G.n[d] = B.hi[d] - B.lo[d] + 1 + 2 * A.S.ng;
    G.llo[d] = Lp.lo[d];
    G.lhi[d] = Lp.hi[d];
    live = live && Lp.lo[d] <= Lp.hi[d];
    nc *= (unsigned)G.n[d];
  }
  G.ncell = nc;
  return live;
}
__device__ __forceinline__ void mcl_cell(const MclGeo& G, unsigned lin, int& i, int& j, int& k) {
  const unsigned r = lin / (unsigned)G.n[0], kk = r / (unsigned)G.n[1];
  i = (int)(lin - r * (unsigned)G.n[0]) + G.slo[0];
  j = (int)(r - kk * (unsigned)G.n[1]) + G.slo[1];
  k = (int)kk + G.slo[2];
}

// ---- 2-D (AMREX_SPACEDIM == 2 builds of the reference: Segmentise, isosurface.cpp:303-406) in the same pipeline.
// A 2-D level is stored as one plane of cells (k = 0); the "cube" of a cell is its square p0 = (i,j), p1 = (i+1,j),
// p2 = (i+1,j+1), p3 = (i,j+1), whose case index is the low nibble of the cube index; square edges e0 = p0p1,
// e1 = p1p2, e2 = p2p3, e3 = p3p0; the segments of a case in Segmentise's order (:352-406):
__device__ __constant__ signed char d_seg[16][4] = {{-1, -1, -1, -1}, {0, 3, -1, -1}, {0, 1, -1, -1}, {1, 3, -1, -1}, {1, 2, -1, -1}, {0, 1, 2, 3},
                                                    {0, 2, -1, -1},   {2, 3, -1, -1}, {2, 3, -1, -1}, {0, 2, -1, -1}, {0, 1, 2, 3},   {1, 2, -1, -1},
                                                    {1, 3, -1, -1},   {0, 1, -1, -1}, {0, 3, -1, -1}, {-1, -1, -1, -1}};
__device__ __constant__ unsigned char d_nseg[16] = {0, 1, 1, 1, 1, 2, 1, 1, 1, 1, 2, 1, 1, 1, 1, 0};
__device__ __constant__ int d_sq_elo[4][2] = {{0, 0}, {1, 0}, {0, 1}, {0, 0}};  // lower endpoint of a square edge
__device__ __constant__ int d_sq_edir[4] = {0, 1, 0, 1};
// first live square, in traversal order (x fastest), that touches the edge (cell, dir), and whether its own
// VertexInterp call runs high -> low along dir: e2 = (p2 -> p3) and e3 = (p3 -> p0) do
template <typename GEO>
__device__ __forceinline__ int first_toucher2(const GEO& G, const unsigned char* live, int i, int j, int k, int dir, bool& reversed) {
  int bi[2], bj[2];
  bool rev[2];
  if (dir == 0) {
    bi[0] = i; bj[0] = j - 1; rev[0] = true;    // e2 of the square below: p2 -> p3
    bi[1] = i; bj[1] = j;     rev[1] = false;   // e0: p0 -> p1
  } else {
    bi[0] = i - 1; bj[0] = j; rev[0] = false;   // e1 of the square to the left: p1 -> p2
    bi[1] = i;     bj[1] = j; rev[1] = true;    // e3: p3 -> p0
  }
  for (int q = 0; q < 2; ++q) {
    if (bi[q] < G.slo[0] || bj[q] < G.slo[1]) continue;
    if (live[lin_of(G, bi[q], bj[q], k)] & 1) { reversed = rev[q]; return q; }
  }
  return -1;
}

// Cell pass (MEASURED, 64 FABs of 130^3 = 1.4e8 cells: 0.76 ms = 2.9 TB/s of the 16 B/cell; without its byte stores
// 0.60 ms, without the LDS exchange 0.74 ms, tile rows 4/8/16 and 16..130 planes per workgroup all within 3 %: the
// two strided read streams are the cost).  Workgroup = 64 x TY cells of a plane marching a z-segment; a cell's cube needs the flags
// (inside: value < iso; masked) of the 2 x 2 cells at i..i+1, j..j+1 of two consecutive planes: the in-plane
// neighbours come through LDS (tiles overlap by one column and one row, the re-read lines are L2 hits), the
// lower plane is carried in registers.  Output cells of a tile: all but its last column / row, plus the FAB's
// last column / row (no cube there, but the cell still owns the edges along the FAB's high faces).
template <int TY>
__global__ __launch_bounds__(64 * TY) void k_mcl_cells(MclArgs A) {
  const int b = blockIdx.y;
  MclGeo G;
  if (!mcl_geo(A, b, G)) return;
  const int nx = G.n[0], ny = G.n[1], nz = G.n[2];
  const int ntx = max(1, (nx - 1 + 62) / 63), nty = max(1, (ny - 1 + TY - 2) / (TY - 1)), ntz = (nz + A.kseg - 1) / A.kseg;
  const unsigned bid = blockIdx.x;
  if (bid >= (unsigned)(ntx * nty * ntz)) return;  // uniform
  const int tx = bid % ntx, ty = (bid / ntx) % nty, tz = bid / (ntx * nty);
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const int i = tx * 63 + lx, j = ty * (TY - 1) + ly;  // FAB-local
  const int ic = min(i, nx - 1), jc = min(j, ny - 1);  // threads past the FAB re-read its last column / row
  const bool own = i < nx && j < ny && (lx < 63 || i == nx - 1) && (ly < TY - 1 || j == ny - 1);
  const int k0 = tz * A.kseg, k1 = min(k0 + A.kseg, nz) - 1;
  const long long nxy = (long long)nx * ny;
  const long long cs = pa_cstride((long long)G.ncell, A.S.ncomp), cm = A.nomask ? 0 : pa_cstride((long long)G.ncell, A.M.ncomp);
  const double* sp = A.S.data + A.S.off[b] + (long long)A.isocomp * cs + ((long long)jc * nx + ic);
  const double* mp = A.nomask ? sp : A.M.data + A.M.off[b] + (long long)A.mcomp * cm + ((long long)jc * nx + ic);
  const double iso = A.iso;
  __shared__ unsigned char sf[2][TY][64];
  const int lx1 = min(lx + 1, 63), ly1 = min(ly + 1, TY - 1);
  // flags of the 2 x 2 cells of one plane in cube-corner order (p0, p1, p2, p3): bits 0-3 inside, bits 4-7 masked
  auto quad = [&](double s, double m, int par) {
    const int f = (s < iso ? 1 : 0) | (m < 0.0 ? 2 : 0);
    sf[par][ly][lx] = (unsigned char)f;
    __syncthreads();
    const int f1 = sf[par][ly][lx1], f2 = sf[par][ly1][lx1], f3 = sf[par][ly1][lx];
    return (f & 1) | ((f1 & 1) << 1) | ((f2 & 1) << 2) | ((f3 & 1) << 3) | ((f & 2) << 3) | ((f1 & 2) << 4) | ((f2 & 2) << 5) | ((f3 & 2) << 6);
  };
  const bool xin = i + 1 < nx, yin = j + 1 < ny;
  const bool okxy = i + G.slo[0] >= G.llo[0] && i + G.slo[0] <= G.lhi[0] && j + G.slo[1] >= G.llo[1] && j + G.slo[1] <= G.lhi[1];
  const long long g0 = A.coff[b];
  const unsigned lin0 = (unsigned)((long long)jc * nx + ic);
  // plane k's results need the flags of plane k+1; planes are requested P at a time, one chunk ahead of the chunk
  // being consumed (each plane costs a barrier, so without the distance every plane waits out a memory round trip)
  constexpr int P = 4;
  const int klast = min(k1 + 1, nz - 1);  // last plane read
  auto load_chunk = [&](int kb, double (&sv)[P], double (&mv)[P]) {
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const long long o = (long long)min(kb + q, nz - 1) * nxy;
      sv[q] = sp[o];
      if (A.nomask) {  // isosurface.cpp:1540-1563 evaluated in place: -1 where the refined cell has an owner on the finer level
        double m = 1.0;
        if (A.has_fine) {
          int p[3] = {(G.slo[0] + ic) * A.ratio, (G.slo[1] + jc) * A.ratio, (G.slo[2] + min(kb + q, nz - 1)) * A.ratio};
          if (wrap_cell(A.LF, p)) {
            const int ow = owner_of(A.LF, p);
            if (ow != -1) m = -1.0;
          }
        }
        mv[q] = m;
      } else {
        mv[q] = mp[o];
      }
    }
  };
  int cur = 0, par = 0;
  auto emit = [&](int k, int up, bool up_in) {  // cell (i, j, k): cur = its plane, up = the plane above
    const int in0 = cur & 1;
    const int kk = k + G.slo[2];
    int cand = ((xin && in0 != ((cur >> 1) & 1)) ? 1 : 0) | ((yin && in0 != ((cur >> 3) & 1)) ? 2 : 0);
    bool ok = okxy && kk >= G.llo[2] && kk <= G.lhi[2];
    int ci;
    if (A.dim2) {  // squares of the plane k = llo[2] only; no z edges, no upper corners
      if (kk != G.llo[2]) cand = 0;
      ok = ok && (cur & 0xF0) == 0;  // Segmentise bails if any corner is masked (:326-327)
      ci = ok ? (cur & 0xF) : 0;
    } else {
      cand |= (up_in && in0 != (up & 1)) ? 4 : 0;
      ok = ok && ((cur | up) & 0xF0) == 0;  // Polygonise bails if any corner is masked (:436-438)
      ci = ok ? ((cur & 0xF) | ((up & 0xF) << 4)) : 0;
    }
    if (own) {
      const long long g = g0 + lin0 + (unsigned long long)k * (unsigned long long)nxy;
      A.lc[g] = (unsigned char)((ok ? 1 : 0) | (cand << 1));
      A.cidx[g] = (unsigned char)ci;
      if (cand || (ci != 0 && ci != (A.dim2 ? 15 : 255))) A.bact[g >> 8] = 1;  // same value from every writer
    }
  };
  auto consume = [&](int kb, const double (&sv)[P], const double (&mv)[P]) {
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const int kp = kb + q;
      if (kp > klast) break;  // uniform
      const int nib = quad(sv[q], mv[q], par);
      par ^= 1;
      if (kp > k0) emit(kp - 1, nib, true);
      cur = nib;
    }
  };
  double sa[P], ma[P], sb[P], mb[P];
  load_chunk(k0, sa, ma);
  for (int kb = k0; kb <= klast; kb += 2 * P) {
    load_chunk(kb + P, sb, mb);
    consume(kb, sa, ma);
    load_chunk(kb + 2 * P, sa, ma);
    consume(kb + P, sb, mb);
  }
  if (k1 + 1 >= nz) emit(k1, 0, false);  // top plane of the FAB: no cube, no z edge
}

// Cell pass, second form (the default).  What the first form above pays for (PMC: 3.64 GB fetched for 2.25 GB of cells):
// its 63-column tiles cut every 130-cell row into three pieces on 8-byte offsets, so most 128-byte lines are fetched by
// two workgroups that sit on different XCDs (different L2s), one row in eight is read twice, and 45 % of the lanes
// idle.  A plane of a FAB is CONTIGUOUS in memory, so here a workgroup owns a slab of whole rows [j0, j0 + rows) and
// walks it as a flat array: lane-contiguous 32-byte loads (4 consecutive cells per thread), one halo row per slab, the
// only partial lines are the two ends of the slab.  The 4 cells of a thread are handled as 4 bytes of one register
// (flags, cube nibbles, candidate bits, cube index: all byte-parallel), neighbours come through LDS as dwords (the
// row above by a funnel shift, its offset q + nx has any alignment), and both result bytes of 4 cells leave as one dword
// each when the FAB's scratch offsets are 4-aligned (nx ny and j0 nx multiples of 4; else byte stores).  Slabs that
// share a halo row, and z segments that share a plane, are neighbours in a numbering that keeps them on one XCD.
// The values of cells whose right / upper neighbour does not exist are never used: such cells are outside every loop
// box (mc_level_impl checks loop hi + 1 <= FAB hi), so their cube index is forced to 0 and their x / y bits are masked.
template <int NT, int GPT, int MM>  // MM: 0 mask multifab, 1 no mask and no finer level, 2 mask = covered by the finer level LF
__device__ __forceinline__ void mcl_cells4_body(const MclArgs& A, const unsigned bidx, unsigned* s_fl) {
  // s_fl: [2][A.ldsw]: one flag byte per cell of the slab + halo row, by plane parity
  const unsigned total = A.tiletab ? A.ntab : (unsigned)A.L.nboxes * (unsigned)A.tiles, chunk = (total + 7u) / 8u;
  const unsigned wg = (bidx & 7u) * chunk + (bidx >> 3);  // XCD x of 8 works through one contiguous run of tiles
  if (wg >= total) return;
  int b, j0, k0, rows = A.rows;
  if (A.tiletab) {  // FABs of different sizes: every FAB has its own slab height (as many rows of ITS width as the workgroup holds) and only the tiles it has
    b = A.tiletab[4 * wg]; j0 = A.tiletab[4 * wg + 1]; rows = A.tiletab[4 * wg + 2]; k0 = A.tiletab[4 * wg + 3];
  } else {
    b = (int)(wg / (unsigned)A.tiles);
    const int tile = (int)(wg - (unsigned)b * (unsigned)A.tiles);
    const int slab = tile % A.nslab, tz = tile / A.nslab;
    j0 = slab * A.rows; k0 = tz * A.kseg;
  }
  MclGeo G;
  if (!mcl_geo(A, b, G)) return;
  const int nx = G.n[0], ny = G.n[1], nz = G.n[2];
  if (j0 >= ny || k0 >= nz) return;  // uniform
  const int own = min(rows, ny - j0), halo = j0 + own < ny ? 1 : 0;
  const int len_own = own * nx, len_all = (own + halo) * nx;
  const int k1 = min(k0 + A.kseg, nz) - 1, klast = min(k1 + 1, nz - 1);
  const long long nxy = (long long)nx * ny;
  const long long cs = pa_cstride((long long)G.ncell, A.S.ncomp), cm = MM ? 0 : pa_cstride((long long)G.ncell, A.M.ncomp);
  const double* sp = A.S.data + A.S.off[b] + (long long)A.isocomp * cs + (long long)j0 * nx;
  const double* mp = MM ? sp : A.M.data + A.M.off[b] + (long long)A.mcomp * cm + (long long)j0 * nx;
  const double iso = A.iso;
  const bool al4 = (nxy & 3) == 0 && (((long long)j0 * nx) & 3) == 0;
  const long long g0 = A.coff[b] + (long long)j0 * nx;
  const int t = threadIdx.x;
  constexpr unsigned M1 = 0x01010101u;
  // per group of 4 cells: byte masks (bit 0 of byte e = cell q + e)
  unsigned XM[GPT], YM[GPT], OKM[GPT], OWN[GPT];
#pragma unroll
  for (int g = 0; g < GPT; ++g) {
    const int q = 4 * (t + NT * g);
    int r = q / nx, c = q - r * nx;
    unsigned xm = 0, ym = 0, okm = 0, ow = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (q + e < len_own) {
        const int i = c, j = j0 + r;
        const bool xin = i + 1 < nx, yin = j + 1 < ny;
        const bool okxy = xin && yin && i + G.slo[0] >= G.llo[0] && i + G.slo[0] <= G.lhi[0] && j + G.slo[1] >= G.llo[1] && j + G.slo[1] <= G.lhi[1];
        ow |= 1u << (8 * e);
        xm |= (xin ? 1u : 0u) << (8 * e);
        ym |= (yin ? 1u : 0u) << (8 * e);
        okm |= (okxy ? 1u : 0u) << (8 * e);
      }
      if (++c == nx) { c = 0; ++r; }
    }
    XM[g] = xm; YM[g] = ym; OKM[g] = okm; OWN[g] = ow;
  }
  // ONE set of value registers: the loads of plane kp + 1 are issued when plane kp's values have been turned into flag
  // bytes, and fly while plane kp goes through LDS and plane kp - 1 is stored
  double vs[GPT][4], vm[GPT][4];
  auto load_plane = [&](int kp) {
    const long long o = (long long)min(kp, nz - 1) * nxy;
#pragma unroll
    for (int g = 0; g < GPT; ++g) {
      const int q = 4 * (t + NT * g);
      if (q + 4 <= len_all) {
        typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
        const d2u a = *(const d2u*)(sp + o + q), c = *(const d2u*)(sp + o + q + 2);
        vs[g][0] = a.x; vs[g][1] = a.y; vs[g][2] = c.x; vs[g][3] = c.y;
        if (MM == 0) {
          const d2u ma = *(const d2u*)(mp + o + q), mc = *(const d2u*)(mp + o + q + 2);
          vm[g][0] = ma.x; vm[g][1] = ma.y; vm[g][2] = mc.x; vm[g][3] = mc.y;
        }
      } else {  // the slab's last, partial group; groups past the slab re-read its last cell
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int qe = min(q + e, len_all - 1);
          vs[g][e] = sp[o + qe];
          if (MM == 0) vm[g][e] = mp[o + qe];
        }
      }
    }
  };
  int mzb[GPT];       // MM == 2: z cell of the finer level's owner map the group's mask bytes mkb were looked up for
  unsigned mkb[GPT];
#pragma unroll
  for (int g = 0; g < GPT; ++g) { mzb[g] = -3; mkb[g] = 0; }
  auto flags_of = [&](int kp, unsigned (&F)[GPT]) {
#pragma unroll
    for (int g = 0; g < GPT; ++g) {
      unsigned f = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) f |= ((vs[g][e] < iso ? 1u : 0u) | ((MM == 0 && vm[g][e] < 0.0) ? 2u : 0u)) << (8 * e);
      if (MM == 2) {  // isosurface.cpp:1540-1563 evaluated in place: masked where the refined cell has an owner on the finer level
        // The owner map of the finer level is constant over cells of g fine cells: a column of cells (fixed i, j) changes its
        // mask only when the plane crosses into another such cell in z -- every g / ratio planes -- so the four lookups of a
        // group are redone only then (wave-uniform: every lane of the workgroup is on the same plane).  Measured before: the
        // pass with the lookups on every plane took twice the time of the unmasked one.
        const int kc = min(kp, nz - 1);
        int pz = (G.slo[2] + kc) * A.ratio;
        bool zin = true;
        {
          const int len = A.LF.domhi[2] - A.LF.domlo[2] + 1;
          if (pz < A.LF.domlo[2] || pz > A.LF.domhi[2]) {
            if (!A.LF.is_per[2]) zin = false;
            else { while (pz < A.LF.domlo[2]) pz += len; while (pz > A.LF.domhi[2]) pz -= len; }
          }
        }
        const int rz = pz - A.LF.mlo[2];
        const int zb = !zin ? -2 : (rz < 0 ? -1 : (A.LF.gshift >= 0 ? (rz >> A.LF.gshift) : rz / A.LF.g));
        if (zb != mzb[g]) {
          mzb[g] = zb;
          unsigned m = 0;
          if (zb >= 0 && zb < A.LF.mn[2]) {
            const int q = 4 * (t + NT * g);
            int r = q / nx, c = q - r * nx;
#pragma unroll 1
            for (int e = 0; e < 4; ++e) {
              if (q + e < len_all) {
                int p[3] = {(G.slo[0] + c) * A.ratio, (G.slo[1] + j0 + r) * A.ratio, pz};
                if (wrap_cell(A.LF, p) && owner_of(A.LF, p) != -1) m |= 2u << (8 * e);
              }
              if (++c == nx) { c = 0; ++r; }
            }
          }
          mkb[g] = m;
        }
        f |= mkb[g];
      }
      F[g] = f;
    }
  };
  unsigned cur[GPT], F[GPT];
#pragma unroll
  for (int g = 0; g < GPT; ++g) cur[g] = 0;
  int par = 0;
  const unsigned actmask = A.dim2 ? 0x07070707u : 0x7F7F7F7Fu;
  auto emit = [&](int k, int g, unsigned cu, unsigned up, bool up_in) {  // cells of group g on plane k: cu = their plane, up = the plane above
    if (OWN[g] == 0) return;
    const int kk = k + G.slo[2];
    const bool kz = kk >= G.llo[2] && kk <= G.lhi[2];
    unsigned X = (cu ^ (cu >> 1)) & XM[g], Y = (cu ^ (cu >> 3)) & YM[g], Z, bad, ci;
    if (A.dim2) {  // squares of the plane k = llo[2] only; no z edges, no upper corners; Segmentise bails if a corner is masked (:326-327)
      if (kk != G.llo[2]) X = Y = 0;
      Z = 0;
      bad = cu & 0xF0F0F0F0u;
      ci = cu & 0x0F0F0F0Fu;
    } else {  // Polygonise bails if any corner is masked (:436-438)
      Z = up_in ? ((cu ^ up) & M1) : 0u;
      bad = (cu | up) & 0xF0F0F0F0u;
      ci = (cu & 0x0F0F0F0Fu) | ((up & 0x0F0F0F0Fu) << 4);
    }
    unsigned tt = bad >> 4;
    tt |= tt >> 1;
    tt |= tt >> 2;
    const unsigned okm = kz ? (~tt & OKM[g]) : 0u;
    ci &= okm * 0xFFu;
    const unsigned lc = okm | (X << 1) | (Y << 2) | (Z << 3);
    const unsigned act = (((X | Y | Z) * 0xFFu) | ((ci ^ (ci >> 1)) & actmask)) & (OWN[g] * 0xFFu);
    const long long gi = g0 + (unsigned long long)k * (unsigned long long)nxy + 4 * (t + NT * g);
    // The two code bytes of a cell are stored only where there is something: a cell that owns a crossing edge or whose live
    // cube is cut by the surface.  Every later kernel reads codes of such cells only, or the live bit of a cube that touches
    // a crossing edge -- which is cut if it is live -- so a zero byte stands for the rest.  The scratch that holds the codes
    // is all zeros between calls (pa_ctx::d_mcz, k_mcl_clean); the pass then writes a few per cent of the level's cells
    // instead of 2 B for every cell (measured before: 0.28 GB of the pass's 1.47 GB on a 512^3 level).
    if (!act) return;
    if (al4 && OWN[g] == M1) {
      *(unsigned*)(A.lc + gi) = lc;
      *(unsigned*)(A.cidx + gi) = ci;
      A.bact[gi >> 8] = 1;  // same value from every writer
    } else {
#pragma unroll 1
      for (int e = 0; e < 4; ++e)
        if (((OWN[g] >> (8 * e)) & 1u) && ((act >> (8 * e)) & 0xFFu)) {
          A.lc[gi + e] = (unsigned char)(lc >> (8 * e));
          A.cidx[gi + e] = (unsigned char)(ci >> (8 * e));
          A.bact[(gi + e) >> 8] = 1;
        }
    }
  };
  load_plane(k0);
  for (int kp = k0; kp <= klast; ++kp) {
    flags_of(kp, F);  // waits for plane kp
    if (kp < klast) load_plane(kp + 1);
    unsigned* sf = s_fl + par * A.ldsw;
#pragma unroll
    for (int g = 0; g < GPT; ++g)
      if (4 * (t + NT * g) < len_all) sf[t + NT * g] = F[g];
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GPT; ++g) {
      if (OWN[g] == 0) continue;
      const int qa = 4 * (t + NT * g) + nx;
      const unsigned nxt = sf[t + NT * g + 1];
      const unsigned F1 = (F[g] >> 8) | (nxt << 24);
      const unsigned lo = sf[qa >> 2], hi = sf[(qa >> 2) + 1];
      const int sh = (qa & 3) * 8;
      const unsigned long long W = (unsigned long long)lo | ((unsigned long long)hi << 32);
      const unsigned F3 = (unsigned)(W >> sh);
      const unsigned F2 = (unsigned)(W >> (sh + 8));  // bytes qa + 1 .. qa + 4 (sh + 8 <= 32)
      const unsigned N = (F[g] & M1) | ((F1 & M1) << 1) | ((F2 & M1) << 2) | ((F3 & M1) << 3) | ((F[g] & (M1 << 1)) << 3) | ((F1 & (M1 << 1)) << 4) |
                         ((F2 & (M1 << 1)) << 5) | ((F3 & (M1 << 1)) << 6);
      if (kp > k0) emit(kp - 1, g, cur[g], N, true);
      cur[g] = N;
    }
    par ^= 1;
  }
  if (k1 + 1 >= nz) {
#pragma unroll
    for (int g = 0; g < GPT; ++g) emit(k1, g, cur[g], 0u, false);  // top plane of the FAB: no cube, no z edge
  }
}

template <int NT, int GPT, int MM>
__global__ __launch_bounds__(NT) void k_mcl_cells4(MclArgs A) {
  extern __shared__ unsigned s_fl[];
  mcl_cells4_body<NT, GPT, MM>(A, blockIdx.x, s_fl);
}

// ---- several levels per launch (pa_mc_hierarchy_fine): level l owns workgroups wg0[l] .. wg0[l+1]-1 of a launch (the host
// fills wg0 per kernel: tiles, 1024-block chunks, FABs, vertices / 256, triangles / 256); the persistent kernels walk the
// levels' marked-block lists one after the other
struct MclBatch {
  int n;
  unsigned wg0[PA_MAXB + 1];
  MclArgs a[PA_MAXB];
  __device__ __forceinline__ int find(unsigned w, unsigned& local) const {
    int l = 0;
    while (l + 1 < n && w >= wg0[l + 1]) ++l;
    local = w - wg0[l];
    return l;
  }
};
template <int NT, int GPT>
__global__ __launch_bounds__(NT) void k_mclb_cells4(MclBatch Bt) {
  extern __shared__ unsigned s_fl[];
  unsigned w;
  const MclArgs A = Bt.a[Bt.find(blockIdx.x, w)];  // by value: the level's arguments in registers, not re-read from the argument segment
  if (A.has_fine) mcl_cells4_body<NT, GPT, 2>(A, w, s_fl);
  else mcl_cells4_body<NT, GPT, 1>(A, w, s_fl);
}

// FAB of a scratch block: last b with coff[b] <= first cell of the block
__device__ __forceinline__ int mcl_box_of(const MclArgs& A, long long cell0) {
  int lo = 0, hi = A.L.nboxes - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (A.coff[mid] <= cell0) lo = mid; else hi = mid - 1;
  }
  return lo;
}
// list of the marked blocks as (block, FAB) pairs (order irrelevant: every block is processed on its own); one atomic per
// WORKGROUP of 1024 blocks (per wave of 64 it was 8192 serialised atomics on one address for a 512^3 level: 35 us)
__device__ __forceinline__ void mcl_active_body(const MclArgs& A, int nblk, unsigned bidx) {
  __shared__ int s_cnt[16], s_base;
  const int q = (int)bidx * 1024 + threadIdx.x;
  const bool on = q < nblk && A.bact[q];
  const unsigned long long m = __ballot(on);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) s_cnt[w] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int i = 0; i < 16; ++i) { const int c = s_cnt[i]; s_cnt[i] = tot; tot += c; }
    s_base = tot ? atomicAdd(A.nact, tot) : 0;
  }
  __syncthreads();
  if (on) {
    const int slot = s_base + s_cnt[w] + __popcll(m & ((1ull << lane) - 1ull));
    A.alist[2 * slot] = q;
    A.alist[2 * slot + 1] = mcl_box_of(A, 256LL * q);
  }
}

__global__ __launch_bounds__(1024) void k_mcl_active(MclArgs A, int nblk) { mcl_active_body(A, nblk, blockIdx.x); }
__global__ __launch_bounds__(1024) void k_mclb_active(MclBatch Bt) {
  unsigned w;
  const MclArgs A = Bt.a[Bt.find(blockIdx.x, w)];
  mcl_active_body(A, (int)(A.coff[A.L.nboxes] / 256), w);
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int wave_excl(int v, int lane) {  // exclusive prefix over the lanes of a wave
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  return x - v;
}

// A marked 256-cell block is one WAVE's work: lane l holds cells 4 l .. 4 l + 3 (their lc / cidx bytes are one dword each),
// sums and prefixes are wave shuffles -- no LDS, no barrier.  (As one workgroup per block, with two barriers per block,
// these two kernels took 0.145 + 0.131 ms on ~1.3e5 marked blocks.)
__device__ __forceinline__ void mcl_count_body(const MclArgs& A) {
  const int nact = *A.nact, lane = threadIdx.x & 63;
  const int nwave = gridDim.x * 4;
  for (int q = blockIdx.x * 4 + (threadIdx.x >> 6); q < nact; q += nwave) {  // marked blocks only (bsum of the others was zeroed)
    const long long blk = A.alist[2 * q];
    const int b = A.alist[2 * q + 1];
    MclGeo G;
    mcl_geo(A, b, G);
    const long long g0 = A.coff[b];
    const unsigned lin0 = (unsigned)(blk * 256 - g0) + 4u * lane;
    const long long g = g0 + lin0;
    unsigned lcw = *(const unsigned*)(A.lc + g), cw = *(const unsigned*)(A.cidx + g);
    unsigned vf = 0;
    int nv = 0, nt = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (lin0 + e >= G.ncell) break;  // the FAB's last block is padded
      const int lcv = (lcw >> (8 * e)) & 0xFF, cand = lcv >> 1;
      int bits = 0;
      if (cand) {
        int i, j, k;
        mcl_cell(G, lin0 + e, i, j, k);
        for (int d = 0; d < 3; ++d) {
          if (!((cand >> d) & 1)) continue;  // edgeTable flags an edge iff its endpoints are on different sides
          bool rev;
          if ((A.dim2 ? first_toucher2(G, A.lc + g0, i, j, k, d, rev) : first_toucher(G, A.lc + g0, i, j, k, d, rev)) >= 0) bits |= (1 << d);
        }
      }
      const int ci = (cw >> (8 * e)) & 0xFF;
      nt += (lcv & 1) ? (A.dim2 ? d_nseg[ci] : c_ntri[ci]) : 0;
      nv += __popc(bits);
      vf |= (unsigned)bits << (8 * e);
    }
    *(unsigned*)(A.vflag + g) = vf;  // (padding cells of the last block: zeros, never read as cells)
    const int tv = wave_sum(nv), tt = wave_sum(nt);
    if (lane == 0) {
      A.bsum[2 * blk] = tv;
      A.bsum[2 * blk + 1] = tt;
    }
  }
}

__global__ __launch_bounds__(256) void k_mcl_count(MclArgs A) { mcl_count_body(A); }
__global__ __launch_bounds__(256) void k_mclb_count(MclBatch Bt) {
  for (int l = 0; l < Bt.n; ++l) {
    const MclArgs A = Bt.a[l];
    mcl_count_body(A);
  }
}

// one workgroup per FAB: exclusive scan of the FAB's block sums in place (FAB-local offsets), totals to tot[b]
__device__ __forceinline__ void mcl_scan_body(const MclArgs& A, const int b) {
  __shared__ long long s_a[1024], s_b[1024];
  const int t = threadIdx.x;
  MclGeo G;
  const bool on = mcl_geo(A, b, G);
  const int nblocks = on ? (int)((G.ncell + 255u) / 256u) : 0;
  int* bsum = A.bsum + 2 * (A.coff[b] / 256);
  const int per = (nblocks + 1023) / 1024;
  const int lo = min(t * per, nblocks), hi = min(lo + per, nblocks);
  long long a = 0, c = 0;
  for (int q = lo; q < hi; ++q) { a += bsum[2 * q]; c += bsum[2 * q + 1]; }
  s_a[t] = a; s_b[t] = c;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const long long xa = t >= o ? s_a[t - o] : 0, xb = t >= o ? s_b[t - o] : 0;
    __syncthreads();
    s_a[t] += xa; s_b[t] += xb;
    __syncthreads();
  }
  long long ea = s_a[t] - a, eb = s_b[t] - c;
  for (int q = lo; q < hi; ++q) {
    const int va = bsum[2 * q], vb = bsum[2 * q + 1];
    bsum[2 * q] = (int)ea; bsum[2 * q + 1] = (int)eb;
    ea += va; eb += vb;
  }
  if (t == 1023) { A.tot[2 * b] = s_a[1023]; A.tot[2 * b + 1] = s_b[1023]; }
}

__global__ __launch_bounds__(1024) void k_mcl_scan(MclArgs A) { mcl_scan_body(A, blockIdx.x); }
__global__ __launch_bounds__(1024) void k_mclb_scan(MclBatch Bt) {
  unsigned w;
  const MclArgs A = Bt.a[Bt.find(blockIdx.x, w)];
  mcl_scan_body(A, (int)w);
}
// first vertex / triangle of every FAB inside its level's part of the output (exclusive prefix of the FAB totals; one
// workgroup per level, a level has at most a few thousand FABs): the host only needs the totals
__global__ __launch_bounds__(256) void k_mclb_base(MclBatch Bt) {
  const MclArgs& A = Bt.a[blockIdx.x];
  long long* base = const_cast<long long*>(A.base);
  __shared__ long long s_v[256], s_t[256];
  const int nb = A.L.nboxes, t = threadIdx.x, per = (nb + 255) / 256;
  const int lo = min(t * per, nb), hi = min(lo + per, nb);
  long long v = 0, c = 0;
  for (int b = lo; b < hi; ++b) {
    const bool on = A.coff[b + 1] > A.coff[b];
    v += on ? A.tot[2 * b] : 0; c += on ? A.tot[2 * b + 1] : 0;
  }
  s_v[t] = v; s_t[t] = c;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const long long xv = t >= o ? s_v[t - o] : 0, xt = t >= o ? s_t[t - o] : 0;
    __syncthreads();
    s_v[t] += xv; s_t[t] += xt;
    __syncthreads();
  }
  long long ev = s_v[t] - v, et = s_t[t] - c;
  for (int b = lo; b < hi; ++b) {
    const bool on = A.coff[b + 1] > A.coff[b];
    base[2 * b] = ev; base[2 * b + 1] = et;
    ev += on ? A.tot[2 * b] : 0; et += on ? A.tot[2 * b + 1] : 0;
  }
}

// Emission.  k_mcl_lists (marked blocks): vertex offset of every cell (kept for the triangles) and one work item per
// vertex / per triangle, parked in the output slot of that vertex (its 24-byte key) / triangle (its 12-byte id triple):
// (scratch cell, FAB, edge direction or triangle number).  k_mcl_verts / k_mcl_tris then run one thread per item --
// a marked block holds ~3 vertices per 256 cells, so per-block emission left 99 % of the lanes idle behind chains of
// dependent loads (measured 0.64 + 0.33 ms for 0.46 M vertices + 0.91 M triangles).
__device__ __forceinline__ void mcl_lists_body(const MclArgs& A, int* vkeys, int* tris) {
  const int nact = *A.nact, lane = threadIdx.x & 63;
  const int nwave = gridDim.x * 4;
  for (int q = blockIdx.x * 4 + (threadIdx.x >> 6); q < nact; q += nwave) {  // one wave per marked block, as k_mcl_count
    const long long blk = A.alist[2 * q];
    const int b = A.alist[2 * q + 1];
    MclGeo G;
    mcl_geo(A, b, G);
    const long long g = blk * 256 + 4 * lane;
    const unsigned lin0 = (unsigned)(g - A.coff[b]);
    const unsigned vf = *(const unsigned*)(A.vflag + g), lcw = *(const unsigned*)(A.lc + g), cw = *(const unsigned*)(A.cidx + g);
    int nvc[4], ntc[4], nv = 0, nt = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool cell = lin0 + e < G.ncell;  // the FAB's last block is padded
      const int ci = (cw >> (8 * e)) & 0xFF;
      nvc[e] = cell ? __popc((vf >> (8 * e)) & 7) : 0;
      ntc[e] = (cell && ((lcw >> (8 * e)) & 1)) ? (A.dim2 ? d_nseg[ci] : c_ntri[ci]) : 0;
      nv += nvc[e];
      nt += ntc[e];
    }
    int vid = A.bsum[2 * blk] + wave_excl(nv, lane);
    int* t = tris + 3LL * (A.base[2 * b + 1] + A.bsum[2 * blk + 1] + wave_excl(nt, lane));
    int* const kb = vkeys + 6LL * A.base[2 * b];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (lin0 + e >= G.ncell) break;
      A.voff[g + e] = vid;
      const long long ge = g + e;
      const int glo = (int)(unsigned)(ge & 0xffffffffLL), ghi = (int)(ge >> 32);
      const int bits = (vf >> (8 * e)) & 7;
      for (int d = 0; d < 3; ++d) {
        if (!(bits & (1 << d))) continue;
        int* k = kb + 6LL * vid;
        k[0] = glo; k[1] = ghi; k[2] = b | (d << 28);
        ++vid;
      }
      for (int r = 0; r < ntc[e]; ++r) { t[0] = glo; t[1] = ghi; t[2] = b | (r << 28); t += 3; }
    }
  }
}

__global__ __launch_bounds__(256) void k_mcl_lists(MclArgs A, int* vkeys, int* tris) { mcl_lists_body(A, vkeys, tris); }
struct MclOut { double* dv[PA_MAXB]; int* dk[PA_MAXB]; int* dt[PA_MAXB]; long long nv[PA_MAXB], nt[PA_MAXB]; };
__global__ __launch_bounds__(256) void k_mclb_lists(MclBatch Bt, MclOut O) {
  for (int l = 0; l < Bt.n; ++l) {
    if (!O.dk[l]) continue;
    const MclArgs A = Bt.a[l];
    mcl_lists_body(A, O.dk[l], O.dt[l]);
  }
}

__device__ __forceinline__ void mcl_verts_body(const MclArgs& A, double* verts, int* vkeys, long long nv, const long long vo) {
  if (vo >= nv) return;
  int* key = vkeys + 6LL * vo;
  const long long g = (long long)(unsigned)key[0] | ((long long)key[1] << 32);
  const int b = key[2] & 0x0fffffff, d = key[2] >> 28;
  MclGeo G;
  mcl_geo(A, b, G);
  const long long g0 = A.coff[b];
  int i, j, k;
  mcl_cell(G, (unsigned)(g - g0), i, j, k);
  const FabView S = mf_view(A.S, A.L.boxes[b], b);
  bool rev = false;
  if (A.dim2) first_toucher2(G, A.lc + g0, i, j, k, d, rev);
  else first_toucher(G, A.lc + g0, i, j, k, d, rev);
  const int hi_i = i + (d == 0), hi_j = j + (d == 1), hi_k = k + (d == 2);
  const int a[3] = {rev ? hi_i : i, rev ? hi_j : j, rev ? hi_k : k};
  const int e[3] = {rev ? i : hi_i, rev ? j : hi_j, rev ? k : hi_k};
  const double v1 = S(a[0], a[1], a[2], A.isocomp), v2 = S(e[0], e[1], e[2], A.isocomp);
  double* o = verts + vo * A.ncomp;
  int mode;  // 0 copy p1, 1 copy p2, 2 interpolate (VI_doIt, isosurface.cpp:257-301)
  if (fabs(A.iso - v1) < PA_EPS_DEF) mode = 0;
  else if (fabs(A.iso - v2) < PA_EPS_DEF) mode = 1;
  else if (fabs(v1 - v2) < PA_EPS_DEF) mode = 0;
  else mode = 2;
  const double mu = mode == 2 ? (A.iso - v1) / (v2 - v1) : 0.0;
  int c0 = 0;
  if (A.xyz) {  // the three coordinate components from the cell indices; the state's components follow
    double x1[3], x2[3];
    mcl_xyz(A, a, x1);
    mcl_xyz(A, e, x2);
    for (int d = 0; d < 3; ++d) o[d] = mode == 0 ? x1[d] : (mode == 1 ? x2[d] : x1[d] + mu * (x2[d] - x1[d]));
    c0 = 3;
  }
  for (int c = 0; c + c0 < A.ncomp; ++c) {
    const double a1 = S(a[0], a[1], a[2], c), a2 = S(e[0], e[1], e[2], c);
    o[c + c0] = mode == 0 ? a1 : (mode == 1 ? a2 : a1 + mu * (a2 - a1));
  }
  key[0] = i; key[1] = j; key[2] = k; key[3] = hi_i; key[4] = hi_j; key[5] = hi_k;
}

__global__ __launch_bounds__(256) void k_mcl_verts(MclArgs A, double* verts, int* vkeys, long long nv) {
  mcl_verts_body(A, verts, vkeys, nv, blockIdx.x * 256LL + threadIdx.x);
}
__global__ __launch_bounds__(256) void k_mclb_verts(MclBatch Bt, MclOut O) {
  unsigned w;
  const int l = Bt.find(blockIdx.x, w);
  const MclArgs A = Bt.a[l];
  mcl_verts_body(A, O.dv[l], O.dk[l], O.nv[l], w * 256LL + threadIdx.x);
}

__device__ __forceinline__ void mcl_tris_body(const MclArgs& A, int* tris, long long nt, const long long to) {
  if (to >= nt) return;
  int* o = tris + 3LL * to;
  const long long g = (long long)(unsigned)o[0] | ((long long)o[1] << 32);
  const int b = o[2] & 0x0fffffff, r = o[2] >> 28;
  MclGeo G;
  mcl_geo(A, b, G);
  const long long g0 = A.coff[b];
  int i, j, k;
  mcl_cell(G, (unsigned)(g - g0), i, j, k);
  const int ci = A.cidx[g];
  int out[3];
  if (A.dim2) {  // segment r of the square: two vertex ids, third entry -1
    for (int q = 0; q < 2; ++q) {
      const int e = d_seg[ci][2 * r + q];
      const long long le = g0 + lin_of(G, i + d_sq_elo[e][0], j + d_sq_elo[e][1], k);
      const int dir = d_sq_edir[e];
      out[q] = A.voff[le] + __popc(A.vflag[le] & ((1 << dir) - 1));
    }
    out[2] = -1;
  } else {
    for (int q = 0; q < 3; ++q) {
      const int e = c_tri[ci][3 * r + q];
      const long long le = g0 + lin_of(G, i + d_elo[e][0], j + d_elo[e][1], k + d_elo[e][2]);
      const int dir = d_edir[e];
      out[q] = A.voff[le] + __popc(A.vflag[le] & ((1 << dir) - 1));
    }
  }
  o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
}

__global__ __launch_bounds__(256) void k_mcl_tris(MclArgs A, int* tris, long long nt) { mcl_tris_body(A, tris, nt, blockIdx.x * 256LL + threadIdx.x); }
__global__ __launch_bounds__(256) void k_mclb_tris(MclBatch Bt, MclOut O) {
  unsigned w;
  const int l = Bt.find(blockIdx.x, w);
  const MclArgs A = Bt.a[l];
  mcl_tris_body(A, O.dt[l], O.nt[l], w * 256LL + threadIdx.x);
}

// the code bytes of the marked blocks back to zero: the invariant of pa_ctx::d_mcz (one wave per block, 4 bytes per lane and array)
__device__ __forceinline__ void mcl_clean_body(const MclArgs& A) {
  const int nact = *A.nact, lane = threadIdx.x & 63;
  const int nwave = gridDim.x * 4;
  for (int q = blockIdx.x * 4 + (threadIdx.x >> 6); q < nact; q += nwave) {
    const long long g = (long long)A.alist[2 * q] * 256 + 4 * lane;
    *(unsigned*)(A.lc + g) = 0u;
    *(unsigned*)(A.cidx + g) = 0u;
  }
}

__global__ __launch_bounds__(256) void k_mcl_clean(MclArgs A) { mcl_clean_body(A); }
__global__ __launch_bounds__(256) void k_mclb_clean(MclBatch Bt) {
  for (int l = 0; l < Bt.n; ++l) mcl_clean_body(Bt.a[l]);
}

// mask of isosurface.cpp:1540-1563: 1, and -1 on cells (ghost cells included) covered by the next finer level
__global__ __launch_bounds__(256) void k_iso_mask(DLevelView L, DMFView M, int comp, DLevelView LF, int has_fine, int ratio) {
  const int b = blockIdx.y;
  const DBox B = L.boxes[b];
  const unsigned nx = B.hi[0] - B.lo[0] + 1 + 2 * M.ng, ny = B.hi[1] - B.lo[1] + 1 + 2 * M.ng, nz = B.hi[2] - B.lo[2] + 1 + 2 * M.ng;
  // grid-stride over the FAB's cells: gridDim.x is sized by the SMALLEST FAB of the level (iso_grid_x), larger ones take more trips
  for (unsigned lin = blockIdx.x * 256u + threadIdx.x; lin < nx * ny * nz; lin += gridDim.x * 256u) {
    const unsigned r = lin / nx, kk = r / ny;
    double v = 1.0;
    if (has_fine) {
      int p[3] = {(B.lo[0] - M.ng + (int)(lin - r * nx)) * ratio, (B.lo[1] - M.ng + (int)(r - kk * ny)) * ratio, (B.lo[2] - M.ng + (int)kk) * ratio};
      if (wrap_cell(LF, p)) {  // periodic images of the coarsened fine boxes mask too (isosurface.cpp:1550-1560)
        const int o = owner_of(LF, p);
        if (o != -1) v = -1.0;
      }
    }
    M.data[M.off[b] + (long long)comp * pa_cstride((long long)nx * ny * nz, M.ncomp) + lin] = v;
  }
}

// cell-centre coordinates of every cell of every grown FAB (isosurface.cpp:1458-1465): (i + 0.5) * dx + plo
struct IsoGeom { double dx[3], plo[3]; };
__global__ __launch_bounds__(256) void k_iso_coords(DLevelView L, DMFView M, int comp0, IsoGeom Q) {
  const int b = blockIdx.y;
  const DBox B = L.boxes[b];
  const unsigned nx = B.hi[0] - B.lo[0] + 1 + 2 * M.ng, ny = B.hi[1] - B.lo[1] + 1 + 2 * M.ng, nz = B.hi[2] - B.lo[2] + 1 + 2 * M.ng;
  const long long cs = pa_cstride((long long)nx * ny * nz, M.ncomp);
  for (unsigned lin = blockIdx.x * 256u + threadIdx.x; lin < nx * ny * nz; lin += gridDim.x * 256u) {  // grid-stride, as k_iso_mask
    const unsigned r = lin / nx, kk = r / ny;
    const int p[3] = {B.lo[0] - M.ng + (int)(lin - r * nx), B.lo[1] - M.ng + (int)(r - kk * ny), B.lo[2] - M.ng + (int)kk};
    double* o = M.data + M.off[b] + (long long)comp0 * cs + lin;
#pragma unroll
    for (int d = 0; d < 3; ++d) o[d * cs] = (p[d] + 0.5) * Q.dx[d] + Q.plo[d];
  }
}

// workgroups per FAB of the per-cell kernels above: enough for the SMALLEST grown FAB of the level (the others loop), so that a
// level of boxes of 32 .. 128 cells per side does not launch the largest FAB's count for every box (2 of 3 workgroups empty on
// the bench's irregular hierarchy); equal boxes: one trip each, as before
static unsigned iso_grid_x(const pa_level* L, int ng) {
  long long nmin = 1LL << 40;
  for (const DBox& B : L->boxes) nmin = std::min(nmin, (long long)(B.hi[0] - B.lo[0] + 1 + 2 * ng) * (B.hi[1] - B.lo[1] + 1 + 2 * ng) * (B.hi[2] - B.lo[2] + 1 + 2 * ng));
  return (unsigned)std::max<long long>(1, (nmin + 255) / 256);
}

extern "C" int pa_iso_coords_level(pa_ctx* ctx, pa_mf* state, int comp0) {
  PaBind bind_(ctx);
  if (!ctx || !state) return pa_fail(ctx, "pa_iso_coords_level: null argument");
  if (comp0 < 0 || comp0 + 3 > state->ncomp) return pa_fail(ctx, "pa_iso_coords_level: component range");
  const pa_level* L = state->lev;
  const long long nmax = (long long)(L->maxn[0] + 2 * state->ng) * (L->maxn[1] + 2 * state->ng) * (L->maxn[2] + 2 * state->ng);
  if (nmax >= (1LL << 31)) return pa_fail(ctx, "pa_iso_coords_level: FAB too large");
  if (L->boxes.empty()) return 0;
  IsoGeom Q;
  for (int d = 0; d < 3; ++d) { Q.dx[d] = L->dx[d]; Q.plo[d] = L->prob_lo[d]; }
  hipLaunchKernelGGL(k_iso_coords, dim3(iso_grid_x(L, state->ng), (unsigned)L->boxes.size()), dim3(256), 0, ctx->stream, L->view, state->view, comp0, Q);
  PA_HIP(hipGetLastError());
  return 0;
}

extern "C" int pa_iso_mask_level(pa_ctx* ctx, pa_mf* mask, int comp, const pa_level* fine, int ratio) {
  PaBind bind_(ctx);
  if (!ctx || !mask) return pa_fail(ctx, "pa_iso_mask_level: null argument");
  if (comp < 0 || comp >= mask->ncomp) return pa_fail(ctx, "pa_iso_mask_level: component range");
  if (fine && ratio < 1) return pa_fail(ctx, "pa_iso_mask_level: bad refinement ratio");
  const pa_level* L = mask->lev;
  const long long nmax = (long long)(L->maxn[0] + 2 * mask->ng) * (L->maxn[1] + 2 * mask->ng) * (L->maxn[2] + 2 * mask->ng);
  if (nmax >= (1LL << 31)) return pa_fail(ctx, "pa_iso_mask_level: FAB too large");
  if (L->boxes.empty()) return 0;
  hipLaunchKernelGGL(k_iso_mask, dim3(iso_grid_x(L, mask->ng), (unsigned)L->boxes.size()), dim3(256), 0, ctx->stream, L->view, mask->view, comp,
                     fine ? fine->view : L->view, fine ? 1 : 0, ratio);
  PA_HIP(hipGetLastError());
  return 0;
}

static int mc_level_impl(pa_ctx* ctx, const pa_mf* state, const pa_mf* mask, int mcomp, const pa_box* loops, int isocomp, double isoval, int64_t* nvert,
                         int64_t* ntri, double** dev_verts, int32_t** dev_vkeys, int32_t** dev_tris, int dim2, int nomask = 0, const pa_level* fine = nullptr,
                         int ratio = 2);
// mask of isosurface.cpp:1540-1563 evaluated inside the cell pass (no mask multifab: half the bytes of the pass)
extern "C" int pa_mc_level_fine(pa_ctx* ctx, const pa_mf* state, const pa_level* fine, int ratio, const pa_box* loops, int isocomp, double isoval,
                                int64_t* nvert, int64_t* ntri, double** dev_verts, int32_t** dev_vkeys, int32_t** dev_tris) {
  PaBind bind_(ctx);
  if (fine && ratio < 1) return pa_fail(ctx, "pa_mc_level_fine: bad refinement ratio");
  return mc_level_impl(ctx, state, state, 0, loops, isocomp, isoval, nvert, ntri, dev_verts, dev_vkeys, dev_tris, 0, 1, fine, ratio);
}
extern "C" int pa_msq_level_fine(pa_ctx* ctx, const pa_mf* state, const pa_level* fine, int ratio, const pa_box* loops, int isocomp, double isoval,
                                 int64_t* nvert, int64_t* nseg, double** dev_verts, int32_t** dev_vkeys, int32_t** dev_segs) {
  PaBind bind_(ctx);
  if (fine && ratio < 1) return pa_fail(ctx, "pa_msq_level_fine: bad refinement ratio");
  return mc_level_impl(ctx, state, state, 0, loops, isocomp, isoval, nvert, nseg, dev_verts, dev_vkeys, dev_segs, 1, 1, fine, ratio);
}
extern "C" int pa_mc_level(pa_ctx* ctx, const pa_mf* state, const pa_mf* mask, int mcomp, const pa_box* loops, int isocomp, double isoval,
                           int64_t* nvert, int64_t* ntri, double** dev_verts, int32_t** dev_vkeys, int32_t** dev_tris) {
  PaBind bind_(ctx);
  return mc_level_impl(ctx, state, mask, mcomp, loops, isocomp, isoval, nvert, ntri, dev_verts, dev_vkeys, dev_tris, 0);
}
extern "C" int pa_msq_level(pa_ctx* ctx, const pa_mf* state, const pa_mf* mask, int mcomp, const pa_box* loops, int isocomp, double isoval,
                            int64_t* nvert, int64_t* nseg, double** dev_verts, int32_t** dev_vkeys, int32_t** dev_segs) {
  PaBind bind_(ctx);
  return mc_level_impl(ctx, state, mask, mcomp, loops, isocomp, isoval, nvert, nseg, dev_verts, dev_vkeys, dev_segs, 1);
}
// ---- one level's pass in two phases, so that several levels share ONE count read-back, ONE output allocation and ONE final
// synchronisation (pa_mc_hierarchy_fine); the single-level entry points run the same two phases back to back.
//   phase 1  cell pass, marked-block list, counts, per-FAB scan; the FAB totals are copied to pinned host memory (async)
//   -- the caller synchronises once, sums the totals of every level and carves the output block --
//   phase 2  work lists, vertices, triangles into the level's part of the block
struct MclWork {
  MclArgs A;
  const pa_mf* state = nullptr;
  int nb = 0, dim2 = 0;
  std::vector<long long> coff, base;
  std::vector<DBox> dl;
  size_t ncell = 0, nblk = 0, hdr = 0, bytes = 0;  // scratch of this level
  long long maxcell = 0, nv = 0, nt = 0;
  int64_t *nvert = nullptr, *ntri = nullptr;
  long long* h_tot = nullptr;                      // [nb][2] in the context's pinned buffer
  long long* d_base = nullptr;
  double* dv = nullptr; int32_t *dk = nullptr, *dt = nullptr;
  bool full_codes = false;
};

static int mc_prepare(pa_ctx* ctx, const pa_mf* state, const pa_mf* mask, int mcomp, const pa_box* loops, int isocomp, double isoval, int64_t* nvert, int64_t* ntri,
                      int dim2, int nomask, const pa_level* fine, int ratio, MclWork& W, const pa_level* xyz_crse = nullptr, int xyz = 0, int cratio = 0) {
  if (!ctx || !state || !mask || !loops || !nvert || !ntri) return pa_fail(ctx, "pa_mc_level: null argument");
  if (state->lev != mask->lev || state->ng != mask->ng) return pa_fail(ctx, "pa_mc_level: state and mask must share the level and the ghost width");
  if (!xyz && state->ncomp < (dim2 ? 3 : 4)) return pa_fail(ctx, "pa_mc_level: state needs the coordinate components + at least one field");
  if (isocomp < 0 || isocomp >= state->ncomp || mcomp < 0 || mcomp >= mask->ncomp) return pa_fail(ctx, "pa_mc_level: component range");
  const pa_level* L = state->lev;
  const int nb = (int)L->boxes.size(), ng = state->ng;
  W.state = state; W.nb = nb; W.dim2 = dim2; W.nvert = nvert; W.ntri = ntri;
  W.coff.assign((size_t)nb + 1, 0);
  W.dl.resize((size_t)nb);
  W.maxcell = 0;
  for (int b = 0; b < nb; ++b) {
    const DBox& B = L->boxes[b];
    long long nc = 1;
    bool on = true;
    for (int d = 0; d < 3; ++d) {
      nc *= B.hi[d] - B.lo[d] + 1 + 2 * ng;
      W.dl[b].lo[d] = loops[b].lo[d];
      W.dl[b].hi[d] = loops[b].hi[d];
      on = on && loops[b].lo[d] <= loops[b].hi[d];
    }
    if (on)
      for (int d = 0; d < 3; ++d) {
        const int up = (dim2 && d == 2) ? 0 : 1;  // squares have no upper plane
        if (loops[b].lo[d] < B.lo[d] - ng || loops[b].hi[d] + up > B.hi[d] + ng) return pa_fail(ctx, "pa_mc_level: loop box + 1 must lie inside the grown FAB");
        if (dim2 && d == 2 && loops[b].lo[d] != loops[b].hi[d]) return pa_fail(ctx, "pa_msq_level: the loop box must be one plane of cells");
      }
    if (nc >= (1LL << 31)) return pa_fail(ctx, "pa_mc_level: FAB too large");
    W.maxcell = std::max(W.maxcell, on ? nc : 0);
    W.coff[b + 1] = W.coff[b] + (on ? (nc + 255) / 256 * 256 : 0);
    nvert[b] = ntri[b] = 0;
  }
  W.ncell = W.nblk = W.hdr = W.bytes = 0;
  if (nb == 0 || W.maxcell == 0) return 0;
  if (nb > 0x0fffffff) return pa_fail(ctx, "pa_mc_level: too many FABs");
  W.ncell = (size_t)W.coff[nb];
  W.nblk = W.ncell / 256;
  if (W.nblk > 0x7fffffffull) return pa_fail(ctx, "pa_mc_level: level too large for one pass");
  W.hdr = ((size_t)nb * (16 + 16 + 24) + ((size_t)nb + 1) * 8 + 255) / 256 * 256;
  W.bytes = (W.hdr + 5 * W.ncell + 17 * W.nblk + 512 + 255) / 256 * 256;  // + 2 B per cell in the context's zeroed code buffer
  MclArgs& A = W.A;
  A.L = L->view; A.S = state->view; A.M = mask->view;
  A.mcomp = mcomp; A.isocomp = isocomp; A.ncomp = state->ncomp; A.iso = isoval;
  A.kseg = 32;
  A.rows = A.nslab = A.tiles = A.ldsw = 0;
  A.dim2 = dim2;
  A.nomask = nomask;
  A.has_fine = (nomask && fine) ? 1 : 0;
  A.ratio = ratio;
  A.LF = fine ? fine->view : L->view;
  A.xyz = xyz;
  A.cratio = 0;
  if (xyz) {
    A.ncomp = 3 + state->ncomp;
    for (int d = 0; d < 3; ++d) {
      A.gdx[d] = L->dx[d];
      A.gplo[d] = L->prob_lo[d];
      A.gdxc[d] = xyz_crse ? xyz_crse->dx[d] : L->dx[d];
    }
    A.cratio = xyz_crse ? cratio : 0;
  }
  return 0;
}

// pinned host memory for the FAB totals of a pass (grow-only)
static long long* mc_pinned(pa_ctx* ctx, size_t n) {
  if (ctx->h_pin_cap < n * sizeof(long long)) {
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    ctx->h_pin = nullptr; ctx->h_pin_cap = 0;
    const size_t want = std::max<size_t>(n * sizeof(long long), 1 << 16);
    if (hipHostMalloc(&ctx->h_pin, want, hipHostMallocDefault) != hipSuccess) { pa_fail(ctx, "pa_mc_level: pinned host allocation failed"); return nullptr; }
    ctx->h_pin_cap = want;
  }
  return (long long*)ctx->h_pin;
}

// Tile table of the slab cell pass for a level whose FABs differ in size (MclArgs::tiletab): {FAB, first row, rows, first plane} per
// workgroup.  The level-wide form cuts every FAB like the LARGEST one -- a 34-wide FAB of a level whose widest is 130 got
// 28-row slabs (a quarter of the 4096 cells a workgroup holds per plane) and the workgroups of the largest FAB's tile count, most of
// which left at once: on the bench's irregular hierarchy the pass ran at 1.8 TB/s against 4.4 on 128^3 boxes.  Here every FAB gets
// as many rows of ITS width as fit and only the tiles it has.  Null for levels of equal boxes (the tuned path).  Cached per level,
// ghost width and planes per workgroup.
static const WgTab* mcl_tiletab(const pa_level* L, int ng, int kseg, int cap_cells) {
  bool same = true;
  for (const DBox& B : L->boxes)
    for (int d = 0; d < 3; ++d) same = same && (B.hi[d] - B.lo[d] + 1 == L->maxn[d]);
  if (same || L->boxes.empty()) return nullptr;
  const long long key = (3LL << 56) | ((long long)ng << 40) | (long long)kseg;
  auto it = L->wgtabs.find(key);
  if (it != L->wgtabs.end()) return it->second->d ? it->second.get() : nullptr;
  std::unique_ptr<WgTab> T(new WgTab());
  std::vector<int> tab;
  for (int b = 0; b < (int)L->boxes.size(); ++b) {
    const DBox& B = L->boxes[b];
    const int nx = B.hi[0] - B.lo[0] + 1 + 2 * ng, ny = B.hi[1] - B.lo[1] + 1 + 2 * ng, nz = B.hi[2] - B.lo[2] + 1 + 2 * ng;
    const int rmax = std::max(4, (cap_cells / nx - 1) / 4 * 4);  // whole rows next to their halo row, multiple of 4 (as the level-wide form)
    const int ns0 = (ny + rmax - 1) / rmax;
    const int rows = std::min(rmax, ((ny + ns0 - 1) / ns0 + 3) / 4 * 4);
    for (int k0 = 0; k0 < nz; k0 += kseg)
      for (int j0 = 0; j0 < ny; j0 += rows) { tab.push_back(b); tab.push_back(j0); tab.push_back(rows); tab.push_back(k0); }
  }
  if (hipMalloc(&T->d, sizeof(int) * tab.size()) == hipSuccess && hipMemcpy(T->d, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice) == hipSuccess) {
    T->n = (unsigned)(tab.size() / 4);
  } else {
    if (T->d) (void)hipFree(T->d);
    T->d = nullptr;
    (void)hipGetLastError();
  }
  const WgTab* raw = T.get();
  L->wgtabs[key] = std::move(T);
  return raw->d ? raw : nullptr;
}

static int mc_phase1(pa_ctx* ctx, MclWork& W, unsigned char* scr, unsigned char* codes) {
  if (W.nb == 0 || W.maxcell == 0) return 0;
  const pa_level* L = W.state->lev;
  const int nb = W.nb, ng = W.state->ng;
  const size_t ncell = W.ncell, nblk = W.nblk;
  MclArgs& A = W.A;
  unsigned char* p = scr;
  A.tot = (long long*)p; p += 16 * (size_t)nb;
  W.d_base = (long long*)p; p += 16 * (size_t)nb;
  long long* d_coff = (long long*)p; p += 8 * ((size_t)nb + 1);
  DBox* d_loops = (DBox*)p;
  p = scr + W.hdr;
  A.voff = (int*)p; p += 4 * ncell;
  A.bsum = (int*)p; p += 8 * nblk;
  A.lc = codes;
  A.cidx = codes + ncell;
  A.vflag = p; p += ncell;
  A.bact = p; p += (nblk + 255) / 256 * 256;
  A.alist = (int*)p; p += 8 * nblk;
  A.nact = (int*)p;
  A.base = W.d_base; A.coff = d_coff; A.loops = d_loops;
  PA_HIP(hipMemcpyAsync(d_coff, W.coff.data(), 8 * ((size_t)nb + 1), hipMemcpyHostToDevice, ctx->stream));
  PA_HIP(hipMemcpyAsync(d_loops, W.dl.data(), sizeof(DBox) * (size_t)nb, hipMemcpyHostToDevice, ctx->stream));
  {
    constexpr int TY = 8;  // tile rows of the first form of the cell pass (FABs too wide for the slab form)
    const int mx = L->maxn[0] + 2 * ng, my = L->maxn[1] + 2 * ng, mz = L->maxn[2] + 2 * ng;
    auto tiles = [&](int ty) { return (unsigned)(std::max(1, (mx - 1 + 62) / 63) * std::max(1, (my - 1 + ty - 2) / (ty - 1)) * ((mz + A.kseg - 1) / A.kseg)); };
    // bact | alist | nact are adjacent and bsum precedes lc: two clears (bsum; bact .. nact)
    PA_HIP(hipMemsetAsync(A.bsum, 0, 8 * nblk, ctx->stream));
    PA_HIP(hipMemsetAsync(A.bact, 0, (size_t)((unsigned char*)A.nact - A.bact) + 4, ctx->stream));
    constexpr int NT4 = 512, GPT4 = 2;  // 4096 cells of a plane per workgroup
    if ((long long)mx * 5 <= 4LL * NT4 * GPT4 && !pa_opt().force_fallbacks) {
      // slab = as many whole rows as fit next to their halo row, evened out over the slabs of the largest FAB, multiple of 4
      const int rmax = std::max(4, ((4 * NT4 * GPT4) / mx - 1) / 4 * 4);
      const int ns0 = (my + rmax - 1) / rmax;
      A.rows = std::min(rmax, ((my + ns0 - 1) / ns0 + 3) / 4 * 4);
      A.nslab = (my + A.rows - 1) / A.rows;
      {  // enough workgroups for 8 per CU in flight, planes re-read at the segment ends <= 1 in 16
        const long long want = 6144;
        const int nseg = (int)std::min<long long>(std::max<long long>(1, mz / 16), std::max<long long>(1, (want + (long long)nb * A.nslab - 1) / ((long long)nb * A.nslab)));
        A.kseg = (mz + nseg - 1) / nseg;
      }
      A.tiles = A.nslab * ((mz + A.kseg - 1) / A.kseg);
      A.ldsw = NT4 * GPT4 + (mx >> 2) + 4;
      if (const WgTab* tt = mcl_tiletab(L, ng, A.kseg, 4 * NT4 * GPT4)) { A.tiletab = tt->d; A.ntab = tt->n; }
      const long long total = A.tiletab ? (long long)A.ntab : (long long)nb * A.tiles;
      if (total > 0x7ffffff0LL) return pa_fail(ctx, "pa_mc_level: level too large for one pass");
      const dim3 g4((unsigned)((total + 7) / 8 * 8));
      const size_t lds4 = 2 * (size_t)A.ldsw * 4;
      if (!A.nomask) hipLaunchKernelGGL((k_mcl_cells4<NT4, GPT4, 0>), g4, dim3(NT4), lds4, ctx->stream, A);
      else if (!A.has_fine) hipLaunchKernelGGL((k_mcl_cells4<NT4, GPT4, 1>), g4, dim3(NT4), lds4, ctx->stream, A);
      else hipLaunchKernelGGL((k_mcl_cells4<NT4, GPT4, 2>), g4, dim3(NT4), lds4, ctx->stream, A);
    } else {
      W.full_codes = true;  // the first form writes the codes of every cell: the code buffer is cleared as a whole afterwards
      hipLaunchKernelGGL((k_mcl_cells<TY>), dim3(tiles(TY), (unsigned)nb), dim3(64 * TY), 0, ctx->stream, A);
    }
    hipLaunchKernelGGL(k_mcl_active, dim3((unsigned)((nblk + 1023) / 1024)), dim3(1024), 0, ctx->stream, A, (int)nblk);
  }
  const dim3 grid(4096);  // persistent over the marked blocks
  hipLaunchKernelGGL(k_mcl_count, grid, dim3(256), 0, ctx->stream, A);
  hipLaunchKernelGGL(k_mcl_scan, dim3((unsigned)nb), dim3(1024), 0, ctx->stream, A);
  PA_HIP(hipGetLastError());
  PA_HIP(hipMemcpyAsync(W.h_tot, A.tot, 16 * (size_t)nb, hipMemcpyDeviceToHost, ctx->stream));
  return 0;
}

// after the synchronisation: per-FAB counts of this level, its bases inside its part of the output
static int mc_counts(pa_ctx* ctx, MclWork& W) {
  W.nv = W.nt = 0;
  if (W.nb == 0 || W.maxcell == 0) return 0;
  W.base.assign(2 * (size_t)W.nb, 0);
  for (int b = 0; b < W.nb; ++b) {
    const bool on = W.coff[b + 1] > W.coff[b];
    W.nvert[b] = on ? W.h_tot[2 * b] : 0;
    W.ntri[b] = on ? W.h_tot[2 * b + 1] : 0;
    if (W.nvert[b] > 0x7fffffffLL || W.ntri[b] > 0x7fffffffLL / 3) return pa_fail(ctx, "pa_mc_level: surface of one FAB too large for 32-bit ids");
    W.base[2 * b] = W.nv; W.base[2 * b + 1] = W.nt;
    W.nv += W.nvert[b]; W.nt += W.ntri[b];
  }
  return 0;
}
// bytes of a level's part of the output block: vertices | keys | triangles, each 256-byte aligned
static void mc_parts(const MclWork& W, size_t& bv, size_t& bk, size_t& bt) {
  bv = ((size_t)W.nv * W.A.ncomp * 8 + 255) / 256 * 256;
  bk = ((size_t)W.nv * 24 + 255) / 256 * 256;
  bt = (std::max<size_t>(8, (size_t)W.nt * 12) + 255) / 256 * 256;
}

static int mc_phase2(pa_ctx* ctx, MclWork& W, unsigned char* part) {
  if (W.nv == 0 && W.nt == 0) return 0;
  size_t bv, bk, bt;
  mc_parts(W, bv, bk, bt);
  W.dv = (double*)part;
  W.dk = (int32_t*)(part + bv);
  W.dt = (int32_t*)(part + bv + bk);
  PA_HIP(hipMemcpyAsync(W.d_base, W.base.data(), 16 * (size_t)W.nb, hipMemcpyHostToDevice, ctx->stream));
  const dim3 grid(4096);
  hipLaunchKernelGGL(k_mcl_lists, grid, dim3(256), 0, ctx->stream, W.A, W.dk, W.dt);
  if (W.nv > 0) hipLaunchKernelGGL(k_mcl_verts, dim3((unsigned)((W.nv + 255) / 256)), dim3(256), 0, ctx->stream, W.A, W.dv, W.dk, W.nv);
  if (W.nt > 0) hipLaunchKernelGGL(k_mcl_tris, dim3((unsigned)((W.nt + 255) / 256)), dim3(256), 0, ctx->stream, W.A, W.dt, W.nt);
  PA_HIP(hipGetLastError());
  return 0;
}

// the output block: a cached one that fits without wasting more than half of itself, else a new one (freed by pa_device_free)
static unsigned char* mc_block(pa_ctx* ctx, size_t need) {
  unsigned char* blockp = nullptr;
  size_t got = 0;
  int best = -1;
  for (int c = 0; c < (int)ctx->surf_cache.size(); ++c)
    if (ctx->surf_cache[c].second >= need && ctx->surf_cache[c].second <= 2 * need + (1u << 20) && (best < 0 || ctx->surf_cache[c].second < ctx->surf_cache[best].second)) best = c;
  if (best >= 0) {
    blockp = (unsigned char*)ctx->surf_cache[best].first;
    got = ctx->surf_cache[best].second;
    ctx->surf_cache.erase(ctx->surf_cache.begin() + best);
  } else {
    if (hipMalloc(&blockp, need) != hipSuccess) {  // make room: drop the cache and try once more
      blockp = nullptr;
      for (auto& c : ctx->surf_cache) (void)hipFree(c.first);
      ctx->surf_cache.clear();
      if (hipMalloc(&blockp, need) != hipSuccess) { pa_fail(ctx, "pa_mc_level: out of device memory for the surface"); return nullptr; }
    }
    got = need;
  }
  ctx->surf_live[blockp] = got;
  return blockp;
}

static_assert(sizeof(MclBatch) + sizeof(MclOut) <= 4000, "kernel arguments of the batched marching-cubes kernels");

// The levels of a hierarchy through ONE set of launches (k_mclb_*): one upload of the levels' tables, one clear, one launch per
// stage for all levels, one read-back of the FAB totals, the FAB bases computed on the device, no final synchronisation (the
// results are complete in stream order: pa_memcpy_d2h and every later call on the context wait for them).
// Scratch: [C: per level coff | loops][T: per level tot][B: per level base][Z: per level bsum | bact | alist | nact][R: per level voff | vflag]
static int mc_run_batched(pa_ctx* ctx, int nlev, MclWork* W) {
  constexpr int NT4 = 512, GPT4 = 2;
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  size_t cb = 0, tb = 0, zb = 0, rb = 0, ncodes = 0;
  for (int l = 0; l < nlev; ++l) {
    const size_t nb = (size_t)W[l].nb;
    cb += al(8 * (nb + 1) + sizeof(DBox) * nb);
    tb += 16 * nb;
    zb += al(8 * W[l].nblk) + al(W[l].nblk) + al(8 * W[l].nblk) + 256;
    rb += al(4 * W[l].ncell) + al(W[l].ncell);
    ncodes += 2 * W[l].ncell;
  }
  tb = al(tb);
  const size_t bb = tb;
  if (ensure_scr(ctx, cb + tb + bb + zb + rb)) return 1;
  PA_TRY_RET(upload_tables(ctx));
  unsigned char* pin = (unsigned char*)mc_pinned(ctx, (cb + tb) / sizeof(long long) + 8);
  if (!pin) return 1;
  if (ctx->mcz_cap < ncodes) {
    if (ctx->d_mcz) (void)hipFree(ctx->d_mcz);
    ctx->d_mcz = nullptr; ctx->mcz_cap = 0;
    PA_HIP(hipMalloc(&ctx->d_mcz, ncodes));
    ctx->mcz_cap = ncodes;
    ctx->mcz_dirty = true;
  }
  if (ctx->mcz_dirty) PA_HIP(hipMemsetAsync(ctx->d_mcz, 0, ctx->mcz_cap, ctx->stream));
  ctx->mcz_dirty = true;
  unsigned char* const S = (unsigned char*)ctx->d_scr;
  unsigned char *pc = S, *pt = S + cb, *pb = S + cb + tb, *pz = S + cb + tb + bb, *pr = S + cb + tb + bb + zb, *pcode = (unsigned char*)ctx->d_mcz;
  unsigned char* hc = pin;  // host image of region C
  MclBatch Bt;
  Bt.n = nlev;
  size_t ldsw = 0;
  for (int l = 0; l < nlev; ++l) {
    MclWork& w = W[l];
    MclArgs& A = w.A;
    const size_t nb = (size_t)w.nb;
    const pa_level* L = w.state->lev;
    const int ng = w.state->ng;
    std::memcpy(hc, w.coff.data(), 8 * (nb + 1));
    std::memcpy(hc + 8 * (nb + 1), w.dl.data(), sizeof(DBox) * nb);
    A.coff = (const long long*)pc;
    A.loops = (const DBox*)(pc + 8 * (nb + 1));
    pc += al(8 * (nb + 1) + sizeof(DBox) * nb); hc += al(8 * (nb + 1) + sizeof(DBox) * nb);
    A.tot = (long long*)pt; w.h_tot = (long long*)(pin + cb + (pt - (S + cb))); pt += 16 * nb;
    w.d_base = (long long*)pb; A.base = w.d_base; pb += 16 * nb;
    A.bsum = (int*)pz; pz += al(8 * w.nblk);
    A.bact = pz; pz += al(w.nblk);
    A.alist = (int*)pz; pz += al(8 * w.nblk);
    A.nact = (int*)pz; pz += 256;
    A.voff = (int*)pr; pr += al(4 * w.ncell);
    A.vflag = pr; pr += al(w.ncell);
    A.lc = pcode; A.cidx = pcode + w.ncell; pcode += 2 * w.ncell;
    // slab form of the cell pass (as mc_phase1)
    const int mx = L->maxn[0] + 2 * ng, my = L->maxn[1] + 2 * ng, mz = L->maxn[2] + 2 * ng;
    const int rmax = std::max(4, ((4 * NT4 * GPT4) / mx - 1) / 4 * 4);
    const int ns0 = (my + rmax - 1) / rmax;
    A.rows = std::min(rmax, ((my + ns0 - 1) / ns0 + 3) / 4 * 4);
    A.nslab = (my + A.rows - 1) / A.rows;
    {
      const long long want = 6144;
      const int nseg = (int)std::min<long long>(std::max<long long>(1, mz / 16), std::max<long long>(1, (want + (long long)nb * A.nslab - 1) / ((long long)nb * A.nslab)));
      A.kseg = (mz + nseg - 1) / nseg;
    }
    A.tiles = A.nslab * ((mz + A.kseg - 1) / A.kseg);
    A.ldsw = NT4 * GPT4 + (mx >> 2) + 4;
    if (const WgTab* tt = mcl_tiletab(L, ng, A.kseg, 4 * NT4 * GPT4)) { A.tiletab = tt->d; A.ntab = tt->n; }
    ldsw = std::max(ldsw, (size_t)A.ldsw);
    Bt.a[l] = A;
  }
  ProfScope prof(ctx, PA_TAG_MC);
  PA_HIP(hipMemcpyAsync(S, pin, cb, hipMemcpyHostToDevice, ctx->stream));
  PA_HIP(hipMemsetAsync(S + cb + tb + bb, 0, zb, ctx->stream));
  auto ranges = [&](auto count) {  // wg0 of a launch from the levels' workgroup counts
    Bt.wg0[0] = 0;
    for (int l = 0; l < nlev; ++l) Bt.wg0[l + 1] = Bt.wg0[l] + (unsigned)count(l);
    return Bt.wg0[nlev];
  };
  unsigned g = ranges([&](int l) { return ((W[l].A.tiletab ? (long long)W[l].A.ntab : (long long)W[l].nb * W[l].A.tiles) + 7) / 8 * 8; });
  hipLaunchKernelGGL((k_mclb_cells4<NT4, GPT4>), dim3(g), dim3(NT4), 2 * ldsw * 4, ctx->stream, Bt);
  g = ranges([&](int l) { return (W[l].nblk + 1023) / 1024; });
  hipLaunchKernelGGL(k_mclb_active, dim3(g), dim3(1024), 0, ctx->stream, Bt);
  hipLaunchKernelGGL(k_mclb_count, dim3(4096), dim3(256), 0, ctx->stream, Bt);
  g = ranges([&](int l) { return W[l].nb; });
  hipLaunchKernelGGL(k_mclb_scan, dim3(g), dim3(1024), 0, ctx->stream, Bt);
  hipLaunchKernelGGL(k_mclb_base, dim3((unsigned)nlev), dim3(256), 0, ctx->stream, Bt);
  PA_HIP(hipGetLastError());
  PA_HIP(hipMemcpyAsync(pin + cb, S + cb, tb, hipMemcpyDeviceToHost, ctx->stream));
  PA_HIP(hipStreamSynchronize(ctx->stream));  // the ONE read-back: FAB totals of every level
  size_t need = 0;
  for (int l = 0; l < nlev; ++l) {
    if (mc_counts(ctx, W[l])) return 1;
    if (W[l].nv || W[l].nt) { size_t bv, bk, bt; mc_parts(W[l], bv, bk, bt); need += bv + bk + bt; }
  }
  MclOut O;
  unsigned char* blockp = nullptr;
  if (need > 0) {
    blockp = mc_block(ctx, need);
    if (!blockp) return 1;
    size_t off = 0;
    for (int l = 0; l < nlev; ++l) {
      O.dv[l] = nullptr; O.dk[l] = nullptr; O.dt[l] = nullptr; O.nv[l] = W[l].nv; O.nt[l] = W[l].nt;
      if (!(W[l].nv || W[l].nt)) continue;
      size_t bv, bk, bt;
      mc_parts(W[l], bv, bk, bt);
      W[l].dv = (double*)(blockp + off); W[l].dk = (int32_t*)(blockp + off + bv); W[l].dt = (int32_t*)(blockp + off + bv + bk);
      O.dv[l] = W[l].dv; O.dk[l] = W[l].dk; O.dt[l] = W[l].dt;
      off += bv + bk + bt;
    }
    hipLaunchKernelGGL(k_mclb_lists, dim3(4096), dim3(256), 0, ctx->stream, Bt, O);
    g = ranges([&](int l) { return (W[l].nv + 255) / 256; });
    if (g) hipLaunchKernelGGL(k_mclb_verts, dim3(g), dim3(256), 0, ctx->stream, Bt, O);
    g = ranges([&](int l) { return (W[l].nt + 255) / 256; });
    if (g) hipLaunchKernelGGL(k_mclb_tris, dim3(g), dim3(256), 0, ctx->stream, Bt, O);
  }
  hipLaunchKernelGGL(k_mclb_clean, dim3(1024), dim3(256), 0, ctx->stream, Bt);
  if (hipGetLastError() != hipSuccess) {  // as mc_run's bail: the pooled block goes back, the levels' pointers are cleared
    if (blockp) { ctx->surf_live.erase(blockp); (void)hipFree(blockp); }
    for (int l = 0; l < nlev; ++l) { W[l].dv = nullptr; W[l].dk = W[l].dt = nullptr; }
    return pa_fail(ctx, "pa_mc_hierarchy_fine: emit kernels failed");
  }
  ctx->mcz_dirty = false;
  return 0;
}

// nlev levels (one for the single-level entry points): see MclWork
static int mc_run(pa_ctx* ctx, int nlev, MclWork* W) {
  {  // all levels in one set of launches when each of them takes the slab form of the cell pass with the mask evaluated in place
    bool batched = nlev > 1 && nlev <= PA_MAXB && !pa_opt().force_fallbacks;
    for (int l = 0; l < nlev && batched; ++l) {
      const int mx = W[l].state->lev->maxn[0] + 2 * W[l].state->ng;
      batched = W[l].nb > 0 && W[l].maxcell > 0 && W[l].A.nomask && !W[l].dim2 && (long long)mx * 5 <= 4LL * 512 * 2;
    }
    if (batched) return mc_run_batched(ctx, nlev, W);
  }
  size_t scr = 0, npin = 0;
  for (int l = 0; l < nlev; ++l) { scr += W[l].bytes; npin += 2 * (size_t)W[l].nb; }
  if (scr == 0) return 0;
  PA_TRY_RET(upload_tables(ctx));
  if (ensure_scr(ctx, scr)) return 1;
  long long* pin = mc_pinned(ctx, npin);
  if (!pin) return 1;
  // the code bytes (2 per cell) live in a buffer that is all zeros between calls: cleared when it is (re)allocated or when a
  // call left it dirty (an error between the cell pass and the clean-up), otherwise only the marked blocks are reset
  size_t ncodes = 0;
  for (int l = 0; l < nlev; ++l) ncodes += 2 * W[l].ncell;
  if (ctx->mcz_cap < ncodes) {
    if (ctx->d_mcz) (void)hipFree(ctx->d_mcz);
    ctx->d_mcz = nullptr; ctx->mcz_cap = 0;
    PA_HIP(hipMalloc(&ctx->d_mcz, ncodes));
    ctx->mcz_cap = ncodes;
    ctx->mcz_dirty = true;
  }
  if (ctx->mcz_dirty) PA_HIP(hipMemsetAsync(ctx->d_mcz, 0, ctx->mcz_cap, ctx->stream));
  ctx->mcz_dirty = true;  // until the clean-up of this call is enqueued
  ProfScope prof(ctx, PA_TAG_MC);
  size_t so = 0, po = 0, co = 0;
  for (int l = 0; l < nlev; ++l) {
    W[l].h_tot = pin + po;
    if (mc_phase1(ctx, W[l], (unsigned char*)ctx->d_scr + so, (unsigned char*)ctx->d_mcz + co)) return 1;
    so += W[l].bytes; po += 2 * (size_t)W[l].nb; co += 2 * W[l].ncell;
  }
  auto clean = [&]() {  // enqueue the reset of the code buffer (after the last reader)
    bool full = false;
    for (int l = 0; l < nlev; ++l) full = full || W[l].full_codes;
    if (full) return;  // stays dirty: the next call clears everything
    for (int l = 0; l < nlev; ++l)
      if (W[l].nb && W[l].maxcell) hipLaunchKernelGGL(k_mcl_clean, dim3(1024), dim3(256), 0, ctx->stream, W[l].A);
    if (hipGetLastError() == hipSuccess) ctx->mcz_dirty = false;
  };
  PA_HIP(hipStreamSynchronize(ctx->stream));  // the ONE count read-back (coff / dl are host vectors of this call)
  size_t need = 0;
  for (int l = 0; l < nlev; ++l) {
    if (mc_counts(ctx, W[l])) return 1;
    if (W[l].nv || W[l].nt) { size_t bv, bk, bt; mc_parts(W[l], bv, bk, bt); need += bv + bk + bt; }
  }
  if (need == 0) { clean(); return 0; }
  unsigned char* blockp = mc_block(ctx, need);
  if (!blockp) return 1;
  auto bail = [&](const std::string& m) { ctx->surf_live.erase(blockp); (void)hipFree(blockp); for (int l = 0; l < nlev; ++l) { W[l].dv = nullptr; W[l].dk = W[l].dt = nullptr; } return pa_fail(ctx, m); };
  size_t off = 0;
  for (int l = 0; l < nlev; ++l) {
    if (!(W[l].nv || W[l].nt)) continue;
    if (mc_phase2(ctx, W[l], blockp + off)) return bail("pa_mc_level: emit kernels failed: " + ctx->err);
    size_t bv, bk, bt;
    mc_parts(W[l], bv, bk, bt);
    off += bv + bk + bt;
  }
  clean();
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail("pa_mc_level: emit kernels failed");  // base vectors are host memory of this call
  return 0;
}

static int mc_level_impl(pa_ctx* ctx, const pa_mf* state, const pa_mf* mask, int mcomp, const pa_box* loops, int isocomp, double isoval, int64_t* nvert,
                         int64_t* ntri, double** dev_verts, int32_t** dev_vkeys, int32_t** dev_tris, int dim2, int nomask, const pa_level* fine, int ratio) {
  if (!ctx || !dev_verts || !dev_vkeys || !dev_tris) return pa_fail(ctx, "pa_mc_level: null argument");
  *dev_verts = nullptr; *dev_vkeys = nullptr; *dev_tris = nullptr;
  MclWork W;
  if (mc_prepare(ctx, state, mask, mcomp, loops, isocomp, isoval, nvert, ntri, dim2, nomask, fine, ratio, W)) return 1;
  if (mc_run(ctx, 1, &W)) return 1;
  *dev_verts = W.dv; *dev_vkeys = W.dk; *dev_tris = W.dt;  // one allocation, base = the vertex array
  return 0;
}

// isosurface.cpp:1434-1728 for ALL levels in one call: every level's cell pass / counts are enqueued back to back, the counts
// of the whole hierarchy are read back ONCE, every level's surface goes into ONE pooled allocation (*block, freed with
// pa_device_free; dev_verts[l] / dev_vkeys[l] / dev_tris[l] point into it, null for a level without surface) and the host
// waits once at the end.  states[l] on level l, masked by level l + 1 (fine_mask[l] != 0, isosurface.cpp:1540-1563) or not
// at all; loops / nvert / ntri: per level, as pa_mc_level_fine.
// pa_mc_hierarchy_fine on states WITHOUT coordinate components (see MclArgs::xyz)
extern "C" int pa_mc_hierarchy_xyz(pa_ctx* ctx, int nlev, const pa_mf* const* fields, const int32_t* fine_mask, int ratio, const pa_box* const* loops, int isocomp,
                                   double isoval, int64_t* const* nvert, int64_t* const* ntri, double** dev_verts, int32_t** dev_vkeys, int32_t** dev_tris, void** block) {
  PaBind bind_(ctx);
  if (!ctx || nlev <= 0 || !fields || !loops || !nvert || !ntri || !dev_verts || !dev_vkeys || !dev_tris || !block) return pa_fail(ctx, "pa_mc_hierarchy_xyz: null argument");
  if (ratio < 2) return pa_fail(ctx, "pa_mc_hierarchy_xyz: bad refinement ratio");
  *block = nullptr;
  std::vector<MclWork> W((size_t)nlev);
  for (int l = 0; l < nlev; ++l) {
    dev_verts[l] = nullptr; dev_vkeys[l] = nullptr; dev_tris[l] = nullptr;
    if (!fields[l]) return pa_fail(ctx, "pa_mc_hierarchy_xyz: null state");
    if (fields[l]->lev->domhi[2] == fields[l]->lev->domlo[2]) return pa_fail(ctx, "pa_mc_hierarchy_xyz: 3-D levels only (marching squares keep their coordinate components)");
    if (l > 0)  // the coarse cell centre needs the coarse level's geometry: level l must be the `ratio` refinement of level l - 1
      for (int d = 0; d < 3; ++d)
        if ((long long)(fields[l]->lev->domhi[d] - fields[l]->lev->domlo[d] + 1) != (long long)ratio * (fields[l - 1]->lev->domhi[d] - fields[l - 1]->lev->domlo[d] + 1) ||
            fields[l]->lev->domlo[d] != ratio * fields[l - 1]->lev->domlo[d])
          return pa_fail(ctx, "pa_mc_hierarchy_xyz: level " + std::to_string(l) + " is not the ratio-" + std::to_string(ratio) + " refinement of level " + std::to_string(l - 1));
    const pa_level* fine = (fine_mask && fine_mask[l] && l + 1 < nlev) ? fields[l + 1]->lev : nullptr;
    if (mc_prepare(ctx, fields[l], fields[l], 0, loops[l], isocomp, isoval, nvert[l], ntri[l], 0, 1, fine, ratio, W[l], l > 0 ? fields[l - 1]->lev : nullptr, 1, ratio)) return 1;
  }
  if (mc_run(ctx, nlev, W.data())) return 1;
  for (int l = 0; l < nlev; ++l) {
    dev_verts[l] = W[l].dv; dev_vkeys[l] = W[l].dk; dev_tris[l] = W[l].dt;
    if (W[l].dv && !*block) *block = W[l].dv;
  }
  return 0;
}

extern "C" int pa_mc_hierarchy_fine(pa_ctx* ctx, int nlev, const pa_mf* const* states, const int32_t* fine_mask, int ratio, const pa_box* const* loops, int isocomp,
                                    double isoval, int64_t* const* nvert, int64_t* const* ntri, double** dev_verts, int32_t** dev_vkeys, int32_t** dev_tris, void** block) {
  PaBind bind_(ctx);
  if (!ctx || nlev <= 0 || !states || !loops || !nvert || !ntri || !dev_verts || !dev_vkeys || !dev_tris || !block) return pa_fail(ctx, "pa_mc_hierarchy_fine: null argument");
  if (ratio < 1) return pa_fail(ctx, "pa_mc_hierarchy_fine: bad refinement ratio");
  *block = nullptr;
  std::vector<MclWork> W((size_t)nlev);
  for (int l = 0; l < nlev; ++l) {
    dev_verts[l] = nullptr; dev_vkeys[l] = nullptr; dev_tris[l] = nullptr;
    if (!states[l]) return pa_fail(ctx, "pa_mc_hierarchy_fine: null state");
    const pa_level* fine = (fine_mask && fine_mask[l] && l + 1 < nlev) ? states[l + 1]->lev : nullptr;
    if (mc_prepare(ctx, states[l], states[l], 0, loops[l], isocomp, isoval, nvert[l], ntri[l], 0, 1, fine, ratio, W[l])) return 1;
  }
  if (mc_run(ctx, nlev, W.data())) return 1;
  for (int l = 0; l < nlev; ++l) {
    dev_verts[l] = W[l].dv; dev_vkeys[l] = W[l].dk; dev_tris[l] = W[l].dt;
    if (W[l].dv && !*block) *block = W[l].dv;  // the first level with a surface starts the block
  }
  return 0;
}
