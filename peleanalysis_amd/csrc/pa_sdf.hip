// pa_sdf.hip -- unsigned distance from the points of a regular grid to a triangle mesh, gfx950.
// Replaces Tools/SDFGen make_level_set3 (makelevelset3.cpp:118-185) as called per FAB from
// isosurface.cpp:1625-1626, and the sign/clip loop isosurface.cpp:1637-1650.  Results are
// bit-identical to the reference's float arithmetic AND to its sequential visiting order:
//
//   exact band (:127-145)  the reference loops over triangles in order and keeps, per grid point,
//       the smallest distance and the FIRST triangle that reached it (strict `<`).  Here: one
//       thread per triangle, `atomicMin` on a 64-bit key (float bits of d << 32 | triangle index);
//       d >= 0 so its bit pattern orders like its value, and ties resolve to the lowest index.
//   sweeps (:169-178, 60-85)  Gauss-Seidel: a point looks at the closest triangle of its 7 upwind
//       neighbours, which the same sweep has already updated.  All 7 lie on earlier hyperplanes
//       u+v+w = const of the sweep's own coordinates, and points of one hyperplane do not read
//       each other, so any order that puts a point after its 7 upwind neighbours gives the
//       sequential result exactly: block wavefronts (k_sdf_sweep_blocks).  2 passes x 8 directions.
//   The intersection counts (:146-165) only feed the sign step that the reference compiles out
//   (`#if 0`, :179-184) and are not computed.
// Float rules: no contraction (Makefile), correctly rounded `/` and sqrtf (hipcc default),
// denormals kept; the two double-precision steps of the reference (:9, :131-133) are kept in double.
#include "pa_internal.h"
#include "pa_fabview.h"
#include <cmath>
#include <cstring>
#include <vector>

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 vscale(float s, V3 w) { return {w.x * s, w.y * s, w.z * s}; }  // vec.h:121-125 (w *= a)
__device__ __forceinline__ float vdot(V3 a, V3 b) {  // vec.h:328-333
  float d = a.x * b.x;
  d += a.y * b.y;
  d += a.z * b.z;
  return d;
}
__device__ __forceinline__ float vmag2(V3 a) {  // vec.h:199-204
  float l = a.x * a.x;
  l += a.y * a.y;
  l += a.z * a.z;
  return l;
}
__device__ __forceinline__ float vdist(V3 a, V3 b) {  // vec.h:211-220
  const float ex = a.x - b.x, ey = a.y - b.y, ez = a.z - b.z;
  float d = ex * ex;
  d += ey * ey;
  d += ez * ez;
  return sqrtf(d);
}

// makelevelset3.cpp:4-18
__device__ __forceinline__ float point_segment_distance(V3 x0, V3 x1, V3 x2) {
  const V3 dx = vsub(x2, x1);
  const double m2 = (double)vmag2(dx);
  float s12 = (float)((double)vdot(vsub(x2, x0), dx) / m2);
  if (s12 < 0) s12 = 0;
  else if (s12 > 1) s12 = 1;
  return vdist(x0, vadd(vscale(s12, x1), vscale(1 - s12, x2)));
}

// makelevelset3.cpp:21-44
__device__ __forceinline__ float point_triangle_distance(V3 x0, V3 x1, V3 x2, V3 x3) {
  const V3 x13 = vsub(x1, x3), x23 = vsub(x2, x3), x03 = vsub(x0, x3);
  const float m13 = vmag2(x13), m23 = vmag2(x23), d = vdot(x13, x23);
  const float det = m13 * m23 - d * d;
  const float invdet = 1.f / (det < 1e-30f ? 1e-30f : det);
  const float a = vdot(x13, x03), b = vdot(x23, x03);
  const float w23 = invdet * (m23 * a - d * b);
  const float w31 = invdet * (m13 * b - d * a);
  const float w12 = 1 - w23 - w31;
  if (w23 >= 0 && w31 >= 0 && w12 >= 0) {
    return vdist(x0, vadd(vadd(vscale(w23, x1), vscale(w31, x2)), vscale(w12, x3)));
  } else {
    float p, q;
    if (w23 > 0) { p = point_segment_distance(x0, x1, x2); q = point_segment_distance(x0, x1, x3); }
    else if (w31 > 0) { p = point_segment_distance(x0, x1, x2); q = point_segment_distance(x0, x2, x3); }
    else { p = point_segment_distance(x0, x1, x3); q = point_segment_distance(x0, x2, x3); }
    return (q < p) ? q : p;
  }
}

struct SdfGrid {  // device-side descriptor of one make_level_set3 call
  long long ntri;
  const unsigned* tri;
  const V3* x;
  float origin[3], dx;
  int ni, nj, nk;
  float* phi;
  unsigned long long* key;  // scratch: (float bits of phi << 32) | closest triangle
  int* ct;                  // scratch: closest triangle (-1: none)
  V3* tv;                   // scratch: the three vertices of every triangle, gathered once per call (k_sdf_gather_tv): the sweeps' chain of
                            // dependent loads is then neighbour's triangle -> its vertices, without the vertex indices in between
};

__device__ __forceinline__ V3 grid_point(const SdfGrid& G, int i, int j, int k) {
  return {i * G.dx + G.origin[0], j * G.dx + G.origin[1], k * G.dx + G.origin[2]};
}
__device__ __forceinline__ float dist_to_tri(const SdfGrid& G, V3 gx, long long t) {
  return point_triangle_distance(gx, G.x[G.tri[3 * t]], G.x[G.tri[3 * t + 1]], G.x[G.tri[3 * t + 2]]);
}

__global__ __launch_bounds__(256) void k_sdf_init(const SdfGrid* grids) {
  const SdfGrid G = grids[blockIdx.y];
  const long long n = (long long)G.ni * G.nj * G.nk;
  const float far = (G.ni + G.nj + G.nk) * G.dx;  // :123
  const unsigned long long k0 = ((unsigned long long)__float_as_uint(far) << 32) | 0xFFFFFFFFull;
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) G.key[q] = k0;
}

__device__ __forceinline__ int clampi(int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); }
__device__ __forceinline__ double min3d(double a, double b, double c) { double m = (b < a) ? b : a; return (c < m) ? c : m; }  // util.h:31-33
__device__ __forceinline__ double max3d(double a, double b, double c) { double m = (a < b) ? b : a; return (m < c) ? c : m; }  // util.h:47-49

// :127-145, thread per triangle
__global__ __launch_bounds__(256) void k_sdf_band(const SdfGrid* grids, int band) {
  const SdfGrid G = grids[blockIdx.y];
  const float far = (G.ni + G.nj + G.nk) * G.dx;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < G.ntri; t += (long long)gridDim.x * blockDim.x) {
    const V3 xp = G.x[G.tri[3 * t]], xq = G.x[G.tri[3 * t + 1]], xr = G.x[G.tri[3 * t + 2]];
    const double o0 = G.origin[0], o1 = G.origin[1], o2 = G.origin[2], dx = G.dx;
    const double fip = ((double)xp.x - o0) / dx, fjp = ((double)xp.y - o1) / dx, fkp = ((double)xp.z - o2) / dx;
    const double fiq = ((double)xq.x - o0) / dx, fjq = ((double)xq.y - o1) / dx, fkq = ((double)xq.z - o2) / dx;
    const double fir = ((double)xr.x - o0) / dx, fjr = ((double)xr.y - o1) / dx, fkr = ((double)xr.z - o2) / dx;
    const int i0 = clampi((int)min3d(fip, fiq, fir) - band, 0, G.ni - 1), i1 = clampi((int)max3d(fip, fiq, fir) + band + 1, 0, G.ni - 1);
    const int j0 = clampi((int)min3d(fjp, fjq, fjr) - band, 0, G.nj - 1), j1 = clampi((int)max3d(fjp, fjq, fjr) + band + 1, 0, G.nj - 1);
    const int k0 = clampi((int)min3d(fkp, fkq, fkr) - band, 0, G.nk - 1), k1 = clampi((int)max3d(fkp, fkq, fkr) + band + 1, 0, G.nk - 1);
    for (int k = k0; k <= k1; ++k)
      for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
          const float d = point_triangle_distance(grid_point(G, i, j, k), xp, xq, xr);
          if (d < far)  // strict, like the reference's first comparison against the upper bound
            atomicMin(&G.key[((long long)k * G.nj + j) * G.ni + i], ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(unsigned)t);
        }
  }
}

// every triangle must refer to existing vertices (a bad index would be an out-of-bounds read): counts offenders
__global__ __launch_bounds__(256) void k_sdf_check(const SdfGrid* grids, const long long* nvert, int* nbad) {
  const SdfGrid G = grids[blockIdx.y];
  const unsigned nv = (unsigned)min(nvert[blockIdx.y], 0xFFFFFFFFll);
  int bad = 0;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < 3 * G.ntri; t += (long long)gridDim.x * blockDim.x) bad += G.tri[t] >= nv;
  if (bad) atomicAdd(nbad, bad);
}

__global__ __launch_bounds__(256) void k_sdf_unpack(const SdfGrid* grids) {
  const SdfGrid G = grids[blockIdx.y];
  const long long n = (long long)G.ni * G.nj * G.nk;
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
    const unsigned long long k = G.key[q];
    G.phi[q] = __uint_as_float((unsigned)(k >> 32));
    G.ct[q] = (int)(unsigned)(k & 0xFFFFFFFFull);  // 0xFFFFFFFF -> -1
  }
}

// tv[3 t + c] = x[tri[3 t + c]] (round 6: one of the three dependent memory levels behind every point of every hyperplane, paid once)
__global__ __launch_bounds__(256) void k_sdf_gather_tv(const SdfGrid* grids) {
  const SdfGrid G = grids[blockIdx.y];
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < 3 * G.ntri; i += (long long)gridDim.x * 256ll) G.tv[i] = G.x[G.tri[i]];
}

// :60-85 + :169-178 + :46-58.  BLOCK WAVEFRONTS (round 6): the sweep's own coordinates (u, v, w) are cut into blocks of B^3 points; a launch
// takes the blocks of one block hyperplane bu + bv + bw = S (blockIdx.y = grid of the batch) -- the kernel boundary makes one plane's
// results visible to every CU of the next (per-XCD L2s are not coherent inside a launch) -- and ONE workgroup walks its block's 3 B - 2
// inner hyperplanes with a barrier between them, the block's closest-triangle indices (with the layer of upwind neighbours) and distances
// in LDS.  A point still comes after its 7 upwind neighbours -- inside the block by the inner hyperplanes, across blocks by the launches
// -- and no point of a sweep is written after a later point of the same sweep read it, so the result is the sequential one, bit for bit
// (tests/golden/sdf_ref.npz).  Until round 6: one launch per hyperplane of POINTS, a thread per point walking its seven candidate
// triangles one after the other -- 16 x (ni + nj + nk - 5) = 6160 launches of 12 us for a 130^3 grid, of which ~6 us were that chain of
// seven dependent evaluations at one wave per SIMD and ~5 the launch-to-launch gap (72.5-76 ms; 7.9 ms per grid in a batch of 16).
// Now 16 x (nbu + nbv + nbw - 2) launches, an inner step = LDS round + ONE global latency + ONE evaluation (eight lanes per point,
// below): 34-39 ms per grid, 7.5 per grid in a batch of 16; with consecutive sweeps overlapping 28.7 / 7.2 (profiles/r06_sdf_blocks.txt).
// CONSECUTIVE SWEEPS OVERLAP: sweep m + 1 may take a block while sweep m is still under way elsewhere, as long as every point it reads or
// writes (itself and its neighbours: one cell in every direction) has been left behind by sweep m.  In launches: block plane S of sweep
// m + 1 goes with plane S + D of sweep m, D = sum over the axes on which the two directions differ of (blocks along that axis - 1), + 4
// (pa_sdf_level_set3 has the derivation) -- a whole sweep when all three axes flip, a third of one when one does.  A launch then carries
// up to three sweeps (blockIdx.z), each on its own block plane.
struct SdfSweeps { int n; int dir[3][3]; int S[3]; };
template <int B>
__global__ __launch_bounds__(B == 8 ? 384 : 128) void k_sdf_sweep_blocks(const SdfGrid* grids, SdfSweeps P) {
  const int di = P.dir[blockIdx.z][0], dj = P.dir[blockIdx.z][1], dk = P.dir[blockIdx.z][2], S = P.S[blockIdx.z];
  const SdfGrid G = grids[blockIdx.y];
  if (G.ntri <= 0) return;
  const int nu = G.ni - 1, nv = G.nj - 1, nw = G.nk - 1;
  if (nu <= 0 || nv <= 0 || nw <= 0) return;
  const int nbu = (nu + B - 1) / B, nbv = (nv + B - 1) / B, nbw = (nw + B - 1) / B;
  if (S > nbu + nbv + nbw - 3) return;
  const int bw_lo = max(0, S - (nbu - 1) - (nbv - 1)), bw_hi = min(nbw - 1, S);
  const int idx = bw_lo * nbv + (int)blockIdx.x;
  if (idx >= (bw_hi + 1) * nbv) return;  // (uniform for the workgroup, as every exit above)
  const int bv = idx % nbv, bw = idx / nbv, bu = S - bv - bw;
  if (bu < 0 || bu >= nbu) return;
  constexpr int B1 = B + 1;
  const int NT = (int)blockDim.x;  // 8 lanes x the largest inner plane's points, rounded up to whole waves (B = 8: 384, B = 4: 128)
  __shared__ int s_ct[B1 * B1 * B1];   // [lw + 1][lv + 1][lu + 1], lu = -1: the upwind neighbour layer (other blocks' points or the grid's rim)
  __shared__ float s_phi[B * B * B];
  const int u0 = bu * B, v0 = bv * B, w0 = bw * B;
  const long long sj = G.ni, sk = (long long)G.ni * G.nj;
  for (int t = threadIdx.x; t < B1 * B1 * B1; t += NT) {
    const int lu = t % B1 - 1, lv = (t / B1) % B1 - 1, lw = t / (B1 * B1) - 1;
    const int u = u0 + lu, v = v0 + lv, w = w0 + lw;  // u = -1: the rim point the first plane reads
    int c = -1;
    if (u < nu && v < nv && w < nw) {
      const int i = di > 0 ? 1 + u : G.ni - 2 - u, j = dj > 0 ? 1 + v : G.nj - 2 - v, k = dk > 0 ? 1 + w : G.nk - 2 - w;
      const long long q = (long long)k * sk + j * sj + i;
      c = G.ct[q];
      if (lu >= 0 && lv >= 0 && lw >= 0) s_phi[(lw * B + lv) * B + lu] = G.phi[q];
    }
    s_ct[t] = c;
  }
  __syncthreads();
  // EIGHT LANES PER POINT: lane m < 7 of a point's group evaluates the distance to the closest triangle of upwind neighbour m (the
  // distances do not depend on phi, the skip rules only on the triangles), then the group's first lane runs the reference's seven strict
  // comparisons in its order on those values -- the same floats compared in the same order, one triangle evaluation deep instead of seven
  // (a point's chain of seven evaluations, ~6 us at one wave per SIMD, was most of a hyperplane launch's 12 us)
  // the points of inner plane s, compact: group g of 8 lanes takes the g-th point of the plane, so a wave whose groups are all past the
  // plane's count skips the step (a fixed column -> group mapping had every wave run every step: 2-3 x the issue slots, which is what a
  // batch of grids pays for)
  constexpr int PMAX = (3 * B * B + 3) / 4 + 1;
  __shared__ unsigned char s_pts[3 * B - 2][PMAX];
  __shared__ int s_cnt[3 * B - 2];
  if (threadIdx.x < 64) {  // the first wave: lane = column (v, w) of the block (B * B <= 64), a ballot per plane
    const int c = (int)threadIdx.x, cv = c % B, cw = c / B;
    const bool colin = c < B * B && v0 + cv < nv && w0 + cw < nw;
    for (int sp = 0; sp < 3 * B - 2; ++sp) {
      const int lu = sp - cv - cw;
      const bool in = colin && lu >= 0 && lu < B && u0 + lu < nu;
      const unsigned long long mk = __ballot(in);
      if (in) s_pts[sp][__popcll(mk & ((1ull << c) - 1ull))] = (unsigned char)c;
      if (c == 0) s_cnt[sp] = __popcll(mk);
    }
  }
  __syncthreads();
  const int grp = (int)threadIdx.x >> 3, m = (int)threadIdx.x & 7;
  const int gslot = min(grp, PMAX - 1);
  int cnt = s_cnt[0], colx = s_pts[0][gslot];
  for (int s = 0; s < 3 * B - 2; ++s) {
    // the next plane's count and point: requested now, they do not wait for this plane's barrier
    const int sn = min(s + 1, 3 * B - 3);
    const int cnt_n = s_cnt[sn], colx_n = s_pts[sn][gslot];
    const int cnt_c = cnt, colx_c = colx;
    cnt = cnt_n; colx = colx_n;
    if ((((int)threadIdx.x >> 6) << 3) >= cnt_c) { __syncthreads(); continue; }  // (uniform for the wave: its first group is past the plane)
    const bool on = grp < cnt_c;  // (the same for the 8 lanes of a group)
    const int colq = on ? colx_c : (int)s_pts[s][0];
    const int lv = colq % B, lw = colq / B, lus = s - lv - lw;
    const int u = u0 + lus, v = v0 + lv, w = w0 + lw;
    const int i = di > 0 ? 1 + u : G.ni - 2 - u, j = dj > 0 ? 1 + v : G.nj - 2 - v, k = dk > 0 ? 1 + w : G.nk - 2 - w;
    const int o = ((lw + 1) * B1 + (lv + 1)) * B1 + (lus + 1);
    const int ct0 = s_ct[o];
    // neighbour m: one step back in u, v, u + v, w, u + w, v + w, u + v + w (makelevelset3.cpp:46-58's order = m + 1 in binary)
    const int mo = (m + 1) & 7;
    const int noff = (mo & 1) + ((mo >> 1) & 1) * B1 + ((mo >> 2) & 1) * B1 * B1;
    const int mine = (on && m < 7) ? s_ct[o - noff] : -1;
    bool skip = mine < 0 || mine == ct0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int other = __shfl(mine, r, 8);
      skip = skip || (r < m && other == mine);
    }
    float d = 0.f;
    if (!skip) {
      const V3 gx = grid_point(G, i, j, k);
      const V3 a = G.tv[3 * (long long)mine], b = G.tv[3 * (long long)mine + 1], c = G.tv[3 * (long long)mine + 2];
      d = point_triangle_distance(gx, a, b, c);
    }
    float phi = on ? s_phi[(lw * B + lv) * B + lus] : 0.f;
    int ct = ct0;
    bool changed = false;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const float dr = __shfl(d, r, 8);
      const int tr = __shfl(mine, r, 8);
      const int sr = __shfl((int)skip, r, 8);
      if (!sr && dr < phi) { phi = dr; ct = tr; changed = true; }
    }
    if (on && m == 0 && changed) {
      const long long q = (long long)k * sk + j * sj + i;
      s_phi[(lw * B + lv) * B + lus] = phi;
      s_ct[o] = ct;
      G.phi[q] = phi;
      G.ct[q] = ct;
    }
    __syncthreads();
  }
}

// (Round 5, second session, measured and not kept: ALL sweeps of a batch in one launch with a barrier per XCD -- the 32 CUs of an XCD share
// one L2, a step's plain stores are written through to it and agent-scope relaxed loads find them there without any cache flush
// (tools/bench/xcdsync.hip: 1-2 us per barrier + step, 0 wrong values in 2000 dependent steps; buffer_inv sc0 does not drop the vector
// cache, sc1 costs 9-30 us).  Bit-identical, but 125-143 ms per 130^3 grid against 76 ms and 15.5 ms per grid in a batch of 16
// against 8.0: a hyperplane of ~10^4 points is throughput work -- ~18 us on the 32 CUs of one XCD, ~3 on the chip -- so the launch
// per hyperplane, whose kernel uses all eight XCDs, stays.)
static int ensure_scr_sdf(pa_ctx* ctx, size_t bytes) {
  if (ctx->scr_cap >= bytes) return 0;
  if (ctx->d_scr) (void)hipFree(ctx->d_scr);
  ctx->d_scr = nullptr;
  ctx->scr_cap = 0;
  PA_HIP(hipMalloc(&ctx->d_scr, bytes));
  ctx->scr_cap = bytes;
  return 0;
}

extern "C" int pa_sdf_level_set3(pa_ctx* ctx, int ngrids, const pa_sdf_grid* grids, int exact_band) {
  PaBind bind_(ctx);
  if (!ctx || ngrids < 0 || (ngrids > 0 && !grids)) return pa_fail(ctx, "pa_sdf_level_set3: null argument");
  if (exact_band < 0) return pa_fail(ctx, "pa_sdf_level_set3: negative exact_band");
  if (ngrids == 0) return 0;
  size_t cells = 0;
  long long max_cells = 0, max_tri = 0;
  for (int g = 0; g < ngrids; ++g) {
    const pa_sdf_grid& S = grids[g];
    if (S.n[0] <= 0 || S.n[1] <= 0 || S.n[2] <= 0) return pa_fail(ctx, "pa_sdf_level_set3: empty grid");
    if (!S.phi) return pa_fail(ctx, "pa_sdf_level_set3: null phi");
    if (S.ntri < 0 || S.nvert < 0 || (S.ntri > 0 && (!S.tri || !S.x))) return pa_fail(ctx, "pa_sdf_level_set3: bad mesh");
    if (S.ntri >= 0xFFFFFFFFll) return pa_fail(ctx, "pa_sdf_level_set3: too many triangles");
    const long long n = (long long)S.n[0] * S.n[1] * S.n[2];
    if (n > (1ll << 31)) return pa_fail(ctx, "pa_sdf_level_set3: grid too large");
    cells += (size_t)n;
    max_cells = std::max(max_cells, n);
    max_tri = std::max(max_tri, (long long)S.ntri);
  }
  // scratch: descriptors + vertex counts | keys (8 B per point) | closest triangle (4 B per point) | gathered triangle vertices (36 B per triangle)
  const size_t desc_bytes = ((size_t)ngrids * (sizeof(SdfGrid) + sizeof(long long)) + 255) / 256 * 256;
  size_t tris = 0;
  for (int g = 0; g < ngrids; ++g) tris += (size_t)grids[g].ntri;
  const size_t tv_at = (desc_bytes + cells * 12 + 255) / 256 * 256;
  if (ensure_scr_sdf(ctx, tv_at + tris * 3 * sizeof(V3) + 256)) return 1;
  unsigned char* base = (unsigned char*)ctx->d_scr;
  unsigned long long* keys = (unsigned long long*)(base + desc_bytes);
  int* cts = (int*)(base + desc_bytes + cells * 8);
  V3* tvs = (V3*)(base + tv_at);
  size_t tat = 0;
  std::vector<SdfGrid> h((size_t)ngrids);
  size_t at = 0;
  for (int g = 0; g < ngrids; ++g) {
    const pa_sdf_grid& S = grids[g];
    SdfGrid& D = h[(size_t)g];
    D.ntri = S.ntri; D.tri = S.tri; D.x = (const V3*)S.x;
    for (int d = 0; d < 3; ++d) D.origin[d] = S.origin[d];
    D.dx = S.dx; D.ni = S.n[0]; D.nj = S.n[1]; D.nk = S.n[2];
    D.phi = S.phi; D.key = keys + at; D.ct = cts + at;
    D.tv = tvs + tat;
    tat += 3 * (size_t)S.ntri;
    at += (size_t)S.n[0] * S.n[1] * S.n[2];
  }
  std::vector<long long> hnv((size_t)ngrids);
  for (int g = 0; g < ngrids; ++g) hnv[(size_t)g] = grids[g].nvert;
  long long* dnv = (long long*)(base + (size_t)ngrids * sizeof(SdfGrid));
  PA_HIP(hipMemcpyAsync(base, h.data(), (size_t)ngrids * sizeof(SdfGrid), hipMemcpyHostToDevice, ctx->stream));
  PA_HIP(hipMemcpyAsync(dnv, hnv.data(), (size_t)ngrids * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
  const SdfGrid* dg = (const SdfGrid*)base;
  const unsigned gx_cells = (unsigned)std::min<long long>((max_cells + 255) / 256, 4096);
  if (max_tri > 0) {  // shape check on the device data before a kernel dereferences an index
    int* dbad = ctx->d_flags + 8;
    int hbad = 0;
    PA_HIP(hipMemsetAsync(dbad, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_sdf_check, dim3((unsigned)std::min<long long>((3 * max_tri + 255) / 256, 4096), (unsigned)ngrids), dim3(256), 0, ctx->stream, dg, dnv, dbad);
    PA_HIP(hipMemcpyAsync(&hbad, dbad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PA_HIP(hipStreamSynchronize(ctx->stream));
    if (hbad) return pa_fail(ctx, "pa_sdf_level_set3: " + std::to_string(hbad) + " triangle vertex indices are out of range");
  } else {
    PA_HIP(hipStreamSynchronize(ctx->stream));  // h goes out of scope
  }
  hipLaunchKernelGGL(k_sdf_init, dim3(gx_cells, (unsigned)ngrids), dim3(256), 0, ctx->stream, dg);
  if (max_tri > 0) {
    const unsigned gx_tri = (unsigned)std::min<long long>((max_tri + 255) / 256, 4096);
    hipLaunchKernelGGL(k_sdf_band, dim3(gx_tri, (unsigned)ngrids), dim3(256), 0, ctx->stream, dg, exact_band);
  }
  hipLaunchKernelGGL(k_sdf_unpack, dim3(gx_cells, (unsigned)ngrids), dim3(256), 0, ctx->stream, dg);
  if (max_tri > 0) hipLaunchKernelGGL(k_sdf_gather_tv, dim3((unsigned)std::min<long long>((3 * max_tri + 255) / 256, 4096), (unsigned)ngrids), dim3(256), 0, ctx->stream, dg);
  if (max_tri > 0) {
    static const int dirs[8][3] = {{1, 1, 1}, {-1, -1, -1}, {1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {-1, 1, -1}, {1, -1, -1}, {-1, 1, 1}};
    int nplanes = 0;
    long long vw_sum = 0;
    for (int g = 0; g < ngrids; ++g) {
      const pa_sdf_grid& S = grids[g];
      if (S.ntri <= 0 || S.n[0] < 2 || S.n[1] < 2 || S.n[2] < 2) continue;  // a grid without triangles stays at its upper bound either way
      nplanes = std::max(nplanes, S.n[0] + S.n[1] + S.n[2] - 5);
      vw_sum += (long long)(S.n[1] - 1) * (S.n[2] - 1);
    }
    // blocks of 8^3 points; 4^3 for a call whose hyperplanes are small (one 130^3 grid: 970 inner steps per sweep against 1078, shorter ones)
    const int blk = vw_sum <= 20000 ? 4 : 8;
    int mb[3] = {0, 0, 0};
    for (int g = 0; g < ngrids; ++g) {
      const pa_sdf_grid& S = grids[g];
      if (S.ntri <= 0 || S.n[0] < 2 || S.n[1] < 2 || S.n[2] < 2) continue;
      for (int d = 0; d < 3; ++d) mb[d] = std::max(mb[d], (S.n[d] - 1 + blk - 1) / blk);
    }
    const int nbp = mb[0] + mb[1] + mb[2] - 2;
    // Start of sweep m + 1 relative to sweep m, in block planes.  A point p of sweep m + 1 (block plane sum_a floor(u'_a / B), u' its
    // coordinates in that sweep) conflicts with sweep m only at points q within one cell of p, which sweep m takes on a block plane
    // <= sum_a floor(u_a(p) / B) + 3; on an axis both sweeps walk the same way u = u', on a flipped one u = nu - 2 - u', so the
    // difference of the two plane indices is at most sum over flipped axes of (blocks - 1): with D = that + 4 every such q is behind
    // sweep m when p is taken (strictly earlier launch).  Sweeps further apart are covered by the sums of their neighbours' offsets.
    int T[16];
    T[0] = 0;
    for (int m = 0; m + 1 < 16 && nplanes > 0; ++m) {
      int d = 4;
      for (int a = 0; a < 3; ++a)
        if (dirs[m % 8][a] != dirs[(m + 1) % 8][a]) d += std::max(mb[a] - 1, 0);
      T[m + 1] = T[m] + std::min(nbp, d);
    }
    for (int m = 2; m + 1 < 16 && nplanes > 0; ++m)  // a launch carries three sweeps at most (the reference's order of directions never
      if (T[m + 1] - T[m - 2] < nbp) {                // has more than two under way: every other transition flips all three axes)
        for (int q = 0; q + 1 < 16; ++q) T[q + 1] = T[q] + nbp;
        break;
      }
    for (int tau = 0; nplanes > 0 && nbp > 0 && tau < T[15] + nbp; ++tau) {
      SdfSweeps P;
      P.n = 0;
      for (int m = 0; m < 16 && P.n < 3; ++m)
        if (tau >= T[m] && tau - T[m] < nbp) {
          for (int a = 0; a < 3; ++a) P.dir[P.n][a] = dirs[m % 8][a];
          P.S[P.n++] = tau - T[m];
        }
      if (!P.n) continue;
      const dim3 gb((unsigned)(mb[1] * mb[2]), (unsigned)ngrids, (unsigned)P.n);
      if (blk == 8) hipLaunchKernelGGL(k_sdf_sweep_blocks<8>, gb, dim3(384), 0, ctx->stream, dg, P);
      else hipLaunchKernelGGL(k_sdf_sweep_blocks<4>, gb, dim3(128), 0, ctx->stream, dg, P);
    }
  }
  PA_HIP(hipGetLastError());
  // the scratch (descriptors) must outlive the kernels; it is only re-used by later calls on this stream
  return 0;
}

// isosurface.cpp:1637-1650: d = sgn * min(dmax, phi), sgn = state(isoComp) < isoVal ? -1 : +1
__global__ __launch_bounds__(256) void k_sdf_signed(FabView S, int isocomp, double isoval, double dmax, const float* __restrict__ phi, DBox vb, FabView D, int dcomp) {
  const int ni = vb.hi[0] - vb.lo[0] + 1, nj = vb.hi[1] - vb.lo[1] + 1, nk = vb.hi[2] - vb.lo[2] + 1;
  const long long n = (long long)ni * nj * nk;
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
    const int iL = (int)(q % ni), jL = (int)((q / ni) % nj), kL = (int)(q / ((long long)ni * nj));
    const int i = vb.lo[0] + iL, j = vb.lo[1] + jL, k = vb.lo[2] + kL;
    const double p = (double)phi[q];
    const double abs_d = (p < dmax) ? p : dmax;  // std::min(dmax, Real(phi))
    const int sgn = S(i, j, k, isocomp) < isoval ? -1 : +1;
    D(i, j, k, dcomp) = sgn * abs_d;
  }
}

extern "C" int pa_sdf_signed_fab(pa_ctx* ctx, pa_box vbox, const float* dev_phi, const pa_fab* state, int isocomp, double isoval, double dmax,
                                 pa_fab* dist, int dcomp) {
  PaBind bind_(ctx);
  if (!ctx || !dev_phi || !state || !dist) return pa_fail(ctx, "pa_sdf_signed_fab: null argument");
  std::string why;
  if (!fab_covers(*state, vbox, 0, isocomp, 1, why) || !fab_covers(*dist, vbox, 0, dcomp, 1, why)) return pa_fail(ctx, "pa_sdf_signed_fab: " + why);
  const long long n = (long long)(vbox.hi[0] - vbox.lo[0] + 1) * (vbox.hi[1] - vbox.lo[1] + 1) * (vbox.hi[2] - vbox.lo[2] + 1);
  const unsigned g = (unsigned)std::min<long long>((n + 255) / 256, 65535);
  hipLaunchKernelGGL(k_sdf_signed, dim3(g), dim3(256), 0, ctx->stream, fab_view(*state), isocomp, isoval, dmax, dev_phi, to_dbox(vbox), fab_view(*dist), dcomp);
  PA_HIP(hipGetLastError());
  return 0;
}
