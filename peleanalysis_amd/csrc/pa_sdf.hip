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
//       each other, so marching hyperplane by hyperplane (one workgroup per grid, a barrier per
//       hyperplane, every point handled by one thread in the reference's neighbour order) gives
//       the sequential result exactly.  2 passes x 8 directions.
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

// :46-58 for one point, the 7 upwind neighbours in the reference's order
__device__ __forceinline__ void relax_point(const SdfGrid& G, int i, int j, int k, int di, int dj, int dk) {
  if (G.ntri <= 0) return;  // a grid of the batch without triangles: every point keeps its upper bound (and there is no triangle 0 to load below)
  const long long sj = G.ni, sk = (long long)G.ni * G.nj, q = (long long)k * sk + j * sj + i;
  const V3 gx = grid_point(G, i, j, k);
  float phi = G.phi[q];
  int ct = G.ct[q];
  const long long nb[7] = {q - di, q - dj * sj, q - di - dj * sj, q - dk * sk, q - di - dk * sk, q - dj * sj - dk * sk, q - di - dj * sj - dk * sk};
  // A triangle whose distance to this point is already known to be >= phi cannot pass the strict
  // `d < phi` test: the point's own closest triangle (d == phi when phi came from it -- phi always is
  // the distance to triangle ct once ct >= 0) and a triangle already tried in this call.  Skipping
  // those evaluations changes nothing in the result; after the first sweeps most neighbours share
  // the point's triangle.
  // The loads of all candidates are issued together -- the 7 neighbours' triangles, then the vertices of those that will be evaluated
  // (from the per-call gather tv: the same floats as x[tri[..]]) -- and only the comparisons run in the reference's order: a chain of
  // two memory latencies per point instead of up to 1 + 2 x 7 (the distances do not depend on phi; the skip rules only on the triangles).
  int tried[7];
  bool ev[7];
  const int ct0 = ct;
  bool changed = false;
#pragma unroll
  for (int m = 0; m < 7; ++m) tried[m] = G.ct[nb[m]];
#pragma unroll
  for (int m = 0; m < 7; ++m) {
    const int t = tried[m];
    bool skip = (t < 0) || (t == ct0);
#pragma unroll
    for (int r = 0; r < m; ++r) skip = skip || (tried[r] == t);
    ev[m] = !skip;
  }
  V3 vx[7][3];
#pragma unroll
  for (int m = 0; m < 7; ++m) {
    const long long t = ev[m] ? (long long)tried[m] : 0;  // triangle 0 exists (the sweeps run only where ntri > 0): a harmless load
#pragma unroll
    for (int c = 0; c < 3; ++c) vx[m][c] = G.tv[3 * t + c];
  }
#pragma unroll
  for (int m = 0; m < 7; ++m) {
    if (ev[m]) {
      const float d = point_triangle_distance(gx, vx[m][0], vx[m][1], vx[m][2]);
      if (d < phi) { phi = d; ct = tried[m]; changed = true; }
    }
  }
  if (changed) {
    G.phi[q] = phi;
    G.ct[q] = ct;
  }
}

// tv[3 t + c] = x[tri[3 t + c]] (round 6: one of the three dependent memory levels behind every point of every hyperplane, paid once)
__global__ __launch_bounds__(256) void k_sdf_gather_tv(const SdfGrid* grids) {
  const SdfGrid G = grids[blockIdx.y];
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < 3 * G.ntri; i += (long long)gridDim.x * 256ll) G.tv[i] = G.x[G.tri[i]];
}

// :60-85 + :169-178, the points of a hyperplane spread over the whole chip: ONE LAUNCH PER HYPERPLANE (all grids of the
// batch side by side, blockIdx.y = grid), the kernel boundary is the barrier between hyperplanes and makes one step's
// results visible to every CU of the next (per-XCD L2s are not coherent inside a launch).  Same points, same neighbour
// order, same arithmetic as the reference's sequential loops: bit-identical to its output (tests/golden/sdf_ref.npz).  16 x (ni + nj + nk - 5) launches of a few microseconds each
// instead of one workgroup walking ~6000 barriers with up to 12 points per thread behind each.
__global__ __launch_bounds__(256) void k_sdf_sweep_plane(const SdfGrid* grids, int di, int dj, int dk, int s) {
  const SdfGrid G = grids[blockIdx.y];
  const int nu = G.ni - 1, nv = G.nj - 1, nw = G.nk - 1;
  if (nu <= 0 || nv <= 0 || nw <= 0 || s >= nu + nv + nw - 2) return;
  const int wlo = max(0, s - (nu - 1) - (nv - 1)), whi = min(nw - 1, s);
  const int idx = wlo * nv + (int)(blockIdx.x * 256u + threadIdx.x);
  if (idx >= (whi + 1) * nv) return;
  const int v = idx % nv, w = idx / nv, u = s - v - w;
  if (u < 0 || u >= nu) return;
  const int i = di > 0 ? 1 + u : G.ni - 2 - u, j = dj > 0 ? 1 + v : G.nj - 2 - v, k = dk > 0 ? 1 + w : G.nk - 2 - w;
  relax_point(G, i, j, k, di, dj, dk);
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
    long long vw = 0;
    for (int g = 0; g < ngrids; ++g) {
      const pa_sdf_grid& S = grids[g];
      if (S.ntri <= 0 || S.n[0] < 2 || S.n[1] < 2 || S.n[2] < 2) continue;  // a grid without triangles stays at its upper bound either way
      nplanes = std::max(nplanes, S.n[0] + S.n[1] + S.n[2] - 5);
      vw = std::max(vw, (long long)(S.n[1] - 1) * (S.n[2] - 1));
    }
    const dim3 gs((unsigned)((vw + 255) / 256), (unsigned)ngrids);
    for (int pass = 0; pass < 2 && nplanes > 0; ++pass)
      for (int s8 = 0; s8 < 8; ++s8)
        for (int sp = 0; sp < nplanes; ++sp)
          hipLaunchKernelGGL(k_sdf_sweep_plane, gs, dim3(256), 0, ctx->stream, dg, dirs[s8][0], dirs[s8][1], dirs[s8][2], sp);
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
