// pa_isomerge.hip -- the global node / element sets of the isosurface (isosurface.cpp:1687-1726 insertion into
// std::set<Node> / std::set<Element>, Node::operator< :834-873 with its 1e-15 tolerance, Element :877-927, the renumbering
// :1751-1812) on the device.  The reference inserts FAB after FAB, vertex after vertex: a vertex that lies within 1e-15
// (Euclidean) of a node already in the set IS that node, otherwise it becomes a new node whose id is its insertion rank;
// elements are id triples rotated so the smallest id leads, degenerate ones dropped, unique, in lexicographic order.
// Restated as data-parallel passes over the RAW vertex sequence (insertion order = fragment order, then vertex order):
//   1. key      position -> cell of a 1e-14 grid (floor(p / H)) -> 64-bit hash; radix sort of (hash, raw index)
//   2. probe    every vertex looks up its own cell, and the neighbour across each face it is within 2e-15 of (what a
//               node within 1e-15 could sit in), and keeps dup = the SMALLEST earlier raw vertex within 1e-15 (or none)
//   3. resolve  dup < 0: the vertex is a new node.  Otherwise it joins dup's node -- provided dup is itself a node.
//               If it is not (closeness is not transitive inside that cluster: a ~ b, b ~ c, a !~ c) the sequential
//               answer depends on the chain, so the call gives up (return code 2) and the caller runs the host merge;
//               with the usual clusters (copies of one edge vertex interpolated by neighbouring FABs, <= 3 ulp apart)
//               this never happens.  Node ids = exclusive scan of the new-node flags.
//   4. elements raw ids -> node ids, rotate, drop degenerate, stable radix sorts (third id, then first:second), unique.
// The predicate sqrt(a*a + b*b + c*c) < 1e-15 of the host merge (tools/common/pa_isomerge.h) is evaluated as
// (a*a + b*b + c*c) <= smax, smax = the largest double whose HOST square root is below 1e-15 (found once by the caller
// side of this file with std::sqrt / std::nextafter), so no device sqrt rounding enters.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <cmath>
#include "pa_internal.h"

namespace {
struct FragTab {
  const double* const* verts;   // [nfrag] device pointers
  const int32_t* const* tris;
  const long long* vbase;       // [nfrag + 1]
  const long long* tbase;       // [nfrag + 1]
  int nfrag, ncomp;
};
__device__ __forceinline__ int frag_of(const long long* base, int n, long long x) {  // last f with base[f] <= x
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (base[mid] <= x) lo = mid; else hi = mid - 1;
  }
  return lo;
}
constexpr double MH = 1.0e-14, MEPS = 1.0e-15;
__device__ __forceinline__ unsigned long long cell_hash(long long x, long long y, long long z) {
  unsigned long long h = (unsigned long long)x * 0x9E3779B97F4A7C15ull;
  h ^= (unsigned long long)y * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
  h ^= (unsigned long long)z * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 32;
  return h;
}

__global__ __launch_bounds__(256) void k_im_keys(FragTab T, long long N, double* P, unsigned long long* hash, int* idx) {
  const long long v = blockIdx.x * 256LL + threadIdx.x;
  if (v >= N) return;
  const int f = frag_of(T.vbase, T.nfrag, v);
  const double* s = T.verts[f] + (v - T.vbase[f]) * T.ncomp;
  const double x = s[0], y = s[1], z = s[2];
  P[3 * v] = x; P[3 * v + 1] = y; P[3 * v + 2] = z;
  hash[v] = cell_hash((long long)floor(x / MH), (long long)floor(y / MH), (long long)floor(z / MH));
  idx[v] = (int)v;
}

__global__ __launch_bounds__(256) void k_im_probe(long long N, const double* P, const unsigned long long* shash, const int* sidx, double smax, int* dup) {
  const long long v = blockIdx.x * 256LL + threadIdx.x;
  if (v >= N) return;
  const double p[3] = {P[3 * v], P[3 * v + 1], P[3 * v + 2]};
  long long g[3];
  int lo[3], hi[3];
  for (int d = 0; d < 3; ++d) {
    g[d] = (long long)floor(p[d] / MH);
    const double r = p[d] - (double)g[d] * MH;  // position inside the cell, up to the rounding of the product
    lo[d] = (r < 2 * MEPS) ? -1 : 0;
    hi[d] = (r > MH - 2 * MEPS) ? 1 : 0;
  }
  int best = -1;
  for (int dz = lo[2]; dz <= hi[2]; ++dz)
    for (int dy = lo[1]; dy <= hi[1]; ++dy)
      for (int dx = lo[0]; dx <= hi[0]; ++dx) {
        const unsigned long long h = cell_hash(g[0] + dx, g[1] + dy, g[2] + dz);
        long long a = 0, b = N;  // lower bound of h
        while (a < b) {
          const long long m = (a + b) >> 1;
          if (shash[m] < h) a = m + 1; else b = m;
        }
        for (; a < N && shash[a] == h; ++a) {
          const int c = sidx[a];
          if (c >= v) break;  // equal hashes keep their raw order (stable sort): nothing earlier follows
          const double qa = P[3 * (long long)c] - p[0], qb = P[3 * (long long)c + 1] - p[1], qc = P[3 * (long long)c + 2] - p[2];
          if (qa * qa + qb * qb + qc * qc <= smax && (best < 0 || c < best)) best = c;
        }
      }
  dup[v] = best;
}

__global__ __launch_bounds__(256) void k_im_flags(long long N, const int* dup, int* isnew, int* ambiguous) {
  const long long v = blockIdx.x * 256LL + threadIdx.x;
  if (v >= N) return;
  const int d = dup[v];
  isnew[v] = d < 0 ? 1 : 0;
  if (d >= 0 && dup[d] >= 0) *ambiguous = 1;
}

__global__ __launch_bounds__(256) void k_im_nodes(FragTab T, long long N, const int* dup, const int* newid, int* nid, double* nodes) {
  const long long v = blockIdx.x * 256LL + threadIdx.x;
  if (v >= N) return;
  const int d = dup[v];
  nid[v] = newid[d < 0 ? v : d];
  if (d >= 0) return;
  const int f = frag_of(T.vbase, T.nfrag, v);
  const double* s = T.verts[f] + (v - T.vbase[f]) * T.ncomp;
  double* o = nodes + (long long)newid[v] * T.ncomp;
  for (int c = 0; c < T.ncomp; ++c) o[c] = s[c];
}

// element t of the raw sequence -> node ids, smallest first (orientation kept); keep = not degenerate
__global__ __launch_bounds__(256) void k_im_elts(FragTab T, long long M, const int* nid, unsigned long long* k01, unsigned* k2, int* keep) {
  const long long t = blockIdx.x * 256LL + threadIdx.x;
  if (t >= M) return;
  const int f = frag_of(T.tbase, T.nfrag, t);
  const int32_t* s = T.tris[f] + 3 * (t - T.tbase[f]);
  const long long vb = T.vbase[f];
  int a = nid[vb + s[0]], b = nid[vb + s[1]], c = nid[vb + s[2]];
  keep[t] = (a != b && b != c && a != c) ? 1 : 0;
  if (b < a && b <= c) { const int x = a; a = b; b = c; c = x; }        // rotate left by one
  else if (c < a && c < b) { const int x = c; c = b; b = a; a = x; }    // rotate left by two
  k01[t] = ((unsigned long long)(unsigned)a << 32) | (unsigned)b;
  k2[t] = (unsigned)c;
}
__global__ __launch_bounds__(256) void k_im_compact(long long M, const int* keep, const int* pos, const unsigned long long* k01, const unsigned* k2,
                                                    unsigned long long* o01, unsigned* o2, int* oidx) {
  const long long t = blockIdx.x * 256LL + threadIdx.x;
  if (t >= M || !keep[t]) return;
  const int q = pos[t];
  o01[q] = k01[t]; o2[q] = k2[t]; oidx[q] = q;
}
__global__ __launch_bounds__(256) void k_im_gather01(long long M, const int* perm, const unsigned long long* k01, unsigned long long* o) {
  const long long t = blockIdx.x * 256LL + threadIdx.x;
  if (t < M) o[t] = k01[perm[t]];
}
__global__ __launch_bounds__(256) void k_im_uniqflag(long long M, const int* perm, const unsigned long long* s01, const unsigned* k2, int* first) {
  const long long t = blockIdx.x * 256LL + threadIdx.x;
  if (t >= M) return;
  first[t] = (t == 0 || s01[t] != s01[t - 1] || k2[perm[t]] != k2[perm[t - 1]]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_im_emit(long long M, const int* perm, const unsigned long long* s01, const unsigned* k2, const int* first, const int* pos,
                                                 int32_t* out) {
  const long long t = blockIdx.x * 256LL + threadIdx.x;
  if (t >= M || !first[t]) return;
  int32_t* o = out + 3LL * pos[t];
  o[0] = (int32_t)(s01[t] >> 32); o[1] = (int32_t)(s01[t] & 0xffffffffull); o[2] = (int32_t)k2[perm[t]];
}

struct Arena {  // carve 256-byte aligned pieces out of one allocation
  unsigned char* p;
  size_t used = 0;
  template <class T> T* take(size_t n) { T* r = (T*)(p + used); used += (n * sizeof(T) + 255) / 256 * 256; return r; }
};
inline size_t al(size_t n, size_t sz) { return (n * sz + 255) / 256 * 256; }
}  // namespace

extern "C" int pa_iso_merge(pa_ctx* ctx, int nfrag, const pa_iso_frag* frags, int ncomp, int64_t* nnodes, double** dev_nodes, int64_t* nelts, int32_t** dev_elts) {
  PaBind bind_(ctx);
  if (!ctx || (nfrag > 0 && !frags) || !nnodes || !dev_nodes || !nelts || !dev_elts) return pa_fail(ctx, "pa_iso_merge: null argument");
  if (ncomp < 3 || nfrag < 0) return pa_fail(ctx, "pa_iso_merge: needs the three coordinates in front of the node data");
  *nnodes = *nelts = 0; *dev_nodes = nullptr; *dev_elts = nullptr;
  std::vector<const double*> hv;
  std::vector<const int32_t*> ht;
  std::vector<long long> vb{0}, tb{0};
  for (int f = 0; f < nfrag; ++f) {
    if (frags[f].nvert < 0 || frags[f].ntri < 0 || (frags[f].nvert > 0 && !frags[f].verts) || (frags[f].ntri > 0 && !frags[f].tris)) return pa_fail(ctx, "pa_iso_merge: bad fragment");
    if (frags[f].nvert == 0 && frags[f].ntri == 0) continue;
    hv.push_back(frags[f].verts); ht.push_back(frags[f].tris);
    vb.push_back(vb.back() + frags[f].nvert); tb.push_back(tb.back() + frags[f].ntri);
  }
  const int nf = (int)hv.size();
  const long long N = vb.back(), M = tb.back();
  if (N == 0) return 0;
  if (N >= 0x7fffffffLL || M >= 0x7fffffffLL) return pa_fail(ctx, "pa_iso_merge: surface too large for 32-bit ids");
  // the largest sum of squares whose host square root is still below the tolerance
  static const double smax = [] {
    double s = MEPS * MEPS;
    while (std::sqrt(s) < MEPS) s = std::nextafter(s, 1.0);
    while (!(std::sqrt(s) < MEPS)) s = std::nextafter(s, 0.0);
    return s;
  }();
  hipStream_t st = ctx->stream;
  // rocPRIM temporary storage
  size_t t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, t1, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)N, 0, 64, st);
  (void)rocprim::exclusive_scan(nullptr, t2, (int*)nullptr, (int*)nullptr, 0, (size_t)std::max(N, M), rocprim::plus<int>(), st);
  if (M > 0) {
    (void)rocprim::radix_sort_pairs(nullptr, t3, (unsigned*)nullptr, (unsigned*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)M, 0, 32, st);
    (void)rocprim::radix_sort_pairs(nullptr, t4, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)M, 0, 64, st);
  }
  t5 = std::max(std::max(t1, t2), std::max(t3, t4));
  const size_t NN = (size_t)N, MM = (size_t)std::max<long long>(M, 1);
  const size_t bytes = al(nf, 8) * 2 + al(nf + 1, 8) * 2 + al(3 * NN, 8) + al(NN, 8) * 2 + al(NN, 4) * 6 + al(MM, 8) * 3 + al(MM, 4) * 7 + al(t5, 1) + 1024;
  unsigned char* work = nullptr;
  PA_HIP(hipMalloc(&work, bytes));
  auto fail = [&](const char* m) { (void)hipFree(work); if (*dev_nodes) { (void)hipFree(*dev_nodes); *dev_nodes = nullptr; } if (*dev_elts) { (void)hipFree(*dev_elts); *dev_elts = nullptr; } return pa_fail(ctx, m); };
#define IM_HIP(x) do { if ((x) != hipSuccess) return fail("pa_iso_merge: " #x " failed"); } while (0)
  Arena A{work};
  const double** d_v = A.take<const double*>(nf);
  const int32_t** d_t = A.take<const int32_t*>(nf);
  long long* d_vb = A.take<long long>(nf + 1);
  long long* d_tb = A.take<long long>(nf + 1);
  double* P = A.take<double>(3 * NN);
  unsigned long long *hash = A.take<unsigned long long>(NN), *shash = A.take<unsigned long long>(NN);
  int *idx = A.take<int>(NN), *sidx = A.take<int>(NN), *dup = A.take<int>(NN), *isnew = A.take<int>(NN), *newid = A.take<int>(NN), *nid = A.take<int>(NN);
  unsigned long long *k01 = A.take<unsigned long long>(MM), *c01 = A.take<unsigned long long>(MM), *s01 = A.take<unsigned long long>(MM);
  unsigned *k2 = A.take<unsigned>(MM), *c2 = A.take<unsigned>(MM), *s2 = A.take<unsigned>(MM);
  int *keep = A.take<int>(MM), *pos = A.take<int>(MM), *cidx = A.take<int>(MM), *perm1 = A.take<int>(MM);
  void* tmp = A.take<unsigned char>(t5);
  int* d_flag = A.take<int>(64);  // [0] ambiguous, [1..2] scan tails
  IM_HIP(hipMemcpyAsync(d_v, hv.data(), 8 * (size_t)nf, hipMemcpyHostToDevice, st));
  IM_HIP(hipMemcpyAsync(d_t, ht.data(), 8 * (size_t)nf, hipMemcpyHostToDevice, st));
  IM_HIP(hipMemcpyAsync(d_vb, vb.data(), 8 * (size_t)(nf + 1), hipMemcpyHostToDevice, st));
  IM_HIP(hipMemcpyAsync(d_tb, tb.data(), 8 * (size_t)(nf + 1), hipMemcpyHostToDevice, st));
  IM_HIP(hipMemsetAsync(d_flag, 0, 256, st));
  const FragTab T{d_v, d_t, d_vb, d_tb, nf, ncomp};
  const dim3 gn((unsigned)((N + 255) / 256)), gm((unsigned)((std::max<long long>(M, 1) + 255) / 256)), blk(256);
  hipLaunchKernelGGL(k_im_keys, gn, blk, 0, st, T, N, P, hash, idx);
  size_t tb1 = t5;
  IM_HIP(rocprim::radix_sort_pairs(tmp, tb1, hash, shash, idx, sidx, (size_t)N, 0, 64, st));  // stable: equal hashes stay in raw order
  hipLaunchKernelGGL(k_im_probe, gn, blk, 0, st, N, P, shash, sidx, smax, dup);
  hipLaunchKernelGGL(k_im_flags, gn, blk, 0, st, N, dup, isnew, d_flag);
  tb1 = t5;
  IM_HIP(rocprim::exclusive_scan(tmp, tb1, isnew, newid, 0, (size_t)N, rocprim::plus<int>(), st));
  int h_flag = 0, h_lastid = 0, h_lastnew = 0;
  IM_HIP(hipMemcpyAsync(&h_flag, d_flag, 4, hipMemcpyDeviceToHost, st));
  IM_HIP(hipMemcpyAsync(&h_lastid, newid + (N - 1), 4, hipMemcpyDeviceToHost, st));
  IM_HIP(hipMemcpyAsync(&h_lastnew, isnew + (N - 1), 4, hipMemcpyDeviceToHost, st));
  IM_HIP(hipStreamSynchronize(st));
  if (h_flag) {  // a cluster whose closeness is not transitive: the sequential rule decides, on the host
    (void)hipFree(work);
    ctx->err = "pa_iso_merge: nodes within the tolerance of each other do not form transitive clusters; use the sequential merge";
    return 2;
  }
  const long long nn = (long long)h_lastid + h_lastnew;
  double* nodes = nullptr;
  IM_HIP(hipMalloc(&nodes, (size_t)nn * ncomp * 8));
  *dev_nodes = nodes;
  hipLaunchKernelGGL(k_im_nodes, gn, blk, 0, st, T, N, dup, newid, nid, nodes);
  long long ne = 0;
  if (M > 0) {
    hipLaunchKernelGGL(k_im_elts, gm, blk, 0, st, T, M, nid, k01, k2, keep);
    tb1 = t5;
    IM_HIP(rocprim::exclusive_scan(tmp, tb1, keep, pos, 0, (size_t)M, rocprim::plus<int>(), st));
    int lp = 0, lk = 0;
    IM_HIP(hipMemcpyAsync(&lp, pos + (M - 1), 4, hipMemcpyDeviceToHost, st));
    IM_HIP(hipMemcpyAsync(&lk, keep + (M - 1), 4, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(k_im_compact, gm, blk, 0, st, M, keep, pos, k01, k2, c01, c2, cidx);
    IM_HIP(hipStreamSynchronize(st));
    const long long M2 = (long long)lp + lk;
    if (M2 > 0) {
      const dim3 g2((unsigned)((M2 + 255) / 256));
      // lexicographic order of (a, b, c): stable sort by c, then stable sort by (a, b)
      tb1 = t5;
      IM_HIP(rocprim::radix_sort_pairs(tmp, tb1, c2, s2, cidx, perm1, (size_t)M2, 0, 32, st));
      hipLaunchKernelGGL(k_im_gather01, g2, blk, 0, st, M2, perm1, c01, k01);  // k01 reused: (a, b) in c-order
      tb1 = t5;
      int* perm2 = cidx;  // reused
      IM_HIP(rocprim::radix_sort_pairs(tmp, tb1, k01, s01, perm1, perm2, (size_t)M2, 0, 64, st));
      hipLaunchKernelGGL(k_im_uniqflag, g2, blk, 0, st, M2, perm2, s01, c2, keep);
      tb1 = t5;
      IM_HIP(rocprim::exclusive_scan(tmp, tb1, keep, pos, 0, (size_t)M2, rocprim::plus<int>(), st));
      IM_HIP(hipMemcpyAsync(&lp, pos + (M2 - 1), 4, hipMemcpyDeviceToHost, st));
      IM_HIP(hipMemcpyAsync(&lk, keep + (M2 - 1), 4, hipMemcpyDeviceToHost, st));
      IM_HIP(hipStreamSynchronize(st));
      ne = (long long)lp + lk;
      int32_t* elts = nullptr;
      IM_HIP(hipMalloc(&elts, (size_t)ne * 12));
      *dev_elts = elts;
      hipLaunchKernelGGL(k_im_emit, g2, blk, 0, st, M2, perm2, s01, c2, keep, pos, elts);
    }
  }
  IM_HIP(hipGetLastError());
  IM_HIP(hipStreamSynchronize(st));
  (void)hipFree(work);
  *nnodes = nn;
  *nelts = ne;
  return 0;
#undef IM_HIP
}
