// membench3.hip -- HBM ceiling of the fused sweep's access pattern on MI355X, with exact coverage
// (TY divides 128; membench2's 13-row case covered 117 of 128 rows and over-reported by 9 %).
// 64 boxes of 128^3, 1 input array (no ghosts), 8 output arrays, FAB layout [box][comp][k][j][i]
// with the padded component stride.  No arithmetic.  Not part of the product.
//   CPL  cells per lane (1: 64-wide tile / 8-B accesses, 2: 128-wide tile = whole rows / 16-B)
//   TY   rows per workgroup (one wavefront per row), marching kseg planes with a barrier per plane
//   NT   1: non-temporal stores, 2: non-temporal loads too
//   LDSB LDS bytes per workgroup (occupancy control)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int CPL, int TY, int NT, int LDSB>
__global__ __launch_bounds__(64 * TY) void k_march(const double* __restrict__ in, double* __restrict__ out, int nb, int kseg, int order, long long boxsz) {
  __shared__ double s_dummy[LDSB / 8];
  if (threadIdx.x == 0) s_dummy[0] = 0;
  constexpr int N = 128;
  const int tx = N / (64 * CPL), ty = N / TY, tz = N / kseg;
  int bid = blockIdx.x;
  const int per_box = tx * ty * tz;
  int b, t;
  if (order == 0) { b = bid / per_box; t = bid % per_box; }  // box-major: consecutive WGs = same box
  else { b = bid % nb; t = bid / nb; }                        // tile-major: consecutive WGs = different boxes
  const int bx = t % tx, by = (t / tx) % ty, bz = t / (tx * ty);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long cell0 = (((long long)(bz * kseg) * N + (by * TY + w)) * N) + bx * 64 * CPL + lane * CPL;
  const double* pi = in + (long long)b * boxsz + cell0;
  double* po = out + (long long)b * 8 * boxsz + cell0;
  typedef double d2 __attribute__((ext_vector_type(2)));
  for (int k = 0; k < kseg; ++k) {
    if (CPL == 2) {
      d2 a = (NT >= 2) ? __builtin_nontemporal_load((const d2*)pi) : *(const d2*)pi;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        d2 v = a + (double)s;
        if (NT >= 1) __builtin_nontemporal_store(v, (d2*)(po + s * boxsz));
        else *(d2*)(po + s * boxsz) = v;
      }
    } else {
      double a = (NT >= 2) ? __builtin_nontemporal_load(pi) : *pi;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (NT >= 1) __builtin_nontemporal_store(a + s, po + s * boxsz);
        else po[s * boxsz] = a + s;
      }
    }
    pi += N * N; po += N * N;
    __syncthreads();
  }
  if (threadIdx.x == 1 && s_dummy[0] == 1.0) out[0] = 0;
}

// 1-D streaming reference: block handles 256 x CPL consecutive cells of every array
template <int CPL, int NT>
__global__ __launch_bounds__(256) void k_stream(const double* __restrict__ in, double* __restrict__ out, long long boxsz, int nb) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const long long per_box = 128LL * 128 * 128 / (256 * CPL);
  const long long bid = blockIdx.x;
  const int b = (int)(bid / per_box);
  const long long c0 = (bid % per_box) * 256 * CPL + threadIdx.x * CPL;
  const double* pi = in + (long long)b * boxsz + c0;
  double* po = out + (long long)b * 8 * boxsz + c0;
  if (CPL == 2) {
    d2 a = (NT >= 2) ? __builtin_nontemporal_load((const d2*)pi) : *(const d2*)pi;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      d2 v = a + (double)s;
      if (NT >= 1) __builtin_nontemporal_store(v, (d2*)(po + s * boxsz));
      else *(d2*)(po + s * boxsz) = v;
    }
  } else {
    double a = (NT >= 2) ? __builtin_nontemporal_load(pi) : *pi;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (NT >= 1) __builtin_nontemporal_store(a + s, po + s * boxsz);
      else po[s * boxsz] = a + s;
    }
  }
}

static double *g_in, *g_out;
static const int nb = 64;
static long long boxsz = 128LL * 128 * 128 + 64;  // MB_PAD=<doubles> overrides the + 64

template <typename F>
int timeit(const char* name, F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9, sum = 0;
  const int NIT = 8;
  for (int it = 0; it < NIT; ++it) {
    CK(hipEventRecord(a));
    launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (it > 0) { sum += ms; if (ms < best) best = ms; }
  }
  CK(hipGetLastError());
  const double bytes = 128.0 * 128 * 128 * nb * 8 * 9;
  printf("%-44s best %.3f ms (%.0f GB/s)  mean %.3f ms\n", name, best, bytes / best / 1e6, sum / (NIT - 1));
  fflush(stdout);
  return 0;
}

template <int CPL, int TY, int NT, int LDSB>
int run(int kseg, int order) {
  char name[128];
  snprintf(name, sizeof name, "march cpl %d ty %2d nt %d lds %6d kseg %3d ord %d", CPL, TY, NT, LDSB, kseg, order);
  const int grid = nb * (128 / (64 * CPL)) * (128 / TY) * (128 / kseg);
  return timeit(name, [&] { hipLaunchKernelGGL((k_march<CPL, TY, NT, LDSB>), dim3(grid), dim3(64 * TY), 0, 0, g_in, g_out, nb, kseg, order, boxsz); });
}
template <int CPL, int NT>
int runs() {
  char name[128];
  snprintf(name, sizeof name, "stream 1-D cpl %d nt %d", CPL, NT);
  const long long grid = (long long)nb * 128 * 128 * 128 / (256 * CPL);
  return timeit(name, [&] { hipLaunchKernelGGL((k_stream<CPL, NT>), dim3((unsigned)grid), dim3(256), 0, 0, g_in, g_out, boxsz, nb); });
}

int main() {
  if (getenv("MB_PAD")) { boxsz = 128LL * 128 * 128 + atoll(getenv("MB_PAD")); printf("component stride pad %lld doubles\n", boxsz - 128LL * 128 * 128); }
  CK(hipMalloc(&g_in, 8 * boxsz * nb));
  CK(hipMalloc(&g_out, 8 * boxsz * nb * 8));
  CK(hipMemset(g_in, 0, 8 * boxsz * nb));
  CK(hipMemset(g_out, 0, 8 * boxsz * nb * 8));
  if (getenv("MB_QUICK")) { runs<1, 0>(); return 0; }
  for (int rep = 0; rep < 2; ++rep) {
    runs<1, 0>(); runs<2, 0>(); runs<2, 1>(); runs<2, 2>();
    run<1, 16, 0, 90000>(128, 0);
    run<1, 16, 1, 90000>(128, 0);
    run<1, 16, 2, 90000>(128, 0);
    run<1, 8, 0, 8>(128, 0);
    run<1, 8, 0, 70000>(128, 0);
    run<2, 16, 0, 90000>(128, 0);
    run<2, 16, 1, 90000>(128, 0);
    run<2, 16, 2, 90000>(128, 0);
    run<2, 8, 0, 70000>(128, 0);
    run<2, 8, 2, 70000>(128, 0);
    run<2, 8, 0, 8>(128, 0);
    run<2, 8, 2, 8>(128, 0);
    run<2, 8, 0, 70000>(64, 0);
    run<2, 8, 0, 70000>(32, 0);
    run<2, 16, 0, 90000>(128, 1);
    run<2, 4, 0, 8>(128, 0);
    run<2, 4, 0, 40000>(128, 0);
  }
  return 0;
}
