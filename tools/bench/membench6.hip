// membench6.hip -- is the tile-marching form of the sweep limited by how many stores a wave may have in flight?
// membench3's marching kernel (64-wide tile, one wavefront per row, 8-B accesses, barrier per plane, 1 workgroup per CU)
// with the input requested D planes ahead: vmcnt retires in issue order, so consuming the load of plane k waits for every
// store issued before it -- with depth D a wave may have D bursts of 8 stores outstanding.  FAB layout [box][comp][cells],
// component stride 16 MiB + MB_PAD doubles (default 256 = the library's 2 KiB).  No arithmetic.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int D, int TY, int LDSB, int NOBAR>
__global__ __launch_bounds__(64 * TY) void k_march(const double* __restrict__ in, double* __restrict__ out, int nb, long long boxsz) {
  __shared__ double s_dummy[LDSB / 8];
  if (threadIdx.x == 0) s_dummy[0] = 0;
  constexpr int N = 128;
  const int ty = N / TY, per_box = 2 * ty;
  const int b = blockIdx.x / per_box, t = blockIdx.x % per_box;
  const int bx = t % 2, by = t / 2;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long cell0 = ((long long)(by * TY + w) * N) + bx * 64 + lane;
  const double* pi = in + (long long)b * boxsz + cell0;
  double* po = out + (long long)b * 8 * boxsz + cell0;
  double a[D];
#pragma unroll
  for (int d = 0; d < D; ++d) a[d] = pi[(long long)d * N * N];
  auto group = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int k = k0 + d;
      double x;
      asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(a[d]));  // the wait lands here; a[d]'s register takes the next request (no copies at the back edge)
      a[d] = pi[(long long)(k + D < N ? k + D : N - 1) * N * N];  // no branch: the wait counts stay static
#pragma unroll
      for (int s = 0; s < 8; ++s) po[s * boxsz + (long long)k * N * N] = x + s;
      if (!NOBAR) __syncthreads();
    }
  };
  group(0);  // peeled: the loop is entered with the steady state's operations in flight, so its static vmcnt(N) are the deep ones
#pragma unroll 1
  for (int k0 = D; k0 < N; k0 += D) group(k0);
  if (threadIdx.x == 1 && s_dummy[0] == 1.0) out[0] = 0;
}

static double *g_in, *g_out;
static const int nb = 64;
static long long boxsz = 128LL * 128 * 128 + 256;

template <int D, int TY, int LDSB, int NOBAR>
int run() {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9, sum = 0;
  const int NIT = 8, grid = nb * 2 * (128 / TY);
  for (int it = 0; it < NIT; ++it) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_march<D, TY, LDSB, NOBAR>), dim3(grid), dim3(64 * TY), 0, 0, g_in, g_out, nb, boxsz);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (it > 0) { sum += ms; if (ms < best) best = ms; }
  }
  CK(hipGetLastError());
  const double bytes = 128.0 * 128 * 128 * nb * 8 * 9;
  printf("depth %d ty %2d lds %6d nobarrier %d: best %.3f ms (%.0f GB/s)  mean %.3f ms\n", D, TY, LDSB, NOBAR, best, bytes / best / 1e6, sum / (NIT - 1));
  fflush(stdout);
  return 0;
}

int main() {
  if (getenv("MB_PAD")) boxsz = 128LL * 128 * 128 + atoll(getenv("MB_PAD"));
  printf("component stride pad %lld doubles\n", boxsz - 128LL * 128 * 128);
  CK(hipMalloc(&g_in, 8 * boxsz * nb));
  CK(hipMalloc(&g_out, 8 * boxsz * nb * 8));
  CK(hipMemset(g_in, 0, 8 * boxsz * nb));
  CK(hipMemset(g_out, 0, 8 * boxsz * nb * 8));
  if (getenv("MB_QUICK")) { run<1, 16, 90000, 0>(); return 0; }
  for (int rep = 0; rep < 2; ++rep) {
    run<1, 16, 90000, 0>(); run<2, 16, 90000, 0>(); run<4, 16, 90000, 0>(); run<8, 16, 90000, 0>(); run<16, 16, 90000, 0>();
    run<1, 16, 90000, 1>(); run<4, 16, 90000, 1>(); run<8, 16, 90000, 1>();
    run<1, 8, 70000, 0>(); run<4, 8, 70000, 0>(); run<8, 8, 70000, 0>();     // 2 workgroups of 8 waves per CU
    run<4, 8, 8, 0>(); run<8, 8, 8, 0>();                                    // 4 workgroups of 8 waves per CU
    run<4, 4, 8, 0>(); run<8, 4, 8, 0>();                                    // 8 workgroups of 4 waves per CU
  }
  return 0;
}
