// membench4.hip -- which feature of the fused sweep's access pattern costs HBM throughput?
// membench3's marching kernel (64-wide tile, one wave per row, barrier per plane, 1 WG per CU)
// plus, one at a time: the ghosted phi layout (132^3, unaligned rows), two halo row waves that only
// load, an edge wave that gathers the two columns beside the tile, 13-row tiles (10 per box, the
// last partial).  No arithmetic.  Reported GB/s = algorithmic 72 B/cell.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// FEAT bits: 1 ghosted phi layout, 2 halo row waves, 4 edge wave, 8 warm-up: 3 extra planes of stores
template <int TY, int FEAT, int NFMA = 0, int NLDS = 0, int NBURN = 0, int NWORK = 0>
__global__ __launch_bounds__(1024) void k_march(const double* __restrict__ in, double* __restrict__ out, int nb, long long boxsz, long long inbox) {
  __shared__ double s_dummy[90000 / 8];
  constexpr int N = 128, G = (FEAT & 1) ? 2 : 0, NP = N + 2 * G;
  constexpr int ty = (N + TY - 1) / TY;
  const int bid = blockIdx.x, per_box = 2 * ty;
  const int b = bid / per_box, t = bid % per_box;
  const int bx = t % 2, by = t / 2;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rows = min(TY, N - by * TY);
  const double* pin = in + (long long)b * inbox;
  double* pout = out + (long long)b * 8 * boxsz;
  double acc = 0;
  if (w < TY) {
    if (w >= rows) { for (int k = 0; k < N; ++k) __syncthreads(); return; }
    const int j = by * TY + w, i = bx * 64 + lane;
    const double* pi = pin + ((long long)(G) * NP + (j + G)) * NP + i + G;
    double* po = pout + ((long long)j) * N + i;
    if (NWORK > 0) {  // prefetched variant: [take plane k] [request plane k+1] [8 stores] [ALU work] [barrier]
      double nxt = *pi, w0 = 1.0, w1 = 2.0, w2 = 3.0, w3 = 4.0;
      for (int k = 0; k < N; ++k) {
        const double a = nxt;
        if (k < N - 1) pi += NP * NP;
        nxt = *pi;
#pragma unroll
        for (int s = 0; s < 8; ++s) po[s * boxsz] = a + s;
        po += N * N;
#pragma unroll
        for (int q = 0; q < NWORK / 4; ++q) {
          w0 = __builtin_fma(w0, 1.0000001, 0.5); w1 = __builtin_fma(w1, 0.9999999, 0.25);
          w2 = __builtin_fma(w2, 1.0000002, 0.125); w3 = __builtin_fma(w3, 0.9999998, 0.0625);
        }
        __syncthreads();
      }
      if ((w0 + w1) + (w2 + w3) == 1.2345e-300) s_dummy[lane] = w0;
      return;
    }
    for (int k = 0; k < N + ((FEAT & 8) ? 3 : 0); ++k) {
      double a = *pi;
      if (NFMA > 0) {  // fp64 work that a memory-bound kernel should hide
        double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
#pragma unroll
        for (int q = 0; q < NFMA / 4; ++q) {
          x0 = __builtin_fma(x0, 1.0000001, 0.5); x1 = __builtin_fma(x1, 0.9999999, 0.25);
          x2 = __builtin_fma(x2, 1.0000002, 0.125); x3 = __builtin_fma(x3, 0.9999998, 0.0625);
        }
        a = (x0 + x1) + (x2 + x3);
      }
      if (NLDS > 0) {
#pragma unroll
        for (int q = 0; q < NLDS; ++q) s_dummy[(q * 1024 + threadIdx.x) % (90000 / 8)] = a + q;
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) po[s * boxsz] = a + s;
      if (k < N - 1) { pi += NP * NP; }
      if (!(FEAT & 8) || k >= 3) po += N * N;
      if (k < N) __syncthreads();
    }
    return;
  }
  if ((FEAT & 2) && w < TY + 2) {  // halo rows: load only
    int j = (w == TY) ? by * TY - 1 : by * TY + rows;
    j = min(max(j, -G), N - 1 + G);
    const double* pi = pin + ((long long)(G) * NP + (j + G)) * NP + bx * 64 + lane + G;
    for (int k = 0; k < N; ++k) {
      acc += *pi; pi += NP * NP;
      if (NBURN > 0) {
        double x0 = acc, x1 = acc + 1, x2 = acc + 2, x3 = acc + 3;
#pragma unroll 8
        for (int q = 0; q < NBURN / 4; ++q) {
          x0 = __builtin_fma(x0, 1.0000001, 0.5); x1 = __builtin_fma(x1, 0.9999999, 0.25);
          x2 = __builtin_fma(x2, 1.0000002, 0.125); x3 = __builtin_fma(x3, 0.9999998, 0.0625);
        }
        acc = (x0 + x1) + (x2 + x3);
      }
      __syncthreads();
    }
    if (acc == 1.2345e-300) s_dummy[lane] = acc;
    return;
  }
  if ((FEAT & 4) && w == TY + 2) {  // edge wave: gathers 2 columns x (rows+2)
    const int l = lane % (2 * (TY + 2)), r = min(l >> 1, rows + 1), side = l & 1;
    int j = by * TY + r - 1, i = side ? bx * 64 + 64 : bx * 64 - 1;
    j = min(max(j, -G), N - 1 + G); i = min(max(i, -G), N - 1 + G);
    const double* pi = pin + ((long long)(G) * NP + (j + G)) * NP + i + G;
    for (int k = 0; k < N; ++k) { acc += *pi; pi += NP * NP; __syncthreads(); }
    if (acc == 1.2345e-300) s_dummy[lane] = acc;
    return;
  }
  for (int k = 0; k < N; ++k) __syncthreads();
}

__global__ void k_fill(double* p, long long n, int mode) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + 12345;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    p[i] = mode ? 300.0 + 1700.0 * (double)(z >> 11) * (1.0 / 9007199254740992.0) : 0.0;
  }
}
static double *g_in, *g_out;
static const int nb = 64;
static long long boxsz = 128LL * 128 * 128 + 64, inbox = 132LL * 132 * 132 + 64;  // MB_PAD=<doubles> overrides the + 64 of boxsz

template <int TY, int FEAT, int NFMA = 0, int NLDS = 0, int NBURN = 0, int NWORK = 0>
int run() {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9, sum = 0;
  const int NIT = 8, grid = nb * 2 * ((128 + TY - 1) / TY);
  for (int it = 0; it < NIT; ++it) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_march<TY, FEAT, NFMA, NLDS, NBURN, NWORK>), dim3(grid), dim3(64 * (TY + 3)), 0, 0, g_in, g_out, nb, boxsz, inbox);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (it > 0) { sum += ms; if (ms < best) best = ms; }
  }
  CK(hipGetLastError());
  const double bytes = 128.0 * 128 * 128 * nb * 8 * 9;
  printf("work %4d burn %4d fma %3d ldsw %d ty %2d ghosted %d halo-rows %d edge-wave %d warmup %d grid %5d: best %.3f ms (%.0f GB/s)  mean %.3f ms\n", NWORK, NBURN, NFMA, NLDS, TY, FEAT & 1, (FEAT >> 1) & 1, (FEAT >> 2) & 1,
         (FEAT >> 3) & 1, grid, best, bytes / best / 1e6, sum / (NIT - 1));
  fflush(stdout);
  return 0;
}

int main(int argc, char**) {
  if (getenv("MB_PAD")) { boxsz = 128LL * 128 * 128 + atoll(getenv("MB_PAD")); printf("component stride pad %lld doubles\n", boxsz - 128LL * 128 * 128); }
  CK(hipMalloc(&g_in, 8 * inbox * nb));
  CK(hipMalloc(&g_out, 8 * boxsz * nb * 8));
  CK(hipMemset(g_in, 0, 8 * inbox * nb));
  CK(hipMemset(g_out, 0, 8 * boxsz * nb * 8));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, g_in, inbox * nb, 1);
  CK(hipDeviceSynchronize());
  if (argc > 1) { run<13, 7>(); return 0; }  // profiling mode: one configuration
  for (int rep = 0; rep < 2; ++rep) {
    run<13, 7>(); run<13, 7, 0, 0, 0, 4>(); run<13, 7, 0, 0, 0, 100>(); run<13, 7, 0, 0, 0, 200>(); run<13, 7, 0, 0, 0, 400>(); run<13, 7, 0, 0, 0, 600>();
  }
  return 0;
}
