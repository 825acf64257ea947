// membench.hip -- calibration of the achievable HBM rate for the access shapes of the fused
// grad->curvature kernel (N-in / M-out streams, 8 or 16 bytes per lane).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename T, int NIN, int NOUT>
__global__ __launch_bounds__(256) void k_streams(const T* __restrict__ in, T* __restrict__ out, long long n, long long stride_in, long long stride_out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    T a = in[i];
#pragma unroll
    for (int s = 1; s < NIN; ++s) { T b = in[i + s * stride_in]; a.x += b.x; }
#pragma unroll
    for (int s = 0; s < NOUT; ++s) { T o = a; o.x += s; out[i + s * stride_out] = o; }
  }
}
struct D1 { double x; };
struct D2 { double x, y; };

template <typename T, int NIN, int NOUT>
__global__ __launch_bounds__(256) void k_streams_nt(const T* __restrict__ in, T* __restrict__ out, long long n, long long stride_in, long long stride_out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    T a;
    a.x = __builtin_nontemporal_load(&in[i].x);
#pragma unroll
    for (int s = 1; s < NIN; ++s) { a.x += __builtin_nontemporal_load(&in[i + s * stride_in].x); }
#pragma unroll
    for (int s = 0; s < NOUT; ++s) {
      double* o = (double*)&out[i + s * stride_out];
      for (int q = 0; q < (int)(sizeof(T) / 8); ++q) __builtin_nontemporal_store(a.x + s + q, o + q);
    }
  }
}

template <typename T, int NIN, int NOUT, bool NT = false>
int run(const char* name, long long cells, int grid) {
  const long long n = cells * sizeof(double) / sizeof(T);
  T *in, *out;
  CK(hipMalloc(&in, sizeof(T) * n * NIN));
  CK(hipMalloc(&out, sizeof(T) * n * NOUT));
  CK(hipMemset(in, 0, sizeof(T) * n * NIN));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int it = 0; it < 12; ++it) {
    CK(hipEventRecord(a));
    if (NT) hipLaunchKernelGGL((k_streams_nt<T, NIN, NOUT>), dim3(grid), dim3(256), 0, 0, in, out, n, n, n);
    else hipLaunchKernelGGL((k_streams<T, NIN, NOUT>), dim3(grid), dim3(256), 0, 0, in, out, n, n, n);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (it > 0 && ms < best) best = ms;
  }
  const double bytes = (double)cells * 8 * (NIN + NOUT);
  printf("%-28s grid %6d: %.3f ms  %.0f GB/s\n", name, grid, best, bytes / best / 1e6);
  CK(hipFree(in)); CK(hipFree(out));
  return 0;
}

int main() {
  const long long cells = 512LL * 512 * 512;
  for (int grid : {4096, 32768, 262144}) {
    run<D1, 2, 8>("8B/lane 2in 8out", cells, grid);
    run<D2, 2, 8>("16B/lane 2in 8out", cells, grid);
    run<D1, 2, 8, true>("8B/lane 2in 8out nt", cells, grid);
    run<D2, 2, 8, true>("16B/lane 2in 8out nt", cells, grid);
    run<D1, 1, 8>("8B/lane 1in 8out", cells, grid);
    run<D2, 1, 8>("16B/lane 1in 8out", cells, grid);
    run<D1, 1, 8, true>("8B/lane 1in 8out nt", cells, grid);
    run<D2, 1, 8, true>("16B/lane 1in 8out nt", cells, grid);
  }
  return 0;
}
