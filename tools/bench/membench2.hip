// membench2.hip -- how much does the tile/marching access pattern cost against HBM?
// Emulates the fused kernel's stores (8 output arrays, FAB layout [box][comp][k][j][i], 128^3
// boxes) and its two input loads with no arithmetic.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// CPL cells per lane (1: 64-wide tile, 8B; 2: 128-wide tile, 16B), TY rows per WG (one wave per row)
template <int CPL, int TY, int NIN, int SYNC = 0, int LDSB = 8>
__global__ void k_march(const double* __restrict__ in, double* __restrict__ out, int nb, int kseg, int order, long long pad) {
  __shared__ double s_dummy[LDSB / 8];
  if (threadIdx.x == 0) s_dummy[0] = 0;
  constexpr int N = 128;
  const int tx = N / (64 * CPL), ty = N / TY, tz = N / kseg;
  int bid = blockIdx.x;
  const int per_box = tx * ty * tz;
  int b, t;
  if (order == 0) { b = bid / per_box; t = bid % per_box; }       // box-major: consecutive WGs = same box
  else { b = bid % nb; t = bid / nb; }                             // tile-major: consecutive WGs = different boxes
  const int bx = t % tx, by = (t / tx) % ty, bz = t / (tx * ty);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long cell0 = (((long long)(bz * kseg) * N + (by * TY + w)) * N) + bx * 64 * CPL + lane * CPL;
  const long long boxsz = (long long)N * N * N + pad;
  const double* pi = in + (long long)b * NIN * boxsz + cell0;
  double* po = out + (long long)b * 8 * boxsz + cell0;
  for (int k = 0; k < kseg; ++k) {
    double v[CPL];
    for (int q = 0; q < CPL; ++q) v[q] = 0;
    for (int s = 0; s < NIN; ++s) {
      if (CPL == 2) { double2 a = *(const double2*)(pi + s * boxsz); v[0] += a.x; v[1] += a.y; }
      else v[0] += pi[s * boxsz];
    }
    for (int s = 0; s < 8; ++s) {
      if (CPL == 2) *(double2*)(po + s * boxsz) = make_double2(v[0] + s, v[1] + s);
      else po[s * boxsz] = v[0] + s;
    }
    pi += N * N; po += N * N;
    if (SYNC) __syncthreads();
  }
  if (threadIdx.x == 1 && s_dummy[0] == 1.0) out[0] = 0;
}

template <int CPL, int TZ, int NIN>
__global__ void k_march_y(const double* __restrict__ in, double* __restrict__ out, int nb, int jseg, int order, long long pad) {
  constexpr int N = 128;
  const int tx = N / (64 * CPL), tz = N / TZ, ty = N / jseg;
  int bid = blockIdx.x;
  const int per_box = tx * ty * tz;
  int b, t;
  if (order == 0) { b = bid / per_box; t = bid % per_box; }
  else { b = bid % nb; t = bid / nb; }
  const int bx = t % tx, bz = (t / tx) % tz, by = t / (tx * tz);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long cell0 = (((long long)(bz * TZ + w) * N + by * jseg) * N) + bx * 64 * CPL + lane * CPL;
  const long long boxsz = (long long)N * N * N + pad;
  const double* pi = in + (long long)b * NIN * boxsz + cell0;
  double* po = out + (long long)b * 8 * boxsz + cell0;
  for (int j = 0; j < jseg; ++j) {
    double v[CPL];
    for (int q = 0; q < CPL; ++q) v[q] = 0;
    for (int s = 0; s < NIN; ++s) {
      if (CPL == 2) { double2 a = *(const double2*)(pi + s * boxsz); v[0] += a.x; v[1] += a.y; }
      else v[0] += pi[s * boxsz];
    }
    for (int s = 0; s < 8; ++s) {
      if (CPL == 2) *(double2*)(po + s * boxsz) = make_double2(v[0] + s, v[1] + s);
      else po[s * boxsz] = v[0] + s;
    }
    pi += N; po += N;
  }
}

template <int CPL, int TY, int NIN, bool MY = false, int SYNC = 0, int LDSB = 8>
int run(const char* name, int nb, int kseg, int order, long long pad = 0) {
  const long long boxsz = 128LL * 128 * 128 + pad;
  double *in, *out;
  CK(hipMalloc(&in, 8 * boxsz * nb * NIN));
  CK(hipMalloc(&out, 8 * boxsz * nb * 8));
  CK(hipMemset(in, 0, 8 * boxsz * nb * NIN));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = nb * (128 / (64 * CPL)) * (128 / TY) * (128 / kseg);
  float best = 1e9;
  for (int it = 0; it < 8; ++it) {
    CK(hipEventRecord(a));
    if (MY) hipLaunchKernelGGL((k_march_y<CPL, TY, NIN>), dim3(grid), dim3(64 * TY), 0, 0, in, out, nb, kseg, order, pad);
    else hipLaunchKernelGGL((k_march<CPL, TY, NIN, SYNC, LDSB>), dim3(grid), dim3(64 * TY), 0, 0, in, out, nb, kseg, order, pad);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (it > 0 && ms < best) best = ms;
  }
  const double bytes = (double)boxsz * nb * 8 * (NIN + 8);
  printf("pad %5lld %-22s cpl %d ty %2d nin %d kseg %3d order %d grid %6d: %.3f ms  %.0f GB/s\n", pad, name, CPL, TY, NIN, kseg, order, grid, best, bytes / best / 1e6);
  CK(hipFree(in)); CK(hipFree(out));
  return 0;
}

int main() {
  const int nb = 64;
  for (int rep = 0; rep < 2; ++rep) {
    run<1, 13, 1>("z 64x13 1in free", nb, 128, 0, 64);
    run<1, 13, 1, false, 1, 8>("z 64x13 1in sync", nb, 128, 0, 64);
    run<1, 13, 1, false, 0, 90000>("z 64x13 1in 1wg/cu", nb, 128, 0, 64);
    run<1, 13, 1, false, 1, 90000>("z 64x13 1in sync 1wg/cu", nb, 128, 0, 64);
    run<1, 16, 1, false, 1, 90000>("z 64x16 1in sync 1wg/cu", nb, 128, 0, 64);
  }
  return 0;
}
