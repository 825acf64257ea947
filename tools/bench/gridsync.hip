// gridsync.hip -- cost of a grid-wide barrier inside one launch on gfx950 (for the distance function's hyperplane sweeps:
// 6192 dependent steps per grid, today one launch each).  (a) cooperative groups grid.sync(); (b) a counter barrier with
// agent-scope release / acquire and a bounded spin; both with a small read-modify-write of shared data between barriers
// so that visibility is checked, not assumed.   build: hipcc --offload-arch=gfx950 -O3 gridsync.hip -o gridsync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every step: element i becomes the sum of its value and its neighbour's of the previous step (needs the barrier to be right)
__global__ void k_coop(int* a, int* b, int n, int steps) {
  cg::grid_group g = cg::this_grid();
  const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int s = 0; s < steps; ++s) {
    int* src = (s & 1) ? b : a;
    int* dst = (s & 1) ? a : b;
    for (int i = t; i < n; i += nt) dst[i] = src[i] + src[(i + 4099) % n];
    g.sync();
  }
}
__global__ void k_spin(int* a, int* b, int n, int steps, unsigned* cnt, int* err) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int s = 0; s < steps; ++s) {
    int* src = (s & 1) ? b : a;
    int* dst = (s & 1) ? a : b;
    for (int i = t; i < n; i += nt) {
      const int x = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), y = __hip_atomic_load(&src[(i + 4099) % n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&dst[i], x + y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(s + 1) * gridDim.x;
      unsigned spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > (1u << 22)) { *err = 1; break; }  // bounded: every wave leaves
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
}

int main() {
  const int n = 16384, steps = 2000;
  int *a, *b, *err; unsigned* cnt;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&err, 4));
  std::vector<int> h(n), ref(n), tmp(n);
  for (int nwg : {32, 64, 128, 256}) {
    for (int mode = 0; mode < 2; ++mode) {
      for (int i = 0; i < n; ++i) h[i] = i % 7;
      CK(hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice));
      CK(hipMemset(cnt, 0, 4)); CK(hipMemset(err, 0, 4));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, 0));
      int nn = n, st = steps;
      if (mode == 0) {
        void* args[] = {&a, &b, &nn, &st};
        CK(hipLaunchCooperativeKernel((const void*)k_coop, dim3(nwg), dim3(256), args, 0, 0));
      } else {
        hipLaunchKernelGGL(k_spin, dim3(nwg), dim3(256), 0, 0, a, b, nn, st, cnt, err);
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      // reference on the host (int arithmetic wraps the same way)
      ref = h;
      for (int s = 0; s < steps; ++s) { for (int i = 0; i < n; ++i) tmp[i] = (int)((unsigned)ref[i] + (unsigned)ref[(i + 4099) % n]); ref.swap(tmp); }
      CK(hipMemcpy(tmp.data(), (steps & 1) ? b : a, n * 4, hipMemcpyDeviceToHost));
      int bad = 0, herr = 0;
      for (int i = 0; i < n; ++i) bad += tmp[i] != ref[i];
      CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      printf("%s  %3d workgroups x 256: %.2f us per step (%d steps), wrong values %d, spin timeout %d\n", mode ? "counter barrier" : "grid.sync()    ", nwg, ms * 1e3 / steps, steps, bad, herr);
    }
  }
  return 0;
}
