// xcdsync.hip -- a barrier among the workgroups of ONE XCD inside one launch, with plain loads / stores between the barriers:
// the distance function's hyperplane sweeps are ~6000 dependent steps of a few microseconds each; a launch per step costs
// ~13 us, a chip-wide barrier needs agent-scope accesses (the eight L2s are not coherent with each other).  The CUs of one
// XCD share ONE L2: stores are written through to it, so a step's results are visible to the other CUs of the same XCD as
// soon as (a) the writer has waited for its stores (s_waitcnt vmcnt(0)), and (b) the reader has dropped its vector cache
// (buffer_inv sc0).  Groups = the workgroups that report the same XCC_ID; each group has its own counter.
// Checks visibility (every step reads what OTHER workgroups of the group wrote in the previous step) and times a step.
// build: hipcc --offload-arch=gfx950 -O3 xcdsync.hip -o xcdsync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }  // HW_REG_XCC_ID[3:0]

struct Sync {
  unsigned registered;        // workgroups that have reported (system scope)
  unsigned members[8];        // per XCD
  int err;
  unsigned pad[22];
  unsigned count[8][32];      // per-XCD barrier counters, one 128-B line each
};

// a[g][i] of step s+1 = a[g][i] + a[g][(i + 4099) % n] of step s, per XCD group g (double buffered)
template <int MODE>  // 0: buffer_inv sc0; 1: buffer_inv sc1; 2: buffer_inv sc0 sc1; 3: agent-scope relaxed atomic loads, no invalidate
__global__ __launch_bounds__(1024) void k_xcd(int* a, int* b, int n, int steps, Sync* S, unsigned* xcc_of_wg) {
  __shared__ unsigned s_slot, s_size, s_xcc;
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id();
    s_xcc = x;
    xcc_of_wg[blockIdx.x] = x;
    s_slot = __hip_atomic_fetch_add(&S->members[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_add(&S->registered, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned spins = 0;
    while (__hip_atomic_load(&S->registered, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < gridDim.x) {
      if (++spins > (1u << 22)) { S->err = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    s_size = __hip_atomic_load(&S->members[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  const unsigned xcc = s_xcc, slot = s_slot, gsize = s_size;
  int* ga = a + (size_t)xcc * n;
  int* gb = b + (size_t)xcc * n;
  unsigned* cnt = &S->count[xcc][0];
  const int t = slot * blockDim.x + threadIdx.x, nt = gsize * blockDim.x;
  for (int s = 0; s < steps; ++s) {
    int* src = (s & 1) ? gb : ga;
    int* dst = (s & 1) ? ga : gb;
    for (int i = t; i < n; i += nt) {
      if (MODE == 3) dst[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + __hip_atomic_load(&src[(i + 4099) % n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else dst[i] = src[i] + src[(i + 4099) % n];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt vmcnt(0): this wave's stores have reached the L2
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(s + 1) * gsize;
      unsigned spins = 0;
      while (__hip_atomic_fetch_add(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > (1u << 20)) { S->err = 2; break; }
      }
    }
    __syncthreads();
    if (MODE == 0) asm volatile("buffer_inv sc0" ::: "memory");  // drop this CU's vector cache: the next step's loads come from the L2
    if (MODE == 1) asm volatile("buffer_inv sc1" ::: "memory");
    if (MODE == 2) asm volatile("buffer_inv sc0 sc1" ::: "memory");
  }
}

int main(int argc, char** argv) {
  const int n = 16384;
  const int steps = argc > 1 ? atoi(argv[1]) : 2000;
  int *a, *b; Sync* S; unsigned* xw;
  CK(hipMalloc(&a, 8 * n * 4)); CK(hipMalloc(&b, 8 * n * 4)); CK(hipMalloc(&S, sizeof(Sync))); CK(hipMalloc(&xw, 4096));
  std::vector<int> h(8 * n), ref(n), tmp(n);
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  for (int nwg : {64, 128, 256}) {
    for (int i = 0; i < 8 * n; ++i) h[i] = (i % n) % 7;
    CK(hipMemcpy(a, h.data(), 8 * n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(S, 0, sizeof(Sync)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    if (mode == 0) hipLaunchKernelGGL(k_xcd<0>, dim3(nwg), dim3(1024), 0, 0, a, b, n, steps, S, xw);
    if (mode == 1) hipLaunchKernelGGL(k_xcd<1>, dim3(nwg), dim3(1024), 0, 0, a, b, n, steps, S, xw);
    if (mode == 2) hipLaunchKernelGGL(k_xcd<2>, dim3(nwg), dim3(1024), 0, 0, a, b, n, steps, S, xw);
    if (mode == 3) hipLaunchKernelGGL(k_xcd<3>, dim3(nwg), dim3(1024), 0, 0, a, b, n, steps, S, xw);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    Sync hs; CK(hipMemcpy(&hs, S, sizeof(Sync), hipMemcpyDeviceToHost));
    std::vector<unsigned> hx(nwg); CK(hipMemcpy(hx.data(), xw, nwg * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) ref[i] = i % 7;
    for (int s = 0; s < steps; ++s) { for (int i = 0; i < n; ++i) tmp[i] = ref[i] + ref[(i + 4099) % n]; ref.swap(tmp); }
    CK(hipMemcpy(h.data(), (steps & 1) ? b : a, 8 * n * 4, hipMemcpyDeviceToHost));
    int bad = 0, rr = 0;
    for (int g = 0; g < 8; ++g) { if (!hs.members[g]) continue; for (int i = 0; i < n; ++i) bad += h[g * n + i] != ref[i]; }
    for (int w = 0; w < nwg; ++w) rr += hx[w] == (unsigned)(w % 8);
    printf("mode %d nwg %3d: members per XCD %u %u %u %u %u %u %u %u, wg i on XCD i%%8: %d of %d, err %d, wrong values %d, %.2f us per step\n", mode, nwg, hs.members[0], hs.members[1], hs.members[2],
           hs.members[3], hs.members[4], hs.members[5], hs.members[6], hs.members[7], rr, nwg, hs.err, bad, ms * 1e3 / steps);
  }
  return 0;
}
