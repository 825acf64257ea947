// membench7.hip -- the WRITE PATTERN of the tile-marching sweeps without their arithmetic: which tile shapes would the memory
// side reward?  One 512^3-cell level of boxes nx x ny x nz, 8 output components per box laid out [comp][k][j][i] (component
// stride padded by 2 KiB like pa_cstride), one input component.  A workgroup owns a tile of wcols columns x R rows and marches
// kseg planes; per plane every wave loads its cells of the input (8 B per lane) and stores them to the 8 components (8 B per
// lane and component).  wave = 64 / wcols rows of wcols columns; ipw = store instructions per wave, component and plane (2:
// every lane serves two rows, ipw * 64 / wcols rows per wave; with xsplit the second instruction serves the other 64-column
// half of a 128-wide row instead).  Workgroups are numbered like MarchArgs order 2 (tiles of a box on one XCD).  A barrier per
// plane paces the waves like the sweep's LDS hand-over does.  Not part of the product.
//   build: hipcc --offload-arch=gfx950 -O3 membench7.hip -o membench7
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Cfg { const char* name; int nx, ny, nz, wcols, waves, ipw, xsplit, kseg; int idle = 0; };  // idle: extra waves that only take part in the barrier (halo / edge waves: the workgroup's size decides how many are resident per CU)

__global__ __launch_bounds__(1024) void k_tiles(const double* __restrict__ in, double* __restrict__ out, int nx, int ny, int nz, int wcols, int ipw, int xsplit, int kseg,
                                                 long long cs, int nboxes, int tiles, int idle, int nsleep) {
  const unsigned per8 = 8u * (unsigned)tiles, g = blockIdx.x / per8, r = blockIdx.x % per8;
  const int box = (int)(g * 8u + (r & 7u)), tile = (int)(r >> 3);
  if (box >= nboxes) return;
  const int waves = (blockDim.x >> 6) - idle, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rpw = 64 / wcols;                       // rows per wave instruction
  const int R = waves * rpw * (xsplit ? 1 : ipw);   // rows per tile
  const int tcols = xsplit ? wcols * ipw : wcols;   // columns per tile
  const int tx = nx / tcols, ty = (ny + R - 1) / R;
  const int bx = tile % tx, by = (tile / tx) % ty, bz = tile / (tx * ty);
  const int k0 = bz * kseg, k1 = min(k0 + kseg, nz);
  const long long boxcells = (long long)nx * ny * nz;
  const double* ib = in + box * boxcells;
  double* ob = out + box * 8 * cs;
  for (int k = k0; k < k1; ++k) {
    for (int q = 0; q < ipw && w < waves; ++q) {
      const int col = bx * tcols + (xsplit ? q * wcols : 0) + lane % wcols;
      int row = by * R + (xsplit ? w * rpw : (w * ipw + q) * rpw) + lane / wcols;
      row = min(row, ny - 1);  // rows past a partial tile repeat the last one (as the sweeps do)
      const long long o = ((long long)k * ny + row) * nx + col;
      const double v = ib[o];
#pragma unroll
      for (int c = 0; c < 8; ++c) ob[c * cs + o] = v + c;
    }
    __syncthreads();
    for (int z = 0; z < nsleep; ++z) __builtin_amdgcn_s_sleep(16);  // ~1024 cycles each: the compute / LDS phase of a real step (argv[1])
  }
}

// the same tiles with the real sweeps' step order: three planes of loads in flight (requested right AFTER the barrier, three steps ahead), the 8 stores of a
// plane issued in one burst at the top of the NEXT step (argv[2] = 1)
__global__ __launch_bounds__(1024) void k_tiles_pref(const double* __restrict__ in, double* __restrict__ out, int nx, int ny, int nz, int wcols, int kseg, long long cs, int nboxes,
                                                      int tiles, int idle, long long pitch_x, long long pitch_xy, int ghost) {
  const unsigned per8 = 8u * (unsigned)tiles, g = blockIdx.x / per8, r = blockIdx.x % per8;
  const int box = (int)(g * 8u + (r & 7u)), tile = (int)(r >> 3);
  if (box >= nboxes) return;
  const int waves = (blockDim.x >> 6) - idle, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rpw = 64 / wcols, R = waves * rpw;
  const int tx = nx / wcols, ty = (ny + R - 1) / R;
  const int bx = tile % tx, by = (tile / tx) % ty, bz = tile / (tx * ty);
  const int k0 = bz * kseg, k1 = min(k0 + kseg, nz);
  const long long boxcells = (long long)nx * ny * nz;
  // input: a FAB with `ghost` ghost layers (pitch_x = nx + 2 ghost, pitch_xy = pitch_x (ny + 2 ghost)), box b at b * pitch_xy * (nz + 2 ghost)
  const double* ib = in + box * pitch_xy * (nz + 2 * ghost) + ((long long)ghost * pitch_xy + ghost * pitch_x + ghost);
  double* ob = out + box * 8 * cs;
  const bool act = w < waves;
  const int col = bx * wcols + lane % wcols;
  const int row = min(by * R + w * rpw + lane / wcols, ny - 1);
  const long long oin = (long long)row * pitch_x + col, oout = (long long)row * nx + col;
  double f0 = 0, f1 = 0, f2 = 0, x = 0;
  if (act) { f0 = ib[(long long)k0 * pitch_xy + oin]; f1 = ib[(long long)min(k0 + 1, nz - 1) * pitch_xy + oin]; f2 = ib[(long long)min(k0 + 2, nz - 1) * pitch_xy + oin]; }
  bool have = false;
  long long kprev = 0;
#define STEP(F, K)                                                                                                           \
  {                                                                                                                          \
    if (act && have) {                                                                                                       \
      _Pragma("unroll") for (int c = 0; c < 8; ++c) ob[c * cs + kprev * nx * ny + oout] = x + c;                             \
    }                                                                                                                        \
    if (act) { asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(F)); }                                                        \
    __syncthreads();                                                                                                         \
    if (act) F = ib[(long long)min((K) + 3, nz - 1) * pitch_xy + oin];                                                       \
    have = true;                                                                                                             \
    kprev = (K);                                                                                                             \
  }
  int k = k0;
  for (; k + 2 < k1; k += 3) { STEP(f0, k) STEP(f1, k + 1) STEP(f2, k + 2) }
  if (k < k1) { STEP(f0, k) if (k + 1 < k1) STEP(f1, k + 1) }
#undef STEP
  if (act && have) {
#pragma unroll
    for (int c = 0; c < 8; ++c) ob[c * cs + kprev * nx * ny + oout] = x + c;
  }
}

int main(int argc, char** argv) {
  const int N = 512;
  const int nsleep = argc > 1 ? atoi(argv[1]) : 0;
  const int pref = argc > 2 ? atoi(argv[2]) : 0;  // 1: k_tiles_pref from plain input, 2: from FABs with 2 ghost layers
  const long long cells = (long long)N * N * N;
  double *in, *out;
  CK(hipMalloc(&in, cells * 8 * 3 / 2 + (1 << 20)));
  CK(hipMalloc(&out, cells * 64 + (1ll << 30)));
  CK(hipMemset(in, 0, cells * 8 * 3 / 2));
  const Cfg cfgs[] = {
      {"64^3 boxes, 64 x 13 tile (the wide sweep on 64^3)", 64, 64, 64, 64, 13, 1, 0, 64},
      {"128^3 boxes, 64 x 13 tile (the headline)", 128, 128, 128, 64, 13, 1, 0, 64},
      {"32^3 boxes, 32 x 16 tile, 2 rows per wave (the narrow sweep)", 32, 32, 32, 32, 8, 1, 0, 32},
      {"64^3 boxes, 64 x 5 tile", 64, 64, 64, 64, 5, 1, 0, 64},
      {"64^3 boxes, 64 x 26 tile: two rows per lane", 64, 64, 64, 64, 13, 2, 0, 64},
      {"128^3 boxes, 128 x 13 tile: both halves of a row per lane", 128, 128, 128, 64, 13, 2, 1, 64},
      {"128^3 boxes, 64 x 26 tile: two rows per lane", 128, 128, 128, 64, 13, 2, 0, 64},
      {"32^3 boxes, 32 x 32 tile: the whole plane of a box per workgroup", 32, 32, 32, 32, 8, 2, 0, 32},
      {"32^3 boxes, 32 x 32 tile, 16 waves", 32, 32, 32, 32, 16, 1, 0, 32},
      {"128^3 boxes, 64 x 16 tile (16 row waves, no halo / edge wave)", 128, 128, 128, 64, 16, 1, 0, 64},
      {"32^3 boxes, 32 x 16 tile + 2 idle waves = 640 threads (the narrow sweep's workgroup)", 32, 32, 32, 32, 8, 1, 0, 32, 2},
      {"32^3 boxes, 32 x 32 tile, two row pairs per lane + 2 idle waves = 640 threads", 32, 32, 32, 32, 8, 2, 0, 32, 2},
      {"32^3 boxes, 32 x 12 tile + 2 idle waves = 512 threads", 32, 32, 32, 32, 6, 1, 0, 32, 2},
      {"128^3 boxes, 64 x 13 tile + 3 idle waves = 1024 threads (the wide sweep's workgroup)", 128, 128, 128, 64, 13, 1, 0, 64, 3},
      {"64^3 boxes, 64 x 13 tile + 3 idle waves = 1024 threads", 64, 64, 64, 64, 13, 1, 0, 64, 3},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
    for (const Cfg& c : cfgs) {
      const long long boxcells = (long long)c.nx * c.ny * c.nz;
      const int nboxes = (int)(cells / boxcells);
      const long long cs = boxcells + 256;  // + 2 KiB
      const int rpw = 64 / c.wcols, R = c.waves * rpw * (c.xsplit ? 1 : c.ipw), tcols = c.xsplit ? c.wcols * c.ipw : c.wcols;
      const int tiles = (c.nx / tcols) * ((c.ny + R - 1) / R) * ((c.nz + c.kseg - 1) / c.kseg);
      const unsigned grid = (unsigned)tiles * 8u * (unsigned)((nboxes + 7) / 8);
      float best = 1e9f;
      for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0, 0));
        if (pref && c.ipw == 1 && !c.xsplit) {
          const int gh = pref == 2 ? 2 : 0;
          hipLaunchKernelGGL(k_tiles_pref, dim3(grid), dim3(64 * (c.waves + c.idle)), 0, 0, in, out, c.nx, c.ny, c.nz, c.wcols, c.kseg, cs, nboxes, tiles, c.idle, (long long)(c.nx + 2 * gh),
                             (long long)(c.nx + 2 * gh) * (c.ny + 2 * gh), gh);
        } else
        hipLaunchKernelGGL(k_tiles, dim3(grid), dim3(64 * (c.waves + c.idle)), 0, 0, in, out, c.nx, c.ny, c.nz, c.wcols, c.ipw, c.xsplit, c.kseg, cs, nboxes, tiles, c.idle, nsleep);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it) best = ms < best ? ms : best;
      }
      if (rep) printf("%-72s rows/tile %2d  contiguous run %5.1f KiB  workgroups %6u x %4d threads: %.3f ms = %.2f TB/s (72 B/cell)\n", c.name, R,
                      (c.nx == tcols ? R : 1) * tcols * 8 / 1024.0, grid, 64 * (c.waves + c.idle), best, cells * 72.0 / (best * 1e-3) / 1e12);
    }
  return 0;
}
