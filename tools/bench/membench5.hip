// membench5.hip -- where does the write rate of this device drop from the memset rate (6.5 TB/s) to the 5.25 TB/s
// the fused sweep's 1-read : 8-write mix gets?  A matrix over
//   S     output streams written concurrently by a workgroup (1, 2, 4, 8); total bytes written are the same for all S
//   RD    the 1/8 read stream on / off
//   W     bytes per lane per store (8 = dwordx2, 16 = dwordx4)
//   ROWS  consecutive wave-rows (256 lanes x W bytes) a workgroup writes into ONE stream before it moves on
//   MODE  0 interleaved (row r of every stream, then row r+1: what the sweep does), 1 stream-major (all rows of
//         stream 0, then stream 1, ...), 2 = persistent grid-stride form of mode 0 (2048 workgroups)
//   PAD   bytes added to the stream stride (0 = streams 2^n apart; the library pads to 512 B mod 16 KiB)
// No arithmetic beyond one add per store.  Not part of the product.  One dispatch per timed iteration, so that a
// rocprofv3 --pmc pass of the same binary gives counters per cell of the matrix (dispatch order = print order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
template <int W> struct Vec;
template <> struct Vec<8> { typedef double T; };
template <> struct Vec<16> { typedef d2 T; };

struct Cfg { int S, rd, W, rows, mode; long long pad; int nt; int fronts = 1; };

// one workgroup = 256 lanes; "unit" u = 8*rows wave-rows of output (rows per stream when S = 8) + rows/… of input
template <int W, bool RD, int NT>
__global__ __launch_bounds__(256) void k_mix(const char* __restrict__ in, char* __restrict__ out, int S, int rows, int mode,
                                             long long stream_stride, long long nunits, int fronts) {
  typedef typename Vec<W>::T V;
  const int rowb = 256 * W;                 // bytes per wave-row of the workgroup
  const int rps = rows * (8 / S);           // rows per stream per unit
  const long long ustep = (mode == 2) ? gridDim.x : nunits;
  for (long long u0 = blockIdx.x; u0 < nunits; u0 += ustep) {
    // fronts > 1: consecutive workgroups work on `fronts` write fronts that each advance sequentially (what co-resident
    // workgroups of a tile-marching kernel do) instead of on one front
    const long long u = fronts > 1 ? (u0 % fronts) * (nunits / fronts) + u0 / fronts : u0;
    V acc = V(0);
    if (RD) {
      // the unit's share of the input: rows wave-rows (1/8 of what it writes)
      const char* pi = in + u * (long long)rows * rowb + threadIdx.x * W;
      for (int r = 0; r < rows; ++r) acc += *(const V*)(pi + (long long)r * rowb);
    }
    char* po = out + u * (long long)rps * rowb + threadIdx.x * W;
    if (mode == 3) {  // FAB layout [box][stream][16 MiB of cells]: stream_stride = component stride inside a box
      const long long upb = (16LL << 20) / ((long long)rps * rowb), bx = u / upb, wu = u - bx * upb;
      po = out + bx * (S * stream_stride) + wu * (long long)rps * rowb + threadIdx.x * W;
    }
    if (mode == 1) {
      for (int s = 0; s < S; ++s)
        for (int r = 0; r < rps; ++r) {
          V v = acc + (double)(s + r);
          if (NT) __builtin_nontemporal_store(v, (V*)(po + s * stream_stride + (long long)r * rowb));
          else *(V*)(po + s * stream_stride + (long long)r * rowb) = v;
        }
    } else {
      for (int r = 0; r < rps; ++r)
        for (int s = 0; s < S; ++s) {
          V v = acc + (double)(s + r);
          if (NT) __builtin_nontemporal_store(v, (V*)(po + s * stream_stride + (long long)r * rowb));
          else *(V*)(po + s * stream_stride + (long long)r * rowb) = v;
        }
    }
  }
}

static char *g_in, *g_out;
static const long long OUT_BYTES = 8LL * 64 * 128 * 128 * 128 * 8;   // 8.59 GB: the 8 outputs of a 512^3 level
static const long long IN_BYTES = OUT_BYTES / 8;
static const long long SLACK = 3LL << 30;

template <int W, bool RD, int NT>
static void launch(const Cfg& c, long long stride, long long nunits) {
  const unsigned grid = (c.mode == 2) ? 2048u : (unsigned)nunits;
  hipLaunchKernelGGL((k_mix<W, RD, NT>), dim3(grid), dim3(256), 0, 0, g_in, g_out, c.S, c.rows, c.mode, stride, nunits, c.fronts);
}

static int run(const Cfg& c, int nit) {
  const long long per_stream = c.mode == 3 ? (16LL << 20) : OUT_BYTES / c.S;
  const long long stride = per_stream + c.pad;
  const long long unit_out = 8LL * c.rows * 256 * c.W;
  const long long nunits = OUT_BYTES / unit_out;
  if ((c.mode == 3 ? stride * c.S * (OUT_BYTES / c.S / per_stream) : stride * (c.S - 1) + per_stream) > OUT_BYTES + SLACK) { printf("pad too large\n"); return 0; }
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f, sum = 0;
  for (int it = 0; it < nit; ++it) {
    CK(hipEventRecord(a));
    if (c.W == 8) { if (c.rd) { if (c.nt) launch<8, true, 1>(c, stride, nunits); else launch<8, true, 0>(c, stride, nunits); }
                    else { if (c.nt) launch<8, false, 1>(c, stride, nunits); else launch<8, false, 0>(c, stride, nunits); } }
    else { if (c.rd) { if (c.nt) launch<16, true, 1>(c, stride, nunits); else launch<16, true, 0>(c, stride, nunits); }
           else { if (c.nt) launch<16, false, 1>(c, stride, nunits); else launch<16, false, 0>(c, stride, nunits); } }
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (it > 0 || nit == 1) { sum += ms; if (ms < best) best = ms; }
  }
  CK(hipGetLastError());
  const double bytes = (double)OUT_BYTES + (c.rd ? (double)IN_BYTES : 0.0);
  if (c.fronts > 1) printf("fronts %6d ", c.fronts);
  printf("S %d rd %d W %2d rows %3d mode %d pad %8lld nt %d | best %.3f ms  %5.0f GB/s | mean %.3f ms\n", c.S, c.rd, c.W, c.rows,
         c.mode, c.pad, c.nt, best, bytes / best / 1e6, sum / (nit > 1 ? nit - 1 : 1));
  fflush(stdout);
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return 0;
}

int main(int argc, char** argv) {
  const int nit = argc > 1 ? atoi(argv[1]) : 6;
  const char* set = argc > 2 ? argv[2] : "all";
  CK(hipMalloc(&g_in, IN_BYTES));
  CK(hipMalloc(&g_out, OUT_BYTES + SLACK));
  CK(hipMemset(g_in, 0, IN_BYTES));
  CK(hipMemset(g_out, 0, OUT_BYTES + SLACK));
  CK(hipDeviceSynchronize());
  // library-style pad: 512 B past a multiple of 16 KiB
  const long long LP = 512;
  if (!strcmp(set, "all") || !strcmp(set, "matrix")) {
    for (int W : {8, 16})
      for (int rd : {0, 1})
        for (int S : {1, 2, 4, 8})
          for (long long pad : {0LL, LP}) {
            if (S == 1 && pad) continue;
            if (run(Cfg{S, rd, W, 1, 0, pad, 0}, nit)) return 1;
          }
  }
  if (!strcmp(set, "all") || !strcmp(set, "shape")) {
    // contiguity per stream visit, order of the stores, persistence, pads at other multiples
    for (int rows : {2, 4, 16})
      for (int mode : {0, 1})
        if (run(Cfg{8, 1, 16, rows, mode, LP, 0}, nit)) return 1;
    for (int rows : {1, 4})
      if (run(Cfg{8, 1, 16, rows, 2, LP, 0}, nit)) return 1;
    for (int S : {1, 8})
      if (run(Cfg{S, 0, 16, 1, 2, S == 1 ? 0 : LP, 0}, nit)) return 1;
    for (long long pad : {64LL, 128LL, 256LL, 1024LL, 2048LL, 4096LL, 8192LL, 4096LL + 512, 65536LL + 512, (1LL << 20) + 512, (2LL << 20) + 4096 + 512})
      if (run(Cfg{8, 1, 16, 1, 0, pad, 0}, nit)) return 1;
    for (int S : {1, 8})
      if (run(Cfg{S, 1, 16, 1, 0, S == 1 ? 0 : LP, 1}, nit)) return 1;
  }
  if (!strcmp(set, "stride")) {
    // the component stride inside a box ([box][comp][cells], 8 streams 16 MiB + pad apart) against streams 1 GiB apart
    if (run(Cfg{8, 1, 8, 1, 0, 512, 0}, nit)) return 1;
    for (long long pad : {0LL, 512LL, 1024LL, 2048LL, 4096LL, 4096LL + 512, 8192LL + 512, 16384LL + 512, 32768LL + 512, 65536LL + 512, (128LL << 10) + 512, (256LL << 10) + 512,
                          (512LL << 10) + 512, (1LL << 20) + 512, (2LL << 20) + 512, (4LL << 20) + 512, (8LL << 20) + 512, (16LL << 20) + 512, (1LL << 20), (2LL << 20), (4LL << 20), (4LL << 20) + 4096, (5LL << 20) + 512})
      if (run(Cfg{8, 1, 8, 1, 3, pad, 0}, nit)) return 1;
    if (run(Cfg{8, 1, 8, 1, 0, 512, 0}, nit)) return 1;
  }
  if (!strcmp(set, "fronts")) {
    // how many sequential write fronts can be open at once before the rate drops?  (boxed layout, 2 KiB pad)
    for (int rows : {1, 4})
      for (int F : {1, 8, 64, 256, 1024, 2048, 4096, 16384, 65536}) {
        Cfg c{8, 1, 8, rows, 3, 2048, 0};
        c.fronts = F;
        if (run(c, nit)) return 1;
      }
  }
  if (!strcmp(set, "fronts3")) {  // three cells for the wide counter pass (tools/prof.sh membench)
    for (int F : {1, 64, 4096}) {
      Cfg c{8, 1, 8, 1, 3, 2048, 0};
      c.fronts = F;
      if (run(c, nit)) return 1;
    }
  }
  return 0;
}
