// pa_fused_march.h -- fused grad->curvature, "k-marching" kernel for gfx950.
//
// Reads ONLY phi (8 B/cell) and writes gx,gy,gz,|g|,Nx,Ny,Nz,K (64 B/cell): the 72 B/cell
// algorithmic traffic of SURVEY 8(d).  The progress variable c = (phi-pmin)*invdenom
// (curvature.cpp:316-320) is formed on the fly from the same loads.
//
// Workgroup = MTY+3 wavefronts over a tile of 64 (x) x MTY (y) columns that marches through
// kseg z-planes:
//   waves 0..MTY+1  one row each: rows j0-1 .. j0+MTY.  Output rows produce the 8 results; the two
//                   outer ("halo") rows only supply c / phi / n_y to their neighbours,
//   wave  MTY+2     the "edge" wave: the columns left/right of the tile (c, phi, n_x only).
// Each thread owns one (i,j) column.  z-neighbours of phi, c and n_z live in registers (rolling
// queues fed by ONE coalesced global load per plane, issued two planes ahead); x/y-neighbours of
// c, phi, n_x, n_y go through a 3-slot LDS ring with ONE barrier per plane.  The flame normal is
// therefore evaluated ~1.2-1.4x per cell instead of 7x (direct form) and never touches HBM.
//
// The per-role loops are branch-free around global memory operations (lanes past the box edge
// mirror the last valid lane instead of being masked; the two warm-up planes store to the first
// output plane and are overwritten in program order) so that hipcc can count the outstanding
// loads/stores and keep the prefetched plane in flight across the barrier (s_waitcnt vmcnt(N)
// instead of vmcnt(0)).
//
// Ghost cells of phi that are NOT valid cells of the level (coarse-fine / physical walls) give a
// c that differs from the reference's (which applies the boundary condition to c itself); every
// result that depends on such a cell lies within two cells of a coarse-fine or wall face and is
// recomputed by k_gradcurv_faces (pa_fused.hip).  Everywhere else results are bit-identical to
// the pass-by-pass path and to the CPU oracle (same operation order: cdiff, normal_from).
#pragma once
#include "pa_fabview.h"

#define PA_MLW 66

__device__ __forceinline__ void normal_from(double cl, double cr, double cs, double cn, double cm, double cc, double cp,
                                            const double dxinv[3], double& nx, double& ny, double& nz) {
  const double gx = cdiff(dxinv[0], cl, cc, cr);
  const double gy = cdiff(dxinv[1], cs, cc, cn);
  const double gz = cdiff(dxinv[2], cm, cc, cp);
  const double sn = sqrt(gx * gx + gy * gy + gz * gz);
  const double ng = -((1e-14 < sn) ? sn : 1e-14);
  nx = gx / ng;
  ny = gy / ng;
  nz = gz / ng;
}

template <int MTY>
struct MarchLds {
  double c[3][MTY + 2][PA_MLW];   // c: x index 0 = left edge column, 1..64 = lanes, llast+2 = right edge column
  double p[3][MTY + 2][PA_MLW];   // phi
  double nx[3][MTY][PA_MLW];      // n_x of rows 1..MTY (+ edge columns)
  double ny[3][MTY + 2][64];      // n_y
  alignas(16) double h[2][MTY + 2][2][6];  // NCG hand-over (pa_fused_march3.h): per plane parity, row and side (N_x of the first three cells, y term, z term of K, -)
};

struct MarchArgs {
  int pcomp, ocomp, kseg;
  double pmin, invdenom, thr;
  // order = 1: 1-D grid, z-segment slowest across ALL boxes (the chip works on the same few planes of
  // every box at a time); order = 0: grid.y = box, all tiles of a box are consecutive
  // order = 2 (k_gradcurv_march3): 1-D grid, XCD-aware.  Workgroups are dealt round-robin over the 8
  // XCDs (blocks b and b+8 share one), so block L = 8*T*g + 8*t + q works on tile t of box 8*g + q:
  // all tiles of a box run on ONE XCD at about the same time and the halo rows / partial lines that
  // neighbouring tiles both read are served by that XCD's L2 instead of being fetched once per XCD.
  int order, nboxes, txy_max, tiles_max;
  int cg = 0;  // host side: launch the CG variant of k_gradcurv_march3 (pa_fused_march3.h)
  // NCG (pa_fused_march3.h, wide CG sweep without the clip): for the first cell behind a special X face the sweep writes N_x of the
  // first three cells and the y and z terms of K into the level's face-major arrays of PAIRS ((double2*)ncg)[pair * ncgs + cgoff + ...],
  // pair = 0 .. 2, so that the fix-up of those cells (k_faces_curv_fast) reads three contiguous streams instead of 8 bytes of five
  // different lines per cell
  double* ncg = nullptr;
  long long ncgs = 0;
  const int* boxlist = nullptr;  // k_gradcurv_march3 / march3n: the launch covers boxes boxlist[0 .. nboxes-1] of the level (null: all of them)
  // k_gradcurv_march3 / march3n on boxes of different sizes: workgroup i works on tile wgtab[2 i + 1] of box wgtab[2 i] (< 0: none).
  // With order 2 every box gets the tile count of the LARGEST box of the launch and the others exit at once -- on a Pele BoxArray
  // (boxes of 32 .. 128 cells per side) 2 of 3 workgroups; the table has none of those and keeps order 2's property (workgroups
  // i and i + 8, one XCD, work on the same box).
  const int* wgtab = nullptr;
  // GOUT variants (curvature.cpp with options: pa_curvature_run): the sweep stores Progress, K, N at out components
  // ocomp .. ocomp + 4 and the cell-centred gradient of c (curvature.cpp:457-490, "cell_normal" before its normalisation:
  // the field do_gaussCurv differentiates again) at components 0 .. 2 of this second multifab instead of grad phi
  double* gdata = nullptr;
  const long long* goff = nullptr;
  int gng = 0;
  // GOUT == 2: one byte per wgtab entry, where the tile stores G (pa_sweep_gneed); null: in every cell
  const unsigned char* gneed = nullptr;
};

// (The first marching kernel, k_gradcurv_march -- requests and stores interleaved, one plane in flight -- lived here until round 6;
// pa_fused_march3.h is its restructured form, measured 2.27 -> 1.92 ms per launch, bit-identical.)
