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
};

template <typename BP, int PA_MTY, int MINW>
__global__ __launch_bounds__(64 * (PA_MTY + 3), MINW) void k_gradcurv_march(BP bp, MarchArgs A) {
  FabView P, O;
  DBox V;
  double dxinv[3];
  constexpr int PA_MROWS = PA_MTY + 2;
  const int box = A.order ? (int)((blockIdx.x / (unsigned)A.txy_max) % (unsigned)A.nboxes) : (int)blockIdx.y;
  if (!bp.get(box, P, O, V, dxinv)) return;
  const int pcomp = A.pcomp, kseg = A.kseg;
  const double pmin = A.pmin, invd = A.invdenom, thr = A.thr;
  const int nx = V.hi[0] - V.lo[0] + 1, ny = V.hi[1] - V.lo[1] + 1, nz = V.hi[2] - V.lo[2] + 1;
  const int tx = (nx + 63) / 64, ty = (ny + PA_MTY - 1) / PA_MTY, tz = (nz + kseg - 1) / kseg;
  unsigned bid = blockIdx.x;
  if (A.order) {
    const unsigned t = bid % (unsigned)A.txy_max, z = bid / ((unsigned)A.txy_max * (unsigned)A.nboxes);
    if (t >= (unsigned)tx * ty || z >= (unsigned)tz) return;
    bid = z * (unsigned)(tx * ty) + t;
  }
  if (bid >= (unsigned)tx * ty * tz) return;  // uniform for the whole workgroup
  const int bx = bid % tx, by = (bid / tx) % ty, bz = bid / (tx * ty);
  const int i0 = V.lo[0] + bx * 64, j0 = V.lo[1] + by * PA_MTY;
  const int k0 = V.lo[2] + bz * kseg, k1 = min(k0 + kseg - 1, V.hi[2]);
  const int iR = min(i0 + 64, V.hi[0] + 1);  // column right of the tile's last valid column
  const int llast = iR - 1 - i0;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rtop = min(PA_MROWS - 1, V.hi[1] + 1 - j0 + 1);  // row slot of the last live row (j = min(j0+MTY, hi_y+1))

  __shared__ MarchLds<PA_MTY> S;
  const long long pps = (long long)P.nx * P.ny;  // plane stride
  const int niter = k1 - k0 + 3;                 // planes k0-1 .. k1+1
#define PA_PROG(x) (((x) - pmin) * invd)         /* curvature.cpp:319 */

  if (w < PA_MROWS && w > rtop) {
    // ---------------------------------------------------------------- dead row (partial tile)
    for (int it = 0; it <= niter; ++it) __syncthreads();
    return;
  }

  if (w < PA_MROWS) {
    // ------------------------------------------------------------------------- row waves
    const int rr = w;
    const int j = j0 + rr - 1;
    const int le = min(lane, llast);  // lanes past the box edge mirror the last valid lane
    const int i = i0 + le;
    const int xs = le + 1;
    const bool halo = (rr == 0) || (rr == rtop);  // supplies neighbours only; one y-neighbour comes from global
    const double* gp = P.p + P.idx(i, j, k0 - 2, pcomp);
    // phi queue: pm = phi(p-2), pc = phi(p-1), p0 = phi(p), p1 = phi(p+1), p2 = (prefetched) phi(p+2)
    double pm = 0, pc = gp[0], p0 = gp[pps], p1 = gp[2 * pps], p2;
    gp += 2 * pps;  // -> phi(k0)
    double cm = PA_PROG(pc), cc = PA_PROG(p0), cp = PA_PROG(p1);  // c at planes p-1, p, p+1  (p = k0-1)
    S.c[0][rr][xs] = cc;
    if (halo) {
      const int jout = (rr == 0) ? j - 1 : j + 1;
      const double* go = P.p + P.idx(i, jout, k0 - 1, pcomp);
      double co = PA_PROG(go[0]), con;
      __syncthreads();
      int sp = 0;
      for (int p = k0 - 1; p <= k1 + 1; ++p) {
        const int sp1 = (sp == 2) ? 0 : sp + 1;
        const long long pi = (p <= k1) ? pps : 0;  // clamp the prefetch on the last plane
        gp += pi; go += pi;
        p2 = gp[0]; con = go[0];
        const double cl = S.c[sp][rr][xs - 1], cr = S.c[sp][rr][xs + 1];
        const double cin = S.c[sp][(rr == 0) ? 1 : rr - 1][xs];
        const double cs = (rr == 0) ? co : cin, cn = (rr == 0) ? cin : co;
        double nxp, nyp, nzp;
        normal_from(cl, cr, cs, cn, cm, cc, cp, dxinv, nxp, nyp, nzp);
        S.ny[sp][rr][lane] = nyp;
        S.c[sp1][rr][xs] = cp;
        S.p[sp][rr][xs] = p0;
        __syncthreads();
        cm = cc; cc = cp; cp = PA_PROG(p2); co = PA_PROG(con);
        p0 = p1; p1 = p2;
        sp = sp1;
      }
      return;
    }
    // output rows
    __syncthreads();
    double nxq = 0, nyq = 0, nzq = 0, nzqm = 0;
    double* op = O.p + O.idx(i, j, k0, A.ocomp);
    const long long ops = (long long)O.nx * O.ny, osc = O.sc;
    int sp = 0;
    // normal at plane p, outputs at plane q = p-1.  The 8 results of a plane are kept in registers
    // and stored DURING the next plane's normal computation, two at a time between its stages:
    // a burst of 8 stores per wave right after the barrier fills the CU's memory pipe and blocks
    // every wave (and the prefetch loads queued behind them) until it drains.  The first three
    // iterations have nothing valid to store yet: they write to plane k0, which the same thread
    // overwrites in program order at p = k0+2.  Keeps the loop free of branches around global
    // memory operations (vmcnt(N) instead of vmcnt(0)).
    double o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0, o5 = 0, o6 = 0, o7 = 0;
#pragma unroll 1
    for (int p = k0 - 1; p <= k1 + 1; ++p) {
      const int sp1 = (sp == 2) ? 0 : sp + 1;
      const int sq = (sp == 0) ? 2 : sp - 1;
      gp += (p <= k1) ? pps : 0;
      p2 = gp[0];
      const double cl = S.c[sp][rr][xs - 1], cr = S.c[sp][rr][xs + 1];
      const double cs = S.c[sp][rr - 1][xs], cn = S.c[sp][rr + 1][xs];
      op[0] = o0;
      __builtin_amdgcn_sched_barrier(0);
      const double ggx = cdiff(dxinv[0], cl, cc, cr);
      const double ggy = cdiff(dxinv[1], cs, cc, cn);
      const double ggz = cdiff(dxinv[2], cm, cc, cp);
      __builtin_amdgcn_sched_barrier(0);
      op[osc] = o1;
      __builtin_amdgcn_sched_barrier(0);
      const double sn = sqrt(ggx * ggx + ggy * ggy + ggz * ggz);
      const double ng = -((1e-14 < sn) ? sn : 1e-14);
      __builtin_amdgcn_sched_barrier(0);
      op[2 * osc] = o2;
      __builtin_amdgcn_sched_barrier(0);
      const double nxp = ggx / ng, nyp = ggy / ng;
      __builtin_amdgcn_sched_barrier(0);
      op[3 * osc] = o3;
      __builtin_amdgcn_sched_barrier(0);
      const double nzp = ggz / ng;
      S.ny[sp][rr][lane] = nyp;
      S.nx[sp][rr - 1][xs] = nxp;
      S.c[sp1][rr][xs] = cp;
      S.p[sp][rr][xs] = p0;
      __builtin_amdgcn_sched_barrier(0);
      op[4 * osc] = o4;
      __syncthreads();
      const double nxl = S.nx[sq][rr - 1][xs - 1], nxr = S.nx[sq][rr - 1][xs + 1];
      const double nys = S.ny[sq][rr - 1][lane], nyn = S.ny[sq][rr + 1][lane];
      double curv = 0.0;
      curv += cdiff(dxinv[0], nxl, nxq, nxr);
      curv += cdiff(dxinv[1], nys, nyq, nyn);
      curv += cdiff(dxinv[2], nzqm, nzq, nzp);
      curv = curv * 0.5;
      __builtin_amdgcn_sched_barrier(0);
      op[5 * osc] = o5;
      __builtin_amdgcn_sched_barrier(0);
      // phi gradient at plane q (pc = phi(q), pm = phi(q-1), p0 = phi(q+1))
      const double pl = S.p[sq][rr][xs - 1], pr = S.p[sq][rr][xs + 1];
      const double ps = S.p[sq][rr - 1][xs], pnn = S.p[sq][rr + 1][xs];
      const double gx = cdiff(dxinv[0], pl, pc, pr);
      const double gy = cdiff(dxinv[1], ps, pc, pnn);
      const double gz = cdiff(dxinv[2], pm, pc, p0);
      __builtin_amdgcn_sched_barrier(0);
      op[6 * osc] = o6;
      __builtin_amdgcn_sched_barrier(0);
      const double gm = sqrt(gx * gx + gy * gy + gz * gz);
      // threshold clip (curvature.cpp:557-566); cm = c at plane q.  thr < 0 never clips.
      const bool clip = (thr >= 0.0) && ((cm < thr) || (cm > 1.0 - thr));
      __builtin_amdgcn_sched_barrier(0);
      op[7 * osc] = o7;
      op += (p >= k0 + 2) ? ops : 0;
      o0 = gx; o1 = gy; o2 = gz; o3 = gm;
      o4 = clip ? 0.0 : nxq;
      o5 = clip ? 0.0 : nyq;
      o6 = clip ? 0.0 : nzq;
      o7 = clip ? 0.0 : curv;
      cm = cc; cc = cp; cp = PA_PROG(p2);
      pm = pc; pc = p0; p0 = p1; p1 = p2;
      nzqm = nzq; nxq = nxp; nyq = nyp; nzq = nzp;
      sp = sp1;
    }
    // results of the last plane (k1)
    op[0] = o0; op[osc] = o1; op[2 * osc] = o2; op[3 * osc] = o3;
    op[4 * osc] = o4; op[5 * osc] = o5; op[6 * osc] = o6; op[7 * osc] = o7;
    return;
  }

  // ----------------------------------------------------------------------------- edge wave
  {
    const int l20 = lane % (2 * PA_MROWS);  // idle lanes mirror the active ones
    const int rr = min(l20 >> 1, rtop);
    const int side = l20 & 1;
    const int j = j0 + rr - 1;
    const int i = side ? iR : i0 - 1;
    const int xs = side ? llast + 2 : 0;
    const int xin = side ? llast + 1 : 1;  // the tile column next to this edge column
    const int rlo = max(rr - 1, 0), rhi = min(rr + 1, PA_MROWS - 1);
    const bool has_n = (rr >= 1 && rr <= PA_MTY);
    const double* gp = P.p + P.idx(i, j, k0 - 2, pcomp);
    const double* go = P.p + P.idx(side ? i + 1 : i - 1, j, k0 - 1, pcomp);
    double pc = gp[0], p0 = gp[pps], p1 = gp[2 * pps], p2;
    gp += 2 * pps;
    double cm = PA_PROG(pc), cc = PA_PROG(p0), cp = PA_PROG(p1);
    double co = PA_PROG(go[0]), con;
    S.c[0][rr][xs] = cc;
    __syncthreads();
    int sp = 0;
    for (int p = k0 - 1; p <= k1 + 1; ++p) {
      const int sp1 = (sp == 2) ? 0 : sp + 1;
      const long long pi = (p <= k1) ? pps : 0;
      gp += pi; go += pi;
      p2 = gp[0]; con = go[0];
      const double inner = S.c[sp][rr][xin];
      const double cl = side ? inner : co, cr = side ? co : inner;
      const double cs = S.c[sp][rlo][xs], cn = S.c[sp][rhi][xs];
      double nxp, nyp, nzp;
      normal_from(cl, cr, cs, cn, cm, cc, cp, dxinv, nxp, nyp, nzp);
      if (has_n) S.nx[sp][rr - 1][xs] = nxp;
      S.c[sp1][rr][xs] = cp;
      S.p[sp][rr][xs] = p0;
      __syncthreads();
      cm = cc; cc = cp; cp = PA_PROG(p2); co = PA_PROG(con);
      p0 = p1; p1 = p2;
      sp = sp1;
    }
  }
#undef PA_PROG
}
