// pa_fused_march3n.h -- fused grad->curvature sweep for NARROW boxes (at most 32 cells in x, AMReX's default
// max_grid_size in 3-D): k_gradcurv_march3's algorithm, arithmetic and step schedule with TWO rows of 32
// columns per wavefront, so that a wave's loads and stores stay 512 contiguous bytes (two adjacent 256-byte rows
// of a 32-wide output FAB) instead of half-empty 64-lane rows (measured with the wide kernel on 32^3 boxes:
// 2.72 ms per 512^3 level against 1.93 ms on 128^3 boxes).
//   waves 0..NRW-1   two output rows each: lanes 0-31 row 2w, lanes 32-63 row 2w+1 of the tile (NRW*2 rows),
//   wave  NRW        the two halo rows: lanes 0-31 the row below the tile, lanes 32-63 the row above,
//   wave  NRW+1      the edge wave: the columns left/right of the tile (c, phi, n_x only).
// Rows past the end of a partial tile are clamped onto the last valid row: such lanes repeat that row's
// computation exactly (same LDS slots, same global addresses) and store the same values again.
// Everything else -- the 3-slot LDS ring, three planes in flight per role, the burst of 8 stores at the top of
// a step and the requests after the barrier, scalar base + lane offset addressing, carried z-fluxes,
// div3_shared -- is as in pa_fused_march3.h (read that header first).  Bit-identical to it and to the oracle.
#pragma once
#include "pa_fused_march3.h"

template <int NRW>
struct MarchLdsN {
  static constexpr int ROWS = 2 * NRW + 2, LW = 34;
  double c[3][ROWS][LW];       // x index 0 = left edge column, 1..32 = columns, llast+2 = right edge column
  double p[3][ROWS][LW];
  double nx[3][2 * NRW][LW];   // n_x of the output rows (+ edge columns)
  double ny[3][ROWS][32];
};

// CG (exact-normal pipeline, see pa_fused_march3.h): the progress variable in the ghost cells behind SPECIAL faces comes from
// the level's compact face-major arrays.  z faces: the output rows take plane lo_z - 1 in the prologue and plane hi_z + 1
// through one select per step; y faces: the halo wave's lanes of that side read the array [plane][x] with their second
// stream (mode 1: the halo row IS the ghost row, the stream runs one plane ahead and supplies its c; mode 2: the tile ends
// one row short of the box, the row BEYOND the halo row is the ghost row); x faces: the edge wave's lanes likewise with the
// array [plane][y].  Both waves address the second stream through per-lane pointers when CG is on.
// GOUT: as in pa_fused_march3.h -- Progress, K, N at out components ocomp .. ocomp + 4 and G into a second multifab.
template <typename BP, int NRW, bool CLIP, bool CG = false, bool GOUT = false>
__device__ __forceinline__ void gradcurv_march3n_body(const BP& bp, const MarchArgs& A, const unsigned bid_x, const unsigned bid_y) {
  FabView P, O;
  DBox V;
  double dxinv[3];
  constexpr int MTY2 = 2 * NRW, MROWS = MTY2 + 2;
  unsigned bid = bid_x;
  int box;
  if (A.wgtab) {
    box = A.wgtab[2 * bid_x];
    if (box < 0) return;
    bid = (unsigned)A.wgtab[2 * bid_x + 1];
  } else if (A.order == 2) {
    const unsigned per8 = 8u * (unsigned)A.tiles_max, g = bid / per8, r = bid % per8;
    box = (int)(g * 8u + (r & 7u));
    bid = r >> 3;
    if (box >= A.nboxes) return;
  } else {
    box = (int)bid_y;
  }
  if (A.boxlist && !A.wgtab) {
    if (box >= A.nboxes) return;
    box = A.boxlist[box];
  }
  if (!bp.get(box, P, O, V, dxinv)) return;
  const int pcomp = A.pcomp, kseg = A.kseg;
  const double pmin = A.pmin, invd = A.invdenom;
  const int nx = V.hi[0] - V.lo[0] + 1, ny = V.hi[1] - V.lo[1] + 1, nz = V.hi[2] - V.lo[2] + 1;
  const int tx = (nx + 31) / 32, ty = (ny + MTY2 - 1) / MTY2, tz = (nz + kseg - 1) / kseg;
  if (bid >= (unsigned)tx * ty * tz) return;  // uniform for the whole workgroup
  const int bx = bid % tx, by = (bid / tx) % ty, bz = bid / (tx * ty);
  const int i0 = V.lo[0] + bx * 32, j0 = V.lo[1] + by * MTY2;
  const int k0 = V.lo[2] + bz * kseg, k1 = min(k0 + kseg - 1, V.hi[2]);
  const int iR = min(i0 + 32, V.hi[0] + 1);  // column right of the tile's last valid column
  const int llast = iR - 1 - i0;
  const int nrows = min(MTY2, V.hi[1] - j0 + 1);  // valid output rows of this tile
  const int rtop = nrows + 1;                     // row slot of the upper halo row
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rsub = lane >> 5, col = lane & 31;
  const double *cgxl = nullptr, *cgxh = nullptr, *cgyl = nullptr, *cgyh = nullptr, *cgzl = nullptr, *cgzh = nullptr;
  int xhmode = 0, yhmode = 0;
  if (CG) {
    if (bx == 0) cgxl = bp.cg_face(box, 0);
    if (iR >= V.hi[0]) { cgxh = bp.cg_face(box, 1); xhmode = cgxh ? (iR == V.hi[0] + 1 ? 1 : 2) : 0; }
    if (by == 0) cgyl = bp.cg_face(box, 2);
    if (j0 + nrows >= V.hi[1]) { cgyh = bp.cg_face(box, 3); yhmode = cgyh ? (j0 + nrows == V.hi[1] + 1 ? 1 : 2) : 0; }
    if (k0 == V.lo[2]) cgzl = bp.cg_face(box, 4);
    if (k1 >= V.hi[2] - 1) cgzh = bp.cg_face(box, 5);
  }

  __shared__ MarchLdsN<NRW> S;
  const long long pps = (long long)P.nx * P.ny * 8;  // plane stride of phi, bytes
  const int pend = k1 + 1;
  const int kfmax = k1 + 2;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
#define PA_PROG(x) (((x) - pmin) * invd) /* curvature.cpp:319 */
#define PA_RUN3(step)                                                                                                  \
  {                                                                                                                    \
    int p = k0 - 1;                                                                                                    \
    _Pragma("unroll 1") for (; p + 2 <= pend; p += 3) {                                                                \
      step(I0{}, p);                                                                                                   \
      step(I1{}, p + 1);                                                                                               \
      step(I2{}, p + 2);                                                                                               \
    }                                                                                                                  \
    if (p <= pend) {                                                                                                   \
      step(I0{}, p);                                                                                                   \
      if (p + 1 <= pend) step(I1{}, p + 1);                                                                            \
    }                                                                                                                  \
  }

  if (w < NRW) {
    // ------------------------------------------------------------------------- output rows
    const int rr = min(1 + 2 * w + rsub, nrows);  // row slot 1..nrows; rows past the tile repeat the last one
    const int le = min(col, llast);
    const int xs = le + 1;
    // wave-uniform bases at the tile's first row, per-lane byte offsets (row inside the tile, column)
    unsigned lo8 = (unsigned)((rr - 1) * P.nx + le) * 8u;
    unsigned so8 = (unsigned)((rr - 1) * O.nx + le) * 8u;
    const char* gp = (const char*)(P.p + P.idx(i0, j0, k0 - 2, pcomp));
    double pc = PA_LDG(gp, lo8), p0 = PA_LDG(gp + pps, lo8), p1 = PA_LDG(gp + 2 * pps, lo8);
    double cm = PA_PROG(pc), cc = PA_PROG(p0), cp = PA_PROG(p1);
    double fzc = zflux(dxinv[2], cm, cc);
    double f[3];
    f[0] = PA_LDG(gp + 3 * pps, lo8);
    f[1] = PA_LDG(gp + 4 * pps, lo8);
    gp += 4 * pps;
    gp += (k0 + 3 <= kfmax) ? pps : 0;
    f[2] = PA_LDG(gp, lo8);
    asm volatile("" ::"v"(f[0]), "v"(f[1]), "v"(f[2]));  // enter the loop with nothing in flight (see pa_fused_march3.h)
    double cgzv = 0.0;
    int pzh = -0x40000000;  // the step whose request becomes c of plane hi_z + 1
    if (CG && cgzl) {
      cc = cgzl[(long long)(j0 + rr - 1 - V.lo[1] + 1) * (nx + 2) + (i0 + le - V.lo[0] + 1)];
      fzc = zflux(dxinv[2], cm, cc);
    }
    if (CG && cgzh) {
      cgzv = cgzh[(long long)(j0 + rr - 1 - V.lo[1] + 1) * (nx + 2) + (i0 + le - V.lo[0] + 1)];
      pzh = V.hi[2] - 1;
    }
    S.c[0][rr][xs] = cc;
    __syncthreads();
    double nxq = 0, nyq = 0, nzq = 0, fzn = 0, fzp = 0;
    char* ob = (char*)(O.p + O.idx(i0, j0, k0, A.ocomp));
    const long long ops = (long long)O.nx * O.ny * 8, osc = O.sc * 8;
    char* ob2 = nullptr;  // GOUT: G's FAB (its own row length, plane and component strides)
    long long ops2 = 0, osc2 = 0;
    unsigned sg8 = 0;
    if (GOUT) {
      static_assert(!GOUT || CG, "GOUT: exact-normal sweep");
      const FabView G2 = mf_view(DMFView{A.gdata, A.goff, 3, A.gng, 0, 0.0, 0.0}, V, box);
      ob2 = (char*)(G2.p + G2.idx(i0, j0, k0, 0));
      ops2 = (long long)G2.nx * G2.ny * 8;
      osc2 = G2.sc * 8;
      sg8 = (unsigned)((rr - 1) * G2.nx + le) * 8u;
    }
    double gxq = 0, gyq = 0, gzq = 0;  // GOUT: G of plane q
    const double thr = A.thr;
    double o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0, o5 = 0, o6 = 0, o7 = 0;
    auto step = [&](auto spc, int p) __attribute__((always_inline)) {
      constexpr int SP = decltype(spc)::value, SP1 = (SP + 1) % 3, SQ = (SP + 2) % 3;
      double x;
      PA_TAKE(x, f[SP]);
      PA_OPAQUE(so8);
      __builtin_amdgcn_sched_barrier(0);
      gp += (p + 5 <= kfmax) ? pps : 0;
      PA_STG(ob, so8, o0); PA_STG(ob + osc, so8, o1); PA_STG(ob + 2 * osc, so8, o2); PA_STG(ob + 3 * osc, so8, o3);
      PA_STG(ob + 4 * osc, so8, o4);
      if (GOUT) {
        PA_OPAQUE(sg8);
        PA_STG(ob2, sg8, o5); PA_STG(ob2 + osc2, sg8, o6); PA_STG(ob2 + 2 * osc2, sg8, o7);
      } else {
        PA_STG(ob + 5 * osc, so8, o5); PA_STG(ob + 6 * osc, so8, o6); PA_STG(ob + 7 * osc, so8, o7);
      }
      __builtin_amdgcn_sched_barrier(0);
      const double cl = S.c[SP][rr][xs - 1], cr = S.c[SP][rr][xs + 1];
      const double cs = S.c[SP][rr - 1][xs], cn = S.c[SP][rr + 1][xs];
      const double ggx = cdiff(dxinv[0], cl, cc, cr);
      const double ggy = cdiff(dxinv[1], cs, cc, cn);
      const double fzh = zflux(dxinv[2], cc, cp);
      const double ggz = favg(fzc, fzh);
      const double sn = sqrt(ggx * ggx + ggy * ggy + ggz * ggz);
      const double ng = -((1e-14 < sn) ? sn : 1e-14);
      double nxp, nyp, nzp;
      div3_shared(ggx, ggy, ggz, ng, nxp, nyp, nzp);
      S.ny[SP][rr][col] = nyp;
      S.nx[SP][rr - 1][xs] = nxp;
      S.c[SP1][rr][xs] = cp;
      if (!GOUT) S.p[SP][rr][xs] = p0;
      __syncthreads();
      PA_OPAQUE(lo8);
      f[SP] = PA_LDG(gp, lo8);  // request for plane p+5, after the barrier
      const double nxl = S.nx[SQ][rr - 1][xs - 1], nxr = S.nx[SQ][rr - 1][xs + 1];
      const double nys = S.ny[SQ][rr - 1][col], nyn = S.ny[SQ][rr + 1][col];
      const double fznh = zflux(dxinv[2], nzq, nzp);
      double curv = 0.0;
      curv += cdiff(dxinv[0], nxl, nxq, nxr);
      curv += cdiff(dxinv[1], nys, nyq, nyn);
      curv += favg(fzn, fznh);
      curv = curv * 0.5;
      double gx = 0, gy = 0, gz = 0, gm = 0, fzph = 0;
      if (!GOUT) {
        const double pl = S.p[SQ][rr][xs - 1], pr = S.p[SQ][rr][xs + 1];
        const double ps = S.p[SQ][rr - 1][xs], pnn = S.p[SQ][rr + 1][xs];
        gx = cdiff(dxinv[0], pl, pc, pr);
        gy = cdiff(dxinv[1], ps, pc, pnn);
        fzph = zflux(dxinv[2], pc, p0);
        gz = favg(fzp, fzph);
        gm = sqrt(gx * gx + gy * gy + gz * gz);
      }
      ob += (p >= k0 + 2) ? ops : 0;
      if (GOUT) {  // [Progress K Nx Ny Nz] + G; cm = c at plane q; Progress and G are not clipped
        ob2 += (p >= k0 + 2) ? ops2 : 0;
        const bool clip = CLIP && ((cm < thr) || (cm > 1.0 - thr));
        o0 = cm;
        o1 = clip ? 0.0 : curv;
        o2 = clip ? 0.0 : nxq;
        o3 = clip ? 0.0 : nyq;
        o4 = clip ? 0.0 : nzq;
        o5 = gxq; o6 = gyq; o7 = gzq;
        gxq = ggx; gyq = ggy; gzq = ggz;
      } else {
      o0 = gx; o1 = gy; o2 = gz; o3 = gm;
      if (CLIP) {  // threshold clip (curvature.cpp:557-566); cm = c at plane q
        const bool clip = (cm < thr) || (cm > 1.0 - thr);
        o4 = clip ? 0.0 : nxq;
        o5 = clip ? 0.0 : nyq;
        o6 = clip ? 0.0 : nzq;
        o7 = clip ? 0.0 : curv;
      } else {
        o4 = nxq; o5 = nyq; o6 = nzq; o7 = curv;
      }
      }
      cm = cc; cc = cp; cp = PA_PROG(x);
      if (CG) cp = (p == pzh) ? cgzv : cp;  // x was phi of plane hi_z + 1
      fzc = fzh; fzn = fznh; fzp = fzph;
      pc = p0; p0 = p1; p1 = x;
      nxq = nxp; nyq = nyp; nzq = nzp;
    };
    PA_RUN3(step)
    PA_STG(ob, so8, o0); PA_STG(ob + osc, so8, o1); PA_STG(ob + 2 * osc, so8, o2); PA_STG(ob + 3 * osc, so8, o3);
    PA_STG(ob + 4 * osc, so8, o4);
    if (GOUT) {
      PA_STG(ob2, sg8, o5); PA_STG(ob2 + osc2, sg8, o6); PA_STG(ob2 + 2 * osc2, sg8, o7);
    } else {
      PA_STG(ob + 5 * osc, so8, o5); PA_STG(ob + 6 * osc, so8, o6); PA_STG(ob + 7 * osc, so8, o7);
    }
    return;
  }

  if (w == NRW) {
    // ------------------------------------------------------------------------- the two halo rows
    const int rr = rsub ? rtop : 0;                 // row slot
    const int j = rsub ? j0 + nrows : j0 - 1;       // its row, and the row beyond it (one y-neighbour comes from global)
    const int jout = rsub ? j + 1 : j - 1;
    const int rin = rsub ? rtop - 1 : 1;            // the tile row next to it
    const int le = min(col, llast);
    const int xs = le + 1;
    const char* gb = (const char*)(P.p + P.idx(i0, P.lo[1], k0 - 2, pcomp));  // wave-uniform: first row of the FAB
    unsigned lo8 = (unsigned)((j - P.lo[1]) * P.nx + le) * 8u;
    unsigned oo8 = (unsigned)((jout - P.lo[1]) * P.nx + le) * 8u;
    const char* gp = gb;
    const char* go = gb + pps;
    // CG: the second stream through a per-lane pointer (the two halves of the wave may sit on different kinds of face)
    const char* gol = gb + pps + oo8;
    long long pso = pps;
    int sh = 0;
#define PA_LDO(d) (CG ? PA_LDG(gol + (d) * pso, 0) : PA_LDG(go + (d) * pps, oo8))
    double p0 = PA_LDG(gp + pps, lo8), p1 = PA_LDG(gp + 2 * pps, lo8);
    double cm = PA_PROG(PA_LDG(gp, lo8)), cc = PA_PROG(p0), cp = PA_PROG(p1);
    const double* cgy = CG ? (rsub ? cgyh : cgyl) : nullptr;
    const int ymode = (CG && cgy) ? (rsub ? yhmode : 1) : 0;  // 1: this row is the ghost row; 2: the row beyond it is
    const bool ysp = ymode == 1;
    if (ymode) {
      const char* cb = (const char*)(cgy + (long long)(k0 - 1 - V.lo[2] + 1) * (nx + 2) + (i0 + le - V.lo[0] + 1));
      pso = (long long)(nx + 2) * 8;
      if (ysp) {
        sh = 1;
        cc = PA_LDG(cb, 0);  // c of plane k0-1
        gol = cb + pso;      // plane k0
      } else {
        gol = cb;            // c of the row beyond, same planes as the phi stream it replaces
      }
    }
    double co = PA_LDO(0);
    if (ysp) cp = co;
    else if (ymode == 0) co = PA_PROG(co);
    double fzc = zflux(dxinv[2], cm, cc);
    double f[3], fo[3];
    __builtin_amdgcn_sched_barrier(0);
    f[0] = PA_LDG(gp + 3 * pps, lo8);
    fo[0] = PA_LDO(1);
    __builtin_amdgcn_sched_barrier(0);
    f[1] = PA_LDG(gp + 4 * pps, lo8);
    fo[1] = PA_LDO(2);
    __builtin_amdgcn_sched_barrier(0);
    gp += 4 * pps;
    gp += (k0 + 3 <= kfmax) ? pps : 0;
    go += 2 * pps;
    go += (k0 + 2 <= pend) ? pps : 0;
    gol += 2 * pso;
    gol += (k0 + 2 + sh <= pend) ? pso : 0;
    f[2] = PA_LDG(gp, lo8);
    fo[2] = PA_LDO(0);
    __builtin_amdgcn_sched_barrier(0);
    S.c[0][rr][xs] = cc;
    __syncthreads();
    auto step = [&](auto spc, int p) __attribute__((always_inline)) {
      constexpr int SP = decltype(spc)::value, SP1 = (SP + 1) % 3;
      double x, xo;
      PA_TAKE(x, f[SP]);
      PA_TAKE(xo, fo[SP]);
      __builtin_amdgcn_sched_barrier(0);
      gp += (p + 5 <= kfmax) ? pps : 0;
      go += (p + 4 <= pend) ? pps : 0;
      if (CG) gol += (p + 4 + sh <= pend) ? pso : 0;
      const double cl = S.c[SP][rr][xs - 1], cr = S.c[SP][rr][xs + 1];
      const double cin = S.c[SP][rin][xs];
      const double cs = rsub ? cin : co, cn = rsub ? co : cin;
      const double ggx = cdiff(dxinv[0], cl, cc, cr);
      const double ggy = cdiff(dxinv[1], cs, cc, cn);
      const double fzh = zflux(dxinv[2], cc, cp);
      const double ggz = favg(fzc, fzh);
      const double sn = sqrt(ggx * ggx + ggy * ggy + ggz * ggz);
      const double ng = -((1e-14 < sn) ? sn : 1e-14);
      S.ny[SP][rr][col] = ggy / ng;
      S.c[SP1][rr][xs] = cp;
      if (!GOUT) S.p[SP][rr][xs] = p0;
      __syncthreads();
      PA_OPAQUE(lo8);
      if (!CG) PA_OPAQUE(oo8);
      f[SP] = PA_LDG(gp, lo8);
      fo[SP] = PA_LDO(0);
      cm = cc; cc = cp; cp = ysp ? xo : PA_PROG(x); co = (ymode == 2) ? xo : PA_PROG(xo);
      fzc = fzh;
      p0 = p1; p1 = x;
    };
    PA_RUN3(step)
#undef PA_LDO
    return;
  }

  // ----------------------------------------------------------------------------- edge wave
  {
    const int l20 = lane % (2 * MROWS);  // idle lanes mirror the active ones
    const int rr = min(l20 >> 1, rtop);
    const int side = l20 & 1;
    const int j = j0 + rr - 1;
    const int i = side ? iR : i0 - 1;
    const int xs = side ? llast + 2 : 0;
    const int xin = side ? llast + 1 : 1;
    const int rlo = max(rr - 1, 0), rhi = min(rr + 1, rtop);
    const bool has_n = (rr >= 1 && rr <= nrows);
    const char* gb = (const char*)(P.p + P.idx(P.lo[0], P.lo[1], k0 - 2, pcomp));
    unsigned og = (unsigned)((j - P.lo[1]) * P.nx + (i - P.lo[0])) * 8u;
    unsigned oo = (unsigned)((j - P.lo[1]) * P.nx + ((side ? i + 1 : i - 1) - P.lo[0])) * 8u;
    const char* gp = gb;
    const char* go = gb + pps;
    const char* gol = gb + pps + oo;  // CG: per-lane pointer of the second stream (see the halo wave)
    long long pso = pps;
    int sh = 0;
#define PA_LDO(d) (CG ? PA_LDG(gol + (d) * pso, 0) : PA_LDG(go + (d) * pps, oo))
    double p0 = PA_LDG(gp + pps, og), p1 = PA_LDG(gp + 2 * pps, og);
    double cm = PA_PROG(PA_LDG(gp, og)), cc = PA_PROG(p0), cp = PA_PROG(p1);
    const double* cgx = CG ? (side ? cgxh : cgxl) : nullptr;
    const int xmode = (CG && cgx) ? (side ? xhmode : 1) : 0;  // 1: this column is the ghost column; 2: the column beyond it is
    const bool xsp = xmode == 1;
    if (xmode) {
      const char* cb = (const char*)(cgx + (long long)(k0 - 1 - V.lo[2] + 1) * (ny + 2) + (j - V.lo[1] + 1));
      pso = (long long)(ny + 2) * 8;
      if (xsp) {
        sh = 1;
        cc = PA_LDG(cb, 0);  // c of plane k0-1
        gol = cb + pso;      // plane k0
      } else {
        gol = cb;
      }
    }
    double co = PA_LDO(0);
    if (xsp) cp = co;
    else if (xmode == 0) co = PA_PROG(co);
    double f[3], fo[3];
    __builtin_amdgcn_sched_barrier(0);
    f[0] = PA_LDG(gp + 3 * pps, og);
    fo[0] = PA_LDO(1);
    __builtin_amdgcn_sched_barrier(0);
    f[1] = PA_LDG(gp + 4 * pps, og);
    fo[1] = PA_LDO(2);
    __builtin_amdgcn_sched_barrier(0);
    gp += 4 * pps;
    gp += (k0 + 3 <= kfmax) ? pps : 0;
    go += 2 * pps;
    go += (k0 + 2 <= pend) ? pps : 0;
    gol += 2 * pso;
    gol += (k0 + 2 + sh <= pend) ? pso : 0;
    f[2] = PA_LDG(gp, og);
    fo[2] = PA_LDO(0);
    __builtin_amdgcn_sched_barrier(0);
    double fzc = zflux(dxinv[2], cm, cc);
    S.c[0][rr][xs] = cc;
    __syncthreads();
    auto step = [&](auto spc, int p) __attribute__((always_inline)) {
      constexpr int SP = decltype(spc)::value, SP1 = (SP + 1) % 3;
      double x, xo;
      PA_TAKE(x, f[SP]);
      PA_TAKE(xo, fo[SP]);
      __builtin_amdgcn_sched_barrier(0);
      gp += (p + 5 <= kfmax) ? pps : 0;
      go += (p + 4 <= pend) ? pps : 0;
      if (CG) gol += (p + 4 + sh <= pend) ? pso : 0;
      const double inner = S.c[SP][rr][xin];
      const double cl = side ? inner : co, cr = side ? co : inner;
      const double cs = S.c[SP][rlo][xs], cn = S.c[SP][rhi][xs];
      const double ggx = cdiff(dxinv[0], cl, cc, cr);
      const double ggy = cdiff(dxinv[1], cs, cc, cn);
      const double fzh = zflux(dxinv[2], cc, cp);
      const double ggz = favg(fzc, fzh);
      const double sn = sqrt(ggx * ggx + ggy * ggy + ggz * ggz);
      const double ng = -((1e-14 < sn) ? sn : 1e-14);
      const double nxp = ggx / ng;
      if (has_n) S.nx[SP][rr - 1][xs] = nxp;
      S.c[SP1][rr][xs] = cp;
      if (!GOUT) S.p[SP][rr][xs] = p0;
      __syncthreads();
      PA_OPAQUE(og);
      if (!CG) PA_OPAQUE(oo);
      f[SP] = PA_LDG(gp, og);
      fo[SP] = PA_LDO(0);
      cm = cc; cc = cp; cp = xsp ? xo : PA_PROG(x); co = (xmode == 2) ? xo : PA_PROG(xo);
      fzc = fzh;
      p0 = p1; p1 = x;
    };
    PA_RUN3(step)
#undef PA_LDO
  }
#undef PA_PROG
#undef PA_RUN3
}

template <typename BP, int NRW, bool CLIP, bool CG = false>
__global__ __launch_bounds__(64 * (NRW + 2), 1) void k_gradcurv_march3n(BP bp, MarchArgs A) {
  gradcurv_march3n_body<BP, NRW, CLIP, CG>(bp, A, blockIdx.x, blockIdx.y);
}

// the CG sweeps of the narrow-box groups of several levels in ONE launch (as k_gradcurv_march3_levels for the wide boxes: no idle
// tail and ramp-up between the levels)
template <int NRW, bool CLIP = false, bool GOUT = false>
__global__ __launch_bounds__(64 * (NRW + 2), 1) void k_gradcurv_march3n_levels(SweepBatch S) {
  int l = 0;
  while (l + 1 < S.n && blockIdx.x >= S.wg0[l + 1]) ++l;
  if (GOUT || (gridDim.y == 1 && !S.prog)) {
    gradcurv_march3n_body<LevelBP2, NRW, CLIP, true, GOUT>(S.bp[l], S.A[l], blockIdx.x - S.wg0[l], 0u);
    return;
  }
  LevelBP2 bp;
  MarchArgs A;
  sweep_slot(S, l, bp, A);
  gradcurv_march3n_body<LevelBP2, NRW, CLIP, true, GOUT>(bp, A, blockIdx.x - S.wg0[l], 0u);
}
