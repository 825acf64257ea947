// pa_fused_march3.h -- fused grad->curvature, k-marching kernel with a 3-plane-deep prefetch.
//
// Same tiling, roles, LDS ring and arithmetic as k_gradcurv_march (pa_fused_march.h; read that
// header first), restructured around what its ISA showed: there every wave consumed, at the end of
// a plane, the load it had issued at the top of the SAME plane (`s_waitcnt vmcnt(8)`), and because
// vmcnt retires in issue order that wait also covered the previous plane's 8 stores.  One memory
// round trip (and one store acknowledge) therefore sat on the barrier-to-barrier critical path of
// every plane, in all three wave roles.  Here
//   * SCHEDULE OF A STEP (the largest single effect, measured 2.27 -> 1.92 ms per launch): all 8 stores of the
//     previous plane go out in ONE burst at the top of the step, and every role issues its request for the
//     plane three steps ahead right AFTER the plane's barrier -- loads and stores never sit next to each other
//     in a wave's instruction stream.  With the requests at the top of the step, directly in front of the
//     stores (the first v3 schedule, kept as DBG 256 for A/B), the same kernel takes 2.27 ms; moving only the
//     output rows' request behind the barrier gives 2.02-2.10, all roles 1.97, plus the burst 1.92.
//   * every role keeps THREE planes in flight (f0,f1,f2): the plane consumed in a step was
//     requested three steps earlier, so a step waits only for stores that are >= 4 planes old
//     (`vmcnt(26)`); the loop is unrolled by 3 = the LDS ring period, which also makes every ring
//     slot a compile-time LDS offset and keeps the in-flight registers free of moves;
//   * global addresses are a wave-uniform base (SGPR pair, advanced with scalar adds) plus one
//     loop-invariant 32-bit lane offset: no per-store 64-bit vector address arithmetic;
//   * the z-direction face fluxes of c, phi and n_z are carried from plane to plane (the high face
//     of plane k is the low face of plane k+1: same operation on the same operands);
//   * the three components of the flame normal share one reciprocal (div3_shared): the exact
//     instruction sequence hipcc emits for an IEEE fp64 division (v_rcp_f64, two Newton steps,
//     quotient, remainder, correction) with its v_div_scale / v_div_fixup steps dropped under a
//     wave-uniform guard that proves them to be no-ops; otherwise the plain `/` is taken.  Results
//     are bit-identical to `/` for every input;
//   * the threshold clip is a template parameter.
// Results are bit-identical to k_gradcurv_march and to the CPU oracle.
#pragma once
#include "pa_dpp.h"
#include <type_traits>
#include "pa_fused_march.h"

// a/d, b/d, c/d, each bit-identical to the IEEE-correct `/` of hipcc.
// hipcc lowers x/y (fp64) to: ys = div_scale(y), xs = div_scale(x), r0 = rcp(ys), two Newton
// steps r <- fma(r, fma(-ys, r, 1), r), q = xs*r, rem = fma(-ys, q, xs), div_fmas(rem, r, q),
// div_fixup.  V_DIV_SCALE_F64 returns its operand unchanged (and VCC = 0, so div_fmas is a plain
// fma) unless: x or y is zero, y is denormal, 1/y is denormal, x/y is denormal or near overflow
// (exponent difference >= 768), or the biased exponent of x is <= 53; V_DIV_FIXUP_F64 passes the
// quotient through unless an operand is zero/inf/nan or the quotient over/underflows.  The guard
// below (2^-100 <= |d| <= 2^100, each numerator zero or of binary exponent >= -900; |numerator|
// <= |d|(1+eps) by construction of d = -max(1e-14, |G|)) excludes all of those, and a zero
// numerator goes through the unscaled sequence to the correctly signed zero.
__device__ __forceinline__ void div3_shared(double a, double b, double c, double d, double& qa, double& qb, double& qc) {
  const double ad = __builtin_fabs(d);
  const int ea = __builtin_amdgcn_frexp_exp(a), eb = __builtin_amdgcn_frexp_exp(b), ec = __builtin_amdgcn_frexp_exp(c);
  const int emin = min(ea, min(eb, ec));
  const bool plain = (ad <= 0x1p100) && (ad >= 0x1p-100) && (emin >= -900);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!plain) == 0ull, 1)) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double ma = a * r, mb = b * r, mc = c * r;
    const double ra = __builtin_fma(-d, ma, a), rb = __builtin_fma(-d, mb, b), rc = __builtin_fma(-d, mc, c);
    qa = __builtin_fma(ra, r, ma);
    qb = __builtin_fma(rb, r, mb);
    qc = __builtin_fma(rc, r, mc);
  } else {
    qa = a / d;
    qb = b / d;
    qc = c / d;
  }
}

// central difference with the low-face flux carried in (cdiff of pa_internal.h split in two)
__device__ __forceinline__ double zflux(double dxinv, double lo, double hi) { return -(dxinv * (hi - lo)); }
__device__ __forceinline__ double favg(double fl, double fh) { return -(0.5 * (fl + fh)); }

// PA_TAKE: move a value that was in flight into a fresh register HERE (the wait for it lands on this
// instruction), so that the register it arrived in can take the next request: the in-flight
// registers then never rotate, and no move of a still-in-flight value ends up at the loop's back edge.
// PA_OPAQUE: keeps the zero-extension of a lane offset inside the loop body, where instruction
// selection can fold it into the `global_* v_off, v_data, s[base]` addressing form.
#define PA_TAKE(dst, src) asm volatile("v_mov_b64 %0, %1" : "=v"(dst) : "v"(src))
#define PA_OPAQUE(x) asm volatile("" : "+v"(x))
#define PA_LDG(base, off) (*(const double*)((const char*)(base) + (off)))
#define PA_STG(base, off, v) (*(double*)((char*)(base) + (off)) = (v))
#define PA_STNT(base, off, v) __builtin_nontemporal_store((v), (double*)((char*)(base) + (off)))
#define PA_STL(base, off, v) do { if (!(DBG & 1) || (v) == 1.2345e-300) { if (DBG & 2048) PA_STNT(base, off, v); else PA_STG(base, off, v); } } while (0)

// DBG (diagnostic builds only, selected with PA_DBG): 256 = the first v3 schedule (requests at the top of a
// step, stores spread over it; results correct); wrong results on purpose: 1 = no global stores in the
// loop (a never-true data-dependent condition keeps the arithmetic alive), 2 = sqrt and divisions
// replaced by additions, 4 = no x/y-neighbour reads from LDS (own values instead); correct results: 2048 = the output
// stores carry the `nt` (non-temporal) bit: measured 2.00 against 1.92 ms per launch, so they do not.
// PAIR: 16-byte stores.  A CU issues `global_store_dwordx2` at only ~7 B/cycle (measured: with 8-B
// stores the sweep left the L2->HBM write interface idle -- TCC_EA0_WRREQ_STALL 0.5 M cycles against
// 41 M for a no-arithmetic emulation of the same pattern -- while its time did not move with the
// arithmetic, the LDS traffic or the prefetch depth).  Lanes 2m and 2m+1 therefore swap one value
// (DPP quad_perm [1,0,3,2]) so that the even lane holds component A of cells 2m, 2m+1 and the odd
// lane component B of the same two cells: one `global_store_dwordx4` writes 512 B of A and 512 B
// of B, four store instructions per plane instead of eight.  Needs full 64-wide tiles, an even row
// length and an even first column in the output FAB (checked on the host, else PAIR = false).
__device__ __forceinline__ double swap_lane_pair(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
typedef double pa_d2 __attribute__((ext_vector_type(2)));
// lane parity `odd`; `off` = even lane: 8*lane, odd lane: 8*(lane-1) + component stride
__device__ __forceinline__ void store_pair(char* base, unsigned off, bool odd, double A, double B) {
  const double recv = swap_lane_pair(odd ? A : B);
  pa_d2 v;
  v.x = odd ? recv : A;
  v.y = odd ? B : recv;
  *(pa_d2*)(base + off) = v;
}

// CG: the progress variable in the ghost cells behind SPECIAL box faces (coarse-fine / wall) is not (phi_ghost - pmin) *
// invdenom but the reference's boundary condition applied to c itself; with CG the kernel takes those values from the
// level's compact face-major arrays (DLevelView::cg, filled by k_prep_faces in pa_fused2.hip), so the flame normal is
// exact in EVERY valid cell and the curvature everywhere except in the first layer behind such a face (whose ghost
// normal comes from the boundary condition on n: pa_fused2.hip).  Where the arrays are read:
//   z faces  output rows: c of plane lo_z - 1 (prologue) / hi_z + 1 (one select per step);
//   y faces  the halo row wave IS the ghost row: its second stream (the row beyond, only needed for the ghost normal,
//            which is irrelevant there) is re-aimed at the array, one plane ahead;
//   x faces  the edge wave's lanes of that side: likewise its stream of the column beyond.
// Values that only feed ghost normals behind special faces (edge ghosts, second ghost layer) may be anything.
// GOUT (pa_curvature_run with options; CG only): no gradient of phi -- the 8 stores of a plane are Progress, K, N (out components
// ocomp .. ocomp + 4) and G = the cell-centred gradient of c, the normal before its normalisation (curvature.cpp:457-490),
// into components 0 .. 2 of a second multifab (MarchArgs::gdata) that do_gaussCurv differentiates again.
// GOUT == 2 (round 6, "KG"): the sweep ALSO forms the Gaussian curvature (curvature.cpp:575-677: Hessian rows = grad(G_d), adjugate,
// Kg = G^T adj(H) G / normgrad^4) and stores it at out component ocomp + 5 -- G of a plane goes through three more LDS rings for its
// x / y neighbours (rows, halo rows and edge columns all form G anyway), its z-neighbours are the planes before and after in
// registers; the operations and their order are k_gauss_curv's (pa_curvopts.hip), the values the ones it would load, so the bits are
// the same in every cell whose six neighbours' G this FAB sees as the level does: every cell but the first layer behind a special face
// (ghost G = the boundary condition on G, not what this sweep forms from ghost c) and the level's irregular cells -- those are
// recomputed from the stored G afterwards (k_gauss_cells).  The separate Gaussian-curvature pass over all cells (1.43 ms per 512^3 level)
// is gone; the sweep stores nine values per cell and plane instead of eight.
template <int MTY>
struct MarchLdsG {
  double gy[3][MTY + 2][PA_MLW], gz[3][MTY + 2][PA_MLW];  // (G_x rides in MarchLds::p, which a GOUT sweep does not use)
};
template <typename BP, int PA_MTY, bool CLIP, bool PAIR, int DBG, bool CG, int GOUT = 0>
__device__ __forceinline__ void gradcurv_march3_body(const BP& bp, const MarchArgs& A, const unsigned bid_x, const unsigned bid_y) {
  static_assert(!GOUT || (CG && !PAIR && DBG == 0), "GOUT: exact-normal sweep, 8-byte stores");
  constexpr bool KG = GOUT == 2;
  FabView P, O;
  DBox V;
  double dxinv[3];
  constexpr int PA_MROWS = PA_MTY + 2;
  constexpr bool OLD_SCHED = (DBG & 256) != 0;  // requests at the top of a step + stores spread over it (first v3 schedule)
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
    box = A.order ? (int)((bid_x / (unsigned)A.txy_max) % (unsigned)A.nboxes) : (int)bid_y;
  }
  if (A.boxlist && !A.wgtab) {
    if (box >= A.nboxes) return;
    box = A.boxlist[box];
  }
  if (!bp.get(box, P, O, V, dxinv)) return;
  int gfl = 1;  // KG: where this tile stores G (pa_sweep_gneed; bit 0: everywhere)
  if constexpr (KG) {
    if (A.gneed && A.wgtab) gfl = A.gneed[bid_x];
  }
  const int pcomp = A.pcomp, kseg = A.kseg;
  const double pmin = A.pmin, invd = A.invdenom;
  const int nx = V.hi[0] - V.lo[0] + 1, ny = V.hi[1] - V.lo[1] + 1, nz = V.hi[2] - V.lo[2] + 1;
  const int tx = (nx + 63) / 64, ty = (ny + PA_MTY - 1) / PA_MTY, tz = (nz + kseg - 1) / kseg;
  if (A.order == 1) {
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
  const int rtop = min(PA_MROWS - 1, V.hi[1] + 1 - j0 + 1);  // row slot of the last live row
  // compact resolved-ghost arrays of the special faces this tile touches (null: ordinary face / tile not at that face)
  const double *cgxl = nullptr, *cgxh = nullptr, *cgyl = nullptr, *cgyh = nullptr, *cgzl = nullptr, *cgzh = nullptr;
  // On the HIGH sides a tile / segment may also end one short of the box (widths 64 t + 1, MTY t + 1 rows, a last
  // segment of one plane): its outermost neighbour row / column / plane is then the last VALID one and the "row beyond"
  // it is the ghost row -- which must come from the array as well (mode 2 below; mode 1 = the neighbour itself is the ghost).
  int xhmode = 0, yhmode = 0;
  if (CG) {
    if (bx == 0) cgxl = bp.cg_face(box, 0);
    if (iR >= V.hi[0]) { cgxh = bp.cg_face(box, 1); xhmode = cgxh ? (iR == V.hi[0] + 1 ? 1 : 2) : 0; }
    if (by == 0) cgyl = bp.cg_face(box, 2);
    if (j0 + rtop - 1 >= V.hi[1]) { cgyh = bp.cg_face(box, 3); yhmode = cgyh ? (j0 + rtop - 1 == V.hi[1] + 1 ? 1 : 2) : 0; }
    if (k0 == V.lo[2]) cgzl = bp.cg_face(box, 4);
    if (k1 >= V.hi[2] - 1) cgzh = bp.cg_face(box, 5);
  }
  // NCG (MarchArgs::ncg): the special x faces of this tile whose first-layer data the sweep mirrors -- the tile must hold the first three
  // columns behind the face (the same rule in k_faces_curv_fast: ncg_face_ok)
  const bool ncgl = CG && !CLIP && A.ncg && cgxl && llast >= 2;
  const bool ncgh = CG && !CLIP && A.ncg && cgxh && xhmode == 1 && llast >= 2;
  const bool ncg_tile = ncgl || ncgh;

  __shared__ MarchLds<PA_MTY> S;
  __shared__ typename std::conditional<KG, MarchLdsG<PA_MTY>, char>::type SG;
  (void)SG;
#define PA_SGX S.p
  const long long pps = (long long)P.nx * P.ny * 8;  // plane stride of phi, bytes
  const int pend = k1 + 1;                            // normals are formed on planes k0-1 .. k1+1
  const int kfmax = k1 + 2;                           // last plane of phi that exists for this segment
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

  if (w < PA_MROWS && w > rtop) {
    // ---------------------------------------------------------------- dead row (partial tile)
    for (int it = 0; it <= pend - (k0 - 1) + 1; ++it) __syncthreads();
    if (ncg_tile) __syncthreads();
    return;
  }

  if (w < PA_MROWS) {
    // ------------------------------------------------------------------------- row waves
    const int rr = w;
    const int j = j0 + rr - 1;
    const int le = min(lane, llast);  // lanes past the box edge mirror the last valid lane
    const int xs = le + 1;
    unsigned lo8 = (unsigned)le * 8u;
    const char* gp = (const char*)(P.p + P.idx(i0, j, k0 - 2, pcomp));  // wave-uniform
    double pc = PA_LDG(gp, lo8), p0 = PA_LDG(gp + pps, lo8), p1 = PA_LDG(gp + 2 * pps, lo8);
    const bool halo = (rr == 0) || (rr == rtop);  // supplies neighbours only; one y-neighbour comes from global
    double cm = PA_PROG(pc), cc = PA_PROG(p0), cp = PA_PROG(p1);  // c at planes p-1, p, p+1  (p = k0-1)
    double fzc = zflux(dxinv[2], cm, cc);                         // low z-face flux of c at plane p
    double f[3];
    if (halo) {
      // requests in the order of the steady state (f, fo per plane): the wait counts the compiler
      // derives for the loop are the minimum over the entry edge and the back edge
      const int jout = (rr == 0) ? j - 1 : j + 1;
      const char* go = (const char*)(P.p + P.idx(i0, jout, k0 - 1, pcomp));
      // CG: this row is the ghost row behind a special y face: the second stream reads the face's compact array
      // [plane][x], one plane AHEAD (it supplies c of plane p+2 at step p instead of c_out of plane p+1)
      const double* cgy = CG ? ((rr == 0) ? cgyl : cgyh) : nullptr;
      const int ymode = (CG && cgy) ? ((rr == 0) ? 1 : yhmode) : 0;  // 1: this row is the ghost row; 2: the row beyond it is
      const bool ysp = ymode == 1;
      const long long pso = ymode ? (long long)(nx + 2) * 8 : pps;  // plane stride of the second stream
      const int sh = ysp ? 1 : 0;
      if (ymode) {
        const char* cb = (const char*)(cgy + (long long)(k0 - 1 - V.lo[2] + 1) * (nx + 2) + (i0 - V.lo[0] + 1));
        if (ysp) {
          cc = PA_LDG(cb, lo8);  // c of plane k0-1
          go = cb + pso;         // plane k0
        } else {
          go = cb;               // c of the row beyond, same planes as the phi stream it replaces
        }
      }
      double co = PA_LDG(go, lo8);
      if (ysp) cp = co;        // c of plane k0
      else if (ymode == 0) co = PA_PROG(co);
      double fo[3];
      __builtin_amdgcn_sched_barrier(0);
      f[0] = PA_LDG(gp + 3 * pps, lo8);
      fo[0] = PA_LDG(go + pso, lo8);
      __builtin_amdgcn_sched_barrier(0);
      f[1] = PA_LDG(gp + 4 * pps, lo8);
      fo[1] = PA_LDG(go + 2 * pso, lo8);
      __builtin_amdgcn_sched_barrier(0);
      gp += 4 * pps;
      gp += (k0 + 3 <= kfmax) ? pps : 0;
      go += 2 * pso;
      go += (k0 + 2 + sh <= pend) ? pso : 0;
      f[2] = PA_LDG(gp, lo8);   // gp -> plane min(k0+3, k1+2), the youngest plane requested
      fo[2] = PA_LDG(go, lo8);  // go -> plane min(k0+2, k1+1)
      __builtin_amdgcn_sched_barrier(0);
      S.c[0][rr][xs] = cc;
      __syncthreads();
      auto step = [&](auto spc, int p) __attribute__((always_inline)) {
        constexpr int SP = decltype(spc)::value, SP1 = (SP + 1) % 3;
        double x, xo;  // phi(min(p+2, k1+2)), phi_out(min(p+1, k1+1))
        PA_TAKE(x, f[SP]);
        PA_TAKE(xo, fo[SP]);
        PA_OPAQUE(lo8);
        __builtin_amdgcn_sched_barrier(0);
        gp += (p + 5 <= kfmax) ? pps : 0;
        go += (p + 4 + sh <= pend) ? pso : 0;
        if (OLD_SCHED) {
          f[SP] = PA_LDG(gp, lo8);
          fo[SP] = PA_LDG(go, lo8);
        }
        const double cl = S.c[SP][rr][xs - 1], cr = S.c[SP][rr][xs + 1];
        const double cin = S.c[SP][(rr == 0) ? 1 : rr - 1][xs];
        const double cs = (rr == 0) ? co : cin, cn = (rr == 0) ? cin : co;
        const double ggx = cdiff(dxinv[0], cl, cc, cr);
        const double ggy = cdiff(dxinv[1], cs, cc, cn);
        const double fzh = zflux(dxinv[2], cc, cp);
        const double ggz = favg(fzc, fzh);
        const double sn = sqrt(ggx * ggx + ggy * ggy + ggz * ggz);
        const double ng = -((1e-14 < sn) ? sn : 1e-14);
        S.ny[SP][rr][lane] = ggy / ng;
        S.c[SP1][rr][xs] = cp;
        if (!GOUT) S.p[SP][rr][xs] = p0;
        if constexpr (KG) { PA_SGX[SP][rr][xs] = ggx; SG.gy[SP][rr][xs] = ggy; SG.gz[SP][rr][xs] = ggz; }
        __syncthreads();
        if (!OLD_SCHED) {
          PA_OPAQUE(lo8);
          f[SP] = PA_LDG(gp, lo8);
          fo[SP] = PA_LDG(go, lo8);
        }
        cm = cc; cc = cp; cp = ysp ? xo : PA_PROG(x); co = (ymode == 2) ? xo : PA_PROG(xo);
        fzc = fzh;
        p0 = p1; p1 = x;
      };
      PA_RUN3(step)
      if (ncg_tile) __syncthreads();
      return;
    }
    // output rows
    f[0] = PA_LDG(gp + 3 * pps, lo8);
    f[1] = PA_LDG(gp + 4 * pps, lo8);
    gp += 4 * pps;
    gp += (k0 + 3 <= kfmax) ? pps : 0;
    f[2] = PA_LDG(gp, lo8);  // gp -> plane min(k0+3, k1+2), the youngest plane requested
    // Enter the loop with nothing in flight: one `s_waitcnt vmcnt(N)` must hold for the first trip and
    // for the steady state, so requests still pending on entry (no stores behind them yet) would pull
    // N down from 26 to 2 and drain the stores every third plane.  Costs one round trip per segment.
    asm volatile("" ::"v"(f[0]), "v"(f[1]), "v"(f[2]));
    // CG: c of the ghost plane behind a special z face from the face's compact array [y][x]
    double cgzv = 0.0;
    int pzh = -0x40000000;  // the step whose request becomes c of plane hi_z + 1
    if (CG && cgzl) {
      cc = PA_LDG((const char*)(cgzl + (long long)(j - V.lo[1] + 1) * (nx + 2) + (i0 - V.lo[0] + 1)), lo8);
      fzc = zflux(dxinv[2], cm, cc);
    }
    if (CG && cgzh) {
      cgzv = PA_LDG((const char*)(cgzh + (long long)(j - V.lo[1] + 1) * (nx + 2) + (i0 - V.lo[0] + 1)), lo8);
      pzh = V.hi[2] - 1;  // step p requests phi of plane p + 2; plane hi_z + 1 is read when k1 >= hi_z - 1
    }
    S.c[0][rr][xs] = cc;
    __syncthreads();
    double nxq = 0, nyq = 0, nzq = 0, fzn = 0, fzp = 0;
    char* ob = (char*)(O.p + O.idx(i0, j, k0, A.ocomp));  // wave-uniform
    const long long ops = (long long)O.nx * O.ny * 8, osc = O.sc * 8;
    char* ob2 = nullptr;  // GOUT: the row of G's FAB (same lane offset, its own plane / component strides)
    long long ops2 = 0, osc2 = 0;
    if (GOUT) {
      const FabView G2 = mf_view(DMFView{A.gdata, A.goff, 3, A.gng, 0, 0.0, 0.0}, V, box);
      ob2 = (char*)(G2.p + G2.idx(i0, j, k0, 0));
      ops2 = (long long)G2.nx * G2.ny * 8;
      osc2 = G2.sc * 8;
    }
    double gxq = 0, gyq = 0, gzq = 0;  // GOUT: G of plane q
    double gxm = 0, gym = 0, gzm = 0, gnq = 1.0, o8 = 0;  // KG: G of plane q - 1, max(1e-14, |G|) of plane q, the Gaussian curvature of plane q
    // KG: G goes to memory only where something reads it afterwards -- the first three layers behind the box faces (the boundary
    // condition on G, its FillBoundary, the fix-up of the first layer: k_gauss_cells; behind x faces only where pa_sweep_gneed says so:
    // three lanes of a row are three partial lines) and the tiles a coarse patch of the finer level gathers from (bit 0)
    bool gsb = true, gs = !KG;
    if constexpr (KG) {
      const int i = i0 + le;
      gsb = (gfl & 1) || ((gfl & 2) && i - V.lo[0] < 3) || ((gfl & 4) && V.hi[0] - i < 3) || (j - V.lo[1] < 3) || (V.hi[1] - j < 3);
    }
    const double thr = A.thr;
    const bool odd = lane & 1;
    unsigned lo16 = odd ? (unsigned)(le - 1) * 8u + (unsigned)osc : (unsigned)le * 8u;
    // normal at plane p, outputs at plane q = p-1.  The 8 results of a plane are kept in registers
    // and stored at the top of the next step; the first three steps have nothing valid to store yet
    // and write to plane k0, which the same thread overwrites in program order.
    double o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0, o5 = 0, o6 = 0, o7 = 0;
    auto step = [&](auto spc, int p) __attribute__((always_inline)) {
      constexpr int SP = decltype(spc)::value, SP1 = (SP + 1) % 3, SQ = (SP + 2) % 3;
      double x;  // phi(min(p+2, k1+2)), requested three steps ago
      PA_TAKE(x, f[SP]);
      PA_OPAQUE(lo8);
      __builtin_amdgcn_sched_barrier(0);
      gp += (p + 5 <= kfmax) ? pps : 0;
      if (OLD_SCHED) f[SP] = PA_LDG(gp, lo8);
      if (!OLD_SCHED && !(DBG & 512)) {  // the 8 results of the previous plane, one burst; the request for plane p+5 follows the barrier
        if (PAIR) {
          PA_OPAQUE(lo16);
          store_pair(ob, lo16, odd, o0, o1); store_pair(ob + 2 * osc, lo16, odd, o2, o3);
          store_pair(ob + 4 * osc, lo16, odd, o4, o5); store_pair(ob + 6 * osc, lo16, odd, o6, o7);
        } else if (GOUT) {
          PA_STL(ob, lo8, o0); PA_STL(ob + osc, lo8, o1); PA_STL(ob + 2 * osc, lo8, o2); PA_STL(ob + 3 * osc, lo8, o3);
          PA_STL(ob + 4 * osc, lo8, o4);
          if constexpr (KG) PA_STL(ob + 5 * osc, lo8, o8);
          if (gs) { PA_STL(ob2, lo8, o5); PA_STL(ob2 + osc2, lo8, o6); PA_STL(ob2 + 2 * osc2, lo8, o7); }
        } else {
          PA_STL(ob, lo8, o0); PA_STL(ob + osc, lo8, o1); PA_STL(ob + 2 * osc, lo8, o2); PA_STL(ob + 3 * osc, lo8, o3);
          PA_STL(ob + 4 * osc, lo8, o4); PA_STL(ob + 5 * osc, lo8, o5); PA_STL(ob + 6 * osc, lo8, o6); PA_STL(ob + 7 * osc, lo8, o7);
        }
        __builtin_amdgcn_sched_barrier(0);  // the burst stays ahead of the plane's arithmetic
      }
      const double cl = (DBG & 4) ? cm : S.c[SP][rr][xs - 1], cr = (DBG & 4) ? cp : S.c[SP][rr][xs + 1];
      const double cs = (DBG & 4) ? cm : S.c[SP][rr - 1][xs], cn = (DBG & 4) ? cp : S.c[SP][rr + 1][xs];
      if (OLD_SCHED) { if (PAIR) { PA_OPAQUE(lo16); store_pair(ob, lo16, odd, o0, o1); } else PA_STL(ob, lo8, o0); }
      if (OLD_SCHED) __builtin_amdgcn_sched_barrier(0);
      const double ggx = cdiff(dxinv[0], cl, cc, cr);
      const double ggy = cdiff(dxinv[1], cs, cc, cn);
      const double fzh = zflux(dxinv[2], cc, cp);
      const double ggz = favg(fzc, fzh);
      if (OLD_SCHED) __builtin_amdgcn_sched_barrier(0);
      if (OLD_SCHED && !PAIR) PA_STL(ob + osc, lo8, o1);
      if (OLD_SCHED) __builtin_amdgcn_sched_barrier(0);
      const double sn = (DBG & 2) ? (ggx * ggx + ggy * ggy + ggz * ggz) : sqrt(ggx * ggx + ggy * ggy + ggz * ggz);
      const double ng = -((1e-14 < sn) ? sn : 1e-14);
      if (OLD_SCHED) __builtin_amdgcn_sched_barrier(0);
      if (OLD_SCHED) { if (PAIR) { PA_OPAQUE(lo16); store_pair(ob + 2 * osc, lo16, odd, o2, o3); } else PA_STL(ob + 2 * osc, lo8, o2); }
      if (OLD_SCHED) __builtin_amdgcn_sched_barrier(0);
      double nxp, nyp, nzp;
      if (DBG & 2) { nxp = ggx + ng; nyp = ggy + ng; nzp = ggz + ng; }
      else div3_shared(ggx, ggy, ggz, ng, nxp, nyp, nzp);
      PA_OPAQUE(lo8);  // new basic block after the guard's branch: see PA_OPAQUE
      if (OLD_SCHED) __builtin_amdgcn_sched_barrier(0);
      if (OLD_SCHED && !PAIR) PA_STL(ob + 3 * osc, lo8, o3);
      if (OLD_SCHED) __builtin_amdgcn_sched_barrier(0);
      S.ny[SP][rr][lane] = nyp;
      S.nx[SP][rr - 1][xs] = nxp;
      S.c[SP1][rr][xs] = cp;
      if (!GOUT) S.p[SP][rr][xs] = p0;
      if constexpr (KG) { PA_SGX[SP][rr][xs] = ggx; SG.gy[SP][rr][xs] = ggy; SG.gz[SP][rr][xs] = ggz; }
      if (OLD_SCHED) __builtin_amdgcn_sched_barrier(0);
      if (OLD_SCHED && !PAIR) PA_STL(ob + 4 * osc, lo8, o4);
      if (DBG & 512) {  // experiment: the burst just before the barrier
        PA_OPAQUE(lo8);
        PA_STL(ob, lo8, o0); PA_STL(ob + osc, lo8, o1); PA_STL(ob + 2 * osc, lo8, o2); PA_STL(ob + 3 * osc, lo8, o3);
        PA_STL(ob + 4 * osc, lo8, o4); PA_STL(ob + 5 * osc, lo8, o5); PA_STL(ob + 6 * osc, lo8, o6); PA_STL(ob + 7 * osc, lo8, o7);
      }
      __syncthreads();
      if (!OLD_SCHED && !(DBG & 1024)) { PA_OPAQUE(lo8); f[SP] = PA_LDG(gp, lo8); }  // request for plane p+5, after the barrier (see the header)
      const double nxl = (DBG & 4) ? nzq : S.nx[SQ][rr - 1][xs - 1], nxr = (DBG & 4) ? nzp : S.nx[SQ][rr - 1][xs + 1];
      const double nys = (DBG & 4) ? nzq : S.ny[SQ][rr - 1][lane], nyn = (DBG & 4) ? nzp : S.ny[SQ][rr + 1][lane];
      const double fznh = zflux(dxinv[2], nzq, nzp);
      double curv = 0.0;
      curv += cdiff(dxinv[0], nxl, nxq, nxr);
      const double cty = cdiff(dxinv[1], nys, nyq, nyn), ctz = favg(fzn, fznh);
      curv += cty;
      curv += ctz;
      curv = curv * 0.5;
      if (CG && !CLIP && ncg_tile) {  // (uniform) NCG: what the fix-up of the first cell behind a special x face needs, handed to the edge wave
        double* hl = &S.h[p & 1][rr][0][0];
        double* hh = &S.h[p & 1][rr][1][0];
        const int xh = llast + 1 - xs;  // 0, 1, 2: the last three columns of the tile (lanes past the box edge mirror the last one)
        // ONE 16-byte LDS store per side: the three lanes next to the face each write a pair, the values of the other lanes by DPP shifts
        if (ncgl) {  // (uniform) lanes 0, 1, 2 = the first, second, third cell behind the low face
          const double n2 = lane_from_right(nxq), tz1 = lane_from_left(ctz), ty2 = lane_from_left(lane_from_left(cty));
          pa_d2 v;
          v.x = lane == 1 ? tz1 : nxq;
          v.y = lane == 0 ? n2 : (lane == 1 ? 0.0 : ty2);
          if (lane <= 2) *(pa_d2*)&hl[lane == 0 ? 0 : (lane == 1 ? 4 : 2)] = v;
        }
        if (ncgh) {  // lanes llast, llast - 1, llast - 2 = the first, second, third cell behind the high face
          const double n2 = lane_from_left(nxq), tz1 = lane_from_right(ctz), ty2 = lane_from_right(lane_from_right(cty));
          const int hk = llast - lane;  // 0, 1, 2
          pa_d2 v;
          v.x = hk == 1 ? tz1 : nxq;
          v.y = hk == 0 ? n2 : (hk == 1 ? 0.0 : ty2);
          if (hk >= 0 && hk <= 2) *(pa_d2*)&hh[hk == 0 ? 0 : (hk == 1 ? 4 : 2)] = v;
        }
        (void)xh;
      }
      if (OLD_SCHED) __builtin_amdgcn_sched_barrier(0);
      if (OLD_SCHED) { if (PAIR) { PA_OPAQUE(lo16); store_pair(ob + 4 * osc, lo16, odd, o4, o5); } else PA_STL(ob + 5 * osc, lo8, o5); }
      if (OLD_SCHED) __builtin_amdgcn_sched_barrier(0);
      // phi gradient at plane q (pc = phi(q), p0 = phi(q+1); fzp = low z-face flux at plane q)
      double gx = 0, gy = 0, gz = 0, gm = 0, fzph = 0;
      if (!GOUT) {
        const double pl = (DBG & 4) ? p1 : S.p[SQ][rr][xs - 1], pr = (DBG & 4) ? p0 : S.p[SQ][rr][xs + 1];
        const double ps = (DBG & 4) ? p1 : S.p[SQ][rr - 1][xs], pnn = (DBG & 4) ? p0 : S.p[SQ][rr + 1][xs];
        gx = cdiff(dxinv[0], pl, pc, pr);
        gy = cdiff(dxinv[1], ps, pc, pnn);
        fzph = zflux(dxinv[2], pc, p0);
        gz = favg(fzp, fzph);
      }
      if (OLD_SCHED) __builtin_amdgcn_sched_barrier(0);
      if (OLD_SCHED && !PAIR) PA_STL(ob + 6 * osc, lo8, o6);
      if (OLD_SCHED) __builtin_amdgcn_sched_barrier(0);
      if (!GOUT) gm = (DBG & 2) ? (gx * gx + gy * gy + gz * gz) : sqrt(gx * gx + gy * gy + gz * gz);
      if (OLD_SCHED) __builtin_amdgcn_sched_barrier(0);
      if (OLD_SCHED) { if (PAIR) { PA_OPAQUE(lo16); store_pair(ob + 6 * osc, lo16, odd, o6, o7); } else PA_STL(ob + 7 * osc, lo8, o7); }
      if (DBG & 1024) { PA_OPAQUE(lo8); f[SP] = PA_LDG(gp, lo8); }  // experiment: the request at the end of the step
      ob += (p >= k0 + 2) ? ops : 0;
      if (GOUT) {  // [Progress K Nx Ny Nz] + G; cm = c at plane q; Progress and G are not clipped (curvature.cpp:557-566 clips K and N)
        ob2 += (p >= k0 + 2) ? ops2 : 0;
        const bool clip = CLIP && ((cm < thr) || (cm > 1.0 - thr));
        if constexpr (KG) {  // k_gauss_curv (pa_curvopts.hip) on plane q: x / y neighbours of G from the rings, z-neighbours from the registers
          double H[3][3];
          H[0][0] = cdiff(dxinv[0], PA_SGX[SQ][rr][xs - 1], gxq, PA_SGX[SQ][rr][xs + 1]);
          H[0][1] = cdiff(dxinv[1], PA_SGX[SQ][rr - 1][xs], gxq, PA_SGX[SQ][rr + 1][xs]);
          H[0][2] = cdiff(dxinv[2], gxm, gxq, ggx);
          H[1][0] = cdiff(dxinv[0], SG.gy[SQ][rr][xs - 1], gyq, SG.gy[SQ][rr][xs + 1]);
          H[1][1] = cdiff(dxinv[1], SG.gy[SQ][rr - 1][xs], gyq, SG.gy[SQ][rr + 1][xs]);
          H[1][2] = cdiff(dxinv[2], gym, gyq, ggy);
          H[2][0] = cdiff(dxinv[0], SG.gz[SQ][rr][xs - 1], gzq, SG.gz[SQ][rr][xs + 1]);
          H[2][1] = cdiff(dxinv[1], SG.gz[SQ][rr - 1][xs], gzq, SG.gz[SQ][rr + 1][xs]);
          H[2][2] = cdiff(dxinv[2], gzm, gzq, ggz);
          const double ax0 = H[1][1] * H[2][2] - H[2][1] * H[1][2];
          const double ay0 = H[1][2] * H[2][0] - H[2][2] * H[1][0];
          const double az0 = H[1][0] * H[2][1] - H[2][0] * H[1][1];
          const double ax1 = H[0][2] * H[2][1] - H[2][2] * H[0][1];
          const double ay1 = H[0][0] * H[2][2] - H[2][0] * H[0][2];
          const double az1 = H[0][1] * H[2][0] - H[2][1] * H[0][0];
          const double ax2 = H[0][1] * H[1][2] - H[1][1] * H[0][2];
          const double ay2 = H[0][2] * H[1][0] - H[1][2] * H[0][0];
          const double az2 = H[0][0] * H[1][1] - H[1][0] * H[0][1];
          const double cx = gxq, cy = gyq, cz = gzq;
          const double kg = (cx * (ax0 * cx + ax1 * cy + ax2 * cz) + cy * (ay0 * cx + ay1 * cy + ay2 * cz) + cz * (az0 * cx + az1 * cy + az2 * cz)) /
                            ((gnq * gnq) * (gnq * gnq));
          o8 = clip ? 0.0 : kg;
          gxm = gxq; gym = gyq; gzm = gzq;
          gnq = (1e-14 < sn) ? sn : 1e-14;  // of plane p = the next step's plane q
          const int q = p - 1;
          gs = q >= k0 && (gsb || q - V.lo[2] < 3 || V.hi[2] - q < 3);
        }
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
    // results of the last plane (k1)
    if (PAIR) {
      store_pair(ob, lo16, odd, o0, o1);
      store_pair(ob + 2 * osc, lo16, odd, o2, o3);
      store_pair(ob + 4 * osc, lo16, odd, o4, o5);
      store_pair(ob + 6 * osc, lo16, odd, o6, o7);
    } else if (GOUT) {
      PA_STG(ob, lo8, o0);
      PA_STG(ob + osc, lo8, o1);
      PA_STG(ob + 2 * osc, lo8, o2);
      PA_STG(ob + 3 * osc, lo8, o3);
      PA_STG(ob + 4 * osc, lo8, o4);
      if constexpr (KG) PA_STG(ob + 5 * osc, lo8, o8);
      if (gs) {
        PA_STG(ob2, lo8, o5);
        PA_STG(ob2 + osc2, lo8, o6);
        PA_STG(ob2 + 2 * osc2, lo8, o7);
      }
    } else {
      PA_STG(ob, lo8, o0);
      PA_STG(ob + osc, lo8, o1);
      PA_STG(ob + 2 * osc, lo8, o2);
      PA_STG(ob + 3 * osc, lo8, o3);
      PA_STG(ob + 4 * osc, lo8, o4);
      PA_STG(ob + 5 * osc, lo8, o5);
      PA_STG(ob + 6 * osc, lo8, o6);
      PA_STG(ob + 7 * osc, lo8, o7);
    }
    if (ncg_tile) __syncthreads();
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
    // uniform base (plane k0-2 of the component) + per-lane in-plane byte offsets (< 4 GiB)
    const char* gb = (const char*)(P.p + P.idx(P.lo[0], P.lo[1], k0 - 2, pcomp));
    unsigned og = (unsigned)((j - P.lo[1]) * P.nx + (i - P.lo[0])) * 8u;
    unsigned oo = (unsigned)((j - P.lo[1]) * P.nx + ((side ? i + 1 : i - 1) - P.lo[0])) * 8u;
    const char* gp = gb;
    // the column beyond the edge column starts at plane k0-1: wave-uniform base + lane offset `oo`.  With CG the stream
    // is addressed through a per-lane pointer instead (gol): the lanes of a side that is a special x face read that
    // face's compact array [plane][y], one plane ahead (see the halo rows)
    const char* go = gb + pps;
    const char* gol = gb + pps + oo;
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
        gol = cb;            // c of the column beyond, same planes as the phi stream it replaces
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
    // NCG: ONE 16-byte store per step and face (the sweep is bound by the store instructions a CU can issue, not by their bytes): lane
    // (pair, row) of the edge wave writes a pair of values of row `row` of the tile -- pair 0 = (N_x of the first, second cell), 1 = (N_x
    // of the third cell, y term of K), 2 = (z term of K, unused) -- into the face's arrays of pairs; a lane without a live cell writes to
    // an element of the arrays' ring that nobody reads, so that the store is unconditional
    const int npair = lane / PA_MTY, nrow = lane % PA_MTY + 1, njr = j0 + nrow - 1;
    pa_d2* nbase[2] = {nullptr, nullptr};
    pa_d2* ndum = nullptr;
    const bool nrow_ok = npair < 3 && njr <= V.hi[1];
    if (CG && ncg_tile) {
      pa_d2* const n2 = (pa_d2*)A.ncg;
      if (ncgl) nbase[0] = n2 + (cgxl - bp.cg_base());
      if (ncgh) nbase[1] = n2 + (cgxh - bp.cg_base());
      ndum = ncgl ? nbase[0] : nbase[1];
    }
    const long long npl = (long long)(ny + 2), ncgs = A.ncgs;
    auto ncg_store = [&](int q2, int par) __attribute__((always_inline)) {
#pragma unroll
      for (int sd = 0; sd < 2; ++sd) {
        if (!nbase[sd]) continue;  // (uniform)
        const pa_d2 v = *(const pa_d2*)&S.h[par][nrow][sd][2 * min(npair, 2)];
        pa_d2* dst = (nrow_ok && q2 >= k0) ? nbase[sd] + (long long)min(npair, 2) * ncgs + (long long)(q2 - V.lo[2] + 1) * npl + (njr - V.lo[1] + 1) : ndum;
        *dst = v;
      }
    };
    auto step_impl = [&](auto spc, int p, auto ncgf) __attribute__((always_inline)) {
      constexpr int SP = decltype(spc)::value, SP1 = (SP + 1) % 3;
      constexpr bool NCGF = decltype(ncgf)::value;
      double x, xo;
      PA_TAKE(x, f[SP]);
      PA_TAKE(xo, fo[SP]);
      PA_OPAQUE(og);
      if (!CG) PA_OPAQUE(oo);
      __builtin_amdgcn_sched_barrier(0);
      gp += (p + 5 <= kfmax) ? pps : 0;
      go += (p + 4 <= pend) ? pps : 0;
      if (CG) gol += (p + 4 + sh <= pend) ? pso : 0;
      if (OLD_SCHED) {
        f[SP] = PA_LDG(gp, og);
        fo[SP] = PA_LDO(0);
      }
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
      if constexpr (KG) { PA_SGX[SP][rr][xs] = ggx; SG.gy[SP][rr][xs] = ggy; SG.gz[SP][rr][xs] = ggz; }
      __syncthreads();
      if (!OLD_SCHED) {
        PA_OPAQUE(og);
        if (!CG) PA_OPAQUE(oo);
        f[SP] = PA_LDG(gp, og);
        fo[SP] = PA_LDO(0);
      }
      if (NCGF) ncg_store(p - 2, (p - 1) & 1);  // the row waves handed plane p - 2 over after the PREVIOUS barrier
      cm = cc; cc = cp; cp = xsp ? xo : PA_PROG(x); co = (xmode == 2) ? xo : PA_PROG(xo);
      fzc = fzh;
      p0 = p1; p1 = x;
    };
    auto step = [&](auto spc, int p) __attribute__((always_inline)) { step_impl(spc, p, std::false_type{}); };
    auto stepn = [&](auto spc, int p) __attribute__((always_inline)) { step_impl(spc, p, std::true_type{}); };
    if (CG && ncg_tile) {
      PA_RUN3(stepn)
      __syncthreads();  // (every wave of an NCG tile ends with this barrier) the last plane's hand-over
      ncg_store(k1, pend & 1);
    } else {
      PA_RUN3(step)
    }
  }
#undef PA_LDO
#undef PA_PROG
#undef PA_RUN3
#undef PA_SGX
}

template <typename BP, int PA_MTY, bool CLIP, bool PAIR = false, int DBG = 0, bool CG = false>
__global__ __launch_bounds__(64 * (PA_MTY + 3), 1) void k_gradcurv_march3(BP bp, MarchArgs A) {
  gradcurv_march3_body<BP, PA_MTY, CLIP, PAIR, DBG, CG>(bp, A, blockIdx.x, blockIdx.y);
}

// The CG sweeps of several levels in ONE launch (exact-normal pipeline: the sweeps of different levels do not depend on
// each other -- only the fix-up afterwards needs the coarser level's normals).  A level of a few boxes is 1-2 rounds of
// workgroups on 256 CUs, each launch ends in a tail of idle CUs, and the next one ramps up again: levels back to back
// in one grid fill those tails (BASELINE config 5's shape, and every rank's share of a sharded hierarchy).  Level l
// owns workgroups wg0[l] .. wg0[l+1]-1, each range a multiple of 8 so that the XCD-aware numbering (order 2) is kept.
// Component slots (blockIdx.y = slot z of a batch of components, pa_gradcurv_run_comps2): phi component + z, outputs + 8 z, the
// slot's set of compact ghost arrays (cg + z cgs[l]) and its progress range prog[2 z], prog[2 z + 1] (device table, null: one slot).
// The sweeps of a batch's components are then ONE launch: on small levels (config 5's shape: 4 x 256^3 per component, a rank's
// share of a sharded hierarchy) a launch per component ended in an idle tail each time.
struct SweepBatch {
  int n;
  unsigned wg0[PA_MAXB + 1];
  LevelBP2 bp[PA_MAXB];
  MarchArgs A[PA_MAXB];
  long long cgs[PA_MAXB] = {};
  const double* prog = nullptr;
};
__device__ __forceinline__ void sweep_slot(const SweepBatch& S, int l, LevelBP2& bp, MarchArgs& A) {
  bp = S.bp[l];
  A = S.A[l];
  const int z = (int)blockIdx.y;
  if (z) {
    A.pcomp += z;
    A.ocomp += 8 * z;
    bp.L.cg += z * S.cgs[l];
  }
  if (S.prog) { A.pmin = S.prog[2 * z]; A.invdenom = S.prog[2 * z + 1]; }
}
template <int PA_MTY, bool CLIP = false, int GOUT = 0>
__global__ __launch_bounds__(64 * (PA_MTY + 3), 1) void k_gradcurv_march3_levels(SweepBatch S) {
  int l = 0;
  while (l + 1 < S.n && blockIdx.x >= S.wg0[l + 1]) ++l;
  if (GOUT || (gridDim.y == 1 && !S.prog)) {  // one component: the arguments straight from the argument segment
    gradcurv_march3_body<LevelBP2, PA_MTY, CLIP, false, 0, true, GOUT>(S.bp[l], S.A[l], blockIdx.x - S.wg0[l], 0u);
    return;
  }
  LevelBP2 bp;
  MarchArgs A;
  sweep_slot(S, l, bp, A);
  gradcurv_march3_body<LevelBP2, PA_MTY, CLIP, false, 0, true, GOUT>(bp, A, blockIdx.x - S.wg0[l], 0u);
}
