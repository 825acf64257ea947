// pa_dist.h -- exchange plans of a sharded hierarchy (see pa_dist.hip).
#pragma once
#include "pa_internal.h"

// one direction of a plan: regions {local box, lo[3], hi[3]} grouped by peer (peers ascending); a peer's regions are
// contiguous in the packed buffer, each region laid out [comp][k][j][i]
struct XSide {
  std::vector<int> peers;
  std::vector<int> first;        // first region of peers[i]; first[npeers] = number of regions
  std::vector<int32_t> regs7;
  std::vector<long long> coff;   // cells (per component) before region r; coff[nreg] = total
  long long maxcells = 0;
  int* d_regs = nullptr;
  long long* d_coff = nullptr;
  ~XSide();
};

struct XPlan {
  XSide send, recv;
  int nlocal = 0;                // same-rank region pairs (coarse-source plans only)
  long long lmax = 0;
  int* d_lsrc = nullptr;
  int* d_ldst = nullptr;
  double* sbuf = nullptr;        // grow-only packed buffers
  double* rbuf = nullptr;
  long long scap = 0, rcap = 0;
  ~XPlan();
};

// coarse data one rank keeps for the coarse-fine stencils of its boxes of one fine level
struct CsPlan {
  XPlan x;                       // coarse level (sharded) -> cs level
  pa_level* cs = nullptr;        // disjoint pieces of the coarse level (null: this rank needs none)
  std::map<int, pa_mf*> mfs;     // multifabs on cs by (component count, slot), ng = 0
  pa_mf* mf(pa_ctx* ctx, int ncomp, int slot = 0);  // slot: distinct buffers of equal component count (phi / normals)
  ~CsPlan();
};

// The LOCAL half of FillBoundary as a list of copy regions (dst box, src box, region, shift), found once per level and
// ghost width on the host, + one (region, chunk of 256 cells) entry per workgroup.  The per-cell form (k_fill_boundary:
// shell index -> cell -> wrap -> owner map -> source box -> two 64-bit FAB indices) spent 150 vector + 170 scalar
// instructions and three dependent loads on every 8 bytes it moved (SQ counters: 0.17 ms of its 0.33 ms on the headline were
// VALU issue alone).  Here the geometry of a region is wave-uniform (scalar registers) and a cell costs two magic-number
// divisions and two address sums.  ok = false (too many boxes for the host search, or a region too large for the 32-bit
// magic division): the caller keeps the per-cell kernel.
struct FbLocal {
  bool ok = false;
  int nreg = 0, nwg = 0;
  int* d_regs = nullptr;   // [nreg][16]: dbox, sbox, lo[3] (dst index space), n[3], shift[3] (src cell = dst cell - shift), ncell, m0, m1, 0, 0
  int* d_wgs = nullptr;    // [nwg][2]: region, chunk
  ~FbLocal();
};
FbLocal* pa_fb_local_plan(pa_ctx* ctx, const pa_level* L, int ng);

// The gather of a fine level's coarse patches (DLevelView::cp) from one coarse level as copy regions: (special face, coarse
// box, rectangle) found once on the host, + one (region, chunk of 256 cells) entry per workgroup.  Patch cells no coarse box
// covers keep the "missing" pattern the patch buffer is filled with when it is allocated.  ok = false: k_cpatch (owner map
// per cell) does the gather.
struct CpPlan {
  bool ok = false;
  int nreg = 0, nwg = 0;
  int* d_regs = nullptr;   // [nreg][12]: face entry, coarse box, patch-local origin (pu, pv), coarse cell of that origin [3], nu, nv, dir, m (magic of nu), 0
  int* d_wgs = nullptr;    // [nwg][2]: region, chunk
  std::vector<int> hregs;  // host copy of the regions (pa_sweep_gneed: the coarse tiles whose G a GOUT == 2 sweep must store in full)
  ~CpPlan();
};
CpPlan* pa_cp_plan(pa_ctx* ctx, const pa_level* F, const pa_level* C);

// A sharded level replicated on every rank (the implicit smoothing solve, pa_smooth.hip: a global iterative solve whose
// dot products, average-down and reflux would otherwise each be a collective): `rep` holds the whole BoxArray unsharded;
// gather = all ranks' valid cells into a multifab on rep (every rank sends its boxes to every other rank; its own by a
// local copy), back = this rank's boxes of a multifab on rep into one on the sharded level (local copies only).
struct RepPlan {
  XPlan gather, back;
  pa_level* rep = nullptr;
  ~RepPlan();
};
RepPlan* pa_rep_plan(pa_ctx* ctx, const pa_level* L);

// Restriction of a sharded fine level onto its sharded coarse level (the distributed smoothing solve, pa_smooth.hip; what
// amrex::average_down and the MLMG flux register move between ranks): `cf` is the fine BoxArray coarsened by the ratio with
// the FINE level's owners, so a rank restricts its own fine boxes without communication; `down` then copies cf's valid cells
// into the coarse level's valid cells (the coarse owner may be another rank), and flux[2 * dir + side] copies the one-cell
// ghost slab behind every special face (dir, side) of cf's boxes -- where the fine rank leaves the average fine flux of each
// coarse face -- onto the coarse cells next to the fine level, through the periodic images of the coarse domain.
struct RsPlan {
  pa_level* cf = nullptr;
  XPlan down;
  XPlan flux[6];
  ~RsPlan();
};
RsPlan* pa_rs_plan(pa_ctx* ctx, const pa_level* F, const pa_level* C, int ratio);

// group > 0: the ncomp components are groups of `group` consecutive ones, group g starting at scomp + g * sgstride in src and at
// dcomp + g * dgstride in dst (the normals of several component slots: 3 of every 8 output components)
struct XJob { XPlan* plan; const pa_mf* src; int scomp; pa_mf* dst; int dcomp; int ncomp; int group = 0, sgstride = 0, dgstride = 0; };

XPlan* pa_fb_plan(pa_ctx* ctx, const pa_level* L, int ng);
CsPlan* pa_cs_plan(pa_ctx* ctx, const pa_level* F, const pa_level* C, int mode, int ng, int halo, int ratio = 2);
// pack -> ONE grouped point-to-point call over all jobs and peers -> unpack; stream-ordered, no host synchronisation
int pa_xexchange(pa_ctx* ctx, int njobs, const XJob* jobs);
int pa_coarse_source(pa_ctx* ctx, const pa_level* fine, const pa_mf* crse, int ccomp, int ncomp, int mode, int ng, int halo, const pa_mf** src, int* scomp, int ratio = 2);
void pa_rccl_destroy(pa_ctx* ctx);
