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

struct XJob { XPlan* plan; const pa_mf* src; int scomp; pa_mf* dst; int dcomp; int ncomp; };

XPlan* pa_fb_plan(pa_ctx* ctx, const pa_level* L, int ng);
CsPlan* pa_cs_plan(pa_ctx* ctx, const pa_level* F, const pa_level* C, int mode, int ng, int halo);
// pack -> ONE grouped point-to-point call over all jobs and peers -> unpack; stream-ordered, no host synchronisation
int pa_xexchange(pa_ctx* ctx, int njobs, const XJob* jobs);
int pa_coarse_source(pa_ctx* ctx, const pa_level* fine, const pa_mf* crse, int ccomp, int ncomp, int mode, int ng, int halo, const pa_mf** src, int* scomp);
void pa_rccl_destroy(pa_ctx* ctx);
