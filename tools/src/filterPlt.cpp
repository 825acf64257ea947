// filterPlt3d -- drop-in for PeleAnalysis Src/filterPlt.cpp (box filter) on MI355X.
//   filterPlt3d.ex infile=<plt> [max_filter_level=<n>] [filter_type=1 (0 none, 1 box, 3/7 and 4/8: 3- and 5-point approximations)] [base_fgr=2] [same_fgr_all_levels=false]
//       [max_grid_size=32] [interp_type=1] [variables="a b"] [is_per="0 0 0"] [exact_filter=0] [retile=1] [allow_unverified_gaussian=0]
// exact_filter=0 (default): the filter as three 1-D passes (tensor-product weights; within 1e-12 * Linf of the reference's
// tap-order sum, HBM-bound); exact_filter=1 (or PA_FILTER_EXACT=1): Filter::apply_filter's (2ng+1)^3 taps in the
// reference's accumulation order, bit for bit with the CPU restatement (3-D build; the 2-D build always sums tap by tap).
// Output: <root>_filtered, same variable names, plotfile time (filterPlt.cpp:222-225).
// The plotfile Header stores no periodicity; like PltFileManager's Geometry it defaults to
// non-periodic unless is_per / geometry.is_periodic is given.
// Built twice: filterPlt3d.ex, and with -DPA_SPACEDIM=2 filterPlt2d.ex = the AMREX_SPACEDIM == 2 build (2-D plotfile in and
// out, is_per of two entries, (2 ng + 1)^2 taps; the level is one plane of cells with z a wall direction).
#include <cstdlib>
#include "../common/pa_team.h"
#ifndef PA_SPACEDIM
#define PA_SPACEDIM 3
#endif

int main(int argc, char** argv) {
  pa::ParmParse pp(argc, argv);
  if (argc < 2 || pp.contains("help")) {
    std::cerr << "usage:\n" << argv[0] << " infile=<plotfilename> \n\tOptions:\n\tmax_filter_level=<n> filter_type=1 base_fgr=<even n> variables=<names>\n";
    return 1;
  }
  std::string infile;
  int finestLevel = 1000, filter_type = 1, fgr = 2, max_grid_size = 32, interp_type = 1;
  bool same_fgr = false;
  pp.get("infile", infile);
  pp.query("max_filter_level", finestLevel);
  pp.query("filter_type", filter_type);
  pp.query("base_fgr", fgr);
  pp.query("same_fgr_all_levels", same_fgr);
  pp.query("max_grid_size", max_grid_size);
  pp.query("interp_type", interp_type);
  int exact_filter = 0;
  pp.query("exact_filter", exact_filter);
  if (exact_filter) setenv("PA_FILTER_EXACT", "1", 1);  // read by the library at every launch
  {  // PelePhysics filter types restated in the library: 0 none, 1 box, 3 / 7 and 4 / 8 the 3- and 5-point approximations; 2 (Gaussian) only on request
    int allow_gauss = 0;
    pp.query("allow_unverified_gaussian", allow_gauss);
    if (filter_type == 2 && !allow_gauss)
      pa::Abort("filter_type 2 (Gaussian) is refused by default: PelePhysics' Filter source is not available to this build, so its weights (ngrow rule, "
                "normalisation) could not be checked and the output could differ from the reference's at 1e-6 .. 1e-5; allow_unverified_gaussian=1 uses "
                "the textbook kernel exp(-6 r^2 / Delta^2) cut at 4 standard deviations");
    if (allow_gauss) setenv("PA_ALLOW_UNVERIFIED_GAUSSIAN", "1", 1);  // read by the library per call
    std::vector<double> wt(40);
    if (pa_filter_weights(filter_type, std::max(fgr, 1), wt.data()) < 0)
      pa::Abort("filter_type " + std::to_string(filter_type) + " is not available in this build (0 none, 1 box, 3 / 7 three-point, 4 / 8 five-point approximations; 2 Gaussian with allow_unverified_gaussian=1)");
    if (filter_type == 2)
      std::cout << "filter_type 2: Gaussian weights from the textbook kernel exp(-6 r^2 / Delta^2), cut at 4 standard deviations -- UNVERIFIED against PelePhysics' Filter" << std::endl;
  }
  if (filter_type == 1 && fgr != 1 && fgr % 2 != 0) pa::Abort("Box filter requires an even filter-to-grid ratio");
  std::vector<int> is_per(3, 0);
#if PA_SPACEDIM == 2
  {
    std::vector<int> p2(2, 0);
    if (!pp.queryarr("is_per", p2, 0, 2)) pp.queryarr("geometry.is_periodic", p2, 0, 2);
    is_per[0] = p2[0];
    is_per[1] = p2[1];
  }
#else
  if (!pp.queryarr("is_per", is_per, 0, 3)) pp.queryarr("geometry.is_periodic", is_per, 0, 3);
#endif
  pa::PhaseTimer tm(pp, PA_SPACEDIM == 2 ? "filterPlt2d" : "filterPlt3d");
  pa::PlotfileHeader H = pa::read_header(infile, PA_SPACEDIM, /* any_ratio: filterPlt.cpp:133,200 take the file's */ true);
  const int Nlev = std::min(finestLevel + 1, H.nlev);
  std::vector<std::string> names;
  std::vector<int> comps;
  if (pp.countval("variables") > 0) {
    pp.getarr("variables", names);
    for (auto& n : names) {
      if (H.comp(n) < 0) pa::Abort("Variable '" + n + "' not found in file");
      comps.push_back(H.comp(n));
    }
  } else {
    names = H.names;
    for (int c = 0; c < (int)names.size(); ++c) comps.push_back(c);
  }
  const int ncomp = (int)names.size();
  pa::AsyncTeam ateam(pp);  // the HIP contexts (ngpus of them, pa_team.h) come up behind the reads
  std::vector<pa::HostMF> host(Nlev), out(Nlev);
  std::vector<pa::Box3> doms;
  std::vector<int> ngs;
  std::vector<std::vector<double>> ws;
  // The output BoxArray is the file's re-chopped to max_grid_size (filterPlt.cpp:141); the data are held and filtered on the
  // same cells merged into large boxes (retile=1, the default; pa_level_retile) -- Filter::apply_filter and the ghost fill are
  // point-wise functions of the level's cells, tests/test_retile.py -- and written back on that BoxArray.
  std::vector<std::vector<pa::Box3>> chopped(Nlev);
  for (int lev = 0; lev < Nlev; ++lev) chopped[lev] = pa::max_size(H.lev[lev].boxes, max_grid_size);
  const std::vector<std::vector<pa::Box3>> tile = pa::retile_levels(chopped, pp, 128);
  std::cout << "Reading data..." << std::endl;
  int fgr_lev = fgr;
  for (int lev = 0; lev < Nlev; ++lev) {
    std::cout << "on level " << lev << std::endl;
    if (!same_fgr && lev > 0) fgr_lev *= H.ref_ratio[lev - 1];  // filterPlt.cpp:132-134
    std::vector<double> w(std::max(fgr_lev + 2, 40));
    const int ng = pa_filter_weights(filter_type, fgr_lev, w.data());
    if (ng < 0 || ng > 16) pa::Abort("filter width on level " + std::to_string(lev) + " (filter-to-grid ratio " + std::to_string(fgr_lev) + ") exceeds 16 ghost cells");
    ngs.push_back(ng);
    ws.push_back(w);
    const std::vector<pa::Box3>& ba = tile[lev];
    host[lev].define(ba, ncomp, ng);
    for (int c = 0; c < ncomp; ++c) pa::read_comp(H, lev, comps[c], host[lev], c);
    for (auto& B : ba) tm.cells += B.numPts();
    out[lev].define(ba, ncomp, 0);
    doms.push_back(H.lev[lev].domain);
  }
  tm.mark("read");
  pa::Team& team = ateam.get();
  tm.mark("hip_context_wait");
  if (team.n > 1) std::cout << "Boxes distributed over " << team.n << " GPUs, transport: " << team.transport << std::endl;
  std::vector<std::vector<int32_t>> owner(Nlev);
  for (int lev = 0; lev < Nlev; ++lev) owner[lev] = pa::shard_boxes(host[lev].boxes, team.n);  // DistributionMapping(ba), filterPlt.cpp:142
  std::cout << "Done!" << std::endl << "FillPatching data..." << std::endl;
  team.run([&](int r) {
    pa::Ctx& ctx = *team.ctx[r];
    std::vector<std::unique_ptr<pa::DevLevel>> dl;
    std::vector<std::unique_ptr<pa::DevMF>> din, dout;
    std::vector<pa::Share> sh;
    for (int lev = 0; lev < Nlev; ++lev) {
      const int ng = ngs[lev];
      sh.emplace_back(host[lev].boxes, owner[lev], r);
      dl.emplace_back(new pa::DevLevel(ctx, host[lev].boxes, H.lev[lev].domain, is_per.data(), H.prob_lo, H.prob_hi, &owner[lev], r, team.n));
      din.emplace_back(new pa::DevMF(ctx, *dl.back(), ncomp, ng));
      dout.emplace_back(new pa::DevMF(ctx, *dl.back(), ncomp, 0));
      if (team.n > 1) {
        pa::HostMF loc;
        sh.back().gather(host[lev], loc);
        ctx.check(pa_mf_upload(ctx.h, din.back()->h, loc.data.data()));
      } else {
        ctx.check(pa_mf_upload(ctx.h, din.back()->h, host[lev].data.data()));
      }
    }
    if (r == 0) tm.mark("upload");
    bool one_ratio = true;  // one ratio argument: hierarchies whose levels differ in ratio take the per-level calls
    for (int lev = 1; lev < Nlev; ++lev) one_ratio = one_ratio && H.ref_ratio[lev - 1] == H.ref_ratio[0];
    if (one_ratio) {  // filterPlt.cpp:159-203 for ALL levels: FillBoundary, FillPatchTwoLevels and foextrap, one launch each
      std::vector<pa_mf*> hm;
      std::vector<int32_t> hg;
      for (int lev = 0; lev < Nlev; ++lev) {
        if (r == 0) std::cout << "on level " << lev << std::endl;
        hm.push_back(din[lev]->h);
        hg.push_back(ngs[lev]);
      }
      ctx.check(pa_fill_ghosts_hierarchy(ctx.h, Nlev, hm.data(), 0, ncomp, hg.data(), Nlev > 1 ? H.ref_ratio[0] : 2, interp_type == 1 ? 1 : 0, 1));
    } else {
      for (int lev = 0; lev < Nlev; ++lev) {
        if (r == 0) std::cout << "on level " << lev << std::endl;
        ctx.check(pa_fill_boundary(ctx.h, din[lev]->h, 0, ncomp, ngs[lev]));
        if (lev > 0) ctx.check(pa_fillpatch_two_levels(ctx.h, din[lev]->h, din[lev - 1]->h, 0, ncomp, ngs[lev], H.ref_ratio[lev - 1], interp_type == 1 ? 1 : 0));
        ctx.check(pa_foextrap(ctx.h, din[lev]->h, 0, ncomp, ngs[lev]));
      }
    }
    if (r == 0) {
      // which summation the library takes (ADVICE: say so in the output): the separable three-pass form differs from the
      // reference's tap order by a few ulp (<= 1e-12 of the field's scale); exact_filter=1 / PA_FILTER_EXACT=1 keeps the tap order
      const char* fe = getenv("PA_FILTER_EXACT");
      std::cout << "Done!" << std::endl << "Filtering data... (" << ((fe && atoi(fe)) ? "tap order of Filter::apply_filter, bit-identical" : "separable form, within 1e-12 of the tap-order sum; exact_filter=1 for the tap order") << ")" << std::endl;
    }
#if PA_SPACEDIM == 2
    for (int lev = 0; lev < Nlev; ++lev) {
      if (r == 0) std::cout << "on level " << lev << std::endl;
      ctx.check(pa_boxfilter_level2d(ctx.h, din[lev]->h, dout[lev]->h, 0, ncomp, ngs[lev], ws[lev].data()));
    }
#else
    {  // every level in one call: the levels are independent and run side by side (pa_boxfilter_hierarchy)
      std::vector<const pa_mf*> hin;
      std::vector<pa_mf*> hout;
      std::vector<int32_t> hng;
      std::vector<const double*> hw;
      for (int lev = 0; lev < Nlev; ++lev) {
        if (r == 0) std::cout << "on level " << lev << std::endl;
        hin.push_back(din[lev]->h); hout.push_back(dout[lev]->h); hng.push_back(ngs[lev]); hw.push_back(ws[lev].data());
      }
      ctx.check(pa_boxfilter_hierarchy(ctx.h, Nlev, hin.data(), hout.data(), 0, ncomp, hng.data(), hw.data()));
    }
#endif
    ctx.check(pa_sync(ctx.h));
    if (pa_bc_errors(ctx.h) != 0) pa::Abort("FillPatchTwoLevels: fine grids are not properly nested in the coarse level");
    if (r == 0) tm.mark("compute");
    for (int lev = 0; lev < Nlev; ++lev) {
      if (team.n > 1) {
        pa::HostMF lo;
        lo.define(sh[lev].boxes, ncomp, 0);
        ctx.check(pa_mf_download(ctx.h, dout[lev]->h, lo.data.data()));
        sh[lev].scatter(lo, out[lev]);
      } else {
        ctx.check(pa_mf_download(ctx.h, dout[lev]->h, out[lev].data.data()));
      }
    }
  });
  std::cout << "Done!" << std::endl << "Saving filtered data..." << std::endl;
  tm.mark("download");
  std::vector<int> steps(Nlev, 0);
  pa::OldOutput old_out;
  old_out.move_away(pa::getFileRoot(infile) + "_filtered", infile, pp);  // UtilCreateCleanDirectory at write time (filterPlt.cpp:52)
  pa::write_plotfile(pa::getFileRoot(infile) + "_filtered", names, doms, H.prob_lo, H.prob_hi, out, H.time, steps, 2, PA_SPACEDIM, nullptr, pa::boxes_if_retiled(chopped, tile));
  tm.mark("write");
  old_out.finish();
  std::cout << "Done!" << std::endl;
  tm.report();
  pa::Finish();
}
