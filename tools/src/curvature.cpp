// curvature3d -- drop-in for PeleAnalysis Src/curvature.cpp on MI355X.
//   curvature3d.ex infile=<plt> [outfile=<root>_K] [finestLevel=<n>] [progressName=temp] [progMin=..] [progMax=..]
//       [useFileMinMax=1] [threshold_prog=0] [threshold_value=1e-4] [do_gaussCurv=0] [do_strain=0]
//       [getStrainTensor=0] [do_velnormal=0] [Aux_Variables="a b"] [sym_dir="0 0 0"] [is_per="1 1 1"]
//       [verbose=0] [fused=1]
// Output components (curvature.cpp:165-236, 796-844): [progressName, (x/y/z_velocity), aux..., Progress,
// SmoothedProgress, MeanCurvature_<v>, FlameNormalX/Y/Z_<v>, GaussianCurvature_<v>, (StrainRate_<v>),
// (ROST_dU?d? x9), (VelFlameNormal)].
// Deviations, all stated: SmoothedProgress / GaussianCurvature are 0.0 when their option is off (the
// reference leaves them uninitialised: quirk Q1); StrainRate keeps the reference's value = div u
// (quirk Q3); the velocities are also read when only do_velnormal is set (the reference reads them
// only under do_strain and then indexes whatever sits at idVst); do_smooth=1 [smoothing_time=1e-7] solves
// the reference's composite implicit diffusion problem to its tolerance (1e-12) with BiCGStab instead of
// AMReX's MLMG (pa_smooth_solve): same field to ~1e-12, not the same iteration history.
// Built twice: curvature3d.ex, and with -DPA_SPACEDIM=2 curvature2d.ex = the AMREX_SPACEDIM == 2 build: 2-D plotfile in and
// out, sym_dir / is_per of two entries, components [progressName, aux..., Progress, SmoothedProgress, MeanCurvature_<v>,
// FlameNormalX/Y_<v>] (no Gaussian curvature in 2-D, curvature.cpp:208-226), MeanCurvature = d(nx)/dx + d(ny)/dy without
// the 0.5 of the 3-D build (:542-546).  The level is one plane of cells with z a homogeneous-Neumann wall
// (pa_curv_params.spacedim = 2); do_strain / getStrainTensor (2 x 2) / do_velnormal work through a zero third velocity
// component on the device; do_gaussCurv (3-D only in the reference) is not available in this build; do_smooth solves on the planes (not refined in z).
#include "../common/pa_team.h"
#ifndef PA_SPACEDIM
#define PA_SPACEDIM 3
#endif

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cerr << "usage:\n" << argv[0] << " infile=<plotfilename> \n\tOptions:\n\tis_per=<L M N> progressName=<name>\n";
    return 1;
  }
  pa::ParmParse pp(argc, argv);
  int verbose = 0, finestLevel = 1000;
  int do_gaussCurv = 0, floorIt = 0, useFileMinMax = 1, do_threshold = 0, do_smooth = 0, do_strain = 0, getStrainTensor = 0, do_velnormal = 0,
      fused = 1;
  std::string progressName = "temp", infile;
  double progMin = 1.0e20, progMax = -1.0e20, threshold = 0.0001;
  pp.query("verbose", verbose);
  pp.get("infile", infile);
  std::string outfile = pa::getFileRoot(infile) + "_K";
  pp.query("outfile", outfile);
  pp.query("finestLevel", finestLevel);
  pp.query("do_gaussCurv", do_gaussCurv);
  pp.query("progressName", progressName);
  pp.query("progMin", progMin);
  pp.query("progMax", progMax);
  pp.query("floorIt", floorIt);
  pp.query("useFileMinMax", useFileMinMax);
  pp.query("threshold_prog", do_threshold);
  pp.query("threshold_value", threshold);
  pp.query("do_smooth", do_smooth);
  pp.query("do_strain", do_strain);
  if (do_strain) pp.query("getStrainTensor", getStrainTensor);
  pp.query("do_velnormal", do_velnormal);
  pp.query("fused", fused);
  double smoothing_time = 1.e-7;  // curvature.cpp:93-94
  pp.query("smoothing_time", smoothing_time);
  const int nAux = pp.countval("Aux_Variables");
  std::cout << "infile = " << infile << "\n" << "reading plt file = " << infile << "\n";
  pa::PhaseTimer tm(pp, PA_SPACEDIM == 2 ? "curvature2d" : "curvature3d");
  pa::PlotfileHeader H = pa::read_header(infile, PA_SPACEDIM);
#if PA_SPACEDIM == 2
  if (do_gaussCurv) pa::Abort("do_gaussCurv is not available in the 2-D build");  // 3-D only in the reference too (curvature.cpp:208-216)
#endif
  finestLevel = std::min(finestLevel, H.nlev - 1);
  const int Nlev = finestLevel + 1;
  const int idC = H.comp(progressName);
  if (idC < 0) pa::Abort("Wrong progress variable name: " + progressName);
  std::vector<std::string> inNames{progressName};
  std::vector<int> inComps{idC};
  const bool need_vel = do_strain || do_velnormal;
  const int idVst = 1;
  if (need_vel)
#if PA_SPACEDIM == 2
    for (const char* v : {"x_velocity", "y_velocity"}) {
#else
    for (const char* v : {"x_velocity", "y_velocity", "z_velocity"}) {
#endif
      if (H.comp(v) < 0) pa::Abort(std::string("Unknown velocity variable name: ") + v);
      inNames.push_back(v);
      inComps.push_back(H.comp(v));
    }
  for (int i = 0; i < nAux; ++i) {
    std::string a;
    pp.get("Aux_Variables", a, i);
    if (H.comp(a) < 0) pa::Abort("Unknown auxiliary variable name: " + a);
    inNames.push_back(a);
    inComps.push_back(H.comp(a));
  }
  const int nCompIn = (int)inNames.size();
#if PA_SPACEDIM == 2
  // the library wants three consecutive velocity components: a zero one is slipped in behind y_velocity on the device
  // (every term it enters is an exact zero); devOf() maps a tool component to its device component
  const int hidden = need_vel ? idVst + 2 : -1;
#else
  const int hidden = -1;
#endif
  auto devOf = [hidden](int c) { return (hidden >= 0 && c >= hidden) ? c + 1 : c; };
  const int nCompDev = nCompIn + (hidden >= 0 ? 1 : 0);
  const int idProg = nCompIn, idSmProg = idProg + 1, idKm = idSmProg + 1, idN = idKm + 1, idKg = idN + 3;
#if PA_SPACEDIM == 2
  int idSR = -1, idROST = -1, idVelNormal = -1, nCompOut = idN + 2;  // no GaussianCurvature slot (curvature.cpp:218-226)
  if (do_strain) { idSR = idN + 2; nCompOut = idSR + 1; }
#else
  int idSR = -1, idROST = -1, idVelNormal = -1, nCompOut = idKg + 1;
  if (do_strain) { idSR = idKg + 1; nCompOut = idSR + 1; }
#endif
  if (getStrainTensor) { idROST = nCompOut; nCompOut = idROST + PA_SPACEDIM * PA_SPACEDIM; }
  if (do_velnormal) { idVelNormal = nCompOut; nCompOut += 1; }
  std::vector<int> sym_dir(3, 0), is_per(3, 1);
#if PA_SPACEDIM == 2
  is_per[2] = 0;  // the plane's normal: a wall with the default Neumann condition
  for (const char* key : {"sym_dir", "is_per"}) {
    std::vector<int> v2;
    if (pp.countval(key)) {
      pp.queryarr(key, v2, 0, 2);
      std::vector<int>& dst = std::string(key) == "sym_dir" ? sym_dir : is_per;
      dst[0] = v2[0];
      dst[1] = v2[1];
    }
  }
#else
  pp.queryarr("sym_dir", sym_dir, 0, 3);
  pp.queryarr("is_per", is_per, 0, 3);
#endif
  int32_t bc[3];
  pa::bc_from_flags(is_per, sym_dir, bc);
  const bool options = do_gaussCurv || do_strain || do_velnormal || do_smooth;
  const int nres = options ? 18 : 8;

  pa::AsyncTeam ateam(pp);  // the HIP contexts (ngpus of them, pa_team.h) come up behind the reads
  std::vector<pa::HostMF> in(Nlev), ostate(Nlev);
  std::vector<pa::Box3> doms;
  // retile=1 (default): data held and swept on the file's cells merged into large boxes (pa_level_retile; identical results in
  // every cell), written back on the file's BoxArray
  const std::vector<std::vector<pa::Box3>> fileBoxes = pa::level_boxes(H, Nlev), tile = pa::retile_levels(fileBoxes, pp);
  for (int lev = 0; lev < Nlev; ++lev) {
    if (verbose) std::cout << "Reading data for level " << lev << "\n";
    in[lev].define(tile[lev], nCompDev, 2);
    for (int c = 0; c < nCompIn; ++c) pa::read_comp(H, lev, inComps[c], in[lev], devOf(c));
    for (auto& B : H.lev[lev].boxes) tm.cells += B.numPts();
    doms.push_back(H.lev[lev].domain);
    ostate[lev].define(tile[lev], nCompOut, 0);
  }
  tm.mark("read");
  pa::Team& team = ateam.get();
  tm.mark("hip_context_wait");
  if (team.n > 1) std::cout << "Boxes distributed over " << team.n << " GPUs, transport: " << team.transport << std::endl;
  std::vector<std::vector<int32_t>> owner(Nlev);
  for (int lev = 0; lev < Nlev; ++lev) owner[lev] = pa::shard_boxes(tile[lev], team.n);  // DistributionMapping(ba), curvature.cpp:289
  // ngpus > 1: the reference's MPI ranks own the FABs DistributionMapping gives them (curvature.cpp:289); here every rank
  // (host thread + GPU) runs the same pipeline on its share and the library fills ghost cells across ranks.  do_smooth: the
  // composite solve is DISTRIBUTED like the reference's MLMG (every rank iterates on its own boxes; restriction, ghost fills,
  // flux register and dot products cross ranks: pa_smooth.hip) -- the one-rank field to the solver tolerance, not its bits
  // (PA_SMOOTH_REPLICATED=1: every rank solves the whole hierarchy, bit-identical).  Downstream of the solve, curvature / normals are an
  // ill-conditioned function (n = G / |G|) of a field that is itself only fixed to ~1e-12 by the solver tolerance (the reference's
  // MLMG solve has the same property): the tests hold the smoothed field to 1e-12, the downstream fields to the DERIVED bound
  // |dn| <= 2 eps S / |G|, |dK| <= 0.5 sum(dxinv) max|dn| (S = |dxinv|_2), and show that the tool's curvature of the ORACLE's
  // smoothed field is the oracle's bit for bit in every cell (tests/test_gpu_smooth.py, test_plotfile_tools.py).
  std::vector<std::string> nnames(inNames);
  nnames.resize(nCompOut);
  nnames[idProg] = "Progress";
  nnames[idSmProg] = "SmoothedProgress";
  nnames[idKm] = "MeanCurvature_" + progressName;
  nnames[idN] = "FlameNormalX_" + progressName;
  nnames[idN + 1] = "FlameNormalY_" + progressName;
#if PA_SPACEDIM == 3
  nnames[idN + 2] = "FlameNormalZ_" + progressName;
  nnames[idKg] = "GaussianCurvature_" + progressName;
#endif
  if (do_strain) nnames[idSR] = "StrainRate_" + progressName;
  if (getStrainTensor) {
    const std::string dirChar[3] = {"x", "y", "z"};
    for (int i = 0; i < PA_SPACEDIM * PA_SPACEDIM; ++i)
      nnames[idROST + i] = "ROST_dU" + dirChar[i / PA_SPACEDIM] + "d" + dirChar[i % PA_SPACEDIM];  // curvature.cpp:815-823
  }
  if (do_velnormal) nnames[idVelNormal] = "VelFlameNormal";
  // one GPU: the writer thread starts now and takes every level as soon as it is assembled and downloaded (pa::LevelGate); the later
  // levels come down while the earlier ones are written
  std::vector<int> isteps(Nlev, 0);
  pa::LevelGate gate;
  const std::function<void(int)> wait_level = [&](int l) { gate.wait(l); };
  const bool overlap_write = team.n == 1;
  pa::OldOutput old_out;
  old_out.move_away(outfile, infile, pp);  // UtilCreateCleanDirectory: header, variables and parameters are validated, the data are read
  std::thread writer;
  if (overlap_write) {
    std::cout << "Writing new data to " << outfile << "\n";
    writer = std::thread([&] { pa::write_plotfile(outfile, nnames, doms, H.prob_lo, H.prob_hi, ostate, 0.0, isteps, 2, PA_SPACEDIM, nullptr, pa::boxes_if_retiled(fileBoxes, tile), &wait_level); });
  }
  team.run([&](int r) {
    pa::Ctx& ctx = *team.ctx[r];
    std::vector<std::unique_ptr<pa::DevLevel>> dl;
    std::vector<std::unique_ptr<pa::DevMF>> dst, dwork, dout;
    std::vector<pa::Share> sh;
    std::vector<pa::HostMF> loc(Nlev);
    for (int lev = 0; lev < Nlev; ++lev) {
      sh.emplace_back(tile[lev], owner[lev], r);
      dl.emplace_back(new pa::DevLevel(ctx, tile[lev], H.lev[lev].domain, is_per.data(), H.prob_lo, H.prob_hi, &owner[lev], r, team.n));
      dst.emplace_back(new pa::DevMF(ctx, *dl.back(), nCompDev, 2));
      dwork.emplace_back(new pa::DevMF(ctx, *dl.back(), 1, 2));
      dout.emplace_back(new pa::DevMF(ctx, *dl.back(), nres, 0));
      if (team.n > 1) sh.back().gather(in[lev], loc[lev]);
      ctx.check(pa_mf_upload(ctx.h, dst.back()->h, (team.n > 1 ? loc[lev] : in[lev]).data.data()));
    }
    if (r == 0) tm.mark("upload");
    std::vector<pa_mf*> s, w, o;
    for (int l = 0; l < Nlev; ++l) { s.push_back(dst[l]->h); w.push_back(dwork[l]->h); o.push_back(dout[l]->h); }
    // progress-variable range (curvature.cpp:139-160): the file min/max over the levels in use, reduced over the ranks
    double pMin = progMin, pMax = progMax;
    if (useFileMinMax) {
      for (int l = 0; l < Nlev; ++l) {
        double a, b;
        ctx.check(pa_minmax_level(ctx.h, s[l], 0, &a, &b));
        pMin = std::min(pMin, a);
        pMax = std::max(pMax, b);
      }
      ctx.check(pa_allreduce(ctx.h, &pMin, 1, 0));  // ParallelDescriptor::ReduceRealMin / Max (curvature.cpp:147-148)
      ctx.check(pa_allreduce(ctx.h, &pMax, 1, 1));
    }
    if (r == 0 && (useFileMinMax || floorIt)) {
      std::cout << "progressName = " << progressName << " at index: " << idC << "\n";
      std::cout << "useFileMinMax = " << useFileMinMax << "\n";
      std::cout << "Min/Max = " << pMin << " / " << pMax << "\n";
    }
    if ((useFileMinMax || floorIt) && pMin >= pMax) pa::Abort("progMin must be less than progMax");
    pa_curv_params P;
    P.prog_min = pMin; P.prog_max = pMax; P.do_threshold = do_threshold; P.threshold = threshold; P.fused = fused;
    P.do_gauss_curv = do_gaussCurv; P.do_strain = do_strain; P.get_strain_tensor = getStrainTensor; P.do_velnormal = do_velnormal; P.vel_comp = idVst;
    P.do_smooth = do_smooth; P.smoothing_time = smoothing_time;
    P.spacedim = PA_SPACEDIM;
    // result layout: fused sweep -> [gx gy gz |g| Nx Ny Nz K]; pass-by-pass with options -> [Progress K Nx Ny Nz Kg SR Vn ROSTx9]
    int rK, rN, rKg = -1, rSR = -1, rVn = -1, rROST = -1;
    if (options) {
      ctx.check(pa_curvature_run(ctx.h, Nlev, s.data(), 0, bc, &P, o.data(), 0));
      if (r == 0 && do_smooth) {
        // the reference's MLMG runs with setVerbose(1) whatever `verbose` says (curvature.cpp:397) and prints its iterations and
        // residuals; this solver is BiCGStab on the same composite operator, so the line names it
        int its = 0;
        double res = 0.0;
        if (pa_smooth_last(ctx.h, &its, &res) == 0)
          std::cout << "Smoothing solve (BiCGStab on the composite operator, multigrid-preconditioned where the step is stiff): " << its << " iterations, resid/bnorm = " << res << "\n";
        if (verbose) std::cout << "Progress variable smoothed successfully \n";
      }
      rK = 1; rN = 2; rKg = 5; rSR = 6; rVn = 7; rROST = 8;
    } else {
      ctx.check(pa_gradcurv_run(ctx.h, Nlev, s.data(), 0, bc, &P, w.data(), o.data(), 0));
      rK = 7; rN = 4;
    }
    ctx.check(pa_sync(ctx.h));
    if (pa_bc_errors(ctx.h) != 0) pa::Abort("coarse-fine boundary: fine grids are not properly nested in the coarse level");
    // the library's work multifabs of these levels (gradient of c, solver vectors) are not needed again: their memory goes to the
    // assembly buffers below
    for (int l = 0; l < Nlev; ++l) (void)pa_level_free_scratch(dl[l]->h);
    if (r == 0) tm.mark("compute");
    // the ghost-free output state (curvature.cpp:833-839) is put together on the device and comes down in one piece:
    // input components (valid cells of the state: the passes only write ghost cells), Progress (curvature.cpp:319, the
    // same two operations as everywhere else), then the results; slots whose option is off stay 0.0
    for (int lev = 0; lev < Nlev; ++lev) {
      pa::DevMF dfin(ctx, *dl[lev], nCompOut, 0);
      ctx.check(pa_mf_setval(ctx.h, dfin.h, 0, nCompOut, 0.0));
      for (int c = 0; c < nCompIn; ++c) ctx.check(pa_mf_copy(ctx.h, dst[lev]->h, devOf(c), dfin.h, c, 1, 0));
      ctx.check(pa_progress_level(ctx.h, dst[lev]->h, 0, pMin, pMax, dfin.h, idProg, 0));
      auto cp = [&](int dstc, int srcc) { ctx.check(pa_mf_copy(ctx.h, dout[lev]->h, srcc, dfin.h, dstc, 1, 0)); };
      cp(idKm, rK);
      for (int d = 0; d < PA_SPACEDIM; ++d) cp(idN + d, rN + d);
      if (do_smooth) cp(idSmProg, 17);
      if (do_gaussCurv) cp(idKg, rKg);
      if (do_strain) cp(idSR, rSR);
      if (getStrainTensor)
        for (int a = 0; a < PA_SPACEDIM; ++a)
          for (int e = 0; e < PA_SPACEDIM; ++e) cp(idROST + a * PA_SPACEDIM + e, rROST + a * 3 + e);  // the library's tensor is 3 x 3 row-major
      if (do_velnormal) cp(idVelNormal, rVn);
      if (team.n > 1) {
        pa::HostMF lo;
        lo.define(sh[lev].boxes, nCompOut, 0);
        ctx.check(pa_mf_download(ctx.h, dfin.h, lo.data.data()));
        sh[lev].scatter(lo, ostate[lev]);
      } else {
        ctx.check(pa_mf_download(ctx.h, dfin.h, ostate[lev].data.data()));
      }
      if (r == 0 && verbose) std::cout << "Mean curvature has been computed on level " << lev << "\n";
      if (overlap_write) gate.done(lev);
    }
  });
  tm.mark("assemble_download");
  if (overlap_write) {
    writer.join();
    tm.mark("write");
    old_out.finish();
    tm.report();
    pa::Finish();
  }
  std::cout << "Writing new data to " << outfile << "\n";
  pa::write_plotfile(outfile, nnames, doms, H.prob_lo, H.prob_hi, ostate, 0.0, isteps, 2, PA_SPACEDIM, nullptr, pa::boxes_if_retiled(fileBoxes, tile));
  tm.mark("write");
  old_out.finish();
  tm.report();
  pa::Finish();
}
