// grad3d -- drop-in for PeleAnalysis Src/grad.cpp on MI355X.
//   grad3d.ex infile=<plt> [gradVar=temp] [finestLevel=<n>] [Aux_Variables="a b"] [sym_dir="0 0 0"]
//             [is_per="1 1 1"] [outfile=<root>_gt] [ngpus=<n>] [gpu_share=0|1] [retile=1|0]
// retile=1 (default): the level data are held and swept on an internal tiling -- the file's cells merged into large boxes
// (pa_level_retile; results identical in every cell, tests/test_retile.py) -- and written back on the file's BoxArray.
// ngpus=<n>: the boxes of every level are dealt to n GPUs (one host thread each, pa_team.h) the way the reference's MPI ranks
// own them (DistributionMapping, grad.cpp:162); the output is byte-identical for every n.
// Output plotfile components: [gradVar, aux..., <v>_gx, <v>_gy, <v>_gz, ||grad<v>||], time 0, steps 0,
// ref ratio 2 (grad.cpp:241-257).  All arithmetic runs in libpeleanalysis_amd (HIP, gfx950).
// Built twice: grad3d.ex, and with -DPA_SPACEDIM=2 grad2d.ex = the AMREX_SPACEDIM == 2 build: 2-D plotfile in and out,
// sym_dir / is_per of two entries, components [gradVar, aux..., <v>_gx, <v>_gy, ||grad<v>||].  The 2-D level is handed
// to the library as one plane of cells with z a homogeneous-Neumann wall: every z difference is an exact zero, so gx,
// gy and sqrt(gx*gx + gy*gy + 0) are the 2-D values bit for bit.
#include "../common/pa_team.h"
#ifndef PA_SPACEDIM
#define PA_SPACEDIM 3
#endif

static void print_usage(char** argv) {
  std::cerr << "usage:\n" << argv[0] << " infile=<plotfilename> \n\tOptions:\n\tis_per=<L M N> gradVar=<name>\n";
  std::exit(1);
}

int main(int argc, char** argv) {
  if (argc < 2) print_usage(argv);
  pa::ParmParse pp(argc, argv);
  if (pp.contains("help")) print_usage(argv);
  std::string gradVar = "temp", infile;
  int finestLevel = 1000;
  pp.get("infile", infile);
  pp.query("gradVar", gradVar);
  pp.query("finestLevel", finestLevel);
  pa::PhaseTimer tm(pp, PA_SPACEDIM == 2 ? "grad2d" : "grad3d");
  std::string outfile = pa::getFileRoot(infile) + "_gt";
  pp.query("outfile", outfile);
  pa::PlotfileHeader H = pa::read_header(infile, PA_SPACEDIM);
  finestLevel = std::min(finestLevel, H.nlev - 1);
  const int Nlev = finestLevel + 1;
  const int idC = H.comp(gradVar);
  if (idC < 0) pa::Abort("Cannot find " + gradVar + " data in pltfile");  // quirk Q9: the reference only prints, then indexes [-1]
  const int nAux = pp.countval("Aux_Variables");
  std::vector<std::string> inNames{gradVar};
  std::vector<int> inComps{idC};
  for (int i = 0; i < nAux; ++i) {
    std::string a;
    pp.get("Aux_Variables", a, i);
    if (H.comp(a) < 0) pa::Abort("Unknown auxiliary variable name: " + a);
    inNames.push_back(a);
    inComps.push_back(H.comp(a));
  }
  const int nCompIn = (int)inNames.size(), idGr = nCompIn, nCompOut = idGr + 4;
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
  std::cout << "Periodicity assumed for this case: " << is_per[0] << " " << is_per[1] << " \n";
#else
  pp.queryarr("sym_dir", sym_dir, 0, 3);
  pp.queryarr("is_per", is_per, 0, 3);
  std::cout << "Periodicity assumed for this case: " << is_per[0] << " " << is_per[1] << " " << is_per[2] << " \n";
#endif
  int32_t bc[3];
  pa::bc_from_flags(is_per, sym_dir, bc);

  pa::AsyncTeam ateam(pp);  // the HIP contexts (ngpus of them) come up behind the reads
  std::vector<pa::HostMF> state(Nlev);
  std::vector<pa::Box3> doms;
  const std::vector<std::vector<pa::Box3>> fileBoxes = pa::level_boxes(H, Nlev), tile = pa::retile_levels(fileBoxes, pp);
  for (int lev = 0; lev < Nlev; ++lev) {
    std::cout << "Reading data for level: " << lev << std::endl;
    state[lev].define(tile[lev], nCompOut, 1);
    for (int c = 0; c < nCompIn; ++c) pa::read_comp(H, lev, inComps[c], state[lev], c);
    for (auto& B : H.lev[lev].boxes) tm.cells += B.numPts();
    doms.push_back(H.lev[lev].domain);
  }
  tm.mark("read");
  pa::Team& team = ateam.get();
  tm.mark("hip_context_wait");
  if (team.n > 1) std::cout << "Boxes distributed over " << team.n << " GPUs, transport: " << team.transport << std::endl;
  std::vector<std::vector<int32_t>> owner(Nlev);
  for (int lev = 0; lev < Nlev; ++lev) owner[lev] = pa::shard_boxes(tile[lev], team.n);  // DistributionMapping(ba), grad.cpp:162
  // output names / components are known before the data: with one GPU the writer thread starts now and takes every level as soon
  // as its download is done (pa::LevelGate), the later levels come down while the earlier ones are written
  std::vector<std::string> nnames(inNames);
  nnames.push_back(gradVar + "_gx");
  nnames.push_back(gradVar + "_gy");
  std::vector<int> ocomps;
  for (int c = 0; c < nCompIn + 2; ++c) ocomps.push_back(c);
#if PA_SPACEDIM == 3
  nnames.push_back(gradVar + "_gz");
  ocomps.push_back(idGr + 2);
#endif
  nnames.push_back("||grad" + gradVar + "||");
  ocomps.push_back(idGr + 3);
  std::vector<int> isteps(Nlev, 0);
  pa::LevelGate gate;
  const std::function<void(int)> wait_level = [&](int l) { gate.wait(l); };
  const bool overlap_write = team.n == 1;
  pa::OldOutput old_out;
  old_out.move_away(outfile, infile, pp);  // UtilCreateCleanDirectory: header, variables and parameters are validated, the data are read
  std::thread writer;
  if (overlap_write) {
    std::cout << "Writing new data to " << outfile << std::endl;
    writer = std::thread([&] { pa::write_plotfile(outfile, nnames, doms, H.prob_lo, H.prob_hi, state, 0.0, isteps, 2, PA_SPACEDIM, &ocomps, pa::boxes_if_retiled(fileBoxes, tile), &wait_level); });
  }
  team.run([&](int r) {  // one rank: its boxes of every level through the library's pipeline (cross-rank ghost fills inside)
    pa::Ctx& ctx = *team.ctx[r];
    std::vector<std::unique_ptr<pa::DevLevel>> dl;
    std::vector<std::unique_ptr<pa::DevMF>> dmf;
    std::vector<pa::Share> sh;
    std::vector<pa::HostMF> loc(Nlev);
    for (int lev = 0; lev < Nlev; ++lev) {
      sh.emplace_back(tile[lev], owner[lev], r);
      dl.emplace_back(new pa::DevLevel(ctx, tile[lev], H.lev[lev].domain, is_per.data(), H.prob_lo, H.prob_hi, &owner[lev], r, team.n));
      dmf.emplace_back(new pa::DevMF(ctx, *dl.back(), nCompOut, 1));
      pa::HostMF& src = team.n > 1 ? loc[lev] : state[lev];
      if (team.n > 1) sh.back().gather(state[lev], loc[lev]);
      ctx.check(pa_mf_upload_comps(ctx.h, dmf.back()->h, src.data.data(), 0, nCompIn));  // the inputs only: the 4 output components are written on the device
    }
    if (r == 0) tm.mark("upload");
    std::vector<pa_mf*> mfs;
    for (auto& m : dmf) mfs.push_back(m->h);
    ctx.check(pa_grad_run(ctx.h, Nlev, mfs.data(), 0, bc, mfs.data(), idGr));  // outputs into the same MultiFab, like grad.cpp
    ctx.check(pa_sync(ctx.h));
    if (pa_bc_errors(ctx.h) != 0) pa::Abort("coarse-fine boundary: fine grids are not properly nested in the coarse level");
    if (r == 0) tm.mark("compute");
    for (int lev = 0; lev < Nlev; ++lev) {
      pa::HostMF& dst = team.n > 1 ? loc[lev] : state[lev];
      ctx.check(pa_mf_download_comps(ctx.h, dmf[lev]->h, dst.data.data(), idGr, 4));  // the outputs only: the inputs are still on the host
      if (team.n > 1) sh[lev].scatter(loc[lev], state[lev]);
      if (overlap_write) gate.done(lev);
    }
  });
  tm.mark("download");
  if (overlap_write) {
    writer.join();
    tm.mark("write");
    old_out.finish();
    tm.report();
    pa::Finish();
  }

  std::cout << "Writing new data to " << outfile << std::endl;
  pa::write_plotfile(outfile, nnames, doms, H.prob_lo, H.prob_hi, state, 0.0, isteps, 2, PA_SPACEDIM, &ocomps, pa::boxes_if_retiled(fileBoxes, tile));
  tm.mark("write");
  old_out.finish();
  tm.report();
  pa::Finish();
}
