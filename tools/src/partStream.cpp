// partStream3d -- drop-in for PeleAnalysis Src/partStream.cpp (streamlines of the velocity field, two per seed,
// RK4 through the AMR hierarchy) on MI355X.
//   partStream3d.ex infile=<plt> (isoFile=<mef> | seedLoc="x y z" | seedRakeNum=<n> seedRakeL="x y z" seedRakeR="x y z")
//       | oneSeedPerCell=1   [nGrow=3] [Nsteps=50] [hRK=0.1] [ngpus=<n> [gpu_share=0|1]]
// The vector field (x/y/z_velocity) gets nGrow ghost layers by FillPatch with piecewise-constant interpolation
// (partStream.cpp:160-177), the lines are traced on the GPU (pa_stream_trace = StreamPC.cpp, bit for bit with the
// CPU restatement) and written in Tecplot ASCII to tec.dat/str_00000.dat (StreamPC.cpp:308-371).
// Deviations, all stated: zones are written in seed order (forward line, then backward line of each seed) -- the
// reference writes them in the order of AMReX's particle tiles after its redistributions; the AMReX particle
// plotfile "junkPlt" is not written; ghost cells outside the (non-periodic) domain hold 0.0 where the reference
// leaves them uninitialised; isoFile accepts the MEF that isosurface writes (label line first) as well as the label-less
// form partStream.cpp:73-82 parses.  oneSeedPerCell (partStream.cpp:21-61): one seed at the centre of every cell, not
// covered by the next finer level, of every grid that contains cell (0,50,107) of its level, levels coarse to fine, grids
// in BoxArray order, cells x fastest ([RECALLED] the particle container's MFIter is untiled by default: tile box = valid box).
// ngpus=<n>: the lines are dealt to n ranks (host threads, one GPU each or sharing with gpu_share=1); every rank holds
// the whole velocity field and traces its share of the seeds, the "a line has left its grid" flag of every step is
// reduced over the ranks (pa_stream_trace_ranks) as StreamPC's Redistribute is collective over the MPI ranks: the
// output is byte-identical for every n.
#include "../common/pa_team.h"
#include <sstream>

static std::vector<double> read_mef_nodes(const std::string& file) {
  std::ifstream f(file, std::ios::binary);
  if (!f) pa::Abort("Unable to open file : " + file);
  std::string l1, l2;
  std::getline(f, l1);
  auto ntok = [](const std::string& s) { std::istringstream is(s); std::string t; int n = 0; while (is >> t) ++n; return n; };
  std::streampos after1 = f.tellg();
  std::getline(f, l2);
  int ncomp;
  {
    std::istringstream is(l2);
    long long a, b;
    if (ntok(l2) == 2 && (is >> a >> b)) { ncomp = ntok(l1); f.seekg(after1); }  // label-less: l1 = names, l2 = "nElts nodesPerElt"
    else ncomp = ntok(l2);                                                        // l1 = label, l2 = names
  }
  long long nElts = 0, npe = 0;
  f >> nElts >> npe;
  std::string fabhdr;
  std::getline(f, fabhdr);  // rest of the "nElts nodesPerElt" line
  std::getline(f, fabhdr);  // FAB (...)((0,0,0) (N-1,0,0) (0,0,0)) ncomp
  const size_t p1 = fabhdr.rfind(") ("), p0 = fabhdr.rfind("((0,0,0) (");
  if (p0 == std::string::npos || p1 == std::string::npos) pa::Abort("isoFile: cannot parse the node FAB header");
  const long long N = std::atoll(fabhdr.c_str() + p0 + 10) + 1;
  if (N <= 0 || ncomp < 3) pa::Abort("isoFile: no nodes / fewer than 3 coordinates");
  std::vector<double> nodes((size_t)N * ncomp), locs((size_t)N * 3);
  f.read((char*)nodes.data(), sizeof(double) * nodes.size());
  if (!f) pa::Abort("isoFile: truncated node data");
  for (long long i = 0; i < N; ++i)
    for (int d = 0; d < 3; ++d) locs[(size_t)i * 3 + d] = nodes[(size_t)i * ncomp + d];
  return locs;
}

int main(int argc, char** argv) {
  pa::ParmParse pp(argc, argv);
  if (argc < 2) {
    std::cerr << "usage:\n" << argv[0] << " infile=<plotfilename> isoFile=<mef> | seedLoc=<x y z> | seedRakeNum=<n> seedRakeL=<x y z> seedRakeR=<x y z>\n";
    return 1;
  }
  std::string infile;
  pp.get("infile", infile);
  pa::PlotfileHeader H = pa::read_header(infile);
  const char* vnames[3] = {"x_velocity", "y_velocity", "z_velocity"};
  int vc[3];
  for (int d = 0; d < 3; ++d) {
    vc[d] = H.comp(vnames[d]);
    if (vc[d] < 0) pa::Abort(std::string("Variable not found in the plotfile: ") + vnames[d]);
  }
  int nGrow = 3, Nsteps = 50;
  double hRK = 0.1;
  pp.query("nGrow", nGrow);
  if (nGrow < 1) pa::Abort("Assertion `nGrow>=1' failed");
  pp.query("Nsteps", Nsteps);
  pp.query("hRK", hRK);
  if (!(hRK >= 0 && hRK <= 0.5)) pa::Abort("Assertion `hRK>=0 && hRK<=0.5' failed");
  // seeds (partStream.cpp:10-117): exactly one of the forms
  const int nc = pp.countval("oneSeedPerCell"), ni = pp.countval("isoFile"), ns = pp.countval("seedLoc"), nrL = pp.countval("seedRakeL"),
            nrR = pp.countval("seedRakeR");
  if (!((nc > 0) ^ ((ni > 0) ^ ((ns > 0) ^ ((nrL > 0) && nrR > 0))))) pa::Abort("Assertion `(nc>0) ^ ((ni>0) ^ ((ns>0) ^ ((nrL>0) && nrR>0)))' failed");
  std::vector<double> locs;
  if (nc > 0) {  // partStream.cpp:21-61
    for (int lev = 0; lev < H.nlev; ++lev) {
      const auto& L = H.lev[lev];
      double dx[3];
      for (int d = 0; d < 3; ++d) dx[d] = (H.prob_hi[d] - H.prob_lo[d]) / (double)(L.domain.hi[d] - L.domain.lo[d] + 1);
      for (const pa::Box3& B : L.boxes) {
        const int tag[3] = {0, 50, 107};
        bool has = true;
        for (int d = 0; d < 3; ++d) has = has && tag[d] >= B.lo[d] && tag[d] <= B.hi[d];
        if (!has) continue;
        for (int k = B.lo[2]; k <= B.hi[2]; ++k)
          for (int j = B.lo[1]; j <= B.hi[1]; ++j)
            for (int i = B.lo[0]; i <= B.hi[0]; ++i) {
              bool covered = false;
              if (lev + 1 < H.nlev)
                for (const pa::Box3& F : H.lev[lev + 1].boxes) {
                  // coarsen(F, 2) contains (i, j, k)?  (floor division: indices are non-negative inside the domain)
                  auto cdiv = [](int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); };
                  if (i >= cdiv(F.lo[0]) && i <= cdiv(F.hi[0]) && j >= cdiv(F.lo[1]) && j <= cdiv(F.hi[1]) && k >= cdiv(F.lo[2]) && k <= cdiv(F.hi[2])) { covered = true; break; }
                }
              if (covered) continue;
              locs.push_back(H.prob_lo[0] + (i + 0.5) * dx[0]);
              locs.push_back(H.prob_lo[1] + (j + 0.5) * dx[1]);
              locs.push_back(H.prob_lo[2] + (k + 0.5) * dx[2]);
            }
      }
    }
  } else if (ni > 0) {
    std::string isoFile;
    pp.get("isoFile", isoFile);
    std::cerr << "Reading isoFile... " << isoFile << std::endl;
    locs = read_mef_nodes(isoFile);
  } else if (ns > 0) {
    std::vector<double> loc;
    if (!pp.queryarr("seedLoc", loc, 0, 3)) pa::Abort("seedLoc not found");
    locs = loc;
  } else {
    int seedRakeNum = 0;
    pp.get("seedRakeNum", seedRakeNum);
    if (seedRakeNum < 2) pa::Abort("Assertion `seedRakeNum >= 2' failed");
    std::vector<double> L, R;
    if (!pp.queryarr("seedRakeL", L, 0, 3)) pa::Abort("seedRakeL not found");
    if (!pp.queryarr("seedRakeR", R, 0, 3)) pa::Abort("seedRakeR not found");
    for (int i = 0; i < seedRakeNum; ++i)
      for (int d = 0; d < 3; ++d) locs.push_back(L[d] + (i / double(seedRakeNum - 1)) * (R[d] - L[d]));
  }
  const long long nseed = (long long)locs.size() / 3;

  pa::Team team(pp);
  const int Nlev = H.nlev;
  const int is_per[3] = {0, 0, 0};
  if (team.n > 1) std::cout << "Lines dealt to " << team.n << " GPUs, transport: " << team.transport << std::endl;
  std::vector<pa::HostMF> hv(Nlev);
  for (int lev = 0; lev < Nlev; ++lev) {
    hv[lev].define(H.lev[lev].boxes, 3, nGrow);
    for (int d = 0; d < 3; ++d) pa::read_comp(H, lev, vc[d], hv[lev], d);
  }
  const auto& Lf = H.lev[Nlev - 1];
  const double dxf = (H.prob_hi[0] - H.prob_lo[0]) / (double)(Lf.domain.hi[0] - Lf.domain.lo[0] + 1);
  const double dt = hRK * dxf;  // partStream.cpp:187
  std::vector<double> pos((size_t)(2 * nseed) * Nsteps * 3);
  team.run([&](int r) {
    pa::Ctx& ctx = *team.ctx[r];
    // the WHOLE vector field on every rank (unsharded levels: ghost fills are local); the seeds are what is dealt
    std::vector<std::unique_ptr<pa::DevLevel>> dl;
    std::vector<std::unique_ptr<pa::DevMF>> dv;
    std::vector<pa_mf*> v;
    for (int lev = 0; lev < Nlev; ++lev) {
      dl.emplace_back(new pa::DevLevel(ctx, H.lev[lev].boxes, H.lev[lev].domain, is_per, H.prob_lo, H.prob_hi));
      dv.emplace_back(new pa::DevMF(ctx, *dl.back(), 3, nGrow));
      ctx.check(pa_mf_upload(ctx.h, dv.back()->h, hv[lev].data.data()));
      ctx.check(pa_fill_boundary(ctx.h, dv[lev]->h, 0, 3, nGrow));
      if (lev > 0) ctx.check(pa_fillpatch_two_levels(ctx.h, dv[lev]->h, dv[lev - 1]->h, 0, 3, nGrow, 2, 0));  // PCInterp
      v.push_back(dv[lev]->h);
    }
    ctx.check(pa_sync(ctx.h));
    if (pa_bc_errors(ctx.h) != 0) pa::Abort("FillPatchTwoLevels: fine grids are not properly nested in the coarse level");
    const long long s0 = nseed * r / team.n, s1 = nseed * (r + 1) / team.n, ns_r = s1 - s0;
    const size_t cnt = (size_t)(2 * ns_r) * Nsteps * 3;
    void* dpos = ns_r > 0 ? pa_device_malloc(ctx.h, (int64_t)cnt * 8) : nullptr;
    if (ns_r > 0 && !dpos) pa::Abort(pa_last_error(ctx.h));
    int32_t nred = 0;
    if (pa_stream_trace_ranks(ctx.h, Nlev, v.data(), 0, ns_r, ns_r > 0 ? locs.data() + 3 * s0 : nullptr, Nsteps, dt, (double*)dpos, &nred, team.n > 1))
      pa::Abort(std::string("bad RK: ") + pa_last_error(ctx.h));
    if (ns_r > 0) {
      ctx.check(pa_memcpy_d2h(ctx.h, pos.data() + (size_t)(2 * s0) * Nsteps * 3, dpos, (int64_t)cnt * 8));
      pa_device_free(ctx.h, dpos);
    }
  });
  // partStream.cpp:197-199 writes the lines a second time as an AMReX particle plotfile ("Writing paticles to junkPlt",
  // ParticleContainer::WritePlotFile).  That binary format lives in AMReX, which is not part of the reference tree, and cannot
  // be restated from a source here -- said on stdout where the reference announces the file, instead of writing a guess.
  std::cout << "Not writing particles to junkPlt (AMReX particle-plotfile format: not available in this build; the line points are all in tec.dat)" << std::endl;
  const std::string tecfile = "tec.dat";
  std::cout << "Writing streamlines in Tecplot ascii format to " << tecfile << std::endl;
  ::mkdir(tecfile.c_str(), 0755);
  if (nseed > 0) {  // StreamPC.cpp:340-370
    std::ofstream ofs(tecfile + "/str_00000.dat");
    if (!ofs) pa::Abort("Unable to create " + tecfile + "/str_00000.dat");
    ofs << "VARIABLES = X Y Z" << '\n';
    for (long long p = 0; p < 2 * nseed; ++p) {
      ofs << "ZONE I=1 J=" << Nsteps << " k=1 FORMAT=POINT\n";
      for (int j = 0; j < Nsteps; ++j) {
        for (int d = 0; d < 3; ++d) ofs << pos[((size_t)p * Nsteps + j) * 3 + d] << " ";
        ofs << '\n';
      }
    }
  }
  return 0;
}
