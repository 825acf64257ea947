// isosurface3d -- drop-in for PeleAnalysis Src/isosurface.cpp (3-D marching cubes on the AMR dual
// grid, MEF output) on MI355X.
//   isosurface3d.ex infile=<plt> [isoVal=1090] [isoCompName=temp] [comps="i j" | sComp=0 nComp=1] [finestLevel=<n>]
//       [is_per="0 0 0"] [outfile_base=<infile>_<comp>_<isoVal>] [writeSurf=1] [computeArea=0] [verbose=0]
// Per FAB the cube classification, ballot/prefix-sum compaction, edge interpolation and triangle
// emission run on the GPU (libpeleanalysis_amd); the global node/element sets are merged on the
// host in FAB order, which reproduces the reference's node numbering on one rank.
//       [nGrow=1] [rm_external_elements=1] [build_distance_function=0 [dmax=<dx0>] [outfile=distance]]
// build_distance_function (isosurface.cpp:1361-1381, 1595-1655, 1731-1748): per FAB the marching-cubes
// triangles of the box grown by nGrow[lev] = int(dmax*1.0000001/dx_lev) feed the GPU distance function
// (pa_sdf_level_set3 = Tools/SDFGen make_level_set3, batched over all FABs of a level), signed with
// the iso component and clipped at dmax; written as a one-component plotfile "distance".
//       [surfFormat=MEF|XDMF] [surface_is_large=0 [chunk_size=32768] [tmpFile=isoTEMPFILE]] [ngpus=<n> [gpu_share=0|1]]
// ngpus=<n> (pa_team.h): the FABs of every level are dealt to n ranks as the reference's MPI ranks own them
// (isosurface.cpp:1441); every rank polygonises its FABs and the fragments are merged in BoxArray order -- the 1-rank
// ordering against which connectivity is defined (the reference gathers per-rank node / element lists to the
// I/O rank and re-uniquifies there, :932-1037, :1838-1878, so ITS node numbering depends on the rank layout) -- so the
// surface file is byte-identical for every n.  collate is accepted and has no effect: the merged surface is always
// written by the one process.  build_distance_function: every rank computes the grids of its own FABs (one make_level_set3 per FAB on the
// FAB's own triangles: no cross-rank data), the distance plotfile is assembled on the host.
// Periodic directions behave as in the reference: ghost cells behind a periodic face keep the coordinates of the
// cells they image (isosurface.cpp:1469 "bad data in periodic directions"; the shift back at :1483-1507 never fires).
#include "../common/pa_team.h"
#include "../common/pa_isomerge.h"
#include <chrono>
#include <future>

int main(int argc, char** argv) {
  pa::ParmParse pp(argc, argv);
  if (argc < 2 || pp.contains("help")) {
    std::cerr << "usage:\n" << argv[0] << " infile=<plotfilename> isoCompName=<name> isoVal=<v> [comps=<list>] [finestLevel=<n>]\n";
    return 1;
  }
  int verbose = 0;
  pp.query("verbose", verbose);
  std::string infile;
  pp.get("infile", infile);
  if (infile.empty()) pa::Abort("Plotfile not specified, Use infile=");
  pa::PlotfileHeader H = pa::read_header(infile, 3, /* any_ratio: isosurface.cpp:1472,1518,1543 take the file's */ true);
  double isoVal = 1090;
  pp.query("isoVal", isoVal);
  std::string isoCompName = "temp";
  pp.query("isoCompName", isoCompName);
  std::vector<int> pltComps;
  if (int nc = pp.countval("comps")) {
    pp.queryarr("comps", pltComps, 0, nc);
  } else {
    int sComp = 0, nComp = 1;
    pp.query("sComp", sComp);
    pp.query("nComp", nComp);
    for (int i = 0; i < nComp; ++i) pltComps.push_back(sComp + i);
  }
  int isoComp = -1;
  for (size_t i = 0; i < pltComps.size(); ++i) {
    if (pltComps[i] < 0 || pltComps[i] >= (int)H.names.size()) pa::Abort("At least one of the components requested is not in pltfile");
    if (H.names[pltComps[i]] == isoCompName) isoComp = (int)i;
  }
  if (isoComp < 0) pa::Abort("isoCompName not in list of variables to read in");
  const int nComp = (int)pltComps.size(), nc = 3 + nComp;
  int finestLevel = H.nlev - 1;
  pp.query("finestLevel", finestLevel);
  finestLevel = std::min(finestLevel, H.nlev - 1);
  const int Nlev = finestLevel + 1;
  int build_distance_function = 0, rm_external_elements = 1;
  pp.query("rm_external_elements", rm_external_elements);
  pp.query("build_distance_function", build_distance_function);
  // ghost layers per level (isosurface.cpp:1368-1382).  Quirk kept: the reference computes the level's
  // cell size from probSize()[lev] -- the domain length of DIRECTION lev -- over the level's x extent.
  std::vector<int> nGrow(Nlev, 1);
  double dmax = (H.prob_hi[0] - H.prob_lo[0]) / (double)(H.lev[0].domain.hi[0] - H.lev[0].domain.lo[0] + 1);
  if (build_distance_function) {
    pp.query("dmax", dmax);
    std::cout << "dmax: " << dmax << std::endl;
    for (int lev = 0; lev < Nlev; ++lev) {
      const int dl = std::min(lev, 2);
      const double dxL = (H.prob_hi[dl] - H.prob_lo[dl]) / (double)(H.lev[lev].domain.hi[0] - H.lev[lev].domain.lo[0] + 1);
      nGrow[lev] = (int)(dmax * (1.0000001) / dxL);
      if (nGrow[lev] < 1) pa::Abort("dmax is smaller than a cell of level " + std::to_string(lev) + ": no ghost layer to polygonise");
    }
  } else {
    pp.query("nGrow", nGrow[0]);
    if (nGrow[0] < 1) pa::Abort("nGrow must be at least 1");
    for (int lev = 1; lev < Nlev; ++lev) nGrow[lev] = nGrow[0];
  }
  std::vector<int> is_per(3, 0);
  pp.queryarr("is_per", is_per, 0, 3);

  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double strt_time_surf = now();
  double t_ctx = 0, t_host = 0, t_up = 0, t_fill = 0, t_mc = 0, t_d2h = 0, t_merge = 0;  // verbose: where "Compute Surface" goes
  double io_time = 0.0;  // plotfile reads (isosurface.cpp:1388-1415); everything else up to the merge is "Compute Surface"
  // the HIP context comes up on a second thread (~0.25 s) while this one reads the plotfile
  double tq = now();
  pa::AsyncTeam ateam(pp);
  std::vector<pa::HostMF> host(Nlev);
  for (int lev = 0; lev < Nlev; ++lev) {
    // host side: the mapped plotfile components only (ghost cells -666: gstate.setVal(-666), isosurface.cpp:1512);
    // the three coordinate components are written on the device (isosurface.cpp:1458-1465)
    tq = now();
    host[lev].define(H.lev[lev].boxes, nComp, nGrow[lev], -666.0);
    t_host += now() - tq;
    const double t_io = now();
    for (int n = 0; n < nComp; ++n) pa::read_comp(H, lev, pltComps[n], host[lev], n);
    io_time += now() - t_io;
  }
  tq = now();
  pa::Team& team = ateam.get();
  t_ctx = now() - tq;  // what was not hidden behind the reads
  if (team.n > 1) std::cout << "Boxes distributed over " << team.n << " GPUs, transport: " << team.transport << std::endl;
  const std::vector<std::vector<int32_t>> owner = pa::shard_levels(H, Nlev, team.n);
  pa::IsoMerger merger(nc);
  std::vector<pa::HostMF> hdist(build_distance_function ? Nlev : 0);
  if (build_distance_function)  // the whole level's distance multifab; every rank fills the FABs it owns (one grid per FAB: no cross-rank data)
    for (int lev = 0; lev < Nlev; ++lev) hdist[lev].define(H.lev[lev].boxes, 1, nGrow[lev]);
  // one FAB's fragment into the global node / element sets: elements with a vertex whose edge is not inside the valid box
  // grown by 1 are dropped, with those vertices (isosurface.cpp:1657-1682; a no-op when nGrow = 1), then isosurface.cpp:1687-1726
  std::vector<double> hv;
  std::vector<int32_t> ht, hk;
  auto merge_box = [&](const pa::Box3& B, int ng, int64_t nv, int64_t nt, const double* pv, const int32_t* pk, const int32_t* pt) {
    hv.assign(pv, pv + nv * nc);
    ht.assign(pt, pt + nt * 3);
    hk.assign(pk, pk + nv * 6);
    long long nvk = nv, ntk = nt;
    if (rm_external_elements && ng > 1) {
      std::vector<int32_t> remap((size_t)nv);
      nvk = 0;
      for (int64_t q = 0; q < nv; ++q) {
        bool in = true;
        for (int d = 0; d < 3; ++d) {
          const int a = hk[(size_t)q * 6 + d], c = hk[(size_t)q * 6 + 3 + d];
          in = in && a >= B.lo[d] - 1 && a <= B.hi[d] + 1 && c >= B.lo[d] - 1 && c <= B.hi[d] + 1;
        }
        remap[(size_t)q] = in ? (int32_t)nvk : -1;
        if (in) {
          if (nvk != q) std::copy(hv.begin() + q * nc, hv.begin() + (q + 1) * nc, hv.begin() + nvk * nc);
          ++nvk;
        }
      }
      ntk = 0;
      for (int64_t t = 0; t < nt; ++t) {
        const int32_t a = remap[(size_t)ht[(size_t)t * 3]], c = remap[(size_t)ht[(size_t)t * 3 + 1]], e = remap[(size_t)ht[(size_t)t * 3 + 2]];
        if (a < 0 || c < 0 || e < 0) continue;
        ht[(size_t)ntk * 3] = a; ht[(size_t)ntk * 3 + 1] = c; ht[(size_t)ntk * 3 + 2] = e;
        ++ntk;
      }
    }
    merger.add(hv.data(), nvk, ht.data(), ntk);
  };
  // per rank and level: the fragments of its FABs as they came off the device (ngpus > 1: merged afterwards in BoxArray order)
  struct LevFrag { std::vector<int> gids; std::vector<int64_t> nvb, ntb; std::vector<double> hva; std::vector<int32_t> hta, hka; double* dv = nullptr; int32_t *dk = nullptr, *dt = nullptr; };
  std::vector<std::vector<LevFrag>> frags(team.n, std::vector<LevFrag>(Nlev));
  std::vector<void*> mc_blocks(team.n, nullptr);  // per rank: the ONE device block that holds the surfaces of all its levels (pa_mc_hierarchy_fine)
  // no per-FAB trimming, no distance function: the node / element sets are built on the device (pa_iso_merge) from the
  // fragments where pa_mc_level_fine left them (ngpus > 1: the other ranks' fragments are first copied to GPU 0); they are
  // only downloaded if the library hands the merge back (PA_ISO_HOST_MERGE=1 forces the host path)
  bool dev_merge = !build_distance_function && !std::getenv("PA_ISO_HOST_MERGE");
  for (int lev = 0; lev < Nlev; ++lev) dev_merge = dev_merge && !(rm_external_elements && nGrow[lev] > 1);
  team.run([&](int r) {
  pa::Ctx& ctx = *team.ctx[r];
  const bool lead = r == 0;  // the phase timers are rank 0's
  std::vector<std::unique_ptr<pa::DevLevel>> dl;
  std::vector<std::unique_ptr<pa::DevMF>> dst;
  std::vector<pa::Share> shares;
  std::vector<pa::HostMF> hloc(Nlev);
  double tq = now();
  // Round 5: the state holds the plotfile components only.  The three coordinate components isosurface.cpp:1458-1478 stores,
  // FillBoundaries and FillPatches are a function of the cell index and of which level covers the cell, so the vertex kernels form
  // them in registers (pa_mc_hierarchy_xyz; same doubles), and the ghost cells of all levels are filled in two launches
  // (pa_fill_ghosts_hierarchy).  Hierarchies whose levels differ in refinement ratio keep the stored-coordinate state and a call
  // per level (xyz_state = false; PA_ISO_XYZ=0 forces it for A/B).
  bool xyz_state = !(std::getenv("PA_ISO_XYZ") && !std::atoi(std::getenv("PA_ISO_XYZ")));
  for (int lev = 1; lev + 1 < Nlev; ++lev) xyz_state = xyz_state && H.ref_ratio[lev] == H.ref_ratio[0];
  const int nst = xyz_state ? nComp : nc, iso_st = xyz_state ? isoComp : 3 + isoComp;  // components of the state, index of the iso field in it
  for (int lev = 0; lev < Nlev; ++lev) {
    const auto& L = H.lev[lev];
    const int ng = nGrow[lev];
    tq = now();
    shares.emplace_back(L.boxes, owner[lev], r);
    dl.emplace_back(new pa::DevLevel(ctx, L.boxes, L.domain, is_per.data(), H.prob_lo, H.prob_hi, &owner[lev], r, team.n));
    dst.emplace_back(new pa::DevMF(ctx, *dl.back(), nst, ng));
    if (team.n > 1) shares.back().gather(host[lev], hloc[lev]);
    if (xyz_state) {
      ctx.check(pa_mf_upload(ctx.h, dst.back()->h, (team.n > 1 ? hloc[lev] : host[lev]).data.data()));
    } else {
      pa::DevMF dfield(ctx, *dl.back(), nComp, ng);
      ctx.check(pa_mf_upload(ctx.h, dfield.h, (team.n > 1 ? hloc[lev] : host[lev]).data.data()));
      ctx.check(pa_iso_coords_level(ctx.h, dst.back()->h, 0));
      ctx.check(pa_mf_copy(ctx.h, dfield.h, 0, dst.back()->h, 3, nComp, ng));
      ctx.check(pa_sync(ctx.h));
    }
    if (lead) t_up += now() - tq;
    if (xyz_state) continue;
    tq = now();
    if (lead) std::cout << "FillPatching the grown structures at level " << lev << "..." << std::endl;
    ctx.check(pa_fill_boundary(ctx.h, dst[lev]->h, 0, nc, ng));
    if (lev > 0) ctx.check(pa_fillpatch_two_levels(ctx.h, dst[lev]->h, dst[lev - 1]->h, 0, nc, ng, H.ref_ratio[lev - 1], 0));  // PCInterp, the file's ratio (isosurface.cpp:1472,1518)
    if (lead) std::cout << "...done FillPatching the grown structures at level " << lev << "..." << std::endl;
    ctx.check(pa_sync(ctx.h));
    if (lead) t_fill += now() - tq;
  }
  if (xyz_state) {  // isosurface.cpp:1468-1524 for ALL levels: FillBoundary, then FillPatchTwoLevels with PCInterp, one launch each
    tq = now();
    std::vector<pa_mf*> hm;
    std::vector<int32_t> hg;
    for (int lev = 0; lev < Nlev; ++lev) {
      if (lead) std::cout << "FillPatching the grown structures at level " << lev << "..." << std::endl;
      hm.push_back(dst[lev]->h);
      hg.push_back(nGrow[lev]);
    }
    ctx.check(pa_fill_ghosts_hierarchy(ctx.h, Nlev, hm.data(), 0, nComp, hg.data(), Nlev > 1 ? H.ref_ratio[0] : 2, 0, 0));
    for (int lev = 0; lev < Nlev; ++lev)
      if (lead) std::cout << "...done FillPatching the grown structures at level " << lev << "..." << std::endl;
    ctx.check(pa_sync(ctx.h));
    if (lead) t_fill += now() - tq;
  }
  ctx.check(pa_sync(ctx.h));
  if (pa_bc_errors(ctx.h) != 0) pa::Abort("FillPatchTwoLevels: fine grids are not properly nested in the coarse level");

  // The MFIter loops of isosurface.cpp:1531-1592 for ALL levels in one call: the cell passes and counts of every level back to
  // back, one read-back of the per-FAB counts, one pooled allocation for the surfaces, no host round trip per level.
  std::vector<std::vector<pa_box>> loops_all(Nlev);
  std::vector<std::vector<int64_t>> nvb_all(Nlev), ntb_all(Nlev);
  std::vector<double*> dv_all(Nlev, nullptr);
  std::vector<int32_t*> dk_all(Nlev, nullptr), dt_all(Nlev, nullptr);
  {
    tq = now();
    std::vector<const pa_mf*> sts(Nlev);
    std::vector<int32_t> fmask(Nlev, 0);
    std::vector<const pa_box*> lp(Nlev);
    std::vector<int64_t*> pnv(Nlev), pnt(Nlev);
    int ratio = 2;
    for (int lev = 0; lev < Nlev; ++lev) {
      const int ng = nGrow[lev];
      const std::vector<pa::Box3>& bxs = shares[lev].boxes;
      const pa::Box3& dom = H.lev[lev].domain;
      // fine-covered mask (isosurface.cpp:1540-1563; all 1 when building the distance function, :1542)
      fmask[lev] = (lev < finestLevel && !build_distance_function) ? 1 : 0;
      if (fmask[lev]) ratio = H.ref_ratio[lev];
      loops_all[lev].resize(std::max<size_t>(bxs.size(), 1));
      for (size_t b = 0; b < bxs.size(); ++b)  // base points: (grown box & domain grown in the periodic directions), high side - 1 (isosurface.cpp:1566-1569)
        for (int d = 0; d < 3; ++d) {
          const int pg = is_per[d] ? ng : 0;  // growPeriodicDomain (isosurface.cpp:1437)
          loops_all[lev][b].lo[d] = std::max(bxs[b].lo[d] - ng, dom.lo[d] - pg);
          loops_all[lev][b].hi[d] = std::min(bxs[b].hi[d] + ng, dom.hi[d] + pg) - 1;
        }
      nvb_all[lev].assign(std::max<size_t>(bxs.size(), 1), 0);
      ntb_all[lev].assign(std::max<size_t>(bxs.size(), 1), 0);
      sts[lev] = dst[lev]->h; lp[lev] = loops_all[lev].data(); pnv[lev] = nvb_all[lev].data(); pnt[lev] = ntb_all[lev].data();
    }
    bool uniform = true;  // one ratio argument: levels with different ratios fall back to a call per level
    for (int lev = 0; lev + 1 < Nlev; ++lev) uniform = uniform && (!fmask[lev] || H.ref_ratio[lev] == ratio);
    if (xyz_state) {
      ctx.check(pa_mc_hierarchy_xyz(ctx.h, Nlev, sts.data(), fmask.data(), Nlev > 1 ? H.ref_ratio[0] : 2, lp.data(), iso_st, isoVal, pnv.data(), pnt.data(), dv_all.data(),
                                    dk_all.data(), dt_all.data(), &mc_blocks[r]));
    } else if (uniform) {
      ctx.check(pa_mc_hierarchy_fine(ctx.h, Nlev, sts.data(), fmask.data(), ratio, lp.data(), 3 + isoComp, isoVal, pnv.data(), pnt.data(), dv_all.data(), dk_all.data(),
                                     dt_all.data(), &mc_blocks[r]));
    } else {
      for (int lev = 0; lev < Nlev; ++lev)
        if (!shares[lev].boxes.empty())
          ctx.check(pa_mc_level_fine(ctx.h, dst[lev]->h, fmask[lev] ? dl[lev + 1]->h : nullptr, fmask[lev] ? H.ref_ratio[lev] : 2, lp[lev], 3 + isoComp, isoVal, pnv[lev], pnt[lev],
                                     &dv_all[lev], &dk_all[lev], &dt_all[lev]));
    }
    ctx.check(pa_sync(ctx.h));
    if (lead) t_mc += now() - tq;
  }
  for (int lev = 0; lev < Nlev; ++lev) {
    const int ng = nGrow[lev];
    struct { const std::vector<pa::Box3>& boxes; const pa::Box3& domain; } L{shares[lev].boxes, H.lev[lev].domain};  // this rank's FABs
    double* base = pa_mf_data(dst[lev]->h);
    // distance function: one grid per FAB that holds triangles, run as one batch per level
    std::unique_ptr<pa::DevMF> ddist;
    std::vector<pa_sdf_grid> grids;
    std::vector<size_t> grid_box;
    std::vector<void*> grid_bufs;
    std::vector<char> has_elts(L.boxes.size(), 0);
    double dxf[3];
    for (int d = 0; d < 3; ++d) dxf[d] = (H.prob_hi[d] - H.prob_lo[d]) / (double)(L.domain.hi[d] - L.domain.lo[d] + 1);
    pa::HostMF hd_loc;  // this rank's FABs of the distance multifab (ngpus = 1: the level's multifab itself)
    std::vector<int64_t> lsoff, lscs;  // offsets of this rank's nc-component device multifab
    if (build_distance_function) {
      if (team.n > 1) hd_loc.define(L.boxes, 1, ng);
      ddist.reset(new pa::DevMF(ctx, *dl[lev], 1, ng));
      std::vector<int32_t> b6(6 * L.boxes.size());
      for (size_t i = 0; i < L.boxes.size(); ++i)
        for (int d = 0; d < 3; ++d) { b6[6 * i + d] = L.boxes[i].lo[d]; b6[6 * i + 3 + d] = L.boxes[i].hi[d]; }
      lsoff.resize(L.boxes.size());
      lscs.resize(L.boxes.size());
      pa_mf_layout((int)L.boxes.size(), b6.data(), nst, ng, lsoff.data(), lscs.data());
    }
    pa::HostMF& hd = team.n > 1 ? hd_loc : (build_distance_function ? hdist[lev] : hd_loc);
    pa::HostMF& hsrc = team.n > 1 ? hloc[lev] : host[lev];  // the plotfile components of this rank's FABs
    // (marching cubes of every level ran above: pa_mc_hierarchy_fine)
    const size_t nb = L.boxes.size();
    std::vector<int64_t> nvb(nvb_all[lev].begin(), nvb_all[lev].begin() + nb), ntb(ntb_all[lev].begin(), ntb_all[lev].begin() + nb);
    double* dv = dv_all[lev];
    int32_t *dk = dk_all[lev], *dt = dt_all[lev];
    tq = now();
    int64_t nvt = 0, ntt = 0;
    for (size_t b = 0; b < nb; ++b) { nvt += nvb[b]; ntt += ntb[b]; }
    LevFrag& F = frags[r][lev];
    F.gids = shares[lev].gids; F.nvb = nvb; F.ntb = ntb;
    std::vector<double>& hva = F.hva;
    std::vector<int32_t>&hta = F.hta, &hka = F.hka;
    F.dv = dv; F.dk = dk; F.dt = dt;
    if (dev_merge) continue;  // the level's surface stays on the device until pa_iso_merge
    hva.resize((size_t)(nvt * nc)); hta.resize((size_t)(ntt * 3)); hka.resize((size_t)(nvt * 6));
    if (nvt > 0) {
      ctx.check(pa_memcpy_d2h(ctx.h, hva.data(), dv, nvt * nc * 8));
      ctx.check(pa_memcpy_d2h(ctx.h, hka.data(), dk, nvt * 6 * 4));
    }
    if (ntt > 0) ctx.check(pa_memcpy_d2h(ctx.h, hta.data(), dt, ntt * 3 * 4));
    if (lead) t_d2h += now() - tq;
    tq = now();
    void* dx3 = nullptr;
    if (build_distance_function && nvt > 0) {  // vertList: Vec3f(loc) rounds to float (isosurface.cpp:1598-1611)
      std::vector<float> xf((size_t)nvt * 3);
      for (int64_t q = 0; q < nvt; ++q)
        for (int d = 0; d < 3; ++d) xf[(size_t)q * 3 + d] = (float)hva[(size_t)q * nc + d];
      dx3 = pa_device_malloc(ctx.h, nvt * 3 * 4);
      if (!dx3) pa::Abort(pa_last_error(ctx.h));
      ctx.check(pa_memcpy_h2d(ctx.h, dx3, xf.data(), nvt * 3 * 4));
      grid_bufs.push_back(dx3);
    }
    int64_t vo = 0, to = 0;
    for (size_t b = 0; b < nb; vo += nvb[b], to += ntb[b], ++b) {
      const pa::Box3& B = L.boxes[b];
      const int64_t nv = nvb[b], nt = ntb[b];
      if (nt <= 0) continue;
      has_elts[b] = 1;
      if (build_distance_function) {
        // vertList / faceList of this FAB BEFORE trimming (isosurface.cpp:1598-1626)
        pa::Box3 g{{B.lo[0] - ng, B.lo[1] - ng, B.lo[2] - ng}, {B.hi[0] + ng, B.hi[1] + ng, B.hi[2] + ng}};
        pa_sdf_grid G;
        G.ntri = nt; G.nvert = nv;
        void* dphi = pa_device_malloc(ctx.h, g.numPts() * 4);
        if (!dphi) pa::Abort(pa_last_error(ctx.h));
        G.tri = (const uint32_t*)(dt + 3 * to);  // local vertex ids in vertCache order = ptID (isosurface.cpp:1602-1611)
        G.x = (const float*)dx3 + 3 * vo;
        for (int d = 0; d < 3; ++d) {
          G.origin[d] = (float)(H.prob_lo[d] + g.lo[d] * dxf[d]);  // local_origin: the box's low NODE (quirk: not the cell centre)
          G.n[d] = g.hi[d] - g.lo[d] + 1;
        }
        G.dx = (float)dxf[0];
        G.phi = (float*)dphi;
        grids.push_back(G);
        grid_box.push_back(b);
        grid_bufs.push_back(dphi);
      }
      if (team.n == 1) merge_box(B, ng, nv, nt, hva.data() + vo * nc, hka.data() + vo * 6, hta.data() + to * 3);
    }
    if (lead) t_merge += now() - tq;
    // (dv, dk, dt live in the rank's pooled block -- or, levels of different ratios, in a block per level -- freed after the loop / the merge)
    if (build_distance_function) {
      ctx.check(pa_sdf_level_set3(ctx.h, (int)grids.size(), grids.data(), 1));
      double* dbase = pa_mf_data(ddist->h);
      for (size_t q = 0; q < grids.size(); ++q) {  // isosurface.cpp:1632-1650
        const size_t b = grid_box[q];
        const pa::Box3& B = L.boxes[b];
        pa_box vb;
        pa_fab fs, fd;
        fs.p = base + lsoff[b]; fs.ncomp = nst; fs.nstride = lscs[b];
        fd.p = dbase + hd.off[b]; fd.ncomp = 1; fd.nstride = hd.cs[b];
        for (int d = 0; d < 3; ++d) { vb.lo[d] = fs.lo[d] = fd.lo[d] = B.lo[d] - ng; vb.hi[d] = fs.hi[d] = fd.hi[d] = B.hi[d] + ng; }
        ctx.check(pa_sdf_signed_fab(ctx.h, vb, grids[q].phi, &fs, iso_st, isoVal, dmax, &fd, 0));
      }
      ctx.check(pa_mf_download(ctx.h, ddist->h, hd.data.data()));
      for (void* p : grid_bufs) pa_device_free(ctx.h, p);
      for (size_t b = 0; b < L.boxes.size(); ++b) {  // FABs without triangles: +-dmax from the first valid cell (isosurface.cpp:1651-1654)
        if (has_elts[b]) continue;
        const pa::Box3& B = L.boxes[b];
        const double v = *hsrc.ptr((int)b, isoComp, B.lo[0], B.lo[1], B.lo[2]) < isoVal ? -dmax : dmax;
        for (int k = B.lo[2] - ng; k <= B.hi[2] + ng; ++k)
          for (int j = B.lo[1] - ng; j <= B.hi[1] + ng; ++j)
            for (int i = B.lo[0] - ng; i <= B.hi[0] + ng; ++i) *hd.ptr((int)b, 0, i, j, k) = v;
      }
      if (team.n > 1) shares[lev].scatter(hd_loc, hdist[lev]);  // disjoint FABs: every rank writes its own
    }
  }
  if (!dev_merge) {  // the surfaces are on the host now
    if (mc_blocks[r]) pa_device_free(ctx.h, mc_blocks[r]);
    else
      for (int lev = 0; lev < Nlev; ++lev) pa_device_free(ctx.h, dv_all[lev]);
    mc_blocks[r] = nullptr;
    for (int lev = 0; lev < Nlev; ++lev) frags[r][lev].dv = nullptr;
  }
  });
  if (team.n > 1 && !dev_merge) {  // BoxArray order = the 1-rank ordering (isosurface.cpp:1531: MFIter over the level's FABs)
    tq = now();
    for (int lev = 0; lev < Nlev; ++lev) {
      std::vector<std::pair<int, int>> where(H.lev[lev].boxes.size());  // global box -> (rank, local index)
      std::vector<std::vector<int64_t>> vo(team.n), to(team.n);
      for (int r = 0; r < team.n; ++r) {
        const LevFrag& F = frags[r][lev];
        vo[r].assign(F.gids.size() + 1, 0); to[r].assign(F.gids.size() + 1, 0);
        for (size_t i = 0; i < F.gids.size(); ++i) { where[F.gids[i]] = {r, (int)i}; vo[r][i + 1] = vo[r][i] + F.nvb[i]; to[r][i + 1] = to[r][i] + F.ntb[i]; }
      }
      for (size_t g = 0; g < where.size(); ++g) {
        const int r = where[g].first, i = where[g].second;
        const LevFrag& F = frags[r][lev];
        if (F.ntb[i] <= 0) continue;
        merge_box(H.lev[lev].boxes[g], nGrow[lev], F.nvb[i], F.ntb[i], F.hva.data() + vo[r][i] * nc, F.hka.data() + vo[r][i] * 6, F.hta.data() + to[r][i] * 3);
      }
    }
    t_merge += now() - tq;
  }
  std::vector<double> dnodes;   // the device merge's results (dev_merge)
  std::vector<int32_t> delts;
  if (dev_merge) {
    tq = now();
    pa::Ctx& ctx = *team.ctx[0];
    // every rank's level blocks on GPU 0 (rank 0's are there already): vertices and triangles, keys are not needed
    std::vector<std::vector<const double*>> bv(team.n, std::vector<const double*>(Nlev, nullptr));
    std::vector<std::vector<const int32_t*>> bt(team.n, std::vector<const int32_t*>(Nlev, nullptr));
    std::vector<void*> copies;
    for (int r = 0; r < team.n; ++r)
      for (int lev = 0; lev < Nlev; ++lev) {
        const LevFrag& F = frags[r][lev];
        int64_t nvt = 0, ntt = 0;
        for (size_t b = 0; b < F.nvb.size(); ++b) { nvt += F.nvb[b]; ntt += F.ntb[b]; }
        if (r == 0 || nvt == 0) { bv[r][lev] = F.dv; bt[r][lev] = F.dt; continue; }
        const int64_t vbytes = (nvt * nc * 8 + 255) / 256 * 256, tbytes = std::max<int64_t>(ntt * 12, 8);
        char* c = (char*)pa_device_malloc(ctx.h, vbytes + tbytes);
        if (!c) pa::Abort(pa_last_error(ctx.h));
        ctx.check(pa_memcpy_d2d(ctx.h, c, F.dv, nvt * nc * 8));
        if (ntt > 0) ctx.check(pa_memcpy_d2d(ctx.h, c + vbytes, F.dt, ntt * 12));
        bv[r][lev] = (const double*)c; bt[r][lev] = (const int32_t*)(c + vbytes);
        copies.push_back(c);
      }
    // insertion order of isosurface.cpp:1531-1726: level by level, FAB by FAB in BoxArray order (= the 1-rank ordering), FABs without elements skipped (:1595)
    struct FragAt { int lev, r; size_t i; int64_t vo, to; };
    std::vector<FragAt> order;
    for (int lev = 0; lev < Nlev; ++lev) {
      std::vector<FragAt> at(H.lev[lev].boxes.size(), FragAt{lev, -1, 0, 0, 0});
      for (int r = 0; r < team.n; ++r) {
        const LevFrag& F = frags[r][lev];
        int64_t vo = 0, to = 0;
        for (size_t i = 0; i < F.gids.size(); vo += F.nvb[i], to += F.ntb[i], ++i) at[F.gids[i]] = FragAt{lev, r, i, vo, to};
      }
      for (const FragAt& a : at)
        if (a.r >= 0 && frags[a.r][lev].ntb[a.i] > 0) order.push_back(a);
    }
    std::vector<pa_iso_frag> fr;
    for (const FragAt& a : order) {
      const LevFrag& F = frags[a.r][a.lev];
      fr.push_back(pa_iso_frag{bv[a.r][a.lev] + a.vo * nc, F.nvb[a.i], bt[a.r][a.lev] + 3 * a.to, F.ntb[a.i]});
    }
    int64_t nn = 0, ne = 0;
    double* dn = nullptr;
    int32_t* de = nullptr;
    const int rc = pa_iso_merge(ctx.h, (int)fr.size(), fr.data(), nc, &nn, &dn, &ne, &de);
    if (rc == 0) {
      dnodes.resize((size_t)(nn * nc));
      delts.resize((size_t)(ne * 3));
      if (nn > 0) ctx.check(pa_memcpy_d2h(ctx.h, dnodes.data(), dn, nn * nc * 8));
      if (ne > 0) ctx.check(pa_memcpy_d2h(ctx.h, delts.data(), de, ne * 12));
      pa_device_free(ctx.h, dn);
      pa_device_free(ctx.h, de);
    } else if (rc == 2) {  // clusters that are not transitive under the tolerance: the sequential rule decides (host)
      if (verbose) std::cout << "  " << pa_last_error(ctx.h) << std::endl;
      dev_merge = false;
      std::vector<double> fv;
      std::vector<int32_t> ft, fk;
      for (const FragAt& a : order) {
        const LevFrag& F = frags[a.r][a.lev];
        const int64_t nv = F.nvb[a.i], nt = F.ntb[a.i];
        fv.resize((size_t)(nv * nc)); ft.resize((size_t)(nt * 3)); fk.assign((size_t)(nv * 6), 0);  // keys: only read by the trimming, which this path excludes
        ctx.check(pa_memcpy_d2h(ctx.h, fv.data(), bv[a.r][a.lev] + a.vo * nc, nv * nc * 8));
        ctx.check(pa_memcpy_d2h(ctx.h, ft.data(), bt[a.r][a.lev] + 3 * a.to, nt * 12));
        merge_box(H.lev[a.lev].boxes[F.gids[a.i]], 1, nv, nt, fv.data(), fk.data(), ft.data());
      }
    } else {
      pa::Abort(pa_last_error(ctx.h));
    }
    for (void* c : copies) pa_device_free(ctx.h, c);
    for (int r = 0; r < team.n; ++r) {
      if (mc_blocks[r]) pa_device_free(team.ctx[r]->h, mc_blocks[r]);
      else
        for (int lev = 0; lev < Nlev; ++lev) pa_device_free(team.ctx[r]->h, frags[r][lev].dv);  // (a block per level: levels of different ratios)
    }
    t_merge += now() - tq;
  }
  if (build_distance_function) {  // isosurface.cpp:1731-1748
    std::string outfile("distance");
    pp.query("outfile", outfile);
    std::vector<pa::Box3> doms;
    std::vector<int> steps;
    for (int lev = 0; lev < Nlev; ++lev) { doms.push_back(H.lev[lev].domain); steps.push_back(H.lev[lev].level_step); }
    pa::write_plotfile(outfile, {"distance"}, doms, H.prob_lo, H.prob_hi, hdist, 0.0 /* isosurface.cpp:1417,1747: the local `Real time = 0` */, steps);
  }
  {  // isosurface.cpp:1756-1771 (one rank: max = min)
    const double surf_time = now() - strt_time_surf - io_time;
    std::cout << "Max Compute Surface time: " << surf_time << '\n' << "Min Compute Surface time: " << surf_time << '\n';
    std::cout << "Max I/O time: " << io_time << '\n' << "Min I/O time: " << io_time << '\n';
    int bench_json = 0;
    pp.query("bench_json", bench_json);
    if (bench_json) {
      long long cells = 0;
      for (int lev = 0; lev < Nlev; ++lev)
        for (auto& B : H.lev[lev].boxes) cells += B.numPts();
      std::cout << "{\"tool\": \"isosurface3d\", \"cells\": " << cells << ", \"phases_s\": {\"read\": " << io_time << ", \"hip_context_wait\": " << t_ctx
                << ", \"host_buffers\": " << t_host << ", \"upload_coords\": " << t_up << ", \"ghost_fill\": " << t_fill << ", \"marching_cubes\": " << t_mc
                << ", \"download\": " << t_d2h << ", \"merge\": " << t_merge << "}}" << std::endl;
    }
    if (verbose)
      std::cout << "  of which: HIP context (not hidden behind the reads) " << t_ctx << ", host buffers " << t_host << ", level tables + upload + coordinates " << t_up << ", ghost fill " << t_fill
                << ", marching cubes " << t_mc << ", download " << t_d2h << ", per-FAB trimming + node/element insertion " << t_merge << '\n';
  }
  const double strt_time_uniq = now();
  if (!dev_merge) merger.finish();
  const std::vector<int32_t> elts = dev_merge ? std::move(delts) : merger.elements();
  const std::vector<double>& nodes = dev_merge ? dnodes : merger.nodes();
  const long long num_nodes = (long long)nodes.size() / nc;
  std::cout << "Uniquify time: " << now() - strt_time_uniq << '\n';  // :1888-1890
  const double strt_time_sout = now();
  int writeSurf = 1, computeArea = 0;
  pp.query("writeSurf", writeSurf);
  pp.query("computeArea", computeArea);
  std::string surfFormat = "MEF";
  pp.query("surfFormat", surfFormat);
  if (surfFormat != "MEF" && surfFormat != "XDMF") pa::Abort("surfFormat must be MEF or XDMF");
  if (computeArea) {  // computed before the element list is released (the reference prints 0 here: quirk Q7)
    const auto& nd = nodes;
    double area = 0;
    for (size_t e = 0; e + 2 < elts.size(); e += 3) {
      const double *a = &nd[(size_t)elts[e] * nc], *b = &nd[(size_t)elts[e + 1] * nc], *c = &nd[(size_t)elts[e + 2] * nc];
      const double ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2], vx = c[0] - a[0], vy = c[1] - a[1], vz = c[2] - a[2];
      const double cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
      area += 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
    }
    std::cout << "Total area = " << area << std::endl;
  }
  if (writeSurf && surfFormat == "XDMF") {  // isosurface.cpp:2135-2229; quirk kept: the default name carries the plotfile TIME, not isoVal
    char buf[72];
    std::snprintf(buf, sizeof buf, "%g", H.time);
    std::string outfile_base = infile + "_" + isoCompName + "_" + std::string(buf);
    pp.query("outfile_base", outfile_base);
    std::vector<std::string> vn;
    for (int n = 0; n < nComp; ++n) vn.push_back(H.names[pltComps[n]]);
    pa::write_xdmf(outfile_base, H.time, isoCompName, isoVal, vn, nodes, elts);
  } else if (writeSurf) {
    std::cout << "...write surface in mef format (mef = Marcs element format)" << std::endl;
    std::cout << "      (Nelts,Nnodes):(" << elts.size() / 3 << ", " << num_nodes << ")" << std::endl;
    std::vector<std::string> vars{"X", "Y", "Z"};
    for (int n = 0; n < nComp; ++n) vars.push_back(H.names[pltComps[n]]);
    char buf[72];
    std::snprintf(buf, sizeof buf, "%g", isoVal);
    std::string outfile_base = infile + "_" + isoCompName + "_" + std::string(buf);
    pp.query("outfile_base", outfile_base);
    // surface_is_large (isosurface.cpp:1919-1999): the reference stages the node data through a file of FABs of at most
    // chunk_size nodes to release its node set before it allocates the final FAB; the MEF is the same either way.  Nothing
    // needs releasing here (nodes are already one array), but the staging file is an observable product of the run (it is
    // never removed), so it is written the same way: Box (0..N-1,0,0) chopped by BoxArray::maxSize(chunk_size), each
    // piece a FAB of nodeSize components whose memory is node-major (:1943-1950).
    int surface_is_large = 0, chunk_size = 32768;
    pp.query("surface_is_large", surface_is_large);
    pp.query("chunk_size", chunk_size);
    if (surface_is_large) {
      if (chunk_size < 1) pa::Abort("chunk_size must be positive");
      std::string tmpFile = "isoTEMPFILE";
      pp.query("tmpFile", tmpFile);
      std::ofstream ost(tmpFile, std::ios::binary);
      if (!ost) pa::Abort("Unable to create " + tmpFile);
      const long long N = num_nodes;
      const long long nparts = (N + chunk_size - 1) / chunk_size, base = nparts ? N / nparts : 0, rem = nparts ? N % nparts : 0;
      if (verbose) std::cout << "  staging vertex data to disk in " << nparts << " chunks..." << std::endl;
      long long lo = 0;
      for (long long q = 0; q < nparts; ++q) {  // BoxArray::maxSize: an even split, the first `rem` pieces one longer
        const long long n = base + (q < rem ? 1 : 0);
        pa::Box3 b{{(int)lo, 0, 0}, {(int)(lo + n - 1), 0, 0}};
        ost << "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))" << pa::box_str(b) << ' ' << nc << "\n";
        ost.write((const char*)(nodes.data() + lo * nc), sizeof(double) * (size_t)(n * nc));
        lo += n;
      }
      if (verbose) std::cout << "  ... data staged." << std::endl;
    }
    std::cout << "  Writing the file..." << std::endl;
    pa::write_mef(outfile_base + ".mef", H.time, vars, nodes, elts);
    std::cout << "            ...done" << std::endl;
  }
  std::cout << "Surface output time: " << now() - strt_time_sout << '\n';  // :2232-2234
  pa::Finish();
}
