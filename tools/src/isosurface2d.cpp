// isosurface2d -- drop-in for the AMREX_SPACEDIM == 2 build of PeleAnalysis Src/isosurface.cpp (isosurface2d.*.ex):
// contour of a 2-D AMR plotfile as line segments (Segmentise, :303-406), merged over levels, written as an MEF file
// with two nodes per element, plus the contour-line assembly (MakeCLines, :1159-1265) and its "Integral:" lines.
//   isosurface2d.ex infile=<plt> isoCompName=<name> isoVal=<v> [comps=<list> | sComp=0 nComp=1] [finestLevel=<n>]
//       [rm_external_elements=1] [nGrow=1] [is_per="0 0"] [writeSurf=1] [surfFormat=MEF|XDMF] [outfile_base=<infile>_<comp>_<isoVal>] [verbose=0]
// The plotfile's plane of cells is handed to the library as boxes with k = 0; the state holds 2 coordinate components
// + the mapped ones; ghost fill, fine-covered mask and the per-FAB loop are the 3-D tool's (tools/src/isosurface.cpp)
// with pa_msq_level_fine in place of pa_mc_level_fine.  build_distance_function aborts as in the reference (:1364-1366);
// surfFormat=XDMF writes the Polyline / XY variant of the 3-D tool's file.
#include "../common/pa_device.h"
#include "../common/pa_isomerge.h"
#include <chrono>
#include <list>

namespace {
// MakeCLines (isosurface.cpp:1159-1265) on 0-based segments; quirk kept: the seed segment of every search that starts a
// new line is popped and never stored
typedef std::pair<int, int> Seg;
typedef std::list<Seg> Line;
std::list<Line> make_clines(const std::vector<int32_t>& elts0) {
  Line segList;
  for (size_t e = 0; e + 1 < elts0.size(); e += 2) segList.push_back(Seg(elts0[e], elts0[e + 1]));
  std::list<Line> cLines;
  if (segList.empty()) return cLines;
  int idx = segList.front().second;
  segList.pop_front();
  cLines.push_back(Line());
  while (!segList.empty()) {
    Line::iterator it = segList.begin();
    for (; it != segList.end(); ++it)
      if (it->first == idx || it->second == idx) break;  // FindMySeg
    if (it != segList.end()) {
      if (it->first == idx) {
        idx = it->second;
        cLines.back().push_back(*it);
      } else {
        idx = it->first;
        cLines.back().push_back(Seg(it->second, it->first));
      }
      segList.erase(it);
    } else {
      cLines.push_back(Line());
      idx = segList.front().second;
      segList.pop_front();
    }
  }
  auto flipped = [](Line& l) {
    l.reverse();
    for (Seg& s : l) std::swap(s.first, s.second);
  };
  bool changed;
  do {
    changed = false;
    for (auto it = cLines.begin(); it != cLines.end(); ++it) {
      if (it->empty()) continue;
      const int idx_l = it->front().first, idx_r = it->back().second;  // read once per outer line (:1217-1218)
      for (auto it1 = cLines.begin(); it1 != cLines.end(); ++it1) {
        if (it1->empty() || it->empty() || it->front() == it1->front()) continue;
        if (idx_r == it1->front().first) {
          it->splice(it->end(), *it1);
          changed = true;
        } else if (idx_r == it1->back().second) {
          flipped(*it1);
          it->splice(it->end(), *it1);
          changed = true;
        } else if (idx_l == it1->front().first) {
          flipped(*it1);
          it->splice(it->begin(), *it1);
          changed = true;
        }
      }
    }
  } while (changed);
  for (auto it = cLines.begin(); it != cLines.end();) it = it->empty() ? cLines.erase(it) : std::next(it);
  std::cerr << "  number of contour lines: " << cLines.size() << std::endl;
  return cLines;
}
}  // namespace

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
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  pa::PlotfileHeader H = pa::read_header(infile, 2);
  double isoVal = 1090;
  pp.query("isoVal", isoVal);
  std::string isoCompName = "temp";
  pp.query("isoCompName", isoCompName);
  std::vector<int> pltComps;
  if (int n = pp.countval("comps")) {
    pp.queryarr("comps", pltComps, 0, n);
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
  const int nComp = (int)pltComps.size(), nc = 2 + nComp;
  int finestLevel = H.nlev - 1;
  pp.query("finestLevel", finestLevel);
  finestLevel = std::min(finestLevel, H.nlev - 1);
  const int Nlev = finestLevel + 1;
  int build_distance_function = 0, rm_external_elements = 1;
  pp.query("rm_external_elements", rm_external_elements);
  pp.query("build_distance_function", build_distance_function);
  if (build_distance_function) pa::Abort("Distance function not worked out for 2D yet");  // isosurface.cpp:1364-1366
  int ng = 1;
  pp.query("nGrow", ng);
  if (ng < 1) pa::Abort("nGrow must be at least 1");
  std::vector<int> is_per(3, 0);
  if (pp.countval("is_per")) {
    std::vector<int> p2;
    pp.queryarr("is_per", p2, 0, 2);
    is_per[0] = p2[0];
    is_per[1] = p2[1];
  }
  std::cout << "Periodicity assumed for this case: " << is_per[0] << " " << is_per[1] << " " << std::endl;

  const double strt_time_surf = now();
  double io_time = 0.0;
  pa::AsyncCtx actx;  // the HIP context comes up behind the reads
  std::vector<std::unique_ptr<pa::DevLevel>> dl;
  std::vector<std::unique_ptr<pa::DevMF>> dst;
  std::vector<pa::HostMF> hosts(Nlev);  // the mapped components; ghost cells -666 (gstate.setVal(-666), isosurface.cpp:1512)
  for (int lev = 0; lev < Nlev; ++lev) {
    hosts[lev].define(H.lev[lev].boxes, nComp, ng, -666.0);
    const double t_io = now();
    for (int n = 0; n < nComp; ++n) pa::read_comp(H, lev, pltComps[n], hosts[lev], n);
    io_time += now() - t_io;
  }
  pa::Ctx& ctx = actx.get();
  for (int lev = 0; lev < Nlev; ++lev) {
    const auto& L = H.lev[lev];
    pa::HostMF& host = hosts[lev];
    dl.emplace_back(new pa::DevLevel(ctx, L.boxes, L.domain, is_per.data(), H.prob_lo, H.prob_hi));
    dst.emplace_back(new pa::DevMF(ctx, *dl.back(), nc, ng));
    pa::DevMF dfield(ctx, *dl.back(), nComp, ng);
    ctx.check(pa_mf_upload(ctx.h, dfield.h, host.data.data()));
    pa::DevMF xyz(ctx, *dl.back(), 3, ng);  // (x, y, unused z) -> comps 0, 1 (isosurface.cpp:1458-1465)
    ctx.check(pa_iso_coords_level(ctx.h, xyz.h, 0));
    ctx.check(pa_mf_copy(ctx.h, xyz.h, 0, dst.back()->h, 0, 2, ng));
    ctx.check(pa_mf_copy(ctx.h, dfield.h, 0, dst.back()->h, 2, nComp, ng));
    std::cout << "FillPatching the grown structures at level " << lev << "..." << std::endl;
    ctx.check(pa_fill_boundary(ctx.h, dst[lev]->h, 0, nc, ng));
    if (lev > 0) ctx.check(pa_fillpatch_two_levels(ctx.h, dst[lev]->h, dst[lev - 1]->h, 0, nc, ng, 2, 0));  // PCInterp
    std::cout << "...done FillPatching the grown structures at level " << lev << "..." << std::endl;
    ctx.check(pa_sync(ctx.h));
  }
  if (pa_bc_errors(ctx.h) != 0) pa::Abort("FillPatchTwoLevels: fine grids are not properly nested in the coarse level");

  pa::IsoMerger merger(nc, 2);
  for (int lev = 0; lev < Nlev; ++lev) {
    const auto& L = H.lev[lev];
    const size_t nb = L.boxes.size();
    std::vector<pa_box> loops(nb);
    for (size_t b = 0; b < nb; ++b) {  // (grown box & domain grown in the periodic directions), high side - 1 (:1566-1569)
      const pa::Box3& B = L.boxes[b];
      for (int d = 0; d < 2; ++d) {
        const int pg = is_per[d] ? ng : 0;
        loops[b].lo[d] = std::max(B.lo[d] - ng, L.domain.lo[d] - pg);
        loops[b].hi[d] = std::min(B.hi[d] + ng, L.domain.hi[d] + pg) - 1;
      }
      loops[b].lo[2] = loops[b].hi[2] = 0;
    }
    std::vector<int64_t> nvb(nb, 0), nsb(nb, 0);
    double* dv = nullptr;
    int32_t *dk = nullptr, *ds = nullptr;
    ctx.check(pa_msq_level_fine(ctx.h, dst[lev]->h, lev < finestLevel ? dl[lev + 1]->h : nullptr, 2, loops.data(), 2 + isoComp, isoVal, nvb.data(), nsb.data(), &dv,
                                &dk, &ds));
    int64_t nvt = 0, nst = 0;
    for (size_t b = 0; b < nb; ++b) { nvt += nvb[b]; nst += nsb[b]; }
    std::vector<double> hva((size_t)(nvt * nc));
    std::vector<int32_t> hsa((size_t)(nst * 3)), hka((size_t)(nvt * 6));
    if (nvt > 0) {
      ctx.check(pa_memcpy_d2h(ctx.h, hva.data(), dv, nvt * nc * 8));
      ctx.check(pa_memcpy_d2h(ctx.h, hka.data(), dk, nvt * 6 * 4));
    }
    if (nst > 0) ctx.check(pa_memcpy_d2h(ctx.h, hsa.data(), ds, nst * 3 * 4));
    pa_device_free(ctx.h, dv);  // one allocation
    int64_t vo = 0, so = 0;
    std::vector<double> hv;
    std::vector<int32_t> hs;
    for (size_t b = 0; b < nb; vo += nvb[b], so += nsb[b], ++b) {
      const pa::Box3& B = L.boxes[b];
      const int64_t nv = nvb[b], ns = nsb[b];
      if (ns <= 0) continue;
      hv.assign(hva.begin() + vo * nc, hva.begin() + (vo + nv) * nc);
      hs.assign(hsa.begin() + so * 3, hsa.begin() + (so + ns) * 3);
      long long nvk = nv, nsk = ns;
      if (rm_external_elements && ng > 1) {  // isosurface.cpp:1657-1682 (a no-op when nGrow = 1)
        std::vector<int32_t> remap((size_t)nv);
        nvk = 0;
        for (int64_t q = 0; q < nv; ++q) {
          bool in = true;
          for (int d = 0; d < 2; ++d) {
            const int a = hka[(size_t)(vo + q) * 6 + d], c = hka[(size_t)(vo + q) * 6 + 3 + d];
            in = in && a >= B.lo[d] - 1 && a <= B.hi[d] + 1 && c >= B.lo[d] - 1 && c <= B.hi[d] + 1;
          }
          remap[(size_t)q] = in ? (int32_t)nvk : -1;
          if (in) {
            if (nvk != q) std::copy(hv.begin() + q * nc, hv.begin() + (q + 1) * nc, hv.begin() + nvk * nc);
            ++nvk;
          }
        }
        nsk = 0;
        for (int64_t t = 0; t < ns; ++t) {
          const int32_t a = remap[(size_t)hs[(size_t)t * 3]], c = remap[(size_t)hs[(size_t)t * 3 + 1]];
          if (a < 0 || c < 0) continue;
          hs[(size_t)nsk * 3] = a; hs[(size_t)nsk * 3 + 1] = c; hs[(size_t)nsk * 3 + 2] = -1;
          ++nsk;
        }
      }
      merger.add(hv.data(), nvk, hs.data(), nsk);
    }
  }
  {  // isosurface.cpp:1756-1771 (one rank: max = min)
    const double surf_time = now() - strt_time_surf - io_time;
    std::cout << "Max Compute Surface time: " << surf_time << '\n' << "Min Compute Surface time: " << surf_time << '\n';
    std::cout << "Max I/O time: " << io_time << '\n' << "Min I/O time: " << io_time << '\n';
  }
  const double strt_time_uniq = now();
  merger.finish();
  const std::vector<int32_t> elts = merger.elements();  // [nElts][2], 0-based
  std::cout << "Uniquify time: " << now() - strt_time_uniq << '\n';
  const double strt_time_sout = now();
  int writeSurf = 1;
  pp.query("writeSurf", writeSurf);
  std::string surfFormat = "MEF";
  pp.query("surfFormat", surfFormat);
  if (surfFormat != "MEF" && surfFormat != "XDMF") pa::Abort("surfFormat must be MEF or XDMF");
  if (writeSurf && surfFormat == "XDMF") {  // isosurface.cpp:2135-2229 (Polyline / XY); quirk kept: the default name carries the plotfile TIME
    char buf[72];
    std::snprintf(buf, sizeof buf, "%g", H.time);
    std::string outfile_base = infile + "_" + isoCompName + "_" + std::string(buf);
    pp.query("outfile_base", outfile_base);
    std::vector<std::string> vn;
    for (int n = 0; n < nComp; ++n) vn.push_back(H.names[pltComps[n]]);
    pa::write_xdmf(outfile_base, H.time, isoCompName, isoVal, vn, merger.nodes(), elts, 2);
  } else if (writeSurf) {
    std::cout << "...write surface in mef format (mef = Marcs element format)" << std::endl;
    std::cout << "      (Nelts,Nnodes):(" << elts.size() / 2 << ", " << merger.num_nodes() << ")" << std::endl;
    // isosurface.cpp:2021-2063: connect the segments into contour lines and integrate the first mapped component times
    // the line normal along each of them (pComp = yComp + 1)
    const std::vector<double>& nd = merger.nodes();
    for (const Line& line : make_clines(elts)) {
      double integral[2] = {0, 0};
      for (const Seg& seg : line) {
        const double *p0 = &nd[(size_t)seg.first * nc], *p1 = &nd[(size_t)seg.second * nc];
        const double x0 = p0[0], x1 = p1[0], y0 = p0[1], y1 = p1[1], avgp0 = p0[2], avgp1 = p1[2];
        const double len = std::sqrt(((x1 - x0) * (x1 - x0)) + (y1 - y0) * (y1 - y0));
        double normal[2] = {0, 0};
        if (len > 0) { normal[0] = (y0 - y1) / len; normal[1] = (x1 - x0) / len; }
        for (int i = 0; i < 2; ++i) integral[i] += normal[i] * 0.5 * (avgp0 + avgp1) * len;
      }
      std::cout << "Integral: " << integral[0] << " " << integral[1] << std::endl;
    }
    std::vector<std::string> vars{"X", "Y"};
    for (int n = 0; n < nComp; ++n) vars.push_back(H.names[pltComps[n]]);
    char buf[72];
    std::snprintf(buf, sizeof buf, "%g", isoVal);
    std::string outfile_base = infile + "_" + isoCompName + "_" + std::string(buf);
    pp.query("outfile_base", outfile_base);
    std::cout << "  Writing the file..." << std::endl;
    {  // isosurface.cpp:2097-2134 with nodesPerElt = 2 and the node FAB on the 2-D box (0..N-1, 0)
      const std::string file = outfile_base + ".mef";
      std::ofstream f(file, std::ios::binary);
      if (!f) pa::Abort("Unable to create " + file);
      char lab[64];
      std::snprintf(lab, sizeof lab, "%g", H.time);
      f << lab << "\n";
      for (size_t c = 0; c < vars.size(); ++c) f << vars[c] << (c + 1 < vars.size() ? " " : "");
      f << "\n" << elts.size() / 2 << " 2\n";
      f << "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))((0,0) (" << merger.num_nodes() - 1 << ",0) (0,0)) " << nc << "\n";
      f.write((const char*)nd.data(), sizeof(double) * nd.size());
      std::vector<int32_t> e1(elts.size());
      for (size_t q = 0; q < elts.size(); ++q) e1[q] = elts[q] + 1;
      f.write((const char*)e1.data(), sizeof(int32_t) * e1.size());
    }
    std::cout << "            ...done" << std::endl;
  }
  std::cout << "Surface output time: " << now() - strt_time_sout << '\n';
  return 0;
}
