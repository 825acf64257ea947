// template3d -- drop-in for PeleAnalysis Src/template.cpp: read every component of a plotfile, copy
// it, write <root>_temp.  Exercises only the plotfile reader / writer (no GPU needed).
//   template3d.ex infile=<plt> [finestLevel=<n>] [is_per="1 1 1"] [retile=0|1]
// retile=1: the data are held on the internal tiling of the other tools (pa_level_retile: the file's cells merged into large boxes;
// read_comp fills a merged box from every file FAB it intersects, write_plotfile gathers every file box back) -- the output must be
// byte-identical to retile=0: the host half of the tools' re-tiling, testable without a GPU (tests/test_plotfile_tools.py)
#include "../common/pa_plotfile.h"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cerr << "usage:\n" << argv[0] << " infile=<plotfilename> \n\tOptions:\n\tis_per=<L M N>\n";
    return 1;
  }
  pa::ParmParse pp(argc, argv);
  if (pp.contains("help")) {
    std::cerr << "usage:\n" << argv[0] << " infile=<plotfilename> \n\tOptions:\n\tis_per=<L M N>\n";
    return 1;
  }
  std::string infile;
  pp.get("infile", infile);
  int finestLevel = 1000;
  pp.query("finestLevel", finestLevel);
  std::vector<int> is_per(3, 1);
  pp.queryarr("is_per", is_per, 0, 3);
  std::cout << "Periodicity assumed for this case: " << is_per[0] << " " << is_per[1] << " " << is_per[2] << " \n";
  pa::PlotfileHeader H = pa::read_header(infile);
  const int Nlev = std::min(finestLevel, H.nlev - 1) + 1;
  std::vector<pa::HostMF> out(Nlev);
  std::vector<pa::Box3> doms;
  std::vector<int> steps(Nlev, 0);
  int retile = 0;
  pp.query("retile", retile);
  const std::vector<std::vector<pa::Box3>> fileBoxes = pa::level_boxes(H, Nlev), tile = retile ? pa::retile_levels(fileBoxes, pp) : fileBoxes;
  for (int lev = 0; lev < Nlev; ++lev) {
    std::cout << "Reading data for level " << lev << std::endl;
    out[lev].define(tile[lev], (int)H.names.size(), 0);
    for (int c = 0; c < (int)H.names.size(); ++c) pa::read_comp(H, lev, c, out[lev], c);
    std::cout << "Data has been read for level " << lev << std::endl;
    doms.push_back(H.lev[lev].domain);
  }
  const std::string outfile = pa::getFileRoot(infile) + "_temp";
  pa::OldOutput old_out;
  old_out.move_away(outfile, infile, pp);  // UtilCreateCleanDirectory: an earlier run's plotfile (levels this run does not write included) is renamed
  std::cout << "Writing new data to " << outfile << std::endl;
  pa::write_plotfile(outfile, H.names, doms, H.prob_lo, H.prob_hi, out, 0.0, steps, 2, 3, nullptr, pa::boxes_if_retiled(fileBoxes, tile));
  old_out.finish();
  return 0;
}
