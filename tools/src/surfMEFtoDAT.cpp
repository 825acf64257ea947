// surfMEFtoDAT3d -- drop-in for PeleAnalysis Src/surfMEFtoDAT.cpp: MEF surface -> Tecplot ASCII (FEPOINT).  Host only
// (the consumer the reference uses to inspect what isosurface wrote).
//   surfMEFtoDAT3d.ex infile=<file.mef> [outfile=<file.dat>] [verbose=0]
#include "../common/pa_plotfile.h"
#include <sstream>

int main(int argc, char** argv) {
  pa::ParmParse pp(argc, argv);
  if (argc < 2 || pp.contains("help")) {
    std::cerr << "usage:\n" << argv[0] << " infile=<name> [options] \n\tOptions:\n\t     outfile=<name>\n";
    return 1;
  }
  int verbose = 0;
  pp.query("verbose", verbose);
  std::string infile;
  pp.get("infile", infile);
  std::ifstream is(infile, std::ios::in | std::ios::binary);
  if (!is) pa::Abort("Unable to open file : " + infile);
  std::string title, line;
  std::getline(is, title);  // parseTitle (:125-130)
  std::getline(is, line);   // parseVarNames (:118-123)
  std::vector<std::string> names;
  {
    std::istringstream ss(line);
    std::string t;
    while (ss >> t) names.push_back(t);
  }
  const int nComp = (int)names.size();
  long long nElts = 0, MYLEN = 0;
  is >> nElts >> MYLEN;
  std::getline(is, line);
  std::getline(is, line);  // FAB header: "... ((0,0,0) (N-1,0,0) (0,0,0)) ncomp"
  const size_t p0 = line.rfind("((0,0,0) (");
  if (p0 == std::string::npos) pa::Abort("cannot parse the node FAB header of " + infile);
  const long long nPts = std::atoll(line.c_str() + p0 + 10) + 1;
  std::vector<double> nodeData((size_t)nPts * nComp);
  is.read((char*)nodeData.data(), sizeof(double) * nodeData.size());
  std::vector<int32_t> connData((size_t)nElts * MYLEN, 0);
  is.read((char*)connData.data(), sizeof(int32_t) * connData.size());
  if (!is) pa::Abort("truncated MEF file " + infile);
  // outfile: the input name with its last extension replaced by .dat (:73-81)
  std::string outfile = infile;
  const size_t dot = outfile.rfind('.');
  if (dot != std::string::npos) outfile = outfile.substr(0, dot);
  outfile += ".dat";
  pp.query("outfile", outfile);
  std::ofstream os(outfile);
  if (!os) pa::Abort("Unable to create " + outfile);
  os << "VARIABLES =";
  for (auto& n : names) os << " " << n;
  os << std::endl;
  os << "ZONE T=\"" << title << "\" N=" << nPts << " E=" << nElts << " F=FEPOINT ET=" << (MYLEN == 2 ? "LINESEG" : "TRIANGLE") << std::endl;
  for (long long i = 0; i < nPts; ++i) {
    for (int k = 0; k < nComp; ++k) os << nodeData[(size_t)i * nComp + k] << " ";
    os << std::endl;
  }
  for (long long i = 0; i < nElts; ++i) {
    for (int k = 0; k < MYLEN; ++k) os << connData[(size_t)i * MYLEN + k] << " ";
    os << std::endl;
  }
  if (verbose) std::cerr << "Wrote " << outfile << std::endl;
  return 0;
}
