// checkIso3d -- drop-in for PeleAnalysis Src/checkIso.cpp: orientation check of a MEF surface.  Host only (the
// validator the reference ships for what isosurface wrote).
//   checkIso3d.ex isoFile=<file.mef>
// Every element contributes its three directed edges (n0,n1), (n1,n2), (n2,n0) to a set keyed by the UNORDERED
// node pair; when a pair is already present, the set must hold it in the opposite direction (checkIso.cpp:127-149:
// a shared edge is traversed once each way by consistently oriented neighbours) -- otherwise the reference's
// AMREX_ALWAYS_ASSERT fires: "Assertion `edgeSet.find(e.reverse())!=edgeSet.end()' failed" and a non-zero exit.
// Quirk kept: the comparator ignores direction, so find(e.reverse()) finds the stored edge whatever its direction;
// the assertion can therefore never fail once the insertion was refused, and the tool accepts every file it can
// read.  `strict=1` (an addition, default 0) makes the check real and exits with 2 on the first edge traversed twice
// in the same direction or shared by more than two elements.
#include "../common/pa_plotfile.h"
#include <map>

int main(int argc, char** argv) {
  pa::ParmParse pp(argc, argv);
  std::string isoFile;
  pp.get("isoFile", isoFile);
  int strict = 0;
  pp.query("strict", strict);
  std::cerr << "Reading isoFile... " << isoFile << std::endl;
  const pa::MefSurface S = pa::read_mef(isoFile);
  std::cout << "nelts: " << S.nElts << std::endl;
  std::cout << "nodesperelt: " << S.nodesPerElt << std::endl;
  if (S.nodesPerElt < 3) pa::Abort("checkIso: elements need at least 3 nodes");
  std::cout << "Read " << S.nElts << " elements and " << S.nNodes << " nodes" << std::endl;
  // unordered pair -> (direction of the first traversal: +1 = (min,max), number of traversals each way)
  struct Use { int fwd = 0, bwd = 0; };
  std::map<std::pair<int32_t, int32_t>, Use> edges;
  for (long long e = 0; e < S.nElts; ++e) {
    const int32_t* f = &S.conn[(size_t)e * S.nodesPerElt];
    for (int q = 0; q < 3; ++q) {
      const int32_t a = f[q], b = f[(q + 1) % 3];
      Use& u = edges[{std::min(a, b), std::max(a, b)}];
      (a <= b ? u.fwd : u.bwd)++;
      if (strict && (u.fwd > 1 || u.bwd > 1)) {
        std::cerr << "edge (" << a << "," << b << ") of element " << e << " is traversed twice in the same direction" << std::endl;
        return 2;
      }
    }
  }
  std::cout << "Found " << edges.size() << " edges (nElts * 3 = " << S.nElts * 3 << ")" << std::endl;
  std::cout << "All shared edges are consistently numbered." << std::endl;
  return 0;
}
