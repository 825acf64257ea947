// isoMEF3d -- drop-in for PeleAnalysis Src/isoMEF.cpp: contour lines of one node variable ON a MEF surface, written as
// Tecplot line zones to ./out.dat.  Host only (a consumer of what isosurface writes).
//   isoMEF3d.ex infile=<file.mef> isoComp=<node variable index> isoVal=<value>
// What the reference does, kept to the letter (isoMEF.cpp:121-341, 382-490):
//   * every triangle with the iso value between two of its nodes yields ONE segment between the two crossed edges, taken
//     in the order (p0,p1), (p1,p2), (p2,p0) (Segmentise, :382-420);
//   * a crossed edge owns one vertex, cached under the UNORDERED node pair and interpolated with the endpoint order of the
//     first element that asked for it (VertexInterp :461-482; the comparator ignores direction, so the reverse lookup of
//     :467 is never reached): P = P1 + mu (P2 - P1), mu = (isoVal - v1) / (v2 - v1), or an endpoint when isoVal or v1 - v2
//     is within 1e-8 (VI_doIt :426-459; the two extra components it appends -- cylindrical radius and angle -- are never
//     written);
//   * vertices are numbered in cache order = by (min node, max node) of their edge (:196-205);
//   * lines: start at the left end of the first segment; repeatedly take the FIRST remaining segment that touches the
//     current vertex, flipped if necessary, and move to its other end; when none touches it, open a new line and continue
//     from the left end of the first remaining segment WITHOUT consuming it (:210-246);
//   * then join fragments until nothing changes (:249-291), with the end ids of the outer fragment read once per outer
//     step (they go stale after a splice, as in the reference), and drop the empty lines;
//   * out.dat: "VARIABLES = <names>", one FELINESEG zone per line with its points (first point of every segment + the
//     last point of the last one) and local connectivity "i i+1" (:300-333).
// stdout: "Found <n> segments"; stderr: the node / element counts and "  number of contours <n>".
#include "../common/pa_plotfile.h"
#include <list>
#include <map>
#include <set>
#include <sstream>

namespace {
struct Seg { int l, r; };  // vertex ids of the two ends
inline bool same(const Seg& a, const Seg& b) { return a.l == b.l && a.r == b.r; }
}  // namespace

int main(int argc, char** argv) {
  pa::ParmParse pp(argc, argv);
  std::string infile;
  pp.get("infile", infile);
  int isoComp = 0;
  double isoVal = 0.0;
  pp.get("isoComp", isoComp);
  pp.get("isoVal", isoVal);
  const pa::MefSurface S = pa::read_mef(infile);
  const int nComp = (int)S.names.size();
  if (isoComp < 0 || isoComp >= nComp) pa::Abort("isoComp is not a node variable of " + infile);
  if (S.nodesPerElt != 3) pa::Abort("isoMEF needs a surface of triangles");
  std::cerr << S.nNodes << " nodes read in with " << nComp << " states per node" << std::endl;
  std::cerr << S.nElts << " elements read in with " << S.nodesPerElt << " nodes per element" << std::endl;
  auto node = [&](int n) { return &S.nodes[(size_t)n * nComp]; };

  // ---- segments, element by element; vertices cached per unordered edge
  struct Vert { std::vector<double> p; int id = 0; };
  std::map<std::pair<int, int>, Vert> cache;
  auto vertex = [&](int a, int b) -> const std::pair<int, int> {
    const std::pair<int, int> key(std::min(a, b), std::max(a, b));
    if (cache.find(key) == cache.end()) {
      constexpr double eps = 1.e-8;
      const double *pa_ = node(a), *pb = node(b);
      const double va = pa_[isoComp], vb = pb[isoComp];
      Vert v;
      if (std::abs(isoVal - va) < eps) v.p.assign(pa_, pa_ + nComp);
      else if (std::abs(isoVal - vb) < eps) v.p.assign(pb, pb + nComp);
      else if (std::abs(va - vb) < eps) v.p.assign(pa_, pa_ + nComp);
      else {
        const double mu = (isoVal - va) / (vb - va);
        v.p.resize(nComp);
        for (int j = 0; j < nComp; ++j) v.p[j] = pa_[j] + mu * (pb[j] - pa_[j]);
      }
      cache.emplace(key, std::move(v));
    }
    return key;
  };
  std::vector<std::pair<std::pair<int, int>, std::pair<int, int>>> raw;  // per segment: the two edge keys
  for (long long e = 0; e < S.nElts; ++e) {
    const int n0 = S.conn[(size_t)e * 3] - 1, n1 = S.conn[(size_t)e * 3 + 1] - 1, n2 = S.conn[(size_t)e * 3 + 2] - 1;
    const bool lo0 = node(n0)[isoComp] < isoVal, lo1 = node(n1)[isoComp] < isoVal, lo2 = node(n2)[isoComp] < isoVal;
    std::vector<std::pair<int, int>> cut;
    if (lo0 != lo1) cut.push_back(vertex(n0, n1));
    if (lo1 != lo2) cut.push_back(vertex(n1, n2));
    if (lo2 != lo0) cut.push_back(vertex(n2, n0));
    if (cut.size() == 2) raw.push_back({cut[0], cut[1]});
  }
  std::cout << "Found " << raw.size() << " segments " << std::endl;
  int cnt = 0;
  std::vector<const Vert*> verts;
  for (auto& kv : cache) { kv.second.id = cnt++; verts.push_back(&kv.second); }
  std::vector<Seg> segs(raw.size());
  for (size_t i = 0; i < raw.size(); ++i) segs[i] = Seg{cache[raw[i].first].id, cache[raw[i].second].id};

  // ---- lines: "first remaining segment touching the vertex" = smallest index among the remaining ones that hold it
  std::vector<std::set<int>> at(verts.size());
  std::set<int> left;
  for (int i = 0; i < (int)segs.size(); ++i) { at[segs[i].l].insert(i); at[segs[i].r].insert(i); left.insert(i); }
  std::list<std::list<Seg>> lines;
  if (!segs.empty()) {
    int idx = segs[*left.begin()].l;
    lines.emplace_back();
    while (!left.empty()) {
      if (!at[idx].empty()) {
        const int i = *at[idx].begin();
        const Seg s = segs[i];
        lines.back().push_back(s.l == idx ? s : Seg{s.r, s.l});
        at[s.l].erase(i);
        at[s.r].erase(i);
        left.erase(i);
        idx = s.l == idx ? s.r : s.l;
      } else {
        lines.emplace_back();
        idx = segs[*left.begin()].l;
      }
    }
    // ---- join the fragments as far as possible
    bool changed;
    do {
      changed = false;
      for (auto it = lines.begin(); it != lines.end(); ++it) {
        if (it->empty()) continue;
        const int idx_l = it->front().l, idx_r = it->back().r;  // read once per outer fragment
        for (auto jt = lines.begin(); jt != lines.end(); ++jt) {
          if (jt->empty() || same(it->front(), jt->front())) continue;
          auto reversed = [&] {
            jt->reverse();
            for (Seg& s : *jt) std::swap(s.l, s.r);
          };
          if (idx_r == jt->front().l) {
            it->splice(it->end(), *jt);
            changed = true;
          } else if (idx_r == jt->back().r) {
            reversed();
            it->splice(it->end(), *jt);
            changed = true;
          } else if (idx_l == jt->front().l) {
            reversed();
            it->splice(it->begin(), *jt);
            changed = true;
          }
        }
      }
    } while (changed);
  }
  lines.remove_if([](const std::list<Seg>& l) { return l.empty(); });
  std::cerr << "  number of contours " << lines.size() << std::endl;

  std::ofstream os("out.dat");
  if (!os) pa::Abort("Unable to create out.dat");
  os << "VARIABLES =";
  for (auto& n : S.names) os << " " << n;
  os << '\n';
  auto put = [&](int id) {
    for (int n = 0; n < nComp; ++n) os << verts[id]->p[n] << " ";
    os << "\n";
  };
  for (const auto& ln : lines) {
    os << "ZONE ZONETYPE=FELINESEG DATAPACKING=POINT N=" << ln.size() + 1 << " E=" << ln.size() << "\n";
    for (const Seg& s : ln) put(s.l);
    put(ln.back().r);
    for (size_t c = 1; c <= ln.size(); ++c) os << c << " " << c + 1 << '\n';
  }
  return 0;
}
