// pa_isomerge.h -- global node / element sets of the isosurface tool (isosurface.cpp:1687-1726,
// 1751-1812, Node :805-873, Element :877-927) without the red-black trees.
//   nodes    unique by position up to the reference's tolerance: two nodes closer than 1e-15
//            (Euclidean) are one node, the FIRST inserted copy is kept, id = insertion order
//   elements node-id triples rotated so the smallest id comes first (orientation preserved),
//            degenerate ones dropped, unique, sorted lexicographically (std::set<Element> order)
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <unordered_map>
#include <vector>

namespace pa {

class IsoMerger {
 public:
  // dim = 2: the 2-D build (positions are (x, y); elements are segments, passed as rows (id0, id1, -1))
  explicit IsoMerger(int ncomp, int dim = 3) : nc_(ncomp), dim_(dim) {}
  // one FAB's fragment: verts [nv][ncomp] in vertCache order, tris [nt][3] local ids
  void add(const double* verts, long long nv, const int32_t* tris, long long nt) {
    std::vector<int32_t> ids((size_t)nv);
    for (long long q = 0; q < nv; ++q) ids[q] = node_id(verts + q * nc_);
    for (long long t = 0; t < nt; ++t) {
      if (dim_ == 2) {  // Element of two ids: the smaller first (std::rotate on two entries), v0 == v1 dropped (:1707-1715)
        const int32_t a = ids[tris[3 * t]], b = ids[tris[3 * t + 1]];
        if (a != b) elts_.push_back({std::min(a, b), std::max(a, b), 0});
        continue;
      }
      std::array<int32_t, 3> v{ids[tris[3 * t]], ids[tris[3 * t + 1]], ids[tris[3 * t + 2]]};
      if (v[0] == v[1] || v[1] == v[2] || v[0] == v[2]) continue;  // degenerate (isosurface.cpp:1723-1724)
      const int s = (int)(std::min_element(v.begin(), v.end()) - v.begin());
      std::rotate(v.begin(), v.begin() + s, v.end());
      elts_.push_back(v);
    }
  }
  void finish() {
    std::sort(elts_.begin(), elts_.end());
    elts_.erase(std::unique(elts_.begin(), elts_.end()), elts_.end());
  }
  const std::vector<double>& nodes() const { return nodes_; }
  long long num_nodes() const { return nc_ ? (long long)nodes_.size() / nc_ : 0; }
  std::vector<int32_t> elements() const {
    std::vector<int32_t> e;
    e.reserve(3 * elts_.size());
    for (auto& v : elts_) { e.push_back(v[0]); e.push_back(v[1]); if (dim_ == 3) e.push_back(v[2]); }
    return e;
  }

 private:
  int nc_, dim_;
  std::vector<double> nodes_;
  std::vector<std::array<int32_t, 3>> elts_;
  struct Key { long long x, y, z; bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; } };
  struct KeyHash { size_t operator()(const Key& k) const { return (size_t)(k.x * 73856093LL ^ k.y * 19349663LL ^ k.z * 83492791LL); } };
  std::unordered_map<Key, std::vector<int32_t>, KeyHash> grid_;

  int32_t node_id(const double* p) {
    constexpr double EPS = 1.0e-15, H = 1.0e-14;
    const double pz = dim_ == 3 ? p[2] : 0.0;
    const Key g{(long long)std::floor(p[0] / H), (long long)std::floor(p[1] / H), (long long)std::floor(pz / H)};
    int32_t best = -1;
    // a node within EPS of p sits in p's hash cell, or in the neighbour across a face p is within EPS of (H = 10 EPS):
    // the other neighbours cannot hold one, so they are not looked up (27 lookups -> 1 for almost every node)
    const double pp[3] = {p[0], p[1], pz};
    const long long gg[3] = {g.x, g.y, g.z};
    int lo[3], hi[3];
    for (int d = 0; d < 3; ++d) {
      const double r = pp[d] - (double)gg[d] * H;  // position inside the cell, up to rounding of the product
      lo[d] = (r < 2 * EPS) ? -1 : 0;
      hi[d] = (r > H - 2 * EPS) ? 1 : 0;
    }
    for (int dz = lo[2]; dz <= hi[2]; ++dz)
      for (int dy = lo[1]; dy <= hi[1]; ++dy)
        for (int dx = lo[0]; dx <= hi[0]; ++dx) {
          if (dim_ == 2 && dz != 0) continue;
          auto it = grid_.find(Key{g.x + dx, g.y + dy, g.z + dz});
          if (it == grid_.end()) continue;
          for (int32_t c : it->second) {
            const double* q = &nodes_[(size_t)c * nc_];
            const double a = q[0] - p[0], b = q[1] - p[1], d = dim_ == 3 ? q[2] - p[2] : 0.0;
            if (std::sqrt(a * a + b * b + d * d) < EPS && (best < 0 || c < best)) best = c;
          }
        }
    if (best >= 0) return best;
    const int32_t id = (int32_t)num_nodes();
    grid_[g].push_back(id);
    nodes_.insert(nodes_.end(), p, p + nc_);
    return id;
  }
};

}  // namespace pa
