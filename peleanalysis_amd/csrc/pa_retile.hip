// pa_retile.hip -- internal re-tiling of a level (host arithmetic only; no GPU code in this file).
//
// grad / curvature / filterPlt results are point-wise functions of the level's CELL SET: FillBoundary hands a cell its
// neighbour's value whichever box holds it, and the masks of MLCellLinOp::applyBC / InterpBndryData test whether a cell is
// covered by the LEVEL, not by which box (grad.cpp:158-213, curvature.cpp:283-326,426-457; filterPlt.cpp:141 re-chops the
// file's BoxArray itself).  The one place a box's shape enters is the normal interpolant at a coarse-fine face, whose order
// drops when the box is fewer than 3 cells thick in the face-normal direction (NX = min(n + 1, 4), oracle/pa_oracle.c
// orc_apply_bc).  So the kernels need not sweep the tiling the plotfile happens to hold: pa_level_retile returns another
// BoxArray with the SAME cells -- boxes that are at least `min_thick` cells thick in every direction merged into the largest
// rectangles the cell set allows (x-runs first: long contiguous rows), thin boxes passed through untouched -- and the
// callers (tools/, bench.py) move FAB data between the two tilings by box intersection on the way in and out.  Marching
// cubes numbers its nodes in box order (isosurface.cpp:1687-1726) and stays on the file's boxes.
// tests/test_retile.py: the oracle is bitwise invariant under such re-tilings (the consistency pin of the recalled
// applyBC / InterpBndryData restatement) and the HIP path on the re-tiled level equals the oracle on the ORIGINAL boxes.
#include "pa_internal.h"
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>
#include "../../include/peleanalysis_amd.h"

namespace {

struct RBox { int lo[3], hi[3]; };

inline long long rcells(const RBox& b) { return (long long)(b.hi[0] - b.lo[0] + 1) * (b.hi[1] - b.lo[1] + 1) * (b.hi[2] - b.lo[2] + 1); }

// cut a chain of consecutive pieces (piece q spans [edge[q], edge[q+1]) in cells) into parts of at most `mx` cells, as even
// as the piece boundaries allow; returns the first piece of every part (+ the end)
std::vector<int> cut_chain(const std::vector<int>& edge, int mx) {
  const int np = (int)edge.size() - 1;
  const long long L = edge[np] - edge[0];
  std::vector<int> cuts{0};
  if (L <= mx) { cuts.push_back(np); return cuts; }
  const long long parts = (L + mx - 1) / mx;
  // even targets snapped to the nearest piece boundary; fall back to greedy filling where a part would exceed mx
  int q = 0;
  for (long long p = 1; p < parts; ++p) {
    const long long target = edge[0] + (L * p + parts / 2) / parts;
    int best = -1;
    for (int r = q + 1; r < np; ++r) {
      if (edge[r] - edge[q] > mx) break;
      if (best < 0 || std::llabs(edge[r] - target) < std::llabs(edge[best] - target)) best = r;
    }
    if (best < 0) break;
    cuts.push_back(best);
    q = best;
  }
  // the tail (and anything the snapping left too long): greedy
  while (edge[np] - edge[q] > mx) {
    int r = q + 1;
    if (r >= np) break;  // a single piece longer than mx stays whole
    while (r + 1 < np && edge[r + 1] - edge[q] <= mx) ++r;
    cuts.push_back(r);
    q = r;
  }
  cuts.push_back(np);
  return cuts;
}

// merge boxes that are adjacent along `dir` and identical in the other two directions (whole boxes only), chains cut at mx
void merge_pass(std::vector<RBox>& bx, int dir, int mx) {
  const int t0 = dir == 0 ? 1 : 0, t1 = dir == 2 ? 1 : 2;
  std::vector<int> ord(bx.size());
  std::iota(ord.begin(), ord.end(), 0);
  auto key = [&](const RBox& b) { return std::array<int, 5>{b.lo[t1], b.hi[t1], b.lo[t0], b.hi[t0], b.lo[dir]}; };
  std::sort(ord.begin(), ord.end(), [&](int a, int b) { return key(bx[a]) < key(bx[b]); });
  std::vector<RBox> out;
  size_t i = 0;
  while (i < ord.size()) {
    size_t j = i;
    std::vector<int> edge{bx[ord[i]].lo[dir]};
    while (true) {
      edge.push_back(bx[ord[j]].hi[dir] + 1);
      if (j + 1 >= ord.size()) break;
      const RBox &a = bx[ord[j]], &b = bx[ord[j + 1]];
      if (a.lo[t0] != b.lo[t0] || a.hi[t0] != b.hi[t0] || a.lo[t1] != b.lo[t1] || a.hi[t1] != b.hi[t1] || b.lo[dir] != a.hi[dir] + 1) break;
      ++j;
    }
    const std::vector<int> cuts = cut_chain(edge, mx);
    for (size_t c = 0; c + 1 < cuts.size(); ++c) {
      RBox m = bx[ord[i]];
      m.lo[dir] = edge[cuts[c]];
      m.hi[dir] = edge[cuts[c + 1]] - 1;
      out.push_back(m);
    }
    i = j + 1;
  }
  bx.swap(out);
}

// An x-run [a, b) of cells as pieces [first, second).  The sweeps work in tiles PA_XTILE = 64 cells wide, one wavefront per row, and
// a tile that is partly filled costs what a full one does (boxes 96 wide: 0.45 of HBM against 0.59 for 128 or 192 wide,
// profiles/r05_retile.txt), while boxes at most 32 wide have a kernel of their own (two rows per wavefront).  So: a remainder
// of 1 .. 32 cells beyond a multiple of 64 becomes a box of its own, and the rest is cut into pieces of whole tiles, as even
// as whole tiles allow, at most mx cells each.  The cuts depend on (a, b) only, so rows with the same run stack in y and z.
constexpr int PA_XTILE = 64;
std::vector<std::pair<int, int>> x_pieces(int a, int b, int mx, int min_thick) {
  std::vector<std::pair<int, int>> out;
  const int L = b - a, r = L % PA_XTILE;
  int wend = b;
  if (L > PA_XTILE && r > 0 && r <= PA_XTILE / 2 && r >= min_thick) wend = b - r;
  const int W = wend - a;
  const int cap = mx >= PA_XTILE ? mx / PA_XTILE * PA_XTILE : mx;
  const int parts = (W + cap - 1) / cap;
  if (parts <= 1) {
    out.push_back({a, wend});
  } else if (cap < PA_XTILE) {  // limits below a tile (tests): pieces of at most the limit, the cuts on EVEN indices -- fine boxes stay aligned to the
    // refinement ratio (the smoothing solve's restriction and the face kernels' 2 x 2 blocks want that; advisor, round 5: [40..52] came out of
    // boxes aligned to 4)
    int pos = a;
    while (wend - pos > cap) {
      int c = pos + cap;
      if ((c & 1) && c - 1 > pos) --c;
      out.push_back({pos, c});
      pos = c;
    }
    out.push_back({pos, wend});
  } else {
    const int units = (W + PA_XTILE - 1) / PA_XTILE;  // tiles; the last one may be ragged
    int pos = a;
    for (int p = 0; p < parts; ++p) {
      const int u = units / parts + (p < units % parts ? 1 : 0);
      const int end = p + 1 == parts ? wend : std::min(wend, pos + u * PA_XTILE);
      if (end > pos) out.push_back({pos, end});
      pos = end;
    }
  }
  if (wend < b) out.push_back({wend, b});
  return out;
}

// the cell set on the grid of all box edges: maximal x-runs, stacked in y where their x-extents agree, then in z
bool retile_grid(const std::vector<RBox>& in, const int mx[3], int min_thick, std::vector<RBox>& out) {
  std::vector<int> xs[3];
  for (int d = 0; d < 3; ++d) {
    for (const RBox& b : in) { xs[d].push_back(b.lo[d]); xs[d].push_back(b.hi[d] + 1); }
    std::sort(xs[d].begin(), xs[d].end());
    xs[d].erase(std::unique(xs[d].begin(), xs[d].end()), xs[d].end());
  }
  const long long g[3] = {(long long)xs[0].size() - 1, (long long)xs[1].size() - 1, (long long)xs[2].size() - 1};
  if (g[0] * g[1] * g[2] > (1ll << 26)) return false;
  std::vector<unsigned char> occ((size_t)(g[0] * g[1] * g[2]), 0);
  auto idx = [&](int d, int v) { return (int)(std::lower_bound(xs[d].begin(), xs[d].end(), v) - xs[d].begin()); };
  for (const RBox& b : in) {
    int a[3], e[3];
    for (int d = 0; d < 3; ++d) { a[d] = idx(d, b.lo[d]); e[d] = idx(d, b.hi[d] + 1); }
    for (int k = a[2]; k < e[2]; ++k)
      for (int j = a[1]; j < e[1]; ++j) std::memset(&occ[(size_t)((k * g[1] + j) * g[0] + a[0])], 1, (size_t)(e[0] - a[0]));
  }
  std::vector<RBox> cur;  // in GRID indices, hi inclusive
  for (int k = 0; k < g[2]; ++k)
    for (int j = 0; j < g[1]; ++j) {
      const unsigned char* row = &occ[(size_t)((k * g[1] + j) * g[0])];
      int i = 0;
      while (i < g[0]) {
        if (!row[i]) { ++i; continue; }
        int e = i;
        while (e + 1 < g[0] && row[e + 1]) ++e;
        for (const auto& pc : x_pieces(xs[0][i], xs[0][e + 1], mx[0], min_thick))
          cur.push_back(RBox{{pc.first, xs[1][j], xs[2][k]}, {pc.second - 1, xs[1][j + 1] - 1, xs[2][k + 1] - 1}});
        i = e + 1;
      }
    }
  merge_pass(cur, 1, mx[1]);
  merge_pass(cur, 2, mx[2]);
  out.swap(cur);
  return true;
}

}  // namespace

extern "C" int pa_level_retile(int nboxes, const int32_t* b6, const int32_t max_size[3], int min_thick, int32_t* out6, int cap) {
  if (nboxes <= 0 || !b6 || !out6 || cap < nboxes) return -1;
  auto identity = [&]() {
    std::memcpy(out6, b6, sizeof(int32_t) * 6 * (size_t)nboxes);
    return nboxes;
  };
  int mx[3];
  for (int d = 0; d < 3; ++d) {
    mx[d] = max_size ? max_size[d] : 128;
    if (mx[d] < 1) return identity();
  }
  if (min_thick < 1) min_thick = 1;
  // a direction in which every box spans the same range (a 2-D level stored as the plane k = 0): nothing is ever merged or
  // cut along it, so the boxes' extent in it never changes and thinness there is no obstacle
  bool flat[3];
  for (int d = 0; d < 3; ++d) {
    flat[d] = true;
    for (int b = 1; b < nboxes && flat[d]; ++b) flat[d] = b6[6 * b + d] == b6[d] && b6[6 * b + 3 + d] == b6[3 + d];
  }
  std::vector<RBox> thick, thin;
  long long cells = 0;
  for (int b = 0; b < nboxes; ++b) {
    RBox B;
    bool t = false;
    for (int d = 0; d < 3; ++d) {
      B.lo[d] = b6[6 * b + d];
      B.hi[d] = b6[6 * b + 3 + d];
      if (B.hi[d] < B.lo[d]) return identity();
      if (!flat[d] && B.hi[d] - B.lo[d] + 1 < min_thick) t = true;
    }
    cells += rcells(B);
    (t ? thin : thick).push_back(B);
  }
  if (thick.size() < 2) return identity();
  auto valid = [&](const std::vector<RBox>& v) {
    long long c = 0;
    for (const RBox& B : v) {
      for (int d = 0; d < 3; ++d)
        if (!flat[d] && B.hi[d] - B.lo[d] + 1 < min_thick) return false;
      c += rcells(B);
    }
    long long ct = 0;
    for (const RBox& B : thin) ct += rcells(B);
    return c + ct == cells && v.size() + thin.size() <= (size_t)cap;
  };
  std::vector<RBox> res;
  bool ok = retile_grid(thick, mx, min_thick, res) && valid(res);
  if (!ok) {  // whole boxes only: always cell-exact and never thinner than its inputs
    res = thick;
    for (int d = 0; d < 3; ++d) merge_pass(res, d, mx[d]);
    ok = valid(res);
  }
  if (!ok) return identity();
  std::sort(res.begin(), res.end(), [](const RBox& a, const RBox& b) {
    return std::array<int, 3>{a.lo[2], a.lo[1], a.lo[0]} < std::array<int, 3>{b.lo[2], b.lo[1], b.lo[0]};
  });
  res.insert(res.end(), thin.begin(), thin.end());
  for (size_t b = 0; b < res.size(); ++b)
    for (int d = 0; d < 3; ++d) { out6[6 * b + d] = res[b].lo[d]; out6[6 * b + 3 + d] = res[b].hi[d]; }
  return (int)res.size();
}

// The limits the tools and bench.py hand to pa_level_retile for a whole hierarchy (measured, profiles/r05_retile.txt): boxes of
// 512 x 256 x 256 where EVERY level then consists of blocks at least 128 cells thick (the nested, regular hierarchies: fewer
// ghost shells and special faces, and no box-box faces in x -- the direction whose ghost strips cost a 128-B line per 16 bytes --
// across a 512-wide level: headline pass 6.4-6.5 ms on the file's 128^3 boxes, 6.14-6.34 on 256^3, 6.05-6.12 on 512 x 256 x 256,
// 6.3 on one 512^3 box per level), otherwise 128^3 on all levels -- on a Pele-like BoxArray larger limits lose (6.93-7.0 ms
// against 7.0-7.1 for 256^3 on the regular level only and 7.05-7.2 for 256^3 / 256 x 128 x 128 everywhere).
// PA_RETILE_MAX="x y z" in the environment overrides.
extern "C" int pa_hierarchy_retile_limits(int nlev, const int32_t* nboxes, const int32_t* const* boxes6, int min_thick, int32_t max_size[3]) {
  if (nlev <= 0 || !nboxes || !boxes6 || !max_size) return -1;
  if (pa_opt().retile_max[0] > 0) {
    for (int d = 0; d < 3; ++d) max_size[d] = pa_opt().retile_max[d];
    return 0;
  }
  const int32_t big[3] = {512, 256, 256};
  bool ok = true;
  for (int l = 0; l < nlev && ok; ++l) {
    if (nboxes[l] <= 0) continue;
    const int cap = 4 * nboxes[l] + 16;
    std::vector<int32_t> out((size_t)cap * 6);
    const int n = pa_level_retile(nboxes[l], boxes6[l], big, min_thick, out.data(), cap);
    if (n < 0) return -1;
    for (int b = 0; b < n && ok; ++b)
      for (int d = 0; d < 3; ++d) {
        const int ext = out[6 * b + 3 + d] - out[6 * b + d] + 1;
        // a direction every box spans alike (2-D levels) does not count
        bool flat = true;
        for (int q = 1; q < n && flat; ++q) flat = out[6 * q + d] == out[d] && out[6 * q + 3 + d] == out[3 + d];
        if (!flat && ext < 128) ok = false;
      }
  }
  for (int d = 0; d < 3; ++d) max_size[d] = ok ? big[d] : 128;
  return 0;
}

// ... for a hierarchy that is sharded over nranks ranks: every rank should keep at least four boxes per level (fewer, larger boxes
// balance worse over the ranks and leave the sweep's per-XCD queues short: rank 0's share of the headline hierarchy, ms per pass
// with 32-us exchanges, 256^3 / 256 x 256 x 128 / 128^3 boxes: 2 ranks 3.38 / 3.56 / 3.59, 4 ranks 2.21 / 1.94 / 2.00, 8 ranks
// 1.31 / 1.16 / 1.06; profiles/r05_sim8_delay.txt).  The largest of 512 x 256 x 256 and those three tilings that leaves >= 4 nranks
// boxes on every level (or as many as the file has), starting at 128^3 wherever the one-rank choice is 128^3.
extern "C" int pa_hierarchy_retile_limits_ranks(int nlev, const int32_t* nboxes, const int32_t* const* boxes6, int min_thick, int nranks, int32_t max_size[3]) {
  if (pa_hierarchy_retile_limits(nlev, nboxes, boxes6, min_thick, max_size) != 0) return -1;
  if (nranks <= 1 || pa_opt().retile_max[0] > 0) return 0;
  // round 6: 64^3 and 32^3 after 128^3 -- a small hierarchy on many ranks (base 64 on 8: one merged box per level, seven ranks
  // idle) keeps boxes for every rank; a level never needs more boxes than the file gave it
  const int32_t cand[6][3] = {{512, 256, 256}, {256, 256, 256}, {256, 256, 128}, {128, 128, 128}, {64, 64, 64}, {32, 32, 32}};
  for (int c = max_size[0] <= 128 ? 3 : 0; c < 6; ++c) {
    bool ok = true;
    for (int l = 0; l < nlev && ok; ++l) {
      if (nboxes[l] <= 0) continue;
      const int cap = 4 * nboxes[l] + 16;
      std::vector<int32_t> out((size_t)cap * 6);
      const int n = pa_level_retile(nboxes[l], boxes6[l], cand[c], min_thick, out.data(), cap);
      ok = n >= std::min(4 * nranks, (int)nboxes[l]) || c == 5;
    }
    if (ok) {
      for (int d = 0; d < 3; ++d) max_size[d] = cand[c][d];
      return 0;
    }
  }
  return 0;
}
