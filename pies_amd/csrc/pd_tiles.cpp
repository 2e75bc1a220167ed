// Tiles of the PD local step (host, once per topology change; see PdTileArrays in pd_kernels.h).
//
// The reference's local step writes one projection per constraint and its right-hand side adds, per node, the ~24
// contributions of the elements around it (Solver.cpp:270-349).  On the device that was a round trip through HBM of one
// 12-byte record per (element, node): 105 + 101 us per local/global iteration at 1M particles, 68 % of the substep
// (profiles/r03_pd1m_kernel_stats.csv).  The tiles keep the sums on chip: spatially close element pairs form a tile, one
// wavefront holds the tile's nodes in LDS and writes one sum per (tile, node).
//
//   order    element pairs sorted along a Morton curve of their centroids (cells of about two element spacings), cut into
//            runs of up to kTileElems pairs that touch at most kTileNodes nodes
//   lists    per tile node, the (element, corner) slots that contribute to it, ascending: the order a lane adds them in
//   records  tile node k of tile t = slot (base + kTileNodes t + k) of the contribution array; a node's incidence list names
//            its tiles' slots in ascending tile order
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <unordered_map>
#include <vector>

#include "device_util.h"

namespace pies {

namespace {
uint64_t spread21(uint64_t v) {  // bits of v (21 of them) to every third position
  v &= 0x1fffffull;
  v = (v | v << 32) & 0x1f00000000ffffull;
  v = (v | v << 16) & 0x1f0000ff0000ffull;
  v = (v | v << 8) & 0x100f00f00f00f00full;
  v = (v | v << 4) & 0x10c30c30c30c30c3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}
}  // namespace

bool pd_plan_tiles(const pies_solver* s, PdTilePlan& out) {
  out = PdTilePlan{};
  const uint32_t ne = static_cast<uint32_t>(s->h_tet.size());
  const uint32_t n = s->nodeCount();
  if (!s->tetVolumePaired || !s->pdLocalPacked || ne == 0) return false;
  uint32_t maxElems = kTileElems;  // (PIES_PD_TILE_ELEMS: 0 = per-(element, node) records, otherwise pairs per tile, at most 128)
  if (const char* e = tuning_env("PIES_PD_TILE_ELEMS")) {
    const int v = std::atoi(e);
    if (v <= 0) return false;
    maxElems = static_cast<uint32_t>(std::min<int>(v, kTileElems));
  }
  // ---- curve order ------------------------------------------------------------------------------------------------
  std::vector<float> cx(3ull * ne);
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t e = 0; e < ne; ++e)
    for (int a = 0; a < 3; ++a) {
      float c = 0.f;
      for (int i = 0; i < 4; ++i) c += s->h_pos[3ull * s->h_tet[e].ids[i] + a];
      c *= 0.25f;
      if (!std::isfinite(c)) c = 0.f;
      cx[3ull * e + a] = c;
      lo[a] = std::min(lo[a], c);
      hi[a] = std::max(hi[a], c);
    }
  double vol = 1.0;
  int dims = 0;
  for (int a = 0; a < 3; ++a)
    if (hi[a] > lo[a]) { vol *= static_cast<double>(hi[a]) - lo[a]; ++dims; }
  // cells that hold ~48 elements of a space-filling mesh (six tetrahedra per lattice cell: cells of two lattice spacings)
  double cell = dims ? std::pow(vol * 48.0 / ne, 1.0 / dims) : 1.0;
  for (int a = 0; a < 3; ++a) cell = std::max(cell, (static_cast<double>(hi[a]) - lo[a]) / 2.0e6);
  if (!(cell > 0.0) || !std::isfinite(cell)) cell = 1.0;
  std::vector<uint64_t> key(ne);
  for (uint32_t e = 0; e < ne; ++e) {
    uint64_t k = 0;
    for (int a = 0; a < 3; ++a) {
      const double q = (static_cast<double>(cx[3ull * e + a]) - lo[a]) / cell;
      k |= spread21(static_cast<uint64_t>(std::min(std::max(q, 0.0), 2097151.0))) << a;
    }
    key[e] = k;
  }
  std::vector<uint32_t> order(ne);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
  // ---- runs ---------------------------------------------------------------------------------------------------------
  std::vector<uint32_t> stamp(n, 0xffffffffu), localOf(n, 0u);
  std::vector<uint32_t> tileNodes;
  std::vector<std::vector<uint16_t>> lists(kTileNodes);
  size_t at = 0;
  while (at < ne) {
    const uint32_t tileId = static_cast<uint32_t>(out.info.size());
    tileNodes.clear();
    size_t end = at;
    while (end < ne && end - at < maxElems) {
      const uint32_t* ids = s->h_tet[order[end]].ids;
      uint32_t fresh = 0;
      for (int i = 0; i < 4; ++i) {
        bool seen = stamp[ids[i]] == tileId;
        for (int j = 0; j < i && !seen; ++j) seen = ids[j] == ids[i];
        if (!seen) ++fresh;
      }
      if (tileNodes.size() + fresh > kTileNodes) break;
      for (int i = 0; i < 4; ++i)
        if (stamp[ids[i]] != tileId) { stamp[ids[i]] = tileId; tileNodes.push_back(ids[i]); }
      ++end;
    }
    if (end == at) return false;  // (cannot happen: an element has four nodes)
    std::sort(tileNodes.begin(), tileNodes.end());  // ascending node index: the tile's position gather walks memory forwards
    const uint32_t nn = static_cast<uint32_t>(tileNodes.size()), cnt = static_cast<uint32_t>(end - at);
    for (uint32_t k = 0; k < nn; ++k) { localOf[tileNodes[k]] = k; lists[k].clear(); }
    const size_t e0 = out.elem.size(), n0 = out.node.size();
    for (uint32_t k = 0; k < cnt; ++k) {
      const uint32_t e = order[at + k];
      const uint32_t* ids = s->h_tet[e].ids;
      out.elem.push_back(e);
      out.local.push_back(localOf[ids[0]] | (localOf[ids[1]] << 8) | (localOf[ids[2]] << 16) | (localOf[ids[3]] << 24));
      for (uint32_t i = 0; i < 4; ++i) lists[localOf[ids[i]]].push_back(static_cast<uint16_t>(k << 2 | i));
    }
    out.elem.resize(e0 + kTileElems, out.elem.back());   // (slots past the tile's count are never used as elements of their own)
    out.local.resize(e0 + kTileElems, out.local.back());
    out.node.insert(out.node.end(), tileNodes.begin(), tileNodes.end());
    out.node.resize(n0 + kTileNodes, tileNodes.back());
    const size_t p0 = out.nptr.size(), i0 = out.inc.size();
    uint32_t run = 0;
    for (uint32_t k = 0; k < nn; ++k) {
      out.nptr.push_back(static_cast<uint16_t>(run));
      out.inc.insert(out.inc.end(), lists[k].begin(), lists[k].end());
      run += static_cast<uint32_t>(lists[k].size());
    }
    out.nptr.resize(p0 + kTileNptr, static_cast<uint16_t>(run));
    out.inc.resize(i0 + 4ull * kTileElems, 0);
    out.info.push_back(nn | cnt << 16);
    out.tileNodes += nn;
    at = end;
  }
  return true;
}

}  // namespace pies
