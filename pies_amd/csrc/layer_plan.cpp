// Schedule LAYERED: the sequential Gauss-Seidel sweeps of tickPBD (Src/Solver.cpp:58-75) as LDS-resident sweeps
// over breadth-first levels of the constraint graph.
//
// Nodes that share a constraint are neighbours in the constraint graph, so a breadth-first levelling puts the
// nodes of every constraint into at most two adjacent levels.  "Group l" = the constraints whose lowest level is
// l; it touches levels l and l+1 only, hence groups l and l+2 share no node: all even groups can be swept
// concurrently, then all odd groups.  Inside a group the constraints are coloured (same conflict rule as
// schedule.cpp) and run colour after colour by one workgroup that keeps the group's two levels of node records
// in LDS, so the sweep costs two launches per container instead of one per colour, and a step between two
// colours is a workgroup barrier instead of a kernel boundary.  Large bodies whose levels are too wide for one
// workgroup get a second levelling across the first: a level is cut into strips, a group into tiles, and the
// tiles of equal (level parity, strip parity) form one of four phases.
//
// The result is that of a sequential sweep over the container in the order [phase 0: group after group, colour
// after colour][phase 1: ...] - the order pies_get_order reports and the oracle replays, bit for bit.  The
// distance container runs its even groups first and the tetrahedral container its odd groups first, so that the
// second distance phase and the first tetrahedral phase (same parity, same resident nodes) share a launch, and
// likewise the last phase of one iteration and the first phase of the next.
//
// Round 4 (second half).  A group's sweep lasts as long as its colours, and a group needs at least as many colours as its busiest
// node has constraints in it.  On a lattice a node's 24 tetrahedra split 12 / 12 between the group below and the group above
// it; a breadth-first levelling of an unstructured mesh puts ~83 % of a node's elements into the group below (an element
// belongs to the group of its LOWEST level, and most elements of a node reach one level down): 38 + 39 colours where the
// node degree would allow 23 + 23.  Two remedies, tried as candidate plans beside the original one (the cheapest is taken; a
// tie keeps the original):  (1) an element whose nodes all lie in ONE level may run with the group below it as well as with
// its own - both keep that level resident - and is dealt to whichever leaves the busiest of its nodes less busy;  (2) levels by
// position instead of by graph distance: slabs along the longest axis as thick as the longest constraint, so that a
// constraint still touches at most two adjacent levels and about half of them lie in a single one.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>

#include "solver_state.h"

namespace pies {
namespace {

struct Ops {  // one container: `stride` node ids per op; bit k of writeMask set when node k is written
  std::vector<uint32_t> ids;
  uint32_t stride = 1, count = 0;
  uint8_t writeMask = 1;
  std::vector<uint16_t> hints[3];  // optional proposed colourings (count entries each, kNoColourHint = none)
};

// First-fit colouring of the ops listed in `sel` (visited in the order `visit` of positions in sel) with node
// indices made local by `loc` (node -> index below `m`).  Returns the number of colours; key[] per position.
constexpr int kWords = 2;  // up to 128 colours inside a group
uint32_t first_fit(const Ops& ops, const std::vector<uint32_t>& sel, const uint32_t* visit, const uint32_t* localOf, uint32_t m,
                   std::vector<uint64_t>& usedW, std::vector<uint64_t>& usedR, std::vector<uint32_t>& key) {
  usedW.assign(static_cast<size_t>(m) * kWords, 0);
  usedR.assign(static_cast<size_t>(m) * kWords, 0);
  uint32_t ncol = 0;
  for (size_t v = 0; v < sel.size(); ++v) {
    const uint32_t p = visit ? visit[v] : static_cast<uint32_t>(v);
    const uint32_t* id = &ops.ids[static_cast<size_t>(sel[p]) * ops.stride];
    uint64_t forbid[kWords] = {0, 0};
    for (uint32_t k = 0; k < ops.stride; ++k) {
      const size_t n = static_cast<size_t>(localOf[id[k]]) * kWords;
      const bool wr = ops.writeMask & (1u << k);
      for (int w = 0; w < kWords; ++w) forbid[w] |= usedW[n + w] | (wr ? usedR[n + w] : 0ull);
    }
    int col = -1;
    for (int w = 0; w < kWords && col < 0; ++w)
      if (~forbid[w]) col = w * 64 + __builtin_ctzll(~forbid[w]);
    if (col < 0) return 0;  // more than 128 colours: give up
    for (uint32_t k = 0; k < ops.stride; ++k) {
      const size_t n = static_cast<size_t>(localOf[id[k]]) * kWords + (col >> 6);
      const uint64_t bit = 1ull << (col & 63);
      if (ops.writeMask & (1u << k)) usedW[n] |= bit;
      else usedR[n] |= bit;
    }
    key[p] = static_cast<uint32_t>(col);
    ncol = std::max(ncol, static_cast<uint32_t>(col) + 1);
  }
  return ncol;
}

// A proposed colouring of the group is taken only when every op has a proposal and no two conflicting ops share a
// colour (same rule as first_fit).  Returns the number of colours, 0 when the proposal is not usable.
uint32_t from_hint(const Ops& ops, const std::vector<uint32_t>& sel, const std::vector<uint16_t>& hint, const uint32_t* localOf,
                   uint32_t m, std::vector<uint32_t>& key) {
  if (hint.size() != ops.count) return 0;
  std::vector<uint64_t> usedW(static_cast<size_t>(m) * kWords, 0), usedR(static_cast<size_t>(m) * kWords, 0);
  uint32_t ncol = 0;
  key.assign(sel.size(), 0);
  for (size_t p = 0; p < sel.size(); ++p) {
    const uint32_t col = hint[sel[p]];
    if (col >= kWords * 64) return 0;
    const uint64_t bit = 1ull << (col & 63);
    const uint32_t* id = &ops.ids[static_cast<size_t>(sel[p]) * ops.stride];
    for (uint32_t k = 0; k < ops.stride; ++k) {
      const size_t n = static_cast<size_t>(localOf[id[k]]) * kWords + (col >> 6);
      const bool wr = ops.writeMask & (1u << k);
      if ((usedW[n] & bit) || (wr && (usedR[n] & bit))) return 0;
    }
    for (uint32_t k = 0; k < ops.stride; ++k) {
      const size_t n = static_cast<size_t>(localOf[id[k]]) * kWords + (col >> 6);
      if (ops.writeMask & (1u << k)) usedW[n] |= bit;
      else usedR[n] |= bit;
    }
    key[p] = col;
    ncol = std::max(ncol, col + 1);
  }
  return ncol;
}

// First fit, then rounds of iterated greedy (Culberson): re-colouring class by class never needs more colours.
// (DSATUR per tile was tried: never fewer colours than this on the lattice or the Delaunay beam, four times the time.)
uint32_t colour_group(const Ops& ops, const std::vector<uint32_t>& sel, const uint32_t* localOf, uint32_t m, int rounds,
                      std::vector<uint32_t>& key) {
  std::vector<uint64_t> usedW, usedR;
  key.assign(sel.size(), 0);
  uint32_t best = first_fit(ops, sel, nullptr, localOf, m, usedW, usedR, key);
  if (best == 0) return 0;
  std::vector<uint32_t> visit(sel.size()), trial(sel.size());
  for (int r = 0; r < rounds; ++r) {
    std::vector<uint32_t> size(best, 0), classOrder(best), rank(best), offs(best + 1, 0);
    for (uint32_t k : key) ++size[k];
    std::iota(classOrder.begin(), classOrder.end(), 0u);
    if (r % 3 == 0) std::reverse(classOrder.begin(), classOrder.end());
    else if (r % 3 == 1) std::stable_sort(classOrder.begin(), classOrder.end(), [&](uint32_t a, uint32_t b) { return size[a] > size[b]; });
    else std::stable_sort(classOrder.begin(), classOrder.end(), [&](uint32_t a, uint32_t b) { return size[a] < size[b]; });
    for (uint32_t k = 0; k < best; ++k) rank[classOrder[k]] = k;
    for (uint32_t k : key) ++offs[rank[k] + 1];
    for (uint32_t k = 0; k < best; ++k) offs[k + 1] += offs[k];
    for (uint32_t p = 0; p < sel.size(); ++p) visit[offs[rank[key[p]]]++] = p;  // stable inside a class
    const uint32_t n = first_fit(ops, sel, visit.data(), localOf, m, usedW, usedR, trial);
    if (n != 0 && n <= best) { key.swap(trial); best = n; }
  }
  return best;
}

}  // namespace

// PIES_LAYER_DEBUG=1: say on stderr why a scene was handed back to the coloured schedule
static bool layer_reject(const char* why) {
  if (const char* e = std::getenv("PIES_LAYER_DEBUG"); e && e[0] == '1') std::fprintf(stderr, "[pies] schedule LAYERED not used: %s\n", why);
  return false;
}

bool build_layer_plan(pies_solver* s) {
  const uint32_t N = s->nodeCount();
  // ---- the containers as id lists ----
  Ops ops[5];
  ops[PIES_POSITION].stride = 1; ops[PIES_POSITION].writeMask = 0x1;
  for (const HostPosition& c : s->h_position) ops[PIES_POSITION].ids.push_back(c.id);
  ops[PIES_DISTANCE].stride = 2; ops[PIES_DISTANCE].writeMask = 0x1;  // node a moves, node b is read (Constraints.cpp:34-36)
  for (const HostDistance& c : s->h_distance) { ops[PIES_DISTANCE].ids.push_back(c.ids[0]); ops[PIES_DISTANCE].ids.push_back(c.ids[1]); }
  ops[PIES_TET].stride = 4; ops[PIES_TET].writeMask = 0xF;
  for (const HostTet& c : s->h_tet) {
    ops[PIES_TET].ids.insert(ops[PIES_TET].ids.end(), c.ids, c.ids + 4);
    for (int a = 0; a < 3; ++a) ops[PIES_TET].hints[a].push_back(c.layerHint[a]);
  }
  ops[PIES_BEND].stride = 4; ops[PIES_BEND].writeMask = 0xF;
  for (const HostBend& c : s->h_bend) ops[PIES_BEND].ids.insert(ops[PIES_BEND].ids.end(), c.ids, c.ids + 4);
  const int kinds[4] = {PIES_POSITION, PIES_DISTANCE, PIES_TET, PIES_BEND};
  for (int k : kinds) ops[k].count = static_cast<uint32_t>(ops[k].ids.size() / ops[k].stride);
  const int linking[3] = {PIES_DISTANCE, PIES_TET, PIES_BEND};
  size_t incidences = 0;
  for (int k : linking) incidences += ops[k].ids.size();
  if (N == 0 || incidences == 0) return layer_reject("no distance / tetrahedral / bend constraints");

  // ---- node -> incident constraints (all linking containers in one list) ----
  std::vector<uint32_t> start(N + 1, 0);
  for (int k : linking)
    for (uint32_t v : ops[k].ids) ++start[v + 1];
  for (uint32_t v = 0; v < N; ++v) start[v + 1] += start[v];
  std::vector<uint32_t> inc(incidences);  // constraint handle: index into the concatenated containers
  uint32_t base[5] = {0, 0, 0, 0, 0};
  {
    std::vector<uint32_t> cur(start.begin(), start.end() - 1);
    uint32_t b = 0;
    for (int k : linking) {
      base[k] = b;
      for (uint32_t c = 0; c < ops[k].count; ++c)
        for (uint32_t j = 0; j < ops[k].stride; ++j) inc[cur[ops[k].ids[static_cast<size_t>(c) * ops[k].stride + j]]++] = b + c;
      b += ops[k].count;
    }
  }
  const uint32_t totalOps = base[PIES_BEND] + ops[PIES_BEND].count;
  auto nodes_of = [&](uint32_t handle, const uint32_t*& ids, uint32_t& stride) {
    const int k = handle >= base[PIES_BEND] ? PIES_BEND : handle >= base[PIES_TET] ? PIES_TET : PIES_DISTANCE;
    stride = ops[k].stride;
    ids = &ops[k].ids[static_cast<size_t>(handle - base[k]) * stride];
  };

  // ---- bounding box, mean edge length ----
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t v = 0; v < N; ++v)
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::min(lo[a], s->h_pos[3 * v + a]);
      hi[a] = std::max(hi[a], s->h_pos[3 * v + a]);
    }
  int axes[3] = {0, 1, 2};  // by decreasing extent
  std::stable_sort(axes, axes + 3, [&](int a, int b) { return hi[a] - lo[a] > hi[b] - lo[b]; });
  double edgeSum = 0.0;
  size_t edges = 0;
  for (int k : linking)
    for (uint32_t c = 0; c < ops[k].count; ++c) {
      const uint32_t* id = &ops[k].ids[static_cast<size_t>(c) * ops[k].stride];
      double d2 = 0.0;
      for (int a = 0; a < 3; ++a) {
        const double d = static_cast<double>(s->h_pos[3 * id[0] + a]) - s->h_pos[3 * id[1] + a];
        d2 += d * d;
      }
      if (std::isfinite(d2)) { edgeSum += std::sqrt(d2); ++edges; }
    }
  const float slack = edges ? static_cast<float>(0.45 * edgeSum / static_cast<double>(edges)) : 0.0f;

  // ---- breadth-first levels from the end face along `axis` (levels become cross-sections), component after component;
  //      nodes without constraints only take part in the per-node steps and are spread over the levels ----
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  auto bfs = [&](int axis, std::vector<uint32_t>& level) -> uint32_t {
    std::vector<uint32_t> byAxis;  // constrained nodes, ascending along the axis (ties: node index)
    for (uint32_t v = 0; v < N; ++v)
      if (start[v + 1] > start[v]) byAxis.push_back(v);
    std::stable_sort(byAxis.begin(), byAxis.end(), [&](uint32_t a, uint32_t b) { return s->h_pos[3 * a + axis] < s->h_pos[3 * b + axis]; });
    level.assign(N, kNone);
    std::vector<uint8_t> seen(totalOps, 0);
    std::vector<uint32_t> frontier, next;
    uint32_t nLevels = 0;
    for (size_t cursor = 0; cursor < byAxis.size(); ++cursor) {
      if (level[byAxis[cursor]] != kNone) continue;
      const float c0 = s->h_pos[3 * byAxis[cursor] + axis];
      frontier.clear();
      for (size_t j = cursor; j < byAxis.size() && !(s->h_pos[3 * byAxis[j] + axis] > c0 + slack); ++j)
        if (level[byAxis[j]] == kNone) { level[byAxis[j]] = nLevels; frontier.push_back(byAxis[j]); }
      while (!frontier.empty()) {
        next.clear();
        for (uint32_t v : frontier)
          for (uint32_t e = start[v]; e < start[v + 1]; ++e) {
            if (seen[inc[e]]) continue;
            seen[inc[e]] = 1;
            const uint32_t* id;
            uint32_t stride;
            nodes_of(inc[e], id, stride);
            for (uint32_t j = 0; j < stride; ++j)
              if (level[id[j]] == kNone) { level[id[j]] = nLevels + 1; next.push_back(id[j]); }
          }
        ++nLevels;
        frontier.swap(next);
      }
    }
    if (nLevels == 0) return 0;
    uint32_t rr = 0;
    for (uint32_t v = 0; v < N; ++v)
      if (level[v] == kNone) level[v] = rr++ % nLevels;
    return nLevels;
  };
  // ---- one candidate plan: tiles, groups and colours for a first levelling; its cost = the colour steps of a sweep ----
  struct Candidate {
    LayerPlan L;
    Plan plans[5];
    uint64_t cost = ~0ull;
    const char* why = nullptr;
    const char* name = "";
  };
  auto reject = [](Candidate& C, const char* why) { C.why = why; return false; };
  auto attempt = [&](const std::vector<uint32_t>& level1, const uint32_t L1, const bool balance, Candidate& C) -> bool {
  LayerPlan& L = C.L;
  std::vector<uint32_t> level2;
  if (L1 < 2) return reject(C, "fewer than two levels");
  L.levels = L1;

  // ---- one strip (a pair of levels fits a workgroup) or strips of a second levelling across the first ----
  uint32_t maxPair = 0;
  {
    std::vector<uint32_t> cnt(L1 + 1, 0);
    for (uint32_t v = 0; v < N; ++v) ++cnt[level1[v]];
    for (uint32_t l = 0; l < L1; ++l) maxPair = std::max(maxPair, cnt[l] + cnt[l + 1]);
  }
  // Measured on the box (substeps/s, one strip / strips / coloured): 20x20x250 680 / - / 333; 50x50x40 290 / 229 / 347;
  // 60x60x100 183 / 235 / 227; 100^3 - / 156 / 116.  One strip pays while a pair of levels is a few waves per colour
  // (long bodies); strips pay once the body is large enough to fill the chip four phases at a time; squat bodies in
  // between are left to the coloured schedule.
  uint32_t oneStripMax = 2560, tileTarget = 1280, stripsMinNodes = 300000;
  if (const char* e = tuning_env("PIES_LAYER_ONE_STRIP_MAX")) oneStripMax = static_cast<uint32_t>(std::atoi(e));
  if (const char* e = tuning_env("PIES_LAYER_TILE_NODES")) tileTarget = static_cast<uint32_t>(std::atoi(e));
  if (const char* e = tuning_env("PIES_LAYER_STRIPS_MIN_NODES")) stripsMinNodes = static_cast<uint32_t>(std::atoi(e));
  tileTarget = std::min(tileTarget, kLayerMaxGroupNodes);
  uint32_t L2 = 1, width = 1;
  if (maxPair <= std::min(oneStripMax, kLayerMaxGroupNodes)) {
    level2.assign(N, 0);
  } else {
    if (N < stripsMinNodes) return reject(C, "a pair of levels is too wide for one workgroup and the body too small for strips");
    L2 = bfs(axes[1], level2);
    if (L2 == 0) return reject(C, "no second levelling");
    // nodes per (level1, level2) cell, prefix sums along level2
    std::vector<uint32_t> pre(static_cast<size_t>(L1 + 1) * (L2 + 1), 0);
    for (uint32_t v = 0; v < N; ++v) ++pre[static_cast<size_t>(level1[v]) * (L2 + 1) + level2[v] + 1];
    for (uint32_t l = 0; l < L1; ++l)
      for (uint32_t m = 0; m < L2; ++m) pre[static_cast<size_t>(l) * (L2 + 1) + m + 1] += pre[static_cast<size_t>(l) * (L2 + 1) + m];
    auto tile_max = [&](uint32_t w) {
      uint32_t worst = 0;
      for (uint32_t l = 0; l < L1; ++l)
        for (uint32_t t = 0; t * w < L2; ++t) {
          const uint32_t m0 = t * w, m1 = std::min((t + 1) * w + 1, L2);  // the strip and the first level of the next one
          uint32_t n = pre[static_cast<size_t>(l) * (L2 + 1) + m1] - pre[static_cast<size_t>(l) * (L2 + 1) + m0];
          if (l + 1 < L1) n += pre[static_cast<size_t>(l + 1) * (L2 + 1) + m1] - pre[static_cast<size_t>(l + 1) * (L2 + 1) + m0];
          worst = std::max(worst, n);
        }
      return worst;
    };
    width = 0;
    for (uint32_t w = L2; w >= 1; --w)
      if (tile_max(w) <= tileTarget) { width = w; break; }
    if (width == 0) {
      if (tile_max(1) > kLayerMaxGroupNodes) return reject(C, "a tile of width 1 does not fit in LDS");
      width = 1;
    }
  }
  const uint32_t S = (L2 + width - 1) / width;  // strips
  L.strips = S;
  L.width = width;

  // ---- node list by (level1, level2, id) ----
  std::vector<uint32_t> cellOff(static_cast<size_t>(L1) * L2 + 1, 0);
  for (uint32_t v = 0; v < N; ++v) ++cellOff[static_cast<size_t>(level1[v]) * L2 + level2[v] + 1];
  for (size_t c = 0; c < static_cast<size_t>(L1) * L2; ++c) cellOff[c + 1] += cellOff[c];
  L.nodeList.resize(N);
  std::vector<uint32_t> posInList(N);
  {
    std::vector<uint32_t> cur(cellOff.begin(), cellOff.end() - 1);
    for (uint32_t v = 0; v < N; ++v) {
      uint32_t& c = cur[static_cast<size_t>(level1[v]) * L2 + level2[v]];
      posInList[v] = c;
      L.nodeList[c++] = v;
    }
  }

  // ---- tiles: phase = 2 * (level parity) + (strip parity); odd level parity starts with the tile "level 0 alone"
  //      (it only matters with one strip, where the per-node steps may run with either parity) ----
  const uint32_t G[2] = {(L1 + 1) / 2, L1 / 2 + 1}, Sp[2] = {(S + 1) / 2, S / 2};
  auto run_of = [&](int64_t l, uint32_t t, uint32_t& first, uint32_t& count) {
    first = count = 0;
    if (l < 0 || l >= static_cast<int64_t>(L1)) return;
    const uint32_t m0 = t * width, m1 = std::min((t + 1) * width + 1, L2);
    first = cellOff[static_cast<size_t>(l) * L2 + m0];
    count = cellOff[static_cast<size_t>(l) * L2 + m1] - first;
  };
  for (int p1 = 0; p1 < 2; ++p1)
    for (int p2 = 0; p2 < 2; ++p2) {
      std::vector<LayerTile>& T = L.tiles[2 * p1 + p2];
      T.resize(static_cast<size_t>(G[p1]) * Sp[p2]);
      for (uint32_t gi = 0; gi < G[p1]; ++gi)
        for (uint32_t ti = 0; ti < Sp[p2]; ++ti) {
          const int64_t l = p1 == 0 ? 2 * static_cast<int64_t>(gi) : 2 * static_cast<int64_t>(gi) - 1;
          LayerTile& tile = T[static_cast<size_t>(gi) * Sp[p2] + ti];
          run_of(l, 2 * ti + p2, tile.first0, tile.count0);
          run_of(l + 1, 2 * ti + p2, tile.first1, tile.count1);
          if (tile.count0 == 0) { tile.first0 = tile.first1; tile.count0 = tile.count1; tile.first1 = tile.count1 = 0; }
          L.maxGroupNodes = std::max(L.maxGroupNodes, tile.count0 + tile.count1);
        }
    }
  if (L.maxGroupNodes > kLayerMaxGroupNodes) return reject(C, "tile larger than the LDS budget");

  // ---- per container: tile of every op, colouring inside the tile, execution order ----
  int rounds = 12;
  if (const char* e = tuning_env("PIES_COLOUR_ROUNDS")) rounds = std::atoi(e);
  const char* noHint = tuning_env("PIES_NO_COLOUR_HINT");
  std::vector<uint32_t> localOf(N, 0);
  auto local_index = [&](const LayerTile& tile, uint32_t v) {
    const uint32_t p = posInList[v];
    return p >= tile.first0 && p < tile.first0 + tile.count0 ? p - tile.first0 : tile.count0 + (p - tile.first1);
  };
  Plan* plans = C.plans;
  // The cost of a plan = the time of a sweep in units of a distance projection's step / 16: a phase lasts as long as its slowest
  // tile, a tile as long as its colour steps, and a colour step as long as one projection of its kind (a tetrahedron's or a bend's
  // ~ 4 distance projections) issued by the wavefronts that share a SIMD (measured, profiles/r05_svd_overlap.txt: 1 / 1.5 / 2.06 /
  // 2.55 for 1 / 2 / 3 / 4 wavefronts per SIMD; round 5: thick slabs with 74 colour steps ran at 210 substeps/s against 270 for
  // breadth-first levels with 71, because their classes are twice as large).
  static const uint64_t weight[5] = {1, 1, 4, 0, 4};
  auto step_time = [](uint32_t n) -> uint64_t {
    const uint32_t w = ((n + 63) / 64 + 3) / 4;  // wavefronts per SIMD of a compute unit
    static const uint64_t t[5] = {0, 16, 24, 33, 41};
    return w <= 4 ? t[w] : 41 + 10ull * (w - 4);
  };
  C.cost = 0;
  for (int k : kinds) {
    const Ops& O = ops[k];
    LayerKind& K = L.kind[k];
    Plan& P = plans[k];
    if (O.count == 0) continue;
    if (k == PIES_POSITION && S > 1) {
      // with strips the per-node steps are launches of their own over the level-ordered copy: the j-th constraint of a
      // node goes to batch j (constraints of one node keep their order, constraints of different nodes commute)
      std::vector<uint32_t> seenCount(N, 0), key(O.count);
      uint32_t nb = 0;
      for (uint32_t c = 0; c < O.count; ++c) { key[c] = seenCount[O.ids[c]]++; nb = std::max(nb, key[c] + 1); }
      std::vector<uint32_t> offs(nb + 1, 0);
      for (uint32_t c = 0; c < O.count; ++c) ++offs[key[c] + 1];
      for (uint32_t b = 0; b < nb; ++b) offs[b + 1] += offs[b];
      P.order.assign(O.count, 0);
      std::vector<uint32_t> cur(offs.begin(), offs.end() - 1);
      for (uint32_t c = 0; c < O.count; ++c) P.order[cur[key[c]]++] = c;
      for (uint32_t b = 0; b < nb; ++b) P.batches.push_back({offs[b], offs[b + 1] - offs[b]});
      for (uint32_t c : P.order) K.local.push_back(posInList[O.ids[c]]);  // index into the level-ordered copy
      continue;
    }
    std::vector<std::vector<uint32_t>> members[4];
    for (int ph = 0; ph < 4; ++ph) members[ph].resize(L.tiles[ph].size());
    // group of every op: the lowest level of its nodes - or, with `balance`, for an op whose nodes share ONE level, that level or
    // the one below, whichever leaves the busiest of its nodes less busy (below[v] / above[v]: ops of this container that node v
    // meets as a node of its group's upper / lower level; the ops that span two levels are counted first)
    std::vector<uint32_t> groupOf(O.count), stripOf(O.count);
    std::vector<uint8_t> alone(O.count, 0);  // dealt to the tile "level 0 alone" (the group below group 0, first tile of the odd parity)
    std::vector<uint32_t> below, above;
    if (balance && k != PIES_POSITION) { below.assign(N, 0); above.assign(N, 0); }
    for (uint32_t c = 0; c < O.count; ++c) {
      uint32_t l = kNone, lmax = 0, m = kNone, mmax = 0;
      for (uint32_t j = 0; j < O.stride; ++j) {
        const uint32_t v = O.ids[static_cast<size_t>(c) * O.stride + j];
        l = std::min(l, level1[v]); lmax = std::max(lmax, level1[v]);
        m = std::min(m, level2[v]); mmax = std::max(mmax, level2[v]);
      }
      if (lmax - l > 1 || mmax - m > 1) return reject(C, "a constraint spans more than two levels");  // cannot happen for breadth-first levellings
      groupOf[c] = l;
      stripOf[c] = m / width;
      if (!below.empty() && lmax != l)
        for (uint32_t j = 0; j < O.stride; ++j) {
          const uint32_t v = O.ids[static_cast<size_t>(c) * O.stride + j];
          ++(level1[v] == l ? above[v] : below[v]);
        }
    }
    if (!below.empty())
      for (uint32_t c = 0; c < O.count; ++c) {
        const uint32_t* id = &O.ids[static_cast<size_t>(c) * O.stride];
        const uint32_t l = groupOf[c];
        bool single = true;
        for (uint32_t j = 0; j < O.stride; ++j) single = single && level1[id[j]] == l;
        if (!single) continue;
        uint32_t stay = 0, down = 0;
        for (uint32_t j = 0; j < O.stride; ++j) { stay = std::max(stay, above[id[j]]); down = std::max(down, below[id[j]]); }
        const bool goDown = down < stay;
        if (goDown && l >= 1) groupOf[c] = l - 1;
        else if (goDown) alone[c] = 1;
        for (uint32_t j = 0; j < O.stride; ++j) ++(goDown ? below[id[j]] : above[id[j]]);
      }
    if (!below.empty())
      if (const char* e = std::getenv("PIES_LAYER_DEBUG"); e && e[0] == '1') {
        uint32_t mb = 0, ma = 0;
        for (uint32_t v = 0; v < N; ++v) { mb = std::max(mb, below[v]); ma = std::max(ma, above[v]); }
        std::fprintf(stderr, "[pies]   container %d: busiest node meets %u ops from the level below, %u from its own\n", k, mb, ma);
      }
    for (uint32_t c = 0; c < O.count; ++c) {
      const uint32_t l = groupOf[c], t = stripOf[c];
      if (k == PIES_POSITION) { members[0][static_cast<size_t>(l / 2) * Sp[0]].push_back(c); continue; }  // one strip: with the even tiles
      const int p1 = alone[c] ? 1 : static_cast<int>(l & 1u), p2 = t & 1u;
      const uint32_t gi = alone[c] ? 0u : p1 ? (l + 1) / 2 : l / 2;
      members[2 * p1 + p2][static_cast<size_t>(gi) * Sp[p2] + t / 2].push_back(c);
    }
    P.order.reserve(O.count);
    K.local.reserve(static_cast<size_t>(O.count) * O.stride);
    for (int step = 0; step < 4; ++step) {
      const int ph = kLayerPhaseOrder[k][step];
      const std::vector<LayerTile>& T = L.tiles[ph];
      std::vector<std::vector<uint32_t>> keys(T.size());
      uint32_t ncol = 0;
      for (size_t g = 0; g < T.size(); ++g) {
        const std::vector<uint32_t>& sel = members[ph][g];
        if (sel.empty()) continue;
        const uint32_t m = T[g].count0 + T[g].count1;
        for (uint32_t i = 0; i < T[g].count0; ++i) localOf[L.nodeList[T[g].first0 + i]] = i;
        for (uint32_t i = 0; i < T[g].count1; ++i) localOf[L.nodeList[T[g].first1 + i]] = T[g].count0 + i;
        uint32_t nc = colour_group(O, sel, localOf.data(), m, rounds, keys[g]);
        if (nc == 0) return reject(C, "more than 128 colours inside a tile");
        if (!(noHint && std::atoi(noHint)))
          for (int a = 0; a < 3; ++a) {
            std::vector<uint32_t> proposed;
            const uint32_t nh = from_hint(O, sel, O.hints[a], localOf.data(), m, proposed);
            if (nh != 0 && nh < nc) { keys[g].swap(proposed); nc = nh; }
          }
        ncol = std::max(ncol, nc);
      }
      K.ncol[ph] = ncol;
      if (ncol == 0) continue;  // no constraint of this container in this phase
      K.colOff[ph].assign(T.size() * (ncol + 1), 0);
      uint64_t slowest = 0;
      for (size_t g = 0; g < T.size(); ++g) {
        const std::vector<uint32_t>& sel = members[ph][g];
        std::vector<uint32_t> offs(ncol + 2, 0);
        for (uint32_t key : keys[g]) ++offs[key + 1];
        for (uint32_t c = 0; c <= ncol; ++c) offs[c + 1] += offs[c];
        uint64_t tileTime = 0;
        for (uint32_t c = 0; c < ncol; ++c) tileTime += step_time(offs[c + 1] - offs[c]);
        slowest = std::max(slowest, tileTime);
        const uint32_t slot0 = static_cast<uint32_t>(P.order.size());
        for (uint32_t c = 0; c <= ncol; ++c) K.colOff[ph][g * (ncol + 1) + c] = slot0 + offs[c];
        for (uint32_t c = 0; c < ncol; ++c)
          if (offs[c + 1] > offs[c]) {
            P.batches.push_back({slot0 + offs[c], offs[c + 1] - offs[c]});
            K.maxClass = std::max(K.maxClass, offs[c + 1] - offs[c]);
          }
        std::vector<uint32_t> sorted(sel.size()), cur(offs.begin(), offs.end() - 1);
        for (size_t p = 0; p < sel.size(); ++p) sorted[cur[keys[g][p]]++] = sel[p];  // stable: host order in a class
        for (uint32_t c : sorted) {
          P.order.push_back(c);
          for (uint32_t j = 0; j < O.stride; ++j) K.local.push_back(local_index(T[g], O.ids[static_cast<size_t>(c) * O.stride + j]));
        }
      }
      C.cost += weight[k] * slowest;
    }
    if (P.order.size() != O.count) return reject(C, "internal: incomplete order");
  }
  return true;
  };

  // ---- the candidates ----
  std::vector<uint32_t> levelBfs, levelSlab;
  const uint32_t L1bfs = bfs(axes[0], levelBfs);
  int want = 2;  // PIES_LAYER_PLAN: 0 = the original plan only, 1 = + the balanced groups, 2 = + slabs by position (default)
  if (const char* e = tuning_env("PIES_LAYER_PLAN")) want = std::atoi(e);
  std::vector<std::unique_ptr<Candidate>> cands;
  auto run = [&](const char* name, const std::vector<uint32_t>& lev, uint32_t nl, bool balance) {
    cands.push_back(std::make_unique<Candidate>());
    cands.back()->name = name;
    if (!attempt(lev, nl, balance, *cands.back())) cands.back()->cost = ~0ull;
  };
  run("breadth-first levels", levelBfs, L1bfs, false);
  // A plan whose groups are balanced already - no phase of a container runs more colours than half the constraints at that
  // container's busiest node, + 1 - cannot gain from the other candidates (a lattice: 12 + 12 colours for 24 tetrahedra at a node);
  // planning a second and a third time is then skipped (1M particles: 5.5 s instead of 13.5 s at finalize)
  if (want >= 1 && cands[0]->cost != ~0ull && !tuning_env("PIES_LAYER_PLAN_FORCE")) {
    bool balanced = true;
    for (int k : linking) {
      if (ops[k].count == 0) continue;
      std::vector<uint32_t> deg(N, 0);
      for (uint32_t v : ops[k].ids) ++deg[v];
      const uint32_t half = (*std::max_element(deg.begin(), deg.end()) + 1u) / 2u;
      for (int ph = 0; ph < 4; ++ph) balanced = balanced && cands[0]->L.kind[k].ncol[ph] <= half + 1u;
    }
    if (balanced) want = 0;
  }
  if (want >= 1) run("breadth-first levels, single-level constraints dealt to either group", levelBfs, L1bfs, true);
  if (want >= 2) {
    // slabs along the longest axis, as thick as the longest constraint (times PIES_LAYER_SLAB, default 1): a constraint touches
    // at most two adjacent slabs
    const int axis = axes[0];
    double longest = 0.0;
    for (int k : linking)
      for (uint32_t c = 0; c < ops[k].count; ++c) {
        const uint32_t* id = &ops[k].ids[static_cast<size_t>(c) * ops[k].stride];
        float mn = INFINITY, mx = -INFINITY;
        for (uint32_t j = 0; j < ops[k].stride; ++j) { mn = std::min(mn, s->h_pos[3 * id[j] + axis]); mx = std::max(mx, s->h_pos[3 * id[j] + axis]); }
        if (std::isfinite(mx - mn)) longest = std::max(longest, static_cast<double>(mx) - mn);
      }
    double scale = 1.0;
    if (const char* e = tuning_env("PIES_LAYER_SLAB")) scale = std::max(1.0, std::atof(e));
    double offset = 0.0;  // the first slab is this fraction of a thickness thinner
    if (const char* e = tuning_env("PIES_LAYER_SLAB_OFFSET")) offset = std::min(0.999, std::max(0.0, std::atof(e)));
    const double d = longest * scale * 1.000001 + 1.0e-6;
    const double extent = static_cast<double>(hi[axis]) - lo[axis];
    if (std::isfinite(extent) && longest > 0.0 && extent / d < 1.0e6) {
      levelSlab.assign(N, 0);
      uint32_t nl = 0;
      for (uint32_t v = 0; v < N; ++v) {
        const double z = static_cast<double>(s->h_pos[3 * v + axis]) - lo[axis];
        levelSlab[v] = std::isfinite(z) ? static_cast<uint32_t>(z / d + offset) : 0u;
        nl = std::max(nl, levelSlab[v] + 1);
      }
      // (two nodes of a constraint are at most `longest` < d apart: their slabs differ by at most one)
      // the candidates are compared by colour steps alone, so slabs must not buy theirs with launches of empty or
      // countless tiles - several bodies apart along the axis leave slabs without a node, a thin long body makes thousands: the
      // candidate only runs when every slab holds a node and there are at most twice as many as breadth-first levels.
      std::vector<uint8_t> seen(nl, 0);
      for (uint32_t v = 0; v < N; ++v) seen[levelSlab[v]] = 1;
      const bool noneEmpty = std::find(seen.begin(), seen.end(), uint8_t(0)) == seen.end();
      if ((noneEmpty && nl <= 2u * L1bfs + 2u) || tuning_env("PIES_LAYER_PLAN_FORCE"))
        run("slabs along the longest axis, single-level constraints dealt to either group", levelSlab, nl, true);
    }
  }
  size_t best = 0;  // (the original plan unless another one saves a twentieth of its colour steps: a lattice keeps its order)
  for (size_t i = 1; i < cands.size(); ++i)
    if (cands[i]->cost < cands[best]->cost && (cands[0]->cost == ~0ull || 20 * cands[i]->cost <= 19 * cands[0]->cost)) best = i;
  if (const char* e = tuning_env("PIES_LAYER_PLAN_FORCE")) {  // tests: take candidate i whatever it costs (if it is a plan at all)
    const size_t i = static_cast<size_t>(std::max(0, std::atoi(e)));
    if (i < cands.size() && cands[i]->cost != ~0ull) best = i;
  }
  if (const char* e = std::getenv("PIES_LAYER_DEBUG"); e && e[0] == '1')
    for (size_t i = 0; i < cands.size(); ++i)
      std::fprintf(stderr, "[pies] LAYERED candidate %zu (%s): %s%s, cost %llu (distance %u + %u, tetrahedra %u + %u colours, %u levels)%s\n", i,
                   cands[i]->name, cands[i]->why ? "rejected: " : "ok", cands[i]->why ? cands[i]->why : "",
                   static_cast<unsigned long long>(cands[i]->cost), cands[i]->L.kind[PIES_DISTANCE].ncol[0], cands[i]->L.kind[PIES_DISTANCE].ncol[2],
                   cands[i]->L.kind[PIES_TET].ncol[0], cands[i]->L.kind[PIES_TET].ncol[2], cands[i]->L.levels, i == best ? "  <- taken" : "");
  if (cands[best]->cost == ~0ull) return layer_reject(cands[0]->why ? cands[0]->why : "no plan");
  LayerPlan& L = cands[best]->L;
  Plan* plans = cands[best]->plans;
  if (const char* e = std::getenv("PIES_LAYER_DEBUG"); e && e[0] == '1') {
    std::fprintf(stderr, "[pies] schedule LAYERED: %u levels, %u strip(s) of width %u, largest tile %u nodes\n", L.levels, L.strips, L.width, L.maxGroupNodes);
    for (int k : kinds)
      std::fprintf(stderr, "[pies]   container %d: colours per phase %u %u %u %u, largest class %u\n", k, L.kind[k].ncol[0], L.kind[k].ncol[1],
                   L.kind[k].ncol[2], L.kind[k].ncol[3], L.kind[k].maxClass);
  }
  for (int k : kinds) s->plan[k] = std::move(plans[k]);
  L.active = true;
  s->layer = std::move(L);
  return true;
}

}  // namespace pies
