// Schedule LAYERED: the sequential Gauss-Seidel sweeps of tickPBD (Src/Solver.cpp:58-75) as LDS-resident sweeps
// over breadth-first levels of the constraint graph.
//
// Nodes that share a constraint are neighbours in the constraint graph, so a breadth-first levelling puts the
// nodes of every constraint into at most two adjacent levels.  "Group l" = the constraints whose lowest level is
// l; it touches levels l and l+1 only, hence groups l and l+2 share no node: all even groups can be swept
// concurrently, then all odd groups.  Inside a group the constraints are coloured (same conflict rule as
// schedule.cpp) and run colour after colour by one workgroup that keeps the group's two levels of node records
// in LDS, so the sweep costs two launches per container instead of one per colour, and a step between two
// colours is a workgroup barrier instead of a kernel boundary.
//
// The result is that of a sequential sweep over the container in the order [phase 0: group after group, colour
// after colour][phase 1: ...] - the order pies_get_order reports and the oracle replays, bit for bit.  The
// distance container runs its even groups first and the tetrahedral container its odd groups first, so that the
// second distance phase and the first tetrahedral phase (same parity, same resident nodes) share a launch, and
// likewise the last phase of one iteration and the first phase of the next.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
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

bool build_layer_plan(pies_solver* s) {
  const uint32_t N = s->nodeCount();
  LayerPlan L;
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
  if (N == 0 || incidences == 0) return false;

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

  // ---- seeds: the end face of the body along its longest axis (levels become cross-sections) ----
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t v = 0; v < N; ++v)
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::min(lo[a], s->h_pos[3 * v + a]);
      hi[a] = std::max(hi[a], s->h_pos[3 * v + a]);
    }
  int axis = 0;
  for (int a = 1; a < 3; ++a)
    if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
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
  std::vector<uint32_t> byAxis;  // constrained nodes, ascending along the axis (ties: node index)
  for (uint32_t v = 0; v < N; ++v)
    if (start[v + 1] > start[v]) byAxis.push_back(v);
  std::stable_sort(byAxis.begin(), byAxis.end(), [&](uint32_t a, uint32_t b) { return s->h_pos[3 * a + axis] < s->h_pos[3 * b + axis]; });

  // ---- breadth-first levels, component after component ----
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  std::vector<uint32_t> level(N, kNone);
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
  if (nLevels < 2) return false;
  {  // nodes without constraints only take part in the per-node steps: spread them over the levels
    uint32_t rr = 0;
    for (uint32_t v = 0; v < N; ++v)
      if (level[v] == kNone) level[v] = rr++ % nLevels;
  }
  L.levels = nLevels;

  // ---- node list by (level, id), groups of both parities ----
  std::vector<uint32_t> levelOff(nLevels + 1, 0);
  for (uint32_t v = 0; v < N; ++v) ++levelOff[level[v] + 1];
  for (uint32_t l = 0; l < nLevels; ++l) levelOff[l + 1] += levelOff[l];
  L.nodeList.resize(N);
  std::vector<uint32_t> posInList(N);
  {
    std::vector<uint32_t> cur(levelOff.begin(), levelOff.end() - 1);
    for (uint32_t v = 0; v < N; ++v) { posInList[v] = cur[level[v]]; L.nodeList[cur[level[v]]++] = v; }
  }
  L.groups[0] = (nLevels + 1) / 2;
  L.groups[1] = nLevels / 2 + 1;
  L.groupOff[0].resize(L.groups[0] + 1);
  for (uint32_t g = 0; g <= L.groups[0]; ++g) L.groupOff[0][g] = levelOff[std::min(2 * g, nLevels)];
  L.groupOff[1].resize(L.groups[1] + 1);
  L.groupOff[1][0] = 0;
  for (uint32_t g = 1; g <= L.groups[1]; ++g) L.groupOff[1][g] = levelOff[std::min(2 * g - 1, nLevels)];
  for (int q = 0; q < 2; ++q)
    for (uint32_t g = 0; g < L.groups[q]; ++g) L.maxGroupNodes = std::max(L.maxGroupNodes, L.groupOff[q][g + 1] - L.groupOff[q][g]);
  if (L.maxGroupNodes > kLayerMaxGroupNodes) return false;  // a pair of levels does not fit in LDS

  // ---- per container: group, colour inside the group, order ----
  int rounds = 12;
  if (const char* e = std::getenv("PIES_COLOUR_ROUNDS")) rounds = std::atoi(e);
  const char* noHint = std::getenv("PIES_NO_COLOUR_HINT");
  std::vector<uint32_t> localOf(N, 0);
  Plan plans[5];
  for (int k : kinds) {
    const Ops& O = ops[k];
    LayerKind& K = L.kind[k];
    Plan& P = plans[k];
    if (O.count == 0) continue;
    // group of an op: parity and index from its lowest level (position constraints run with the even groups)
    std::vector<std::vector<uint32_t>> members[2];
    members[0].resize(L.groups[0]);
    members[1].resize(L.groups[1]);
    for (uint32_t c = 0; c < O.count; ++c) {
      uint32_t l = kNone, lmax = 0;
      for (uint32_t j = 0; j < O.stride; ++j) {
        l = std::min(l, level[O.ids[static_cast<size_t>(c) * O.stride + j]]);
        lmax = std::max(lmax, level[O.ids[static_cast<size_t>(c) * O.stride + j]]);
      }
      if (lmax - l > 1) return false;  // cannot happen for a breadth-first levelling
      if (k == PIES_POSITION) members[0][l / 2].push_back(c);
      else if (l & 1u) members[1][(l + 1) / 2].push_back(c);
      else members[0][l / 2].push_back(c);
    }
    P.order.reserve(O.count);
    K.local.reserve(static_cast<size_t>(O.count) * O.stride);
    std::vector<std::vector<uint32_t>> keys[2];
    for (int phase = 0; phase < 2; ++phase) {
      const int q = (kLayerFirstParity[k] + phase) & 1;
      keys[q].resize(L.groups[q]);
      uint32_t ncol = 0;
      for (uint32_t g = 0; g < L.groups[q]; ++g) {
        const std::vector<uint32_t>& sel = members[q][g];
        if (sel.empty()) continue;
        const uint32_t n0 = L.groupOff[q][g], m = L.groupOff[q][g + 1] - n0;
        for (uint32_t i = 0; i < m; ++i) localOf[L.nodeList[n0 + i]] = i;
        uint32_t nc = colour_group(O, sel, localOf.data(), m, rounds, keys[q][g]);
        if (nc == 0) return false;
        if (!(noHint && std::atoi(noHint)))
          for (int a = 0; a < 3; ++a) {
            std::vector<uint32_t> proposed;
            const uint32_t nh = from_hint(O, sel, O.hints[a], localOf.data(), m, proposed);
            if (nh != 0 && nh < nc) { keys[q][g].swap(proposed); nc = nh; }
          }
        ncol = std::max(ncol, nc);
      }
      K.ncol[phase] = ncol;
      K.colOff[phase].assign(static_cast<size_t>(L.groups[q]) * (ncol + 1), 0);
      for (uint32_t g = 0; g < L.groups[q]; ++g) {
        const std::vector<uint32_t>& sel = members[q][g];
        const uint32_t n0 = L.groupOff[q][g];
        std::vector<uint32_t> offs(ncol + 2, 0);
        for (uint32_t key : keys[q][g]) ++offs[key + 1];
        for (uint32_t c = 0; c <= ncol; ++c) offs[c + 1] += offs[c];
        const uint32_t slot0 = static_cast<uint32_t>(P.order.size());
        for (uint32_t c = 0; c <= ncol; ++c) K.colOff[phase][static_cast<size_t>(g) * (ncol + 1) + c] = slot0 + offs[c];
        for (uint32_t c = 0; c < ncol; ++c)
          if (offs[c + 1] > offs[c]) {
            P.batches.push_back({slot0 + offs[c], offs[c + 1] - offs[c]});
            K.maxClass = std::max(K.maxClass, offs[c + 1] - offs[c]);
          }
        std::vector<uint32_t> sorted(sel.size()), cur(offs.begin(), offs.end() - 1);
        for (size_t p = 0; p < sel.size(); ++p) sorted[cur[keys[q][g][p]]++] = sel[p];  // stable: host order in a class
        for (uint32_t c : sorted) {
          P.order.push_back(c);
          for (uint32_t j = 0; j < O.stride; ++j) K.local.push_back(posInList[O.ids[static_cast<size_t>(c) * O.stride + j]] - n0);
        }
      }
      if (k == PIES_POSITION) break;  // one phase only
    }
    if (P.order.size() != O.count) return false;
  }
  for (int k : kinds) s->plan[k] = std::move(plans[k]);
  L.active = true;
  s->layer = std::move(L);
  return true;
}

}  // namespace pies
