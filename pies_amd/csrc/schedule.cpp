// Mapping of the reference's sequential Gauss-Seidel sweeps (Src/Solver.cpp:58-75: every
// projectNodePositions call reads and immediately writes node.position, Constraints.h:121-129) onto
// conflict-free device batches.
//
// Two constraints conflict when one writes a node the other reads or writes.  The result of a
// sequential sweep depends only on the relative order of conflicting pairs, so:
//   EXACT    : batch = dependency level of the host (container) order.  Every conflicting pair keeps
//              its order => bit-identical to the sequential sweep in container order.
//   COLOURED : first-fit graph colouring.  Batches are independent sets, executed colour after
//              colour => bit-identical to a sequential sweep over the container re-ordered by colour
//              (stable), which is the order pies_get_order reports.
#include <algorithm>
#include <cstdlib>
#include <numeric>

#include "solver_state.h"

namespace pies {

static void plan_from_keys(const std::vector<uint32_t>& key, uint32_t nkeys, Plan& out) {
  const uint32_t n = static_cast<uint32_t>(key.size());
  std::vector<uint32_t> offs(nkeys + 1, 0);
  for (uint32_t c = 0; c < n; ++c) ++offs[key[c] + 1];
  for (uint32_t k = 0; k < nkeys; ++k) offs[k + 1] += offs[k];
  out.order.assign(n, 0);
  std::vector<uint32_t> cur(offs.begin(), offs.end() - 1);
  for (uint32_t c = 0; c < n; ++c) out.order[cur[key[c]]++] = c;  // stable: host order inside a batch
  out.batches.clear();
  for (uint32_t k = 0; k < nkeys; ++k)
    if (offs[k + 1] > offs[k]) out.batches.push_back({offs[k], offs[k + 1] - offs[k]});
}

static uint32_t levels(const OpView& ops, uint32_t nodeCount, std::vector<uint32_t>& key) {
  // lastW[n]: level of the last op that wrote n; lastR[n]: highest level of any op that read n.
  std::vector<uint32_t> lastW(nodeCount, 0), lastR(nodeCount, 0);  // 0 = none; levels are 1-based here
  uint32_t maxLevel = 0;
  for (uint32_t c = 0; c < ops.count; ++c) {
    const uint32_t* id = ops.ids + static_cast<size_t>(c) * ops.stride;
    uint32_t lv = 0;
    for (uint32_t k = 0; k < ops.stride; ++k) {
      const uint32_t n = id[k];
      lv = std::max(lv, lastW[n]);
      if (ops.writeMask & (1u << k)) lv = std::max(lv, lastR[n]);
    }
    ++lv;
    for (uint32_t k = 0; k < ops.stride; ++k) {
      const uint32_t n = id[k];
      if (ops.writeMask & (1u << k)) lastW[n] = lv;
      else lastR[n] = std::max(lastR[n], lv);
    }
    key[c] = lv - 1;
    maxLevel = std::max(maxLevel, lv);
  }
  return maxLevel;
}

// First-fit colouring of the ops visited in the order `visit` (nullptr = host order).
static bool colours(const OpView& ops, uint32_t nodeCount, const uint32_t* visit, std::vector<uint32_t>& key, uint32_t& ncolours) {
  constexpr int kWords = 4;  // up to 256 colours
  std::vector<uint64_t> usedW(static_cast<size_t>(nodeCount) * kWords, 0), usedR(static_cast<size_t>(nodeCount) * kWords, 0);
  ncolours = 0;
  for (uint32_t v = 0; v < ops.count; ++v) {
    const uint32_t c = visit ? visit[v] : v;
    const uint32_t* id = ops.ids + static_cast<size_t>(c) * ops.stride;
    uint64_t forbid[kWords] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < ops.stride; ++k) {
      const size_t n = static_cast<size_t>(id[k]) * kWords;
      const bool wr = ops.writeMask & (1u << k);
      for (int w = 0; w < kWords; ++w) forbid[w] |= usedW[n + w] | (wr ? usedR[n + w] : 0ull);
    }
    int col = -1;
    for (int w = 0; w < kWords && col < 0; ++w)
      if (~forbid[w]) col = w * 64 + __builtin_ctzll(~forbid[w]);
    if (col < 0) return false;
    for (uint32_t k = 0; k < ops.stride; ++k) {
      const size_t n = static_cast<size_t>(id[k]) * kWords;
      const uint64_t bit = 1ull << (col & 63);
      if (ops.writeMask & (1u << k)) usedW[n + (col >> 6)] |= bit;
      else usedR[n + (col >> 6)] |= bit;
    }
    key[c] = static_cast<uint32_t>(col);
    ncolours = std::max(ncolours, static_cast<uint32_t>(col) + 1);
  }
  return true;
}

// Iterated greedy (Culberson): re-running first-fit with the ops grouped by their current colour class can
// never use more colours, and visiting the classes in a different order (largest first / reversed) lets
// small classes dissolve into earlier ones.  Fewer colours = fewer dependent launches per sweep.
static bool colours_iterated(const OpView& ops, uint32_t nodeCount, std::vector<uint32_t>& key, uint32_t& ncolours) {
  if (!colours(ops, nodeCount, nullptr, key, ncolours)) return false;
  int rounds = 12;
  if (const char* e = std::getenv("PIES_COLOUR_ROUNDS")) rounds = std::atoi(e);
  std::vector<uint32_t> visit(ops.count), best = key, trial(ops.count);
  uint32_t bestN = ncolours;
  for (int r = 0; r < rounds; ++r) {
    // class sizes of the best colouring so far
    std::vector<uint32_t> size(bestN, 0), classOrder(bestN);
    for (uint32_t c = 0; c < ops.count; ++c) ++size[best[c]];
    std::iota(classOrder.begin(), classOrder.end(), 0u);
    if (r % 3 == 0) std::reverse(classOrder.begin(), classOrder.end());
    else if (r % 3 == 1) std::stable_sort(classOrder.begin(), classOrder.end(), [&](uint32_t a, uint32_t b) { return size[a] > size[b]; });
    else std::stable_sort(classOrder.begin(), classOrder.end(), [&](uint32_t a, uint32_t b) { return size[a] < size[b]; });
    std::vector<uint32_t> rank(bestN), offs(bestN + 1, 0);
    for (uint32_t k = 0; k < bestN; ++k) rank[classOrder[k]] = k;
    for (uint32_t c = 0; c < ops.count; ++c) ++offs[rank[best[c]] + 1];
    for (uint32_t k = 0; k < bestN; ++k) offs[k + 1] += offs[k];
    for (uint32_t c = 0; c < ops.count; ++c) visit[offs[rank[best[c]]]++] = c;  // stable inside a class
    uint32_t n = 0;
    if (!colours(ops, nodeCount, visit.data(), trial, n)) break;
    if (n <= bestN) { best.swap(trial); bestN = n; }
  }
  key.swap(best);
  ncolours = bestN;
  return true;
}

void build_plan(const OpView& ops, uint32_t nodeCount, int schedule, Plan& out) {
  out.order.clear();
  out.batches.clear();
  if (ops.count == 0) return;
  std::vector<uint32_t> key(ops.count, 0);
  uint32_t nkeys = 0;
  if (schedule < 0) {  // order-independent use (PD local step): one batch, host order
    plan_from_keys(key, 1, out);
    return;
  }
  if (schedule == PIES_SCHEDULE_COLOURED && colours_iterated(ops, nodeCount, key, nkeys)) {
    plan_from_keys(key, nkeys, out);
    return;
  }
  nkeys = levels(ops, nodeCount, key);  // EXACT, and the fallback when >256 colours would be needed
  plan_from_keys(key, nkeys, out);
}

}  // namespace pies
