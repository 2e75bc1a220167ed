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

// A colouring proposed by the lattice factories (scene.cpp) is taken only after it has been checked here, op by
// op, against the same conflict rule first-fit uses; any unhinted op or clash rejects the whole proposal.
static bool colours_from_hint(const OpView& ops, uint32_t nodeCount, std::vector<uint32_t>& key, uint32_t& ncolours) {
  if (!ops.hint) return false;
  constexpr int kWords = 4;
  std::vector<uint64_t> usedW(static_cast<size_t>(nodeCount) * kWords, 0), usedR(static_cast<size_t>(nodeCount) * kWords, 0);
  ncolours = 0;
  for (uint32_t c = 0; c < ops.count; ++c) {
    const uint32_t col = ops.hint[c];
    if (col >= kWords * 64) return false;
    const uint64_t bit = 1ull << (col & 63);
    const uint32_t* id = ops.ids + static_cast<size_t>(c) * ops.stride;
    for (uint32_t k = 0; k < ops.stride; ++k) {
      const size_t n = static_cast<size_t>(id[k]) * kWords + (col >> 6);
      const bool wr = ops.writeMask & (1u << k);
      if ((usedW[n] & bit) || (wr && (usedR[n] & bit))) return false;
    }
    for (uint32_t k = 0; k < ops.stride; ++k) {
      const size_t n = static_cast<size_t>(id[k]) * kWords + (col >> 6);
      if (ops.writeMask & (1u << k)) usedW[n] |= bit;
      else usedR[n] |= bit;
    }
    key[c] = col;
    ncolours = std::max(ncolours, col + 1);
  }
  return true;
}

// DSATUR (Brelaz): always colour the op whose neighbours already use the most distinct colours (ties: most
// conflicting neighbours, then host order), with the smallest free colour.  On regular meshes the saturation
// front sweeps through the lattice and finds the periodic optimum first-fit misses (24 colours for the
// six-tetrahedra-per-cell box, the maximum number of tetrahedra at a node, where first-fit needs 30).
static bool colours_dsatur(const OpView& ops, uint32_t nodeCount, std::vector<uint32_t>& key, uint32_t& ncolours) {
  constexpr int kWords = 4;
  const uint32_t n = ops.count, st = ops.stride;
  // node -> incident (op, slot) lists
  std::vector<uint32_t> start(nodeCount + 1, 0);
  for (size_t e = 0; e < static_cast<size_t>(n) * st; ++e) ++start[ops.ids[e] + 1];
  for (uint32_t v = 0; v < nodeCount; ++v) start[v + 1] += start[v];
  std::vector<uint32_t> inc(static_cast<size_t>(n) * st);
  {
    std::vector<uint32_t> cur(start.begin(), start.end() - 1);
    for (uint32_t c = 0; c < n; ++c)
      for (uint32_t k = 0; k < st; ++k) inc[cur[ops.ids[static_cast<size_t>(c) * st + k]]++] = c * st + k;
  }
  auto writes = [&](uint32_t slot) { return (ops.writeMask >> slot) & 1u; };
  std::vector<uint64_t> forbid(static_cast<size_t>(n) * kWords, 0);
  std::vector<uint32_t> sat(n, 0), deg(n, 0);
  for (uint32_t c = 0; c < n; ++c)
    for (uint32_t k = 0; k < st; ++k) {
      const uint32_t v = ops.ids[static_cast<size_t>(c) * st + k];
      deg[c] += start[v + 1] - start[v] - 1;  // upper bound (read-read pairs included): tie-break only
    }
  // bucket queue on saturation with lazy deletion; inside a bucket a heap on (degree desc, index asc)
  using Item = std::pair<uint32_t, uint32_t>;  // (degree, ~index) max-heap
  std::vector<std::vector<Item>> bucket(kWords * 64 + 1);
  for (uint32_t c = 0; c < n; ++c) bucket[0].push_back({deg[c], ~c});
  std::make_heap(bucket[0].begin(), bucket[0].end());
  std::vector<uint8_t> done(n, 0);
  int top = 0;
  ncolours = 0;
  for (uint32_t coloured = 0; coloured < n;) {
    while (top >= 0 && bucket[top].empty()) --top;
    if (top < 0) return false;
    std::pop_heap(bucket[top].begin(), bucket[top].end());
    const uint32_t c = ~bucket[top].back().second;
    bucket[top].pop_back();
    if (done[c] || sat[c] != static_cast<uint32_t>(top)) continue;  // stale entry
    const uint64_t* f = &forbid[static_cast<size_t>(c) * kWords];
    int col = -1;
    for (int w = 0; w < kWords && col < 0; ++w)
      if (~f[w]) col = w * 64 + __builtin_ctzll(~f[w]);
    if (col < 0) return false;
    done[c] = 1;
    ++coloured;
    key[c] = static_cast<uint32_t>(col);
    ncolours = std::max(ncolours, static_cast<uint32_t>(col) + 1);
    const uint64_t bit = 1ull << (col & 63);
    for (uint32_t k = 0; k < st; ++k) {
      const uint32_t v = ops.ids[static_cast<size_t>(c) * st + k];
      const bool wr = writes(k);
      for (uint32_t e = start[v]; e < start[v + 1]; ++e) {
        const uint32_t d = inc[e] / st;
        if (done[d] || !(wr || writes(inc[e] % st))) continue;
        uint64_t& word = forbid[static_cast<size_t>(d) * kWords + (col >> 6)];
        if (word & bit) continue;
        word |= bit;
        const uint32_t sd = ++sat[d];
        bucket[sd].push_back({deg[d], ~d});
        std::push_heap(bucket[sd].begin(), bucket[sd].end());
        if (static_cast<int>(sd) > top) top = static_cast<int>(sd);
      }
    }
  }
  return true;
}

// Iterated greedy (Culberson): re-running first-fit with the ops grouped by their current colour class can
// never use more colours, and visiting the classes in a different order (largest first / reversed) lets
// small classes dissolve into earlier ones.  Fewer colours = fewer dependent launches per sweep.
static bool colours_iterated(const OpView& ops, uint32_t nodeCount, std::vector<uint32_t>& key, uint32_t& ncolours) {
  if (!colours(ops, nodeCount, nullptr, key, ncolours)) return false;
  if (const char* e = tuning_env("PIES_COLOUR_DSATUR"); e && std::atoi(e)) {
    std::vector<uint32_t> k2(ops.count);
    uint32_t n2 = 0;
    if (colours_dsatur(ops, nodeCount, k2, n2) && n2 <= ncolours) { key.swap(k2); ncolours = n2; }
  }
  int rounds = 12;
  if (const char* e = tuning_env("PIES_COLOUR_ROUNDS")) rounds = std::atoi(e);
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
  const char* noHint = tuning_env("PIES_NO_COLOUR_HINT");
  if (!(noHint && std::atoi(noHint))) {
    uint32_t n = 0;
    if (colours_from_hint(ops, nodeCount, trial, n) && n < ncolours) { key.swap(trial); ncolours = n; }
  }
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
