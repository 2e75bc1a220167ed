// Schedule EXACT as one dependency DAG over the whole substep.
//
// tickPBD (Src/Solver.cpp:58-136) is a fixed sequence of operations on node positions: per iteration the position,
// distance, tetrahedral and bend containers in container order, [the node-node collision pass,] the floor clamp of
// every node.  Two operations commute unless one writes a node the other touches, so any execution that keeps the
// relative order of conflicting operations gives the sequential result bit for bit.  An operation's level is one more
// than the highest level of the operations it must follow; a level is one launch (k_wave).  Unlike per-sweep levels
// (schedule.cpp `levels`, one barrier per container and iteration) a sweep's tail overlaps the next sweeps' heads:
// 2 879 launches instead of 49 322 for 20 iterations of the 100k-particle beam, the same result.
// The collision pass reads and writes arbitrary nodes, so it cuts the DAG into segments.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "solver_state.h"

namespace pies {

namespace {
struct Stream {  // one container, in the order the reference visits it
  const uint32_t* ids;   // count * stride node ids, HOST order
  const uint32_t* slot;  // host index -> slot of the device arrays (inverse of Plan::order)
  uint32_t stride, count;
  uint8_t writeMask;
};
}  // namespace

// Returns false (and leaves `out` inactive) when the DAG would be too large to be worth holding.
bool build_wave_plan(const pies_solver* s, WavePlan& out) {
  out = WavePlan{};
  const uint32_t n = s->nodeCount();
  const uint32_t iters = s->opt.iterations;
  const uint64_t perIter = (s->releaseHinge ? 0ull : s->h_position.size()) + s->h_distance.size() + s->h_tet.size() + s->h_bend.size() + n;
  const uint64_t total = perIter * iters;
  if (n == 0 || iters == 0 || total == 0 || total > kWaveMaxOps) return false;

  // host-order ids and host index -> slot maps of the four containers
  std::vector<uint32_t> ids[4], slot[4];
  auto invert = [&](int kind, int type, size_t count) {  // kind: 0..3 in visiting order; type: the container's plan
    slot[kind].resize(count);
    const std::vector<uint32_t>& order = s->plan[type].order;
    for (uint32_t k = 0; k < count; ++k) slot[kind][order[k]] = k;
  };
  ids[0].resize(s->h_position.size());
  for (size_t i = 0; i < ids[0].size(); ++i) ids[0][i] = s->h_position[i].id;
  ids[1].resize(2 * s->h_distance.size());
  for (size_t i = 0; i < s->h_distance.size(); ++i) { ids[1][2 * i] = s->h_distance[i].ids[0]; ids[1][2 * i + 1] = s->h_distance[i].ids[1]; }
  ids[2].resize(4 * s->h_tet.size());
  for (size_t i = 0; i < s->h_tet.size(); ++i) std::memcpy(&ids[2][4 * i], s->h_tet[i].ids, 16);
  ids[3].resize(4 * s->h_bend.size());
  for (size_t i = 0; i < s->h_bend.size(); ++i) std::memcpy(&ids[3][4 * i], s->h_bend[i].ids, 16);
  invert(0, PIES_POSITION, s->h_position.size());
  invert(1, PIES_DISTANCE, s->h_distance.size());
  invert(2, PIES_TET, s->h_tet.size());
  invert(3, PIES_BEND, s->h_bend.size());
  const Stream streams[4] = {
      {ids[0].data(), slot[0].data(), 1, s->releaseHinge ? 0u : (uint32_t)s->h_position.size(), 0x1},  // Solver.cpp:59-63
      {ids[1].data(), slot[1].data(), 2, (uint32_t)s->h_distance.size(), 0x1},  // a distance projection moves node a only
      {ids[2].data(), slot[2].data(), 4, (uint32_t)s->h_tet.size(), 0xF},
      {ids[3].data(), slot[3].data(), 4, (uint32_t)s->h_bend.size(), 0xF}};

  // ---- pass 1: the level of every operation, in sequential order --------------------------------------------
  std::vector<uint32_t> level(total);
  std::vector<uint32_t> lastW(n, 0), lastR(n, 0);  // highest level that wrote / read the node (0 = none; levels are 1-based)
  uint32_t base = 0, maxLevel = 0;                 // operations after a barrier start above `base`
  uint64_t at = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    for (const Stream& st : streams)
      for (uint32_t c = 0; c < st.count; ++c) {
        const uint32_t* id = st.ids + static_cast<size_t>(c) * st.stride;
        uint32_t lv = base;
        for (uint32_t k = 0; k < st.stride; ++k) {
          lv = std::max(lv, lastW[id[k]]);
          if (st.writeMask & (1u << k)) lv = std::max(lv, lastR[id[k]]);
        }
        ++lv;
        for (uint32_t k = 0; k < st.stride; ++k) {
          if (st.writeMask & (1u << k)) lastW[id[k]] = lv;
          else lastR[id[k]] = std::max(lastR[id[k]], lv);
        }
        level[at++] = lv;
        maxLevel = std::max(maxLevel, lv);
      }
    if (s->nodeCollisions) {  // Solver.cpp:81-130 sits between the constraints and the floor clamp
      out.barrierAfter.push_back(maxLevel);
      base = maxLevel;
    }
    for (uint32_t i = 0; i < n; ++i) {  // Solver.cpp:132-136
      const uint32_t lv = std::max(base, std::max(lastW[i], lastR[i])) + 1;
      lastW[i] = lv;
      level[at++] = lv;
      maxLevel = std::max(maxLevel, lv);
    }
  }

  // ---- pass 2: counting sort by (level, kind) -----------------------------------------------------------------
  out.levels.assign(maxLevel, WaveLevel{});
  at = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    for (int kind = 0; kind < 4; ++kind)
      for (uint32_t c = 0; c < streams[kind].count; ++c) ++out.levels[level[at++] - 1].cnt[kind];
    for (uint32_t i = 0; i < n; ++i) ++out.levels[level[at++] - 1].cnt[4];
  }
  uint64_t run = 0;
  for (WaveLevel& L : out.levels)
    for (int kind = 0; kind < kWaveKinds; ++kind) {
      L.off[kind] = static_cast<uint32_t>(run);
      run += L.cnt[kind];
    }
  out.index.resize(total);
  std::vector<uint32_t> fill(static_cast<size_t>(maxLevel) * kWaveKinds, 0);
  at = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    for (int kind = 0; kind < 4; ++kind)
      for (uint32_t c = 0; c < streams[kind].count; ++c) {
        const uint32_t lv = level[at++] - 1;
        out.index[out.levels[lv].off[kind] + fill[static_cast<size_t>(lv) * kWaveKinds + kind]++] = streams[kind].slot[c];
      }
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t lv = level[at++] - 1;
      out.index[out.levels[lv].off[4] + fill[static_cast<size_t>(lv) * kWaveKinds + 4]++] = i;
    }
  }
  if (const char* e = std::getenv("PIES_LAYER_DEBUG"); e && e[0] == '1') {  // how large the levels are (development aid)
    uint32_t small[5] = {0, 0, 0, 0, 0};  // levels with at most 64, 256, 1024, 4096 operations; larger
    for (const WaveLevel& L : out.levels) {
      uint64_t ops = 0;
      for (int kind = 0; kind < kWaveKinds; ++kind) ops += L.cnt[kind];
      ++small[ops <= 64 ? 0 : ops <= 256 ? 1 : ops <= 1024 ? 2 : ops <= 4096 ? 3 : 4];
    }
    std::fprintf(stderr, "[pies] schedule EXACT: %zu levels; operations per level <= 64: %u, <= 256: %u, <= 1024: %u, <= 4096: %u, more: %u\n",
                 out.levels.size(), small[0], small[1], small[2], small[3], small[4]);
  }
  out.active = true;
  return true;
}

}  // namespace pies
