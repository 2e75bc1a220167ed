// Launch wrappers of the node-node collision kernels (hash_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "kernels.h"

namespace pies {

// counters[]: [0] cells in use [1] cell entries E [3] failure flags [4..30] groups per pass [31] resolved pairs
// [32] ticket (k_collide_flow) [33] epoch [34..36] / [37..39] bounding box of the cell ranges (min / max, as int)
// [44] progress of k_collide_reference (nodes visited, diagnostics) [46..47] candidates tested (64 bit, statistics)
constexpr uint32_t kCounterUsed = 0, kCounterEntries = 1, kCounterFlags = 3, kCounterPass0 = 4, kCounterPairs = 31, kCounterTicket = 32,
                   kCounterEpoch = 33, kCounterBoxMin = 34, kCounterBoxMax = 37, kCounterSortPasses = 40, kCounterDense = 41 /* a cell holds more than kMaxBucket nodes: the group order leaves the pass to the sequential loop */, kCounterProgress = 44, kCounterCandidates = 46 /* 64 bit: [46], [47] */,
                   kHashCounters = 64;
// failure flags: 1 non-finite position, 2 cell index overflow, (4: until round 4 more than kMaxBucket nodes in a cell - now kCounterDense), 8 k_collide_flow wait timed
// out, 128 more cell entries than reserved  (16, 32, 64 belong to the triangle grid's word, tri_kernels.h)
constexpr uint32_t kRadixTile = 4096;  // entries per workgroup of a radix pass (256 threads x 16)

struct HashArrays {
  uint32_t n;           // nodes
  uint32_t maxEntries;  // reserved (cell, node) entries
  uint32_t capacity;    // cell index slots, power of two
  uint32_t mask;
  // per node
  int4* rng;            // min cell x, y, z and packed lengths (8 bits each) at build time
  uint32_t* entCount;   // n + 1: cells the node overlaps (lx * ly * lz), then their exclusive prefix sum in entOff
  uint32_t* entOff;
  uint32_t* scanSums;   // tile sums of the prefix sum
  int* boxPart;         // per workgroup of k_grid_range: bounding box of its nodes' cell ranges (min x y z, max x y z)
  // (cell key, node) entries: node-major before the sort, ascending (key, node) after it; ping-pong buffers
  uint64_t* key[2];
  uint32_t* val[2];     // node index | 0x80000000 when the cell is the node's minimum cell
  uint32_t* hist;       // radix digit counts, [256][workgroups]
  // cell index: open addressing on the exact key -> bucket [start, end) in the sorted entries
  uint64_t* keys;
  uint32_t *start, *end;
  uint32_t* gcnt;       // nodes whose minimum cell it is (a "group")
  uint32_t* used;       // slots in use
  uint32_t* done;       // per slot: epoch in which the cell's group was resolved (k_collide_flow)
  uint32_t* passList;   // 27 x n group slots
  uint32_t* counters;
};

// zero + range + prefix sum + emit + radix sort + cell index; returns the number of launches
// sortPasses: radix passes captured for the entries' sort; a pass takes up to 11 key bits (grid_box, hash_device.h)
// groups: group sizes and pass lists of the group order (k_grid_groups; also its "too many nodes in a cell" latch) - the pair order
// over ranges of at most two cells looks at the buckets itself (k_pair_groups) and builds without it
uint32_t launch_hash_build(hipStream_t st, const HashArrays& H, const NodeArrays& nd, float gridSpacing, uint32_t sortPasses, bool groups = true);
constexpr uint32_t kMaxBucket = 2048;  // nodes overlapping one cell beyond which the parallel orders hand the pass to the sequential loop (the
                                       // reference's PBD loop has no limit and no latch, Solver.cpp:81-130; until round 4 this was a failure)
// the resolve of Solver.cpp:85-130 in the parallel visiting order; returns the number of launches.
// rearm: the launch first resets the work queue of k_collide_flow (a replay without a new hash build, profile passes)
// gridSpacing: for the sequential loop that takes a pass over a pile the group order cannot run (kCounterDense)
uint32_t launch_collide(hipStream_t st, const HashArrays& H, const NodeArrays& nd, float gridSpacing, float friction, float staticThreshold,
                        bool rearm = false);
// the same loop in the reference's order: ascending node index, range from the current position (one sequential chain)
// gate != nullptr: a device word; the kernel returns at once while it is 0 (the fallback of launch_collide_turns)
uint32_t launch_collide_reference(hipStream_t st, const HashArrays& H, const NodeArrays& nd, float gridSpacing, float friction,
                                  float staticThreshold, const uint32_t* gate = nullptr);

}  // namespace pies
