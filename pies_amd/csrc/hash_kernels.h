// Launch wrappers of the node-node collision kernels (hash_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "kernels.h"

namespace pies {

constexpr uint32_t kCounterTicket = 32, kCounterEpoch = 33, kHashCounters = 64;
struct HashArrays {
  uint32_t n;         // nodes (capacity of every per-pass list)
  uint32_t capacity;  // table slots, power of two
  uint32_t mask;
  uint64_t* keys;     // packed cell id, ~0 = empty
  uint32_t *cnt, *start, *fill;     // nodes overlapping the cell
  uint32_t *gcnt, *gstart, *gfill;  // nodes whose minimum cell it is (a "group")
  uint32_t* used;                   // slots in use
  uint32_t* counters;               // [0] used slots [1] bucket entries [2] grouped nodes [3] failure flag [4..30] groups per pass [31] resolved pairs [32] ticket [33] epoch
  uint32_t* done;                   // per table slot: epoch in which the cell's group was resolved (k_collide_flow)
  uint32_t* passList;               // 27 x n group slots
  uint32_t* nodeSlot;               // n x 8: table slot of each cell of the node's range
  int4* rng;                        // per node: min cell x,y,z and packed lengths
  uint32_t *bucket, *bucketSorted;  // 8n
  uint32_t *group, *groupSorted;    // n
};

// reset + count + alloc + fill + sort; returns the number of launches
uint32_t launch_hash_build(hipStream_t st, const HashArrays& H, const NodeArrays& nd, float gridSpacing);
// the 27 resolve passes (Solver.cpp:85-130); returns the number of launches
uint32_t launch_collide(hipStream_t st, const HashArrays& H, const NodeArrays& nd, float friction, float staticThreshold);

}  // namespace pies
