// Launch wrappers of the pair-ordered node-node resolve (pair_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "hash_kernels.h"
#include "kernels.h"

namespace pies {

// control words of a pass
constexpr uint32_t kPairLeft = 3;       // nodes that left their slack in this pass (entries of `left`)
constexpr uint32_t kPairFlags = 4;      // 1: an unlisted pair may have touched (k_pair_verify)  2: list storage overflow
constexpr uint32_t kPairRetry = 5;      // the pass is being repeated
constexpr uint32_t kPairRounds = 6;     // rounds the last pass needed (diagnostics)
constexpr uint32_t kPairRetries = 7;    // lifetime: passes repeated
constexpr uint32_t kPairInexact = 8;    // lifetime: passes in which a node left its slack in the repeat as well (the result may differ
                                        // from the documented order by visits that were filtered out)
constexpr uint32_t kPairEdges = 9;      // pairs listed by the last pass (diagnostics)
constexpr uint32_t kPairDeepest = 12;    // most levels any pass needed since the host last looked (it sizes the captured launches from it)
constexpr uint32_t kPairGroups = 10;     // groups of this grid (entries of `grp`)
constexpr uint32_t kPairSpilled = 11;    // groups the small list kernel passed on to the large one (entries of `spill`)
constexpr uint32_t kPairFallback = 13;   // the sequential loop runs this pass: by turns it could not be proved exact twice, or (either order) a pile
                                         // the lists do not hold (more than 1 024 partners of a node, more entries than reserved)
constexpr uint32_t kPairFallbacks = 14;  // lifetime: passes the sequential loop had to run
constexpr uint32_t kPairBarrier = 16;   // [16] counter and [17] abort word of the grid barrier of a repeated pass (k_pair_repeat)
constexpr uint32_t kPairWords = 32;

constexpr uint32_t kPairLists = 64;     // the frontier is kept as this many sub-lists: a wavefront appends to one of them, so that the
                                        // appends of a level are spread over 64 counters (same-address atomics take ~8 ns each)
constexpr uint32_t kPairStripes = 1024; // resolved-pair statistics are counted in stripes for the same reason
constexpr uint32_t kPairPad = 32;       // counters that many wavefronts add to sit 128 bytes apart: atomics on one cache line take their turns
                                        // (measured: 500 000 adds on 64 adjacent words 2.4 ms, on 64 lines 60 us)
constexpr uint32_t kPairPools = 64;     // list storage is handed out from this many pools (a wavefront of the list kernel uses one)

struct PairArrays {
  uint32_t n;
  uint32_t poolCap;        // list entries per pool
  float4 *node;            // 4 per node, one cache line: [0] x, y, z, invMass  [1] vx, vy, vz, radius  [2] position when the grid was
                           // built (the filter's distances, the slack test, the repeat), slack (how far the node may stray from there before
                           // the filtered lists stop being exact; persistent)  [3] first list entry, entries, current entry | round in
                           // which the node got there << 16, the current entry itself
  float4 *vel0;            // velocities when the pass started (for the repeat)
  uint32_t *exc;           // per node: how far it has strayed in this pass (float bits)
  uint32_t byIndex;        // the lists are for the reference order by turns: a node's entries ascending by the partner's INDEX
  uint32_t *turnCnt;       // reference order by turns: per node, the members of its turn (partners + itself) that are not at it yet
  uint32_t *nbr;           // list entries: other node | (shared cells - 1) << 28, a node's entries ascending by pair key
  uint32_t *nbrM;          // only for scenes with ranges wider than two cells per axis: shared cells of an entry (nbr then holds the node alone)
  float4 *bq;              // per node: position when the grid was built, radius + slack: all the list kernel gathers of a candidate
  uint32_t frCap;          // entries of one sub-list of the frontier
  uint32_t *fr[2];         // frontier: the nodes that moved on to a new entry in the last round, kPairLists sub-lists of frCap entries
  uint32_t *frCount;       // [3][kPairLists] entries of the sub-lists of round r, at r % 3
  uint32_t *hitStripe;     // resolved pairs of this pass, striped
  uint32_t *stat;          // [0..127] candidates looked at (64 x 64 bit), [128..191] listed entries: striped counters of the list kernel
  uint4 *grp;              // 4 per group: the buckets of the 2x2x2 cells above the group's cell (k_pair_groups)
  uint32_t *spill;         // groups for the large list kernel
  uint32_t *left;          // the nodes that left their slack in this pass
  uint32_t *pool;          // entries handed out per pool
  uint32_t* ctl;
};

// One pass of Solver.cpp:85-130 in the pair order: save, lists, start, `rounds` level launches, the tail
// that finishes whatever levels are left, and the (normally skipped) repeat with the widest slack.  Returns the launches.
uint32_t launch_collide_pairs(hipStream_t st, const HashArrays& H, const PairArrays& P, const NodeArrays& nd, float gridSpacing, float friction,
                              float staticThreshold, uint32_t rounds);

// One pass of Solver.cpp:85-130 in the REFERENCE's order - ascending node index, the node's range from its live position, buckets in
// dx, dy, dz order, every overlapping visit resolved at once - executed by dependency levels of whole TURNS (a node's turn = all of
// its visits): the turns whose members (the node and the partners within reach) have all had their earlier turns share no node,
// so a level is one launch.  Same result as k_collide_reference's single chain, bit for bit (see pair_kernels.hip).
uint32_t launch_collide_turns(hipStream_t st, const HashArrays& H, const PairArrays& P, const NodeArrays& nd, float gridSpacing, float friction,
                              float staticThreshold, uint32_t rounds);

}  // namespace pies
