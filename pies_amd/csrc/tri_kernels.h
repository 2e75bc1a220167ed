// Point-triangle collision pipeline of the PD substep (reference: Src/Solver.cpp:680-875 detection,
// Src/CollisionDetection.cpp:227-302 CCD, Src/CollisionConstraint.cpp:67-194 constraint) -- launch wrappers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "kernels.h"

namespace pies {

// A triangle's box over position and previous position spans up to 50 cells per axis when it is inserted (TriCompRange,
// Solver.cpp:974-976) and up to 20 when it searches (sweptTriRange, :672-674); a longer range is EMPTY, as in the
// reference.
//
// Round 4, second half: the reference's grid lists a triangle in EVERY cell of its range (18-27 cells for a moving triangle of a
// unit lattice: a million (cell, triangle) entries and 2.2 M atomic operations per substep at 42k triangles, and every pair of
// neighbours met in ~10 shared cells).  The device keeps each triangle ONCE, in the cell of its range's minimum corner, and a
// searching triangle looks at every cell a partner's minimum corner can lie in (its own range, grown downwards by the longest
// range inserted - 1): the same pairs, each met once, and the number of cells they share - which the reference's list repeats
// a hit by - is the volume of the two ranges' intersection.  Three size classes keep the window small when a few triangles are
// much longer than the rest: class k has cells of 1, 4 and 16 world cells (TriGridLevel::shift) and takes the triangles whose range spans
// at most kTriLevelExt of them per axis (the last class takes every insertable triangle: 50 world cells are 5 cells of 16).  Cells are
// slots of a table indexed by the cell coordinates modulo the table's (power-of-two) dimensions: two cells far apart may
// share a slot, which only adds candidates that the exact range test drops.
constexpr uint32_t kTriInsertMaxCells = 50, kTriSearchMaxCells = 20;
constexpr int kTriLevels = 3;
constexpr uint32_t kTriLevelExt = 4;
struct TriGridLevel {
  uint32_t base;       // first slot of the level in TriArrays::cell
  uint32_t shift;      // world cells per cell, log2
  uint32_t lx, ly, lz; // table dimensions, log2
};
constexpr float kTriContactW = 10000.0f;                        // PointTriangleCollisionConstraint::w (CollisionConstraint.h:32)

struct TriArrays {
  uint32_t nt;           // surface triangles (0 = pipeline disabled)
  uint32_t threadCount;  // SolverOptions::threadCount: contacts are listed thread by thread (Solver.cpp:714,852)
  const uint32_t* tris;      // 3 node ids per triangle
  // triangle grid (minimum-corner cells in three size classes, see above)
  TriGridLevel level[kTriLevels];
  uint32_t slots;      // slots of the three tables together (a multiple of 2048)
  uint32_t* cellCnt;   // per slot: triangles listed
  uint32_t* cellStart; // slots + 1: first entry of the slot (entries are in slot order: a row of cells along z is one stretch)
  uint32_t* tileSum;   // per 2048 slots: triangles listed
  uint32_t *cellOf, *posIn;  // per triangle: its slot (0xffffffff: not inserted) and its place among the slot's triangles
  float4* ent;         // 4 per listed triangle, slot by slot: {box min of its six corner positions, bits(triangle)}, {box max, regular ? 1 : 0},
                       // {bits: min cell, packed lengths}, {bits: node ids} - all a searching triangle needs of a partner, in one line
  float4* boxOf;       // the same four records per triangle (k_tri_box), copied into ent once the slots have their storage
  uint32_t* counters;  // [0] [1] - [2] contacts [3] failure flag [4] nodes with contacts [5] incidences
                       // [8] merged row entries [6] dependency levels of the contact list [7] form of the sequential passes: 0 on an LDS copy of the
                       // touched nodes, 2 through L2, 1 more than kTriMaxLevels levels (single-wavefront walk)
                       // [9] hit records [10..12] longest range (in cells of the class) listed per class [13..15] triangles per class
  int4* rng;           // per triangle: min cell, packed lengths
  uint2* work;         // pairs left for the CCD: {triangle, partner | corners to test << 29}, in 64 lists of maxWork / 64
  uint32_t maxWork;
  uint32_t* workCnt;   // the lists' lengths, 16 words apart
  uint32_t* head;      // per triangle: its first hit record (0xffffffff: none)
  uint4* pool;         // hit records {triangle, partner, which of the triangle's corners hit (bits 0-2), next record of the triangle}; maxContacts
  // contacts of the current substep, in the reference's list order
  uint32_t maxContacts;
  uint32_t *cntTri, *offTri;  // contacts per triangle and their list offsets, indexed by merge_rank(triangle)
  uint4* ids;        // a, b, c, d
  float4* contrib;   // 4 per contact: w * (AtA p)_i
  // per node: the contacts it takes part in, ascending (contact << 2 | local index)
  uint32_t *incCnt, *incStart, *incFill, *usedNodes, *inc, *incSorted;
  uint32_t* incPos;    // per (contact << 2 | local): its position in the node's sorted list
  // dependency levels of the contact list (k_tri_levels): the sequential passes run level by level
  int* lastLevel;      // per node: level of the last contact seen that touches it (-1 between substeps)
  uint32_t* lvl;       // per contact
  uint32_t* lvOrder;   // contacts bucketed by level
  uint32_t* lvStart;   // kTriMaxLevels + 1 offsets into lvOrder
  // merged contact rows (k_contact_csr; contact-heavy graph variant): per node with contacts the distinct columns of its
  // row of the contact matrix, ascending, and -w * (how often the column occurs); rowLen = 0xffffffff: not merged
  uint32_t *rowStart, *rowLen, *rowCol;
  float* rowCoef;
  uint32_t* usedBits;  // bitmap over the nodes: takes part in a contact of this substep (usedNodes is read off it, ascending)
  uint32_t* nodeSlot;  // per node with contacts: its index in usedNodes (= its record in the LDS copy of the sequential passes)
  uint2* lvSlots;      // per entry of lvOrder: the four nodeSlots of the contact, 16 bit each
};
constexpr uint32_t kTriNil = 0xffffffffu;
constexpr uint32_t kWorkShards = 64;   // lists of the CCD's work list (k_tri_pairs)
constexpr uint32_t kGridTile = 2048;   // slots per tile of the prefix sum over the slots (256 threads x 8)
// Before the grid is built (rides in the substep's first launch, k_pd_predict: one launch less): the slots the last substep listed triangles in, the tile sums and the substep's counters back to zero
// ([0] and [4]-[8] are zeroed by k_tri_box, which runs before their first use: this kernel's own workgroups still read [4]),
// the per-node incidence counts of the last substep's contacts.
__device__ __forceinline__ void tri_reset(const TriArrays& T, uint32_t tid, uint32_t stride) {
  const uint32_t usedNodes = T.counters[4];
  if (tid < 16u && (tid == 1u || tid == 2u || tid >= 9u)) T.counters[tid] = 0;  // ([3] is the sticky failure flag)
  for (uint32_t b = tid; b < T.slots / kGridTile; b += stride) T.tileSum[b] = 0;
  if (tid < kWorkShards) T.workCnt[16u * tid] = 0;
  for (uint32_t t = tid; t < T.nt; t += stride) {
    const uint32_t s = T.cellOf[t];  // (the slots of the last substep; several triangles of a slot store the same 0)
    if (s != kTriNil) T.cellCnt[s] = 0;
  }
  for (uint32_t u = tid; u < usedNodes; u += stride) {
    const uint32_t n = T.usedNodes[u];
    T.incCnt[n] = 0;
    T.usedBits[n >> 5] = 0;  // (several nodes of a word: every writer stores the same 0)
  }
}

constexpr uint32_t kTriMaxLevels = 2048;  // longer chains (one node in thousands of contacts) take the single-wavefront path

struct PdArrays;

// after the predict kernel: grid build, detection (pairs + hit records, scan, contact list), per-node incidence + diagonal; returns launches
// (levelsInLine, contact-light variant only: the dependency levels of the list are computed by the same launch as the list -
// launch_tri_levels must not be called for that substep)
uint32_t launch_tri_detect(hipStream_t st, const TriArrays& T, const NodeArrays& nd, const float* kdiag, float* cdiag, float* dinv,
                           float threshold, float thickness, bool mergedRows, bool levelsInLine);
// dependency levels of the contact list for the sequential passes (may run on another stream beside the local/global iterations)
void launch_tri_levels(hipStream_t st, const TriArrays& T);
void launch_pd_local_tri(hipStream_t st, const TriArrays& T, const float4* pos, float thickness);
// every stabilisation iteration of the substep in one launch: the passes over the contact list with the floor snap of the list's
// nodes (nstatic / statp: launch_pd_rhs's floor targets) between them; the snap of all other nodes is launch_pd_stabilize's, once, behind it
void launch_tri_stabilize(hipStream_t st, const TriArrays& T, const NodeArrays& nd, float thickness, const uint32_t* nstatic, const float4* statp,
                          uint32_t iterations);
// nstatic != nullptr: the pass ends with the floor friction (Solver.cpp:473-484) of the nodes of the contact list; the other nodes
// got theirs in launch_pd_velocity (usedBits)
void launch_tri_friction(hipStream_t st, const TriArrays& T, const NodeArrays& nd, float friction, float staticThreshold, const uint32_t* nstatic = nullptr);

}  // namespace pies
