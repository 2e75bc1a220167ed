// Point-triangle collision pipeline of the PD substep (reference: Src/Solver.cpp:680-875 detection,
// Src/CollisionDetection.cpp:227-302 CCD, Src/CollisionConstraint.cpp:67-194 constraint) -- launch wrappers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "kernels.h"

namespace pies {

// A triangle's box over position and previous position spans up to 50 cells per axis when it is inserted (TriCompRange,
// Solver.cpp:974-976) and up to 20 when it searches (sweptTriRange, :672-674); a longer range is EMPTY, as in the
// reference.  Storage is reserved for kTriMaxEntries (cell, triangle) entries per triangle on average (a triangle of a
// simulation mesh spans 1-8 world-unit cells); the first kTriMaxEntries cells of a triangle have their index slot cached
// between the count and the fill pass, the rest are looked up again.  More entries in total than reserved latch a failure.
constexpr uint32_t kTriInsertMaxCells = 50, kTriSearchMaxCells = 20;
constexpr uint32_t kTriMaxEntries = 64;
constexpr float kTriContactW = 10000.0f;                        // PointTriangleCollisionConstraint::w (CollisionConstraint.h:32)

struct TriArrays {
  uint32_t nt;           // surface triangles (0 = pipeline disabled)
  uint32_t threadCount;  // SolverOptions::threadCount: contacts are listed thread by thread (Solver.cpp:714,852)
  const uint32_t* tris;      // 3 node ids per triangle
  // triangle grid (exact-key cell table, world-unit cells)
  uint32_t capacity, mask;
  uint64_t* keys;
  uint32_t *cnt, *start, *fill, *used;
  uint32_t* counters;  // [0] used cells [1] bucket entries [2] contacts [3] failure flag [4] nodes with contacts [5] incidences
                       // [8] merged row entries [6] dependency levels of the contact list [7] form of the sequential passes: 0 on an LDS copy of the
                       // touched nodes, 2 through L2, 1 more than kTriMaxLevels levels (single-wavefront walk)
  uint32_t* triSlot;   // nt x kTriMaxEntries
  int4* rng;           // per triangle: min cell, packed lengths
  float4* box;         // 3 per triangle (k_tri_count): {box min of its six corner positions, regular ? 1 : 0}, {box max, bits(node 0)},
                       // {bits(node 1), bits(node 2), -, -}: what the detection's reject test needs of a candidate, in one place
  uint32_t *bucket, *bucketSorted;
  uint32_t maxEntries;  // (cell, triangle) entries reserved in bucket / bucketSorted
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
constexpr uint32_t kTriMaxLevels = 2048;  // longer chains (one node in thousands of contacts) take the single-wavefront path

struct PdArrays;

// after the predict kernel: grid build, detection (count, scan, fill), per-node incidence + diagonal; returns launches
uint32_t launch_tri_detect(hipStream_t st, const TriArrays& T, const NodeArrays& nd, const float* kdiag, float* cdiag, float* dinv,
                           float threshold, float thickness, bool mergedRows);
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
