// Launch wrappers of the Projective-Dynamics kernels (pd_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "kernels.h"
#include "tri_kernels.h"

namespace pies {

constexpr uint32_t kCgRowBlocks = 256;  // extra blocks of k_cg_ap that sum the contact rows (graph variant useCAp)
constexpr uint32_t kCgInitBlocks = 4096;  // most workgroups of k_cg1_init (CgArrays::npartsI; partI holds that many partials)
constexpr uint32_t kCgBlocks = 1024;  // CG launch shape: <= 1024 blocks x 256 threads (4 per CU), grid-stride

// One contribution record: w * (A^T B p)_i, packed to 12 bytes (global_load/store_dwordx3).
struct Vec3f {
  float x, y, z;
};

struct CgArrays {
  uint32_t n;
  uint32_t nparts;  // blocks per CG launch (<= kCgBlocks)
  uint32_t* partCount;  // one-launch-per-iteration form: how many workgroups wrote part1[0], part1[1] and k_cg1_first's three sums in
                        // part1[0] - the launches of a solve are not all the same width (only the last one, whose continuation
                        // synchronises its workgroups with a grid barrier, is bound by what the device holds at once)
  uint32_t npartsI; // blocks of the launch that writes partI (the one-launch-per-iteration form's k_cg1_init: no grid barrier,
                    // so not bound by what the device holds at once; = nparts everywhere else)
  // K in sliced ELL form: a slice is one wavefront's rows, 64 / lanesPerRow of them; lane L = lanesPerRow * r + q of the slice
  // takes entries q, q + lanesPerRow, ... of its row r, entry step j of the slice sits at sliceOff[s] + 64 j + L, so a
  // wavefront's loads of col/val are contiguous.  Rows are padded to the slice's longest row with (col = the row itself, val = 0).
  uint32_t lanesPerRow;      // 1 (large systems) or 4
  const uint32_t* sliceOff;  // slices + 1 offsets (in entries)
  const uint32_t* col;
  const float* val;
  // Row dictionary (pd_setup.cpp; nullptr: none): rows with the same stencil - the same column offsets from the row's own index
  // and the same values, which is every interior row of a lattice and every class of boundary row - share one copy of it.  A row
  // is then one word (where its stencil starts and how long it is), and the SpMV reads (offset, value) pairs that stay in the vector cache instead of 8 bytes per
  // stored entry from HBM.  Same entries in the same order as the SELL arrays: the sums are bit for bit the same.
  const uint32_t* rowStencil;  // per row: first pair of its stencil | pairs << 24 (one word: no second look-up on the row's chain of loads)
  const int2* stencil;         // (column - row, bits of the value)
  // Windowed SELL (round 5; pd_setup.cpp build_window_matrix; wRows == 0: not built).  The rows are cut into chunks of wRows
  // consecutive rows; ONE WORKGROUP takes a chunk: it stages the vector entries of the chunk's window - its own rows first (slot
  // = row - first row), then the distinct columns outside the chunk in ascending order (the halo) - in LDS as 16-byte slots, and
  // the rows then gather from LDS through 16-bit slot indices.  What streams from HBM per stored entry is the value + a 16-bit
  // slot (6 bytes instead of 8), a column's vector entries are loaded once per chunk instead of once per row that names it (3-7
  // loads per row instead of 15), and the loads that remain are runs of consecutive columns.  Inside a chunk the rows may be
  // sorted by length (wPerm: local row of (slice, lane); nullptr: the natural order), so that a slice is padded to ITS longest
  // row only (an unstructured mesh: 1.33 -> 1.0x).  Entries keep their order inside a row (ascending column): a row's sum is
  // bit for bit the SELL arrays' and the dictionary's.
  uint32_t wRows, wChunks, wLdsSlots;  // rows per chunk (a multiple of 64), chunks, slots of the largest window
  const uint2* wChunk;       // per chunk: {first halo entry, halo entries}
  const uint32_t* wBase;     // per chunk: what its 16-bit halo columns are relative to
  const uint16_t* wHalo16;   // halo columns - base (nullptr: wHalo32 holds the columns themselves)
  const uint32_t* wHalo32;
  const uint32_t* wSliceOff; // slices + 1 offsets (in entries) of wVal / wIdx: slice s of chunk c is number c * wRows / 64 + s
  const float* wVal;
  const uint16_t* wIdx;
  const uint16_t* wPerm;     // per (chunk, slice, lane): the row's place in its chunk
  float* cdiag;  // diagonal of the collision matrix (floor contacts)
  float* dinv;   // 1 / diag(K + C)
  float4 *r, *z, *p[2], *ap;
  float *partI, *partA, *partB, *partBnext;
  float *partB0, *partB1;  // the ping-pong pair under fixed names (partB / partBnext are re-pointed per iteration)
  // point-triangle contacts of the substep (null when the pipeline is off): per node, ascending (contact<<2 | local)
  const uint32_t *tIncCnt, *tIncStart, *tInc;
  const uint4* tIds;
  // contact-heavy substeps (the host switches the graph variant, capi.cpp): the contact part of a row is summed by a
  // whole wavefront in a pass of its own (k_contact_rows) into cAp instead of by the row's lane inside the SpMV
  const uint32_t* tUsed;       // nodes that take part in contacts
  const uint32_t* tUsedCount;  // how many (device counter)
  float4* cAp;
  int useCAp;
  // merged contact rows (tri_kernels.hip k_contact_csr): distinct columns of a node's contact row, -w * multiplicity
  const uint32_t *rowStart, *rowLen, *rowCol;
  const float* rowCoef;
  float* scal;   // rz[2][3], bb[3], iterations, [10] the solve is over (converged, or went on inside its last launch), [11] where
                 // its final residual partials are
  uint32_t* ticket;  // [0] grid barrier counter of the solve's last k_cg_update, [1] its abort word (zeroed by k_cg_init)
  float* stats;  // over the tick: [0] max relative residual^2 of its solves, [1] max iterations, [2] solves, [3] solves that ended
                 // above the tolerance; since the buffers were built: [4] solves above the tolerance, [5] solves
  float tol2;    // squared relative tolerance of the solve whose statistics are being closed
  // ---- one launch per CG iteration (pd_cg1_kernels.hip; Chronopoulos-Gear's single-reduction form) ----
  // the preconditioned vectors t = D^-1 r, c = D^-1 s (s = (K + C) p by recurrence), a = D^-1 w (w = (K + C) t) as 12-byte
  // records in ping-pong pairs (an iteration's rows read their neighbours' old values while other workgroups write the new
  // ones), the search direction p, and per workgroup {r.t, w.t, r.r} of the three columns
  Vec3f *t1[2], *c1[2], *a1[2], *p1;
  float* part1[2];
  const float* kdiag;  // diagonal of K (the residual is r = (kdiag + cdiag) t)
  int single;          // the captured solves are of this form (solve_statistics: where a solve's final partials are)
};

// ShapeMatchingConstraint data (fp64 like the reference) and the fp64 contribution slots of shape and
// goal matching (the reference adds float w * double projection to the float force vector)
struct ShapeArrays {
  uint32_t count;          // shape constraints
  const uint32_t* off;     // count+1 entry offsets
  const uint32_t* node;    // per entry
  const double* mat;       // 3 per entry, centred material coordinates
  const double* qinv;      // 9 per constraint, row-major
  double* quat;            // 4 per constraint (w,x,y,z), warm-started across iterations and ticks
  const float* w;          // per constraint
};

struct RhsArrays {
  const float4* msn;
  const Vec3f* contrib;
  const uint32_t *incPtr, *incSlot;
  const double4* contribD;
  const uint32_t *incPtrD, *incSlotD;
  const float4* pos;
  const uint32_t* nstatic;
  float4* statp;
  const uint32_t *tIncCnt, *tIncStart, *tInc;
  const float4* tContrib;
  const uint32_t* tUsedCount;  // nodes in contacts this substep (device word; 0: nobody reads tIncCnt)
  uint32_t n;
};

// Tile-resident local step (pd_tiles.cpp plans, k_pd_local_tiles runs): the element pairs are cut into tiles of up to 128
// spatially close elements that touch at most 128 nodes; ONE WAVEFRONT takes a tile: it holds the positions of the tile's nodes
// in LDS, projects the tile's elements two per lane, parks every element's four contributions in LDS and then adds them up
// node by node - lane k owns tile nodes k and k + 64 and walks their incidence lists in a fixed order (deterministic, no
// atomics, no barrier between wavefronts).  What leaves the chip is ONE 12-byte sum per (tile, node) instead of one record
// per (element, node): 3-4 records per node for the right-hand side to add up instead of 24 (Solver.cpp:270-349).
// Everything of a tile sits at a fixed stride (no descriptor to chase): tile t owns node[128 t ..], local[128 t ..],
// inc[512 t ..], nptr[132 t ..] and partial[128 t ..].
constexpr uint32_t kTileNodes = 128;  // nodes a tile may touch
constexpr uint32_t kTileElems = 128;  // element pairs of a tile (two per lane)
constexpr uint32_t kTileNptr = 132;   // stride of the per-node list offsets (129 used)
struct PdTileArrays {
  uint32_t ntiles;  // 0: the local step writes per-(element, node) records (k_pd_local_tet_pair)
  const uint32_t* info;   // per tile: nodes | elements << 16
  const uint32_t* node;   // global node index per tile node
  const uint32_t* local;  // per element pair: four 8-bit tile-local node indices
  const uint16_t* dict;   // per element pair: index into the rest dictionary, or nullptr: the four arrays below
  const float4 *q0, *q1, *q2, *vq2;
  const uint16_t* nptr;   // per tile node: first entry of its list in `inc` (nodes + 1 offsets)
  const uint16_t* inc;    // per (element, corner) of the tile, grouped by node, ascending element: element << 2 | corner
  Vec3f* partial;         // per tile node: the sum over the tile's elements (records of the right-hand side's gather)
};

struct PdArrays {
  ShapeArrays shape;
  double4* contribD;        // shape entries, then goal entries: (x, y, z, w)
  const uint32_t* incPtrD;  // per node: fp64 slots in reference order (shape constraints, then goal)
  const uint32_t* incSlotD;
  float4* msn;
  float4* rhs;
  float4* statp;
  Vec3f* contrib;
  const uint32_t* incPtr;
  const uint32_t* incSlot;
  const float4* tContrib;  // 4 per contact: w * (AtA p)_i
  TriArrays tri;
  const uint32_t* triCount;
  uint32_t* nstatic;
  const float* kdiag;
  PdTileArrays tiles;
  uint32_t rhsLanes;  // lanes of k_pd_rhs that share a node: 4 (a node gathers its ~24 per-constraint records) or 1 (tile sums)
  CgArrays cg;
};

void launch_pd_predict(hipStream_t st, const NodeArrays& nd, const PdArrays& pd, float h, float contactHeight, bool triReset);
void launch_pd_local_distance(hipStream_t st, const float4* pos, const uint2* ids, const float2* rw, Vec3f* contrib, uint32_t count);
// the node-pair extension container (CollisionConstraint.cpp:7-65; Solver.cpp:398-428)
constexpr float kNodePairW = 100000.0f;  // Include/Pies/CollisionConstraint.h:14
void launch_pd_local_node_pair(hipStream_t st, const float4* pos, const float* radius, const uint2* ids, Vec3f* contrib, uint32_t count);
void launch_pd_node_pair_friction(hipStream_t st, const float4* pos, float4* vel, const float* radius, const uint2* ids, uint32_t count,
                                  float friction, float staticThreshold);
void launch_pd_local_tet(hipStream_t st, bool volume, const float4* pos, const uint4* ids, const float4* q0, const float4* q1,
                         const float4* q2, Vec3f* contrib, uint32_t count);
// strain + volume constraints over identical elements (same ids, same Qinv), fused
// tri != nullptr: the launch also runs the local step of the point-triangle contacts (launch_pd_local_tri's work) in a few
// extra workgroups
void launch_pd_local_tet_pair(hipStream_t st, const float4* pos, const uint4* ids, const float4* q0, const float4* q1, const float4* q2,
                              const float4* vq2, Vec3f* contribTet, Vec3f* contribVol, uint32_t count, const TriArrays* tri = nullptr,
                              float thickness = 0.0f, bool packed = true, const uint16_t* dictIndex = nullptr,
                              const float4* dictTable = nullptr);  // packed: two elements per lane; dict*: pd_local_packed.h RestDictionary
// the strain + volume local step over tiles (see PdTileArrays); dictTable: the rest dictionary's table when T.dict is set;
// tri as in launch_pd_local_tet_pair
void launch_pd_local_tiles(hipStream_t st, const float4* pos, const PdTileArrays& T, const float4* dictTable, const TriArrays* tri = nullptr,
                           float thickness = 0.0f);
void launch_pd_local_bend(hipStream_t st, const float4* pos, const uint4* ids, const float2* angle_w, Vec3f* contrib, uint32_t count);
void launch_pd_local_shape(hipStream_t st, const float4* pos, const PdArrays& pd);
void launch_pd_rhs(hipStream_t st, const NodeArrays& nd, const PdArrays& pd);
RhsArrays rhs_arrays(const NodeArrays& nd, const PdArrays& pd);
// part: -1 = the solve; profile passes: 1 = the SpMV (+ direction update) kernels only, 0 = the vector-update kernels only
// first / last: the first and the last solve of a substep (a solve's statistics are closed by the next solve's first
// kernel, the last one's by a launch of its own)
// neverExit: the iterations do not take the converged early exit (in-situ timing); hook(ctx, class) is called before and
// after every k_cg_ap (PIES_KERNEL_PD_SPMV) and k_cg_update (PIES_KERNEL_PD_CG_UPDATE) launch; overflowIters: iterations a
// solve that is still above the tolerance after its maxIters captured ones may go on for inside the last launch (see
// k_cg_update)
void launch_pd_solve(hipStream_t st, const NodeArrays& nd, const PdArrays& pd, int maxIters, float tol, int part = -1, bool first = true,
                     bool last = true, bool neverExit = false, void (*hook)(void*, int) = nullptr, void* hookCtx = nullptr,
                     int overflowIters = 0);
// The same solve with ONE launch per CG iteration (pd_cg1_kernels.hip): k_cg1_init (residual; with fuseRhs it evaluates the
// right-hand side itself and launch_pd_rhs is not needed), k_cg1_first (w = (K + C) D^-1 r), then `iters` launches of
// k_cg1_iter, each a whole iteration - scalars from the previous launch's partial sums, the neighbours' new preconditioned
// residual recomputed while it is gathered, x / r / p / s updated for the launch's own rows, the next dot products.  Contact
// rows are summed inline (the contact-light graph variant); one lane per row.  hook brackets the k_cg1_iter launches
// (PIES_KERNEL_PD_SPMV).
void launch_pd_solve1(hipStream_t st, const NodeArrays& nd, const PdArrays& pd, int iters, float tol, bool first, bool last, bool fuseRhs,
                      bool neverExit = false, void (*hook)(void*, int) = nullptr, void* hookCtx = nullptr, int overflowIters = 0);
void launch_cg_finish(hipStream_t st, const CgArrays& A);  // the statistics of a substep's last solve (A.partB: its final partials)
// closeSolve: one more workgroup closes the statistics of the substep's last solve (k_cg_finish's work; maxIters = that solve's
// captured iterations, tol its tolerance)
void launch_pd_stabilize(hipStream_t st, const NodeArrays& nd, const PdArrays& pd, bool closeSolve = false, int maxIters = 0, float tol = 0.0f,
                         bool single = false);  // single: that solve was launch_pd_solve1's
// staticFriction = false leaves the floor friction (Solver.cpp:473-484) to the contacts' friction pass (launch_tri_friction), which the
// reference runs after the point-triangle friction
void launch_pd_velocity(hipStream_t st, const NodeArrays& nd, const PdArrays& pd, float h, float damping, float gravity,
                        float friction, float staticThreshold, bool staticFriction, const uint32_t* usedBits = nullptr);
// usedBits (the point-triangle pipeline's bitmap of nodes in contacts): floor friction for the nodes outside it only

// workgroups of k_cg_update the device holds at once (0: unknown); the CG kernels' grid stays below it, see grid_barrier
inline uint32_t window_lds_bytes(const CgArrays& A) { return (A.wLdsSlots + A.wRows) * 16u; }  // the largest window + a chunk's rows of 16 bytes (window_rows)
uint32_t cg_update_resident_blocks(int device);
uint32_t cg1_iter_resident_blocks(int device, uint32_t windowLdsBytes = 0);  // windowLdsBytes != 0: the windowed kernel with that much LDS

}  // namespace pies
