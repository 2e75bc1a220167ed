// Internal state of one pies_solver handle: host mirror of the scene (authoritative until the first
// tick / after a read-back), schedules, HBM buffers, the captured substep graph.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pies_hip.h"
#include "kernels.h"
#include "pd_kernels.h"
#include "hash_kernels.h"
#include "pair_kernels.h"

namespace pies {

struct HostPosition {
  uint32_t id;
  float target[3];
  float w;
};
constexpr uint16_t kNoColourHint = 0xFFFFu;
struct HostDistance {
  uint32_t ids[2];
  float target;
  float w;
  uint16_t hint;  // colour proposed by a lattice factory (schedule.cpp verifies it), kNoColourHint otherwise
};
struct HostTet {  // TetrahedralConstraint and VolumeConstraint share the rest data
  uint32_t ids[4];
  float qinv[9];  // column-major (glm layout): qinv[3*col + row]
  float lo, hi;   // minStrain/maxStrain or minOmega/maxOmega
  float w;
  float AtA[16];  // row-major 4x4, A^T A (B = I so AtB = A^T)
  float A[16];    // row-major 4x4
  uint16_t hint;  // see HostDistance
  // colours proposed for schedule LAYERED when the breadth-first levels are lattice layers perpendicular to x, y or z
  // (layer_plan.cpp verifies a proposal group by group)
  uint16_t layerHint[3] = {kNoColourHint, kNoColourHint, kNoColourHint};
};
struct HostBend {
  uint32_t ids[4];
  float angle;
  float w;
};
struct HostNodePair {  // CollisionConstraint (Src/CollisionConstraint.cpp:7-65): an extension container, pies_add_node_pair_constraints
  uint32_t ids[2];
};

struct HostShape {  // ShapeMatchingConstraint (Src/ShapeMatchingConstraint.cpp:6-48)
  std::vector<uint32_t> ids;
  std::vector<double> mat;  // 3 x n material coordinates, centred (column i at mat[3i..])
  double Qinv[9];           // row-major
  float w;
};
struct HostGoal {  // GoalMatchingConstraint (Src/ShapeMatchingConstraint.cpp:124-137)
  std::vector<uint32_t> ids;
  std::vector<float> mat;  // 3 x n world positions at creation
  float transform[16];     // column-major mat4
  float w;
};
struct HostFixedRegion {  // Solver::FixedRegion (Include/Pies/Solver.h:147-151)
  float invInitialTransform[16];
  uint32_t goal;
};

struct Batch {
  uint32_t start, count;
};

// Execution plan of one constraint container: slot -> host index, and conflict-free batches of slots.
struct Plan {
  std::vector<uint32_t> order;
  std::vector<Batch> batches;
};

// Whole-substep dependency levels of schedule EXACT (wavefront.cpp)
constexpr uint64_t kWaveMaxOps = 1ull << 27;  // 512 MB of level indices: beyond this the per-sweep levels are used
struct WavePlan {
  bool active = false;
  std::vector<WaveLevel> levels;
  std::vector<uint32_t> index;         // items of every level, kind after kind
  std::vector<uint32_t> barrierAfter;  // with node-node collisions: the collision pass of iteration i runs after this many levels
};

// Schedule LAYERED (layer_plan.cpp): breadth-first levels of the constraint graph.  A constraint's nodes lie in
// two adjacent levels, so the constraints whose lowest level is l ("group l") touch levels l and l+1 only and
// groups of equal parity are independent: one workgroup sweeps one group with its nodes resident in LDS, colour
// after colour (a colouring inside the group), and a container's sweep is two launches (even groups, odd groups).
struct LayerTile {  // the nodes a workgroup keeps in LDS: two runs of the level-ordered node list (one per level)
  uint32_t first0, count0, first1, count1;
};
struct LayerKind {
  uint32_t ncol[4] = {0, 0, 0, 0};   // colours per tile in each phase (padded to the phase's maximum; 0 = phase unused)
  std::vector<uint32_t> colOff[4];   // per tile of the phase: ncol+1 absolute slot offsets
  std::vector<uint32_t> local;       // stride x count tile-local node indices (16 bit each), slot order
  uint32_t maxClass = 0;             // largest colour class of any tile
};
// A tile's phase = 2 * (parity of its lowest level) + (parity of its strip).  Order in which a container runs its phases:
// the distance container ends where the tetrahedral one starts and vice versa, so those phases share a launch.
constexpr int kLayerPhaseOrder[5][4] = {{0, 1, 2, 3}, {0, 1, 2, 3}, {3, 2, 1, 0}, {0, 1, 2, 3}, {0, 1, 2, 3}};
constexpr uint32_t kLayerMaxGroupNodes = 7936;  // 20 B per node (record + radius) + the colour offsets in the 160 KB LDS
struct LayerPlan {
  bool active = false;
  uint32_t levels = 0;         // breadth-first levels along the longest axis
  uint32_t strips = 1;         // strips of `width` levels of the second levelling (1: a pair of levels is one tile)
  uint32_t width = 1;
  uint32_t maxGroupNodes = 0;  // nodes of the largest tile
  std::vector<uint32_t> nodeList;     // node ids sorted by (level, second level, id)
  std::vector<LayerTile> tiles[4];    // per phase
  LayerKind kind[5];                  // indexed by PIES_POSITION .. PIES_BEND (PIES_VOLUME unused)
};
struct LayerDevice {
  uint32_t* nodeList = nullptr;
  float4* lpos = nullptr;
  float* lrad = nullptr;
  pies::LayerTile* tiles[4] = {nullptr, nullptr, nullptr, nullptr};
  uint32_t* colOff[5][4] = {};
  uint32_t* pc_lid = nullptr;  // 1 x 16 bit in a word
  uint32_t* dc_lid = nullptr;  // a | b << 16
  uint2* tc_lid = nullptr;     // (n1 | n2 << 16, n3 | n4 << 16)
  uint2* bc_lid = nullptr;
};

// In-situ timing of one kernel class (pies_profile_in_situ): whole substeps are launched eagerly and every launch of the
// class is bracketed by two events on the solver's stream.
struct Probe {
  int kernel = -1;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> events;  // begin, end, begin, end, ...
  size_t used = 0;
  bool failed = false;
  void mark() {
    if (used == events.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) { failed = true; return; }
      events.push_back(e);
    }
    if (hipEventRecord(events[used++], stream) != hipSuccess) failed = true;
  }
};

// Tiles of the PD strain + volume local step (pd_tiles.cpp; device form: PdTileArrays), fixed strides per tile
struct PdTilePlan {
  std::vector<uint32_t> info;    // per tile: nodes | elements << 16
  std::vector<uint32_t> node;    // kTileNodes per tile: global node index (unused slots repeat the last node)
  std::vector<uint32_t> elem;    // kTileElems per tile: host element index (unused slots repeat the last element)
  std::vector<uint32_t> local;   // kTileElems per tile
  std::vector<uint16_t> nptr;    // kTileNptr per tile
  std::vector<uint16_t> inc;     // 4 * kTileElems per tile
  uint64_t tileNodes = 0;        // sum of the tiles' node counts
};

template <class T> struct DevArray {
  T* p = nullptr;
  size_t n = 0;
};

}  // namespace pies

struct pies_solver {
  pies_options_t opt;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t sideStream = nullptr;  // second branch of the PD substep: the dependency levels of the contact list (needed only by the
                                     // sequential passes at the end of the substep) are computed beside the local/global iterations
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  bool triLevelsForked = true;
  bool pdLocalPacked = true;    // strain + volume local step two elements per lane in packed fp32 (PIES_PD_LOCAL_PACKED=0: one per lane)
  bool pdSingleCg = true;       // the global step's CG with one launch per iteration where it applies (PIES_PD_CG_SINGLE=0: the two-launch form everywhere)
  bool pdSingleCgRows = true;   // ... also in the contact-heavy graph variant (merged contact rows gathered inline)
  bool pdFuseRhs = true;        // its residual kernel evaluates the right-hand side when a node's records are tile sums (PIES_PD_FUSE_RHS=0: k_pd_rhs)
  bool pcgOverflow = true;      // a solve above the tolerance after its captured iterations goes on inside the last launch (PIES_PCG_OVERFLOW=0: off)
  bool pcgPinned = false;       // PIES_PCG_BUDGET
  uint32_t pcgPinnedBudget = 32;
  uint32_t asyncSinceSync = 0;  // PD ticks queued by pies_tick_async since the last host synchronisation
  std::string error;

  bool releaseHinge = false;
  bool nodeCollisions = true;
  bool collideFast = true;         // every node's cell range spans at most 2 cells per axis: the parallel visiting order may run
  uint16_t* d_pairDictIndex = nullptr;  // PD, paired elements: index of the element's set of constants (rest dictionary), or nullptr
  float4* d_pairDictTable = nullptr;
  uint32_t pairDictSets = 0;
  std::vector<uint16_t> h_pairDictIndex;  // host copy (the tile plan stores it in tile order)
  uint32_t pdTileRecords = 0;            // sum of the tiles' node counts
  uint32_t pdTiles = 0;                  // PD: tiles of the strain + volume local step (0: per-(element, node) records)
  float pdWindowPadding = 0.0f;         // PD: stored / real entries of the windowed matrix (0: not built)
  uint32_t pdWindowEntries = 0;         // its stored entries
  uint32_t pdWindowHalo = 0;            // its halo entries over all chunks
  uint32_t pdRowStencils = 0;           // PD: distinct rows of the system matrix in its row dictionary (0: none)
  bool tetVolumePaired = false;    // PD: h_volume[k] and h_tet[k] are the same element for every k (fused local step)
  bool triangleCollisions = true;  // PD point-triangle CCD contacts (Solver.cpp:693-797); extension flag to switch off
  bool simFailed = false;
  int schedule = PIES_SCHEDULE_DEFAULT;
  int collisionOrderFlag = -1;     // PIES_FLAG_COLLISION_ORDER: -1 follows the schedule (EXACT: reference order, otherwise pair order)
  uint32_t sortPasses = 3;         // radix passes captured per node-grid build (up to 11 key bits each); follows the scene's cell box
  uint32_t sortCalm = 0;
  bool pairRoundsPinned = false;   // pies_set_collision_rounds: the count is the host's
  uint32_t pairCalm = 0;           // synchronisations in a row at which fewer level launches would have done
  uint32_t pairRounds = 96;        // level launches captured per pair-ordered pass (a tail kernel finishes deeper orders)

  // ---- host mirror ----
  std::vector<float> h_pos, h_prev, h_vel;  // n x 3
  std::vector<float> h_radius, h_invMass;   // n
  std::vector<pies::HostPosition> h_position;
  std::vector<pies::HostDistance> h_distance;
  std::vector<pies::HostTet> h_tet, h_volume;
  std::vector<pies::HostBend> h_bend;
  std::vector<pies::HostNodePair> h_nodePair;  // extension container (PD): pies_add_node_pair_constraints
  std::vector<pies::HostShape> h_shape;
  std::vector<pies::HostGoal> h_goal;
  std::vector<pies::HostFixedRegion> h_fixedRegions;
  bool goalDirty = false;  // a goal transform changed: refresh its projected positions in HBM
  std::vector<uint32_t> h_triangles;  // 3 per triangle
  std::vector<uint32_t> h_lines;
  uint32_t constraintId = 0;

  bool sceneDirty = true;    // topology/rest data changed: rebuild plans + upload everything
  // which host mirror arrays are older than the device's (bit 0 positions, 1 previous positions, 2 velocities):
  // a read of one array copies that array only, through the pinned staging buffer
  uint32_t stale = 0;
  bool graphDirty = false;   // only the launch sequence changed (releaseHinge, CG budget): re-capture, no re-upload
  bool hostNodesDirty = false;  // host node state edited (pies_write_nodes): upload nodes only

  // ---- plans ----
  pies::Plan plan[5];  // PIES_POSITION .. PIES_BEND
  pies::WavePlan wave;  // schedule EXACT, PBD: levels of the whole-substep DAG
  uint32_t* d_waveIndex = nullptr;
  pies::LayerPlan layer;  // schedule LAYERED, PBD
  pies::LayerDevice d_layer;

  // ---- HBM ----
  pies::NodeArrays nd{nullptr, nullptr, nullptr, nullptr, 0};
  uint32_t* d_pc_id = nullptr;
  float4* d_pc_tw = nullptr;
  uint2* d_dc_ids = nullptr;
  float2* d_dc_rw = nullptr;
  uint4* d_tc_ids = nullptr;
  float4 *d_tc_q0 = nullptr, *d_tc_q1 = nullptr, *d_tc_q2 = nullptr;
  uint4* d_bc_ids = nullptr;
  float2* d_bc_aw = nullptr;
  uint2* d_np_ids = nullptr;  // node-pair extension (PD)
  uint4* d_vc_ids = nullptr;  // volume constraints (PD only), host order
  float4 *d_vc_q0 = nullptr, *d_vc_q1 = nullptr, *d_vc_q2 = nullptr;

  // ---- node-node collisions (PBD) ----
  pies::HashArrays hash{};
  pies::PairArrays pairs{};  // pair-ordered resolve

  // ---- Projective Dynamics ----
  pies::PdArrays pd{};
  uint32_t slotBase[6] = {0, 0, 0, 0, 0, 0};  // first contribution slot of each container (5: the node-pair extension)
  uint32_t pd_nnz = 0;
  uint32_t goalSlotBase = 0;  // first fp64 contribution slot of the goal constraints
  float pcgTol = 3.0e-7f;     // relative residual ||r|| / ||b|| per coordinate column
  uint32_t pcgMaxIters = 128; // upper bound of CG iterations per global step (thousands of w = 1e4 contacts need 40+)
  uint32_t pcgBudget = 32;    // iterations currently captured in the graph (adapted to what the solves use)
  uint32_t pcgCalm = 0;       // synchronisations in the current observation window (all solves converged)
  uint32_t pcgWindowMax = 0;  // most CG iterations any solve used in that window
  uint32_t pcgRecent[3] = {0, 0, 0};  // ... and at the last three synchronisations
  bool triFastRows = false;     // PD graph variant: contact rows of the SpMV summed by k_contact_rows (many contacts)
  uint32_t triQuiet = 0;        // synchronisations without a contact while that variant is active
  uint32_t pcgCooldown = 0;   // synchronisations left before the budget may shrink again after a solve ran out
  std::vector<void*> allocations;
  float4* h_stage = nullptr;  // pinned staging for the per-tick position read-back
  float* d_pack = nullptr;    // n x 3: a node array's x, y, z packed for the read-back (freed with the device state)
  size_t h_stage_n = 0;
  // ---- render-state export (Solver.h:42-71): frame k is copied out while frame k+1 computes ----
  hipStream_t copyStream = nullptr;
  hipEvent_t evTick[2] = {nullptr, nullptr};    // frame's positions are in d_export
  hipEvent_t evCopied[2] = {nullptr, nullptr};  // frame's positions are in h_export[frame & 1]
  float4* h_export[2] = {nullptr, nullptr};     // pinned
  size_t h_export_n = 0;
  float4* d_export = nullptr;
  uint64_t frameBegun = 0;      // frames begun so far (frame ids start at 1)
  uint64_t frameAcquired = 0;   // frame the host currently holds (0: none)
  // ---- PD: a substep whose solve ends above the tolerance is run again with a larger CG budget (pies_tick) ----
  float4 *snapPos = nullptr, *snapPrev = nullptr, *snapVel = nullptr;
  double* snapQuat = nullptr;
  bool pcgRetry = true;
  uint32_t pcgRetries = 0;         // substeps run again since the handle was created
  uint64_t pcgShortSolves = 0;     // solves left above the tolerance (budget at its ceiling, or asynchronous ticks)

  hipGraph_t graph = nullptr;
  hipGraphExec_t graphExec = nullptr;
  std::vector<std::pair<hipGraph_t, hipGraphExec_t>> retiredGraphs;  // profile-pass graphs, freed with the handle
  // PD: the substep graph exists once per (captured CG iterations, contact-row variant) - a ladder of budgets 2, 4, 8, ... up
  // to the ceiling of pies_set_pcg, instantiated together - so that following the solves means launching another executable
  // graph, not capturing and instantiating one in the middle of a frame (~11 ms).  graph / graphExec then alias an entry.
  struct PdGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; uint32_t counts[32] = {0}; };
  std::map<uint64_t, PdGraph> pdLadder;
  bool graphFromLadder = false;
  uint32_t launchCounts[PIES_KERNEL_COUNT] = {};
  pies::Probe* probe = nullptr;  // set for the duration of pies_profile_in_situ

  uint32_t nodeCount() const { return static_cast<uint32_t>(h_radius.size()); }
};

namespace pies {
// scene.cpp
int fail(pies_solver* s, int code, const std::string& msg);
// schedule.cpp : op = up to 4 node ids; bit i of writeMask set if node i is written.
struct OpView {
  const uint32_t* ids;  // count * stride entries
  uint32_t stride;      // node ids per op
  uint32_t count;
  uint8_t writeMask;    // same for every op of a container
  const uint16_t* hint = nullptr;  // optional proposed colouring (count entries)
};
void build_plan(const OpView& ops, uint32_t nodeCount, int schedule, Plan& out);
bool build_wave_plan(const pies_solver* s, WavePlan& out);
// layer_plan.cpp : fills s->layer and the four PBD plans; false (nothing changed) when the scene does not suit it
bool build_layer_plan(pies_solver* s);
}  // namespace pies
