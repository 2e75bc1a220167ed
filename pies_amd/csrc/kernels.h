// Launch wrappers of the HIP kernels (implemented in *.hip).  Host-callable, graph-capturable:
// no allocation, no synchronisation, everything on the caller's stream.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pies {

// Tuning / diagnostic switches that change what is captured (graph variants, CG budget, tile sizes, ...).  They are set with
// pies_set_tuning (process wide, by name) - NOT read from the environment: a host's environment cannot alter a shipped solver.
// Returns the value as a string, or nullptr when the switch is not set.  (capi.cpp)
const char* tuning_env(const char* name);

struct NodeArrays {
  float4* pos;    // x, y, z, invMass
  float4* prev;   // x, y, z, (unused)
  float4* vel;    // x, y, z, (unused)
  float* radius;
  uint32_t n;
};

// One dependency level of the whole-substep DAG of schedule EXACT (wavefront.cpp): items of kind k are
// index[off[k] .. off[k] + cnt[k]), slots of the container arrays (node indices for the floor clamp).
constexpr int kWaveKinds = 5;  // position, distance, tet, bend, floor clamp: the order tickPBD visits them
struct WaveLevel {
  uint32_t off[kWaveKinds];
  uint32_t cnt[kWaveKinds];
};
struct WaveData {  // the containers' device arrays, in plan (slot) order
  const uint32_t* pc_id;
  const float4* pc_tw;
  const uint2* dc_ids;
  const float2* dc_rw;
  const uint4* tc_ids;
  const float4 *tc_q0, *tc_q1, *tc_q2;
  const uint4* bc_ids;
  const float2* bc_aw;
};
void launch_wave(hipStream_t st, const NodeArrays& nd, float floorHeight, const uint32_t* index, const WaveLevel& L, const WaveData& W);

// Schedule LAYERED (layer_plan.cpp, layer_kernels.hip): one launch = the groups of one parity, one workgroup per
// group with the group's node records in LDS, running a short list of segments (a container phase or a per-node
// step) colour after colour.
enum { LAYER_POSITION = 0, LAYER_DISTANCE = 1, LAYER_TET = 2, LAYER_BEND = 3, LAYER_FLOOR = 4, LAYER_PREDICT = 5, LAYER_VELOCITY = 6 };
constexpr int kLayerMaxSegs = 6;
constexpr int kLayerMaxCols = 128;  // colours of one segment inside a group (layer_plan.cpp gives up beyond)
struct LayerSeg {
  uint32_t kind;
  uint32_t ncol;           // colours per group (0 for the per-node kinds)
  const uint32_t* colOff;  // groups x (ncol+1) slot offsets
};
struct LayerLaunch {
  uint32_t phase, groups, nseg, maxClass;  // phase = 2 * (level parity) + (strip parity); groups = tiles of that phase
  uint32_t loadGlobal, storeGlobal;  // node records from / to the node array instead of the layer-ordered copy
  const uint4* tileList;             // = LayerData::tiles[phase], resolved by launch_layer (one dependent load less at the kernel's start)
  uint32_t needRadius;               // a floor / velocity segment is in the launch (set by launch_layer)
  LayerSeg seg[kLayerMaxSegs];
#ifdef PIES_EXPERIMENTS
  uint32_t stampSlot;  // diagnostic build: the launch's place in the time-stamp buffer (layer_kernels.hip)
#endif
};
struct LayerData {
  const uint32_t* nodeList;
  const uint4* tiles[4];  // per phase: (first0, count0, first1, count1) = two runs of the level-ordered node list
  uint32_t maxGroupNodes;
  float4* lpos;       // node records in nodeList order: valid between the layer launches of a substep
  const float* lrad;  // radii in nodeList order
  const uint32_t* pc_lid;
  const float4* pc_tw;
  const uint32_t* dc_lid;
  const float2* dc_rw;
  const uint2* tc_lid;
  const float4 *tc_q0, *tc_q1, *tc_q2;
  const uint2* bc_lid;
  const float2* bc_aw;
};
struct LayerParams {  // scalars of the per-node steps
  float floorHeight, dt, gravity, damping, friction;
};
// Returns hipSuccess or the error of the attribute call that raises the kernel's LDS limit.
hipError_t layer_prepare(uint32_t maxGroupNodes);
void launch_layer(hipStream_t st, const NodeArrays& nd, const LayerData& D, const LayerLaunch& L, const LayerParams& P);
// Per-node steps on the level-ordered copy, for bodies whose levels are cut into strips (no tile partition covers every
// node exactly once there): predict reads the node array and fills the copy, velocity writes the node array back.
void launch_lpredict(hipStream_t st, const NodeArrays& nd, const LayerData& D, const LayerParams& P);
void launch_lvelocity(hipStream_t st, const NodeArrays& nd, const LayerData& D, const LayerParams& P);
void launch_lfloor(hipStream_t st, const NodeArrays& nd, const LayerData& D, const LayerParams& P);
void launch_lposition(hipStream_t st, const LayerData& D, uint32_t start, uint32_t count);
void launch_lcopy(hipStream_t st, const NodeArrays& nd, const LayerData& D, bool toNodeArray);

// an empty kernel (one wavefront): calibrates the cost of an event bracket (pies_profile_in_situ)
void launch_noop(hipStream_t st);
// x, y, z of n four-float records, packed (the host mirrors' layout)
void launch_pack_xyz(hipStream_t st, const float4* src, float* dst, uint32_t n);
// Solver.cpp:47-52
void launch_predict(hipStream_t st, const NodeArrays& nd, float dt, float gravity);
// Solver.cpp:132-136
void launch_floor(hipStream_t st, const NodeArrays& nd, float floorHeight);
// Solver.cpp:140-158
void launch_velocity(hipStream_t st, const NodeArrays& nd, float dt, float damping, float friction, float floorHeight);

// One conflict-free batch [start, start+count) of a constraint container.
// PositionConstraint: Constraints.h:121-129 + Constraints.cpp:58-63
void launch_position(hipStream_t st, float4* pos, const uint32_t* ids, const float4* target_w, uint32_t start,
                     uint32_t count);
// DistanceConstraint: Constraints.h:121-129 + Constraints.cpp:11-37
void launch_distance(hipStream_t st, float4* pos, const uint2* ids, const float2* rest_w, uint32_t start, uint32_t count);
// TetrahedralConstraint: Constraints.h:121-129 + Constraints.cpp:76-128
void launch_tet(hipStream_t st, float4* pos, const uint4* ids, const float4* q0, const float4* q1, const float4* q2,
                uint32_t start, uint32_t count);
// BendConstraint: Constraints.h:121-129 + Constraints.cpp:312-366
void launch_bend(hipStream_t st, float4* pos, const uint4* ids, const float2* angle_w, uint32_t start, uint32_t count);

}  // namespace pies
