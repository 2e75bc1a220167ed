// Launch wrappers of the HIP kernels (implemented in *.hip).  Host-callable, graph-capturable:
// no allocation, no synchronisation, everything on the caller's stream.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pies {

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
