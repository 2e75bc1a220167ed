// PBD substep kernels for gfx950: prediction, per-constraint projections (one lane = one constraint of
// a conflict-free batch), floor clamp, velocity update.
//
// Data layout: node state is struct-of-arrays of 16-byte records (pos4 = x,y,z,invMass; prev4; vel4;
// radius[]), so the streaming kernels move exactly one dwordx4 per lane per array, fully coalesced,
// and a constraint's node gather is one 16-byte load per node.  Constraint records are stored in
// schedule order (batch after batch), as SoA of 8/16-byte records, so every per-constraint read is
// coalesced; only the node gather/scatter is indexed.
//
// These kernels are bound by memory latency/bandwidth (gather -> <=1 kFLOP of 3x3 algebra -> scatter);
// there is no GEMM-shaped work, hence no MFMA.
#include <cstdlib>

#include "dev_math.h"
#include "kernels.h"
#include "pbd_project.h"

namespace pies {

constexpr int kBlock = 256;  // streaming kernels: 4 waves of 64
#ifndef PIES_PROJ_BLOCK
#define PIES_PROJ_BLOCK 256
#endif
constexpr int kProjBlock = PIES_PROJ_BLOCK;  // projection kernels (64 measured no faster at 100k, slower at 1M)

static inline dim3 grid_for(uint32_t n, int block) { return dim3((n + block - 1) / block); }

#define PIES_LAUNCH(kernel, block, n_items, st, ...) \
  hipLaunchKernelGGL(kernel, grid_for(n_items, block), dim3(block), 0, st, __VA_ARGS__)

// ----------------------------------------------------------------------------------------------
// Solver.cpp:47-52   prev = pos;  pos += v*dt + (0,-g,0)*dt*dt
// 48 B/node: read pos, vel; write prev, pos.
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_predict(float4* __restrict__ pos, float4* __restrict__ prev,
                                                    const float4* __restrict__ vel, uint32_t n, float dt, float g) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float4 p = pos[i];
  const float4 v = vel[i];
  prev[i] = make_float4(p.x, p.y, p.z, 0.0f);
  predict_core(p, v, dt, g);
  pos[i] = p;
}

// Solver.cpp:132-136
PIES_DEV void clamp_floor(float4* __restrict__ pos, const float* __restrict__ radius, uint32_t i, float floorHeight) {
  float4 p = pos[i];
  if (floor_core(p, radius[i], floorHeight)) pos[i] = p;
}
__global__ void __launch_bounds__(kBlock) k_floor(float4* __restrict__ pos, const float* __restrict__ radius, uint32_t n,
                                                  float floorHeight) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  clamp_floor(pos, radius, i, floorHeight);
}

// Solver.cpp:140-158   v = (1-damping)*(pos-prev)/dt, floor friction with the hard-coded speed 5.0
__global__ void __launch_bounds__(kBlock) k_velocity(const float4* __restrict__ pos, const float4* __restrict__ prev,
                                                     float4* __restrict__ vel, const float* __restrict__ radius, uint32_t n,
                                                     float dt, float damping, float friction, float floorHeight) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  vel[i] = velocity_core(pos[i], prev[i], radius[i], dt, damping, friction, floorHeight);
}

// ----------------------------------------------------------------------------------------------
// PositionConstraint (Constraints.cpp:58-63 through Constraints.h:121-129): pos += w*(fixed - pos)
// ----------------------------------------------------------------------------------------------
PIES_DEV void project_position(float4* __restrict__ pos, const uint32_t* __restrict__ ids, const float4* __restrict__ target_w, uint32_t c) {
  const uint32_t id = ids[c];
  const float4 tw = target_w[c];
  float4 p = pos[id];
  position_core(p, tw);
  pos[id] = p;
}
__global__ void __launch_bounds__(kProjBlock) k_position(float4* __restrict__ pos, const uint32_t* __restrict__ ids,
                                                     const float4* __restrict__ target_w, uint32_t start, uint32_t count) {
  const uint32_t t = xcd_block(blockIdx.x, gridDim.x) * kProjBlock + threadIdx.x;
  if (t >= count) return;
  project_position(pos, ids, target_w, start + t);
}

// ----------------------------------------------------------------------------------------------
// DistanceConstraint (Constraints.cpp:11-37): only node a moves, by the full correction.
// 52 algorithmic B/projection: ids 8 + rest,w 8 + two position reads 24 + one position write 12.
// ----------------------------------------------------------------------------------------------
PIES_DEV void project_distance(float4* __restrict__ pos, const uint2* __restrict__ ids, const float2* __restrict__ rest_w, uint32_t c) {
  const uint2 id = ids[c];
  const float2 rw = rest_w[c];
  float4 a = pos[id.x];
  const float4 b = pos[id.y];
  distance_core(a, b, rw);
  pos[id.x] = a;
}
__global__ void __launch_bounds__(kProjBlock) k_distance(float4* __restrict__ pos, const uint2* __restrict__ ids,
                                                     const float2* __restrict__ rest_w, uint32_t start, uint32_t count) {
  const uint32_t t = xcd_block(blockIdx.x, gridDim.x) * kProjBlock + threadIdx.x;
  if (t >= count) return;
  project_distance(pos, ids, rest_w, start + t);
}

// ----------------------------------------------------------------------------------------------
// TetrahedralConstraint (Constraints.cpp:76-128) applied as a PBD projection (Constraints.h:121-129).
// 160 algorithmic B/projection: ids 16 + Qinv 36 + (min,max,w) 12 + 4 position reads 48 + 4 writes 48.
// Record layout: q0 = Qinv col0 + Qinv[1][0], q1 = Qinv[1][1..2] + Qinv[2][0..1], q2 = Qinv[2][2], min, max, w.
// ----------------------------------------------------------------------------------------------
template <int VARIANT>
PIES_DEV void project_tet(float4* __restrict__ pos, const uint4* __restrict__ ids, const float4* __restrict__ q0,
                          const float4* __restrict__ q1, const float4* __restrict__ q2, uint32_t c) {
  const uint4 id = ids[c];
  const float4 a0 = q0[c], a1 = q1[c], a2 = q2[c];
  float4 x1 = pos[id.x], x2 = pos[id.y], x3 = pos[id.z], x4 = pos[id.w];
  tet_core<VARIANT>(x1, x2, x3, x4, a0, a1, a2);
  pos[id.x] = x1;
  pos[id.y] = x2;
  pos[id.z] = x3;
  pos[id.w] = x4;
}
template <int VARIANT>
__global__ void __launch_bounds__(kProjBlock) k_tet(float4* __restrict__ pos, const uint4* __restrict__ ids,
                                                const float4* __restrict__ q0, const float4* __restrict__ q1,
                                                const float4* __restrict__ q2, uint32_t start, uint32_t count) {
  const uint32_t t = xcd_block(blockIdx.x, gridDim.x) * kProjBlock + threadIdx.x;
  if (t >= count) return;
  project_tet<VARIANT>(pos, ids, q0, q1, q2, start + t);
}

// ----------------------------------------------------------------------------------------------
// BendConstraint (Constraints.cpp:312-366): dihedral-angle projection, mass weighted.
// 136 algorithmic B/projection: ids 16 + angle,w 8 + 4 x (pos 12 + invMass 4) reads 64 + 4 writes 48.
// ----------------------------------------------------------------------------------------------
PIES_DEV void project_bend(float4* __restrict__ pos, const uint4* __restrict__ ids, const float2* __restrict__ angle_w, uint32_t c) {
  const uint4 id = ids[c];
  const float2 aw = angle_w[c];
  float4 x1 = pos[id.x], x2 = pos[id.y], x3 = pos[id.z], x4 = pos[id.w];
  if (!bend_core(x1, x2, x3, x4, aw)) return;
  pos[id.x] = x1;
  pos[id.y] = x2;
  pos[id.z] = x3;
  pos[id.w] = x4;
}
__global__ void __launch_bounds__(kProjBlock) k_bend(float4* __restrict__ pos, const uint4* __restrict__ ids,
                                                 const float2* __restrict__ angle_w, uint32_t start, uint32_t count) {
  const uint32_t t = xcd_block(blockIdx.x, gridDim.x) * kProjBlock + threadIdx.x;
  if (t >= count) return;
  project_bend(pos, ids, angle_w, start + t);
}

// ----------------------------------------------------------------------------------------------
// One dependency level of the whole-substep DAG (schedule EXACT, wavefront.cpp): the projections and floor clamps
// whose predecessors in the reference's sequential order have all run, whatever their type or iteration.  Blocks
// are homogeneous: the first ceil(cnt[0]/256) blocks run position projections, the next ones distance, ...
// index[off[k] + t] is the slot of the t-th item of kind k (a node index for the floor clamp).
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kProjBlock) k_wave(float4* __restrict__ pos, const float* __restrict__ radius, float floorHeight,
                                                     const uint32_t* __restrict__ index, WaveLevel L, WaveData W) {
  uint32_t b = blockIdx.x;
#pragma unroll
  for (int kind = 0; kind < kWaveKinds; ++kind) {
    const uint32_t nb = (L.cnt[kind] + kProjBlock - 1) / kProjBlock;
    if (b < nb) {
      const uint32_t t = b * kProjBlock + threadIdx.x;
      if (t >= L.cnt[kind]) return;
      const uint32_t c = index[L.off[kind] + t];
      if (kind == 0) project_position(pos, W.pc_id, W.pc_tw, c);
      else if (kind == 1) project_distance(pos, W.dc_ids, W.dc_rw, c);
      else if (kind == 2) project_tet<0>(pos, W.tc_ids, W.tc_q0, W.tc_q1, W.tc_q2, c);
      else if (kind == 3) project_bend(pos, W.bc_ids, W.bc_aw, c);
      else clamp_floor(pos, radius, c, floorHeight);
      return;
    }
    b -= nb;
  }
}

// ----------------------------------------------------------------------------------------------
void launch_wave(hipStream_t st, const NodeArrays& nd, float floorHeight, const uint32_t* index, const WaveLevel& L, const WaveData& W) {
  uint32_t blocks = 0;
  for (int kind = 0; kind < kWaveKinds; ++kind) blocks += (L.cnt[kind] + kProjBlock - 1) / kProjBlock;
  if (blocks == 0) return;
  hipLaunchKernelGGL(k_wave, dim3(blocks), dim3(kProjBlock), 0, st, nd.pos, nd.radius, floorHeight, index, L, W);
}
void launch_predict(hipStream_t st, const NodeArrays& nd, float dt, float gravity) {
  if (nd.n == 0) return;
  PIES_LAUNCH(k_predict, kBlock, nd.n, st, nd.pos, nd.prev, nd.vel, nd.n, dt, gravity);
}
void launch_floor(hipStream_t st, const NodeArrays& nd, float floorHeight) {
  if (nd.n == 0) return;
  PIES_LAUNCH(k_floor, kBlock, nd.n, st, nd.pos, nd.radius, nd.n, floorHeight);
}
void launch_velocity(hipStream_t st, const NodeArrays& nd, float dt, float damping, float friction, float floorHeight) {
  if (nd.n == 0) return;
  PIES_LAUNCH(k_velocity, kBlock, nd.n, st, nd.pos, nd.prev, nd.vel, nd.radius, nd.n, dt, damping, friction, floorHeight);
}
__global__ void k_noop() {}
// the x, y, z of n records of four floats, packed: what the host mirrors keep (pies_tick's read-back moves 12 bytes per node)
__global__ void __launch_bounds__(kBlock) k_pack_xyz(const float4* __restrict__ src, float* __restrict__ dst, uint32_t n) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float4 v = src[i];
  dst[3 * i] = v.x;
  dst[3 * i + 1] = v.y;
  dst[3 * i + 2] = v.z;
}
void launch_pack_xyz(hipStream_t st, const float4* src, float* dst, uint32_t n) {
  if (n) hipLaunchKernelGGL(k_pack_xyz, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, src, dst, n);
}
void launch_noop(hipStream_t st) { hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, st); }
void launch_position(hipStream_t st, float4* pos, const uint32_t* ids, const float4* target_w, uint32_t start,
                     uint32_t count) {
  if (count == 0) return;
  PIES_LAUNCH(k_position, kProjBlock, count, st, pos, ids, target_w, start, count);
}
void launch_distance(hipStream_t st, float4* pos, const uint2* ids, const float2* rest_w, uint32_t start, uint32_t count) {
  if (count == 0) return;
  PIES_LAUNCH(k_distance, kProjBlock, count, st, pos, ids, rest_w, start, count);
}
void launch_tet(hipStream_t st, float4* pos, const uint4* ids, const float4* q0, const float4* q1, const float4* q2,
                uint32_t start, uint32_t count) {
  if (count == 0) return;
#ifdef PIES_EXPERIMENTS  // timing experiments that change the arithmetic: never part of the product build (build.py)
  static const int variant = [] { const char* e = getenv("PIES_EXP_TET"); return e ? atoi(e) : 0; }();
  if (variant == 1) { PIES_LAUNCH(k_tet<1>, kProjBlock, count, st, pos, ids, q0, q1, q2, start, count); return; }
#endif
  PIES_LAUNCH(k_tet<0>, kProjBlock, count, st, pos, ids, q0, q1, q2, start, count);
}
void launch_bend(hipStream_t st, float4* pos, const uint4* ids, const float2* angle_w, uint32_t start, uint32_t count) {
  if (count == 0) return;
  PIES_LAUNCH(k_bend, kProjBlock, count, st, pos, ids, angle_w, start, count);
}

}  // namespace pies
