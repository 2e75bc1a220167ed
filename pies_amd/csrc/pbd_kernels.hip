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
  const float gx = (0.0f * dt) * dt, gy = (-g * dt) * dt, gz = (0.0f * dt) * dt;
  p.x = p.x + (v.x * dt + gx);
  p.y = p.y + (v.y * dt + gy);
  p.z = p.z + (v.z * dt + gz);
  pos[i] = p;
}

// Solver.cpp:132-136
PIES_DEV void clamp_floor(float4* __restrict__ pos, const float* __restrict__ radius, uint32_t i, float floorHeight) {
  float4 p = pos[i];
  const float r = radius[i];
  if (p.y - r < floorHeight) {
    p.y = floorHeight + r;
    pos[i] = p;
  }
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
  const float4 p = pos[i];
  const float4 q = prev[i];
  const float k = 1.0f - damping;
  float vx = (k * (p.x - q.x)) / dt;
  float vy = (k * (p.y - q.y)) / dt;
  float vz = (k * (p.z - q.z)) / dt;
  if (p.y - radius[i] <= floorHeight) {
    const float l = sqrtf(vx * vx + vz * vz);
    if (l < 5.0f) {
      vx = 0.0f;
      vz = 0.0f;
    } else {
      vx *= 1.0f - friction;
      vz *= 1.0f - friction;
    }
  }
  vel[i] = make_float4(vx, vy, vz, 0.0f);
}

// ----------------------------------------------------------------------------------------------
// PositionConstraint (Constraints.cpp:58-63 through Constraints.h:121-129): pos += w*(fixed - pos)
// ----------------------------------------------------------------------------------------------
PIES_DEV void project_position(float4* __restrict__ pos, const uint32_t* __restrict__ ids, const float4* __restrict__ target_w, uint32_t c) {
  const uint32_t id = ids[c];
  const float4 tw = target_w[c];
  float4 p = pos[id];
  p.x += tw.w * (tw.x - p.x);
  p.y += tw.w * (tw.y - p.y);
  p.z += tw.w * (tw.z - p.z);
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
  const float dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
  const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
  float ux = 1.0f, uy = 0.0f, uz = 0.0f;
  if (dist > 0.00001f) {
    ux = dx / dist;
    uy = dy / dist;
    uz = dz / dist;
  }
  const float nd = -(rw.x - dist);  // -disp
  const float px = a.x + nd * ux, py = a.y + nd * uy, pz = a.z + nd * uz;
  a.x += rw.y * (px - a.x);
  a.y += rw.y * (py - a.y);
  a.z += rw.y * (pz - a.z);
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

  const float qi[3][3] = {{a0.x, a0.y, a0.z}, {a0.w, a1.x, a1.y}, {a1.z, a1.w, a2.x}};  // [col][row]
  const float minStrain = a2.y, maxStrain = a2.z, w = a2.w;
  const float P[3][3] = {{x2.x - x1.x, x2.y - x1.y, x2.z - x1.z},
                         {x3.x - x1.x, x3.y - x1.y, x3.z - x1.z},
                         {x4.x - x1.x, x4.y - x1.y, x4.z - x1.z}};
  float F[3][3];
  mat3_mul_cm(P, qi, F);
  // the reference hands Eigen the matrix F_(r,c) = F[r][c] (Constraints.cpp:93-95)
  if (VARIANT == 1) {  // experiment: no SVD at all (memory/latency floor)
    x2.x += w * (F[0][0] - x2.x); x3.y += w * (F[1][1] - x3.y); x4.z += w * (F[2][2] - x4.z); x1.x += w * (F[0][1] - x1.x);
    pos[id.x] = x1; pos[id.y] = x2; pos[id.z] = x3; pos[id.w] = x4;
    return;
  }
  Svd3 d;
  svd3(F, d);
  float s[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) s[i] = clampf(d.s[i], minStrain, maxStrain);
  if (det3_cm(F) < 0.0f) {  // flip the smallest singular value (Constraints.cpp:106-108)
    int k = 0;
    float m = d.s[0];
    if (d.s[1] <= m) { k = 1; m = d.s[1]; }
    if (d.s[2] <= m) { k = 2; }
    s[0] = (k == 0) ? -s[0] : s[0];
    s[1] = (k == 1) ? -s[1] : s[1];
    s[2] = (k == 2) ? -s[2] : s[2];
  }
  float Fh[3][3];
  svd3_recompose(d, s, Fh);
  // projected = (0, Fh row 0, Fh row 1, Fh row 2) (Constraints.cpp:113-127); pos += w*(proj - pos)
  x1.x += w * (0.0f - x1.x);
  x1.y += w * (0.0f - x1.y);
  x1.z += w * (0.0f - x1.z);
  x2.x += w * (Fh[0][0] - x2.x);
  x2.y += w * (Fh[0][1] - x2.y);
  x2.z += w * (Fh[0][2] - x2.z);
  x3.x += w * (Fh[1][0] - x3.x);
  x3.y += w * (Fh[1][1] - x3.y);
  x3.z += w * (Fh[1][2] - x3.z);
  x4.x += w * (Fh[2][0] - x4.x);
  x4.y += w * (Fh[2][1] - x4.y);
  x4.z += w * (Fh[2][2] - x4.z);
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
struct V3 {
  float x, y, z;
};
PIES_DEV V3 sub(const float4& a, const float4& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PIES_DEV V3 crossv(const V3& a, const V3& b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
PIES_DEV float dotv(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PIES_DEV V3 addv(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PIES_DEV V3 subv(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PIES_DEV V3 mulv(const V3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PIES_DEV V3 divv(const V3& a, float s) { return {a.x / s, a.y / s, a.z / s}; }
PIES_DEV V3 negv(const V3& a) { return {-a.x, -a.y, -a.z}; }

PIES_DEV void project_bend(float4* __restrict__ pos, const uint4* __restrict__ ids, const float2* __restrict__ angle_w, uint32_t c) {
  const uint4 id = ids[c];
  const float2 aw = angle_w[c];
  float4 x1 = pos[id.x], x2 = pos[id.y], x3 = pos[id.z], x4 = pos[id.w];
  const V3 p2 = sub(x2, x1), p3 = sub(x3, x1), p4 = sub(x4, x1);
  const V3 c23 = crossv(p2, p3), c24 = crossv(p2, p4);
  const float l23 = sqrtf(dotv(c23, c23)), l24 = sqrtf(dotv(c24, c24));
  const V3 n1 = divv(c23, l23), n2 = divv(c24, l24);
  const float d = dotv(n1, n2);
  // acos in double, rounded once: the correctly rounded float value, identical on host and device
  const float C = static_cast<float>(acos(static_cast<double>(d))) - aw.x;
  const V3 q3 = divv(addv(crossv(p2, n2), mulv(crossv(n1, p2), d)), l23);
  const V3 q4 = divv(addv(crossv(p2, n1), mulv(crossv(n2, p2), d)), l24);
  const V3 q2 = subv(negv(divv(addv(crossv(p3, n2), mulv(crossv(n1, p3), d)), l23)),
                     divv(addv(crossv(p4, n1), mulv(crossv(n2, p4), d)), l24));
  const V3 q1 = subv(subv(negv(q2), q3), q4);
  const float wSum = x1.w + x2.w + x3.w + x4.w;
  const float qSq = dotv(q1, q1) + dotv(q2, q2) + dotv(q3, q3) + dotv(q4, q4);
  const float num = sqrtf(fmaxf(1.0f - d * d, 0.0f)) * C;
  if (qSq < 0.00001f) return;  // projection = current positions: pos += w*0
  const float w = aw.y;
  const V3 d1 = divv(mulv(mulv(negv(q1), 4 * x1.w / wSum), num), qSq);
  const V3 d2 = divv(mulv(mulv(negv(q2), 4 * x2.w / wSum), num), qSq);
  const V3 d3 = divv(mulv(mulv(negv(q3), 4 * x3.w / wSum), num), qSq);
  const V3 d4 = divv(mulv(mulv(negv(q4), 4 * x4.w / wSum), num), qSq);
  // projected_i = x_i + d_i ; pos_i += w * (projected_i - pos_i)
  x1.x += w * ((x1.x + d1.x) - x1.x); x1.y += w * ((x1.y + d1.y) - x1.y); x1.z += w * ((x1.z + d1.z) - x1.z);
  x2.x += w * ((x2.x + d2.x) - x2.x); x2.y += w * ((x2.y + d2.y) - x2.y); x2.z += w * ((x2.z + d2.z) - x2.z);
  x3.x += w * ((x3.x + d3.x) - x3.x); x3.y += w * ((x3.y + d3.y) - x3.y); x3.z += w * ((x3.z + d3.z) - x3.z);
  x4.x += w * ((x4.x + d4.x) - x4.x); x4.y += w * ((x4.y + d4.y) - x4.y); x4.z += w * ((x4.z + d4.z) - x4.z);
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
  static const int variant = [] { const char* e = getenv("PIES_EXP_TET"); return e ? atoi(e) : 0; }();
  if (variant == 1) { PIES_LAUNCH(k_tet<1>, kProjBlock, count, st, pos, ids, q0, q1, q2, start, count); return; }
  PIES_LAUNCH(k_tet<0>, kProjBlock, count, st, pos, ids, q0, q1, q2, start, count);
}
void launch_bend(hipStream_t st, float4* pos, const uint4* ids, const float2* angle_w, uint32_t start, uint32_t count) {
  if (count == 0) return;
  PIES_LAUNCH(k_bend, kProjBlock, count, st, pos, ids, angle_w, start, count);
}

}  // namespace pies
