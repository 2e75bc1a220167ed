// Projective-Dynamics substep kernels for gfx950 (Src/Solver.cpp:162-486 of the reference).
//
//   predict      pos += h v ; Msn_h2 = pos/invMass/h^2 ; floor-contact detection ; system diagonal
//   local step   the strain + volume element pairs: tile-resident (k_pd_local_tiles, round 4) - one wavefront per tile of 128
//                pairs, two elements per lane in packed fp32 (pd_local_packed.h), contributions parked in LDS and added up node
//                by node: one 12-byte sum per (tile, node) leaves the chip.  Every other container (and scenes without pairs):
//                one lane = one constraint, the whole container in ONE launch -> w*(A^T B p)_i per (constraint, node)
//                (Solver.cpp:270-308).  The contacts' local step rides in a few extra workgroups of the element launch.
//   rhs          Msn_h2 + the node's records (tile sums, per-constraint records, contact / goal / floor terms; Solver.cpp:310-349),
//                gathered without atomics: by one lane inside the residual kernel of the one-launch CG when a node has a few tile
//                sums (pd_cg1_kernels.hip), by four lanes per node in k_pd_rhs otherwise - deterministic, not the reference's term order
//   global step  Jacobi-preconditioned CG on (K + C) x = rhs for the 3 coordinate columns at once (the reference factors K + C
//                with a sparse Cholesky every substep, Solver.cpp:258-262,356): pd_cg1_kernels.hip (one launch per iteration),
//                pd_cg_kernels.hip (two launches; the experiments with several lanes per row), helpers in pd_cg_device.h
//   velocity     v = (1-d)(pos-prev)/h + h f/m ; prev = pos ; floor friction
//
// Everything here is bandwidth/latency bound (gathers, streams, SpMV at ~15 nnz/row, 3x3 algebra): no MFMA.
#include <cstdint>

#include <cstdlib>

#include "dev_math.h"
#include "pd_kernels.h"
#include "pd_cg_device.h"
#include "pd_rhs_device.h"

namespace pies {


static inline dim3 grid_for(uint32_t n) { return dim3((n + kBlock - 1) / kBlock); }

// ------------------------------------------------------------------------------------------------------
// Solver.cpp:229-238 + floor part of the detection (:829-834, one contact per (triangle,node) incidence)
// + diagonal of the collision matrix (:254-259) and the Jacobi preconditioner.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_pd_predict(float4* __restrict__ pos, const float4* __restrict__ vel,
                                                       float4* __restrict__ msn, const uint32_t* __restrict__ triCount,
                                                       uint32_t* __restrict__ nstatic, const float* __restrict__ kdiag,
                                                       float* __restrict__ cdiag, float* __restrict__ dinv, uint32_t n, float h,
                                                       float h2, float contactHeight, TriArrays tri) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (tri.nt) tri_reset(tri, i, gridDim.x * kBlock);  // the point-triangle pipeline's reset rides along (tri_kernels.h)
  if (i >= n) return;
  float4 p = pos[i];
  const float4 v = vel[i];
  p.x = p.x + h * v.x;
  p.y = p.y + h * v.y;
  p.z = p.z + h * v.z;
  pos[i] = p;
  msn[i] = make_float4((p.x / p.w) / h2, (p.y / p.w) / h2, (p.z / p.w) / h2, 0.0f);
  const uint32_t ns = (p.y < contactHeight) ? triCount[i] : 0u;
  nstatic[i] = ns;
  float cd = 0.0f;
  for (uint32_t k = 0; k < ns; ++k) cd += kStaticW;  // coeffRef(n,n) += w once per contact
  cdiag[i] = cd;
  dinv[i] = 1.0f / (kdiag[i] + cd);
}

// ------------------------------------------------------------------------------------------------------
// Local steps.  contrib[slotBase + i*count + c] = w * (A^T B p)_i   (Constraints.h:89-105); one plane of
// `count` records per local node index i, so that every store instruction writes 16 contiguous bytes per lane
// ------------------------------------------------------------------------------------------------------
// DistanceConstraint: A = B = [[.5,-.5],[-.5,.5]]  =>  A^T B = [[.5,-.5],[-.5,.5]] exactly.
__global__ void __launch_bounds__(kBlock) k_pd_local_distance(const float4* __restrict__ pos, const uint2* __restrict__ ids,
                                                              const float2* __restrict__ rest_w, Vec3f* __restrict__ contrib,
                                                              uint32_t count) {
  const uint32_t c = xcd_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
  if (c >= count) return;
  const uint2 id = ids[c];
  const float2 rw = rest_w[c];
  const float4 a = pos[id.x];
  const float4 b = pos[id.y];
  const float dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
  const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
  float ux = 1.0f, uy = 0.0f, uz = 0.0f;
  if (dist > 0.00001f) {
    ux = dx / dist;
    uy = dy / dist;
    uz = dz / dist;
  }
  const float nd = -(rw.x - dist);
  const float p0x = a.x + nd * ux, p0y = a.y + nd * uy, p0z = a.z + nd * uz;  // projected[0]; projected[1] = b
  const float w = rw.y;
  // (AtB p)_0 = .5 p0 + (-.5) p1 ; (AtB p)_1 = (-.5) p0 + .5 p1 ; accumulated from 0 like the reference's product
  contrib[c] = Vec3f{w * ((0.0f + 0.5f * p0x) + -0.5f * b.x), w * ((0.0f + 0.5f * p0y) + -0.5f * b.y),
                     w * ((0.0f + 0.5f * p0z) + -0.5f * b.z)};
  contrib[count + c] = Vec3f{w * ((0.0f + -0.5f * p0x) + 0.5f * b.x), w * ((0.0f + -0.5f * p0y) + 0.5f * b.y),
                             w * ((0.0f + -0.5f * p0z) + 0.5f * b.z)};
}

// CollisionConstraint (node-node; CollisionConstraint.cpp:10-41 and :49-65), the extension container of
// pies_add_node_pair_constraints: both nodes' projected positions, weighted (A = B = I: contribution_i = w * projected_i).
__global__ void __launch_bounds__(kBlock) k_pd_local_node_pair(const float4* __restrict__ pos, const float* __restrict__ radius,
                                                               const uint2* __restrict__ ids, Vec3f* __restrict__ contrib, uint32_t count) {
  const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= count) return;
  const uint2 id = ids[c];
  const float4 a = pos[id.x], b = pos[id.y];  // (.w = invMass)
  float pax = a.x, pay = a.y, paz = a.z, pbx = b.x, pby = b.y, pbz = b.z;
  const float dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
  const float distSq = dx * dx + dy * dy + dz * dz;
  const float r = radius[id.x] + radius[id.y];
  if (distSq < r * r) {
    const float dist = sqrtf(distSq);
    const float dispLength = r - dist;
    float ex, ey, ez;
    if (dist > 0.00001f) { ex = dispLength * dx / dist; ey = dispLength * dy / dist; ez = dispLength * dz / dist; }
    else { ex = dispLength; ey = 0.0f; ez = 0.0f; }
    const float wSum = a.w + b.w;
    pax -= ex * a.w / wSum; pay -= ey * a.w / wSum; paz -= ez * a.w / wSum;
    pbx += ex * b.w / wSum; pby += ey * b.w / wSum; pbz += ez * b.w / wSum;
  }
  contrib[c] = Vec3f{kNodePairW * pax, kNodePairW * pay, kNodePairW * paz};
  contrib[count + c] = Vec3f{kNodePairW * pbx, kNodePairW * pby, kNodePairW * pbz};
}
// The friction loop over the node-node constraints (Solver.cpp:398-428), in list order: pairs may share nodes, so the list is walked
// by one lane (an extension container a host fills by hand; the live collision constraints have their own level-ordered passes).
__global__ void k_pd_node_pair_friction(const float4* __restrict__ pos, float4* __restrict__ vel, const float* __restrict__ radius,
                                        const uint2* __restrict__ ids, uint32_t count, float frictionOpt, float staticThreshold) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (uint32_t c = 0; c < count; ++c) {
    const uint2 id = ids[c];
    const float4 a = pos[id.x], b = pos[id.y];
    const float dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    if (dist > radius[id.x] + radius[id.y]) continue;
    const float nx = dx / dist, ny = dy / dist, nz = dz / dist;
    float4 va = vel[id.x], vb = vel[id.y];
    const float rx = vb.x - va.x, ry = vb.y - va.y, rz = vb.z - va.z;
    const float rn = rx * nx + ry * ny + rz * nz;
    const float px = rx - rn * nx, py = ry - rn * ny, pz = rz - rn * nz;
    float friction = -frictionOpt;
    if (sqrtf(px * px + py * py + pz * pz) < staticThreshold) friction = 1.0f;
    const float wSum = a.w + b.w;
    va.x += -friction * px * a.w / wSum; va.y += -friction * py * a.w / wSum; va.z += -friction * pz * a.w / wSum;
    vb.x += friction * px * b.w / wSum; vb.y += friction * py * b.w / wSum; vb.z += friction * pz * b.w / wSum;
    vel[id.x] = va;
    vel[id.y] = vb;
  }
}

// Constraints.cpp:186-203.  The reference divides the three components by `den`; here one reciprocal (a correctly rounded
// division) and three products - the last bit of D may differ, PD parity is by tolerance, and the
// ten iterations of this loop were a quarter of the local step's instructions with three divisions each.
PIES_DEV void compute_d(const float s[3], float omegaMin, float omegaMax, float D[3]) {
  D[0] = D[1] = D[2] = 0.0f;
  for (int it = 0; it < 10; ++it) {
    const float sx = s[0] + D[0], sy = s[1] + D[1], sz = s[2] + D[2];
    const float product = sx * sy * sz;
    const float omega = clampf(product, omegaMin, omegaMax);
    const float C = product - omega;
    const float gx = sy * sz, gy = sx * sz, gz = sx * sy;
    const float num = (gx * D[0] + gy * D[1] + gz * D[2]) - C;
    const float den = gx * gx + gy * gy + gz * gz;
    const float q = num / den;
    D[0] = q * gx;
    D[1] = q * gy;
    D[2] = q * gz;
  }
}

// TetrahedralConstraint (VOLUME = false, Constraints.cpp:76-128) and VolumeConstraint (VOLUME = true,
// :205-255).  B = I and A = [0 ; Qinv_(r,k) D] (Constraints.cpp:141-175), so
// (A^T p)_0 = -(q_r0+q_r1+q_r2) weighted sum, (A^T p)_{1+c} = sum_r Qinv[r][c] p_{1+r}; p_0 = 0.
struct TetFrame {
  float qi[3][3];  // Qinv, [col][row]
  float F[3][3];
  Svd3 d;
};
PIES_DEV void tet_frame(const float4* __restrict__ pos, const uint4 id, const float4 a0, const float4 a1, const float4 a2, TetFrame& t) {
  const float4 x1 = pos[id.x], x2 = pos[id.y], x3 = pos[id.z], x4 = pos[id.w];
  const float qi[3][3] = {{a0.x, a0.y, a0.z}, {a0.w, a1.x, a1.y}, {a1.z, a1.w, a2.x}};
  const float P[3][3] = {{x2.x - x1.x, x2.y - x1.y, x2.z - x1.z},
                         {x3.x - x1.x, x3.y - x1.y, x3.z - x1.z},
                         {x4.x - x1.x, x4.y - x1.y, x4.z - x1.z}};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) t.qi[i][j] = qi[i][j];
  mat3_mul_cm(P, t.qi, t.F);
  svd3(t.F, t.d);
}
// singular values of the projection: VolumeConstraint (:205-255) or TetrahedralConstraint (:76-128)
template <bool VOLUME> PIES_DEV void tet_project(const TetFrame& t, float lo, float hi, float s[3]) {
  if (VOLUME) {
    float D[3];
    compute_d(t.d.s, lo, hi, D);
    s[0] = t.d.s[0] + D[0];
    s[1] = t.d.s[1] + D[1];
    s[2] = t.d.s[2] + D[2];
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) s[i] = clampf(t.d.s[i], lo, hi);
    if (det3_cm(t.F) < 0.0f) {
      int k = 0;
      float m = t.d.s[0];
      if (t.d.s[1] <= m) { k = 1; m = t.d.s[1]; }
      if (t.d.s[2] <= m) { k = 2; }
      s[0] = (k == 0) ? -s[0] : s[0];
      s[1] = (k == 1) ? -s[1] : s[1];
      s[2] = (k == 2) ? -s[2] : s[2];
    }
  }
}
// contribution_i = w * (A^T p)_i with p = (0, Fh[0], Fh[1], Fh[2]), Fh = U diag(s) V^T
PIES_DEV void tet_records(const TetFrame& t, const float s[3], float w, float rec[4][3]) {
  float Fh[3][3];
  svd3_recompose(t.d, s, Fh);
  // A[1+r][0] = ((0 + -q_r0) + -q_r1) + -q_r2 ; A[1+r][1+c] = q_rc with q_rc = Qinv[r][c] (reference's row-major read)
  float A0[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) A0[r] = ((0.0f + t.qi[r][0] * -1.0f) + t.qi[r][1] * -1.0f) + t.qi[r][2] * -1.0f;
  float out[4][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    // sum over rows m = 0..3 of A[m][i] * p_m, starting from 0; row 0 of A and p_0 are zero
    out[0][k] = ((0.0f + A0[0] * Fh[0][k]) + A0[1] * Fh[1][k]) + A0[2] * Fh[2][k];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) out[1 + cc][k] = ((0.0f + t.qi[0][cc] * Fh[0][k]) + t.qi[1][cc] * Fh[1][k]) + t.qi[2][cc] * Fh[2][k];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) rec[i][k] = w * out[i][k];
}
PIES_DEV void tet_emit(const TetFrame& t, const float s[3], float w, Vec3f* __restrict__ contrib, uint32_t count, uint32_t c) {
  float rec[4][3];
  tet_records(t, s, w, rec);
#pragma unroll
  for (int i = 0; i < 4; ++i) contrib[i * count + c] = Vec3f{rec[i][0], rec[i][1], rec[i][2]};
}

template <bool VOLUME>
__global__ void __launch_bounds__(kBlock) k_pd_local_tet(const float4* __restrict__ pos, const uint4* __restrict__ ids,
                                                         const float4* __restrict__ q0, const float4* __restrict__ q1,
                                                         const float4* __restrict__ q2, Vec3f* __restrict__ contrib,
                                                         uint32_t count) {
  const uint32_t c = xcd_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
  if (c >= count) return;
  const float4 a2 = q2[c];
  TetFrame t;
  tet_frame(pos, ids[c], q0[c], q1[c], a2, t);
  float s[3];
  tet_project<VOLUME>(t, a2.y, a2.z, s);
  tet_emit(t, s, a2.w, contrib, count, c);
}

// A tetrahedral-strain and a volume constraint over the same element (createTetBox adds them in pairs,
// PrimitiveUtilities.cpp:401-514; pd_setup.cpp checks ids and Qinv are identical): one gather, one SVD, both
// projections - the arithmetic of each is exactly that of its own kernel above.
PIES_DEV void local_tet_pair(const float4* __restrict__ pos, const uint4* __restrict__ ids, const float4* __restrict__ q0,
                             const float4* __restrict__ q1, const float4* __restrict__ q2, const float4* __restrict__ vq2,
                             Vec3f* __restrict__ contribTet, uint32_t count, uint32_t tetBlocks) {
  const uint32_t c = xcd_block(blockIdx.x, tetBlocks) * kBlock + threadIdx.x;  // (the launch may carry extra workgroups behind these)
  if (c >= count) return;
  const float4 a2 = q2[c], v2 = vq2[c];
  TetFrame t;
  tet_frame(pos, ids[c], q0[c], q1[c], a2, t);
  // One record per corner: the strain and the volume contribution of the element added up (the right-hand side gathers 4
  // records per element pair instead of 8; the volume container's slots stay unused).  Both projections share U and V, and
  // w (A^T p) is linear in the projected singular values: the two are combined BEFORE the recomposition,
  // w_a U diag(s_a) V^T + w_b U diag(s_b) V^T = U diag(w_a s_a + w_b s_b) V^T - one recomposition and one A^T product per
  // element instead of two (this kernel is bound by VALU issue at 100k particles: 2 230 instructions per lane, 23 us).
  float sa[3], sb[3];
  tet_project<false>(t, a2.y, a2.z, sa);
  tet_project<true>(t, v2.y, v2.z, sb);
  const float sc[3] = {fmaf(a2.w, sa[0], v2.w * sb[0]), fmaf(a2.w, sa[1], v2.w * sb[1]), fmaf(a2.w, sa[2], v2.w * sb[2])};
  float rec[4][3];
  tet_records(t, sc, 1.0f, rec);
#pragma unroll
  for (int i = 0; i < 4; ++i) contribTet[i * count + c] = Vec3f{rec[i][0], rec[i][1], rec[i][2]};
}
// (Asking for 5 wavefronts per SIMD - 77 VGPRs instead of 108, no spill - measured the same on one box: 26.0 / 186 us
// against 26.1 / 192 us at 100k / 5.8M element pairs.)
// Local step of the point-triangle contacts (CollisionConstraint.cpp:86-124) and w * (AtA p)_i (:176-194): the body of
// tri_kernels.hip's k_pd_local_tri (same IEEE sequence), run by a few extra workgroups of the fused strain + volume launch
// so that a substep without contacts does not pay a launch boundary per local/global iteration for it.
template <uint32_t BLOCK> PIES_DEV void local_tri_contacts(const TriArrays& T, const float4* __restrict__ pos, float thickness, uint32_t block, uint32_t nblocks) {
  const uint32_t M = T.counters[2];
  for (uint32_t c = block * BLOCK + threadIdx.x; c < M; c += nblocks * BLOCK) {
    const uint4 id = T.ids[c];
    const float4 q[4] = {pos[id.x], pos[id.y], pos[id.z], pos[id.w]};
    float p[4][3] = {{q[0].x, q[0].y, q[0].z}, {q[1].x, q[1].y, q[1].z}, {q[2].x, q[2].y, q[2].z}, {q[3].x, q[3].y, q[3].z}};
    const float rel[3] = {p[0][0] - p[1][0], p[0][1] - p[1][1], p[0][2] - p[1][2]};
    const float a[3] = {p[2][0] - p[1][0], p[2][1] - p[1][1], p[2][2] - p[1][2]};
    const float b[3] = {p[3][0] - p[1][0], p[3][1] - p[1][1], p[3][2] - p[1][2]};
    const float cr[3] = {a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]};
    const float inv = 1.0f / sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
    const float n[3] = {cr[0] * inv, cr[1] * inv, cr[2] * inv};
    const float nDotP = n[0] * rel[0] + n[1] * rel[1] + n[2] * rel[2];
    if (nDotP < thickness) {
      const float d = thickness - nDotP;
      p[0][0] = p[0][0] + d * n[0];
      p[0][1] = p[0][1] + d * n[1];
      p[0][2] = p[0][2] + d * n[2];
    }
    // AtA = [[3,-1,-1,-1],[-1,1,0,0],[-1,0,1,0],[-1,0,0,1]], products accumulated from 0 in column order
    const float AtA[4][4] = {{3.f, -1.f, -1.f, -1.f}, {-1.f, 1.f, 0.f, 0.f}, {-1.f, 0.f, 1.f, 0.f}, {-1.f, 0.f, 0.f, 1.f}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ax += AtA[i][k] * p[k][0];
        ay += AtA[i][k] * p[k][1];
        az += AtA[i][k] * p[k][2];
      }
      T.contrib[4 * c + i] = make_float4(kTriContactW * ax, kTriContactW * ay, kTriContactW * az, 0.f);
    }
  }
}
constexpr uint32_t kTriLocalBlocks = 64;  // extra workgroups of the fused launch that sweep the contact list
}  // namespace pies
#include "pd_local_packed.h"
namespace pies {
// PACKED: two elements per lane in packed fp32 (pd_local_packed.h; the default), tetBlocks workgroups cover ceil(count / 2)
// lanes; DICT: the elements' constants come from the rest dictionary
template <bool PACKED, bool DICT>
__global__ void __launch_bounds__(kBlock) k_pd_local_tet_pair(const float4* __restrict__ pos, const uint4* __restrict__ ids,
                                                              const float4* __restrict__ q0, const float4* __restrict__ q1,
                                                              const float4* __restrict__ q2, const float4* __restrict__ vq2,
                                                              RestDictionary dict, Vec3f* __restrict__ contribTet, uint32_t count, TriArrays T,
                                                              float thickness, uint32_t tetBlocks) {
  if (blockIdx.x >= tetBlocks) {  // uniform per workgroup
    local_tri_contacts<kBlock>(T, pos, thickness, blockIdx.x - tetBlocks, kTriLocalBlocks);
    return;
  }
  if (PACKED) {
    const uint32_t c0 = 2u * (xcd_block(blockIdx.x, tetBlocks) * kBlock + threadIdx.x);
    if (c0 >= count) return;
    local_tet_pair_packed<DICT>(pos, ids, q0, q1, q2, vq2, dict, contribTet, count, c0, c0 + 1u < count ? c0 + 1u : c0);
  } else {
    local_tet_pair(pos, ids, q0, q1, q2, vq2, contribTet, count, tetBlocks);
  }
}

// The strain + volume local step over tiles (PdTileArrays, pd_kernels.h): one wavefront = one workgroup = one tile of up to 128
// element pairs on up to 128 nodes.  Node positions are staged in LDS, the elements are projected two per lane
// (pair_project_packed), every element's four contributions are parked in LDS, and lane k adds up the records of tile nodes k
// and k + 64 in list order.  A single wavefront's LDS operations execute in order, so the barrier below is only the
// compiler's and the hardware's s_waitcnt; wavefronts never wait for each other, and the chip balances 4 213 tiles of config 3
// like it balanced the 4 213 wavefronts of k_pd_local_tet_pair.
// (Round 5: config 3's 4 470 tiles are 1.09 rounds of the 4 096 wavefronts the chip holds at 114 registers and 9 KB of LDS each.  Five
// wavefronts per SIMD - the staged positions sharing their LDS with the parked contributions, 7 KB, and __launch_bounds__(64, 5): 96
// registers, 27 spilled words - put every tile in one round and were slower: 22.1 against 16.5 us per launch, 1 920 against 2 190
// substeps/s.)
constexpr uint32_t kTileLanes = 64;
template <bool DICT>
__global__ void __launch_bounds__(kTileLanes) k_pd_local_tiles(PdTileArrays T, const float4* __restrict__ pos, const float4* __restrict__ dictTable,
                                                               TriArrays tri, float thickness, uint32_t tileBlocks) {
  if (blockIdx.x >= tileBlocks) {  // uniform per workgroup
    local_tri_contacts<kTileLanes>(tri, pos, thickness, blockIdx.x - tileBlocks, gridDim.x - tileBlocks);
    return;
  }
  __shared__ float4 sPos[kTileNodes];
  __shared__ float sX[4 * kTileElems], sY[4 * kTileElems], sZ[4 * kTileElems];  // [corner][element]
  __shared__ uint16_t sInc[4 * kTileElems];
  const uint32_t t = xcd_block(blockIdx.x, tileBlocks), lane = threadIdx.x;
  const uint32_t info = T.info[t];
  const uint32_t nn = info & 0xffffu, cnt = info >> 16;
  // everything the tile needs is requested before anything is waited for
  const uint32_t g0 = T.node[kTileNodes * t + lane], g1 = T.node[kTileNodes * t + 64u + lane];
  const uint2 lab = reinterpret_cast<const uint2*>(T.local + static_cast<size_t>(kTileElems) * t)[lane];  // elements 2 lane, 2 lane + 1
  const uint4 incw = reinterpret_cast<const uint4*>(T.inc + 4ull * kTileElems * t)[lane];                 // 8 list entries
  const uint16_t* np = T.nptr + static_cast<size_t>(kTileNptr) * t;
  const uint32_t b0 = np[lane], e0 = np[lane + 1u], b1 = np[64u + lane], e1 = np[65u + lane];
  float4 a0, a1, a2, av, b0c, b1c, b2c, bvc;
  if (DICT) {
    const uint32_t dd = reinterpret_cast<const uint32_t*>(T.dict + static_cast<size_t>(kTileElems) * t)[lane];
    const uint32_t ka = 4u * (dd & 0xffffu), kb = 4u * (dd >> 16);
    a0 = dictTable[ka]; a1 = dictTable[ka + 1]; a2 = dictTable[ka + 2]; av = dictTable[ka + 3];
    b0c = dictTable[kb]; b1c = dictTable[kb + 1]; b2c = dictTable[kb + 2]; bvc = dictTable[kb + 3];
  } else {
    const size_t ea = static_cast<size_t>(kTileElems) * t + 2u * lane, eb = ea + 1u;
    a0 = T.q0[ea]; a1 = T.q1[ea]; a2 = T.q2[ea]; av = T.vq2[ea];
    b0c = T.q0[eb]; b1c = T.q1[eb]; b2c = T.q2[eb]; bvc = T.vq2[eb];
  }
  sPos[lane] = pos[g0];
  sPos[64u + lane] = pos[g1];
  reinterpret_cast<uint4*>(sInc)[lane] = incw;
  __syncthreads();
  const uint32_t na[4] = {lab.x & 0xffu, (lab.x >> 8) & 0xffu, (lab.x >> 16) & 0xffu, lab.x >> 24};
  const uint32_t nb[4] = {lab.y & 0xffu, (lab.y >> 8) & 0xffu, (lab.y >> 16) & 0xffu, lab.y >> 24};
  // (bounds build: an element's local node indices lie inside the tile's node list)
  const bool inb = PIES_IN_BOUNDS(2u * lane >= cnt || max(max(na[0], na[1]), max(na[2], na[3])) < max(nn, 1u), 21u) &&
                   PIES_IN_BOUNDS(2u * lane + 1u >= cnt || max(max(nb[0], nb[1]), max(nb[2], nb[3])) < max(nn, 1u), 22u);
  (void)inb;
  const float4 xa[4] = {sPos[PIES_CLAMP_INDEX(na[0], kTileNodes)], sPos[PIES_CLAMP_INDEX(na[1], kTileNodes)],
                        sPos[PIES_CLAMP_INDEX(na[2], kTileNodes)], sPos[PIES_CLAMP_INDEX(na[3], kTileNodes)]};
  const float4 xb[4] = {sPos[PIES_CLAMP_INDEX(nb[0], kTileNodes)], sPos[PIES_CLAMP_INDEX(nb[1], kTileNodes)],
                        sPos[PIES_CLAMP_INDEX(nb[2], kTileNodes)], sPos[PIES_CLAMP_INDEX(nb[3], kTileNodes)]};
  f2 rec[4][3];
  pair_project_packed(xa, xb, a0, a1, a2, av, b0c, b1c, b2c, bvc, rec);
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // [corner][element]: a lane's two elements are neighbours, 8 bytes per store
    reinterpret_cast<f2*>(sX + kTileElems * i)[lane] = rec[i][0];
    reinterpret_cast<f2*>(sY + kTileElems * i)[lane] = rec[i][1];
    reinterpret_cast<f2*>(sZ + kTileElems * i)[lane] = rec[i][2];
  }
  __syncthreads();
  // node pass: list entries are (element << 2 | corner); slots of elements past the tile's count are in no list
  (void)cnt;
  auto sum = [&](uint32_t b, uint32_t e, float& ax, float& ay, float& az) {
    ax = 0.f; ay = 0.f; az = 0.f;
    for (uint32_t r = b; r < e; ++r) {
      const uint32_t v = sInc[r];
      const uint32_t at = kTileElems * (v & 3u) + (v >> 2);
      if (!PIES_IN_BOUNDS(r < 4u * kTileElems && (v >> 2) < kTileElems, 23u)) continue;
      ax += sX[at]; ay += sY[at]; az += sZ[at];
    }
  };
  float ax, ay, az;
  if (lane < nn) {
    sum(b0, e0, ax, ay, az);
    T.partial[static_cast<size_t>(kTileNodes) * t + lane] = Vec3f{ax, ay, az};
  }
  if (64u + lane < nn) {
    sum(b1, e1, ax, ay, az);
    T.partial[static_cast<size_t>(kTileNodes) * t + 64u + lane] = Vec3f{ax, ay, az};
  }
}

// BendConstraint in PD (Constraints.cpp:312-366): A = B = I, contribution = w * projected_i.
struct V3 {
  float x, y, z;
};
PIES_DEV V3 sub4(const float4& a, const float4& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PIES_DEV V3 crossv(const V3& a, const V3& b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
PIES_DEV float dotv(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PIES_DEV V3 addv(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PIES_DEV V3 subv(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PIES_DEV V3 mulv(const V3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PIES_DEV V3 divv(const V3& a, float s) { return {a.x / s, a.y / s, a.z / s}; }
PIES_DEV V3 negv(const V3& a) { return {-a.x, -a.y, -a.z}; }

__global__ void __launch_bounds__(kBlock) k_pd_local_bend(const float4* __restrict__ pos, const uint4* __restrict__ ids,
                                                          const float2* __restrict__ angle_w, Vec3f* __restrict__ contrib,
                                                          uint32_t count) {
  const uint32_t c = xcd_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
  if (c >= count) return;
  const uint4 id = ids[c];
  const float2 aw = angle_w[c];
  const float4 x1 = pos[id.x], x2 = pos[id.y], x3 = pos[id.z], x4 = pos[id.w];
  const V3 p2 = sub4(x2, x1), p3 = sub4(x3, x1), p4 = sub4(x4, x1);
  const V3 c23 = crossv(p2, p3), c24 = crossv(p2, p4);
  const float l23 = sqrtf(dotv(c23, c23)), l24 = sqrtf(dotv(c24, c24));
  const V3 n1 = divv(c23, l23), n2 = divv(c24, l24);
  const float d = dotv(n1, n2);
  const float C = static_cast<float>(acos(static_cast<double>(d))) - aw.x;
  const V3 q3 = divv(addv(crossv(p2, n2), mulv(crossv(n1, p2), d)), l23);
  const V3 q4 = divv(addv(crossv(p2, n1), mulv(crossv(n2, p2), d)), l24);
  const V3 q2 = subv(negv(divv(addv(crossv(p3, n2), mulv(crossv(n1, p3), d)), l23)), divv(addv(crossv(p4, n1), mulv(crossv(n2, p4), d)), l24));
  const V3 q1 = subv(subv(negv(q2), q3), q4);
  const float wSum = x1.w + x2.w + x3.w + x4.w;
  const float qSq = dotv(q1, q1) + dotv(q2, q2) + dotv(q3, q3) + dotv(q4, q4);
  const float num = sqrtf(fmaxf(1.0f - d * d, 0.0f)) * C;
  V3 pr[4] = {{x1.x, x1.y, x1.z}, {x2.x, x2.y, x2.z}, {x3.x, x3.y, x3.z}, {x4.x, x4.y, x4.z}};
  if (!(qSq < 0.00001f)) {
    pr[0] = addv(pr[0], divv(mulv(mulv(negv(q1), 4 * x1.w / wSum), num), qSq));
    pr[1] = addv(pr[1], divv(mulv(mulv(negv(q2), 4 * x2.w / wSum), num), qSq));
    pr[2] = addv(pr[2], divv(mulv(mulv(negv(q3), 4 * x3.w / wSum), num), qSq));
    pr[3] = addv(pr[3], divv(mulv(mulv(negv(q4), 4 * x4.w / wSum), num), qSq));
  }
  const float w = aw.y;
#pragma unroll
  for (int i = 0; i < 4; ++i) contrib[i * count + c] = Vec3f{w * pr[i].x, w * pr[i].y, w * pr[i].z};
}

// ShapeMatchingConstraint::projectToAuxiliaryVariable (ShapeMatchingConstraint.cpp:96-122), one workgroup
// per constraint: float centroid, P = sum (x - c) r^T / invMass in fp64, F = P Qinv, warm-started
// rotation extraction (:75-94; the 1e-9 is added to the reciprocal, as in the reference), projected = R r + c.
// The centroid and P are block reductions, so their rounding differs from the reference's sequential sums
// at the 1e-7 / 1e-16 level (PD parity is a tolerance anyway).
__global__ void __launch_bounds__(kBlock) k_pd_local_shape(const float4* __restrict__ pos, ShapeArrays S, double4* __restrict__ contribD) {
  __shared__ double red[kBlock / 64][12];
  __shared__ double sh[16];  // centroid (3), R (9)
  const uint32_t k = blockIdx.x;
  const uint32_t e0 = S.off[k], n = S.off[k + 1] - e0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // centroid: weight * position accumulated in float per lane, then reduced
  const float weight = 1.0f / static_cast<float>(n);
  float cx = 0.f, cy = 0.f, cz = 0.f;
  for (uint32_t e = threadIdx.x; e < n; e += kBlock) {
    const float4 p = pos[S.node[e0 + e]];
    cx += weight * p.x; cy += weight * p.y; cz += weight * p.z;
  }
  double acc[12] = {cx, cy, cz, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] += __shfl_xor(acc[q], off, 64);
  if (lane == 0) { red[wave][0] = acc[0]; red[wave][1] = acc[1]; red[wave][2] = acc[2]; }
  __syncthreads();
  if (threadIdx.x < 3) sh[threadIdx.x] = static_cast<double>(static_cast<float>(((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]));
  __syncthreads();
  const float comx = static_cast<float>(sh[0]), comy = static_cast<float>(sh[1]), comz = static_cast<float>(sh[2]);
  // P
  double P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t e = threadIdx.x; e < n; e += kBlock) {
    const float4 p = pos[S.node[e0 + e]];
    const double l[3] = {static_cast<double>(p.x - comx), static_cast<double>(p.y - comy), static_cast<double>(p.z - comz)};
    const double im = static_cast<double>(p.w);
    const double* m = S.mat + 3 * static_cast<size_t>(e0 + e);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) P[3 * r + c] += l[r] * m[c] / im;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
    for (int q = 0; q < 9; ++q) P[q] += __shfl_xor(P[q], off, 64);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int q = 0; q < 9; ++q) red[wave][q] = P[q];
  __syncthreads();
  if (threadIdx.x == 0) {
    double Pm[9], F[9];
    for (int q = 0; q < 9; ++q) Pm[q] = ((red[0][q] + red[1][q]) + red[2][q]) + red[3][q];
    const double* Qi = S.qinv + 9 * static_cast<size_t>(k);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) F[3 * r + c] = Pm[3 * r] * Qi[c] + Pm[3 * r + 1] * Qi[3 + c] + Pm[3 * r + 2] * Qi[6 + c];
    double* qp = S.quat + 4 * static_cast<size_t>(k);
    double qw = qp[0], qx = qp[1], qy = qp[2], qz = qp[3];
    double R[9];
    auto qmat = [&]() {
      const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
      const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
      R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
      R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
      R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
    };
    for (int iter = 0; iter < 100; ++iter) {
      qmat();
      double om[3] = {0, 0, 0}, dsum = 0;
      for (int c = 0; c < 3; ++c) {
        const double r0 = R[c], r1 = R[3 + c], r2 = R[6 + c], a0 = F[c], a1 = F[3 + c], a2 = F[6 + c];
        om[0] += r1 * a2 - r2 * a1;
        om[1] += r2 * a0 - r0 * a2;
        om[2] += r0 * a1 - r1 * a0;
        dsum += r0 * a0 + r1 * a1 + r2 * a2;
      }
      const double f = 1.0 / fabs(dsum) + 1.0e-9;
      om[0] *= f; om[1] *= f; om[2] *= f;
      const double wn = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
      if (wn < 1.0e-9) break;
      const double ha = 0.5 * wn, shf = sin(ha), ch = cos(ha);
      const double ax = (1.0 / wn) * om[0], ay = (1.0 / wn) * om[1], az = (1.0 / wn) * om[2];
      const double dw = ch, dx = shf * ax, dy = shf * ay, dz = shf * az;
      const double nw = dw * qw - dx * qx - dy * qy - dz * qz, nx = dw * qx + dx * qw + dy * qz - dz * qy;
      const double ny = dw * qy + dy * qw + dz * qx - dx * qz, nz = dw * qz + dz * qw + dx * qy - dy * qx;
      const double nn = sqrt(nw * nw + nx * nx + ny * ny + nz * nz);
      qw = nw / nn; qx = nx / nn; qy = ny / nn; qz = nz / nn;
    }
    qp[0] = qw; qp[1] = qx; qp[2] = qy; qp[3] = qz;
    qmat();
    for (int q = 0; q < 9; ++q) sh[3 + q] = R[q];
  }
  __syncthreads();
  const double wk = static_cast<double>(S.w[k]);
  const double T0 = static_cast<double>(comx), T1 = static_cast<double>(comy), T2 = static_cast<double>(comz);
  for (uint32_t e = threadIdx.x; e < n; e += kBlock) {
    const double* m = S.mat + 3 * static_cast<size_t>(e0 + e);
    contribD[e0 + e] = make_double4((sh[3] * m[0] + sh[4] * m[1] + sh[5] * m[2]) + T0, (sh[6] * m[0] + sh[7] * m[1] + sh[8] * m[2]) + T1,
                                    (sh[9] * m[0] + sh[10] * m[1] + sh[11] * m[2]) + T2, wk);
  }
}

// ------------------------------------------------------------------------------------------------------
// Right-hand side (Solver.cpp:266, 310-349): one lane per node, contributions in the reference's order
// (position, distance, tet, volume, bend, ... then the floor contacts), so the float sum is the reference's.
// Also evaluates the floor projection (CollisionConstraint.cpp:447-455: clamps to y >= 0, not floorHeight).
// ------------------------------------------------------------------------------------------------------
template <uint32_t LANES>
__global__ void __launch_bounds__(kBlock) k_pd_rhs(RhsArrays R, float4* __restrict__ rhs) {
  const uint32_t i = (xcd_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x) / LANES, sub = threadIdx.x & (LANES - 1);
  const float4 f = rhs_of_node<LANES>(R, i, sub, i < R.n);
  if (i < R.n && sub == 0) rhs[i] = f;
}

// ------------------------------------------------------------------------------------------------------
// Solver.cpp:367-383 (floor snap; tri/edge stabilisation is a later row) -- idempotent, applied once
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_pd_stabilize(float4* __restrict__ pos, const float4* __restrict__ statp,
                                                         const uint32_t* __restrict__ nstatic, uint32_t n, CgArrays A, int closeSolve) {
  if (closeSolve && blockIdx.x + 1u == gridDim.x) {  // one workgroup behind the others: the statistics of the substep's last solve
    solve_statistics(A, A.partB);
    return;
  }
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (nstatic[i]) {
    const float4 s = statp[i];
    float4 p = pos[i];
    p.x = s.x;
    p.y = s.y;
    p.z = s.z;
    pos[i] = p;
  }
}

// Solver.cpp:386-395 + floor friction :473-484 (once per contact instance)
__global__ void __launch_bounds__(kBlock) k_pd_velocity(const float4* __restrict__ pos, float4* __restrict__ prev,
                                                        float4* __restrict__ vel, const uint32_t* __restrict__ nstatic, uint32_t n,
                                                        float h, float damping, float gravity, float friction,
                                                        float staticThreshold, bool staticFriction, const uint32_t* __restrict__ usedBits) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  // usedBits: the floor friction of the nodes that are in a point-triangle contact comes after the contacts' friction
  // (launch_tri_friction applies it); every other node gets it here
  if (usedBits) staticFriction = ((usedBits[i >> 5] >> (i & 31u)) & 1u) == 0u;
  const float4 p = pos[i];
  const float4 q = prev[i];
  const float k = 1.0f - damping;
  // node.force = (0,-g,0)/invMass ; v = (1-d)(pos-prev)/h + h*force*invMass
  const float fx = 0.0f / p.w, fy = -gravity / p.w, fz = 0.0f / p.w;
  float vx = (k * (p.x - q.x)) / h + (h * fx) * p.w;
  float vy = (k * (p.y - q.y)) / h + (h * fy) * p.w;
  float vz = (k * (p.z - q.z)) / h + (h * fz) * p.w;
  const uint32_t ns = staticFriction ? nstatic[i] : 0u;
  for (uint32_t c = 0; c < ns; ++c) {
    const float px = vx, pz = vz;  // perpVel = (vx, 0, vz)
    float fr = friction;
    if (sqrtf(px * px + 0.0f * 0.0f + pz * pz) < staticThreshold) fr = 1.0f;
    vx += -fr * px;
    vy += -fr * 0.0f;
    vz += -fr * pz;
  }
  prev[i] = make_float4(p.x, p.y, p.z, 0.f);
  vel[i] = make_float4(vx, vy, vz, 0.f);
}

// ------------------------------------------------------------------------------------------------------
void launch_pd_predict(hipStream_t st, const NodeArrays& nd, const PdArrays& pd, float h, float contactHeight, bool triReset) {
  if (nd.n == 0) return;
  TriArrays tri = pd.tri;
  if (!triReset) tri.nt = 0;
  hipLaunchKernelGGL(k_pd_predict, grid_for(nd.n), dim3(kBlock), 0, st, nd.pos, nd.vel, pd.msn, pd.triCount, pd.nstatic, pd.kdiag,
                     pd.cg.cdiag, pd.cg.dinv, nd.n, h, h * h, contactHeight, tri);
}
void launch_pd_local_node_pair(hipStream_t st, const float4* pos, const float* radius, const uint2* ids, Vec3f* contrib, uint32_t count) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_pd_local_node_pair, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, st, pos, radius, ids, contrib, count);
}
void launch_pd_node_pair_friction(hipStream_t st, const float4* pos, float4* vel, const float* radius, const uint2* ids, uint32_t count,
                                  float friction, float staticThreshold) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_pd_node_pair_friction, dim3(1), dim3(64), 0, st, pos, vel, radius, ids, count, friction, staticThreshold);
}
void launch_pd_local_distance(hipStream_t st, const float4* pos, const uint2* ids, const float2* rw, Vec3f* contrib, uint32_t count) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_pd_local_distance, grid_for(count), dim3(kBlock), 0, st, pos, ids, rw, contrib, count);
}
void launch_pd_local_tet(hipStream_t st, bool volume, const float4* pos, const uint4* ids, const float4* q0, const float4* q1,
                         const float4* q2, Vec3f* contrib, uint32_t count) {
  if (count == 0) return;
  if (volume) hipLaunchKernelGGL(k_pd_local_tet<true>, grid_for(count), dim3(kBlock), 0, st, pos, ids, q0, q1, q2, contrib, count);
  else hipLaunchKernelGGL(k_pd_local_tet<false>, grid_for(count), dim3(kBlock), 0, st, pos, ids, q0, q1, q2, contrib, count);
}
RhsArrays rhs_arrays(const NodeArrays& nd, const PdArrays& pd) {
  return RhsArrays{pd.msn, pd.contrib, pd.incPtr, pd.incSlot, pd.contribD, pd.incPtrD, pd.incSlotD, nd.pos, pd.nstatic, pd.statp,
                   pd.cg.tIncCnt, pd.cg.tIncStart, pd.cg.tInc, pd.tContrib, pd.cg.tUsedCount, nd.n};
}
void launch_pd_rhs(hipStream_t st, const NodeArrays& nd, const PdArrays& pd) {
  if (nd.n == 0) return;
  if (pd.rhsLanes == 1 && !pd.cg.useCAp) hipLaunchKernelGGL(k_pd_rhs<1>, grid_for(nd.n), dim3(kBlock), 0, st, rhs_arrays(nd, pd), pd.rhs);  // (many contacts: four lanes share a node's contact records)
  else hipLaunchKernelGGL(k_pd_rhs<4>, dim3((nd.n + kBlock / 4 - 1) / (kBlock / 4)), dim3(kBlock), 0, st, rhs_arrays(nd, pd), pd.rhs);
}
PIES_BOUNDS_REPORT(pd)
void launch_pd_local_tet_pair(hipStream_t st, const float4* pos, const uint4* ids, const float4* q0, const float4* q1, const float4* q2,
                              const float4* vq2, Vec3f* contribTet, Vec3f* contribVol, uint32_t count, const TriArrays* tri,
                              float thickness, bool packed, const uint16_t* dictIndex, const float4* dictTable) {
  if (count == 0) return;
  (void)contribVol;  // the pair's two contributions are added into the strain constraint's records
  const bool withTri = tri && tri->nt;
  const RestDictionary dict{dictIndex, dictTable};
  const TriArrays T = withTri ? *tri : TriArrays{};
  if (packed) {
    const uint32_t tetBlocks = grid_for((count + 1u) / 2u).x;
    const dim3 grid(tetBlocks + (withTri ? kTriLocalBlocks : 0u));
    if (dictIndex) hipLaunchKernelGGL((k_pd_local_tet_pair<true, true>), grid, dim3(kBlock), 0, st, pos, ids, q0, q1, q2, vq2, dict, contribTet, count, T, thickness, tetBlocks);
    else hipLaunchKernelGGL((k_pd_local_tet_pair<true, false>), grid, dim3(kBlock), 0, st, pos, ids, q0, q1, q2, vq2, dict, contribTet, count, T, thickness, tetBlocks);
  } else {
    const uint32_t tetBlocks = grid_for(count).x;
    hipLaunchKernelGGL((k_pd_local_tet_pair<false, false>), dim3(tetBlocks + (withTri ? kTriLocalBlocks : 0u)), dim3(kBlock), 0, st, pos, ids, q0, q1, q2, vq2,
                       dict, contribTet, count, T, thickness, tetBlocks);
  }
}
void launch_pd_local_tiles(hipStream_t st, const float4* pos, const PdTileArrays& T, const float4* dictTable, const TriArrays* tri, float thickness) {
  if (T.ntiles == 0) return;
  const bool withTri = tri && tri->nt;
  const TriArrays Tr = withTri ? *tri : TriArrays{};
  const dim3 grid(T.ntiles + (withTri ? 4u * kTriLocalBlocks : 0u));  // (the contacts' local step: the same number of lanes as before)
  if (T.dict) hipLaunchKernelGGL(k_pd_local_tiles<true>, grid, dim3(kTileLanes), 0, st, T, pos, dictTable, Tr, thickness, T.ntiles);
  else hipLaunchKernelGGL(k_pd_local_tiles<false>, grid, dim3(kTileLanes), 0, st, T, pos, dictTable, Tr, thickness, T.ntiles);
}
void launch_pd_local_bend(hipStream_t st, const float4* pos, const uint4* ids, const float2* angle_w, Vec3f* contrib, uint32_t count) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_pd_local_bend, grid_for(count), dim3(kBlock), 0, st, pos, ids, angle_w, contrib, count);
}
void launch_pd_local_shape(hipStream_t st, const float4* pos, const PdArrays& pd) {
  if (pd.shape.count == 0) return;
  hipLaunchKernelGGL(k_pd_local_shape, dim3(pd.shape.count), dim3(kBlock), 0, st, pos, pd.shape, pd.contribD);
}
void launch_pd_stabilize(hipStream_t st, const NodeArrays& nd, const PdArrays& pd, bool closeSolve, int maxIters, float tol, bool single) {
  if (nd.n == 0) return;
  CgArrays A = pd.cg;
  A.tol2 = tol * tol;
  A.single = single ? 1 : 0;
  float* pb[2] = {pd.cg.partB, pd.cg.partBnext};
  A.partB = single ? pd.cg.part1[maxIters & 1] : pb[maxIters & 1];  // where the last solve's final residual partials are (launch_pd_solve / launch_pd_solve1)
  hipLaunchKernelGGL(k_pd_stabilize, dim3(grid_for(nd.n).x + (closeSolve ? 1u : 0u)), dim3(kBlock), 0, st, nd.pos, pd.statp, pd.nstatic, nd.n, A,
                     closeSolve ? 1 : 0);
}
void launch_pd_velocity(hipStream_t st, const NodeArrays& nd, const PdArrays& pd, float h, float damping, float gravity,
                        float friction, float staticThreshold, bool staticFriction, const uint32_t* usedBits) {
  if (nd.n == 0) return;
  hipLaunchKernelGGL(k_pd_velocity, grid_for(nd.n), dim3(kBlock), 0, st, nd.pos, nd.prev, nd.vel, pd.nstatic, nd.n, h, damping, gravity,
                     friction, staticThreshold, staticFriction, usedBits);
}

}  // namespace pies