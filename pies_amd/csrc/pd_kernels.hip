// Projective-Dynamics substep kernels for gfx950 (Src/Solver.cpp:162-486 of the reference).
//
//   predict      pos += h v ; Msn_h2 = pos/invMass/h^2 ; floor-contact detection ; system diagonal
//   local step   one lane = one constraint, all constraints of a container in ONE launch (the local step
//                only reads positions, Solver.cpp:270-308) -> writes w*(A^T B p)_i per (constraint, node)
//                (strain + volume constraints over the same elements share one gather and one SVD; the contacts'
//                local step rides in a few extra workgroups of that launch)
//   rhs          four lanes = one node: Msn_h2 + the node's contribution records (Solver.cpp:310-349), gathered without
//                atomics: interleaved partial sums combined pairwise - deterministic, not the reference's term order
//   global step  Jacobi-preconditioned CG on (K + C) x = rhs for the 3 coordinate columns at once; K in sliced ELL
//                form (one row per lane, 64 rows per slice); the reference factors K + C with a sparse Cholesky every
//                substep (Solver.cpp:258-262,356) -- CG to a relative residual replaces the direct solve; the graph holds a
//                budget of iterations per solve, a converged solve's remaining launches return on one flag word, a solve
//                that needs more goes on inside its last launch (k_cg_update, grid barriers), and a substep whose solve still ends above
//                the tolerance is run again by pies_tick (capi.cpp)
//   velocity     v = (1-d)(pos-prev)/h + h f/m ; prev = pos ; floor friction
//
// Everything here is bandwidth/latency bound (gathers, streams, SpMV at ~15 nnz/row, 3x3 algebra): no MFMA.
#include <cstdint>

#include <cstdlib>

#include "dev_math.h"
#include "pd_kernels.h"

namespace pies {

constexpr int kBlock = 256;
constexpr float kStaticW = 10000.0f;  // StaticCollisionConstraint::w (Include/Pies/CollisionConstraint.h:78)

static inline dim3 grid_for(uint32_t n) { return dim3((n + kBlock - 1) / kBlock); }

// ------------------------------------------------------------------------------------------------------
// Solver.cpp:229-238 + floor part of the detection (:829-834, one contact per (triangle,node) incidence)
// + diagonal of the collision matrix (:254-259) and the Jacobi preconditioner.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_pd_predict(float4* __restrict__ pos, const float4* __restrict__ vel,
                                                       float4* __restrict__ msn, const uint32_t* __restrict__ triCount,
                                                       uint32_t* __restrict__ nstatic, const float* __restrict__ kdiag,
                                                       float* __restrict__ cdiag, float* __restrict__ dinv, uint32_t n, float h,
                                                       float h2, float contactHeight) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
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
PIES_DEV void local_tri_contacts(const TriArrays& T, const float4* __restrict__ pos, float thickness, uint32_t block, uint32_t nblocks) {
  const uint32_t M = T.counters[2];
  for (uint32_t c = block * kBlock + threadIdx.x; c < M; c += nblocks * kBlock) {
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
    local_tri_contacts(T, pos, thickness, blockIdx.x - tetBlocks, kTriLocalBlocks);
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
#ifndef PIES_RHS_LANES
#define PIES_RHS_LANES 4
#endif
// lanes that share a node's gather (each with four records in flight).  Measured per launch at 100k / 1M nodes with ~24
// records per node: 16 lanes 13.1 / 136 us, 8 lanes 10.2 / 95, 4 lanes 9.3 / 88, 2 lanes 10.6 / 105.
constexpr uint32_t kRhsLanes = PIES_RHS_LANES;
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
// The right-hand side of node i, by the kRhsLanes lanes that share it (`sub` = the lane's place among them; every lane of the
// wavefront calls these, lanes past the last node with live = false).  The value is complete in the lane with sub == 0.
// kRhsLanes lanes per node: lane `sub` adds up the records sub, sub + kRhsLanes, ... of the node's slot list (four
// slot indices and records per lane in flight at once), then the partial sums are combined pairwise.  The reference adds
// the same terms one after the other; the difference is fp32 rounding of a ~50-term sum (PD parity is by
// tolerance).  (Measured at 100k nodes: 1 lane/node 60 us; 16 lanes with the terms added in
// list order by one lane 24 us; visiting nodes in Morton order was slower than index order.)
// rhs_gather: this lane's share of the records [b, e) of the slot list, added to (ax, ay, az)
PIES_DEV void rhs_gather(const RhsArrays& R, uint32_t b, uint32_t e, uint32_t sub, float& ax, float& ay, float& az) {
  for (uint32_t k = b + sub; k < e; k += 4 * kRhsLanes) {  // four records per lane in flight: slot indices first, then the records
    uint32_t slot[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) slot[u] = (k + kRhsLanes * u < e) ? R.incSlot[k + kRhsLanes * u] : 0xffffffffu;
    Vec3f c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = (slot[u] != 0xffffffffu) ? R.contrib[slot[u]] : Vec3f{0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ax += c[u].x;
      ay += c[u].y;
      az += c[u].z;
    }
  }
}
// rhs_finish: the lanes' partial sums combined and added to f (= M s_n / h^2 of the node), then the terms one lane adds: contacts,
// shape / goal matching, floor
PIES_DEV float4 rhs_finish(const RhsArrays& R, uint32_t i, uint32_t sub, bool live, float4 f, float ax, float ay, float az) {
#pragma unroll
  for (int off = kRhsLanes / 2; off >= 1; off >>= 1) {
    ax += __shfl_xor(ax, off, kRhsLanes);
    ay += __shfl_xor(ay, off, kRhsLanes);
    az += __shfl_xor(az, off, kRhsLanes);
  }
  f.x += ax;
  f.y += ay;
  f.z += az;
  float tx = 0.f, ty = 0.f, tz = 0.f;
  if (R.tIncCnt && *R.tUsedCount != 0u) {  // point-triangle contacts (Solver.cpp:337-340): a node of a contact patch takes part in tens of contacts; the
    // team's lanes add its records like the ones above (four in flight per lane, partial sums combined pairwise) - one lane
    // walking the list made this launch 55 us with 29k contacts (2 dependent loads per record), the rest of it takes 10
    const uint32_t tc = live ? R.tIncCnt[i] : 0u;
    const uint32_t ts = tc ? R.tIncStart[i] : 0u;
    for (uint32_t k = sub; k < tc; k += 4 * kRhsLanes) {
      uint32_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (k + kRhsLanes * u < tc) ? R.tInc[ts + k + kRhsLanes * u] : 0xffffffffu;
      float4 c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = (v[u] != 0xffffffffu) ? R.tContrib[4 * (v[u] >> 2) + (v[u] & 3u)] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        tx += c[u].x;
        ty += c[u].y;
        tz += c[u].z;
      }
    }
#pragma unroll
    for (int off = kRhsLanes / 2; off >= 1; off >>= 1) {
      tx += __shfl_xor(tx, off, kRhsLanes);
      ty += __shfl_xor(ty, off, kRhsLanes);
      tz += __shfl_xor(tz, off, kRhsLanes);
    }
  }
  if (!live || sub != 0) return f;
  if (R.incPtrD) {  // shape then goal matching: force += float w * double projection (ShapeMatchingConstraint.cpp:58-72,147-161)
    const uint32_t ed = R.incPtrD[i + 1];
    for (uint32_t k = R.incPtrD[i]; k < ed; ++k) {
      const double4 c = R.contribD[R.incSlotD[k]];
      f.x = static_cast<float>(static_cast<double>(f.x) + c.w * c.x);
      f.y = static_cast<float>(static_cast<double>(f.y) + c.w * c.y);
      f.z = static_cast<float>(static_cast<double>(f.z) + c.w * c.z);
    }
  }
  f.x += tx;  // (after the shape-matching terms, as in the reference's loop order)
  f.y += ty;
  f.z += tz;
  const uint32_t ns = R.nstatic[i];
  if (ns) {
    float4 p = R.pos[i];
    if (p.y < 0.0f) p.y = 0.0f;
    R.statp[i] = p;
    const float cx = kStaticW * p.x, cy = kStaticW * p.y, cz = kStaticW * p.z;
    for (uint32_t k = 0; k < ns; ++k) {
      f.x += cx;
      f.y += cy;
      f.z += cz;
    }
  }
  return f;
}
PIES_DEV float4 rhs_of_node(const RhsArrays& R, uint32_t i, uint32_t sub, bool live) {
  const float4 f = live ? R.msn[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t b = live ? R.incPtr[i] : 0u, e = live ? R.incPtr[i + 1] : 0u;
  float ax = 0.f, ay = 0.f, az = 0.f;
  rhs_gather(R, b, e, sub, ax, ay, az);
  return rhs_finish(R, i, sub, live, f, ax, ay, az);
}
__global__ void __launch_bounds__(kBlock) k_pd_rhs(RhsArrays R, float4* __restrict__ rhs) {
  const uint32_t i = (xcd_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x) / kRhsLanes, sub = threadIdx.x & (kRhsLanes - 1);
  const float4 f = rhs_of_node(R, i, sub, i < R.n);
  if (i < R.n && sub == 0) rhs[i] = f;
}

// ------------------------------------------------------------------------------------------------------
// Jacobi-preconditioned CG, 3 right-hand sides at once.  Launch shape: kCgBlocks blocks of 256 threads,
// grid-stride.  Dot products: per-block partials, re-reduced in a fixed order by every block of the
// consuming kernel (deterministic, no atomics, no extra launch).
//   partB[b] = { rz[3], rr[3] }  of the current residual     (written by init / update)
//   partA[b] = { pAp[3] }                                    (written by ap)
//   scal     = { rz[2][3] ping-pong, bb[3] }                 (written by block 0 of ap)
// ------------------------------------------------------------------------------------------------------
struct Red6 {
  float v[6];
};

template <int NV> PIES_DEV void block_reduce_partials(const float* __restrict__ part, int stride, uint32_t nparts, float out[NV]) {
  __shared__ float lds[4][NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = 0.0f;
  for (uint32_t t = threadIdx.x; t < nparts; t += kBlock)  // fixed order: the same sum in every block
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] += part[t * stride + k];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] += __shfl_xor(acc[k], off, 64);
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) lds[wave][k] = acc[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) out[k] = ((lds[0][k] + lds[1][k]) + lds[2][k]) + lds[3][k];
  __syncthreads();
}

template <int NV> PIES_DEV void block_write_partial(const float acc_in[NV], float* __restrict__ part, int stride) {
  __shared__ float lds[4][NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = acc_in[k];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] += __shfl_xor(acc[k], off, 64);
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) lds[wave][k] = acc[k];
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) part[blockIdx.x * stride + k] = ((lds[0][k] + lds[1][k]) + lds[2][k]) + lds[3][k];
  __syncthreads();
}

PIES_DEV bool all_converged(const float rr[3], const float bb[3], float tol2) {
  return rr[0] <= tol2 * bb[0] && rr[1] <= tol2 * bb[1] && rr[2] <= tol2 * bb[2];
}

// Off-diagonal part of the contact blocks w*AtA for row i (the diagonal 3w / w is in cdiag): the point couples
// to the three triangle nodes with -w, each triangle node to the point with -w.  FETCH(j) returns the vector at j.
// The terms are added in contact-list order; the loads of four contacts are requested together (index, ids, then
// the vectors - three dependent trips per batch instead of per contact: a node of a contact patch sits in tens of
// contacts, and one lane walking them one by one made the SpMV ten times slower than without contacts).
template <class Fetch> PIES_DEV void contact_row(const CgArrays& A, uint32_t i, Fetch fetch, float& sx, float& sy, float& sz) {
  if (!A.tIncCnt || *A.tUsedCount == 0u) return;  // (no contact in this substep: one uniform word instead of a load per row)
  const uint32_t tc = A.tIncCnt[i];
  if (!tc) return;
  if (A.useCAp && A.rowLen) {  // the merged row of this substep exists (contact-heavy variant; reached from the CG continuation)
    const uint32_t len = A.rowLen[i];
    if (len != 0xffffffffu) {
      const uint32_t off = A.rowStart[i];
      for (uint32_t t = 0; t < len; ++t) {
        const float coef = A.rowCoef[off + t];
        float q[3];
        fetch(A.rowCol[off + t], q[0], q[1], q[2]);
        sx = fmaf(coef, q[0], sx); sy = fmaf(coef, q[1], sy); sz = fmaf(coef, q[2], sz);
      }
      return;
    }
  }
  const uint32_t ts = A.tIncStart[i];
  constexpr int kAhead = 4;
  for (uint32_t k0 = 0; k0 < tc; k0 += kAhead) {
    uint32_t v[kAhead];
    uint4 id[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) v[u] = A.tInc[ts + min(k0 + u, tc - 1)];  // clamped: unconditional loads
#pragma unroll
    for (int u = 0; u < kAhead; ++u) id[u] = A.tIds[v[u] >> 2];
    float q[kAhead][3][3];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const bool point = (v[u] & 3u) == 0u;  // the point's row couples to the three triangle nodes, their rows to the point
      fetch(point ? id[u].y : id[u].x, q[u][0][0], q[u][0][1], q[u][0][2]);
      fetch(id[u].z, q[u][1][0], q[u][1][1], q[u][1][2]);
      fetch(id[u].w, q[u][2][0], q[u][2][1], q[u][2][2]);
    }
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      if (k0 + u >= tc) break;
      const int terms = (v[u] & 3u) == 0u ? 3 : 1;
#pragma unroll
      for (int t = 0; t < 3; ++t)
        if (t < terms) {
          sx = fmaf(-kTriContactW, q[u][t][0], sx);
          sy = fmaf(-kTriContactW, q[u][t][1], sy);
          sz = fmaf(-kTriContactW, q[u][t][2], sz);
        }
    }
  }
}

// The slices a wavefront sweeps: blocks that share an XCD (equal blockIdx % 8, see xcd_block) take one contiguous
// part of the matrix, so the rows a slice gathers from were fetched into that XCD's L2 by its neighbours.
struct SliceSweep {
  uint32_t begin, end, step;
};
template <int LPR> PIES_DEV SliceSweep slice_sweep(uint32_t n, uint32_t nblocks) {  // nblocks: the launch's SpMV blocks (the first ones)
  constexpr uint32_t kRows = 64u / LPR;  // rows of a slice
  const uint32_t nslices = (n + kRows - 1u) / kRows;
  const uint32_t labels = nblocks < 8u ? nblocks : 8u;
  const uint32_t x = blockIdx.x % labels, xb = blockIdx.x / labels;
  const uint32_t nbx = (nblocks - x + labels - 1u) / labels;  // blocks carrying this label
  const uint32_t segBeg = static_cast<uint32_t>((static_cast<uint64_t>(nslices) * x) / labels);
  const uint32_t segEnd = static_cast<uint32_t>((static_cast<uint64_t>(nslices) * (x + 1u)) / labels);
  return {segBeg + xb * (kBlock / 64u) + (threadIdx.x >> 6), segEnd, nbx * (kBlock / 64u)};
}

// Statistics at the end of a solve (max relative residual over the tick's solves, iterations of the solve): run by one
// block, either of k_cg_finish or - for every solve but the last of a substep - of the next solve's k_cg_init, which
// saves a launch per local/global iteration.  prevPartB holds the finished solve's final residual partials.
PIES_DEV void solve_statistics(const CgArrays& A, const float* __restrict__ prevPartB) {
  float red[6];
  // a solve that converged before its last captured iteration left its final partials where k_cg_ap found them
  // (scal[10] = 1, scal[11] = 0: partI, 1 / 2: the ping-pong pair); otherwise the last k_cg_update wrote prevPartB
  const bool done = A.scal[10] != 0.0f;
  const int where = static_cast<int>(A.scal[11]);
  if (done && where == 0) block_reduce_partials<6>(A.partI, 9, A.nparts, red);
  else block_reduce_partials<6>(done ? (where == 1 ? A.partB0 : A.partB1) : prevPartB, 6, A.nparts, red);
  if (threadIdx.x == 0) {
    float worst = 0.f;
    bool above = false;  // the very test the CG kernels take their early exit on (all_converged)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float bb = A.scal[6 + c];
      const float rel = bb > 0.f ? red[3 + c] / bb : 0.f;
      worst = fmaxf(worst, rel);
      above = above || !(red[3 + c] <= A.tol2 * bb);
    }
    A.stats[0] = fmaxf(A.stats[0], worst);  // max over solves of ||r||^2 / ||b||^2
    A.stats[1] = fmaxf(A.stats[1], A.scal[9]);
    A.stats[2] += 1.0f;
    const float ranShort = above ? 1.0f : 0.0f;  // the solve used its whole captured budget and is still above the tolerance
    A.stats[3] += ranShort;
    // lifetime counters, as 64-bit integers in the words [4..5] and [6..7] (a float stops counting at 2^24)
    unsigned long long* life = reinterpret_cast<unsigned long long*>(A.stats + 4);
    life[0] += above ? 1ull : 0ull;
    life[1] += 1ull;
  }
}

// Contact part of (K + C) v for the nodes that take part in contacts, one wavefront per node: lane t takes the node's
// incidences t, t + 64, ... (list order inside a lane), the 64 partial sums are combined pairwise.  Used by the
// contact-heavy graph variant only (CgArrays::useCAp): a node of a contact patch sits in tens to hundreds of contacts, and
// the row's single lane walking them inside the SpMV made one CG iteration ~10x longer.
// the wavefront's sum over one node's contact rows; fetch(j, qx, qy, qz) reads the vector
template <class Fetch> PIES_DEV void contact_rows_of_node(const CgArrays& A, uint32_t node, uint32_t lane, Fetch fetch, float& sx, float& sy, float& sz) {
  const uint32_t tc = A.tIncCnt[node], ts = A.tIncStart[node];
  sx = 0.f; sy = 0.f; sz = 0.f;
  const uint32_t len = A.rowLen ? A.rowLen[node] : 0xffffffffu;
  if (len != 0xffffffffu) {  // merged row: one gather of the distinct columns (a handful per node), coefficient = -w * multiplicity
    const uint32_t off = A.rowStart[node];
    for (uint32_t t = lane; t < len; t += 64) {
      const float coef = A.rowCoef[off + t];
      float q[3];
      fetch(A.rowCol[off + t], q[0], q[1], q[2]);
      sx = fmaf(coef, q[0], sx); sy = fmaf(coef, q[1], sy); sz = fmaf(coef, q[2], sz);
    }
  } else
  for (uint32_t t = lane; t < tc; t += 64) {
    const uint32_t v = A.tInc[ts + t];
    const uint4 id = A.tIds[v >> 2];
    const bool point = (v & 3u) == 0u;
    float q0[3], q1[3] = {0.f, 0.f, 0.f}, q2[3] = {0.f, 0.f, 0.f};
    fetch(point ? id.y : id.x, q0[0], q0[1], q0[2]);
    if (point) { fetch(id.z, q1[0], q1[1], q1[2]); fetch(id.w, q2[0], q2[1], q2[2]); }
    sx = fmaf(-kTriContactW, q0[0], sx); sy = fmaf(-kTriContactW, q0[1], sy); sz = fmaf(-kTriContactW, q0[2], sz);
    if (point) {
      sx = fmaf(-kTriContactW, q1[0], sx); sy = fmaf(-kTriContactW, q1[1], sy); sz = fmaf(-kTriContactW, q1[2], sz);
      sx = fmaf(-kTriContactW, q2[0], sx); sy = fmaf(-kTriContactW, q2[1], sy); sz = fmaf(-kTriContactW, q2[2], sz);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sx += __shfl_xor(sx, off, 64);
    sy += __shfl_xor(sy, off, 64);
    sz += __shfl_xor(sz, off, 64);
  }
}
// (K + C) x for the nodes with contacts, before k_cg_init (the residual needs the complete row; inside the CG iterations
// the contact rows are summed by extra blocks of k_cg_ap itself)
__global__ void __launch_bounds__(kBlock) k_contact_rows(CgArrays A, const float4* __restrict__ x) {
  const uint32_t used = *A.tUsedCount, lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = (gridDim.x * kBlock) >> 6;
  for (uint32_t u = wave; u < used; u += nwaves) {
    const uint32_t node = A.tUsed[u];
    float sx, sy, sz;
    contact_rows_of_node(A, node, lane, [&](uint32_t j, float& qx, float& qy, float& qz) { const float4 v = x[j]; qx = v.x; qy = v.y; qz = v.z; }, sx, sy, sz);
    if (lane == 0) A.cAp[node] = make_float4(sx, sy, sz, 0.f);
  }
}

// the LPR lanes of a row combine their partial sums (pairwise; every lane of the row ends with the total)
template <int LPR> PIES_DEV void row_combine(float& sx, float& sy, float& sz) {
#pragma unroll
  for (int off = LPR / 2; off >= 1; off >>= 1) {
    sx += __shfl_xor(sx, off, LPR);
    sy += __shfl_xor(sy, off, LPR);
    sz += __shfl_xor(sz, off, LPR);
  }
}

// r = f - (K + C) x ; z = D^-1 r ; partB = {rz, rr} ; partI = {bb}.   LPR lanes per row (sliced ELL, see CgArrays).
// prevPartB != nullptr: an extra block closes the previous solve's statistics (its scal[] entries are still intact:
// this solve's k_cg_ap(0) is the first kernel to overwrite them).
template <int LPR>
__global__ void __launch_bounds__(kBlock) k_cg_init(CgArrays A, const float4* __restrict__ x, const float4* __restrict__ f,
                                                    const float* __restrict__ prevPartB) {
  if (blockIdx.x == A.nparts) {  // one block behind the SpMV blocks: bookkeeping only (inside block 0 it delayed that block's rows by 2-3 us)
    if (prevPartB) solve_statistics(A, prevPartB);
    if (threadIdx.x == 0) {
      A.scal[10] = 0.0f;  // this solve has not converged yet (read by k_cg_ap / k_cg_update)
      A.ticket[0] = 0u;   // grid barrier counter of the solve's last k_cg_update
      A.ticket[1] = 0u;   // its abort word
    }
    return;
  }
  const uint32_t lane = threadIdx.x & 63u;
  const SliceSweep sw = slice_sweep<LPR>(A.n, A.nparts);
  float acc9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t sl = sw.begin; sl < sw.end; sl += sw.step) {
    const uint32_t i = sl * (64u / LPR) + lane / LPR;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (LPR == 1 && A.rowStencil) {  // row dictionary: the row's (column - row, value) pairs, shared by every row like it
      if (i < A.n) {
        const uint32_t rs = A.rowStencil[i];
        const uint32_t b = rs & 0xffffffu, e = b + (rs >> 24);
#pragma unroll 4
        for (uint32_t k = b; k < e; ++k) {
          const int2 p = A.stencil[k];
          const float a = __int_as_float(p.y);
          const float4 xj = x[static_cast<uint32_t>(static_cast<int>(i) + p.x)];
          sx = fmaf(a, xj.x, sx);
          sy = fmaf(a, xj.y, sy);
          sz = fmaf(a, xj.z, sz);
        }
      }
    } else {
      const uint32_t off = A.sliceOff[sl], width = (A.sliceOff[sl + 1] - off) >> 6;
#pragma unroll 4
      for (uint32_t k = 0; k < width; ++k) {
        const uint32_t at = off + (k << 6) + lane;
        const float a = A.val[at];
        const float4 xj = x[A.col[at]];
        sx = fmaf(a, xj.x, sx);
        sy = fmaf(a, xj.y, sy);
        sz = fmaf(a, xj.z, sz);
      }
    }
    row_combine<LPR>(sx, sy, sz);
    if (i < A.n && lane % LPR == 0u) {
      if (A.useCAp) {
        if (*A.tUsedCount != 0u && A.tIncCnt[i]) {
          const float4 c = A.cAp[i]; sx += c.x; sy += c.y; sz += c.z;
        }
      } else {
        contact_row(A, i, [&](uint32_t j, float& px, float& py, float& pz) { const float4 v = x[j]; px = v.x; py = v.y; pz = v.z; }, sx, sy, sz);
      }
      const float4 xi = x[i], fi = f[i];
      const float cd = A.cdiag[i], di = A.dinv[i];
      const float rx = fi.x - fmaf(cd, xi.x, sx), ry = fi.y - fmaf(cd, xi.y, sy), rz = fi.z - fmaf(cd, xi.z, sz);
      const float zx = di * rx, zy = di * ry, zz = di * rz;
      A.r[i] = make_float4(rx, ry, rz, 0.f);
      A.z[i] = make_float4(zx, zy, zz, 0.f);
      acc9[0] += rx * zx; acc9[1] += ry * zy; acc9[2] += rz * zz;
      acc9[3] += rx * rx; acc9[4] += ry * ry; acc9[5] += rz * rz;
      acc9[6] += fi.x * fi.x; acc9[7] += fi.y * fi.y; acc9[8] += fi.z * fi.z;
    }
  }
  block_write_partial<9>(acc9, A.partI, 9);
}

// The SpMV rows of iteration k for this workgroup's slices: p = z + beta p_old (written), Ap = (K + C) p (written; the contact
// part only when inlineContacts, i.e. summed by the row's lane), acc += p.Ap.
template <int LPR> PIES_DEV void cg_ap_rows(const CgArrays& A, int k, const float beta[3], bool inlineContacts, float acc[3]) {
  const float4* __restrict__ pold = A.p[(k + 1) & 1];
  float4* __restrict__ pnew = A.p[k & 1];
  const uint32_t lane = threadIdx.x & 63u;
  const SliceSweep sw = slice_sweep<LPR>(A.n, A.nparts);
  for (uint32_t sl = sw.begin; sl < sw.end; sl += sw.step) {
    const uint32_t i = sl * (64u / LPR) + lane / LPR;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (LPR == 1 && A.rowStencil) {  // row dictionary (see k_cg_init)
      if (i < A.n) {
        const uint32_t rs = A.rowStencil[i];
        const uint32_t sb = rs & 0xffffffu, se = sb + (rs >> 24);
        if (k > 0) {
#pragma unroll 4
          for (uint32_t q = sb; q < se; ++q) {
            const int2 pr = A.stencil[q];
            const float a = __int_as_float(pr.y);
            const uint32_t j = static_cast<uint32_t>(static_cast<int>(i) + pr.x);
            const float4 zj = A.z[j], pj = pold[j];
            sx = fmaf(a, fmaf(beta[0], pj.x, zj.x), sx);
            sy = fmaf(a, fmaf(beta[1], pj.y, zj.y), sy);
            sz = fmaf(a, fmaf(beta[2], pj.z, zj.z), sz);
          }
        } else {
#pragma unroll 4
          for (uint32_t q = sb; q < se; ++q) {
            const int2 pr = A.stencil[q];
            const float a = __int_as_float(pr.y);
            const float4 zj = A.z[static_cast<uint32_t>(static_cast<int>(i) + pr.x)];
            sx = fmaf(a, zj.x, sx);
            sy = fmaf(a, zj.y, sy);
            sz = fmaf(a, zj.z, sz);
          }
        }
      }
    } else {
    const uint32_t off = A.sliceOff[sl], width = (A.sliceOff[sl + 1] - off) >> 6;
    if (k > 0) {
#pragma unroll 4
      for (uint32_t kk = 0; kk < width; ++kk) {
        const uint32_t at = off + (kk << 6) + lane;
        const float a = A.val[at];
        const uint32_t j = A.col[at];
        const float4 zj = A.z[j], pj = pold[j];
        sx = fmaf(a, fmaf(beta[0], pj.x, zj.x), sx);
        sy = fmaf(a, fmaf(beta[1], pj.y, zj.y), sy);
        sz = fmaf(a, fmaf(beta[2], pj.z, zj.z), sz);
      }
    } else {
#pragma unroll 4
      for (uint32_t kk = 0; kk < width; ++kk) {
        const uint32_t at = off + (kk << 6) + lane;
        const float a = A.val[at];
        const float4 zj = A.z[A.col[at]];
        sx = fmaf(a, zj.x, sx);
        sy = fmaf(a, zj.y, sy);
        sz = fmaf(a, zj.z, sz);
      }
    }
    }
    row_combine<LPR>(sx, sy, sz);
    if (i < A.n && lane % LPR == 0u) {
      if (inlineContacts) {
        contact_row(A, i, [&](uint32_t j, float& qx, float& qy, float& qz) {
          const float4 zj = A.z[j];
          qx = zj.x; qy = zj.y; qz = zj.z;
          if (k > 0) {
            const float4 pj = pold[j];
            qx = fmaf(beta[0], pj.x, qx); qy = fmaf(beta[1], pj.y, qy); qz = fmaf(beta[2], pj.z, qz);
          }
        }, sx, sy, sz);
      }
      const float4 zi = A.z[i];
      float px = zi.x, py = zi.y, pz = zi.z;
      if (k > 0) {
        const float4 pi = pold[i];
        px = fmaf(beta[0], pi.x, px);
        py = fmaf(beta[1], pi.y, py);
        pz = fmaf(beta[2], pi.z, pz);
      }
      const float cd = A.cdiag[i];
      const float ax = fmaf(cd, px, sx), ay = fmaf(cd, py, sy), az = fmaf(cd, pz, sz);
      pnew[i] = make_float4(px, py, pz, 0.f);
      A.ap[i] = make_float4(ax, ay, az, 0.f);
      acc[0] += px * ax;
      acc[1] += py * ay;
      acc[2] += pz * az;
    }
  }
}

// iteration k:  beta = rz_k / rz_{k-1} (0 for k = 0) ; p = z + beta p_old ; Ap = (K + C) p ; partA = {pAp}
// With useCAp the launch carries kCgRowBlocks extra blocks behind the nparts SpMV blocks: they sum the contact rows of p
// (one wavefront per node, written to cAp) and add their share of p.Ap to partA; k_cg_update adds cAp to Ap.  (Round 2: a
// launch of their own before every k_cg_ap, 100 launches per substep of a contact scene.)
template <int LPR> __global__ void __launch_bounds__(kBlock) k_cg_ap(CgArrays A, int k, float tol2) {
  // The solve converged in an earlier iteration: nothing to do, and nothing to read but this word (without it every block
  // of the remaining captured launches re-reduced the residual partials to find that out: 4.7 instead of 2.5 us per launch
  // at 100k rows, and half of config 3's CG launches are such exits).
  if (A.scal[10] != 0.0f) return;
  float red[9];
  float rz[3], rr[3], bb[3];
  if (k == 0) {
    block_reduce_partials<9>(A.partI, 9, A.nparts, red);
#pragma unroll
    for (int c = 0; c < 3; ++c) { rz[c] = red[c]; rr[c] = red[3 + c]; bb[c] = red[6 + c]; }
  } else {
    block_reduce_partials<6>(A.partB, 6, A.nparts, red);
#pragma unroll
    for (int c = 0; c < 3; ++c) { rz[c] = red[c]; rr[c] = red[3 + c]; bb[c] = A.scal[6 + c]; }
  }
  if (k == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    A.scal[6] = bb[0];
    A.scal[7] = bb[1];
    A.scal[8] = bb[2];
    A.scal[9] = 0.0f;  // iterations started in this solve
  }
  if (all_converged(rr, bb, tol2)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      A.scal[11] = k == 0 ? 0.0f : static_cast<float>(1 + (k & 1));  // where the final residual partials are (solve_statistics)
      A.scal[10] = 1.0f;
    }
    return;
  }
  float beta[3] = {0.f, 0.f, 0.f};
  if (k > 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float old = A.scal[3 * ((k - 1) & 1) + c];
      beta[c] = old > 0.0f ? rz[c] / old : 0.0f;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) A.scal[3 * (k & 1) + c] = rz[c];
    A.scal[9] = static_cast<float>(k + 1);
  }
  const float4* __restrict__ pold = A.p[(k + 1) & 1];
  const uint32_t lane = threadIdx.x & 63u;
  float acc[3] = {0, 0, 0};
  if (blockIdx.x >= A.nparts) {  // contact rows of p = z + beta p_old
    auto fetch = [&](uint32_t j, float& qx, float& qy, float& qz) {
      const float4 zj = A.z[j];
      qx = zj.x; qy = zj.y; qz = zj.z;
      if (k > 0) {
        const float4 pj = pold[j];
        qx = fmaf(beta[0], pj.x, qx); qy = fmaf(beta[1], pj.y, qy); qz = fmaf(beta[2], pj.z, qz);
      }
    };
    const uint32_t used = *A.tUsedCount;
    const uint32_t wave = ((blockIdx.x - A.nparts) * kBlock + threadIdx.x) >> 6, nwaves = ((gridDim.x - A.nparts) * kBlock) >> 6;
    for (uint32_t u = wave; u < used; u += nwaves) {
      const uint32_t node = A.tUsed[u];
      float sx, sy, sz;
      contact_rows_of_node(A, node, lane, fetch, sx, sy, sz);
      if (lane == 0) {
        A.cAp[node] = make_float4(sx, sy, sz, 0.f);
        float px, py, pz;
        fetch(node, px, py, pz);
        acc[0] += px * sx; acc[1] += py * sy; acc[2] += pz * sz;
      }
    }
    block_write_partial<3>(acc, A.partA, 3);
    return;
  }
  cg_ap_rows<LPR>(A, k, beta, !A.useCAp, acc);
  block_write_partial<3>(acc, A.partA, 3);
}

// x += alpha p ; r -= alpha Ap ; z = D^-1 r for this workgroup's rows; acc += {r.z, r.r} per column
PIES_DEV void cg_update_rows(const CgArrays& A, float4* __restrict__ x, int k, const float alpha[3], bool addCAp, float acc[6]) {
  const float4* __restrict__ p = A.p[k & 1];
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < A.n; i += gridDim.x * kBlock) {
    const float4 pi = p[i];
    float4 api = A.ap[i];
    if (addCAp && *A.tUsedCount != 0u && A.tIncCnt[i]) {
      const float4 c = A.cAp[i];
      api.x += c.x; api.y += c.y; api.z += c.z;
    }
    float4 xi = x[i], ri = A.r[i];
    xi.x = fmaf(alpha[0], pi.x, xi.x);
    xi.y = fmaf(alpha[1], pi.y, xi.y);
    xi.z = fmaf(alpha[2], pi.z, xi.z);
    ri.x = fmaf(-alpha[0], api.x, ri.x);
    ri.y = fmaf(-alpha[1], api.y, ri.y);
    ri.z = fmaf(-alpha[2], api.z, ri.z);
    const float di = A.dinv[i];
    const float zx = di * ri.x, zy = di * ri.y, zz = di * ri.z;
    x[i] = xi;
    A.r[i] = ri;
    A.z[i] = make_float4(zx, zy, zz, 0.f);
    acc[0] += ri.x * zx; acc[1] += ri.y * zy; acc[2] += ri.z * zz;
    acc[3] += ri.x * ri.x; acc[4] += ri.y * ri.y; acc[5] += ri.z * ri.z;
  }
}

// Barrier across the workgroups of a launch whose workgroups are all resident (k_cg_update: at most 1024 of 256 threads with
// a few hundred bytes of LDS, the chip holds 5 x 256 of them).  `counter` only grows (the solve's first kernel zeroes it);
// release before the arrival, acquire after the last one, as a grid-wide synchronisation has to.  The wait is bounded: a
// workgroup that gives up returns false and leaves, the others follow at their next barrier (the solve then stays where it
// was and is counted as short).
PIES_DEV bool grid_barrier(uint32_t* counter, uint32_t nblocks, uint32_t& passed) {
  __shared__ uint32_t sOk;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(counter, 1u);
    const uint32_t target = (passed + 1u) * nblocks;
    uint32_t spins = 0, ok = 1u;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 20)) { ok = 0u; break; }  // ~1 s
    }
    // a workgroup that gives up says so in the abort word; one that arrives late and finds the counter already past its
    // target (the others have left) must not run an iteration alone: everybody checks the word after the wait
    if (!ok) __hip_atomic_store(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (__hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = 0u;
    __threadfence();
    sOk = ok;
  }
  __syncthreads();
  ++passed;
  return sOk != 0u;
}

// The continuation's grid barrier needs every workgroup of the launch resident at once: how many k_cg_update workgroups the
// device holds (pd_setup.cpp sizes the CG kernels' grid below it)
uint32_t cg_update_resident_blocks(int device);

// alpha = rz_k / pAp ; x += alpha p ; r -= alpha Ap ; z = D^-1 r ; partB = {rz_{k+1}, rr_{k+1}}.
// overflow > 0: this is the solve's last captured iteration.  If the residual is still above the tolerance after it (new
// contacts stiffened the system since the budget was chosen, and the host has not looked yet), the launch goes on: its
// workgroups run up to `overflow` more iterations themselves, a grid barrier where the captured path has a kernel boundary
// (k_cg_ap's rows, barrier, these rows, barrier), the contact rows summed lane by lane.  An iteration costs about what a
// captured one does, so neither pies_tick nor a blind queue of pies_tick_async calls feeds an unconverged solve into the
// next substep, and the host raises the captured budget at its next look.  (The first version let the last workgroup to
// finish go on alone: 4 ms per iteration at 125k rows, a second per frame at a contact onset.)
__global__ void __launch_bounds__(kBlock) k_cg_update(CgArrays A, float4* __restrict__ x, int k, float tol2, int overflow) {
  // k_cg_ap(k) has looked at the residual of iteration k: had the solve converged, it would have set the flag and produced
  // nothing.  (Until the flag existed this kernel re-reduced the residual partials to take the same decision: one block-wide
  // reduction per launch for nothing.)
  if (A.scal[10] != 0.0f) return;
  float red[9];
  float rr[3], bb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) bb[c] = A.scal[6 + c];  // (written by k_cg_ap(0) of this solve)
  const bool rows = A.useCAp && A.tIncCnt;
  float pap[3];
  block_reduce_partials<3>(A.partA, 3, A.nparts + (rows ? kCgRowBlocks : 0u), pap);
  float alpha[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float rzk = A.scal[3 * (k & 1) + c];
    alpha[c] = pap[c] > 0.0f ? rzk / pap[c] : 0.0f;
  }
  float acc[6] = {0, 0, 0, 0, 0, 0};
  // blocks run concurrently, so the new residual partials go to the other half of a ping-pong pair
  cg_update_rows(A, x, k, alpha, rows, acc);
  block_write_partial<6>(acc, A.partBnext, 6);
  if (overflow <= 0 || A.lanesPerRow != 1u) return;
  // ---- the iterations beyond the captured ones --------------------------------------------------------------------
  float* const pb[2] = {A.partB0, A.partB1};
  uint32_t passed = 0;
  int kk = k + 1;
  float rzOld[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) rzOld[c] = A.scal[3 * (k & 1) + c];  // (kept in registers: block 0 does not have to publish them)
  for (;;) {
    if (!grid_barrier(A.ticket, gridDim.x, passed)) return;  // the partials of iteration kk - 1 are complete
    float rz[3];
    block_reduce_partials<6>(pb[kk & 1], 6, A.nparts, red);
#pragma unroll
    for (int c = 0; c < 3; ++c) { rz[c] = red[c]; rr[c] = red[3 + c]; }
    if (all_converged(rr, bb, tol2) || kk >= k + 1 + overflow) break;
    float beta[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) beta[c] = rzOld[c] > 0.0f ? rz[c] / rzOld[c] : 0.0f;
    float a3[3] = {0, 0, 0};
    cg_ap_rows<1>(A, kk, beta, true, a3);
    block_write_partial<3>(a3, A.partA, 3);
    if (!grid_barrier(A.ticket, gridDim.x, passed)) return;
    block_reduce_partials<3>(A.partA, 3, A.nparts, pap);
#pragma unroll
    for (int c = 0; c < 3; ++c) alpha[c] = pap[c] > 0.0f ? rz[c] / pap[c] : 0.0f;
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[c] = 0.0f;
    cg_update_rows(A, x, kk, alpha, false, acc);
    block_write_partial<6>(acc, pb[(kk + 1) & 1], 6);
#pragma unroll
    for (int c = 0; c < 3; ++c) rzOld[c] = rz[c];
    ++kk;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // where the final residual partials are, and how many iterations it took
    A.scal[9] = static_cast<float>(kk);
    A.scal[11] = static_cast<float>(1 + (kk & 1));
    A.scal[10] = 1.0f;
  }
}

// end of the last solve of a substep: its statistics
__global__ void __launch_bounds__(kBlock) k_cg_finish(CgArrays A) { solve_statistics(A, A.partB); }

uint32_t cg_update_resident_blocks(int device) {
  int perCu = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_cg_update, kBlock, 0) != hipSuccess) return 0;
  return static_cast<uint32_t>(std::max(0, perCu)) * static_cast<uint32_t>(std::max(0, prop.multiProcessorCount));
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
void launch_pd_predict(hipStream_t st, const NodeArrays& nd, const PdArrays& pd, float h, float contactHeight) {
  if (nd.n == 0) return;
  hipLaunchKernelGGL(k_pd_predict, grid_for(nd.n), dim3(kBlock), 0, st, nd.pos, nd.vel, pd.msn, pd.triCount, pd.nstatic, pd.kdiag,
                     pd.cg.cdiag, pd.cg.dinv, nd.n, h, h * h, contactHeight);
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
static RhsArrays rhs_arrays(const NodeArrays& nd, const PdArrays& pd) {
  return RhsArrays{pd.msn, pd.contrib, pd.incPtr, pd.incSlot, pd.contribD, pd.incPtrD, pd.incSlotD, nd.pos, pd.nstatic, pd.statp,
                   pd.cg.tIncCnt, pd.cg.tIncStart, pd.cg.tInc, pd.tContrib, pd.cg.tUsedCount, nd.n};
}
void launch_pd_rhs(hipStream_t st, const NodeArrays& nd, const PdArrays& pd) {
  if (nd.n == 0) return;
  hipLaunchKernelGGL(k_pd_rhs, dim3((nd.n + kBlock / kRhsLanes - 1) / (kBlock / kRhsLanes)), dim3(kBlock), 0, st, rhs_arrays(nd, pd), pd.rhs);
}
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
void launch_pd_local_bend(hipStream_t st, const float4* pos, const uint4* ids, const float2* angle_w, Vec3f* contrib, uint32_t count) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_pd_local_bend, grid_for(count), dim3(kBlock), 0, st, pos, ids, angle_w, contrib, count);
}
void launch_pd_local_shape(hipStream_t st, const float4* pos, const PdArrays& pd) {
  if (pd.shape.count == 0) return;
  hipLaunchKernelGGL(k_pd_local_shape, dim3(pd.shape.count), dim3(kBlock), 0, st, pos, pd.shape, pd.contribD);
}
void launch_pd_solve(hipStream_t st, const NodeArrays& nd, const PdArrays& pd, int maxIters, float tol, int part, bool first, bool last,
                     bool neverExit, void (*hook)(void*, int), void* hookCtx, int overflowIters) {
  if (nd.n == 0) return;
  CgArrays A = pd.cg;
  const dim3 grid(A.nparts), block(kBlock);
  const bool rows = A.useCAp && A.tIncCnt;
  const dim3 agrid(A.nparts + (rows ? kCgRowBlocks : 0u));  // k_cg_ap: SpMV blocks + contact-row blocks
  if (part >= 0) {  // profile pass: one kind of kernel only, never taking the converged early exit
    (void)hipMemsetAsync(A.scal + 10, 0, sizeof(float), st);  // (the last real solve may have left "converged" behind)
    for (int k = 0; k < maxIters; ++k) {
      if (part == 1) {
        if (A.lanesPerRow == 4) hipLaunchKernelGGL(k_cg_ap<4>, agrid, block, 0, st, A, k, -1.0f);
        else if (A.lanesPerRow == 2) hipLaunchKernelGGL(k_cg_ap<2>, agrid, block, 0, st, A, k, -1.0f);
        else if (A.lanesPerRow == 8) hipLaunchKernelGGL(k_cg_ap<8>, agrid, block, 0, st, A, k, -1.0f);
        else hipLaunchKernelGGL(k_cg_ap<1>, agrid, block, 0, st, A, k, -1.0f);
      }
      else hipLaunchKernelGGL(k_cg_update, grid, block, 0, st, A, nd.pos, k, -1.0f, 0);
    }
    return;
  }
  const float tol2 = neverExit ? -1.0f : tol * tol;
  A.tol2 = tol * tol;
  float* pb[2] = {pd.cg.partB, pd.cg.partBnext};
  // every solve of a substep runs the same number of iterations, so the previous solve left its final partials here
  if (rows) hipLaunchKernelGGL(k_contact_rows, dim3(256), block, 0, st, A, nd.pos);
  const float* prevB = first ? nullptr : pb[maxIters & 1];
  const dim3 igrid(A.nparts + 1u);  // k_cg_init: SpMV blocks + the bookkeeping block
  if (A.lanesPerRow == 4) hipLaunchKernelGGL(k_cg_init<4>, igrid, block, 0, st, A, nd.pos, pd.rhs, prevB);
  else if (A.lanesPerRow == 2) hipLaunchKernelGGL(k_cg_init<2>, igrid, block, 0, st, A, nd.pos, pd.rhs, prevB);
  else if (A.lanesPerRow == 8) hipLaunchKernelGGL(k_cg_init<8>, igrid, block, 0, st, A, nd.pos, pd.rhs, prevB);
  else hipLaunchKernelGGL(k_cg_init<1>, igrid, block, 0, st, A, nd.pos, pd.rhs, prevB);
  for (int k = 0; k < maxIters; ++k) {
    A.partB = pb[k & 1];       // residual partials of iteration k (k = 0 reads partI instead)
    A.partBnext = pb[(k + 1) & 1];
    if (hook) hook(hookCtx, 14);  // PIES_KERNEL_PD_SPMV
    if (A.lanesPerRow == 4) hipLaunchKernelGGL(k_cg_ap<4>, agrid, block, 0, st, A, k, tol2);
    else if (A.lanesPerRow == 2) hipLaunchKernelGGL(k_cg_ap<2>, agrid, block, 0, st, A, k, tol2);
    else if (A.lanesPerRow == 8) hipLaunchKernelGGL(k_cg_ap<8>, agrid, block, 0, st, A, k, tol2);
    else hipLaunchKernelGGL(k_cg_ap<1>, agrid, block, 0, st, A, k, tol2);
    if (hook) { hook(hookCtx, 14); hook(hookCtx, 15); }  // PIES_KERNEL_PD_CG_UPDATE
    hipLaunchKernelGGL(k_cg_update, grid, block, 0, st, A, nd.pos, k, tol2, k + 1 == maxIters && !neverExit ? overflowIters : 0);
    if (hook) hook(hookCtx, 15);
  }
  if (!last) return;
  A.partB = pb[maxIters & 1];
  hipLaunchKernelGGL(k_cg_finish, dim3(1), block, 0, st, A);
}
void launch_pd_stabilize(hipStream_t st, const NodeArrays& nd, const PdArrays& pd, bool closeSolve, int maxIters, float tol) {
  if (nd.n == 0) return;
  CgArrays A = pd.cg;
  A.tol2 = tol * tol;
  float* pb[2] = {pd.cg.partB, pd.cg.partBnext};
  A.partB = pb[maxIters & 1];  // where the last solve's final residual partials are (launch_pd_solve)
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
