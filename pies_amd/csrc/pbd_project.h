// The PBD projections on node records held in registers (gfx950).  The global-memory kernels of pbd_kernels.hip
// (one launch per conflict-free batch) and the LDS-resident layer kernel of layer_kernels.hip both run these, so
// every schedule executes the same IEEE sequence per constraint; they differ only in where the node records live.
#pragma once
#include "dev_math.h"

namespace pies {

// Solver.cpp:47-52   prev = pos;  pos += v*dt + (0,-g,0)*dt*dt
PIES_DEV void predict_core(float4& p, const float4 v, const float dt, const float g) {
  const float gx = (0.0f * dt) * dt, gy = (-g * dt) * dt, gz = (0.0f * dt) * dt;
  p.x = p.x + (v.x * dt + gx);
  p.y = p.y + (v.y * dt + gy);
  p.z = p.z + (v.z * dt + gz);
}

// Solver.cpp:140-158   v = (1-damping)*(pos-prev)/dt, floor friction with the hard-coded speed 5.0
PIES_DEV float4 velocity_core(const float4 p, const float4 q, const float r, const float dt, const float damping,
                              const float friction, const float floorHeight) {
  const float k = 1.0f - damping;
  float vx = (k * (p.x - q.x)) / dt;
  float vy = (k * (p.y - q.y)) / dt;
  float vz = (k * (p.z - q.z)) / dt;
  if (p.y - r <= floorHeight) {
    const float l = sqrtf(vx * vx + vz * vz);
    if (l < 5.0f) {
      vx = 0.0f;
      vz = 0.0f;
    } else {
      vx *= 1.0f - friction;
      vz *= 1.0f - friction;
    }
  }
  return make_float4(vx, vy, vz, 0.0f);
}

// PositionConstraint (Constraints.cpp:58-63 through Constraints.h:121-129): pos += w*(fixed - pos)
PIES_DEV void position_core(float4& p, const float4 tw) {
  p.x += tw.w * (tw.x - p.x);
  p.y += tw.w * (tw.y - p.y);
  p.z += tw.w * (tw.z - p.z);
}

// DistanceConstraint (Constraints.cpp:11-37): only node a moves, by the full correction.  rw = (rest, w)
PIES_DEV void distance_core(float4& a, const float4 b, const float2 rw) {
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
}

// TetrahedralConstraint (Constraints.cpp:76-128) applied as a PBD projection (Constraints.h:121-129).
// Record layout: a0 = Qinv col0 + Qinv[1][0], a1 = Qinv[1][1..2] + Qinv[2][0..1], a2 = Qinv[2][2], min, max, w.
template <int VARIANT>
PIES_DEV void tet_core(float4& x1, float4& x2, float4& x3, float4& x4, const float4 a0, const float4 a1, const float4 a2) {
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
    return;
  }
  Svd3 d;
  if (VARIANT == 2) svd3_jacobi(F, d);  // experiment: the iteration from V = I (rounds 1-5; tools/svd_bench.hip)
  else svd3(F, d);
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
}

// BendConstraint (Constraints.cpp:312-366): dihedral-angle projection, mass weighted.  aw = (angle, w).
// Returns false when the projection leaves the nodes where they are (nothing to store).
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

PIES_DEV bool bend_core(float4& x1, float4& x2, float4& x3, float4& x4, const float2 aw) {
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
  if (qSq < 0.00001f) return false;  // projection = current positions: pos += w*0
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
  return true;
}

// Solver.cpp:132-136
PIES_DEV bool floor_core(float4& p, const float r, const float floorHeight) {
  if (p.y - r < floorHeight) {
    p.y = floorHeight + r;
    return true;
  }
  return false;
}

}  // namespace pies
