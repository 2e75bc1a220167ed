// The right-hand side of one node (Solver.cpp:266, 310-349), shared by k_pd_rhs (pd_kernels.hip) and by the residual
// kernel of the one-launch-per-iteration CG when it evaluates the right-hand side itself (pd_cg1_kernels.hip).
#pragma once
#include <cstdint>

#include "dev_math.h"
#include "pd_kernels.h"

namespace pies {

constexpr float kStaticW = 10000.0f;  // StaticCollisionConstraint::w (Include/Pies/CollisionConstraint.h:78)

// lanes that share a node's gather (each with four records in flight).  Measured per launch at 100k / 1M nodes with ~24
// records per node: 16 lanes 13.1 / 136 us, 8 lanes 10.2 / 95, 4 lanes 9.3 / 88, 2 lanes 10.6 / 105.
// (k_pd_rhs<4> is the gather of the per-(constraint, node) records; one lane per node serves scenes whose records are tile sums,
// two or three per node, and the residual kernel that evaluates the right-hand side itself)
// The right-hand side of node i, by the kRhsLanes lanes that share it (`sub` = the lane's place among them; every lane of the
// wavefront calls these, lanes past the last node with live = false).  The value is complete in the lane with sub == 0.
// kRhsLanes lanes per node: lane `sub` adds up the records sub, sub + kRhsLanes, ... of the node's slot list (four
// slot indices and records per lane in flight at once), then the partial sums are combined pairwise.  The reference adds
// the same terms one after the other; the difference is fp32 rounding of a ~50-term sum (PD parity is by
// tolerance).  (Measured at 100k nodes: 1 lane/node 60 us; 16 lanes with the terms added in
// list order by one lane 24 us; visiting nodes in Morton order was slower than index order.)
// rhs_gather: this lane's share of the records [b, e) of the slot list, added to (ax, ay, az)
template <uint32_t kRhsLanes> PIES_DEV void rhs_gather(const RhsArrays& R, uint32_t b, uint32_t e, uint32_t sub, float& ax, float& ay, float& az) {
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
template <uint32_t kRhsLanes> PIES_DEV float4 rhs_finish(const RhsArrays& R, uint32_t i, uint32_t sub, bool live, float4 f, float ax, float ay, float az) {
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
template <uint32_t kRhsLanes> PIES_DEV float4 rhs_of_node(const RhsArrays& R, uint32_t i, uint32_t sub, bool live) {
  const float4 f = live ? R.msn[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t b = live ? R.incPtr[i] : 0u, e = live ? R.incPtr[i + 1] : 0u;
  float ax = 0.f, ay = 0.f, az = 0.f;
  rhs_gather<kRhsLanes>(R, b, e, sub, ax, ay, az);
  return rhs_finish<kRhsLanes>(R, i, sub, live, f, ax, ay, az);
}

}  // namespace pies
