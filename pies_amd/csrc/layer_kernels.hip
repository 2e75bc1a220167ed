// Schedule LAYERED on gfx950: one workgroup sweeps one group of constraints (layer_plan.cpp) with the group's
// node records resident in LDS.
//
// A launch covers the groups of one parity (they share no node).  The workgroup copies its two levels of node
// records (16 B each) from HBM into LDS once, runs the launch's segments - phases of the distance / tetrahedral /
// bend / position containers colour after colour, and the per-node steps of the substep (predict, floor clamp,
// velocity) - and writes the records back once.  Between two colours stands a workgroup barrier (~0.1 us) where
// the batch-per-launch schedules pay a kernel boundary plus two dependent HBM round trips (~3.8 us); constraint
// records (local ids 4-8 B, rest data) stream from HBM and the next colour's are requested before the current
// colour is computed, so their latency hides behind the arithmetic.
//
// Arithmetic per constraint is pbd_project.h's, the same as the global-memory kernels'.  No MFMA: 3x3 algebra per
// lane.  Algorithmic bytes are those of the projections executed; HBM traffic is lower because a node
// record is read and written once per launch instead of once per incident constraint.
#include <cstdlib>

#include "dev_math.h"
#include "kernels.h"
#include "pbd_project.h"

namespace pies {

constexpr uint32_t kOffStride = kLayerMaxCols + 2;  // colour offsets of one segment in LDS (+1 end, +1 look-ahead)

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the global loads in flight
// (s_waitcnt vmcnt(0)), i.e. the constraint records requested ahead for the next colours.
PIES_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int kDistPreload = 12;
constexpr int kBatch = 4;  // node records a lane requests before it consumes the first (the load phase is written out for 4)  // colours of a distance segment whose records a lane requests up front

PIES_DEV uint32_t lo16(uint32_t v) { return v & 0xFFFFu; }
PIES_DEV uint32_t hi16(uint32_t v) { return v >> 16; }

template <int BLOCK, int TETV>
__global__ void __launch_bounds__(BLOCK) k_layer(NodeArrays nd, LayerData D, LayerLaunch L, LayerParams P) {
  extern __shared__ float4 lds[];
  float4* __restrict__ sp = lds;                                                  // node records of the group
  float* __restrict__ srad = reinterpret_cast<float*>(sp + D.maxGroupNodes);      // their radii
  uint32_t* __restrict__ soff = reinterpret_cast<uint32_t*>(srad + D.maxGroupNodes);  // colour offsets, per segment
  const uint32_t tid = threadIdx.x;
  const uint32_t g = xcd_block(blockIdx.x, gridDim.x);  // neighbouring groups (shared levels) meet in one XCD's L2
  const uint32_t n0 = D.groupOff[L.parity][g];
  const uint32_t m = D.groupOff[L.parity][g + 1] - n0;
  const uint32_t* __restrict__ nodes = D.nodeList + n0;
  if (m == 0) return;  // uniform: an empty group (odd group 0 of a one-level body)

  bool needRadius = false;
  for (uint32_t s = 0; s < L.nseg; ++s) needRadius |= L.seg[s].kind == LAYER_FLOOR || L.seg[s].kind == LAYER_VELOCITY;
  // Node records in: from the layer-ordered copy (the group's range is contiguous: coalesced) or, in the first launch
  // of a substep / after a collision pass, gathered from the node array.  Four requests per lane are in flight before
  // the first is consumed (clamped indices keep the loads unconditional, so nothing waits at a branch join).
  if (L.loadGlobal) {
    for (uint32_t base = 0; base < m; base += kBatch * BLOCK) {
      uint32_t v[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) v[k] = nodes[min(base + k * BLOCK + tid, m - 1)];
      const float4 r0 = nd.pos[v[0]], r1 = nd.pos[v[1]], r2 = nd.pos[v[2]], r3 = nd.pos[v[3]];
      const uint32_t i = base + tid;
      if (i < m) sp[i] = r0;
      if (i + BLOCK < m) sp[i + BLOCK] = r1;
      if (i + 2 * BLOCK < m) sp[i + 2 * BLOCK] = r2;
      if (i + 3 * BLOCK < m) sp[i + 3 * BLOCK] = r3;
    }
  } else {
    const float4* __restrict__ src = D.lpos + n0;
    for (uint32_t base = 0; base < m; base += kBatch * BLOCK) {
      const uint32_t i = base + tid;
      const float4 r0 = src[min(i, m - 1)], r1 = src[min(i + BLOCK, m - 1)], r2 = src[min(i + 2 * BLOCK, m - 1)],
                   r3 = src[min(i + 3 * BLOCK, m - 1)];
      if (i < m) sp[i] = r0;
      if (i + BLOCK < m) sp[i + BLOCK] = r1;
      if (i + 2 * BLOCK < m) sp[i + 2 * BLOCK] = r2;
      if (i + 3 * BLOCK < m) sp[i + 3 * BLOCK] = r3;
    }
  }
  if (needRadius)
    for (uint32_t i = tid; i < m; i += BLOCK) srad[i] = D.lrad[n0 + i];
  for (uint32_t s = 0; s < L.nseg; ++s) {
    const uint32_t nc = L.seg[s].ncol;
    if (nc == 0) continue;
    const uint32_t* __restrict__ src = L.seg[s].colOff + static_cast<size_t>(g) * (nc + 1);
    for (uint32_t c = tid; c <= nc + 1; c += BLOCK) soff[s * kOffStride + c] = src[c <= nc ? c : nc];
  }
  __syncthreads();

  for (uint32_t s = 0; s < L.nseg; ++s) {
    const uint32_t kind = L.seg[s].kind, ncol = L.seg[s].ncol;
    const uint32_t* __restrict__ off = soff + s * kOffStride;
    if (kind == LAYER_TET) {
      // the record of the next colour is requested before this colour's SVD (unconditionally, from a clamped slot:
      // a conditional load would have to be waited for where the branches join)
      const uint32_t last = off[ncol] > off[0] ? off[ncol] - 1 : 0;  // a valid slot (the segment is not empty)
      uint32_t lo = off[0], hi = off[1];
      bool have = lo + tid < hi;
      uint32_t t0 = min(lo + tid, last);
      uint2 id = D.tc_lid[t0];
      float4 a0 = D.tc_q0[t0], a1 = D.tc_q1[t0], a2 = D.tc_q2[t0];
      for (uint32_t c = 0; c < ncol; ++c) {
        const uint32_t nlo = hi, nhi = off[c + 2];
        const bool nhave = nlo + tid < nhi;
        t0 = min(nlo + tid, last);
        const uint2 nid = D.tc_lid[t0];
        const float4 b0 = D.tc_q0[t0], b1 = D.tc_q1[t0], b2 = D.tc_q2[t0];
        if (have) {
          const uint32_t i1 = lo16(id.x), i2 = hi16(id.x), i3 = lo16(id.y), i4 = hi16(id.y);
          float4 x1 = sp[i1], x2 = sp[i2], x3 = sp[i3], x4 = sp[i4];
          tet_core<TETV>(x1, x2, x3, x4, a0, a1, a2);
          sp[i1] = x1; sp[i2] = x2; sp[i3] = x3; sp[i4] = x4;
        }
        for (uint32_t t = lo + tid + BLOCK; t < hi; t += BLOCK) {  // classes larger than the workgroup
          const uint2 jd = D.tc_lid[t];
          const uint32_t i1 = lo16(jd.x), i2 = hi16(jd.x), i3 = lo16(jd.y), i4 = hi16(jd.y);
          float4 x1 = sp[i1], x2 = sp[i2], x3 = sp[i3], x4 = sp[i4];
          tet_core<TETV>(x1, x2, x3, x4, D.tc_q0[t], D.tc_q1[t], D.tc_q2[t]);
          sp[i1] = x1; sp[i2] = x2; sp[i3] = x3; sp[i4] = x4;
        }
        lds_barrier();
        lo = nlo; hi = nhi; have = nhave; id = nid; a0 = b0; a1 = b1; a2 = b2;
      }
    } else if (kind == LAYER_DISTANCE) {
      // a projection is ~40 instructions, far shorter than a record's way from HBM: every lane requests its record
      // of each of the first kDistPreload colours at once and the latency is paid once per segment
      const uint32_t last = off[ncol] > off[0] ? off[ncol] - 1 : 0;
      uint32_t id[kDistPreload];
      float2 rw[kDistPreload];
#pragma unroll
      for (int c = 0; c < kDistPreload; ++c) {
        const uint32_t cc = min(static_cast<uint32_t>(c), ncol - 1);
        const uint32_t t = min(off[cc] + tid, last);
        id[c] = D.dc_lid[t];
        rw[c] = D.dc_rw[t];
      }
#pragma unroll
      for (int c = 0; c < kDistPreload; ++c) {
        if (static_cast<uint32_t>(c) < ncol) {
          if (off[c] + tid < off[c + 1]) {
            float4 a = sp[lo16(id[c])];
            distance_core(a, sp[hi16(id[c])], rw[c]);
            sp[lo16(id[c])] = a;
          }
          for (uint32_t t = off[c] + tid + BLOCK; t < off[c + 1]; t += BLOCK) {  // classes larger than the workgroup
            const uint32_t jd = D.dc_lid[t];
            float4 a = sp[lo16(jd)];
            distance_core(a, sp[hi16(jd)], D.dc_rw[t]);
            sp[lo16(jd)] = a;
          }
          lds_barrier();
        }
      }
      for (uint32_t c = kDistPreload; c < ncol; ++c) {
        for (uint32_t t = off[c] + tid; t < off[c + 1]; t += BLOCK) {
          const uint32_t jd = D.dc_lid[t];
          float4 a = sp[lo16(jd)];
          distance_core(a, sp[hi16(jd)], D.dc_rw[t]);
          sp[lo16(jd)] = a;
        }
        lds_barrier();
      }
    } else if (kind == LAYER_BEND) {
      for (uint32_t c = 0; c < ncol; ++c) {
        for (uint32_t t = off[c] + tid; t < off[c + 1]; t += BLOCK) {
          const uint2 jd = D.bc_lid[t];
          const uint32_t i1 = lo16(jd.x), i2 = hi16(jd.x), i3 = lo16(jd.y), i4 = hi16(jd.y);
          float4 x1 = sp[i1], x2 = sp[i2], x3 = sp[i3], x4 = sp[i4];
          if (bend_core(x1, x2, x3, x4, D.bc_aw[t])) { sp[i1] = x1; sp[i2] = x2; sp[i3] = x3; sp[i4] = x4; }
        }
        __syncthreads();
      }
    } else if (kind == LAYER_POSITION) {
      for (uint32_t c = 0; c < ncol; ++c) {
        for (uint32_t t = off[c] + tid; t < off[c + 1]; t += BLOCK) {
          const uint32_t i = D.pc_lid[t];
          float4 p = sp[i];
          position_core(p, D.pc_tw[t]);
          sp[i] = p;
        }
        __syncthreads();
      }
    } else if (kind == LAYER_FLOOR) {
      for (uint32_t i = tid; i < m; i += BLOCK) {
        float4 p = sp[i];
        if (floor_core(p, srad[i], P.floorHeight)) sp[i] = p;
      }
      __syncthreads();
    } else if (kind == LAYER_PREDICT) {
      for (uint32_t base = 0; base < m; base += kBatch * BLOCK) {
        uint32_t v[kBatch];
        float4 vel[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) v[k] = nodes[min(base + k * BLOCK + tid, m - 1)];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) vel[k] = nd.vel[v[k]];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
          const uint32_t i = base + k * BLOCK + tid;
          if (i < m) {
            float4 p = sp[i];
            nd.prev[v[k]] = make_float4(p.x, p.y, p.z, 0.0f);
            predict_core(p, vel[k], P.dt, P.gravity);
            sp[i] = p;
          }
        }
      }
      __syncthreads();
    } else if (kind == LAYER_VELOCITY) {
      for (uint32_t base = 0; base < m; base += kBatch * BLOCK) {
        uint32_t v[kBatch];
        float4 prev[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) v[k] = nodes[min(base + k * BLOCK + tid, m - 1)];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) prev[k] = nd.prev[v[k]];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
          const uint32_t i = base + k * BLOCK + tid;
          if (i < m) nd.vel[v[k]] = velocity_core(sp[i], prev[k], srad[i], P.dt, P.damping, P.friction, P.floorHeight);
        }
      }
    }
  }
  // node records out: to the layer-ordered copy, or scattered to the node array (last launch of the substep, before
  // a collision pass)
  if (L.storeGlobal) {
    for (uint32_t base = 0; base < m; base += kBatch * BLOCK) {
      uint32_t v[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) v[k] = nodes[min(base + k * BLOCK + tid, m - 1)];
#pragma unroll
      for (int k = 0; k < kBatch; ++k)
        if (base + k * BLOCK + tid < m) nd.pos[v[k]] = sp[base + k * BLOCK + tid];
    }
  } else {
    for (uint32_t i = tid; i < m; i += BLOCK) D.lpos[n0 + i] = sp[i];
  }
}

static size_t layer_lds_bytes(uint32_t maxGroupNodes) {
  return static_cast<size_t>(maxGroupNodes) * (sizeof(float4) + sizeof(float)) + kLayerMaxSegs * kOffStride * sizeof(uint32_t);
}

hipError_t layer_prepare(uint32_t maxGroupNodes) {
  const int bytes = static_cast<int>(layer_lds_bytes(maxGroupNodes));
  if (bytes <= 64 * 1024) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer<256, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer<512, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer<1024, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e;
}

void launch_layer(hipStream_t st, const NodeArrays& nd, const LayerData& D, const LayerLaunch& L, const LayerParams& P) {
  if (L.groups == 0 || L.nseg == 0) return;
  const size_t lds = layer_lds_bytes(D.maxGroupNodes);
  static const int variant = [] { const char* e = getenv("PIES_EXP_TET"); return e ? atoi(e) : 0; }();
  static const uint32_t forceBlock = [] { const char* e = getenv("PIES_LAYER_BLOCK"); return e ? (uint32_t)atoi(e) : 0u; }();
  const uint32_t want = forceBlock ? forceBlock : L.maxClass;
  if (variant == 1) { hipLaunchKernelGGL((k_layer<256, 1>), dim3(L.groups), dim3(256), lds, st, nd, D, L, P); return; }  // experiment: no SVD
  if (want <= 256) hipLaunchKernelGGL((k_layer<256, 0>), dim3(L.groups), dim3(256), lds, st, nd, D, L, P);
  else if (want <= 512) hipLaunchKernelGGL((k_layer<512, 0>), dim3(L.groups), dim3(512), lds, st, nd, D, L, P);
  else hipLaunchKernelGGL((k_layer<1024, 0>), dim3(L.groups), dim3(1024), lds, st, nd, D, L, P);
}

}  // namespace pies
