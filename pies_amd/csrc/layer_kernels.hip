// Schedule LAYERED on gfx950: one workgroup sweeps one tile of constraints (layer_plan.cpp) with the tile's node
// records resident in LDS.
//
// A launch covers the tiles of one phase (they share no node).  The workgroup copies its node records (16 B each;
// two runs of the level-ordered copy, one per level) from HBM into LDS once, runs the launch's segments - phases of
// the distance / tetrahedral / bend / position containers colour after colour and, when a level is one tile, the
// per-node steps of the substep (predict, floor clamp, velocity) - and writes the records back once.  Between two colours stands a workgroup barrier (~0.1 us) where
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
constexpr int kDistPreloadMax = 12;  // colours of a distance segment whose records a lane requests up front (fewer in the 128-register variants)
constexpr int kBatch = 4;         // node records a lane requests before it consumes the first (the load phase is written out for 4)

#ifdef PIES_EXPERIMENTS  // in-kernel time stamps of a diagnostic build (tools/layer_timeline.py): never part of the product build
__device__ unsigned long long* g_layer_stamps = nullptr;  // [launch slot][tile][kStampsPerTile]
constexpr int kStampsPerTile = 128;
#define PIES_STAMP_DECL unsigned long long* stampBase = nullptr; int stampIdx = 0; \
  if (g_layer_stamps && tid == 0) stampBase = g_layer_stamps + (static_cast<size_t>(L.stampSlot) * 4096u + g) * kStampsPerTile;
#define PIES_STAMP() do { if (stampBase && stampIdx < kStampsPerTile) stampBase[stampIdx++] = __builtin_amdgcn_s_memtime(); } while (0)
#define PIES_STAMP_REAL() do { if (stampBase && stampIdx < kStampsPerTile) stampBase[stampIdx++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PIES_STAMP_DECL
#define PIES_STAMP()
#define PIES_STAMP_REAL()
#endif

PIES_DEV uint32_t lo16(uint32_t v) { return v & 0xFFFFu; }
PIES_DEV uint32_t hi16(uint32_t v) { return v >> 16; }

// WPE: wavefronts per SIMD the kernel is compiled for (the second argument of HIP's __launch_bounds__).  A launch with fewer tiles
// than compute units (config 2: 125) lasts as long as one wavefront's instruction stream and takes all the registers that shorten
// it (1: up to 256 at 256 / 512 threads); a launch with several tiles per compute unit (1M particles: 400, the unstructured beam) is
// bound by throughput and wants two 512-thread workgroups resident per compute unit: 4 wavefronts per SIMD, 128 registers.
template <int BLOCK, int TETV, int WPE = 1>
__global__ void __launch_bounds__(BLOCK, WPE) k_layer(NodeArrays nd, LayerData D, LayerLaunch L, LayerParams P) {
  constexpr int kDistPreload = (WPE > 1 || BLOCK > 512) ? 6 : kDistPreloadMax;
  constexpr int kDistAhead = 3;  // colours of a distance segment whose records are in flight
  extern __shared__ float4 lds[];
  float4* __restrict__ sp = lds;                                                  // node records of the group
  float* __restrict__ srad = reinterpret_cast<float*>(sp + D.maxGroupNodes);      // their radii
  uint32_t* __restrict__ soff = reinterpret_cast<uint32_t*>(srad + D.maxGroupNodes);  // colour offsets, per segment
  const uint32_t tid = threadIdx.x;
  // (the grid size and the phase's tile list come with the launch record: the kernel's first loads are then ONE line of its
  // arguments and the tile's descriptor - gridDim.x lives in the hidden arguments, another cache line, and D.tiles[L.phase] is a
  // pointer fetched through a pointer: each a dependent scalar round trip of ~0.2 us before the first node record is requested)
  const uint32_t g = xcd_block(blockIdx.x, L.groups);  // neighbouring groups (shared levels) meet in one XCD's L2
  const uint4 tile = L.tileList[g];
  const uint32_t m = tile.y + tile.w;
  if (m == 0) return;  // uniform: an empty tile
  if (!PIES_IN_BOUNDS(m <= D.maxGroupNodes, 10u)) return;  // (uniform) the tile's node records fit the LDS the launch asked for
  PIES_STAMP_DECL
  PIES_STAMP_REAL();
  PIES_STAMP();
  // LDS index -> position in the level-ordered node list (two runs: the tile's part of its two levels)
  auto lp = [&](uint32_t i) { return i < tile.y ? tile.x + i : tile.z + (i - tile.y); };

  // The prologue is ONE round trip to memory: every load that depends only on the tile descriptor - the colour offsets of the
  // launch's segments, the radii, the first four node records of every lane - is requested before anything is waited for, then
  // LDS is written.  (Rounds 2-5 ran node records -> radii -> colour offsets as three load-wait-store loops behind each other and
  // found out whether the radii are needed by walking the segment list with scalar loads: ~1 us of a 1.9-us prologue.)
  const bool needRadius = L.needRadius != 0;
  // (all six segment records are read unconditionally - unused ones hold ncol = 0, launch_layer sees to that - so that their
  // scalar loads are issued together instead of one dependent round trip per segment)
  uint32_t segCols[kLayerMaxSegs];
  const uint32_t* segOff[kLayerMaxSegs];
#pragma unroll
  for (int s = 0; s < kLayerMaxSegs; ++s) { segCols[s] = L.seg[s].ncol; segOff[s] = L.seg[s].colOff; }
  uint32_t offReg[kLayerMaxSegs];
  float radReg[kBatch];
  auto request_early = [&]() {  // (called inside both branches of the node load below: a branch waits for every load in flight)
#pragma unroll
    for (int s = 0; s < kLayerMaxSegs; ++s) {
      offReg[s] = 0u;
      const uint32_t nc = segCols[s];
      if (nc != 0 && tid <= nc + 1) offReg[s] = (segOff[s] + static_cast<size_t>(g) * (nc + 1))[min(tid, nc)];
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) radReg[k] = needRadius ? D.lrad[lp(min(k * BLOCK + tid, m - 1))] : 0.0f;
  };
  // Node records in: from the level-ordered copy (the tile's two runs are contiguous: coalesced) or, with one strip, in
  // the first launch of a substep / after a collision pass, gathered from the node array.  Four requests per lane are in flight before
  // the first is consumed (clamped indices keep the loads unconditional, so nothing waits at a branch join).
  // (the first four records of every lane outside the loop: a loop header waits for every load in flight)
  auto load4 = [&](uint32_t base, float4& r0, float4& r1, float4& r2, float4& r3) {
    const uint32_t i = base + tid;
    if (L.loadGlobal) {
      uint32_t v[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) v[k] = D.nodeList[lp(min(i + k * BLOCK, m - 1))];
      r0 = nd.pos[v[0]]; r1 = nd.pos[v[1]]; r2 = nd.pos[v[2]]; r3 = nd.pos[v[3]];
    } else {
      r0 = D.lpos[lp(min(i, m - 1))]; r1 = D.lpos[lp(min(i + BLOCK, m - 1))]; r2 = D.lpos[lp(min(i + 2 * BLOCK, m - 1))];
      r3 = D.lpos[lp(min(i + 3 * BLOCK, m - 1))];
    }
  };
  auto store4 = [&](uint32_t base, const float4& r0, const float4& r1, const float4& r2, const float4& r3) {
    const uint32_t i = base + tid;
    if (i < m) sp[i] = r0;
    if (i + BLOCK < m) sp[i + BLOCK] = r1;
    if (i + 2 * BLOCK < m) sp[i + 2 * BLOCK] = r2;
    if (i + 3 * BLOCK < m) sp[i + 3 * BLOCK] = r3;
  };
  {
    float4 r0, r1, r2, r3;
    if (L.loadGlobal) {  // (uniform)
      request_early();
      uint32_t v[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) v[k] = D.nodeList[lp(min(tid + k * BLOCK, m - 1))];
      r0 = nd.pos[v[0]]; r1 = nd.pos[v[1]]; r2 = nd.pos[v[2]]; r3 = nd.pos[v[3]];
    } else {
      request_early();
      r0 = D.lpos[lp(min(tid, m - 1))]; r1 = D.lpos[lp(min(tid + BLOCK, m - 1))]; r2 = D.lpos[lp(min(tid + 2 * BLOCK, m - 1))];
      r3 = D.lpos[lp(min(tid + 3 * BLOCK, m - 1))];
    }
    store4(0u, r0, r1, r2, r3);
  }
  for (uint32_t base = kBatch * BLOCK; base < m; base += kBatch * BLOCK) {  // (tiles of more than 4 x BLOCK nodes)
    float4 r0, r1, r2, r3;
    load4(base, r0, r1, r2, r3);
    store4(base, r0, r1, r2, r3);
  }
  if (needRadius) {
#pragma unroll
    for (int k = 0; k < kBatch; ++k)
      if (k * BLOCK + tid < m) srad[k * BLOCK + tid] = radReg[k];
    for (uint32_t i = kBatch * BLOCK + tid; i < m; i += BLOCK) srad[i] = D.lrad[lp(i)];  // (tiles of more than 4 x BLOCK nodes)
  }
#pragma unroll
  for (int s = 0; s < kLayerMaxSegs; ++s) {
    const uint32_t nc = segCols[s];
    if (nc != 0) {
      if (tid <= nc + 1) soff[s * kOffStride + tid] = offReg[s];
      for (uint32_t c = tid + BLOCK; c <= nc + 1; c += BLOCK) soff[s * kOffStride + c] = (segOff[s] + static_cast<size_t>(g) * (nc + 1))[min(c, nc)];
    }
  }
  __syncthreads();
  PIES_STAMP();

  for (uint32_t s = 0; s < L.nseg; ++s) {
    const uint32_t kind = L.seg[s].kind, ncol = L.seg[s].ncol;
    const uint32_t* __restrict__ off = soff + s * kOffStride;
    if (kind == LAYER_TET && (WPE > 1 || BLOCK > 512)) {
      // the 128-register variants: no record is held across a projection (the next colour's records in flight cost 14 registers,
      // and with four wavefronts per SIMD another wavefront's arithmetic covers the load)
      for (uint32_t c = 0; c < ncol; ++c) {
        for (uint32_t t = off[c] + tid; t < off[c + 1]; t += BLOCK) {
          const uint2 jd = D.tc_lid[t];
          const uint32_t i1 = lo16(jd.x), i2 = hi16(jd.x), i3 = lo16(jd.y), i4 = hi16(jd.y);
          float4 x1 = sp[i1], x2 = sp[i2], x3 = sp[i3], x4 = sp[i4];
          tet_core<TETV>(x1, x2, x3, x4, D.tc_q0[t], D.tc_q1[t], D.tc_q2[t]);
          sp[i1] = x1; sp[i2] = x2; sp[i3] = x3; sp[i4] = x4;
        }
        lds_barrier();
      }
    } else if (kind == LAYER_TET) {
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
        if (have && PIES_IN_BOUNDS(max(max(lo16(id.x), hi16(id.x)), max(lo16(id.y), hi16(id.y))) < m, 11u)) {
          const uint32_t i1 = lo16(id.x), i2 = hi16(id.x), i3 = lo16(id.y), i4 = hi16(id.y);
          float4 x1 = sp[i1], x2 = sp[i2], x3 = sp[i3], x4 = sp[i4];
#ifdef PIES_EXPERIMENTS
          if (stampBase) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PIES_STAMP(); }  // the gather has landed
#endif
          tet_core<TETV>(x1, x2, x3, x4, a0, a1, a2);
#ifdef PIES_EXPERIMENTS
          if (stampBase) { asm volatile("" : "+v"(x1.x), "+v"(x2.x), "+v"(x3.x), "+v"(x4.x)); PIES_STAMP(); }  // the projection is done
#endif
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
        PIES_STAMP();
        lo = nlo; hi = nhi; have = nhave; id = nid; a0 = b0; a1 = b1; a2 = b2;
      }
    } else if (kind == LAYER_DISTANCE) {
      // a projection is ~40 instructions, far shorter than a record's way from HBM: every lane keeps the records of kDistAhead
      // colours in flight - those of colour c + kDistAhead are requested when colour c begins - so the latency is paid once per
      // segment.  (Rounds 2-5 requested all kDistPreload colours at once: a tile's 36 KB of records are ~290 cache lines against
      // the ~64 a compute unit keeps in flight, and the first colour waited for most of them - 2.1 us against 0.48 for the others
      // in the in-kernel stamps; 1.3 us now.  2, 3, 4 and 6 ahead measure the same.)
      const uint32_t last = off[ncol] > off[0] ? off[ncol] - 1 : 0;
      uint32_t id[kDistPreload];
      float2 rw[kDistPreload];
#pragma unroll
      for (int c = 0; c < kDistPreload; ++c) { id[c] = 0u; rw[c] = make_float2(0.f, 0.f); }
#pragma unroll
      for (int c = 0; c < kDistAhead; ++c) {
        const uint32_t cc = min(static_cast<uint32_t>(c), ncol - 1);
        const uint32_t t = min(off[cc] + tid, last);
        id[c] = D.dc_lid[t];
        rw[c] = D.dc_rw[t];
      }
#pragma unroll
      for (int c = 0; c < kDistPreload; ++c) {
        if (static_cast<uint32_t>(c) < ncol) {
          if (c + kDistAhead < kDistPreload) {  // (compile time) the records of colour c + kDistAhead, requested kDistAhead colours ahead
            const uint32_t cc = min(static_cast<uint32_t>(c + kDistAhead), ncol - 1);
            const uint32_t t = min(off[cc] + tid, last);
            id[(c + kDistAhead) % kDistPreload] = D.dc_lid[t];
            rw[(c + kDistAhead) % kDistPreload] = D.dc_rw[t];
          }
          if (off[c] + tid < off[c + 1] && PIES_IN_BOUNDS(max(lo16(id[c]), hi16(id[c])) < m, 12u)) {
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
          PIES_STAMP();
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
          if (!PIES_IN_BOUNDS(i < m, 13u)) continue;
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
        for (int k = 0; k < kBatch; ++k) v[k] = D.nodeList[lp(min(base + k * BLOCK + tid, m - 1))];
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
        for (int k = 0; k < kBatch; ++k) v[k] = D.nodeList[lp(min(base + k * BLOCK + tid, m - 1))];
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
      for (int k = 0; k < kBatch; ++k) v[k] = D.nodeList[lp(min(base + k * BLOCK + tid, m - 1))];
#pragma unroll
      for (int k = 0; k < kBatch; ++k)
        if (base + k * BLOCK + tid < m) nd.pos[v[k]] = sp[base + k * BLOCK + tid];
    }
  } else {
    for (uint32_t i = tid; i < m; i += BLOCK) D.lpos[lp(i)] = sp[i];
  }
  PIES_STAMP();
  PIES_STAMP_REAL();
}

// ---- per-node steps over the level-ordered copy (bodies cut into strips) -----------------------------------------
constexpr int kNodeBlock = 256;
__global__ void __launch_bounds__(kNodeBlock) k_lpredict(NodeArrays nd, const uint32_t* __restrict__ nodeList, float4* __restrict__ lpos,
                                                         float dt, float g) {
  const uint32_t i = blockIdx.x * kNodeBlock + threadIdx.x;
  if (i >= nd.n) return;
  const uint32_t v = nodeList[i];
  float4 p = nd.pos[v];
  nd.prev[v] = make_float4(p.x, p.y, p.z, 0.0f);
  predict_core(p, nd.vel[v], dt, g);
  lpos[i] = p;
}
__global__ void __launch_bounds__(kNodeBlock) k_lvelocity(NodeArrays nd, const uint32_t* __restrict__ nodeList, const float4* __restrict__ lpos,
                                                          const float* __restrict__ lrad, LayerParams P) {
  const uint32_t i = blockIdx.x * kNodeBlock + threadIdx.x;
  if (i >= nd.n) return;
  const uint32_t v = nodeList[i];
  const float4 p = lpos[i];
  nd.vel[v] = velocity_core(p, nd.prev[v], lrad[i], P.dt, P.damping, P.friction, P.floorHeight);
  nd.pos[v] = p;
}
__global__ void __launch_bounds__(kNodeBlock) k_lfloor(float4* __restrict__ lpos, const float* __restrict__ lrad, uint32_t n, float floorHeight) {
  const uint32_t i = blockIdx.x * kNodeBlock + threadIdx.x;
  if (i >= n) return;
  float4 p = lpos[i];
  if (floor_core(p, lrad[i], floorHeight)) lpos[i] = p;
}
__global__ void __launch_bounds__(kNodeBlock) k_lposition(float4* __restrict__ lpos, const uint32_t* __restrict__ lid,
                                                          const float4* __restrict__ target_w, uint32_t start, uint32_t count) {
  const uint32_t t = blockIdx.x * kNodeBlock + threadIdx.x;
  if (t >= count) return;
  const uint32_t i = lid[start + t];
  float4 p = lpos[i];
  position_core(p, target_w[start + t]);
  lpos[i] = p;
}
template <bool TO_NODES>
__global__ void __launch_bounds__(kNodeBlock) k_lcopy(float4* __restrict__ pos, const uint32_t* __restrict__ nodeList, float4* __restrict__ lpos,
                                                      uint32_t n) {
  const uint32_t i = blockIdx.x * kNodeBlock + threadIdx.x;
  if (i >= n) return;
  if (TO_NODES) pos[nodeList[i]] = lpos[i];
  else lpos[i] = pos[nodeList[i]];
}
static dim3 node_grid(uint32_t n) { return dim3((n + kNodeBlock - 1) / kNodeBlock); }
void launch_lpredict(hipStream_t st, const NodeArrays& nd, const LayerData& D, const LayerParams& P) {
  if (nd.n) hipLaunchKernelGGL(k_lpredict, node_grid(nd.n), dim3(kNodeBlock), 0, st, nd, D.nodeList, D.lpos, P.dt, P.gravity);
}
void launch_lvelocity(hipStream_t st, const NodeArrays& nd, const LayerData& D, const LayerParams& P) {
  if (nd.n) hipLaunchKernelGGL(k_lvelocity, node_grid(nd.n), dim3(kNodeBlock), 0, st, nd, D.nodeList, D.lpos, D.lrad, P);
}
void launch_lfloor(hipStream_t st, const NodeArrays& nd, const LayerData& D, const LayerParams& P) {
  if (nd.n) hipLaunchKernelGGL(k_lfloor, node_grid(nd.n), dim3(kNodeBlock), 0, st, D.lpos, D.lrad, nd.n, P.floorHeight);
}
void launch_lposition(hipStream_t st, const LayerData& D, uint32_t start, uint32_t count) {
  if (count) hipLaunchKernelGGL(k_lposition, node_grid(count), dim3(kNodeBlock), 0, st, D.lpos, D.pc_lid, D.pc_tw, start, count);
}
void launch_lcopy(hipStream_t st, const NodeArrays& nd, const LayerData& D, bool toNodeArray) {
  if (nd.n == 0) return;
  if (toNodeArray) hipLaunchKernelGGL(k_lcopy<true>, node_grid(nd.n), dim3(kNodeBlock), 0, st, nd.pos, D.nodeList, D.lpos, nd.n);
  else hipLaunchKernelGGL(k_lcopy<false>, node_grid(nd.n), dim3(kNodeBlock), 0, st, nd.pos, D.nodeList, D.lpos, nd.n);
}

#ifdef PIES_EXPERIMENTS
extern "C" int pies_exp_layer_stamps(unsigned long long* deviceBuffer) {  // 64 x 4096 x 128 x 8 bytes, or nullptr
  return hipMemcpyToSymbol(HIP_SYMBOL(g_layer_stamps), &deviceBuffer, sizeof(deviceBuffer)) == hipSuccess ? 0 : 1;
}
#endif
PIES_BOUNDS_REPORT(layer)
static size_t layer_lds_bytes(uint32_t maxGroupNodes) {
  return static_cast<size_t>(maxGroupNodes) * (sizeof(float4) + sizeof(float)) + kLayerMaxSegs * kOffStride * sizeof(uint32_t);
}

hipError_t layer_prepare(uint32_t maxGroupNodes) {
  const int bytes = static_cast<int>(layer_lds_bytes(maxGroupNodes));
  if (bytes <= 64 * 1024) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer<256, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer<512, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer<512, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer<256, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer<1024, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e;
}

void launch_layer(hipStream_t st, const NodeArrays& nd, const LayerData& D, const LayerLaunch& L0, const LayerParams& P) {
  if (L0.groups == 0 || L0.nseg == 0) return;
  LayerLaunch L = L0;
  L.tileList = D.tiles[L.phase];
  for (uint32_t s = L.nseg; s < static_cast<uint32_t>(kLayerMaxSegs); ++s) L.seg[s] = LayerSeg{};  // (the kernel reads all six)
  L.needRadius = 0u;
  for (uint32_t s = 0; s < L.nseg; ++s) L.needRadius |= (L.seg[s].kind == LAYER_FLOOR || L.seg[s].kind == LAYER_VELOCITY) ? 1u : 0u;
  const size_t lds = layer_lds_bytes(D.maxGroupNodes);
#ifdef PIES_EXPERIMENTS  // timing experiments that change the work done: never part of the product build (build.py)
  const int skipMask = [] { const char* e = getenv("PIES_EXP_LAYER_SKIP"); return e ? atoi(e) : 0; }();
  if (skipMask) {
    L.nseg = 0;
    for (uint32_t s = 0; s < L0.nseg; ++s)
      if (!((skipMask >> L0.seg[s].kind) & 1)) L.seg[L.nseg++] = L0.seg[s];
  }
  {
    static unsigned int slot = 0;  // the launch's place in the stamp buffer (64 launches are kept)
    L.stampSlot = slot++ & 63u;
  }
  const int variant = [] { const char* e = getenv("PIES_EXP_TET"); return e ? atoi(e) : 0; }();
  if (variant == 1) { hipLaunchKernelGGL((k_layer<256, 1>), dim3(L.groups), dim3(256), lds, st, nd, D, L, P); return; }  // no SVD
#endif
  const uint32_t forceBlock = [] { const char* e = tuning_env("PIES_LAYER_BLOCK"); return e ? (uint32_t)atoi(e) : 0u; }();  // speed only
  // A workgroup as wide as the launch's largest colour class runs every colour in one round - what counts while the launch has
  // fewer tiles than the chip has compute units (config 2: 125).  With more tiles than compute units (strips: 400 at 1M particles)
  // throughput counts: a workgroup of 1 024 threads leaves the unit to itself and its barriers idle, two of 512 hide each other's
  // (measured at 1M particles: 162 substeps/s with the class-wide choice, 187 with 512 throughout, 174 with 256, 138 with 1 024).
  uint32_t want = forceBlock ? forceBlock : L.maxClass;
  if (!forceBlock && L.groups > 256u) want = std::min<uint32_t>(want, 512u);  // (256 compute units)
  const bool throughput = L.groups > 256u;  // several tiles per compute unit
  if (want <= 256 && throughput) hipLaunchKernelGGL((k_layer<256, 0, 4>), dim3(L.groups), dim3(256), lds, st, nd, D, L, P);
  else if (want <= 256) hipLaunchKernelGGL((k_layer<256, 0>), dim3(L.groups), dim3(256), lds, st, nd, D, L, P);
  else if (want <= 512 && throughput) hipLaunchKernelGGL((k_layer<512, 0, 4>), dim3(L.groups), dim3(512), lds, st, nd, D, L, P);
  else if (want <= 512) hipLaunchKernelGGL((k_layer<512, 0>), dim3(L.groups), dim3(512), lds, st, nd, D, L, P);
  else hipLaunchKernelGGL((k_layer<1024, 0>), dim3(L.groups), dim3(1024), lds, st, nd, D, L, P);
}

}  // namespace pies
