// Node-node collision broad phase and resolve for the PBD substep (reference: Include/Pies/SpatialHash.h,
// Src/Solver.cpp:81-130 and :877-901).
//
// Broad phase.  The reference keeps a parallel_flat_hash_map<CellId, vector<Node*>> that is cleared and
// rebuilt every solver iteration by 16 threads which each scan all nodes.  Here the grid is rebuilt on the
// device in five small kernels: (1) every node computes its cell range with the reference's NodeCompRange
// arithmetic and counts itself into an open-addressing table keyed by the exact cell id (no hash
// aliasing: full 63-bit key compare); (2) buckets get contiguous storage by bump allocation from the
// final counts; (3) nodes are filled in; (4) each bucket is sorted by node index -- which is exactly the
// bucket order of the reference (its insert threads scan the nodes in index order).  Only cells that
// were used are touched when the table is reset.
//
// Resolve.  The reference visits nodes 0..N-1 sequentially and, per node, every bucket of its cell range
// in dx,dy,dz order, resolving each overlapping pair immediately (positions and velocities of both
// nodes).  That loop is order dependent, so the device fixes a *documented* order that exposes
// parallelism (DESIGN.md "Node-node collisions"): nodes are grouped by the minimum cell of their range;
// groups whose minimum cells agree modulo 3 on every axis touch disjoint node sets (a range spans at
// most 2 cells per axis), so the 27 residue classes are 27 passes; inside a pass one wavefront owns one
// group and visits its nodes in ascending index.  Per visited node the wave walks the buckets exactly
// like the reference: 64 candidates at a time are tested in parallel, and hits are resolved one by one
// in lane (= bucket) order, re-testing the remaining candidates after each resolve because the visiting
// node has moved.  The per-pair arithmetic is the reference's, operation for operation.
#include <cstdint>
#include <cstdlib>

#include "cell_table.h"
#include "hash_kernels.h"

namespace pies {

constexpr int kBlock = 256;
constexpr uint32_t kMaxBucket = 2048;  // nodes overlapping one cell before the simulation is declared failed

static inline dim3 grid_for(uint32_t n) { return dim3((n + kBlock - 1) / kBlock); }

// ---- reset: only the slots the previous build used ------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_hash_reset(HashArrays H) {
  const uint32_t used = H.counters[0];
  for (uint32_t u = blockIdx.x * kBlock + threadIdx.x; u < used; u += gridDim.x * kBlock) {
    const uint32_t s = H.used[u];
    H.keys[s] = kEmpty;
    H.cnt[s] = 0;
    H.gcnt[s] = 0;
  }
}
__global__ void k_hash_zero(HashArrays H) {
  const uint32_t t = threadIdx.x;
  if (t < 3) H.counters[t] = 0;            // used, total entries, total grouped nodes
  if (t >= 4 && t < 4 + 27) H.counters[t] = 0;  // groups per pass      (counters[3] = sticky failure flag)
  if (t == kCounterTicket) H.counters[t] = 0;   // k_collide_flow's work queue
  if (t == kCounterEpoch) H.counters[t] += 1;   // completion stamps of earlier builds are stale by construction
}

// ---- count: NodeCompRange (Solver.cpp:877-901) + insertion into the cell table ---------------------
__global__ void __launch_bounds__(kBlock) k_hash_count(HashArrays H, const float4* __restrict__ pos, const float* __restrict__ radius,
                                                       uint32_t n, float scale) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float4 p = pos[i];
  const float R = (radius[i] + 0.5f) / scale;
  const float gx = p.x / scale - R, gy = p.y / scale - R, gz = p.z / scale - R;
  const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
  const float twoR = 2 * R;
  uint32_t lx = static_cast<uint32_t>(ceilf((gx - fx) + twoR));
  uint32_t ly = static_cast<uint32_t>(ceilf((gy - fy) + twoR));
  uint32_t lz = static_cast<uint32_t>(ceilf((gz - fz) + twoR));
  const bool finite = (fabsf(fx) < 1.0e6f) && (fabsf(fy) < 1.0e6f) && (fabsf(fz) < 1.0e6f);  // also false for NaN
  if (lx > 50 || ly > 50 || lz > 50) lx = ly = lz = 0;  // the reference returns an empty range (Solver.cpp:896-898)
  if (!finite || lx > 2 || ly > 2 || lz > 2) {          // outside what this build supports: latch the failure
    atomicOr(&H.counters[3], 1u);
    lx = ly = lz = 0;
  }
  const int mx = finite ? static_cast<int>(fx) : 0, my = finite ? static_cast<int>(fy) : 0, mz = finite ? static_cast<int>(fz) : 0;
  H.rng[i] = make_int4(mx, my, mz, static_cast<int>(lx | (ly << 8) | (lz << 16)));
  uint32_t e = 0;
  for (uint32_t dx = 0; dx < lx; ++dx)
    for (uint32_t dy = 0; dy < ly; ++dy)
      for (uint32_t dz = 0; dz < lz; ++dz, ++e) {
        const uint64_t key = pack_cell(mx + (int)dx, my + (int)dy, mz + (int)dz);
        uint32_t h = hash_cell(key, H.mask);
        uint32_t slot = 0xffffffffu;
        for (int probe = 0; probe < 4096; ++probe) {
          const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&H.keys[h]), kEmpty, key);
          if (old == kEmpty) {
            H.used[atomicAdd(&H.counters[0], 1u)] = h;
            slot = h;
            break;
          }
          if (old == key) { slot = h; break; }
          h = (h + 1) & H.mask;
        }
        if (slot == 0xffffffffu) { atomicOr(&H.counters[3], 2u); H.nodeSlot[i * 8 + e] = slot; continue; }
        atomicAdd(&H.cnt[slot], 1u);
        if (e == 0) atomicAdd(&H.gcnt[slot], 1u);  // this cell is the node's minimum cell
        H.nodeSlot[i * 8 + e] = slot;
      }
}

// ---- alloc: contiguous storage per bucket / per group, pass lists ----------------------------------
__global__ void __launch_bounds__(kBlock) k_hash_alloc(HashArrays H) {
  const uint32_t used = H.counters[0];
  for (uint32_t u = blockIdx.x * kBlock + threadIdx.x; u < used; u += gridDim.x * kBlock) {
    const uint32_t s = H.used[u];
    if (H.cnt[s] > kMaxBucket) atomicOr(&H.counters[3], 4u);  // runaway pile-up: latch, like Solver.cpp:741-755
    H.start[s] = atomicAdd(&H.counters[1], H.cnt[s]);
    H.fill[s] = 0;
    const uint32_t gc = H.gcnt[s];
    if (gc) {
      H.gstart[s] = atomicAdd(&H.counters[2], gc);
      H.gfill[s] = 0;
      const uint64_t key = H.keys[s];
      const int x = static_cast<int>((key >> 42) & 0x1fffff) - kCoordBias, y = static_cast<int>((key >> 21) & 0x1fffff) - kCoordBias,
                z = static_cast<int>(key & 0x1fffff) - kCoordBias;
      const uint32_t pass = static_cast<uint32_t>(mod3(x) + 3 * mod3(y) + 9 * mod3(z));
      H.passList[static_cast<size_t>(pass) * H.n + atomicAdd(&H.counters[4 + pass], 1u)] = s;
    }
  }
}

// ---- fill ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_hash_fill(HashArrays H, uint32_t n) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int4 rg = H.rng[i];
  const uint32_t lx = rg.w & 0xff, ly = (rg.w >> 8) & 0xff, lz = (rg.w >> 16) & 0xff;
  const uint32_t ne = lx * ly * lz;
  for (uint32_t e = 0; e < ne; ++e) {
    const uint32_t s = H.nodeSlot[i * 8 + e];
    if (s == 0xffffffffu) continue;
    H.bucket[H.start[s] + atomicAdd(&H.fill[s], 1u)] = i;
    if (e == 0) H.group[H.gstart[s] + atomicAdd(&H.gfill[s], 1u)] = i;
  }
}

// ---- sort: ascending node index inside every bucket and every group (rank sort, one wave per cell) ---
__global__ void __launch_bounds__(kBlock) k_hash_sort(HashArrays H) {
  if (H.counters[3]) return;  // failed: the host latches _simFailed
  const uint32_t used = H.counters[0];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = (gridDim.x * kBlock) >> 6;
  for (uint32_t u = wave; u < used; u += nwaves) {
    const uint32_t s = H.used[u];
    rank_sort(H.bucket, H.bucketSorted, H.start[s], H.cnt[s], lane);
    const uint32_t gc = H.gcnt[s];
    if (gc) rank_sort(H.group, H.groupSorted, H.gstart[s], gc, lane);
  }
}

// ---- resolve ---------------------------------------------------------------------------------------
// Node state is read and written through agent-scope relaxed atomics (L2-served, write-through): within a
// pass exactly one wavefront touches a given node, and that wavefront must see its own earlier writes.
PIES_DEV float ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PIES_DEV void st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PIES_DEV float bcast(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// One resolved pair (Solver.cpp:92-125).  Every argument is wave uniform: (pi, vi) is the visiting node, (pj, wj)
// the other node's state before the pair; `self` marks the node meeting itself (quirk Q3).  On return (oj, vj) is
// the other node's new state (for a self pair it has already been folded into pi, vi).
// The pair needs 15 IEEE divisions; instead of every lane computing all of them, lane t computes one (the three
// components of dir on lanes 0-2, then the twelve corrections `(coef * vec_k) * mass / wSum` on lanes 0-11) and the
// quotients are broadcast: the same operations on the same operands, so the same bits.
struct PairState {
  float pix, piy, piz, vix, viy, viz;
};
struct LaneRole {
  int k3;    // component this lane divides
  int kind;  // 0: node position, 1: other position, 2: node velocity, 3: other velocity
};
PIES_DEV float sel3(int k, float x, float y, float z) { return k == 0 ? x : (k == 1 ? y : z); }
PIES_DEV void resolve_pair(PairState& a, float imi, bool self, float hdx, float hdy, float hdz, float hdist, float hdisp, float himj,
                           float pjx, float pjy, float pjz, float wjx, float wjy, float wjz, float friction, float staticThreshold,
                           const LaneRole role, float& ojx, float& ojy, float& ojz, float& vjx, float& vjy, float& vjz) {
  float ux = 1.0f, uy = 0.0f, uz = 0.0f;
  if (hdist > 0.00001f) {
    const float quot = sel3(role.k3, hdx, hdy, hdz) / hdist;
    ux = bcast(quot, 0); uy = bcast(quot, 1); uz = bcast(quot, 2);
  }
  const float wSum = imi + himj;
  const float sa = 0.85f * -hdisp, sb = 0.85f * hdisp;
  // friction works on the velocities as they are before the pair
  vjx = self ? a.vix : wjx; vjy = self ? a.viy : wjy; vjz = self ? a.viz : wjz;
  const float rx = vjx - a.vix, ry = vjy - a.viy, rz = vjz - a.viz;
  const float rd = rx * ux + ry * uy + rz * uz;
  const float qx = rx - rd * ux, qy = ry - rd * uy, qz = rz - rd * uz;
  float fr = friction;
  if (staticThreshold > 0.0f)  // sqrt(x) < t is false for every t <= 0
    if (sqrtf(qx * qx + qy * qy + qz * qz) < staticThreshold) fr = 1.0f;
  const float vec = role.kind < 2 ? sel3(role.k3, ux, uy, uz) : sel3(role.k3, qx, qy, qz);
  const float coef = role.kind == 0 ? sa : (role.kind == 1 ? sb : (role.kind == 2 ? -fr : fr));
  const float mass = (role.kind & 1) ? himj : imi;
  const float corr = ((coef * vec) * mass) / wSum;
  // node.position += 0.85f * -disp * dir * node.invMass / wSum
  a.pix += bcast(corr, 0); a.piy += bcast(corr, 1); a.piz += bcast(corr, 2);
  // other.position += 0.85f * disp * dir * other.invMass / wSum   (other may be the node itself)
  ojx = self ? a.pix : pjx; ojy = self ? a.piy : pjy; ojz = self ? a.piz : pjz;
  ojx += bcast(corr, 3); ojy += bcast(corr, 4); ojz += bcast(corr, 5);
  a.vix += bcast(corr, 6); a.viy += bcast(corr, 7); a.viz += bcast(corr, 8);
  if (self) {
    a.pix = ojx; a.piy = ojy; a.piz = ojz;
    a.vix += bcast(corr, 9); a.viy += bcast(corr, 10); a.viz += bcast(corr, 11);
  } else {
    vjx += bcast(corr, 9); vjy += bcast(corr, 10); vjz += bcast(corr, 11);
  }
}

// Resolve of one group straight from global memory: every candidate's state is fetched again for every visiting
// node.  Only used for groups whose neighbourhood does not fit the LDS staging of k_collide (dense pile-ups).
PIES_DEV uint32_t collide_group_global(const HashArrays& H, float* pos, float* vel, const float* __restrict__ radius, uint32_t gslot,
                                       int lane, const LaneRole role, float friction, float staticThreshold) {
  uint32_t resolved = 0;
  const uint32_t gs = H.gstart[gslot], gc = H.gcnt[gslot];
  for (uint32_t k = 0; k < gc; ++k) {
    const uint32_t i = H.groupSorted[gs + k];
    PairState a = {ld(pos + 4 * i), ld(pos + 4 * i + 1), ld(pos + 4 * i + 2), ld(vel + 4 * i), ld(vel + 4 * i + 1), ld(vel + 4 * i + 2)};
    const float imi = ld(pos + 4 * i + 3);
    const float ri = radius[i];
    const int4 rg = H.rng[i];
    const uint32_t lx = rg.w & 0xff, ly = (rg.w >> 8) & 0xff, lz = (rg.w >> 16) & 0xff;
    for (uint32_t dx = 0; dx < lx; ++dx)
      for (uint32_t dy = 0; dy < ly; ++dy)
        for (uint32_t dz = 0; dz < lz; ++dz) {
          const uint32_t cs = find_cell(H.keys, H.mask, pack_cell(rg.x + (int)dx, rg.y + (int)dy, rg.z + (int)dz));
          if (cs == 0xffffffffu) continue;
          const uint32_t bs = H.start[cs], bc = H.cnt[cs];
          for (uint32_t base = 0; base < bc; base += 64) {
            const bool valid = base + lane < bc;
            const uint32_t j = valid ? H.bucketSorted[bs + base + lane] : 0xffffffffu;
            float pjx = 0.f, pjy = 0.f, pjz = 0.f, imj = 1.f, rj = 0.f, wjx = 0.f, wjy = 0.f, wjz = 0.f;
            if (valid) {  // candidate state up front: a resolve then needs no further loads
              pjx = ld(pos + 4 * j); pjy = ld(pos + 4 * j + 1); pjz = ld(pos + 4 * j + 2); imj = ld(pos + 4 * j + 3);
              wjx = ld(vel + 4 * j); wjy = ld(vel + 4 * j + 1); wjz = ld(vel + 4 * j + 2);
              rj = radius[j];
            }
            int cursor = 0;
            for (;;) {
              if (valid && j == i) { pjx = a.pix; pjy = a.piy; pjz = a.piz; }  // the self pair sees the node's current position
              const float ddx = pjx - a.pix, ddy = pjy - a.piy, ddz = pjz - a.piz;
              const float dist = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
              const float disp = ri + rj - dist;
              const bool hit = valid && lane >= cursor && disp > 0.0f;
              const unsigned long long m = __ballot(hit);
              if (m == 0ull) break;
              const int l = __builtin_ctzll(m);
              const uint32_t hj = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(j), l));
              float ojx, ojy, ojz, vjx, vjy, vjz;
              resolve_pair(a, imi, hj == i, bcast(ddx, l), bcast(ddy, l), bcast(ddz, l), bcast(dist, l), bcast(disp, l), bcast(imj, l),
                           bcast(pjx, l), bcast(pjy, l), bcast(pjz, l), bcast(wjx, l), bcast(wjy, l), bcast(wjz, l), friction,
                           staticThreshold, role, ojx, ojy, ojz, vjx, vjy, vjz);
              if (hj != i && lane == l) {
                st(pos + 4 * hj, ojx); st(pos + 4 * hj + 1, ojy); st(pos + 4 * hj + 2, ojz);
                st(vel + 4 * hj, vjx); st(vel + 4 * hj + 1, vjy); st(vel + 4 * hj + 2, vjz);
              }
              ++resolved;
              cursor = l + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are in L2 before its next loads
          }
        }
    if (lane == 0) {
      st(pos + 4 * i, a.pix); st(pos + 4 * i + 1, a.piy); st(pos + 4 * i + 2, a.piz);
      st(vel + 4 * i, a.vix); st(vel + 4 * i + 1, a.viy); st(vel + 4 * i + 2, a.viz);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  return resolved;
}

// LDS staging of one group's neighbourhood.  All nodes of a group share their minimum cell, so everything they can
// touch lies in the 2x2x2 cells above it: the distinct nodes of those buckets (typically ~100) are fetched ONCE
// into a wave-private LDS table (open addressing on the node index), the visiting order of collide_group_global is
// replayed on the LDS copies, and the touched nodes are written back at the end.  Same arithmetic, same order.
#ifndef PIES_COL_BLOCK
#define PIES_COL_BLOCK 128
#endif
constexpr int kColBlock = PIES_COL_BLOCK;            // wavefronts of a block, each with its own table
#ifndef PIES_COL_SLOTS
#define PIES_COL_SLOTS 512
#endif
constexpr uint32_t kColSlots = PIES_COL_SLOTS;       // table capacity per wavefront (power of two)
#ifndef PIES_COL_MAX_UNIQUE
#define PIES_COL_MAX_UNIQUE 384
#endif
constexpr uint32_t kColMaxUnique = PIES_COL_MAX_UNIQUE;   // live entries allowed: an interior group of BASELINE config 4 (spacing 0.9, cells of 2.0) sees 216-343 distinct nodes
constexpr uint32_t kColMaxEntries = 1024; // bucket entries of the 8 cells
constexpr uint32_t kColEmpty = 0xffffffffu, kColDirty = 0x80000000u;
constexpr uint32_t kColMaxSpins = 1u << 18;  // polls of one completion stamp before the wait is declared dead (~0.3 s)
struct ColTable {
  uint32_t key[kColSlots];  // node index | kColDirty
  float px[kColSlots], py[kColSlots], pz[kColSlots], im[kColSlots], vx[kColSlots], vy[kColSlots], vz[kColSlots], r[kColSlots];
  uint16_t ent[kColMaxEntries];  // table slot of every bucket entry, cell after cell
};
constexpr int kColSlotBits = kColSlots == 256 ? 8 : kColSlots == 512 ? 9 : kColSlots == 1024 ? 10 : 11;
static_assert((1u << kColSlotBits) == kColSlots, "kColSlots must be 256, 512, 1024 or 2048");
PIES_DEV uint32_t col_hash(uint32_t j) { return (j * 2654435761u) >> (32 - kColSlotBits); }
PIES_DEV uint32_t col_find(const uint32_t* key, uint32_t j) {  // j is present
  uint32_t h = col_hash(j);
  while ((key[h] & ~kColDirty) != j) h = (h + 1) & (kColSlots - 1);
  return h;
}

// One group on the staged path (or, for a dense neighbourhood, the unstaged one).  Node state is read and written
// through agent-scope (sc1) loads and stores: in k_collide_flow the previous owner of a node may be a wavefront
// of the same launch on another XCD.  Returns the number of resolved pairs.
PIES_DEV uint32_t collide_group(const HashArrays& H, ColTable& T, float* pos, float* vel, const float* __restrict__ radius, uint32_t gslot,
                                int lane, const LaneRole role, float friction, float staticThreshold, int forceGlobal) {
  uint32_t resolved = 0;
  const uint32_t gs = H.gstart[gslot], gc = H.gcnt[gslot];
  if (gc == 0) return 0;
  // ---- the 2x2x2 cells above the group's cell: lanes 0..7 look one up each ------------------------------
  const int4 rg0 = H.rng[H.groupSorted[gs]];
  uint32_t myStart = 0, myCnt = 0;
  if (lane < 8) {
    const uint32_t cs = find_cell(H.keys, H.mask, pack_cell(rg0.x + ((lane >> 2) & 1), rg0.y + ((lane >> 1) & 1), rg0.z + (lane & 1)));
    if (cs != 0xffffffffu) { myStart = H.start[cs]; myCnt = H.cnt[cs]; }
  }
  uint32_t cStart[8], cCnt[8], cOff[9];
  cOff[0] = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    cStart[c] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(myStart), c));
    cCnt[c] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(myCnt), c));
    cOff[c + 1] = cOff[c] + cCnt[c];
  }
  bool staged = !forceGlobal && cOff[8] <= kColMaxEntries;
  if (staged) {
    for (uint32_t t = lane; t < kColSlots; t += 64) T.key[t] = kColEmpty;
    uint32_t unique = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      for (uint32_t base = 0; base < cCnt[c] && staged; base += 64) {
        bool fresh = false;
        if (base + lane < cCnt[c]) {
          const uint32_t j = H.bucketSorted[cStart[c] + base + lane];
          uint32_t h = col_hash(j);
          for (;;) {  // at most kColMaxUnique + 64 live entries: the probe ends
            const uint32_t old = atomicCAS(&T.key[h], kColEmpty, j);
            if (old == kColEmpty) { fresh = true; break; }
            if (old == j) break;
            h = (h + 1) & (kColSlots - 1);
          }
          T.ent[cOff[c] + base + lane] = static_cast<uint16_t>(h);
        }
        unique += static_cast<uint32_t>(__popcll(__ballot(fresh)));
        if (unique > kColMaxUnique) staged = false;
      }
    }
  }
  if (!staged) {
    return collide_group_global(H, pos, vel, radius, gslot, lane, role, friction, staticThreshold);
  }
  for (uint32_t t = lane; t < kColSlots; t += 64) {
    const uint32_t j = T.key[t];
    if (j == kColEmpty) continue;
    T.px[t] = ld(pos + 4 * j); T.py[t] = ld(pos + 4 * j + 1); T.pz[t] = ld(pos + 4 * j + 2); T.im[t] = ld(pos + 4 * j + 3);
    T.vx[t] = ld(vel + 4 * j); T.vy[t] = ld(vel + 4 * j + 1); T.vz[t] = ld(vel + 4 * j + 2);
    T.r[t] = radius[j];
  }
  // ---- the visiting order of collide_group_global on the staged copies -----------------------------------
  for (uint32_t k = 0; k < gc; ++k) {
    const uint32_t i = H.groupSorted[gs + k];
    const uint32_t si = col_find(T.key, i);
    PairState a = {T.px[si], T.py[si], T.pz[si], T.vx[si], T.vy[si], T.vz[si]};
    const float imi = T.im[si], ri = T.r[si];
    const uint32_t rw = static_cast<uint32_t>(H.rng[i].w);
    const uint32_t lx = rw & 0xff, ly = (rw >> 8) & 0xff, lz = (rw >> 16) & 0xff;
    for (uint32_t dx = 0; dx < lx; ++dx)
      for (uint32_t dy = 0; dy < ly; ++dy)
        for (uint32_t dz = 0; dz < lz; ++dz) {
          const uint32_t c = dx * 4 + dy * 2 + dz;
          uint32_t off = 0, bc = 0;
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (c == static_cast<uint32_t>(q)) { off = cOff[q]; bc = cCnt[q]; }
          for (uint32_t base = 0; base < bc; base += 64) {
            const bool valid = base + lane < bc;
            const uint32_t sj = valid ? T.ent[off + base + lane] : 0u;
            const uint32_t j = valid ? (T.key[sj] & ~kColDirty) : 0xffffffffu;
            float pjx = 0.f, pjy = 0.f, pjz = 0.f, imj = 1.f, rj = 0.f, wjx = 0.f, wjy = 0.f, wjz = 0.f;
            if (valid) {
              pjx = T.px[sj]; pjy = T.py[sj]; pjz = T.pz[sj]; imj = T.im[sj];
              wjx = T.vx[sj]; wjy = T.vy[sj]; wjz = T.vz[sj];
              rj = T.r[sj];
            }
            int cursor = 0;
            for (;;) {
              if (valid && j == i) { pjx = a.pix; pjy = a.piy; pjz = a.piz; }  // the self pair sees the node's current position
              const float ddx = pjx - a.pix, ddy = pjy - a.piy, ddz = pjz - a.piz;
              const float dist = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
              const float disp = ri + rj - dist;
              const bool hit = valid && lane >= cursor && disp > 0.0f;
              const unsigned long long m = __ballot(hit);
              if (m == 0ull) break;
              const int l = __builtin_ctzll(m);
              const uint32_t hj = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(j), l));
              float ojx, ojy, ojz, vjx, vjy, vjz;
              resolve_pair(a, imi, hj == i, bcast(ddx, l), bcast(ddy, l), bcast(ddz, l), bcast(dist, l), bcast(disp, l), bcast(imj, l),
                           bcast(pjx, l), bcast(pjy, l), bcast(pjz, l), bcast(wjx, l), bcast(wjy, l), bcast(wjz, l), friction,
                           staticThreshold, role, ojx, ojy, ojz, vjx, vjy, vjz);
              if (hj != i && lane == l) {
                T.px[sj] = ojx; T.py[sj] = ojy; T.pz[sj] = ojz;
                T.vx[sj] = vjx; T.vy[sj] = vjy; T.vz[sj] = vjz;
                T.key[sj] = j | kColDirty;
              }
              ++resolved;
              cursor = l + 1;
            }
          }
        }
    if (lane == 0) {
      T.px[si] = a.pix; T.py[si] = a.piy; T.pz[si] = a.piz;
      T.vx[si] = a.vix; T.vy[si] = a.viy; T.vz[si] = a.viz;
      T.key[si] = i | kColDirty;
    }
  }
  // ---- write the touched nodes back --------------------------------------------------------------------
  for (uint32_t t = lane; t < kColSlots; t += 64) {
    const uint32_t kj = T.key[t];
    if (kj == kColEmpty || !(kj & kColDirty)) continue;
    const uint32_t j = kj & ~kColDirty;
    st(pos + 4 * j, T.px[t]); st(pos + 4 * j + 1, T.py[t]); st(pos + 4 * j + 2, T.pz[t]);
    st(vel + 4 * j, T.vx[t]); st(vel + 4 * j + 1, T.vy[t]); st(vel + 4 * j + 2, T.vz[t]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the group's stores have left the wavefront
  return resolved;
}

// The 27 residue classes as 27 launches: inside a launch no two groups share a node.
__global__ void __launch_bounds__(kColBlock) k_collide(HashArrays H, float4* pos4, float4* vel4, const float* __restrict__ radius,
                                                       uint32_t pass, float friction, float staticThreshold, int forceGlobal) {
  __shared__ ColTable tables[kColBlock / 64];
  ColTable& T = tables[threadIdx.x >> 6];
  float* pos = reinterpret_cast<float*>(pos4);
  float* vel = reinterpret_cast<float*>(vel4);
  const int lane = threadIdx.x & 63;
  const LaneRole role = {lane % 3, lane / 3};
  const uint32_t wave = (blockIdx.x * kColBlock + threadIdx.x) >> 6, nwaves = (gridDim.x * kColBlock) >> 6;
  if (H.counters[3]) return;  // failed: the host latches _simFailed
  const uint32_t ngroups = H.counters[4 + pass];
  uint32_t resolved = 0;  // statistics, one atomic per wave at the end (a per-pair atomic on one word serialises the chip)
  for (uint32_t g = wave; g < ngroups; g += nwaves)
    resolved += collide_group(H, T, pos, vel, radius, H.passList[static_cast<size_t>(pass) * H.n + g], lane, role, friction,
                              staticThreshold, forceGlobal);
  if (lane == 0 && resolved) atomicAdd(&H.counters[31], resolved);
}

// The same order of conflicting groups in ONE launch.  Groups are handed out through a ticket counter in pass-major
// order; before a wavefront touches its group it waits until every group of an earlier pass within two cells (the
// only ones that can share a node with it) has published its completion stamp.  A ticket's predecessors were all
// taken earlier by wavefronts that are resident and running, so the wait always ends; it is bounded anyway and a
// timeout latches the failure flag, which every wait loop polls.  Against the 27 launches this removes the barrier
// after each pass (a pass lasted as long as its slowest group, with ~1.7 wavefronts per SIMD).
PIES_DEV uint32_t ldu(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void __launch_bounds__(kColBlock) k_collide_flow(HashArrays H, float4* pos4, float4* vel4, const float* __restrict__ radius,
                                                            float friction, float staticThreshold, int forceGlobal) {
  __shared__ ColTable tables[kColBlock / 64];
  ColTable& T = tables[threadIdx.x >> 6];
  float* pos = reinterpret_cast<float*>(pos4);
  float* vel = reinterpret_cast<float*>(vel4);
  const int lane = threadIdx.x & 63;
  const LaneRole role = {lane % 3, lane / 3};
  if (H.counters[3]) return;
  const uint32_t epoch = H.counters[kCounterEpoch];
  // inclusive prefix of the groups per pass, lane p holding pass p
  uint32_t incl = lane < 27 ? H.counters[4 + lane] : 0u;
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const uint32_t v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 26));
  uint32_t resolved = 0;
  for (;;) {
    uint32_t ticket = 0;
    if (lane == 0) ticket = atomicAdd(&H.counters[kCounterTicket], 1u);
    ticket = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ticket)));
    if (ticket >= total) break;
    const uint32_t pass = static_cast<uint32_t>(__popcll(__ballot(lane < 27 && incl <= ticket)));
    const uint32_t before = pass ? static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), pass - 1)) : 0u;
    const uint32_t gslot = H.passList[static_cast<size_t>(pass) * H.n + (ticket - before)];
    // ---- wait for the conflicting groups of earlier passes -------------------------------------------------
    const uint64_t key = H.keys[gslot];
    const int x = static_cast<int>((key >> 42) & 0x1fffff) - kCoordBias, y = static_cast<int>((key >> 21) & 0x1fffff) - kCoordBias,
              z = static_cast<int>(key & 0x1fffff) - kCoordBias;
    bool gaveUp = false;
    for (int q = lane; q < 125; q += 64) {
      const int dx = q % 5 - 2, dy = (q / 5) % 5 - 2, dz = q / 25 - 2;
      if (dx == 0 && dy == 0 && dz == 0) continue;
      const uint32_t other = static_cast<uint32_t>(mod3(x + dx) + 3 * mod3(y + dy) + 9 * mod3(z + dz));
      if (other >= pass) continue;
      const uint32_t os = find_cell(H.keys, H.mask, pack_cell(x + dx, y + dy, z + dz));
      if (os == 0xffffffffu || H.gcnt[os] == 0u) continue;
      uint32_t spins = 0;
      while (ldu(H.done + os) != epoch) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 255u) == 0u && (spins > kColMaxSpins || ldu(H.counters + 3))) { gaveUp = true; break; }
      }
    }
    if (__ballot(gaveUp)) {  // a predecessor never finished (or the simulation failed elsewhere): latch and leave
      if (lane == 0) atomicOr(&H.counters[3], 8u);
      break;
    }
    resolved += collide_group(H, T, pos, vel, radius, gslot, lane, role, friction, staticThreshold, forceGlobal);
    if (lane == 0) __hip_atomic_store(H.done + gslot, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (lane == 0 && resolved) atomicAdd(&H.counters[31], resolved);
}

// ----------------------------------------------------------------------------------------------------
uint32_t launch_hash_build(hipStream_t st_, const HashArrays& H, const NodeArrays& nd, float scale) {
  if (nd.n == 0) return 0;
  const dim3 wide(std::min<uint32_t>(2048u, (H.capacity / 8 + kBlock - 1) / kBlock));
  hipLaunchKernelGGL(k_hash_reset, wide, dim3(kBlock), 0, st_, H);
  hipLaunchKernelGGL(k_hash_zero, dim3(1), dim3(64), 0, st_, H);
  hipLaunchKernelGGL(k_hash_count, grid_for(nd.n), dim3(kBlock), 0, st_, H, nd.pos, nd.radius, nd.n, scale);
  hipLaunchKernelGGL(k_hash_alloc, wide, dim3(kBlock), 0, st_, H);
  hipLaunchKernelGGL(k_hash_fill, grid_for(nd.n), dim3(kBlock), 0, st_, H, nd.n);
  hipLaunchKernelGGL(k_hash_sort, wide, dim3(kBlock), 0, st_, H);
  return 6;
}

uint32_t launch_collide(hipStream_t st_, const HashArrays& H, const NodeArrays& nd, float friction, float staticThreshold) {
  if (nd.n == 0) return 0;
  const dim3 grid(std::max<uint32_t>(1u, std::min<uint32_t>(2048u, (nd.n / 8 + 1) / 2)));
  auto flag = [](const char* name) { const char* e = std::getenv(name); return e && e[0] == '1' ? 1 : 0; };
  const int forceGlobal = flag("PIES_COLLIDE_GLOBAL");  // diagnostics, read when the substep is captured
  const int passes = flag("PIES_COLLIDE_PASSES");
  if (!passes) {
    hipLaunchKernelGGL(k_collide_flow, grid, dim3(kColBlock), 0, st_, H, nd.pos, nd.vel, nd.radius, friction, staticThreshold, forceGlobal);
    return 1;
  }
  for (uint32_t pass = 0; pass < 27; ++pass)
    hipLaunchKernelGGL(k_collide, grid, dim3(kColBlock), 0, st_, H, nd.pos, nd.vel, nd.radius, pass, friction, staticThreshold, forceGlobal);
  return 27;
}

}  // namespace pies
