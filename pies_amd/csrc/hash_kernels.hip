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

__global__ void __launch_bounds__(kBlock) k_collide(HashArrays H, float4* pos4, float4* vel4, const float* __restrict__ radius,
                                                    uint32_t pass, float friction, float staticThreshold) {
  float* pos = reinterpret_cast<float*>(pos4);
  float* vel = reinterpret_cast<float*>(vel4);
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = (gridDim.x * kBlock) >> 6;
  if (H.counters[3]) return;  // failed: the host latches _simFailed
  const uint32_t ngroups = H.counters[4 + pass];
  uint32_t resolved = 0;  // statistics, one atomic per wave at the end (a per-pair atomic on one word serialises the chip)
  for (uint32_t g = wave; g < ngroups; g += nwaves) {
    const uint32_t gslot = H.passList[static_cast<size_t>(pass) * H.n + g];
    const uint32_t gs = H.gstart[gslot], gc = H.gcnt[gslot];
    for (uint32_t k = 0; k < gc; ++k) {
      const uint32_t i = H.groupSorted[gs + k];
      float pix = ld(pos + 4 * i), piy = ld(pos + 4 * i + 1), piz = ld(pos + 4 * i + 2);
      const float imi = ld(pos + 4 * i + 3);
      float vix = ld(vel + 4 * i), viy = ld(vel + 4 * i + 1), viz = ld(vel + 4 * i + 2);
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
                if (valid && j == i) { pjx = pix; pjy = piy; pjz = piz; }  // the self pair sees the node's current position
                const float ddx = pjx - pix, ddy = pjy - piy, ddz = pjz - piz;
                const float dist = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
                const float disp = ri + rj - dist;
                const bool hit = valid && lane >= cursor && disp > 0.0f;
                const unsigned long long m = __ballot(hit);
                if (m == 0ull) break;
                const int l = __builtin_ctzll(m);
                // the hit pair, made wave uniform (Solver.cpp:92-125)
                const float hdx = bcast(ddx, l), hdy = bcast(ddy, l), hdz = bcast(ddz, l);
                const float hdist = bcast(dist, l), hdisp = bcast(disp, l), himj = bcast(imj, l);
                const uint32_t hj = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(j), l));
                float ux = 1.0f, uy = 0.0f, uz = 0.0f;
                if (hdist > 0.00001f) { ux = hdx / hdist; uy = hdy / hdist; uz = hdz / hdist; }
                const float wSum = imi + himj;
                const float sa = 0.85f * -hdisp, sb = 0.85f * hdisp;
                // node.position += 0.85f * -disp * dir * node.invMass / wSum
                pix += ((sa * ux) * imi) / wSum; piy += ((sa * uy) * imi) / wSum; piz += ((sa * uz) * imi) / wSum;
                // other.position += 0.85f * disp * dir * other.invMass / wSum   (other may be the node itself)
                float ojx = (hj == i) ? pix : bcast(pjx, l), ojy = (hj == i) ? piy : bcast(pjy, l), ojz = (hj == i) ? piz : bcast(pjz, l);
                ojx += ((sb * ux) * himj) / wSum; ojy += ((sb * uy) * himj) / wSum; ojz += ((sb * uz) * himj) / wSum;
                // friction on the velocities
                float vjx = bcast(wjx, l), vjy = bcast(wjy, l), vjz = bcast(wjz, l);
                if (hj == i) { vjx = vix; vjy = viy; vjz = viz; }
                const float rx = vjx - vix, ry = vjy - viy, rz = vjz - viz;
                const float rd = rx * ux + ry * uy + rz * uz;
                const float qx = rx - rd * ux, qy = ry - rd * uy, qz = rz - rd * uz;
                float fr = friction;
                if (sqrtf(qx * qx + qy * qy + qz * qz) < staticThreshold) fr = 1.0f;
                vix += ((-fr * qx) * imi) / wSum; viy += ((-fr * qy) * imi) / wSum; viz += ((-fr * qz) * imi) / wSum;
                if (hj == i) {
                  pix = ojx; piy = ojy; piz = ojz;
                  vix += ((fr * qx) * himj) / wSum; viy += ((fr * qy) * himj) / wSum; viz += ((fr * qz) * himj) / wSum;
                } else {
                  vjx += ((fr * qx) * himj) / wSum; vjy += ((fr * qy) * himj) / wSum; vjz += ((fr * qz) * himj) / wSum;
                  if (lane == l) {
                    st(pos + 4 * hj, ojx); st(pos + 4 * hj + 1, ojy); st(pos + 4 * hj + 2, ojz);
                    st(vel + 4 * hj, vjx); st(vel + 4 * hj + 1, vjy); st(vel + 4 * hj + 2, vjz);
                  }
                }
                ++resolved;
                cursor = l + 1;
              }
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are in L2 before its next loads
            }
          }
      if (lane == 0) {
        st(pos + 4 * i, pix); st(pos + 4 * i + 1, piy); st(pos + 4 * i + 2, piz);
        st(vel + 4 * i, vix); st(vel + 4 * i + 1, viy); st(vel + 4 * i + 2, viz);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
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
  const dim3 grid(std::max<uint32_t>(1u, std::min<uint32_t>(1024u, (nd.n / 8 + 3) / 4)));
  for (uint32_t pass = 0; pass < 27; ++pass)
    hipLaunchKernelGGL(k_collide, grid, dim3(kBlock), 0, st_, H, nd.pos, nd.vel, nd.radius, pass, friction, staticThreshold);
  return 27;
}

}  // namespace pies
