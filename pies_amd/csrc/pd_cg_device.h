// Device helpers shared by the CG kernels of the PD global step (pd_cg_kernels.hip: the two-launch iteration,
// pd_cg1_kernels.hip: one launch per iteration) and by the kernels that close a solve's statistics.
#pragma once
#include <cstdint>

#include "dev_math.h"
#include "pd_kernels.h"

namespace pies {

constexpr int kBlock = 256;

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
// (kAhead: contacts whose loads are requested together; the one-launch-per-iteration kernels, whose fetch is three gathers, take 1:
// twelve fetches in flight cost them 60 registers for a path a contact-light substep takes for a handful of rows)
// ACC: float, or double in the residual kernels (the row's contact terms are w = 1e4 times coordinates and cancel against the
// right-hand side like the elastic terms do: summed in fp32 with thousands of contacts the residual - and with it the solution -
// carried 2.6 x the deviation of the reference's own fp32 solve from the fp64 yardstick; tests/test_tri_collisions_gpu.py)
template <int kAhead = 4, class ACC = float, class Fetch> PIES_DEV void contact_row(const CgArrays& A, uint32_t i, Fetch fetch, ACC& sx, ACC& sy, ACC& sz) {
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
        sx = fma(static_cast<ACC>(coef), static_cast<ACC>(q[0]), sx); sy = fma(static_cast<ACC>(coef), static_cast<ACC>(q[1]), sy);
        sz = fma(static_cast<ACC>(coef), static_cast<ACC>(q[2]), sz);
      }
      return;
    }
  }
  const uint32_t ts = A.tIncStart[i];
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
          sx = fma(static_cast<ACC>(-kTriContactW), static_cast<ACC>(q[u][t][0]), sx);
          sy = fma(static_cast<ACC>(-kTriContactW), static_cast<ACC>(q[u][t][1]), sy);
          sz = fma(static_cast<ACC>(-kTriContactW), static_cast<ACC>(q[u][t][2]), sz);
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
  if (done && where == 0 && A.single) {  // (k_cg1_first left the norms in scal[18..20]: partI belongs to the next solve by now)
#pragma unroll
    for (int c = 0; c < 3; ++c) red[3 + c] = A.scal[18 + c];
  }
  else if (done && where == 0) block_reduce_partials<6>(A.partI, 9, A.nparts, red);
  else if (A.single) {  // one launch per iteration: {r.t, w.t, r.r} per workgroup, the residual norms in the last three
    const float* last = done ? A.part1[where - 1] : prevPartB;
    block_reduce_partials<3>(last + 6, 9, A.partCount[last == A.part1[1] ? 1 : 0], red + 3);
  }
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

// the LPR lanes of a row combine their partial sums (pairwise; every lane of the row ends with the total)
template <int LPR> PIES_DEV void row_combine(float& sx, float& sy, float& sz) {
#pragma unroll
  for (int off = LPR / 2; off >= 1; off >>= 1) {
    sx += __shfl_xor(sx, off, LPR);
    sy += __shfl_xor(sy, off, LPR);
    sz += __shfl_xor(sz, off, LPR);
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

}  // namespace pies
