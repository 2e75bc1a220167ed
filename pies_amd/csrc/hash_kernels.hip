// Node-node collision broad phase and resolve for the PBD substep (reference: Include/Pies/SpatialHash.h,
// Src/Solver.cpp:81-130 and :877-901).
//
// Broad phase.  The reference keeps a parallel_flat_hash_map<CellId, vector<Node*>> that is cleared and rebuilt every
// solver iteration by 16 threads which each scan all nodes.  Here the grid is rebuilt on the device as a sort:
//   (1) k_grid_range   every node computes its cell range with the reference's NodeCompRange arithmetic (up to 50 cells
//                      per axis, like the reference) and the number of cells it overlaps; the bounding box of all
//                      ranges is reduced with one atomic per wavefront;
//   (2) prefix sum     of the per-node cell counts -> each node's first entry (three small kernels);
//   (3) k_grid_emit    one (cell key, node) entry per overlapped cell, node-major.  The key packs the cell's coordinates
//                      relative to the bounding box with just the bits the box needs, so a scene of 23 x 45 x 45 cells
//                      sorts on 17 bits;
//   (4) radix sort     least-significant-digit, 8 bits per pass, stable: histogram, prefix sum of the (digit, workgroup)
//                      counts, scatter with ballot-based ranks.  Only the passes the key width needs do work (the
//                      captured graph holds all eight; the rest exit at once).  Stable + node-major input = every
//                      bucket lists its nodes in ascending index, which is the reference's bucket order (its insert
//                      threads scan the nodes in index order);
//   (5) k_grid_cells   bucket boundaries from neighbouring keys; each bucket's [start, end) goes into an exact-key
//                      open-addressing index (full 64-bit compare: two cells never alias, the reference's map is exact
//                      as well) that the resolve uses to find a cell's bucket with one or two probes.
//
// Resolve.  Two orders:
//  * k_collide_reference - the reference's loop as it stands (Solver.cpp:85-130): nodes 0..N-1 in ascending index, each
//    querying the cell range of its CURRENT position (SpatialHash.h:101-127), buckets in dx,dy,dz order, every overlapping
//    pair resolved at once (positions and velocities of both nodes, the self pair included).  The loop is one dependent
//    chain, so it runs on one wavefront: 64 candidates are tested at a time and hits are resolved one by one in bucket
//    order, re-testing the rest after each resolve because the visiting node has moved.  Bit-identical to the loop; slow.
//  * k_collide_flow - a documented order that exposes parallelism (DESIGN.md "Node-node collisions"): nodes are grouped
//    by the minimum cell of their range; groups whose minimum cells agree modulo 3 on every axis touch disjoint node
//    sets (needs ranges of at most 2 cells per axis), so the 27 residue classes are 27 passes; inside a pass one
//    wavefront owns one group and visits its nodes in ascending index, each with the range it was inserted with.
// The per-pair arithmetic is the reference's, operation for operation, in both.
#include <climits>
#include <cstdint>
#include <cstdlib>

#include "cell_table.h"
#include "hash_kernels.h"
#include "hash_device.h"

namespace pies {

constexpr int kBlock = 256;

static inline dim3 grid_for(uint32_t n) { return dim3((n + kBlock - 1) / kBlock); }

// ---- reset: only the index slots the previous build used; counters; bounding box --------------------------------
__global__ void __launch_bounds__(kBlock) k_grid_reset(HashArrays H) {
  const uint32_t used = H.counters[kCounterUsed];
  for (uint32_t u = blockIdx.x * kBlock + threadIdx.x; u < used; u += gridDim.x * kBlock) {
    const uint32_t s = H.used[u];
    H.keys[s] = kEmpty;
  }
}
// ---- range: NodeCompRange + cells per node + bounding box ---------------------------------------------------------
PIES_DEV int wave_min(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
PIES_DEV int wave_max(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}
__global__ void __launch_bounds__(kBlock) k_grid_range(HashArrays H, const float4* __restrict__ pos, const float* __restrict__ radius,
                                                       uint32_t n, float scale) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (blockIdx.x == 0) {  // the build's counters (k_grid_zero's work: nothing in this launch reads them; k_grid_reset, which needs the
    const uint32_t t = threadIdx.x;  // previous build's cell count, runs before it)
    if (t == kCounterUsed || t == kCounterEntries || t == 2 || t == kCounterDense) H.counters[t] = 0;
    if (t >= kCounterPass0 && t < kCounterPass0 + 27) H.counters[t] = 0;
    if (t == kCounterTicket) H.counters[t] = 0;
    if (t == kCounterEpoch) H.counters[t] += 1;
  }
  int mx = 0, my = 0, mz = 0;
  uint32_t lx = 0, ly = 0, lz = 0;
  if (i < n) {
    const float4 p = pos[i];
    if (!node_range(p.x, p.y, p.z, radius[i], scale, mx, my, mz, lx, ly, lz)) atomicOr(&H.counters[kCounterFlags], 1u);
    H.rng[i] = make_int4(mx, my, mz, static_cast<int>(lx | (ly << 8) | (lz << 16)));
    H.entCount[i] = lx * ly * lz;
  }
  if (i == n) H.entCount[n] = 0;  // so that the exclusive prefix sum ends with the total
  const bool have = i < n && lx * ly * lz != 0u;
  int lo[3] = {have ? mx : INT_MAX, have ? my : INT_MAX, have ? mz : INT_MAX};
  int hi[3] = {have ? mx + static_cast<int>(lx) - 1 : INT_MIN, have ? my + static_cast<int>(ly) - 1 : INT_MIN,
               have ? mz + static_cast<int>(lz) - 1 : INT_MIN};
  // bounding box of the workgroup's ranges -> boxPart[workgroup]; k_grid_box reduces the partial boxes (six global
  // atomics per wavefront on six words took 0.5 ms at 500k nodes)
  __shared__ int part[4][6];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = wave_min(lo[a]);
    hi[a] = wave_max(hi[a]);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { part[threadIdx.x >> 6][a] = lo[a]; part[threadIdx.x >> 6][3 + a] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    int v = part[0][a];
    for (int w = 1; w < 4; ++w) v = a < 3 ? min(v, part[w][a]) : max(v, part[w][a]);
    H.boxPart[blockIdx.x * 6 + a] = v;
  }
}
// ---- exclusive prefix sum of uint32 (tile sums, scan of the sums, add) -----------------------------------------
constexpr uint32_t kScanTile = 2048;  // 256 threads x 8
PIES_DEV uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds, uint32_t& total) {  // 256 threads; lds: 8 words
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off, 64);
    if (lane >= static_cast<uint32_t>(off)) incl += t;
  }
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < wave; ++w) base += lds[w];
  total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return base + incl - v;
}
// the bounding box of the nodes' cell ranges from the workgroups' partial boxes, and the sort this build runs (k_grid_box's
// work for a workgroup of kBlock threads: it rides behind the tiles of k_scan_tiles, one launch less per rebuild)
PIES_DEV void grid_box_reduce(const HashArrays& H, uint32_t nparts, uint32_t sortPasses) {
  __shared__ int red[kBlock / 64][6];
  int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
  for (uint32_t b = threadIdx.x; b < nparts; b += kBlock) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = min(lo[a], H.boxPart[b * 6 + a]);
      hi[a] = max(hi[a], H.boxPart[b * 6 + 3 + a]);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = wave_min(lo[a]);
    hi[a] = wave_max(hi[a]);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { red[threadIdx.x >> 6][a] = lo[a]; red[threadIdx.x >> 6][3 + a] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    int v = red[0][a];
    for (int w = 1; w < kBlock / 64; ++w) v = a < 3 ? min(v, red[w][a]) : max(v, red[w][a]);
    H.counters[(a < 3 ? kCounterBoxMin : kCounterBoxMax - 3) + a] = static_cast<uint32_t>(v);
  }
  if (threadIdx.x == 0) {  // the sort of this build: the captured passes, and whether they can hold the box's key
    H.counters[kCounterSortPasses] = sortPasses;
    uint32_t total = 0;
    bool empty = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int lo_ = red[0][a], hi_ = red[0][3 + a];
      for (int w = 1; w < kBlock / 64; ++w) { lo_ = min(lo_, red[w][a]); hi_ = max(hi_, red[w][3 + a]); }
      if (hi_ < lo_) empty = true;
      const uint32_t ext = empty ? 0u : static_cast<uint32_t>(hi_ - lo_);
      total += ext ? 32u - static_cast<uint32_t>(__builtin_clz(ext)) : 0u;
    }
    if (!empty && total > kRadixMaxDigit * sortPasses) atomicOr(&H.counters[kCounterFlags], 512u);
  }
}
__global__ void __launch_bounds__(kBlock) k_scan_tiles(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n,
                                                       uint32_t* __restrict__ sums, HashArrays H, uint32_t boxParts, uint32_t sortPasses) {
  if (boxParts && blockIdx.x + 1u == gridDim.x) {  // (uniform per workgroup)
    grid_box_reduce(H, boxParts, sortPasses);
    return;
  }
  __shared__ uint32_t lds[8];
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 8u;
  uint32_t v[8], s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v[k] = base + k < n ? in[base + k] : 0u;
    s += v[k];
  }
  uint32_t total;
  uint32_t run = block_exclusive_scan(s, lds, total);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(1024) k_scan_sums(uint32_t* __restrict__ sums, uint32_t m) {  // one workgroup, in place
  __shared__ uint32_t part[1024];
  const uint32_t t = threadIdx.x, per = (m + 1023u) / 1024u;
  const uint32_t lo = min(m, t * per), hi = min(m, lo + per);
  uint32_t s = 0;
  for (uint32_t k = lo; k < hi; ++k) s += sums[k];
  part[t] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partial sums
    const uint32_t a = t >= off ? part[t - off] : 0u;
    __syncthreads();
    part[t] += a;
    __syncthreads();
  }
  uint32_t run = part[t] - s;
  for (uint32_t k = lo; k < hi; ++k) {
    const uint32_t v = sums[k];
    sums[k] = run;
    run += v;
  }
}
// ---- emit: one (cell key, node) entry per overlapped cell, node-major --------------------------------------------
// A wavefront's 64 nodes own one contiguous stretch of the entry list (entOff is node-major).  The lanes write it slot by slot -
// lane l takes slots l, l + 64, ... of the stretch and finds the node a slot belongs to by a binary search over the lanes'
// offsets - so that a store instruction covers 64 consecutive entries.  (One lane writing its own node's entries one after the
// other put 64 separate 8-byte pieces into every store: 33 us for 4 M entries.)
__global__ void __launch_bounds__(kBlock) k_grid_emit(HashArrays H, uint32_t n) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (i == 0) {
    const uint32_t total = H.entOff[n] + H.scanSums[n / kScanTile];
    if (total > H.maxEntries) atomicOr(&H.counters[kCounterFlags], 128u);
    H.counters[kCounterEntries] = min(total, H.maxEntries);
  }
  const bool live = i < n;
  const int4 rg = live ? H.rng[i] : make_int4(0, 0, 0, 0);
  const uint32_t lx = rg.w & 0xff, ly = (rg.w >> 8) & 0xff, lz = (rg.w >> 16) & 0xff;
  const uint32_t base = live ? H.entOff[i] + H.scanSums[i / kScanTile] : 0u;  // (the tile sums are added here: k_scan_add's launch)
  uint32_t cnt = live ? lx * ly * lz : 0u;
  if (base + cnt > H.maxEntries) cnt = 0;  // flagged above: the host latches the failure
  const GridBox B = grid_box(H.counters);
  // offsets of the lanes' nodes inside the wavefront's stretch
  uint32_t incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  const uint32_t pre = incl - cnt, total = __shfl(incl, 63, 64);
  const uint32_t waveBase = __shfl(base, 0, 64);  // (lane 0 is live whenever any lane of the wavefront is)
  for (uint32_t j0 = 0; j0 < total; j0 += 64) {
    const uint32_t j = j0 + static_cast<uint32_t>(lane);
    // the node slot j belongs to: the last lane whose offset is <= j (lanes without entries share their offset with the next)
    uint32_t l = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      const uint32_t pm = __shfl(pre, static_cast<int>(l) + step, 64);
      if (pm <= j) l += static_cast<uint32_t>(step);
    }
    const uint32_t pl = __shfl(pre, static_cast<int>(l), 64);
    const int ox = __shfl(rg.x, static_cast<int>(l), 64), oy = __shfl(rg.y, static_cast<int>(l), 64), oz = __shfl(rg.z, static_cast<int>(l), 64);
    const uint32_t ow = static_cast<uint32_t>(__shfl(rg.w, static_cast<int>(l), 64));
    const uint32_t ob = __shfl(base, static_cast<int>(l), 64);
    if (j >= total) continue;
    const uint32_t oly = (ow >> 8) & 0xff, olz = (ow >> 16) & 0xff, olx = ow & 0xff;
    const uint32_t e = j - pl, lyz = oly * olz;
    const uint32_t dx = e / lyz, r = e - dx * lyz, dy = r / olz, dz = r - dy * olz;
    const uint32_t node = blockIdx.x * kBlock + (threadIdx.x & ~63u) + l;
    const uint32_t side = (dx ? 4u : 0u) | (dy ? 2u : 0u) | (dz ? 1u : 0u);
    const uint32_t twoLong = (olx == 2u ? 4u : 0u) | (oly == 2u ? 2u : 0u) | (olz == 2u ? 1u : 0u);
    const uint64_t key = box_key(B, ox + static_cast<int>(dx), oy + static_cast<int>(dy), oz + static_cast<int>(dz));
    const uint32_t val = node | (twoLong << kLongShift) | (side << kSideShift) | (e == 0 ? kMinFlag : 0u);  // e == 0: the node's minimum cell
    // A key of at most 32 bits (any scene but one that spans billions of cells) travels through the sort in one word with its
    // value: 8 bytes per entry and pass instead of 8 + 4.  The last pass writes the values out for the kernels that read them.
    if (B.packed) H.key[0][ob + e] = (key << 32) | val;
    else { H.key[0][ob + e] = key; H.val[0][ob + e] = val; }
  }
  (void)waveBase;
}

// ---- radix sort of the entries by key: 8 bits per pass, stable ------------------------------------------------
// Pass p reads buffer p & 1 and writes the other one.  A workgroup owns kRadixTile consecutive entries; wavefront w of
// it the w-th quarter, sixteen rounds of 64 consecutive entries - so (workgroup, wavefront, round, lane) is the input order.
__global__ void __launch_bounds__(kBlock) k_radix_hist(HashArrays H, uint32_t pass, uint32_t nblkMax) {
  __shared__ uint32_t h[1u << kRadixMaxDigit];
  const GridBox B = grid_box(H.counters);
  if (pass >= grid_passes(B)) return;
  const uint32_t bins = 1u << B.digit, shift = B.digit * pass + (B.packed ? 32u : 0u);
  const uint32_t E = H.counters[kCounterEntries];
  const uint32_t blk = blockIdx.x;
  if (blk * kRadixTile >= E) return;
  const uint64_t* __restrict__ src = H.key[pass & 1u];
  for (uint32_t d = threadIdx.x; d < bins; d += kBlock) h[d] = 0;
  __syncthreads();
  uint64_t kk[kRadixTile / kBlock];  // all sixteen loads in flight before the first LDS atomic
#pragma unroll
  for (uint32_t k = 0; k < kRadixTile / kBlock; ++k) {
    const uint32_t t = blk * kRadixTile + k * kBlock + threadIdx.x;
    kk[k] = t < E ? src[t] : ~0ull;
  }
#pragma unroll
  for (uint32_t k = 0; k < kRadixTile / kBlock; ++k) {
    const uint32_t t = blk * kRadixTile + k * kBlock + threadIdx.x;
    if (t < E) atomicAdd(&h[static_cast<uint32_t>(kk[k] >> shift) & (bins - 1u)], 1u);
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < bins; d += kBlock) H.hist[d * nblkMax + blk] = h[d];
}
// Prefix sums of the (digit, workgroup) counts in digit-major order, in two steps: workgroup d of this kernel turns row d
// (the counts of digit d over the sorting workgroups) into its exclusive prefix and leaves the row's total in
// hist[256 * nblkMax + d]; the scatter kernel adds the exclusive prefix of the 256 totals.
__global__ void __launch_bounds__(kBlock) k_radix_scan(HashArrays H, uint32_t pass, uint32_t nblkMax) {
  __shared__ uint32_t lds[8];
  const GridBox B = grid_box(H.counters);
  if (pass >= grid_passes(B) || blockIdx.x >= (1u << B.digit)) return;
  const uint32_t E = H.counters[kCounterEntries];
  const uint32_t nblk = (E + kRadixTile - 1u) / kRadixTile;
  uint32_t* __restrict__ row = H.hist + static_cast<size_t>(blockIdx.x) * nblkMax;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nblk; base += kBlock) {
    const uint32_t k = base + threadIdx.x;
    const uint32_t v = k < nblk ? row[k] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(v, lds, total);
    if (k < nblk) row[k] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) H.hist[(1u << kRadixMaxDigit) * nblkMax + blockIdx.x] = carry;
}
__global__ void __launch_bounds__(kBlock) k_radix_scatter(HashArrays H, uint32_t pass, uint32_t nblkMax) {
  constexpr uint32_t kBins = 1u << kRadixMaxDigit;
  __shared__ uint32_t cnt[4][kBins];  // per wavefront: running count of a digit, then the wavefront's base inside the workgroup
  __shared__ uint32_t gbase[kBins];
  __shared__ uint32_t scanLds[8];
  const GridBox B = grid_box(H.counters);
  if (pass >= grid_passes(B)) return;
  const uint32_t bins = 1u << B.digit, shift = B.digit * pass + (B.packed ? 32u : 0u);
  const bool packed = B.packed, last = pass + 1u == grid_passes(B);
  const uint32_t E = H.counters[kCounterEntries];
  const uint32_t blk = blockIdx.x;
  if (blk * kRadixTile >= E) return;
  const uint64_t* __restrict__ skey = H.key[pass & 1u];
  const uint32_t* __restrict__ sval = H.val[pass & 1u];
  uint64_t* __restrict__ dkey = H.key[(pass & 1u) ^ 1u];
  uint32_t* __restrict__ dval = H.val[(pass & 1u) ^ 1u];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  constexpr int kRounds = kRadixTile / kBlock;  // 16
  for (uint32_t d = lane; d < bins; d += 64u) cnt[wave][d] = 0;
  __builtin_amdgcn_wave_barrier();
  uint64_t key[kRounds];
  uint32_t val[kRounds], rank[kRounds];
  const uint32_t first = blk * kRadixTile + wave * (kRadixTile / 4u);
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const uint32_t t = first + static_cast<uint32_t>(r) * 64u + lane;
    const bool valid = t < E;
    key[r] = valid ? skey[t] : 0ull;
    val[r] = (valid && !packed) ? sval[t] : 0u;
  }
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const uint32_t t = first + static_cast<uint32_t>(r) * 64u + lane;
    const bool valid = t < E;
    const uint32_t d = static_cast<uint32_t>(key[r] >> shift) & (bins - 1u);
    unsigned long long peers = __ballot(valid);  // lanes of this round holding the same digit
#pragma unroll
    for (int b = 0; b < static_cast<int>(kRadixMaxDigit); ++b) {
      if (static_cast<uint32_t>(b) < B.digit) {  // (uniform)
        const bool bit = (d >> b) & 1u;
        const unsigned long long m = __ballot(valid && bit);
        peers &= bit ? m : ~m;
      }
    }
    rank[r] = 0;
    if (valid) {
      const uint32_t before = cnt[wave][d];
      rank[r] = before + static_cast<uint32_t>(__popcll(peers & ((1ull << lane) - 1ull)));
      if (static_cast<uint32_t>(__builtin_ctzll(peers)) == lane) cnt[wave][d] = before + static_cast<uint32_t>(__popcll(peers));
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {  // per digit: the wavefronts' bases inside the workgroup, and the workgroup's base in the output.  A thread takes
    // kBins / kBlock consecutive digits (their totals scanned in the thread, the threads' sums across the workgroup)
    constexpr uint32_t kPer = kBins / kBlock;  // 8
    const uint32_t d0 = threadIdx.x * kPer;
    uint32_t tot[kPer], mine = 0;
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
      const uint32_t d = d0 + q;
      tot[q] = d < bins ? H.hist[kBins * nblkMax + d] : 0u;
      mine += tot[q];
      if (d < bins) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint32_t c = cnt[w][d];
          cnt[w][d] = run;
          run += c;
        }
      }
    }
    uint32_t all;
    uint32_t digitBase = block_exclusive_scan(mine, scanLds, all);  // entries with a smaller digit
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
      const uint32_t d = d0 + q;
      if (d < bins) gbase[d] = digitBase + H.hist[d * nblkMax + blk];
      digitBase += tot[q];
    }
  }
  __syncthreads();
  // (Round 4 measured the tile put into sorted order in LDS first and written out from there, so that a store instruction
  // covers the runs of a few digits instead of 64 entries in 64 cache lines: 41.8 us per pass against 36.5 - the pass is not
  // bound by its store pattern, and the 32 KB of staging cost a resident workgroup per CU.)
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const uint32_t t = first + static_cast<uint32_t>(r) * 64u + lane;
    if (t < E) {
      const uint32_t d = static_cast<uint32_t>(key[r] >> shift) & (bins - 1u);
      const uint32_t at = gbase[d] + cnt[wave][d] + rank[r];
      dkey[at] = key[r];
      if (!packed) dval[at] = val[r];
      else if (last) dval[at] = static_cast<uint32_t>(key[r]);  // (the values, for the kernels behind the sort)
    }
  }
}

// ---- cells: bucket boundaries -> cell index; groups per pass ------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_grid_cells(HashArrays H) {
  // a workgroup takes kRadixTile consecutive entries; the cells it creates are collected in LDS and appended to the list
  // of cells in use with ONE global atomic (47k appends on one word took 0.4 ms at 500k nodes; round 4 tried one entry per
  // thread - 16 000 workgroups, 16 000 appends: 184 us against 38).  A thread's sixteen entries are requested together.
  __shared__ uint32_t made[kRadixTile];  // at most one cell per entry of the tile (buckets of one node)
  __shared__ uint32_t nmade, base;
  const uint32_t E = H.counters[kCounterEntries];
  const uint32_t first = blockIdx.x * kRadixTile;
  if (first >= E) return;
  if (threadIdx.x == 0) nmade = 0;
  __syncthreads();
  const GridBox B = grid_box(H.counters);
  const uint32_t fb = grid_passes(B) & 1u;  // the buffer the last pass wrote
  const uint64_t* __restrict__ key = H.key[fb];
  const uint32_t ksh = B.packed ? 32u : 0u;  // (a packed entry: the key is the upper word)
  const bool direct = direct_index(H, B);
  constexpr uint32_t kRounds = kRadixTile / kBlock;
  uint64_t kp[kRounds], kc[kRounds], kn[kRounds];
#pragma unroll
  for (uint32_t r = 0; r < kRounds; ++r) {
    const uint32_t t = first + r * kBlock + threadIdx.x;
    const uint32_t tc = min(t, E - 1u);
    kc[r] = key[tc] >> ksh;
    kp[r] = key[tc ? tc - 1u : 0u] >> ksh;
    kn[r] = key[min(tc + 1u, E - 1u)] >> ksh;
  }
#pragma unroll
  for (uint32_t r = 0; r < kRounds; ++r) {
    const uint32_t t = first + r * kBlock + threadIdx.x;
    if (t >= E) continue;
    const uint64_t k = kc[r];
    const bool head = t == 0 || kp[r] != k;
    const bool tail = t + 1 == E || kn[r] != k;
    if (!(head || tail)) continue;
    bool created;
    uint32_t slot;
    if (direct) {  // the key is the slot (hash_device.h: direct_index): the bucket's head entry creates the cell
      slot = static_cast<uint32_t>(k);
      created = head;
      if (head) H.keys[slot] = k;
    } else {
      slot = insert_cell(H.keys, H.mask, k, created);  // whichever of the bucket's two ends comes first creates it
    }
    if (slot == 0xffffffffu) { atomicOr(&H.counters[kCounterFlags], 2u); continue; }
    if (created) made[atomicAdd(&nmade, 1u)] = slot;
    if (head) H.start[slot] = t;
    if (tail) H.end[slot] = t + 1;
  }
  __syncthreads();
  if (threadIdx.x == 0 && nmade) base = atomicAdd(&H.counters[kCounterUsed], nmade);
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < nmade; k += kBlock) H.used[base + k] = made[k];
}
// one lane per cell in use: its group size (entries carrying kMinFlag) and its place in the pass list.  Places are handed
// out per workgroup: one global atomic per (workgroup, pass) instead of one per cell on 27 words.
__global__ void __launch_bounds__(kBlock) k_grid_groups(HashArrays H) {
  __shared__ uint32_t lcount[27], lbase[27];
  const uint32_t used = H.counters[kCounterUsed];
  const GridBox B = grid_box(H.counters);
  const uint32_t* __restrict__ val = H.val[grid_passes(B) & 1u];
  for (uint32_t first = blockIdx.x * kBlock; first < used; first += gridDim.x * kBlock) {  // uniform per workgroup
    if (threadIdx.x < 27) lcount[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t u = first + threadIdx.x;
    uint32_t s = 0, pass = 0, rank = 0, gc = 0;
    if (u < used) {
      s = H.used[u];
      const uint32_t bs = H.start[s], be = H.end[s];
      if (be - bs > kMaxBucket) H.counters[kCounterDense] = 1u;  // a pile the group order's tables do not hold: the sequential loop takes the pass
      for (uint32_t e = bs; e < be; ++e) gc += val[e] >> 31;
      H.gcnt[s] = gc;
      if (gc) {
        int x, y, z;
        box_cell(B, H.keys[s], x, y, z);
        pass = static_cast<uint32_t>(mod3(x) + 3 * mod3(y) + 9 * mod3(z));
        rank = atomicAdd(&lcount[pass], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < 27 && lcount[threadIdx.x]) lbase[threadIdx.x] = atomicAdd(&H.counters[kCounterPass0 + threadIdx.x], lcount[threadIdx.x]);
    __syncthreads();
    if (gc) H.passList[static_cast<size_t>(pass) * H.n + lbase[pass] + rank] = s;
    __syncthreads();
  }
}

// ---- resolve ---------------------------------------------------------------------------------------
// The unstaged path reads and writes node state through agent-scope relaxed atomics (L2-served, write-through): within a
// pass exactly one wavefront touches a given node, and that wavefront must see its own earlier writes.  (The staged path
// moves whole records with ordinary 16-byte loads and stores between the acquire fence behind the wait and the release
// store of the completion stamp, group_resolve.)
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

// The nodes of a group are the entries of its own cell's bucket that carry kMinFlag (ascending node index, like the
// bucket).  Returns the bucket-relative position of the next one at or after `from`, or cnt.
PIES_DEV uint32_t next_min_entry(const uint32_t* __restrict__ val, uint32_t start, uint32_t cnt, uint32_t from, int lane) {
  for (uint32_t base = from & ~63u; base < cnt; base += 64) {
    const uint32_t e = base + static_cast<uint32_t>(lane);
    const bool f = e < cnt && e >= from && (val[start + e] & kMinFlag) != 0u;
    const unsigned long long m = __ballot(f);
    if (m) return base + static_cast<uint32_t>(__builtin_ctzll(m));
  }
  return cnt;
}

// One bucket [bs, bs + bc) met by visiting node i, whose state `a` lives in registers: 64 candidates are tested at a time,
// hits are resolved one by one in bucket order and the rest re-tested (the visiting node has moved).  Candidate state
// comes straight from global memory.  Returns the number of resolved pairs.
PIES_DEV uint32_t collide_bucket_global(const uint32_t* __restrict__ val, uint32_t bs, uint32_t bc, uint32_t i, PairState& a, float imi, float ri,
                                        float* pos, float* vel, const float* __restrict__ radius, int lane, const LaneRole role,
                                        float friction, float staticThreshold, uint32_t& tested) {
  uint32_t resolved = 0;
  tested += bc;  // statistics: candidates this visit looks at (SURVEY 8d counts 16 B per candidate neighbour)
  for (uint32_t base = 0; base < bc; base += 64) {
    const bool valid = base + lane < bc;
    const uint32_t j = valid ? (val[bs + base + lane] & kNodeMask) : 0xffffffffu;
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
  return resolved;
}

// Resolve of one group straight from global memory: every candidate's state is fetched again for every visiting
// node.  Only used for groups whose neighbourhood does not fit the LDS staging of k_collide (dense pile-ups).
PIES_DEV uint32_t collide_group_global(const HashArrays& H, const GridBox& B, const uint32_t* __restrict__ val, float* pos, float* vel,
                                       const float* __restrict__ radius, uint32_t gslot, int lane, const LaneRole role, float friction,
                                       float staticThreshold, uint32_t& tested) {
  uint32_t resolved = 0;
  const uint32_t gs = H.start[gslot], gn = H.end[gslot] - gs;
  for (uint32_t ge = next_min_entry(val, gs, gn, 0, lane); ge < gn; ge = next_min_entry(val, gs, gn, ge + 1, lane)) {
    const uint32_t i = val[gs + ge] & kNodeMask;
    PairState a = {ld(pos + 4 * i), ld(pos + 4 * i + 1), ld(pos + 4 * i + 2), ld(vel + 4 * i), ld(vel + 4 * i + 1), ld(vel + 4 * i + 2)};
    const float imi = ld(pos + 4 * i + 3);
    const float ri = radius[i];
    const int4 rg = H.rng[i];
    const uint32_t lx = rg.w & 0xff, ly = (rg.w >> 8) & 0xff, lz = (rg.w >> 16) & 0xff;
    for (uint32_t dx = 0; dx < lx; ++dx)
      for (uint32_t dy = 0; dy < ly; ++dy)
        for (uint32_t dz = 0; dz < lz; ++dz) {
          const uint32_t cs = find_bucket(H, B, rg.x + (int)dx, rg.y + (int)dy, rg.z + (int)dz);
          if (cs == 0xffffffffu) continue;
          const uint32_t bs = H.start[cs];
          resolved += collide_bucket_global(val, bs, H.end[cs] - bs, i, a, imi, ri, pos, vel, radius, lane, role, friction, staticThreshold, tested);
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
constexpr uint32_t kColMaxSpins = 1u << 22;  // default polls of one completion stamp before the wait is declared dead (~5 s);
                                             // PIES_COLLIDE_SPIN_LIMIT overrides it (0 = wait for ever; PIES_PROFILER_SAFE=1 implies 0:
                                             // counter collection serialises and slows the launch)
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

// One group on the staged path (or, for a dense neighbourhood, the unstaged one), in two halves.
//  * group_prepare reads only what the grid build left behind (buckets, entries, ranges): the group's eight buckets, the
//    table of distinct nodes, the table slot of every bucket entry, and - in registers, one own-cell entry per lane - the
//    group's own nodes with their ranges and slots.  k_collide_flow runs it BEFORE it waits for the conflicting groups of
//    earlier passes, so this half (about a quarter of a group's time) is off the chain of the 27 passes.
//  * group_resolve loads the node state, replays the visiting order of collide_group_global on the copies and writes the
//    touched nodes back.  Node state is read and written through agent-scope (sc1) loads and stores: in k_collide_flow the
//    previous owner of a node may be a wavefront of the same launch on another XCD.  Returns the number of resolved pairs.
struct GroupPlan {
  uint32_t cStart[8], cCnt[8], cOff[9];  // the 2x2x2 buckets above the group's cell (wave uniform)
  bool empty, staged, ownInLanes;
  uint32_t ownVal, ownRng, ownSlot;      // lane e: entry e of the own cell (cell 0), its range word and table slot
};
PIES_DEV void group_prepare(const HashArrays& H, const GridBox& B, const uint32_t* __restrict__ val, ColTable& T, uint32_t gslot, int lane,
                            int forceGlobal, GroupPlan& P) {
  P.empty = H.gcnt[gslot] == 0;
  P.staged = false;
  P.ownInLanes = false;
  P.ownVal = P.ownRng = P.ownSlot = 0;
  if (P.empty) return;
  // ---- the 2x2x2 cells above the group's cell: lanes 0..7 look one up each ------------------------------
  int gx, gy, gz;
  box_cell(B, H.keys[gslot], gx, gy, gz);
  uint32_t myStart = 0, myCnt = 0;
  if (lane < 8) {
    const uint32_t cs = find_bucket(H, B, gx + ((lane >> 2) & 1), gy + ((lane >> 1) & 1), gz + (lane & 1));
    if (cs != 0xffffffffu) { myStart = H.start[cs]; myCnt = H.end[cs] - myStart; }
  }
  P.cOff[0] = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    P.cStart[c] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(myStart), c));
    P.cCnt[c] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(myCnt), c));
    P.cOff[c + 1] = P.cOff[c] + P.cCnt[c];
  }
  bool staged = !forceGlobal && P.cOff[8] <= kColMaxEntries;
  if (staged) {
    for (uint32_t t = lane; t < kColSlots; t += 64) T.key[t] = kColEmpty;
    uint32_t unique = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      for (uint32_t base = 0; base < P.cCnt[c] && staged; base += 64) {
        bool fresh = false;
        if (base + lane < P.cCnt[c]) {
          const uint32_t v = val[P.cStart[c] + base + lane];
          const uint32_t j = v & kNodeMask;
          uint32_t h = col_hash(j);
          for (;;) {  // at most kColMaxUnique + 64 live entries: the probe ends
            const uint32_t old = atomicCAS(&T.key[h], kColEmpty, j);
            if (old == kColEmpty) { fresh = true; break; }
            if (old == j) break;
            h = (h + 1) & (kColSlots - 1);
          }
          T.ent[P.cOff[c] + base + lane] = static_cast<uint16_t>(h);
          if (c == 0 && base == 0) {  // the own cell's first 64 entries stay in the lanes
            P.ownVal = v;
            P.ownSlot = h;
            if (v & kMinFlag) P.ownRng = static_cast<uint32_t>(H.rng[j].w);
          }
        }
        unique += static_cast<uint32_t>(__popcll(__ballot(fresh)));
        if (unique > kColMaxUnique) staged = false;
      }
    }
  }
  P.staged = staged;
  P.ownInLanes = staged && P.cCnt[0] <= 64u;
}
PIES_DEV uint32_t group_resolve(const HashArrays& H, const GridBox& B, const uint32_t* __restrict__ val, ColTable& T, float* pos, float* vel,
                                const float* __restrict__ radius, uint32_t gslot, int lane, const LaneRole role, float friction,
                                float staticThreshold, const GroupPlan& P, uint32_t& tested) {
  uint32_t resolved = 0;
  if (P.empty) return 0;
  if (!P.staged) {
    return collide_group_global(H, B, val, pos, vel, radius, gslot, lane, role, friction, staticThreshold, tested);
  }
  const uint32_t* cStart = P.cStart;
  const uint32_t* cCnt = P.cCnt;
  const uint32_t* cOff = P.cOff;
  // Node records as two 16-byte loads (and stores, below).  Coherence with the wavefronts that owned these nodes earlier
  // in the launch, possibly on another XCD, is the memory model's: the caller's acquire fence at agent scope after the
  // wait orders these loads behind the predecessors' stores, which their release store of the completion stamp published
  // (round 1 read and wrote eight relaxed agent-scope dwords per node instead).
  const float4* pos4 = reinterpret_cast<const float4*>(pos);
  const float4* vel4 = reinterpret_cast<const float4*>(vel);
  for (uint32_t t = lane; t < kColSlots; t += 64) {
    const uint32_t j = T.key[t];
    if (j == kColEmpty) continue;
    const float4 pj = pos4[j], vj = vel4[j];
    T.px[t] = pj.x; T.py[t] = pj.y; T.pz[t] = pj.z; T.im[t] = pj.w;
    T.vx[t] = vj.x; T.vy[t] = vj.y; T.vz[t] = vj.z;
    T.r[t] = radius[j];
  }
  // ---- the visiting order of collide_group_global on the staged copies -----------------------------------
  // (the group's own cell is cell 0 of the eight)
  // the group's own nodes in ascending entry order: from the lanes when the own cell has at most 64 entries (no global
  // load and no table probe per node), otherwise looked up one by one
  unsigned long long own = P.ownInLanes ? __ballot(static_cast<uint32_t>(lane) < cCnt[0] && (P.ownVal & kMinFlag) != 0u) : 0ull;
  for (uint32_t ge = P.ownInLanes ? (own ? static_cast<uint32_t>(__builtin_ctzll(own)) : cCnt[0]) : next_min_entry(val, cStart[0], cCnt[0], 0, lane);
       ge < cCnt[0];
       ge = P.ownInLanes ? ((own &= own - 1ull) ? static_cast<uint32_t>(__builtin_ctzll(own)) : cCnt[0]) : next_min_entry(val, cStart[0], cCnt[0], ge + 1, lane)) {
    uint32_t i, si, rw;
    if (P.ownInLanes) {
      i = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(P.ownVal), static_cast<int>(ge))) & kNodeMask;
      si = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(P.ownSlot), static_cast<int>(ge)));
      rw = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(P.ownRng), static_cast<int>(ge)));
    } else {
      i = val[cStart[0] + ge] & kNodeMask;
      si = col_find(T.key, i);
      rw = static_cast<uint32_t>(H.rng[i].w);
    }
    PairState a = {T.px[si], T.py[si], T.pz[si], T.vx[si], T.vy[si], T.vz[si]};
    const float imi = T.im[si], ri = T.r[si];
    const uint32_t lx = rw & 0xff, ly = (rw >> 8) & 0xff, lz = (rw >> 16) & 0xff;
    for (uint32_t dx = 0; dx < lx; ++dx)
      for (uint32_t dy = 0; dy < ly; ++dy)
        for (uint32_t dz = 0; dz < lz; ++dz) {
          const uint32_t c = dx * 4 + dy * 2 + dz;
          uint32_t off = 0, bc = 0;
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (c == static_cast<uint32_t>(q)) { off = cOff[q]; bc = cCnt[q]; }
          tested += bc;
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
    reinterpret_cast<float4*>(pos)[j] = make_float4(T.px[t], T.py[t], T.pz[t], T.im[t]);
    reinterpret_cast<float4*>(vel)[j] = make_float4(T.vx[t], T.vy[t], T.vz[t], 0.0f);  // (the fourth component of a velocity record is 0 everywhere)
  }
  return resolved;  // (the caller's release store of the completion stamp publishes these stores)
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
  if (H.counters[3] || H.counters[kCounterDense]) return;  // failed: the host latches _simFailed; a pile: the sequential loop runs the pass
  const GridBox B = grid_box(H.counters);
  const uint32_t* __restrict__ val = H.val[grid_passes(B) & 1u];
  const uint32_t ngroups = H.counters[4 + pass];
  uint32_t resolved = 0, tested = 0;  // statistics, one atomic per wave at the end (a per-pair atomic on one word serialises the chip)
  for (uint32_t g = wave; g < ngroups; g += nwaves)
  {
    const uint32_t gslot = H.passList[static_cast<size_t>(pass) * H.n + g];
    GroupPlan P;
    group_prepare(H, B, val, T, gslot, lane, forceGlobal, P);
    resolved += group_resolve(H, B, val, T, pos, vel, radius, gslot, lane, role, friction, staticThreshold, P, tested);
  }
  if (lane == 0 && resolved) atomicAdd(&H.counters[31], resolved);
  if (lane == 0 && tested) atomicAdd(reinterpret_cast<unsigned long long*>(&H.counters[kCounterCandidates]), static_cast<unsigned long long>(tested));
}

// The same order of conflicting groups in ONE launch.  Groups are handed out through a ticket counter in pass-major
// order; before a wavefront touches its group it waits until every group of an earlier pass within two cells (the
// only ones that can share a node with it) has published its completion stamp.  A ticket's predecessors were all
// taken earlier by wavefronts that are resident and running, so the wait always ends; it is bounded anyway and a
// timeout latches the failure flag, which every wait loop polls.  Against the 27 launches this removes the barrier
// after each pass (a pass lasted as long as its slowest group, with ~1.7 wavefronts per SIMD).
PIES_DEV uint32_t ldu(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void __launch_bounds__(kColBlock) k_collide_flow(HashArrays H, float4* pos4, float4* vel4, const float* __restrict__ radius,
                                                            float friction, float staticThreshold, int forceGlobal, uint32_t maxSpins) {
  __shared__ ColTable tables[kColBlock / 64];
  ColTable& T = tables[threadIdx.x >> 6];
  float* pos = reinterpret_cast<float*>(pos4);
  float* vel = reinterpret_cast<float*>(vel4);
  const int lane = threadIdx.x & 63;
  const LaneRole role = {lane % 3, lane / 3};
  if (H.counters[3] || H.counters[kCounterDense]) return;
  const GridBox B = grid_box(H.counters);
  const uint32_t* __restrict__ val = H.val[grid_passes(B) & 1u];
  const uint32_t epoch = H.counters[kCounterEpoch];
  // inclusive prefix of the groups per pass, lane p holding pass p
  uint32_t incl = lane < 27 ? H.counters[4 + lane] : 0u;
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const uint32_t v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 26));
  uint32_t resolved = 0, tested = 0;
  for (;;) {
    uint32_t ticket = 0;
    if (lane == 0) ticket = atomicAdd(&H.counters[kCounterTicket], 1u);
    ticket = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ticket)));
    if (ticket >= total) break;
    const uint32_t pass = static_cast<uint32_t>(__popcll(__ballot(lane < 27 && incl <= ticket)));
    const uint32_t before = pass ? static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), pass - 1)) : 0u;
    const uint32_t gslot = H.passList[static_cast<size_t>(pass) * H.n + (ticket - before)];
    // ---- what does not depend on the predecessors' results: buckets, table of distinct nodes, own entries ------------
    GroupPlan P;
    group_prepare(H, B, val, T, gslot, lane, forceGlobal, P);
    // ---- wait for the conflicting groups of earlier passes -------------------------------------------------
    int x, y, z;
    box_cell(B, H.keys[gslot], x, y, z);
    bool gaveUp = false;
    for (int q = lane; q < 125; q += 64) {
      const int dx = q % 5 - 2, dy = (q / 5) % 5 - 2, dz = q / 25 - 2;
      if (dx == 0 && dy == 0 && dz == 0) continue;
      const uint32_t other = static_cast<uint32_t>(mod3(x + dx) + 3 * mod3(y + dy) + 9 * mod3(z + dz));
      if (other >= pass) continue;
      const uint32_t os = find_bucket(H, B, x + dx, y + dy, z + dz);
      if (os == 0xffffffffu || H.gcnt[os] == 0u) continue;
      uint32_t spins = 0;
      while (ldu(H.done + os) != epoch) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 255u) == 0u && ((maxSpins && spins > maxSpins) || ldu(H.counters + 3))) { gaveUp = true; break; }
      }
    }
    if (__ballot(gaveUp)) {  // a predecessor never finished (or the simulation failed elsewhere): latch and leave
      if (lane == 0) atomicOr(&H.counters[3], 8u);
      break;
    }
    // the stamps were polled relaxed; this fence orders every later load of the wavefront after them (pairs with the
    // release store below): the predecessors' node writes are visible by the memory model, not by cache behaviour
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    resolved += group_resolve(H, B, val, T, pos, vel, radius, gslot, lane, role, friction, staticThreshold, P, tested);
    if (lane == 0) __hip_atomic_store(H.done + gslot, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (lane == 0 && resolved) atomicAdd(&H.counters[31], resolved);
  if (lane == 0 && tested) atomicAdd(reinterpret_cast<unsigned long long*>(&H.counters[kCounterCandidates]), static_cast<unsigned long long>(tested));
}


// The reference's loop as it stands (Solver.cpp:85-130, SpatialHash.h:101-127): nodes in ascending index; a node's
// cell range comes from its position when its turn starts (the node may have been moved by earlier pairs of this
// pass), buckets are those of the grid built at the start of the iteration, visited in dx, dy, dz order.  The loop is
// one dependent chain: one wavefront runs it.  Lanes look up to 64 cells of the range at once.
__global__ void __launch_bounds__(64) k_collide_reference(HashArrays H, float4* pos4, float4* vel4, const float* __restrict__ radius, uint32_t n,
                                                          float scale, float friction, float staticThreshold, const uint32_t* __restrict__ gate,
                                                          unsigned long long budget) {
  if (gate && *gate == 0u) return;        // (the fallback of a parallel order: only when that pass asks for it)
  if (H.counters[kCounterFlags]) return;  // failed: the host latches _simFailed
  if (gate) {
    // What the pass costs here: the sum over the cells of (nodes in the cell)^2 candidate tests, ~15 M of them per second on this one
    // wavefront.  A pile whose pass would take minutes (BASELINE config 2 with the node-node pass on collapses 100 000 nodes into a few
    // cells - quirk Q2 -: 10^10 tests per iteration, hours per tick in the reference as well) is declared failed instead of keeping
    // the device busy for longer than a host waits: the one limit of this build the reference does not have (PIES_FALLBACK_VISITS, 1e9).
    const uint32_t used = H.counters[kCounterUsed];
    unsigned long long est = 0;
    for (uint32_t u = threadIdx.x; u < used; u += 64u) {
      const uint32_t sl = H.used[u];
      const unsigned long long len = H.end[sl] - H.start[sl];
      est += len * len;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) est += __shfl_xor(est, o, 64);
    if (est > budget) {
      if (threadIdx.x == 0) atomicOr(&H.counters[kCounterFlags], 4u);
      return;
    }
  }
  float* pos = reinterpret_cast<float*>(pos4);
  float* vel = reinterpret_cast<float*>(vel4);
  const int lane = threadIdx.x & 63;
  const LaneRole role = {lane % 3, lane / 3};
  const GridBox B = grid_box(H.counters);
  const uint32_t* __restrict__ val = H.val[grid_passes(B) & 1u];
  uint32_t resolved = 0, tested = 0;
  for (uint32_t i = 0; i < n; ++i) {
    PairState a = {ld(pos + 4 * i), ld(pos + 4 * i + 1), ld(pos + 4 * i + 2), ld(vel + 4 * i), ld(vel + 4 * i + 1), ld(vel + 4 * i + 2)};
    const float imi = ld(pos + 4 * i + 3);
    const float ri = radius[i];
    int mx, my, mz;
    uint32_t lx, ly, lz;
    if (!node_range(a.pix, a.piy, a.piz, ri, scale, mx, my, mz, lx, ly, lz)) {
      if (lane == 0) atomicOr(&H.counters[kCounterFlags], 1u);
      break;
    }
    const uint32_t ncell = lx * ly * lz;
    for (uint32_t cbase = 0; cbase < ncell; cbase += 64) {
      const uint32_t c = cbase + static_cast<uint32_t>(lane);
      uint32_t myStart = 0, myCnt = 0;
      if (c < ncell) {  // cell c of the range, dz fastest (SpatialHash.h:108-125)
        const uint32_t dz = c % lz, dy = (c / lz) % ly, dx = c / (lz * ly);
        const uint32_t cs = find_bucket(H, B, mx + static_cast<int>(dx), my + static_cast<int>(dy), mz + static_cast<int>(dz));
        if (cs != 0xffffffffu) { myStart = H.start[cs]; myCnt = H.end[cs] - myStart; }
      }
      const uint32_t here = min(64u, ncell - cbase);
      for (uint32_t q = 0; q < here; ++q) {
        const uint32_t bs = static_cast<uint32_t>(__shfl(static_cast<int>(myStart), static_cast<int>(q), 64));
        const uint32_t bc = static_cast<uint32_t>(__shfl(static_cast<int>(myCnt), static_cast<int>(q), 64));
        if (bc) resolved += collide_bucket_global(val, bs, bc, i, a, imi, ri, pos, vel, radius, lane, role, friction, staticThreshold, tested);
      }
    }
    if (lane == 0) {
      st(pos + 4 * i, a.pix); st(pos + 4 * i + 1, a.piy); st(pos + 4 * i + 2, a.piz);
      st(vel + 4 * i, a.vix); st(vel + 4 * i + 1, a.viy); st(vel + 4 * i + 2, a.viz);
      if ((i & 1023u) == 0u) H.counters[kCounterProgress] = i;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (lane == 0 && resolved) atomicAdd(&H.counters[kCounterPairs], resolved);
  if (lane == 0 && tested) atomicAdd(reinterpret_cast<unsigned long long*>(&H.counters[kCounterCandidates]), static_cast<unsigned long long>(tested));
}

// resets the work queue of k_collide_flow without a new hash build (profile replays)
__global__ void k_collide_rearm(HashArrays H) {
  if (threadIdx.x == 0) {
    H.counters[kCounterTicket] = 0;
    H.counters[kCounterEpoch] += 1;
  }
}

// ----------------------------------------------------------------------------------------------------
uint32_t launch_hash_build(hipStream_t st_, const HashArrays& H, const NodeArrays& nd, float scale, uint32_t sortPasses, bool groups) {
  if (nd.n == 0) return 0;
  const uint32_t n = nd.n;
  uint32_t launches = 0;
  const dim3 wide(std::min<uint32_t>(2048u, (H.capacity / 8 + kBlock - 1) / kBlock));
  // (round 4: three launches fewer - the counters are zeroed by the first workgroup of k_grid_range, the bounding box is reduced
  // by an extra workgroup of k_scan_tiles, the tile sums are added by k_grid_emit itself)
  hipLaunchKernelGGL(k_grid_reset, wide, dim3(kBlock), 0, st_, H); ++launches;
  hipLaunchKernelGGL(k_grid_range, grid_for(n + 1), dim3(kBlock), 0, st_, H, nd.pos, nd.radius, n, scale); ++launches;
  const uint32_t m = n + 1, tiles = (m + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(k_scan_tiles, dim3(tiles + 1u), dim3(kBlock), 0, st_, H.entCount, H.entOff, m, H.scanSums, H, grid_for(n + 1).x, sortPasses); ++launches;
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st_, H.scanSums, tiles); ++launches;
  hipLaunchKernelGGL(k_grid_emit, grid_for(n), dim3(kBlock), 0, st_, H, n); ++launches;
  const uint32_t nblkMax = (H.maxEntries + kRadixTile - 1) / kRadixTile;
  for (uint32_t pass = 0; pass < sortPasses; ++pass) {  // the key's bits are dealt evenly to the passes (grid_box)
    hipLaunchKernelGGL(k_radix_hist, dim3(nblkMax), dim3(kBlock), 0, st_, H, pass, nblkMax);
    hipLaunchKernelGGL(k_radix_scan, dim3(1u << kRadixMaxDigit), dim3(kBlock), 0, st_, H, pass, nblkMax);
    hipLaunchKernelGGL(k_radix_scatter, dim3(nblkMax), dim3(kBlock), 0, st_, H, pass, nblkMax);
    launches += 3;
  }
  hipLaunchKernelGGL(k_grid_cells, dim3(nblkMax), dim3(kBlock), 0, st_, H); ++launches;
  if (groups) { hipLaunchKernelGGL(k_grid_groups, wide, dim3(kBlock), 0, st_, H); ++launches; }
  return launches;
}

uint32_t launch_collide(hipStream_t st_, const HashArrays& H, const NodeArrays& nd, float scale, float friction, float staticThreshold, bool rearm) {
  if (nd.n == 0) return 0;
  const dim3 grid(std::max<uint32_t>(1u, std::min<uint32_t>(2048u, (nd.n / 8 + 1) / 2)));
  auto flag = [](const char* name) { const char* e = tuning_env(name); return e && e[0] == '1' ? 1 : 0; };
  const int forceGlobal = flag("PIES_COLLIDE_GLOBAL");  // diagnostics, read when the substep is captured
  const int passes = flag("PIES_COLLIDE_PASSES");
  if (!passes) {
    uint32_t maxSpins = kColMaxSpins;
    if (const char* e = tuning_env("PIES_COLLIDE_SPIN_LIMIT")) maxSpins = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
    else if (const char* e = std::getenv("PIES_PROFILER_SAFE"); e && e[0] == '1') maxSpins = 0;
    if (rearm) hipLaunchKernelGGL(k_collide_rearm, dim3(1), dim3(64), 0, st_, H);
    hipLaunchKernelGGL(k_collide_flow, grid, dim3(kColBlock), 0, st_, H, nd.pos, nd.vel, nd.radius, friction, staticThreshold, forceGlobal,
                       maxSpins);
    // a cell with more nodes than the group order's tables hold: the pass in the reference's own order (returns at once otherwise)
    return 1 + launch_collide_reference(st_, H, nd, scale, friction, staticThreshold, H.counters + kCounterDense);
  }
  for (uint32_t pass = 0; pass < 27; ++pass)
    hipLaunchKernelGGL(k_collide, grid, dim3(kColBlock), 0, st_, H, nd.pos, nd.vel, nd.radius, pass, friction, staticThreshold, forceGlobal);
  return 27 + launch_collide_reference(st_, H, nd, scale, friction, staticThreshold, H.counters + kCounterDense);
}

uint32_t launch_collide_reference(hipStream_t st_, const HashArrays& H, const NodeArrays& nd, float scale, float friction, float staticThreshold,
                                  const uint32_t* gate) {
  if (nd.n == 0) return 0;
  unsigned long long budget = 1000000000ull;  // candidate tests a fallback pass may cost (a minute on the one wavefront; BASELINE config 4: 3.4e8)
  if (const char* e = tuning_env("PIES_FALLBACK_VISITS")) budget = std::strtoull(e, nullptr, 10);
  hipLaunchKernelGGL(k_collide_reference, dim3(1), dim3(64), 0, st_, H, nd.pos, nd.vel, nd.radius, nd.n, scale, friction, staticThreshold, gate, budget);
  return 1;
}

}  // namespace pies
