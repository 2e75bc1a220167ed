// Exact-key open-addressing cell table shared by the node grid (hash_kernels.hip) and the triangle grid
// (tri_kernels.hip): a cell id (three signed 21-bit coordinates) is packed into 63 bits and compared in
// full, so two different cells never alias -- the reference's hash map is exact as well.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pies {

#ifndef PIES_DEV
#define PIES_DEV __device__ __forceinline__
#endif

constexpr uint64_t kEmpty = ~0ull;
constexpr int kCoordBias = 1 << 20;  // cell coordinates are packed as 21-bit biased integers

PIES_DEV uint64_t pack_cell(int x, int y, int z) {
  return (static_cast<uint64_t>(static_cast<uint32_t>(x + kCoordBias)) << 42) |
         (static_cast<uint64_t>(static_cast<uint32_t>(y + kCoordBias)) << 21) | static_cast<uint64_t>(static_cast<uint32_t>(z + kCoordBias));
}
PIES_DEV uint32_t hash_cell(uint64_t k, uint32_t mask) {
  k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
  k ^= k >> 27; k *= 0x94d049bb133111ebull;
  k ^= k >> 31;
  return static_cast<uint32_t>(k) & mask;
}
PIES_DEV int mod3(int v) { int m = v % 3; return m < 0 ? m + 3 : m; }

// read-only lookup (the table is static while it is used)
PIES_DEV uint32_t find_cell(const uint64_t* __restrict__ keys, uint32_t mask, uint64_t key) {
  uint32_t h = hash_cell(key, mask);
  for (int probe = 0; probe < 4096; ++probe) {
    const uint64_t k = keys[h];
    if (k == key) return h;
    if (k == kEmpty) return 0xffffffffu;
    h = (h + 1) & mask;
  }
  return 0xffffffffu;
}

// find-or-insert with a 64-bit compare-and-swap; `created` tells the caller to record a newly used slot
PIES_DEV uint32_t insert_cell(uint64_t* keys, uint32_t mask, uint64_t key, bool& created) {
  uint32_t h = hash_cell(key, mask);
  created = false;
  for (int probe = 0; probe < 4096; ++probe) {
    const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[h]), kEmpty, key);
    if (old == kEmpty) { created = true; return h; }
    if (old == key) return h;
    h = (h + 1) & mask;
  }
  return 0xffffffffu;
}

// ascending rank sort of c values (one wave; c is small: a bucket or a per-node list)
// rankOfValue (optional): rankOfValue[v] = position of value v in its sorted list (the values are distinct indices)
PIES_DEV void rank_sort(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t start, uint32_t c, int lane,
                        uint32_t* __restrict__ rankOfValue = nullptr) {
  for (uint32_t e = lane; e < c; e += 64) {
    const uint32_t v = in[start + e];
    uint32_t rank = 0;
    for (uint32_t f = 0; f < c; ++f) rank += (in[start + f] < v) ? 1u : 0u;
    out[start + rank] = v;
    if (rankOfValue) rankOfValue[v] = rank;
  }
}


}  // namespace pies
