// Device-side helpers of the node grid shared by hash_kernels.hip (grid build, group-ordered and reference-ordered
// resolve) and pair_kernels.hip (pair-ordered resolve).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "cell_table.h"
#include "hash_kernels.h"

namespace pies {

// entry value: node index (25 bits) | whether the node's range is two cells long, per axis (bit 27 x, 26 y, 25 z; only
// meaningful where no range is longer: the pair-ordered resolve) | which side of the node's range the cell is on, per axis (bit
// 30 x, 29 y, 28 z: set when the cell is not the range's minimum on that axis) | kMinFlag in the node's minimum cell
constexpr uint32_t kMinFlag = 0x80000000u, kNodeMask = 0x01ffffffu;
constexpr uint32_t kSideShift = 28, kLongShift = 25;

// ---- the cell box of a build: origin and bits per axis of the packed key --------------------------------------
struct GridBox {
  int mn[3];
  uint32_t ext[3];   // max - min per axis
  uint32_t bits[3];
  uint32_t passes, digit;  // the build's radix sort: passes captured in the substep graph, key bits a pass sorts by
  bool packed;             // the key has at most 32 bits: an entry travels through the sort as ONE 64-bit word, key << 32 | value
  bool empty;
};
constexpr uint32_t kRadixMaxDigit = 11;  // bits per pass at most (2048 bins)
PIES_DEV GridBox grid_box(const uint32_t* __restrict__ counters) {
  GridBox B;
  B.empty = false;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    B.mn[a] = static_cast<int>(counters[kCounterBoxMin + a]);
    const int mx = static_cast<int>(counters[kCounterBoxMax + a]);
    if (mx < B.mn[a]) B.empty = true;
    B.ext[a] = B.empty ? 0u : static_cast<uint32_t>(mx - B.mn[a]);
    B.bits[a] = B.ext[a] ? 32u - static_cast<uint32_t>(__builtin_clz(B.ext[a])) : 0u;
  }
  // The key's bits are dealt evenly to the passes the host captured (launch_hash_build): a box of 17 key bits is sorted in two
  // passes of 9 bits, not in three of 8 with five more launches that find nothing to do.  More bits than kRadixMaxDigit per
  // captured pass is latched by k_grid_box (the host follows the box at its synchronisations, with bits to spare).
  B.passes = counters[kCounterSortPasses];
  const uint32_t total = B.bits[0] + B.bits[1] + B.bits[2];
  B.digit = B.passes ? min(kRadixMaxDigit, (total + B.passes - 1u) / B.passes) : 0u;
  B.packed = total <= 32u;
  return B;
}
PIES_DEV uint32_t grid_passes(const GridBox& B) { return B.passes; }
PIES_DEV bool in_box(const GridBox& B, int x, int y, int z) {
  return !B.empty && x >= B.mn[0] && y >= B.mn[1] && z >= B.mn[2] && static_cast<uint32_t>(x - B.mn[0]) <= B.ext[0] &&
         static_cast<uint32_t>(y - B.mn[1]) <= B.ext[1] && static_cast<uint32_t>(z - B.mn[2]) <= B.ext[2];
}
PIES_DEV uint64_t box_key(const GridBox& B, int x, int y, int z) {  // in_box(x, y, z)
  return (static_cast<uint64_t>(static_cast<uint32_t>(x - B.mn[0])) << (B.bits[1] + B.bits[2])) |
         (static_cast<uint64_t>(static_cast<uint32_t>(y - B.mn[1])) << B.bits[2]) | static_cast<uint64_t>(static_cast<uint32_t>(z - B.mn[2]));
}
PIES_DEV void box_cell(const GridBox& B, uint64_t key, int& x, int& y, int& z) {
  z = B.mn[2] + static_cast<int>(key & ((1ull << B.bits[2]) - 1ull));
  y = B.mn[1] + static_cast<int>((key >> B.bits[2]) & ((1ull << B.bits[1]) - 1ull));
  x = B.mn[0] + static_cast<int>(key >> (B.bits[1] + B.bits[2]));
}
// bucket of cell (x, y, z): index slot or ~0
// A build whose packed key is shorter than the index (every scene but one that spans more cells than it has index slots:
// 17 bits against 2^23 slots in BASELINE config 4) uses the key itself as the slot - no hashing, no probing, and no
// compare-and-swap when the index is built (k_grid_cells).  The same test on both sides, from the build's own box.
PIES_DEV bool direct_index(const HashArrays& H, const GridBox& B) {
  const uint32_t total = B.bits[0] + B.bits[1] + B.bits[2];
  return total <= 30u && (1u << total) <= H.mask + 1u;
}
PIES_DEV uint32_t find_bucket(const HashArrays& H, const GridBox& B, int x, int y, int z) {
  if (!in_box(B, x, y, z)) return 0xffffffffu;
  const uint64_t key = box_key(B, x, y, z);
  if (direct_index(H, B)) {
    const uint32_t h = static_cast<uint32_t>(key);
    return H.keys[h] == key ? h : 0xffffffffu;
  }
  return find_cell(H.keys, H.mask, key);
}

// NodeCompRange (Solver.cpp:877-901).  Returns false for a non-finite position; an over-long range is empty, like the
// reference's (:896-898).
PIES_DEV bool node_range(float px, float py, float pz, float radius, float scale, int& mx, int& my, int& mz, uint32_t& lx, uint32_t& ly,
                         uint32_t& lz) {
  const float R = (radius + 0.5f) / scale;
  const float gx = px / scale - R, gy = py / scale - R, gz = pz / scale - R;
  const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
  const float twoR = 2 * R;
  const float cx = ceilf((gx - fx) + twoR), cy = ceilf((gy - fy) + twoR), cz = ceilf((gz - fz) + twoR);
  const bool finite = (fabsf(fx) < 1.0e6f) && (fabsf(fy) < 1.0e6f) && (fabsf(fz) < 1.0e6f) && (cx >= 0.0f) && (cy >= 0.0f) && (cz >= 0.0f) &&
                      (cx < 1.0e6f) && (cy < 1.0e6f) && (cz < 1.0e6f);  // false for NaN as well
  mx = finite ? static_cast<int>(fx) : 0;
  my = finite ? static_cast<int>(fy) : 0;
  mz = finite ? static_cast<int>(fz) : 0;
  lx = finite ? static_cast<uint32_t>(cx) : 0u;
  ly = finite ? static_cast<uint32_t>(cy) : 0u;
  lz = finite ? static_cast<uint32_t>(cz) : 0u;
  if (lx > 50 || ly > 50 || lz > 50) lx = ly = lz = 0;
  return finite;
}

}  // namespace pies
