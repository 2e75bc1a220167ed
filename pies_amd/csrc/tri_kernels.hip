// Point-triangle collisions of the Projective-Dynamics substep.
//
//   detection  Src/Solver.cpp:680-875: the reference inserts every surface triangle into a world-unit cell grid over the
//              AABB of its nodes' current and previous positions (TriCompRange :942-979, sweptTriRange :639-677);
//              for every triangle, every triangle without a common node found in the cells of its swept range
//              is tested with three point-triangle CCDs (CollisionDetection.cpp:227-302).  Contacts are listed
//              in the reference's order: thread by thread (triangle id modulo threadCount), triangle by
//              triangle, cell by cell (dx,dy,dz), bucket entry by entry (ascending triangle id), point by
//              point -- including the duplicates a pair produces when it shares several cells.  The device finds the
//              same pairs from a grid that holds every triangle once (tri_kernels.h) and writes the same list.
//   constraint CollisionConstraint.cpp:67-194: differential coordinates w.r.t. the point (A = B), projection
//              along the triangle normal, w = 1e4; its 4x4 block is added to the system matrix through a
//              per-node list of contacts (diagonal into cdiag, off-diagonals applied inside the SpMV).
//   sequential parts (stabilisation :367-383 via CollisionConstraint.cpp:126-162, friction Solver.cpp:431-471)
//              are order dependent Gauss-Seidel passes over the contact list.  They run level by level of the list's
//              dependency DAG (two contacts conflict when they share a node; k_tri_levels) - the sequential result -
//              on an LDS copy of the touched nodes when these fit, through L2 otherwise; lists with more than
//              kTriMaxLevels levels are walked by one wavefront, 64 contacts at a time.
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "cell_table.h"
#include "tri_kernels.h"

namespace pies {

constexpr int kBlock = 256;
constexpr int kTriTeam = 16;  // lanes sharing one triangle's candidate list in the counting pass
static inline dim3 grid_for(uint32_t n) { return dim3((n + kBlock - 1) / kBlock); }

struct F3 {
  float x, y, z;
};
PIES_DEV F3 f3(float x, float y, float z) { return {x, y, z}; }
PIES_DEV F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PIES_DEV F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PIES_DEV F3 operator*(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PIES_DEV F3 operator*(float s, F3 a) { return {s * a.x, s * a.y, s * a.z}; }
PIES_DEV F3 operator/(F3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
PIES_DEV F3 neg(F3 a) { return {-a.x, -a.y, -a.z}; }
PIES_DEV float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PIES_DEV F3 cross(F3 a, F3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
PIES_DEV F3 normalize(F3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
PIES_DEV F3 xyz(const float4& v) { return {v.x, v.y, v.z}; }

// inverse(mat3(c0, c1, c2)) * v with glm's cofactor formula (columns c0, c1, c2)
PIES_DEV F3 solve_columns(F3 c0, F3 c1, F3 c2, F3 v) {
  const float m00 = c0.x, m01 = c0.y, m02 = c0.z, m10 = c1.x, m11 = c1.y, m12 = c1.z, m20 = c2.x, m21 = c2.y, m22 = c2.z;
  const float det = +m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02) + m20 * (m01 * m12 - m11 * m02);
  const float ood = 1.0f / det;
  const float i00 = +(m11 * m22 - m21 * m12) * ood, i10 = -(m10 * m22 - m20 * m12) * ood, i20 = +(m10 * m21 - m20 * m11) * ood;
  const float i01 = -(m01 * m22 - m21 * m02) * ood, i11 = +(m00 * m22 - m20 * m02) * ood, i21 = -(m00 * m21 - m20 * m01) * ood;
  const float i02 = +(m01 * m12 - m11 * m02) * ood, i12 = -(m00 * m12 - m10 * m02) * ood, i22 = +(m00 * m11 - m10 * m01) * ood;
  return {i00 * v.x + i10 * v.y + i20 * v.z, i01 * v.x + i11 * v.y + i21 * v.z, i02 * v.x + i12 * v.y + i22 * v.z};
}
PIES_DEV bool bary_outside(F3 b) { return (0.0f > b.x) || (b.x > 1.0f) || (0.0f > b.y) || (b.y > 1.0f) || (b.x + b.y > 1.0f); }

struct Cubic {
  float c3, c2, c1, c0;
};
PIES_DEV float eval(const Cubic& c, float t) {
  const float t2 = t * t;
  return c.c3 * t2 * t + c.c2 * t2 + c.c1 * t + c.c0;
}
PIES_DEV void expand_term(float a0, float b0, float c0, float ad, float bd, float cd, Cubic& e) {  // CollisionDetection.cpp:209-221
  e.c3 += ad * bd * cd;
  e.c2 += ad * bd * c0 + a0 * bd * cd + ad * b0 * cd;
  e.c1 += ad * b0 * c0 + a0 * bd * c0 + a0 * b0 * cd;
  e.c0 += a0 * b0 * c0;
}
// smallest real root of a genuine cubic in [0,1]: monotone pieces between the critical points, first sign
// change bisected 32 times (stands in for the companion-matrix eigenvalues of CollisionDetection.cpp:189-204)
PIES_DEV bool cubic_smallest_root01(const Cubic& c, float& root) {
  float cuts[4];
  int nc = 0;
  cuts[nc++] = 0.0f;
  const float a = 3.0f * c.c3, b = 2.0f * c.c2, cc = c.c1;
  const float disc = b * b - 4.0f * a * cc;
  if (disc > 0.0f) {
    const float sq = sqrtf(disc);
    float t0 = (-b - sq) / (2.0f * a), t1 = (-b + sq) / (2.0f * a);
    if (t0 > t1) { const float tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 > 0.0f && t0 < 1.0f) cuts[nc++] = t0;
    if (t1 > 0.0f && t1 < 1.0f) cuts[nc++] = t1;
  }
  cuts[nc++] = 1.0f;
  for (int k = 0; k + 1 < nc; ++k) {
    float lo = cuts[k], hi = cuts[k + 1];
    float flo = eval(c, lo), fhi = eval(c, hi);
    if (flo == 0.0f) { root = lo; return true; }
    if (flo * fhi > 0.0f) continue;
    for (int it = 0; it < 32; ++it) {
      const float mid = 0.5f * (lo + hi);
      const float fm = eval(c, mid);
      if ((flo < 0.0f) == (fm < 0.0f) && fm != 0.0f) { lo = mid; flo = fm; } else { hi = mid; fhi = fm; }
    }
    root = hi;
    return true;
  }
  return false;
}
PIES_DEV bool find_root(const Cubic& c, float& t) {  // CollisionDetection.cpp:143-205
  if (c.c3 == 0.0f) {
    if (c.c2 == 0.0f) {
      if (c.c1 == 0.0f) {
        if (c.c0 == 0.0f) { t = 0.0f; return true; }
        return false;
      }
      t = -c.c0 / c.c1;
      return t >= 0.0f && t <= 1.0f;
    }
    const float disc = c.c1 * c.c1 - 4.0f * c.c2 * c.c0;
    if (disc < 0.0f) return false;
    const float sq = sqrtf(disc);
    t = (-c.c1 - sq) / (2.0f * c.c2);
    if (t > 1.0f) return false;
    if (t < 0.0f) t = (-c.c1 + sq) / (2.0f * c.c2);
    return t >= 0.0f && t <= 1.0f;
  }
  return cubic_smallest_root01(c, t);
}
// CollisionDetection.cpp:227-302
PIES_DEV bool point_triangle_ccd(F3 ap0, F3 ab0, F3 ac0, F3 ap1, F3 ab1, F3 ac1, float threshold) {
  const F3 n0 = normalize(cross(ab0, ac0)), n1 = normalize(cross(ab1, ac1));
  const float d0 = dot(n0, ap0), d1 = dot(n1, ap1);
  if (d0 * d1 >= 0.0f) {
    if (d1 >= 0.0f && d1 < threshold) return !bary_outside(solve_columns(ab1, ac1, n1, ap1));
    return false;
  }
  const F3 apd = ap1 - ap0, abd = ab1 - ab0, acd = ac1 - ac0;
  Cubic e{0.f, 0.f, 0.f, 0.f};
  expand_term(ap0.x, ab0.y, ac0.z, apd.x, abd.y, acd.z, e);
  expand_term(-ap0.x, ac0.y, ab0.z, -apd.x, acd.y, abd.z, e);
  expand_term(-ab0.x, ap0.y, ac0.z, -abd.x, apd.y, acd.z, e);
  expand_term(ab0.x, ac0.y, ap0.z, abd.x, acd.y, apd.z, e);
  expand_term(ac0.x, ap0.y, ab0.z, acd.x, apd.y, abd.z, e);
  expand_term(-ac0.x, ab0.y, ap0.z, -acd.x, abd.y, apd.z, e);
  float t;
  if (!find_root(e, t)) return false;
  const F3 apt = ap0 + t * apd, abt = ab0 + t * abd, act = ac0 + t * acd;
  const F3 n = normalize(cross(abt, act));
  return !bary_outside(solve_columns(abt, act, n, apt));
}

// ---- triangle grid (tri_kernels.h: every triangle once, in the cell of its range's minimum corner) ----------------------
constexpr uint32_t kNil = kTriNil;
PIES_DEV uint32_t grid_slot(const TriGridLevel& L, int x, int y, int z) {  // cell coordinates of the level -> slot (modulo the table)
  const uint32_t ux = static_cast<uint32_t>(x) & ((1u << L.lx) - 1u), uy = static_cast<uint32_t>(y) & ((1u << L.ly) - 1u),
                 uz = static_cast<uint32_t>(z) & ((1u << L.lz) - 1u);
  return L.base + ((((ux << L.ly) | uy) << L.lz) | uz);
}
// cells of the level a range of `len` world cells from `m` spans
PIES_DEV uint32_t level_extent(int m, uint32_t len, uint32_t shift) { return static_cast<uint32_t>(((m + static_cast<int>(len) - 1) >> shift) - (m >> shift)) + 1u; }

// TriCompRange / sweptTriRange: floor(min), ceil(max) - floor(min) over position and prevPosition, world units; the triangle's
// record for the detection (box of its six corner positions, "both normals are non-zero"); its size class and slot.
__global__ void __launch_bounds__(kBlock) k_tri_box(TriArrays T, const float4* __restrict__ pos, const float4* __restrict__ prev) {
  __shared__ uint32_t sExt[kTriLevels], sCnt[kTriLevels];
  if (threadIdx.x < kTriLevels) { sExt[threadIdx.x] = 0; sCnt[threadIdx.x] = 0; }
  __syncthreads();
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  uint32_t slot = kNil;
  int cls = -1;
  uint32_t ext = 0;
  if (t < T.nt) {
    const uint32_t i0 = T.tris[3 * t], i1 = T.tris[3 * t + 1], i2 = T.tris[3 * t + 2];
    const F3 b1 = xyz(pos[i0]), c1 = xyz(pos[i1]), d1 = xyz(pos[i2]);
    const F3 b0 = xyz(prev[i0]), c0 = xyz(prev[i1]), d0 = xyz(prev[i2]);
    const float pv[3][3] = {{b1.x, b1.y, b1.z}, {c1.x, c1.y, c1.z}, {d1.x, d1.y, d1.z}};
    const float qv[3][3] = {{b0.x, b0.y, b0.z}, {c0.x, c0.y, c0.z}, {d0.x, d0.y, d0.z}};
    float mn[3], mx[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (i == 0) { mx[k] = pv[i][k]; mn[k] = pv[i][k]; }
        mx[k] = fmaxf(pv[i][k], mx[k]); mx[k] = fmaxf(qv[i][k], mx[k]);
        mn[k] = fminf(pv[i][k], mn[k]); mn[k] = fminf(qv[i][k], mn[k]);
      }
    }
    int m[3];
    uint32_t len[3];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float f = floorf(mn[k]);
      ok = ok && (fabsf(f) < 1.0e6f) && (fabsf(mx[k]) < 1.0e6f);
      m[k] = ok ? static_cast<int>(f) : 0;
      len[k] = ok ? static_cast<uint32_t>(ceilf(mx[k]) - static_cast<float>(static_cast<long long>(f))) : 0u;
    }
    if (!ok) atomicOr(&T.counters[3], 32u);  // non-finite
    if (!ok || len[0] > kTriInsertMaxCells || len[1] > kTriInsertMaxCells || len[2] > kTriInsertMaxCells)
      len[0] = len[1] = len[2] = 0;  // the reference returns an empty range (Solver.cpp:974-976)
    T.rng[t] = make_int4(m[0], m[1], m[2], static_cast<int>(len[0] | (len[1] << 8) | (len[2] << 16)));
    const F3 nn0 = cross(c0 - b0, d0 - b0), nn1 = cross(c1 - b1, d1 - b1);
    const bool regular = dot(nn0, nn0) > 0.0f && dot(nn1, nn1) > 0.0f;
    T.boxOf[4 * t] = make_float4(mn[0], mn[1], mn[2], __uint_as_float(t));
    T.boxOf[4 * t + 1] = make_float4(mx[0], mx[1], mx[2], regular ? 1.0f : 0.0f);
    T.boxOf[4 * t + 2] = make_float4(__int_as_float(m[0]), __int_as_float(m[1]), __int_as_float(m[2]), __uint_as_float(len[0] | (len[1] << 8) | (len[2] << 16)));
    T.boxOf[4 * t + 3] = make_float4(__uint_as_float(i0), __uint_as_float(i1), __uint_as_float(i2), 0.0f);
    T.head[t] = kNil;
    if (len[0] * len[1] * len[2] != 0u) {
      int k = 0;
#pragma unroll
      for (int l = kTriLevels - 1; l >= 0; --l) {  // the finest class whose cells the range spans at most kTriLevelExt of per axis
        const uint32_t sh = T.level[l].shift;
        const uint32_t e = max(max(level_extent(m[0], len[0], sh), level_extent(m[1], len[1], sh)), level_extent(m[2], len[2], sh));
        if (l == kTriLevels - 1 || e <= kTriLevelExt) { k = l; ext = e; }
      }
      cls = k;
      const uint32_t sh = T.level[k].shift;
      slot = grid_slot(T.level[k], m[0] >> sh, m[1] >> sh, m[2] >> sh);
      T.posIn[t] = atomicAdd(&T.cellCnt[slot], 1u);
    }
    T.cellOf[t] = slot;
  }
  if (blockIdx.x == 0 && (threadIdx.x == 0 || (threadIdx.x >= 4 && threadIdx.x <= 8))) T.counters[threadIdx.x] = 0;
  // the tiles' sums for the prefix sum over the slots: the triangles of a wavefront are neighbours, so their slots lie in a few
  // tiles - one atomic per wavefront and tile
  for (unsigned long long todo = __ballot(slot != kNil); todo;) {
    const uint32_t tile = __shfl(slot, __builtin_ctzll(todo), 64) / kGridTile;
    const unsigned long long same = __ballot(slot != kNil && slot / kGridTile == tile);
    if ((threadIdx.x & 63u) == static_cast<uint32_t>(__builtin_ctzll(same))) atomicAdd(&T.tileSum[tile], static_cast<uint32_t>(__popcll(same)));
    todo &= ~same;
  }
  // the classes' statistics: per wavefront a count per class and - only while it still raises it - the longest range
#pragma unroll
  for (int k = 0; k < kTriLevels; ++k) {
    const unsigned long long in = __ballot(cls == k);
    if (in == 0ull) continue;
    if ((threadIdx.x & 63u) == static_cast<uint32_t>(__builtin_ctzll(in))) atomicAdd(&sCnt[k], static_cast<uint32_t>(__popcll(in)));
    if (cls == k && sExt[k] < ext) atomicMax(&sExt[k], ext);
  }
  __syncthreads();
  if (threadIdx.x < kTriLevels && sCnt[threadIdx.x]) {
    atomicMax(&T.counters[10 + threadIdx.x], sExt[threadIdx.x]);
    atomicAdd(&T.counters[13 + threadIdx.x], sCnt[threadIdx.x]);
  }
}
// first entry of every slot = the exclusive prefix sum of the slots' counts, in slot order, so that the slots of a row of cells
// (consecutive z) have their entries in one stretch.  The sums of tiles of kGridTile slots come from k_tri_box; every tile adds up
// the sums before it and scans its own slots.
__global__ void __launch_bounds__(kBlock) k_tri_starts(TriArrays T) {
  __shared__ uint32_t red[kBlock / 64], wsum[kBlock / 64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t before = 0;
  for (uint32_t b = threadIdx.x; b < blockIdx.x; b += kBlock) before += T.tileSum[b];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) before += __shfl_xor(before, off, 64);
  if (lane == 0u) red[wave] = before;
  // this thread's 8 consecutive slots
  const size_t first = static_cast<size_t>(kGridTile) * blockIdx.x + 8u * threadIdx.x;
  const uint4 a = *reinterpret_cast<const uint4*>(T.cellCnt + first), b = *reinterpret_cast<const uint4*>(T.cellCnt + first + 4);
  const uint32_t mine = (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(incl, off, 64);
    if (lane >= static_cast<uint32_t>(off)) incl += v;
  }
  if (lane == 63u) wsum[wave] = incl;
  __syncthreads();
  uint32_t run = (red[0] + red[1]) + (red[2] + red[3]);
  for (uint32_t w = 0; w < wave; ++w) run += wsum[w];
  run += incl - mine;
  uint4 oa, ob;
  oa.x = run; oa.y = oa.x + a.x; oa.z = oa.y + a.y; oa.w = oa.z + a.z;
  ob.x = oa.w + a.w; ob.y = ob.x + b.x; ob.z = ob.y + b.y; ob.w = ob.z + b.z;
  *reinterpret_cast<uint4*>(T.cellStart + first) = oa;
  *reinterpret_cast<uint4*>(T.cellStart + first + 4) = ob;
  if (blockIdx.x == gridDim.x - 1u && threadIdx.x == kBlock - 1u) T.cellStart[T.slots] = ob.w + b.w;
}
__global__ void __launch_bounds__(kBlock) k_tri_fill(TriArrays T) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= T.nt) return;
  const uint32_t s = T.cellOf[t];
  if (s == kNil) return;
  const uint32_t at = T.cellStart[s] + T.posIn[t];
#pragma unroll
  for (int i = 0; i < 4; ++i) T.ent[4 * at + i] = T.boxOf[4 * t + i];
}

// Position of triangle t in the reference's merge order: thread (t mod T) owns triangles t, t + T, ... and the
// per-thread lists are concatenated thread after thread (Solver.cpp:714, 852).  cntTri / offTri are indexed by it.
PIES_DEV uint32_t merge_rank(uint32_t t, uint32_t nt, uint32_t threads) {
  const uint32_t th = t % threads, q = nt / threads, rem = nt % threads;
  return th * q + min(th, rem) + t / threads;
}

struct CellBox {  // a range of world cells [lo, hi) per axis
  int x0, y0, z0, x1, y1, z1;
};
PIES_DEV CellBox range_box(const int4& r, uint32_t lx, uint32_t ly, uint32_t lz) {
  return {r.x, r.y, r.z, r.x + static_cast<int>(lx), r.y + static_cast<int>(ly), r.z + static_cast<int>(lz)};
}
PIES_DEV int4 ent_range(const float4& e) { return make_int4(__float_as_int(e.x), __float_as_int(e.y), __float_as_int(e.z), __float_as_int(e.w)); }
PIES_DEV CellBox insert_box(const int4& r) { return range_box(r, r.w & 0xff, (r.w >> 8) & 0xff, (r.w >> 16) & 0xff); }
PIES_DEV CellBox meet(const CellBox& a, const CellBox& b) {
  return {max(a.x0, b.x0), max(a.y0, b.y0), max(a.z0, b.z0), min(a.x1, b.x1), min(a.y1, b.y1), min(a.z1, b.z1)};
}
PIES_DEV bool empty(const CellBox& b) { return b.x1 <= b.x0 || b.y1 <= b.y0 || b.z1 <= b.z0; }
PIES_DEV uint32_t volume(const CellBox& b) { return static_cast<uint32_t>((b.x1 - b.x0) * (b.y1 - b.y0) * (b.z1 - b.z0)); }
PIES_DEV bool holds(const CellBox& b, int x, int y, int z) { return x >= b.x0 && x < b.x1 && y >= b.y0 && y < b.y1 && z >= b.z0 && z < b.z1; }
// cells of the (non-empty) box b that come before cell (x, y, z) when a range is walked x-major (dx, dy, dz: Solver.cpp:722-724)
PIES_DEV uint32_t cells_before(const CellBox& b, int x, int y, int z) {
  const int nx = b.x1 - b.x0, ny = b.y1 - b.y0, nz = b.z1 - b.z0;
  int n = min(max(x - b.x0, 0), nx) * ny * nz;
  if (x >= b.x0 && x < b.x1) {
    n += min(max(y - b.y0, 0), ny) * nz;
    if (y >= b.y0 && y < b.y1) n += min(max(z - b.z0, 0), nz);
  }
  return static_cast<uint32_t>(n);
}

// the window of class L a partner of a triangle with range box S can be listed in: its minimum corner lies in S grown downwards
// by (the longest range listed in the class - 1) cells of the class
struct Window {
  int x0, y0, z0;
  uint32_t wx, wy, wz;
};
PIES_DEV Window window_of(const CellBox& S, uint32_t shift, uint32_t longest) {
  Window w;
  const int grow = static_cast<int>(longest) - 1;
  w.x0 = (S.x0 >> shift) - grow; w.y0 = (S.y0 >> shift) - grow; w.z0 = (S.z0 >> shift) - grow;
  w.wx = static_cast<uint32_t>(((S.x1 - 1) >> shift) - w.x0 + 1);
  w.wy = static_cast<uint32_t>(((S.y1 - 1) >> shift) - w.y0 + 1);
  w.wz = static_cast<uint32_t>(((S.z1 - 1) >> shift) - w.z0 + 1);
  return w;
}

// The entries of one row of a window - the cells (x, y, z0 .. z0 + wz) of a class - are one stretch of the entry list, or two
// when the row runs past the end of the table's z axis and goes on at its start.
struct RowRun {
  uint32_t s1, c1, s2, c2;
};
PIES_DEV RowRun row_run(const TriArrays& T, const TriGridLevel& L, int x, int y, int z0, uint32_t wz) {
  const uint32_t ux = static_cast<uint32_t>(x) & ((1u << L.lx) - 1u), uy = static_cast<uint32_t>(y) & ((1u << L.ly) - 1u);
  const uint32_t nz = 1u << L.lz, uz = static_cast<uint32_t>(z0) & (nz - 1u);
  const uint32_t row = L.base + (((ux << L.ly) | uy) << L.lz);
  RowRun r;
  r.s1 = T.cellStart[row + uz];
  if (uz + wz <= nz) {
    r.c1 = T.cellStart[row + uz + wz] - r.s1;
    r.s2 = 0; r.c2 = 0;
  } else {
    r.c1 = T.cellStart[row + nz] - r.s1;
    r.s2 = T.cellStart[row];
    r.c2 = T.cellStart[row + (uz + wz - nz)] - r.s2;
  }
  return r;
}

// Solver.cpp:751-755 - a bucket of more than 1000 triangles fails the sim - for a triangle whose windows list more than 1000
// triangles (otherwise no cell of its range can be in that many ranges): the exact number of ranges every cell of its range lies
// in.  Rare (a collapsed mesh) and slow: per cell of the range, every listed triangle of the windows.
template <int TEAM>
PIES_DEV bool crowded_cell(const TriArrays& T, const CellBox& S, uint32_t member) {
  for (int x = S.x0; x < S.x1; ++x)
    for (int y = S.y0; y < S.y1; ++y)
      for (int z = S.z0; z < S.z1; ++z) {
        uint32_t cover = 0;
        for (int k = 0; k < kTriLevels; ++k) {
          if (T.counters[13 + k] == 0u) continue;
          const Window W = window_of(S, T.level[k].shift, T.counters[10 + k]);
          const uint32_t rows = W.wx * W.wy;
          for (uint32_t w = member; w < rows; w += TEAM) {
            const uint32_t dx = w / W.wy, dy = w - dx * W.wy;
            const RowRun r = row_run(T, T.level[k], W.x0 + (int)dx, W.y0 + (int)dy, W.z0, W.wz);
            for (uint32_t e = 0; e < r.c1 + r.c2; ++e) {
              const uint32_t at = e < r.c1 ? r.s1 + e : r.s2 + (e - r.c1);
              if (holds(insert_box(ent_range(T.ent[4 * at + 2])), x, y, z)) ++cover;
            }
          }
        }
#pragma unroll
        for (int off = TEAM / 2; off >= 1; off >>= 1) cover += __shfl_xor(cover, off, TEAM);
        if (cover > 1000u) return true;
        if (T.counters[3]) return false;  // somebody else has failed the sim already
      }
  return false;
}

// ---- detection (Solver.cpp:714-797).  The reference walks the cells of a triangle's range and tests every triangle listed
// there without a common node, corner by corner; a pair that shares c cells is tested c times with the same result.  Two launches:
//
// k_tri_pairs  TEAM lanes share a triangle.  The team walks the windows of the three size classes row by row, TEAM rows at a time
//              (a lane per row: where its entries are and how many), forms the running offsets and walks the rows' entries as ONE
//              list, TEAM at a time; a partner is met once: its box - in the entry - against the box of the triangle's three
//              swept corners first (most partners of a window end here), then the ranges' intersection (none: the reference
//              never pairs them), the common nodes, the corners one by one.  What is left - a pair and the corners to test - goes
//              into the work list.  Nothing here is heavy (with the CCD inlined the kernel held 192 registers - two wavefronts per
//              SIMD - and took 43 us of a 470 us substep; 20 us now, the CCD's launch 8 on top of its floor).
// k_tri_ccd    one lane per entry of the work list: the CCD of the listed corners (Solver.cpp:775-797); a pair with hits goes
//              into the pool of hit records {triangle, partner, corners}, chained per triangle, and adds (hits x shared cells)
//              to the triangle's contacts (cntTri).  k_tri_emit writes the list from the records.
constexpr uint32_t kPairStage = 1024;  // pairs a workgroup (16 triangles) collects in LDS; a triangle of a moving lattice leaves ~15
// The work list is kWorkShards lists with a counter each, 64 bytes apart (workCnt[16 s]): atomics on ONE word take their turns at
// ~9 ns each on this chip (measured: 29 000 of them made a 10 us kernel a 260 us one), and so do the loads of its neighbours in
// the line; a workgroup appends to the list of its index modulo kWorkShards.
PIES_DEV uint32_t shard_capacity(const TriArrays& T) { return T.maxWork / kWorkShards; }
PIES_DEV void work_append(const TriArrays& T, uint32_t shard, uint2 item) {
  for (uint32_t tries = 0; tries < kWorkShards; ++tries, shard = (shard + 1u) % kWorkShards) {  // (a full list passes the item on)
    const uint32_t at = atomicAdd(&T.workCnt[16u * shard], 1u);
    if (at < shard_capacity(T)) { T.work[shard * shard_capacity(T) + at] = item; return; }
  }
  atomicOr(&T.counters[3], 64u);
}
template <int TEAM>
__global__ void __launch_bounds__(kBlock) k_tri_pairs(TriArrays T, const float4* __restrict__ pos, const float4* __restrict__ prev, float threshold) {
  // the workgroup's part of the work list is collected in LDS and appended at the end: one atomic per workgroup on the list's
  // counter (an append per wavefront and round of candidates put 200 000 atomics on that one word: 370 us)
  __shared__ uint2 sWork[kPairStage];
  __shared__ uint32_t sCount, sBase;
  if (threadIdx.x == 0) sCount = 0;
  __syncthreads();
  const uint32_t t = (blockIdx.x * kBlock + threadIdx.x) / TEAM, member = threadIdx.x % TEAM;
  bool search = t < T.nt && T.counters[3] == 0u;  // a team lies inside one wavefront and stays together
  int4 rg = make_int4(0, 0, 0, 0);
  uint32_t lx = 0, ly = 0, lz = 0;
  if (search) {
    if (member == 0) T.cntTri[merge_rank(t, T.nt, T.threadCount)] = 0;
    rg = T.rng[t];
    lx = rg.w & 0xff; ly = (rg.w >> 8) & 0xff; lz = (rg.w >> 16) & 0xff;
    if (lx > kTriSearchMaxCells || ly > kTriSearchMaxCells || lz > kTriSearchMaxCells) lx = ly = lz = 0;  // sweptTriRange: empty (Solver.cpp:672-674)
    const uint32_t ncell = lx * ly * lz;
    if (ncell == 0u || ncell > 1000u) {
      // Solver.cpp:741-745: more than 1000 buckets in a range fails the sim - every cell of a searching triangle's range has a
      // bucket, the triangle's own entry
      if (ncell && member == 0) atomicOr(&T.counters[3], 16u);
      search = false;
    }
  }
  if (search) {
  const CellBox S = range_box(rg, lx, ly, lz);
  const uint32_t ia[3] = {T.tris[3 * t], T.tris[3 * t + 1], T.tris[3 * t + 2]};
  F3 slo[3], shi[3];  // the corners' swept segments as boxes
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const F3 p1 = xyz(pos[ia[i]]), p0 = xyz(prev[ia[i]]);
    slo[i] = {fminf(p0.x, p1.x), fminf(p0.y, p1.y), fminf(p0.z, p1.z)};
    shi[i] = {fmaxf(p0.x, p1.x), fmaxf(p0.y, p1.y), fmaxf(p0.z, p1.z)};
  }
  const F3 tlo = {fminf(fminf(slo[0].x, slo[1].x), slo[2].x), fminf(fminf(slo[0].y, slo[1].y), slo[2].y), fminf(fminf(slo[0].z, slo[1].z), slo[2].z)};
  const F3 thi = {fmaxf(fmaxf(shi[0].x, shi[1].x), shi[2].x), fmaxf(fmaxf(shi[0].y, shi[1].y), shi[2].y), fmaxf(fmaxf(shi[0].z, shi[1].z), shi[2].z)};
  uint32_t listed = 0;
  for (int k = 0; k < kTriLevels; ++k) {
    if (T.counters[13 + k] == 0u) continue;
    const TriGridLevel L = T.level[k];
    const Window W = window_of(S, L.shift, T.counters[10 + k]);
    const uint32_t rows = W.wx * W.wy;
    const float invWy = 1.0f / static_cast<float>(W.wy);
    for (uint32_t w0 = 0; w0 < rows; w0 += TEAM) {
      // this lane's row of the round
      const uint32_t w = w0 + member;
      RowRun run = {0u, 0u, 0u, 0u};
      if (w < rows) {
        const uint32_t dx = static_cast<uint32_t>((static_cast<float>(w) + 0.5f) * invWy), dy = w - dx * W.wy;  // (rows < 2^10: exact)
        run = row_run(T, L, W.x0 + (int)dx, W.y0 + (int)dy, W.z0, W.wz);
      }
      const uint32_t bc = run.c1 + run.c2;
      // offsets of the round's rows in the round's candidate list
      uint32_t incl = bc;
#pragma unroll
      for (int off = 1; off < TEAM; off <<= 1) {
        const uint32_t below = __shfl_up(incl, off, TEAM);
        if (static_cast<int>(member) >= off) incl += below;
      }
      const uint32_t pre = incl - bc, total = __shfl(incl, TEAM - 1, TEAM);
      listed += total;
      for (uint32_t k0 = 0; k0 < total; k0 += TEAM) {  // (total is the team's: its lanes stay together)
        const uint32_t kk = k0 + member;
        // where candidate kk sits: its row is the last one of the round whose offset is <= kk (empty rows share their
        // offset with the next one, so "the last" is never one of them): a binary search over the team's lanes (every
        // lane takes part in every shuffle - a lane past the end of the list searches too and drops the result)
        uint32_t l = 0;
#pragma unroll
        for (int step = TEAM / 2; step >= 1; step >>= 1) {
          const uint32_t pm = __shfl(pre, static_cast<int>(l) + step, TEAM);
          if (pm <= kk) l += static_cast<uint32_t>(step);
        }
        const uint32_t pl = __shfl(pre, static_cast<int>(l), TEAM);
        const uint32_t rs1 = __shfl(run.s1, static_cast<int>(l), TEAM), rc1 = __shfl(run.c1, static_cast<int>(l), TEAM), rs2 = __shfl(run.s2, static_cast<int>(l), TEAM);
        uint32_t o = 0, tests = 0;
        if (kk < total) {
          const uint32_t at = kk - pl < rc1 ? rs1 + (kk - pl) : rs2 + (kk - pl - rc1);
          const float4 e0 = T.ent[4 * at], e1 = T.ent[4 * at + 1], e2 = T.ent[4 * at + 2], e3 = T.ent[4 * at + 3];  // (all four at once: no load waits for another)
          // Conservative reject before anything else.  A hit puts a corner, at some time in [0,1], within `threshold` of a
          // point of the moving partner (proximity branch: at t = 1; crossing branch: on it at the root), so the corner's
          // swept segment must meet the box of the partner's six corner positions grown by the threshold; the margin adds 5 %
          // and 1e-3 of the box on top of that, orders of magnitude above the rounding of the barycentric test.  First all
          // three segments at once (their common box), then one by one.  A partner with a vanishing normal (NaN inside the
          // CCD, which then cannot say "outside") is never rejected here, nor is anything non-finite: every comparison below
          // is false for NaN.
          const bool regular = e1.w != 0.0f;
          const F3 lo = {e0.x, e0.y, e0.z}, hi = {e1.x, e1.y, e1.z};
          const float margin = 1.05f * threshold + 1.0e-3f * fmaxf(fmaxf(hi.x - lo.x, hi.y - lo.y), hi.z - lo.z);
          const bool apartAll = (thi.x < lo.x - margin) || (tlo.x > hi.x + margin) || (thi.y < lo.y - margin) || (tlo.y > hi.y + margin) ||
                                (thi.z < lo.z - margin) || (tlo.z > hi.z + margin);
          if (!(regular && apartAll)) {
            o = __float_as_uint(e0.w);
            if (!empty(meet(S, insert_box(ent_range(e2))))) {  // (no cell lists both: the reference never sees the pair)
              const uint32_t ib = __float_as_uint(e3.x), ic = __float_as_uint(e3.y), idd = __float_as_uint(e3.z);
              bool common = false;
#pragma unroll
              for (int i = 0; i < 3; ++i) common = common || ia[i] == ib || ia[i] == ic || ia[i] == idd;
              if (!common) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                  const bool apart = (shi[i].x < lo.x - margin) || (slo[i].x > hi.x + margin) || (shi[i].y < lo.y - margin) || (slo[i].y > hi.y + margin) ||
                                     (shi[i].z < lo.z - margin) || (slo[i].z > hi.z + margin);
                  if (!(regular && apart)) tests |= 1u << i;
                }
              }
            }
          }
        }
        // the pairs that are left: one LDS append per wavefront for those of its lanes
        const unsigned long long keep = __ballot(tests != 0u);
        if (keep) {
          const int leader = __builtin_ctzll(keep);
          uint32_t base = 0;
          if (static_cast<int>(threadIdx.x & 63u) == leader) base = atomicAdd(&sCount, static_cast<uint32_t>(__popcll(keep)));
          base = __shfl(base, leader, 64);
          if (tests) {
            const uint32_t at = base + static_cast<uint32_t>(__popcll(keep & ((1ull << (threadIdx.x & 63u)) - 1ull)));
            const uint2 item = make_uint2(t, o | (tests << 29));
            if (at < kPairStage) sWork[at] = item;
            else work_append(T, blockIdx.x % kWorkShards, item);  // (more than the stage holds: straight to the list)
          }
        }
      }
    }
  }
  // Solver.cpp:751-755 (a bucket of more than 1000 triangles): the triangles whose ranges hold a cell are all listed in the windows
  if (listed > 1000u && crowded_cell<TEAM>(T, S, member) && member == 0) atomicOr(&T.counters[3], 16u);
  }
  __syncthreads();
  const uint32_t staged = min(sCount, kPairStage);
  // The workgroup's list, and - when that one is full (a mesh folded locally fills ONE list while the others are nearly empty:
  // ADVICE r4) - the lists behind it, each taking what it has room for; only when every list is full is the work list as a whole
  // (128 pairs per triangle) exhausted, which is as fatal as the contact list it would overflow.  (A list's counter may run past its
  // capacity: k_tri_ccd reads min(counter, capacity) entries, and exactly those were written.)
  const uint32_t cap = shard_capacity(T);
  uint32_t done = 0;
  for (uint32_t tries = 0; tries < kWorkShards && done < staged; ++tries) {  // (workgroup uniform)
    const uint32_t shard = (blockIdx.x + tries) % kWorkShards;
    __syncthreads();
    if (threadIdx.x == 0) sBase = atomicAdd(&T.workCnt[16u * shard], staged - done);
    __syncthreads();
    const uint32_t base = min(sBase, cap), fit = min(staged - done, cap - base);
    for (uint32_t i = threadIdx.x; i < fit; i += kBlock) T.work[shard * cap + base + i] = sWork[done + i];
    done += fit;
  }
  if (done < staged && threadIdx.x == 0) atomicOr(&T.counters[3], 64u);
}

__global__ void __launch_bounds__(kBlock) k_tri_ccd(TriArrays T, const float4* __restrict__ pos, const float4* __restrict__ prev, float threshold) {
  if (T.counters[3]) return;
  // workgroup b works on list b % kWorkShards, with the other workgroups of that list
  const uint32_t shard = blockIdx.x % kWorkShards, cap = shard_capacity(T);
  const uint32_t n = min(T.workCnt[16u * shard], cap);
  for (uint32_t w = (blockIdx.x / kWorkShards) * kBlock + threadIdx.x; w < n; w += (gridDim.x / kWorkShards) * kBlock) {
    const uint2 item = T.work[shard * cap + w];
    const uint32_t t = item.x, o = item.y & 0x1fffffffu, tests = item.y >> 29;
    const uint32_t ia[3] = {T.tris[3 * t], T.tris[3 * t + 1], T.tris[3 * t + 2]};
    const uint32_t ib = T.tris[3 * o], ic = T.tris[3 * o + 1], idd = T.tris[3 * o + 2];
    const F3 b1 = xyz(pos[ib]), c1 = xyz(pos[ic]), d1 = xyz(pos[idd]);
    const F3 b0 = xyz(prev[ib]), c0 = xyz(prev[ic]), d0 = xyz(prev[idd]);
    uint32_t hits = 0;
    for (int i = 0; i < 3; ++i) {  // (not unrolled: one copy of the CCD)
      if (!(tests >> i & 1u)) continue;
      const F3 p1 = xyz(pos[ia[i]]), p0 = xyz(prev[ia[i]]);
      if (point_triangle_ccd(p0 - b0, c0 - b0, d0 - b0, p1 - b1, c1 - b1, d1 - b1, threshold)) hits |= 1u << i;
    }
    // the hit records: one append per wavefront
    const unsigned long long hot = __ballot(hits != 0u);
    if (hot == 0ull) continue;
    const int leader = __builtin_ctzll(hot);
    uint32_t base = 0;
    if (static_cast<int>(threadIdx.x & 63u) == leader) base = atomicAdd(&T.counters[9], static_cast<uint32_t>(__popcll(hot)));
    base = __shfl(base, leader, 64);
    if (hits == 0u) continue;
    const int4 rg = T.rng[t];
    const uint32_t shared = volume(meet(insert_box(rg), insert_box(T.rng[o])));  // (a triangle on the work list searched: its range is the listed one)
    atomicAdd(&T.cntTri[merge_rank(t, T.nt, T.threadCount)], static_cast<uint32_t>(__popc(hits)) * shared);
    const uint32_t rec = base + static_cast<uint32_t>(__popcll(hot & ((1ull << (threadIdx.x & 63u)) - 1ull)));
    if (rec < T.maxContacts) T.pool[rec] = make_uint4(t, o, hits, atomicExch(&T.head[t], rec));
    else atomicOr(&T.counters[3], 64u);  // more records than contacts fit the list: the list overflows as well
  }
}

// The contact list from the hit records (Solver.cpp:721-797 order: the cells of the triangle's range x-major, in each cell the
// listed triangles by ascending index, for each the triangle's corners 0, 1, 2).  One thread per record {t, o, corners}: for every
// cell both ranges hold, the entry's position = the triangle's offset + what all of the triangle's records (its chain) put in
// front of it: a record's hits x (the cells it shares with t that come earlier + this cell, if it holds it and the partner's
// index is smaller).
PIES_DEV void tri_emit(const TriArrays& T, uint32_t first, uint32_t stride) {
  const uint32_t nrec = min(T.counters[9], T.maxContacts);
  for (uint32_t r = first; r < nrec; r += stride) {
    const uint4 rec = T.pool[r];
    const uint32_t t = rec.x, o = rec.y;
    const int4 rg = T.rng[t];
    const CellBox S = insert_box(rg);  // (a triangle with records searched: its range is the one it was listed with)
    const CellBox mine = meet(S, insert_box(T.rng[o]));
    const uint32_t base = T.offTri[merge_rank(t, T.nt, T.threadCount)];
    const uint32_t ia[3] = {T.tris[3 * t], T.tris[3 * t + 1], T.tris[3 * t + 2]};
    const uint32_t ib = T.tris[3 * o], ic = T.tris[3 * o + 1], idd = T.tris[3 * o + 2];
    for (int x = mine.x0; x < mine.x1; ++x)
      for (int y = mine.y0; y < mine.y1; ++y)
        for (int z = mine.z0; z < mine.z1; ++z) {
          uint32_t c = base;
          for (uint32_t q = T.head[t]; q != kNil;) {
            const uint4 other = T.pool[q];
            const CellBox theirs = meet(S, insert_box(T.rng[other.y]));
            c += static_cast<uint32_t>(__popc(other.z)) * (cells_before(theirs, x, y, z) + ((other.y < o && holds(theirs, x, y, z)) ? 1u : 0u));
            q = other.w;
          }
#pragma unroll
          for (int i = 0; i < 3; ++i)
            if (rec.z >> i & 1u) {
              if (c < T.maxContacts) T.ids[c] = make_uint4(ia[i], ib, ic, idd);
              ++c;
            }
        }
  }
}

// exclusive scan of the per-triangle counts in the reference's merge order (one block).  Every wavefront takes one contiguous
// sixteenth of the counts: a coalesced sweep for its sum, the sixteen sums in LDS, and - only for a wavefront whose range holds
// contacts at all - a second sweep that writes the offsets (wave-wide scans over 64 counts at a time).  A substep without
// contacts is one sweep of loads that do not depend on each other (the first version gave every thread 41 consecutive counts
// and a 10-step Hillis-Steele scan over the threads: 10 us at 42k triangles).
__global__ void __launch_bounds__(kBlock) k_tri_emit(TriArrays T) {
  if (T.counters[3]) return;
  tri_emit(T, blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock);
}
// (a workgroup of 1024; returns the length of the contact list)
PIES_DEV uint32_t tri_scan(const TriArrays& T, uint32_t* wsum) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, nt = T.nt;
  const uint32_t per = (((nt + 15u) / 16u) + 63u) & ~63u;
  const uint32_t lo = min(nt, wave * per), hi = min(nt, lo + per);
  uint32_t sum = 0;
  {  // four independent 16-byte loads per lane and round: the sweep is a chain of load latencies (one word per lane and round took
     // 38 us for the 100k triangles of config 5's body)
    const uint4* c4 = reinterpret_cast<const uint4*>(T.cntTri + lo);  // (lo is a multiple of 64)
    const uint32_t quads = (hi - lo) / 4u;
    uint32_t q = lane;
    for (; q + 192u < quads; q += 256u) {
      const uint4 a = c4[q], b = c4[q + 64u], c = c4[q + 128u], d = c4[q + 192u];
      sum += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
    }
    for (; q < quads; q += 64u) { const uint4 a = c4[q]; sum += (a.x + a.y) + (a.z + a.w); }
    for (uint32_t r = lo + 4u * quads + lane; r < hi; r += 64u) sum += T.cntTri[r];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0u) wsum[wave] = sum;
  __syncthreads();
  uint32_t run = 0, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < 16u; ++w) {
    const uint32_t v = wsum[w];
    if (w < wave) run += v;
    total += v;
  }
  if (sum != 0u) {  // (offsets of a range without contacts are never read: the fill pass skips empty triangles)
    for (uint32_t r0 = lo; r0 < hi; r0 += 64u) {
      const uint32_t r = r0 + lane;
      const uint32_t c = r < hi ? T.cntTri[r] : 0u;
      if (__builtin_amdgcn_ballot_w64(c != 0u) == 0ull) continue;  // (64 triangles without a contact: their offsets are never read)
      uint32_t inc = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(inc, off, 64);
        if (lane >= static_cast<uint32_t>(off)) inc += v;
      }
      if (r < hi) T.offTri[r] = run + inc - c;
      run += __shfl(inc, 63, 64);
    }
  }
  if (tid == 0u) {
    if (total > T.maxContacts) atomicOr(&T.counters[3], 64u);  // contact list overflow: latch
    T.counters[2] = min(total, T.maxContacts);
  }
  return total;
}
__global__ void __launch_bounds__(1024) k_tri_scan(TriArrays T) {
  __shared__ uint32_t wsum[16];
  if (T.counters[9] == 0u) {  // no hit record, no contact: the usual substep
    if (threadIdx.x == 0u) T.counters[2] = 0u;
    return;
  }
  tri_scan(T, wsum);
}

// ---- per-node incidence of the contacts + their diagonal blocks ---------------------------------------------
// Five steps with a device-wide dependency between them (count, used, alloc, fill, sort).  Each is a device function over
// (first item, stride); the contact-heavy graph variant runs them as five launches, the other one as ONE launch of a single
// workgroup (k_inc_all) - a substep with few or no contacts pays one launch for the lot instead of five.
PIES_DEV void inc_count(const TriArrays& T, float* __restrict__ cdiag, uint32_t first, uint32_t stride) {
  const uint32_t M = T.counters[2];
  for (uint32_t c = first; c < M; c += stride) {
    const uint4 id = T.ids[c];
    const uint32_t n[4] = {id.x, id.y, id.z, id.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (atomicAdd(&T.incCnt[n[i]], 1u) == 0u) atomicOr(&T.usedBits[n[i] >> 5], 1u << (n[i] & 31u));
      // diag(w A^T A) = w * (3, 1, 1, 1); multiples of 1e4 add exactly in float, so the order is irrelevant
      atomicAdd(&cdiag[n[i]], kTriContactW * (i == 0 ? 3.0f : 1.0f));
    }
  }
}
// The nodes that take part in contacts, in ascending order, from the bitmap inc_count marked them in: one workgroup of 1024, a
// thread per run of bitmap words (popcounts, a prefix sum over the threads, then the set bits in order).  An append in
// arrival order would be cheaper, but the order decides which wavefront sums which contact rows (k_cg_ap), i.e. the rounding
// of p.Ap: results would differ from run to run.
PIES_DEV void inc_used(const TriArrays& T, uint32_t words, uint32_t* part) {
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = (words + 1023u) / 1024u;
  const uint32_t lo = min(words, tid * chunk), hi = min(words, lo + chunk);
  uint32_t sum = 0;
  for (uint32_t w = lo; w < hi; ++w) sum += static_cast<uint32_t>(__popc(T.usedBits[w]));
  part[tid] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
    const uint32_t v = tid >= off ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t at = tid ? part[tid - 1] : 0u;
  if (sum)
    for (uint32_t w = lo; w < hi; ++w) {
      uint32_t bits = T.usedBits[w];
      while (bits) {
        const uint32_t b = static_cast<uint32_t>(__ffs(bits)) - 1u;
        T.usedNodes[at++] = (w << 5) + b;
        bits &= bits - 1u;
      }
    }
  if (tid == 1023) T.counters[4] = part[1023];
}
PIES_DEV void inc_alloc(const TriArrays& T, const float* __restrict__ kdiag, const float* __restrict__ cdiag, float* __restrict__ dinv,
                        uint32_t first, uint32_t stride) {
  const uint32_t used = T.counters[4];
  for (uint32_t u = first; u < used; u += stride) {
    const uint32_t n = T.usedNodes[u];
    T.incStart[n] = atomicAdd(&T.counters[5], T.incCnt[n]);
    T.incFill[n] = 0;
    T.nodeSlot[n] = u;  // the node's place in the LDS copy of the sequential passes
    dinv[n] = 1.0f / (kdiag[n] + cdiag[n]);
  }
}
PIES_DEV void inc_fill(const TriArrays& T, uint32_t first, uint32_t stride) {
  const uint32_t M = T.counters[2];
  for (uint32_t c = first; c < M; c += stride) {
    const uint4 id = T.ids[c];
    const uint32_t n[4] = {id.x, id.y, id.z, id.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) T.inc[T.incStart[n[i]] + atomicAdd(&T.incFill[n[i]], 1u)] = (c << 2) | static_cast<uint32_t>(i);
  }
}
PIES_DEV void inc_sort(const TriArrays& T, uint32_t wave, uint32_t nwaves) {
  const uint32_t used = T.counters[4];
  const int lane = threadIdx.x & 63;
  for (uint32_t u = wave; u < used; u += nwaves) {
    const uint32_t n = T.usedNodes[u];
    rank_sort(T.inc, T.incSorted, T.incStart[n], T.incCnt[n], lane, T.incPos);
  }
}
__global__ void __launch_bounds__(kBlock) k_inc_count(TriArrays T, float* __restrict__ cdiag) {
  inc_count(T, cdiag, blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock);
}
__global__ void __launch_bounds__(1024) k_inc_used(TriArrays T, uint32_t words) {
  __shared__ uint32_t part[1024];
  inc_used(T, words, part);
}
__global__ void __launch_bounds__(kBlock) k_inc_alloc(TriArrays T, const float* __restrict__ kdiag, const float* __restrict__ cdiag,
                                                      float* __restrict__ dinv) {
  inc_alloc(T, kdiag, cdiag, dinv, blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock);
}
__global__ void __launch_bounds__(kBlock) k_inc_fill(TriArrays T) { inc_fill(T, blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock); }
__global__ void __launch_bounds__(kBlock) k_inc_sort(TriArrays T) {
  inc_sort(T, (blockIdx.x * kBlock + threadIdx.x) >> 6, (gridDim.x * kBlock) >> 6);
}
// the five steps by one workgroup: a step's stores are in L2, and this compute unit's vector cache holds none of the lines
// they went to, before the next step starts (the fence), and every wavefront has finished the step (the barrier)
PIES_DEV void step_boundary() {
  __threadfence();
  __syncthreads();
}
__global__ void __launch_bounds__(1024) k_inc_all(TriArrays T, const float* __restrict__ kdiag, float* __restrict__ cdiag, float* __restrict__ dinv,
                                                  uint32_t words) {
  __shared__ uint32_t part[1024];
  if (T.counters[2] == 0u) return;  // no contact in this substep (k_tri_box has cleared the counts this chain would write)
  inc_count(T, cdiag, threadIdx.x, 1024u);
  step_boundary();
  inc_used(T, words, part);
  step_boundary();
  inc_alloc(T, kdiag, cdiag, dinv, threadIdx.x, 1024u);
  step_boundary();
  inc_fill(T, threadIdx.x, 1024u);
  step_boundary();
  inc_sort(T, threadIdx.x >> 6, 16u);
}
// The graph variant for substeps with few or no contacts: the offsets of the contact list, the list and the incidence chain by
// ONE workgroup (three launches fewer; a substep without a hit record - the usual one - leaves at the first line).
PIES_DEV bool tri_tail(const TriArrays& T, const float* __restrict__ kdiag, float* __restrict__ cdiag, float* __restrict__ dinv, uint32_t words,
                       uint32_t* part) {  // (a workgroup of 1024; part: 1024 words of LDS; false: no contact list in this substep)
  if (T.counters[9] == 0u) {
    if (threadIdx.x == 0u) T.counters[2] = 0u;
    return false;
  }
  const uint32_t total = tri_scan(T, part);
  if (total > T.maxContacts || T.counters[3]) return false;  // (latched: the tick ends as a failure)
  step_boundary();
  tri_emit(T, threadIdx.x, 1024u);
  step_boundary();
  inc_count(T, cdiag, threadIdx.x, 1024u);
  step_boundary();
  inc_used(T, words, part);
  step_boundary();
  inc_alloc(T, kdiag, cdiag, dinv, threadIdx.x, 1024u);
  step_boundary();
  inc_fill(T, threadIdx.x, 1024u);
  step_boundary();
  inc_sort(T, threadIdx.x >> 6, 16u);
  return true;
}
__global__ void __launch_bounds__(1024) k_tri_tail(TriArrays T, const float* __restrict__ kdiag, float* __restrict__ cdiag, float* __restrict__ dinv,
                                                   uint32_t words) {
  __shared__ uint32_t part[1024];
  tri_tail(T, kdiag, cdiag, dinv, words, part);
}

// ---- merged contact rows -------------------------------------------------------------------------------------
// The off-diagonal part of a node's row of the contact matrix, w * (AtA)_i. summed over its contacts, with equal columns
// merged: a node of a contact patch sits in tens of contacts over a handful of triangles, so its ~100 row entries name
// ~15 distinct nodes.  The CG sums these rows once per iteration (k_cg_ap's row blocks): with the merged form a row is one
// gather of at most a wavefront's width instead of three dependent loads per contact.  One wavefront per node: the raw
// columns go through a small LDS hash with counts (integer adds: exact), the distinct ones are sorted by column (ranks
// among at most kRowMaxUnique entries), so the stored order - and with it the rounding of the sum - does not depend on
// which lane got where first.  A node with more distinct columns than kRowMaxUnique keeps the contact-by-contact form
// (rowLen = kRowUnmerged).  Storage: 6 entries per contact at most.
constexpr uint32_t kRowSlots = 512, kRowMaxUnique = 256, kRowUnmerged = 0xffffffffu;
__global__ void __launch_bounds__(kBlock) k_contact_csr(TriArrays T, uint32_t maxUnique) {
  __shared__ uint32_t hkey[kBlock / 64][kRowSlots];
  __shared__ uint32_t hcnt[kBlock / 64][kRowSlots];
  __shared__ uint32_t ucol[kBlock / 64][kRowMaxUnique], ucnt[kBlock / 64][kRowMaxUnique];
  const uint32_t used = T.counters[4];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = (gridDim.x * kBlock) >> 6;
  for (uint32_t u = wave; u < used; u += nwaves) {
    const uint32_t node = T.usedNodes[u];
    const uint32_t tc = T.incCnt[node], ts = T.incStart[node];
    for (uint32_t t = lane; t < kRowSlots; t += 64) { hkey[w][t] = 0xffffffffu; hcnt[w][t] = 0; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t fresh = 0;
    for (uint32_t base = 0; base < tc; base += 64) {
      const uint32_t t = base + lane;
      uint32_t cols[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
      if (t < tc) {
        const uint32_t v = T.incSorted[ts + t];
        const uint4 id = T.ids[v >> 2];
        if ((v & 3u) == 0u) { cols[0] = id.y; cols[1] = id.z; cols[2] = id.w; }  // the point's row couples to the triangle's nodes,
        else cols[0] = id.x;                                                       // their rows to the point
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        bool mine = false;
        if (cols[q] != 0xffffffffu && fresh <= maxUnique) {
          uint32_t h = (cols[q] * 2654435761u) >> 23;  // 9 bits
          for (;;) {
            const uint32_t old = atomicCAS(&hkey[w][h], 0xffffffffu, cols[q]);
            if (old == 0xffffffffu) { mine = true; break; }
            if (old == cols[q]) break;
            h = (h + 1u) & (kRowSlots - 1u);
          }
          atomicAdd(&hcnt[w][h], 1u);
        }
        fresh += static_cast<uint32_t>(__popcll(__ballot(mine)));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (fresh > maxUnique) {  // (uniform: every lane counted the same ballots)
      if (lane == 0) T.rowLen[node] = kRowUnmerged;
      continue;
    }
    // compact the occupied slots, then order them by column
    uint32_t nu = 0;
    for (uint32_t base = 0; base < kRowSlots; base += 64) {
      const uint32_t k = hkey[w][base + lane];
      const bool occ = k != 0xffffffffu;
      const unsigned long long m = __ballot(occ);
      if (occ) {
        const uint32_t at = nu + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
        ucol[w][at] = k;
        ucnt[w][at] = hcnt[w][base + lane];
      }
      nu += static_cast<uint32_t>(__popcll(m));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t off = 0;
    if (lane == 0) off = atomicAdd(&T.counters[8], nu);
    off = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(off)));
    for (uint32_t t = lane; t < nu; t += 64) {
      const uint32_t c = ucol[w][t];
      uint32_t rank = 0;
      for (uint32_t o = 0; o < nu; ++o) rank += ucol[w][o] < c ? 1u : 0u;  // columns are distinct
      T.rowCol[off + rank] = c;
      T.rowCoef[off + rank] = -kTriContactW * static_cast<float>(ucnt[w][t]);
    }
    if (lane == 0) { T.rowStart[node] = off; T.rowLen[node] = nu; }
  }
}

// ---- local step (CollisionConstraint.cpp:86-124) and w * (AtA p)_i (:176-194) ------------------------------
__global__ void __launch_bounds__(kBlock) k_pd_local_tri(TriArrays T, const float4* __restrict__ pos, float thickness) {
  const uint32_t M = T.counters[2];
  for (uint32_t c = blockIdx.x * kBlock + threadIdx.x; c < M; c += gridDim.x * kBlock) {
    const uint4 id = T.ids[c];
    F3 p[4] = {xyz(pos[id.x]), xyz(pos[id.y]), xyz(pos[id.z]), xyz(pos[id.w])};
    const F3 rel = p[0] - p[1];
    const F3 n = normalize(cross(p[2] - p[1], p[3] - p[1]));
    const float nDotP = dot(n, rel);
    if (nDotP < thickness) p[0] = p[0] + (thickness - nDotP) * n;
    // AtA = [[3,-1,-1,-1],[-1,1,0,0],[-1,0,1,0],[-1,0,0,1]], products accumulated from 0 in column order
    const float AtA[4][4] = {{3.f, -1.f, -1.f, -1.f}, {-1.f, 1.f, 0.f, 0.f}, {-1.f, 0.f, 1.f, 0.f}, {-1.f, 0.f, 0.f, 1.f}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ax += AtA[i][k] * p[k].x;
        ay += AtA[i][k] * p[k].y;
        az += AtA[i][k] * p[k].z;
      }
      T.contrib[4 * c + i] = make_float4(kTriContactW * ax, kTriContactW * ay, kTriContactW * az, 0.f);
    }
  }
}

// ---- sequential passes over the contact list, one wavefront ------------------------------------------------
PIES_DEV float ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PIES_DEV void st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PIES_DEV F3 ld3(const float* base, uint32_t node) { return {ld(base + 4 * node), ld(base + 4 * node + 1), ld(base + 4 * node + 2)}; }
PIES_DEV void st3(float* base, uint32_t node, F3 v) { st(base + 4 * node, v.x); st(base + 4 * node + 1, v.y); st(base + 4 * node + 2, v.z); }

// dependency level of every contact of a 64-contact window (contacts that share a node keep their list order)
PIES_DEV int window_levels(bool valid, const uint4& id, int lane, int& maxLevel, int base = 0) {
  int level = base;
  for (int m = 0; m < 63; ++m) {
    const int lm = __builtin_amdgcn_readlane(level, m);
    const uint32_t mx = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(id.x), m));
    const uint32_t my = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(id.y), m));
    const uint32_t mz = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(id.z), m));
    const uint32_t mw = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(id.w), m));
    const uint32_t mine[4] = {id.x, id.y, id.z, id.w};
    bool share = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) share = share || mine[k] == mx || mine[k] == my || mine[k] == mz || mine[k] == mw;
    if (valid && lane > m && share && mx != 0xffffffffu) level = max(level, lm + 1);
  }
  int mxl = valid ? level : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mxl = max(mxl, __shfl_xor(mxl, off, 64));
  maxLevel = mxl;
  return level;
}

// One contact of a sequential pass.  MODE 0: PointTriangleCollisionConstraint::stabilizeCollisions
// (CollisionConstraint.cpp:126-162); MODE 1: point-triangle friction (Solver.cpp:431-471).  The node state is reached
// through an accessor (L2 or the workgroup's LDS copy): im(l), pos(l), set_pos(l, v), second(l) / set_second(l, v) = the
// previous position (MODE 0) or the velocity (MODE 1) of the contact's node l = 0 (point), 1..3 (triangle).
template <int MODE, class IO>
PIES_DEV void tri_contact_step(IO& io, float thickness, float friction, float staticThreshold) {
  const float imA = io.im(0), imB = io.im(1), imC = io.im(2), imD = io.im(3);
  const F3 pa = io.pos(0), pb = io.pos(1), pc = io.pos(2), pd = io.pos(3);
  const F3 n = normalize(cross(pc - pb, pd - pb));
  const float wTri = imB + imC + imD, wSum = imA + wTri;
  if (MODE == 0) {
    const float nDotP = dot(n, pa - pb);
    if (nDotP < thickness) {
      const F3 disp = (thickness - nDotP) * n;
      const F3 da = disp * imA / wSum, dt = disp * wTri / wSum;
      io.set_pos(0, pa + da); io.set_pos(1, pb - dt); io.set_pos(2, pc - dt); io.set_pos(3, pd - dt);
      io.set_second(0, io.second(0) + da); io.set_second(1, io.second(1) - dt);
      io.set_second(2, io.second(2) - dt); io.set_second(3, io.second(3) - dt);
    }
  } else {
    const F3 va = io.second(0), vb = io.second(1), vc = io.second(2), vd = io.second(3);
    const F3 avg = (vb + vc + vd) / 3.0f;
    const F3 rel = va - avg;
    const float vDotN = dot(rel, n);
    const F3 perp = rel - vDotN * n;
    float fr = friction;
    if (sqrtf(dot(perp, perp)) < staticThreshold) fr = 1.0f;
    const F3 dv = (-fr) * perp - (1.1f * fminf(vDotN, 0.0f)) * n;
    const F3 ndv = neg(dv);
    io.set_second(0, va + dv * imA / wSum);
    io.set_second(1, vb + ndv * wTri / wSum);
    io.set_second(2, vc + ndv * wTri / wSum);
    io.set_second(3, vd + ndv * wTri / wSum);
  }
}
// node state in L2 (agent-scope loads and stores)
struct GlobalNodes {
  uint32_t id[4];
  float *p, *q;
  PIES_DEV float im(int l) const { return ld(p + 4 * id[l] + 3); }
  PIES_DEV F3 pos(int l) const { return ld3(p, id[l]); }
  PIES_DEV void set_pos(int l, F3 v) const { st3(p, id[l], v); }
  PIES_DEV F3 second(int l) const { return ld3(q, id[l]); }
  PIES_DEV void set_second(int l, F3 v) const { st3(q, id[l], v); }
};
template <int MODE>
PIES_DEV void tri_contact_step(const uint4 id, float* pos, float* prev, float* vel, float thickness, float friction, float staticThreshold) {
  GlobalNodes io = {{id.x, id.y, id.z, id.w}, pos, MODE == 0 ? prev : vel};
  tri_contact_step<MODE>(io, thickness, friction, staticThreshold);
}
// node state in the workgroup's LDS copy (records of four floats; the fourth of P is the inverse mass)
struct LdsNodes {
  uint32_t slot[4];
  float4 *P, *Q;
  PIES_DEV float im(int l) const { return P[slot[l]].w; }
  PIES_DEV F3 pos(int l) const { const float4 v = P[slot[l]]; return {v.x, v.y, v.z}; }
  PIES_DEV void set_pos(int l, F3 v) const { float4& d = P[slot[l]]; d.x = v.x; d.y = v.y; d.z = v.z; }
  PIES_DEV F3 second(int l) const { const float4 v = Q[slot[l]]; return {v.x, v.y, v.z}; }
  PIES_DEV void set_second(int l, F3 v) const { float4& d = Q[slot[l]]; d.x = v.x; d.y = v.y; d.z = v.z; }
};
// workgroup barrier that waits for this wavefront's LDS traffic only (requests to global memory stay in flight)
PIES_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Dependency levels of the whole contact list, once per substep: level(c) = 1 + the highest level among the earlier
// contacts that share a node with c, so that running the list level by level, in any order inside a level, is the
// reference's sequential pass.  A node's contacts are already listed in ascending order (incSorted) with every
// incidence's position in that list (incPos), so the earlier contact that matters for node n is the list entry before
// c: at most four predecessors per contact, all with smaller indices.  The list is relaxed 1024 contacts at a time,
// in order: predecessors in earlier chunks are final, those inside the chunk are iterated to the fixed point
// (as many rounds as the longest chain inside the chunk; levels live in LDS, 16 bit).  Lists longer than
// kLevelsLdsCap take the older walk: one wavefront, 64 contacts at a time, levels inside the window with 63
// cross-lane broadcasts on top of a per-node "level of the last earlier contact" array.  Then the contacts are
// bucketed by level.  More than kTriMaxLevels levels (thousands of contacts on one node) raise counters[7] and the
// passes fall back to the single-wavefront walk.
constexpr int kSeqBlock = 1024;
constexpr uint32_t kLevelsLdsCap = 49152;  // contacts whose 16-bit levels fit next to the histogram in LDS
// The fast path (the touched nodes fit the LDS copy of the sequential passes, at most kSeqLdsNodes, and the list has at most
// kLvMaxContacts entries): Kahn's algorithm with the NODES as owners.  A thread owns up to four touched nodes and walks each
// node's sorted contact list (incSorted) with a private cursor; heads[c] (one byte per contact, LDS) counts on how many of
// its four nodes contact c is the first unprocessed entry.  Round r: every owner looks at the contact its list stands at
// (heads == 4: all four owners see the same snapshot, so all four agree), barrier, the owners of ready contacts move on
// (heads of the next entry + 1) and the owner of the contact's point appends it to level r, barrier.  A round is two
// barriers and a few LDS operations, its list entries were requested four to eight rounds earlier; the number of rounds is
// the number of levels (the old path relaxed 1024 contacts at a time: the sum of the chunks' chain lengths, 6x as many
// rounds on a 29k-contact patch: 1.35 ms against 0.2 ms).
constexpr uint32_t kSeqLdsNodes = 4096, kLvMaxContacts = 65536;
PIES_DEV uint32_t ldu32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <uint32_t kLvNodesPerThread>
PIES_DEV void levels_by_node_owners(const TriArrays& T, uint32_t* heads32, uint32_t* sCnt, const uint32_t M, const uint32_t used, const int tid) {
  // Per owned node: `head` = the list entry the node stands at (position k), nx[0..3] = the entries behind it, pend[0..3]
  // = the four after those, requested at the last refill.  Global memory is only touched at refills - every four rounds,
  // by every thread at once (a node consumes at most one entry per round) - so that no wavefront waits for a load inside
  // a round; the level lists are written with plain stores.
  const uint8_t* heads8 = reinterpret_cast<const uint8_t*>(heads32);
  for (uint32_t w = tid; w < (M + 3) / 4; w += kSeqBlock) heads32[w] = 0;
  if (tid < 3) sCnt[tid] = 0;
  __syncthreads();
  uint32_t base[kLvNodesPerThread], cnt[kLvNodesPerThread], k[kLvNodesPerThread], kRefill[kLvNodesPerThread], head[kLvNodesPerThread];
  uint32_t nx[kLvNodesPerThread][4], pend[kLvNodesPerThread][4];
  auto entry_at = [&](uint32_t j, uint32_t at) {  // list entry `at` of owned node j (any in-range entry when past the end)
    const uint32_t last = cnt[j] ? cnt[j] - 1u : 0u;
    return T.incSorted[base[j] + min(at, last)];
  };
#pragma unroll
  for (uint32_t j = 0; j < kLvNodesPerThread; ++j) {
    const uint32_t u = tid + j * kSeqBlock;
    base[j] = 0; cnt[j] = 0; k[j] = 0; kRefill[j] = 0;
    if (u < used) {
      const uint32_t n = T.usedNodes[u];
      base[j] = T.incStart[n];
      cnt[j] = T.incCnt[n];
    }
    head[j] = entry_at(j, 0);
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) { nx[j][i] = entry_at(j, 1 + i); pend[j][i] = entry_at(j, 5 + i); }
    if (cnt[j]) {
      const uint32_t c = head[j] >> 2;
      atomicAdd(&heads32[c >> 2], 1u << (8u * (c & 3u)));
    }
  }
  if (tid == 0) T.lvStart[0] = 0;
  __syncthreads();
  uint32_t total = 0, r = 0;
  bool stuck = false;
  while (total < M && !stuck) {
    // refill: s = entries consumed since the last one (0..4); the entries requested then have had four rounds to arrive
#pragma unroll
    for (uint32_t j = 0; j < kLvNodesPerThread; ++j) {
      const uint32_t s = k[j] - kRefill[j];
      const uint32_t n0 = s < 4u ? nx[j][0] : pend[j][0];
      const uint32_t n1 = s < 3u ? nx[j][1] : (s == 3u ? pend[j][0] : pend[j][1]);
      const uint32_t n2 = s < 2u ? nx[j][2] : (s == 2u ? pend[j][0] : s == 3u ? pend[j][1] : pend[j][2]);
      const uint32_t n3 = s < 1u ? nx[j][3] : (s == 1u ? pend[j][0] : s == 2u ? pend[j][1] : s == 3u ? pend[j][2] : pend[j][3]);
      nx[j][0] = n0; nx[j][1] = n1; nx[j][2] = n2; nx[j][3] = n3;
      kRefill[j] = k[j];
      // (a node that consumed nothing since the last refill has its four requested entries already: a node of a 28k-contact
      // patch moves on in one round of ten, and 16 uncoalesced requests per thread every four rounds - 16 000 from ONE compute
      // unit - cost a sixth of the kernel: 1 063 -> 887 us for the 610 levels of that patch.  Fewer owning threads with more
      // nodes each are slower: 512 threads 1 126 us, 256 threads 2 906 us.  ONE barrier per round instead of two - two copies of the
      // counters, a round reads one and adds to the other, the copy it read gets the addition a round later - is slower too: 938 us:
      // a round is its LDS operations and its ~150 instructions per wavefront, sixteen wavefronts on one compute unit.)
      if (s != 0u) {
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) pend[j][i] = entry_at(j, k[j] + 5u + i);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // (unrolled: a loop here makes the compiler drain the refill's loads before entering it)
      if (total >= M) break;
      if (r >= kTriMaxLevels) { stuck = true; break; }
      bool ready[kLvNodesPerThread];
      uint32_t nEmit = 0;
#pragma unroll
      for (uint32_t j = 0; j < kLvNodesPerThread; ++j) {
        ready[j] = k[j] < cnt[j] && heads8[head[j] >> 2] == 4;
        nEmit += (ready[j] && (head[j] & 3u) == 0u) ? 1u : 0u;
      }
      lds_barrier();
      uint32_t at = nEmit ? total + atomicAdd(&sCnt[r % 3u], nEmit) : 0u;
#pragma unroll
      for (uint32_t j = 0; j < kLvNodesPerThread; ++j) {
        if (!ready[j]) continue;
        if ((head[j] & 3u) == 0u) T.lvOrder[at++] = head[j] >> 2;
        ++k[j];
        head[j] = nx[j][0]; nx[j][0] = nx[j][1]; nx[j][1] = nx[j][2]; nx[j][2] = nx[j][3];
        if (k[j] < cnt[j]) {
          const uint32_t c = head[j] >> 2;
          atomicAdd(&heads32[c >> 2], 1u << (8u * (c & 3u)));
        }
      }
      lds_barrier();
      const uint32_t made = sCnt[r % 3u];
      if (made == 0u) { stuck = true; break; }  // (a contact that names a node twice never becomes ready: the walk takes over)
      total += made;
      if (tid == 0) { sCnt[(r + 2u) % 3u] = 0; T.lvStart[r + 1u] = total; }
      ++r;
    }
  }
  if (stuck) {
    if (tid == 0) { T.counters[6] = 1u << 20; T.counters[7] = 1; }
    return;
  }
  if (tid == 0) { T.counters[6] = r; T.counters[7] = 0; }
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the level lists have left the wavefronts
  __syncthreads();
  for (uint32_t q = tid; q < M; q += kSeqBlock) {  // the LDS slots of every contact's nodes, in level order
    const uint4 id = T.ids[ldu32(T.lvOrder + q)];
    T.lvSlots[q] = make_uint2(T.nodeSlot[id.x] | (T.nodeSlot[id.y] << 16), T.nodeSlot[id.z] | (T.nodeSlot[id.w] << 16));
  }
}

PIES_DEV void tri_levels(const TriArrays& T, int ldsForm) {  // (a workgroup of kSeqBlock)
  __shared__ __align__(16) uint32_t raw[(kTriMaxLevels + 1) + kLevelsLdsCap / 2 + 8];
  static_assert(sizeof(raw) >= kLvMaxContacts + 64, "heads[] of the fast path must fit");
  __shared__ int sMaxLevel;
  uint32_t* hist = raw;
  uint16_t* slv = reinterpret_cast<uint16_t*>(raw + kTriMaxLevels + 2);
  const uint32_t M = T.counters[2];
  const int tid = threadIdx.x;
  if (ldsForm && M != 0 && M <= kLvMaxContacts && T.counters[4] <= kSeqLdsNodes) {
    const uint32_t used = T.counters[4];  // nodes per thread: as few as the touched nodes need (a round's cost is per slot)
    if (used <= kSeqBlock) levels_by_node_owners<1>(T, raw + 4, raw, M, used, tid);
    else if (used <= 2 * kSeqBlock) levels_by_node_owners<2>(T, raw + 4, raw, M, used, tid);
    else levels_by_node_owners<4>(T, raw + 4, raw, M, used, tid);
    return;
  }
  for (int b = tid; b <= static_cast<int>(kTriMaxLevels); b += kSeqBlock) hist[b] = 0;
  if (tid == 0) sMaxLevel = -1;
  __syncthreads();
  if (M == 0) {
    if (tid == 0) { T.counters[6] = 0; T.counters[7] = 0; }
    return;
  }
  const bool inLds = M <= kLevelsLdsCap;
  if (inLds) {
    int top = -1;
    bool overflow = false;
    for (uint32_t base = 0; base < M; base += kSeqBlock) {
      const uint32_t c = base + tid;
      const bool valid = c < M;
      uint32_t pred[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      if (valid) {
        const uint4 id = T.ids[c];
        const uint32_t node[4] = {id.x, id.y, id.z, id.w};
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          const uint32_t rank = T.incPos[4 * c + l];
          if (rank) pred[l] = T.incSorted[T.incStart[node[l]] + rank - 1] >> 2;
        }
        slv[c] = 0;
      }
      __syncthreads();
      int cur = 0;
      for (;;) {
        int lv = 0;
        if (valid) {
#pragma unroll
          for (int l = 0; l < 4; ++l)
            if (pred[l] != 0xffffffffu) lv = max(lv, static_cast<int>(slv[pred[l]]) + 1);
        }
        const bool changed = valid && lv != cur;
        if (changed) { cur = lv; slv[c] = static_cast<uint16_t>(min(lv, 65535)); }
        if (!__syncthreads_or(changed ? 1 : 0)) break;
      }
      if (valid) { top = max(top, cur); overflow = overflow || cur >= 65535; }
    }
    // block maximum of the levels
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) top = max(top, __shfl_xor(top, off, 64));
    if ((tid & 63) == 0) atomicMax(&sMaxLevel, overflow ? 1 << 20 : top);
    __syncthreads();
  } else {
    if (tid < 64) {
      int top = -1;
      for (uint32_t base = 0; base < M; base += 64) {
        const bool valid = base + tid < M;
        const uint4 id = valid ? T.ids[base + tid] : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
        int bl = 0;
        if (valid) {
          const int a = __hip_atomic_load(T.lastLevel + id.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int b = __hip_atomic_load(T.lastLevel + id.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int c = __hip_atomic_load(T.lastLevel + id.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int d = __hip_atomic_load(T.lastLevel + id.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bl = max(max(a, b), max(c, d)) + 1;
        }
        int maxLevel;
        const int level = window_levels(valid, id, tid, maxLevel, bl);
        if (valid) {
          T.lvl[base + tid] = static_cast<uint32_t>(level);
          atomicMax(T.lastLevel + id.x, level); atomicMax(T.lastLevel + id.y, level);
          atomicMax(T.lastLevel + id.z, level); atomicMax(T.lastLevel + id.w, level);
        }
        top = max(top, maxLevel);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the window's levels are in L2 before the next window reads them
      }
      if (tid == 0) sMaxLevel = top;
    }
    __syncthreads();
    for (uint32_t c = tid; c < M; c += kSeqBlock) {  // behind us: the per-node levels go back to -1 for the next substep
      const uint4 id = T.ids[c];
      T.lastLevel[id.x] = -1; T.lastLevel[id.y] = -1; T.lastLevel[id.z] = -1; T.lastLevel[id.w] = -1;
    }
  }
  const int levels = sMaxLevel + 1;
  if (levels > static_cast<int>(kTriMaxLevels)) {
    if (tid == 0) { T.counters[6] = static_cast<uint32_t>(min(levels, 1 << 20)); T.counters[7] = 1; }
    return;
  }
  auto level_of = [&](uint32_t c) { return inLds ? static_cast<uint32_t>(slv[c]) : T.lvl[c]; };
  for (uint32_t c = tid; c < M; c += kSeqBlock) atomicAdd(&hist[level_of(c) + 1], 1u);
  __syncthreads();
  if (tid == 0) {
    for (int b = 0; b < levels; ++b) hist[b + 1] += hist[b];  // hist[b] = first slot of level b
    T.counters[6] = static_cast<uint32_t>(levels);
    T.counters[7] = 2;  // level by level through L2
  }
  __syncthreads();
  for (int b = tid; b <= levels; b += kSeqBlock) T.lvStart[b] = hist[b];
  __syncthreads();
  for (uint32_t c = tid; c < M; c += kSeqBlock) T.lvOrder[atomicAdd(&hist[level_of(c)], 1u)] = c;  // any order inside a level
}
__global__ void __launch_bounds__(kSeqBlock) k_tri_levels(TriArrays T, int ldsForm) { tri_levels(T, ldsForm); }
// The contact-light graph variant, levels in line: list offsets, list, incidence chain and dependency levels by ONE workgroup (a
// substep without a hit record leaves at the first line; k_tri_box has zeroed what the levels would report).
static_assert(kSeqBlock == 1024, "k_tri_tail_levels runs the tail's steps and the levels with one workgroup size");
__global__ void __launch_bounds__(1024) k_tri_tail_levels(TriArrays T, const float* __restrict__ kdiag, float* __restrict__ cdiag, float* __restrict__ dinv,
                                                          uint32_t words, int ldsForm) {
  __shared__ uint32_t part[1024];
  if (!tri_tail(T, kdiag, cdiag, dinv, words, part)) return;
  step_boundary();
  tri_levels(T, ldsForm);
}

// A sequential pass over the contact list (stabilisation or friction), level by level: the contacts of a level share no
// node.  Three forms, chosen by k_tri_levels (counters[7]):
//  0  the touched nodes (at most kSeqLdsNodes) are copied into LDS, the levels run on the copy with a barrier that waits
//     for LDS traffic only, the copy is written back at the end.  A level costs its arithmetic and one LDS round trip
//     (0.43 us: the dependent chain cross product - square root - division - dot product - divisions of one contact)
//     instead of one or two L2 round trips on top of it (0.75 us stabilisation, 2.0 us friction on a 29k-contact patch);
//     four wavefronts work (levels are 20-110 contacts wide), the other twelve help with the copy and leave.  The
//     contacts' node slots are staged through LDS kSeqChunk at a time.  (Measured and dropped: four lanes per contact,
//     one vector component each over DPP quad permutes - bit-identical, less than half the instructions, the same
//     270 us per pass: a level waits for latencies, not for issue slots; one working wavefront without barriers: slower,
//     levels wider than 64 contacts take two rounds.)
//  2  node state through agent-scope (L2) loads and stores, a level ends with the stores drained and a workgroup barrier.
//  1  more levels than kTriMaxLevels: one wavefront walks the list window by window.
#ifndef PIES_SEQ_WORKERS
#define PIES_SEQ_WORKERS 256
#endif
constexpr int kSeqWorkers = PIES_SEQ_WORKERS;
constexpr uint32_t kSeqChunk = 2048;
// between two levels: with one working wavefront its LDS operations are already in order; with several, a barrier
PIES_DEV void level_barrier() {
  if (kSeqWorkers > 64) lds_barrier();
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// floor friction of one node (Solver.cpp:473-484, once per floor contact of the node; the arithmetic of k_pd_velocity's floor-friction loop)
PIES_DEV F3 floor_friction(F3 v, uint32_t ns, float friction, float staticThreshold) {
  for (uint32_t c = 0; c < ns; ++c) {
    const float px = v.x, pz = v.z;
    float fr = friction;
    if (sqrtf(px * px + 0.0f * 0.0f + pz * pz) < staticThreshold) fr = 1.0f;
    v.x += -fr * px;
    v.y += -fr * 0.0f;
    v.z += -fr * pz;
  }
  return v;
}
// MODE 0 runs ALL the stabilisation iterations of the substep (Solver.cpp:367-383: every iteration is a pass over the contacts
// followed by the floor snap of every node with a floor contact): the snap of a node puts it where the right-hand side kernel
// left its target (statp) and is idempotent, and a pass only touches the nodes of the list (usedNodes) - so the snap of those
// nodes runs here, between the passes, and the snap of all the others once, in k_pd_stabilize behind this kernel.  (Until
// round 3 the host launched pass and snap `iterations` times: eight launches of ~4.7 us in a substep without a single contact.)
// MODE 1 (friction) ends with the floor friction of the list's nodes, which the reference applies after the contacts' friction;
// k_pd_velocity, before this kernel, has applied it to every node that is in no contact (usedBits).
template <int MODE>
__global__ void __launch_bounds__(kSeqBlock) k_tri_sequential(TriArrays T, float4* pos4, float4* prev4, float4* vel4, float thickness,
                                                              float friction, float staticThreshold, const uint32_t* __restrict__ nstatic,
                                                              const float4* __restrict__ statp, uint32_t iterations) {
  __shared__ float4 P[kSeqLdsNodes], Q[kSeqLdsNodes];
  __shared__ uint2 sSlots[kSeqChunk];
  __shared__ uint32_t sLv[kTriMaxLevels + 1];
  float* pos = reinterpret_cast<float*>(pos4);
  float* prev = reinterpret_cast<float*>(prev4);
  float* vel = reinterpret_cast<float*>(vel4);
  const int tid = threadIdx.x;
  const uint32_t M = T.counters[2];
  if (M == 0) return;
  const uint32_t form = T.counters[7], used = T.counters[4];
  const bool snap = MODE == 0 && nstatic != nullptr, floorFr = MODE == 1 && nstatic != nullptr;
  if (form == 0) {
    const uint32_t levels = T.counters[6];
    float4* second4 = MODE == 0 ? prev4 : vel4;
    for (uint32_t u = tid; u < used; u += kSeqBlock) {
      const uint32_t n = T.usedNodes[u];
      P[u] = pos4[n];
      Q[u] = second4[n];
    }
    for (uint32_t l = tid; l <= levels; l += kSeqBlock) sLv[l] = T.lvStart[l];
    __syncthreads();
    if (tid >= kSeqWorkers) return;  // (a wavefront that has ended no longer counts at the barrier)
    uint32_t k0 = 0, kEnd = 0;
    for (uint32_t it = 0; it < iterations; ++it) {
      if (M > kSeqChunk) k0 = kEnd = 0;  // (a list that fits one chunk stays staged)
      for (uint32_t lv = 0; lv < levels; ++lv) {
        const uint32_t lo = sLv[lv], hi = sLv[lv + 1];
        uint32_t seg = lo;
        while (seg < hi) {
          if (min(hi, seg + kSeqChunk) > kEnd) {  // stage the node slots of the next kSeqChunk contacts
            k0 = seg;
            kEnd = min(M, seg + kSeqChunk);
            for (uint32_t q = k0 + tid; q < kEnd; q += kSeqWorkers) sSlots[q - k0] = T.lvSlots[q];
            level_barrier();
          }
          const uint32_t segEnd = min(hi, kEnd);
          for (uint32_t q = seg + tid; q < segEnd; q += kSeqWorkers) {
            const uint2 sl = sSlots[q - k0];
            LdsNodes io = {{sl.x & 0xffffu, sl.x >> 16, sl.y & 0xffffu, sl.y >> 16}, P, Q};
            tri_contact_step<MODE>(io, thickness, friction, staticThreshold);
          }
          level_barrier();
          seg = segEnd;
        }
      }
      if (snap) {
        for (uint32_t u = tid; u < used; u += kSeqWorkers) {
          const uint32_t n = T.usedNodes[u];
          if (nstatic[n]) {
            const float4 sp = statp[n];
            float4& d = P[u];
            d.x = sp.x; d.y = sp.y; d.z = sp.z;
          }
        }
        level_barrier();
      }
    }
    for (uint32_t u = tid; u < used; u += kSeqWorkers) {
      const uint32_t n = T.usedNodes[u];
      if (MODE == 0) pos4[n] = P[u];
      float4 q = Q[u];
      if (floorFr) {
        const F3 v = floor_friction(F3{q.x, q.y, q.z}, nstatic[n], friction, staticThreshold);
        q.x = v.x; q.y = v.y; q.z = v.z;
      }
      second4[n] = q;
    }
    return;
  }
  if (form == 1) {  // too many levels: one wavefront walks the list window by window
    if (tid >= 64) return;
    for (uint32_t it = 0; it < iterations; ++it) {
      for (uint32_t base = 0; base < M; base += 64) {
        const bool valid = base + tid < M;
        const uint4 id = valid ? T.ids[base + tid] : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
        int maxLevel;
        const int level = window_levels(valid, id, tid, maxLevel);
        for (int lv = 0; lv <= maxLevel; ++lv) {
          if (valid && level == lv) tri_contact_step<MODE>(id, pos, prev, vel, thickness, friction, staticThreshold);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      if (snap) {
        for (uint32_t u = tid; u < used; u += 64) {
          const uint32_t n = T.usedNodes[u];
          if (nstatic[n]) { const float4 sp = statp[n]; st3(pos, n, F3{sp.x, sp.y, sp.z}); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    if (floorFr)
      for (uint32_t u = tid; u < used; u += 64) {
        const uint32_t n = T.usedNodes[u];
        const uint32_t ns = nstatic[n];
        if (ns) st3(vel, n, floor_friction(ld3(vel, n), ns, friction, staticThreshold));
      }
    return;
  }
  const uint32_t levels = T.counters[6];
  for (uint32_t it = 0; it < iterations; ++it) {
    for (uint32_t lv = 0; lv < levels; ++lv) {
      const uint32_t lo = T.lvStart[lv], hi = T.lvStart[lv + 1];
      for (uint32_t k = lo + tid; k < hi; k += kSeqBlock)
        tri_contact_step<MODE>(T.ids[T.lvOrder[k]], pos, prev, vel, thickness, friction, staticThreshold);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (snap) {
      for (uint32_t u = tid; u < used; u += kSeqBlock) {
        const uint32_t n = T.usedNodes[u];
        if (nstatic[n]) { const float4 sp = statp[n]; st3(pos, n, F3{sp.x, sp.y, sp.z}); }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  if (floorFr)
    for (uint32_t u = tid; u < used; u += kSeqBlock) {
      const uint32_t n = T.usedNodes[u];
      const uint32_t ns = nstatic[n];
      if (ns) st3(vel, n, floor_friction(ld3(vel, n), ns, friction, staticThreshold));
    }
}

// ------------------------------------------------------------------------------------------------------------
static int tri_lds_form() {
  const char* e = tuning_env("PIES_TRI_LDS");  // diagnostics, read when the substep is captured: 0 = sequential passes through L2
  return e && e[0] == '0' ? 0 : 1;
}
uint32_t launch_tri_detect(hipStream_t st_, const TriArrays& T, const NodeArrays& nd, const float* kdiag, float* cdiag, float* dinv,
                           float threshold, float /*thickness*/, bool mergedRows, bool levelsInLine) {
  if (T.nt == 0) return 0;
  const dim3 wide(std::min<uint32_t>(1024u, (T.nt * 8 + kBlock - 1) / kBlock)), blk(kBlock);
  hipLaunchKernelGGL(k_tri_box, grid_for(T.nt), blk, 0, st_, T, nd.pos, nd.prev);
  hipLaunchKernelGGL(k_tri_starts, dim3(T.slots / kGridTile), blk, 0, st_, T);
  hipLaunchKernelGGL(k_tri_fill, grid_for(T.nt), blk, 0, st_, T);
  int team = kTriTeam;  // PIES_TRI_TEAM: lanes per triangle in k_tri_pairs (8, 16, 32; speed only)
  if (const char* e = tuning_env("PIES_TRI_TEAM")) team = std::atoi(e);
  if (team == 8) hipLaunchKernelGGL((k_tri_pairs<8>), grid_for(T.nt * 8), blk, 0, st_, T, nd.pos, nd.prev, threshold);
  else if (team == 32) hipLaunchKernelGGL((k_tri_pairs<32>), grid_for(T.nt * 32), blk, 0, st_, T, nd.pos, nd.prev, threshold);
  else hipLaunchKernelGGL((k_tri_pairs<kTriTeam>), grid_for(T.nt * kTriTeam), blk, 0, st_, T, nd.pos, nd.prev, threshold);
  hipLaunchKernelGGL(k_tri_ccd, dim3(1024), blk, 0, st_, T, nd.pos, nd.prev, threshold);  // (a multiple of kWorkShards)
  if (!mergedRows) {  // the variant for substeps with few or no contacts: list offsets, list and incidence chain as one launch of one workgroup
    if (levelsInLine) {  // ... and the dependency levels with them (launch_tri_levels is not called then)
      hipLaunchKernelGGL(k_tri_tail_levels, dim3(1), dim3(1024), 0, st_, T, kdiag, cdiag, dinv, (nd.n + 31u) / 32u, tri_lds_form());
      return 6;
    }
    hipLaunchKernelGGL(k_tri_tail, dim3(1), dim3(1024), 0, st_, T, kdiag, cdiag, dinv, (nd.n + 31u) / 32u);
    return 6;
  }
  hipLaunchKernelGGL(k_tri_scan, dim3(1), dim3(1024), 0, st_, T);
  hipLaunchKernelGGL(k_tri_emit, dim3(std::min<uint32_t>(256u, (T.maxContacts + kBlock - 1) / kBlock)), blk, 0, st_, T);
  const dim3 cgrid(std::min<uint32_t>(256u, (T.maxContacts + kBlock - 1) / kBlock));
  hipLaunchKernelGGL(k_inc_count, cgrid, blk, 0, st_, T, cdiag);
  hipLaunchKernelGGL(k_inc_used, dim3(1), dim3(1024), 0, st_, T, (nd.n + 31u) / 32u);
  hipLaunchKernelGGL(k_inc_alloc, cgrid, blk, 0, st_, T, kdiag, cdiag, dinv);
  hipLaunchKernelGGL(k_inc_fill, cgrid, blk, 0, st_, T);
  hipLaunchKernelGGL(k_inc_sort, cgrid, blk, 0, st_, T);
  if (mergedRows) {
    uint32_t maxUnique = kRowMaxUnique;  // diagnostics: PIES_ROW_MAX_UNIQUE lowers it (rows with more distinct columns stay unmerged)
    if (const char* e = tuning_env("PIES_ROW_MAX_UNIQUE")) maxUnique = std::min<uint32_t>(kRowMaxUnique, static_cast<uint32_t>(std::max(0, std::atoi(e))));
    hipLaunchKernelGGL(k_contact_csr, cgrid, blk, 0, st_, T, maxUnique);
  }
  return 13;
}
void launch_tri_levels(hipStream_t st_, const TriArrays& T) {
  if (T.nt == 0) return;
  hipLaunchKernelGGL(k_tri_levels, dim3(1), dim3(kSeqBlock), 0, st_, T, tri_lds_form());
}
void launch_pd_local_tri(hipStream_t st_, const TriArrays& T, const float4* pos, float thickness) {
  if (T.nt == 0) return;
  const dim3 cgrid(std::min<uint32_t>(256u, (T.maxContacts + kBlock - 1) / kBlock));
  hipLaunchKernelGGL(k_pd_local_tri, cgrid, dim3(kBlock), 0, st_, T, pos, thickness);
}
void launch_tri_stabilize(hipStream_t st_, const TriArrays& T, const NodeArrays& nd, float thickness, const uint32_t* nstatic, const float4* statp,
                          uint32_t iterations) {
  if (T.nt == 0 || iterations == 0) return;
  hipLaunchKernelGGL(k_tri_sequential<0>, dim3(1), dim3(kSeqBlock), 0, st_, T, nd.pos, nd.prev, nd.vel, thickness, 0.0f, 0.0f, nstatic, statp, iterations);
}
void launch_tri_friction(hipStream_t st_, const TriArrays& T, const NodeArrays& nd, float friction, float staticThreshold, const uint32_t* nstatic) {
  if (T.nt == 0) return;
  hipLaunchKernelGGL(k_tri_sequential<1>, dim3(1), dim3(kSeqBlock), 0, st_, T, nd.pos, nd.prev, nd.vel, 0.0f, friction, staticThreshold, nstatic,
                     static_cast<const float4*>(nullptr), 1u);
}

}  // namespace pies
