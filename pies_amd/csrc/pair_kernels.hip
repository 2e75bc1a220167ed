// Node-node resolve of the PBD substep (Src/Solver.cpp:85-130) in the PAIR ORDER.
//
// The reference's loop lets node i meet node j once per grid cell both were inserted into, in each direction, itself
// included (quirk Q3), and resolves every overlapping meeting at once.  The result depends on the order of the meetings
// that share a node and on nothing else.  The pair order keeps every meeting and re-orders them pair by pair:
//   1. every node's meetings with itself (one per cell of its range);
//   2. the unordered pairs {i < j} whose inserted ranges share m > 0 cells, in ascending order of pair_key (a class from the
//      pair's direction and place when the grid was built - pairs of one class seldom share a node -, then a 64-bit mix of
//      the two indices: a total order), each as m visits of i to j followed by m visits of j to i.  Every visit tests the live
//      positions, like the reference's.
// The oracle replays exactly this with a sort and a sequential loop (FLAG_COLLISION_RULE = 2).  On the device the order is
// executed by dependency levels: a pair's turn comes when it is the next unprocessed pair in the key-sorted lists of BOTH
// its nodes, and the pairs whose turn has come share no node, so a level is one data-parallel launch.  The classed key keeps the chains short: 30-45 levels for the 3-4 M pairs of
// BASELINE config 4 (60-80 with the hash alone), against 27 passes x 350 dependent visits per group in the group order.
//
// Filter.  Of the ~300 nodes that share a cell with a node only ~20 are near enough to ever touch it.  A pair is listed
// only if its distance at grid-build time is below r_i + r_j + s_i + s_j, and every node is checked to stay within its
// slack s_i of its build-time position whenever it has been moved: while that holds, every unlisted visit is a miss
// (|p_i - p_j| >= d0 - s_i - s_j >= r_i + r_j) and the result is that of the full order.  The slack is per node and follows
// what the node did in the passes before.  A node that leaves its slack is put on a list; after the pass one wavefront per
// listed node looks at the unlisted nodes it shares a cell with and tests, with the largest excursions both nodes had in the
// pass, whether the two can have touched (d0 - e_i - e_j < r_i + r_j).  Only if one such pair exists is the pass repeated from
// the saved state, the nodes that left their slack now listing every node they share a cell with; a repeat that fails the
// same test is counted (pies_get_collision_health: passes_inexact).
#include <map>
#include <mutex>
#include <utility>
#include <climits>
#include <cstdlib>
#include <cstdint>

#include "dev_math.h"
#include "hash_device.h"
#include "pair_kernels.h"

namespace pies {

constexpr int kBlock = 256;
constexpr uint32_t kMaxCand = 512;                         // distinct nodes of a group's 2x2x2 cells (BASELINE config 4: 216-343)
constexpr uint32_t kMaxOwn = 256;                          // nodes of one group
constexpr uint32_t kMaxDeg = 1024;                         // listed partners of one node
constexpr uint32_t kPairNodeMask = 0x01ffffffu;            // partner index (n < 2^25); bits 28-31 hold (shared cells - 1)
// Bits 25-27 of a record's current entry carry the low three bits of the cursor it belongs to.  A record is four words, written
// and read with one 16-byte access each, and a lane looks at the records of nodes that other lanes may be moving on in the same
// launch.  The protocol is safe with the old or the new record; an experiment that made lanes read records later in a launch
// (a lane going on to its node's next pair: half the levels, but each three times as long - dropped) showed readers that got the
// cursor / stamp word of one version with the entry word of the next.  The tag makes such a view recognisable; a reader that
// gets one leaves the pair alone, like one that finds the node moved on in this round (whoever moved it sees to it).
constexpr uint32_t kPairTagShift = 25;
PIES_DEV bool rec_consistent(const uint4& r) { return ((r.w >> kPairTagShift) & 7u) == (r.z & 7u); }
constexpr int kBuildWaves = 2;                            // wavefronts of a workgroup of the list kernel: one group at a time

// oracle/ora_math.h: pair_key - direction class and parity of the pair from the positions the grid was built from, then
// murmur3's 64-bit finaliser over (i << 32 | j); i < j, (pix, ..) the lower node's position
PIES_DEV uint64_t pair_key(uint32_t i, uint32_t j, float pix, float piy, float piz, float pjx, float pjy, float pjz) {
  uint64_t k = (static_cast<uint64_t>(i) << 32) | j;
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  const float dx = pjx - pix, dy = pjy - piy, dz = pjz - piz;
  const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
  const float lim = 0.41421356f * fmaxf(ax, fmaxf(ay, az));
  int qx = ax > lim ? (dx < 0.0f ? -1 : 1) : 0, qy = ay > lim ? (dy < 0.0f ? -1 : 1) : 0, qz = az > lim ? (dz < 0.0f ? -1 : 1) : 0;
  const int lead = qx != 0 ? qx : (qy != 0 ? qy : qz);
  if (lead < 0) { qx = -qx; qy = -qy; qz = -qz; }
  const float fx = static_cast<float>(qx), fy = static_cast<float>(qy), fz = static_cast<float>(qz);
  const float qq = fmaxf(fx * fx + fy * fy + fz * fz, 1.0f);
  const float ui = (pix * fx + piy * fy + piz * fz) / qq, uj = (pjx * fx + pjy * fy + pjz * fz) / qq;
  const float len = fmaxf(fabsf(uj - ui), 0.001f);
  const float t = fminf(fmaxf(floorf(fminf(ui, uj) / len), -1.0e9f), 1.0e9f);
  const uint32_t parity = static_cast<uint32_t>(static_cast<long long>(t)) & 1u;
  const uint32_t cls = static_cast<uint32_t>((qx + 1) * 9 + (qy + 1) * 3 + (qz + 1)) * 2u + parity;  // < 54
  return (static_cast<uint64_t>(cls) << 58) | (k >> 6);
}

// the key of the pair {i, j} in either order of the arguments: ONE evaluation of pair_key on swapped operands (a wavefront's lanes
// hold pairs of both orders: `i < j ? pair_key(i, j, ..) : pair_key(j, i, ..)` made it run both - ~150 instructions each, a quarter
// of the list kernel's)
PIES_DEV uint64_t pair_key_of(uint32_t i, uint32_t j, float pix, float piy, float piz, float pjx, float pjy, float pjz) {
  const bool low = i < j;
  return pair_key(low ? i : j, low ? j : i, low ? pix : pjx, low ? piy : pjy, low ? piz : pjz, low ? pjx : pix, low ? pjy : piy, low ? pjz : piz);
}

// The reference order by turns looks a node's range up from its LIVE position when its turn starts, so a node can be met in a cell its
// partner was not inserted into when the lists were built.  The lists hold the pairs within reach that share a cell of the two INSERTED
// ranges; the ranges cover [p - r - 0.5, p + r + 0.5] per axis (Solver.cpp:877-901), so two nodes whose inserted ranges share no cell
// are r_i + r_j + 1 apart on some axis and can only touch in a pass in which their excursions add up to 1: while every node's
// excursion stays below kTurnMaxExcursion the lists (and k_pair_verify's test of the unlisted nodes that DO share a cell) cover every
// visit that can hit.  The nodes that stray further are looked at one by one after the pass (k_pair_verify); only a pass in which
// one of them can have touched a node it shares no inserted cell with is left to the sequential loop.
constexpr float kTurnMaxExcursion = 0.45f;
PIES_DEV float lane_value(float v, uint32_t srcLane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), static_cast<int>(srcLane))); }
// slack of a node after a pass in which it strayed `exc` from its build-time position
PIES_DEV float next_slack(float exc, float previous, float r) {
  const float want = fmaxf(2.0f * exc + 0.1f * r, 0.4f * r);
  return previous < 1.0e30f ? fmaxf(want, 0.95f * previous) : want;
}

// ---- one visit (Solver.cpp:88-126), all of it in one lane -------------------------------------------------------------
struct NodeState {
  float px, py, pz, w, vx, vy, vz, r;
};
// node a visits node b (a != b).  Returns whether the pair was resolved.
PIES_DEV bool visit(NodeState& a, NodeState& b, float friction, float staticThreshold) {
  const float dx = b.px - a.px, dy = b.py - a.py, dz = b.pz - a.pz;
  const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
  const float disp = a.r + b.r - dist;
  if (!(disp > 0.0f)) return false;
  float ux = 1.0f, uy = 0.0f, uz = 0.0f;
  if (dist > 0.00001f) { ux = dx / dist; uy = dy / dist; uz = dz / dist; }
  const float wSum = a.w + b.w;
  const float sa = 0.85f * -disp, sb = 0.85f * disp;
  const float rx = b.vx - a.vx, ry = b.vy - a.vy, rz = b.vz - a.vz;
  const float rd = rx * ux + ry * uy + rz * uz;
  const float qx = rx - rd * ux, qy = ry - rd * uy, qz = rz - rd * uz;
  float fr = friction;
  if (staticThreshold > 0.0f)  // sqrt(x) < t is false for every t <= 0
    if (sqrtf(qx * qx + qy * qy + qz * qz) < staticThreshold) fr = 1.0f;
  a.px += ((sa * ux) * a.w) / wSum; a.py += ((sa * uy) * a.w) / wSum; a.pz += ((sa * uz) * a.w) / wSum;
  b.px += ((sb * ux) * b.w) / wSum; b.py += ((sb * uy) * b.w) / wSum; b.pz += ((sb * uz) * b.w) / wSum;
  a.vx += ((-fr * qx) * a.w) / wSum; a.vy += ((-fr * qy) * a.w) / wSum; a.vz += ((-fr * qz) * a.w) / wSum;
  b.vx += ((fr * qx) * b.w) / wSum; b.vy += ((fr * qy) * b.w) / wSum; b.vz += ((fr * qz) * b.w) / wSum;
  return true;
}
// a node meets itself (quirk Q3): `other` aliases `node`, so the second update of each line sees the first
PIES_DEV bool visit_self(NodeState& a, float friction, float staticThreshold) {
  const float dx = a.px - a.px, dy = a.py - a.py, dz = a.pz - a.pz;
  const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
  const float disp = a.r + a.r - dist;
  if (!(disp > 0.0f)) return false;
  float ux = 1.0f, uy = 0.0f, uz = 0.0f;
  if (dist > 0.00001f) { ux = dx / dist; uy = dy / dist; uz = dz / dist; }
  const float wSum = a.w + a.w;
  const float sa = 0.85f * -disp, sb = 0.85f * disp;
  const float rx = a.vx - a.vx, ry = a.vy - a.vy, rz = a.vz - a.vz;
  const float rd = rx * ux + ry * uy + rz * uz;
  const float qx = rx - rd * ux, qy = ry - rd * uy, qz = rz - rd * uz;
  float fr = friction;
  if (staticThreshold > 0.0f)
    if (sqrtf(qx * qx + qy * qy + qz * qz) < staticThreshold) fr = 1.0f;
  a.px += ((sa * ux) * a.w) / wSum; a.py += ((sa * uy) * a.w) / wSum; a.pz += ((sa * uz) * a.w) / wSum;
  a.px += ((sb * ux) * a.w) / wSum; a.py += ((sb * uy) * a.w) / wSum; a.pz += ((sb * uz) * a.w) / wSum;
  a.vx += ((-fr * qx) * a.w) / wSum; a.vy += ((-fr * qy) * a.w) / wSum; a.vz += ((-fr * qz) * a.w) / wSum;
  a.vx += ((fr * qx) * a.w) / wSum; a.vy += ((fr * qy) * a.w) / wSum; a.vz += ((fr * qz) * a.w) / wSum;
  return true;
}
// ---- the pass's own node records: 64 bytes = one cache line per node ------------------------------------------------------
//   [0] x, y, z, invMass          [1] vx, vy, vz, radius          [2] position when the grid was built (x, y, z), slack
//   [3] first list entry, entries, current entry | round in which the node got there << 16, the current entry itself
// A level touches a node through this line only (the level kernels are bound by the number of scattered memory transactions).
PIES_DEV NodeState load_node(const float4* __restrict__ node, uint32_t i) {
  const float4 p = node[4u * i], v = node[4u * i + 1u];
  return NodeState{p.x, p.y, p.z, p.w, v.x, v.y, v.z, v.w};
}
PIES_DEV void store_node(float4* node, uint32_t i, const NodeState& a) {
  node[4u * i] = make_float4(a.px, a.py, a.pz, a.w);
  node[4u * i + 1u] = make_float4(a.vx, a.vy, a.vz, a.r);
}
PIES_DEV uint4 load_rec(const float4* __restrict__ node, uint32_t i) { return reinterpret_cast<const uint4*>(node)[4u * i + 3u]; }
PIES_DEV void store_rec(float4* node, uint32_t i, uint4 r) { reinterpret_cast<uint4*>(node)[4u * i + 3u] = r; }
// a node has been moved: its excursion from the position the lists were built from (p0.xyz) is kept as a maximum; the first time
// it leaves its slack (p0.w) it is put on the list k_pair_verify works through
PIES_DEV void note_excursion(const PairArrays& P, uint32_t i, const NodeState& a, const float4 p0) {
  const float dx = a.px - p0.x, dy = a.py - p0.y, dz = a.pz - p0.z;
  const float e = sqrtf(dx * dx + dy * dy + dz * dz);
  const float thr = 0.999f * p0.w;
  const float old = __uint_as_float(atomicMax(&P.exc[i], __float_as_uint(e)));  // (non-negative floats order like their bits; NaN sorts above everything)
  if (!(e <= thr) && old <= thr) {
    const uint32_t at = atomicAdd(&P.ctl[kPairLeft], 1u);
    if (at < P.n) P.left[at] = i;
  }
}
// the same for a caller that owns the node for the whole launch step (no other lane can touch it) and has read its excursion already
PIES_DEV void note_excursion_owned(const PairArrays& P, uint32_t i, const NodeState& a, const float4 p0, uint32_t oldBits) {
  const float dx = a.px - p0.x, dy = a.py - p0.y, dz = a.pz - p0.z;
  const float e = sqrtf(dx * dx + dy * dy + dz * dz);
  const float thr = 0.999f * p0.w;
  if (__float_as_uint(e) > oldBits) P.exc[i] = __float_as_uint(e);
  if (!(e <= thr) && __uint_as_float(oldBits) <= thr) {
    const uint32_t at = atomicAdd(&P.ctl[kPairLeft], 1u);
    if (at < P.n) P.left[at] = i;
  }
}
// resolved pairs are counted per wavefront into one of kPairStripes words (k_pair_check adds them up)
PIES_DEV void count_hits(const PairArrays& P, uint32_t hits, int lane) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) hits += __shfl_xor(hits, o, 64);
  if (lane == 0 && hits) atomicAdd(&P.hitStripe[((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % kPairStripes) * kPairPad], hits);
}
// every node's meetings with itself: it is in every bucket of its own range, once per cell (quirk Q3)
PIES_DEV uint32_t self_visits(const HashArrays& H, const PairArrays& P, uint32_t i, NodeState& a, const float4 p0, float friction, float staticThreshold) {
  const int4 rg = H.rng[i];
  const uint32_t m = (rg.w & 0xff) * ((rg.w >> 8) & 0xff) * ((rg.w >> 16) & 0xff);
  uint32_t hits = 0;
  for (uint32_t q = 0; q < m; ++q) hits += visit_self(a, friction, staticThreshold) ? 1u : 0u;
  if (hits) note_excursion(P, i, a, p0);
  return hits;
}

// ---- save: the state the pass starts from, control words, the meetings of every node with itself -------------------------
__global__ void __launch_bounds__(kBlock) k_pair_save(HashArrays H, PairArrays P, const float4* __restrict__ pos, const float4* __restrict__ vel,
                                                      const float* __restrict__ radius, float friction, float staticThreshold) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (i == 0) {
    P.ctl[kPairFlags] = 0;
    P.ctl[kPairRetry] = 0;
    P.ctl[kPairRounds] = 0;
    P.ctl[kPairEdges] = 0;
    P.ctl[kPairGroups] = 0;
    P.ctl[kPairSpilled] = 0;
    P.ctl[kPairFallback] = 0;
    P.ctl[kPairBarrier] = 0;  // (the grid barrier of the levels behind the captured launches: counter, abort word)
    P.ctl[kPairBarrier + 1] = 0;
  }
  if (i < kPairPools) P.pool[i * kPairPad] = 0;
  if (i < 3u * kPairLists) P.frCount[i * kPairPad] = 0;
  if (i < 64u) { P.stat[i * kPairPad] = 0; P.stat[i * kPairPad + 1u] = 0; P.stat[i * kPairPad + 2u] = 0; }
  uint32_t hits = 0;
  if (i < P.n && !H.counters[kCounterFlags]) {
    const float4 p = pos[i], v = vel[i];
    const float r = radius[i];
    float sl = P.node[4u * i + 2u].w;
    if (!(sl > 0.0f)) sl = 0.5f * r;  // first pass after pies_finalize
    const float4 p0 = make_float4(p.x, p.y, p.z, sl);
    P.node[4u * i + 2u] = p0;
    P.bq[i] = make_float4(p.x, p.y, p.z, r + sl);
    P.vel0[i] = v;
    P.exc[i] = 0u;
    NodeState a{p.x, p.y, p.z, p.w, v.x, v.y, v.z, r};
    if (!P.byIndex) hits = self_visits(H, P, i, a, p0, friction, staticThreshold);
    store_node(P.node, i, a);
    // (by turns: a node meets itself inside its own turn, at its place in the bucket; a node no list is written for - an empty range -
    // has its own turn as its only event)
    store_rec(P.node, i, P.byIndex ? make_uint4(0u, 0u, 0u, i) : make_uint4(0u, 0u, 0u, 0u));
    if (P.byIndex) P.turnCnt[i] = 1u;
  }
  count_hits(P, hits, lane);
}
// the same for the repeat of a pass: the saved state is back in place (k_pair_check), the meetings with itself again
__global__ void __launch_bounds__(kBlock) k_pair_self(HashArrays H, PairArrays P, float friction, float staticThreshold) {
  if (!P.ctl[kPairRetry] || H.counters[kCounterFlags]) return;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t hits = 0;
  if (i < P.n && P.byIndex) {  // by turns: the records start over (k_pair_save)
    store_rec(P.node, i, make_uint4(0u, 0u, 0u, i));
    P.turnCnt[i] = 1u;
  } else if (i < P.n) {
    NodeState a = load_node(P.node, i);
    hits = self_visits(H, P, i, a, P.node[4u * i + 2u], friction, staticThreshold);
    if (hits) store_node(P.node, i, a);
  }
  count_hits(P, hits, lane);
}

// ---- groups: one descriptor per group (the nodes whose minimum cell is the same cell) -------------------------------------
// The list kernel works group by group; what it needs of a group - the buckets of the 2x2x2 cells above the group's cell - is
// looked up here, one lane per cell in use, where the dependent look-ups of different cells overlap.
//   descriptor: [0..7] first entry of the eight buckets   [8..11] their lengths, 16 bits each   [12] the cell's slot
__global__ void __launch_bounds__(kBlock) k_pair_groups(HashArrays H, PairArrays P, uint32_t repeat) {
  if (repeat && !P.ctl[kPairRetry]) return;
  if (H.counters[kCounterFlags]) return;
  __shared__ uint32_t lcount, lbase;
  const uint32_t used = H.counters[kCounterUsed];
  const GridBox B = grid_box(H.counters);
  const uint32_t* __restrict__ val = H.val[grid_passes(B) & 1u];
  for (uint32_t first = blockIdx.x * kBlock; first < used; first += gridDim.x * kBlock) {  // (workgroup uniform)
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
    const uint32_t u = first + threadIdx.x;
    uint32_t s = 0, rank = 0;
    bool have = false;
    if (u < used) {
      s = H.used[u];
      // a group exists where some node has its minimum cell (kMinFlag).  (k_grid_groups leaves that count in gcnt for the
      // group order; builds for the pair order skip that launch and look here.)
      const uint32_t bs = H.start[s], be = H.end[s];
      if (be - bs > 0xffffu) atomicOr(&P.ctl[kPairFlags], 2u);  // (a descriptor holds 16-bit bucket lengths; such a pile is the sequential loop's)
      for (uint32_t e = bs; e < be && !have; ++e) have = (val[e] >> 31) != 0u;
      if (have) rank = atomicAdd(&lcount, 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0 && lcount) lbase = atomicAdd(&P.ctl[kPairGroups], lcount);
    __syncthreads();
    if (have) {
      int gx, gy, gz;
      box_cell(B, H.keys[s], gx, gy, gz);
      uint32_t st[8], cn[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t cs = find_bucket(H, B, gx + ((c >> 2) & 1), gy + ((c >> 1) & 1), gz + (c & 1));
        st[c] = cs != 0xffffffffu ? H.start[cs] : 0u;
        cn[c] = cs != 0xffffffffu ? min(H.end[cs] - st[c], 0xffffu) : 0u;  // (a bucket holds at most 2048 nodes)
      }
      uint4* d = P.grp + 4ull * (lbase + rank);
      d[0] = make_uint4(st[0], st[1], st[2], st[3]);
      d[1] = make_uint4(st[4], st[5], st[6], st[7]);
      d[2] = make_uint4(cn[0] | (cn[1] << 16), cn[2] | (cn[3] << 16), cn[4] | (cn[5] << 16), cn[6] | (cn[7] << 16));
      d[3] = make_uint4(s, 0u, 0u, 0u);
    }
    __syncthreads();
  }
}

// ---- lists: one workgroup per group -------------------------------------------------------------------------------------
// Everything a node of the group can share a cell with sits in the buckets of the 2x2x2 cells above the group's cell.  A node
// is in several of them; it is taken from the one that is the minimum corner of what its range shares with the block, which
// the side bits of its entry decide without a look at the node: (cell's offset in the block) & (cell's side in the node's
// range) == 0 on every axis.  The workgroup's wavefronts share the table of candidates and split the group's own nodes.
// The kernel exists in two sizes.  The small one (a group of BASELINE config 4 has ~300 candidates, a node ~20 partners) keeps
// twelve groups in flight per CU - the kernel is bound by the chain of dependent look-ups of a group, not by arithmetic - and
// passes groups that do not fit on to a list the large one works through.
template <uint32_t MAXC, uint32_t MAXD, uint32_t MAXOWN>
struct BuildLds {
  uint32_t id[MAXC];                                 // the distinct candidates: node index
  float px[MAXC], py[MAXC], pz[MAXC], rs[MAXC];      // position at grid-build time, radius + slack
  uint32_t rg[MAXC];                                 // (min cell - group cell + 1) per axis, 2 bits each; (length - 1) per axis from bit 8
  uint16_t own[MAXOWN];                              // candidates that are the group's own nodes
  uint32_t ncand, nown, spill;
  uint32_t looked[8];                                // candidates in the cells of a range of (1 or 2) x (1 or 2) x (1 or 2) cells from the group's cell
  uint64_t lk[kBuildWaves][MAXD];                    // one node's partners: pair key
  uint32_t le[kBuildWaves][MAXD];                    //                     partner | (shared cells - 1) << 28
  uint16_t near[kBuildWaves][2][MAXD];               // table slots of the candidates within reach of the wavefront's two nodes (first sweep)
};

// cells two ranges share on one axis: [a0, a0 + la) and [b0, b0 + lb)
PIES_DEV uint32_t shared_cells(int a0, uint32_t la, int b0, uint32_t lb) {
  const int lo = max(a0, b0), hi = min(a0 + static_cast<int>(la), b0 + static_cast<int>(lb));
  return hi > lo ? static_cast<uint32_t>(hi - lo) : 0u;
}

// appends the accepted candidates of the wavefront's lanes to the node's partner list in LDS; returns the new length
template <uint32_t MAXD>
PIES_DEV uint32_t push_partners(uint64_t* lk, uint32_t* le, uint32_t d, bool accept, uint32_t i, uint32_t j, uint32_t m, float pix, float piy, float piz,
                                float pjx, float pjy, float pjz, int lane, bool byIndex) {
  const unsigned long long mask = __ballot(accept);
  if (accept) {
    const uint32_t at = d + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
    if (at < MAXD) {
      // (the reference order by turns walks a node's partners in ascending index: the partner's index is the key)
      lk[at] = byIndex ? static_cast<uint64_t>(j) : pair_key_of(i, j, pix, piy, piz, pjx, pjy, pjz);
      le[at] = j | ((m - 1u) << 28);
    }
  }
  return d + static_cast<uint32_t>(__popcll(mask));
}

// sorts the d partners by key (rank sort: the keys are distinct) and writes the node's list into the wavefront's pool
PIES_DEV void write_list(const PairArrays& P, const uint64_t* lk, const uint32_t* le, uint32_t i, uint32_t d, uint32_t pool, int lane,
                         const uint32_t* lm = nullptr) {
  uint32_t at = 0;
  for (uint32_t tries = 0;; ++tries) {  // a full pool (long lists of one dense group) passes the node on to the next one
    if (lane == 0 && d) at = atomicAdd(&P.pool[pool * kPairPad], d);
    at = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(at)));
    if (at + d <= P.poolCap) break;
    if (tries + 1u == kPairPools) {  // (the node keeps an empty list; its partners wait for it for ever: flagged, the host latches the failure)
      if (lane == 0) atomicOr(&P.ctl[kPairFlags], 2u);
      return;
    }
    pool = (pool + 1u) % kPairPools;
  }
  const uint32_t off = pool * P.poolCap + at;
  __builtin_amdgcn_wave_barrier();
  uint32_t firstEntry = 0;
  bool haveFirst = false;
  for (uint32_t e = lane; e < d; e += 64) {
    const uint64_t k = lk[e];
    uint32_t rank = 0;
    for (uint32_t f = 0; f < d; ++f) rank += lk[f] < k ? 1u : 0u;
    const uint32_t v = le[e];
    P.nbr[off + rank] = v;
    if (lm) P.nbrM[off + rank] = lm[e];
    if (rank == 0u) { firstEntry = v; haveFirst = true; }
  }
  // the node's record: first entry, entries, cursor 0 reached in round 0, the current entry itself (from the lane that holds it)
  const unsigned long long who = __ballot(haveFirst);
  const uint32_t v0 = who ? static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(firstEntry), __builtin_ctzll(who))) : 0u;
  if (P.byIndex) {
    // reference order by turns (k_turn_round): entries | partners with a lower index << 16, event 0, the node whose turn the first
    // event is (the first partner if it has a lower index, else the node itself); the turn waits for its d + 1 members
    uint32_t below = 0;
    for (uint32_t e = lane; e < d; e += 64) below += lk[e] < static_cast<uint64_t>(i) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) below += __shfl_xor(below, o, 64);
    if (lane == 0) {
      store_rec(P.node, i, make_uint4(off, d | (below << 16), 0u, below ? (v0 & kPairNodeMask) : i));
      P.turnCnt[i] = d + 1u;
    }
  } else if (lane == 0) store_rec(P.node, i, make_uint4(off, d, 0u, v0));
  __builtin_amdgcn_wave_barrier();
}

// BIG: the groups the small kernel passed on (and, beyond MAXC candidates, node by node straight from the buckets)
// (The small instance asks for six wavefronts per SIMD: with two nodes per wavefront it took 100 registers - four wavefronts per
// SIMD, eight groups in flight per compute unit instead of twelve - and the third fewer instructions bought 10 %; at 80 registers
// and 11 spilled words config 4 went 95 / 137 -> 100 / 143.  The level kernel at the same setting was slower: 94 / 136.)
template <uint32_t MAXC, uint32_t MAXD, uint32_t MAXOWN, bool BIG>
__global__ void __launch_bounds__(64 * kBuildWaves, BIG ? 1 : 6) k_pair_build(HashArrays H, PairArrays P, uint32_t repeat) {
  __shared__ BuildLds<MAXC, MAXD, MAXOWN> L;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (repeat && !P.ctl[kPairRetry]) return;
  if (H.counters[kCounterFlags]) return;
  const uint32_t ngroups = BIG ? min(P.ctl[kPairSpilled], P.n) : min(P.ctl[kPairGroups], P.n);
  if (ngroups == 0u) return;
  const GridBox B = grid_box(H.counters);
  const uint32_t* __restrict__ val = H.val[grid_passes(B) & 1u];
  const uint32_t pool = (blockIdx.x * kBuildWaves + wv) % kPairPools;
  uint64_t* lk = L.lk[wv];
  uint32_t* le = L.le[wv];
  uint64_t tested = 0;
  uint32_t edges = 0;
  uint64_t testedAll = 0;
  uint32_t edgesAll = 0;
  for (uint32_t u = blockIdx.x; u < ngroups; u += gridDim.x, testedAll += tested, edgesAll += edges) {  // (workgroup uniform)
    tested = 0;
    edges = 0;
    const uint32_t g = BIG ? P.spill[u] : u;
    const uint4* __restrict__ desc = P.grp + 4ull * g;
    const uint4 d0 = desc[0], d1 = desc[1], d2 = desc[2];
    const uint32_t cStart[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    const uint32_t cCnt[8] = {d2.x & 0xffffu, d2.x >> 16, d2.y & 0xffffu, d2.y >> 16, d2.z & 0xffffu, d2.z >> 16, d2.w & 0xffffu, d2.w >> 16};
    if (threadIdx.x == 0) { L.ncand = 0; L.nown = 0; L.spill = 0; }
    if (threadIdx.x < 8u) {  // (statistics: what the reference's loop would look at for a node of this group, by the lengths of its range)
      uint32_t sum = 0;
#pragma unroll
      for (uint32_t c = 0; c < 8u; ++c)
        if ((c & ~threadIdx.x) == 0u) sum += cCnt[c];
      L.looked[threadIdx.x] = sum;
    }
    __syncthreads();
    // ---- the distinct nodes of the eight buckets, each from its canonical cell; the group's own nodes.  Cells are dealt to the
    // wavefronts; a wavefront reserves a run of the table per 64 entries (the order of the table does not matter)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if ((c % kBuildWaves) != wv) continue;
      for (uint32_t base = 0; base < cCnt[c]; base += 64) {
        uint32_t v = 0;
        bool take = false;
        if (base + lane < cCnt[c]) {
          v = val[cStart[c] + base + lane];
          take = ((v >> kSideShift) & 7u & static_cast<uint32_t>(c)) == 0u;
        }
        const unsigned long long tm = __ballot(take);
        const bool mine = take && c == 0 && (v & kMinFlag) != 0u;
        const unsigned long long mm = __ballot(mine);
        uint32_t at = 0, ao = 0;
        if (lane == 0) {
          at = atomicAdd(&L.ncand, static_cast<uint32_t>(__popcll(tm)));
          if (mm) ao = atomicAdd(&L.nown, static_cast<uint32_t>(__popcll(mm)));
        }
        at = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(at))) + static_cast<uint32_t>(__popcll(tm & ((1ull << lane) - 1ull)));
        ao = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ao))) + static_cast<uint32_t>(__popcll(mm & ((1ull << lane) - 1ull)));
        if (take && at < MAXC) {
          L.id[at] = v & kNodeMask;
          // the node's range relative to the group's cell, from the entry alone: its minimum is this cell minus the side bits
          const uint32_t side = (v >> kSideShift) & 7u, two = (v >> kLongShift) & 7u, cc = static_cast<uint32_t>(c);
          const uint32_t mnx = ((cc >> 2) & 1u) + 1u - ((side >> 2) & 1u), mny = ((cc >> 1) & 1u) + 1u - ((side >> 1) & 1u), mnz = (cc & 1u) + 1u - (side & 1u);
          L.rg[at] = mnx | (mny << 2) | (mnz << 4) | (((two >> 2) & 1u) << 8) | (((two >> 1) & 1u) << 14) | ((two & 1u) << 20);
        }
        if (mine && ao < MAXOWN && at < MAXC) L.own[ao] = static_cast<uint16_t>(at);
      }
    }
    __syncthreads();
    const uint32_t ncand = L.ncand, nown = L.nown;
    if (ncand > MAXC || nown > MAXOWN) {
      if (!BIG) {  // does not fit the small kernel: the large one takes the group
        if (threadIdx.x == 0) P.spill[atomicAdd(&P.ctl[kPairSpilled], 1u)] = g;
        __syncthreads();
        continue;
      }
      // ---- a dense neighbourhood: candidates straight from the buckets, node by node (wavefront 0 alone).  A partner sits in
      // several of the node's cells; it is taken where the cell is the minimum corner of what the two ranges share.
      if (wv == 0) {
        int gx, gy, gz;
        box_cell(B, H.keys[desc[3].x], gx, gy, gz);
        for (uint32_t ge = 0; ge < cCnt[0]; ++ge) {
          const uint32_t v = val[cStart[0] + ge];
          if (!(v & kMinFlag)) continue;  // (wave uniform)
          const uint32_t i = v & kNodeMask;
          const float4 pi = P.node[4u * i + 2u];
          const float rsi = P.node[4u * i + 1u].w + pi.w;
          const int4 rgi = H.rng[i];
          const uint32_t lxi = rgi.w & 0xff, lyi = (rgi.w >> 8) & 0xff, lzi = (rgi.w >> 16) & 0xff;
          uint32_t d = 0;
          for (uint32_t dx = 0; dx < lxi; ++dx)
            for (uint32_t dy = 0; dy < lyi; ++dy)
              for (uint32_t dz = 0; dz < lzi; ++dz) {
                const uint32_t c = (dx * 4 + dy * 2 + dz) & 7u;
                uint32_t bs = 0, bc = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                  if (c == static_cast<uint32_t>(q)) { bs = cStart[q]; bc = cCnt[q]; }
                tested += bc;
                const int cx = gx + static_cast<int>(dx), cy = gy + static_cast<int>(dy), cz = gz + static_cast<int>(dz);
                for (uint32_t base = 0; base < bc; base += 64) {
                  bool accept = false;
                  uint32_t j = 0, m = 0;
                  float4 pj = make_float4(0.f, 0.f, 0.f, 0.f);
                  if (base + lane < bc) {
                    j = val[bs + base + lane] & kNodeMask;
                    if (j != i) {
                      const int4 rgj = H.rng[j];
                      if (cx == max(rgi.x, rgj.x) && cy == max(rgi.y, rgj.y) && cz == max(rgi.z, rgj.z)) {
                        m = shared_cells(rgi.x, lxi, rgj.x, rgj.w & 0xff) * shared_cells(rgi.y, lyi, rgj.y, (rgj.w >> 8) & 0xff) *
                            shared_cells(rgi.z, lzi, rgj.z, (rgj.w >> 16) & 0xff);
                        pj = P.node[4u * j + 2u];
                        const float ddx = pj.x - pi.x, ddy = pj.y - pi.y, ddz = pj.z - pi.z;
                        const float cut = 1.001f * (rsi + (P.node[4u * j + 1u].w + pj.w));
                        accept = m != 0u && !(ddx * ddx + ddy * ddy + ddz * ddz >= cut * cut);
                      }
                    }
                  }
                  d = push_partners<MAXD>(lk, le, d, accept, i, j, m, pi.x, pi.y, pi.z, pj.x, pj.y, pj.z, lane, P.byIndex != 0u);
                }
              }
          __builtin_amdgcn_wave_barrier();
          if (d > MAXD) {  // a pile-up beyond anything a simulation survives: latch, like the > 2048 nodes in a cell of the grid
            if (lane == 0) atomicOr(&P.ctl[kPairFlags], 2u);
            d = MAXD;
          }
          write_list(P, lk, le, i, d, pool, lane);
          edges += d;
        }
      }
      __syncthreads();
      continue;
    }
    for (uint32_t t = threadIdx.x; t < ncand; t += 64 * kBuildWaves) {
      const float4 p = P.bq[L.id[t]];  // (the one gather per candidate)
      L.px[t] = p.x; L.py[t] = p.y; L.pz[t] = p.z; L.rs[t] = p.w;
    }
    __syncthreads();
    // Two own nodes per turn of a wavefront.  The first sweep - the distance test over all ~300 candidates - runs node after node
    // with all 64 lanes and leaves the slots of the ~25 within reach in LDS; when both nodes have at most 32 of them (nearly
    // always) the second sweep - shared cells and the pair key, 60 % of the instructions of a candidate round - and the rank sort
    // run for both at once, a half of the wavefront each: the kernel is bound by VALU issue (~350 wavefront instructions per
    // node) and these two parts kept 25 and 11 of 64 lanes busy.
    bool stop = false;
    for (uint32_t o = 2u * static_cast<uint32_t>(wv); o < nown && !stop; o += 2u * kBuildWaves) {
      const bool haveB = o + 1u < nown;
      uint32_t nnAB[2] = {0u, 0u};
      for (uint32_t h = 0; h < (haveB ? 2u : 1u); ++h) {  // ---- first sweep
        const uint32_t si = L.own[o + h];
        const float pix = L.px[si], piy = L.py[si], piz = L.pz[si], rsi = L.rs[si];
        uint16_t* near = L.near[wv][h];
        // a lane keeps the rounds in which its candidate was within reach as bits and the slots are compacted once at the end (a
        // ballot, two bit counts and a bounds test per round were two thirds of a round's instructions); up to 32 rounds (MAXC <= 2048)
        uint32_t bits = 0;
        for (uint32_t base = 0, r = 0; base < ncand; base += 64, ++r) {
          const uint32_t t = base + static_cast<uint32_t>(lane);
          if (t < ncand && t != si) {
            const float ddx = L.px[t] - pix, ddy = L.py[t] - piy, ddz = L.pz[t] - piz;
            const float cut = 1.001f * (rsi + L.rs[t]);  // (wide for a node that left its slack in the first attempt)
            if (!(ddx * ddx + ddy * ddy + ddz * ddz >= cut * cut)) bits |= 1u << r;
          }
        }
        const uint32_t mine = static_cast<uint32_t>(__popc(bits));
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t up = __shfl_up(incl, off, 64);
          if (lane >= off) incl += up;
        }
        const uint32_t nn = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
        uint32_t at = incl - mine;
        while (bits) {
          const uint32_t r = static_cast<uint32_t>(__builtin_ctz(bits));
          bits &= bits - 1u;
          if (at < MAXD) near[at] = static_cast<uint16_t>(r * 64u + static_cast<uint32_t>(lane));
          ++at;
        }
        nnAB[h] = nn;
      }
      __builtin_amdgcn_wave_barrier();
      // the candidates the reference's loop would look at for a node (statistics: SURVEY 8d counts 16 B for each): the cells
      // (dx, dy, dz) below its range's lengths, from the group's table (a loop over cCnt[] here cost 150-260 instructions per node)
      auto looked_of = [&](uint32_t rgi) { return L.looked[(((rgi >> 8) & 1u) << 2) | (((rgi >> 14) & 1u) << 1) | ((rgi >> 20) & 1u)]; };
      if (!BIG && haveB && nnAB[0] <= 32u && nnAB[1] <= 32u && P.byIndex == 0u) {
        // ---- second sweep and lists of both nodes, a half of the wavefront each
        const uint32_t h = static_cast<uint32_t>(lane) >> 5, hl = static_cast<uint32_t>(lane) & 31u;
        const unsigned long long halfMask = 0xffffffffull << (32u * h);
        const uint32_t si = L.own[o + h];
        const uint32_t i = L.id[si];
        const float pix = L.px[si], piy = L.py[si], piz = L.pz[si];
        const uint32_t rgi = L.rg[si];
        const uint32_t lxi = ((rgi >> 8) & 63u) + 1u, lyi = ((rgi >> 14) & 63u) + 1u, lzi = ((rgi >> 20) & 63u) + 1u;
        const uint32_t nn = h ? nnAB[1] : nnAB[0];
        uint64_t* lkh = lk + h * (MAXD / 2u);
        uint32_t* leh = le + h * (MAXD / 2u);
        bool accept = false;
        uint32_t j = 0, m = 0;
        float pjx = 0.f, pjy = 0.f, pjz = 0.f;
        if (hl < nn) {
          const uint32_t t = L.near[wv][h][hl];
          pjx = L.px[t]; pjy = L.py[t]; pjz = L.pz[t];
          j = L.id[t];
          const uint32_t rgj = L.rg[t];
          m = shared_cells(0, lxi, static_cast<int>(rgj & 3u) - 1, ((rgj >> 8) & 63u) + 1u) *
              shared_cells(0, lyi, static_cast<int>((rgj >> 2) & 3u) - 1, ((rgj >> 14) & 63u) + 1u) *
              shared_cells(0, lzi, static_cast<int>((rgj >> 4) & 3u) - 1, ((rgj >> 20) & 63u) + 1u);
          accept = m != 0u;
        }
        const unsigned long long am = __ballot(accept) & halfMask;
        const uint32_t d = static_cast<uint32_t>(__popcll(am));
        if (accept) {
          const uint32_t at = static_cast<uint32_t>(__popcll(am & ((1ull << lane) - 1ull)));
          lkh[at] = pair_key_of(i, j, pix, piy, piz, pjx, pjy, pjz);
          leh[at] = j | ((m - 1u) << 28);
        }
        __builtin_amdgcn_wave_barrier();
        // storage from a pool (a full pool passes the node on to the next one, like write_list)
        uint32_t poolH = pool, at = 0;
        bool placed = false, failed = false;
        for (uint32_t tries = 0; tries < kPairPools; ++tries) {
          uint32_t got = 0;
          if (!placed && hl == 0u && d) got = atomicAdd(&P.pool[poolH * kPairPad], d);
          got = static_cast<uint32_t>(__shfl(static_cast<int>(got), static_cast<int>(32u * h), 64));
          if (!placed) {
            if (got + d <= P.poolCap) { at = got; placed = true; }
            else if (tries + 1u == kPairPools) { failed = true; placed = true; }
            else poolH = (poolH + 1u) % kPairPools;
          }
          if (__ballot(!placed) == 0ull) break;
        }
        if (failed && hl == 0u) atomicOr(&P.ctl[kPairFlags], 2u);  // (the node keeps an empty list: flagged, the pass goes to the sequential loop)
        const uint32_t off = poolH * P.poolCap + at;
        uint32_t firstEntry = 0;
        bool haveFirst = false;
        const uint32_t dMax = max(static_cast<uint32_t>(__shfl(static_cast<int>(d), 0, 64)), static_cast<uint32_t>(__shfl(static_cast<int>(d), 32, 64)));
        if (!failed) {
          const uint64_t k = hl < d ? lkh[hl] : 0ull;
          uint32_t rank = 0;
          for (uint32_t f = 0; f < dMax; ++f) rank += (f < d && lkh[min(f, MAXD / 2u - 1u)] < k) ? 1u : 0u;
          if (hl < d) {
            const uint32_t v = leh[hl];
            P.nbr[off + rank] = v;
            if (rank == 0u) { firstEntry = v; haveFirst = true; }
          }
        }
        const unsigned long long who = __ballot(haveFirst) & halfMask;
        const uint32_t v0 = static_cast<uint32_t>(__shfl(static_cast<int>(firstEntry), who ? __builtin_ctzll(who) : 0, 64));
        if (hl == 0u && !failed) store_rec(P.node, i, make_uint4(off, d, 0u, who ? v0 : 0u));
        __builtin_amdgcn_wave_barrier();
        tested += looked_of(L.rg[L.own[o]]) + looked_of(L.rg[L.own[o + 1u]]);
        edges += static_cast<uint32_t>(__shfl(static_cast<int>(d), 0, 64)) + static_cast<uint32_t>(__shfl(static_cast<int>(d), 32, 64));
        continue;
      }
      for (uint32_t h = 0; h < (haveB ? 2u : 1u) && !stop; ++h) {  // ---- a node at a time (long lists, the large kernel, the order by turns)
        const uint32_t si = L.own[o + h];
        const uint32_t i = L.id[si];
        const float pix = L.px[si], piy = L.py[si], piz = L.pz[si];
        const uint32_t rgi = L.rg[si];
        const uint32_t lxi = ((rgi >> 8) & 63u) + 1u, lyi = ((rgi >> 14) & 63u) + 1u, lzi = ((rgi >> 20) & 63u) + 1u;
        const uint16_t* near = L.near[wv][h];
        const uint32_t nn = nnAB[h];
        uint32_t d = nn > MAXD ? MAXD + 1u : 0u;  // (more within reach than a list holds: handled below like a list that is too long)
        for (uint32_t base = 0; base < nn && nn <= MAXD; base += 64) {
          const uint32_t e = base + static_cast<uint32_t>(lane);
          bool accept = false;
          uint32_t j = 0, m = 0;
          float pjx = 0.f, pjy = 0.f, pjz = 0.f;
          if (e < nn) {
            const uint32_t t = near[e];
            pjx = L.px[t]; pjy = L.py[t]; pjz = L.pz[t];
            j = L.id[t];
            const uint32_t rgj = L.rg[t];
            m = shared_cells(0, lxi, static_cast<int>(rgj & 3u) - 1, ((rgj >> 8) & 63u) + 1u) *
                shared_cells(0, lyi, static_cast<int>((rgj >> 2) & 3u) - 1, ((rgj >> 14) & 63u) + 1u) *
                shared_cells(0, lzi, static_cast<int>((rgj >> 4) & 3u) - 1, ((rgj >> 20) & 63u) + 1u);
            accept = m != 0u;
          }
          d = push_partners<MAXD>(lk, le, d, accept, i, j, m, pix, piy, piz, pjx, pjy, pjz, lane, P.byIndex != 0u);
        }
        __builtin_amdgcn_wave_barrier();
        if (d > MAXD) {
          if (!BIG) { if (lane == 0) L.spill = 1; stop = true; break; }  // (the large kernel redoes the group; lists written so far are replaced)
          if (lane == 0) atomicOr(&P.ctl[kPairFlags], 2u);  // a pile-up beyond anything a simulation survives: latch
          d = MAXD;
        }
        tested += looked_of(rgi);
        write_list(P, lk, le, i, d, pool, lane);
        edges += d;
      }
    }
    __syncthreads();  // (the table is reused by the next group)
    if (!BIG && L.spill) {
      if (threadIdx.x == 0) P.spill[atomicAdd(&P.ctl[kPairSpilled], 1u)] = g;
      tested = 0;  // (what this group has counted so far is counted again by the large kernel)
      edges = 0;
      __syncthreads();
    }
  }
  // statistics, striped (same-address atomics of 16 000 wavefronts would take longer than the lists)
  if (lane == 0 && testedAll) atomicAdd(reinterpret_cast<unsigned long long*>(&P.stat[(blockIdx.x % 64u) * kPairPad]), static_cast<unsigned long long>(testedAll));
  if (lane == 0 && edgesAll) atomicAdd(&P.stat[(blockIdx.x % 64u) * kPairPad + 2u], edgesAll);
}

// ---- lists for scenes whose ranges are wider than two cells per axis (gridSpacing < 2 (r + 0.5): NodeCompRange allows up to 50
// cells per axis): one wavefront per NODE walks the buckets of the node's own range.  A partner sits in several of them; it is
// taken where the cell is the minimum corner of what the two ranges share (a look at the partner's range: one more gather per
// candidate than the 2x2x2 path needs).  The number of shared cells does not fit the entry's four bits: it goes to nbrM.
struct WideLds {
  uint64_t lk[kMaxDeg];
  uint32_t le[kMaxDeg], lm[kMaxDeg];
};
__global__ void __launch_bounds__(64) k_pair_build_wide(HashArrays H, PairArrays P, uint32_t repeat) {
  __shared__ WideLds L;
  const int lane = threadIdx.x;
  if (repeat && !P.ctl[kPairRetry]) return;
  if (H.counters[kCounterFlags]) return;
  const GridBox B = grid_box(H.counters);
  const uint32_t* __restrict__ val = H.val[grid_passes(B) & 1u];
  const uint32_t pool = blockIdx.x % kPairPools;
  uint64_t tested = 0;
  uint32_t edges = 0;
  for (uint32_t i = blockIdx.x; i < P.n; i += gridDim.x) {
    const int4 rgi = H.rng[i];
    const uint32_t lxi = rgi.w & 0xff, lyi = (rgi.w >> 8) & 0xff, lzi = (rgi.w >> 16) & 0xff;
    if (lxi * lyi * lzi == 0u) continue;  // an over-long range is empty (Solver.cpp:896-898): the node visits nothing
    const float4 pi = P.node[4u * i + 2u];
    const float rsi = P.node[4u * i + 1u].w + pi.w;
    uint32_t d = 0;
    for (uint32_t dx = 0; dx < lxi; ++dx)
      for (uint32_t dy = 0; dy < lyi; ++dy)
        for (uint32_t dz = 0; dz < lzi; ++dz) {
          const int cx = rgi.x + static_cast<int>(dx), cy = rgi.y + static_cast<int>(dy), cz = rgi.z + static_cast<int>(dz);
          const uint32_t cs = find_bucket(H, B, cx, cy, cz);
          if (cs == 0xffffffffu) continue;
          const uint32_t bs = H.start[cs], bc = H.end[cs] - bs;
          tested += bc;
          for (uint32_t base = 0; base < bc; base += 64) {
            bool accept = false;
            uint32_t j = 0, m = 0;
            float4 pj = make_float4(0.f, 0.f, 0.f, 0.f);
            if (base + lane < bc) {
              j = val[bs + base + lane] & kNodeMask;
              if (j != i) {
                const int4 rgj = H.rng[j];
                if (cx == max(rgi.x, rgj.x) && cy == max(rgi.y, rgj.y) && cz == max(rgi.z, rgj.z)) {
                  m = shared_cells(rgi.x, lxi, rgj.x, rgj.w & 0xff) * shared_cells(rgi.y, lyi, rgj.y, (rgj.w >> 8) & 0xff) *
                      shared_cells(rgi.z, lzi, rgj.z, (rgj.w >> 16) & 0xff);
                  pj = P.node[4u * j + 2u];
                  const float ddx = pj.x - pi.x, ddy = pj.y - pi.y, ddz = pj.z - pi.z;
                  const float cut = 1.001f * (rsi + (P.node[4u * j + 1u].w + pj.w));
                  accept = m != 0u && !(ddx * ddx + ddy * ddy + ddz * ddz >= cut * cut);
                }
              }
            }
            const unsigned long long mask = __ballot(accept);
            if (accept) {
              const uint32_t at = d + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
              if (at < kMaxDeg) {
                L.lk[at] = P.byIndex ? static_cast<uint64_t>(j) : pair_key_of(i, j, pi.x, pi.y, pi.z, pj.x, pj.y, pj.z);
                L.le[at] = j;
                L.lm[at] = m;
              }
            }
            d += static_cast<uint32_t>(__popcll(mask));
          }
        }
    __builtin_amdgcn_wave_barrier();
    if (d > kMaxDeg) {  // a pile-up beyond anything a simulation survives: latch
      if (lane == 0) atomicOr(&P.ctl[kPairFlags], 2u);
      d = kMaxDeg;
    }
    write_list(P, L.lk, L.le, i, d, pool, lane, L.lm);
    edges += d;
  }
  if (lane == 0 && tested) atomicAdd(reinterpret_cast<unsigned long long*>(&P.stat[(blockIdx.x % 64u) * kPairPad]), static_cast<unsigned long long>(tested));
  if (lane == 0 && edges) atomicAdd(&P.stat[(blockIdx.x % 64u) * kPairPad + 2u], edges);
}

// The frontier of a round is kept as kPairLists sub-lists.  A wavefront works on chunks of 64 consecutive positions of their
// concatenation and appends to the sub-list its chunk is dealt to (chunk index modulo kPairLists).
struct FrontierView {
  uint32_t incl;   // lane s: entries of sub-lists 0 .. s
  uint32_t total;
};
PIES_DEV FrontierView frontier_view(const PairArrays& P, uint32_t round, int lane) {
  FrontierView v;
  if (round == 1u) { v.incl = 0; v.total = P.n; return v; }  // (all nodes, by index)
  uint32_t c = min(__hip_atomic_load(&P.frCount[((round % 3u) * kPairLists + static_cast<uint32_t>(lane)) * kPairPad], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), P.frCap);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(c, off, 64);
    if (lane >= off) c += t;
  }
  v.incl = c;
  v.total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(c), 63));
  return v;
}
// position e of the concatenated sub-lists -> the node (e < total)
PIES_DEV uint32_t frontier_node(const PairArrays& P, const FrontierView& v, uint32_t round, uint32_t e) {
  if (round == 1u) return e;
  uint32_t lo = 0;  // number of sub-lists that end at or before e: binary search over the lanes' inclusive sums
#pragma unroll
  for (int bit = 32; bit >= 1; bit >>= 1) {
    const uint32_t probe = lo + static_cast<uint32_t>(bit) - 1u;
    const uint32_t end = __shfl(v.incl, static_cast<int>(probe), 64);
    if (end <= e) lo += static_cast<uint32_t>(bit);
  }
  const uint32_t prev = __shfl(v.incl, static_cast<int>(lo ? lo - 1u : 0u), 64);  // (every lane shuffles: a lane must not read from an idle one)
  const uint32_t before = lo ? prev : 0u;
  return P.fr[round & 1u][lo * P.frCap + (e - before)];
}

PIES_DEV void process_frontier(const HashArrays& H, const PairArrays& P, float friction, float staticThreshold, uint32_t round, uint32_t first,
                               uint32_t step, const FrontierView& view, int lane, uint32_t& hits) {
  float4* node = P.node;
  uint32_t* next = P.fr[(round + 1u) & 1u];
  uint32_t* nextCount = P.frCount + ((round + 1u) % 3u) * kPairLists * kPairPad;
  const uint32_t stampNow = round & 0xffffu, stampPrev = (round - 1u) & 0xffffu;
  const uint32_t count = view.total;
  // moves a node on to its next entry (reached in this round); returns whether it has one
  auto move_on = [&](uint32_t i, const uint4 r) {
    const uint32_t c = (r.z & 0xffffu) + 1u;
    const uint32_t entry = (c < r.y ? P.nbr[r.x + c] : 0u) | ((c & 7u) << kPairTagShift);
    store_rec(node, i, make_uint4(r.x, r.y, c | (stampNow << 16), entry));
    return c < r.y;
  };
  for (uint32_t base = first - static_cast<uint32_t>(lane); base < count; base += step) {  // (wave uniform trip count)
    const uint32_t e = base + static_cast<uint32_t>(lane);
    bool moveX = false, moveY = false;
    uint32_t x = 0, y = 0;
    const uint32_t xe = frontier_node(P, view, round, min(e, count - 1u));  // (every lane takes part in the shuffles)
    if (e < count) {
      x = xe;
      const uint4 rx = load_rec(node, x);
      // (x's own record must still be the one it reached in the last round: its partner's lane may have moved it on already)
      if (rec_consistent(rx) && (rx.z & 0xffffu) < rx.y && (rx.z >> 16) == stampPrev) {
        y = rx.w & kPairNodeMask;
        const uint4 ry = load_rec(node, y);
        const uint32_t sy = ry.z >> 16;
        bool take = rec_consistent(ry) && (ry.z & 0xffffu) < ry.y && sy != stampNow && (ry.w & kPairNodeMask) == x;
        if (take && sy == stampPrev && y < x) take = false;  // y is in this frontier as well and takes the pair
        if (take) {
          const bool xLow = x < y;
          const uint32_t lo = xLow ? x : y, hi = xLow ? y : x;
          NodeState a = load_node(node, lo), b = load_node(node, hi);
          const float dx = b.px - a.px, dy = b.py - a.py, dz = b.pz - a.pz;
          const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
          if (a.r + b.r - dist > 0.0f) {
            const float4 a0 = node[4u * lo + 2u], b0 = node[4u * hi + 2u];
            const uint32_t m = P.nbrM ? P.nbrM[rx.x + (rx.z & 0xffffu)] : (rx.w >> 28) + 1u;  // (wide ranges keep the count beside the entry)
            uint32_t h = 0;
            for (uint32_t t = 0; t < m; ++t) h += visit(a, b, friction, staticThreshold) ? 1u : 0u;
            for (uint32_t t = 0; t < m; ++t) h += visit(b, a, friction, staticThreshold) ? 1u : 0u;
            store_node(node, lo, a);
            store_node(node, hi, b);
            note_excursion(P, lo, a, a0);
            note_excursion(P, hi, b, b0);
            hits += h;
          }
          moveX = move_on(x, rx);
          moveY = move_on(y, ry);
        }
      }
    }
    // the nodes that moved on and have entries left go to the sub-list this chunk is dealt to (one atomic per wavefront; a
    // sub-list takes at most 128 nodes from each of its chunks: frCap covers that)
    const unsigned long long mx = __ballot(moveX), my = __ballot(moveY);
    const uint32_t nx = static_cast<uint32_t>(__popcll(mx)), ny = static_cast<uint32_t>(__popcll(my));
    if (nx + ny) {
      const uint32_t sub = (base >> 6) % kPairLists;
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(&nextCount[sub * kPairPad], nx + ny);
      at = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(at)));
      uint32_t* dst = next + static_cast<size_t>(sub) * P.frCap;
      const uint32_t ix = at + static_cast<uint32_t>(__popcll(mx & ((1ull << lane) - 1ull)));
      const uint32_t iy = at + nx + static_cast<uint32_t>(__popcll(my & ((1ull << lane) - 1ull)));
      if (moveX && ix < P.frCap) dst[ix] = x;
      if (moveY && iy < P.frCap) dst[iy] = y;
    }
  }
}

// One level as a launch of 256-thread workgroups.  Half the lanes of a frontier find that they have nothing to do (the other
// node's lane takes the pair, or the partner is not there yet), and a lane that takes a pair runs up to sixteen visits of 350
// instructions while the rest of its wavefront waits - with one wavefront per workgroup the 1 125 wavefronts of a settled
// level of config 4 each carried a few such lanes, and the hundred SIMDs that got two of them set the level's time.  Here the
// workgroup's four wavefronts look at their nodes, the lanes that take a pair put it into LDS, and the pairs are dealt out
// again densely: the first wavefronts get full loads, the others leave.  Half as many wavefronts run visits, one per SIMD.
constexpr uint32_t kRoundBlock = 256;
struct TakenPair {
  uint32_t x, y;
  uint4 rx, ry;
};
__global__ void __launch_bounds__(kRoundBlock) k_pair_round(HashArrays H, PairArrays P, float friction, float staticThreshold, uint32_t round, uint32_t repeat) {
  __shared__ TakenPair taken[kRoundBlock];
  __shared__ uint32_t waveTook[kRoundBlock / 64];
  if (repeat && !P.ctl[kPairRetry]) return;
  if (H.counters[kCounterFlags]) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const FrontierView view = frontier_view(P, round, lane);
  if (blockIdx.x == 0 && wv == 0) {
    P.frCount[(((round + 2u) % 3u) * kPairLists + static_cast<uint32_t>(lane)) * kPairPad] = 0;  // the lists of the round after the next (read by the previous launch, filled by the next)
    if (lane == 0 && view.total) { P.ctl[kPairRounds] = round; if (!repeat && round > P.ctl[kPairDeepest]) P.ctl[kPairDeepest] = round; }
  }
  const uint32_t count = view.total;
  if (count == 0u) return;  // (a pass that will be repeated is finished all the same: it finds every node that leaves its slack)
  float4* node = P.node;
  uint32_t* next = P.fr[(round + 1u) & 1u];
  uint32_t* nextCount = P.frCount + ((round + 1u) % 3u) * kPairLists * kPairPad;
  const uint32_t stampNow = round & 0xffffu, stampPrev = (round - 1u) & 0xffffu;
  auto move_on = [&](uint32_t i, const uint4 r) {
    const uint32_t c = (r.z & 0xffffu) + 1u;
    const uint32_t entry = (c < r.y ? P.nbr[r.x + c] : 0u) | ((c & 7u) << kPairTagShift);
    store_rec(node, i, make_uint4(r.x, r.y, c | (stampNow << 16), entry));
    return c < r.y;
  };
  uint32_t hits = 0;
  for (uint32_t chunk = blockIdx.x; chunk * kRoundBlock < count; chunk += gridDim.x) {  // (workgroup uniform)
    const uint32_t e = chunk * kRoundBlock + threadIdx.x;
    // ---- who takes a pair
    bool take = false;
    uint32_t x = 0, y = 0;
    uint4 rx = make_uint4(0u, 0u, 0u, 0u), ry = rx;
    const uint32_t xe = frontier_node(P, view, round, min(e, count - 1u));  // (every lane takes part in the shuffles)
    if (e < count) {
      x = xe;
      rx = load_rec(node, x);
      // (x's own record must still be the one it reached in the last round: its partner's lane may have moved it on already)
      if (rec_consistent(rx) && (rx.z & 0xffffu) < rx.y && (rx.z >> 16) == stampPrev) {
        y = rx.w & kPairNodeMask;
        ry = load_rec(node, y);
        const uint32_t sy = ry.z >> 16;
        take = rec_consistent(ry) && (ry.z & 0xffffu) < ry.y && sy != stampNow && (ry.w & kPairNodeMask) == x;
        if (take && sy == stampPrev && y < x) take = false;  // y is in this frontier as well and takes the pair
      }
    }
    // ---- the taken pairs, densely
    const unsigned long long tm = __ballot(take);
    if (lane == 0) waveTook[wv] = static_cast<uint32_t>(__popcll(tm));
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kRoundBlock / 64; ++w) {
      const uint32_t c = waveTook[w];
      if (w < static_cast<uint32_t>(wv)) before += c;
      total += c;
    }
    if (take) taken[before + static_cast<uint32_t>(__popcll(tm & ((1ull << lane) - 1ull)))] = TakenPair{x, y, rx, ry};
    __syncthreads();
    // ---- the visits
    bool moveX = false, moveY = false;
    if (threadIdx.x < total) {
      const TakenPair t = taken[threadIdx.x];
      x = t.x; y = t.y;
      const bool xLow = x < y;
      const uint32_t lo = xLow ? x : y, hi = xLow ? y : x;
      NodeState a = load_node(node, lo), b = load_node(node, hi);
      const float dx = b.px - a.px, dy = b.py - a.py, dz = b.pz - a.pz;
      const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
      if (a.r + b.r - dist > 0.0f) {
        const float4 a0 = node[4u * lo + 2u], b0 = node[4u * hi + 2u];
        const uint32_t m = P.nbrM ? P.nbrM[t.rx.x + (t.rx.z & 0xffffu)] : (t.rx.w >> 28) + 1u;  // (wide ranges keep the count beside the entry)
        uint32_t h = 0;
        for (uint32_t v = 0; v < m; ++v) h += visit(a, b, friction, staticThreshold) ? 1u : 0u;
        for (uint32_t v = 0; v < m; ++v) h += visit(b, a, friction, staticThreshold) ? 1u : 0u;
        store_node(node, lo, a);
        store_node(node, hi, b);
        note_excursion(P, lo, a, a0);
        note_excursion(P, hi, b, b0);
        hits += h;
      }
      moveX = move_on(x, t.rx);
      moveY = move_on(y, t.ry);
    }
    // the nodes that moved on and have entries left go to the sub-list this wavefront's chunk is dealt to (one atomic per
    // wavefront; a sub-list takes at most 128 nodes from each of its chunks: frCap covers that)
    if (static_cast<uint32_t>(wv) * 64u < total) {  // (wavefront uniform)
      const unsigned long long mx = __ballot(moveX), my = __ballot(moveY);
      const uint32_t nx = static_cast<uint32_t>(__popcll(mx)), ny = static_cast<uint32_t>(__popcll(my));
      if (nx + ny) {
        const uint32_t sub = (chunk * (kRoundBlock / 64) + static_cast<uint32_t>(wv)) % kPairLists;
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(&nextCount[sub * kPairPad], nx + ny);
        at = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(at)));
        uint32_t* dst = next + static_cast<size_t>(sub) * P.frCap;
        const uint32_t ix = at + static_cast<uint32_t>(__popcll(mx & ((1ull << lane) - 1ull)));
        const uint32_t iy = at + nx + static_cast<uint32_t>(__popcll(my & ((1ull << lane) - 1ull)));
        if (moveX && ix < P.frCap) dst[ix] = x;
        if (moveY && iy < P.frCap) dst[iy] = y;
      }
    }
    __syncthreads();  // (the table is reused by the next chunk)
  }
  count_hits(P, hits, lane);
}

// ---- the same level with FOUR LANES PER PAIR (round 5) ---------------------------------------------------------------------
// A lane that takes a pair runs up to sixteen visits of ~350 instructions - fifteen correctly rounded divisions each - while a level
// waits for its slowest lane.  Here a pair is taken by a quad of lanes, lane k holding component k of the positions and velocities
// (lane 3 idles along with a copy of component 0): the three sums of a visit (|d|^2, r.u, |q|^2) are quad permutes (v_mov_dpp) added
// in visit()'s order, every lane divides for its own component only - five divisions instead of fifteen, ~120 instructions
// instead of ~350 -, same operations on the same operands: the bits of visit().  A workgroup looks at 64 frontier nodes (its first
// wavefront), the pairs taken go through LDS, and all four wavefronts run their visits.
// Measured on BASELINE config 4: a settled frame 9.16 -> 8.76-8.9 ms (burst window 71.8 -> 73.0 substeps/s): a third of the
// instructions in the slowest lane bought 4 %, because the levels are bound by the NUMBER of wavefront instructions issued (per-level
// counters: profiles/r05_levels_pmc_config4_before.txt), and a quad wavefront holds 16 pairs where a lane-per-pair one holds 64: what
// pays is full wavefronts of quads with equal numbers of visits - the sorted table of pair_level4.
PIES_DEV float quad_lane(float v, int j) {  // the value of lane j (0-3) of the lane's quad
  switch (j) {
    case 0: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x00, 0xf, 0xf, false));
    case 1: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x55, 0xf, 0xf, false));
    case 2: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xaa, 0xf, 0xf, false));
    default: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xff, 0xf, 0xf, false));
  }
}
PIES_DEV float quad_sum3(float t) { return (quad_lane(t, 0) + quad_lane(t, 1)) + quad_lane(t, 2); }  // x + y + z in visit()'s order
struct QuadNode {
  float p, v;  // component k of position and velocity
  float w, r;  // inverse mass, radius (every lane)
};
// node a visits node b (visit(), one component per lane; the result of the overlap test is the same in the quad's lanes)
PIES_DEV bool visit_quad(QuadNode& a, QuadNode& b, int k, float friction, float staticThreshold) {
  const float d = b.p - a.p;
  const float dist = sqrtf(quad_sum3(d * d));
  const float disp = a.r + b.r - dist;
  if (!(disp > 0.0f)) return false;
  float u = k == 0 ? 1.0f : 0.0f;
  if (dist > 0.00001f) u = d / dist;
  const float wSum = a.w + b.w;
  const float sa = 0.85f * -disp, sb = 0.85f * disp;
  const float r = b.v - a.v;
  const float rd = quad_sum3(r * u);
  const float q = r - rd * u;
  float fr = friction;
  if (staticThreshold > 0.0f)  // sqrt(x) < t is false for every t <= 0
    if (sqrtf(quad_sum3(q * q)) < staticThreshold) fr = 1.0f;
  a.p += ((sa * u) * a.w) / wSum;
  b.p += ((sb * u) * b.w) / wSum;
  a.v += ((-fr * q) * a.w) / wSum;
  b.v += ((fr * q) * b.w) / wSum;
  return true;
}
PIES_DEV float comp4(const float4 v, int k) { return k == 1 ? v.y : (k == 2 ? v.z : v.x); }  // (lane 3: a copy of component 0)

constexpr uint32_t kQuadNodes = 64;     // frontier nodes a wavefront looks at per turn of a workgroup's loop
constexpr uint32_t kQuadLookMax = 4;    // wavefronts of a workgroup that look (PIES_PAIR_LOOK_WAVES: 1, 2 or 4): the table holds 64 pairs for each
constexpr uint32_t kQuadBins = 9;       // pairs by visits per side: 8 and more ... 1, and the ones that do not overlap
struct QuadTable {
  TakenPair taken[kQuadNodes * kQuadLookMax];
  uint32_t bins[64];  // [bin * look + wavefront] (9 x 4 used)
};
// one level by the workgroups of the calling launch (all threads of a workgroup call: barriers); T: the workgroup's LDS; look: the
// wavefronts of the workgroup that look at frontier nodes (at most blockDim.x / 64)
PIES_DEV void pair_level4(const HashArrays& H, const PairArrays& P, float friction, float staticThreshold, uint32_t round, const FrontierView& view,
                          QuadTable& T, uint32_t look, uint32_t& hits) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t count = view.total;
  float4* node = P.node;
  uint32_t* next = P.fr[(round + 1u) & 1u];
  uint32_t* nextCount = P.frCount + ((round + 1u) % 3u) * kPairLists * kPairPad;
  const uint32_t stampNow = round & 0xffffu, stampPrev = (round - 1u) & 0xffffu;
  const int k = lane & 3;
  const uint32_t per = kQuadNodes * look;  // frontier nodes per turn
  TakenPair* taken = T.taken;
  for (uint32_t chunk = blockIdx.x; chunk * per < count; chunk += gridDim.x) {  // (workgroup uniform)
    // ---- who takes a pair: the first wavefronts look at the chunk's frontier nodes, 64 each
    bool take = false;
    uint32_t x = 0, y = 0, key = 0, rank = 0;
    uint4 rx = make_uint4(0u, 0u, 0u, 0u), ry = rx;
    if (static_cast<uint32_t>(wv) < look) {
      const uint32_t e = chunk * per + static_cast<uint32_t>(wv) * kQuadNodes + static_cast<uint32_t>(lane);
      const uint32_t xe = frontier_node(P, view, round, min(e, count - 1u));  // (every lane takes part in the shuffles)
      if (e < count) {
        x = xe;
        rx = load_rec(node, x);
        // (x's own record must still be the one it reached in the last round: its partner's lane may have moved it on already)
        if (rec_consistent(rx) && (rx.z & 0xffffu) < rx.y && (rx.z >> 16) == stampPrev) {
          y = rx.w & kPairNodeMask;
          ry = load_rec(node, y);
          const uint32_t sy = ry.z >> 16;
          take = rec_consistent(ry) && (ry.z & 0xffffu) < ry.y && sy != stampNow && (ry.w & kPairNodeMask) == x;
          if (take && sy == stampPrev && y < x) take = false;  // y is in this frontier as well and takes the pair
        }
      }
      // The pairs go into the table sorted: the ones that overlap first, by descending number of shared cells (= visits from either
      // side), the ones that do not (most, once a pile has settled: they only move on) last.  A wavefront of the visits runs as long
      // as its busiest quad, and the level launches of config 4 turned out to be bound by VALU issue (profiles/
      // r05_levels_pmc_config4.txt: 20 M wavefront instructions in level 1, SQ_ACTIVE_INST_ANY x 8 resident wavefronts > a SIMD's
      // cycles): in arrival order a wavefront's sixteen pairs held ~6 overlapping ones with 1-8 shared cells each.  The overlap test
      // is visit()'s own (same operands, same order), on lines the records' loads have just brought in.
      if (take) {
        const bool xLow = x < y;
        const uint32_t lo = xLow ? x : y, hi = xLow ? y : x;
        const float4 ap = node[4u * lo], av = node[4u * lo + 1u], bp = node[4u * hi], bv = node[4u * hi + 1u];
        const float dx = bp.x - ap.x, dy = bp.y - ap.y, dz = bp.z - ap.z;
        const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
        if (av.w + bv.w - dist > 0.0f) {
          const uint32_t m = P.nbrM ? P.nbrM[rx.x + (rx.z & 0xffffu)] : (rx.w >> 28) + 1u;
          key = min(m, kQuadBins - 1u);
        }
      }
      // (bin b of the table = key kQuadBins - 1 - b: descending; the counts lie bin by bin, wavefront by wavefront: the table's order)
#pragma unroll
      for (uint32_t b = 0; b < kQuadBins; ++b) {
        const bool mine = take && key == kQuadBins - 1u - b;
        const unsigned long long mk = __ballot(mine);
        if (mine) rank = static_cast<uint32_t>(__popcll(mk & ((1ull << lane) - 1ull)));
        if (lane == 0) T.bins[b * look + static_cast<uint32_t>(wv)] = static_cast<uint32_t>(__popcll(mk));
      }
    }
    __syncthreads();
    // a pair's place: the pairs of the bins before its own, its bin's pairs of the wavefronts before its own, the lanes before it -
    // a prefix sum over the 9 x look counts, one per lane
    uint32_t total, nOv;  // pairs taken; the ones of them that overlap (they come first)
    {
      const uint32_t cells = kQuadBins * look;
      const uint32_t c = static_cast<uint32_t>(lane) < cells ? T.bins[lane] : 0u;
      uint32_t incl = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
      }
      total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
      nOv = static_cast<uint32_t>(__shfl(incl, static_cast<int>((kQuadBins - 1u) * look - 1u), 64));
      const uint32_t mine = (kQuadBins - 1u - key) * look + min(static_cast<uint32_t>(wv), look - 1u);
      const uint32_t at = static_cast<uint32_t>(__shfl(incl - c, static_cast<int>(mine), 64));
      if (take) taken[at + rank] = TakenPair{x, y, rx, ry};
    }
    __syncthreads();
    // the nodes of a wavefront's pairs that moved on and have entries left go to the sub-list their 64 places of the table are dealt
    // to (at most 128 nodes for every 64 frontier nodes looked at: frCap covers that); one atomic per wavefront and turn
    auto append = [&](bool moveX, bool moveY, uint32_t x, uint32_t y, uint32_t place) {
      const unsigned long long mx = __ballot(moveX), my = __ballot(moveY);
      const uint32_t nx = static_cast<uint32_t>(__popcll(mx)), ny = static_cast<uint32_t>(__popcll(my));
      if (nx + ny) {
        const uint32_t sub = (chunk * look + (place >> 6)) % kPairLists;
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(&nextCount[sub * kPairPad], nx + ny);
        at = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(at)));
        uint32_t* dst = next + static_cast<size_t>(sub) * P.frCap;
        const uint32_t ix = at + static_cast<uint32_t>(__popcll(mx & ((1ull << lane) - 1ull)));
        const uint32_t iy = at + nx + static_cast<uint32_t>(__popcll(my & ((1ull << lane) - 1ull)));
        if (moveX && ix < P.frCap) dst[ix] = x;
        if (moveY && iy < P.frCap) dst[iy] = y;
        // a sub-list that is full: the node's remaining pairs would never be visited - the pass goes to the sequential loop
        // instead (flag 2, like turn_advance; the capacity leaves ~120 entries of margin, this is the net under it)
        if (lane == 0 && at + nx + ny > P.frCap) atomicOr(&P.ctl[kPairFlags], 2u);
      }
    };
    // ---- the visits: the workgroup's quads take the overlapping pairs, blockDim.x / 4 at a time (a workgroup of one wavefront: 16; of four: 64)
    const uint32_t quads = blockDim.x >> 2;
    for (uint32_t q0 = 0; q0 < nOv; q0 += quads) {  // (workgroup uniform)
      const uint32_t q = q0 + (threadIdx.x >> 2);
      bool moveX = false, moveY = false;
      uint32_t x = 0, y = 0;
      if (q < nOv) {
        const TakenPair t = taken[q];
        x = t.x; y = t.y;
        const bool xLow = x < y;
        const uint32_t lo = xLow ? x : y, hi = xLow ? y : x;
        // everything the pair may need is requested at once (one round trip instead of four dependent ones: the positions the
        // lists were built from, the excursions so far and the next list entries used to be fetched when they were needed)
        const float4 ap = node[4u * lo], av = node[4u * lo + 1u], bp = node[4u * hi], bv = node[4u * hi + 1u];
        const float4 a0 = node[4u * lo + 2u], b0 = node[4u * hi + 2u];
        const uint32_t ea = P.exc[lo], eb = P.exc[hi];  // (a node is in one pair of a level: nobody else touches its excursion now)
        const uint32_t cx = (t.rx.z & 0xffffu) + 1u, cy = (t.ry.z & 0xffffu) + 1u;
        const uint32_t nextX = P.nbr[t.rx.x + min(cx, t.rx.y - 1u)], nextY = P.nbr[t.ry.x + min(cy, t.ry.y - 1u)];  // (unconditional: clamped)
        QuadNode a{comp4(ap, k), comp4(av, k), ap.w, av.w}, b{comp4(bp, k), comp4(bv, k), bp.w, bv.w};
        const float d = b.p - a.p;
        const float dist = sqrtf(quad_sum3(d * d));
        if (a.r + b.r - dist > 0.0f) {
          const uint32_t m = P.nbrM ? P.nbrM[t.rx.x + (t.rx.z & 0xffffu)] : (t.rx.w >> 28) + 1u;  // (wide ranges keep the count beside the entry)
          uint32_t h = 0;
          for (uint32_t v = 0; v < m; ++v) h += visit_quad(a, b, k, friction, staticThreshold) ? 1u : 0u;
          for (uint32_t v = 0; v < m; ++v) h += visit_quad(b, a, k, friction, staticThreshold) ? 1u : 0u;
          // the quad's first lane puts the components together and stores the two nodes
          const float apy = quad_lane(a.p, 1), apz = quad_lane(a.p, 2), avy = quad_lane(a.v, 1), avz = quad_lane(a.v, 2);
          const float bpy = quad_lane(b.p, 1), bpz = quad_lane(b.p, 2), bvy = quad_lane(b.v, 1), bvz = quad_lane(b.v, 2);
          if (k == 0) {
            const NodeState na{a.p, apy, apz, a.w, a.v, avy, avz, a.r}, nb{b.p, bpy, bpz, b.w, b.v, bvy, bvz, b.r};
            store_node(node, lo, na);
            store_node(node, hi, nb);
            note_excursion_owned(P, lo, na, a0, ea);
            note_excursion_owned(P, hi, nb, b0, eb);
            hits += h;
          }
        }
        if (k == 0) {
          store_rec(node, x, make_uint4(t.rx.x, t.rx.y, cx | (stampNow << 16), (cx < t.rx.y ? nextX : 0u) | ((cx & 7u) << kPairTagShift)));
          store_rec(node, y, make_uint4(t.ry.x, t.ry.y, cy | (stampNow << 16), (cy < t.ry.y ? nextY : 0u) | ((cy & 7u) << kPairTagShift)));
          moveX = cx < t.rx.y;
          moveY = cy < t.ry.y;
        }
      }
      const uint32_t first = q0 + static_cast<uint32_t>(wv) * 16u;  // the wavefront's sixteen places of the table
      if (first < nOv) append(moveX, moveY, x, y, first);  // (wavefront uniform)
    }
    // ---- the pairs that do not overlap only move on: a lane each, the table's 64-places at a time (the first of them may begin
    //      with overlapping pairs: the quads had those)
    for (uint32_t base = (nOv & ~63u) + static_cast<uint32_t>(wv) * 64u; base < total; base += blockDim.x) {  // (wavefront uniform)
      const uint32_t p = base + static_cast<uint32_t>(lane);
      bool moveX = false, moveY = false;
      uint32_t x = 0, y = 0;
      if (p >= nOv && p < total) {
        const TakenPair t = taken[p];
        x = t.x; y = t.y;
        const uint32_t cx = (t.rx.z & 0xffffu) + 1u, cy = (t.ry.z & 0xffffu) + 1u;
        const uint32_t nextX = P.nbr[t.rx.x + min(cx, t.rx.y - 1u)], nextY = P.nbr[t.ry.x + min(cy, t.ry.y - 1u)];  // (unconditional: clamped)
        store_rec(node, x, make_uint4(t.rx.x, t.rx.y, cx | (stampNow << 16), (cx < t.rx.y ? nextX : 0u) | ((cx & 7u) << kPairTagShift)));
        store_rec(node, y, make_uint4(t.ry.x, t.ry.y, cy | (stampNow << 16), (cy < t.ry.y ? nextY : 0u) | ((cy & 7u) << kPairTagShift)));
        moveX = cx < t.rx.y;
        moveY = cy < t.ry.y;
      }
      append(moveX, moveY, x, y, base);
    }
    __syncthreads();  // (the table is reused by the next chunk)
  }
}

__global__ void __launch_bounds__(kRoundBlock) k_pair_round4(HashArrays H, PairArrays P, float friction, float staticThreshold, uint32_t round, uint32_t repeat,
                                                             uint32_t look) {
  __shared__ QuadTable table;
  if (repeat && !P.ctl[kPairRetry]) return;
  if (H.counters[kCounterFlags]) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const FrontierView view = frontier_view(P, round, lane);
  if (blockIdx.x == 0 && wv == 0) {
    P.frCount[(((round + 2u) % 3u) * kPairLists + static_cast<uint32_t>(lane)) * kPairPad] = 0;  // the lists of the round after the next
    if (lane == 0 && view.total) { P.ctl[kPairRounds] = round; if (!repeat && round > P.ctl[kPairDeepest]) P.ctl[kPairDeepest] = round; }
  }
  if (view.total == 0u) return;
  uint32_t hits = 0;
  // as many looking wavefronts as it takes to give every workgroup of the launch one turn (a small frontier in chunks of 256 would
  // leave most compute units idle: the levels 26-36 of a settled pass of config 4 took 20 us instead of 14)
  const uint32_t need = (view.total + kQuadNodes * gridDim.x - 1u) / (kQuadNodes * gridDim.x);
  pair_level4(H, P, friction, staticThreshold, round, view, table, min(look, max(need, 1u)), hits);
  count_hits(P, hits, lane);
}

// ---- the levels of a REPEATED pass in one launch ----------------------------------------------------------------------------------
// A pass is repeated when a node left its slack and an unlisted pair may have touched: twice in the first three ticks of BASELINE
// config 4, never afterwards.  Until round 5 every pass captured its repeat's level launches all the same - half as many again as
// the first attempt's, 72 per settled pass of config 4, each returning on the "no repeat" word for the price of its dispatch (0.7 ms
// of a 9-ms tick).  Now the repeat's levels are ONE launch: its workgroups - all resident - run level after level with a grid
// barrier where the captured launches have a kernel boundary (release, counter, bounded wait, acquire: the CG continuation's
// barrier, pd_cg_device.h), slower per level than a launch and run twice in a simulation's life.  A wait that times out hands the
// pass to the sequential loop (flag 2).
PIES_DEV bool pair_grid_barrier(uint32_t* counter, uint32_t nblocks, uint32_t& passed) {
  __shared__ uint32_t sOk;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(counter, 1u);
    const uint32_t target = (passed + 1u) * nblocks;
    uint32_t spins = 0, ok = 1u;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 14)) { ok = 0u; break; }  // ~16 ms: a starved barrier (not all workgroups resident) gives the pass to the sequential loop quickly
    }
    // (a workgroup that gives up says so; one that arrives late and finds the counter past its target checks the word)
    if (!ok) __hip_atomic_store(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (__hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = 0u;
    __threadfence();
    sOk = ok;
  }
  __syncthreads();
  ++passed;
  return sOk != 0u;
}
// (repeat = 0: the same launch as the TAIL of the first attempt - whatever levels are left behind the captured launches, from
// `firstRound` on.  Until round 5 that was one workgroup (k_pair_tail): a pass deeper than the captured count - the first ticks of a
// scene, a pile that forms between two looks of the host - ran its surplus levels on one compute unit, ten times slower than the
// launches (config 4 pinned at 48 captured levels: 8 substeps/s instead of 100).)
__global__ void __launch_bounds__(kRoundBlock) k_pair_repeat(HashArrays H, PairArrays P, float friction, float staticThreshold, uint32_t firstRound,
                                                             uint32_t repeat) {
  __shared__ QuadTable table;
  if (repeat && !P.ctl[kPairRetry]) return;
  if (!repeat && P.ctl[kPairRetry]) return;
  if (H.counters[kCounterFlags]) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t hits = 0, passed = 0;
  for (uint32_t round = firstRound;; ++round) {
    const FrontierView view = frontier_view(P, round, lane);
    if (view.total == 0u) break;  // (the same words in every workgroup: all leave together)
    if (blockIdx.x == 0 && wv == 0) {
      __hip_atomic_store(&P.frCount[(((round + 2u) % 3u) * kPairLists + static_cast<uint32_t>(lane)) * kPairPad], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0) { P.ctl[kPairRounds] = round; if (!repeat && round > P.ctl[kPairDeepest]) P.ctl[kPairDeepest] = round; }
    }
    const uint32_t need = (view.total + kQuadNodes * gridDim.x - 1u) / (kQuadNodes * gridDim.x);  // (looking wavefronts: as in k_pair_round4)
    pair_level4(H, P, friction, staticThreshold, round, view, table, min(kQuadLookMax, max(need, 1u)), hits);
    if (!pair_grid_barrier(P.ctl + kPairBarrier, gridDim.x, passed)) {
      if (threadIdx.x == 0) atomicOr(&P.ctl[kPairFlags], 2u);
      break;
    }
  }
  count_hits(P, hits, lane);
}
// workgroups of k_pair_repeat the device holds at once (its grid barrier needs all of them resident); half of that, so that a second
// solver on the card leaves room
uint32_t pair_repeat_blocks(int device) {
  int perCu = 0, dev = device;
  hipDeviceProp_t prop;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_pair_repeat, kRoundBlock, 0) != hipSuccess) return 0;
  return static_cast<uint32_t>(std::max(0, perCu)) * static_cast<uint32_t>(std::max(0, prop.multiProcessorCount)) / 2u;
}

uint32_t turn_finish_blocks(int device);
// The grid-barrier kernels' resident workgroup counts, per DEVICE (a process may hold handles on several): looked up for
// the device that is current at the launch (the solver's), computed once per device under a mutex.
static uint32_t resident_blocks_of_current_device(int which, uint32_t (*count)(int)) {
  static std::mutex mu;
  static std::map<std::pair<int, int>, uint32_t> cache;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(which, dev);
  const auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const uint32_t v = count(dev);
  cache[key] = v;
  return v;
}

// Whatever levels are left after the captured rounds (and all levels of a repeated pass): one workgroup, a workgroup barrier
// where the rounds have a kernel boundary.  Global memory written before the barrier is visible to the workgroup after it.
__global__ void __launch_bounds__(1024) k_pair_tail(HashArrays H, PairArrays P, float friction, float staticThreshold, uint32_t round, uint32_t repeat) {
  if (repeat && !P.ctl[kPairRetry]) return;
  if (!repeat && P.ctl[kPairRetry]) return;
  if (H.counters[kCounterFlags]) return;
  const int lane = threadIdx.x & 63;
  uint32_t hits = 0;
  for (;; ++round) {
    const FrontierView view = frontier_view(P, round, lane);
    if (view.total == 0u) break;  // (the same words for every wavefront: all leave together)
    __syncthreads();  // every wavefront has read the counts before the lists after the next are cleared
    if (threadIdx.x < kPairLists) __hip_atomic_store(&P.frCount[(((round + 2u) % 3u) * kPairLists + threadIdx.x) * kPairPad], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) { P.ctl[kPairRounds] = round; if (!repeat && round > P.ctl[kPairDeepest]) P.ctl[kPairDeepest] = round; }
    process_frontier(H, P, friction, staticThreshold, round, threadIdx.x, blockDim.x, view, lane, hits);
    __threadfence();
    __syncthreads();
  }
  count_hits(P, hits, lane);
}

// ---- after the pass: can an unlisted pair have touched? -----------------------------------------------------------------
// One wavefront per node that left its slack: the nodes it shares a cell with but did not list (d0 >= cut) are tested with the
// largest excursions of the pass.
__global__ void __launch_bounds__(kBlock) k_pair_verify(HashArrays H, PairArrays P, uint32_t repeat, float scale) {
  if (repeat && !P.ctl[kPairRetry]) return;
  if (H.counters[kCounterFlags]) return;
  const int lane = threadIdx.x & 63;
  const GridBox B = grid_box(H.counters);
  const uint32_t* __restrict__ val = H.val[grid_passes(B) & 1u];
  const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = (gridDim.x * kBlock) >> 6;
  if (P.byIndex) {
    // By turns: the nodes that strayed far (see kTurnMaxExcursion) may have walked buckets of cells their inserted range does not
    // hold and met nodes no list knows.  One wavefront per such node looks at every node inserted into a cell its live range can
    // have reached - the range of a sphere of radius r + excursion around its build-time position - that shares NO inserted cell
    // with it, and tests with the largest excursions of the pass whether the two can have touched.  If so the pass is the
    // sequential loop's (a repeat would list the same pairs).
    bool lost = false;
    for (uint32_t base = wave * 64u; base < P.n; base += nwaves * 64u) {  // (wavefront uniform)
      const uint32_t mine = base + static_cast<uint32_t>(lane);
      const float em = mine < P.n ? __uint_as_float(P.exc[mine]) : 0.0f;
      unsigned long long far = __ballot(!(em < kTurnMaxExcursion));
      while (far) {
        const uint32_t q = static_cast<uint32_t>(__builtin_ctzll(far));
        far &= far - 1ull;
        const uint32_t i = base + q;
        const float4 pi = P.node[4u * i + 2u];
        const float ri = P.node[4u * i + 1u].w, ei = lane_value(em, q);
        const int4 rgi = H.rng[i];
        int mx, my, mz;
        uint32_t lx, ly, lz;
        if (!node_range(pi.x, pi.y, pi.z, ri + ei, scale, mx, my, mz, lx, ly, lz) || lx * ly * lz == 0u) { lost = true; continue; }
        for (uint32_t c = 0; c < lx * ly * lz; ++c) {
          const uint32_t cs = find_bucket(H, B, mx + static_cast<int>(c / (lz * ly)), my + static_cast<int>((c / lz) % ly), mz + static_cast<int>(c % lz));
          if (cs == 0xffffffffu) continue;
          const uint32_t bs = H.start[cs], bc = H.end[cs] - bs;
          for (uint32_t b0 = 0; b0 < bc; b0 += 64u) {
            if (b0 + lane >= bc) continue;
            const uint32_t j = val[bs + b0 + lane] & kNodeMask;
            if (j == i) continue;
            const int4 rgj = H.rng[j];
            if (shared_cells(rgi.x, rgi.w & 0xff, rgj.x, rgj.w & 0xff) * shared_cells(rgi.y, (rgi.w >> 8) & 0xff, rgj.y, (rgj.w >> 8) & 0xff) *
                    shared_cells(rgi.z, (rgi.w >> 16) & 0xff, rgj.z, (rgj.w >> 16) & 0xff) != 0u)
              continue;  // (they share an inserted cell: the lists and the test below cover the pair)
            const float4 pj = P.node[4u * j + 2u];
            const float ddx = pj.x - pi.x, ddy = pj.y - pi.y, ddz = pj.z - pi.z;
            const float reach = 1.001f * (ri + P.node[4u * j + 1u].w + ei + __uint_as_float(P.exc[j]));
            if (!(ddx * ddx + ddy * ddy + ddz * ddz >= reach * reach)) lost = true;
          }
        }
      }
    }
    if (__ballot(lost) && lane == 0) atomicOr(&P.ctl[kPairFlags], 2u);
  }
  const uint32_t count = min(P.ctl[kPairLeft], P.n);
  if (count == 0u) return;
  bool bad = false;
  for (uint32_t u = wave; u < count; u += nwaves) {
    const uint32_t i = P.left[u];
    const float4 pi = P.node[4u * i + 2u];
    const float ri = P.node[4u * i + 1u].w, ei = __uint_as_float(P.exc[i]);
    const int4 rg = H.rng[i];
    const uint32_t lx = rg.w & 0xff, ly = (rg.w >> 8) & 0xff, lz = (rg.w >> 16) & 0xff;
    for (uint32_t dx = 0; dx < lx; ++dx)
      for (uint32_t dy = 0; dy < ly; ++dy)
        for (uint32_t dz = 0; dz < lz; ++dz) {
          const uint32_t cs = find_bucket(H, B, rg.x + static_cast<int>(dx), rg.y + static_cast<int>(dy), rg.z + static_cast<int>(dz));
          if (cs == 0xffffffffu) continue;
          const uint32_t bs = H.start[cs], bc = H.end[cs] - bs;
          for (uint32_t base = 0; base < bc; base += 64) {
            if (base + lane >= bc) continue;
            const uint32_t j = val[bs + base + lane] & kNodeMask;
            if (j == i) continue;
            const float4 pj = P.node[4u * j + 2u];
            const float rj = P.node[4u * j + 1u].w;
            const float ddx = pj.x - pi.x, ddy = pj.y - pi.y, ddz = pj.z - pi.z;
            const float d2 = ddx * ddx + ddy * ddy + ddz * ddz;
            const float cut = 1.001f * ((ri + pi.w) + (rj + pj.w));
            if (!(d2 >= cut * cut)) continue;  // listed: it was visited
            const float reach = 1.001f * (ri + rj + ei + __uint_as_float(P.exc[j]));
            if (!(d2 >= reach * reach)) bad = true;  // the two may have touched while the pair was skipped
          }
        }
  }
  if (__ballot(bad) && lane == 0) atomicOr(&P.ctl[kPairFlags], 1u);
}

// first = after the first attempt.  A failed verification puts the saved state back and arms the repeat, in which the nodes
// that left their slack get the room they took in the first attempt and more.  Otherwise (and after the repeat) the result goes
// back to the solver's node arrays and every node's slack follows what it did.
__global__ void __launch_bounds__(kBlock) k_pair_check(HashArrays H, PairArrays P, float4* pos, float4* vel, uint32_t first) {
  const uint32_t flags = P.ctl[kPairFlags];
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const bool overflow = (flags & 2u) != 0u;  // the lists are incomplete: the pass is the sequential loop's, nothing of it is kept
  const bool repeat = first && (flags & 1u) && !overflow;
  if (!first && !P.ctl[kPairRetry]) return;  // nothing was repeated: the first check has done everything
  if (i < P.n && !H.counters[kCounterFlags]) {
    const float4 p0 = P.node[4u * i + 2u];
    const float e = __uint_as_float(P.exc[i]), sl = p0.w;
    if (repeat) {
      const float4 v0 = P.vel0[i];
      const float r = P.node[4u * i + 1u].w;
      P.node[4u * i] = make_float4(p0.x, p0.y, p0.z, P.node[4u * i].w);
      P.node[4u * i + 1u] = make_float4(v0.x, v0.y, v0.z, r);
      store_rec(P.node, i, make_uint4(0u, 0u, 0u, 0u));
      P.exc[i] = 0u;
      if (!(e <= 0.999f * sl)) {  // (the repeat follows the first attempt's course until a newly listed pair touches)
        const float room = 2.0f * e + 0.2f * r;
        P.node[4u * i + 2u].w = room;
        P.bq[i].w = r + room;
      }
    } else {
      const float4 p = P.node[4u * i], v = P.node[4u * i + 1u];
      // (by turns, a pass that could not be proved exact in its repeat either is left to the sequential loop: pos / vel keep the
      // state the pass started from)
      if (!(P.byIndex && !first && (flags & 1u)) && !overflow) {
        pos[i] = p;
        vel[i] = make_float4(v.x, v.y, v.z, 0.0f);  // (the fourth component of a velocity record is 0 everywhere)
      }
      P.node[4u * i + 2u].w = next_slack(e, sl, v.w);
    }
  }
  // the pass's resolved pairs go to the statistics when its result stands (a pass that is repeated counts once; one that is left to
  // the sequential loop is counted by that loop)
  if (blockIdx.x == 0 && !repeat && !(P.byIndex && !first && (flags & 1u)) && !overflow) {
    uint32_t sum = 0;
    for (uint32_t k = threadIdx.x; k < kPairStripes; k += kBlock) { sum += P.hitStripe[k * kPairPad]; P.hitStripe[k * kPairPad] = 0; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&H.counters[kCounterPairs], sum);
    if (threadIdx.x < 64) {  // candidates looked at (64 bit) and listed entries
      unsigned long long c = *reinterpret_cast<unsigned long long*>(&P.stat[threadIdx.x * kPairPad]);
      uint32_t e = P.stat[threadIdx.x * kPairPad + 2u];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { c += __shfl_xor(c, o, 64); e += __shfl_xor(e, o, 64); }
      if (threadIdx.x == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(&H.counters[kCounterCandidates]), c);
        P.ctl[kPairEdges] = e;
      }
    }
  } else if (blockIdx.x == 0) {
    for (uint32_t k = threadIdx.x; k < kPairStripes; k += kBlock) P.hitStripe[k * kPairPad] = 0;
    if (threadIdx.x < 64) { P.stat[threadIdx.x * kPairPad] = 0; P.stat[threadIdx.x * kPairPad + 1u] = 0; P.stat[threadIdx.x * kPairPad + 2u] = 0; }
  }
  if (i != 0 || first) return;
  if (flags & 1u) P.ctl[kPairInexact] += 1;
  if ((P.byIndex && (flags & 1u)) || (flags & 2u)) { P.ctl[kPairFallback] = 1u; P.ctl[kPairFallbacks] += 1u; }  // (list storage overflow in the repeat)
  P.ctl[kPairLeft] = 0;
}
// (one thread, after every block of the first k_pair_check has read the flags)
__global__ void k_pair_arm(HashArrays H, PairArrays P) {
  if (P.ctl[kPairFlags] & 1u) {
    if (threadIdx.x < kPairPools) P.pool[threadIdx.x * kPairPad] = 0;
    for (uint32_t k = threadIdx.x; k < 3u * kPairLists; k += 64) P.frCount[k * kPairPad] = 0;
  }
  if (threadIdx.x != 0) return;
  const uint32_t flags = P.ctl[kPairFlags];
  P.ctl[kPairLeft] = 0;
  if (flags & 2u) {  // a pile the lists do not hold (more than 1 024 partners of a node, more entries than reserved): no repeat - the
    P.ctl[kPairFallback] = 1u;  // sequential loop runs the pass from the state it started with (the reference has no such limit)
    P.ctl[kPairFallbacks] += 1u;
    return;
  }
  if (!(flags & 1u)) return;
  P.ctl[kPairRetry] = 1;
  P.ctl[kPairBarrier] = 0;      // k_pair_repeat's grid barrier: counter, abort word
  P.ctl[kPairBarrier + 1] = 0;
  P.ctl[kPairRetries] += 1;
  P.ctl[kPairFlags] = 0;
  P.ctl[kPairEdges] = 0;
  P.ctl[kPairSpilled] = 0;
}

// ======================================================================================================================
// The REFERENCE's order (Solver.cpp:85-130) by dependency levels of TURNS
// ======================================================================================================================
// The reference's loop gives every node its turn in ascending index: the node's cell range is looked up from its LIVE position
// (SpatialHash.h:101-106), the buckets of that range are walked in dx, dy, dz order and their nodes in ascending index, and every
// overlapping visit is resolved at once.  k_collide_reference (hash_kernels.hip) runs that as one chain on one wavefront: 20 us
// per node.  Here the same turns run by dependency levels:
//   * who a turn can touch: the node's partners within reach when the grid was built (the pair order's filtered lists, sorted by
//     INDEX here; a visit to anybody else is a miss as long as every node stays within its slack - the pair order's proof obligation,
//     checked by the same k_pair_verify, repeated with wider slacks, and left to the sequential kernel when that fails too);
//   * a node j lives through the turns of its partners below it, its own turn, the turns of its partners above it - in that order.
//     Turn i may run when every member of it (i and its partners) has had all its earlier events.  Two turns that are ready at the
//     same time share no member, so a level is one launch, one wavefront per turn;
//   * inside a turn the wavefront holds the partners in its lanes (ascending index).  For every cell of the live range, in order:
//     the lanes whose partner was INSERTED into that cell test the overlap against the node's current state, the lowest hit is
//     resolved (Solver.cpp:88-126, the same visit() as the pair order's), the node's new state goes to every lane, the lanes above
//     test again; the node meets itself at its place in the bucket (quirk Q3).  The visits that are skipped are the ones that miss.
// The result is the sequential loop's, bit for bit: tests/test_collisions_gpu.py runs both against the oracle's plain loop.
constexpr uint32_t kTurnDone = 0xffffffffu;
constexpr uint32_t kTurnBlock = 256;

// the node whose turn event t of node j is: partners below j, j itself, partners above j
PIES_DEV uint32_t turn_event_node(const PairArrays& P, uint32_t j, uint32_t off, uint32_t d, uint32_t below, uint32_t t) {
  if (t > d) return kTurnDone;
  if (t == below) return j;
  return P.nbr[off + (t < below ? t : t - 1u)] & kPairNodeMask;
}
PIES_DEV bool cell_in_range(const int4 rg, int cx, int cy, int cz) {
  const int lx = rg.w & 0xff, ly = (rg.w >> 8) & 0xff, lz = (rg.w >> 16) & 0xff;
  return cx >= rg.x && cx < rg.x + lx && cy >= rg.y && cy < rg.y + ly && cz >= rg.z && cz < rg.z + lz;
}

// One resolved visit of a turn (Solver.cpp:92-125) with its fifteen divisions dealt to the lanes: `a` (wave uniform) visits the node
// held by lane `src` (self: itself).  visit() runs all of it in one lane - ~350 instructions, a turn of config 4 resolves ~20 visits
// one after the other: 15 us of a 45-us level.  Here lane t computes ONE quotient (the three components of the direction on lanes
// 0-2, then the twelve corrections ((coef * vec_k) * mass) / wSum on lanes 0-11) and the quotients are broadcast: the same operations
// on the same operands in the same order as visit() / visit_self(), so the same bits (k_collide_reference's resolve_pair does the
// same).  The overlap is known (the caller's test): the visit resolves.
PIES_DEV float pick3(int k, float x, float y, float z) { return k == 0 ? x : (k == 1 ? y : z); }
PIES_DEV void visit_wide(NodeState& a, NodeState& b, uint32_t src, bool self, float friction, float staticThreshold, int lane) {
  NodeState o = a;
  if (!self) o = NodeState{lane_value(b.px, src), lane_value(b.py, src), lane_value(b.pz, src), lane_value(b.w, src),
                           lane_value(b.vx, src), lane_value(b.vy, src), lane_value(b.vz, src), lane_value(b.r, src)};
  const float dx = o.px - a.px, dy = o.py - a.py, dz = o.pz - a.pz;
  const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
  const float disp = a.r + o.r - dist;
  const int k3 = lane % 3, kind = (lane / 3) & 3;
  float ux = 1.0f, uy = 0.0f, uz = 0.0f;
  if (dist > 0.00001f) {
    const float quot = pick3(k3, dx, dy, dz) / dist;
    ux = lane_value(quot, 0); uy = lane_value(quot, 1); uz = lane_value(quot, 2);
  }
  const float wSum = a.w + o.w;
  const float sa = 0.85f * -disp, sb = 0.85f * disp;
  const float rx = o.vx - a.vx, ry = o.vy - a.vy, rz = o.vz - a.vz;
  const float rd = rx * ux + ry * uy + rz * uz;
  const float qx = rx - rd * ux, qy = ry - rd * uy, qz = rz - rd * uz;
  float fr = friction;
  if (staticThreshold > 0.0f)  // sqrt(x) < t is false for every t <= 0
    if (sqrtf(qx * qx + qy * qy + qz * qz) < staticThreshold) fr = 1.0f;
  const float vec = kind < 2 ? pick3(k3, ux, uy, uz) : pick3(k3, qx, qy, qz);
  const float coef = kind == 0 ? sa : (kind == 1 ? sb : (kind == 2 ? -fr : fr));
  const float mass = (kind & 1) ? o.w : a.w;
  const float corr = ((coef * vec) * mass) / wSum;
  if (self) {  // (visit_self: the second update of each line sees the first)
    a.px += lane_value(corr, 0); a.py += lane_value(corr, 1); a.pz += lane_value(corr, 2);
    a.px += lane_value(corr, 3); a.py += lane_value(corr, 4); a.pz += lane_value(corr, 5);
    a.vx += lane_value(corr, 6); a.vy += lane_value(corr, 7); a.vz += lane_value(corr, 8);
    a.vx += lane_value(corr, 9); a.vy += lane_value(corr, 10); a.vz += lane_value(corr, 11);
    return;
  }
  a.px += lane_value(corr, 0); a.py += lane_value(corr, 1); a.pz += lane_value(corr, 2);
  o.px += lane_value(corr, 3); o.py += lane_value(corr, 4); o.pz += lane_value(corr, 5);
  a.vx += lane_value(corr, 6); a.vy += lane_value(corr, 7); a.vz += lane_value(corr, 8);
  o.vx += lane_value(corr, 9); o.vy += lane_value(corr, 10); o.vz += lane_value(corr, 11);
  if (lane == static_cast<int>(src)) b = o;
}

// One bucket of a turn: the lanes with inCell hold the bucket's partners of node a (ascending index over the lanes), selfAt = the
// lane before which the node meets itself (64: behind the last lane; kTurnDone: not in this bucket / not in this batch).
PIES_DEV void turn_cell(NodeState& a, NodeState& b, bool inCell, uint32_t selfAt, int lane, float friction, float staticThreshold, uint32_t& hits,
                        bool& aMoved, bool& bMoved) {
  unsigned long long pending = __ballot(inCell);
  for (;;) {
    bool hit = false;
    if ((pending >> lane) & 1ull) {  // the overlap test of visit(), on the live states
      const float dx = b.px - a.px, dy = b.py - a.py, dz = b.pz - a.pz;
      const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
      hit = a.r + b.r - dist > 0.0f;
    }
    const unsigned long long hm = __ballot(hit);
    const uint32_t first = hm ? static_cast<uint32_t>(__builtin_ctzll(hm)) : 64u;
    if (selfAt != kTurnDone && selfAt <= first) {  // everything below the node's own place has missed: it meets itself
      if (a.r + a.r > 0.0f) {  // (visit_self's test: the distance to itself is 0)
        visit_wide(a, b, 0u, true, friction, staticThreshold, lane);
        ++hits;
        aMoved = true;
      }
      pending = selfAt >= 64u ? 0ull : pending & ~((1ull << selfAt) - 1ull);
      selfAt = kTurnDone;
      continue;  // (its state may have changed: the lanes above test again)
    }
    if (first >= 64u) break;
    visit_wide(a, b, first, false, friction, staticThreshold, lane);
    if (lane == static_cast<int>(first)) bMoved = true;
    ++hits;
    aMoved = true;
    pending = first >= 63u ? 0ull : pending & ~((2ull << first) - 1ull);
  }
}

// the members of a finished turn move on: `j` (lane-held, valid where have) to its next event; a node whose turn has all its
// members waiting for it goes to the next frontier
PIES_DEV void turn_advance(const PairArrays& P, bool have, uint32_t j, uint32_t* __restrict__ next, uint32_t* __restrict__ nextCount, uint32_t sub, int lane) {
  uint32_t ready = kTurnDone;
  if (have) {
    const uint4 r = load_rec(P.node, j);
    const uint32_t d = r.y & 0xffffu, below = r.y >> 16, t = r.z + 1u;
    const uint32_t nxt = turn_event_node(P, j, r.x, d, below, t);
    store_rec(P.node, j, make_uint4(r.x, r.y, t, nxt));
    if (nxt != kTurnDone && atomicSub(&P.turnCnt[nxt], 1u) == 1u) ready = nxt;
  }
  const unsigned long long rm = __ballot(ready != kTurnDone);
  if (rm) {
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(&nextCount[sub * kPairPad], static_cast<uint32_t>(__popcll(rm)));
    at = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(at))) + static_cast<uint32_t>(__popcll(rm & ((1ull << lane) - 1ull)));
    if (ready != kTurnDone) {
      if (at < P.frCap) next[static_cast<size_t>(sub) * P.frCap + at] = ready;
      else atomicOr(&P.ctl[kPairFlags], 2u);  // (frontier storage: the host latches the failure)
    }
  }
}

// the turn of node i, by one wavefront; returns the visits it resolved
PIES_DEV uint32_t run_turn(const HashArrays& H, const PairArrays& P, uint32_t i, float scale, float friction, float staticThreshold, uint32_t* next,
                           uint32_t* nextCount, uint32_t sub, int lane) {
  float4* node = P.node;
  const uint4 ri = load_rec(node, i);
  const uint32_t off = ri.x, d = ri.y & 0xffffu, below = ri.y >> 16;
  NodeState a = load_node(node, i);
  const int4 rgi = H.rng[i];
  int mx, my, mz;
  uint32_t lx, ly, lz;
  if (!node_range(a.px, a.py, a.pz, a.r, scale, mx, my, mz, lx, ly, lz)) {  // (the sequential loop stops there, Solver.cpp's would not: latched)
    if (lane == 0) atomicOr(&H.counters[kCounterFlags], 1u);
    lx = ly = lz = 0;
  }
  const uint32_t ncell = lx * ly * lz;
  // (a node that was inserted with an empty - over-long - range has no list; should its live range hold cells, the lists cannot serve)
  if (ncell != 0u && (rgi.w & 0xffffff) == 0 && lane == 0) atomicOr(&P.ctl[kPairFlags], 1u);
  uint32_t hits = 0;
  bool aMoved = false;
  if (d < 64u) {  // the partners in the lanes' registers for the whole turn (lane d: the node itself, for the bookkeeping behind the turn)
    const bool have = static_cast<uint32_t>(lane) < d;
    const uint32_t j = have ? P.nbr[off + lane] & kPairNodeMask : 0u;
    NodeState b = have ? load_node(node, j) : NodeState{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int4 rgj = have ? H.rng[j] : make_int4(0, 0, 0, 0);
    // (the build-time positions and the excursions so far of the members, for the bookkeeping behind the turn: requested now -
    // behind the turn a load and a returning atomic were two more dependent round trips; a member belongs to this turn alone
    // for the whole level, so its excursion is read and written plainly)
    const uint32_t mj0 = have ? j : i;
    const float4 p0m = static_cast<uint32_t>(lane) <= d ? node[4u * mj0 + 2u] : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t excm = static_cast<uint32_t>(lane) <= d ? P.exc[mj0] : 0u;
    // what the members need when the turn is over - their records (the same cache line as their state) and the nodes whose turns
    // their NEXT events are - is requested now, beside the states: behind the turn these would be two more dependent round trips
    const bool member = static_cast<uint32_t>(lane) <= d;
    const uint32_t mj = have ? j : i;
    const uint4 mr = member ? (have ? load_rec(node, j) : ri) : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t mnext = member ? turn_event_node(P, mj, mr.x, mr.y & 0xffffu, mr.y >> 16, mr.z + 1u) : kTurnDone;
    bool bMoved = false;
    for (uint32_t c = 0; c < ncell; ++c) {  // dz fastest (SpatialHash.h:108-125)
      const int cx = mx + static_cast<int>(c / (lz * ly)), cy = my + static_cast<int>((c / lz) % ly), cz = mz + static_cast<int>(c % lz);
      turn_cell(a, b, have && cell_in_range(rgj, cx, cy, cz), cell_in_range(rgi, cx, cy, cz) ? below : kTurnDone, lane, friction, staticThreshold,
                hits, aMoved, bMoved);
    }
    if (bMoved) {
      store_node(node, j, b);
      note_excursion_owned(P, j, b, p0m, excm);
    }
    if (aMoved && static_cast<uint32_t>(lane) == d) {  // (lane d holds the node's own build-time position and excursion)
      store_node(node, i, a);
      note_excursion_owned(P, i, a, p0m, excm);
    }
    // every member moves on to its next event; a node whose turn has all its members waiting for it goes to the next frontier
    uint32_t ready = kTurnDone;
    if (member) {
      store_rec(node, mj, make_uint4(mr.x, mr.y, mr.z + 1u, mnext));
      if (mnext != kTurnDone && atomicSub(&P.turnCnt[mnext], 1u) == 1u) ready = mnext;
    }
    const unsigned long long rm = __ballot(ready != kTurnDone);
    if (rm) {
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(&nextCount[sub * kPairPad], static_cast<uint32_t>(__popcll(rm)));
      at = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(at))) + static_cast<uint32_t>(__popcll(rm & ((1ull << lane) - 1ull)));
      if (ready != kTurnDone) {
        if (at < P.frCap) next[static_cast<size_t>(sub) * P.frCap + at] = ready;
        else atomicOr(&P.ctl[kPairFlags], 2u);
      }
    }
    return hits;
  } else {  // a dense neighbourhood: 64 partners at a time, their states through memory (a partner sits in one batch)
    for (uint32_t c = 0; c < ncell; ++c) {
      const int cx = mx + static_cast<int>(c / (lz * ly)), cy = my + static_cast<int>((c / lz) % ly), cz = mz + static_cast<int>(c % lz);
      const bool selfIn = cell_in_range(rgi, cx, cy, cz);
      for (uint32_t base = 0; base < d; base += 64u) {
        const bool have = base + static_cast<uint32_t>(lane) < d;
        const uint32_t j = have ? P.nbr[off + base + lane] & kPairNodeMask : 0u;
        NodeState b = have ? load_node(node, j) : NodeState{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int4 rgj = have ? H.rng[j] : make_int4(0, 0, 0, 0);
        bool bMoved = false;
        // the node's own place: before lane below - base of the batch that holds it, behind the last batch when every partner is below
        uint32_t selfAt = kTurnDone;
        if (selfIn && below >= base && (below < base + 64u || (below == d && base + 64u >= d))) selfAt = below - base;
        turn_cell(a, b, have && cell_in_range(rgj, cx, cy, cz), selfAt, lane, friction, staticThreshold, hits, aMoved, bMoved);
        if (bMoved) {
          store_node(node, j, b);
          note_excursion(P, j, b, node[4u * j + 2u]);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (aMoved && lane == 0) {
      store_node(node, i, a);
      note_excursion(P, i, a, node[4u * i + 2u]);
    }
    for (uint32_t base = 0; base < d; base += 64u) {
      const bool have = base + static_cast<uint32_t>(lane) < d;
      turn_advance(P, have, have ? P.nbr[off + base + lane] & kPairNodeMask : 0u, next, nextCount, sub, lane);
    }
  }
  turn_advance(P, lane == 0, i, next, nextCount, sub, lane);  // the node itself: on to its first partner above it
  return hits;
}

// the first events: every node tells the node whose turn its first event is that it is waiting; turns with all members waiting
// make the first frontier (the lists of round 2: round 1 means "every node" to frontier_view)
__global__ void __launch_bounds__(kBlock) k_turn_first(HashArrays H, PairArrays P, uint32_t repeat) {
  if (repeat && !P.ctl[kPairRetry]) return;
  if (H.counters[kCounterFlags]) return;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t ready = kTurnDone;
  if (i < P.n) {
    const uint32_t f = load_rec(P.node, i).w;
    if (f != kTurnDone && atomicSub(&P.turnCnt[f], 1u) == 1u) ready = f;
  }
  const unsigned long long rm = __ballot(ready != kTurnDone);
  if (rm) {
    const uint32_t sub = ((blockIdx.x * kBlock + threadIdx.x) >> 6) % kPairLists;
    uint32_t* nextCount = P.frCount + (2u % 3u) * kPairLists * kPairPad;
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(&nextCount[sub * kPairPad], static_cast<uint32_t>(__popcll(rm)));
    at = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(at))) + static_cast<uint32_t>(__popcll(rm & ((1ull << lane) - 1ull)));
    if (ready != kTurnDone) {
      if (at < P.frCap) P.fr[0][static_cast<size_t>(sub) * P.frCap + at] = ready;
      else atomicOr(&P.ctl[kPairFlags], 2u);
    }
  }
}

// one level: the turns of the frontier of `round`, one wavefront each
__global__ void __launch_bounds__(kTurnBlock) k_turn_round(HashArrays H, PairArrays P, float scale, float friction, float staticThreshold, uint32_t round,
                                                           uint32_t repeat) {
  if (repeat && !P.ctl[kPairRetry]) return;
  if (H.counters[kCounterFlags]) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const FrontierView view = frontier_view(P, round, lane);
  if (blockIdx.x == 0 && wv == 0) {
    P.frCount[(((round + 2u) % 3u) * kPairLists + static_cast<uint32_t>(lane)) * kPairPad] = 0;  // the lists of the round after the next
    if (lane == 0 && view.total) { P.ctl[kPairRounds] = round - 1u; if (!repeat && round - 1u > P.ctl[kPairDeepest]) P.ctl[kPairDeepest] = round - 1u; }
  }
  const uint32_t count = view.total;
  if (count == 0u) return;
  uint32_t* next = P.fr[(round + 1u) & 1u];
  uint32_t* nextCount = P.frCount + ((round + 1u) % 3u) * kPairLists * kPairPad;
  const uint32_t wavesPerGrid = gridDim.x * (kTurnBlock / 64u);
  uint32_t hits = 0;
  for (uint32_t e = blockIdx.x * (kTurnBlock / 64u) + static_cast<uint32_t>(wv); e < count; e += wavesPerGrid) {  // (wavefront uniform)
    const uint32_t i = frontier_node(P, view, round, e);
    hits += run_turn(H, P, i, scale, friction, staticThreshold, next, nextCount, e % kPairLists, lane);
  }
  if (lane == 0 && hits) atomicAdd(&P.hitStripe[((blockIdx.x * (kTurnBlock / 64u) + static_cast<uint32_t>(wv)) % kPairStripes) * kPairPad], hits);
}

// whatever levels are left after the captured launches (and all levels of a repeated pass): one workgroup, a workgroup barrier
// where the levels have a kernel boundary (slow, never wrong)
__global__ void __launch_bounds__(1024) k_turn_tail(HashArrays H, PairArrays P, float scale, float friction, float staticThreshold, uint32_t round,
                                                    uint32_t repeat) {
  if (repeat && !P.ctl[kPairRetry]) return;
  if (!repeat && P.ctl[kPairRetry]) return;
  if (H.counters[kCounterFlags]) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t hits = 0;
  for (;; ++round) {
    const FrontierView view = frontier_view(P, round, lane);
    if (view.total == 0u) break;  // (the same words for every wavefront: all leave together)
    __syncthreads();              // every wavefront has read the counts before the lists after the next are cleared
    if (threadIdx.x < kPairLists) __hip_atomic_store(&P.frCount[(((round + 2u) % 3u) * kPairLists + threadIdx.x) * kPairPad], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) { P.ctl[kPairRounds] = round - 1u; if (!repeat && round - 1u > P.ctl[kPairDeepest]) P.ctl[kPairDeepest] = round - 1u; }
    uint32_t* next = P.fr[(round + 1u) & 1u];
    uint32_t* nextCount = P.frCount + ((round + 1u) % 3u) * kPairLists * kPairPad;
    for (uint32_t e = static_cast<uint32_t>(wv); e < view.total; e += blockDim.x / 64u) {
      const uint32_t i = frontier_node(P, view, round, e);
      hits += run_turn(H, P, i, scale, friction, staticThreshold, next, nextCount, e % kPairLists, lane);
    }
    __threadfence();
    __syncthreads();
  }
  if (lane == 0 && hits) atomicAdd(&P.hitStripe[(static_cast<uint32_t>(wv) % kPairStripes) * kPairPad], hits);
}

// The same with resident workgroups and a grid barrier between the levels (k_pair_repeat's): a pass deeper than the captured launches
// no longer finishes on one compute unit, and a repeated pass runs here whole.
__global__ void __launch_bounds__(kTurnBlock) k_turn_finish(HashArrays H, PairArrays P, float scale, float friction, float staticThreshold, uint32_t round,
                                                            uint32_t repeat) {
  if (repeat && !P.ctl[kPairRetry]) return;
  if (!repeat && P.ctl[kPairRetry]) return;
  if (H.counters[kCounterFlags]) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t wavesPerGrid = gridDim.x * (kTurnBlock / 64u);
  uint32_t hits = 0, passed = 0;
  for (;; ++round) {
    const FrontierView view = frontier_view(P, round, lane);
    if (view.total == 0u) break;  // (the same words in every workgroup: all leave together)
    if (blockIdx.x == 0 && wv == 0) {
      __hip_atomic_store(&P.frCount[(((round + 2u) % 3u) * kPairLists + static_cast<uint32_t>(lane)) * kPairPad], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0) { P.ctl[kPairRounds] = round - 1u; if (!repeat && round - 1u > P.ctl[kPairDeepest]) P.ctl[kPairDeepest] = round - 1u; }
    }
    uint32_t* next = P.fr[(round + 1u) & 1u];
    uint32_t* nextCount = P.frCount + ((round + 1u) % 3u) * kPairLists * kPairPad;
    for (uint32_t e = blockIdx.x * (kTurnBlock / 64u) + static_cast<uint32_t>(wv); e < view.total; e += wavesPerGrid) {  // (wavefront uniform)
      const uint32_t i = frontier_node(P, view, round, e);
      hits += run_turn(H, P, i, scale, friction, staticThreshold, next, nextCount, e % kPairLists, lane);
    }
    if (!pair_grid_barrier(P.ctl + kPairBarrier, gridDim.x, passed)) {
      if (threadIdx.x == 0) atomicOr(&P.ctl[kPairFlags], 2u);
      break;
    }
  }
  if (lane == 0 && hits) atomicAdd(&P.hitStripe[((blockIdx.x * (kTurnBlock / 64u) + static_cast<uint32_t>(wv)) % kPairStripes) * kPairPad], hits);
}
uint32_t turn_finish_blocks(int device) {
  int perCu = 0, dev = device;
  hipDeviceProp_t prop;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_turn_finish, kTurnBlock, 0) != hipSuccess) return 0;
  return static_cast<uint32_t>(std::max(0, perCu)) * static_cast<uint32_t>(std::max(0, prop.multiProcessorCount)) / 2u;
}

uint32_t launch_collide_turns(hipStream_t st, const HashArrays& H, const PairArrays& Pin, const NodeArrays& nd, float gridSpacing, float friction,
                              float staticThreshold, uint32_t rounds) {
  if (nd.n == 0) return 0;
  PairArrays P = Pin;
  P.byIndex = 1u;
  const uint32_t n = nd.n;
  uint32_t launches = 0;
  const dim3 perNode((n + kBlock - 1) / kBlock);
  const dim3 groups(std::max<uint32_t>(1u, std::min<uint32_t>(8192u, n / 8 + 1)));
  // a level of BASELINE config 4 holds a few hundred turns: one wavefront each
  const dim3 level(std::max<uint32_t>(1u, std::min<uint32_t>(1024u, (n / 64u + kTurnBlock / 64u) / (kTurnBlock / 64u))));
  hipLaunchKernelGGL(k_pair_save, perNode, dim3(kBlock), 0, st, H, P, nd.pos, nd.vel, nd.radius, friction, staticThreshold); ++launches;
  const bool wide = P.nbrM != nullptr;  // ranges of more than two cells per axis: lists node by node
  if (!wide) { hipLaunchKernelGGL(k_pair_groups, dim3(std::min<uint32_t>(2048u, (H.capacity / 8 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, H, P, 0u); ++launches; }
  for (uint32_t repeat = 0; repeat < 2; ++repeat) {
    if (repeat) { hipLaunchKernelGGL(k_pair_self, perNode, dim3(kBlock), 0, st, H, P, friction, staticThreshold); ++launches; }
    if (wide) {
      hipLaunchKernelGGL(k_pair_build_wide, dim3(std::max<uint32_t>(1u, std::min<uint32_t>(16384u, n))), dim3(64), 0, st, H, P, repeat); ++launches;
    } else {
      hipLaunchKernelGGL((k_pair_build<384, 96, 64, false>), groups, dim3(64 * kBuildWaves), 0, st, H, P, repeat); ++launches;
      hipLaunchKernelGGL((k_pair_build<kMaxCand, kMaxDeg, kMaxOwn, true>), dim3(512), dim3(64 * kBuildWaves), 0, st, H, P, repeat); ++launches;
    }
    hipLaunchKernelGGL(k_turn_first, perNode, dim3(kBlock), 0, st, H, P, repeat); ++launches;
    // (a repeated pass - rare: a node left its slack and an unlisted pair may have touched - runs all its levels in the tail kernel)
    // PIES_TURN_LEVEL_LAUNCHES=0: no captured level launches at all - every level behind the grid barrier
    uint32_t captured = repeat ? 0u : rounds;
    if (const char* e = tuning_env("PIES_TURN_LEVEL_LAUNCHES"); e && e[0] == '0') captured = 0u;
    for (uint32_t r = 2; r < 2u + captured; ++r) {
      hipLaunchKernelGGL(k_turn_round, level, dim3(kTurnBlock), 0, st, H, P, gridSpacing, friction, staticThreshold, r, repeat); ++launches;
    }
    const uint32_t residentFinish = resident_blocks_of_current_device(1, turn_finish_blocks);
    uint32_t finishCap = 128u;  // PIES_TURN_FINISH_BLOCKS: workgroups behind the grid barrier (a level of config 4 holds ~460 turns, one wavefront each; measured with every level behind the barrier: 512 workgroups 293 ms per tick, 128: 217, 64: 284 - captured launches: 160-172)
    if (const char* e = tuning_env("PIES_TURN_FINISH_BLOCKS")) { const int v = std::atoi(e); if (v >= 1 && v <= 4096) finishCap = static_cast<uint32_t>(v); }
    const uint32_t finishBlocks = std::min<uint32_t>(std::min<uint32_t>(finishCap, residentFinish), level.x);
    if (finishBlocks) { hipLaunchKernelGGL(k_turn_finish, dim3(finishBlocks), dim3(kTurnBlock), 0, st, H, P, gridSpacing, friction, staticThreshold, 2u + captured, repeat); ++launches; }
    else { hipLaunchKernelGGL(k_turn_tail, dim3(1), dim3(1024), 0, st, H, P, gridSpacing, friction, staticThreshold, 2u + captured, repeat); ++launches; }
    hipLaunchKernelGGL(k_pair_verify, dim3(64), dim3(kBlock), 0, st, H, P, repeat, gridSpacing); ++launches;
    hipLaunchKernelGGL(k_pair_check, perNode, dim3(kBlock), 0, st, H, P, nd.pos, nd.vel, repeat ? 0u : 1u); ++launches;
    if (!repeat) { hipLaunchKernelGGL(k_pair_arm, dim3(1), dim3(64), 0, st, H, P); ++launches; }
  }
  // a pass that could not be proved exact twice: the sequential loop on the state the pass started from (it returns at once otherwise)
  launches += launch_collide_reference(st, H, nd, gridSpacing, friction, staticThreshold, P.ctl + kPairFallback);
  return launches;
}

uint32_t launch_collide_pairs(hipStream_t st, const HashArrays& H, const PairArrays& P, const NodeArrays& nd, float gridSpacing, float friction,
                              float staticThreshold, uint32_t rounds) {
  if (nd.n == 0) return 0;
  const uint32_t n = nd.n;
  uint32_t launches = 0;
  const dim3 perNode((n + kBlock - 1) / kBlock);
  const dim3 groups(std::max<uint32_t>(1u, std::min<uint32_t>(8192u, n / 8 + 1)));
  // workgroups of a level launch: a settled level of config 4 has ~280 chunks of frontier nodes, and 60 % of the captured level
  // launches find nothing to do (the repeat's, and the spare ones of the first attempt) - an empty launch costs what its dispatch
  // costs.  Measured on config 4 (burst / settled substeps/s): cap 256: 55.5 / 76.3, 512: 66.7 / 91.3, 1 024: 68.2 / 97.3, 2 048 (rounds
  // 3's): 65.3 / 94.2.  PIES_PAIR_LEVEL_BLOCKS sets the cap (chunks beyond it are taken in a grid-stride loop).
  uint32_t levelCap = 1024u;
  if (const char* e = tuning_env("PIES_PAIR_LEVEL_BLOCKS")) { const int v = std::atoi(e); if (v >= 1 && v <= 65535) levelCap = static_cast<uint32_t>(v); }
  const dim3 level(std::max<uint32_t>(1u, std::min<uint32_t>(levelCap, (n + kRoundBlock - 1u) / kRoundBlock)));
  // the level launches of the REPEAT almost always find "no repeat" and return: a small grid makes that cheap (a repeat that does
  // run takes its frontier in a grid-stride loop)
  uint32_t repeatCap = levelCap;
  if (const char* e = tuning_env("PIES_PAIR_REPEAT_BLOCKS")) { const int v = std::atoi(e); if (v >= 1 && v <= 65535) repeatCap = static_cast<uint32_t>(v); }
  const dim3 levelRepeat(std::max<uint32_t>(1u, std::min<uint32_t>(repeatCap, level.x)));
  // four lanes per pair (k_pair_round4): a workgroup takes 64 frontier nodes per round of its loop; PIES_PAIR_QUADS=0: one lane per pair
  bool quads = true;
  if (const char* e = tuning_env("PIES_PAIR_QUADS")) quads = e[0] != '0';
  // threads of a level's workgroups (PIES_PAIR_QUAD_THREADS: 64, 128 or 256): a workgroup looks at 64 frontier nodes per turn of its
  // loop whatever its size, and as many workgroups as the chip holds at once take part (8 of 256 threads per compute unit).
  // Measured on config 4 (burst / settled), before the table was sorted: 256: 75.6 / 109.9, 128: 68.8 / 102.0, 64: 63.4 / 94.3 - more,
  // smaller workgroups put every chunk of a level in flight at once and were slower, and requesting all of a pair's operands at once
  // (one round trip instead of four) changed nothing (75.2 / 109.1): a level is bound by VALU issue, not by its round trips.
  uint32_t threads4 = kRoundBlock;
  if (const char* e = tuning_env("PIES_PAIR_QUAD_THREADS")) { const int v = std::atoi(e); if (v == 64 || v == 128 || v == 256) threads4 = static_cast<uint32_t>(v); }
  uint32_t cap4 = 1024u * (kRoundBlock / threads4);  // (measured on config 4 with 256 threads and the sorted table, burst / settled substeps/s: 512: 81 / 116, 768: 87 / 121, 1 024: 90 / 127, 1 536: 86 / 123, 2 048: 82 / 117)
  if (const char* e = tuning_env("PIES_PAIR_QUAD_BLOCKS")) { const int v = std::atoi(e); if (v >= 1 && v <= 65535) cap4 = static_cast<uint32_t>(v); }
  // wavefronts of a level's workgroup that look at frontier nodes, 64 each (PIES_PAIR_LOOK_WAVES: 1, 2 or 4): the pairs they take are
  // sorted by visits across the whole workgroup, so more of them make the visiting wavefronts denser and more alike
  uint32_t look4 = 4u;
  if (const char* e = tuning_env("PIES_PAIR_LOOK_WAVES")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4) look4 = static_cast<uint32_t>(v); }
  look4 = std::min(look4, threads4 / 64u);
  const dim3 level4(std::max<uint32_t>(1u, std::min<uint32_t>(cap4, (n + kQuadNodes * look4 - 1u) / (kQuadNodes * look4))));
  const dim3 levelRepeat4(std::max<uint32_t>(1u, std::min<uint32_t>(repeatCap, level4.x)));
  // the repeat's levels in one launch of resident workgroups (PIES_PAIR_REPEAT_LAUNCHES=1: captured level launches as in rounds 3-4)
  const uint32_t residentRepeat = resident_blocks_of_current_device(0, pair_repeat_blocks);
  uint32_t repeatBlocks = std::min<uint32_t>(std::min<uint32_t>(512u, residentRepeat), level4.x);
  if (const char* e = tuning_env("PIES_PAIR_REPEAT_LAUNCHES"); e && e[0] == '1') repeatBlocks = 0;
  hipLaunchKernelGGL(k_pair_save, perNode, dim3(kBlock), 0, st, H, P, nd.pos, nd.vel, nd.radius, friction, staticThreshold); ++launches;
  const bool wide = P.nbrM != nullptr;  // ranges of more than two cells per axis: lists node by node
  if (!wide) { hipLaunchKernelGGL(k_pair_groups, dim3(std::min<uint32_t>(2048u, (H.capacity / 8 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, H, P, 0u); ++launches; }
  for (uint32_t repeat = 0; repeat < 2; ++repeat) {
    if (repeat) { hipLaunchKernelGGL(k_pair_self, perNode, dim3(kBlock), 0, st, H, P, friction, staticThreshold); ++launches; }
    if (wide) {
      hipLaunchKernelGGL(k_pair_build_wide, dim3(std::max<uint32_t>(1u, std::min<uint32_t>(16384u, n))), dim3(64), 0, st, H, P, repeat); ++launches;
    } else {
      hipLaunchKernelGGL((k_pair_build<384, 96, 64, false>), groups, dim3(64 * kBuildWaves), 0, st, H, P, repeat); ++launches;
      hipLaunchKernelGGL((k_pair_build<kMaxCand, kMaxDeg, kMaxOwn, true>), dim3(512), dim3(64 * kBuildWaves), 0, st, H, P, repeat); ++launches;
    }
    // (the repeat lists more partners and runs deeper: half as many launches again; they return at once - 2.5 us each - when
    // nothing is repeated)
    if (repeat && quads && repeatBlocks) {  // the repeat's levels: one launch (k_pair_repeat runs until the frontier is empty)
      hipLaunchKernelGGL(k_pair_repeat, dim3(repeatBlocks), dim3(kRoundBlock), 0, st, H, P, friction, staticThreshold, 1u, 1u); ++launches;
    } else if (quads && repeatBlocks) {  // the first attempt: captured level launches, then whatever is left in one launch of resident workgroups
      for (uint32_t r = 1; r <= rounds; ++r) {
        hipLaunchKernelGGL(k_pair_round4, level4, dim3(threads4), 0, st, H, P, friction, staticThreshold, r, 0u, look4);
        ++launches;
      }
      hipLaunchKernelGGL(k_pair_repeat, dim3(repeatBlocks), dim3(kRoundBlock), 0, st, H, P, friction, staticThreshold, rounds + 1u, 0u); ++launches;
    } else {
      const uint32_t captured = repeat ? rounds + rounds / 2u : rounds;
      for (uint32_t r = 1; r <= captured; ++r) {
        if (quads) hipLaunchKernelGGL(k_pair_round4, repeat ? levelRepeat4 : level4, dim3(threads4), 0, st, H, P, friction, staticThreshold, r, repeat, look4);
        else hipLaunchKernelGGL(k_pair_round, repeat ? levelRepeat : level, dim3(kRoundBlock), 0, st, H, P, friction, staticThreshold, r, repeat);
        ++launches;
      }
      hipLaunchKernelGGL(k_pair_tail, dim3(1), dim3(1024), 0, st, H, P, friction, staticThreshold, captured + 1u, repeat); ++launches;
    }
    hipLaunchKernelGGL(k_pair_verify, dim3(64), dim3(kBlock), 0, st, H, P, repeat, gridSpacing); ++launches;
    hipLaunchKernelGGL(k_pair_check, perNode, dim3(kBlock), 0, st, H, P, nd.pos, nd.vel, repeat ? 0u : 1u); ++launches;
    if (!repeat) { hipLaunchKernelGGL(k_pair_arm, dim3(1), dim3(64), 0, st, H, P); ++launches; }
  }
  // a pile the lists do not hold: the pass in the reference's own order, from the state it started with (returns at once otherwise)
  launches += launch_collide_reference(st, H, nd, gridSpacing, friction, staticThreshold, P.ctl + kPairFallback);
  return launches;
}

}  // namespace pies
