// Projective-Dynamics precompute (host, once per topology change) -- the counterpart of
// Solver::tickPD's rebuild block (Src/Solver.cpp:168-221 of the reference):
//   * K = diag(1/(invMass h^2)) + sum_c w_c A_c^T A_c, accumulated in the reference's container order
//     (position, distance, tet, volume, [shape, goal,] bend) with float `+=`, stored as CSR;
//   * per node, the list of (constraint, local index) contribution slots in the order the reference adds
//     them to the force vector (Solver.cpp:310-349), so the device right-hand side sums in that order;
//   * per node, the number of surface-triangle incidences (one floor contact each, Solver.cpp:829-834).
// The factorisation itself is replaced by CG on the device (pd_kernels.hip).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <vector>

#include "device_util.h"

namespace pies {
struct int2_host {
  int x, y;
  bool operator<(const int2_host& o) const { return x != o.x ? x < o.x : y < o.y; }
};
static_assert(sizeof(int2_host) == 8, "layout of int2");

namespace {
struct Entry {
  uint32_t col;
  float val;
};
}  // namespace

// The windowed form of the system matrix (CgArrays: wRows ...).  false: not built - a window beyond what a workgroup may hold in
// LDS or beyond 16-bit slot indices (an unstructured mesh numbered without locality); the SELL arrays serve such a scene.
struct WindowMatrix {
  uint32_t rows = 0, chunks = 0, ldsSlots = 0;
  bool sorted = false, halo16 = true;
  std::vector<uint2> chunk;
  std::vector<uint32_t> base, halo32, sliceOff;
  std::vector<uint16_t> halo16v, idx, perm;
  std::vector<float> val;
  double padding = 1.0;  // stored entries / matrix entries
};
constexpr uint32_t kWindowMaxSlots = 6144;  // 96 KB of LDS: one workgroup per compute unit is still resident beside another kernel's

bool build_window_matrix(uint32_t n, const std::vector<uint32_t>& rowptr, const std::vector<uint32_t>& col, const std::vector<float>& val,
                         uint32_t R, int sortMode, WindowMatrix& W) {
  if (n == 0 || R == 0 || R % 64u) return false;
  W.rows = R;
  W.chunks = (n + R - 1u) / R;
  const uint32_t spc = R / 64u;  // slices per chunk
  // (1) per chunk: the halo (distinct columns outside the chunk, ascending)
  W.chunk.resize(W.chunks);
  W.base.assign(W.chunks, 0u);
  std::vector<uint32_t> tmp;
  std::vector<std::vector<uint32_t>> halos(W.chunks);
  for (uint32_t c = 0; c < W.chunks; ++c) {
    const uint32_t r0 = c * R, r1 = std::min(n, r0 + R);
    tmp.clear();
    for (uint32_t k = rowptr[r0]; k < rowptr[r1]; ++k)
      if (col[k] < r0 || col[k] >= r1) tmp.push_back(col[k]);
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    if (R + tmp.size() > kWindowMaxSlots) return false;
    W.ldsSlots = std::max<uint32_t>(W.ldsSlots, R + static_cast<uint32_t>(tmp.size()));
    if (!tmp.empty() && tmp.back() - tmp.front() > 0xffffu) W.halo16 = false;
    halos[c] = tmp;
  }
  if (const char* e = tuning_env("PIES_PD_WINDOW_HALO32"); e && e[0] == '1') W.halo16 = false;  // tests: the 32-bit halo list on a small mesh
  uint32_t at = 0;
  for (uint32_t c = 0; c < W.chunks; ++c) {
    W.chunk[c] = make_uint2(at, static_cast<uint32_t>(halos[c].size()));
    W.base[c] = halos[c].empty() ? 0u : halos[c].front();
    at += static_cast<uint32_t>(halos[c].size());
  }
  if (W.halo16) {
    W.halo16v.reserve(at);
    for (uint32_t c = 0; c < W.chunks; ++c)
      for (uint32_t j : halos[c]) W.halo16v.push_back(static_cast<uint16_t>(j - W.base[c]));
  } else {
    W.halo32.reserve(at);
    for (uint32_t c = 0; c < W.chunks; ++c) W.halo32.insert(W.halo32.end(), halos[c].begin(), halos[c].end());
  }
  // (2) the rows' order inside a chunk: by length (descending, stable) when the natural order pads a tenth or more
  auto len = [&](uint32_t i) { return i < n ? rowptr[i + 1] - rowptr[i] : 0u; };
  uint64_t padNatural = 0;
  for (uint32_t sl = 0; sl < W.chunks * spc; ++sl) {
    uint32_t w = 0;
    for (uint32_t l = 0; l < 64u; ++l) w = std::max(w, len(sl * 64u + l));
    padNatural += 64ull * w;
  }
  const uint64_t nnz = rowptr[n];
  W.sorted = sortMode == 1 || (sortMode < 0 && padNatural * 10ull >= nnz * 11ull);
  std::vector<uint16_t> perm(static_cast<size_t>(W.chunks) * R);
  for (uint32_t c = 0; c < W.chunks; ++c) {
    uint16_t* pm = perm.data() + static_cast<size_t>(c) * R;
    for (uint32_t k = 0; k < R; ++k) pm[k] = static_cast<uint16_t>(k);
    if (W.sorted) std::stable_sort(pm, pm + R, [&](uint16_t a, uint16_t b) { return len(c * R + a) > len(c * R + b); });
  }
  // (3) the slices
  W.sliceOff.assign(static_cast<size_t>(W.chunks) * spc + 1u, 0u);
  for (uint32_t sl = 0; sl < W.chunks * spc; ++sl) {
    const uint32_t c = sl / spc;
    uint32_t w = 0;
    for (uint32_t l = 0; l < 64u; ++l) w = std::max(w, len(c * R + perm[static_cast<size_t>(sl) * 64u + l]));
    W.sliceOff[sl + 1] = W.sliceOff[sl] + 64u * w;
  }
  const size_t stored = W.sliceOff.back();
  W.padding = nnz ? static_cast<double>(stored) / static_cast<double>(nnz) : 1.0;
  W.val.assign(stored, 0.0f);
  W.idx.assign(stored, 0);
  for (uint32_t sl = 0; sl < W.chunks * spc; ++sl) {
    const uint32_t c = sl / spc, r0 = c * R, r1 = std::min(n, r0 + R);
    const std::vector<uint32_t>& h = halos[c];
    const uint32_t width = (W.sliceOff[sl + 1] - W.sliceOff[sl]) / 64u;
    for (uint32_t l = 0; l < 64u; ++l) {
      const uint32_t lr = perm[static_cast<size_t>(sl) * 64u + l], i = r0 + lr;
      const uint32_t b = i < n ? rowptr[i] : 0u, ln = len(i);
      for (uint32_t k = 0; k < width; ++k) {
        const size_t e = static_cast<size_t>(W.sliceOff[sl]) + 64u * k + l;
        if (k < ln) {
          const uint32_t j = col[b + k];
          W.val[e] = val[b + k];
          W.idx[e] = static_cast<uint16_t>(j >= r0 && j < r1 ? j - r0 : R + static_cast<uint32_t>(std::lower_bound(h.begin(), h.end(), j) - h.begin()));
        } else {
          W.idx[e] = static_cast<uint16_t>(lr);  // padding: the row's own slot, value 0
        }
      }
    }
  }
  if (W.sorted) W.perm.swap(perm);
  return true;
}

int pd_build(pies_solver* s) {
  const uint32_t n = s->nodeCount();
  const float h = s->opt.fixedTimestepSize / s->opt.timeSubsteps;
  const float h2 = h * h;

  // ---- K, row by row, entries appended in the reference's accumulation order -------------------------
  std::vector<std::vector<Entry>> rows(n);
  for (uint32_t i = 0; i < n; ++i) rows[i].push_back({i, 1.0f / (s->h_invMass[i] * h2)});  // Solver.cpp:179-182
  for (const HostPosition& c : s->h_position) rows[c.id].push_back({c.id, c.w * 1.0f});    // A = B = 1
  for (const HostDistance& c : s->h_distance) {                                            // A^T A = [[.5,-.5],[-.5,.5]]
    const float AtA[2][2] = {{0.5f * 0.5f + -0.5f * -0.5f, 0.5f * -0.5f + -0.5f * 0.5f},
                             {-0.5f * 0.5f + 0.5f * -0.5f, -0.5f * -0.5f + 0.5f * 0.5f}};
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) rows[c.ids[i]].push_back({c.ids[j], c.w * AtA[i][j]});
  }
  for (const std::vector<HostTet>* list : {&s->h_tet, &s->h_volume})
    for (const HostTet& c : *list)
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) rows[c.ids[i]].push_back({c.ids[j], c.w * c.AtA[4 * i + j]});
  for (const HostShape& c : s->h_shape)  // ShapeMatchingConstraint.cpp:50-56
    for (uint32_t id : c.ids) rows[id].push_back({id, c.w});
  for (const HostGoal& c : s->h_goal)  // :139-145
    for (uint32_t id : c.ids) rows[id].push_back({id, c.w});
  for (const HostBend& c : s->h_bend)  // A = I: w on the diagonal (the off-diagonal terms are w*0)
    for (uint32_t id : c.ids) rows[id].push_back({id, c.w * 1.0f});
  for (const HostNodePair& c : s->h_nodePair)  // CollisionConstraint.cpp:43-47 (extension container: the pairs are the scene's, so
    for (uint32_t id : c.ids) rows[id].push_back({id, kNodePairW});  // their diagonal terms are part of K, not rebuilt every substep)

  std::vector<uint32_t> rowptr(n + 1, 0), col;
  std::vector<float> val, kdiag(n, 0.f);
  col.reserve(static_cast<size_t>(n) * 16);
  val.reserve(static_cast<size_t>(n) * 16);
  for (uint32_t i = 0; i < n; ++i) {
    std::vector<Entry>& r = rows[i];
    std::stable_sort(r.begin(), r.end(), [](const Entry& a, const Entry& b) { return a.col < b.col; });
    for (size_t k = 0; k < r.size();) {
      float acc = r[k].val;  // coeffRef(i,j) starts at 0: 0 + v == v
      size_t m = k + 1;
      for (; m < r.size() && r[m].col == r[k].col; ++m) acc += r[m].val;
      col.push_back(r[k].col);
      val.push_back(acc);
      if (r[k].col == i) kdiag[i] = acc;
      k = m;
    }
    rowptr[i + 1] = static_cast<uint32_t>(col.size());
    std::vector<Entry>().swap(r);
  }
  s->pd_nnz = static_cast<uint32_t>(col.size());
  // CSR -> sliced ELL (see CgArrays).  A slice is one wavefront's worth of rows: 64 rows with one lane each.  (Several lanes
  // per row - lane L r + q taking entries q, q + L, ... of row r, partial sums combined by shuffles - were measured at 100k
  // rows: k_cg_ap 5.4 us with 1 lane per row, 7.8 / 10.4 / 14.5 us with 2 / 4 / 8: the extra wavefronts only add gather
  // instructions.  PIES_SELL_LANES keeps the experiment available.)  Entries keep their order (ascending column).
  uint32_t lpr = 1u;
  if (const char* e = tuning_env("PIES_SELL_LANES")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) lpr = static_cast<uint32_t>(v); }
  const uint32_t rps = 64u / lpr;
  const uint32_t nslices = (n + rps - 1u) / rps;
  std::vector<uint32_t> sliceOff(nslices + 1, 0), sellCol;
  std::vector<float> sellVal;
  for (uint32_t sl = 0; sl < nslices; ++sl) {
    uint32_t width = 0;
    for (uint32_t i = rps * sl; i < std::min(n, rps * sl + rps); ++i) width = std::max(width, rowptr[i + 1] - rowptr[i]);
    sliceOff[sl + 1] = sliceOff[sl] + 64u * ((width + lpr - 1u) / lpr);
  }
  sellCol.resize(sliceOff[nslices]);
  sellVal.assign(sliceOff[nslices], 0.0f);
  for (uint32_t sl = 0; sl < nslices; ++sl) {
    const uint32_t steps = (sliceOff[sl + 1] - sliceOff[sl]) / 64u;
    for (uint32_t r = 0; r < rps; ++r) {
      const uint32_t i = rps * sl + r, self = std::min(i, n - 1u);
      const uint32_t b = i < n ? rowptr[i] : 0u, len = i < n ? rowptr[i + 1] - rowptr[i] : 0u;
      for (uint32_t k = 0; k < steps * lpr; ++k) {
        const size_t at = static_cast<size_t>(sliceOff[sl]) + 64u * (k / lpr) + (r * lpr + k % lpr);
        sellCol[at] = k < len ? col[b + k] : self;
        if (k < len) sellVal[at] = val[b + k];
      }
    }
  }

  // ---- contribution slots and per-node incidence lists -------------------------------------------------
  const uint32_t cnt[6] = {(uint32_t)s->h_position.size(), (uint32_t)s->h_distance.size(), (uint32_t)s->h_tet.size(),
                           (uint32_t)s->h_volume.size(), (uint32_t)s->h_bend.size(), (uint32_t)s->h_nodePair.size()};
  const uint32_t arity[6] = {1, 2, 4, 4, 4, 2};
  // strain + volume element pairs in tiles: one record per (tile, node) instead of four per element (pd_tiles.cpp)
  PdTilePlan tiles;
  const bool tiled = pd_plan_tiles(s, tiles);
  s->pdTiles = tiled ? static_cast<uint32_t>(tiles.info.size()) : 0u;
  s->pdTileRecords = tiled ? static_cast<uint32_t>(tiles.tileNodes) : 0u;
  uint32_t total = 0;
  for (int t = 0; t < 6; ++t) {
    s->slotBase[t] = total;
    if (!(tiled && (t == PIES_TET || t == PIES_VOLUME))) total += cnt[t] * arity[t];
  }
  const uint32_t tileSlotBase = total;
  if (tiled) total += static_cast<uint32_t>(tiles.node.size());  // (kTileNodes slots per tile)
  std::vector<uint32_t> incPtr(n + 1, 0);
  auto for_each_incidence = [&](auto&& fn) {
    // reference order of setupGlobalForceVector calls: position, distance, tet, volume, bend (Solver.cpp:310-327)
    for (uint32_t c = 0; c < cnt[0]; ++c) fn(s->h_position[c].id, s->slotBase[0] + c);
    for (uint32_t c = 0; c < cnt[1]; ++c)
      for (uint32_t i = 0; i < 2; ++i) fn(s->h_distance[c].ids[i], s->slotBase[1] + i * cnt[1] + c);
    for (uint32_t c = 0; c < (tiled ? 0u : cnt[2]); ++c)
      for (uint32_t i = 0; i < 4; ++i) fn(s->h_tet[c].ids[i], s->slotBase[2] + i * cnt[2] + c);
    if (tiled)  // a node's tile sums, in ascending tile order
      for (uint32_t t = 0; t < tiles.info.size(); ++t)
        for (uint32_t k = 0; k < (tiles.info[t] & 0xffffu); ++k) fn(tiles.node[static_cast<size_t>(kTileNodes) * t + k], tileSlotBase + kTileNodes * t + k);
    // strain + volume constraints over identical elements (tetVolumePaired): the fused local step adds the volume
    // contribution into the strain constraint's record, so the volume slots are not gathered
    for (uint32_t c = 0; c < (s->tetVolumePaired ? 0u : cnt[3]); ++c)
      for (uint32_t i = 0; i < 4; ++i) fn(s->h_volume[c].ids[i], s->slotBase[3] + i * cnt[3] + c);
    for (uint32_t c = 0; c < cnt[4]; ++c)
      for (uint32_t i = 0; i < 4; ++i) fn(s->h_bend[c].ids[i], s->slotBase[4] + i * cnt[4] + c);
    for (uint32_t c = 0; c < cnt[5]; ++c)  // (the reference adds collision terms after every other container, Solver.cpp:337-349)
      for (uint32_t i = 0; i < 2; ++i) fn(s->h_nodePair[c].ids[i], s->slotBase[5] + i * cnt[5] + c);
  };
  for_each_incidence([&](uint32_t node, uint32_t) { ++incPtr[node + 1]; });
  for (uint32_t i = 0; i < n; ++i) incPtr[i + 1] += incPtr[i];
  std::vector<uint32_t> incSlot(incPtr[n]), cur(incPtr.begin(), incPtr.end() - 1);
  for_each_incidence([&](uint32_t node, uint32_t slot) { incSlot[cur[node]++] = slot; });

  // fp64 slots: shape entries, then goal entries (reference order :329-335)
  std::vector<uint32_t> shOff(1, 0), shNode;
  std::vector<double> shMat, shQinv, shQuat;
  std::vector<float> shW;
  for (const HostShape& c : s->h_shape) {
    shNode.insert(shNode.end(), c.ids.begin(), c.ids.end());
    shMat.insert(shMat.end(), c.mat.begin(), c.mat.end());
    shQinv.insert(shQinv.end(), c.Qinv, c.Qinv + 9);
    shQuat.insert(shQuat.end(), {1.0, 0.0, 0.0, 0.0});  // _currentRotation = identity (ShapeMatchingConstraint.cpp:14)
    shW.push_back(c.w);
    shOff.push_back(static_cast<uint32_t>(shNode.size()));
  }
  s->goalSlotBase = static_cast<uint32_t>(shNode.size());
  uint32_t dtotal = s->goalSlotBase;
  for (const HostGoal& c : s->h_goal) dtotal += static_cast<uint32_t>(c.ids.size());
  std::vector<uint32_t> incPtrD(n + 1, 0);
  auto for_each_d = [&](auto&& fn) {
    uint32_t slot = 0;
    for (const HostShape& c : s->h_shape)
      for (uint32_t id : c.ids) fn(id, slot++);
    for (const HostGoal& c : s->h_goal)
      for (uint32_t id : c.ids) fn(id, slot++);
  };
  for_each_d([&](uint32_t node, uint32_t) { ++incPtrD[node + 1]; });
  for (uint32_t i = 0; i < n; ++i) incPtrD[i + 1] += incPtrD[i];
  std::vector<uint32_t> incSlotD(incPtrD[n]), curD(incPtrD.begin(), incPtrD.end() - 1);
  for_each_d([&](uint32_t node, uint32_t slot) { incSlotD[curD[node]++] = slot; });

  std::vector<uint32_t> triCount(n, 0);
  for (uint32_t id : s->h_triangles) ++triCount[id];

  // position constraints project to a constant: w * (A^T B p) = w * (0 + 1*target)
  std::vector<Vec3f> contrib0(cnt[0]);
  for (uint32_t c = 0; c < cnt[0]; ++c) {
    const HostPosition& p = s->h_position[c];
    contrib0[c] = Vec3f{p.w * (0.0f + 1.0f * p.target[0]), p.w * (0.0f + 1.0f * p.target[1]), p.w * (0.0f + 1.0f * p.target[2])};
  }

  // ---- HBM ---------------------------------------------------------------------------------------------
  PdArrays& pd = s->pd;
  CgArrays& cg = pd.cg;
  cg.n = n;
  // blocks of every CG launch: one per four slices (4 wavefronts per block).  Measured at 100k rows (substeps/s of config 3):
  // 64 blocks 1039, 128: 1188, 196: 1242, 256: 1268, 391 (this formula): 1246 - fewer blocks make the per-kernel re-reduction of
  // the partial dot products cheaper and the SpMV slower, by about the same amount.
  cg.nparts = std::max(1u, std::min(kCgBlocks, (nslices + 3u) / 4u));
  if (const char* e = tuning_env("PIES_CG_BLOCKS")) { const int v = std::atoi(e); if (v >= 1 && v <= (int)kCgBlocks) cg.nparts = static_cast<uint32_t>(v); }
  // k_cg_update's in-kernel continuation synchronises its workgroups with a grid barrier: the grid must fit the device at once
  // (occupancy of the kernel x compute units of THIS device; half of it, so that a second solver on the card leaves room)
  if (s->device >= 0) {
    const uint32_t resident = std::min(cg_update_resident_blocks(s->device), cg1_iter_resident_blocks(s->device));
    if (resident >= 2) cg.nparts = std::max(1u, std::min(cg.nparts, resident / 2u));
  }
  // The launches of the one-launch-per-iteration form that have no grid barrier (all but a solve's last): as many workgroups as the
  // rows ask for, whatever the device holds at once (1M rows: 1 024 instead of the 512 the continuation's barrier allows - the
  // launches are gathers, and more wavefronts in flight hide more of them)
  // (two rows per thread and more: at 250k rows 977 workgroups were slower than the 512 the barrier allows - every workgroup
  // re-reduces every workgroup's partial sums - at 1M rows 1 024 are 12 % faster than 512)
  cg.npartsI = std::max(cg.nparts, std::min(kCgBlocks, (nslices + 7u) / 8u));
  if (const char* e = tuning_env("PIES_CG_INIT_BLOCKS")) { const int v = std::atoi(e); if (v >= 1 && v <= (int)kCgInitBlocks) cg.npartsI = static_cast<uint32_t>(v); }
  uint32_t *d_rowptr, *d_col, *d_incPtr, *d_incSlot, *d_tri;
  float *d_val, *d_kdiag;
  if (int rc = upload(s, sliceOff, &d_rowptr)) return rc;
  if (int rc = upload(s, sellCol, &d_col)) return rc;
  if (int rc = upload(s, sellVal, &d_val)) return rc;
  if (int rc = upload(s, kdiag, &d_kdiag)) return rc;
  if (int rc = upload(s, incPtr, &d_incPtr)) return rc;
  if (int rc = upload(s, incSlot, &d_incSlot)) return rc;
  if (int rc = upload(s, triCount, &d_tri)) return rc;

  cg.sliceOff = d_rowptr; cg.col = d_col; cg.val = d_val;
  // Row dictionary: every row as (column - row, value) pairs; rows with equal sequences share an entry.  Used when the scene
  // has few distinct rows (a lattice: the interior row and the classes of boundary rows, a few hundred at most).
  cg.rowStencil = nullptr; cg.stencil = nullptr;
  {
    const char* e = tuning_env("PIES_PD_ROW_DICT");
    if (lpr == 1u && n != 0 && !(e && e[0] == '0')) {
      std::map<std::vector<int2_host>, uint16_t> ids;
      std::vector<uint32_t> rowStencil(n);
      std::vector<uint32_t> stencilAt(1, 0);
      std::vector<int2_host> stencil;
      bool ok = true;
      std::vector<int2_host> key;
      for (uint32_t i = 0; ok && i < n; ++i) {
        key.clear();
        for (uint32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
          int2_host p;
          p.x = static_cast<int>(col[k]) - static_cast<int>(i);
          std::memcpy(&p.y, &val[k], sizeof(float));
          key.push_back(p);
        }
        auto it = ids.find(key);
        if (it == ids.end()) {
          if (ids.size() >= 4096 || (ids.size() + 1) * 8 > n || key.size() > 255 || stencil.size() + key.size() >= (1u << 24)) { ok = false; break; }  // no real compression: the SELL arrays
          it = ids.emplace(key, static_cast<uint16_t>(ids.size())).first;
          stencil.insert(stencil.end(), key.begin(), key.end());
          stencilAt.push_back(static_cast<uint32_t>(stencil.size()));
        }
        rowStencil[i] = stencilAt[it->second] | (static_cast<uint32_t>(key.size()) << 24);
      }
      if (ok) {
        uint32_t* d_rs;
        int2_host* d_st;
        if (int rc = upload(s, rowStencil, &d_rs)) return rc;
        if (int rc = upload(s, stencil, &d_st)) return rc;
        cg.rowStencil = d_rs; cg.stencil = reinterpret_cast<const int2*>(d_st);
        s->pdRowStencils = static_cast<uint32_t>(ids.size());
      }
    }
  }
  cg.lanesPerRow = lpr;
  // Windowed SELL for the scenes that stream their matrix (no row dictionary: unstructured meshes, lattices with per-node
  // materials): see CgArrays.  PIES_PD_WINDOW=0 keeps the SELL arrays, =2 builds it beside a row dictionary as well (the kernels
  // then take the window); PIES_CG_CHUNK_ROWS = rows per chunk (64 ... 1024; default 256: a 100k-row system still fills the chip
  // with one chunk per workgroup); PIES_PD_WINDOW_SORT = 0 / 1 forces the rows' order inside a chunk.
  cg.wRows = cg.wChunks = cg.wLdsSlots = 0;
  cg.wChunk = nullptr; cg.wBase = nullptr; cg.wHalo16 = nullptr; cg.wHalo32 = nullptr; cg.wSliceOff = nullptr; cg.wVal = nullptr;
  cg.wIdx = nullptr; cg.wPerm = nullptr;
  s->pdWindowPadding = 0.0f; s->pdWindowHalo = 0; s->pdWindowEntries = 0;
  {
    int mode = 1;
    if (const char* e = tuning_env("PIES_PD_WINDOW")) mode = std::atoi(e);
    uint32_t R = 256;
    if (const char* e = tuning_env("PIES_CG_CHUNK_ROWS")) { const int v = std::atoi(e); if (v >= 64 && v <= 1024 && v % 64 == 0) R = static_cast<uint32_t>(v); }
    int sortMode = -1;
    if (const char* e = tuning_env("PIES_PD_WINDOW_SORT")) sortMode = std::atoi(e) ? 1 : 0;
    WindowMatrix W;
    if (lpr == 1u && n != 0 && s->device >= 0 && (mode == 2 || (mode == 1 && !cg.rowStencil)) && build_window_matrix(n, rowptr, col, val, R, sortMode, W)) {
      uint2* d_chunk; uint32_t *d_base, *d_h32, *d_so; uint16_t *d_h16, *d_idx, *d_perm; float* d_wv;
      if (int rc = upload(s, W.chunk, &d_chunk)) return rc;
      if (int rc = upload(s, W.base, &d_base)) return rc;
      if (int rc = upload(s, W.halo16v, &d_h16)) return rc;
      if (int rc = upload(s, W.halo32, &d_h32)) return rc;
      if (int rc = upload(s, W.sliceOff, &d_so)) return rc;
      if (int rc = upload(s, W.val, &d_wv)) return rc;
      if (int rc = upload(s, W.idx, &d_idx)) return rc;
      if (int rc = upload(s, W.perm, &d_perm)) return rc;
      cg.wRows = W.rows; cg.wChunks = W.chunks; cg.wLdsSlots = W.ldsSlots;
      cg.wChunk = d_chunk; cg.wBase = d_base; cg.wHalo16 = W.halo16 ? d_h16 : nullptr; cg.wHalo32 = W.halo16 ? nullptr : d_h32;
      if (W.halo16 && W.halo16v.empty()) { cg.wHalo16 = nullptr; cg.wHalo32 = nullptr; }  // (no halo at all: a single chunk)
      cg.wSliceOff = d_so; cg.wVal = d_wv; cg.wIdx = d_idx; cg.wPerm = W.sorted ? d_perm : nullptr;
      s->pdWindowPadding = static_cast<float>(W.padding);
      s->pdWindowEntries = static_cast<uint32_t>(W.val.size());
      s->pdWindowHalo = static_cast<uint32_t>(W.halo16 ? W.halo16v.size() : W.halo32.size());
      // no more workgroups than chunks; the continuation's grid must fit the device beside the window's LDS
      cg.nparts = std::max(1u, std::min(cg.nparts, cg.wChunks));
      cg.npartsI = std::max(cg.nparts, std::min(cg.npartsI, cg.wChunks));
      const uint32_t resident = cg1_iter_resident_blocks(s->device, window_lds_bytes(cg));
      if (resident >= 2) cg.nparts = std::max(1u, std::min(cg.nparts, resident / 2u));
    }
  }
  // lanes of k_pd_rhs per node: four for the ~24 per-element records of a node, one when they are a few tile sums
  pd.rhsLanes = n && incPtr[n] <= 6ull * n ? 1u : 4u;
  if (const char* e = tuning_env("PIES_PD_RHS_LANES")) pd.rhsLanes = std::atoi(e) == 1 ? 1u : 4u;
  pd.tiles = PdTileArrays{};
  if (tiled) {
    PdTileArrays& T = pd.tiles;
    uint32_t *d_info, *d_node, *d_local;
    uint16_t *d_nptr, *d_inc;
    if (int rc = upload(s, tiles.info, &d_info)) return rc;
    if (int rc = upload(s, tiles.node, &d_node)) return rc;
    if (int rc = upload(s, tiles.local, &d_local)) return rc;
    if (int rc = upload(s, tiles.nptr, &d_nptr)) return rc;
    if (int rc = upload(s, tiles.inc, &d_inc)) return rc;
    T.ntiles = static_cast<uint32_t>(tiles.info.size());
    T.info = d_info; T.node = d_node; T.local = d_local; T.nptr = d_nptr; T.inc = d_inc;
    if (!s->h_pairDictIndex.empty() && s->d_pairDictTable) {
      std::vector<uint16_t> idx(tiles.elem.size());
      for (size_t k = 0; k < idx.size(); ++k) idx[k] = s->h_pairDictIndex[tiles.elem[k]];
      uint16_t* d_idx;
      if (int rc = upload(s, idx, &d_idx)) return rc;
      T.dict = d_idx;
    } else {
      std::vector<float4> q0(tiles.elem.size()), q1(q0.size()), q2(q0.size()), vq2(q0.size());
      for (size_t k = 0; k < q0.size(); ++k) {
        const HostTet &a = s->h_tet[tiles.elem[k]], &b = s->h_volume[tiles.elem[k]];
        q0[k] = make_float4(a.qinv[0], a.qinv[1], a.qinv[2], a.qinv[3]);
        q1[k] = make_float4(a.qinv[4], a.qinv[5], a.qinv[6], a.qinv[7]);
        q2[k] = make_float4(a.qinv[8], a.lo, a.hi, a.w);
        vq2[k] = make_float4(b.qinv[8], b.lo, b.hi, b.w);
      }
      float4 *d0, *d1, *d2, *d3;
      if (int rc = upload(s, q0, &d0)) return rc;
      if (int rc = upload(s, q1, &d1)) return rc;
      if (int rc = upload(s, q2, &d2)) return rc;
      if (int rc = upload(s, vq2, &d3)) return rc;
      T.q0 = d0; T.q1 = d1; T.q2 = d2; T.vq2 = d3;
    }
  }
  pd.kdiag = d_kdiag; pd.incPtr = d_incPtr; pd.incSlot = d_incSlot; pd.triCount = d_tri;
  pd.contribD = nullptr; pd.incPtrD = nullptr; pd.incSlotD = nullptr;
  pd.shape = ShapeArrays{};
  if (dtotal) {
    uint32_t *d_ipd, *d_isd, *d_off, *d_node;
    double *d_mat, *d_qinv, *d_quat;
    float* d_w;
    if (int rc = upload(s, incPtrD, &d_ipd)) return rc;
    if (int rc = upload(s, incSlotD, &d_isd)) return rc;
    if (int rc = dev_alloc(s, dtotal, &pd.contribD, true)) return rc;
    pd.incPtrD = d_ipd; pd.incSlotD = d_isd;
    if (!s->h_shape.empty()) {
      if (int rc = upload(s, shOff, &d_off)) return rc;
      if (int rc = upload(s, shNode, &d_node)) return rc;
      if (int rc = upload(s, shMat, &d_mat)) return rc;
      if (int rc = upload(s, shQinv, &d_qinv)) return rc;
      if (int rc = upload(s, shQuat, &d_quat)) return rc;
      if (int rc = upload(s, shW, &d_w)) return rc;
      pd.shape = ShapeArrays{(uint32_t)s->h_shape.size(), d_off, d_node, d_mat, d_qinv, d_quat, d_w};
    }
    s->goalDirty = true;
  }
  if (int rc = dev_alloc(s, n, &pd.msn)) return rc;
  if (int rc = dev_alloc(s, n, &pd.rhs)) return rc;
  if (int rc = dev_alloc(s, n, &pd.statp, true)) return rc;
  if (int rc = dev_alloc(s, n, &pd.nstatic, true)) return rc;
  if (int rc = dev_alloc(s, std::max<size_t>(total, 1), &pd.contrib, true)) return rc;
  if (tiled) pd.tiles.partial = pd.contrib + tileSlotBase;
  if (int rc = dev_alloc(s, n, &cg.cdiag, true)) return rc;
  if (int rc = dev_alloc(s, n, &cg.dinv, true)) return rc;
  if (int rc = dev_alloc(s, n, &cg.r)) return rc;
  if (int rc = dev_alloc(s, n, &cg.z)) return rc;
  if (int rc = dev_alloc(s, n, &cg.p[0], true)) return rc;
  if (int rc = dev_alloc(s, n, &cg.p[1], true)) return rc;
  if (int rc = dev_alloc(s, n, &cg.ap)) return rc;
  if (int rc = dev_alloc(s, kCgInitBlocks * 9, &cg.partI, true)) return rc;
  if (int rc = dev_alloc(s, 4, &cg.partCount, true)) return rc;
  if (int rc = dev_alloc(s, (kCgBlocks + kCgRowBlocks) * 3, &cg.partA, true)) return rc;
  if (int rc = dev_alloc(s, kCgBlocks * 6, &cg.partB, true)) return rc;
  if (int rc = dev_alloc(s, kCgBlocks * 6, &cg.partBnext, true)) return rc;
  cg.partB0 = cg.partB;
  cg.partB1 = cg.partBnext;
  if (int rc = dev_alloc(s, 32, &cg.scal, true)) return rc;
  // one launch per iteration (pd_cg1_kernels.hip): 12-byte vector records in ping-pong pairs, partial sums of 9 per workgroup
  for (int b = 0; b < 2; ++b) {
    if (int rc = dev_alloc(s, n, &cg.t1[b], true)) return rc;
    if (int rc = dev_alloc(s, n, &cg.c1[b], true)) return rc;
    if (int rc = dev_alloc(s, n, &cg.a1[b], true)) return rc;
    // (k_cg1_first / k_cg1_iter run npartsI workgroups wide, and PIES_CG_INIT_BLOCKS may raise that to kCgInitBlocks)
    if (int rc = dev_alloc(s, kCgInitBlocks * 9, &cg.part1[b], true)) return rc;
  }
  if (int rc = dev_alloc(s, n, &cg.p1, true)) return rc;
  cg.kdiag = d_kdiag;
  cg.single = 0;
  if (int rc = dev_alloc(s, 2, &cg.ticket, true)) return rc;
  if (int rc = dev_alloc(s, 8, &cg.stats, true)) return rc;
  // ---- point-triangle contact pipeline (Solver.cpp:680-875) ------------------------------------------------
  pd.tri = TriArrays{};
  cg.tIncCnt = cg.tIncStart = cg.tInc = nullptr;
  cg.tIds = nullptr;
  cg.tUsed = cg.tUsedCount = nullptr;
  cg.rowStart = cg.rowLen = cg.rowCol = nullptr;
  cg.rowCoef = nullptr;
  cg.cAp = nullptr;
  cg.useCAp = 0;
  pd.tContrib = nullptr;
  const uint32_t nt = static_cast<uint32_t>(s->h_triangles.size() / 3);
  if (s->triangleCollisions && nt) {
    TriArrays& T = pd.tri;
    T.nt = nt;
    T.threadCount = std::max(1u, s->opt.threadCount);
    uint32_t* d_tris;
    if (int rc = upload(s, s->h_triangles, &d_tris)) return rc;
    T.tris = d_tris;
    // The cell tables of the three size classes (tri_kernels.h): slots are indexed by the cell coordinates modulo the table's
    // dimensions - powers of two, the scene's extent at finalize where that fits 2^22 slots, and never fewer than a
    // search window is long (23 cells of the finest class, 9 and 7 of the coarser ones), so that no window meets a slot twice.
    // A scene that outgrows its table shares slots between distant cells: more candidates for the exact range test, nothing else.
    float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};
    for (uint32_t i = 0; i < n; ++i)
      for (int k = 0; k < 3; ++k) {
        const float v = s->h_pos[3ull * i + k];
        if (!(std::fabs(v) < 1.0e6f)) continue;
        lo[k] = i == 0 ? v : std::min(lo[k], v);
        hi[k] = i == 0 ? v : std::max(hi[k], v);
      }
    uint32_t lg[3];
    for (int k = 0; k < 3; ++k) {
      const double want = (static_cast<double>(hi[k]) - lo[k]) + 2.0;
      lg[k] = 5;
      while (lg[k] < 10 && (1u << lg[k]) < want) ++lg[k];
    }
    while (lg[0] + lg[1] + lg[2] > 22) {  // the longest axis gives way first
      int k = lg[0] >= lg[1] && lg[0] >= lg[2] ? 0 : lg[1] >= lg[2] ? 1 : 2;
      --lg[k];
    }
    static const uint32_t shifts[kTriLevels] = {0, 2, 4}, floorLg[kTriLevels] = {5, 4, 3};
    uint32_t slots = 0;
    for (int l = 0; l < kTriLevels; ++l) {
      TriGridLevel& L = T.level[l];
      L.base = slots;
      L.shift = shifts[l];
      L.lx = std::max(floorLg[l], lg[0] > shifts[l] ? lg[0] - shifts[l] : 0u);
      L.ly = std::max(floorLg[l], lg[1] > shifts[l] ? lg[1] - shifts[l] : 0u);
      L.lz = std::max(floorLg[l], lg[2] > shifts[l] ? lg[2] - shifts[l] : 0u);
      slots += 1u << (L.lx + L.ly + L.lz);
    }
    // (a hit between two long triangles is listed once per cell they share - up to 1000 times: a floor under the list keeps a
    // handful of floor quads from overflowing a small scene)
    T.maxContacts = std::max<uint32_t>(16 * nt + 1024, 1u << 16);
    slots = (slots + 2047u) & ~2047u;  // whole tiles of the prefix sum
    T.slots = slots;
    if (int rc = dev_alloc(s, slots, &T.cellCnt, true)) return rc;
    if (int rc = dev_alloc(s, slots + 1ull, &T.cellStart, true)) return rc;
    if (int rc = dev_alloc(s, slots / 2048u + 1u, &T.tileSum, true)) return rc;
    if (int rc = dev_alloc(s, nt, &T.cellOf)) return rc;
    HIP_TRY(s, hipMemsetAsync(T.cellOf, 0xFF, static_cast<size_t>(nt) * sizeof(uint32_t), s->stream));
    if (int rc = dev_alloc(s, nt, &T.posIn, true)) return rc;
    if (int rc = dev_alloc(s, 4ull * nt, &T.ent, true)) return rc;
    if (int rc = dev_alloc(s, 4ull * nt, &T.boxOf, true)) return rc;
    if (int rc = dev_alloc(s, 16, &T.counters, true)) return rc;
    if (int rc = dev_alloc(s, nt, &T.rng, true)) return rc;
    if (int rc = dev_alloc(s, nt, &T.head, true)) return rc;
    if (int rc = dev_alloc(s, T.maxContacts, &T.pool, true)) return rc;
    if (nt >= (1u << 29)) return fail(s, PIES_ERR_UNSUPPORTED, "too many surface triangles for the contact work list");
    T.maxWork = static_cast<uint32_t>(std::min<uint64_t>((128ull * nt + 65536ull) & ~63ull, 1ull << 30));  // 64 lists of a 64th each
    if (int rc = dev_alloc(s, 64u * 16u, &T.workCnt, true)) return rc;
    if (int rc = dev_alloc(s, T.maxWork, &T.work, true)) return rc;
    if (int rc = dev_alloc(s, nt, &T.cntTri, true)) return rc;
    if (int rc = dev_alloc(s, nt, &T.offTri, true)) return rc;
    if (int rc = dev_alloc(s, T.maxContacts, &T.ids, true)) return rc;
    if (int rc = dev_alloc(s, 4ull * T.maxContacts, &T.contrib, true)) return rc;
    if (int rc = dev_alloc(s, n, &T.incCnt, true)) return rc;
    if (int rc = dev_alloc(s, n, &T.incStart, true)) return rc;
    if (int rc = dev_alloc(s, n, &T.incFill, true)) return rc;
    if (int rc = dev_alloc(s, n, &T.usedNodes, true)) return rc;
    if (int rc = dev_alloc(s, 4ull * T.maxContacts, &T.inc, true)) return rc;
    if (int rc = dev_alloc(s, 4ull * T.maxContacts, &T.incSorted, true)) return rc;
    if (int rc = dev_alloc(s, 4ull * T.maxContacts, &T.incPos, true)) return rc;
    if (int rc = dev_alloc(s, n, &T.lastLevel)) return rc;
    HIP_TRY(s, hipMemsetAsync(T.lastLevel, 0xFF, static_cast<size_t>(n) * sizeof(int), s->stream));  // -1
    if (int rc = dev_alloc(s, T.maxContacts, &T.lvl, true)) return rc;
    if (int rc = dev_alloc(s, T.maxContacts, &T.lvOrder, true)) return rc;
    if (int rc = dev_alloc(s, kTriMaxLevels + 1, &T.lvStart, true)) return rc;
    if (int rc = dev_alloc(s, (n + 31ull) / 32 + 1, &T.usedBits, true)) return rc;
    if (int rc = dev_alloc(s, n, &T.nodeSlot, true)) return rc;
    if (int rc = dev_alloc(s, n, &T.rowStart, true)) return rc;
    if (int rc = dev_alloc(s, n, &T.rowLen, true)) return rc;
    if (int rc = dev_alloc(s, 6ull * T.maxContacts, &T.rowCol, true)) return rc;
    if (int rc = dev_alloc(s, 6ull * T.maxContacts, &T.rowCoef, true)) return rc;
    if (int rc = dev_alloc(s, T.maxContacts, &T.lvSlots, true)) return rc;
    cg.tIncCnt = T.incCnt; cg.tIncStart = T.incStart; cg.tInc = T.incSorted; cg.tIds = T.ids;
    cg.tUsed = T.usedNodes; cg.tUsedCount = T.counters + 4;
    cg.rowStart = T.rowStart; cg.rowLen = T.rowLen; cg.rowCol = T.rowCol; cg.rowCoef = T.rowCoef;
    if (int rc = dev_alloc(s, n, &cg.cAp, true)) return rc;
    if (const char* e = tuning_env("PIES_TRI_FAST_ROWS")) s->triFastRows = std::atoi(e) != 0;  // tests: force a variant from the first tick
    cg.useCAp = s->triFastRows ? 1 : 0;
    pd.tContrib = T.contrib;
  }
  if (!contrib0.empty())
    HIP_TRY(s, hipMemcpyAsync(pd.contrib + s->slotBase[0], contrib0.data(), contrib0.size() * sizeof(Vec3f),
                              hipMemcpyHostToDevice, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return pd_upload_goals(s);
}

// GoalMatchingConstraint::projectToAuxiliaryVariable (ShapeMatchingConstraint.cpp:163-173): projected =
// transform * (material, 1) in float (glm mat4*vec4), stored as double.  The targets only change when a
// transform does, so they are evaluated on the host and copied into the fp64 contribution slots.
int pd_upload_goals(pies_solver* s) {
  if (!s->goalDirty || !s->pd.contribD) { s->goalDirty = false; return PIES_OK; }
  std::vector<double4> out;
  for (const HostGoal& c : s->h_goal)
    for (size_t i = 0; i < c.ids.size(); ++i) {
      const float x = c.mat[3 * i], y = c.mat[3 * i + 1], z = c.mat[3 * i + 2];
      const float* m = c.transform;
      float o[3];
      for (int r = 0; r < 3; ++r) o[r] = (m[r] * x + m[4 + r] * y) + (m[8 + r] * z + m[12 + r] * 1.0f);
      out.push_back(make_double4(o[0], o[1], o[2], c.w));
    }
  if (!out.empty()) {
    HIP_TRY(s, hipMemcpyAsync(s->pd.contribD + s->goalSlotBase, out.data(), out.size() * sizeof(double4), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(s, hipStreamSynchronize(s->stream));
  }
  s->goalDirty = false;
  return PIES_OK;
}

}  // namespace pies
