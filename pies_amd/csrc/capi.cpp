// C ABI (include/pies_hip.h): handle lifetime, upload to HBM, substep graph capture, tick, state access.
// The substep itself is Solver::tickPBD (Src/Solver.cpp:40-160) / tickPD (:162-486) re-expressed as a
// fixed sequence of kernel launches captured once into a hipGraph: at 100k particles a conflict-free
// batch runs for a few microseconds, so un-graphed launches would be host-bound.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "device_util.h"

#include <map>
#include <memory>
#include <mutex>

using namespace pies;

namespace pies {

// pies_set_tuning's registry (see kernels.h: tuning_env).  Values live as long as the process: a pointer handed out stays valid.
static std::mutex& tuning_mutex() { static std::mutex m; return m; }
// Every value ever set is kept in an append-only list of heap strings, and the map points at the current one: a pointer handed
// out stays valid and its bytes never change, whatever pies_set_tuning does afterwards.  (pies_set_tuning must still not race
// with pies_finalize / a capture of the same process: the switches are read at different times of a handle's life.)
static std::vector<std::unique_ptr<std::string>>& tuning_values() { static std::vector<std::unique_ptr<std::string>> v; return v; }
static std::map<std::string, const std::string*>& tuning_current() { static std::map<std::string, const std::string*> m; return m; }
const char* tuning_env(const char* name) {
  std::lock_guard<std::mutex> lock(tuning_mutex());
  auto it = tuning_current().find(name);
  return it == tuning_current().end() || !it->second || it->second->empty() ? nullptr : it->second->c_str();
}

// PIES_PROFILER_SAFE=1 (set by the profiling scripts): rocprofv3 7.2 on this pool segfaults when tens of
// thousands of graph kernel nodes are queued without a synchronisation, or when a graph is destroyed
// while it traces.  In this mode every tick is followed by a stream synchronisation and graphs are only
// released with the process.
static bool under_profiler() {
  static const bool v = [] { const char* e = std::getenv("PIES_PROFILER_SAFE"); return e && e[0] == '1'; }();
  return v;
}

static void destroy_graph(pies_solver* s) {
  // graphs are not destroyed while a profiler is attached (rocprofv3 7.2 crashes on graph destruction)
  if (!under_profiler()) {
    if (!s->graphFromLadder) {
      if (s->graphExec) (void)hipGraphExecDestroy(s->graphExec);
      if (s->graph) (void)hipGraphDestroy(s->graph);
    }
    for (auto& kv : s->pdLadder) {
      if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
      if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    for (auto& ge : s->retiredGraphs) {
      (void)hipGraphExecDestroy(ge.second);
      (void)hipGraphDestroy(ge.first);
    }
    s->retiredGraphs.clear();
  }
  s->pdLadder.clear();
  s->graphFromLadder = false;
  s->graphExec = nullptr;
  s->graph = nullptr;
}

static void free_device(pies_solver* s) {
  destroy_graph(s);
  for (void* p : s->allocations) (void)hipFree(p);
  s->allocations.clear();
  s->nd = NodeArrays{nullptr, nullptr, nullptr, nullptr, 0};
  s->d_pc_id = nullptr; s->d_pc_tw = nullptr;
  s->d_dc_ids = nullptr; s->d_dc_rw = nullptr;
  s->d_tc_ids = nullptr; s->d_tc_q0 = s->d_tc_q1 = s->d_tc_q2 = nullptr;
  s->d_bc_ids = nullptr; s->d_bc_aw = nullptr;
  s->d_vc_ids = nullptr; s->d_vc_q0 = s->d_vc_q1 = s->d_vc_q2 = nullptr;
  s->pd = PdArrays{};
  s->hash = HashArrays{};
  s->pairs = PairArrays{};
  s->d_pairDictIndex = nullptr;
  s->d_pairDictTable = nullptr;
  s->pairDictSets = 0;
  s->pdRowStencils = 0;
  s->d_layer = LayerDevice{};
  s->snapPos = s->snapPrev = s->snapVel = nullptr;
  s->snapQuat = nullptr;
}

static int upload_nodes(pies_solver* s) {
  const uint32_t n = s->nodeCount();
  std::vector<float4> pos(n), prev(n), vel(n);
  for (uint32_t i = 0; i < n; ++i) {
    pos[i] = make_float4(s->h_pos[3 * i], s->h_pos[3 * i + 1], s->h_pos[3 * i + 2], s->h_invMass[i]);
    prev[i] = make_float4(s->h_prev[3 * i], s->h_prev[3 * i + 1], s->h_prev[3 * i + 2], 0.f);
    vel[i] = make_float4(s->h_vel[3 * i], s->h_vel[3 * i + 1], s->h_vel[3 * i + 2], 0.f);
  }
  if (n) {
    HIP_TRY(s, hipMemcpyAsync(s->nd.pos, pos.data(), n * sizeof(float4), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(s, hipMemcpyAsync(s->nd.prev, prev.data(), n * sizeof(float4), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(s, hipMemcpyAsync(s->nd.vel, vel.data(), n * sizeof(float4), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(s, hipMemcpyAsync(s->nd.radius, s->h_radius.data(), n * sizeof(float), hipMemcpyHostToDevice, s->stream));
    std::vector<float> lrad;
    if (s->d_layer.lrad && s->layer.nodeList.size() == n) {  // schedule LAYERED keeps the radii in level order as well
      lrad.resize(n);
      for (uint32_t i = 0; i < n; ++i) lrad[i] = s->h_radius[s->layer.nodeList[i]];
      HIP_TRY(s, hipMemcpyAsync(s->d_layer.lrad, lrad.data(), n * sizeof(float), hipMemcpyHostToDevice, s->stream));
    }
    HIP_TRY(s, hipStreamSynchronize(s->stream));  // the staging vectors die with this scope
  }
  s->hostNodesDirty = false;
  s->stale = 0;
  return PIES_OK;
}

// Host mirror <- HBM: the arrays of `mask` (bit 0 positions, 1 previous positions, 2 velocities) that are stale, one
// copy each through the pinned staging buffer.
static int download_nodes(pies_solver* s, uint32_t mask = 7u) {
  const uint32_t n = s->nd.n;
  mask &= s->stale;
  if (n == 0 || !s->h_stage) { s->stale = 0; return PIES_OK; }
  float* dst[3] = {s->h_pos.data(), s->h_prev.data(), s->h_vel.data()};
  const float4* src[3] = {s->nd.pos, s->nd.prev, s->nd.vel};
  for (int a = 0; a < 3; ++a) {
    if (!(mask & (1u << a))) continue;
    HIP_TRY(s, hipMemcpyAsync(s->h_stage, src[a], n * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    for (uint32_t i = 0; i < n; ++i) {
      dst[a][3 * i] = s->h_stage[i].x;
      dst[a][3 * i + 1] = s->h_stage[i].y;
      dst[a][3 * i + 2] = s->h_stage[i].z;
    }
    s->stale &= ~(1u << a);
  }
  return PIES_OK;
}

int scene_sync_host(pies_solver* s) {
  if (s->device == PIES_DEVICE_NONE) return PIES_OK;
  if (hipSetDevice(s->device) != hipSuccess) return fail(s, PIES_ERR_HIP, "hipSetDevice failed");
  return download_nodes(s);
}

// Schedule LAYERED: the substep as a list of layer launches (tiles of one phase, LDS resident), collision passes and
// - for bodies cut into strips - per-node launches over the level-ordered copy.  Walks tickPBD's order
// (Solver.cpp:45-159) and packs consecutive steps that run on the same phase into one launch: the distance container
// runs its phases in the order (0,1,2,3), the tetrahedral one (3,2,1,0), bend (0,1,2,3).  With one strip only phases
// 0 (even levels) and 2 (odd levels) exist and both cover every node, so the per-node steps ride along and an
// iteration without bend constraints or collisions is two launches: [tet even of the previous iteration, floor clamp,
// position, distance even] and [distance odd, tet odd].
enum { ITEM_LAYER = 0, ITEM_COLLIDE, ITEM_LPREDICT, ITEM_LVELOCITY, ITEM_LFLOOR, ITEM_LPOSITION, ITEM_TO_NODES, ITEM_FROM_NODES };
struct LayerItem {
  int type = ITEM_LAYER;
  LayerLaunch launch{};
  uint64_t bytes = 0;  // algorithmic bytes of the launch (SURVEY 8d per-unit figures)
  Batch batch{0, 0};   // ITEM_LPOSITION
};
static void build_layer_program(const pies_solver* s, std::vector<LayerItem>& prog) {
  const LayerPlan& L = s->layer;
  const bool oneStrip = L.strips == 1;
  constexpr uint32_t kCollide = 0xFFFFFFFFu;
  struct Step { uint32_t kind; int phase; int container; };  // phase -1: a per-node step
  std::vector<Step> steps;
  steps.push_back({LAYER_PREDICT, -1, -1});
  for (uint32_t it = 0; it < s->opt.iterations; ++it) {
    if (!s->releaseHinge && !s->h_position.empty()) steps.push_back({LAYER_POSITION, oneStrip ? 0 : -1, PIES_POSITION});
    const int cont[3] = {PIES_DISTANCE, PIES_TET, PIES_BEND};
    const uint32_t lk[3] = {LAYER_DISTANCE, LAYER_TET, LAYER_BEND};
    for (int c = 0; c < 3; ++c)
      for (int k = 0; k < 4; ++k) {
        const int ph = kLayerPhaseOrder[cont[c]][k];
        if (L.kind[cont[c]].ncol[ph]) steps.push_back({lk[c], ph, cont[c]});
      }
    if (s->nodeCollisions) steps.push_back({kCollide, -1, -1});
    steps.push_back({LAYER_FLOOR, -1, -1});
  }
  steps.push_back({LAYER_VELOCITY, -1, -1});

  const uint64_t N = s->nd.n;
  LayerItem cur;
  bool open = false;
  auto flush = [&] { if (open) prog.push_back(cur); open = false; cur = LayerItem{}; };
  auto plain = [&](int type) { LayerItem it; it.type = type; prog.push_back(it); };
  for (size_t i = 0; i < steps.size(); ++i) {
    const Step& st = steps[i];
    if (st.kind == kCollide) {  // the collision pass works on the node array
      flush();
      if (!oneStrip) plain(ITEM_TO_NODES);
      plain(ITEM_COLLIDE);
      if (!oneStrip) plain(ITEM_FROM_NODES);
      continue;
    }
    if (st.phase < 0 && !oneStrip) {  // strips: no phase's tiles cover every node exactly once
      flush();
      if (st.kind == LAYER_POSITION) {
        for (const Batch& b : s->plan[PIES_POSITION].batches) { LayerItem it; it.type = ITEM_LPOSITION; it.batch = b; it.bytes = 44ull * b.count; prog.push_back(it); }
      } else {
        LayerItem it;
        it.type = st.kind == LAYER_PREDICT ? ITEM_LPREDICT : st.kind == LAYER_FLOOR ? ITEM_LFLOOR : ITEM_LVELOCITY;
        it.bytes = (st.kind == LAYER_PREDICT ? 48u : st.kind == LAYER_FLOOR ? 20u : 40u) * N;
        prog.push_back(it);
      }
      continue;
    }
    int q = st.phase;
    if (open && (q < 0 || q == (int)cur.launch.phase) && cur.launch.nseg < (uint32_t)kLayerMaxSegs) {
      q = cur.launch.phase;
    } else {
      flush();
      if (q < 0) {  // a per-node step opens a launch: take the phase of the next container step so that it can join
        q = 0;
        for (size_t j = i + 1; j < steps.size(); ++j) {
          if (steps[j].kind == kCollide) break;
          if (steps[j].phase >= 0) { q = steps[j].phase; break; }
        }
      }
      open = true;
      cur.type = ITEM_LAYER;
      cur.launch.phase = (uint32_t)q;
      cur.launch.groups = (uint32_t)L.tiles[q].size();
      cur.launch.maxClass = 1;
    }
    LayerSeg& seg = cur.launch.seg[cur.launch.nseg++];
    seg.kind = st.kind;
    seg.ncol = 0;
    seg.colOff = nullptr;
    if (st.container >= 0) {
      const LayerKind& K = L.kind[st.container];
      seg.ncol = K.ncol[q];
      seg.colOff = s->d_layer.colOff[st.container][q];
      cur.launch.maxClass = std::max(cur.launch.maxClass, K.maxClass);
      const uint64_t count = K.colOff[q].empty() ? 0 : K.colOff[q].back() - K.colOff[q].front();
      const uint64_t perUnit = st.container == PIES_POSITION ? 44 : st.container == PIES_DISTANCE ? 52 : st.container == PIES_TET ? 160 : 136;
      cur.bytes += perUnit * count;
    } else {
      cur.bytes += (st.kind == LAYER_PREDICT ? 48u : st.kind == LAYER_FLOOR ? 20u : 40u) * N;
    }
  }
  flush();
  // One strip: the node array is the source of the first launch and of every launch after a collision pass, and the
  // destination of the last launch and of every launch before a collision pass.  Strips: always the copy.
  for (size_t i = 0; i < prog.size(); ++i) {
    if (prog[i].type != ITEM_LAYER) continue;
    prog[i].launch.loadGlobal = oneStrip && (i == 0 || prog[i - 1].type == ITEM_COLLIDE) ? 1u : 0u;
    prog[i].launch.storeGlobal = oneStrip && (i + 1 == prog.size() || prog[i + 1].type == ITEM_COLLIDE) ? 1u : 0u;
  }
}

// Solver.cpp:85-130 in the reference's order (schedule EXACT, the flag, or ranges wider than two cells) or in the
// parallel visiting order of DESIGN.md section 6
// (cell, node) entries of the collision grid: NodeCompRange spans ceil(fract + 2R) <= 1 + ceil(2R) cells per axis (an
// over-long range is empty, Solver.cpp:896-898), which only depends on the radius, so the sum bounds any positions.  8
// per node for the reference's defaults.  fast: every range spans at most 2 cells per axis (2R <= 1).
static void collision_grid_bound(const pies_solver* s, uint64_t& entries, bool& fast) {
  entries = 0;
  fast = true;
  for (float r : s->h_radius) {
    const float twoR = 2.0f * ((r + 0.5f) / s->opt.gridSpacing);
    const uint64_t len = std::isfinite(twoR) && twoR >= 0.0f && twoR < 64.0f ? std::min<uint64_t>(50, 1 + static_cast<uint64_t>(std::ceil(twoR))) : 0;
    entries += len * len * len;
    if (!(twoR <= 1.0f)) fast = false;
  }
}
static void probe_mark(pies_solver* s, int k) {
  if (s->probe && s->probe->kernel == k) s->probe->mark();
}
// PIES_COLLISION_ORDER_*: the reference's loop for ranges wider than two cells per axis (the parallel orders need 2R <= 1),
// under schedule EXACT, or when asked for; otherwise the pair order (or, when asked for, the group order of rounds 1-2)
static int collision_order(const pies_solver* s) {
  int order = s->collisionOrderFlag >= 0 ? s->collisionOrderFlag
                                          : (s->schedule == PIES_SCHEDULE_EXACT ? PIES_COLLISION_ORDER_REFERENCE : PIES_COLLISION_ORDER_PAIRS);
  // the group order needs ranges of at most two cells per axis (2R <= 1); the pair order lists node by node beyond that
  if (order == PIES_COLLISION_ORDER_GROUPS && !s->collideFast) order = PIES_COLLISION_ORDER_REFERENCE;
  return order;
}
static bool needs_grid_groups(const pies_solver* s) { return collision_order(s) != PIES_COLLISION_ORDER_PAIRS || !s->collideFast; }
static uint32_t enqueue_collide(pies_solver* s, bool rearm = false) {
  switch (collision_order(s)) {
    case PIES_COLLISION_ORDER_REFERENCE:
      return launch_collide_reference(s->stream, s->hash, s->nd, s->opt.gridSpacing, s->opt.friction, s->opt.staticFrictionThreshold);
    case PIES_COLLISION_ORDER_GROUPS:
      return launch_collide(s->stream, s->hash, s->nd, s->opt.friction, s->opt.staticFrictionThreshold, rearm);
    default:
      return launch_collide_pairs(s->stream, s->hash, s->pairs, s->nd, s->opt.friction, s->opt.staticFrictionThreshold, s->pairRounds);
  }
}

static void enqueue_layered_substep(pies_solver* s, int only, uint32_t* counts, uint64_t* units) {
  hipStream_t st = s->stream;
  std::vector<LayerItem> prog;
  build_layer_program(s, prog);
  const LayerDevice& d = s->d_layer;
  LayerData D{};
  D.nodeList = d.nodeList;
  for (int ph = 0; ph < 4; ++ph) D.tiles[ph] = reinterpret_cast<const uint4*>(d.tiles[ph]);
  D.maxGroupNodes = s->layer.maxGroupNodes;
  D.lpos = d.lpos; D.lrad = d.lrad;
  D.pc_lid = d.pc_lid; D.pc_tw = s->d_pc_tw;
  D.dc_lid = d.dc_lid; D.dc_rw = s->d_dc_rw;
  D.tc_lid = d.tc_lid; D.tc_q0 = s->d_tc_q0; D.tc_q1 = s->d_tc_q1; D.tc_q2 = s->d_tc_q2;
  D.bc_lid = d.bc_lid; D.bc_aw = s->d_bc_aw;
  const float dt = s->opt.fixedTimestepSize / s->opt.timeSubsteps;
  const LayerParams P = {s->opt.floorHeight, dt, s->opt.gravity, s->opt.damping, s->opt.friction};
  int cur = -1;  // class of the launch being made: with a probe attached only that class's units are tallied
  auto ON = [&](int k) { const bool on = only < 0 || only == k; cur = k; if (on) probe_mark(s, k); return on; };
  auto C = [&](int k) { probe_mark(s, k); if (counts) ++counts[k]; };
  auto U = [&](uint64_t u) { if (units && (only >= 0 || (s->probe && s->probe->kernel == cur))) *units += u; };
  for (const LayerItem& item : prog) {
    switch (item.type) {
      case ITEM_COLLIDE: {  // Solver.cpp:81-130
        uint32_t nb = 34, nc = 1;
        if (ON(PIES_KERNEL_HASH)) { nb = launch_hash_build(st, s->hash, s->nd, s->opt.gridSpacing, s->sortPasses, needs_grid_groups(s)); U(s->nd.n); }
        probe_mark(s, PIES_KERNEL_HASH);
        if (ON(PIES_KERNEL_COLLIDE)) { nc = enqueue_collide(s, only == PIES_KERNEL_COLLIDE); U(s->nd.n); }
        probe_mark(s, PIES_KERNEL_COLLIDE);
        if (counts) { counts[PIES_KERNEL_HASH] += nb; counts[PIES_KERNEL_COLLIDE] += nc; }
        break;
      }
      case ITEM_LAYER:
        if (ON(PIES_KERNEL_LAYER)) { launch_layer(st, s->nd, D, item.launch, P); U(item.bytes); }
        C(PIES_KERNEL_LAYER);
        break;
      case ITEM_LPREDICT:
        if (ON(PIES_KERNEL_PREDICT)) { launch_lpredict(st, s->nd, D, P); U(s->nd.n); }
        C(PIES_KERNEL_PREDICT);
        break;
      case ITEM_LVELOCITY:
        if (ON(PIES_KERNEL_VELOCITY)) { launch_lvelocity(st, s->nd, D, P); U(s->nd.n); }
        C(PIES_KERNEL_VELOCITY);
        break;
      case ITEM_LFLOOR:
        if (ON(PIES_KERNEL_FLOOR)) { launch_lfloor(st, s->nd, D, P); U(s->nd.n); }
        C(PIES_KERNEL_FLOOR);
        break;
      case ITEM_LPOSITION:
        if (ON(PIES_KERNEL_POSITION)) { launch_lposition(st, D, item.batch.start, item.batch.count); U(item.batch.count); }
        C(PIES_KERNEL_POSITION);
        break;
      case ITEM_TO_NODES:
      case ITEM_FROM_NODES:
        if (only < 0) launch_lcopy(st, s->nd, D, item.type == ITEM_TO_NODES);
        break;
    }
  }
}

// One PBD substep as a launch sequence (Solver.cpp:45-159).  `timer`/`timedKernel` select one kernel
// class for per-dispatch timing (profile pass); counts (optional) tallies launches per class.
static void enqueue_pbd_substep(pies_solver* s, int only, uint32_t* counts, uint64_t* units = nullptr) {
  hipStream_t st = s->stream;
  const float dt = s->opt.fixedTimestepSize / s->opt.timeSubsteps;
  int cur = -1;
  auto C = [&](int k) { probe_mark(s, k); if (counts) ++counts[k]; };
  auto ON = [&](int k) { const bool on = only < 0 || only == k; cur = k; if (on) probe_mark(s, k); return on; };  // profile pass: one kernel class only
  auto U = [&](uint64_t u) { if (units && (only >= 0 || (s->probe && s->probe->kernel == cur))) *units += u; };

  if (s->layer.active) { enqueue_layered_substep(s, only, counts, units); return; }
  if (ON(PIES_KERNEL_PREDICT)) { launch_predict(st, s->nd, dt, s->opt.gravity); U(s->nd.n); }
  C(PIES_KERNEL_PREDICT);
  if (s->wave.active) {  // schedule EXACT: the levels of the whole-substep DAG, cut by the collision passes
    const WaveData W = {s->d_pc_id, s->d_pc_tw, s->d_dc_ids, s->d_dc_rw, s->d_tc_ids, s->d_tc_q0, s->d_tc_q1, s->d_tc_q2, s->d_bc_ids, s->d_bc_aw};
    size_t barrier = 0;
    auto collide = [&] {
      uint32_t nb = 34, nc = 1;
      if (ON(PIES_KERNEL_HASH)) { nb = launch_hash_build(st, s->hash, s->nd, s->opt.gridSpacing, s->sortPasses, needs_grid_groups(s)); U(s->nd.n); }
      probe_mark(s, PIES_KERNEL_HASH);
      if (ON(PIES_KERNEL_COLLIDE)) { nc = enqueue_collide(s, only == PIES_KERNEL_COLLIDE); U(s->nd.n); }
      probe_mark(s, PIES_KERNEL_COLLIDE);
      if (counts) { counts[PIES_KERNEL_HASH] += nb; counts[PIES_KERNEL_COLLIDE] += nc; }
    };
    for (size_t l = 0; l < s->wave.levels.size(); ++l) {
      for (; barrier < s->wave.barrierAfter.size() && s->wave.barrierAfter[barrier] == l; ++barrier) collide();
      const WaveLevel& L = s->wave.levels[l];
      if (ON(PIES_KERNEL_WAVE)) {
        launch_wave(st, s->nd, s->opt.floorHeight, s->d_waveIndex, L, W);
        U((uint64_t)L.cnt[0] + L.cnt[1] + L.cnt[2] + L.cnt[3] + L.cnt[4]);
      }
      C(PIES_KERNEL_WAVE);
    }
    for (; barrier < s->wave.barrierAfter.size(); ++barrier) collide();
  }
  for (uint32_t it = 0; it < (s->wave.active ? 0u : s->opt.iterations); ++it) {  // one launch per batch, sweep after sweep
    if (!s->releaseHinge)
      for (const Batch& b : s->plan[PIES_POSITION].batches) {
        if (ON(PIES_KERNEL_POSITION)) { launch_position(st, s->nd.pos, s->d_pc_id, s->d_pc_tw, b.start, b.count); U(b.count); }
        C(PIES_KERNEL_POSITION);
      }
    for (const Batch& b : s->plan[PIES_DISTANCE].batches) {
      if (ON(PIES_KERNEL_DISTANCE)) { launch_distance(st, s->nd.pos, s->d_dc_ids, s->d_dc_rw, b.start, b.count); U(b.count); }
      C(PIES_KERNEL_DISTANCE);
    }
    for (const Batch& b : s->plan[PIES_TET].batches) {
      if (ON(PIES_KERNEL_TET)) { launch_tet(st, s->nd.pos, s->d_tc_ids, s->d_tc_q0, s->d_tc_q1, s->d_tc_q2, b.start, b.count); U(b.count); }
      C(PIES_KERNEL_TET);
    }
    for (const Batch& b : s->plan[PIES_BEND].batches) {
      if (ON(PIES_KERNEL_BEND)) { launch_bend(st, s->nd.pos, s->d_bc_ids, s->d_bc_aw, b.start, b.count); U(b.count); }
      C(PIES_KERNEL_BEND);
    }
    if (s->nodeCollisions) {  // Solver.cpp:81-130
      uint32_t nb = 34, nc = 1;
      if (ON(PIES_KERNEL_HASH)) { nb = launch_hash_build(st, s->hash, s->nd, s->opt.gridSpacing, s->sortPasses, needs_grid_groups(s)); U(s->nd.n); }
      probe_mark(s, PIES_KERNEL_HASH);
      if (ON(PIES_KERNEL_COLLIDE)) { nc = enqueue_collide(s, only == PIES_KERNEL_COLLIDE); U(s->nd.n); }
      probe_mark(s, PIES_KERNEL_COLLIDE);
      if (counts) { counts[PIES_KERNEL_HASH] += nb; counts[PIES_KERNEL_COLLIDE] += nc; }
    }
    if (ON(PIES_KERNEL_FLOOR)) { launch_floor(st, s->nd, s->opt.floorHeight); U(s->nd.n); }
    C(PIES_KERNEL_FLOOR);
  }
  if (ON(PIES_KERNEL_VELOCITY)) { launch_velocity(st, s->nd, dt, s->opt.damping, s->opt.friction, s->opt.floorHeight); U(s->nd.n); }
  C(PIES_KERNEL_VELOCITY);
}

// The global step's CG with one launch per iteration (pd_cg1_kernels.hip): the contact-light graph variant with one lane per
// matrix row; the contact-heavy variant (contact rows summed by extra workgroups) keeps the two-launch form.
static bool pd_single_cg(const pies_solver* s) {
  // (the contact-heavy variant: the rows' contact parts are the merged rows k_contact_csr builds every substep - a gather of a
  // handful of distinct columns by the row's lane; PIES_PD_CG_SINGLE_ROWS=0 keeps the two-launch form with its extra workgroups there)
  return s->pdSingleCg && (!s->pd.cg.useCAp || s->pdSingleCgRows) && s->pd.cg.lanesPerRow == 1u;
}

// One PD substep as a launch sequence (Solver.cpp:228-485).  `only` >= 0 (profile pass) launches one kernel
// class of the tetrahedral pipeline; units tallies the work items of the launches made.
static void enqueue_pd_substep(pies_solver* s, int only = -1, uint32_t* counts = nullptr, uint64_t* units = nullptr) {
  hipStream_t st = s->stream;
  const float h = s->opt.fixedTimestepSize / s->opt.timeSubsteps;
  const PdArrays& pd = s->pd;
  int cur = -1;
  auto ON = [&](int k) { const bool on = only < 0 || only == k; cur = k; if (on) probe_mark(s, k); return on; };
  auto C = [&](int k, uint32_t n = 1) { if (k != PIES_KERNEL_PD_SPMV && k != PIES_KERNEL_PD_CG_UPDATE) probe_mark(s, k); if (counts) counts[k] += n; };
  auto U = [&](uint64_t u) { if (units && (only >= 0 || (s->probe && s->probe->kernel == cur))) *units += u; };
  const uint32_t nDist = (uint32_t)s->h_distance.size(), nTet = (uint32_t)s->h_tet.size(), nVol = (uint32_t)s->h_volume.size();
  if (ON(PIES_KERNEL_PD_PREDICT)) { launch_pd_predict(st, s->nd, pd, h, s->opt.floorHeight + s->opt.collisionThickness); U(s->nd.n); }
  C(PIES_KERNEL_PD_PREDICT);
  const bool tri = pd.tri.nt != 0;
  // the statistics of the substep's last solve are closed by an extra workgroup of the floor-snap launch when there is one
  const bool statsInStabilize = only < 0 && s->opt.collisionStabilizationIterations > 0 && s->opt.iterations > 0 && s->nd.n != 0;
  if (tri && only < 0) {  // Solver.cpp:240, 245-248: detection, contact list, their blocks of the system matrix
    launch_tri_detect(st, pd.tri, s->nd, pd.kdiag, pd.cg.cdiag, pd.cg.dinv, s->opt.collisionThresholdDistance, s->opt.collisionThickness,
                      pd.cg.useCAp != 0);
    // The dependency levels of the list (one workgroup, up to 1 ms with tens of thousands of contacts) are only needed by
    // the sequential passes behind the local/global iterations: a second branch of the substep, joined there.
    // Only in the contact-heavy graph variant: a fork and join inside a hipGraph costs about 100 us per replay (measured:
    // config 3, no contact, 1 257 -> 1 120 substeps/s with the branch; 29k contacts, 197 -> 234 with it).
    const char* e = tuning_env("PIES_TRI_SIDE");  // diagnostics: 0 = always in line, 1 = always beside
    s->triLevelsForked = e ? e[0] != '0' : s->triFastRows;
    if (s->triLevelsForked) {
      (void)hipEventRecord(s->evFork, st);
      (void)hipStreamWaitEvent(s->sideStream, s->evFork, 0);
      launch_tri_levels(s->sideStream, pd.tri);
      (void)hipEventRecord(s->evJoin, s->sideStream);
    } else {
      launch_tri_levels(st, pd.tri);
    }
  }
  for (uint32_t it = 0; it < s->opt.iterations; ++it) {
    // local step (Solver.cpp:270-308): position constraints project to a constant, uploaded once
    if (nDist && ON(PIES_KERNEL_PD_LOCAL_DISTANCE)) {
      launch_pd_local_distance(st, s->nd.pos, s->d_dc_ids, s->d_dc_rw, pd.contrib + s->slotBase[PIES_DISTANCE], nDist);
      U(nDist);
    }
    if (nDist) C(PIES_KERNEL_PD_LOCAL_DISTANCE);
    if (s->tetVolumePaired && pd.tiles.ntiles) {  // tile-resident: one sum per (tile, node) leaves the chip (pd_tiles.cpp)
      if (ON(PIES_KERNEL_PD_LOCAL_TET)) {
        launch_pd_local_tiles(st, s->nd.pos, pd.tiles, s->d_pairDictTable, tri && only < 0 ? &pd.tri : nullptr, s->opt.collisionThickness);
        U(nTet);
      }
      C(PIES_KERNEL_PD_LOCAL_TET);
    } else if (s->tetVolumePaired) {  // both projections in one launch, accounted to the strain class
      if (ON(PIES_KERNEL_PD_LOCAL_TET)) {
        launch_pd_local_tet_pair(st, s->nd.pos, s->d_tc_ids, s->d_tc_q0, s->d_tc_q1, s->d_tc_q2, s->d_vc_q2,
                                 pd.contrib + s->slotBase[PIES_TET], pd.contrib + s->slotBase[PIES_VOLUME], nTet,
                                 tri && only < 0 ? &pd.tri : nullptr, s->opt.collisionThickness, s->pdLocalPacked,
                                 s->pdLocalPacked ? s->d_pairDictIndex : nullptr, s->d_pairDictTable);  // + the contacts' local step
        U(nTet);
      }
      C(PIES_KERNEL_PD_LOCAL_TET);
    } else {
    if (nTet && ON(PIES_KERNEL_PD_LOCAL_TET)) {
      launch_pd_local_tet(st, false, s->nd.pos, s->d_tc_ids, s->d_tc_q0, s->d_tc_q1, s->d_tc_q2, pd.contrib + s->slotBase[PIES_TET], nTet);
      U(nTet);
    }
    if (nTet) C(PIES_KERNEL_PD_LOCAL_TET);
    if (nVol && ON(PIES_KERNEL_PD_LOCAL_VOLUME)) {
      launch_pd_local_tet(st, true, s->nd.pos, s->d_vc_ids, s->d_vc_q0, s->d_vc_q1, s->d_vc_q2, pd.contrib + s->slotBase[PIES_VOLUME], nVol);
      U(nVol);
    }
    if (nVol) C(PIES_KERNEL_PD_LOCAL_VOLUME);
    }
    if (only < 0) {
      launch_pd_local_bend(st, s->nd.pos, s->d_bc_ids, s->d_bc_aw, pd.contrib + s->slotBase[PIES_BEND], (uint32_t)s->h_bend.size());
      launch_pd_local_shape(st, s->nd.pos, pd);                        // goal targets are constants between transform updates
      if (tri && !(s->tetVolumePaired && nTet)) launch_pd_local_tri(st, pd.tri, s->nd.pos, s->opt.collisionThickness);  // Solver.cpp:298-300
    }
    // Solver.cpp:266, 310-349.  (When a node's records are a few tile sums, the residual kernel of the one-launch-per-iteration
    // CG evaluates the right-hand side itself.)
    // (contact-heavy variant: a node's contact records are gathered by four lanes in k_pd_rhs, not by the residual kernel's one)
    const bool single = pd_single_cg(s), fuseRhs = single && only < 0 && pd.rhsLanes == 1 && s->pdFuseRhs && !pd.cg.useCAp;
    if (!fuseRhs) {
      if (ON(PIES_KERNEL_PD_RHS)) { launch_pd_rhs(st, s->nd, pd); U(s->nd.n); }
      C(PIES_KERNEL_PD_RHS);
    } else if (counts) {
      counts[PIES_KERNEL_PD_RHS] += 1;  // (the residual kernel that evaluates the right-hand side is counted - and bracketed - as this class)
    }
    const int overflow = s->pcgOverflow ? (int)(s->pcgMaxIters > s->pcgBudget ? s->pcgMaxIters - s->pcgBudget : 0u) : 0;
    const bool lastSolve = it + 1 == s->opt.iterations && !statsInStabilize;
    if (only < 0) {  // Solver.cpp:356-364
      const bool probed = s->probe && (s->probe->kernel == PIES_KERNEL_PD_SPMV || s->probe->kernel == PIES_KERNEL_PD_CG_UPDATE);
      // a probed solve never takes the converged early exit: every bracketed launch does a full SpMV / vector update
      auto hook = s->probe ? [](void* ctx, int cls) { probe_mark(static_cast<pies_solver*>(ctx), cls); } : (void (*)(void*, int))nullptr;
      if (fuseRhs && units && s->probe && s->probe->kernel == PIES_KERNEL_PD_RHS) *units += s->nd.n;
      if (single) launch_pd_solve1(st, s->nd, pd, (int)s->pcgBudget, s->pcgTol, it == 0, lastSolve, fuseRhs, probed, hook, s, overflow);
      else launch_pd_solve(st, s->nd, pd, (int)s->pcgBudget, s->pcgTol, -1, it == 0, lastSolve, probed, hook, s, overflow);
      if (probed && units) *units += (uint64_t)s->nd.n * s->pcgBudget;
    }
    else if (only == PIES_KERNEL_PD_SPMV) {
      if (single) launch_pd_solve1(st, s->nd, pd, (int)s->pcgBudget, s->pcgTol, true, false, false, true);
      else launch_pd_solve(st, s->nd, pd, (int)s->pcgBudget, 0.f, 1);
      U((uint64_t)s->nd.n * s->pcgBudget);
    }
    else if (only == PIES_KERNEL_PD_CG_UPDATE && !single) { launch_pd_solve(st, s->nd, pd, (int)s->pcgBudget, 0.f, 0); U((uint64_t)s->nd.n * s->pcgBudget); }
    C(PIES_KERNEL_PD_SPMV, s->pcgBudget);
    if (!single) C(PIES_KERNEL_PD_CG_UPDATE, s->pcgBudget);
  }
  if (tri && only < 0) {  // :367-383: every stabilisation iteration is a sequential pass over the contacts, then the floor snap
    if (s->triLevelsForked) (void)hipStreamWaitEvent(st, s->evJoin, 0);
    // all iterations in one launch (the floor snap of the contacts' nodes between the passes), then the snap of everybody else:
    // idempotent, so once is what the reference's `iterations` times come to
    launch_tri_stabilize(st, pd.tri, s->nd, s->opt.collisionThickness, pd.nstatic, pd.statp, s->opt.collisionStabilizationIterations);
    if (s->opt.collisionStabilizationIterations > 0) launch_pd_stabilize(st, s->nd, pd, statsInStabilize, (int)s->pcgBudget, s->pcgTol, pd_single_cg(s));
    // velocities, then the contacts' friction (:431-471), then the floor friction (:473-484).  The floor friction of a node that is
    // in no contact does not wait for the contacts: the velocity kernel applies it; the contacts' pass ends with that of its own nodes
    launch_pd_velocity(st, s->nd, pd, h, s->opt.damping, s->opt.gravity, s->opt.friction, s->opt.staticFrictionThreshold, false, pd.tri.usedBits);
    launch_tri_friction(st, pd.tri, s->nd, s->opt.friction, s->opt.staticFrictionThreshold, pd.nstatic);
  } else {
    if (only < 0 && s->opt.collisionStabilizationIterations > 0) launch_pd_stabilize(st, s->nd, pd, statsInStabilize, (int)s->pcgBudget, s->pcgTol, pd_single_cg(s));  // the floor snap is idempotent
    if (ON(PIES_KERNEL_PD_VELOCITY)) {
      launch_pd_velocity(st, s->nd, pd, h, s->opt.damping, s->opt.gravity, s->opt.friction, s->opt.staticFrictionThreshold, true);
      U(s->nd.n);
    }
  }
  C(PIES_KERNEL_PD_VELOCITY);
}

static void enqueue_substep(pies_solver* s, uint32_t* counts) {
  if (s->opt.solver == PIES_SOLVER_PD) enqueue_pd_substep(s, -1, counts);
  else enqueue_pbd_substep(s, -1, counts);
}

// The budget ladder: captured CG iterations per solve of the graphs instantiated together - 1, 2, 3, 4, 6, 8, 12, 16, ... up to
// the ceiling of pies_set_pcg (the one-launch-per-iteration form pays one launch per unused iteration, so the low rungs are
// close together); ladder_rung = the rung that holds `budget` iterations
static std::vector<uint32_t> ladder_rungs(const pies_solver* s) {
  std::vector<uint32_t> r;
  if (s->pcgMaxIters > 1) r.push_back(1);  // (a body at rest: the solves end at their first look at the residual)
  for (uint32_t b = 2; b < s->pcgMaxIters; b *= 2) {
    r.push_back(b);
    if (b + b / 2 < s->pcgMaxIters) r.push_back(b + b / 2);
  }
  r.push_back(s->pcgMaxIters);
  return r;
}
static uint32_t ladder_rung(const pies_solver* s, uint32_t budget) {
  for (uint32_t r : ladder_rungs(s))
    if (r >= budget) return r;
  return s->pcgMaxIters;
}
static bool uses_ladder(const pies_solver* s) {
  if (s->opt.solver != PIES_SOLVER_PD || s->pcgPinned || under_profiler()) return false;
  const char* e = tuning_env("PIES_NO_GRAPH");
  return !(e && e[0] == '1');
}
// makes the ladder entry of (pcgBudget, triFastRows) the graph pies_tick launches
static int select_pd_graph(pies_solver* s) {
  s->pcgBudget = ladder_rung(s, s->pcgBudget);
  s->pd.cg.useCAp = s->triFastRows ? 1 : 0;
  auto it = s->pdLadder.find(static_cast<uint64_t>(s->pcgBudget) | (static_cast<uint64_t>(s->triFastRows ? 1 : 0) << 32));
  if (it == s->pdLadder.end()) return fail(s, PIES_ERR_STATE, "no captured graph for this CG budget");
  s->graph = it->second.graph;
  s->graphExec = it->second.exec;
  s->graphFromLadder = true;
  { const char* e = tuning_env("PIES_TRI_SIDE"); s->triLevelsForked = e ? e[0] != '0' : s->triFastRows; }  // (as the entry was captured)
  std::memcpy(s->launchCounts, it->second.counts, sizeof(s->launchCounts));
  return PIES_OK;
}

static int capture_graph(pies_solver* s) {
  destroy_graph(s);
  std::memset(s->launchCounts, 0, sizeof(s->launchCounts));
  if (s->nd.n == 0) return PIES_OK;
  if (const char* e = tuning_env("PIES_NO_GRAPH"); e && e[0] == '1') {
    // count launches without running them: a capture that is thrown away
    HIP_TRY(s, hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    enqueue_substep(s, s->launchCounts);
    hipGraph_t tmp = nullptr;
    HIP_TRY(s, hipStreamEndCapture(s->stream, &tmp));
    if (tmp) (void)hipGraphDestroy(tmp);
    return PIES_OK;
  }
  if (uses_ladder(s)) {
    // every rung, both contact-row variants when the scene has surface triangles: captured and instantiated now, so that
    // following the solves later never instantiates a graph in the middle of a frame
    const uint32_t wantBudget = s->pcgBudget;
    const bool wantRows = s->triFastRows;
    const std::vector<uint32_t> rungs = ladder_rungs(s);
    const int variants = s->pd.tri.nt ? 2 : 1;
    for (int v = 0; v < variants; ++v)
      for (uint32_t r : rungs) {
        const bool rows = variants == 2 ? v != 0 : wantRows;
        s->pcgBudget = r;
        s->triFastRows = rows;
        s->pd.cg.useCAp = rows ? 1 : 0;
        pies_solver::PdGraph g;
        HIP_TRY(s, hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
        enqueue_substep(s, g.counts);
        hipError_t e = hipStreamEndCapture(s->stream, &g.graph);
        if (e != hipSuccess) return fail(s, PIES_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        HIP_TRY(s, hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
        s->pdLadder[static_cast<uint64_t>(r) | (static_cast<uint64_t>(rows ? 1 : 0) << 32)] = g;
      }
    s->pcgBudget = wantBudget;
    s->triFastRows = wantRows;
    return select_pd_graph(s);
  }
  HIP_TRY(s, hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
  enqueue_substep(s, s->launchCounts);
  hipError_t e = hipStreamEndCapture(s->stream, &s->graph);
  if (e != hipSuccess) return fail(s, PIES_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
  HIP_TRY(s, hipGraphInstantiate(&s->graphExec, s->graph, nullptr, nullptr, 0));
  return PIES_OK;
}

// The graph holds a fixed number of CG iterations per solve (converged solves early-exit the rest).  At
// every host synchronisation the budget follows what the solves needed: it starts at 32, shrinks to (most
// iterations used over the last 8+ synchronisations) + a third of that (at least 2), + a quarter (at least 1) after 24,
// and quadruples (at least 32, at most pcgMaxIters = 128 by default) when a solve ran out of iterations above the tolerance.
static int adapt_pcg_budget(pies_solver* s) {
  s->asyncSinceSync = 0;  // (called right after a host synchronisation)
  if (s->opt.solver != PIES_SOLVER_PD || !s->pd.cg.stats || !s->graphExec || s->sceneDirty || under_profiler()) return PIES_OK;
  if (s->pcgPinned) return PIES_OK;  // PIES_PCG_BUDGET: tests of the overflow path keep the captured budget where they put it
  float st[4] = {0, 0, 0, 0};
  HIP_TRY(s, hipMemcpyAsync(st, s->pd.cg.stats, sizeof(st), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  if (st[2] == 0.0f) return PIES_OK;  // no solve since the statistics were reset
  const uint32_t used = static_cast<uint32_t>(st[1]);
  const bool converged = st[0] <= s->pcgTol * s->pcgTol;
  uint32_t budget = s->pcgBudget;
  if (!converged && budget < s->pcgMaxIters) {
    // a solve ran out of iterations above the tolerance (new contacts stiffen the system at once): back to the full
    // budget now - the reference's solve is a direct one - and no shrinking for the next 24 synchronisations (a converged
    // solve's unused captured iterations return on one flag word: a generous budget costs 2.5 us per unused launch)
    budget = std::min(s->pcgMaxIters, std::max(32u, 4u * budget));  // 2..8 -> 32 -> 128: at most two short substeps
    s->pcgCalm = 0;
    s->pcgWindowMax = 0;
    s->pcgCooldown = 24;
  } else if (used > budget || (pd_single_cg(s) && used == budget && budget < s->pcgMaxIters)) {
    // converged, but only because the last launch went on by itself (k_cg_update's continuation): capture what it needed
    budget = std::min(s->pcgMaxIters, used + std::max(pd_single_cg(s) ? 1u : 2u, (used + 2u) / 3u));
    s->pcgCalm = 0;
    s->pcgWindowMax = 0;
    s->pcgCooldown = 8;
  } else if (s->pcgCooldown > 0) {
    --s->pcgCooldown;
  } else if (converged) {
    // the most iterations any solve used over the window of synchronisations since the last change
    s->pcgWindowMax = std::max(s->pcgWindowMax, used);
    ++s->pcgCalm;
    bool restart = false;
    // spare iterations on top of the most any solve of the window used: a third of it (at least two) after 8 calm
    // synchronisations, a quarter (at least one) after 24.  (Round 2 took + 2 / + 1 flat: a contact patch whose solves use
    // 5-8 iterations then sat at 8 and ran short on the next fluctuation - back to 32 for 60 frames.)
    // (the one-launch-per-iteration form needs one launch beyond the iterations a solve uses - the one that finds it converged)
    const uint32_t spare8 = std::max(pd_single_cg(s) ? 1u : 2u, (s->pcgWindowMax + 2u) / 3u), spare24 = std::max(1u, (s->pcgWindowMax + 3u) / 4u);
    if (s->pcgCalm >= 8 && s->pcgWindowMax + spare8 < budget) { budget = s->pcgWindowMax + spare8; restart = true; }
    else if (s->pcgCalm >= 24) {
      if (s->pcgWindowMax + spare24 < budget) budget = s->pcgWindowMax + spare24;
      restart = true;  // the window never looks back further than 24 synchronisations
    }
    if (restart) { s->pcgCalm = 0; s->pcgWindowMax = 0; }
    // (every spare iteration is two launches per local/global iteration that exit at once)
  }
  // Graph variant for contact-heavy substeps: the contact rows of the SpMV get a pass of their own (k_contact_rows, one
  // wavefront per node).  Either variant is correct with any number of contacts; the switch only follows what the last
  // substep saw (on at 512 contacts, off after 120 synchronisations without any).
  bool fastRows = s->triFastRows;
  if (s->pd.tri.nt && s->pd.tri.counters) {
    const int force = [] { const char* e = tuning_env("PIES_TRI_FAST_ROWS"); return e ? std::atoi(e) : -1; }();
    uint32_t contacts = 0;
    HIP_TRY(s, hipMemcpyAsync(&contacts, s->pd.tri.counters + 2, sizeof(contacts), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    if (contacts >= 512) { fastRows = true; s->triQuiet = 0; }
    else if (contacts == 0 && fastRows && ++s->triQuiet >= 120) { fastRows = false; s->triQuiet = 0; }
    else if (contacts != 0) s->triQuiet = 0;
    if (force >= 0) fastRows = force != 0;
  }
  if (const char* e = std::getenv("PIES_PCG_DEBUG"); e && e[0] == '1')
    std::fprintf(stderr, "[pies] pcg: residual^2 %.3g (%s) iterations %u budget %u -> %u calm %u cooldown %u contact rows %s\n", st[0],
                 converged ? "ok" : "short", used, s->pcgBudget, budget, s->pcgCalm, s->pcgCooldown, fastRows ? "pass" : "inline");
  if (!s->pdLadder.empty()) budget = ladder_rung(s, budget);
  if (budget != s->pcgBudget || fastRows != s->triFastRows) {
    s->pcgBudget = budget;
    s->triFastRows = fastRows;
    s->pd.cg.useCAp = fastRows ? 1 : 0;
    return s->pdLadder.empty() ? capture_graph(s) : select_pd_graph(s);  // (another executable graph of the ladder: no capture)
  }
  return PIES_OK;
}

// Pair-ordered node-node pass: the captured level launches follow what the passes need, at host synchronisations (a launch
// that finds nothing to do costs 2.5 us, and each pass captures the launches twice - once for its repeat).  Deeper orders than
// captured are finished by the single-workgroup tail kernel, so a short count is slow, never wrong.
static int adapt_pair_rounds(pies_solver* s) {
  if (!s->pairs.ctl || s->pairRoundsPinned || !s->graphExec || s->sceneDirty || under_profiler()) return PIES_OK;
  if (s->opt.solver != PIES_SOLVER_PBD || !s->nodeCollisions) return PIES_OK;
  uint32_t deepest = 0;
  HIP_TRY(s, hipMemcpyAsync(&deepest, s->pairs.ctl + kPairDeepest, sizeof(deepest), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  if (deepest == 0) return PIES_OK;  // no pass since the last look
  HIP_TRY(s, hipMemsetAsync(s->pairs.ctl + kPairDeepest, 0, sizeof(uint32_t), s->stream));
  uint32_t rounds = s->pairRounds;
  const uint32_t want = std::min(1024u, ((deepest + deepest / 4u + 8u + 15u) / 16u) * 16u);  // a quarter and eight more, in steps of 16
  if (deepest > rounds) { rounds = want; s->pairCalm = 0; }
  else if (want < rounds) { if (++s->pairCalm >= 3) { rounds = want; s->pairCalm = 0; } }
  else s->pairCalm = 0;
  if (rounds != s->pairRounds) {
    s->pairRounds = rounds;
    return capture_graph(s);
  }
  return PIES_OK;
}

// Radix passes of the node grid's sort: a pass takes up to 11 bits of the cell key, whose width follows the cell box of the
// scene.  The host looks at the box at its synchronisations and captures enough passes for five more bits than it saw (a box 32
// times the volume); a build whose key does not fit the captured passes latches a failure (k_grid_box).
static uint32_t sort_passes_for(uint32_t keyBits) { return std::max(1u, std::min(6u, (keyBits + 5u + 10u) / 11u)); }
static int adapt_sort_passes(pies_solver* s) {
  if (!s->hash.counters || s->sceneDirty || under_profiler()) return PIES_OK;
  if (s->opt.solver != PIES_SOLVER_PBD || !s->nodeCollisions) return PIES_OK;
  int box[6];
  HIP_TRY(s, hipMemcpyAsync(box, s->hash.counters + kCounterBoxMin, sizeof(box), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  uint32_t bits = 0;
  for (int a = 0; a < 3; ++a) {
    if (box[3 + a] < box[a]) return PIES_OK;  // no build yet (or an empty one)
    uint32_t ext = static_cast<uint32_t>(box[3 + a] - box[a]);
    while (ext) { ++bits; ext >>= 1; }
  }
  const uint32_t want = sort_passes_for(bits);
  uint32_t passes = s->sortPasses;
  if (want > passes) { passes = want; s->sortCalm = 0; }
  else if (want < passes) { if (++s->sortCalm >= 8) { passes = want; s->sortCalm = 0; } }
  else s->sortCalm = 0;
  if (passes != s->sortPasses) {
    s->sortPasses = passes;
    if (s->graphExec) return capture_graph(s);  // (without a captured graph the next tick's launches take the new count)
  }
  return PIES_OK;
}

static int poll_failure(pies_solver* s) {
  uint32_t* flagWord = s->hash.counters ? s->hash.counters + 3 : s->pd.tri.counters ? s->pd.tri.counters + 3 : nullptr;
  if (s->simFailed || !flagWord || s->device == PIES_DEVICE_NONE) return PIES_OK;
  uint32_t flag = 0;
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipMemcpyAsync(&flag, flagWord, sizeof(flag), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  if (flag) {  // like the reference's latch (Solver.cpp:741-755, 853-856): tick becomes a no-op
    s->simFailed = true;
    s->error = flag & 2    ? "collision grid overflow (more cells or (cell, triangle) entries than reserved)"
               : flag & 4  ? "more than 2048 nodes overlap one grid cell (runaway pile-up)"
               : flag & 16 ? "more than 1000 triangles in one grid cell, or more than 1000 cells in a triangle's search range (the reference's safety latches, Solver.cpp:741-755)"
               : flag & 32 ? "a triangle's swept bounding box is non-finite"
               : flag & 64 ? "point-triangle contact list overflow"
               : flag & 8  ? "node-node collision pass: the wait for a neighbouring group timed out (PIES_COLLIDE_SPIN_LIMIT)"
               : flag & 128 ? "node-node collision grid: more cell entries than the build reserves (sized from the radii at finalize)"
               : flag & 256 ? "node-node collision pass: more than 512 nodes within reach of one node, or more pairs than reserved (runaway pile-up)"
               : flag & 512 ? "node-node collision grid: the scene's cell box outgrew the captured sort passes between two synchronisations (32 times its volume)"
                          : "a node left the supported cell range (non-finite position)";
  }
  return PIES_OK;
}

}  // namespace pies

extern "C" {

int pies_abi_version(void) { return PIES_ABI_VERSION; }

void pies_default_options(pies_options_t* o) {
  if (!o) return;
  o->fixedTimestepSize = 0.012f;
  o->timeSubsteps = 1;
  o->iterations = 4;
  o->collisionStabilizationIterations = 4;
  o->collisionThresholdDistance = 0.1f;
  o->collisionThickness = 0.05f;
  o->gravity = 10.0f;
  o->damping = 0.006f;
  o->friction = 0.01f;
  o->staticFrictionThreshold = 0.f;
  o->floorHeight = 0.0f;
  o->gridSpacing = 2.0f;
  o->threadCount = 8;
  o->solver = PIES_SOLVER_PD;
}

// PIES_SCHEDULE overrides PIES_SCHEDULE_DEFAULT for new handles (not an explicit pies_set_schedule)
static void apply_schedule_environment(pies_solver* s) {
  if (const char* e = tuning_env("PIES_PCG_OVERFLOW")) s->pcgOverflow = e[0] != '0';
  if (const char* e = tuning_env("PIES_PD_LOCAL_PACKED")) s->pdLocalPacked = e[0] != '0';
  if (const char* e = tuning_env("PIES_PD_CG_SINGLE")) s->pdSingleCg = e[0] != '0';
  if (const char* e = tuning_env("PIES_PD_FUSE_RHS")) s->pdFuseRhs = e[0] != '0';
  if (const char* e = tuning_env("PIES_PD_CG_SINGLE_ROWS")) s->pdSingleCgRows = e[0] != '0';
  if (const char* e = tuning_env("PIES_PCG_BUDGET")) {  // diagnostics: the captured CG iterations, never adapted
    const int v = std::atoi(e);
    if (v >= 1 && v <= 4096) { s->pcgPinned = true; s->pcgPinnedBudget = static_cast<uint32_t>(v); s->pcgBudget = std::min(s->pcgMaxIters, s->pcgPinnedBudget); }
  }
  if (const char* e = std::getenv("PIES_SCHEDULE")) {
    if (!std::strcmp(e, "exact")) s->schedule = PIES_SCHEDULE_EXACT;
    else if (!std::strcmp(e, "coloured")) s->schedule = PIES_SCHEDULE_COLOURED;
    else if (!std::strcmp(e, "layered")) s->schedule = PIES_SCHEDULE_LAYERED;
  }
}

int pies_create(const pies_options_t* options, int device, pies_solver_t** out) {
  if (!out) return PIES_ERR_INVALID;
  *out = nullptr;
  if (device == PIES_DEVICE_NONE) {  // scene/plan inspection only: every call that would compute fails
    pies_solver* s = new pies_solver();
    if (options) s->opt = *options; else pies_default_options(&s->opt);
    if (s->opt.timeSubsteps == 0) s->opt.timeSubsteps = 1;
    s->device = PIES_DEVICE_NONE;
    apply_schedule_environment(s);
    *out = s;
    return PIES_OK;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return PIES_ERR_HIP;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PIES_ERR_HIP;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PIES_ERR_HIP;  // kernels are built for gfx950 only
  if (hipSetDevice(device) != hipSuccess) return PIES_ERR_HIP;
  pies_solver* s = new pies_solver();
  if (options) s->opt = *options; else pies_default_options(&s->opt);
  if (s->opt.timeSubsteps == 0) s->opt.timeSubsteps = 1;
  s->device = device;
  apply_schedule_environment(s);
  if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&s->sideStream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&s->evFork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s->evJoin, hipEventDisableTiming) != hipSuccess) {
    if (s->evFork) (void)hipEventDestroy(s->evFork);
    if (s->sideStream) (void)hipStreamDestroy(s->sideStream);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return PIES_ERR_HIP;
  }
  *out = s;
  return PIES_OK;
}

int pies_destroy(pies_solver_t* s) {
  if (!s) return PIES_OK;
  if (s->device == PIES_DEVICE_NONE) { delete s; return PIES_OK; }
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->sideStream) (void)hipStreamSynchronize(s->sideStream);
  if (s->copyStream) (void)hipStreamSynchronize(s->copyStream);
  free_device(s);
  if (s->h_stage) (void)hipHostFree(s->h_stage);
  for (int b = 0; b < 2; ++b) {
    if (s->h_export[b]) (void)hipHostFree(s->h_export[b]);
    if (s->evTick[b]) (void)hipEventDestroy(s->evTick[b]);
    if (s->evCopied[b]) (void)hipEventDestroy(s->evCopied[b]);
  }
  if (s->d_export) (void)hipFree(s->d_export);
  if (s->copyStream) (void)hipStreamDestroy(s->copyStream);
  if (s->evFork) (void)hipEventDestroy(s->evFork);
  if (s->evJoin) (void)hipEventDestroy(s->evJoin);
  if (s->sideStream) (void)hipStreamDestroy(s->sideStream);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
  return PIES_OK;
}

int pies_clear(pies_solver_t* s) {
  if (!s) return PIES_ERR_INVALID;
  if (s->device != PIES_DEVICE_NONE) {
    (void)hipSetDevice(s->device);
    (void)hipStreamSynchronize(s->stream);
    free_device(s);
  }
  s->h_pos.clear(); s->h_prev.clear(); s->h_vel.clear(); s->h_radius.clear(); s->h_invMass.clear();
  s->h_position.clear(); s->h_distance.clear(); s->h_tet.clear(); s->h_volume.clear(); s->h_bend.clear();
  s->h_triangles.clear(); s->h_lines.clear();
  s->h_shape.clear(); s->h_goal.clear();  // like the reference, the fixed-region list survives clear() (Solver.cpp:488-507)
  for (Plan& p : s->plan) { p.order.clear(); p.batches.clear(); }
  s->constraintId = 0;
  s->sceneDirty = true;
  s->stale = 0;
  s->hostNodesDirty = false;
  return PIES_OK;  // like the reference, the failure latch is not reset (Solver.cpp:488-507)
}

const char* pies_last_error(const pies_solver_t* s) { return s ? s->error.c_str() : "null handle"; }

int pies_get_options(const pies_solver_t* s, pies_options_t* out) {
  if (!s || !out) return PIES_ERR_INVALID;
  *out = s->opt;
  return PIES_OK;
}

int pies_set_flag(pies_solver_t* s, int flag, int value) {
  if (!s) return PIES_ERR_INVALID;
  if (flag == PIES_FLAG_REFERENCE_COLLISION_ORDER || flag == PIES_FLAG_COLLISION_ORDER) {
    int v = value;
    if (flag == PIES_FLAG_REFERENCE_COLLISION_ORDER) v = value != 0 ? PIES_COLLISION_ORDER_REFERENCE : PIES_COLLISION_ORDER_PAIRS;
    if (v != PIES_COLLISION_ORDER_REFERENCE && v != PIES_COLLISION_ORDER_GROUPS && v != PIES_COLLISION_ORDER_PAIRS)
      return fail(s, PIES_ERR_INVALID, "pies_set_flag: unknown collision order");
    if (s->collisionOrderFlag != v) {
      s->collisionOrderFlag = v;
      if (!s->sceneDirty) s->graphDirty = true;  // same buffers, another resolve kernel in the captured substep
    }
    return PIES_OK;
  }
  bool* target = flag == PIES_FLAG_RELEASE_HINGE       ? &s->releaseHinge
                 : flag == PIES_FLAG_NODE_COLLISIONS   ? &s->nodeCollisions
                 : flag == PIES_FLAG_TRIANGLE_COLLISIONS ? &s->triangleCollisions
                                                         : nullptr;
  if (!target) return fail(s, PIES_ERR_INVALID, "pies_set_flag: unknown flag");
  if (*target != (value != 0)) {
    *target = value != 0;
    // releaseHinge only drops the position-constraint launches (Solver.cpp:59): with per-container batches (COLOURED,
    // LAYERED, PD) the plans and buffers stay valid and the substep is captured again; schedule EXACT bakes the
    // position constraints into its dependency levels, and the collision flags decide which buffers exist
    const bool captureOnly = flag == PIES_FLAG_RELEASE_HINGE && !s->sceneDirty && (s->opt.solver == PIES_SOLVER_PD || !s->wave.active);
    if (captureOnly) s->graphDirty = true;
    else {
      if (int rc = scene_sync_host(s)) return rc;
      s->sceneDirty = true;
    }
  }
  return PIES_OK;
}

int pies_set_solver(pies_solver_t* s, int solver) {
  if (!s) return PIES_ERR_INVALID;
  if (solver != PIES_SOLVER_PBD && solver != PIES_SOLVER_PD) return fail(s, PIES_ERR_INVALID, "pies_set_solver: unknown solver");
  if (solver != s->opt.solver) {
    if (int rc = scene_sync_host(s)) return rc;
    s->opt.solver = solver;
    s->sceneDirty = true;
  }
  return PIES_OK;
}

int pies_set_schedule(pies_solver_t* s, int schedule) {
  if (!s) return PIES_ERR_INVALID;
  if (schedule != PIES_SCHEDULE_EXACT && schedule != PIES_SCHEDULE_COLOURED && schedule != PIES_SCHEDULE_LAYERED) return fail(s, PIES_ERR_INVALID, "unknown schedule");
  if (schedule != s->schedule) {
    if (int rc = scene_sync_host(s)) return rc;
    s->schedule = schedule;
    s->collisionOrderFlag = -1;  // the node-node order follows the schedule again
    s->sceneDirty = true;
  }
  return PIES_OK;
}

int pies_set_pcg(pies_solver_t* s, float rel_tol, uint32_t max_iters) {
  if (!s || !(rel_tol >= 0.0f) || max_iters == 0 || max_iters > 4096) return fail(s, PIES_ERR_INVALID, "pies_set_pcg: bad argument");
  if (rel_tol != s->pcgTol || max_iters != s->pcgMaxIters) {
    s->pcgTol = rel_tol;
    s->pcgMaxIters = max_iters;
    s->pcgBudget = std::min(max_iters, s->pcgPinned ? s->pcgPinnedBudget : 32u);
    s->graphDirty = true;  // the captured launch sequence changes, nothing else
  }
  return PIES_OK;
}

int pies_set_pcg_retry(pies_solver_t* s, int enabled) {
  if (!s) return PIES_ERR_INVALID;
  s->pcgRetry = enabled != 0;
  return PIES_OK;
}

int pies_get_pcg_stats(pies_solver_t* s, float* max_rel_residual, uint32_t* max_iters_used, uint32_t* solves) {
  if (!s) return PIES_ERR_INVALID;
  float st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (s->opt.solver == PIES_SOLVER_PD && s->pd.cg.stats) {
    HIP_TRY(s, hipSetDevice(s->device));
    HIP_TRY(s, hipMemcpyAsync(st, s->pd.cg.stats, sizeof(st), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(s, hipStreamSynchronize(s->stream));
  }
  if (max_rel_residual) *max_rel_residual = std::sqrt(st[0]);
  if (max_iters_used) *max_iters_used = static_cast<uint32_t>(st[1]);
  if (solves) *solves = static_cast<uint32_t>(st[2]);
  return PIES_OK;
}

int pies_get_pcg_health(pies_solver_t* s, uint64_t* short_solves, uint64_t* solves_total, uint32_t* substeps_retried, uint32_t* budget) {
  if (!s) return PIES_ERR_INVALID;
  float st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (s->opt.solver == PIES_SOLVER_PD && s->pd.cg.stats) {
    HIP_TRY(s, hipSetDevice(s->device));
    HIP_TRY(s, hipMemcpyAsync(st, s->pd.cg.stats, sizeof(st), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(s, hipStreamSynchronize(s->stream));
  }
  // stats[4], [5]: solves left above the tolerance / solves run by the substeps whose result was kept (a substep
  // that pies_tick ran again takes its counts back), since the buffers were built
  uint64_t life[2];
  std::memcpy(life, st + 4, sizeof(life));  // (64-bit integer counters in the words [4..5] and [6..7])
  if (short_solves) *short_solves = life[0];
  if (solves_total) *solves_total = life[1];
  if (substeps_retried) *substeps_retried = s->pcgRetries;
  if (budget) *budget = s->pcgBudget;
  return PIES_OK;
}

static int build_plans(pies_solver* s, int sched) {
  const uint32_t n = s->nodeCount();
  s->layer = LayerPlan{};
  s->wave = WavePlan{};
  if (sched == PIES_SCHEDULE_LAYERED) {
    if (build_layer_plan(s)) return PIES_OK;
    sched = PIES_SCHEDULE_COLOURED;  // wide bodies (two levels do not fit in LDS), scenes without constraints
  }
  std::vector<uint32_t> ids;
  ids.resize(s->h_position.size());
  for (size_t i = 0; i < ids.size(); ++i) ids[i] = s->h_position[i].id;
  build_plan({ids.data(), 1, (uint32_t)s->h_position.size(), 0x1}, n, sched, s->plan[PIES_POSITION]);
  ids.resize(2 * s->h_distance.size());
  for (size_t i = 0; i < s->h_distance.size(); ++i) { ids[2 * i] = s->h_distance[i].ids[0]; ids[2 * i + 1] = s->h_distance[i].ids[1]; }
  // a distance projection moves node a only (Constraints.cpp:34-36); node b is read
  std::vector<uint16_t> hint(s->h_distance.size());
  for (size_t i = 0; i < hint.size(); ++i) hint[i] = s->h_distance[i].hint;
  build_plan({ids.data(), 2, (uint32_t)s->h_distance.size(), 0x1, hint.data()}, n, sched, s->plan[PIES_DISTANCE]);
  ids.resize(4 * s->h_tet.size());
  for (size_t i = 0; i < s->h_tet.size(); ++i) std::memcpy(&ids[4 * i], s->h_tet[i].ids, 16);
  hint.resize(s->h_tet.size());
  for (size_t i = 0; i < hint.size(); ++i) hint[i] = s->h_tet[i].hint;
  build_plan({ids.data(), 4, (uint32_t)s->h_tet.size(), 0xF, hint.data()}, n, sched, s->plan[PIES_TET]);
  ids.resize(4 * s->h_bend.size());
  for (size_t i = 0; i < s->h_bend.size(); ++i) std::memcpy(&ids[4 * i], s->h_bend[i].ids, 16);
  build_plan({ids.data(), 4, (uint32_t)s->h_bend.size(), 0xF}, n, sched, s->plan[PIES_BEND]);
  const char* noWave = tuning_env("PIES_NO_WAVEFRONT");
  if (sched == PIES_SCHEDULE_EXACT && !(noWave && noWave[0] == '1')) build_wave_plan(s, s->wave);
  return PIES_OK;
}

int pies_finalize(pies_solver_t* s) {
  if (!s) return PIES_ERR_INVALID;
  if (s->device == PIES_DEVICE_NONE) {  // host-only handle: plans can be inspected, nothing is uploaded
    if (s->sceneDirty) build_plans(s, s->opt.solver == PIES_SOLVER_PD ? -1 : s->schedule);
    s->sceneDirty = false;
    return PIES_OK;
  }
  HIP_TRY(s, hipSetDevice(s->device));
  if (!s->sceneDirty) {
    if (s->hostNodesDirty) return upload_nodes(s);
    return PIES_OK;
  }
  const bool isPD = s->opt.solver == PIES_SOLVER_PD;
  const bool collide = s->nodeCollisions && !isPD;
  // The parallel visiting order of the node-node pass needs ranges of at most 2 cells per axis (true for the reference
  // defaults r = 0.5, spacing 2); other scenes run the pass in the reference's own order (one sequential chain, any range
  // up to the reference's 50 cells per axis).
  s->collideFast = true;
  if (collide) { uint64_t e; collision_grid_bound(s, e, s->collideFast); }
  if (int rc = download_nodes(s)) return rc;
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  free_device(s);

  const uint32_t n = s->nodeCount();
  // ---- plans (PD's local step is order independent: one batch per container, host order) ----
  build_plans(s, isPD ? -1 : s->schedule);
  // ---- node arrays ----
  if (n) {
    void* p;
    HIP_TRY(s, hipMalloc(&p, n * sizeof(float4))); s->allocations.push_back(p); s->nd.pos = (float4*)p;
    HIP_TRY(s, hipMalloc(&p, n * sizeof(float4))); s->allocations.push_back(p); s->nd.prev = (float4*)p;
    HIP_TRY(s, hipMalloc(&p, n * sizeof(float4))); s->allocations.push_back(p); s->nd.vel = (float4*)p;
    HIP_TRY(s, hipMalloc(&p, n * sizeof(float))); s->allocations.push_back(p); s->nd.radius = (float*)p;
    s->nd.n = n;
    if (s->h_stage_n < n) {
      if (s->h_stage) (void)hipHostFree(s->h_stage);
      s->h_stage = nullptr;
      HIP_TRY(s, hipHostMalloc((void**)&s->h_stage, n * sizeof(float4), hipHostMallocDefault));
      s->h_stage_n = n;
    }
  }
  if (int rc = upload_nodes(s)) return rc;
  // ---- constraint records, in schedule order ----
  {
    const Plan& pl = s->plan[PIES_POSITION];
    std::vector<uint32_t> id(pl.order.size());
    std::vector<float4> tw(pl.order.size());
    for (size_t k = 0; k < pl.order.size(); ++k) {
      const HostPosition& c = s->h_position[pl.order[k]];
      id[k] = c.id;
      tw[k] = make_float4(c.target[0], c.target[1], c.target[2], c.w);
    }
    if (int rc = upload(s, id, &s->d_pc_id)) return rc;
    if (int rc = upload(s, tw, &s->d_pc_tw)) return rc;
    HIP_TRY(s, hipStreamSynchronize(s->stream));
  }
  {
    const Plan& pl = s->plan[PIES_DISTANCE];
    std::vector<uint2> id(pl.order.size());
    std::vector<float2> rw(pl.order.size());
    for (size_t k = 0; k < pl.order.size(); ++k) {
      const HostDistance& c = s->h_distance[pl.order[k]];
      id[k] = make_uint2(c.ids[0], c.ids[1]);
      rw[k] = make_float2(c.target, c.w);
    }
    if (int rc = upload(s, id, &s->d_dc_ids)) return rc;
    if (int rc = upload(s, rw, &s->d_dc_rw)) return rc;
    HIP_TRY(s, hipStreamSynchronize(s->stream));
  }
  {
    const Plan& pl = s->plan[PIES_TET];
    std::vector<uint4> id(pl.order.size());
    std::vector<float4> q0(pl.order.size()), q1(pl.order.size()), q2(pl.order.size());
    for (size_t k = 0; k < pl.order.size(); ++k) {
      const HostTet& c = s->h_tet[pl.order[k]];
      id[k] = make_uint4(c.ids[0], c.ids[1], c.ids[2], c.ids[3]);
      q0[k] = make_float4(c.qinv[0], c.qinv[1], c.qinv[2], c.qinv[3]);
      q1[k] = make_float4(c.qinv[4], c.qinv[5], c.qinv[6], c.qinv[7]);
      q2[k] = make_float4(c.qinv[8], c.lo, c.hi, c.w);
    }
    if (int rc = upload(s, id, &s->d_tc_ids)) return rc;
    if (int rc = upload(s, q0, &s->d_tc_q0)) return rc;
    if (int rc = upload(s, q1, &s->d_tc_q1)) return rc;
    if (int rc = upload(s, q2, &s->d_tc_q2)) return rc;
    HIP_TRY(s, hipStreamSynchronize(s->stream));
  }
  {
    const Plan& pl = s->plan[PIES_BEND];
    std::vector<uint4> id(pl.order.size());
    std::vector<float2> aw(pl.order.size());
    for (size_t k = 0; k < pl.order.size(); ++k) {
      const HostBend& c = s->h_bend[pl.order[k]];
      id[k] = make_uint4(c.ids[0], c.ids[1], c.ids[2], c.ids[3]);
      aw[k] = make_float2(c.angle, c.w);
    }
    if (int rc = upload(s, id, &s->d_bc_ids)) return rc;
    if (int rc = upload(s, aw, &s->d_bc_aw)) return rc;
    HIP_TRY(s, hipStreamSynchronize(s->stream));
  }
  if (s->layer.active && !isPD) {
    const LayerPlan& L = s->layer;
    LayerDevice& d = s->d_layer;
    int maxLds = 0;
    HIP_TRY(s, hipDeviceGetAttribute(&maxLds, hipDeviceAttributeMaxSharedMemoryPerBlock, s->device));
    if (static_cast<size_t>(L.maxGroupNodes) * 20 + 4096 > static_cast<size_t>(maxLds))
      return fail(s, PIES_ERR_UNSUPPORTED, "schedule LAYERED: the device's LDS is smaller than this build assumes");
    HIP_TRY(s, layer_prepare(L.maxGroupNodes));
    if (int rc = upload(s, L.nodeList, &d.nodeList)) return rc;
    if (int rc = dev_alloc(s, L.nodeList.size(), &d.lpos, true)) return rc;
    {
      std::vector<float> lrad(L.nodeList.size());
      for (size_t i = 0; i < lrad.size(); ++i) lrad[i] = s->h_radius[L.nodeList[i]];
      if (int rc = upload(s, lrad, &d.lrad)) return rc;
    }
    for (int ph = 0; ph < 4; ++ph)
      if (int rc = upload(s, L.tiles[ph], &d.tiles[ph])) return rc;
    for (int k = 0; k < 5; ++k)
      for (int ph = 0; ph < 4; ++ph)
        if (int rc = upload(s, L.kind[k].colOff[ph], &d.colOff[k][ph])) return rc;
    if (int rc = upload(s, L.kind[PIES_POSITION].local, &d.pc_lid)) return rc;
    {
      const std::vector<uint32_t>& l = L.kind[PIES_DISTANCE].local;
      std::vector<uint32_t> packed(l.size() / 2);
      for (size_t k = 0; k < packed.size(); ++k) packed[k] = l[2 * k] | (l[2 * k + 1] << 16);
      if (int rc = upload(s, packed, &d.dc_lid)) return rc;
    }
    for (int k : {PIES_TET, PIES_BEND}) {
      const std::vector<uint32_t>& l = L.kind[k].local;
      std::vector<uint2> packed(l.size() / 4);
      for (size_t c = 0; c < packed.size(); ++c) packed[c] = make_uint2(l[4 * c] | (l[4 * c + 1] << 16), l[4 * c + 2] | (l[4 * c + 3] << 16));
      if (int rc = upload(s, packed, k == PIES_TET ? &d.tc_lid : &d.bc_lid)) return rc;
    }
    HIP_TRY(s, hipStreamSynchronize(s->stream));
  }
  s->d_waveIndex = nullptr;
  if (s->wave.active && !isPD) {
    if (int rc = upload(s, s->wave.index, &s->d_waveIndex)) return rc;
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    std::vector<uint32_t>().swap(s->wave.index);  // the levels (offsets, counts) stay on the host; the items live in HBM
  }
  if (collide && n) {
    HashArrays& H = s->hash;
    H.n = n;
    uint64_t entries = 0;
    bool fast = true;
    collision_grid_bound(s, entries, fast);
    if (n >= (1u << 25)) return fail(s, PIES_ERR_UNSUPPORTED, "node-node collisions: more than 2^25 nodes");
    if (entries > 0x7fff0000ull) return fail(s, PIES_ERR_UNSUPPORTED, "node-node collisions: more than 2^31 (cell, node) entries (gridSpacing is tiny against the radii)");
    H.maxEntries = static_cast<uint32_t>(entries + 64);
    {  // sort passes to start with: from the cell box of the scene as it stands (adapt_sort_passes follows it from there)
      float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      float rmax = 0.0f;
      for (float r : s->h_radius) if (std::isfinite(r)) rmax = std::max(rmax, r);
      const size_t stride = s->h_pos.size() / std::max<size_t>(1, n);
      for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
          const float v = s->h_pos[i * stride + a];
          if (std::isfinite(v)) { lo[a] = std::min(lo[a], v); hi[a] = std::max(hi[a], v); }
        }
      uint32_t bits = 0;
      for (int a = 0; a < 3; ++a) {
        const double cells = hi[a] >= lo[a] ? (static_cast<double>(hi[a]) - lo[a] + 2.0 * (rmax + 0.5)) / s->opt.gridSpacing + 2.0 : 1.0;
        uint64_t ext = static_cast<uint64_t>(std::min(cells, 4.0e9));
        while (ext) { ++bits; ext >>= 1; }
      }
      s->sortPasses = sort_passes_for(bits);
      s->sortCalm = 0;
    }
    uint32_t cap = 1024;
    const uint64_t want = s->collideFast ? 16ull * n : 2ull * H.maxEntries;  // distinct cells <= 8n resp. <= entries: load factor <= 0.5
    while (cap < want && cap < (1u << 30)) cap <<= 1;
    H.capacity = cap;
    H.mask = cap - 1;
    if (int rc = dev_alloc(s, n, &H.rng, true)) return rc;
    if (int rc = dev_alloc(s, n + 1ull, &H.entCount, true)) return rc;
    if (int rc = dev_alloc(s, n + 1ull, &H.entOff, true)) return rc;
    if (int rc = dev_alloc(s, (n + 1ull) / 2048 + 2, &H.scanSums, true)) return rc;
    if (int rc = dev_alloc(s, 6ull * ((n + 1ull + 255) / 256), &H.boxPart, true)) return rc;
    for (int b = 0; b < 2; ++b) {
      if (int rc = dev_alloc(s, H.maxEntries, &H.key[b], true)) return rc;
      if (int rc = dev_alloc(s, H.maxEntries, &H.val[b], true)) return rc;
    }
    if (int rc = dev_alloc(s, 2048ull * ((H.maxEntries + kRadixTile - 1) / kRadixTile) + 2048, &H.hist, true)) return rc;  // (digit, workgroup) counts of a pass + the digit totals
    if (int rc = dev_alloc(s, cap, &H.keys)) return rc;
    HIP_TRY(s, hipMemsetAsync(H.keys, 0xFF, static_cast<size_t>(cap) * sizeof(uint64_t), s->stream));
    if (int rc = dev_alloc(s, cap, &H.start, true)) return rc;
    if (int rc = dev_alloc(s, cap, &H.end, true)) return rc;
    if (int rc = dev_alloc(s, cap, &H.gcnt, true)) return rc;
    if (int rc = dev_alloc(s, cap, &H.done, true)) return rc;
    if (int rc = dev_alloc(s, std::min<uint64_t>(cap, H.maxEntries), &H.used, true)) return rc;
    if (int rc = dev_alloc(s, kHashCounters, &H.counters, true)) return rc;
    if (int rc = dev_alloc(s, 27ull * n, &H.passList, true)) return rc;
    s->pairs = PairArrays{};
    {  // pair order: every node's list of partners (in pools), the frontier of the level launches

      PairArrays& P = s->pairs;
      P.n = n;
      // list entries: 96 per node on average (BASELINE config 4 lists 15-50), in kPairPools pools; a small scene may list every
      // pair (a body that has collapsed into a few cells: quirk Q2 does that to a tetrahedral PBD body within a tick)
      // (only a small scene: the n * n floor used to apply to every scene of 8 192 nodes and more - 270 MB per handle)
      const uint64_t everyPair = n <= 8192u ? static_cast<uint64_t>(n) * n : 0ull;
      P.poolCap = static_cast<uint32_t>(std::min<uint64_t>((std::max<uint64_t>(96ull * n, everyPair) + 65536) / kPairPools + 4096, 0x7fff0000ull / kPairPools));
      if (int rc = dev_alloc(s, 4ull * n, &P.node, true)) return rc;
      if (int rc = dev_alloc(s, n, &P.vel0)) return rc;
      if (int rc = dev_alloc(s, n, &P.exc, true)) return rc;
      if (int rc = dev_alloc(s, static_cast<size_t>(P.poolCap) * kPairPools, &P.nbr)) return rc;
      if (!s->collideFast)  // ranges wider than two cells per axis: the shared-cell count of an entry does not fit its four bits
        if (int rc = dev_alloc(s, static_cast<size_t>(P.poolCap) * kPairPools, &P.nbrM)) return rc;
      P.frCap = n / 32 + 256;  // a chunk of 64 lanes appends at most 128 nodes to the one sub-list it is dealt to
      for (int b = 0; b < 2; ++b)
        if (int rc = dev_alloc(s, static_cast<size_t>(P.frCap) * kPairLists, &P.fr[b])) return rc;
      if (int rc = dev_alloc(s, 3ull * kPairLists * kPairPad, &P.frCount, true)) return rc;
      if (int rc = dev_alloc(s, static_cast<size_t>(kPairStripes) * kPairPad, &P.hitStripe, true)) return rc;
      if (int rc = dev_alloc(s, n, &P.bq)) return rc;
      if (int rc = dev_alloc(s, 64ull * kPairPad, &P.stat, true)) return rc;
      if (int rc = dev_alloc(s, 4ull * n, &P.grp)) return rc;
      if (int rc = dev_alloc(s, n, &P.spill)) return rc;
      if (int rc = dev_alloc(s, n, &P.left, true)) return rc;
      if (int rc = dev_alloc(s, static_cast<size_t>(kPairPools) * kPairPad, &P.pool, true)) return rc;
      if (int rc = dev_alloc(s, kPairWords, &P.ctl, true)) return rc;
    }
    HIP_TRY(s, hipStreamSynchronize(s->stream));
  }
  if (isPD) {
    std::vector<uint4> id(s->h_volume.size());
    std::vector<float4> q0(id.size()), q1(id.size()), q2(id.size());
    for (size_t k = 0; k < id.size(); ++k) {
      const HostTet& c = s->h_volume[k];
      id[k] = make_uint4(c.ids[0], c.ids[1], c.ids[2], c.ids[3]);
      q0[k] = make_float4(c.qinv[0], c.qinv[1], c.qinv[2], c.qinv[3]);
      q1[k] = make_float4(c.qinv[4], c.qinv[5], c.qinv[6], c.qinv[7]);
      q2[k] = make_float4(c.qinv[8], c.lo, c.hi, c.w);
    }
    if (int rc = upload(s, id, &s->d_vc_ids)) return rc;
    if (int rc = upload(s, q0, &s->d_vc_q0)) return rc;
    if (int rc = upload(s, q1, &s->d_vc_q1)) return rc;
    if (int rc = upload(s, q2, &s->d_vc_q2)) return rc;
    // strain and volume constraints added pairwise over the same elements share one gather and one SVD
    s->tetVolumePaired = !s->h_tet.empty() && s->h_tet.size() == s->h_volume.size();
    for (size_t k = 0; s->tetVolumePaired && k < s->h_tet.size(); ++k) {
      const HostTet &a = s->h_tet[s->plan[PIES_TET].order[k]], &b = s->h_volume[k];
      s->tetVolumePaired = std::memcmp(a.ids, b.ids, sizeof(a.ids)) == 0 && std::memcmp(a.qinv, b.qinv, sizeof(a.qinv)) == 0;
    }
    if (const char* e = tuning_env("PIES_NO_TET_PAIRS"); e && e[0] == '1') s->tetVolumePaired = false;
    // Rest dictionary: the 64 bytes of constants of an element pair are the same for every element of one shape and material.
    // With few distinct sets (a createTetBox lattice: one per orientation) the local step reads a 16-bit index per element.
    s->d_pairDictIndex = nullptr;
    s->d_pairDictTable = nullptr;
    s->pairDictSets = 0;
    s->h_pairDictIndex.clear();
    const char* de = tuning_env("PIES_PD_REST_DICT");
    if (s->tetVolumePaired && !(de && de[0] == '0')) {
      struct Set { float v[16]; bool operator<(const Set& o) const { return std::memcmp(v, o.v, sizeof(v)) < 0; } };
      std::map<Set, uint16_t> sets;
      std::vector<uint16_t> index(id.size());
      std::vector<float4> table;
      bool ok = true;
      for (size_t k = 0; ok && k < id.size(); ++k) {
        const HostTet &a = s->h_tet[s->plan[PIES_TET].order[k]], &b = s->h_volume[k];
        Set key;
        std::memcpy(key.v, a.qinv, 9 * sizeof(float));
        key.v[9] = a.lo; key.v[10] = a.hi; key.v[11] = a.w;
        key.v[12] = b.qinv[8]; key.v[13] = b.lo; key.v[14] = b.hi; key.v[15] = b.w;
        auto it = sets.find(key);
        if (it == sets.end()) {
          if (sets.size() >= 4096 || (sets.size() + 1) * 16 > id.size()) { ok = false; break; }  // no real compression: per-element arrays
          it = sets.emplace(key, static_cast<uint16_t>(sets.size())).first;
          table.push_back(make_float4(key.v[0], key.v[1], key.v[2], key.v[3]));
          table.push_back(make_float4(key.v[4], key.v[5], key.v[6], key.v[7]));
          table.push_back(make_float4(key.v[8], key.v[9], key.v[10], key.v[11]));
          table.push_back(make_float4(key.v[12], key.v[13], key.v[14], key.v[15]));
        }
        index[k] = it->second;
      }
      s->h_pairDictIndex.clear();
      if (ok && !index.empty()) {
        s->h_pairDictIndex = index;
        if (int rc = upload(s, index, &s->d_pairDictIndex)) return rc;
        if (int rc = upload(s, table, &s->d_pairDictTable)) return rc;
        s->pairDictSets = static_cast<uint32_t>(sets.size());
      }
    }
    if (int rc = pd_build(s)) return rc;
    if (n) {  // input of a substep, kept until its solves are known to have met the tolerance (pd_tick_checked)
      if (int rc = dev_alloc(s, n, &s->snapPos)) return rc;
      if (int rc = dev_alloc(s, n, &s->snapPrev)) return rc;
      if (int rc = dev_alloc(s, n, &s->snapVel)) return rc;
      if (s->pd.shape.count)
        if (int rc = dev_alloc(s, 4ull * s->pd.shape.count, &s->snapQuat)) return rc;
    }
  }
  if (int rc = capture_graph(s)) return rc;
  s->sceneDirty = false;
  s->graphDirty = false;
  return PIES_OK;
}

// Brings HBM and the captured graph up to date with the host-side scene.
static int ensure_ready(pies_solver* s) {
  if (s->sceneDirty || s->hostNodesDirty)
    if (int rc = pies_finalize(s)) return rc;
  HIP_TRY(s, hipSetDevice(s->device));
  if (s->graphDirty) {
    HIP_TRY(s, hipStreamSynchronize(s->stream));  // the old graph may still be running
    if (int rc = capture_graph(s)) return rc;
    s->graphDirty = false;
  }
  return PIES_OK;
}

static int launch_substep(pies_solver* s) {
  if (s->graphExec) {
    HIP_TRY(s, hipGraphLaunch(s->graphExec, s->stream));
  } else {  // PIES_NO_GRAPH=1: eager launches (debug / tracing)
    enqueue_substep(s, nullptr);
    HIP_TRY(s, hipGetLastError());
  }
  return PIES_OK;
}

int pies_tick_async(pies_solver_t* s) {
  if (!s) return PIES_ERR_INVALID;
  if (s->device == PIES_DEVICE_NONE) return fail(s, PIES_ERR_HIP, "host-only handle (PIES_DEVICE_NONE): there is no CPU solver");
  if (s->simFailed) return PIES_OK;  // Solver.cpp:26-28
  if (int rc = ensure_ready(s)) return rc;
  if (s->nd.n == 0) return PIES_OK;
  if (s->opt.solver == PIES_SOLVER_PD) {
    // The captured CG budget can only follow the solves at a host synchronisation.  A caller that queues tick after tick
    // without one gets one here every 16 ticks (the queue drains once, ~50 us of idle device): measured without it, 150
    // queued ticks of a contact scene whose budget had settled at 8 ran short when the contacts re-bound (14 iterations
    // needed), the unconverged positions fed the next tick, and the simulation ended in the failure latch.
    if (s->asyncSinceSync >= 16) {
      if (int rc = pies_synchronize(s)) return rc;
      if (s->simFailed) return PIES_OK;
    }
    ++s->asyncSinceSync;
    if (s->goalDirty)
      if (int rc = pd_upload_goals(s)) return rc;
    if (s->asyncSinceSync == 1) HIP_TRY(s, hipMemsetAsync(s->pd.cg.stats, 0, 4 * sizeof(float), s->stream));
  } else if (s->nodeCollisions && s->hash.counters) {
    // The node grid's captured radix passes hold the scene's cell box plus five key bits, and the host can only follow a
    // growing box at a synchronisation (adapt_sort_passes): a caller that queues PBD ticks blindly gets one every 16 ticks,
    // like the PD path above, so that a burst that spreads the particles never outruns the captured passes.
    if (s->asyncSinceSync >= 16) {
      if (int rc = pies_synchronize(s)) return rc;
      if (s->simFailed) return PIES_OK;
    }
    ++s->asyncSinceSync;
  }
  for (uint32_t sub = 0; sub < s->opt.timeSubsteps; ++sub)
    if (int rc = launch_substep(s)) return rc;
  if (under_profiler()) HIP_TRY(s, hipStreamSynchronize(s->stream));
  s->stale = 7u;
  return PIES_OK;
}

int pies_synchronize(pies_solver_t* s) {
  if (!s) return PIES_ERR_INVALID;
  if (s->device == PIES_DEVICE_NONE) return PIES_OK;
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  s->asyncSinceSync = 0;
  if (int rc = poll_failure(s)) return rc;  // a loop of pies_tick_async learns here that the simulation failed
  if (int rc = adapt_pair_rounds(s)) return rc;
  if (int rc = adapt_sort_passes(s)) return rc;
  return adapt_pcg_budget(s);
}

// Projective Dynamics, synchronous tick: the reference's global step is a direct solve (Solver.cpp:258-262, 356), the
// device's a CG with a captured iteration budget.  A substep in which a solve ends above the tolerance (new contacts
// stiffen the system from one substep to the next) is therefore not kept: the node state is put back and the substep
// runs again with four times the budget, up to the ceiling of pies_set_pcg.
static int pd_tick_checked(pies_solver* s) {
  const uint32_t n = s->nd.n;
  float before[8], after[8];
  HIP_TRY(s, hipMemsetAsync(s->pd.cg.stats, 0, 4 * sizeof(float), s->stream));
  for (uint32_t sub = 0; sub < s->opt.timeSubsteps; ++sub) {
    HIP_TRY(s, hipMemcpyAsync(before, s->pd.cg.stats, sizeof(before), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(s, hipMemcpyAsync(s->snapPos, s->nd.pos, n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
    HIP_TRY(s, hipMemcpyAsync(s->snapPrev, s->nd.prev, n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
    HIP_TRY(s, hipMemcpyAsync(s->snapVel, s->nd.vel, n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
    if (s->snapQuat)
      HIP_TRY(s, hipMemcpyAsync(s->snapQuat, s->pd.shape.quat, 4ull * s->pd.shape.count * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    for (;;) {
      if (int rc = launch_substep(s)) return rc;
      HIP_TRY(s, hipMemcpyAsync(after, s->pd.cg.stats, sizeof(after), hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(s, hipStreamSynchronize(s->stream));
      const bool ranShort = after[3] > before[3];
      if (!ranShort || s->pcgBudget >= s->pcgMaxIters || s->pcgPinned) {
        if (ranShort) s->pcgShortSolves += static_cast<uint64_t>(after[3] - before[3]);
        break;
      }
      // put the substep's input back (the statistics too: the attempt does not count) and capture a larger budget
      HIP_TRY(s, hipMemcpyAsync(s->nd.pos, s->snapPos, n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
      HIP_TRY(s, hipMemcpyAsync(s->nd.prev, s->snapPrev, n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
      HIP_TRY(s, hipMemcpyAsync(s->nd.vel, s->snapVel, n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
      if (s->snapQuat)
        HIP_TRY(s, hipMemcpyAsync(s->pd.shape.quat, s->snapQuat, 4ull * s->pd.shape.count * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
      HIP_TRY(s, hipMemcpyAsync(s->pd.cg.stats, before, sizeof(before), hipMemcpyHostToDevice, s->stream));
      HIP_TRY(s, hipStreamSynchronize(s->stream));
      s->pcgBudget = std::min(s->pcgMaxIters, std::max(32u, 4u * s->pcgBudget));
      s->pcgCalm = 0;
      s->pcgWindowMax = 0;
      s->pcgCooldown = 24;
      ++s->pcgRetries;
      if (const char* e = std::getenv("PIES_PCG_DEBUG"); e && e[0] == '1')
        std::fprintf(stderr, "[pies] pcg: substep ran short (residual^2 %.3g): again with budget %u\n", after[0], s->pcgBudget);
      if (int rc = s->pdLadder.empty() ? capture_graph(s) : select_pd_graph(s)) return rc;
    }
  }
  return PIES_OK;
}

int pies_tick(pies_solver_t* s) {
  if (!s) return PIES_ERR_INVALID;
  if (s->simFailed) return PIES_OK;
  if (s->opt.solver == PIES_SOLVER_PD && s->pcgRetry && s->device != PIES_DEVICE_NONE && !under_profiler()) {
    if (int rc = ensure_ready(s)) return rc;
    if (s->nd.n == 0) return PIES_OK;
    if (s->goalDirty)
      if (int rc = pd_upload_goals(s)) return rc;
    if (int rc = pd_tick_checked(s)) return rc;
    s->stale = 7u;
  } else {
    if (int rc = pies_tick_async(s)) return rc;
  }
  const uint32_t n = s->nd.n;
  if (n == 0) return PIES_OK;
  // Solver.cpp:157 : _vertices[i].position = position -- one D2H copy per tick; the host mirror's positions are
  // current afterwards (pies_read_nodes / pies_read_positions_strided copy from it without touching the device)
  if (int rc = download_nodes(s, 1u)) return rc;
  if (int rc = poll_failure(s)) return rc;
  if (int rc = adapt_pair_rounds(s)) return rc;
  if (int rc = adapt_sort_passes(s)) return rc;
  return adapt_pcg_budget(s);
}

// ---- render-state export: frame k leaves through a copy stream while frame k+1 computes -----------------------
static int export_prepare(pies_solver* s) {
  const uint32_t n = s->nd.n;
  if (!s->copyStream) {
    HIP_TRY(s, hipStreamCreateWithFlags(&s->copyStream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
      HIP_TRY(s, hipEventCreateWithFlags(&s->evTick[b], hipEventDisableTiming));
      HIP_TRY(s, hipEventCreateWithFlags(&s->evCopied[b], hipEventDisableTiming));
    }
  }
  if (s->h_export_n < n || !s->d_export) {
    HIP_TRY(s, hipStreamSynchronize(s->copyStream));
    for (int b = 0; b < 2; ++b) {
      if (s->h_export[b]) (void)hipHostFree(s->h_export[b]);
      s->h_export[b] = nullptr;
      HIP_TRY(s, hipHostMalloc((void**)&s->h_export[b], std::max<size_t>(n, 1) * sizeof(float4), hipHostMallocDefault));
    }
    if (s->d_export) (void)hipFree(s->d_export);
    s->d_export = nullptr;
    HIP_TRY(s, hipMalloc((void**)&s->d_export, std::max<size_t>(n, 1) * sizeof(float4)));
    s->h_export_n = n;
  }
  return PIES_OK;
}

int pies_tick_begin(pies_solver_t* s, uint64_t* frame) {
  if (!s || !frame) return PIES_ERR_INVALID;
  *frame = 0;
  if (s->device == PIES_DEVICE_NONE) return fail(s, PIES_ERR_HIP, "host-only handle (PIES_DEVICE_NONE): there is no CPU solver");
  const uint64_t f = s->frameBegun + 1;
  if (s->frameAcquired && s->frameAcquired + 2 <= f)
    return fail(s, PIES_ERR_STATE, "pies_tick_begin: the frame two ticks back is still acquired (pies_export_release it first)");
  if (int rc = pies_tick_async(s)) return rc;  // a failed simulation still hands out (unchanged) frames
  if (int rc = export_prepare(s)) return rc;
  const uint32_t n = s->nd.n;
  const int b = static_cast<int>(f & 1u);
  if (n) {
    // d_export is free once the previous frame's D2H copy has read it; by now that copy finished long ago
    if (f > 1) HIP_TRY(s, hipStreamWaitEvent(s->stream, s->evCopied[b ^ 1], 0));
    HIP_TRY(s, hipMemcpyAsync(s->d_export, s->nd.pos, n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
  }
  HIP_TRY(s, hipEventRecord(s->evTick[b], s->stream));
  HIP_TRY(s, hipStreamWaitEvent(s->copyStream, s->evTick[b], 0));
  if (n) HIP_TRY(s, hipMemcpyAsync(s->h_export[b], s->d_export, n * sizeof(float4), hipMemcpyDeviceToHost, s->copyStream));
  HIP_TRY(s, hipEventRecord(s->evCopied[b], s->copyStream));
  s->frameBegun = f;
  *frame = f;
  return PIES_OK;
}

int pies_export_acquire(pies_solver_t* s, uint64_t frame, const float** pos4, uint32_t* n) {
  if (!s || !pos4) return PIES_ERR_INVALID;
  *pos4 = nullptr;
  if (n) *n = 0;
  if (frame == 0 || frame > s->frameBegun || frame + 2 <= s->frameBegun)
    return fail(s, PIES_ERR_STATE, "pies_export_acquire: only the last two frames begun are held");
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipEventSynchronize(s->evCopied[frame & 1u]));
  s->frameAcquired = frame;
  *pos4 = reinterpret_cast<const float*>(s->h_export[frame & 1u]);
  if (n) *n = s->nd.n;
  return PIES_OK;
}

int pies_export_release(pies_solver_t* s, uint64_t frame) {
  if (!s) return PIES_ERR_INVALID;
  if (s->frameAcquired == frame) s->frameAcquired = 0;
  return PIES_OK;
}

int pies_read_positions_strided(pies_solver_t* s, void* dst, uint64_t stride_bytes, uint32_t n) {
  if (!s || (!dst && n) || stride_bytes < 3 * sizeof(float)) return PIES_ERR_INVALID;
  if (n != s->nodeCount()) return fail(s, PIES_ERR_INVALID, "pies_read_positions_strided: n does not match the node count");
  if (s->device != PIES_DEVICE_NONE) {
    HIP_TRY(s, hipSetDevice(s->device));
    if (int rc = download_nodes(s, 1u)) return rc;
  }
  char* out = static_cast<char*>(dst);
  for (uint32_t i = 0; i < n; ++i) std::memcpy(out + static_cast<size_t>(i) * stride_bytes, &s->h_pos[3 * static_cast<size_t>(i)], 3 * sizeof(float));
  return PIES_OK;
}

int pies_failed(pies_solver_t* s, int* failed) {
  if (!s || !failed) return PIES_ERR_INVALID;
  if (int rc = poll_failure(s)) return rc;
  *failed = s->simFailed ? 1 : 0;
  return PIES_OK;
}

int pies_get_tri_contacts(pies_solver_t* s, uint32_t* ids, uint32_t capacity, uint32_t* count) {
  if (!s || !count) return PIES_ERR_INVALID;
  *count = 0;
  if (s->device == PIES_DEVICE_NONE || !s->pd.tri.counters) return PIES_OK;
  HIP_TRY(s, hipSetDevice(s->device));
  uint32_t m = 0;
  HIP_TRY(s, hipMemcpyAsync(&m, s->pd.tri.counters + 2, sizeof(m), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  *count = m;
  if (ids && m) {
    if (m > capacity) return fail(s, PIES_ERR_INVALID, "pies_get_tri_contacts: capacity too small");
    HIP_TRY(s, hipMemcpyAsync(ids, s->pd.tri.ids, static_cast<size_t>(m) * sizeof(uint4), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(s, hipStreamSynchronize(s->stream));
  }
  return PIES_OK;
}

int pies_collision_pairs(pies_solver_t* s, uint64_t* pairs) {
  if (!s || !pairs) return PIES_ERR_INVALID;
  *pairs = 0;
  if (!s->hash.counters) return PIES_OK;
  uint32_t v = 0;
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipMemcpyAsync(&v, s->hash.counters + 31, sizeof(v), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  HIP_TRY(s, hipMemsetAsync(s->hash.counters + 31, 0, sizeof(v), s->stream));
  *pairs = v;
  return PIES_OK;
}

int pies_debug_pair_state(pies_solver_t* s, float* slack, float* excursion, uint32_t* degree, uint32_t n) {
  if (!s || !s->pairs.ctl || n != s->pairs.n) return PIES_ERR_INVALID;
  std::vector<float4> node(4ull * n);
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  HIP_TRY(s, hipMemcpy(node.data(), s->pairs.node, node.size() * sizeof(float4), hipMemcpyDeviceToHost));
  if (excursion) HIP_TRY(s, hipMemcpy(excursion, s->pairs.exc, n * sizeof(float), hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < n; ++i) {
    if (slack) slack[i] = node[4ull * i + 2].w;
    if (degree) std::memcpy(&degree[i], &node[4ull * i + 3].y, sizeof(uint32_t));
  }
  return PIES_OK;
}

int pies_set_tuning(const char* name, const char* value) {
  if (!name || std::strncmp(name, "PIES_", 5) != 0) return PIES_ERR_INVALID;
  std::lock_guard<std::mutex> lock(tuning_mutex());
  if (value && value[0]) {
    tuning_values().push_back(std::make_unique<std::string>(value));
    tuning_current()[name] = tuning_values().back().get();
  } else {
    tuning_current().erase(name);
  }
  return PIES_OK;
}

int pies_set_collision_rounds(pies_solver_t* s, uint32_t rounds) {
  if (!s) return PIES_ERR_INVALID;
  if (rounds > 4096) return fail(s, PIES_ERR_INVALID, "pies_set_collision_rounds: at most 4096");
  s->pairRoundsPinned = true;  // an explicit count is kept (the library follows the passes by itself otherwise)
  if (rounds != s->pairRounds) {
    s->pairRounds = rounds;
    if (!s->sceneDirty) s->graphDirty = true;
  }
  return PIES_OK;
}

int pies_get_collision_health(pies_solver_t* s, uint32_t* rounds, uint32_t* pairs_listed, uint32_t* passes_repeated, uint32_t* passes_inexact) {
  if (!s) return PIES_ERR_INVALID;
  uint32_t v[kPairWords] = {0};
  if (s->pairs.ctl) {
    HIP_TRY(s, hipSetDevice(s->device));
    HIP_TRY(s, hipMemcpyAsync(v, s->pairs.ctl, sizeof(v), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(s, hipStreamSynchronize(s->stream));
  }
  if (rounds) *rounds = v[kPairRounds];
  if (pairs_listed) *pairs_listed = v[kPairEdges] / 2;
  if (passes_repeated) *passes_repeated = v[kPairRetries];
  if (passes_inexact) *passes_inexact = v[kPairInexact];
  return PIES_OK;
}

int pies_collision_stats(pies_solver_t* s, uint64_t* pairs, uint64_t* candidates) {
  if (!s) return PIES_ERR_INVALID;
  if (pairs) *pairs = 0;
  if (candidates) *candidates = 0;
  if (!s->hash.counters) return PIES_OK;
  uint32_t v[kHashCounters];
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipMemcpyAsync(v, s->hash.counters, sizeof(v), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  HIP_TRY(s, hipMemsetAsync(s->hash.counters + kCounterPairs, 0, sizeof(uint32_t), s->stream));
  HIP_TRY(s, hipMemsetAsync(s->hash.counters + kCounterCandidates, 0, 2 * sizeof(uint32_t), s->stream));
  if (pairs) *pairs = v[kCounterPairs];
  if (candidates) *candidates = static_cast<uint64_t>(v[kCounterCandidates]) | (static_cast<uint64_t>(v[kCounterCandidates + 1]) << 32);
  return PIES_OK;
}

int pies_count(const pies_solver_t* s, int what, uint32_t* out) {
  if (!s || !out) return PIES_ERR_INVALID;
  switch (what) {
    case PIES_POSITION: *out = (uint32_t)s->h_position.size(); break;
    case PIES_DISTANCE: *out = (uint32_t)s->h_distance.size(); break;
    case PIES_TET: *out = (uint32_t)s->h_tet.size(); break;
    case PIES_VOLUME: *out = (uint32_t)s->h_volume.size(); break;
    case PIES_BEND: *out = (uint32_t)s->h_bend.size(); break;
    case PIES_SHAPE: *out = (uint32_t)s->h_shape.size(); break;
    case PIES_GOAL: *out = (uint32_t)s->h_goal.size(); break;
    case PIES_TRIANGLES: *out = (uint32_t)(s->h_triangles.size() / 3); break;
    case PIES_LINES: *out = (uint32_t)s->h_lines.size(); break;
    case PIES_NODES: *out = s->nodeCount(); break;
    case PIES_SYSTEM_NNZ: *out = s->pd_nnz; break;
    case PIES_REST_SETS: *out = s->pdLocalPacked && s->d_pairDictIndex ? s->pairDictSets : 0u; break;
    case PIES_ROW_STENCILS: *out = s->pd.cg.rowStencil ? s->pdRowStencils : 0u; break;
    case PIES_PD_TILES: *out = s->pd.tiles.ntiles; break;
    case PIES_PD_TILE_RECORDS: *out = s->pd.tiles.ntiles ? s->pdTileRecords : 0u; break;
    case PIES_PD_CG_SINGLE: *out = s->opt.solver == PIES_SOLVER_PD && pd_single_cg(s) ? 1u : 0u; break;
    default: return PIES_ERR_INVALID;
  }
  return PIES_OK;
}

int pies_read_nodes(pies_solver_t* s, int what, float* out, uint32_t n) {
  if (!s || (!out && n)) return PIES_ERR_INVALID;
  if (n != s->nodeCount()) return fail(s, PIES_ERR_INVALID, "pies_read_nodes: n does not match the node count");
  if (s->device != PIES_DEVICE_NONE && what >= PIES_NODE_POSITION && what <= PIES_NODE_VELOCITY) {
    HIP_TRY(s, hipSetDevice(s->device));
    if (int rc = download_nodes(s, 1u << what)) return rc;  // the requested array only
  }
  const std::vector<float>* src = nullptr;
  switch (what) {
    case PIES_NODE_POSITION: src = &s->h_pos; break;
    case PIES_NODE_PREV_POSITION: src = &s->h_prev; break;
    case PIES_NODE_VELOCITY: src = &s->h_vel; break;
    case PIES_NODE_RADIUS: src = &s->h_radius; break;
    case PIES_NODE_INV_MASS: src = &s->h_invMass; break;
    default: return fail(s, PIES_ERR_INVALID, "pies_read_nodes: unknown selector");
  }
  if (!src->empty()) std::memcpy(out, src->data(), src->size() * sizeof(float));
  return PIES_OK;
}

int pies_write_nodes(pies_solver_t* s, int what, const float* in, uint32_t n) {
  if (!s || (!in && n)) return PIES_ERR_INVALID;
  if (n != s->nodeCount()) return fail(s, PIES_ERR_INVALID, "pies_write_nodes: n does not match the node count");
  if (int rc = scene_sync_host(s)) return rc;
  std::vector<float>* dst = nullptr;
  switch (what) {
    case PIES_NODE_POSITION: dst = &s->h_pos; break;
    case PIES_NODE_PREV_POSITION: dst = &s->h_prev; break;
    case PIES_NODE_VELOCITY: dst = &s->h_vel; break;
    case PIES_NODE_RADIUS: dst = &s->h_radius; break;
    case PIES_NODE_INV_MASS: dst = &s->h_invMass; break;
    default: return fail(s, PIES_ERR_INVALID, "pies_write_nodes: unknown selector");
  }
  if (!dst->empty()) std::memcpy(dst->data(), in, dst->size() * sizeof(float));
  s->hostNodesDirty = true;
  if (what == PIES_NODE_RADIUS && s->hash.counters && !s->sceneDirty) {  // the collision grid is sized from the radii
    uint64_t entries;
    bool fast;
    collision_grid_bound(s, entries, fast);
    if (entries + 64 > s->hash.maxEntries || fast != s->collideFast) s->sceneDirty = true;
  }
  return PIES_OK;
}

int pies_get_ids(const pies_solver_t* s, int type, uint32_t* out, uint32_t capacity) {
  if (!s || !out) return PIES_ERR_INVALID;
  size_t k = 0;
  auto put = [&](uint32_t v) { if (k < capacity) out[k] = v; ++k; };
  switch (type) {
    case PIES_POSITION: for (auto& c : s->h_position) put(c.id); break;
    case PIES_DISTANCE: for (auto& c : s->h_distance) { put(c.ids[0]); put(c.ids[1]); } break;
    case PIES_TET: for (auto& c : s->h_tet) for (uint32_t v : c.ids) put(v); break;
    case PIES_VOLUME: for (auto& c : s->h_volume) for (uint32_t v : c.ids) put(v); break;
    case PIES_BEND: for (auto& c : s->h_bend) for (uint32_t v : c.ids) put(v); break;
    case PIES_TRIANGLES: for (uint32_t v : s->h_triangles) put(v); break;
    case PIES_LINES: for (uint32_t v : s->h_lines) put(v); break;
    default: return PIES_ERR_INVALID;
  }
  return k <= capacity ? PIES_OK : PIES_ERR_INVALID;
}

int pies_get_group(const pies_solver_t* s, int type, uint32_t index, uint32_t* ids, uint32_t capacity, uint32_t* count) {
  if (!s || !count || (type != PIES_SHAPE && type != PIES_GOAL)) return PIES_ERR_INVALID;
  const std::vector<uint32_t>* v = nullptr;
  if (type == PIES_SHAPE && index < s->h_shape.size()) v = &s->h_shape[index].ids;
  if (type == PIES_GOAL && index < s->h_goal.size()) v = &s->h_goal[index].ids;
  if (!v) return PIES_ERR_INVALID;
  *count = static_cast<uint32_t>(v->size());
  if (ids) {
    if (v->size() > capacity) return PIES_ERR_INVALID;
    std::copy(v->begin(), v->end(), ids);
  }
  return PIES_OK;
}

int pies_get_rest(const pies_solver_t* s, int type, float* out, uint32_t capacity) {
  if (!s || !out) return PIES_ERR_INVALID;
  size_t k = 0;
  auto put = [&](float v) { if (k < capacity) out[k] = v; ++k; };
  switch (type) {
    case PIES_DISTANCE: for (auto& c : s->h_distance) put(c.target); break;
    case PIES_TET: for (auto& c : s->h_tet) for (float v : c.qinv) put(v); break;
    case PIES_VOLUME: for (auto& c : s->h_volume) for (float v : c.qinv) put(v); break;
    case PIES_BEND: for (auto& c : s->h_bend) put(c.angle); break;
    default: return PIES_ERR_INVALID;
  }
  return k <= capacity ? PIES_OK : PIES_ERR_INVALID;
}

int pies_get_order(pies_solver_t* s, int type, uint32_t* order, uint32_t capacity) {
  if (!s || !order || type < PIES_POSITION || type > PIES_BEND) return PIES_ERR_INVALID;
  if (s->sceneDirty)
    if (int rc = pies_finalize(s)) return rc;
  const Plan& pl = s->plan[type];
  if (pl.order.size() > capacity) return fail(s, PIES_ERR_INVALID, "pies_get_order: capacity too small");
  if (!pl.order.empty()) std::memcpy(order, pl.order.data(), pl.order.size() * sizeof(uint32_t));
  return PIES_OK;
}

int pies_get_batches(pies_solver_t* s, int type, uint32_t* offs, uint32_t capacity, uint32_t* n_batches) {
  if (!s || !n_batches || type < PIES_POSITION || type > PIES_BEND) return PIES_ERR_INVALID;
  if (s->sceneDirty)
    if (int rc = pies_finalize(s)) return rc;
  const Plan& pl = s->plan[type];
  *n_batches = (uint32_t)pl.batches.size();
  if (offs) {
    if (pl.batches.size() + 1 > capacity) return fail(s, PIES_ERR_INVALID, "pies_get_batches: capacity too small");
    for (size_t b = 0; b < pl.batches.size(); ++b) offs[b] = pl.batches[b].start;
    offs[pl.batches.size()] = pl.batches.empty() ? 0 : pl.batches.back().start + pl.batches.back().count;
  }
  return PIES_OK;
}

int pies_launch_counts(pies_solver_t* s, uint32_t* out) {
  if (!s || !out) return PIES_ERR_INVALID;
  if (s->sceneDirty)
    if (int rc = pies_finalize(s)) return rc;
  std::memcpy(out, s->launchCounts, sizeof(s->launchCounts));
  return PIES_OK;
}

// Node state kept aside while a measurement pass steps the solver, put back afterwards.
namespace {
struct StateGuard {
  pies_solver* s;
  float4 *pos = nullptr, *prev = nullptr, *vel = nullptr;
  bool ok = false;
  explicit StateGuard(pies_solver* s_) : s(s_) {
    const size_t bytes = static_cast<size_t>(s->nd.n) * sizeof(float4);
    if (!bytes) { ok = true; return; }
    if (hipMalloc((void**)&pos, bytes) != hipSuccess || hipMalloc((void**)&prev, bytes) != hipSuccess || hipMalloc((void**)&vel, bytes) != hipSuccess) return;
    ok = hipMemcpyAsync(pos, s->nd.pos, bytes, hipMemcpyDeviceToDevice, s->stream) == hipSuccess &&
         hipMemcpyAsync(prev, s->nd.prev, bytes, hipMemcpyDeviceToDevice, s->stream) == hipSuccess &&
         hipMemcpyAsync(vel, s->nd.vel, bytes, hipMemcpyDeviceToDevice, s->stream) == hipSuccess;
  }
  ~StateGuard() {
    const size_t bytes = static_cast<size_t>(s->nd.n) * sizeof(float4);
    if (ok && bytes) {
      (void)hipMemcpyAsync(s->nd.pos, pos, bytes, hipMemcpyDeviceToDevice, s->stream);
      (void)hipMemcpyAsync(s->nd.prev, prev, bytes, hipMemcpyDeviceToDevice, s->stream);
      (void)hipMemcpyAsync(s->nd.vel, vel, bytes, hipMemcpyDeviceToDevice, s->stream);
      (void)hipStreamSynchronize(s->stream);
    }
    if (pos) (void)hipFree(pos);
    if (prev) (void)hipFree(prev);
    if (vel) (void)hipFree(vel);
  }
};
}  // namespace

int pies_profile_substep(pies_solver_t* s, int kernel, uint32_t* launches, double* total_ms, uint64_t* units) {
  if (!s || kernel < 0 || kernel >= PIES_KERNEL_COUNT) return PIES_ERR_INVALID;
  if (s->device == PIES_DEVICE_NONE) return fail(s, PIES_ERR_HIP, "host-only handle");
  const bool isPD = s->opt.solver == PIES_SOLVER_PD;
  const bool pdClass = kernel >= PIES_KERNEL_PD_PREDICT && kernel <= PIES_KERNEL_PD_VELOCITY;
  if (isPD != pdClass) return fail(s, PIES_ERR_INVALID, "pies_profile_substep: kernel class of the other solver");
  if (int rc = ensure_ready(s)) return rc;
  if (launches) *launches = 0;
  if (total_ms) *total_ms = 0.0;
  if (units) *units = 0;
  if (s->nd.n == 0 || s->launchCounts[kernel] == 0) return PIES_OK;
  // a graph holding ONLY this class's launches of one substep, replayed back to back: the launches form one dependent
  // chain, so wall time / launches is the per-launch device time incl. the kernel boundary.  The working set of one
  // class usually fits the caches: these are isolated-replay times, NOT bandwidth figures (pies_profile_in_situ).
  uint64_t u = 0;
  const int reps = 5;
  auto enqueue_one = [&](uint64_t* units_) {
    if (isPD) enqueue_pd_substep(s, kernel, nullptr, units_);
    else enqueue_pbd_substep(s, kernel, nullptr, units_);
  };
  StateGuard keep(s);
  if (!keep.ok) return fail(s, PIES_ERR_HIP, "pies_profile_substep: no memory to keep the node state aside");
  struct Scope {  // everything the pass creates is released on every return path
    hipStream_t st;
    bool capturing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    ~Scope() {
      if (capturing) { hipGraph_t tmp = nullptr; (void)hipStreamEndCapture(st, &tmp); if (tmp) (void)hipGraphDestroy(tmp); }
      if (ev0) (void)hipEventDestroy(ev0);
      if (ev1) (void)hipEventDestroy(ev1);
      if (ge) (void)hipGraphExecDestroy(ge);
      if (g) (void)hipGraphDestroy(g);
    }
  } sc{s->stream};
  HIP_TRY(s, hipEventCreate(&sc.ev0));
  HIP_TRY(s, hipEventCreate(&sc.ev1));
  const bool eager = under_profiler();  // rocprofv3 7.2 segfaults on a second graph instantiation: launch eagerly there
  if (eager) {
    enqueue_one(&u);
    HIP_TRY(s, hipGetLastError());
  } else {
    HIP_TRY(s, hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    sc.capturing = true;
    enqueue_one(&u);
    sc.capturing = false;
    HIP_TRY(s, hipStreamEndCapture(s->stream, &sc.g));
    HIP_TRY(s, hipGraphInstantiate(&sc.ge, sc.g, nullptr, nullptr, 0));
    HIP_TRY(s, hipGraphLaunch(sc.ge, s->stream));  // warm
  }
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  HIP_TRY(s, hipEventRecord(sc.ev0, s->stream));
  for (int r = 0; r < reps; ++r) {
    if (eager) { enqueue_one(nullptr); HIP_TRY(s, hipGetLastError()); }
    else HIP_TRY(s, hipGraphLaunch(sc.ge, s->stream));
  }
  HIP_TRY(s, hipEventRecord(sc.ev1, s->stream));
  HIP_TRY(s, hipEventSynchronize(sc.ev1));
  float evMs = 0.0f;
  HIP_TRY(s, hipEventElapsedTime(&evMs, sc.ev0, sc.ev1));
  const uint32_t n = s->launchCounts[kernel];
  if (launches) *launches = n * reps;
  if (total_ms) *total_ms = evMs;
  if (units) *units = u * reps;
  return PIES_OK;
}

int pies_profile_in_situ(pies_solver_t* s, int kernel, uint32_t substeps, uint32_t* launches, double* total_ms, uint64_t* units,
                         double* bracket_overhead_ms) {
  if (!s || kernel < 0 || kernel >= PIES_KERNEL_COUNT || substeps == 0) return PIES_ERR_INVALID;
  if (s->device == PIES_DEVICE_NONE) return fail(s, PIES_ERR_HIP, "host-only handle");
  if (int rc = ensure_ready(s)) return rc;
  if (launches) *launches = 0;
  if (total_ms) *total_ms = 0.0;
  if (units) *units = 0;
  if (s->nd.n == 0) return PIES_OK;
  StateGuard keep(s);
  if (!keep.ok) return fail(s, PIES_ERR_HIP, "pies_profile_in_situ: no memory to keep the node state aside");
  Probe probe;
  probe.kernel = kernel;
  probe.stream = s->stream;
  struct Clear { pies_solver* s; Probe* p; ~Clear() { s->probe = nullptr; for (hipEvent_t e : p->events) (void)hipEventDestroy(e); } } clear{s, &probe};
  const bool isPD = s->opt.solver == PIES_SOLVER_PD;
  uint64_t u = 0;
  if (isPD) enqueue_pd_substep(s, -1, nullptr, nullptr);  // warm: caches and clocks as in a running simulation
  else enqueue_pbd_substep(s, -1, nullptr, nullptr);
  HIP_TRY(s, hipGetLastError());
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  s->probe = &probe;
  for (uint32_t r = 0; r < substeps; ++r) {
    if (isPD) enqueue_pd_substep(s, -1, nullptr, &u);
    else enqueue_pbd_substep(s, -1, nullptr, &u);
    HIP_TRY(s, hipGetLastError());
  }
  s->probe = nullptr;
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  if (probe.failed || (probe.used & 1u)) return fail(s, PIES_ERR_HIP, "pies_profile_in_situ: event bookkeeping failed");
  double ms = 0.0;
  for (size_t k = 0; k + 1 < probe.used; k += 2) {
    float e = 0.0f;
    HIP_TRY(s, hipEventElapsedTime(&e, probe.events[k], probe.events[k + 1]));
    ms += e;
  }
  if (launches) *launches = static_cast<uint32_t>(probe.used / 2);
  if (total_ms) *total_ms = ms;
  if (units) *units = u;
  if (bracket_overhead_ms) {
    // What a bracket costs by itself (the two event packets, the wait for the kernel's end-of-kernel cache write-back): the
    // same brackets around ONE and around TWO launches of an empty kernel; the difference is the empty kernel, the rest the
    // overhead.  Short kernels (a few microseconds) are dominated by it.
    const int reps = 24;
    double one = 0.0, two = 0.0;
    launch_noop(s->stream);  // first launch of the kernel: code object load, not part of the calibration
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    for (int count = 1; count <= 2; ++count) {
      probe.used = 0;
      for (int r = 0; r < reps; ++r) {
        probe.mark();
        for (int c = 0; c < count; ++c) launch_noop(s->stream);
        probe.mark();
      }
      HIP_TRY(s, hipGetLastError());
      HIP_TRY(s, hipStreamSynchronize(s->stream));
      double best = 1.0e30;  // the overhead is a floor: the smallest bracket is the one no other activity disturbed
      for (size_t k = 0; k + 1 < probe.used; k += 2) {
        float e = 0.0f;
        HIP_TRY(s, hipEventElapsedTime(&e, probe.events[k], probe.events[k + 1]));
        best = std::min(best, static_cast<double>(e));
      }
      (count == 1 ? one : two) = best;
    }
    *bracket_overhead_ms = std::max(0.0, one - std::max(0.0, two - one));
  }
  s->stale = 7u;
  return PIES_OK;
}

}  // extern "C"
