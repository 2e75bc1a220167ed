// The substep as a launch sequence - Solver::tickPBD (Src/Solver.cpp:40-160) / tickPD (:162-486) as a fixed list of kernel
// launches -, its capture into hipGraphs (one per CG budget and contact-row variant for PD), and the adaptations that follow
// the scene at host synchronisations: captured CG iterations, level launches of the pair order, radix passes of the node grid.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

#include "capi_internal.h"

using namespace pies;

namespace pies {

// PIES_PROFILER_SAFE=1 (set by the profiling scripts): rocprofv3 7.2 on this pool segfaults when tens of
// thousands of graph kernel nodes are queued without a synchronisation, or when a graph is destroyed
// while it traces.  In this mode every tick is followed by a stream synchronisation and graphs are only
// released with the process.
bool under_profiler() {
  static const bool v = [] { const char* e = std::getenv("PIES_PROFILER_SAFE"); return e && e[0] == '1'; }();
  return v;
}

void destroy_graph(pies_solver* s) {
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
void build_layer_program(const pies_solver* s, std::vector<LayerItem>& prog) {
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
void collision_grid_bound(const pies_solver* s, uint64_t& entries, bool& fast) {
  entries = 0;
  fast = true;
  for (float r : s->h_radius) {
    const float twoR = 2.0f * ((r + 0.5f) / s->opt.gridSpacing);
    const uint64_t len = std::isfinite(twoR) && twoR >= 0.0f && twoR < 64.0f ? std::min<uint64_t>(50, 1 + static_cast<uint64_t>(std::ceil(twoR))) : 0;
    entries += len * len * len;
    if (!(twoR <= 1.0f)) fast = false;
  }
}
void probe_mark(pies_solver* s, int k) {
  if (s->probe && s->probe->kernel == k) s->probe->mark();
}
// PIES_COLLISION_ORDER_*: the reference's loop for ranges wider than two cells per axis (the parallel orders need 2R <= 1),
// under schedule EXACT, or when asked for; otherwise the pair order (or, when asked for, the group order of rounds 1-2)
int collision_order(const pies_solver* s) {
  int order = s->collisionOrderFlag >= 0 ? s->collisionOrderFlag
                                          : (s->schedule == PIES_SCHEDULE_EXACT ? PIES_COLLISION_ORDER_REFERENCE : PIES_COLLISION_ORDER_PAIRS);
  // the group order needs ranges of at most two cells per axis (2R <= 1); the pair order lists node by node beyond that
  if (order == PIES_COLLISION_ORDER_GROUPS && !s->collideFast) order = PIES_COLLISION_ORDER_REFERENCE;
  return order;
}
// The reference's order runs by dependency levels of turns (pair_kernels.hip: launch_collide_turns) from 1 024 nodes on; below that -
// and with PIES_REFERENCE_TURNS=0 - as the single chain of k_collide_reference, which is also the turns' fallback.
bool reference_by_turns(const pies_solver* s) {
  if (const char* e = tuning_env("PIES_REFERENCE_TURNS")) return e[0] != '0' && s->pairs.turnCnt != nullptr;
  return s->nd.n >= 1024u && s->pairs.turnCnt != nullptr;
}
// (the group lists of k_grid_groups: the group order's, and the list kernels' for ranges wider than two cells per axis)
bool needs_grid_groups(const pies_solver* s) {
  const int order = collision_order(s);
  if (order == PIES_COLLISION_ORDER_GROUPS) return true;
  if (order == PIES_COLLISION_ORDER_REFERENCE && !reference_by_turns(s)) return false;
  return !s->collideFast;
}
uint32_t enqueue_collide(pies_solver* s, bool rearm) {
  switch (collision_order(s)) {
    case PIES_COLLISION_ORDER_REFERENCE:
      if (reference_by_turns(s))
        return launch_collide_turns(s->stream, s->hash, s->pairs, s->nd, s->opt.gridSpacing, s->opt.friction, s->opt.staticFrictionThreshold, s->pairRounds);
      return launch_collide_reference(s->stream, s->hash, s->nd, s->opt.gridSpacing, s->opt.friction, s->opt.staticFrictionThreshold);
    case PIES_COLLISION_ORDER_GROUPS:
      return launch_collide(s->stream, s->hash, s->nd, s->opt.gridSpacing, s->opt.friction, s->opt.staticFrictionThreshold, rearm);
    default:
      return launch_collide_pairs(s->stream, s->hash, s->pairs, s->nd, s->opt.gridSpacing, s->opt.friction, s->opt.staticFrictionThreshold, s->pairRounds);
  }
}

void enqueue_layered_substep(pies_solver* s, int only, uint32_t* counts, uint64_t* units) {
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
void enqueue_pbd_substep(pies_solver* s, int only, uint32_t* counts, uint64_t* units) {
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
bool pd_single_cg(const pies_solver* s) {
  // (the contact-heavy variant: the rows' contact parts are the merged rows k_contact_csr builds every substep - a gather of a
  // handful of distinct columns by the row's lane; PIES_PD_CG_SINGLE_ROWS=0 keeps the two-launch form with its extra workgroups there)
  return s->pdSingleCg && (!s->pd.cg.useCAp || s->pdSingleCgRows) && s->pd.cg.lanesPerRow == 1u;
}

// One PD substep as a launch sequence (Solver.cpp:228-485).  `only` >= 0 (profile pass) launches one kernel
// class of the tetrahedral pipeline; units tallies the work items of the launches made.
void enqueue_pd_substep(pies_solver* s, int only, uint32_t* counts, uint64_t* units) {
  hipStream_t st = s->stream;
  const float h = s->opt.fixedTimestepSize / s->opt.timeSubsteps;
  const PdArrays& pd = s->pd;
  int cur = -1;
  auto ON = [&](int k) { const bool on = only < 0 || only == k; cur = k; if (on) probe_mark(s, k); return on; };
  auto C = [&](int k, uint32_t n = 1) { if (k != PIES_KERNEL_PD_SPMV && k != PIES_KERNEL_PD_CG_UPDATE) probe_mark(s, k); if (counts) counts[k] += n; };
  auto U = [&](uint64_t u) { if (units && (only >= 0 || (s->probe && s->probe->kernel == cur))) *units += u; };
  const uint32_t nDist = (uint32_t)s->h_distance.size(), nTet = (uint32_t)s->h_tet.size(), nVol = (uint32_t)s->h_volume.size();
  if (ON(PIES_KERNEL_PD_PREDICT)) { launch_pd_predict(st, s->nd, pd, h, s->opt.floorHeight + s->opt.collisionThickness, pd.tri.nt != 0 && only < 0); U(s->nd.n); }
  C(PIES_KERNEL_PD_PREDICT);
  const bool tri = pd.tri.nt != 0;
  // the statistics of the substep's last solve are closed by an extra workgroup of the floor-snap launch when there is one
  const bool statsInStabilize = only < 0 && s->opt.collisionStabilizationIterations > 0 && s->opt.iterations > 0 && s->nd.n != 0;
  if (tri && only < 0) {  // Solver.cpp:240, 245-248: detection, contact list, their blocks of the system matrix
    const char* side = tuning_env("PIES_TRI_SIDE");  // diagnostics: 0 = always in line, 1 = always beside
    s->triLevelsForked = side ? side[0] != '0' : s->triFastRows;
    const bool levelsInLine = !s->triLevelsForked && pd.cg.useCAp == 0;  // the contact-light variant computes them with the list
    launch_tri_detect(st, pd.tri, s->nd, pd.kdiag, pd.cg.cdiag, pd.cg.dinv, s->opt.collisionThresholdDistance, s->opt.collisionThickness,
                      pd.cg.useCAp != 0, levelsInLine);
    // The dependency levels of the list (one workgroup, up to 1 ms with tens of thousands of contacts) are only needed by
    // the sequential passes behind the local/global iterations: a second branch of the substep, joined there.
    // Only in the contact-heavy graph variant: a fork and join inside a hipGraph costs about 100 us per replay (measured:
    // config 3, no contact, 1 257 -> 1 120 substeps/s with the branch; 29k contacts, 197 -> 234 with it).
    if (s->triLevelsForked) {
      (void)hipEventRecord(s->evFork, st);
      (void)hipStreamWaitEvent(s->sideStream, s->evFork, 0);
      launch_tri_levels(s->sideStream, pd.tri);
      (void)hipEventRecord(s->evJoin, s->sideStream);
    } else if (!levelsInLine) {
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
      launch_pd_local_node_pair(st, s->nd.pos, s->nd.radius, s->d_np_ids, pd.contrib + s->slotBase[5], (uint32_t)s->h_nodePair.size());
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
      // a probed solve never takes the converged early exit: every bracketed launch does a full SpMV / vector update
      // (PIES_PCG_NEVER_EXIT=1: the same for every solve of a captured substep - a diagnostics switch for profiler passes, so that a
      // trace of a body at rest holds working CG launches instead of early exits; a converged column stands still, cg1_scalars)
      const char* neverExitEnv = tuning_env("PIES_PCG_NEVER_EXIT");  // (read when the substep is captured)
      const bool neverExitTuning = neverExitEnv && neverExitEnv[0] == '1';
      const bool probed = neverExitTuning || (s->probe && (s->probe->kernel == PIES_KERNEL_PD_SPMV || s->probe->kernel == PIES_KERNEL_PD_CG_UPDATE));
      auto hook = s->probe ? [](void* ctx, int cls) { probe_mark(static_cast<pies_solver*>(ctx), cls); } : (void (*)(void*, int))nullptr;
      if (fuseRhs && units && s->probe && s->probe->kernel == PIES_KERNEL_PD_RHS) *units += s->nd.n;
      if (single) launch_pd_solve1(st, s->nd, pd, (int)s->pcgBudget, s->pcgTol, it == 0, lastSolve, fuseRhs, probed, hook, s, overflow);
      else launch_pd_solve(st, s->nd, pd, (int)s->pcgBudget, s->pcgTol, -1, it == 0, lastSolve, probed, hook, s, overflow);
      if (probed && s->probe && units) *units += (uint64_t)s->nd.n * s->pcgBudget;
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
    if (only < 0) launch_pd_node_pair_friction(st, s->nd.pos, s->nd.vel, s->nd.radius, s->d_np_ids, (uint32_t)s->h_nodePair.size(), s->opt.friction,
                                               s->opt.staticFrictionThreshold);  // Solver.cpp:398-428 comes before the triangles' (:431-471)
    launch_tri_friction(st, pd.tri, s->nd, s->opt.friction, s->opt.staticFrictionThreshold, pd.nstatic);
  } else {
    if (only < 0 && s->opt.collisionStabilizationIterations > 0) launch_pd_stabilize(st, s->nd, pd, statsInStabilize, (int)s->pcgBudget, s->pcgTol, pd_single_cg(s));  // the floor snap is idempotent
    if (ON(PIES_KERNEL_PD_VELOCITY)) {
      launch_pd_velocity(st, s->nd, pd, h, s->opt.damping, s->opt.gravity, s->opt.friction, s->opt.staticFrictionThreshold, true);
      U(s->nd.n);
    }
    if (only < 0) launch_pd_node_pair_friction(st, s->nd.pos, s->nd.vel, s->nd.radius, s->d_np_ids, (uint32_t)s->h_nodePair.size(), s->opt.friction,
                                               s->opt.staticFrictionThreshold);
  }
  C(PIES_KERNEL_PD_VELOCITY);
}

void enqueue_substep(pies_solver* s, uint32_t* counts) {
  if (s->opt.solver == PIES_SOLVER_PD) enqueue_pd_substep(s, -1, counts);
  else enqueue_pbd_substep(s, -1, counts);
}

// The budget ladder: captured CG iterations per solve of the graphs instantiated together - 1, 2, 3, 4, 6, 8, 12, 16, ... up to
// the ceiling of pies_set_pcg (the one-launch-per-iteration form pays one launch per unused iteration, so the low rungs are
// close together); ladder_rung = the rung that holds `budget` iterations
std::vector<uint32_t> ladder_rungs(const pies_solver* s) {
  std::vector<uint32_t> r;
  if (s->pcgMaxIters > 1) r.push_back(1);  // (a body at rest: the solves end at their first look at the residual)
  for (uint32_t b = 2; b < s->pcgMaxIters; b *= 2) {
    r.push_back(b);
    if (b + b / 2 < s->pcgMaxIters) r.push_back(b + b / 2);
  }
  r.push_back(s->pcgMaxIters);
  return r;
}
uint32_t ladder_rung(const pies_solver* s, uint32_t budget) {
  for (uint32_t r : ladder_rungs(s))
    if (r >= budget) return r;
  return s->pcgMaxIters;
}
bool uses_ladder(const pies_solver* s) {
  if (s->opt.solver != PIES_SOLVER_PD || s->pcgPinned || under_profiler()) return false;
  const char* e = tuning_env("PIES_NO_GRAPH");
  return !(e && e[0] == '1');
}
// makes the ladder entry of (pcgBudget, triFastRows) the graph pies_tick launches
int select_pd_graph(pies_solver* s) {
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

int capture_graph(pies_solver* s) {
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
int adapt_pcg_budget(pies_solver* s) {
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
    // A budget far above what the solves use comes down in two steps: to TWICE (most used + spare) after three calm
    // synchronisations, to (most used + spare) after eight more.  (Round 4: the budget is switched between instantiated graphs and
    // a solve that needs more than was captured goes on inside its last launch, so coming down early costs a slower frame at
    // worst; staying at 32 for fourteen frames after a contact onset cost 30 launches x 2.5 us per solve that found nothing to do.)
    s->pcgRecent[s->pcgCalm % 3u] = used;  // (the last three synchronisations: what the solves use NOW, whatever the window saw before)
    const uint32_t recent = std::max(s->pcgRecent[0], std::max(s->pcgRecent[1], s->pcgRecent[2]));
    const uint32_t spare3 = std::max(pd_single_cg(s) ? 1u : 2u, (recent + 2u) / 3u);
    if (s->pcgCalm >= 3 && 2u * (recent + spare3) < budget && !s->pdLadder.empty()) { budget = 2u * (recent + spare3); restart = true; }
    else if (s->pcgCalm >= 8 && s->pcgWindowMax + spare8 < budget) { budget = s->pcgWindowMax + spare8; restart = true; }
    else if (s->pcgCalm >= 24) {
      if (s->pcgWindowMax + spare24 < budget) budget = s->pcgWindowMax + spare24;
      restart = true;  // the window never looks back further than 24 synchronisations
    }
    if (restart) { s->pcgCalm = 0; s->pcgWindowMax = 0; s->pcgRecent[0] = s->pcgRecent[1] = s->pcgRecent[2] = 0; }
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
int adapt_pair_rounds(pies_solver* s) {
  if (!s->pairs.ctl || s->pairRoundsPinned || !s->graphExec || s->sceneDirty || under_profiler()) return PIES_OK;
  if (s->opt.solver != PIES_SOLVER_PBD || !s->nodeCollisions) return PIES_OK;
  uint32_t deepest = 0;
  HIP_TRY(s, hipMemcpyAsync(&deepest, s->pairs.ctl + kPairDeepest, sizeof(deepest), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  if (deepest == 0) return PIES_OK;  // no pass since the last look
  HIP_TRY(s, hipMemsetAsync(s->pairs.ctl + kPairDeepest, 0, sizeof(uint32_t), s->stream));
  uint32_t rounds = s->pairRounds;
  // (the reference's order by turns runs deeper than the pair order: ~850 levels per pass of config 4)
  const uint32_t cap = collision_order(s) == PIES_COLLISION_ORDER_REFERENCE ? 4096u : 1024u;
  // (round 5: eight more than the deepest pass seen, in steps of 8 - a quarter more on top of that were 24 launches per settled pass of
  // config 4 that found nothing to do, 40 with the repeat's half again: VERDICT r4 item 6)
  const uint32_t want = std::min(cap, ((deepest + 8u + 7u) / 8u) * 8u);
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
uint32_t sort_passes_for(uint32_t keyBits) { return std::max(1u, std::min(6u, (keyBits + 5u + 10u) / 11u)); }
int adapt_sort_passes(pies_solver* s) {
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

int poll_failure(pies_solver* s) {
  uint32_t* flagWord = s->hash.counters ? s->hash.counters + 3 : s->pd.tri.counters ? s->pd.tri.counters + 3 : nullptr;
  if (s->simFailed || !flagWord || s->device == PIES_DEVICE_NONE) return PIES_OK;
  uint32_t flag = 0;
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipMemcpyAsync(&flag, flagWord, sizeof(flag), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  if (flag) {  // like the reference's latch (Solver.cpp:741-755, 853-856): tick becomes a no-op
    s->simFailed = true;
    s->error = flag & 2    ? "node-node collision grid overflow (more cells or entries than reserved)"
               : flag & 4  ? "runaway pile-up: a node-node pass that only the sequential loop can run (more than 1024 nodes within reach of one node, or more than 2048 in a cell under the group order) would cost more candidate tests than PIES_FALLBACK_VISITS allows (1e9: about a minute)"
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
