// C ABI (include/pies_hip.h): handle lifetime, upload to HBM, tick, state access (the launch sequence of a substep and its
// graph capture: substep_graph.cpp; timing passes: profiling.cpp; pies_set_tuning: tuning.cpp).
// The substep itself is Solver::tickPBD (Src/Solver.cpp:40-160) / tickPD (:162-486) re-expressed as a
// fixed sequence of kernel launches captured once into a hipGraph: at 100k particles a conflict-free
// batch runs for a few microseconds, so un-graphed launches would be host-bound.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "capi_internal.h"

#include <map>
#include <memory>
#include <mutex>

using namespace pies;

namespace pies {

// strain and volume constraints added pairwise over the same elements (createTetBox and addTriMeshVolume add them that way,
// PrimitiveUtilities.cpp:401-514): one gather, one SVD, one tile plan for both
bool tet_volume_pairs(const pies_solver* s) {
  bool paired = !s->h_tet.empty() && s->h_tet.size() == s->h_volume.size();
  const bool planned = s->plan[PIES_TET].order.size() == s->h_tet.size();  // (PD: host order; a handle that has not been finalized has no plan yet)
  for (size_t k = 0; paired && k < s->h_tet.size(); ++k) {
    const HostTet &a = s->h_tet[planned ? s->plan[PIES_TET].order[k] : k], &b = s->h_volume[k];
    paired = std::memcmp(a.ids, b.ids, sizeof(a.ids)) == 0 && std::memcmp(a.qinv, b.qinv, sizeof(a.qinv)) == 0;
  }
  if (const char* e = tuning_env("PIES_NO_TET_PAIRS"); e && e[0] == '1') paired = false;
  return paired;
}

void free_device(pies_solver* s) {
  destroy_graph(s);
  for (void* p : s->allocations) (void)hipFree(p);
  s->allocations.clear();
  s->nd = NodeArrays{nullptr, nullptr, nullptr, nullptr, 0};
  s->d_pack = nullptr;
  s->d_pc_id = nullptr; s->d_pc_tw = nullptr;
  s->d_dc_ids = nullptr; s->d_dc_rw = nullptr;
  s->d_tc_ids = nullptr; s->d_tc_q0 = s->d_tc_q1 = s->d_tc_q2 = nullptr;
  s->d_bc_ids = nullptr; s->d_bc_aw = nullptr;
  s->d_np_ids = nullptr;
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
  if (n == 0 || !s->h_stage || !s->d_pack) { s->stale = 0; return PIES_OK; }
  float* dst[3] = {s->h_pos.data(), s->h_prev.data(), s->h_vel.data()};
  const float4* src[3] = {s->nd.pos, s->nd.prev, s->nd.vel};
  for (int a = 0; a < 3; ++a) {
    if (!(mask & (1u << a))) continue;
    // packed on the device: 12 bytes per node cross the bus, and the mirror is one memcpy from the pinned stage (measured on
    // config 2: 654 ticks/s against 637 with four floats per node and an unpacking loop; the asynchronous export stays the
    // fast way out, 680)
    launch_pack_xyz(s->stream, src[a], s->d_pack, n);
    HIP_TRY(s, hipMemcpyAsync(s->h_stage, s->d_pack, 3ull * n * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    std::memcpy(dst[a], s->h_stage, 3ull * n * sizeof(float));
    s->stale &= ~(1u << a);
  }
  return PIES_OK;
}

int scene_sync_host(pies_solver* s) {
  if (s->device == PIES_DEVICE_NONE) return PIES_OK;
  if (hipSetDevice(s->device) != hipSuccess) return fail(s, PIES_ERR_HIP, "hipSetDevice failed");
  return download_nodes(s);
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
  s->h_position.clear(); s->h_distance.clear(); s->h_tet.clear(); s->h_volume.clear(); s->h_bend.clear(); s->h_nodePair.clear();
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
    HIP_TRY(s, hipMalloc(&p, 3ull * n * sizeof(float))); s->allocations.push_back(p); s->d_pack = (float*)p;
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
  if (isPD && !s->h_nodePair.empty()) {  // the node-pair extension (container order: a pair's slot is its index)
    std::vector<uint2> id(s->h_nodePair.size());
    for (size_t k = 0; k < id.size(); ++k) id[k] = make_uint2(s->h_nodePair[k].ids[0], s->h_nodePair[k].ids[1]);
    if (int rc = upload(s, id, &s->d_np_ids)) return rc;
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
      if (int rc = dev_alloc(s, n, &P.turnCnt, true)) return rc;
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
    s->tetVolumePaired = tet_volume_pairs(s);
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
int pies_internal_ensure_ready(pies_solver* s) {
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
  if (int rc = pies_internal_ensure_ready(s)) return rc;
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
    if (int rc = pies_internal_ensure_ready(s)) return rc;
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

int pies_get_tri_grid_stats(pies_solver_t* s, uint32_t out[8]) {
  if (!s || !out) return PIES_ERR_INVALID;
  for (int i = 0; i < 8; ++i) out[i] = 0;
  if (s->device == PIES_DEVICE_NONE || !s->pd.tri.counters) return PIES_OK;
  HIP_TRY(s, hipSetDevice(s->device));
  uint32_t c[16];
  HIP_TRY(s, hipMemcpyAsync(c, s->pd.tri.counters, sizeof(c), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  uint32_t lists[64 * 16];
  HIP_TRY(s, hipMemcpyAsync(lists, s->pd.tri.workCnt, sizeof(lists), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  for (int i = 0; i < 64; ++i) out[0] += lists[16 * i];
  out[1] = c[9];
  for (int k = 0; k < 3; ++k) { out[2 + k] = c[10 + k]; out[5 + k] = c[13 + k]; }
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

int pies_get_collision_fallbacks(pies_solver_t* s, uint32_t* passes) {
  if (!s || !passes) return PIES_ERR_INVALID;
  *passes = 0;
  if (s->pairs.ctl) {
    HIP_TRY(s, hipSetDevice(s->device));
    HIP_TRY(s, hipMemcpyAsync(passes, s->pairs.ctl + kPairFallbacks, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(s, hipStreamSynchronize(s->stream));
  }
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
    case PIES_NODE_PAIRS: *out = (uint32_t)s->h_nodePair.size(); break;
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
    case PIES_PD_WINDOW_ENTRIES: *out = s->pd.cg.wRows ? s->pdWindowEntries : 0u; break;
    case PIES_PD_WINDOW_HALO: *out = s->pd.cg.wRows ? s->pdWindowHalo : 0u; break;
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
    case PIES_NODE_PAIRS: for (auto& c : s->h_nodePair) { put(c.ids[0]); put(c.ids[1]); } break;
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

// The tile plan of the PD strain + volume local step (pd_tiles.cpp), from the host-side scene alone: works on host-only handles.
int pies_get_pd_tile_plan(pies_solver_t* s, uint32_t* n_tiles, uint32_t* info, uint32_t* node, uint32_t* elem, uint32_t* local, uint16_t* nptr,
                          uint16_t* inc, uint32_t tile_capacity) {
  if (!s || !n_tiles) return PIES_ERR_INVALID;
  const bool paired = s->tetVolumePaired;
  s->tetVolumePaired = tet_volume_pairs(s);
  PdTilePlan plan;
  const bool ok = pd_plan_tiles(s, plan);
  s->tetVolumePaired = paired;
  *n_tiles = ok ? static_cast<uint32_t>(plan.info.size()) : 0u;
  if (!ok || !info) return PIES_OK;
  if (plan.info.size() > tile_capacity) return fail(s, PIES_ERR_INVALID, "pies_get_pd_tile_plan: capacity too small");
  std::memcpy(info, plan.info.data(), plan.info.size() * sizeof(uint32_t));
  if (node) std::memcpy(node, plan.node.data(), plan.node.size() * sizeof(uint32_t));
  if (elem) std::memcpy(elem, plan.elem.data(), plan.elem.size() * sizeof(uint32_t));
  if (local) std::memcpy(local, plan.local.data(), plan.local.size() * sizeof(uint32_t));
  if (nptr) std::memcpy(nptr, plan.nptr.data(), plan.nptr.size() * sizeof(uint16_t));
  if (inc) std::memcpy(inc, plan.inc.data(), plan.inc.size() * sizeof(uint16_t));
  return PIES_OK;
}

int pies_launch_counts(pies_solver_t* s, uint32_t* out) {
  if (!s || !out) return PIES_ERR_INVALID;
  if (s->sceneDirty)
    if (int rc = pies_finalize(s)) return rc;
  std::memcpy(out, s->launchCounts, sizeof(s->launchCounts));
  return PIES_OK;
}
}  // extern "C"


#ifdef PIES_BOUNDS
// The diagnostic build's record of device-side bounds violations (dev_math.h PIES_IN_BOUNDS), per kernel file: out[2 k] = the first
// failing site, out[2 k + 1] = how many, for k = layer, pd, cg1.  Reading clears.  tests/conftest.py asks after the session.
extern "C" int pies_exp_bounds_layer(unsigned int*);
extern "C" int pies_exp_bounds_pd(unsigned int*);
extern "C" int pies_exp_bounds_cg1(unsigned int*);
extern "C" int pies_exp_bounds_report(unsigned int* out6) {
  return pies_exp_bounds_layer(out6) | pies_exp_bounds_pd(out6 + 2) | pies_exp_bounds_cg1(out6 + 4);
}
#endif
