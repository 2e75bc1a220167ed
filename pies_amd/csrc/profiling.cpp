// Timing passes of the C ABI: pies_profile_substep (isolated replay of one kernel class) and pies_profile_in_situ (whole
// substeps launched eagerly with HIP events around every launch of one class).
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

extern "C" {

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
  if (int rc = pies_internal_ensure_ready(s)) return rc;
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
  if (int rc = pies_internal_ensure_ready(s)) return rc;
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
