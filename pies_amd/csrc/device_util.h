// Small HBM helpers shared by the host translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "solver_state.h"

namespace pies {

#define HIP_TRY(s, expr)                                                                    \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess)                                                                  \
      return ::pies::fail((s), PIES_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
  } while (0)

template <class T> int dev_alloc(pies_solver* s, size_t count, T** d, bool zero = false) {
  *d = nullptr;
  if (count == 0) return PIES_OK;
  void* p = nullptr;
  HIP_TRY(s, hipMalloc(&p, count * sizeof(T)));
  s->allocations.push_back(p);
  if (zero) HIP_TRY(s, hipMemsetAsync(p, 0, count * sizeof(T), s->stream));
  *d = static_cast<T*>(p);
  return PIES_OK;
}

// Allocates and copies; synchronises the stream so the caller's staging vector may die.
template <class T> int upload(pies_solver* s, const std::vector<T>& h, T** d) {
  if (int rc = dev_alloc(s, h.size(), d)) return rc;
  if (h.empty()) return PIES_OK;
  HIP_TRY(s, hipMemcpyAsync(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return PIES_OK;
}

int pd_build(pies_solver* s);         // pd_setup.cpp
int pd_upload_goals(pies_solver* s);  // pd_setup.cpp
bool pd_plan_tiles(const pies_solver* s, PdTilePlan& out);  // pd_tiles.cpp; false: the scene keeps per-(element, node) records

}  // namespace pies
