// Internals shared by the translation units behind the C ABI (capi.cpp: handle lifetime, upload, tick, state access;
// substep_graph.cpp: the substep as a launch sequence, graph capture, the adaptations that follow the scene; profiling.cpp: the
// timing passes; tuning.cpp: the registry of pies_set_tuning).
#pragma once
#include <vector>

#include "device_util.h"

namespace pies {

// ---- substep_graph.cpp ----
bool under_profiler();              // PIES_PROFILER_SAFE=1
void destroy_graph(pies_solver* s);
void collision_grid_bound(const pies_solver* s, uint64_t& entries, bool& fast);
void probe_mark(pies_solver* s, int k);
int collision_order(const pies_solver* s);
bool needs_grid_groups(const pies_solver* s);
uint32_t enqueue_collide(pies_solver* s, bool rearm = false);
void enqueue_layered_substep(pies_solver* s, int only, uint32_t* counts, uint64_t* units);
void enqueue_pbd_substep(pies_solver* s, int only, uint32_t* counts, uint64_t* units = nullptr);
bool pd_single_cg(const pies_solver* s);
void enqueue_pd_substep(pies_solver* s, int only = -1, uint32_t* counts = nullptr, uint64_t* units = nullptr);
void enqueue_substep(pies_solver* s, uint32_t* counts);
std::vector<uint32_t> ladder_rungs(const pies_solver* s);
uint32_t ladder_rung(const pies_solver* s, uint32_t budget);
bool uses_ladder(const pies_solver* s);
int select_pd_graph(pies_solver* s);
int capture_graph(pies_solver* s);
int adapt_pcg_budget(pies_solver* s);
int adapt_pair_rounds(pies_solver* s);
uint32_t sort_passes_for(uint32_t keyBits);
int adapt_sort_passes(pies_solver* s);
int poll_failure(pies_solver* s);
// ---- capi.cpp ----
void free_device(pies_solver* s);
bool tet_volume_pairs(const pies_solver* s);


}  // namespace pies

// brings HBM and the captured graph up to date with the host-side scene (capi.cpp; C linkage like its callers, not part of the ABI)
extern "C" int pies_internal_ensure_ready(pies_solver* s);
