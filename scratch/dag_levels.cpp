// experiment: number of dependency levels of one PBD substep when the dependency DAG spans constraint types and
// iterations (EXACT order), against the per-sweep levelling the library uses today
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdint>
#include "pies_hip.h"
int main(int argc, char** argv) {
  uint32_t W = 20, H = 20, D = 250, iters = 20;
  if (argc > 3) { W = atoi(argv[1]); H = atoi(argv[2]); D = atoi(argv[3]); }
  pies_options_t o; pies_default_options(&o); o.solver = PIES_SOLVER_PBD; o.iterations = iters;
  pies_solver_t* s; if (pies_create(&o, PIES_DEVICE_NONE, &s)) return 1;
  float t[3] = {0, 5, 0}, v[3] = {0, 0, 0};
  pies_create_tet_box(s, W, H, D, t, 1.0f, v, 0.05f, 1.0f, 0);
  pies_create_box(s, W, H, D, t, 1.0f, 0.5f, 1, 0, 0);
  uint32_t n, nd, nt; pies_count(s, PIES_NODES, &n); pies_count(s, PIES_DISTANCE, &nd); pies_count(s, PIES_TET, &nt);
  std::vector<uint32_t> di(2ull * nd), ti(4ull * nt);
  pies_get_ids(s, PIES_DISTANCE, di.data(), di.size()); pies_get_ids(s, PIES_TET, ti.data(), ti.size());
  std::vector<uint32_t> od(nd), ot(nt);
  if (argc > 4) {  // coloured order
    pies_set_schedule(s, PIES_SCHEDULE_COLOURED);
    pies_get_order(s, PIES_DISTANCE, od.data(), nd); pies_get_order(s, PIES_TET, ot.data(), nt);
    std::vector<uint32_t> d2(di.size()), t2(ti.size());
    for (uint32_t k = 0; k < nd; ++k) { d2[2*k] = di[2*od[k]]; d2[2*k+1] = di[2*od[k]+1]; }
    for (uint32_t k = 0; k < nt; ++k) for (int j = 0; j < 4; ++j) t2[4*k+j] = ti[4*ot[k]+j];
    di.swap(d2); ti.swap(t2);
  }
  std::vector<uint32_t> lastW(n, 0), lastR(n, 0);
  uint32_t maxLevel = 0; uint64_t ops = 0;
  std::vector<uint32_t> hist;
  auto op = [&](const uint32_t* id, int stride, unsigned writeMask) {
    uint32_t lv = 0;
    for (int k = 0; k < stride; ++k) { lv = std::max(lv, lastW[id[k]]); if (writeMask >> k & 1) lv = std::max(lv, lastR[id[k]]); }
    ++lv;
    for (int k = 0; k < stride; ++k) { if (writeMask >> k & 1) lastW[id[k]] = lv; else lastR[id[k]] = std::max(lastR[id[k]], lv); }
    if (lv > maxLevel) { maxLevel = lv; hist.resize(lv + 1, 0); }
    ++hist[lv]; ++ops;
  };
  for (uint32_t it = 0; it < iters; ++it) {
    for (uint32_t c = 0; c < nd; ++c) op(&di[2 * c], 2, 1);
    for (uint32_t c = 0; c < nt; ++c) op(&ti[4 * c], 4, 15);
    for (uint32_t i = 0; i < n; ++i) op(&i, 1, 1);  // floor clamp
    printf("after iteration %u: %u levels\n", it + 1, maxLevel);
  }
  printf("ops %llu levels %u  avg ops/level %.0f\n", (unsigned long long)ops, maxLevel, double(ops) / maxLevel);
  uint32_t small = 0; for (uint32_t l = 1; l <= maxLevel; ++l) if (hist[l] < 256) ++small;
  printf("levels with < 256 ops: %u\n", small);
  return 0;
}
