// pies_set_tuning: process-wide tuning and diagnostic switches, by name (include/pies_hip.h).
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

}  // namespace pies

extern "C" {

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

}  // extern "C"
