// Pies::Node (reference Include/Pies/Node.h:8-20).  On the device the same state lives as
// struct-of-arrays in HBM; this AoS record is what hosts see.
#pragma once
#include <cstdint>

#include "glm_compat.h"

namespace Pies {
struct Node {
  uint32_t id = 0;
  glm::vec3 position{};
  glm::vec3 prevPosition{};
  glm::vec3 velocity{};
  glm::vec3 force{};
  float radius = 0.1f;
  float invMass = 1.0f;
};
}  // namespace Pies
