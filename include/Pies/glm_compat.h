// glm at the boundary: Pies::Solver's public signatures use glm::vec3 / glm::mat4 (reference
// Include/Pies/Solver.h:76-116).  Hosts that have glm get the real types; otherwise a minimal
// layout-compatible stand-in is provided (vec3 = 3 packed floats, mat4 = 16 floats column-major), which is
// all the boundary needs -- the solver itself never does glm arithmetic on the host.
#pragma once

#if __has_include(<glm/glm.hpp>)
#include <glm/glm.hpp>
#else
namespace glm {
struct vec3 {
  float x = 0.f, y = 0.f, z = 0.f;
  vec3() = default;
  vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
  explicit vec3(float s) : x(s), y(s), z(s) {}
  float& operator[](int i) { return (&x)[i]; }
  const float& operator[](int i) const { return (&x)[i]; }
};
struct vec4 {
  float x = 0.f, y = 0.f, z = 0.f, w = 0.f;
  float& operator[](int i) { return (&x)[i]; }
  const float& operator[](int i) const { return (&x)[i]; }
};
struct mat4 {
  vec4 c[4];  // column-major, like glm
  mat4() = default;
  explicit mat4(float d) { c[0].x = d; c[1].y = d; c[2].z = d; c[3].w = d; }
  vec4& operator[](int i) { return c[i]; }
  const vec4& operator[](int i) const { return c[i]; }
};
}  // namespace glm
#endif
