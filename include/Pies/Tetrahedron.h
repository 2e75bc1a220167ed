#pragma once
#include <cstdint>
namespace Pies {
struct Tetrahedron {  // reference Include/Pies/Tetrahedron.h
  uint32_t nodeIds[4];
};
}  // namespace Pies
