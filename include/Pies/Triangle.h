#pragma once
#include <cstdint>
namespace Pies {
struct Triangle {  // reference Include/Pies/Triangle.h
  uint32_t nodeIds[3];
};
}  // namespace Pies
