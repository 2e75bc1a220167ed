// One rotation of the product's one-sided Jacobi SVD (dev_math.h jacobi_pair<0,1>), isolated for an ISA listing
// (tools/rotation_isa.py -> profiles/r04_svd_isa_counts.txt)
#include <hip/hip_runtime.h>
#include "../pies_amd/csrc/dev_math.h"
using namespace pies;
extern "C" __global__ void one_rotation(float* io) {
  Svd3 d;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) { d.b[i][k] = io[(3 * i + k) * 64 + t]; d.v[i][k] = io[(9 + 3 * i + k) * 64 + t]; }
  asm volatile("; ---- rotation begin");
  jacobi_pair<0, 1>(d);
  asm volatile("; ---- rotation end");
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) { io[(3 * i + k) * 64 + t] = d.b[i][k]; io[(9 + 3 * i + k) * 64 + t] = d.v[i][k]; }
}
