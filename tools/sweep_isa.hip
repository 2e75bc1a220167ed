// One whole decomposition (svd3 vs svd3_pk) isolated for instruction counts (profiles/r04_svd_isa_counts.txt)
#include <hip/hip_runtime.h>
#include "../pies_amd/csrc/dev_math.h"
using namespace pies;
template <bool PK> __device__ __forceinline__ void run(float* io) {
  float a[3][3];
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) a[i][k] = io[(3 * i + k) * 64 + t];
  Svd3 d;
  if (PK) svd3_pk(a, d); else svd3(a, d);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) { io[(3 * i + k) * 64 + t] = d.b[i][k]; io[(9 + 3 * i + k) * 64 + t] = d.v[i][k]; }
  io[18 * 64 + t] = d.s[0] + d.s[1] + d.s[2];
}
extern "C" __global__ void svd_scalar(float* io) { run<false>(io); }
extern "C" __global__ void svd_packed(float* io) { run<true>(io); }
