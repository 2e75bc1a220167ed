// Micro-benchmark of the PBD tetrahedral projection (pbd_project.h tet_core) - development aid, not part of the product.
// Two decompositions: VARIANT 0 = the product (closed-form start, round 6), VARIANT 2 = the plain one-sided Jacobi iteration from
// V = I (rounds 1-5).  Two element classes:
//   healthy    a sheared, stretched element (what a perturbed rest state looks like: the plain iteration takes 3-4 sweeps)
//   flattened  all four nodes in the plane y = const (BASELINE config 2 after its first tick: every element lies on the floor,
//              one column of F is exactly zero)
// One workgroup per compute unit runs STEPS dependent projections per lane, 1 to 4 wavefronts per SIMD: one wavefront per SIMD is
// k_layer's situation at 100k particles (a colour step lasts as long as one wavefront's instruction stream), four per SIMD is
// the throughput limit (time / 4 = the issue time of the stream).
//
// build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I pies_amd/csrc -I include tools/svd_bench.hip -o /tmp/svd_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "pbd_project.h"

using namespace pies;

template <int VARIANT, bool FLAT> __global__ void k_chain(float4* out, int steps) {
  const int t = threadIdx.x + blockIdx.x * blockDim.x;
  const float e = 0.001f * static_cast<float>(t % 97);
  const float y0 = FLAT ? 0.475f : 0.02f;
  float4 x1 = make_float4(0.01f + e, y0, -0.01f, 1.f), x2 = make_float4(1.1f + e, FLAT ? y0 : 0.05f, 0.02f, 1.f);
  float4 x3 = make_float4(-0.03f, FLAT ? y0 : 0.93f + e, 0.04f, 1.f), x4 = make_float4(0.02f, FLAT ? y0 : -0.04f, 1.07f - e, 1.f);
  const float4 a0 = make_float4(1.f, 0.f, 0.f, 0.f), a1 = make_float4(1.f, 0.f, 0.f, 0.f), a2 = make_float4(1.f, 0.8f, 1.0f, 0.05f);
  for (int s = 0; s < steps; ++s) {
    tet_core<VARIANT>(x1, x2, x3, x4, a0, a1, a2);
    // keep the element from collapsing towards the origin (quirk Q2) so that every step does a full decomposition
    x2.x += 1.0f; x4.z += 1.0f;
    x4.x -= 0.27f; x2.z -= 0.19f;
    if (FLAT) { x1.y = y0; x2.y = y0; x3.y = y0; x4.y = y0; x3.x += 0.4f; x3.z += 0.6f; }  // (the floor clamp)
    else { x3.y += 1.0f; x2.y += 0.31f; x3.z += 0.23f; }
  }
  out[t] = make_float4(x1.x + x2.x, x1.y + x3.y, x1.z + x4.z, x2.y + x3.z + x4.x);
}

template <int VARIANT, bool FLAT> static void run(const char* name, float4* d) {
  const int cus = 256, steps = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::printf("%s\n", name);
  for (int waves = 1; waves <= 4; ++waves) {
    const int threads = 256 * waves;
    hipLaunchKernelGGL((k_chain<VARIANT, FLAT>), dim3(cus), dim3(threads), 0, 0, d, steps);  // warm
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL((k_chain<VARIANT, FLAT>), dim3(cus), dim3(threads), 0, 0, d, steps);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    std::vector<float4> h(4);
    hipMemcpy(h.data(), d, sizeof(float4) * 4, hipMemcpyDeviceToHost);
    std::printf("  %d wavefront(s) per SIMD: %.3f us per projection step  (checksum %.6f)\n", waves, 1e3 * best / steps,
                h[0].x + h[1].y + h[2].z + h[3].w);
  }
}

int main() {
  float4* d;
  hipMalloc(&d, sizeof(float4) * 256 * 1024);
  run<2, false>("healthy element, plain Jacobi iteration from V = I (rounds 1-5)", d);
  run<0, false>("healthy element, closed-form start (round 6)", d);
  run<2, true>("flattened element (y = const), plain Jacobi iteration", d);
  run<0, true>("flattened element (y = const), closed-form start's direct path", d);
  return 0;
}
