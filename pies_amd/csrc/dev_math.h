// Device-side 3x3 arithmetic for the constraint projections (gfx950).
// Everything lives in registers: fixed-size arrays are only ever indexed by compile-time constants
// (after unrolling), so nothing is demoted to scratch.  All code is compiled with -ffp-contract=off:
// the arithmetic is the plain IEEE single-precision sequence written here, which is what makes the
// device results reproducible bit for bit on a host.
#pragma once
#include <hip/hip_runtime.h>

namespace pies {

#define PIES_DEV __device__ __forceinline__

constexpr int kSvdSweeps = 5;       // one-sided Jacobi sweeps; 4 already reach fp32 round-off on 3x3
constexpr float kSvdTiny = 1.0e-18f;

struct Svd3 {
  float b[3][3];  // b[i] = column i of A*V (= s_i u_i)
  float v[3][3];  // v[i] = column i of V
  float s[3];
};

template <int P, int Q> PIES_DEV void jacobi_pair(Svd3& d) {
  const float alpha = d.b[P][0] * d.b[P][0] + d.b[P][1] * d.b[P][1] + d.b[P][2] * d.b[P][2];
  const float beta = d.b[Q][0] * d.b[Q][0] + d.b[Q][1] * d.b[Q][1] + d.b[Q][2] * d.b[Q][2];
  const float gamma = d.b[P][0] * d.b[Q][0] + d.b[P][1] * d.b[Q][1] + d.b[P][2] * d.b[Q][2];
  float cs = 1.0f, sn = 0.0f;
  if (gamma != 0.0f) {
    const float zeta = (beta - alpha) / (2.0f * gamma);
    float t = 1.0f / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
    if (zeta < 0.0f) t = -t;
    cs = 1.0f / sqrtf(1.0f + t * t);
    sn = cs * t;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float x = d.b[P][k], y = d.b[Q][k];
    d.b[P][k] = cs * x - sn * y;
    d.b[Q][k] = sn * x + cs * y;
    const float vx = d.v[P][k], vy = d.v[Q][k];
    d.v[P][k] = cs * vx - sn * vy;
    d.v[Q][k] = sn * vx + cs * vy;
  }
}

// a[r][c]: row-major input.  A*V = B with orthogonal columns; s_i = |b_i|.
PIES_DEV void svd3(const float a[3][3], Svd3& d) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      d.b[i][k] = a[k][i];
      d.v[i][k] = (i == k) ? 1.0f : 0.0f;
    }
#pragma unroll
  for (int sweep = 0; sweep < kSvdSweeps; ++sweep) {
    jacobi_pair<0, 1>(d);
    jacobi_pair<0, 2>(d);
    jacobi_pair<1, 2>(d);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
    d.s[i] = sqrtf(d.b[i][0] * d.b[i][0] + d.b[i][1] * d.b[i][1] + d.b[i][2] * d.b[i][2]);
}

template <int K, int I, int J> PIES_DEV void complete_u(float u[3][3], float sg) {
  u[K][0] = sg * (u[I][1] * u[J][2] - u[I][2] * u[J][1]);
  u[K][1] = sg * (u[I][2] * u[J][0] - u[I][0] * u[J][2]);
  u[K][2] = sg * (u[I][0] * u[J][1] - u[I][1] * u[J][0]);
}

// out[r][c] = sum_i (u_i[r] * snew[i]) * v_i[c],  u_i = b_i / s_i.  A collapsed direction
// (s_i <= kSvdTiny) gets u_i from the oriented completion of the other two; two collapsed
// directions are dropped.
PIES_DEV void svd3_recompose(const Svd3& d, const float snew[3], float out[3][3]) {
  float u[3][3];
  const bool ok0 = d.s[0] > kSvdTiny, ok1 = d.s[1] > kSvdTiny, ok2 = d.s[2] > kSvdTiny;
  const float i0 = ok0 ? 1.0f / d.s[0] : 0.0f;
  const float i1 = ok1 ? 1.0f / d.s[1] : 0.0f;
  const float i2 = ok2 ? 1.0f / d.s[2] : 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    u[0][k] = d.b[0][k] * i0;
    u[1][k] = d.b[1][k] * i1;
    u[2][k] = d.b[2][k] * i2;
  }
  const int nbad = (ok0 ? 0 : 1) + (ok1 ? 0 : 1) + (ok2 ? 0 : 1);
  if (nbad == 1) {  // rare: a flattened element
    const float detv = d.v[0][0] * (d.v[1][1] * d.v[2][2] - d.v[1][2] * d.v[2][1]) -
                       d.v[0][1] * (d.v[1][0] * d.v[2][2] - d.v[1][2] * d.v[2][0]) +
                       d.v[0][2] * (d.v[1][0] * d.v[2][1] - d.v[1][1] * d.v[2][0]);
    const float sg = detv < 0.0f ? -1.0f : 1.0f;
    if (!ok0) complete_u<0, 1, 2>(u, sg);
    else if (!ok1) complete_u<1, 2, 0>(u, sg);
    else complete_u<2, 0, 1>(u, sg);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      out[r][c] = (u[0][r] * snew[0]) * d.v[0][c] + (u[1][r] * snew[1]) * d.v[1][c] + (u[2][r] * snew[2]) * d.v[2][c];
}

PIES_DEV float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// F = P * Qinv in the column-major convention of the reference's glm types:
// f[c][r] = p[0][r]*q[c][0] + p[1][r]*q[c][1] + p[2][r]*q[c][2].
PIES_DEV void mat3_mul_cm(const float p[3][3], const float q[3][3], float f[3][3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) f[c][r] = p[0][r] * q[c][0] + p[1][r] * q[c][1] + p[2][r] * q[c][2];
}

PIES_DEV float det3_cm(const float m[3][3]) {
  return +m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) +
         m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
}

}  // namespace pies
