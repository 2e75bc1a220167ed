// ISA-level instruction counts of 3x3 SVD variants for k_layer's tetrahedral projection.
//   A  one-sided (Hestenes) Jacobi on the columns of F, sweeps until converged        = the product's svd3 (dev_math.h)
//   B  two-sided Jacobi on S = F^T F with the exact rotation (two rsqrt_nr), V accumulated as a matrix, fixed sweeps
//   C  two-sided Jacobi on S with McAdams' approximate Givens rotation (one rsqrt_nr, half-angle clamp at pi/8), V accumulated
//      as a quaternion, fixed sweeps (Computing the SVD of 3x3 matrices with minimal branching ..., 2011)
// Each kernel runs ONE sweep (three rotations) on registers; `python tools/svd_isa_counts.py` prints the VALU instructions of one
// sweep.  B and C also need S = F^T F up front (18 instructions) and B = F V afterwards (15): charged in the table of DESIGN.md.
#include <hip/hip_runtime.h>
#include "../pies_amd/csrc/dev_math.h"
using namespace pies;

__global__ void sweep_A(float* io) {
  Svd3 d;
  for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) { d.b[i][k] = io[threadIdx.x * 18 + 3 * i + k]; d.v[i][k] = io[threadIdx.x * 18 + 9 + 3 * i + k]; }
  jacobi_pair<0, 1>(d); jacobi_pair<0, 2>(d); jacobi_pair<1, 2>(d);
  for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) { io[threadIdx.x * 18 + 3 * i + k] = d.b[i][k]; io[threadIdx.x * 18 + 9 + 3 * i + k] = d.v[i][k]; }
}

struct Sym { float s00, s11, s22, s01, s02, s12; float v[3][3]; };
template <int P, int Q> __device__ __forceinline__ void rot_B(Sym& m) {
  float& spp = P == 0 ? m.s00 : m.s11; float& sqq = Q == 1 ? m.s11 : m.s22;
  float& spq = (P == 0 && Q == 1) ? m.s01 : (P == 0 ? m.s02 : m.s12);
  float& spr = (P == 0 && Q == 1) ? m.s02 : (P == 0 ? m.s01 : m.s01);
  float& sqr = (P == 0 && Q == 1) ? m.s12 : (P == 0 ? m.s12 : m.s02);
  if (!(spq * spq > fmaf(kSvdTol2, spp * sqq, kSvdTiny2))) return;
  const float delta = sqq - spp, g2 = spq + spq;
  const float hw = fmaf(delta, delta, g2 * g2);
  const float h = hw * rsqrt_nr(hw);
  const float c1 = h + fabsf(delta), s1 = delta < 0.0f ? -g2 : g2;
  const float inv = rsqrt_nr(fmaf(c1, c1, s1 * s1));
  const float cs = c1 * inv, sn = s1 * inv;
  const float cc = cs * cs, ss = sn * sn, cs2 = 2.0f * cs * sn;
  const float npp = fmaf(cc, spp, fmaf(-cs2, spq, ss * sqq)), nqq = fmaf(ss, spp, fmaf(cs2, spq, cc * sqq));
  const float npr = fmaf(cs, spr, -(sn * sqr)), nqr = fmaf(sn, spr, cs * sqr);
  spp = npp; sqq = nqq; spq = 0.0f; spr = npr; sqr = nqr;
#pragma unroll
  for (int k = 0; k < 3; ++k) { const float x = m.v[P][k], y = m.v[Q][k]; m.v[P][k] = fmaf(cs, x, -(sn * y)); m.v[Q][k] = fmaf(sn, x, cs * y); }
}
__global__ void sweep_B(float* io) {
  Sym m; const float* p = io + threadIdx.x * 18;
  m.s00 = p[0]; m.s11 = p[1]; m.s22 = p[2]; m.s01 = p[3]; m.s02 = p[4]; m.s12 = p[5];
  for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) m.v[i][k] = p[9 + 3 * i + k];
  rot_B<0, 1>(m); rot_B<0, 2>(m); rot_B<1, 2>(m);
  float* q = io + threadIdx.x * 18;
  q[0] = m.s00; q[1] = m.s11; q[2] = m.s22; q[3] = m.s01; q[4] = m.s02; q[5] = m.s12;
  for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) q[9 + 3 * i + k] = m.v[i][k];
}

struct SymQ { float s00, s11, s22, s01, s02, s12; float qx, qy, qz, qw; };
template <int P, int Q> __device__ __forceinline__ void rot_C(SymQ& m) {
  float& spp = P == 0 ? m.s00 : m.s11; float& sqq = Q == 1 ? m.s11 : m.s22;
  float& spq = (P == 0 && Q == 1) ? m.s01 : (P == 0 ? m.s02 : m.s12);
  float& spr = (P == 0 && Q == 1) ? m.s02 : (P == 0 ? m.s01 : m.s01);
  float& sqr = (P == 0 && Q == 1) ? m.s12 : (P == 0 ? m.s12 : m.s02);
  float ch = 2.0f * (spp - sqq), sh = spq;
  const bool big = 5.82842712474619f * sh * sh < ch * ch;
  const float w = rsqrt_nr(fmaf(ch, ch, sh * sh + kSvdTiny2));
  ch = big ? w * ch : 0.9238795325112867f; sh = big ? w * sh : 0.3826834323650898f;
  const float c = fmaf(ch, ch, -(sh * sh)), s = 2.0f * ch * sh;
  const float cc = c * c, ss = s * s, cs2 = 2.0f * c * s;
  const float npp = fmaf(cc, spp, fmaf(cs2, spq, ss * sqq)), nqq = fmaf(ss, spp, fmaf(-cs2, spq, cc * sqq));
  const float npq = fmaf(c * s, sqq - spp, (cc - ss) * spq);
  const float npr = fmaf(c, spr, s * sqr), nqr = fmaf(-s, spr, c * sqr);
  spp = npp; sqq = nqq; spq = npq; spr = npr; sqr = nqr;
  // quaternion of the rotation about the axis perpendicular to (P, Q): (sh on that axis, ch)
  float ax = 0.f, ay = 0.f, az = 0.f;
  if (P == 0 && Q == 1) az = sh; else if (P == 0) ay = -sh; else ax = sh;
  const float nx = fmaf(m.qw, ax, fmaf(m.qx, ch, fmaf(m.qy, az, -(m.qz * ay))));
  const float ny = fmaf(m.qw, ay, fmaf(m.qy, ch, fmaf(m.qz, ax, -(m.qx * az))));
  const float nz = fmaf(m.qw, az, fmaf(m.qz, ch, fmaf(m.qx, ay, -(m.qy * ax))));
  const float nw = fmaf(m.qw, ch, -fmaf(m.qx, ax, fmaf(m.qy, ay, m.qz * az)));
  m.qx = nx; m.qy = ny; m.qz = nz; m.qw = nw;
}
__global__ void sweep_C(float* io) {
  SymQ m; const float* p = io + threadIdx.x * 18;
  m.s00 = p[0]; m.s11 = p[1]; m.s22 = p[2]; m.s01 = p[3]; m.s02 = p[4]; m.s12 = p[5]; m.qx = p[6]; m.qy = p[7]; m.qz = p[8]; m.qw = p[9];
  rot_C<0, 1>(m); rot_C<0, 2>(m); rot_C<1, 2>(m);
  float* q = io + threadIdx.x * 18;
  q[0] = m.s00; q[1] = m.s11; q[2] = m.s22; q[3] = m.s01; q[4] = m.s02; q[5] = m.s12; q[6] = m.qx; q[7] = m.qy; q[8] = m.qz; q[9] = m.qw;
}
