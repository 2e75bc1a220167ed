// Device-side 3x3 arithmetic for the constraint projections (gfx950).
// Everything lives in registers: fixed-size arrays are only ever indexed by compile-time constants
// (after unrolling), so nothing is demoted to scratch.  All code is compiled with -ffp-contract=off:
// the arithmetic is the plain IEEE single-precision sequence written here, which is what makes the
// device results reproducible bit for bit on a host.
#pragma once
#include <hip/hip_runtime.h>

namespace pies {

#define PIES_DEV __device__ __forceinline__

// Workgroups are dealt round-robin over the 8 XCDs (observed, MI355X_MICROARCH.md "Workgroup dispatch"), each
// with a private 4 MiB L2.  Gather kernels whose work items are stored in mesh order use this bijective
// relabelling so that one XCD processes one contiguous eighth of the items and neighbouring items' node
// records meet in the same L2.  Speed only: any placement gives the same result.
PIES_DEV uint32_t xcd_block(uint32_t bid, uint32_t nwg) {
  const uint32_t q = nwg >> 3, r = nwg & 7u, x = bid & 7u;
  return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + (bid >> 3);
}

// One-sided (Hestenes) Jacobi SVD of a 3x3: column pairs of B = A*V are rotated until every pair is
// orthogonal to working precision, |b_p.b_q| <= kSvdTol |b_p||b_q|  (at most kSvdMaxSweeps sweeps; a
// typical deformation gradient needs 2-3 rotating sweeps plus the final check sweep).  Per rotation:
// two rsqrt_nr, everything else fused multiply-adds; no division or square-root instruction in the sweep.
#ifndef PIES_SVD_MAX_SWEEPS
#define PIES_SVD_MAX_SWEEPS 8
#endif
constexpr int kSvdMaxSweeps = PIES_SVD_MAX_SWEEPS;
constexpr float kSvdTol = 4.76837158203125e-07f;  // 4 * 2^-23
constexpr float kSvdTol2 = kSvdTol * kSvdTol;
constexpr float kSvdTiny = 1.0e-18f;
constexpr float kSvdTiny2 = 1.0e-36f;

struct Svd3 {
  float b[3][3];  // b[i] = column i of A*V (= s_i u_i)
  float v[3][3];  // v[i] = column i of V
  float s[3];   // |b_i|
  float rs[3];  // 1 / |b_i|
};

// 1/sqrt(x) for the SVD from an integer seed and three Newton steps (relative error ~1e-7): only *, fma and integer
// operations, so a host reproduces it bit for bit - and 14 instructions where a correctly rounded square root
// followed by a correctly rounded division takes 31 (measured: k_tet 5.98 -> 5.15 us per launch at 100k particles).
// Used where the SVD needs a normalisation, never in arithmetic restated from the reference.
PIES_DEV float rsqrt_nr(float x) {
  float y = __int_as_float(0x5f3759df - (__float_as_int(x) >> 1));
  const float hx = 0.5f * x;
  y = y * fmaf(-hx, y * y, 1.5f);
  y = y * fmaf(-hx, y * y, 1.5f);
  y = y * fmaf(-hx, y * y, 1.5f);
  return y;
}
PIES_DEV float dot3f(const float x[3], const float y[3]) { return fmaf(x[2], y[2], fmaf(x[1], y[1], x[0] * y[0])); }

template <int P, int Q> PIES_DEV bool jacobi_pair(Svd3& d) {
  const float alpha = dot3f(d.b[P], d.b[P]);
  const float beta = dot3f(d.b[Q], d.b[Q]);
  const float gamma = dot3f(d.b[P], d.b[Q]);
  // + kSvdTiny2: a column whose squared norm has fallen to ~1e-36 is numerically zero (a collapsed element); since
  // gamma^2 <= alpha*beta the pair is then skipped instead of sending rsqrt_nr out of its range.  For every other
  // pair the fused sum rounds to kSvdTol2 * (alpha * beta) itself.
  if (!(gamma * gamma > fmaf(kSvdTol2, alpha * beta, kSvdTiny2))) return false;
  const float delta = beta - alpha;
  const float g2 = gamma + gamma;
  const float hw = fmaf(delta, delta, g2 * g2);
  const float h = hw * rsqrt_nr(hw);
  const float c1 = h + fabsf(delta);        // ~ cos(theta)
  const float s1 = delta < 0.0f ? -g2 : g2; // ~ sin(theta):  tan = sign(delta)*2g / (|delta| + h)
  const float inv = rsqrt_nr(fmaf(c1, c1, s1 * s1));
  const float cs = c1 * inv, sn = s1 * inv;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float x = d.b[P][k], y = d.b[Q][k];
    d.b[P][k] = fmaf(cs, x, -(sn * y));
    d.b[Q][k] = fmaf(sn, x, cs * y);
    const float vx = d.v[P][k], vy = d.v[Q][k];
    d.v[P][k] = fmaf(cs, vx, -(sn * vy));
    d.v[Q][k] = fmaf(sn, vx, cs * vy);
  }
  return true;
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
  for (int sweep = 0; sweep < kSvdMaxSweeps; ++sweep) {
    const bool r01 = jacobi_pair<0, 1>(d);
    const bool r02 = jacobi_pair<0, 2>(d);
    const bool r12 = jacobi_pair<1, 2>(d);
    if (!(r01 || r02 || r12)) break;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float n2 = dot3f(d.b[i], d.b[i]);
    d.rs[i] = n2 > kSvdTiny2 ? rsqrt_nr(n2) : 0.0f;  // a collapsed direction: s = 0, handled by svd3_recompose
    d.s[i] = n2 * d.rs[i];
  }
}

// ---- the same decomposition with the rotations written on register PAIRS (round 4) ---------------------------------------
// The compiler packs a rotation's application into v_pk_mul_f32 / v_pk_fma_f32 already, but it assembles the operands of
// every packed instruction with v_mov_b32: 21 moves in the 80 VALU instructions of a rotation (profiles/r04_svd_isa_counts.txt)
// - and a colour step of k_layer lasts as long as its wavefronts' instruction streams.  A packed fp32 instruction takes each
// operand from EITHER half of a 64-bit register pair (op_sel), so no value ever has to be moved: the columns live in pairs,
//     sweep start:  A[k] = (col 0, -)   B[k] = (col 1, col 2)             k = component, likewise for V
//     (0,1): x = A.lo, y = B.lo -> N1 = (col 0', col 1');  (0,2): x = N1.lo, y = B.hi -> N2 = (col 0'', col 2');
//     (1,2): x = N1.hi, y = N2.hi -> N3 = (col 1'', col 2'');  next sweep: A = N2, B = N3
// and one rotation of (x, y) is   t = (-sn * y, cs * y);  (x', y') = (cs * x + t.lo, sn * x + t.hi)   - the very IEEE
// operations of jacobi_pair (a product, a negation, a fused multiply-add), so the result is bit for bit that of svd3.
// A lane whose pair needs no rotation while another lane's does goes through the same instructions with (cs, sn) = (1, 0):
// x * 1 - 0 and 0 * x + y return x and y; when no lane needs it the pair is only re-packed (v_pk_mov_b32).
typedef float pk2 __attribute__((ext_vector_type(2)));
template <int HX, int HY> PIES_DEV pk2 pk_rotate(const pk2 sc, const pk2 xp, const pk2 yp) {  // sc = (cs, sn); x = xp[HX], y = yp[HY]
  pk2 t, r;
  if (HY == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(t) : "v"(sc), "v"(yp));
  else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(t) : "v"(sc), "v"(yp));
  if (HX == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(sc), "v"(xp), "v"(t));
  else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(sc), "v"(xp), "v"(t));
  return r;
}
template <int HX, int HY> PIES_DEV pk2 pk_pair(const pk2 xp, const pk2 yp) {  // (xp[HX], yp[HY])
  pk2 r;
  if (HX == 0 && HY == 0) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(r) : "v"(xp), "v"(yp));
  else if (HX == 0 && HY == 1) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(xp), "v"(yp));
  else if (HX == 1 && HY == 0) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(xp), "v"(yp));
  else asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(r) : "v"(xp), "v"(yp));
  return r;
}
// one rotation of the column pair (x, y) = (bx[k][HX], by[k][HY]) and of V's; out[k] = (x', y').  Returns jacobi_pair's flag.
template <int HX, int HY>
PIES_DEV bool jacobi_pair_pk(const pk2 bx[3], const pk2 by[3], const pk2 vx[3], const pk2 vy[3], pk2 bo[3], pk2 vo[3]) {
  const float x[3] = {bx[0][HX], bx[1][HX], bx[2][HX]}, y[3] = {by[0][HY], by[1][HY], by[2][HY]};
  const float alpha = dot3f(x, x);
  const float beta = dot3f(y, y);
  const float gamma = dot3f(x, y);
  const bool need = gamma * gamma > fmaf(kSvdTol2, alpha * beta, kSvdTiny2);
  if (__builtin_amdgcn_ballot_w64(need) != 0ull) {  // (uniform: some lane of the wavefront rotates)
    const float delta = beta - alpha;
    const float g2 = gamma + gamma;
    const float hw = fmaf(delta, delta, g2 * g2);
    const float h = hw * rsqrt_nr(hw);
    const float c1 = h + fabsf(delta);
    const float s1 = delta < 0.0f ? -g2 : g2;
    const float inv = rsqrt_nr(fmaf(c1, c1, s1 * s1));
    const pk2 sc = {need ? c1 * inv : 1.0f, need ? s1 * inv : 0.0f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      bo[k] = pk_rotate<HX, HY>(sc, bx[k], by[k]);
      vo[k] = pk_rotate<HX, HY>(sc, vx[k], vy[k]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      bo[k] = pk_pair<HX, HY>(bx[k], by[k]);
      vo[k] = pk_pair<HX, HY>(vx[k], vy[k]);
    }
  }
  return need;
}
PIES_DEV void svd3_pk(const float a[3][3], Svd3& d) {
  pk2 A[3], B[3], VA[3], VB[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    A[k] = pk2{a[k][0], 0.0f};
    B[k] = pk2{a[k][1], a[k][2]};
    VA[k] = pk2{k == 0 ? 1.0f : 0.0f, 0.0f};
    VB[k] = pk2{k == 1 ? 1.0f : 0.0f, k == 2 ? 1.0f : 0.0f};
  }
  for (int sweep = 0; sweep < kSvdMaxSweeps; ++sweep) {
    pk2 N1[3], V1[3], N2[3], V2[3], N3[3], V3[3];
    const bool r01 = jacobi_pair_pk<0, 0>(A, B, VA, VB, N1, V1);
    const bool r02 = jacobi_pair_pk<0, 1>(N1, B, V1, VB, N2, V2);
    const bool r12 = jacobi_pair_pk<1, 1>(N1, N2, V1, V2, N3, V3);
#pragma unroll
    for (int k = 0; k < 3; ++k) { A[k] = N2[k]; B[k] = N3[k]; VA[k] = V2[k]; VB[k] = V3[k]; }
    if (!(r01 || r02 || r12)) break;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    d.b[0][k] = A[k][0]; d.b[1][k] = B[k][0]; d.b[2][k] = B[k][1];
    d.v[0][k] = VA[k][0]; d.v[1][k] = VB[k][0]; d.v[2][k] = VB[k][1];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float n2 = dot3f(d.b[i], d.b[i]);
    d.rs[i] = n2 > kSvdTiny2 ? rsqrt_nr(n2) : 0.0f;
    d.s[i] = n2 * d.rs[i];
  }
}

template <int K, int I, int J> PIES_DEV void complete_t(const Svd3& d, float t[3][3], float sg) {
  float ui[3], uj[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    ui[c] = d.b[I][c] / d.s[I];
    uj[c] = d.b[J][c] / d.s[J];
  }
  t[K][0] = sg * (ui[1] * uj[2] - ui[2] * uj[1]);
  t[K][1] = sg * (ui[2] * uj[0] - ui[0] * uj[2]);
  t[K][2] = sg * (ui[0] * uj[1] - ui[1] * uj[0]);
}

// out[r][c] = sum_i (b_i[r] * snew[i]/s_i) * v_i[c]  ( = U diag(snew) V^T ).  A collapsed direction
// (s_i <= kSvdTiny) takes u_i from the oriented completion of the other two; two collapsed
// directions are dropped.
PIES_DEV void svd3_recompose(const Svd3& d, const float snew[3], float out[3][3]) {
  float t[3][3];
  const bool ok0 = d.s[0] > kSvdTiny, ok1 = d.s[1] > kSvdTiny, ok2 = d.s[2] > kSvdTiny;
  const float g0 = ok0 ? snew[0] * d.rs[0] : 0.0f;
  const float g1 = ok1 ? snew[1] * d.rs[1] : 0.0f;
  const float g2 = ok2 ? snew[2] * d.rs[2] : 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    t[0][k] = d.b[0][k] * g0;
    t[1][k] = d.b[1][k] * g1;
    t[2][k] = d.b[2][k] * g2;
  }
  const int nbad = (ok0 ? 0 : 1) + (ok1 ? 0 : 1) + (ok2 ? 0 : 1);
  if (nbad == 1) {  // rare: a flattened element
    const float detv = d.v[0][0] * (d.v[1][1] * d.v[2][2] - d.v[1][2] * d.v[2][1]) -
                       d.v[0][1] * (d.v[1][0] * d.v[2][2] - d.v[1][2] * d.v[2][0]) +
                       d.v[0][2] * (d.v[1][0] * d.v[2][1] - d.v[1][1] * d.v[2][0]);
    const float sgn = detv < 0.0f ? -1.0f : 1.0f;
    if (!ok0) complete_t<0, 1, 2>(d, t, sgn * snew[0]);
    else if (!ok1) complete_t<1, 2, 0>(d, t, sgn * snew[1]);
    else complete_t<2, 0, 1>(d, t, sgn * snew[2]);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) out[r][c] = fmaf(t[2][r], d.v[2][c], fmaf(t[1][r], d.v[1][c], t[0][r] * d.v[0][c]));
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
