// Device-side 3x3 arithmetic for the constraint projections (gfx950).
// Everything lives in registers: fixed-size arrays are only ever indexed by compile-time constants
// (after unrolling), so nothing is demoted to scratch.  All code is compiled with -ffp-contract=off:
// the arithmetic is the plain IEEE single-precision sequence written here, which is what makes the
// device results reproducible bit for bit on a host.
#pragma once
#include <hip/hip_runtime.h>

namespace pies {

#define PIES_DEV __device__ __forceinline__

// Device-side bounds checks of a diagnostic build (python -m pies_amd.build --bounds: -DPIES_BOUNDS; libpies_hip_bounds.so, never
// loaded by the product): PIES_IN_BOUNDS(cond, site) is `cond`, and a false one is RECORDED (first failing site id + a count, read by
// pies_exp_bounds_report) and the guarded access skipped - no trap: a faulting kernel can take the whole node down.  In the product
// build the macro is the constant true and the guards compile away.  Sites: 1x k_layer, 2x PD local tiles, 3x the windowed CG
// rows, 4x the node-node pair lists, 5x the triangle contact lists.
#ifdef PIES_BOUNDS
static __device__ unsigned int g_pies_bounds[2];  // (one per translation unit: the library is not built as relocatable device code)
PIES_DEV bool pies_bounds_note(bool ok, unsigned int site) {
  if (!ok) {
    atomicCAS(&g_pies_bounds[0], 0u, site);
    atomicAdd(&g_pies_bounds[1], 1u);
  }
  return ok;
}
#define PIES_IN_BOUNDS(cond, site) (::pies::pies_bounds_note((cond), (site)))
#define PIES_CLAMP_INDEX(i, n) ((i) < (n) ? (i) : 0u)
// in every kernel file: the host function that reads (and clears) the file's record: out[0] first failing site, out[1] count
#define PIES_BOUNDS_REPORT(name)                                                                                     \
  extern "C" int pies_exp_bounds_##name(unsigned int* out) {                                                         \
    unsigned int zero[2] = {0u, 0u};                                                                                 \
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(::pies::g_pies_bounds), sizeof(zero)) != hipSuccess) return 1;           \
    return hipMemcpyToSymbol(HIP_SYMBOL(::pies::g_pies_bounds), zero, sizeof(zero)) == hipSuccess ? 0 : 1;           \
  }
#else
#define PIES_IN_BOUNDS(cond, site) (true)
#define PIES_CLAMP_INDEX(i, n) (i)
#define PIES_BOUNDS_REPORT(name)
#endif

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

// a pair (alpha, beta, gamma) = (|b_p|^2, |b_q|^2, b_p.b_q) is out of tolerance.  + kSvdTiny2: a column whose squared
// norm has fallen to ~1e-36 is numerically zero (a collapsed element); since gamma^2 <= alpha*beta the pair is then skipped
// instead of sending rsqrt_nr out of its range.  For every other pair the fused sum rounds to kSvdTol2 * (alpha * beta).
PIES_DEV bool pair_needs(float alpha, float beta, float gamma) { return gamma * gamma > fmaf(kSvdTol2, alpha * beta, kSvdTiny2); }

// the rotation that makes columns P and Q of B orthogonal, applied to B and V
template <int P, int Q> PIES_DEV void jacobi_rotate(Svd3& d, const float alpha, const float beta, const float gamma) {
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
}
// Test of the pair (P, Q) and, where needed, its rotation.  The rotation stands behind a WAVE-UNIFORM branch: left to itself
// the compiler turns "if (needed) rotate" into straight-line code with selects (profiles/r06_layer_isa.txt: 68 instructions per
// pair whether anything rotates or not), and most tests of a decomposition - the certifying sweep, the pairs of a flattened
// element - rotate nothing in any lane.  With the branch a test costs its 13 instructions.
template <int P, int Q> PIES_DEV bool jacobi_pair(Svd3& d) {
  const float alpha = dot3f(d.b[P], d.b[P]);
  const float beta = dot3f(d.b[Q], d.b[Q]);
  const float gamma = dot3f(d.b[P], d.b[Q]);
  const bool need = pair_needs(alpha, beta, gamma);
  if (__builtin_amdgcn_ballot_w64(need) == 0ull) return false;  // (uniform: no lane of the wavefront rotates)
  if (need) jacobi_rotate<P, Q>(d, alpha, beta, gamma);
  return need;
}
PIES_DEV void svd3_finish(Svd3& d) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float n2 = dot3f(d.b[i], d.b[i]);
    const bool ok = n2 > kSvdTiny2;
    const float r = rsqrt_nr(ok ? n2 : 1.0f);  // (straight-line: the three Newton chains interleave instead of standing behind three branches)
    d.rs[i] = ok ? r : 0.0f;                   // a collapsed direction: s = 0, handled by svd3_recompose
    d.s[i] = n2 * d.rs[i];
  }
}

// The plain iteration from V = I (rounds 1-5): a[r][c] row-major.  A*V = B with orthogonal columns; s_i = |b_i|.
PIES_DEV void svd3_jacobi(const float a[3][3], Svd3& d) {
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
  svd3_finish(d);
}

// ---- round 6: the iteration is STARTED where it would end -----------------------------------------------------------------
// From V = I a generic deformation gradient takes 3-4 rotating sweeps (9-12 rotations of ~80 instructions); what a sweep
// converges to is the eigenvector frame of S = A^T A, and for a symmetric 3x3 that frame has a closed form:
//   * S's most isolated eigenvalue lambda = q +- 2 p cos(acos(|r|) / 3)  (q = tr S / 3, 6 p^2 = |S - qI|_F^2,
//     r = det(S - qI) / 2 p^3; the largest eigenvalue when r >= 0, else the smallest - the one whose cosine is insensitive to
//     r, so near-double eigenvalues cost no accuracy).  cos(acos(x)/3) on [0, 1] is analytic (the root of 4c^3 - 3c = x above
//     sqrt(3)/2): a degree-7 polynomial, |error| < 8e-8; no inverse trigonometric function is evaluated.
//   * its eigenvector n = the largest column of adj(S - lambda I) (the cross product of two rows), normalised;
//   * V0 = [t1, t2, n] with (t1, t2) the branch-free orthonormal completion of Duff et al. (JCGT 2017), det V0 = +1;
//   * B = A V0, and ONE rotation of the pair (0, 1) diagonalises what is left (a 2x2 problem: exact).
// The same sweeps as before then run until one passes without a rotation - they CERTIFY the result to the same tolerance
// as in rounds 1-5 and repair whatever the closed form left (ill-conditioned A: S squares the condition number), so accuracy
// is that of the one-sided iteration, not of the closed form.  On a perturbed rest state: 1 rotation + 3 tests instead of
// 10-12 rotations + 12 tests.  An element with at most one pair out of tolerance (rest state; a flattened element
// whose zero column is a coordinate axis - BASELINE config 2 after its first tick) skips the closed form: that pair is
// rotated from the entries of S.  Every choice is made per element from its own data: bit-reproducible on a host.
constexpr float kCos3[8] = {8.660253882e-01f, 1.666651964e-01f,  -4.807964712e-02f, 2.440584078e-02f,
                            -1.432729699e-02f, 7.718813606e-03f, -2.961986931e-03f, 5.536798271e-04f};
// 1 / t for t in [1, 2]: a linear seed (error < 1/17) and three Newton steps
PIES_DEV float recip12(float t) {
  float y = fmaf(-0.47058824f, t, 1.4117647f);
  y = y * fmaf(-t, y, 2.0f);
  y = y * fmaf(-t, y, 2.0f);
  y = y * fmaf(-t, y, 2.0f);
  return y;
}
// 1 / x to ~2e-4 (integer seed, two Newton steps), any sign: enough for a correction that is itself of the order of 1e-6
PIES_DEV float recip_rough(float x) {
  float y = __int_as_float(0x7EF311C7 - (__float_as_int(x) & 0x7fffffff));
  y = y * fmaf(-fabsf(x), y, 2.0f);
  y = y * fmaf(-fabsf(x), y, 2.0f);
  return __builtin_copysignf(y, x);
}
// The small-angle form of the three rotations, applied TOGETHER and unnormalised: B <- B (I + T), V <- V (I + T) with T
// antisymmetric, t_pq = g_pq / (|b_q|^2 - |b_p|^2) from one snapshot of the six inner products (|t| < 2.5e-4: the columns grow by
// t^2 / 2 < 3e-8, the cross terms are of second order).  What the closed-form frame leaves between its isolated direction and the
// other two - and what the full rotation of the pair (0, 1) leaves between a long and a short column - is a rounding-level
// angle (~1e-6), but one that an ill-conditioned element (a nearly flat one: its smallest column 20x shorter than the others -
// every element of BASELINE config 2 between the floor clamps) fails the relative test on.  This takes it out for ~85
// instructions in every lane; left to the certifying sweeps it costs a rotating sweep (3 x 70) and another clean one in every
// WAVEFRONT that holds such an element (measured on config 2: 18 % of the elements, i.e. every wavefront).
PIES_DEV void jacobi_polish(Svd3& d) {
  const float n0 = dot3f(d.b[0], d.b[0]), n1 = dot3f(d.b[1], d.b[1]), n2 = dot3f(d.b[2], d.b[2]);
  const float g01 = dot3f(d.b[0], d.b[1]), g02 = dot3f(d.b[0], d.b[2]), g12 = dot3f(d.b[1], d.b[2]);
  float t01 = g01 * recip_rough(n1 - n0), t02 = g02 * recip_rough(n2 - n0), t12 = g12 * recip_rough(n2 - n1);
  if (!(fabsf(t01) < 2.5e-4f)) t01 = 0.0f;  // (also NaN: equal norms)
  if (!(fabsf(t02) < 2.5e-4f)) t02 = 0.0f;
  if (!(fabsf(t12) < 2.5e-4f)) t12 = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float x = d.b[0][k], y = d.b[1][k], z = d.b[2][k];
    d.b[0][k] = fmaf(-t02, z, fmaf(-t01, y, x));
    d.b[1][k] = fmaf(-t12, z, fmaf(t01, x, y));
    d.b[2][k] = fmaf(t12, y, fmaf(t02, x, z));
    const float vx = d.v[0][k], vy = d.v[1][k], vz = d.v[2][k];
    d.v[0][k] = fmaf(-t02, vz, fmaf(-t01, vy, vx));
    d.v[1][k] = fmaf(-t12, vz, fmaf(t01, vx, vy));
    d.v[2][k] = fmaf(t12, vy, fmaf(t02, vx, vz));
  }
}
// a[r][c]: row-major input.  A*V = B with orthogonal columns; s_i = |b_i|.
PIES_DEV void svd3(const float a[3][3], Svd3& d) {
  float A[3][3];  // A[i] = column i
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) A[i][k] = a[k][i];
  const float s00 = dot3f(A[0], A[0]), s11 = dot3f(A[1], A[1]), s22 = dot3f(A[2], A[2]);
  const float s01 = dot3f(A[0], A[1]), s02 = dot3f(A[0], A[2]), s12 = dot3f(A[1], A[2]);
  const bool n01 = pair_needs(s00, s11, s01), n02 = pair_needs(s00, s22, s02), n12 = pair_needs(s11, s22, s12);
  const int cnt = (n01 ? 1 : 0) + (n02 ? 1 : 0) + (n12 ? 1 : 0);
  float q = 0.0f, d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, p2 = 0.0f;
  if (cnt >= 2) {
    q = ((s00 + s11) + s22) * 0.333333343f;
    d0 = s00 - q; d1 = s11 - q; d2 = s22 - q;
    const float p1 = fmaf(s12, s12, fmaf(s02, s02, s01 * s01));
    p2 = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, p1 + p1)));  // 6 p^2
  }
  if (cnt >= 2 && p2 > 1.0e-30f && p2 < 1.0e16f) {  // (the range in which nothing below leaves single precision)
    const float w = p2 * 0.166666672f;
    const float ip = rsqrt_nr(w);
    const float p = w * ip;
    const float det = fmaf(d0, fmaf(d1, d2, -(s12 * s12)), fmaf(s02, fmaf(s01, s12, -(d1 * s02)), -(s01 * fmaf(s01, d2, -(s12 * s02)))));
    const float r = ((0.5f * det) * ip) * (ip * ip);
    const float x = fminf(fabsf(r), 1.0f);
    float c = kCos3[7];
#pragma unroll
    for (int k = 6; k >= 0; --k) c = fmaf(c, x, kCos3[k]);
    const float lam = q + __builtin_copysignf((p + p) * c, r);
    const float m00 = s00 - lam, m11 = s11 - lam, m22 = s22 - lam;
    const float c00 = fmaf(m11, m22, -(s12 * s12)), c11 = fmaf(m00, m22, -(s02 * s02)), c22 = fmaf(m00, m11, -(s01 * s01));
    const float c01 = fmaf(s02, s12, -(s01 * m22)), c02 = fmaf(s01, s12, -(s02 * m11)), c12 = fmaf(s01, s02, -(s12 * m00));
    const float a0 = fabsf(c00), a1 = fabsf(c11), a2 = fabsf(c22);
    const bool k0 = a0 >= a1 && a0 >= a2, k1 = !k0 && a1 >= a2;
    const float v0 = k0 ? c00 : (k1 ? c01 : c02), v1 = k0 ? c01 : (k1 ? c11 : c12), v2 = k0 ? c02 : (k1 ? c12 : c22);
    const float n2 = fmaf(v2, v2, fmaf(v1, v1, v0 * v0));
    const bool okn = n2 > kSvdTiny2;
    const float in = rsqrt_nr(okn ? n2 : 1.0f);
    const float nx = okn ? v0 * in : 0.0f, ny = okn ? v1 * in : 0.0f, nz = okn ? v2 * in : 1.0f;
    const float sg = __builtin_copysignf(1.0f, nz);
    const float aa = -recip12(fabsf(nz) + 1.0f) * sg;  // -1 / (sg + nz)
    const float bb = (nx * ny) * aa;
    const float V[3][3] = {{fmaf(sg * nx, nx * aa, 1.0f), sg * bb, -(sg * nx)}, {bb, fmaf(ny, ny * aa, sg), -ny}, {nx, ny, nz}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        d.v[i][k] = V[i][k];
        d.b[i][k] = fmaf(A[2][k], V[i][2], fmaf(A[1][k], V[i][1], A[0][k] * V[i][0]));
      }
    (void)jacobi_pair<0, 1>(d);
    jacobi_polish(d);
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        d.b[i][k] = A[i][k];
        d.v[i][k] = (i == k) ? 1.0f : 0.0f;
      }
    if (n01) jacobi_rotate<0, 1>(d, s00, s11, s01);
    else if (n02) jacobi_rotate<0, 2>(d, s00, s22, s02);
    else if (n12) jacobi_rotate<1, 2>(d, s11, s22, s12);
  }
  // The certifying sweeps: all three pairs are tested on one snapshot of the six inner products; a clean snapshot ends the
  // decomposition and its squared norms are the singular values' (one branch and 30 instructions where three separate tests
  // cost three branches and 41, and svd3_finish's norms another 9); otherwise a sweep of the plain iteration runs.
  float n0 = s00, n1 = s11, n2 = s22;  // (cnt == 0: B = A, nothing was rotated)
  if (cnt != 0) {
    bool clean = false;
    for (int sweep = 0; sweep < kSvdMaxSweeps; ++sweep) {
      n0 = dot3f(d.b[0], d.b[0]); n1 = dot3f(d.b[1], d.b[1]); n2 = dot3f(d.b[2], d.b[2]);
      const float g02 = dot3f(d.b[0], d.b[2]), g12 = dot3f(d.b[1], d.b[2]), g01 = dot3f(d.b[0], d.b[1]);
      const bool t02 = pair_needs(n0, n2, g02), t12 = pair_needs(n1, n2, g12), t01 = pair_needs(n0, n1, g01);
      clean = !(int(t02) | int(t12) | int(t01));  // (no short circuit: one branch)
      if (clean) break;
      (void)jacobi_pair<0, 2>(d);
      (void)jacobi_pair<1, 2>(d);
      (void)jacobi_pair<0, 1>(d);
    }
    if (!clean) { n0 = dot3f(d.b[0], d.b[0]); n1 = dot3f(d.b[1], d.b[1]); n2 = dot3f(d.b[2], d.b[2]); }  // (sweeps exhausted)
  }
  const float nn[3] = {n0, n1, n2};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const bool ok = nn[i] > kSvdTiny2;
    const float r = rsqrt_nr(ok ? nn[i] : 1.0f);
    d.rs[i] = ok ? r : 0.0f;  // a collapsed direction: s = 0, handled by svd3_recompose
    d.s[i] = nn[i] * d.rs[i];
  }
}

// t[K] = sg * (u_I x u_J): the direction a collapsed column K leaves open, oriented so that det(U) det(V) = +1
// (V is a rotation: the identity or the completed frame of svd3, times Givens rotations)
template <int K, int I, int J> PIES_DEV void complete_t(const Svd3& d, float t[3][3], float sg) {
  float ui[3], uj[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    ui[c] = d.b[I][c] * d.rs[I];
    uj[c] = d.b[J][c] * d.rs[J];
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
  if (nbad == 1) {  // a flattened element
    if (!ok0) complete_t<0, 1, 2>(d, t, snew[0]);
    else if (!ok1) complete_t<1, 2, 0>(d, t, snew[1]);
    else complete_t<2, 0, 1>(d, t, snew[2]);
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
