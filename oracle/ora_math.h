// TEST INFRASTRUCTURE ONLY -- CPU oracle for the Pies hot path. PARITY UNPINNED (see README.md).
//
// Minimal vector/matrix arithmetic restating the *published* glm semantics the reference relies on
// (g-truc/glm, un-vendored submodule, commit unpinned: /root/reference/.gitmodules:1-3).
// Only the operations the hot path uses are restated, with glm's documented evaluation order:
//   dot(a,b)      = a.x*b.x + a.y*b.y + a.z*b.z           (left-associated)
//   length(v)     = sqrt(dot(v,v))
//   normalize(v)  = v * (1/sqrt(dot(v,v)))                 (glm::inversesqrt = 1/sqrt)
//   mat3          = 3 columns, m[c][r]                     (column-major)
//   inverse(mat3) = cofactors * (1/det)
// Nothing under oracle/ may be linked, imported or executed by the product (pies_amd/, include/).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace ora {

struct vec3 {
  float x = 0.f, y = 0.f, z = 0.f;
  vec3() = default;
  vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
  explicit vec3(float s) : x(s), y(s), z(s) {}
  float& operator[](int i) { return (&x)[i]; }
  const float& operator[](int i) const { return (&x)[i]; }
};

inline vec3 operator+(const vec3& a, const vec3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(const vec3& a, const vec3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(const vec3& a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(const vec3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, const vec3& a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator*(const vec3& a, const vec3& b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator/(const vec3& a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3& operator+=(vec3& a, const vec3& b) { a = a + b; return a; }
inline vec3& operator-=(vec3& a, const vec3& b) { a = a - b; return a; }

inline float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(const vec3& a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(const vec3& a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline vec3 cross(const vec3& a, const vec3& b) {
  return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
// Order of the node pairs in collision rule 2 (the device's pair order): pairs are classed by the direction from the lower to
// the higher node when the grid was built - 13 classes of 45-degree sectors around the 26 lattice directions, orientation
// folded - and by the parity of the pair's place along that direction (in units of the pair's own extent); inside a class
// murmur3's 64-bit finaliser over (i << 32 | j) decides.  Pairs of one class seldom share a node (on a lattice never), so
// the chains of pairs that share nodes stay about as short as the number of classes.  The key is a total order: the hash
// part alone is a bijection of the pair.  Plain IEEE single precision, no contraction: the device computes the same bits.
inline uint64_t pair_key(uint32_t i, uint32_t j, const vec3& pi, const vec3& pj) {  // i < j
  uint64_t k = (static_cast<uint64_t>(i) << 32) | j;
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  const float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
  const float ax = std::fabs(dx), ay = std::fabs(dy), az = std::fabs(dz);
  const float lim = 0.41421356f * std::fmax(ax, std::fmax(ay, az));
  int qx = ax > lim ? (dx < 0.0f ? -1 : 1) : 0, qy = ay > lim ? (dy < 0.0f ? -1 : 1) : 0, qz = az > lim ? (dz < 0.0f ? -1 : 1) : 0;
  const int lead = qx != 0 ? qx : (qy != 0 ? qy : qz);
  if (lead < 0) { qx = -qx; qy = -qy; qz = -qz; }
  const float fx = static_cast<float>(qx), fy = static_cast<float>(qy), fz = static_cast<float>(qz);
  const float qq = std::fmax(fx * fx + fy * fy + fz * fz, 1.0f);
  const float ui = (pi.x * fx + pi.y * fy + pi.z * fz) / qq, uj = (pj.x * fx + pj.y * fy + pj.z * fz) / qq;
  const float len = std::fmax(std::fabs(uj - ui), 0.001f);
  const float t = std::fmin(std::fmax(std::floor(std::fmin(ui, uj) / len), -1.0e9f), 1.0e9f);
  const uint32_t parity = static_cast<uint32_t>(static_cast<long long>(t)) & 1u;  // (NaN positions: the grid has failed already)
  const uint32_t cls = static_cast<uint32_t>((qx + 1) * 9 + (qy + 1) * 3 + (qz + 1)) * 2u + parity;  // < 54
  return (static_cast<uint64_t>(cls) << 58) | (k >> 6);
}

inline float clampf(float v, float lo, float hi) { return std::fmin(std::fmax(v, lo), hi); }
inline float fractf(float v) { return v - std::floor(v); }

// Column-major 3x3: c[col][row], like glm::mat3.
struct mat3 {
  vec3 c[3];
  mat3() = default;
  mat3(const vec3& c0, const vec3& c1, const vec3& c2) { c[0] = c0; c[1] = c1; c[2] = c2; }
  vec3& operator[](int i) { return c[i]; }
  const vec3& operator[](int i) const { return c[i]; }
};

// glm operator*(mat3, mat3): result[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2]
inline mat3 operator*(const mat3& A, const mat3& B) {
  mat3 R;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r)
      R[c][r] = A[0][r] * B[c][0] + A[1][r] * B[c][1] + A[2][r] * B[c][2];
  return R;
}

// glm operator*(mat3, vec3): m[0]*v.x + m[1]*v.y + m[2]*v.z per row
inline vec3 operator*(const mat3& m, const vec3& v) {
  return {m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z,
          m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z,
          m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z};
}

inline float determinant(const mat3& m) {
  return +m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) -
         m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) +
         m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
}

inline mat3 inverse(const mat3& m) {
  float ood = 1.0f / determinant(m);
  mat3 I;
  I[0][0] = +(m[1][1] * m[2][2] - m[2][1] * m[1][2]) * ood;
  I[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]) * ood;
  I[2][0] = +(m[1][0] * m[2][1] - m[2][0] * m[1][1]) * ood;
  I[0][1] = -(m[0][1] * m[2][2] - m[2][1] * m[0][2]) * ood;
  I[1][1] = +(m[0][0] * m[2][2] - m[2][0] * m[0][2]) * ood;
  I[2][1] = -(m[0][0] * m[2][1] - m[2][0] * m[0][1]) * ood;
  I[0][2] = +(m[0][1] * m[1][2] - m[1][1] * m[0][2]) * ood;
  I[1][2] = -(m[0][0] * m[1][2] - m[1][0] * m[0][2]) * ood;
  I[2][2] = +(m[0][0] * m[1][1] - m[1][0] * m[0][1]) * ood;
  return I;
}

inline mat3 transpose(const mat3& m) {
  mat3 T;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) T[c][r] = m[r][c];
  return T;
}

// Column-major 4x4 (only used for region transforms; setup time).
struct mat4 {
  float m[16];  // m[4*col + row]
};
// glm::inverse(mat4): the published cofactor algorithm (glm/detail/func_matrix.inl, compute_inverse<4,4>).
inline mat4 inverse(const mat4& A) {
  auto m = [&](int c, int r) { return A.m[4 * c + r]; };
  float C00 = m(2, 2) * m(3, 3) - m(3, 2) * m(2, 3), C02 = m(1, 2) * m(3, 3) - m(3, 2) * m(1, 3), C03 = m(1, 2) * m(2, 3) - m(2, 2) * m(1, 3);
  float C04 = m(2, 1) * m(3, 3) - m(3, 1) * m(2, 3), C06 = m(1, 1) * m(3, 3) - m(3, 1) * m(1, 3), C07 = m(1, 1) * m(2, 3) - m(2, 1) * m(1, 3);
  float C08 = m(2, 1) * m(3, 2) - m(3, 1) * m(2, 2), C10 = m(1, 1) * m(3, 2) - m(3, 1) * m(1, 2), C11 = m(1, 1) * m(2, 2) - m(2, 1) * m(1, 2);
  float C12 = m(2, 0) * m(3, 3) - m(3, 0) * m(2, 3), C14 = m(1, 0) * m(3, 3) - m(3, 0) * m(1, 3), C15 = m(1, 0) * m(2, 3) - m(2, 0) * m(1, 3);
  float C16 = m(2, 0) * m(3, 2) - m(3, 0) * m(2, 2), C18 = m(1, 0) * m(3, 2) - m(3, 0) * m(1, 2), C19 = m(1, 0) * m(2, 2) - m(2, 0) * m(1, 2);
  float C20 = m(2, 0) * m(3, 1) - m(3, 0) * m(2, 1), C22 = m(1, 0) * m(3, 1) - m(3, 0) * m(1, 1), C23 = m(1, 0) * m(2, 1) - m(2, 0) * m(1, 1);
  float F0[4] = {C00, C00, C02, C03}, F1[4] = {C04, C04, C06, C07}, F2[4] = {C08, C08, C10, C11};
  float F3[4] = {C12, C12, C14, C15}, F4[4] = {C16, C16, C18, C19}, F5[4] = {C20, C20, C22, C23};
  float V0[4] = {m(1, 0), m(0, 0), m(0, 0), m(0, 0)}, V1[4] = {m(1, 1), m(0, 1), m(0, 1), m(0, 1)};
  float V2[4] = {m(1, 2), m(0, 2), m(0, 2), m(0, 2)}, V3[4] = {m(1, 3), m(0, 3), m(0, 3), m(0, 3)};
  const float SA[4] = {+1, -1, +1, -1}, SB[4] = {-1, +1, -1, +1};
  float I[4][4];
  for (int r = 0; r < 4; ++r) {
    I[0][r] = (V1[r] * F0[r] - V2[r] * F1[r] + V3[r] * F2[r]) * SA[r];
    I[1][r] = (V0[r] * F0[r] - V2[r] * F3[r] + V3[r] * F4[r]) * SB[r];
    I[2][r] = (V0[r] * F1[r] - V1[r] * F3[r] + V3[r] * F5[r]) * SA[r];
    I[3][r] = (V0[r] * F2[r] - V1[r] * F4[r] + V2[r] * F5[r]) * SB[r];
  }
  float Dot1 = (m(0, 0) * I[0][0] + m(0, 1) * I[1][0]) + (m(0, 2) * I[2][0] + m(0, 3) * I[3][0]);
  float ood = 1.0f / Dot1;
  mat4 R;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) R.m[4 * c + r] = I[c][r] * ood;
  return R;
}
// glm operator*(mat4, mat4): per column, ((a0*b0 + a1*b1) + a2*b2) + a3*b3
inline mat4 operator*(const mat4& a, const mat4& b) {
  mat4 R;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      R.m[4 * c + r] = ((a.m[r] * b.m[4 * c] + a.m[4 + r] * b.m[4 * c + 1]) + a.m[8 + r] * b.m[4 * c + 2]) + a.m[12 + r] * b.m[4 * c + 3];
  return R;
}
// glm operator*(mat4, vec4) sums as (m0*x + m1*y) + (m2*z + m3*w); here w = 1.
inline void mul_point(const mat4& M, const vec3& p, float out[4]) {
  for (int r = 0; r < 4; ++r)
    out[r] = (M.m[0 + r] * p.x + M.m[4 + r] * p.y) + (M.m[8 + r] * p.z + M.m[12 + r] * 1.0f);
}

// ---------------------------------------------------------------------------------------------
// 3x3 SVD by one-sided (Hestenes) Jacobi.  The reference calls Eigen::JacobiSVD<Matrix3f> with
// full U and V (/root/reference/Src/Constraints.cpp:97-99, 225-227); Eigen (fork
// nithinp7/eigen.git, /root/reference/.gitmodules:10-12) is an empty, un-vendored submodule, so its
// rounding cannot be reproduced.  What the hot path consumes is only  U * f(S) * V^T, which is
// independent of the SVD algorithm and of the sign/order conventions of U and V; this routine
// therefore restates the mathematical definition:  A*V = B with orthogonal columns,
// s_i = |b_i|, u_i = b_i / s_i.
//
// Column pairs are rotated until every pair is orthogonal to working precision
// (|b_p.b_q| <= kSvdTol * |b_p||b_q|), like Eigen's JacobiSVD sweeps until no off-diagonal exceeds
// its threshold.  The rotation for a pair with a = |b_p|^2, b = |b_q|^2, g = b_p.b_q is the one
// that zeroes the new inner product:  tan(theta) = sign(b-a)*2g / (|b-a| + sqrt((b-a)^2 + 4g^2)).
// Products and sums inside this routine use fused multiply-adds (std::fmaf is exact and portable).
// ---------------------------------------------------------------------------------------------
#ifndef ORA_SVD_MAX_SWEEPS
#define ORA_SVD_MAX_SWEEPS 8
#endif
constexpr int kSvdMaxSweeps = ORA_SVD_MAX_SWEEPS;
#ifndef ORA_SVD_TOL
#define ORA_SVD_TOL 4.76837158203125e-07f /* 4 * 2^-23 */
#endif
constexpr float kSvdTol2 = ORA_SVD_TOL * ORA_SVD_TOL;
constexpr float kSvdTiny = 1.0e-18f;   // a singular value below this is a collapsed direction
constexpr float kSvdTiny2 = 1.0e-36f;  // squared norms (or their products) below this: numerically zero columns

struct Svd3 {
  float b[3][3];  // b[i] = i-th column of A*V  (= s_i * u_i)
  float v[3][3];  // v[i] = i-th column of V
  float s[3];     // singular values (>= 0, unsorted)
  float rs[3];    // 1 / s[i]
  int sweeps = 0;      // sweeps of the certifying loop that rotated, + 1 (statistics)
  int rotations = 0;   // rotations applied (statistics)
  bool closed_form = false;
};

// 1/sqrt(x) from an integer seed and three Newton steps (relative error ~1e-7).  Only *, fma and integer operations:
// the same bits on any IEEE machine.  The SVD normalises with it instead of sqrt + division (the SVD is this
// repository's own restatement of Eigen's JacobiSVD, see the header comment; nothing restated literally from the
// reference uses it).
inline float rsqrt_nr(float x) {
  int32_t i;
  std::memcpy(&i, &x, 4);
  i = 0x5f3759df - (i >> 1);
  float y;
  std::memcpy(&y, &i, 4);
  const float hx = 0.5f * x;
  y = y * std::fmaf(-hx, y * y, 1.5f);
  y = y * std::fmaf(-hx, y * y, 1.5f);
  y = y * std::fmaf(-hx, y * y, 1.5f);
  return y;
}

inline float dot3f(const float* x, const float* y) { return std::fmaf(x[2], y[2], std::fmaf(x[1], y[1], x[0] * y[0])); }

inline bool pair_needs(float alpha, float beta, float gamma) {
  // + kSvdTiny2: numerically zero columns (collapsed element) are never rotated against; for every other pair the fused
  // sum rounds to kSvdTol2 * (alpha * beta) itself
  return gamma * gamma > std::fmaf(kSvdTol2, alpha * beta, kSvdTiny2);
}
inline void jacobi_rotate(Svd3& d, int p, int q, float alpha, float beta, float gamma) {
  float* bp = d.b[p];
  float* bq = d.b[q];
  const float delta = beta - alpha;
  const float g2 = gamma + gamma;
  const float hw = std::fmaf(delta, delta, g2 * g2);
  const float h = hw * rsqrt_nr(hw);
  const float c1 = h + std::fabs(delta);           // proportional to cos(theta)
  const float s1 = delta < 0.0f ? -g2 : g2;        // proportional to sin(theta)
  const float inv = rsqrt_nr(std::fmaf(c1, c1, s1 * s1));
  const float cs = c1 * inv, sn = s1 * inv;
  for (int k = 0; k < 3; ++k) {
    const float x = bp[k], y = bq[k];
    bp[k] = std::fmaf(cs, x, -(sn * y));
    bq[k] = std::fmaf(sn, x, cs * y);
    const float vx = d.v[p][k], vy = d.v[q][k];
    d.v[p][k] = std::fmaf(cs, vx, -(sn * vy));
    d.v[q][k] = std::fmaf(sn, vx, cs * vy);
  }
}
inline bool jacobi_pair(Svd3& d, int p, int q) {
  const float alpha = dot3f(d.b[p], d.b[p]);
  const float beta = dot3f(d.b[q], d.b[q]);
  const float gamma = dot3f(d.b[p], d.b[q]);
  if (!pair_needs(alpha, beta, gamma)) return false;
  jacobi_rotate(d, p, q, alpha, beta, gamma);
  return true;
}
inline void svd3_finish(Svd3& d) {
  for (int i = 0; i < 3; ++i) {
    const float n2 = dot3f(d.b[i], d.b[i]);
    d.rs[i] = n2 > kSvdTiny2 ? rsqrt_nr(n2) : 0.0f;
    d.s[i] = n2 * d.rs[i];
  }
}

// The plain iteration from V = I (what rounds 1-5 ran).  a[r][c] row-major input.
inline Svd3 svd3_jacobi(const float a[3][3]) {
  Svd3 d;
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) {
      d.b[i][k] = a[k][i];
      d.v[i][k] = (i == k) ? 1.0f : 0.0f;
    }
  for (d.sweeps = 0; d.sweeps < kSvdMaxSweeps; ++d.sweeps) {
    bool r01 = jacobi_pair(d, 0, 1);
    bool r02 = jacobi_pair(d, 0, 2);
    bool r12 = jacobi_pair(d, 1, 2);
    d.rotations += (r01 ? 1 : 0) + (r02 ? 1 : 0) + (r12 ? 1 : 0);
    if (!(r01 || r02 || r12)) break;
  }
  svd3_finish(d);
  return d;
}

// Round 6: the iteration is started where it would end.  S = A^T A is symmetric 3x3, so the frame the sweeps converge to
// has a closed form: the eigenvector n of S's most isolated eigenvalue (trigonometric solution of the characteristic cubic,
// with cos(acos(x)/3) on [0, 1] as a degree-7 polynomial; n = the largest column of adj(S - lambda I)), an orthonormal
// completion V0 = [t1, t2, n] (Duff et al., JCGT 2017), B = A V0 and one rotation of the pair (0, 1).  The sweeps then run
// until one passes without a rotation: they certify every pair to the same tolerance as before and repair what the closed
// form left (ill-conditioned A), so the result is that of the one-sided iteration.  An element with at most one pair out
// of tolerance takes that rotation from the entries of S instead.  The device (dev_math.h svd3) executes this very IEEE
// sequence.  Only U f(S) V^T is consumed, which does not depend on how the decomposition was found.
constexpr float kCos3[8] = {8.660253882e-01f, 1.666651964e-01f,  -4.807964712e-02f, 2.440584078e-02f,
                            -1.432729699e-02f, 7.718813606e-03f, -2.961986931e-03f, 5.536798271e-04f};
inline float recip12(float t) {  // 1 / t for t in [1, 2]
  float y = std::fmaf(-0.47058824f, t, 1.4117647f);
  y = y * std::fmaf(-t, y, 2.0f);
  y = y * std::fmaf(-t, y, 2.0f);
  y = y * std::fmaf(-t, y, 2.0f);
  return y;
}
// 1 / x to ~2e-4 (integer seed, two Newton steps), any sign: enough for a correction that is itself of the order of 1e-6
inline float recip_rough(float x) {
  int32_t i;
  std::memcpy(&i, &x, 4);
  i = 0x7EF311C7 - (i & 0x7fffffff);
  float y;
  std::memcpy(&y, &i, 4);
  y = y * std::fmaf(-std::fabs(x), y, 2.0f);
  y = y * std::fmaf(-std::fabs(x), y, 2.0f);
  return std::copysign(y, x);
}
// The small-angle form of the three rotations, applied together and unnormalised: B <- B (I + T), V <- V (I + T), T antisymmetric,
// t_pq = g_pq / (|b_q|^2 - |b_p|^2) from one snapshot of the six inner products (|t| < 2.5e-4: the columns grow by t^2 / 2 < 3e-8).
// What the closed-form frame leaves between its isolated direction and the other two - and what the full rotation of the pair
// (0, 1) leaves between a long and a short column - is a rounding-level angle (~1e-6), but one that an ill-conditioned element
// (a nearly flat one) fails the relative test on; this takes it out without a rotating sweep.
inline void jacobi_polish(Svd3& d) {
  const float n0 = dot3f(d.b[0], d.b[0]), n1 = dot3f(d.b[1], d.b[1]), n2 = dot3f(d.b[2], d.b[2]);
  const float g01 = dot3f(d.b[0], d.b[1]), g02 = dot3f(d.b[0], d.b[2]), g12 = dot3f(d.b[1], d.b[2]);
  float t01 = g01 * recip_rough(n1 - n0), t02 = g02 * recip_rough(n2 - n0), t12 = g12 * recip_rough(n2 - n1);
  if (!(std::fabs(t01) < 2.5e-4f)) t01 = 0.0f;  // (also NaN: equal norms)
  if (!(std::fabs(t02) < 2.5e-4f)) t02 = 0.0f;
  if (!(std::fabs(t12) < 2.5e-4f)) t12 = 0.0f;
  for (int k = 0; k < 3; ++k) {
    const float x = d.b[0][k], y = d.b[1][k], z = d.b[2][k];
    d.b[0][k] = std::fmaf(-t02, z, std::fmaf(-t01, y, x));
    d.b[1][k] = std::fmaf(-t12, z, std::fmaf(t01, x, y));
    d.b[2][k] = std::fmaf(t12, y, std::fmaf(t02, x, z));
    const float vx = d.v[0][k], vy = d.v[1][k], vz = d.v[2][k];
    d.v[0][k] = std::fmaf(-t02, vz, std::fmaf(-t01, vy, vx));
    d.v[1][k] = std::fmaf(-t12, vz, std::fmaf(t01, vx, vy));
    d.v[2][k] = std::fmaf(t12, vy, std::fmaf(t02, vx, vz));
  }
}
inline Svd3 svd3(const float a[3][3]) {
  Svd3 d;
  float A[3][3];  // A[i] = column i
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) A[i][k] = a[k][i];
  const float s00 = dot3f(A[0], A[0]), s11 = dot3f(A[1], A[1]), s22 = dot3f(A[2], A[2]);
  const float s01 = dot3f(A[0], A[1]), s02 = dot3f(A[0], A[2]), s12 = dot3f(A[1], A[2]);
  const bool n01 = pair_needs(s00, s11, s01), n02 = pair_needs(s00, s22, s02), n12 = pair_needs(s11, s22, s12);
  const int cnt = (n01 ? 1 : 0) + (n02 ? 1 : 0) + (n12 ? 1 : 0);
  const float q = ((s00 + s11) + s22) * 0.333333343f;
  const float d0 = s00 - q, d1 = s11 - q, d2 = s22 - q;
  const float p1 = std::fmaf(s12, s12, std::fmaf(s02, s02, s01 * s01));
  const float p2 = std::fmaf(d0, d0, std::fmaf(d1, d1, std::fmaf(d2, d2, p1 + p1)));
  if (cnt >= 2 && p2 > 1.0e-30f && p2 < 1.0e16f) {
    const float w = p2 * 0.166666672f;
    const float ip = rsqrt_nr(w);
    const float p = w * ip;
    const float det = std::fmaf(d0, std::fmaf(d1, d2, -(s12 * s12)),
                                std::fmaf(s02, std::fmaf(s01, s12, -(d1 * s02)), -(s01 * std::fmaf(s01, d2, -(s12 * s02)))));
    const float r = ((0.5f * det) * ip) * (ip * ip);
    const float x = std::fmin(std::fabs(r), 1.0f);
    float c = kCos3[7];
    for (int k = 6; k >= 0; --k) c = std::fmaf(c, x, kCos3[k]);
    const float lam = q + std::copysign((p + p) * c, r);
    const float m00 = s00 - lam, m11 = s11 - lam, m22 = s22 - lam;
    const float c00 = std::fmaf(m11, m22, -(s12 * s12)), c11 = std::fmaf(m00, m22, -(s02 * s02)), c22 = std::fmaf(m00, m11, -(s01 * s01));
    const float c01 = std::fmaf(s02, s12, -(s01 * m22)), c02 = std::fmaf(s01, s12, -(s02 * m11)), c12 = std::fmaf(s01, s02, -(s12 * m00));
    const float a0 = std::fabs(c00), a1 = std::fabs(c11), a2 = std::fabs(c22);
    const bool k0 = a0 >= a1 && a0 >= a2, k1 = !k0 && a1 >= a2;
    const float v0 = k0 ? c00 : (k1 ? c01 : c02), v1 = k0 ? c01 : (k1 ? c11 : c12), v2 = k0 ? c02 : (k1 ? c12 : c22);
    const float n2 = std::fmaf(v2, v2, std::fmaf(v1, v1, v0 * v0));
    const bool okn = n2 > kSvdTiny2;
    const float in = rsqrt_nr(okn ? n2 : 1.0f);
    const float nx = okn ? v0 * in : 0.0f, ny = okn ? v1 * in : 0.0f, nz = okn ? v2 * in : 1.0f;
    const float sg = std::copysign(1.0f, nz);
    const float aa = -recip12(std::fabs(nz) + 1.0f) * sg;  // -1 / (sg + nz)
    const float bb = (nx * ny) * aa;
    const float V[3][3] = {{std::fmaf(sg * nx, nx * aa, 1.0f), sg * bb, -(sg * nx)}, {bb, std::fmaf(ny, ny * aa, sg), -ny}, {nx, ny, nz}};
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) {
        d.v[i][k] = V[i][k];
        d.b[i][k] = std::fmaf(A[2][k], V[i][2], std::fmaf(A[1][k], V[i][1], A[0][k] * V[i][0]));
      }
    d.closed_form = true;
    d.rotations += jacobi_pair(d, 0, 1) ? 1 : 0;
    jacobi_polish(d);
  } else {
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) {
        d.b[i][k] = A[i][k];
        d.v[i][k] = (i == k) ? 1.0f : 0.0f;
      }
    if (n01) jacobi_rotate(d, 0, 1, s00, s11, s01);
    else if (n02) jacobi_rotate(d, 0, 2, s00, s22, s02);
    else if (n12) jacobi_rotate(d, 1, 2, s11, s22, s12);
    d.rotations += cnt != 0 ? 1 : 0;
  }
  // the certifying sweeps: the three pairs tested on one snapshot of the six inner products; a clean snapshot ends the
  // decomposition (its squared norms are the singular values'), otherwise a sweep of the plain iteration runs
  float n0 = s00, n1 = s11, n2 = s22;
  if (cnt != 0) {
    bool clean = false;
    for (d.sweeps = 0; d.sweeps < kSvdMaxSweeps; ++d.sweeps) {
      n0 = dot3f(d.b[0], d.b[0]); n1 = dot3f(d.b[1], d.b[1]); n2 = dot3f(d.b[2], d.b[2]);
      const float g02 = dot3f(d.b[0], d.b[2]), g12 = dot3f(d.b[1], d.b[2]), g01 = dot3f(d.b[0], d.b[1]);
      clean = !(pair_needs(n0, n2, g02) || pair_needs(n1, n2, g12) || pair_needs(n0, n1, g01));
      if (clean) break;
      const bool r02 = jacobi_pair(d, 0, 2);
      const bool r12 = jacobi_pair(d, 1, 2);
      const bool r01 = jacobi_pair(d, 0, 1);
      d.rotations += (r02 ? 1 : 0) + (r12 ? 1 : 0) + (r01 ? 1 : 0);
    }
    if (!clean) { n0 = dot3f(d.b[0], d.b[0]); n1 = dot3f(d.b[1], d.b[1]); n2 = dot3f(d.b[2], d.b[2]); }
  }
  const float nn[3] = {n0, n1, n2};
  for (int i = 0; i < 3; ++i) {
    const bool ok = nn[i] > kSvdTiny2;
    const float r = rsqrt_nr(ok ? nn[i] : 1.0f);
    d.rs[i] = ok ? r : 0.0f;
    d.s[i] = nn[i] * d.rs[i];
  }
  return d;
}


// out[r][c] = sum_i u_i[r] * snew[i] * v_i[c], with u_i = b_i / s_i, evaluated as
// sum_i (b_i[r] * (snew[i]/s_i)) * v_i[c].  A direction whose singular value underflows (collapsed
// element) gets u_i from the oriented completion of the other two; if two collapse the terms are
// dropped (the reference's result is arbitrary there as well).
inline void svd3_recompose(const Svd3& d, const float snew[3], float out[3][3]) {
  float t[3][3];  // t[i] = u_i * snew[i]
  bool ok[3];
  int nbad = 0;
  for (int i = 0; i < 3; ++i) {
    ok[i] = d.s[i] > kSvdTiny;
    if (!ok[i]) ++nbad;
    float g = ok[i] ? snew[i] * d.rs[i] : 0.0f;
    for (int k = 0; k < 3; ++k) t[i][k] = d.b[i][k] * g;
  }
  if (nbad == 1) {
    int k = !ok[0] ? 0 : (!ok[1] ? 1 : 2);
    int i = (k + 1) % 3, j = (k + 2) % 3;
    float ui[3], uj[3];
    for (int c = 0; c < 3; ++c) { ui[c] = d.b[i][c] * d.rs[i]; uj[c] = d.b[j][c] * d.rs[j]; }
    // orientation of the completion: det(U) * det(V) = +1, and V is a rotation by construction (the identity or the
    // completed frame, times Givens rotations)
    float sg = snew[k];
    t[k][0] = sg * (ui[1] * uj[2] - ui[2] * uj[1]);
    t[k][1] = sg * (ui[2] * uj[0] - ui[0] * uj[2]);
    t[k][2] = sg * (ui[0] * uj[1] - ui[1] * uj[0]);
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      out[r][c] = std::fmaf(t[2][r], d.v[2][c], std::fmaf(t[1][r], d.v[1][c], t[0][r] * d.v[0][c]));
}

}  // namespace ora
