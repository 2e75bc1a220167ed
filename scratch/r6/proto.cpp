// Prototype of the analytic-start SVD (round 6): CPU, same IEEE sequence the device would run.
// g++ -O2 -ffp-contract=off -shared -fPIC proto.cpp -o proto.so
#include <cmath>
#include <cstdint>
#include <cstring>

static inline float rsqrt_nr(float x) {
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
// 1/t for t in [1,2]
static inline float recip12(float t) {
  float y = std::fmaf(-0.47058824f, t, 1.4117647f);  // 24/17 - 8/17 t: max rel err 1/17
  y = y * std::fmaf(-t, y, 2.0f);
  y = y * std::fmaf(-t, y, 2.0f);
  y = y * std::fmaf(-t, y, 2.0f);
  return y;
}
static inline float dot3f(const float* x, const float* y) { return std::fmaf(x[2], y[2], std::fmaf(x[1], y[1], x[0] * y[0])); }

static float kTol = 4.76837158203125e-07f;
static const float kTiny2 = 1.0e-36f, kTiny = 1.0e-18f;
struct Svd3 {
  float b[3][3], v[3][3], s[3], rs[3];
  int rot[8];
};
static int g_mode = 0;
static const double* g_lam = nullptr; static int g_idx = 0;

static inline bool jacobi_pair(Svd3& d, int p, int q) {
  float* bp = d.b[p];
  float* bq = d.b[q];
  const float alpha = dot3f(bp, bp), beta = dot3f(bq, bq), gamma = dot3f(bp, bq);
  if (!(gamma * gamma > std::fmaf(kTol * kTol, alpha * beta, kTiny2))) return false;
  const float delta = beta - alpha;
  const float g2 = gamma + gamma;
  const float hw = std::fmaf(delta, delta, g2 * g2);
  const float h = hw * rsqrt_nr(hw);
  const float c1 = h + std::fabs(delta);
  const float s1 = delta < 0.0f ? -g2 : g2;
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
  return true;
}
static inline void finish(Svd3& d) {
  for (int i = 0; i < 3; ++i) {
    const float n2 = dot3f(d.b[i], d.b[i]);
    d.rs[i] = n2 > kTiny2 ? rsqrt_nr(n2) : 0.0f;
    d.s[i] = n2 * d.rs[i];
  }
}
static void svd_old(const float a[3][3], Svd3& d) {
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) { d.b[i][k] = a[k][i]; d.v[i][k] = i == k ? 1.0f : 0.0f; }
  for (int i = 0; i < 8; ++i) d.rot[i] = 0;
  for (int sw = 0; sw < 8; ++sw) {
    bool r01 = jacobi_pair(d, 0, 1), r02 = jacobi_pair(d, 0, 2), r12 = jacobi_pair(d, 1, 2);
    d.rot[0] += r01 + r02 + r12;
    d.rot[1]++;
    if (!(r01 || r02 || r12)) break;
  }
  finish(d);
}

static const float kC[8] = {8.660253882e-01f, 1.666651964e-01f, -4.807964712e-02f, 2.440584078e-02f,
                            -1.432729699e-02f, 7.718813606e-03f, -2.961986931e-03f, 5.536798271e-04f};

// v2: S = A^T A first; <= 1 pair out of tolerance -> plain Jacobi; else analytic start, then the certified loop
static inline bool pair_test(float a, float b, float g) { return g * g > std::fmaf(kTol * kTol, a * b, kTiny2); }
static void svd_new(const float a[3][3], Svd3& d) {
  float A[3][3];  // A[i] = column i
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) A[i][k] = a[k][i];
  for (int i = 0; i < 8; ++i) d.rot[i] = 0;
  const float s00 = dot3f(A[0], A[0]), s11 = dot3f(A[1], A[1]), s22 = dot3f(A[2], A[2]);
  const float s01 = dot3f(A[0], A[1]), s02 = dot3f(A[0], A[2]), s12 = dot3f(A[1], A[2]);
  const int cnt = pair_test(s00, s11, s01) + pair_test(s00, s22, s02) + pair_test(s11, s22, s12);
  const float q = ((s00 + s11) + s22) * 0.333333343f;
  const float d0 = s00 - q, d1 = s11 - q, d2 = s22 - q;
  const float p1 = std::fmaf(s12, s12, std::fmaf(s02, s02, s01 * s01));
  const float p2 = std::fmaf(d0, d0, std::fmaf(d1, d1, std::fmaf(d2, d2, p1 + p1)));
  const bool analytic = cnt >= 2 && p2 > 1.0e-30f && p2 < 1.0e16f;
  d.rot[6] = analytic;
  d.rot[7] = cnt;
  if (!analytic) {
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) { d.b[i][k] = A[i][k]; d.v[i][k] = i == k ? 1.0f : 0.0f; }
    if (cnt == 0) { finish(d); return; }
  } else {
    float n[3] = {0.0f, 0.0f, 1.0f};
    const float w = p2 * 0.166666672f;
    const float ip = rsqrt_nr(w);
    const float p = w * ip;
    const float det = std::fmaf(d0, std::fmaf(d1, d2, -(s12 * s12)), std::fmaf(s02, std::fmaf(s01, s12, -(d1 * s02)), -(s01 * std::fmaf(s01, d2, -(s12 * s02)))));
    const float r = ((0.5f * det) * ip) * (ip * ip);
    const float x = std::fmin(std::fabs(r), 1.0f);
    float c = kC[7];
    for (int k = 6; k >= 0; --k) c = std::fmaf(c, x, kC[k]);
    const float lam = q + std::copysign((p + p) * c, r);
    const float m00 = s00 - lam, m11 = s11 - lam, m22 = s22 - lam;
    const float c00 = std::fmaf(m11, m22, -(s12 * s12)), c11 = std::fmaf(m00, m22, -(s02 * s02)), c22 = std::fmaf(m00, m11, -(s01 * s01));
    const float c01 = std::fmaf(s02, s12, -(s01 * m22)), c02 = std::fmaf(s01, s12, -(s02 * m11)), c12 = std::fmaf(s01, s02, -(s12 * m00));
    const float a0 = std::fabs(c00), a1 = std::fabs(c11), a2 = std::fabs(c22);
    float v0, v1, v2;
    if (a0 >= a1 && a0 >= a2) { v0 = c00; v1 = c01; v2 = c02; }
    else if (a1 >= a2) { v0 = c01; v1 = c11; v2 = c12; }
    else { v0 = c02; v1 = c12; v2 = c22; }
    const float n2 = std::fmaf(v2, v2, std::fmaf(v1, v1, v0 * v0));
    if (n2 > kTiny2) {
      const float in = rsqrt_nr(n2);
      n[0] = v0 * in; n[1] = v1 * in; n[2] = v2 * in;
    }
    const float sg = std::copysign(1.0f, n[2]);
    const float aa = -recip12(std::fabs(n[2]) + 1.0f) * sg;  // -1/(sg + z)
    const float bb = (n[0] * n[1]) * aa;
    float V[3][3];
    V[0][0] = std::fmaf(sg * n[0], n[0] * aa, 1.0f); V[0][1] = sg * bb; V[0][2] = -(sg * n[0]);
    V[1][0] = bb; V[1][1] = std::fmaf(n[1], n[1] * aa, sg); V[1][2] = -n[1];
    V[2][0] = n[0]; V[2][1] = n[1]; V[2][2] = n[2];
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) {
        d.v[i][k] = V[i][k];
        d.b[i][k] = std::fmaf(A[2][k], V[i][2], std::fmaf(A[1][k], V[i][1], A[0][k] * V[i][0]));
      }
  }
  for (int sw = 0; sw < 8; ++sw) {
    bool r01 = jacobi_pair(d, 0, 1), r02 = jacobi_pair(d, 0, 2), r12 = jacobi_pair(d, 1, 2);
    d.rot[0] += r01 + r02 + r12;
    d.rot[1]++;
    if (sw == 0) { d.rot[3] = r02; d.rot[4] = r12; d.rot[2] = r01; }
    if (sw == 1) { d.rot[5] = r01 + r02 + r12; }
    if (!(r01 || r02 || r12)) break;
  }
  finish(d);
}

static void recompose(const Svd3& d, const float snew[3], float out[3][3]) {
  float t[3][3];
  for (int i = 0; i < 3; ++i) {
    const bool ok = d.s[i] > kTiny;
    const float g = ok ? snew[i] * d.rs[i] : 0.0f;
    for (int k = 0; k < 3; ++k) t[i][k] = d.b[i][k] * g;
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out[r][c] = std::fmaf(t[2][r], d.v[2][c], std::fmaf(t[1][r], d.v[1][c], t[0][r] * d.v[0][c]));
}
static float det3(const float a[3][3]) {
  return a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) + a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
}

extern "C" {
void set_tol(float t) { kTol = t; }
// a: n x 9 row-major; out: n x 9 fixed(F); stats: n x 8 ints
void set_dbg(int m, const double* lam) { g_mode = m; g_lam = lam; }
void fixed_batch(int mode, int n, const float* a, float lo, float hi, float* out, int* stats, float* sv) {
  for (int t = 0; t < n; ++t) {
    float A[3][3], O[3][3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[r][c] = a[9 * t + 3 * r + c];
    Svd3 d; g_idx = t;
    if (mode == 0) svd_old(A, d); else svd_new(A, d);
    float s[3];
    for (int i = 0; i < 3; ++i) s[i] = std::fmin(std::fmax(d.s[i], lo), hi);
    if (det3(A) < 0.0f) {
      int k = 0;
      if (d.s[1] <= d.s[k]) k = 1;
      if (d.s[2] <= d.s[k]) k = 2;
      s[k] = -s[k];
    }
    recompose(d, s, O);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out[9 * t + 3 * r + c] = O[r][c];
    for (int i = 0; i < 8; ++i) stats[8 * t + i] = d.rot[i];
    for (int i = 0; i < 3; ++i) sv[3 * t + i] = d.s[i];
  }
}
}
