// The PD local step of a strain + volume element pair, TWO ELEMENTS PER LANE in packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 process two floats per lane and issue slot on gfx950).  k_pd_local_tet_pair is bound by VALU issue at 100k
// particles - ~2 000 instructions per element, 23 us per launch, every wavefront of the chip busy - so the way to shorten it is
// fewer instructions per element: the same one-sided Jacobi SVD (dev_math.h), projections and A^T products, written on
// two-component vectors.  Differences from the scalar code are roundings only (PD parity is by tolerance):
// a column pair is rotated when EITHER element needs it (the other one's rotation is replaced by the identity), so the sweep
// counts of the two are those of the slower one, which does nothing to a converged decomposition.
// Included by pd_kernels.hip only (after dev_math.h and the scalar local step).
#pragma once

namespace pies {

typedef float f2 __attribute__((ext_vector_type(2)));

PIES_DEV f2 splat2(float a) { return f2{a, a}; }
PIES_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
PIES_DEV f2 abs2(f2 a) { return __builtin_elementwise_abs(a); }
PIES_DEV f2 sel2(bool cx, bool cy, f2 a, f2 b) { return f2{cx ? a.x : b.x, cy ? a.y : b.y}; }
PIES_DEV f2 clamp2(f2 v, f2 lo, f2 hi) { return __builtin_elementwise_min(__builtin_elementwise_max(v, lo), hi); }

// rsqrt_nr (dev_math.h) on both components
PIES_DEV f2 rsqrt_nr2(f2 x) {
  f2 y = {__int_as_float(0x5f3759df - (__float_as_int(x.x) >> 1)), __int_as_float(0x5f3759df - (__float_as_int(x.y) >> 1))};
  const f2 hx = splat2(0.5f) * x, c = splat2(1.5f);
  y = y * fma2(-hx, y * y, c);
  y = y * fma2(-hx, y * y, c);
  y = y * fma2(-hx, y * y, c);
  return y;
}
// 1 / x: the hardware's approximation and one Newton step (~1 ulp)
PIES_DEV f2 rcp2(f2 x) {
  const f2 r = {__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
  return r * fma2(-x, r, splat2(2.0f));
}
PIES_DEV f2 dot3p(const f2 x[3], const f2 y[3]) { return fma2(x[2], y[2], fma2(x[1], y[1], x[0] * y[0])); }

struct Svd3P {
  f2 b[3][3], v[3][3], s[3], rs[3];
};

template <int P, int Q> PIES_DEV bool jacobi_pair_p(Svd3P& d) {
  const f2 alpha = dot3p(d.b[P], d.b[P]), beta = dot3p(d.b[Q], d.b[Q]), gamma = dot3p(d.b[P], d.b[Q]);
  const f2 thr = fma2(splat2(kSvdTol2), alpha * beta, splat2(kSvdTiny2)), gg = gamma * gamma;
  const bool nx = gg.x > thr.x, ny = gg.y > thr.y;
  if (!(nx || ny)) return false;
  const f2 delta = beta - alpha, g2 = gamma + gamma;
  const f2 hw = fma2(delta, delta, g2 * g2);
  const f2 h = hw * rsqrt_nr2(hw);
  const f2 c1 = h + abs2(delta);
  const f2 s1 = {delta.x < 0.0f ? -g2.x : g2.x, delta.y < 0.0f ? -g2.y : g2.y};
  const f2 inv = rsqrt_nr2(fma2(c1, c1, s1 * s1));
  const f2 cs = sel2(nx, ny, c1 * inv, splat2(1.0f)), sn = sel2(nx, ny, s1 * inv, splat2(0.0f));
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const f2 x = d.b[P][k], y = d.b[Q][k];
    d.b[P][k] = fma2(cs, x, -(sn * y));
    d.b[Q][k] = fma2(sn, x, cs * y);
    const f2 vx = d.v[P][k], vy = d.v[Q][k];
    d.v[P][k] = fma2(cs, vx, -(sn * vy));
    d.v[Q][k] = fma2(sn, vx, cs * vy);
  }
  return true;
}

PIES_DEV void svd3p(const f2 a[3][3], Svd3P& d) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      d.b[i][k] = a[k][i];
      d.v[i][k] = splat2((i == k) ? 1.0f : 0.0f);
    }
  for (int sweep = 0; sweep < kSvdMaxSweeps; ++sweep) {
    const bool r01 = jacobi_pair_p<0, 1>(d);
    const bool r02 = jacobi_pair_p<0, 2>(d);
    const bool r12 = jacobi_pair_p<1, 2>(d);
    if (!(r01 || r02 || r12)) break;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const f2 n2 = dot3p(d.b[i], d.b[i]);
    d.rs[i] = sel2(n2.x > kSvdTiny2, n2.y > kSvdTiny2, rsqrt_nr2(n2), splat2(0.0f));
    d.s[i] = n2 * d.rs[i];
  }
}

// one component of the packed decomposition as the scalar structure (for the rare scalar paths)
template <int E> PIES_DEV void svd3p_extract(const Svd3P& d, Svd3& o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      o.b[i][k] = d.b[i][k][E];
      o.v[i][k] = d.v[i][k][E];
    }
    o.s[i] = d.s[i][E];
    o.rs[i] = d.rs[i][E];
  }
}

// U diag(snew) V^T of both elements (svd3_recompose); an element with a collapsed direction takes the scalar routine
PIES_DEV void svd3p_recompose(const Svd3P& d, const f2 snew[3], f2 out[3][3]) {
  f2 t[3][3];
  bool badx = false, bady = false;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const bool okx = d.s[i].x > kSvdTiny, oky = d.s[i].y > kSvdTiny;
    badx = badx || !okx;
    bady = bady || !oky;
    const f2 g = sel2(okx, oky, snew[i] * d.rs[i], splat2(0.0f));
#pragma unroll
    for (int k = 0; k < 3; ++k) t[i][k] = d.b[i][k] * g;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) out[r][c] = fma2(t[2][r], d.v[2][c], fma2(t[1][r], d.v[1][c], t[0][r] * d.v[0][c]));
  if (badx) {  // rare: a flattened element
    Svd3 e;
    svd3p_extract<0>(d, e);
    const float sn[3] = {snew[0].x, snew[1].x, snew[2].x};
    float o[3][3];
    svd3_recompose(e, sn, o);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) out[r][c].x = o[r][c];
  }
  if (bady) {
    Svd3 e;
    svd3p_extract<1>(d, e);
    const float sn[3] = {snew[0].y, snew[1].y, snew[2].y};
    float o[3][3];
    svd3_recompose(e, sn, o);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) out[r][c].y = o[r][c];
  }
}

// VolumeConstraint's ten Newton-like iterations (compute_d) on both elements
PIES_DEV void compute_d_p(const f2 s[3], f2 omegaMin, f2 omegaMax, f2 D[3]) {
  D[0] = D[1] = D[2] = splat2(0.0f);
  for (int it = 0; it < 10; ++it) {
    const f2 sx = s[0] + D[0], sy = s[1] + D[1], sz = s[2] + D[2];
    const f2 product = sx * sy * sz;
    const f2 C = product - clamp2(product, omegaMin, omegaMax);
    const f2 gx = sy * sz, gy = sx * sz, gz = sx * sy;
    const f2 num = (gx * D[0] + gy * D[1] + gz * D[2]) - C;
    const f2 den = gx * gx + gy * gy + gz * gz;
    const f2 q = num * rcp2(den);
    D[0] = q * gx;
    D[1] = q * gy;
    D[2] = q * gz;
  }
}

// TetrahedralConstraint's projection of the singular values of one element (tet_project<false>): clamp, and the smallest one
// negated in an inverted element
PIES_DEV void strain_project(const float ds[3], float detF, float lo, float hi, float s[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) s[i] = clampf(ds[i], lo, hi);
  if (detF < 0.0f) {
    int k = 0;
    float m = ds[0];
    if (ds[1] <= m) { k = 1; m = ds[1]; }
    if (ds[2] <= m) { k = 2; }
    s[0] = (k == 0) ? -s[0] : s[0];
    s[1] = (k == 1) ? -s[1] : s[1];
    s[2] = (k == 2) ? -s[2] : s[2];
  }
}

// The constants of an element pair - Qinv, the strain limits and weight, the volume limits and weight: 64 bytes - are
// identical for all elements of one shape and material (a createTetBox lattice has one set per orientation of its
// tetrahedra).  When a scene holds few distinct sets (pd_rest_dictionary, capi.cpp) the elements carry a 16-bit index into a
// table of sets instead: 2 bytes per element from HBM instead of 64, the table stays in the vector cache.
struct RestDictionary {
  const uint16_t* index;  // per element pair; nullptr: no dictionary, the per-element arrays are read
  const float4* table;    // 4 per set: q0, q1, q2 (strain), q2 (volume)
};

// The strain + volume projection of two elements at once: positions xa[4] / xb[4], constants (Qinv 9, strain lo / hi / w;
// volume lo / hi / w in the last record) -> rec[i][k] = the sum of both constraints' w (A^T p)_i, component k, elements (x, y)
PIES_DEV void pair_project_packed(const float4 xa[4], const float4 xb[4], const float4 a0, const float4 a1, const float4 a2, const float4 av,
                                  const float4 b0, const float4 b1, const float4 b2, const float4 bv, f2 rec[4][3]) {
  // Qinv [col][row] (tet_frame)
  const f2 qi[3][3] = {{f2{a0.x, b0.x}, f2{a0.y, b0.y}, f2{a0.z, b0.z}},
                       {f2{a0.w, b0.w}, f2{a1.x, b1.x}, f2{a1.y, b1.y}},
                       {f2{a1.z, b1.z}, f2{a1.w, b1.w}, f2{a2.x, b2.x}}};
  const f2 x1[3] = {f2{xa[0].x, xb[0].x}, f2{xa[0].y, xb[0].y}, f2{xa[0].z, xb[0].z}};
  const f2 P[3][3] = {{f2{xa[1].x, xb[1].x} - x1[0], f2{xa[1].y, xb[1].y} - x1[1], f2{xa[1].z, xb[1].z} - x1[2]},
                      {f2{xa[2].x, xb[2].x} - x1[0], f2{xa[2].y, xb[2].y} - x1[1], f2{xa[2].z, xb[2].z} - x1[2]},
                      {f2{xa[3].x, xb[3].x} - x1[0], f2{xa[3].y, xb[3].y} - x1[1], f2{xa[3].z, xb[3].z} - x1[2]}};
  f2 F[3][3];  // F = P * Qinv, column-major (mat3_mul_cm)
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) F[c][r] = fma2(P[2][r], qi[c][2], fma2(P[1][r], qi[c][1], P[0][r] * qi[c][0]));
  Svd3P d;
  svd3p(F, d);
  // strain projection (per element: a clamp and a sign)
  const f2 det = F[0][0] * (F[1][1] * F[2][2] - F[2][1] * F[1][2]) - F[1][0] * (F[0][1] * F[2][2] - F[2][1] * F[0][2]) +
                 F[2][0] * (F[0][1] * F[1][2] - F[1][1] * F[0][2]);
  float sax[3], say[3];
  {
    const float dsx[3] = {d.s[0].x, d.s[1].x, d.s[2].x}, dsy[3] = {d.s[0].y, d.s[1].y, d.s[2].y};
    strain_project(dsx, det.x, a2.y, a2.z, sax);
    strain_project(dsy, det.y, b2.y, b2.z, say);
  }
  // volume projection
  f2 D[3];
  compute_d_p(d.s, f2{av.y, bv.y}, f2{av.z, bv.z}, D);
  // w_strain s_strain + w_volume s_volume: one recomposition, one A^T product (see local_tet_pair)
  const f2 ws = {a2.w, b2.w}, wv = {av.w, bv.w};
  f2 sc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) sc[i] = fma2(ws, f2{sax[i], say[i]}, wv * (d.s[i] + D[i]));
  f2 Fh[3][3];
  svd3p_recompose(d, sc, Fh);
  // (A^T p)_0 = sum_r A0[r] Fh[r], A0[r] = -(q_r0 + q_r1 + q_r2); (A^T p)_{1+c} = sum_r q_rc Fh[r]   (tet_records)
  f2 A0[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) A0[r] = -((qi[r][0] + qi[r][1]) + qi[r][2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rec[0][k] = fma2(A0[2], Fh[2][k], fma2(A0[1], Fh[1][k], A0[0] * Fh[0][k]));
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) rec[1 + cc][k] = fma2(qi[2][cc], Fh[2][k], fma2(qi[1][cc], Fh[1][k], qi[0][cc] * Fh[0][k]));
  }
}

// elements c0 and c1 (c1 == c0 for the odd one out: its result is not written twice)
template <bool DICT>
PIES_DEV void local_tet_pair_packed(const float4* __restrict__ pos, const uint4* __restrict__ ids, const float4* __restrict__ q0,
                                    const float4* __restrict__ q1, const float4* __restrict__ q2, const float4* __restrict__ vq2,
                                    const RestDictionary& dict, Vec3f* __restrict__ contribTet, uint32_t count, uint32_t c0, uint32_t c1) {
  const uint4 ia = ids[c0], ib = ids[c1];
  float4 a0, a1, a2, av, b0, b1, b2, bv;
  if (DICT) {
    const uint32_t ka = 4u * dict.index[c0], kb = 4u * dict.index[c1];
    a0 = dict.table[ka]; a1 = dict.table[ka + 1]; a2 = dict.table[ka + 2]; av = dict.table[ka + 3];
    b0 = dict.table[kb]; b1 = dict.table[kb + 1]; b2 = dict.table[kb + 2]; bv = dict.table[kb + 3];
  } else {
    a0 = q0[c0]; a1 = q1[c0]; a2 = q2[c0]; av = vq2[c0];
    b0 = q0[c1]; b1 = q1[c1]; b2 = q2[c1]; bv = vq2[c1];
  }
  const float4 xa[4] = {pos[ia.x], pos[ia.y], pos[ia.z], pos[ia.w]};
  const float4 xb[4] = {pos[ib.x], pos[ib.y], pos[ib.z], pos[ib.w]};
  f2 rec[4][3];
  pair_project_packed(xa, xb, a0, a1, a2, av, b0, b1, b2, bv, rec);
  if (c1 == c0 + 1u) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      contribTet[i * count + c0] = Vec3f{rec[i][0].x, rec[i][1].x, rec[i][2].x};
      contribTet[i * count + c1] = Vec3f{rec[i][0].y, rec[i][1].y, rec[i][2].y};
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) contribTet[i * count + c0] = Vec3f{rec[i][0].x, rec[i][1].x, rec[i][2].x};
  }
}

}  // namespace pies
