// Scene construction (setup time, host only): node append, constraint factories, lattice primitives.
// Mirrors the reference's host-side API so that node numbering, constraint order and rest state are
// the ones a Pies host would get:
//   addNodes                 Src/PrimitiveUtilities.cpp:42-75
//   create*Constraint        Src/Constraints.cpp:39-56, 65-74, 130-184, 257-310, 368-394
//   createTetBox/createBox/createSheet/createBendSheet   Src/PrimitiveUtilities.cpp:330-976, 1127-1289
#include <cmath>
#include <cstring>

#include "solver_state.h"

namespace pies {

int fail(pies_solver* s, int code, const std::string& msg) {
  if (s) s->error = msg;
  return code;
}

namespace {

struct P3 {
  float x, y, z;
};
inline P3 operator-(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot3(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline P3 cross3(P3 a, P3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline P3 normalize3(P3 a) {
  float inv = 1.0f / std::sqrt(dot3(a, a));
  return {a.x * inv, a.y * inv, a.z * inv};
}
inline P3 node_pos(const pies_solver* s, uint32_t id) { return {s->h_pos[3 * id], s->h_pos[3 * id + 1], s->h_pos[3 * id + 2]}; }

// Inverse of the 3x3 whose columns are e1, e2, e3; result column-major q[3*col+row] (cofactors / det).
void inverse_columns(P3 e1, P3 e2, P3 e3, float q[9]) {
  const float m[3][3] = {{e1.x, e1.y, e1.z}, {e2.x, e2.y, e2.z}, {e3.x, e3.y, e3.z}};  // m[col][row]
  const float det = +m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) +
                    m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
  const float ood = 1.0f / det;
  float I[3][3];
  I[0][0] = +(m[1][1] * m[2][2] - m[2][1] * m[1][2]) * ood;
  I[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]) * ood;
  I[2][0] = +(m[1][0] * m[2][1] - m[2][0] * m[1][1]) * ood;
  I[0][1] = -(m[0][1] * m[2][2] - m[2][1] * m[0][2]) * ood;
  I[1][1] = +(m[0][0] * m[2][2] - m[2][0] * m[0][2]) * ood;
  I[2][1] = -(m[0][0] * m[2][1] - m[2][0] * m[0][1]) * ood;
  I[0][2] = +(m[0][1] * m[1][2] - m[1][1] * m[0][2]) * ood;
  I[1][2] = -(m[0][0] * m[1][2] - m[1][0] * m[0][2]) * ood;
  I[2][2] = +(m[0][0] * m[1][1] - m[1][0] * m[0][1]) * ood;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) q[3 * c + r] = I[c][r];
}

bool ids_ok(const pies_solver* s, const uint32_t* ids, size_t count) {
  const uint32_t n = s->nodeCount();
  for (size_t i = 0; i < count; ++i)
    if (ids[i] >= n) return false;
  return true;
}

// Rest data shared by the tetrahedral-strain and volume factories (Constraints.cpp:140-176, 266-302):
// Qinv = inverse([x2-x1, x3-x1, x4-x1]);  A = [0 ; Qinv_(r,k) * D] with the reference's row-major
// reading of the column-major inverse (diffToBary_(r,k) = diffToBary[r][k]);  B = I.
void tet_matrices(HostTet& t);
HostTet make_tet(const pies_solver* s, const uint32_t ids[4], float w, float lo, float hi) {
  HostTet t{};
  t.hint = kNoColourHint;
  std::memcpy(t.ids, ids, sizeof(t.ids));
  const P3 x1 = node_pos(s, ids[0]);
  inverse_columns(node_pos(s, ids[1]) - x1, node_pos(s, ids[2]) - x1, node_pos(s, ids[3]) - x1, t.qinv);
  t.lo = lo;
  t.hi = hi;
  t.w = w;
  tet_matrices(t);
  return t;
}
// A = [0; Qinv^T D] and A^T A from the rest data (Constraints.cpp:157-184; also after pies_set_rest replaced Qinv)
void tet_matrices(HostTet& t) {
  const float D[3][4] = {{-1, 1, 0, 0}, {-1, 0, 1, 0}, {-1, 0, 0, 1}};
  for (int j = 0; j < 4; ++j) t.A[j] = 0.0f;
  for (int r = 0; r < 3; ++r)
    for (int j = 0; j < 4; ++j) {
      float acc = 0.f;
      for (int k = 0; k < 3; ++k) acc += t.qinv[3 * r + k] * D[k][j];
      t.A[4 * (1 + r) + j] = acc;
    }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float acc = 0.f;
      for (int k = 0; k < 4; ++k) acc += t.A[4 * k + i] * t.A[4 * k + j];
      t.AtA[4 * i + j] = acc;
    }
}

void push_distance(pies_solver* s, uint32_t a, uint32_t b, float w, uint16_t hint = kNoColourHint) {
  HostDistance d;
  d.hint = hint;
  d.ids[0] = a;
  d.ids[1] = b;
  const P3 diff = node_pos(s, b) - node_pos(s, a);
  d.target = std::sqrt(dot3(diff, diff));
  d.w = w;
  s->h_distance.push_back(d);
  ++s->constraintId;
}
void push_position(pies_solver* s, uint32_t id, float w) {
  HostPosition p;
  p.id = id;
  p.target[0] = s->h_pos[3 * id];
  p.target[1] = s->h_pos[3 * id + 1];
  p.target[2] = s->h_pos[3 * id + 2];
  p.w = w;
  s->h_position.push_back(p);
  ++s->constraintId;
}
void push_bend(pies_solver* s, const uint32_t ids[4], float w) {
  HostBend b;
  std::memcpy(b.ids, ids, sizeof(b.ids));
  const P3 x1 = node_pos(s, ids[0]);
  const P3 p2 = node_pos(s, ids[1]) - x1, p3 = node_pos(s, ids[2]) - x1, p4 = node_pos(s, ids[3]) - x1;
  b.angle = static_cast<float>(std::acos(static_cast<double>(dot3(normalize3(cross3(p2, p3)), normalize3(cross3(p2, p4))))));
  b.w = w;
  s->h_bend.push_back(b);
  ++s->constraintId;
}

uint32_t push_node(pies_solver* s, P3 p, P3 v, float radius, float invMass) {
  const uint32_t id = s->nodeCount();
  s->h_pos.insert(s->h_pos.end(), {p.x, p.y, p.z});
  s->h_prev.insert(s->h_prev.end(), {p.x, p.y, p.z});
  s->h_vel.insert(s->h_vel.end(), {v.x, v.y, v.z});
  s->h_radius.push_back(radius);
  s->h_invMass.push_back(invMass);
  return id;
}

// glm::inverse(mat4) (cofactor expansion), column-major m[4*col+row]; setup time only.
void inverse_mat4(const float m[16], float out[16]) {
  auto M = [&](int c, int r) { return m[4 * c + r]; };
  const float c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3), c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3), c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
  const float c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3), c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3), c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
  const float c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2), c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2), c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
  const float c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3), c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3), c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
  const float c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2), c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2), c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
  const float c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1), c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1), c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
  const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
  const float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
  const float v0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)}, v1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)};
  const float v2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)}, v3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
  const float sa[4] = {+1, -1, +1, -1}, sb[4] = {-1, +1, -1, +1};
  float inv[4][4];  // inv[col][row]
  for (int r = 0; r < 4; ++r) {
    inv[0][r] = (v1[r] * f0[r] - v2[r] * f1[r] + v3[r] * f2[r]) * sa[r];
    inv[1][r] = (v0[r] * f0[r] - v2[r] * f3[r] + v3[r] * f4[r]) * sb[r];
    inv[2][r] = (v0[r] * f1[r] - v1[r] * f3[r] + v3[r] * f5[r]) * sa[r];
    inv[3][r] = (v0[r] * f2[r] - v1[r] * f4[r] + v2[r] * f5[r]) * sb[r];
  }
  const float dot1 = (M(0, 0) * inv[0][0] + M(0, 1) * inv[1][0]) + (M(0, 2) * inv[2][0] + M(0, 3) * inv[3][0]);
  const float ood = 1.0f / dot1;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) out[4 * c + r] = inv[c][r] * ood;
}

// glm mat4 * vec4(p, 1): (m0*x + m1*y) + (m2*z + m3*1)
inline void mul_point(const float m[16], P3 p, float out[4]) {
  for (int r = 0; r < 4; ++r) out[r] = (m[r] * p.x + m[4 + r] * p.y) + (m[8 + r] * p.z + m[12 + r] * 1.0f);
}
inline void mul_mat4(const float a[16], const float b[16], float out[16]) {  // glm a*b, column by column
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      out[4 * c + r] = ((a[r] * b[4 * c] + a[4 + r] * b[4 * c + 1]) + a[8 + r] * b[4 * c + 2]) + a[12 + r] * b[4 * c + 3];
}

bool inside_unit_box(const float worldToRegion[16], P3 p) {
  float l[4];
  mul_point(worldToRegion, p, l);
  return -1.0f <= l[0] && l[0] <= 1.0f && -1.0f <= l[1] && l[1] <= 1.0f && -1.0f <= l[2] && l[2] <= 1.0f;
}

void inverse3d(const double m[3][3], double o[9]) {
  const double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1], c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2], c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
  const double id = 1.0 / (m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02);
  o[0] = c00 * id; o[1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id; o[2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
  o[3] = c01 * id; o[4] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id; o[5] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
  o[6] = c02 * id; o[7] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id; o[8] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
}

// ShapeMatchingConstraint constructor (ShapeMatchingConstraint.cpp:6-48): unweighted float centroid of the
// material coordinates, Q = sum (r r^T)/invMass (products in float, sum in double), Qinv in double.
void push_shape(pies_solver* s, const std::vector<uint32_t>& ids, const std::vector<P3>& material, float w) {
  HostShape c;
  c.ids = ids;
  c.w = w;
  const size_t n = material.size();
  c.mat.assign(3 * n, 0.0);
  P3 com{0.f, 0.f, 0.f};
  const float weight = 1.0f / static_cast<float>(n);
  for (const P3& m : material) { com.x += weight * m.x; com.y += weight * m.y; com.z += weight * m.z; }
  double Q[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (size_t i = 0; i < n; ++i) {
    const float mc[3] = {material[i].x - com.x, material[i].y - com.y, material[i].z - com.z};
    for (int k = 0; k < 3; ++k) c.mat[3 * i + k] = mc[k];
    const float im = s->h_invMass[ids[i]];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) Q[r][cc] += (mc[r] * mc[cc]) / im;
  }
  inverse3d(Q, c.Qinv);
  s->h_shape.push_back(std::move(c));
}

void push_goal(pies_solver* s, const std::vector<uint32_t>& ids, float w) {
  HostGoal c;
  c.ids = ids;
  c.w = w;
  c.mat.resize(3 * ids.size());
  for (size_t i = 0; i < ids.size(); ++i)
    for (int k = 0; k < 3; ++k) c.mat[3 * i + k] = s->h_pos[3 * ids[i] + k];
  for (int i = 0; i < 16; ++i) c.transform[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  s->h_goal.push_back(std::move(c));
}

// Grid::gridIdToNodeId (PrimitiveUtilities.cpp:35-38)
inline uint32_t lattice_id(uint32_t H, uint32_t D, uint32_t first, uint32_t x, uint32_t y, uint32_t z) {
  return z + D * (y + H * x) + first;
}

// The 12 x (cells per face) surface triangles shared by createBox and createTetBox.
void lattice_surface(pies_solver* s, uint32_t W, uint32_t H, uint32_t D, uint32_t first) {
  auto G = [&](uint32_t x, uint32_t y, uint32_t z) { return lattice_id(H, D, first, x, y, z); };
  auto tri = [&](uint32_t a, uint32_t b, uint32_t c) { s->h_triangles.insert(s->h_triangles.end(), {a, b, c}); };
  for (uint32_t i = 0; i + 1 < W; ++i)
    for (uint32_t j = 0; j + 1 < H; ++j) {
      tri(G(i, j, 0), G(i + 1, j + 1, 0), G(i + 1, j, 0));
      tri(G(i, j, 0), G(i, j + 1, 0), G(i + 1, j + 1, 0));
      tri(G(i, j, D - 1), G(i + 1, j, D - 1), G(i + 1, j + 1, D - 1));
      tri(G(i, j, D - 1), G(i + 1, j + 1, D - 1), G(i, j + 1, D - 1));
    }
  for (uint32_t i = 0; i + 1 < W; ++i)
    for (uint32_t k = 0; k + 1 < D; ++k) {
      tri(G(i, 0, k), G(i + 1, 0, k), G(i + 1, 0, k + 1));
      tri(G(i, 0, k), G(i + 1, 0, k + 1), G(i, 0, k + 1));
      tri(G(i, H - 1, k), G(i + 1, H - 1, k + 1), G(i + 1, H - 1, k));
      tri(G(i, H - 1, k), G(i, H - 1, k + 1), G(i + 1, H - 1, k + 1));
    }
  for (uint32_t j = 0; j + 1 < H; ++j)
    for (uint32_t k = 0; k + 1 < D; ++k) {
      tri(G(0, j, k), G(0, j + 1, k + 1), G(0, j + 1, k));
      tri(G(0, j, k), G(0, j, k + 1), G(0, j + 1, k + 1));
      tri(G(W - 1, j, k), G(W - 1, j + 1, k), G(W - 1, j + 1, k + 1));
      tri(G(W - 1, j, k), G(W - 1, j + 1, k + 1), G(W - 1, j, k + 1));
    }
}

void lines_from_distances(pies_solver* s, size_t firstDistance) {
  for (size_t i = firstDistance; i < s->h_distance.size(); ++i) {
    s->h_lines.push_back(s->h_distance[i].ids[0]);
    s->h_lines.push_back(s->h_distance[i].ids[1]);
  }
}

}  // namespace

// declared for capi.cpp
int scene_sync_host(pies_solver* s);  // brings the host mirror up to date before a scene edit

}  // namespace pies

using namespace pies;

#define PIES_CHECK_HANDLE(s) \
  if (!(s)) return PIES_ERR_INVALID
#define PIES_BEGIN_EDIT(s)                      \
  PIES_CHECK_HANDLE(s);                         \
  if (int rc__ = scene_sync_host(s)) return rc__; \
  (s)->sceneDirty = true

extern "C" {

int pies_add_nodes_ex(pies_solver_t* s, uint32_t n, const float* pos, const float* vel, const float* radius,
                      const float* inv_mass, uint32_t* first_id) {
  PIES_BEGIN_EDIT(s);
  if (n && !pos) return fail(s, PIES_ERR_INVALID, "pies_add_nodes: pos is NULL");
  const uint32_t first = s->nodeCount();
  if (static_cast<uint64_t>(first) + n > 0xFFFFFFF0ull) return fail(s, PIES_ERR_INVALID, "pies_add_nodes: too many nodes");
  for (uint32_t i = 0; i < n; ++i)
    push_node(s, {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]},
              vel ? P3{vel[3 * i], vel[3 * i + 1], vel[3 * i + 2]} : P3{0.f, 0.f, 0.f}, radius ? radius[i] : 0.5f,
              inv_mass ? inv_mass[i] : 1.0f);
  if (first_id) *first_id = first;
  return PIES_OK;
}

int pies_add_nodes(pies_solver_t* s, uint32_t n, const float* pos, uint32_t* first_id) {
  return pies_add_nodes_ex(s, n, pos, nullptr, nullptr, nullptr, first_id);
}

int pies_add_position_constraints(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w) {
  PIES_BEGIN_EDIT(s);
  if (n && (!ids || !ids_ok(s, ids, n))) return fail(s, PIES_ERR_INVALID, "position constraint: bad node id");
  for (uint32_t i = 0; i < n; ++i) push_position(s, ids[i], w);
  return PIES_OK;
}

int pies_add_distance_constraints(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w) {
  PIES_BEGIN_EDIT(s);
  if (n && (!ids || !ids_ok(s, ids, 2ull * n))) return fail(s, PIES_ERR_INVALID, "distance constraint: bad node id");
  for (uint32_t i = 0; i < n; ++i) push_distance(s, ids[2 * i], ids[2 * i + 1], w);
  return PIES_OK;
}

int pies_add_tet_constraints(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w, float min_strain, float max_strain) {
  PIES_BEGIN_EDIT(s);
  if (n && (!ids || !ids_ok(s, ids, 4ull * n))) return fail(s, PIES_ERR_INVALID, "tet constraint: bad node id");
  for (uint32_t i = 0; i < n; ++i) {
    s->h_tet.push_back(make_tet(s, ids + 4 * i, w, min_strain, max_strain));
    ++s->constraintId;
  }
  return PIES_OK;
}

int pies_add_volume_constraints(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w, float compression, float stretching) {
  PIES_BEGIN_EDIT(s);
  if (n && (!ids || !ids_ok(s, ids, 4ull * n))) return fail(s, PIES_ERR_INVALID, "volume constraint: bad node id");
  for (uint32_t i = 0; i < n; ++i) {
    s->h_volume.push_back(make_tet(s, ids + 4 * i, w, compression, stretching));
    ++s->constraintId;
  }
  return PIES_OK;
}

int pies_add_bend_constraints(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w) {
  PIES_BEGIN_EDIT(s);
  if (n && (!ids || !ids_ok(s, ids, 4ull * n))) return fail(s, PIES_ERR_INVALID, "bend constraint: bad node id");
  for (uint32_t i = 0; i < n; ++i) push_bend(s, ids + 4 * i, w);
  return PIES_OK;
}

int pies_add_node_pair_constraints(pies_solver_t* s, uint32_t n, const uint32_t* ids) {
  PIES_BEGIN_EDIT(s);
  if (n && (!ids || !ids_ok(s, ids, 2ull * n))) return fail(s, PIES_ERR_INVALID, "node-pair constraint: bad node id");
  for (uint32_t i = 0; i < n; ++i) {
    if (ids[2 * i] == ids[2 * i + 1]) return fail(s, PIES_ERR_INVALID, "node-pair constraint: a node paired with itself");
    s->h_nodePair.push_back(pies::HostNodePair{{ids[2 * i], ids[2 * i + 1]}});
  }
  return PIES_OK;
}

/* Rest data of constraints [first, first + n) of a container, replaced (the counterpart of pies_get_rest, same layout): a host that
 * restores a saved scene, or pre-strains a material, does not have to move the nodes into the rest pose first.  The reference's
 * factories take the rest pose from the node positions at creation (Constraints.cpp:39-56, 130-184, 257-310, 368-394). */
int pies_set_rest(pies_solver_t* s, int type, uint32_t first, uint32_t n, const float* rest) {
  PIES_BEGIN_EDIT(s);
  if (n && !rest) return fail(s, PIES_ERR_INVALID, "pies_set_rest: rest is NULL");
  auto in_range = [&](size_t size) { return static_cast<uint64_t>(first) + n <= size; };
  switch (type) {
    case PIES_DISTANCE:
      if (!in_range(s->h_distance.size())) return fail(s, PIES_ERR_INVALID, "pies_set_rest: range beyond the container");
      for (uint32_t i = 0; i < n; ++i) s->h_distance[first + i].target = rest[i];
      break;
    case PIES_TET:
    case PIES_VOLUME: {
      std::vector<HostTet>& list = type == PIES_TET ? s->h_tet : s->h_volume;
      if (!in_range(list.size())) return fail(s, PIES_ERR_INVALID, "pies_set_rest: range beyond the container");
      for (uint32_t i = 0; i < n; ++i) {
        std::memcpy(list[first + i].qinv, rest + 9ull * i, sizeof(float) * 9);
        tet_matrices(list[first + i]);
      }
      break;
    }
    case PIES_BEND:
      if (!in_range(s->h_bend.size())) return fail(s, PIES_ERR_INVALID, "pies_set_rest: range beyond the container");
      for (uint32_t i = 0; i < n; ++i) s->h_bend[first + i].angle = rest[i];
      break;
    default:
      return fail(s, PIES_ERR_INVALID, "pies_set_rest: container without rest data");
  }
  return PIES_OK;
}

int pies_add_triangles(pies_solver_t* s, uint32_t n, const uint32_t* ids) {
  PIES_BEGIN_EDIT(s);
  if (n && (!ids || !ids_ok(s, ids, 3ull * n))) return fail(s, PIES_ERR_INVALID, "triangle: bad node id");
  s->h_triangles.insert(s->h_triangles.end(), ids, ids + 3ull * n);
  return PIES_OK;
}

int pies_create_tet_box(pies_solver_t* s, uint32_t W, uint32_t H, uint32_t D, const float tr[3], float scale,
                        const float vel[3], float w, float mass, uint32_t flags) {
  PIES_BEGIN_EDIT(s);
  if (!tr || !vel || W == 0 || H == 0 || D == 0) return fail(s, PIES_ERR_INVALID, "pies_create_tet_box: bad argument");
  const uint32_t first = s->nodeCount();
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j)
      for (uint32_t k = 0; k < D; ++k)
        push_node(s, {scale * float(i) + tr[0], scale * float(j) + tr[1], scale * float(k) + tr[2]}, {vel[0], vel[1], vel[2]},
                  0.95f * 0.5f * scale, 1.0f / mass);
  auto G = [&](uint32_t x, uint32_t y, uint32_t z) { return lattice_id(H, D, first, x, y, z); };
  for (uint32_t i = 0; i + 1 < W; ++i)
    for (uint32_t j = 0; j + 1 < H; ++j)
      for (uint32_t k = 0; k + 1 < D; ++k) {
        const uint32_t n000 = G(i, j, k), n001 = G(i, j, k + 1), n010 = G(i, j + 1, k), n011 = G(i, j + 1, k + 1);
        const uint32_t n100 = G(i + 1, j, k), n101 = G(i + 1, j, k + 1), n110 = G(i + 1, j + 1, k), n111 = G(i + 1, j + 1, k + 1);
        // the six tetrahedra of the cell, all sharing the 000-111 diagonal (PrimitiveUtilities.cpp:401-514)
        const uint32_t q[6][4] = {{n000, n001, n011, n111}, {n000, n010, n011, n111}, {n000, n001, n101, n111},
                                  {n000, n100, n101, n111}, {n000, n010, n110, n111}, {n000, n100, n110, n111}};
        // Proposed colouring: tetrahedron e of a cell runs 000 -> e_a -> e_a+e_b -> 111, so two tetrahedra of
        // the same kind share a node only when their cells differ by e_a, e_b, e_c, e_a+e_b, e_b+e_c or 111,
        // none of which has i+j+k = 0 mod 4: 6 x 4 = 24 colours, the number of tetrahedra at an inner node.
        const uint16_t cellColour = static_cast<uint16_t>((i + j + k) & 3u);
        // Schedule LAYERED colours one layer of cells at a time: 12 tetrahedra of a layer meet at a node and 12 colours
        // suffice, periodic with period 2 in both in-layer directions (tables found by tools/layer_colour_tables.py).
        // kLayerTetColour[axis][2 * (u & 1) + (v & 1)][e]: u, v = cell indices along the two in-layer axes (ascending)
        static const uint8_t kLayerTetColour[3][4][6] = {
            {{6, 5, 11, 4, 10, 8}, {0, 4, 7, 5, 1, 3}, {8, 9, 7, 2, 1, 6}, {3, 2, 11, 9, 10, 0}},
            {{6, 5, 1, 4, 10, 0}, {8, 4, 7, 5, 11, 3}, {8, 9, 10, 2, 1, 3}, {6, 2, 11, 9, 7, 0}},
            {{3, 6, 1, 10, 8, 0}, {0, 2, 7, 5, 9, 3}, {11, 2, 8, 5, 1, 4}, {4, 6, 9, 10, 7, 11}},
        };
        const uint32_t uv[3] = {2 * (j & 1u) + (k & 1u), 2 * (i & 1u) + (k & 1u), 2 * (i & 1u) + (j & 1u)};
        for (int e = 0; e < 6; ++e) {
          s->h_tet.push_back(make_tet(s, q[e], w, 0.8f, 1.0f));
          s->h_tet.back().hint = static_cast<uint16_t>(4 * e + cellColour);
          for (int a = 0; a < 3; ++a) s->h_tet.back().layerHint[a] = kLayerTetColour[a][uv[a]][e];
          if (flags & 1u) s->h_volume.push_back(make_tet(s, q[e], w, 1.0f, 1.0f));
          s->constraintId += 2;
        }
      }
  if (flags & 2u) lattice_surface(s, W, H, D, first);
  return PIES_OK;
}

int pies_create_box(pies_solver_t* s, uint32_t W, uint32_t H, uint32_t D, const float tr[3], float scale, float w,
                    int existing, uint32_t existing_first, uint32_t flags) {
  PIES_BEGIN_EDIT(s);
  if (W == 0 || H == 0 || D == 0) return fail(s, PIES_ERR_INVALID, "pies_create_box: bad argument");
  uint32_t first = existing ? existing_first : s->nodeCount();
  if (existing) {
    if (static_cast<uint64_t>(first) + static_cast<uint64_t>(W) * H * D > s->nodeCount())
      return fail(s, PIES_ERR_INVALID, "pies_create_box: existing lattice out of range");
  } else {
    if (!tr) return fail(s, PIES_ERR_INVALID, "pies_create_box: translation is NULL");
    for (uint32_t i = 0; i < W; ++i)
      for (uint32_t j = 0; j < H; ++j)
        for (uint32_t k = 0; k < D; ++k)
          push_node(s, {scale * float(i) + tr[0], scale * float(j) + tr[1], scale * float(k) + tr[2]}, {0.f, 0.f, 0.f},
                    0.5f * scale, 1.0f);
  }
  auto G = [&](uint32_t x, uint32_t y, uint32_t z) { return lattice_id(H, D, first, x, y, z); };
  const size_t firstDistance = s->h_distance.size();
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j)
      for (uint32_t k = 0; k < D; ++k) {
        // Proposed colouring (9 colours; a node is moved by 7 of these constraints and read by 7 more):
        // colour = (x + 2y + 4z of the moved node + g[kind]) mod 9, found by exhaustive search over linear
        // forms so that neither two constraints moving one node nor a mover and a reader of it coincide.
        auto col = [&](uint32_t x, uint32_t y, uint32_t z, uint32_t kind) {
          static const uint32_t g[7] = {0, 1, 3, 5, 4, 2, 6};
          return static_cast<uint16_t>((x + 2u * y + 4u * z + g[kind]) % 9u);
        };
        if (i + 1 < W) push_distance(s, G(i, j, k), G(i + 1, j, k), w, col(i, j, k, 0));
        if (j + 1 < H) push_distance(s, G(i, j, k), G(i, j + 1, k), w, col(i, j, k, 1));
        if (k + 1 < D) push_distance(s, G(i, j, k), G(i, j, k + 1), w, col(i, j, k, 2));
        if (i + 1 < W && j + 1 < H && k + 1 < D) {  // the four body diagonals of the cell
          push_distance(s, G(i, j, k), G(i + 1, j + 1, k + 1), w, col(i, j, k, 3));
          push_distance(s, G(i + 1, j, k), G(i, j + 1, k + 1), w, col(i + 1, j, k, 4));
          push_distance(s, G(i, j + 1, k), G(i + 1, j, k + 1), w, col(i, j + 1, k, 5));
          push_distance(s, G(i, j, k + 1), G(i + 1, j + 1, k), w, col(i, j, k + 1, 6));
        }
      }
  if (flags & 2u) lattice_surface(s, W, H, D, first);
  lines_from_distances(s, firstDistance);
  return PIES_OK;
}

int pies_create_sheet(pies_solver_t* s, uint32_t W, uint32_t H, const float tr[3], float scale, float mass, float w) {
  PIES_BEGIN_EDIT(s);
  if (!tr || W == 0 || H == 0) return fail(s, PIES_ERR_INVALID, "pies_create_sheet: bad argument");
  const uint32_t first = s->nodeCount();
  auto G = [&](uint32_t x, uint32_t y) { return lattice_id(H, 1, first, x, y, 0); };
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j) {
      const uint32_t id = push_node(s, {scale * float(i) + tr[0], scale * 0.0f + tr[1], scale * float(j) + tr[2]}, {0.f, 0.f, 0.f},
                                    0.5f * scale, 1.0f / mass);
      if (i == 0 || i == W - 1 || j == 0 || j == H - 1) push_position(s, id, w);  // pinned border
    }
  const size_t firstDistance = s->h_distance.size();
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j) {
      if (i + 1 < W) push_distance(s, G(i, j), G(i + 1, j), w);
      if (j + 1 < H) push_distance(s, G(i, j), G(i, j + 1), w);
      if (i + 1 < W && j + 1 < H) {
        push_distance(s, G(i, j), G(i + 1, j + 1), w);
        push_distance(s, G(i + 1, j), G(i, j + 1), w);
      }
    }
  for (uint32_t i = 0; i + 1 < W; ++i)
    for (uint32_t j = 0; j + 1 < H; ++j) {
      s->h_triangles.insert(s->h_triangles.end(), {G(i, j), G(i + 1, j + 1), G(i + 1, j)});
      s->h_triangles.insert(s->h_triangles.end(), {G(i, j), G(i, j + 1), G(i + 1, j + 1)});
    }
  lines_from_distances(s, firstDistance);
  return PIES_OK;
}

int pies_create_bend_sheet(pies_solver_t* s, uint32_t W, uint32_t H, const float tr[3], float scale, float w) {
  PIES_BEGIN_EDIT(s);
  if (!tr || W == 0 || H == 0) return fail(s, PIES_ERR_INVALID, "pies_create_bend_sheet: bad argument");
  const uint32_t first = s->nodeCount();
  auto G = [&](uint32_t x, uint32_t y) { return lattice_id(H, 1, first, x, y, 0); };
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j) {
      const uint32_t id = push_node(s, {scale * float(i) + tr[0], scale * 0.0f + tr[1], scale * float(j) + tr[2]}, {0.f, 0.f, 0.f},
                                    0.5f * scale, 1.0f);
      if (i < 3) push_position(s, id, w);  // clamped strip
    }
  const size_t firstDistance = s->h_distance.size();
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j) {
      if (i + 1 < W) push_distance(s, G(i, j), G(i + 1, j), w);
      if (j + 1 < H) push_distance(s, G(i, j), G(i, j + 1), w);
      if (i + 1 < W && j + 1 < H) push_distance(s, G(i, j), G(i + 1, j + 1), w);
    }
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j) {
      if (i + 1 < W && j + 1 < H) {
        const uint32_t q[4] = {G(i, j), G(i + 1, j + 1), G(i + 1, j), G(i, j + 1)};
        push_bend(s, q, w);
      }
      if (i + 2 < W && j + 2 < H) {
        const uint32_t qa[4] = {G(i + 1, j), G(i + 1, j + 1), G(i, j), G(i + 2, j + 1)};
        const uint32_t qb[4] = {G(i, j + 1), G(i + 1, j + 1), G(i, j), G(i + 1, j + 2)};
        push_bend(s, qa, w);
        push_bend(s, qb, w);
      }
    }
  for (uint32_t i = 0; i + 1 < W; ++i)
    for (uint32_t j = 0; j + 1 < H; ++j) {
      s->h_triangles.insert(s->h_triangles.end(), {G(i, j), G(i + 1, j + 1), G(i + 1, j)});
      s->h_triangles.insert(s->h_triangles.end(), {G(i, j), G(i, j + 1), G(i + 1, j + 1)});
    }
  lines_from_distances(s, firstDistance);
  return PIES_OK;
}

// ShapeMatchingConstraint over explicit nodes: material coordinates = the nodes' current positions
// (what createShapeMatching* and addLinkedRegions pass, PrimitiveUtilities.cpp:149-156,1027-1036)
int pies_add_shape_constraint(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w) {
  PIES_BEGIN_EDIT(s);
  if (n == 0 || !ids || !ids_ok(s, ids, n)) return fail(s, PIES_ERR_INVALID, "shape constraint: bad node id");
  std::vector<uint32_t> v(ids, ids + n);
  std::vector<P3> mc(n);
  for (uint32_t i = 0; i < n; ++i) mc[i] = node_pos(s, ids[i]);
  push_shape(s, v, mc, w);
  return PIES_OK;
}

int pies_add_goal_constraint(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w, uint32_t* goal_index) {
  PIES_BEGIN_EDIT(s);
  if (n && (!ids || !ids_ok(s, ids, n))) return fail(s, PIES_ERR_INVALID, "goal constraint: bad node id");
  if (goal_index) *goal_index = static_cast<uint32_t>(s->h_goal.size());
  push_goal(s, std::vector<uint32_t>(ids, ids + n), w);
  return PIES_OK;
}

// GoalMatchingConstraint::setTransform (ShapeMatchingConstraint.cpp:175-177); no re-capture needed
int pies_set_goal_transform(pies_solver_t* s, uint32_t goal, const float m16[16]) {
  PIES_CHECK_HANDLE(s);
  if (!m16 || goal >= s->h_goal.size()) return fail(s, PIES_ERR_INVALID, "pies_set_goal_transform: bad goal index");
  std::memcpy(s->h_goal[goal].transform, m16, 16 * sizeof(float));
  s->goalDirty = true;
  return PIES_OK;
}

// Solver::addFixedRegions (PrimitiveUtilities.cpp:77-112): one goal constraint per region over the nodes
// inside its unit box.
int pies_add_fixed_regions(pies_solver_t* s, uint32_t n, const float* mats16, float w) {
  PIES_BEGIN_EDIT(s);
  if (n && !mats16) return fail(s, PIES_ERR_INVALID, "pies_add_fixed_regions: NULL matrices");
  for (uint32_t k = 0; k < n; ++k) {
    HostFixedRegion region;
    inverse_mat4(mats16 + 16 * k, region.invInitialTransform);
    region.goal = static_cast<uint32_t>(s->h_goal.size());
    std::vector<uint32_t> inside;
    for (uint32_t i = 0; i < s->nodeCount(); ++i)
      if (inside_unit_box(region.invInitialTransform, node_pos(s, i))) inside.push_back(i);
    push_goal(s, inside, w);
    s->h_fixedRegions.push_back(region);
  }
  return PIES_OK;
}

// Solver::updateFixedRegions (PrimitiveUtilities.cpp:114-128): transform = current * inverse(initial)
int pies_update_fixed_regions(pies_solver_t* s, uint32_t n, const float* mats16) {
  PIES_CHECK_HANDLE(s);
  if (n != s->h_fixedRegions.size() || (n && !mats16)) return fail(s, PIES_ERR_INVALID, "pies_update_fixed_regions: region count mismatch");
  for (uint32_t k = 0; k < n; ++k) {
    float t[16];
    mul_mat4(mats16 + 16 * k, s->h_fixedRegions[k].invInitialTransform, t);
    std::memcpy(s->h_goal[s->h_fixedRegions[k].goal].transform, t, sizeof(t));
  }
  s->goalDirty = true;
  return PIES_OK;
}

// Solver::addLinkedRegions (PrimitiveUtilities.cpp:130-162): one shape-matching constraint per region with >= 3 nodes
int pies_add_linked_regions(pies_solver_t* s, uint32_t n, const float* mats16, float w) {
  PIES_BEGIN_EDIT(s);
  if (n && !mats16) return fail(s, PIES_ERR_INVALID, "pies_add_linked_regions: NULL matrices");
  for (uint32_t k = 0; k < n; ++k) {
    float inv[16];
    inverse_mat4(mats16 + 16 * k, inv);
    std::vector<uint32_t> ids;
    std::vector<P3> mc;
    for (uint32_t i = 0; i < s->nodeCount(); ++i)
      if (inside_unit_box(inv, node_pos(s, i))) { ids.push_back(i); mc.push_back(node_pos(s, i)); }
    if (mc.size() >= 3) push_shape(s, ids, mc, w);
  }
  return PIES_OK;
}

// Solver::createShapeMatchingBox (PrimitiveUtilities.cpp:985-1048): the scale argument is overridden to 0.5,
// invMass 1/10, zero velocity (the reference ignores initialVelocity), one constraint over the whole lattice.
int pies_create_shape_matching_box(pies_solver_t* s, const float tr[3], uint32_t countX, uint32_t countY, uint32_t countZ, float w) {
  PIES_BEGIN_EDIT(s);
  if (!tr || countX == 0 || countY == 0 || countZ == 0) return fail(s, PIES_ERR_INVALID, "pies_create_shape_matching_box: bad argument");
  const float scale = 0.5f;
  const uint32_t first = s->nodeCount();
  for (uint32_t i = 0; i < countX; ++i)
    for (uint32_t j = 0; j < countY; ++j)
      for (uint32_t k = 0; k < countZ; ++k)
        push_node(s, {scale * float(i) + tr[0], scale * float(j) + tr[1], scale * float(k) + tr[2]}, {0.f, 0.f, 0.f}, 0.5f * scale, 1.0f / 10.0f);
  const uint32_t n = countX * countY * countZ;
  std::vector<uint32_t> ids(n);
  std::vector<P3> mc(n);
  for (uint32_t i = 0; i < n; ++i) { ids[i] = first + i; mc[i] = node_pos(s, first + i); }
  push_shape(s, ids, mc, w);
  return PIES_OK;
}

// Solver::createShapeMatchingSheet (PrimitiveUtilities.cpp:1050-1125): W x H x 1 lattice in the xy plane,
// overlapping 3x3 patches, one shape-matching constraint per patch (reference: 50 x 50).
int pies_create_shape_matching_sheet(pies_solver_t* s, uint32_t W, uint32_t H, const float tr[3], float scale, float w) {
  PIES_BEGIN_EDIT(s);
  if (!tr || W < 3 || H < 3) return fail(s, PIES_ERR_INVALID, "pies_create_shape_matching_sheet: bad argument");
  constexpr uint32_t pw = 3, ph = 3;
  struct Patch { std::vector<uint32_t> ids; std::vector<P3> mc; };
  std::vector<Patch> patches((W / pw) * (H / ph));
  auto add = [&](uint32_t patch, uint32_t id, P3 p) {
    if (patch < patches.size()) { patches[patch].ids.push_back(id); patches[patch].mc.push_back(p); }
  };
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j) {
      const P3 p{scale * float(i) + tr[0], scale * float(j) + tr[1], scale * 0.0f + tr[2]};
      const uint32_t id = push_node(s, p, {0.f, 0.f, 0.f}, 0.5f * scale, 1.0f);
      add(i / pw * ph + j / ph, id, p);
      if ((i % pw) == (pw - 1) && i < (W - 1)) add((1 + i / pw) * ph + j / ph, id, p);
      if ((j % ph) == (ph - 1) && j < (H - 1)) add(i / pw * ph + j / ph + 1, id, p);
    }
  for (const Patch& pt : patches)
    if (!pt.ids.empty()) push_shape(s, pt.ids, pt.mc, w);
  return PIES_OK;
}

}  // extern "C"
