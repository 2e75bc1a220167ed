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
HostTet make_tet(const pies_solver* s, const uint32_t ids[4], float w, float lo, float hi) {
  HostTet t{};
  std::memcpy(t.ids, ids, sizeof(t.ids));
  const P3 x1 = node_pos(s, ids[0]);
  inverse_columns(node_pos(s, ids[1]) - x1, node_pos(s, ids[2]) - x1, node_pos(s, ids[3]) - x1, t.qinv);
  t.lo = lo;
  t.hi = hi;
  t.w = w;
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
  return t;
}

void push_distance(pies_solver* s, uint32_t a, uint32_t b, float w) {
  HostDistance d;
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
        for (int e = 0; e < 6; ++e) {
          s->h_tet.push_back(make_tet(s, q[e], w, 0.8f, 1.0f));
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
        if (i + 1 < W) push_distance(s, G(i, j, k), G(i + 1, j, k), w);
        if (j + 1 < H) push_distance(s, G(i, j, k), G(i, j + 1, k), w);
        if (k + 1 < D) push_distance(s, G(i, j, k), G(i, j, k + 1), w);
        if (i + 1 < W && j + 1 < H && k + 1 < D) {  // the four body diagonals of the cell
          push_distance(s, G(i, j, k), G(i + 1, j + 1, k + 1), w);
          push_distance(s, G(i + 1, j, k), G(i, j + 1, k + 1), w);
          push_distance(s, G(i, j + 1, k), G(i + 1, j, k + 1), w);
          push_distance(s, G(i, j, k + 1), G(i + 1, j + 1, k), w);
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

}  // extern "C"
