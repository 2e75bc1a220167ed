// TEST INFRASTRUCTURE ONLY.  CPU oracle: a plain, single-threaded restatement of the reference
// solver loop of nithinp7/Pies, used only by tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg as the *checker*.  The product (pies_amd/, include/) never links, imports or
// executes anything in this directory.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path, and it
// cannot be built here (glm, Eigen, parallel-hashmap, tetgen are empty un-vendored submodules,
// /root/reference/.gitmodules:1-12).  This restatement is pinned instead by analytic known-answer
// tests and fp64 numpy/scipy golden vectors (tests/golden/, generator committed beside them).
//
// Every function cites the reference lines it follows (paths relative to /root/reference).
// Build: oracle/Makefile  ->  oracle/_build/libpies_oracle.so  (g++ -O2 -ffp-contract=off).
#include "ora_math.h"

#include <algorithm>
#include <array>
#include <cassert>
#include <xmmintrin.h>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

using namespace ora;

namespace {

// Include/Pies/Solver.h:23-38 (same fields, order and defaults; `solver`: 0 = PBD, 1 = PD)
struct Options {
  float fixedTimestepSize = 0.012f;
  uint32_t timeSubsteps = 1;
  uint32_t iterations = 4;
  uint32_t collisionStabilizationIterations = 4;
  float collisionThresholdDistance = 0.1f;
  float collisionThickness = 0.05f;
  float gravity = 10.0f;
  float damping = 0.006f;
  float friction = 0.01f;
  float staticFrictionThreshold = 0.f;
  float floorHeight = 0.0f;
  float gridSpacing = 2.0f;
  uint32_t threadCount = 8;
  int32_t solver = 1;
};

// Include/Pies/Node.h:8-20
struct Node {
  uint32_t id = 0;
  vec3 position, prevPosition, velocity, force;
  float radius = 0.1f;
  float invMass = 1.0f;
};

// ---------------------------------------------------------------------------------------------
// Include/Pies/Constraints.h:20-145 -- generic constraint: w, AtA, AtB, node ids, aux variable.
// ---------------------------------------------------------------------------------------------
template <int N> struct ConBase {
  uint32_t id = 0;
  float w = 1.0f;
  float AtA[N][N];
  float AtB[N][N];
  std::array<uint32_t, N> nodeIds;
  std::array<vec3, N> projected;

  // Constraints.h:49-61 : _AtA = A^T A, _AtB = A^T B
  void init(uint32_t id_, float w_, const float A[N][N], const float B[N][N],
            const std::array<uint32_t, N>& ids) {
    id = id_;
    w = w_;
    nodeIds = ids;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        float sa = 0.f, sb = 0.f;
        for (int k = 0; k < N; ++k) {
          sa += A[k][i] * A[k][j];
          sb += A[k][i] * B[k][j];
        }
        AtA[i][j] = sa;
        AtB[i][j] = sb;
      }
    for (auto& p : projected) p = vec3(0.f);
  }
};

// Constraints.cpp:11-37
struct DistanceCon : ConBase<2> {
  float targetDistance = 0.f;
  void project(const std::vector<Node>& nodes, std::array<vec3, 2>& out) const {
    const Node& a = nodes[nodeIds[0]];
    const Node& b = nodes[nodeIds[1]];
    out[0] = a.position;
    out[1] = b.position;
    vec3 diff = b.position - a.position;
    float dist = length(diff);
    vec3 dir(1.0f, 0.0f, 0.0f);
    if (dist > 0.00001f) dir = diff / dist;
    float disp = targetDistance - dist;
    out[0] += -disp * dir;  // only node a moves, by the full correction (Constraints.cpp:34)
  }
};

// Constraints.cpp:58-63
struct PositionCon : ConBase<1> {
  vec3 fixedPosition;
  void project(const std::vector<Node>&, std::array<vec3, 1>& out) const { out[0] = fixedPosition; }
};

// Shared by Constraints.cpp:85-99 and :214-227: P = [x2-x1, x3-x1, x4-x1] (columns), F = P*Qinv,
// Eigen F_ filled row-major from glm column-major => F_(r,c) = F[r][c].
static inline void tet_deformation(const std::vector<Node>& nodes, const std::array<uint32_t, 4>& ids,
                                   const mat3& Qinv, mat3& F, float F_[3][3]) {
  const Node& x1 = nodes[ids[0]];
  const Node& x2 = nodes[ids[1]];
  const Node& x3 = nodes[ids[2]];
  const Node& x4 = nodes[ids[3]];
  mat3 P(x2.position - x1.position, x3.position - x1.position, x4.position - x1.position);
  F = P * Qinv;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) F_[r][c] = F[r][c];
}

// Constraints.cpp:111-127 / :238-254 : Fhat = U*S'*V^T ; P1 = transpose(mat3 from Fhat columns);
// projected = (0, P1[0], P1[1], P1[2]).  mat3(Fhat(0,0),Fhat(1,0),Fhat(2,0), ...) has glm column c
// = Eigen column c of Fhat, so P1[c][r] = Fhat(c, r).
static inline void tet_emit(const float Fhat[3][3], std::array<vec3, 4>& out) {
  out[0] = vec3(0.0f);
  for (int c = 0; c < 3; ++c) out[1 + c] = vec3(Fhat[c][0], Fhat[c][1], Fhat[c][2]);
}

// Statistics of the decompositions the strain projection ran (what a workload asks of svd3: ora_svd_stats).  Not thread safe:
// read after single-threaded ticks only.
static bool g_svd_plain = false;  // ora_set_flag(*, 5, 1): the plain iteration from V = I (a process-wide replay switch: statistics, and the test that
                                  // both decompositions give the same projections)
static uint64_t g_svd_stats[8];  // calls, closed-form starts, rotations, certifying sweeps, sweeps histogram (1, 2, 3, >= 4)
static inline void svd_stats_add(const Svd3& d) {
  g_svd_stats[0] += 1;
  g_svd_stats[1] += d.closed_form ? 1 : 0;
  g_svd_stats[2] += static_cast<uint64_t>(d.rotations);
  const int sw = d.sweeps + 1;  // sweeps run, the clean one included (0 + 1 for an element that needed none: see svd3)
  g_svd_stats[3] += static_cast<uint64_t>(sw);
  g_svd_stats[4 + (sw >= 4 ? 3 : sw - 1)] += 1;
}

// Constraints.cpp:76-128
struct TetCon : ConBase<4> {
  mat3 Qinv;
  float minStrain = 0.8f, maxStrain = 1.0f;
  void project(const std::vector<Node>& nodes, std::array<vec3, 4>& out) const {
    mat3 F;
    float F_[3][3];
    tet_deformation(nodes, nodeIds, Qinv, F, F_);
    Svd3 d = g_svd_plain ? svd3_jacobi(F_) : svd3(F_);
    svd_stats_add(d);
    float s[3];
    for (int i = 0; i < 3; ++i) s[i] = clampf(d.s[i], minStrain, maxStrain);
    if (determinant(F) < 0.0f) {
      // Eigen sorts singular values descending, so singularValues[2] is the smallest one
      // (Constraints.cpp:106-108); the flip is applied after clamping.
      int k = 0;
      if (d.s[1] <= d.s[k]) k = 1;
      if (d.s[2] <= d.s[k]) k = 2;
      s[k] *= -1.0f;
    }
    float Fhat[3][3];
    svd3_recompose(d, s, Fhat);
    tet_emit(Fhat, out);
  }
};

// Constraints.cpp:186-203
static vec3 computeD(const vec3& sigma, float omegaMin, float omegaMax) {
  const uint32_t COMP_D_ITERS = 10;
  vec3 D(0.0f);
  for (uint32_t i = 0; i < COMP_D_ITERS; ++i) {
    vec3 sp = sigma + D;
    float product = sp.x * sp.y * sp.z;
    float omega = clampf(product, omegaMin, omegaMax);
    float C = product - omega;
    vec3 gradC(sp.y * sp.z, sp.x * sp.z, sp.x * sp.y);
    D = (dot(gradC, D) - C) * gradC / dot(gradC, gradC);
  }
  return D;
}

// Constraints.cpp:205-255
struct VolumeCon : ConBase<4> {
  mat3 Qinv;
  float minOmega = 1.0f, maxOmega = 1.0f;
  void project(const std::vector<Node>& nodes, std::array<vec3, 4>& out) const {
    mat3 F;
    float F_[3][3];
    tet_deformation(nodes, nodeIds, Qinv, F, F_);
    Svd3 d = svd3(F_);
    vec3 D = computeD(vec3(d.s[0], d.s[1], d.s[2]), minOmega, maxOmega);
    float s[3] = {d.s[0] + D.x, d.s[1] + D.y, d.s[2] + D.z};
    float Fhat[3][3];
    svd3_recompose(d, s, Fhat);
    tet_emit(Fhat, out);
  }
};

// The reference calls acos on a float (Constraints.cpp:339, 385).  libm acosf implementations differ by
// an ulp between platforms; evaluating in double and rounding once gives the correctly rounded float
// result (up to double-rounding ties), which any conforming acosf is within 1 ulp of, and which a
// device can reproduce exactly.
static inline float acos_f(float d) { return static_cast<float>(std::acos(static_cast<double>(d))); }

// Constraints.cpp:312-366
struct BendCon : ConBase<4> {
  float initialAngle = 0.f;
  void project(const std::vector<Node>& nodes, std::array<vec3, 4>& out) const {
    const Node& x1 = nodes[nodeIds[0]];
    const Node& x2 = nodes[nodeIds[1]];
    const Node& x3 = nodes[nodeIds[2]];
    const Node& x4 = nodes[nodeIds[3]];
    vec3 p2 = x2.position - x1.position;
    vec3 p3 = x3.position - x1.position;
    vec3 p4 = x4.position - x1.position;
    vec3 p2Xp3 = cross(p2, p3);
    vec3 p2Xp4 = cross(p2, p4);
    float l23 = length(p2Xp3);
    float l24 = length(p2Xp4);
    vec3 n1 = p2Xp3 / l23;
    vec3 n2 = p2Xp4 / l24;
    float d = dot(n1, n2);
    float d2 = d * d;
    float C = acos_f(d) - initialAngle;
    out[0] = x1.position;
    out[1] = x2.position;
    out[2] = x3.position;
    out[3] = x4.position;
    vec3 q3 = (cross(p2, n2) + (cross(n1, p2) * d)) / l23;
    vec3 q4 = (cross(p2, n1) + (cross(n2, p2) * d)) / l24;
    vec3 q2 = -((cross(p3, n2) + (cross(n1, p3) * d)) / l23) - ((cross(p4, n1) + (cross(n2, p4) * d)) / l24);
    vec3 q1 = -q2 - q3 - q4;
    float wSum = x1.invMass + x2.invMass + x3.invMass + x4.invMass;
    float qSq = dot(q1, q1) + dot(q2, q2) + dot(q3, q3) + dot(q4, q4);
    float num = std::sqrt(std::fmax(1.0f - d2, 0.0f)) * C;
    if (qSq < 0.00001f) return;
    out[0] += -q1 * (4 * x1.invMass / wSum) * (num) / qSq;
    out[1] += -q2 * (4 * x2.invMass / wSum) * (num) / qSq;
    out[2] += -q3 * (4 * x3.invMass / wSum) * (num) / qSq;
    out[3] += -q4 * (4 * x4.invMass / wSum) * (num) / qSq;
  }
};

// ---------------------------------------------------------------------------------------------
// Src/ShapeMatchingConstraint.cpp:6-122  (double precision where the reference is)
// ---------------------------------------------------------------------------------------------
struct Quatd {
  double w = 1, x = 0, y = 0, z = 0;
};
static Quatd qmul(const Quatd& a, const Quatd& b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
static void qmat(const Quatd& q, double R[3][3]) {
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz;       R[0][2] = txz + twy;
  R[1][0] = txy + twz;       R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
  R[2][0] = txz - twy;       R[2][1] = tyz + twx;       R[2][2] = 1 - (txx + tyy);
}
static void inv3d(const double m[3][3], double o[3][3]) {
  double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
  double c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2];
  double c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
  double det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02;
  double id = 1.0 / det;
  o[0][0] = c00 * id;
  o[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
  o[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
  o[1][0] = c01 * id;
  o[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id;
  o[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
  o[2][0] = c02 * id;
  o[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
  o[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
}

// ShapeMatchingConstraint.cpp:75-94 (Mueller et al., rotation extraction).  Note the 1e-9 is added
// to the reciprocal, not to the denominator (:84-87).
static void extractRotation(const double A[3][3], Quatd& q, unsigned maxIter) {
  for (unsigned iter = 0; iter < maxIter; ++iter) {
    double R[3][3];
    qmat(q, R);
    double om[3] = {0, 0, 0};
    double dsum = 0;
    for (int c = 0; c < 3; ++c) {
      double r[3] = {R[0][c], R[1][c], R[2][c]};
      double a[3] = {A[0][c], A[1][c], A[2][c]};
      om[0] += r[1] * a[2] - r[2] * a[1];
      om[1] += r[2] * a[0] - r[0] * a[2];
      om[2] += r[0] * a[1] - r[1] * a[0];
      dsum += r[0] * a[0] + r[1] * a[1] + r[2] * a[2];
    }
    double f = 1.0 / std::fabs(dsum) + 1.0e-9;
    om[0] *= f; om[1] *= f; om[2] *= f;
    double w = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    if (w < 1.0e-9) break;
    double ax[3] = {(1.0 / w) * om[0], (1.0 / w) * om[1], (1.0 / w) * om[2]};
    double ha = 0.5 * w;
    double sh = std::sin(ha);
    Quatd dq{std::cos(ha), sh * ax[0], sh * ax[1], sh * ax[2]};
    q = qmul(dq, q);
    double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    q.w /= n; q.x /= n; q.y /= n; q.z /= n;
  }
}

struct ShapeCon {
  std::vector<uint32_t> ids;
  std::vector<double> mat;   // 3 x n, column i at mat[3*i..]
  std::vector<double> proj;  // 3 x n
  double Qinv[3][3];
  Quatd rot;
  float w = 1.f;

  // ShapeMatchingConstraint.cpp:6-48
  void init(const std::vector<Node>& nodes, const std::vector<uint32_t>& indices,
            const std::vector<vec3>& materialCoordinates, float w_) {
    ids = indices;
    w = w_;
    size_t n = materialCoordinates.size();
    mat.assign(3 * n, 0.0);
    proj.assign(3 * n, 0.0);
    vec3 com(0.0f);
    float weight = 1.0f / static_cast<float>(n);
    for (const vec3& c : materialCoordinates) com += weight * c;
    double Q[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (size_t i = 0; i < n; ++i) {
      vec3 mc = materialCoordinates[i] - com;
      mat[3 * i + 0] = mc.x; mat[3 * i + 1] = mc.y; mat[3 * i + 2] = mc.z;
      // glm::outerProduct(c, r)[col][row] = c[row]*r[col]; here c == r == mc; then / invMass
      float im = nodes[indices[i]].invMass;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Q[r][c] += (mc[r] * mc[c]) / im;
    }
    inv3d(Q, Qinv);
  }

  // ShapeMatchingConstraint.cpp:96-122
  void project(const std::vector<Node>& nodes) {
    vec3 com(0.0f);
    float weight = 1.0f / static_cast<float>(ids.size());
    for (uint32_t id : ids) com += weight * nodes[id].position;
    double P[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (size_t i = 0; i < ids.size(); ++i) {
      const Node& node = nodes[ids[i]];
      vec3 lc = node.position - com;
      double l[3] = {lc.x, lc.y, lc.z};
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) P[r][c] += l[r] * mat[3 * i + c] / node.invMass;
    }
    double F[3][3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) F[r][c] = P[r][0] * Qinv[0][c] + P[r][1] * Qinv[1][c] + P[r][2] * Qinv[2][c];
    extractRotation(F, rot, 100);
    double R[3][3];
    qmat(rot, R);
    double T[3] = {com.x, com.y, com.z};
    for (size_t i = 0; i < ids.size(); ++i)
      for (int r = 0; r < 3; ++r)
        proj[3 * i + r] = (R[r][0] * mat[3 * i] + R[r][1] * mat[3 * i + 1] + R[r][2] * mat[3 * i + 2]) + T[r];
  }
};

// ShapeMatchingConstraint.cpp:124-177
struct GoalCon {
  std::vector<uint32_t> ids;
  std::vector<vec3> mat;
  std::vector<double> proj;
  mat4 transform;
  float w = 1.f;
  void init(const std::vector<Node>& nodes, const std::vector<uint32_t>& indices, float w_) {
    ids = indices;
    w = w_;
    mat.resize(ids.size());
    proj.assign(3 * ids.size(), 0.0);
    for (size_t i = 0; i < ids.size(); ++i) mat[i] = nodes[ids[i]].position;
    for (int i = 0; i < 16; ++i) transform.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  }
  void project() {
    for (size_t i = 0; i < mat.size(); ++i) {
      float o[4];
      mul_point(transform, mat[i], o);
      proj[3 * i + 0] = o[0]; proj[3 * i + 1] = o[1]; proj[3 * i + 2] = o[2];
    }
  }
};

// Src/CollisionConstraint.cpp:7-65 (node-node, PD); w = 1e5 (Include/Pies/CollisionConstraint.h:14).  The reference never
// creates one: its only source, Solver::_parallelComputeCollisions (Solver.cpp:509-637), is never called, and tickPD calls none of
// the three methods - only the friction loop (Solver.cpp:398-428) walks the always-empty list.  Here the constraint is an EXTENSION
// container (ora_add_node_pairs, default empty) wired the way the reference wires its live collision constraints (triangle,
// static): projection in the local step, w on both diagonal entries of the per-substep collision matrix, w * projected into the
// right-hand side, and the friction loop after the velocity update.
struct NodePairCollision {
  float w = 100000.0f;
  uint32_t nodeIds[2] = {0, 0};
  vec3 projectedPositions[2];
  // CollisionConstraint.cpp:10-41
  void project(const std::vector<Node>& nodes) {
    const Node& nodeA = nodes[nodeIds[0]];
    const Node& nodeB = nodes[nodeIds[1]];
    projectedPositions[0] = nodeA.position;
    projectedPositions[1] = nodeB.position;
    vec3 diff = nodeB.position - nodeA.position;
    float distSq = dot(diff, diff);
    float r = nodeA.radius + nodeB.radius;
    float rSq = r * r;
    if (distSq >= rSq) return;
    float dist = std::sqrt(distSq);
    float dispLength = r - dist;
    vec3 disp;
    if (dist > 0.00001f) disp = dispLength * diff / dist;
    else disp = vec3(dispLength, 0.0f, 0.0f);
    float wSum = nodeA.invMass + nodeB.invMass;
    projectedPositions[0] -= disp * nodeA.invMass / wSum;
    projectedPositions[1] += disp * nodeB.invMass / wSum;
  }
};

// Src/CollisionConstraint.cpp:439-463 (floor); w = 1e4 (Include/Pies/CollisionConstraint.h:78)
struct StaticCollision {
  float w = 10000.0f;
  uint32_t nodeId = 0;
  vec3 projectedPosition;
};

// Src/CollisionConstraint.cpp:67-194; w = 1e4, AtA from the differential-coordinate A (:74-83)
struct TriCollision {
  float w = 10000.0f;
  uint32_t nodeIds[4];
  vec3 projectedPositions[4];
  float AtA[4][4];
  float thickness = 0.01f;
  bool colliding = false;
};

// ---------------------------------------------------------------------------------------------
// Include/Pies/SpatialHash.h:61-199.  The phmap container only affects iteration order of *cells*,
// which the solver never iterates; bucket order is ascending value index (each of the 16 insert
// threads scans values in order and owns disjoint submaps, SpatialHash.h:141-176).
// ---------------------------------------------------------------------------------------------
struct CellId {
  int64_t x, y, z;
  bool operator==(const CellId& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct CellHash {
  size_t operator()(const CellId& id) const noexcept {
    int64_t h = (id.x * 92837111) ^ (id.y * 689287499) ^ (id.z * 283923481);  // SpatialHash.h:30
    return static_cast<size_t>(std::abs(h));
  }
};
struct CellRange {
  int64_t minX = 0, minY = 0, minZ = 0;
  uint32_t lengthX = 0, lengthY = 0, lengthZ = 0;
};

// Src/Solver.cpp:877-901
static CellRange nodeCompRange(const Node& node, float scale) {
  float radiusPadding = 0.5f;
  float gridLocalRadius = (node.radius + radiusPadding) / scale;
  vec3 gridLocalPos = node.position / scale;
  vec3 gridLocalMin = gridLocalPos - vec3(gridLocalRadius);
  CellRange range{};
  range.minX = static_cast<int64_t>(std::floor(gridLocalMin.x));
  range.minY = static_cast<int64_t>(std::floor(gridLocalMin.y));
  range.minZ = static_cast<int64_t>(std::floor(gridLocalMin.z));
  float twoR = 2 * gridLocalRadius;
  range.lengthX = static_cast<uint32_t>(std::ceil(fractf(gridLocalMin.x) + twoR));
  range.lengthY = static_cast<uint32_t>(std::ceil(fractf(gridLocalMin.y) + twoR));
  range.lengthZ = static_cast<uint32_t>(std::ceil(fractf(gridLocalMin.z) + twoR));
  if (range.lengthX > 50 || range.lengthY > 50 || range.lengthZ > 50) return {};
  return range;
}

struct NodeHash {
  // 16 sub-maps like phmap's parallel_flat_hash_map<..., 4> the reference relies on (SpatialHash.h:134-176): a cell lives
  // in sub-map hash % 16.  Single-threaded insertion fills them in one scan; the reference's scheme (`threads` = 16) has
  // every thread scan ALL nodes and keep the cells of the sub-map it owns - sixteen times the range arithmetic, no locks.
  // Either way a bucket lists its nodes in ascending index.
  static constexpr uint32_t kSub = 16;
  std::unordered_map<CellId, std::vector<uint32_t>, CellHash> sub[kSub];
  int threads = 1;
  static uint32_t owner(const CellId& c) { return static_cast<uint32_t>(CellHash()(c) % kSub); }
  void clear() { for (auto& m : sub) m.clear(); }
  // SpatialHash.h:129-189
  void bulkInsert(const std::vector<Node>& nodes, float scale) {
    if (threads <= 1) {
      for (uint32_t i = 0; i < nodes.size(); ++i) {
        CellRange r = nodeCompRange(nodes[i], scale);
        for (uint32_t dx = 0; dx < r.lengthX; ++dx)
          for (uint32_t dy = 0; dy < r.lengthY; ++dy)
            for (uint32_t dz = 0; dz < r.lengthZ; ++dz) {
              CellId c{r.minX + dx, r.minY + dy, r.minZ + dz};
              sub[owner(c)][c].push_back(i);
            }
      }
      return;
    }
#pragma omp parallel for num_threads(kSub) schedule(static, 1)
    for (int t = 0; t < static_cast<int>(kSub); ++t) {  // thread t owns sub-map t (SpatialHash.h:141-176)
      for (uint32_t i = 0; i < nodes.size(); ++i) {
        CellRange r = nodeCompRange(nodes[i], scale);
        for (uint32_t dx = 0; dx < r.lengthX; ++dx)
          for (uint32_t dy = 0; dy < r.lengthY; ++dy)
            for (uint32_t dz = 0; dz < r.lengthZ; ++dz) {
              CellId c{r.minX + dx, r.minY + dy, r.minZ + dz};
              if (owner(c) == static_cast<uint32_t>(t)) sub[t][c].push_back(i);
            }
      }
    }
  }
  const std::vector<uint32_t>* find(const CellId& c) const {
    const auto& m = sub[owner(c)];
    auto it = m.find(c);
    return it != m.end() ? &it->second : nullptr;
  }
  void findCollisions(const CellRange& r, std::vector<const std::vector<uint32_t>*>& out) const {
    for (uint32_t dx = 0; dx < r.lengthX; ++dx)
      for (uint32_t dy = 0; dy < r.lengthY; ++dy)
        for (uint32_t dz = 0; dz < r.lengthZ; ++dz)
          if (const auto* b = find(CellId{r.minX + dx, r.minY + dy, r.minZ + dz})) out.push_back(b);
  }
  // SpatialHash.h:101-127
  void findCollisions(const Node& node, float scale, std::vector<const std::vector<uint32_t>*>& out) const {
    findCollisions(nodeCompRange(node, scale), out);
  }
};

// ---------------------------------------------------------------------------------------------
// Src/CollisionDetection.cpp:14-302 -- point-triangle continuous collision detection.
// ---------------------------------------------------------------------------------------------
struct CubicExpression {
  float cubicCoeff = 0.f, quadCoeff = 0.f, linearCoeff = 0.f, constCoeff = 0.f;
  float eval(float t) const {  // CollisionDetection.cpp:20-23
    float t2 = t * t;
    return cubicCoeff * t2 * t + quadCoeff * t2 + linearCoeff * t + constCoeff;
  }
};

// CollisionDetection.cpp:209-221
static inline void expandTerm(float a0, float b0, float c0, float ad, float bd, float cd, CubicExpression& e) {
  e.cubicCoeff += ad * bd * cd;
  e.quadCoeff += ad * bd * c0 + a0 * bd * cd + ad * b0 * cd;
  e.linearCoeff += ad * b0 * c0 + a0 * bd * c0 + a0 * b0 * cd;
  e.constCoeff += a0 * b0 * c0;
}

// Smallest real root of a genuine cubic in [0,1].  The reference asks Eigen::PolynomialSolver<float,3>
// (companion-matrix eigenvalues, CollisionDetection.cpp:189-204; Eigen is absent) for all roots and keeps
// the smallest real one in [0,1].  Restated from the definition: [0,1] is split at the critical points of
// the cubic into monotone pieces, and the first piece whose end values differ in sign (or vanish) is
// bisected a fixed 32 times.  (A double root that only touches zero is invisible to both approaches up to
// rounding.)
static bool cubic_smallest_root01(const CubicExpression& c, float& root) {
  float cuts[4];
  int nc = 0;
  cuts[nc++] = 0.0f;
  const float a = 3.0f * c.cubicCoeff, b = 2.0f * c.quadCoeff, cc = c.linearCoeff;
  const float disc = b * b - 4.0f * a * cc;
  if (disc > 0.0f) {
    const float sq = std::sqrt(disc);
    float t0 = (-b - sq) / (2.0f * a), t1 = (-b + sq) / (2.0f * a);
    if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 > 0.0f && t0 < 1.0f) cuts[nc++] = t0;
    if (t1 > 0.0f && t1 < 1.0f) cuts[nc++] = t1;
  }
  cuts[nc++] = 1.0f;
  for (int k = 0; k + 1 < nc; ++k) {
    float lo = cuts[k], hi = cuts[k + 1];
    float flo = c.eval(lo), fhi = c.eval(hi);
    if (flo == 0.0f) { root = lo; return true; }
    if (flo * fhi > 0.0f) continue;
    if (fhi == 0.0f && k + 2 == nc) { /* root at the right end */ }
    for (int it = 0; it < 32; ++it) {
      const float mid = 0.5f * (lo + hi);
      const float fm = c.eval(mid);
      if ((flo < 0.0f) == (fm < 0.0f) && fm != 0.0f) { lo = mid; flo = fm; } else { hi = mid; fhi = fm; }
    }
    root = hi;
    return true;
  }
  return false;
}

// CollisionDetection.cpp:143-205
static bool findRootInInterval(const CubicExpression& c, float& t) {
  if (c.cubicCoeff == 0.0f) {
    if (c.quadCoeff == 0.0f) {
      if (c.linearCoeff == 0.0f) {
        if (c.constCoeff == 0.0f) { t = 0.0f; return true; }
        return false;
      }
      t = -c.constCoeff / c.linearCoeff;
      return t >= 0.0f && t <= 1.0f;
    }
    float disc = c.linearCoeff * c.linearCoeff - 4.0f * c.quadCoeff * c.constCoeff;
    if (disc < 0.0f) return false;
    float sq = std::sqrt(disc);
    t = (-c.linearCoeff - sq) / (2.0f * c.quadCoeff);
    if (t > 1.0f) return false;
    if (t < 0.0f) t = (-c.linearCoeff + sq) / (2.0f * c.quadCoeff);
    return t >= 0.0f && t <= 1.0f;
  }
  return cubic_smallest_root01(c, t);
}

static inline bool bary_outside(const vec3& b) {  // CollisionDetection.cpp:251-253, 295-297
  return (0.0 > b.x) || (b.x > 1.0) || (0.0 > b.y) || (b.y > 1.0) || (b.x + b.y > 1.0);
}

// CollisionDetection.cpp:227-302.  Returns true and t when a contact is reported.
static bool pointTriangleCCD(const vec3& ap0, const vec3& ab0, const vec3& ac0, const vec3& ap1, const vec3& ab1,
                             const vec3& ac1, float thresholdDistance, float& tOut) {
  vec3 n0 = normalize(cross(ab0, ac0));
  vec3 n1 = normalize(cross(ab1, ac1));
  float nDotP0 = dot(n0, ap0);
  float nDotP1 = dot(n1, ap1);
  if (nDotP0 * nDotP1 >= 0.0f) {
    if (nDotP1 >= 0.0f && nDotP1 < thresholdDistance) {
      vec3 bc = inverse(mat3(ab1, ac1, n1)) * ap1;
      if (bary_outside(bc)) return false;
      tOut = 0.0f;
      return true;
    }
    return false;
  }
  vec3 apd = ap1 - ap0, abd = ab1 - ab0, acd = ac1 - ac0;
  CubicExpression e;
  expandTerm(ap0.x, ab0.y, ac0.z, apd.x, abd.y, acd.z, e);
  expandTerm(-ap0.x, ac0.y, ab0.z, -apd.x, acd.y, abd.z, e);
  expandTerm(-ab0.x, ap0.y, ac0.z, -abd.x, apd.y, acd.z, e);
  expandTerm(ab0.x, ac0.y, ap0.z, abd.x, acd.y, apd.z, e);
  expandTerm(ac0.x, ap0.y, ab0.z, acd.x, apd.y, abd.z, e);
  expandTerm(-ac0.x, ab0.y, ap0.z, -acd.x, abd.y, apd.z, e);
  float t;
  if (!findRootInInterval(e, t)) return false;
  vec3 apt = ap0 + t * apd, abt = ab0 + t * abd, act = ac0 + t * acd;
  vec3 n = normalize(cross(abt, act));
  vec3 bc = inverse(mat3(abt, act, n)) * apt;
  if (bary_outside(bc)) return false;
  tOut = t;
  return true;
}

// Src/Solver.cpp:639-677 (cap 20) and :942-979 (cap 50): AABB over position and prevPosition of the three
// nodes, in WORLD units (the computed grid-scaled x1..x3 are unused in the reference).
static CellRange triRange(const std::vector<Node>& nodes, const std::array<uint32_t, 3>& tri, uint32_t cap) {
  vec3 mx = nodes[tri[0]].position, mn = nodes[tri[0]].position;
  for (uint32_t i = 0; i < 3; ++i) {
    const Node& node = nodes[tri[i]];
    for (int k = 0; k < 3; ++k) {
      mx[k] = std::fmax(node.position[k], mx[k]);
      mx[k] = std::fmax(node.prevPosition[k], mx[k]);
      mn[k] = std::fmin(node.position[k], mn[k]);
      mn[k] = std::fmin(node.prevPosition[k], mn[k]);
    }
  }
  CellRange r{};
  r.minX = static_cast<int64_t>(std::floor(mn.x));
  r.minY = static_cast<int64_t>(std::floor(mn.y));
  r.minZ = static_cast<int64_t>(std::floor(mn.z));
  r.lengthX = static_cast<uint32_t>(std::ceil(mx.x) - r.minX);
  r.lengthY = static_cast<uint32_t>(std::ceil(mx.y) - r.minY);
  r.lengthZ = static_cast<uint32_t>(std::ceil(mx.z) - r.minZ);
  if (r.lengthX > cap || r.lengthY > cap || r.lengthZ > cap) return {};
  return r;
}

// ---------------------------------------------------------------------------------------------
// fp32 banded Cholesky on a reverse-Cuthill-McKee ordering.  Stands in for
// Eigen::SimplicialLLT<SparseMatrix<float>> (Src/Solver.cpp:213-215, 258-262, 356): same exact
// factor-and-solve semantics in fp32; fill-reducing ordering and elimination order differ (Eigen
// source absent), which only changes rounding.
// ---------------------------------------------------------------------------------------------
struct SparseSym {  // full symmetric matrix as per-row ordered maps (setup-time only)
  std::vector<std::map<uint32_t, float>> rows;
  void resize(size_t n) { rows.assign(n, {}); }
  float& ref(uint32_t i, uint32_t j) { return rows[i][j]; }
};

// R = float: the reference's arithmetic (SimplicialLLT<SparseMatrix<float>>).  R = double: the YARDSTICK of the PD tolerance
// (FLAG_PD_SOLVE_FP64): the same fp32 matrix and right-hand side, factorised and substituted in double, the solution rounded
// to float once - what the linear solve would return without the fp32 round-off of a system whose entries are ~ m/h^2.
template <class R> struct BandedLLTOf {
  uint32_t n = 0, bw = 0;
  std::vector<uint32_t> perm, iperm;  // perm[new] = old
  std::vector<R> L;                   // column-major band: L(i, j) at L[j*(bw+1) + (i - j)] for j <= i <= j+bw

  static uint32_t bandwidth_of(const SparseSym& A, const std::vector<uint32_t>& ip) {
    uint32_t b = 0;
    for (uint32_t i = 0; i < A.rows.size(); ++i)
      for (auto& kv : A.rows[i]) {
        uint32_t a = ip[i], c = ip[kv.first];
        b = std::max(b, a > c ? a - c : c - a);
      }
    return b;
  }

  // Ordering with the smallest bandwidth among reverse Cuthill-McKee and the three sorts by a coordinate (`xyz`: 3 per
  // node; for a beam the sort along its axis gives one cross-section, where Cuthill-McKee from a corner gives diagonal
  // shells three times as wide).  Only rounding depends on the choice.
  void order(const SparseSym& A, const float* xyz = nullptr) {
    n = static_cast<uint32_t>(A.rows.size());
    perm.clear();
    std::vector<char> seen(n, 0);
    std::vector<uint32_t> deg(n);
    for (uint32_t i = 0; i < n; ++i) deg[i] = static_cast<uint32_t>(A.rows[i].size());
    for (uint32_t start = 0; start < n; ++start) {
      if (seen[start]) continue;
      seen[start] = 1;
      size_t head = perm.size();
      perm.push_back(start);
      while (head < perm.size()) {
        uint32_t u = perm[head++];
        std::vector<uint32_t> nb;
        for (auto& kv : A.rows[u])
          if (!seen[kv.first]) { seen[kv.first] = 1; nb.push_back(kv.first); }
        std::sort(nb.begin(), nb.end(), [&](uint32_t a, uint32_t b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
        for (uint32_t v : nb) perm.push_back(v);
      }
    }
    std::reverse(perm.begin(), perm.end());
    iperm.assign(n, 0);
    for (uint32_t i = 0; i < n; ++i) iperm[perm[i]] = i;
    bw = bandwidth_of(A, iperm);
    if (xyz) {
      for (int axis = 0; axis < 3; ++axis) {
        std::vector<uint32_t> p(n), ip(n);
        std::iota(p.begin(), p.end(), 0u);
        std::stable_sort(p.begin(), p.end(), [&](uint32_t a, uint32_t b) { return xyz[3 * a + axis] < xyz[3 * b + axis]; });
        for (uint32_t i = 0; i < n; ++i) ip[p[i]] = i;
        const uint32_t b = bandwidth_of(A, ip);
        if (b < bw) { bw = b; perm.swap(p); iperm.swap(ip); }
      }
    }
  }

  // sum of a[k] * b[k] over [0, m): sixteen interleaved partial sums combined in a fixed order, so that the compiler may use
  // vector registers without being allowed to reassociate
  static R dot(const R* __restrict a, const R* __restrict b, ptrdiff_t m) {
    R acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    ptrdiff_t k = 0;
    for (; k + 16 <= m; k += 16)
      for (int u = 0; u < 16; ++u) acc[u] += a[k + u] * b[k + u];
    R tail = 0;
    for (; k < m; ++k) tail += a[k] * b[k];
    for (int w = 8; w >= 1; w >>= 1)
      for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
    return acc[0] + tail;
  }

  // Column-major band: L(i, j), j <= i <= j + bw, at L[j * (bw + 1) + (i - j)].  Right-looking (outer-product) form: after
  // column k is scaled, every later column j of the window takes  L(j.., j) -= L(j, k) * L(j.., k)  - contiguous updates
  // without a reduction, which the compiler vectorises as they stand (2.5e11 multiply-adds at 250k nodes).
  // Fill-in decays geometrically away from K's sparsity pattern (K is dominated by m/h^2) and ends as subnormal numbers,
  // which x86 multiplies in microcode ~100x slower: they are flushed to zero while the factorisation and the
  // substitutions run (differences of 1e-38 in entries of order 1e3).
  struct FlushSubnormals {
    unsigned saved;
    FlushSubnormals() : saved(_mm_getcsr()) { _mm_setcsr(saved | 0x8040u); }
    ~FlushSubnormals() { _mm_setcsr(saved); }
  };

  bool factor(const SparseSym& A) {
    FlushSubnormals ftz;
    const size_t W = static_cast<size_t>(bw) + 1;
    L.assign(static_cast<size_t>(n) * W, R(0));
    for (uint32_t io = 0; io < n; ++io)
      for (auto& kv : A.rows[io]) {
        uint32_t i = iperm[io], j = iperm[kv.first];
        if (j <= i) L[j * W + (i - j)] = kv.second;
      }
    for (uint32_t k = 0; k < n; ++k) {
      R* __restrict ck = L.data() + k * W;
      if (!(ck[0] > R(0))) return false;
      const R d = std::sqrt(ck[0]);
      ck[0] = d;
      const uint32_t m = std::min<uint32_t>(bw, n - 1 - k);  // rows k+1 .. k+m below the diagonal
      for (uint32_t r = 1; r <= m; ++r) ck[r] = ck[r] / d;
      for (uint32_t r = 1; r <= m; ++r) {  // column j = k + r, rows j .. k + m
        const R f = ck[r];
        if (f == R(0)) continue;
        R* __restrict cj = L.data() + (static_cast<size_t>(k) + r) * W;
        const R* __restrict src = ck + r;
        const uint32_t len = m - r + 1;
        for (uint32_t t = 0; t < len; ++t) cj[t] -= f * src[t];
      }
    }
    return true;
  }

  // x (old ordering, stride 1 column) <- A^-1 b
  template <class B> void solve(const B* b, float* x) const {
    FlushSubnormals ftz;
    const size_t W = static_cast<size_t>(bw) + 1;
    std::vector<R> y(n);
    for (uint32_t i = 0; i < n; ++i) y[i] = b[perm[i]];
    for (uint32_t k = 0; k < n; ++k) {  // L y = b, column by column
      const R* __restrict ck = L.data() + k * W;
      const R v = y[k] / ck[0];
      y[k] = v;
      const uint32_t m = std::min<uint32_t>(bw, n - 1 - k);
      R* __restrict yy = y.data() + k + 1;
      for (uint32_t r = 0; r < m; ++r) yy[r] -= ck[r + 1] * v;
    }
    for (uint32_t k = n; k-- > 0;) {  // L^T x = y: row k of L^T is column k of L
      const R* ck = L.data() + k * W;
      const uint32_t m = std::min<uint32_t>(bw, n - 1 - k);
      y[k] = (y[k] - dot(ck + 1, y.data() + k + 1, m)) / ck[0];
    }
    for (uint32_t i = 0; i < n; ++i) x[perm[i]] = static_cast<float>(y[i]);
  }
};
using BandedLLT = BandedLLTOf<float>;

}  // namespace

// ---------------------------------------------------------------------------------------------
// The solver (Include/Pies/Solver.h:40-199, Src/Solver.cpp)
// ---------------------------------------------------------------------------------------------
struct ora_solver {
  Options opt;
  uint32_t constraintId = 0;
  bool releaseHinge = false;
  bool nodeCollisions = true;  // extension: false skips Solver.cpp:81-130 (not a reference option)
  bool triangleCollisions = true;  // extension: false skips the point-triangle CCD of Solver.cpp:693-797
  bool simFailed = false;

  std::vector<Node> nodes;
  std::vector<PositionCon> positionCons;
  std::vector<DistanceCon> distanceCons;
  std::vector<TetCon> tetCons;
  std::vector<VolumeCon> volumeCons;
  std::vector<BendCon> bendCons;
  std::vector<ShapeCon> shapeCons;
  std::vector<GoalCon> goalCons;
  std::vector<std::array<uint32_t, 3>> triangles;
  std::vector<uint32_t> lines;
  std::vector<std::array<uint32_t, 4>> tets;

  struct FixedRegion { mat4 invInitialTransform; uint32_t goal; };  // Solver.h:147-151
  std::vector<FixedRegion> fixedRegions;

  // optional replay of a device collision visiting order (empty = reference order 0..N-1)
  std::vector<uint32_t> collisionOrder;
  // collisionRule 0: the reference's loop (ascending node index, query range from the node's current
  // position).  1: the device's documented rule (DESIGN.md "Node-node collisions"): nodes are visited
  // pass by pass, pass = (min cell mod 3) per axis at hash-build time, then by min cell, then ascending
  // index, and a node queries the cell range it was *inserted* with.  Same per-pair arithmetic.
  // 2: the device's pair order ("pair order"): the same visits - node i meets node j once per cell that
  // the ranges they were inserted with share, in each direction, itself included (quirk Q3) - re-ordered pair by pair:
  // first every node's visits to itself (ascending index), then the unordered pairs {i < j} whose ranges share m > 0 cells
  // in ascending order of pair_key (ora_math.h: direction class and parity from the positions the grid was built from, then a
  // 64-bit mix of the two indices), each as m visits of i to j followed by m visits of j to i.  A pure re-ordering of rule 1's
  // visits; every visit tests the live positions like the reference's loop.
  int collisionRule = 0;

  // Optional multi-core replay for the all-cores CPU baseline (bench.py): conflict-free batches of each container
  // (ora_set_batches, taken from a coloured device plan) are swept with `threads` OpenMP threads.  A batch touches every
  // node at most once, so the result is bit-identical to the single-threaded sweep in the same order.
  int threads = 1;
  std::vector<uint32_t> batchOffs[5];
  // the reference's own threading (CPU baseline): 16 threads for the node-hash insert (SpatialHash.h:134), threadCount
  // threads for the PD collision detection (Solver.h:36, Solver.cpp:558-566); everything else is single-threaded there
  bool referenceThreads = false;

  NodeHash hashNodes;
  std::vector<StaticCollision> staticCollisions;
  std::vector<TriCollision> triCollisions;
  std::vector<NodePairCollision> nodePairs;  // extension: Solver.h:186 _collisions, which the reference never fills

  // PD state (Solver.h:163-171)
  bool pdDirty = true;
  uint32_t previousNodeCount = 0;
  SparseSym stiffness;
  BandedLLT llt;
  BandedLLTOf<double> llt64;  // FLAG_PD_SOLVE_FP64: the yardstick solve
  std::vector<double> force64;
  bool solveFp64 = false;
  bool orderedWithContacts = false;
  std::vector<float> state, force, msn;  // N x 3, column-major like Eigen::MatrixXf
  uint64_t stat_collision_pairs = 0;

  void tick();
  void tickPBD();
  void tickPD();
  void detectPD();
};

// Constraints.h:121-129
template <class C, int N> static inline void projectNodePositions(C& c, std::vector<Node>& nodes) {
  std::array<vec3, N> fixed;
  c.project(nodes, fixed);
  for (int i = 0; i < N; ++i) {
    Node& node = nodes[c.nodeIds[i]];
    node.position += c.w * (fixed[i] - node.position);
  }
}

// Src/Solver.cpp:25-38
void ora_solver::tick() {
  if (simFailed) return;
  if (opt.solver == 0) tickPBD(); else tickPD();
}

// One Gauss-Seidel sweep over a container: sequential (the reference's loop), or batch by batch with OpenMP when
// conflict-free batches were supplied and threads > 1 (same result: no two constraints of a batch share a written node).
template <class C, int N> static void sweep(std::vector<C>& cons, std::vector<Node>& nodes, const std::vector<uint32_t>& offs, int threads) {
  if (threads > 1 && offs.size() >= 2 && offs.back() == cons.size()) {
    for (size_t b = 0; b + 1 < offs.size(); ++b) {
      const int64_t lo = offs[b], hi = offs[b + 1];
#pragma omp parallel for num_threads(threads) schedule(static)
      for (int64_t k = lo; k < hi; ++k) projectNodePositions<C, N>(cons[static_cast<size_t>(k)], nodes);
    }
    return;
  }
  for (C& c : cons) projectNodePositions<C, N>(c, nodes);
}

// Src/Solver.cpp:40-160
void ora_solver::tickPBD() {
  float deltaTime = opt.fixedTimestepSize / opt.timeSubsteps;
  const int64_t nNodes = static_cast<int64_t>(nodes.size());
  for (uint32_t substep = 0; substep < opt.timeSubsteps; ++substep) {
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
    for (int64_t k = 0; k < nNodes; ++k) {  // :47-52
      Node& node = nodes[static_cast<size_t>(k)];
      node.prevPosition = node.position;
      node.position += node.velocity * deltaTime + vec3(0.0f, -opt.gravity, 0.0f) * deltaTime * deltaTime;
    }
    for (uint32_t i = 0; i < opt.iterations; ++i) {
      if (!releaseHinge) sweep<PositionCon, 1>(positionCons, nodes, batchOffs[0], threads);  // :59-63
      sweep<DistanceCon, 2>(distanceCons, nodes, batchOffs[1], threads);                       // :65-67
      sweep<TetCon, 4>(tetCons, nodes, batchOffs[2], threads);                                 // :69-71
      sweep<BendCon, 4>(bendCons, nodes, batchOffs[4], threads);                               // :73-75

      if (nodeCollisions) {
        hashNodes.clear();                                 // :81
        hashNodes.bulkInsert(nodes, opt.gridSpacing);      // :82
        std::vector<const std::vector<uint32_t>*> scratch;
        const size_t n = nodes.size();
        std::vector<CellRange> inserted;
        if (collisionRule == 1) {
          inserted.resize(n);
          for (size_t k = 0; k < n; ++k) inserted[k] = nodeCompRange(nodes[k], opt.gridSpacing);
          auto mod3 = [](int64_t v) { return static_cast<int>(((v % 3) + 3) % 3); };
          auto colour = [&](const CellRange& r) { return mod3(r.minX) + 3 * mod3(r.minY) + 9 * mod3(r.minZ); };
          collisionOrder.resize(n);
          std::iota(collisionOrder.begin(), collisionOrder.end(), 0u);
          std::stable_sort(collisionOrder.begin(), collisionOrder.end(), [&](uint32_t a, uint32_t b) {
            const CellRange &ra = inserted[a], &rb = inserted[b];
            int ca = colour(ra), cb = colour(rb);
            if (ca != cb) return ca < cb;
            if (ra.minX != rb.minX) return ra.minX < rb.minX;
            if (ra.minY != rb.minY) return ra.minY < rb.minY;
            if (ra.minZ != rb.minZ) return ra.minZ < rb.minZ;
            return a < b;
          });
        }
        // one visit of `node` to `other` (:88-126), whatever the order the visits come in
        auto visit = [&](Node& node, Node* other) {
          vec3 diff = other->position - node.position;
          float dist = length(diff);
          float disp = node.radius + other->radius - dist;
          if (disp <= 0.0) return;
          ++stat_collision_pairs;
          vec3 dir(1.0f, 0.0f, 0.0f);
          if (dist > 0.00001f) dir = diff / dist;
          float wSum = node.invMass + other->invMass;
          node.position += 0.85f * -disp * dir * node.invMass / wSum;
          other->position += 0.85f * disp * dir * other->invMass / wSum;
          vec3 relativeVelocity = other->velocity - node.velocity;
          vec3 perpVel = relativeVelocity - dot(relativeVelocity, dir) * dir;
          float friction = opt.friction;
          if (length(perpVel) < opt.staticFrictionThreshold) friction = 1.0f;
          node.velocity += -friction * perpVel * node.invMass / wSum;
          other->velocity += friction * perpVel * other->invMass / wSum;
        };
        if (collisionRule == 2) {
          inserted.resize(n);
          for (size_t k = 0; k < n; ++k) inserted[k] = nodeCompRange(nodes[k], opt.gridSpacing);
          // cells two inserted ranges share, per axis and in total
          auto shared = [](int64_t a0, uint32_t la, int64_t b0, uint32_t lb) {
            const int64_t lo = std::max(a0, b0), hi = std::min(a0 + static_cast<int64_t>(la), b0 + static_cast<int64_t>(lb));
            return hi > lo ? static_cast<uint32_t>(hi - lo) : 0u;
          };
          struct Pair { uint64_t key; uint32_t i, j, m; };
          std::vector<Pair> pairs;  // (listed before any visit: the key looks at the positions the grid was built from)
          for (uint32_t a = 0; a < n; ++a) {
            const CellRange& ra = inserted[a];
            // every j > a sharing a cell with a is found in the bucket of the shared box's minimum corner, exactly once
            for (uint32_t dx = 0; dx < ra.lengthX; ++dx)
              for (uint32_t dy = 0; dy < ra.lengthY; ++dy)
                for (uint32_t dz = 0; dz < ra.lengthZ; ++dz) {
                  const CellId c{ra.minX + dx, ra.minY + dy, ra.minZ + dz};
                  const std::vector<uint32_t>* bucket = hashNodes.find(c);
                  if (!bucket) continue;
                  for (uint32_t b : *bucket) {
                    if (b <= a) continue;
                    const CellRange& rb = inserted[b];
                    if (c.x != std::max(ra.minX, rb.minX) || c.y != std::max(ra.minY, rb.minY) || c.z != std::max(ra.minZ, rb.minZ)) continue;
                    const uint32_t m = shared(ra.minX, ra.lengthX, rb.minX, rb.lengthX) * shared(ra.minY, ra.lengthY, rb.minY, rb.lengthY) *
                                       shared(ra.minZ, ra.lengthZ, rb.minZ, rb.lengthZ);
                    pairs.push_back(Pair{pair_key(a, b, nodes[a].position, nodes[b].position), a, b, m});
                  }
                }
          }
          for (size_t k = 0; k < n; ++k) {  // a node is in every bucket of its own range: it meets itself once per cell
            const CellRange& r = inserted[k];
            const uint32_t m = r.lengthX * r.lengthY * r.lengthZ;
            for (uint32_t t = 0; t < m; ++t) visit(nodes[k], &nodes[k]);
          }
          std::sort(pairs.begin(), pairs.end(), [](const Pair& x, const Pair& y) { return x.key < y.key; });  // (the key is a bijection of (i, j))
          for (const Pair& pr : pairs) {
            for (uint32_t t = 0; t < pr.m; ++t) visit(nodes[pr.i], &nodes[pr.j]);
            for (uint32_t t = 0; t < pr.m; ++t) visit(nodes[pr.j], &nodes[pr.i]);
          }
        } else
        for (size_t oi = 0; oi < n; ++oi) {                // :86-130
          const uint32_t ni = collisionOrder.empty() ? static_cast<uint32_t>(oi) : collisionOrder[oi];
          Node& node = nodes[ni];
          if (collisionRule == 1) hashNodes.findCollisions(inserted[ni], scratch);
          else hashNodes.findCollisions(node, opt.gridSpacing, scratch);
          for (const std::vector<uint32_t>* bucket : scratch)
            for (uint32_t otherId : *bucket) visit(node, &nodes[otherId]);
          scratch.clear();
        }
      }
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
      for (int64_t k = 0; k < nNodes; ++k) {  // :132-136
        Node& node = nodes[static_cast<size_t>(k)];
        if (node.position.y - node.radius < opt.floorHeight) node.position.y = opt.floorHeight + node.radius;
      }
    }
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
    for (int64_t k = 0; k < nNodes; ++k) {  // :140-158
      Node& node = nodes[static_cast<size_t>(k)];
      node.velocity = (1.0f - opt.damping) * (node.position - node.prevPosition) / deltaTime;
      if (node.position.y - node.radius <= opt.floorHeight) {
        float l = std::sqrt(node.velocity.x * node.velocity.x + node.velocity.z * node.velocity.z);
        if (l < 5.0f) {
          node.velocity.x = 0.0f;
          node.velocity.z = 0.0f;
        } else {
          node.velocity.x *= 1.0f - opt.friction;
          node.velocity.z *= 1.0f - opt.friction;
        }
      }
    }
  }
}

// Constraints.h:70-81
template <class C, int N> static void addStiffness(const C& c, SparseSym& K) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) K.ref(c.nodeIds[i], c.nodeIds[j]) += c.w * c.AtA[i][j];
}
// Constraints.h:89-105
template <class C, int N, class F> static void addForce(const C& c, F* f, size_t n) {
  for (int i = 0; i < N; ++i) {
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int k = 0; k < N; ++k) {
      ax += c.AtB[i][k] * c.projected[k].x;
      ay += c.AtB[i][k] * c.projected[k].y;
      az += c.AtB[i][k] * c.projected[k].z;
    }
    uint32_t id = c.nodeIds[i];
    f[id] += c.w * ax;
    f[n + id] += c.w * ay;
    f[2 * n + id] += c.w * az;
  }
}

// Src/Solver.cpp:680-875: triangle hash, swept-range candidate search, three point-triangle CCD tests per
// candidate pair, floor contacts, per-thread lists merged thread by thread.
void ora_solver::detectPD() {
  staticCollisions.clear();
  triCollisions.clear();
  const uint32_t T = std::max(1u, opt.threadCount);
  std::unordered_map<CellId, std::vector<uint32_t>, CellHash> triHash;
  if (triangleCollisions) {  // :693, SpatialHash.h:129-189 with TriCompRange
    for (uint32_t ti = 0; ti < triangles.size(); ++ti) {
      CellRange r = triRange(nodes, triangles[ti], 50);
      for (uint32_t dx = 0; dx < r.lengthX; ++dx)
        for (uint32_t dy = 0; dy < r.lengthY; ++dy)
          for (uint32_t dz = 0; dz < r.lengthZ; ++dz) triHash[CellId{r.minX + dx, r.minY + dy, r.minZ + dz}].push_back(ti);
    }
  }
  // threadCount threads, triangle ids strided (:714-715); with referenceThreads the loop really runs on that many OpenMP
  // threads like the reference's std::threads, otherwise thread after thread - the merge (:852-874) is in thread order
  // either way, so the lists are the same
  std::vector<std::vector<TriCollision>> perThreadTri(T);
  std::vector<std::vector<StaticCollision>> perThreadStatic(T);
  std::vector<char> perThreadFailed(T, 0);
#pragma omp parallel for num_threads(static_cast<int>(T)) schedule(static, 1) if (referenceThreads && T > 1)
  for (int ti = 0; ti < static_cast<int>(T); ++ti) {
    const uint32_t t = static_cast<uint32_t>(ti);
    std::vector<TriCollision>& mine = perThreadTri[t];
    std::vector<StaticCollision>& mineStatic = perThreadStatic[t];
    bool failed = false;
    std::vector<const std::vector<uint32_t>*> buckets;
    for (size_t triId = t; triId < triangles.size() && !failed; triId += T) {
      const auto& tri = triangles[triId];
      if (triangleCollisions) {
        CellRange range = triRange(nodes, tri, 20);  // sweptTriRange
        for (uint32_t dx = 0; dx < range.lengthX; ++dx)
          for (uint32_t dy = 0; dy < range.lengthY; ++dy)
            for (uint32_t dz = 0; dz < range.lengthZ; ++dz) {
              auto it = triHash.find(CellId{range.minX + dx, range.minY + dy, range.minZ + dz});
              if (it != triHash.end()) buckets.push_back(&it->second);
            }
        if (buckets.size() > 1000) { failed = true; break; }  // :741-745
        for (const std::vector<uint32_t>* bucket : buckets) {
          for (uint32_t other : *bucket) {
            if (bucket->size() > 1000) { failed = true; break; }  // :751-755
            const auto& ot = triangles[other];
            bool common = false;
            for (uint32_t i = 0; i < 3; ++i)
              for (uint32_t j = 0; j < 3; ++j)
                if (tri[i] == ot[j]) common = true;
            if (common) continue;
            const Node& nodeB = nodes[ot[0]];
            const Node& nodeC = nodes[ot[1]];
            const Node& nodeD = nodes[ot[2]];
            for (uint32_t i = 0; i < 3; ++i) {
              const Node& nodeA = nodes[tri[i]];
              float tt;
              if (!pointTriangleCCD(nodeA.prevPosition - nodeB.prevPosition, nodeC.prevPosition - nodeB.prevPosition,
                                    nodeD.prevPosition - nodeB.prevPosition, nodeA.position - nodeB.position,
                                    nodeC.position - nodeB.position, nodeD.position - nodeB.position,
                                    opt.collisionThresholdDistance, tt))
                continue;
              TriCollision c;  // CollisionConstraint.cpp:67-84
              c.nodeIds[0] = nodeA.id; c.nodeIds[1] = nodeB.id; c.nodeIds[2] = nodeC.id; c.nodeIds[3] = nodeD.id;
              c.thickness = opt.collisionThickness;
              const float A[4][4] = {{0, 0, 0, 0}, {-1, 1, 0, 0}, {-1, 0, 1, 0}, {-1, 0, 0, 1}};
              for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) {
                  float acc = 0.f;
                  for (int k = 0; k < 4; ++k) acc += A[k][a] * A[k][b];
                  c.AtA[a][b] = acc;
                }
              mine.push_back(c);
            }
          }
          if (failed) break;
        }
        buckets.clear();
        if (failed) break;
      }
      for (uint32_t i = 0; i < 3; ++i) {  // :829-834
        const Node& node = nodes[tri[i]];
        if (node.position.y < opt.floorHeight + opt.collisionThickness) {
          StaticCollision sc;
          sc.nodeId = node.id;
          mineStatic.push_back(sc);
        }
      }
    }
    perThreadFailed[t] = failed ? 1 : 0;
  }
  for (uint32_t t = 0; t < T; ++t) {
    if (perThreadFailed[t]) { simFailed = true; return; }  // :853-856
    triCollisions.insert(triCollisions.end(), perThreadTri[t].begin(), perThreadTri[t].end());
    staticCollisions.insert(staticCollisions.end(), perThreadStatic[t].begin(), perThreadStatic[t].end());
  }
}

// CollisionConstraint.cpp:86-124 (project) / :126-162 (stabilize): normal of triangle b,c,d; push node a out
static inline bool tri_contact_normal(const std::vector<Node>& nodes, const TriCollision& c, vec3& disp) {
  const Node& A = nodes[c.nodeIds[0]];
  const Node& B = nodes[c.nodeIds[1]];
  const Node& C = nodes[c.nodeIds[2]];
  const Node& D = nodes[c.nodeIds[3]];
  vec3 p = A.position - B.position;
  vec3 n = normalize(cross(C.position - B.position, D.position - B.position));
  float nDotP = dot(n, p);
  if (nDotP < c.thickness) {
    disp = (c.thickness - nDotP) * n;
    return true;
  }
  return false;
}

static std::vector<float> node_xyz(const std::vector<Node>& nodes) {
  std::vector<float> p(3 * nodes.size());
  for (size_t i = 0; i < nodes.size(); ++i) { p[3 * i] = nodes[i].position.x; p[3 * i + 1] = nodes[i].position.y; p[3 * i + 2] = nodes[i].position.z; }
  return p;
}

// Src/Solver.cpp:162-486
void ora_solver::tickPD() {
  uint32_t nodeCount = static_cast<uint32_t>(nodes.size());
  const size_t n = nodeCount;
  float h = opt.fixedTimestepSize / opt.timeSubsteps;
  float h2 = h * h;

  if (pdDirty || previousNodeCount != nodeCount) {  // :168 (dirty flag: benign divergence, Q9)
    previousNodeCount = nodeCount;
    pdDirty = false;
    stiffness.resize(n);
    for (uint32_t i = 0; i < nodeCount; ++i) stiffness.ref(i, i) = 1.0f / (nodes[i].invMass * h2);  // :179-182
    for (auto& c : positionCons) addStiffness<PositionCon, 1>(c, stiffness);
    for (auto& c : distanceCons) addStiffness<DistanceCon, 2>(c, stiffness);
    for (auto& c : tetCons) addStiffness<TetCon, 4>(c, stiffness);
    for (auto& c : volumeCons) addStiffness<VolumeCon, 4>(c, stiffness);
    for (auto& c : shapeCons)
      for (uint32_t id : c.ids) stiffness.ref(id, id) += c.w;  // ShapeMatchingConstraint.cpp:50-56
    for (auto& c : goalCons)
      for (uint32_t id : c.ids) stiffness.ref(id, id) += c.w;  // :139-145
    for (auto& c : bendCons) addStiffness<BendCon, 4>(c, stiffness);
    llt.order(stiffness, node_xyz(nodes).data());
    state.assign(3 * n, 0.f);
    force.assign(3 * n, 0.f);
    msn.assign(3 * n, 0.f);
  }

  for (Node& node : nodes) node.force = vec3(0.0f, -opt.gravity, 0.0f) / node.invMass;  // :224-226

  for (uint32_t substep = 0; substep < opt.timeSubsteps; ++substep) {
    for (uint32_t i = 0; i < nodeCount; ++i) {  // :229-238
      Node& node = nodes[i];
      node.position += h * node.velocity;
      vec3 m = node.position / node.invMass / h2;
      msn[i] = m.x; msn[n + i] = m.y; msn[2 * n + i] = m.z;
    }

    detectPD();  // :240

    // :242-262  C = sum of the collision blocks (tri, edge, static), then K + C, re-factor
    SparseSym col;
    col.resize(n);
    for (const TriCollision& c : triCollisions)  // CollisionConstraint.cpp:164-174
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) col.ref(c.nodeIds[i], c.nodeIds[j]) += c.w * c.AtA[i][j];
    for (const StaticCollision& c : staticCollisions) col.ref(c.nodeId, c.nodeId) += c.w;  // :441-445
    for (const NodePairCollision& c : nodePairs) {  // CollisionConstraint.cpp:43-47 (extension: see NodePairCollision)
      col.ref(c.nodeIds[0], c.nodeIds[0]) += c.w;
      col.ref(c.nodeIds[1], c.nodeIds[1]) += c.w;
    }
    SparseSym sys = stiffness;
    for (uint32_t i = 0; i < nodeCount; ++i)
      for (const auto& kv : col.rows[i]) {
        auto it = stiffness.rows[i].find(kv.first);
        sys.ref(i, kv.first) = (it != stiffness.rows[i].end() ? it->second : 0.0f) + kv.second;
      }
    if (!triCollisions.empty()) llt.order(sys, node_xyz(nodes).data());  // contact blocks fall outside K's band
    else if (llt.bw == 0 || orderedWithContacts) llt.order(stiffness, node_xyz(nodes).data());
    orderedWithContacts = !triCollisions.empty();
    const bool timing = std::getenv("PIES_ORACLE_TIMING") != nullptr;
    const auto tFactor0 = std::chrono::steady_clock::now();
    if (solveFp64) {  // same ordering, the factor in double
      llt64.n = llt.n; llt64.bw = llt.bw; llt64.perm = llt.perm; llt64.iperm = llt.iperm;
      if (!llt64.factor(sys)) { simFailed = true; return; }
    } else
    if (!llt.factor(sys)) { simFailed = true; return; }
    if (timing)
      std::fprintf(stderr, "[oracle] n %u bandwidth %u: factorisation %.2f s\n", nodeCount, llt.bw,
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - tFactor0).count());

    for (uint32_t iter = 0; iter < opt.iterations; ++iter) {
      force = msn;  // :266
      // local step :270-308
      for (auto& c : positionCons) c.project(nodes, c.projected);
      for (auto& c : distanceCons) c.project(nodes, c.projected);
      for (auto& c : tetCons) c.project(nodes, c.projected);
      for (auto& c : bendCons) c.project(nodes, c.projected);
      for (auto& c : volumeCons) c.project(nodes, c.projected);
      for (auto& c : shapeCons) c.project(nodes);
      for (auto& c : goalCons) c.project();
      for (auto& c : triCollisions) {  // CollisionConstraint.cpp:86-124
        for (int k = 0; k < 4; ++k) c.projectedPositions[k] = nodes[c.nodeIds[k]].position;
        vec3 disp;
        c.colliding = tri_contact_normal(nodes, c, disp);
        if (c.colliding) c.projectedPositions[0] += disp;
      }
      for (auto& c : staticCollisions) {  // CollisionConstraint.cpp:447-455
        const Node& node = nodes[c.nodeId];
        c.projectedPosition = node.position;
        if (node.position.y < 0.0f) c.projectedPosition.y = 0.0f;
      }
      for (auto& c : nodePairs) c.project(nodes);
      // RHS :310-349.  `f` is the reference's float force vector - or, for the fp64 yardstick (FLAG_PD_SOLVE_FP64), a double
      // copy of it: the contributions w * (AtB p) are the same fp32 values, but they are ADDED without the round-off of a
      // float accumulator that starts at M s_n / h^2 ~ 1e6 (an ulp of 0.1 per addition, ~50 additions per node)
      auto accumulate = [&](auto* f) {
      for (auto& c : positionCons) addForce<PositionCon, 1>(c, f, n);
      for (auto& c : distanceCons) addForce<DistanceCon, 2>(c, f, n);
      for (auto& c : tetCons) addForce<TetCon, 4>(c, f, n);
      for (auto& c : volumeCons) addForce<VolumeCon, 4>(c, f, n);
      for (auto& c : bendCons) addForce<BendCon, 4>(c, f, n);
      for (auto& c : shapeCons)
        for (size_t i = 0; i < c.ids.size(); ++i) {
          uint32_t id = c.ids[i];
          f[id] += c.w * c.proj[3 * i + 0];
          f[n + id] += c.w * c.proj[3 * i + 1];
          f[2 * n + id] += c.w * c.proj[3 * i + 2];
        }
      for (auto& c : goalCons)
        for (size_t i = 0; i < c.ids.size(); ++i) {
          uint32_t id = c.ids[i];
          f[id] += c.w * c.proj[3 * i + 0];
          f[n + id] += c.w * c.proj[3 * i + 1];
          f[2 * n + id] += c.w * c.proj[3 * i + 2];
        }
      for (auto& c : triCollisions)  // CollisionConstraint.cpp:176-194: A == B, so AtB p = AtA p
        for (int i = 0; i < 4; ++i) {
          float ax = 0.f, ay = 0.f, az = 0.f;
          for (int k = 0; k < 4; ++k) {
            ax += c.AtA[i][k] * c.projectedPositions[k].x;
            ay += c.AtA[i][k] * c.projectedPositions[k].y;
            az += c.AtA[i][k] * c.projectedPositions[k].z;
          }
          uint32_t id = c.nodeIds[i];
          f[id] += c.w * ax;
          f[n + id] += c.w * ay;
          f[2 * n + id] += c.w * az;
        }
      for (auto& c : staticCollisions) {  // CollisionConstraint.cpp:457-463
        f[c.nodeId] += c.w * c.projectedPosition.x;
        f[n + c.nodeId] += c.w * c.projectedPosition.y;
        f[2 * n + c.nodeId] += c.w * c.projectedPosition.z;
      }
      for (auto& c : nodePairs)  // CollisionConstraint.cpp:49-65 (every thread's share: both nodes)
        for (int k = 0; k < 2; ++k) {
          f[c.nodeIds[k]] += c.w * c.projectedPositions[k].x;
          f[n + c.nodeIds[k]] += c.w * c.projectedPositions[k].y;
          f[2 * n + c.nodeIds[k]] += c.w * c.projectedPositions[k].z;
        }
      };
      if (solveFp64) {
        force64.assign(force.begin(), force.end());
        accumulate(force64.data());
      } else {
        accumulate(force.data());
      }
      // global step :356-364
      for (int col = 0; col < 3; ++col) {
        if (solveFp64) llt64.solve(&force64[col * n], &state[col * n]);
        else llt.solve(&force[col * n], &state[col * n]);
      }
      for (uint32_t i = 0; i < nodeCount; ++i) nodes[i].position = vec3(state[i], state[n + i], state[2 * n + i]);
    }

    for (uint32_t ci = 0; ci < opt.collisionStabilizationIterations; ++ci) {  // :367-383
      for (auto& c : triCollisions) {  // CollisionConstraint.cpp:126-162
        vec3 disp;
        if (!tri_contact_normal(nodes, c, disp)) continue;
        Node& A = nodes[c.nodeIds[0]];
        Node& B = nodes[c.nodeIds[1]];
        Node& C = nodes[c.nodeIds[2]];
        Node& D = nodes[c.nodeIds[3]];
        float wTriSum = B.invMass + C.invMass + D.invMass;
        float wSum = A.invMass + wTriSum;
        A.position += disp * A.invMass / wSum;
        B.position -= disp * wTriSum / wSum;
        C.position -= disp * wTriSum / wSum;
        D.position -= disp * wTriSum / wSum;
        A.prevPosition += disp * A.invMass / wSum;
        B.prevPosition -= disp * wTriSum / wSum;
        C.prevPosition -= disp * wTriSum / wSum;
        D.prevPosition -= disp * wTriSum / wSum;
      }
      for (auto& c : staticCollisions) nodes[c.nodeId].position = c.projectedPosition;
    }

    for (uint32_t i = 0; i < nodeCount; ++i) {  // :386-395
      Node& node = nodes[i];
      node.velocity = (1.0f - opt.damping) * (node.position - node.prevPosition) / h + h * node.force * node.invMass;
      node.prevPosition = node.position;
    }
    for (const NodePairCollision& collision : nodePairs) {  // :398-428
      Node& a = nodes[collision.nodeIds[0]];
      Node& b = nodes[collision.nodeIds[1]];
      vec3 diff = b.position - a.position;
      float dist = length(diff);
      if (dist > a.radius + b.radius) continue;
      vec3 nrm = diff / dist;
      vec3 relativeVelocity = b.velocity - a.velocity;
      vec3 perpVel = relativeVelocity - dot(relativeVelocity, nrm) * nrm;
      float friction = -opt.friction;
      if (length(perpVel) < opt.staticFrictionThreshold) friction = 1.0f;
      float wSum = a.invMass + b.invMass;
      a.velocity += -friction * perpVel * a.invMass / wSum;
      b.velocity += friction * perpVel * b.invMass / wSum;
    }
    for (const TriCollision& col : triCollisions) {  // :431-471
      Node& a = nodes[col.nodeIds[0]];
      Node& b = nodes[col.nodeIds[1]];
      Node& c = nodes[col.nodeIds[2]];
      Node& d = nodes[col.nodeIds[3]];
      vec3 avgTriVelocity = (b.velocity + c.velocity + d.velocity) / 3.0f;
      vec3 nn = normalize(cross(c.position - b.position, d.position - b.position));
      vec3 relativeVelocity = a.velocity - avgTriVelocity;
      float vDotN = dot(relativeVelocity, nn);
      vec3 normVel = vDotN * nn;
      vec3 perpVel = relativeVelocity - normVel;
      float friction = opt.friction;
      if (length(perpVel) < opt.staticFrictionThreshold) friction = 1.0f;
      float triWSum = b.invMass + c.invMass + d.invMass;
      float wSum = a.invMass + triWSum;
      vec3 dv = -friction * perpVel - 1.1f * std::fmin(vDotN, 0.0f) * nn;
      a.velocity += dv * a.invMass / wSum;
      b.velocity += -dv * triWSum / wSum;
      c.velocity += -dv * triWSum / wSum;
      d.velocity += -dv * triWSum / wSum;
    }
    for (const StaticCollision& c : staticCollisions) {  // :473-484
      Node& node = nodes[c.nodeId];
      vec3 perpVel(node.velocity.x, 0.0f, node.velocity.z);
      float friction = opt.friction;
      if (length(perpVel) < opt.staticFrictionThreshold) friction = 1.0f;
      node.velocity += -friction * perpVel;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Factories (Src/Constraints.cpp:39-56, 65-74, 130-184, 257-310, 368-394)
// ---------------------------------------------------------------------------------------------
static DistanceCon makeDistance(uint32_t id, const Node& a, const Node& b, float w) {
  float A[2][2] = {{0.5f, -0.5f}, {-0.5f, 0.5f}};
  DistanceCon c;
  c.init(id, w, A, A, {a.id, b.id});
  c.targetDistance = length(b.position - a.position);
  return c;
}
static PositionCon makePosition(uint32_t id, const Node& n, float w) {
  float I[1][1] = {{1.0f}};
  PositionCon c;
  c.init(id, w, I, I, {n.id});
  c.fixedPosition = n.position;
  return c;
}
// A = [0 ; diffToBary_ * worldToDiff] with diffToBary_(r,c) = diffToBary[r][c] (glm indices), B = I
static void tetA(const Node& x1, const Node& x2, const Node& x3, const Node& x4, mat3& Qinv, float A[4][4]) {
  mat3 baryToDiff(x2.position - x1.position, x3.position - x1.position, x4.position - x1.position);
  Qinv = inverse(baryToDiff);
  const float D[3][4] = {{-1, 1, 0, 0}, {-1, 0, 1, 0}, {-1, 0, 0, 1}};
  for (int j = 0; j < 4; ++j) A[0][j] = 0.0f;
  for (int r = 0; r < 3; ++r)
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
      for (int k = 0; k < 3; ++k) s += Qinv[r][k] * D[k][j];
      A[1 + r][j] = s;
    }
}
static const float I4[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
static TetCon makeTet(uint32_t id, float w, const Node& x1, const Node& x2, const Node& x3, const Node& x4,
                      float minStrain, float maxStrain) {
  TetCon c;
  float A[4][4];
  tetA(x1, x2, x3, x4, c.Qinv, A);
  c.init(id, w, A, I4, {x1.id, x2.id, x3.id, x4.id});
  c.minStrain = minStrain;
  c.maxStrain = maxStrain;
  return c;
}
static VolumeCon makeVolume(uint32_t id, float w, const Node& x1, const Node& x2, const Node& x3, const Node& x4,
                            float compression, float stretching) {
  VolumeCon c;
  float A[4][4];
  tetA(x1, x2, x3, x4, c.Qinv, A);
  c.init(id, w, A, I4, {x1.id, x2.id, x3.id, x4.id});
  c.minOmega = compression;
  c.maxOmega = stretching;
  return c;
}
static BendCon makeBend(uint32_t id, float w, const Node& x1, const Node& x2, const Node& x3, const Node& x4) {
  vec3 p2 = x2.position - x1.position;
  vec3 p3 = x3.position - x1.position;
  vec3 p4 = x4.position - x1.position;
  vec3 n1 = normalize(cross(p2, p3));
  vec3 n2 = normalize(cross(p2, p4));
  BendCon c;
  c.init(id, w, I4, I4, {x1.id, x2.id, x3.id, x4.id});
  c.initialAngle = acos_f(dot(n1, n2));
  return c;
}

// Src/PrimitiveUtilities.cpp:35-38
static inline uint32_t gid(uint32_t H, uint32_t D, size_t off, uint32_t x, uint32_t y, uint32_t z) {
  return z + D * (y + H * x) + static_cast<uint32_t>(off);
}

// Src/PrimitiveUtilities.cpp:524-606 / :734-816 (identical in createBox and createTetBox)
static void boxSurface(ora_solver* s, uint32_t W, uint32_t H, uint32_t D, size_t off) {
  auto G = [&](uint32_t x, uint32_t y, uint32_t z) { return gid(H, D, off, x, y, z); };
  auto& T = s->triangles;
  for (uint32_t i = 0; i + 1 < W; ++i)
    for (uint32_t j = 0; j + 1 < H; ++j) {
      T.push_back({G(i, j, 0), G(i + 1, j + 1, 0), G(i + 1, j, 0)});
      T.push_back({G(i, j, 0), G(i, j + 1, 0), G(i + 1, j + 1, 0)});
      T.push_back({G(i, j, D - 1), G(i + 1, j, D - 1), G(i + 1, j + 1, D - 1)});
      T.push_back({G(i, j, D - 1), G(i + 1, j + 1, D - 1), G(i, j + 1, D - 1)});
    }
  for (uint32_t i = 0; i + 1 < W; ++i)
    for (uint32_t k = 0; k + 1 < D; ++k) {
      T.push_back({G(i, 0, k), G(i + 1, 0, k), G(i + 1, 0, k + 1)});
      T.push_back({G(i, 0, k), G(i + 1, 0, k + 1), G(i, 0, k + 1)});
      T.push_back({G(i, H - 1, k), G(i + 1, H - 1, k + 1), G(i + 1, H - 1, k)});
      T.push_back({G(i, H - 1, k), G(i, H - 1, k + 1), G(i + 1, H - 1, k + 1)});
    }
  for (uint32_t j = 0; j + 1 < H; ++j)
    for (uint32_t k = 0; k + 1 < D; ++k) {
      T.push_back({G(0, j, k), G(0, j + 1, k + 1), G(0, j + 1, k)});
      T.push_back({G(0, j, k), G(0, j, k + 1), G(0, j + 1, k + 1)});
      T.push_back({G(W - 1, j, k), G(W - 1, j + 1, k), G(W - 1, j + 1, k + 1)});
      T.push_back({G(W - 1, j, k), G(W - 1, j + 1, k + 1), G(W - 1, j, k + 1)});
    }
}

extern "C" {

ora_solver* ora_create(const void* options) {
  ora_solver* s = new ora_solver();
  if (options) std::memcpy(&s->opt, options, sizeof(Options));
  return s;
}
void ora_destroy(ora_solver* s) { delete s; }
uint32_t ora_options_size() { return sizeof(Options); }

// flag: 0 = releaseHinge, 1 = nodeCollisions (extension), 2 = collision visiting rule (0 reference, 1 device)
void ora_set_flag(ora_solver* s, int flag, int value) {
  if (flag == 0) s->releaseHinge = value != 0;
  if (flag == 1) s->nodeCollisions = value != 0;
  if (flag == 2) { s->collisionRule = value; s->collisionOrder.clear(); }
  if (flag == 3) s->triangleCollisions = value != 0;
  if (flag == 4) s->solveFp64 = value != 0;  // FLAG_PD_SOLVE_FP64
  if (flag == 5) g_svd_plain = value != 0;   // FLAG_SVD_PLAIN (process-wide)
}
int ora_failed(ora_solver* s) { return s->simFailed ? 1 : 0; }

// generic node append (ids offset by current count, like every add*/create* in PrimitiveUtilities.cpp)
uint32_t ora_add_nodes_raw(ora_solver* s, uint32_t n, const float* pos, const float* vel, const float* radius,
                           const float* invMass) {
  uint32_t base = static_cast<uint32_t>(s->nodes.size());
  for (uint32_t i = 0; i < n; ++i) {
    Node node;
    node.id = base + i;
    node.position = vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
    node.prevPosition = node.position;
    node.velocity = vel ? vec3(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2]) : vec3(0.f);
    node.radius = radius ? radius[i] : 0.5f;
    node.invMass = invMass ? invMass[i] : 1.0f;
    s->nodes.push_back(node);
  }
  s->pdDirty = true;
  return base;
}
// Src/PrimitiveUtilities.cpp:42-75 : mass 1, radius 0.5, velocity 0
uint32_t ora_add_nodes(ora_solver* s, uint32_t n, const float* pos) {
  return ora_add_nodes_raw(s, n, pos, nullptr, nullptr, nullptr);
}

void ora_add_distance(ora_solver* s, uint32_t n, const uint32_t* ids, float w) {
  for (uint32_t i = 0; i < n; ++i)
    s->distanceCons.push_back(makeDistance(s->constraintId++, s->nodes[ids[2 * i]], s->nodes[ids[2 * i + 1]], w));
  s->pdDirty = true;
}
void ora_add_position(ora_solver* s, uint32_t n, const uint32_t* ids, float w) {
  for (uint32_t i = 0; i < n; ++i) s->positionCons.push_back(makePosition(s->constraintId++, s->nodes[ids[i]], w));
  s->pdDirty = true;
}
void ora_add_tet(ora_solver* s, uint32_t n, const uint32_t* ids, float w, float minStrain, float maxStrain) {
  for (uint32_t i = 0; i < n; ++i)
    s->tetCons.push_back(makeTet(s->constraintId++, w, s->nodes[ids[4 * i]], s->nodes[ids[4 * i + 1]],
                                 s->nodes[ids[4 * i + 2]], s->nodes[ids[4 * i + 3]], minStrain, maxStrain));
  s->pdDirty = true;
}
void ora_add_volume(ora_solver* s, uint32_t n, const uint32_t* ids, float w, float compression, float stretching) {
  for (uint32_t i = 0; i < n; ++i)
    s->volumeCons.push_back(makeVolume(s->constraintId++, w, s->nodes[ids[4 * i]], s->nodes[ids[4 * i + 1]],
                                       s->nodes[ids[4 * i + 2]], s->nodes[ids[4 * i + 3]], compression, stretching));
  s->pdDirty = true;
}
// extension: n node-node CollisionConstraints (CollisionConstraint.cpp:7-8: ids only; w stays the type's 1e5)
void ora_add_node_pairs(ora_solver* s, uint32_t n, const uint32_t* ids) {
  for (uint32_t i = 0; i < n; ++i) {
    NodePairCollision c;
    c.nodeIds[0] = ids[2 * i];
    c.nodeIds[1] = ids[2 * i + 1];
    s->nodePairs.push_back(c);
  }
}
void ora_add_bend(ora_solver* s, uint32_t n, const uint32_t* ids, float w) {
  for (uint32_t i = 0; i < n; ++i)
    s->bendCons.push_back(makeBend(s->constraintId++, w, s->nodes[ids[4 * i]], s->nodes[ids[4 * i + 1]],
                                   s->nodes[ids[4 * i + 2]], s->nodes[ids[4 * i + 3]]));
  s->pdDirty = true;
}
// material coordinates = current positions of the listed nodes (as addLinkedRegions / createShapeMatching* do)
void ora_add_shape(ora_solver* s, uint32_t n, const uint32_t* ids, float w) {
  std::vector<uint32_t> idx(ids, ids + n);
  std::vector<vec3> mc(n);
  for (uint32_t i = 0; i < n; ++i) mc[i] = s->nodes[ids[i]].position;
  ShapeCon c;
  c.init(s->nodes, idx, mc, w);
  s->shapeCons.push_back(std::move(c));
  s->pdDirty = true;
}
void ora_add_goal(ora_solver* s, uint32_t n, const uint32_t* ids, float w) {
  std::vector<uint32_t> idx(ids, ids + n);
  GoalCon c;
  c.init(s->nodes, idx, w);
  s->goalCons.push_back(std::move(c));
  s->pdDirty = true;
}
void ora_set_goal_transform(ora_solver* s, uint32_t goal, const float* m16) {
  std::memcpy(s->goalCons[goal].transform.m, m16, 16 * sizeof(float));
}

static bool inside_unit_box(const mat4& worldToRegion, const vec3& p) {
  float l[4];
  mul_point(worldToRegion, p, l);
  return -1.0f <= l[0] && l[0] <= 1.0f && -1.0f <= l[1] && l[1] <= 1.0f && -1.0f <= l[2] && l[2] <= 1.0f;
}
// Solver::addFixedRegions (PrimitiveUtilities.cpp:77-112)
void ora_add_fixed_regions(ora_solver* s, uint32_t n, const float* mats16, float w) {
  for (uint32_t k = 0; k < n; ++k) {
    mat4 regionToWorld;
    std::memcpy(regionToWorld.m, mats16 + 16 * k, 16 * sizeof(float));
    ora_solver::FixedRegion region;
    region.invInitialTransform = inverse(regionToWorld);
    region.goal = static_cast<uint32_t>(s->goalCons.size());
    std::vector<uint32_t> constrained;
    for (const Node& node : s->nodes)
      if (inside_unit_box(region.invInitialTransform, node.position)) constrained.push_back(node.id);
    GoalCon c;
    c.init(s->nodes, constrained, w);
    s->goalCons.push_back(std::move(c));
    s->fixedRegions.push_back(region);
  }
  s->pdDirty = true;
}
// Solver::updateFixedRegions (PrimitiveUtilities.cpp:114-128)
void ora_update_fixed_regions(ora_solver* s, uint32_t n, const float* mats16) {
  if (n != s->fixedRegions.size()) return;
  for (uint32_t i = 0; i < n; ++i) {
    mat4 cur;
    std::memcpy(cur.m, mats16 + 16 * i, 16 * sizeof(float));
    s->goalCons[s->fixedRegions[i].goal].transform = cur * s->fixedRegions[i].invInitialTransform;
  }
}
// Solver::addLinkedRegions (PrimitiveUtilities.cpp:130-162)
void ora_add_linked_regions(ora_solver* s, uint32_t n, const float* mats16, float w) {
  for (uint32_t k = 0; k < n; ++k) {
    mat4 region;
    std::memcpy(region.m, mats16 + 16 * k, 16 * sizeof(float));
    mat4 worldToRegion = inverse(region);
    std::vector<vec3> mc;
    std::vector<uint32_t> idx;
    for (const Node& node : s->nodes)
      if (inside_unit_box(worldToRegion, node.position)) { mc.push_back(node.position); idx.push_back(node.id); }
    if (mc.size() >= 3) {
      ShapeCon c;
      c.init(s->nodes, idx, mc, w);
      s->shapeCons.push_back(std::move(c));
    }
  }
  s->pdDirty = true;
}
// Solver::createShapeMatchingBox (PrimitiveUtilities.cpp:985-1048); scale is overridden to 0.5 (:995)
void ora_create_shape_matching_box(ora_solver* s, const float* tr, uint32_t cx, uint32_t cy, uint32_t cz, float w) {
  const float scale = 0.5f;
  size_t off = s->nodes.size();
  vec3 t(tr[0], tr[1], tr[2]);
  for (uint32_t i = 0; i < cx; ++i)
    for (uint32_t j = 0; j < cy; ++j)
      for (uint32_t k = 0; k < cz; ++k) {
        Node node;
        node.id = gid(cy, cz, off, i, j, k);
        node.position = scale * vec3(float(i), float(j), float(k)) + t;
        node.prevPosition = node.position;
        node.velocity = vec3(0.0f);
        node.radius = 0.5f * scale;
        node.invMass = 1.0f / 10.0f;
        s->nodes.push_back(node);
      }
  std::vector<uint32_t> idx(cx * cy * cz);
  std::vector<vec3> mc(idx.size());
  for (uint32_t i = 0; i < idx.size(); ++i) { idx[i] = static_cast<uint32_t>(off) + i; mc[i] = s->nodes[idx[i]].position; }
  ShapeCon c;
  c.init(s->nodes, idx, mc, w);
  s->shapeCons.push_back(std::move(c));
  s->pdDirty = true;
}
// Solver::createShapeMatchingSheet (PrimitiveUtilities.cpp:1050-1125), generalised to W x H (reference 50 x 50)
void ora_create_shape_matching_sheet(ora_solver* s, uint32_t W, uint32_t H, const float* tr, float scale, float w) {
  const uint32_t pw = 3, ph = 3;
  struct Patch { std::vector<vec3> mc; std::vector<uint32_t> idx; };
  std::vector<Patch> patches((W / pw) * (H / ph));
  size_t off = s->nodes.size();
  vec3 t(tr[0], tr[1], tr[2]);
  auto add = [&](uint32_t p, uint32_t id, const vec3& pos) { if (p < patches.size()) { patches[p].mc.push_back(pos); patches[p].idx.push_back(id); } };
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j) {
      Node node;
      node.id = gid(H, 1, off, i, j, 0);
      node.position = scale * vec3(float(i), float(j), 0.0f) + t;
      node.prevPosition = node.position;
      node.velocity = vec3(0.0f);
      node.radius = 0.5f * scale;
      node.invMass = 1.0f;
      s->nodes.push_back(node);
      add(i / pw * ph + j / ph, node.id, node.position);
      if ((i % pw) == (pw - 1) && i < (W - 1)) add((1 + i / pw) * ph + j / ph, node.id, node.position);
      if ((j % ph) == (ph - 1) && j < (H - 1)) add(i / pw * ph + j / ph + 1, node.id, node.position);
    }
  for (const Patch& p : patches) {
    if (p.idx.empty()) continue;  // the reference also builds constraints for empty patches; they touch no node
    ShapeCon c;
    c.init(s->nodes, p.idx, p.mc, w);
    s->shapeCons.push_back(std::move(c));
  }
  s->pdDirty = true;
}
// node ids of shape (5) / goal (6) constraint k
uint32_t ora_group_size(ora_solver* s, int type, uint32_t k) {
  return type == 5 ? (uint32_t)s->shapeCons[k].ids.size() : (uint32_t)s->goalCons[k].ids.size();
}
void ora_group_ids(ora_solver* s, int type, uint32_t k, uint32_t* out) {
  const auto& ids = type == 5 ? s->shapeCons[k].ids : s->goalCons[k].ids;
  std::copy(ids.begin(), ids.end(), out);
}
void ora_add_triangles(ora_solver* s, uint32_t n, const uint32_t* ids) {
  for (uint32_t i = 0; i < n; ++i) s->triangles.push_back({ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]});
}

// Generalised Solver::createTetBox (PrimitiveUtilities.cpp:330-618) on a W x H x D lattice.
// flags bit0: also push VolumeConstraints (the reference always does, :408-414); bit1: triangles.
void ora_create_tet_box(ora_solver* s, uint32_t W, uint32_t H, uint32_t D, const float* tr, float scale,
                        const float* vel, float w, float mass, uint32_t flags) {
  size_t off = s->nodes.size();
  vec3 t(tr[0], tr[1], tr[2]), v(vel[0], vel[1], vel[2]);
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j)
      for (uint32_t k = 0; k < D; ++k) {
        Node node;
        node.id = gid(H, D, off, i, j, k);
        node.position = scale * vec3(float(i), float(j), float(k)) + t;
        node.prevPosition = node.position;
        node.velocity = v;
        node.radius = 0.95f * 0.5f * scale;
        node.invMass = 1.0f / mass;
        s->nodes.push_back(node);
      }
  auto G = [&](uint32_t x, uint32_t y, uint32_t z) { return gid(H, D, off, x, y, z); };
  for (uint32_t i = 0; i + 1 < W; ++i)
    for (uint32_t j = 0; j + 1 < H; ++j)
      for (uint32_t k = 0; k + 1 < D; ++k) {
        uint32_t n000 = G(i, j, k), n001 = G(i, j, k + 1), n010 = G(i, j + 1, k), n011 = G(i, j + 1, k + 1);
        uint32_t n100 = G(i + 1, j, k), n101 = G(i + 1, j, k + 1), n110 = G(i + 1, j + 1, k), n111 = G(i + 1, j + 1, k + 1);
        const uint32_t q[6][4] = {{n000, n001, n011, n111}, {n000, n010, n011, n111}, {n000, n001, n101, n111},
                                  {n000, n100, n101, n111}, {n000, n010, n110, n111}, {n000, n100, n110, n111}};
        for (int e = 0; e < 6; ++e) {
          auto& N = s->nodes;
          s->tetCons.push_back(makeTet(s->constraintId++, w, N[q[e][0]], N[q[e][1]], N[q[e][2]], N[q[e][3]], 0.8f, 1.0f));
          if (flags & 1u)
            s->volumeCons.push_back(makeVolume(s->constraintId++, w, N[q[e][0]], N[q[e][1]], N[q[e][2]], N[q[e][3]], 1.0f, 1.0f));
          else
            s->constraintId++;
          s->tets.push_back({q[e][0], q[e][1], q[e][2], q[e][3]});
        }
      }
  if (flags & 2u) boxSurface(s, W, H, D, off);
  s->pdDirty = true;
}

// Generalised Solver::createBox (PrimitiveUtilities.cpp:620-847): distance constraints only.
// mode 0: new nodes (radius 0.5*scale, invMass 1); mode 1: constraints over an existing lattice at `off`.
void ora_create_box(ora_solver* s, uint32_t W, uint32_t H, uint32_t D, const float* tr, float scale, float w,
                    int existing, uint32_t existingOff, uint32_t flags) {
  size_t off = existing ? existingOff : s->nodes.size();
  if (!existing) {
    vec3 t(tr[0], tr[1], tr[2]);
    for (uint32_t i = 0; i < W; ++i)
      for (uint32_t j = 0; j < H; ++j)
        for (uint32_t k = 0; k < D; ++k) {
          Node node;
          node.id = gid(H, D, off, i, j, k);
          node.position = scale * vec3(float(i), float(j), float(k)) + t;
          node.prevPosition = node.position;
          node.velocity = vec3(0.0f);
          node.radius = 0.5f * scale;
          node.invMass = 1.0f;
          s->nodes.push_back(node);
        }
  }
  auto G = [&](uint32_t x, uint32_t y, uint32_t z) { return gid(H, D, off, x, y, z); };
  size_t firstDist = s->distanceCons.size();
  auto push = [&](uint32_t a, uint32_t b) {
    s->distanceCons.push_back(makeDistance(s->constraintId++, s->nodes[a], s->nodes[b], w));
  };
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j)
      for (uint32_t k = 0; k < D; ++k) {
        if (i < W - 1) push(G(i, j, k), G(i + 1, j, k));
        if (j < H - 1) push(G(i, j, k), G(i, j + 1, k));
        if (k < D - 1) push(G(i, j, k), G(i, j, k + 1));
        if (i < W - 1 && j < H - 1 && k < D - 1) {
          push(G(i, j, k), G(i + 1, j + 1, k + 1));
          push(G(i + 1, j, k), G(i, j + 1, k + 1));
          push(G(i, j + 1, k), G(i + 1, j, k + 1));
          push(G(i, j, k + 1), G(i + 1, j + 1, k));
        }
      }
  if (flags & 2u) boxSurface(s, W, H, D, off);
  for (size_t i = firstDist; i < s->distanceCons.size(); ++i) {
    s->lines.push_back(s->distanceCons[i].nodeIds[0]);
    s->lines.push_back(s->distanceCons[i].nodeIds[1]);
  }
  s->pdDirty = true;
}

// Solver::createSheet (PrimitiveUtilities.cpp:849-976), generalised to W x H.
void ora_create_sheet(ora_solver* s, uint32_t W, uint32_t H, const float* tr, float scale, float mass, float w) {
  size_t off = s->nodes.size();
  vec3 t(tr[0], tr[1], tr[2]);
  auto G = [&](uint32_t x, uint32_t y) { return gid(H, 1, off, x, y, 0); };
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j) {
      Node node;
      node.id = G(i, j);
      node.position = scale * vec3(float(i), 0.f, float(j)) + t;
      node.prevPosition = node.position;
      node.velocity = vec3(0.0f);
      node.radius = 0.5f * scale;
      node.invMass = 1.0f / mass;
      s->nodes.push_back(node);
      if (i == 0 || i == W - 1 || j == 0 || j == H - 1)
        s->positionCons.push_back(makePosition(s->constraintId++, s->nodes.back(), w));
    }
  size_t firstDist = s->distanceCons.size();
  auto push = [&](uint32_t a, uint32_t b) {
    s->distanceCons.push_back(makeDistance(s->constraintId++, s->nodes[a], s->nodes[b], w));
  };
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j) {
      if (i < W - 1) push(G(i, j), G(i + 1, j));
      if (j < H - 1) push(G(i, j), G(i, j + 1));
      if (i < W - 1 && j < H - 1) {
        push(G(i, j), G(i + 1, j + 1));
        push(G(i + 1, j), G(i, j + 1));
      }
    }
  for (uint32_t i = 0; i + 1 < W; ++i)
    for (uint32_t j = 0; j + 1 < H; ++j) {
      s->triangles.push_back({G(i, j), G(i + 1, j + 1), G(i + 1, j)});
      s->triangles.push_back({G(i, j), G(i, j + 1), G(i + 1, j + 1)});
    }
  for (size_t i = firstDist; i < s->distanceCons.size(); ++i) {
    s->lines.push_back(s->distanceCons[i].nodeIds[0]);
    s->lines.push_back(s->distanceCons[i].nodeIds[1]);
  }
  s->pdDirty = true;
}

// Solver::createBendSheet (PrimitiveUtilities.cpp:1127-1289), generalised to W x H.
void ora_create_bend_sheet(ora_solver* s, uint32_t W, uint32_t H, const float* tr, float scale, float w) {
  size_t off = s->nodes.size();
  vec3 t(tr[0], tr[1], tr[2]);
  auto G = [&](uint32_t x, uint32_t y) { return gid(H, 1, off, x, y, 0); };
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j) {
      Node node;
      node.id = G(i, j);
      node.position = scale * vec3(float(i), 0.f, float(j)) + t;
      node.prevPosition = node.position;
      node.velocity = vec3(0.0f);
      node.radius = 0.5f * scale;
      node.invMass = 1.0f;
      s->nodes.push_back(node);
      if (i < 3) s->positionCons.push_back(makePosition(s->constraintId++, s->nodes.back(), w));
    }
  size_t firstDist = s->distanceCons.size();
  auto push = [&](uint32_t a, uint32_t b) {
    s->distanceCons.push_back(makeDistance(s->constraintId++, s->nodes[a], s->nodes[b], w));
  };
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j) {
      if (i < W - 1) push(G(i, j), G(i + 1, j));
      if (j < H - 1) push(G(i, j), G(i, j + 1));
      if (i < W - 1 && j < H - 1) push(G(i, j), G(i + 1, j + 1));
    }
  auto& N = s->nodes;
  for (uint32_t i = 0; i < W; ++i)
    for (uint32_t j = 0; j < H; ++j) {
      if (i < W - 1 && j < H - 1)
        s->bendCons.push_back(makeBend(s->constraintId++, w, N[G(i, j)], N[G(i + 1, j + 1)], N[G(i + 1, j)], N[G(i, j + 1)]));
      if (i + 2 < W && j + 2 < H) {
        s->bendCons.push_back(makeBend(s->constraintId++, w, N[G(i + 1, j)], N[G(i + 1, j + 1)], N[G(i, j)], N[G(i + 2, j + 1)]));
        s->bendCons.push_back(makeBend(s->constraintId++, w, N[G(i, j + 1)], N[G(i + 1, j + 1)], N[G(i, j)], N[G(i + 1, j + 2)]));
      }
    }
  for (uint32_t i = 0; i + 1 < W; ++i)
    for (uint32_t j = 0; j + 1 < H; ++j) {
      s->triangles.push_back({G(i, j), G(i + 1, j + 1), G(i + 1, j)});
      s->triangles.push_back({G(i, j), G(i, j + 1), G(i + 1, j + 1)});
    }
  for (size_t i = firstDist; i < s->distanceCons.size(); ++i) {
    s->lines.push_back(s->distanceCons[i].nodeIds[0]);
    s->lines.push_back(s->distanceCons[i].nodeIds[1]);
  }
  s->pdDirty = true;
}

// Reorder a constraint container: new[i] = old[perm[i]].  type: 0 position, 1 distance, 2 tet,
// 3 volume, 4 bend.  Sequential Gauss-Seidel over the permuted container is what a coloured device
// schedule computes, so the oracle replays device schedules through this call.
void ora_permute(ora_solver* s, int type, const uint32_t* perm, uint32_t n) {
  auto apply = [&](auto& v) {
    assert(v.size() == n);
    auto old = v;
    for (uint32_t i = 0; i < n; ++i) v[i] = old[perm[i]];
  };
  switch (type) {
    case 0: apply(s->positionCons); break;
    case 1: apply(s->distanceCons); break;
    case 2: apply(s->tetCons); break;
    case 3: apply(s->volumeCons); break;
    case 4: apply(s->bendCons); break;
  }
  if (type >= 0 && type <= 4) s->batchOffs[type].clear();  // batches describe an order: gone with it
  s->pdDirty = true;
}
// Conflict-free batches of a container (after ora_permute): n_batches + 1 slot offsets.  Used only with ora_set_threads > 1.
void ora_set_batches(ora_solver* s, int type, const uint32_t* offs, uint32_t n_batches) {
  if (type < 0 || type > 4) return;
  s->batchOffs[type].assign(offs, offs + (n_batches ? n_batches + 1 : 0));
}
void ora_set_threads(ora_solver* s, int threads) { s->threads = threads < 1 ? 1 : threads; }
void ora_set_reference_threads(ora_solver* s, int on) {
  s->referenceThreads = on != 0;
  s->hashNodes.threads = on ? 16 : 1;
}
void ora_set_collision_order(ora_solver* s, const uint32_t* order, uint32_t n) {
  s->collisionOrder.assign(order, order + n);
}

uint32_t ora_count(ora_solver* s, int what) {
  switch (what) {
    case 0: return (uint32_t)s->positionCons.size();
    case 1: return (uint32_t)s->distanceCons.size();
    case 2: return (uint32_t)s->tetCons.size();
    case 3: return (uint32_t)s->volumeCons.size();
    case 4: return (uint32_t)s->bendCons.size();
    case 5: return (uint32_t)s->shapeCons.size();
    case 6: return (uint32_t)s->goalCons.size();
    case 7: return (uint32_t)s->triangles.size();
    case 8: return (uint32_t)s->lines.size();
    case 9: return (uint32_t)s->nodes.size();
    case 10: return (uint32_t)s->staticCollisions.size();
    case 11: return (uint32_t)s->triCollisions.size();
    case 18: return (uint32_t)s->nodePairs.size();
  }
  return 0;
}
uint64_t ora_stat_collision_pairs(ora_solver* s) { return s->stat_collision_pairs; }

// what: 0 position, 1 prevPosition, 2 velocity (n x 3), 3 radius, 4 invMass (n)
void ora_get(ora_solver* s, int what, float* out) {
  for (size_t i = 0; i < s->nodes.size(); ++i) {
    const Node& n = s->nodes[i];
    const vec3* v = what == 0 ? &n.position : what == 1 ? &n.prevPosition : &n.velocity;
    if (what <= 2) { out[3 * i] = v->x; out[3 * i + 1] = v->y; out[3 * i + 2] = v->z; }
    else out[i] = what == 3 ? n.radius : n.invMass;
  }
}
void ora_set(ora_solver* s, int what, const float* in) {
  for (size_t i = 0; i < s->nodes.size(); ++i) {
    Node& n = s->nodes[i];
    if (what <= 2) {
      vec3 v(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
      if (what == 0) n.position = v; else if (what == 1) n.prevPosition = v; else n.velocity = v;
    } else if (what == 3) n.radius = in[i];
    else { n.invMass = in[i]; s->pdDirty = true; }
  }
}
// constraint node ids, flattened; type as in ora_permute, 7 = triangles, 8 = lines
void ora_get_ids(ora_solver* s, int type, uint32_t* out) {
  size_t k = 0;
  auto dump = [&](auto& v) { for (auto& c : v) for (uint32_t id : c.nodeIds) out[k++] = id; };
  switch (type) {
    case 0: dump(s->positionCons); break;
    case 1: dump(s->distanceCons); break;
    case 2: dump(s->tetCons); break;
    case 3: dump(s->volumeCons); break;
    case 4: dump(s->bendCons); break;
    case 7: for (auto& t : s->triangles) for (uint32_t id : t) out[k++] = id; break;
    case 8: for (uint32_t id : s->lines) out[k++] = id; break;
  }
}
// per-constraint rest data: distance -> target (1), tet/volume -> Qinv as glm column-major (9)
void ora_get_rest(ora_solver* s, int type, float* out) {
  size_t k = 0;
  if (type == 1) for (auto& c : s->distanceCons) out[k++] = c.targetDistance;
  if (type == 2) for (auto& c : s->tetCons) for (int col = 0; col < 3; ++col) for (int r = 0; r < 3; ++r) out[k++] = c.Qinv[col][r];
  if (type == 3) for (auto& c : s->volumeCons) for (int col = 0; col < 3; ++col) for (int r = 0; r < 3; ++r) out[k++] = c.Qinv[col][r];
  if (type == 4) for (auto& c : s->bendCons) out[k++] = c.initialAngle;
}

void ora_tick(ora_solver* s) { s->tick(); }

// ------------------------------- single-operation entry points (KATs) ------------------------
// a: row-major 3x3; out: row-major U*diag(snew)*V^T with snew = clamp(s) (+ flip) as the tet functor
void ora_svd_stats(uint64_t* out, int reset) {
  for (int i = 0; i < 8; ++i) out[i] = g_svd_stats[i];
  if (reset) for (int i = 0; i < 8; ++i) g_svd_stats[i] = 0;
}
int ora_svd3(const float* a, float* s_out, float* b_out, float* v_out) {
  float A[3][3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[r][c] = a[3 * r + c];
  Svd3 d = svd3(A);
  for (int i = 0; i < 3; ++i) {
    s_out[i] = d.s[i];
    for (int k = 0; k < 3; ++k) { b_out[3 * i + k] = d.b[i][k]; v_out[3 * i + k] = d.v[i][k]; }
  }
  return d.sweeps;
}
// out (row-major) = U * diag(snew) * V^T for the decomposition of a
void ora_svd3_recompose(const float* a, const float* snew, float* out) {
  float A[3][3], O[3][3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[r][c] = a[3 * r + c];
  Svd3 d = svd3(A);
  svd3_recompose(d, snew, O);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out[3 * r + c] = O[r][c];
}
static std::vector<Node> four(const float* x) {
  std::vector<Node> n(4);
  for (int i = 0; i < 4; ++i) { n[i].id = i; n[i].position = vec3(x[3 * i], x[3 * i + 1], x[3 * i + 2]); }
  return n;
}
// x: 4 positions; qinv: glm column-major 3x3; out: 4 projected vec3
void ora_project_tet(const float* x, const float* qinv, float minStrain, float maxStrain, float* out) {
  auto n = four(x);
  TetCon c;
  c.nodeIds = {0, 1, 2, 3};
  for (int col = 0; col < 3; ++col) for (int r = 0; r < 3; ++r) c.Qinv[col][r] = qinv[3 * col + r];
  c.minStrain = minStrain; c.maxStrain = maxStrain;
  std::array<vec3, 4> p;
  c.project(n, p);
  for (int i = 0; i < 4; ++i) { out[3 * i] = p[i].x; out[3 * i + 1] = p[i].y; out[3 * i + 2] = p[i].z; }
}
void ora_project_volume(const float* x, const float* qinv, float minOmega, float maxOmega, float* out) {
  auto n = four(x);
  VolumeCon c;
  c.nodeIds = {0, 1, 2, 3};
  for (int col = 0; col < 3; ++col) for (int r = 0; r < 3; ++r) c.Qinv[col][r] = qinv[3 * col + r];
  c.minOmega = minOmega; c.maxOmega = maxOmega;
  std::array<vec3, 4> p;
  c.project(n, p);
  for (int i = 0; i < 4; ++i) { out[3 * i] = p[i].x; out[3 * i + 1] = p[i].y; out[3 * i + 2] = p[i].z; }
}
void ora_project_distance(const float* x, float target, float* out) {
  std::vector<Node> n(2);
  for (int i = 0; i < 2; ++i) { n[i].id = i; n[i].position = vec3(x[3 * i], x[3 * i + 1], x[3 * i + 2]); }
  DistanceCon c;
  c.nodeIds = {0, 1};
  c.targetDistance = target;
  std::array<vec3, 2> p;
  c.project(n, p);
  for (int i = 0; i < 2; ++i) { out[3 * i] = p[i].x; out[3 * i + 1] = p[i].y; out[3 * i + 2] = p[i].z; }
}
void ora_project_bend(const float* x, const float* invMass, float angle, float* out) {
  auto n = four(x);
  for (int i = 0; i < 4; ++i) n[i].invMass = invMass[i];
  BendCon c;
  c.nodeIds = {0, 1, 2, 3};
  c.initialAngle = angle;
  std::array<vec3, 4> p;
  c.project(n, p);
  for (int i = 0; i < 4; ++i) { out[3 * i] = p[i].x; out[3 * i + 1] = p[i].y; out[3 * i + 2] = p[i].z; }
}
// rest-state helpers for KATs: Qinv (glm column-major) and AtA (row-major 4x4) of a tet
void ora_tet_rest(const float* x, float* qinv, float* AtA) {
  auto n = four(x);
  TetCon c = makeTet(0, 1.0f, n[0], n[1], n[2], n[3], 0.8f, 1.0f);
  for (int col = 0; col < 3; ++col) for (int r = 0; r < 3; ++r) qinv[3 * col + r] = c.Qinv[col][r];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) AtA[4 * i + j] = c.AtA[i][j];
}
// CollisionDetection.cpp:227-302; v: ap0, ab0, ac0, ap1, ab1, ac1 (18 floats); returns 1 and *t on contact
int ora_point_triangle_ccd(const float* v, float threshold, float* t) {
  auto V = [&](int k) { return vec3(v[3 * k], v[3 * k + 1], v[3 * k + 2]); };
  float tt = -1.0f;
  bool hit = pointTriangleCCD(V(0), V(1), V(2), V(3), V(4), V(5), threshold, tt);
  *t = tt;
  return hit ? 1 : 0;
}
// contacts of the last detection: n x 4 node ids (a, b, c, d)
void ora_get_tri_collisions(ora_solver* s, uint32_t* out) {
  size_t k = 0;
  for (const TriCollision& c : s->triCollisions) for (uint32_t id : c.nodeIds) out[k++] = id;
}
// Src/Solver.cpp:877-901 : out = {minX,minY,minZ,lenX,lenY,lenZ}
void ora_node_range(const float* pos, float radius, float scale, int64_t* out) {
  Node n;
  n.position = vec3(pos[0], pos[1], pos[2]);
  n.radius = radius;
  CellRange r = nodeCompRange(n, scale);
  out[0] = r.minX; out[1] = r.minY; out[2] = r.minZ;
  out[3] = r.lengthX; out[4] = r.lengthY; out[5] = r.lengthZ;
}

}  // extern "C"
