// Pies::Solver -- drop-in for the reference's public class (Include/Pies/Solver.h:21-116), implemented on
// the MI355X through the C ABI in pies_hip.h (link libpies_hip.so).  Same names, signatures, defaults and
// ownership rules; hosts (PiesForAlthea / PiesForMaya) recompile against this header.
//
//   Pies::SolverOptions opt; opt.solver = Pies::SolverName::PBD;
//   Pies::Solver solver(opt);
//   solver.createTetBox({0, 5, 0}, 1.0f, {0, 0, 0}, 0.05f, 1.0f, false);
//   solver.tick(dt);                       // positions are current in getVertices() when this returns
//
// Differences from the reference, all documented in DESIGN.md: tick() runs on the GPU; getters return
// host mirrors refreshed once per tick (Solver.cpp:157,393 refresh per substep, which nothing can observe);
// addTriMeshVolume needs tetgen, which is not part of this build (addTetMeshVolume takes tetgen's output arrays, or
// just elements, and does what addTriMeshVolume does after its tetrahedralize() call); errors surface as std::runtime_error carrying pies_last_error().
// Shape/goal matching (regions, createShapeMatching*) are Projective-Dynamics constraints, as in the reference.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../pies_hip.h"
#include "Node.h"
#include "Tetrahedron.h"
#include "Triangle.h"
#include "glm_compat.h"

namespace Pies {
enum class SolverName { PBD, PD };  // Solver.h:21

struct SolverOptions {  // Solver.h:23-38, field for field
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
  SolverName solver = SolverName::PD;
};
static_assert(sizeof(SolverOptions) == sizeof(pies_options_t), "SolverOptions must mirror pies_options_t");

class Solver {
public:
  struct Vertex {  // Solver.h:42-49: consumed directly as a vertex/instance stream by the renderers
    glm::vec3 position{};
    float radius{};
    glm::vec3 baseColor{};
    float roughness{};
    float metallic{};
  };

  bool renderStateDirty = true;
  bool releaseHinge = false;
  // extensions (not in the reference): which device schedule maps the Gauss-Seidel sweeps, and whether the
  // PBD node-node pass runs (the reference always runs it).  PIES_SCHEDULE_EXACT sweeps the containers in the order
  // the constraints were added and the colliding nodes in ascending index - the reference's order, bit for bit the
  // sequential loop; the default (PIES_SCHEDULE_DEFAULT = LAYERED) is the fast one bench.py reports, a sweep over the
  // same constraints in another order (pies_hip.h; DESIGN.md section 3 states how far apart the results are).
  // -1 (default) leaves the handle's own choice alone: PIES_SCHEDULE_DEFAULT, or what PIES_SCHEDULE=exact|coloured|layered
  // in the environment selected when the handle was opened.
  int schedule = -1;
  bool nodeCollisions = true;       // PBD node-node pass (Solver.cpp:81-130)
  bool triangleCollisions = true;   // PD point-triangle contacts (Solver.cpp:693-797)

  // Like the reference's (Solver.h:54-55: `Solver() = default`, a converting constructor from the options): constructing a
  // Solver touches no device.  The device handle is opened by the first call that needs it (add*/create*/tick), which is
  // where a missing gfx950 device surfaces (std::runtime_error; there is no CPU fallback).
  Solver() = default;
  Solver(const SolverOptions& options, int device = 0) : _options(options), _device(device) {}
  Solver(Solver&& rhs) noexcept { *this = std::move(rhs); }
  Solver& operator=(Solver&& rhs) noexcept {
    if (this != &rhs) {
      if (_h) pies_destroy(_h);
      _h = rhs._h;
      rhs._h = nullptr;
      _options = rhs._options;
      _device = rhs._device;
      for (int k = 0; k < 5; ++k) _pushed[k] = rhs._pushed[k];
      _vertices = std::move(rhs._vertices);
      _lines = std::move(rhs._lines);
      _triangles = std::move(rhs._triangles);
      _frames[0] = rhs._frames[0];
      _frames[1] = rhs._frames[1];
      renderStateDirty = rhs.renderStateDirty;
      releaseHinge = rhs.releaseHinge;
      schedule = rhs.schedule;
      nodeCollisions = rhs.nodeCollisions;
      triangleCollisions = rhs.triangleCollisions;
    }
    return *this;
  }
  Solver(const Solver&) = delete;
  Solver& operator=(const Solver&) = delete;
  ~Solver() {
    if (_h) pies_destroy(_h);
  }

  // Solver.cpp:25-38.  The argument is ignored like in the reference (the step is fixedTimestepSize).
  void tick(float /*deltaTime*/) { _tickAs(_options.solver); }
  // Solver.cpp:40, :162: the reference's tickPBD / tickPD run the NAMED solver whatever SolverOptions::solver says.  So do
  // these: the handle is switched to that solver (pies_set_solver: the device buffers are rebuilt on the next tick if it
  // differs from the one the last tick ran) and stays there until another tick* asks for the other one.
  void tickPBD(float /*deltaTime*/) { _tickAs(SolverName::PBD); }
  void tickPD(float /*deltaTime*/) { _tickAs(SolverName::PD); }
  // tick() in two halves, so that a host can do its own work (or begin the next tick) while this one's positions travel:
  // beginTick queues the substeps and the asynchronous copy of their result, endTick waits for that copy and refreshes
  // getVertices().  At most two ticks may be begun before the first is ended (pies_tick_begin in pies_hip.h).
  void beginTick() {
    _pushState(_options.solver);
    uint64_t f = 0;
    _ck(pies_tick_begin(_handle(), &f));
    if (_frames[0] == 0) _frames[0] = f; else _frames[1] = f;
  }
  void endTick() {
    if (_frames[0] == 0) return;
    const uint64_t f = _frames[0];
    _frames[0] = _frames[1];
    _frames[1] = 0;
    const float* p = nullptr;
    uint32_t n = 0;
    _ck(pies_export_acquire(_handle(), f, &p, &n));
    const size_t m = n < _vertices.size() ? n : _vertices.size();
    for (size_t i = 0; i < m; ++i) _vertices[i].position = glm::vec3(p[4 * i], p[4 * i + 1], p[4 * i + 2]);
    _ck(pies_export_release(_handle(), f));
  }

  const std::vector<Vertex>& getVertices() const { return _vertices; }
  const std::vector<uint32_t>& getLines() const { return _lines; }
  const std::vector<Triangle>& getTriangles() const { return _triangles; }
  const SolverOptions& getOptions() const { return _options; }

  void clear() {
    if (_h) _ck(pies_clear(_h));
    _vertices.clear();
    _lines.clear();
    _triangles.clear();
    renderStateDirty = true;
  }

  // ---- importing meshes (PrimitiveUtilities.cpp:42-328) ----
  void addNodes(const std::vector<glm::vec3>& vertices) {
    std::vector<float> p = _flatten(vertices);
    _ck(pies_add_nodes(_handle(), static_cast<uint32_t>(vertices.size()), p.data(), nullptr));
    _syncRenderState();
  }
  void addTriMeshVolume(const std::vector<glm::vec3>&, const std::vector<uint32_t>&, const glm::vec3&, float, float, float, float,
                        float, float, float) {
    throw std::runtime_error("Pies::Solver::addTriMeshVolume needs tetgen (not part of this build): tetrahedralise offline and "
                             "call addTetMeshVolume");
  }
  // addTriMeshVolume after its tetrahedralize() call (PrimitiveUtilities.cpp:243-328), for hosts that mesh offline:
  // `vertices`, `tetIndices` (4 per element), `triFaceIndices` (3 per face) and `face2tet` (2 per face, -1 = no element on
  // that side) are tetgen's pointlist, tetrahedronlist, trifacelist and face2tetlist.  Like the reference: a face with an
  // element on both sides is skipped, boundary faces become triangles with the winding SWITCHED to (v0, v2, v1) so that
  // normals point outward (:255-266); mass = density, radius 0.5 (:176-177, :281-282); one strain and one volume
  // constraint per element when the respective stiffness is not 0 (:293-315).  An empty `face2tet` treats every face as a
  // boundary face.
  void addTetMeshVolume(const std::vector<glm::vec3>& vertices, const std::vector<uint32_t>& tetIndices,
                        const std::vector<uint32_t>& triFaceIndices, const std::vector<int>& face2tet, const glm::vec3& initialVelocity,
                        float density, float strainStiffness, float minStrain, float maxStrain, float volumeStiffness, float compression,
                        float stretching) {
    std::vector<uint32_t> surface;
    const size_t nf = triFaceIndices.size() / 3;
    for (size_t i = 0; i < nf; ++i) {
      if (!face2tet.empty() && face2tet[2 * i] >= 0 && face2tet[2 * i + 1] >= 0) continue;
      surface.push_back(triFaceIndices[3 * i]);
      surface.push_back(triFaceIndices[3 * i + 2]);  // switched winding
      surface.push_back(triFaceIndices[3 * i + 1]);
    }
    _addTetMesh(vertices, tetIndices, surface, initialVelocity, density, strainStiffness, minStrain, maxStrain, volumeStiffness,
                compression, stretching);
  }
  // The same without a face list: the boundary is derived from the elements (a face that belongs to exactly one element,
  // elements in order, faces opposite vertex 0, 1, 2, 3) and every triangle is wound so that its normal points away from the
  // element's fourth vertex, i.e. outward.
  void addTetMeshVolume(const std::vector<glm::vec3>& vertices, const std::vector<uint32_t>& tetIndices, const glm::vec3& initialVelocity,
                        float density, float strainStiffness, float minStrain, float maxStrain, float volumeStiffness, float compression,
                        float stretching) {
    struct Face { uint32_t key[3]; uint32_t tet, opposite; };
    static const int kFace[4][3] = {{1, 2, 3}, {0, 3, 2}, {0, 1, 3}, {0, 2, 1}};
    std::vector<Face> faces;
    const size_t nt = tetIndices.size() / 4;
    for (size_t t = 0; t < nt; ++t)
      for (uint32_t f = 0; f < 4; ++f) {
        Face fc{{tetIndices[4 * t + kFace[f][0]], tetIndices[4 * t + kFace[f][1]], tetIndices[4 * t + kFace[f][2]]}, static_cast<uint32_t>(t), f};
        if (fc.key[0] > fc.key[1]) std::swap(fc.key[0], fc.key[1]);
        if (fc.key[1] > fc.key[2]) std::swap(fc.key[1], fc.key[2]);
        if (fc.key[0] > fc.key[1]) std::swap(fc.key[0], fc.key[1]);
        faces.push_back(fc);
      }
    std::vector<size_t> order(faces.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    auto less = [&](size_t a, size_t b) {
      for (int k = 0; k < 3; ++k)
        if (faces[a].key[k] != faces[b].key[k]) return faces[a].key[k] < faces[b].key[k];
      return a < b;
    };
    std::sort(order.begin(), order.end(), less);
    std::vector<char> boundary(faces.size(), 0);
    for (size_t i = 0; i < order.size();) {
      size_t j = i + 1;
      while (j < order.size() && faces[order[i]].key[0] == faces[order[j]].key[0] && faces[order[i]].key[1] == faces[order[j]].key[1] &&
             faces[order[i]].key[2] == faces[order[j]].key[2])
        ++j;
      if (j == i + 1) boundary[order[i]] = 1;
      i = j;
    }
    std::vector<uint32_t> surface;
    for (size_t i = 0; i < faces.size(); ++i) {
      if (!boundary[i]) continue;
      const uint32_t t = faces[i].tet, f = faces[i].opposite;
      uint32_t a = tetIndices[4 * t + kFace[f][0]], b = tetIndices[4 * t + kFace[f][1]], c = tetIndices[4 * t + kFace[f][2]];
      const glm::vec3 &pa = vertices[a], &pb = vertices[b], &pc = vertices[c], &pd = vertices[tetIndices[4 * t + f]];
      const float ux = pb[0] - pa[0], uy = pb[1] - pa[1], uz = pb[2] - pa[2], vx = pc[0] - pa[0], vy = pc[1] - pa[1], vz = pc[2] - pa[2];
      const float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
      if (nx * (pd[0] - pa[0]) + ny * (pd[1] - pa[1]) + nz * (pd[2] - pa[2]) > 0.0f) std::swap(b, c);  // normal towards the inside: flip
      surface.push_back(a); surface.push_back(b); surface.push_back(c);
    }
    _addTetMesh(vertices, tetIndices, surface, initialVelocity, density, strainStiffness, minStrain, maxStrain, volumeStiffness,
                compression, stretching);
  }
  void addFixedRegions(const std::vector<glm::mat4>& regionMatrices, float w) {
    std::vector<float> m = _flattenMats(regionMatrices);
    _ck(pies_add_fixed_regions(_handle(), static_cast<uint32_t>(regionMatrices.size()), m.data(), w));
  }
  void updateFixedRegions(const std::vector<glm::mat4>& regionMatrices) {
    std::vector<float> m = _flattenMats(regionMatrices);
    // the reference asserts and returns on a size mismatch (PrimitiveUtilities.cpp:115-118); here it throws
    _ck(pies_update_fixed_regions(_handle(), static_cast<uint32_t>(regionMatrices.size()), m.data()));
  }
  void addLinkedRegions(const std::vector<glm::mat4>& regionMatrices, float w) {
    std::vector<float> m = _flattenMats(regionMatrices);
    _ck(pies_add_linked_regions(_handle(), static_cast<uint32_t>(regionMatrices.size()), m.data(), w));
  }

  // ---- primitives (PrimitiveUtilities.cpp:330-1289), reference grid sizes ----
  void createBox(const glm::vec3& translation, float scale, float w) {
    const float t[3] = {translation[0], translation[1], translation[2]};
    _ck(pies_create_box(_handle(), 5, 5, 5, t, scale, w, 0, 0, 2u));
    _syncRenderState();
  }
  void createTetBox(const glm::vec3& translation, float scale, const glm::vec3& initialVelocity, float w, float mass, bool hinged) {
    const float t[3] = {translation[0], translation[1], translation[2]}, v[3] = {initialVelocity[0], initialVelocity[1], initialVelocity[2]};
    if (hinged) _ck(pies_create_tet_box(_handle(), 10, 2, 10, t, scale, v, w, mass, 3u));
    else _ck(pies_create_tet_box(_handle(), 3, 3, 3, t, scale, v, w, mass, 3u));
    _syncRenderState();
  }
  void createSheet(const glm::vec3& translation, float scale, float mass, float k) {
    const float t[3] = {translation[0], translation[1], translation[2]};
    _ck(pies_create_sheet(_handle(), 20, 20, t, scale, mass, k));
    _syncRenderState();
  }
  // like the reference, `scale` and `initialVelocity` are accepted and ignored (PrimitiveUtilities.cpp:995,1015)
  void createShapeMatchingBox(const glm::vec3& translation, uint32_t countX, uint32_t countY, uint32_t countZ, float /*scale*/,
                              const glm::vec3& /*initialVelocity*/, float w) {
    const float t[3] = {translation[0], translation[1], translation[2]};
    _ck(pies_create_shape_matching_box(_handle(), t, countX, countY, countZ, w));
    _syncRenderState();
  }
  void createShapeMatchingSheet(const glm::vec3& translation, float scale, const glm::vec3& /*initialVelocity*/, float w) {
    const float t[3] = {translation[0], translation[1], translation[2]};
    _ck(pies_create_shape_matching_sheet(_handle(), 50, 50, t, scale, w));
    _syncRenderState();
  }
  void createBendSheet(const glm::vec3& translation, float scale, float w) {
    const float t[3] = {translation[0], translation[1], translation[2]};
    _ck(pies_create_bend_sheet(_handle(), 10, 10, t, scale, w));
    _syncRenderState();
  }

  pies_solver_t* handle() { return _handle(); }  // escape hatch to the C ABI (schedules, PCG settings, statistics)

private:
  static float _randf() { return static_cast<float>(double(std::rand()) / RAND_MAX); }  // cosmetics only (PrimitiveUtilities.cpp:10-12)
  static std::vector<float> _flatten(const std::vector<glm::vec3>& v) {
    std::vector<float> p(3 * v.size());
    for (size_t i = 0; i < v.size(); ++i) { p[3 * i] = v[i][0]; p[3 * i + 1] = v[i][1]; p[3 * i + 2] = v[i][2]; }
    return p;
  }
  static std::vector<float> _flattenMats(const std::vector<glm::mat4>& v) {
    std::vector<float> m(16 * v.size());
    for (size_t k = 0; k < v.size(); ++k)
      for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) m[16 * k + 4 * c + r] = v[k][c][r];
    return m;
  }
  // nodes, constraints and (already outward-wound) surface triangles of a tetrahedral mesh (PrimitiveUtilities.cpp:268-326)
  void _addTetMesh(const std::vector<glm::vec3>& vertices, const std::vector<uint32_t>& tetIndices, const std::vector<uint32_t>& surface,
                   const glm::vec3& initialVelocity, float density, float strainStiffness, float minStrain, float maxStrain,
                   float volumeStiffness, float compression, float stretching) {
    const uint32_t n = static_cast<uint32_t>(vertices.size());
    std::vector<float> p = _flatten(vertices), v(3 * size_t(n)), r(n, 0.5f), im(n, 1.0f / density);
    for (uint32_t i = 0; i < n; ++i) { v[3 * i] = initialVelocity[0]; v[3 * i + 1] = initialVelocity[1]; v[3 * i + 2] = initialVelocity[2]; }
    uint32_t first = 0;
    _ck(pies_add_nodes_ex(_handle(), n, p.data(), v.data(), r.data(), im.data(), &first));
    std::vector<uint32_t> t(tetIndices), f(surface);
    for (uint32_t& id : t) id += first;
    for (uint32_t& id : f) id += first;
    const uint32_t nt = static_cast<uint32_t>(t.size() / 4);
    if (strainStiffness != 0.0f) _ck(pies_add_tet_constraints(_handle(), nt, t.data(), strainStiffness, minStrain, maxStrain));
    if (volumeStiffness != 0.0f) _ck(pies_add_volume_constraints(_handle(), nt, t.data(), volumeStiffness, compression, stretching));
    _ck(pies_add_triangles(_handle(), static_cast<uint32_t>(f.size() / 3), f.data()));
    _syncRenderState();
  }
  void _ck(int rc) const {
    if (rc != PIES_OK) throw std::runtime_error(std::string("Pies::Solver: ") + pies_last_error(_h));
  }
  // after an add*/create*: new vertices get one colour per primitive, lines/triangles are re-read
  void _syncRenderState() {
    uint32_t n = 0, nl = 0, nt = 0;
    _ck(pies_count(_handle(), PIES_NODES, &n));
    _ck(pies_count(_handle(), PIES_LINES, &nl));
    _ck(pies_count(_handle(), PIES_TRIANGLES, &nt));
    const size_t old = _vertices.size();
    _vertices.resize(n);
    std::vector<float> pos(3 * size_t(n)), rad(n);
    if (n) {
      _ck(pies_read_nodes(_handle(), PIES_NODE_POSITION, pos.data(), n));
      _ck(pies_read_nodes(_handle(), PIES_NODE_RADIUS, rad.data(), n));
    }
    const glm::vec3 colour(_randf(), _randf(), _randf());
    const float roughness = _randf(), metallic = static_cast<float>(std::rand() % 2);
    for (size_t i = 0; i < n; ++i) {
      _vertices[i].position = glm::vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
      _vertices[i].radius = rad[i];
      if (i >= old) { _vertices[i].baseColor = colour; _vertices[i].roughness = roughness; _vertices[i].metallic = metallic; }
    }
    _lines.resize(nl);
    if (nl) _ck(pies_get_ids(_handle(), PIES_LINES, _lines.data(), nl));
    _triangles.resize(nt);
    if (nt) _ck(pies_get_ids(_handle(), PIES_TRIANGLES, &_triangles[0].nodeIds[0], 3 * nt));
    renderStateDirty = true;
  }
  // Solver.cpp:157,393: _vertices[i].position = node.position.  pies_tick has already brought the positions to the
  // host (one pinned D2H copy per tick); this writes them straight into the vertex stream.
  void _refreshPositions() {
    const uint32_t n = static_cast<uint32_t>(_vertices.size());
    if (!n) return;
    _ck(pies_read_positions_strided(_handle(), &_vertices[0].position, sizeof(Vertex), n));
  }

  // opens the device handle on first use
  pies_solver_t* _handle() {
    if (!_h) {
      pies_options_t o;
      static_assert(sizeof(o) == sizeof(_options), "layout");
      std::memcpy(&o, &_options, sizeof(o));
      if (pies_create(&o, _device, &_h) != PIES_OK) throw std::runtime_error("Pies::Solver: no gfx950 HIP device (there is no CPU fallback)");
    }
    return _h;
  }
  // the public flags, the schedule and the solver to run reach the handle when they differ from what it was last given
  void _pushState(SolverName solver) {
    pies_solver_t* h = _handle();
    const int now[5] = {releaseHinge ? 1 : 0, nodeCollisions ? 1 : 0, triangleCollisions ? 1 : 0, schedule, static_cast<int>(solver)};
    if (now[0] != _pushed[0]) _ck(pies_set_flag(h, PIES_FLAG_RELEASE_HINGE, now[0]));
    if (now[1] != _pushed[1]) _ck(pies_set_flag(h, PIES_FLAG_NODE_COLLISIONS, now[1]));
    if (now[2] != _pushed[2]) _ck(pies_set_flag(h, PIES_FLAG_TRIANGLE_COLLISIONS, now[2]));
    if (now[3] != _pushed[3] && now[3] >= 0) _ck(pies_set_schedule(h, now[3]));
    if (now[4] != _pushed[4]) _ck(pies_set_solver(h, now[4]));
    for (int k = 0; k < 5; ++k) _pushed[k] = now[k];
  }
  void _tickAs(SolverName solver) {
    _pushState(solver);
    _ck(pies_tick(_handle()));
    _refreshPositions();
  }

  pies_solver_t* _h = nullptr;
  SolverOptions _options;
  int _device = 0;
  int _pushed[5] = {-2, -2, -2, -2, -2};  // releaseHinge, nodeCollisions, triangleCollisions, schedule, solver as last given to the handle
  std::vector<Vertex> _vertices;
  std::vector<uint32_t> _lines;
  std::vector<Triangle> _triangles;
  uint64_t _frames[2] = {0, 0};  // ticks begun and not yet ended (beginTick / endTick)
};
}  // namespace Pies
