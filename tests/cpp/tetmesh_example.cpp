// addTetMeshVolume through the drop-in class: the post-tetrahedralize part of Solver::addTriMeshVolume
// (PrimitiveUtilities.cpp:243-328) fed with tetgen-style arrays.  The mesh is a 3 x 3 x 3 node lattice (spacing 1) cut
// into 6 tetrahedra per cell; the face list holds every face once, boundary faces wound INWARD like tetgen's trifacelist
// (the reference switches the winding at :263-266), interior faces with both neighbours set.
//   argv[1] = "faces": the overload with trifacelist + face2tetlist;  "derive": the overload that derives the boundary.
// Prints the triangles and the node positions after 5 PD ticks (hex floats), which tests/test_dropin_cpp.py compares with
// the oracle's run of the same mesh.  Exit code 0 on success.
#include <Pies/Solver.h>

#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>

int main(int argc, char** argv) {
  const bool derive = argc > 1 && std::strcmp(argv[1], "derive") == 0;
  const int N = 3;
  std::vector<glm::vec3> verts;
  for (int x = 0; x < N; ++x)
    for (int y = 0; y < N; ++y)
      for (int z = 0; z < N; ++z) verts.push_back(glm::vec3(0.3f + x, 1.5f + y, 0.2f + z));
  auto id = [&](int x, int y, int z) { return static_cast<uint32_t>(z + N * (y + N * x)); };
  std::vector<uint32_t> tets;
  static const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  for (int x = 0; x + 1 < N; ++x)
    for (int y = 0; y + 1 < N; ++y)
      for (int z = 0; z + 1 < N; ++z)
        for (const auto& p : perm) {  // Kuhn split: walk from the cell's corner to the opposite one, one axis at a time
          int c[3] = {x, y, z};
          uint32_t v[4];
          v[0] = id(c[0], c[1], c[2]);
          for (int k = 0; k < 3; ++k) { ++c[p[k]]; v[k + 1] = id(c[0], c[1], c[2]); }
          // positive orientation
          const glm::vec3 &a = verts[v[0]], &b = verts[v[1]], &cc = verts[v[2]], &d = verts[v[3]];
          const float ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2], vx = cc[0] - a[0], vy = cc[1] - a[1], vz = cc[2] - a[2];
          const float wx = d[0] - a[0], wy = d[1] - a[1], wz = d[2] - a[2];
          const float det = ux * (vy * wz - vz * wy) - uy * (vx * wz - vz * wx) + uz * (vx * wy - vy * wx);
          if (det < 0) std::swap(v[2], v[3]);
          for (uint32_t q : v) tets.push_back(q);
        }
  // tetgen-style faces: every face once, face2tet = the one or two elements it belongs to; boundary faces wound inward
  std::map<std::array<uint32_t, 3>, std::array<int, 2>> adj;
  std::map<std::array<uint32_t, 3>, std::array<uint32_t, 4>> first;  // face as listed (a, b, c) + opposite vertex
  static const int kFace[4][3] = {{1, 2, 3}, {0, 3, 2}, {0, 1, 3}, {0, 2, 1}};
  for (size_t t = 0; t < tets.size() / 4; ++t)
    for (int f = 0; f < 4; ++f) {
      std::array<uint32_t, 3> key = {tets[4 * t + kFace[f][0]], tets[4 * t + kFace[f][1]], tets[4 * t + kFace[f][2]]};
      const std::array<uint32_t, 4> listed = {key[0], key[1], key[2], tets[4 * t + f]};
      std::sort(key.begin(), key.end());
      auto it = adj.find(key);
      if (it == adj.end()) { adj[key] = {static_cast<int>(t), -1}; first[key] = listed; }
      else it->second[1] = static_cast<int>(t);
    }
  std::vector<uint32_t> faces;
  std::vector<int> face2tet;
  size_t nBoundary = 0;
  for (const auto& kv : adj) {
    std::array<uint32_t, 4> l = first[kv.first];
    // wind the face so that its normal points TOWARDS the element's fourth vertex (inward), tetgen's convention here
    const glm::vec3 &a = verts[l[0]], &b = verts[l[1]], &c = verts[l[2]], &d = verts[l[3]];
    const float ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2], vx = c[0] - a[0], vy = c[1] - a[1], vz = c[2] - a[2];
    const float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    if (nx * (d[0] - a[0]) + ny * (d[1] - a[1]) + nz * (d[2] - a[2]) < 0.0f) std::swap(l[1], l[2]);
    faces.push_back(l[0]); faces.push_back(l[1]); faces.push_back(l[2]);
    face2tet.push_back(kv.second[0]); face2tet.push_back(kv.second[1]);
    if (kv.second[1] < 0) ++nBoundary;
  }

  Pies::SolverOptions options;  // reference defaults: PD
  options.iterations = 5;
  Pies::Solver solver(options);
  solver.addNodes({glm::vec3(9.0f, 3.0f, 9.0f)});  // something before the mesh: ids are offset (PrimitiveUtilities.cpp:243)
  const float density = 2.5f;
  if (derive) solver.addTetMeshVolume(verts, tets, glm::vec3(0.0f, -1.0f, 0.0f), density, 1.0f, 0.8f, 1.0f, 1.0f, 1.0f, 1.0f);
  else solver.addTetMeshVolume(verts, tets, faces, face2tet, glm::vec3(0.0f, -1.0f, 0.0f), density, 1.0f, 0.8f, 1.0f, 1.0f, 1.0f, 1.0f);

  const auto& tris = solver.getTriangles();
  if (tris.size() != nBoundary || nBoundary != 2u * 6u * (N - 1) * (N - 1)) return 2;
  if (solver.getVertices().size() != verts.size() + 1) return 3;
  // every surface triangle is wound outward: its normal points away from the mesh's centre
  const float cx = 0.3f + 1.0f, cy = 1.5f + 1.0f, cz = 0.2f + 1.0f;
  for (const Pies::Triangle& t : tris) {
    const glm::vec3 &a = solver.getVertices()[t.nodeIds[0]].position, &b = solver.getVertices()[t.nodeIds[1]].position,
                    &c = solver.getVertices()[t.nodeIds[2]].position;
    const float ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2], vx = c[0] - a[0], vy = c[1] - a[1], vz = c[2] - a[2];
    const float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    const float mx = (a[0] + b[0] + c[0]) / 3 - cx, my = (a[1] + b[1] + c[1]) / 3 - cy, mz = (a[2] + b[2] + c[2]) / 3 - cz;
    if (!(nx * mx + ny * my + nz * mz > 0.0f)) return 4;
    if (t.nodeIds[0] == 0 || t.nodeIds[1] == 0 || t.nodeIds[2] == 0) return 5;  // ids are offset by the node added before
  }
  uint32_t nTet = 0, nVol = 0;
  pies_count(solver.handle(), PIES_TET, &nTet);
  pies_count(solver.handle(), PIES_VOLUME, &nVol);
  if (nTet != tets.size() / 4 || nVol != nTet) return 6;
  std::vector<float> im(verts.size() + 1), rad(verts.size() + 1);
  pies_read_nodes(solver.handle(), PIES_NODE_INV_MASS, im.data(), static_cast<uint32_t>(im.size()));
  pies_read_nodes(solver.handle(), PIES_NODE_RADIUS, rad.data(), static_cast<uint32_t>(rad.size()));
  if (im[1] != 1.0f / density || rad[1] != 0.5f || solver.getVertices()[1].radius != 0.5f) return 7;

  for (const Pies::Triangle& t : tris) std::printf("tri %u %u %u\n", t.nodeIds[0], t.nodeIds[1], t.nodeIds[2]);
  for (uint32_t q = 0; q < tets.size(); q += 4) std::printf("tet %u %u %u %u\n", tets[q] + 1, tets[q + 1] + 1, tets[q + 2] + 1, tets[q + 3] + 1);
  for (int i = 0; i < 5; ++i) solver.tick(0.0f);
  for (const auto& v : solver.getVertices()) std::printf("pos %a %a %a\n", v.position[0], v.position[1], v.position[2]);
  std::printf("tetmesh ok\n");
  return 0;
}
