// Compiles against the drop-in header exactly like a Pies host would (PiesForAlthea-style usage):
// build a scene through the reference's public API, tick, read vertices.  Exit code 0 on success.
#include <Pies/Solver.h>

#include <cmath>
#include <cstdio>
#include <cstring>

int main(int argc, char** argv) {
  const bool pd = argc > 1 && std::strcmp(argv[1], "pd") == 0;
  Pies::SolverOptions options;
  options.solver = pd ? Pies::SolverName::PD : Pies::SolverName::PBD;
  options.iterations = 6;
  Pies::Solver solver(options);
  solver.createTetBox(glm::vec3(0.0f, 4.0f, 0.0f), 1.0f, glm::vec3(0.0f), pd ? 1.0f : 0.002f, 1.0f, false);
  solver.createBox(glm::vec3(8.0f, 3.0f, 0.0f), 1.0f, 0.5f);
  solver.addNodes({glm::vec3(20.0f, 2.0f, 0.0f), glm::vec3(20.6f, 2.0f, 0.0f)});
  const size_t n = solver.getVertices().size();
  if (n != 27 + 125 + 2 || solver.getTriangles().empty() || solver.getLines().empty() || !solver.renderStateDirty) return 2;
  solver.renderStateDirty = false;
  const float y0 = solver.getVertices()[0].position[1];
  for (int i = 0; i < 10; ++i) solver.tick(0.016f);
  const Pies::Solver::Vertex& v = solver.getVertices()[0];
  if (!(std::isfinite(v.position[0]) && v.position[1] < y0)) return 3;  // the box fell under gravity
  std::printf("dropin ok: %zu vertices, %zu triangles, y %.4f -> %.4f (%s)\n", n, solver.getTriangles().size(), y0, v.position[1],
              pd ? "PD" : "PBD");
  // tick() in two halves (beginTick queues the substeps and the asynchronous export, endTick waits for the frame): two
  // frames in flight, and the vertices end where three plain ticks of an identical solver end
  Pies::Solver twin(options);
  twin.createTetBox(glm::vec3(0.0f, 4.0f, 0.0f), 1.0f, glm::vec3(0.0f), pd ? 1.0f : 0.002f, 1.0f, false);
  twin.createBox(glm::vec3(8.0f, 3.0f, 0.0f), 1.0f, 0.5f);
  twin.addNodes({glm::vec3(20.0f, 2.0f, 0.0f), glm::vec3(20.6f, 2.0f, 0.0f)});
  for (int i = 0; i < 10; ++i) twin.tick(0.016f);
  for (int i = 0; i < 3; ++i) solver.tick(0.016f);
  twin.beginTick();
  twin.beginTick();
  twin.endTick();
  twin.beginTick();
  twin.endTick();
  twin.endTick();
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k)
      if (twin.getVertices()[i].position[k] != solver.getVertices()[i].position[k]) return 5;
  // the reference's construction forms: copy-initialisation from the options (its constructor is not explicit), a
  // default-constructed solver (no device is opened until something needs one) that is move-assigned later
  Pies::Solver fromOptions = options;
  Pies::Solver late;
  if (late.getOptions().iterations != 4 || !late.getVertices().empty()) return 6;
  late = std::move(fromOptions);
  if (late.getOptions().iterations != 6) return 7;
  // tickPBD / tickPD run the NAMED solver whatever options.solver says (Solver.cpp:40, :162): the same scene ticked with
  // tickPBD under PD options ends where a PBD solver's tick() ends
  Pies::SolverOptions other = options;
  other.solver = pd ? Pies::SolverName::PBD : Pies::SolverName::PD;
  Pies::Solver named(other), plain(options);
  for (Pies::Solver* s : {&named, &plain}) s->createTetBox(glm::vec3(0.0f, 4.0f, 0.0f), 1.0f, glm::vec3(0.0f), pd ? 1.0f : 0.002f, 1.0f, false);
  for (int i = 0; i < 3; ++i) {
    if (pd) named.tickPD(0.0f); else named.tickPBD(0.0f);
    plain.tick(0.0f);
  }
  for (size_t i = 0; i < plain.getVertices().size(); ++i)
    for (int k = 0; k < 3; ++k)
      if (named.getVertices()[i].position[k] != plain.getVertices()[i].position[k]) return 8;
  Pies::Solver moved(std::move(solver));
  moved.tick(0.0f);
  moved.clear();
  return moved.getVertices().empty() ? 0 : 4;
}
