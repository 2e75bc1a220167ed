/* pies_hip.h -- C ABI of the MI355X (gfx950) implementation of the Pies solver loop.
 *
 * The reference (nithinp7/Pies) has no FFI: hosts include <Pies/Solver.h> and call the C++ class
 * (Include/Pies/Solver.h:40-116 in the reference tree).  This header is the boundary a binding for that
 * class would use: plain pointers and sizes, no C++ or torch types.  `include/Pies/Solver.h` in this
 * repository is the C++ drop-in class written on top of it.  Each entry point cites the reference
 * interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - every call returns 0 (PIES_OK) or a PIES_ERR_* code; pies_last_error() gives the text;
 *  - one handle = one HIP device + one stream; calls on one handle are serialised by the caller
 *    (the reference solver is single-caller as well);
 *  - inputs are copied, no caller buffer is retained;
 *  - node ids returned/accepted are global (offset by the node count at the time of the add*),
 *    exactly like the reference's add and create functions (Src/PrimitiveUtilities.cpp:53-57,358);
 *  - there is no CPU fallback: without a gfx950 device pies_create fails.
 */
#ifndef PIES_HIP_H
#define PIES_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: PIES_SCHEDULE_LAYERED, PIES_KERNEL_LAYER (pies_launch_counts now fills PIES_KERNEL_COUNT = 19 entries)
 * 3: pies_tick_begin / pies_export_acquire / pies_export_release, pies_read_positions_strided, pies_get_pcg_health,
 *    pies_set_pcg_retry, PIES_FLAG_REFERENCE_COLLISION_ORDER, pies_profile_in_situ; PIES_SCHEDULE_DEFAULT */
#define PIES_ABI_VERSION 4

typedef struct pies_solver pies_solver_t;

/* Pies::SolverOptions, field for field (Include/Pies/Solver.h:23-38). solver: 0 = PBD, 1 = PD. */
typedef struct pies_options {
  float fixedTimestepSize;                   /* 0.012f */
  uint32_t timeSubsteps;                     /* 1 */
  uint32_t iterations;                       /* 4 */
  uint32_t collisionStabilizationIterations; /* 4 */
  float collisionThresholdDistance;          /* 0.1f */
  float collisionThickness;                  /* 0.05f */
  float gravity;                             /* 10.0f */
  float damping;                             /* 0.006f */
  float friction;                            /* 0.01f */
  float staticFrictionThreshold;             /* 0.f */
  float floorHeight;                         /* 0.0f */
  float gridSpacing;                         /* 2.0f */
  uint32_t threadCount;                      /* 8 (accepted; only orders PD floor contacts) */
  int32_t solver;                            /* 1 (PD) */
} pies_options_t;

enum {
  PIES_OK = 0,
  PIES_ERR_INVALID = 1,     /* bad argument (null, out-of-range node id, size mismatch) */
  PIES_ERR_HIP = 2,         /* HIP runtime error, or no gfx950 device */
  PIES_ERR_STATE = 3,       /* call not valid in the handle's current state */
  PIES_ERR_UNSUPPORTED = 4  /* feature not available in this build */
};

enum { PIES_SOLVER_PBD = 0, PIES_SOLVER_PD = 1 }; /* Pies::SolverName, Solver.h:21 */

/* constraint containers, in the order Solver::tickPBD / tickPD visit them */
enum {
  PIES_POSITION = 0, /* PositionConstraint      Constraints.cpp:58-74  */
  PIES_DISTANCE = 1, /* DistanceConstraint      Constraints.cpp:11-56  */
  PIES_TET = 2,      /* TetrahedralConstraint   Constraints.cpp:76-184 */
  PIES_VOLUME = 3,   /* VolumeConstraint        Constraints.cpp:186-310 (PD only) */
  PIES_BEND = 4,     /* BendConstraint          Constraints.cpp:312-394 */
  PIES_SHAPE = 5,    /* ShapeMatchingConstraint ShapeMatchingConstraint.cpp:6-122 (PD only) */
  PIES_GOAL = 6,     /* GoalMatchingConstraint  ShapeMatchingConstraint.cpp:124-177 (PD only) */
  PIES_TRIANGLES = 7,
  PIES_LINES = 8,
  PIES_NODES = 9,
  PIES_SYSTEM_NNZ = 10, /* pies_count only: stored entries of the PD system matrix (after pies_finalize) */
  PIES_REST_SETS = 11,  /* pies_count only: distinct sets of element constants in the PD local step's rest dictionary (0: the
                         * per-element arrays are read; after pies_finalize) */
  PIES_ROW_STENCILS = 12, /* pies_count only: distinct rows in the row dictionary of the PD system matrix (0: SELL arrays only) */
  PIES_PD_TILES = 13,     /* pies_count only: tiles of the PD strain + volume local step (0: one record per (element, node)) */
  PIES_PD_TILE_RECORDS = 14, /* pies_count only: (tile, node) sums the right-hand side gathers */
  PIES_PD_CG_SINGLE = 15, /* pies_count only: 1 when the captured global step runs one launch per CG iteration */
  PIES_PD_WINDOW_ENTRIES = 16, /* pies_count only: stored entries (padding included) of the windowed system matrix the CG iterations
                                  stream - value + 16-bit window slot each -, 0: not built (row dictionary, or the SELL arrays) */
  PIES_PD_WINDOW_HALO = 17,    /* pies_count only: its halo entries over all chunks (columns staged in LDS besides a chunk's own rows) */
  PIES_NODE_PAIRS = 18         /* CollisionConstraint (node-node, PD)  CollisionConstraint.cpp:7-65: an EXTENSION container, see
                                  pies_add_node_pair_constraints */
};

/* How the sequential Gauss-Seidel sweeps of tickPBD (Solver.cpp:58-75) are mapped to the device.
 *  EXACT    : dependency levels of the container order; bit-identical to processing the containers
 *             sequentially in the order the host added the constraints (the reference's semantics).
 *  COLOURED : greedy graph colouring; identical to a sequential sweep over the containers re-ordered
 *             colour by colour (pies_get_order returns that order).  Fewer, larger launches.
 *  LAYERED  : breadth-first levels of the constraint graph; the constraints between two adjacent levels are
 *             swept colour by colour by one workgroup with the nodes resident in LDS (two launches per
 *             iteration; wide bodies are cut into strips by a second levelling: four phases per container).
 *             Identical to a sequential sweep in the order pies_get_order returns.  Falls back to COLOURED
 *             for scenes without distance / tetrahedral / bend constraints and for squat bodies of fewer than
 *             300k nodes whose cross-sections do not fit one workgroup (measured slower there). */
enum { PIES_SCHEDULE_EXACT = 0, PIES_SCHEDULE_COLOURED = 1, PIES_SCHEDULE_LAYERED = 2 };
/* What pies_create and Pies::Solver start with: the schedule bench.py's headline figure is measured on.  Every schedule
 * is a Gauss-Seidel sweep over the same constraints with the same per-constraint arithmetic; LAYERED and COLOURED
 * visit them in another order than the host added them, EXACT in exactly that order (and then also runs the node-node
 * pass in the reference's order, see PIES_FLAG_REFERENCE_COLLISION_ORDER).  bench.py reports how far the orders are
 * apart (`order_deviation`) and the EXACT throughput beside the headline.  The environment variable
 * PIES_SCHEDULE=exact|coloured|layered overrides the default at pies_create (not an explicit pies_set_schedule). */
#define PIES_SCHEDULE_DEFAULT PIES_SCHEDULE_LAYERED

enum {
  PIES_FLAG_RELEASE_HINGE = 0,  /* Solver::releaseHinge (Solver.h:52, Solver.cpp:59) */
  PIES_FLAG_NODE_COLLISIONS = 1,    /* extension, default 1: 0 skips the PBD node-node pass (Solver.cpp:81-130) */
  PIES_FLAG_TRIANGLE_COLLISIONS = 2, /* extension, default 1: 0 skips the PD point-triangle contacts (Solver.cpp:693-797) */
  /* PBD node-node pass (Solver.cpp:85-130).  1: the reference's loop -- nodes in ascending index, each querying the cell
   * range of its *current* position, every overlapping pair resolved at once -- executed as one sequential chain on
   * the device (bit-identical to the loop, slow).  0: the parallel visiting rule of DESIGN.md section 6 (same contact
   * set and per-pair arithmetic, different order).  Default: 1 under PIES_SCHEDULE_EXACT, 0 otherwise; setting the
   * flag overrides that until the schedule changes. */
  PIES_FLAG_REFERENCE_COLLISION_ORDER = 3,
  /* The order of the PBD node-node pass, by name (value: PIES_COLLISION_ORDER_*).  Every order uses the reference's per-pair
   * arithmetic; the loop is order dependent (Solver.cpp:85-130 moves both nodes at once).
   *  REFERENCE : the reference's loop (see above): node i meets every node of every bucket of the cell range recomputed from
   *              i's LIVE position when its turn comes (SpatialHash.h:101-106), in each direction, itself included.
   *  PAIRS and GROUPS are a documented deviation: the same per-pair arithmetic, but node i meets node j once per grid cell
   *  both were INSERTED into at the start of the iteration - the same meetings unless a node crosses a cell boundary
   *  inside the pass, and a different order.
   *  PAIRS     : default under COLOURED / LAYERED.  A node's meetings with itself first, then the pairs {i < j} in
   *              ascending order of a 64-bit mix of (i, j), each as its m visits of i to j and m visits of j to i; executed
   *              by dependency levels, one lane per pair.
   *  GROUPS    : rounds 1-2's order (nodes grouped by minimum cell, 27 residue classes, ascending index inside a group);
   *              needs cell ranges of at most two cells per axis: a scene with gridSpacing < 2 (r_max + 0.5) runs REFERENCE instead.
   * PIES_FLAG_REFERENCE_COLLISION_ORDER = 0 selects PAIRS. */
  PIES_FLAG_COLLISION_ORDER = 4
};
enum { PIES_COLLISION_ORDER_REFERENCE = 0, PIES_COLLISION_ORDER_GROUPS = 1, PIES_COLLISION_ORDER_PAIRS = 2 };

/* node state selectors for pies_read_nodes / pies_write_nodes */
enum { PIES_NODE_POSITION = 0, PIES_NODE_PREV_POSITION = 1, PIES_NODE_VELOCITY = 2, PIES_NODE_RADIUS = 3, PIES_NODE_INV_MASS = 4 };

/* ---- lifetime ------------------------------------------------------------------------------ */
/* device == PIES_DEVICE_NONE creates a host-only handle: scenes can be built and schedules inspected
 * (pies_get_ids/_rest/_order/_batches), every call that would compute returns PIES_ERR_HIP. */
#define PIES_DEVICE_NONE (-1)
/* Solver::Solver(const SolverOptions&) (Solver.cpp:11-17).  options == NULL -> reference defaults. */
int pies_create(const pies_options_t* options, int device, pies_solver_t** out);
/* Solver::~Solver (Solver.cpp:19-23) */
int pies_destroy(pies_solver_t* s);
/* Solver::clear (Solver.cpp:488-507); also resets the PD system. */
int pies_clear(pies_solver_t* s);
const char* pies_last_error(const pies_solver_t* s);
int pies_abi_version(void);
void pies_default_options(pies_options_t* out);
/* Solver::getOptions (Solver.h:71) */
int pies_get_options(const pies_solver_t* s, pies_options_t* out);

/* ---- scene construction (setup time; host side) --------------------------------------------- */
/* Solver::addNodes (PrimitiveUtilities.cpp:42-75): mass 1, radius 0.5, velocity 0.  pos: n x 3. */
int pies_add_nodes(pies_solver_t* s, uint32_t n, const float* pos, uint32_t* first_id);
/* Bulk node append with explicit state; vel/radius/inv_mass may be NULL (0 / 0.5 / 1). */
int pies_add_nodes_ex(pies_solver_t* s, uint32_t n, const float* pos, const float* vel, const float* radius,
                      const float* inv_mass, uint32_t* first_id);
/* create*Constraint factories (Constraints.cpp:39-56, 65-74, 130-184, 257-310, 368-394): rest state is
 * taken from the nodes' current positions, like the reference factories do. ids: n x {1,2,4,4,4}. */
int pies_add_position_constraints(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w);
int pies_add_distance_constraints(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w);
int pies_add_tet_constraints(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w, float min_strain, float max_strain);
int pies_add_volume_constraints(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w, float compression, float stretching);
int pies_add_bend_constraints(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w);
/* EXTENSION (not reachable in the reference): n node-node CollisionConstraints over the node pairs ids[2 i], ids[2 i + 1]
 * (Src/CollisionConstraint.cpp:7-65, w = 1e5 of Include/Pies/CollisionConstraint.h:14).  The reference's only source of these
 * constraints, Solver::_parallelComputeCollisions (Solver.cpp:509-637), is never called, and tickPD calls none of the type's three
 * methods - only its friction loop (Solver.cpp:398-428) walks the always-empty list.  Here a host may list pairs itself; PD then
 * treats them the way it treats the live collision constraints: projection in every local step (:10-41), w on both diagonal entries
 * of the system (:43-47), w * projected in the right-hand side (:49-65), the friction loop after the velocity update.  PBD scenes
 * ignore the list (the PBD tick has its own node-node pass, Solver.cpp:85-130).  Default: none. */
int pies_add_node_pair_constraints(pies_solver_t* s, uint32_t n, const uint32_t* ids);
/* ShapeMatchingConstraint over the listed nodes (ShapeMatchingConstraint.cpp:6-48); material coordinates
 * are the nodes' current positions, as createShapeMatching* / addLinkedRegions pass them.  PD only. */
int pies_add_shape_constraint(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w);
/* GoalMatchingConstraint over the listed nodes (ShapeMatchingConstraint.cpp:124-137).  PD only. */
int pies_add_goal_constraint(pies_solver_t* s, uint32_t n, const uint32_t* ids, float w, uint32_t* goal_index);
/* GoalMatchingConstraint::setTransform (:175-177); m16 column-major like glm::mat4 */
int pies_set_goal_transform(pies_solver_t* s, uint32_t goal, const float m16[16]);
/* Solver::addFixedRegions / updateFixedRegions / addLinkedRegions (PrimitiveUtilities.cpp:77-162);
 * mats16: n column-major 4x4 region-to-world matrices (the region is the unit box [-1,1]^3) */
int pies_add_fixed_regions(pies_solver_t* s, uint32_t n, const float* mats16, float w);
int pies_update_fixed_regions(pies_solver_t* s, uint32_t n, const float* mats16);
int pies_add_linked_regions(pies_solver_t* s, uint32_t n, const float* mats16, float w);
/* Solver::createShapeMatchingBox (:985-1048; spacing fixed to 0.5 and mass 10 like the reference) and
 * Solver::createShapeMatchingSheet (:1050-1125; reference 50 x 50, 3x3 patches) */
int pies_create_shape_matching_box(pies_solver_t* s, const float translation[3], uint32_t count_x, uint32_t count_y, uint32_t count_z, float w);
int pies_create_shape_matching_sheet(pies_solver_t* s, uint32_t W, uint32_t H, const float translation[3], float scale, float w);
/* Solver::_triangles entries (Solver.h:194); ids: n x 3 */
int pies_add_triangles(pies_solver_t* s, uint32_t n, const uint32_t* ids);
/* Solver::createTetBox (PrimitiveUtilities.cpp:330-618) on a W x H x D lattice (reference: 3x3x3, hinged
 * 10x2x10).  flags bit0: push VolumeConstraints too (the reference always does); bit1: surface triangles. */
int pies_create_tet_box(pies_solver_t* s, uint32_t W, uint32_t H, uint32_t D, const float translation[3], float scale,
                        const float velocity[3], float w, float mass, uint32_t flags);
/* Solver::createBox (PrimitiveUtilities.cpp:620-847) on a W x H x D lattice (reference: 5x5x5).
 * existing != 0: only add the distance constraints, over an existing lattice starting at node existing_first. */
int pies_create_box(pies_solver_t* s, uint32_t W, uint32_t H, uint32_t D, const float translation[3], float scale,
                    float w, int existing, uint32_t existing_first, uint32_t flags);
/* Solver::createSheet (:849-976, reference 20x20) and Solver::createBendSheet (:1127-1289, reference 10x10) */
int pies_create_sheet(pies_solver_t* s, uint32_t W, uint32_t H, const float translation[3], float scale, float mass, float w);
int pies_create_bend_sheet(pies_solver_t* s, uint32_t W, uint32_t H, const float translation[3], float scale, float w);

/* ---- configuration -------------------------------------------------------------------------- */
int pies_set_flag(pies_solver_t* s, int flag, int value);
/* Solver::tickPBD / Solver::tickPD (Solver.h:62-63) run the named solver whatever SolverOptions::solver says: this switches
 * which loop pies_tick runs (PIES_SOLVER_PBD / PIES_SOLVER_PD).  The device buffers of the other loop are built at the next tick
 * (a full pies_finalize), so alternating every tick is correct but slow.  Node state carries over. */
int pies_set_solver(pies_solver_t* s, int solver);
int pies_set_schedule(pies_solver_t* s, int schedule);
/* PD global step: the reference solves (K + C) x = rhs with a sparse Cholesky factorisation rebuilt every
 * substep (Solver.cpp:258-262,356); here it is a Jacobi-preconditioned CG.  rel_tol bounds ||r||/||rhs|| per
 * coordinate column, max_iters the upper bound of CG iterations per solve (defaults 3e-7, 128; the captured graph
 * holds as many as the recent solves needed plus one or two). */
int pies_set_pcg(pies_solver_t* s, float rel_tol, uint32_t max_iters);
/* Over the last pies_tick, or over the pies_tick_async calls since the last synchronisation: largest ||r||/||rhs|| left by
 * any solve, most CG iterations any solve used, solves run. */
int pies_get_pcg_stats(pies_solver_t* s, float* max_rel_residual, uint32_t* max_iters_used, uint32_t* solves);
/* The reference's global step is exact (Solver.cpp:356).  pies_tick therefore does not keep a substep in which a solve
 * ended above rel_tol: the substep's input is restored and it runs again with four times the captured CG budget, up to
 * max_iters (default on; 0 switches the re-run off).  pies_tick_async cannot look at a substep before the next one is
 * queued; both paths count what was left above the tolerance:
 *   short_solves      solves of kept substeps that ended above rel_tol since pies_finalize (0 = every solve met it),
 *   solves_total      solves of kept substeps,
 *   substeps_retried  substeps pies_tick ran again,
 *   budget            CG iterations per solve in the captured graph right now. */
int pies_set_pcg_retry(pies_solver_t* s, int enabled);
int pies_get_pcg_health(pies_solver_t* s, uint64_t* short_solves, uint64_t* solves_total, uint32_t* substeps_retried, uint32_t* budget);
/* Builds schedules, uploads to HBM and captures the substep graph.  Called implicitly by pies_tick
 * when the scene changed. */
int pies_finalize(pies_solver_t* s);

/* ---- the hot path --------------------------------------------------------------------------- */
/* Solver::tick (Solver.cpp:25-38): timeSubsteps substeps of tickPBD / tickPD, then positions are copied
 * back so that pies_read_nodes / Solver::getVertices are current (Solver.cpp:157,393).  No-op once failed. */
int pies_tick(pies_solver_t* s);
/* Same work, but neither the host copy-back nor a stream synchronisation: state stays in HBM.  (Projective Dynamics: the
 * captured CG budget can only follow the solves at a host synchronisation, so every 16th call without one in between
 * synchronises first.) */
int pies_tick_async(pies_solver_t* s);
/* Waits for the queued ticks, then latches a device-side failure (pies_failed) and adapts the CG budget. */
int pies_synchronize(pies_solver_t* s);
/* Render-state export without a stall (Solver::getVertices, Solver.h:42-71; Solver.cpp:157,393 write _vertices every
 * substep): pies_tick_begin queues one tick plus the copy of its final positions into one of two pinned host buffers
 * (a copy stream moves frame k while the compute stream already runs frame k+1) and returns at once with the frame's
 * id (1, 2, ...).  pies_export_acquire blocks until that frame's copy has landed and returns n x 4 floats
 * (x, y, z, invMass); the pointer stays valid until pies_export_release or until two more frames have begun.  A
 * third pies_tick_begin while the oldest frame is still acquired fails with PIES_ERR_STATE. */
int pies_tick_begin(pies_solver_t* s, uint64_t* frame);
int pies_export_acquire(pies_solver_t* s, uint64_t frame, const float** pos4, uint32_t* n);
int pies_export_release(pies_solver_t* s, uint64_t frame);
/* _simFailed latch (Solver.cpp:26-28,853-856) */
int pies_failed(pies_solver_t* s, int* failed);
/* Point-triangle contacts of the last PD substep (Solver::_triCollisions, Solver.h:187), in list order:
 * 4 node ids each (point a, triangle b c d).  ids may be NULL to query the count. */
int pies_get_tri_contacts(pies_solver_t* s, uint32_t* ids, uint32_t capacity, uint32_t* count);
/* Broad phase of the last PD substep's point-triangle detection (diagnostics; synchronises).  The device lists every surface
 * triangle once, in the cell of the minimum corner of its swept range, in one of three size classes (cells of 1, 4 and 16 world
 * cells; pies_amd/csrc/tri_kernels.h) - the reference lists it in every cell of the range (Solver.cpp:692, SpatialHash.h:141-176);
 * the pairs and the contact list are the same.  out[0] = pairs handed to the CCD (Solver.cpp:775-797 runs it for every pair
 * without a common node in a shared cell), out[1] = pairs with at least one hit, out[2..4] = longest range listed per class (in
 * cells of the class), out[5..7] = triangles listed per class. */
int pies_get_tri_grid_stats(pies_solver_t* s, uint32_t out[8]);
/* Node-node pairs resolved by the PBD collision pass since the last call (statistics). */
int pies_collision_pairs(pies_solver_t* s, uint64_t* pairs);
/* The same plus the candidates the pass looked at (bucket entries visited, the unit of SURVEY 8d's "16 B per candidate
 * neighbour"); both counters restart. */
int pies_collision_stats(pies_solver_t* s, uint64_t* pairs, uint64_t* candidates);
/* Pair order: level launches captured per pass.  By default the library follows what the passes need at host synchronisations
 * (it starts at 96; BASELINE config 4 settles at 48 in the pair order, at ~1 100 in the reference's order by turns); a call pins the count.  A single-workgroup kernel finishes deeper
 * orders than captured: slow, never wrong. */
int pies_set_collision_rounds(pies_solver_t* s, uint32_t rounds);
/* Tuning and diagnostic switches, process wide, by name (value NULL or "" unsets): graph variants and sizes that tests and
 * profiling scripts pin - PIES_PCG_BUDGET, PIES_PCG_OVERFLOW, PIES_TRI_FAST_ROWS, PIES_TRI_LDS, PIES_ROW_MAX_UNIQUE, PIES_TRI_SIDE, PIES_TRI_TEAM,
 * PIES_NO_GRAPH, PIES_NO_WAVEFRONT, PIES_NO_TET_PAIRS, PIES_PD_LOCAL_PACKED (0: one element per lane in the PD strain + volume step),
 * PIES_PD_REST_DICT (0: per-element constants instead of the rest dictionary), PIES_PD_ROW_DICT (0: the PD system matrix as SELL
 * arrays only, no row dictionary), PIES_LAYER_PLAN (0: schedule LAYERED's original plan only, 1: + single-level constraints dealt to either group, 2: + slabs by position; default 2) / _PLAN_FORCE (candidate index) / _SLAB / _SLAB_OFFSET, PIES_LAYER_ONE_STRIP_MAX / _TILE_NODES / _STRIPS_MIN_NODES / PIES_LAYER_BLOCK,
 * PIES_PD_TILE_ELEMS (0: per-(element, node) records instead of the tile-resident local step), PIES_PD_CG_SINGLE / _SINGLE_ROWS (0: the
 * two-launch CG everywhere / in the contact-heavy variant), PIES_PD_FUSE_RHS (0: k_pd_rhs), PIES_PD_RHS_LANES,
 * PIES_COLOUR_ROUNDS, PIES_COLOUR_DSATUR, PIES_NO_COLOUR_HINT, PIES_SELL_LANES, PIES_CG_BLOCKS, PIES_COLLIDE_GLOBAL / _PASSES /
 * _SPIN_LIMIT; round 5: PIES_PD_WINDOW (0: no windowed matrix, 2: also beside a row dictionary) / _WINDOW_KERNELS (1 iterations, 2 first
 * product, 4 residual; default 3) / _WINDOW_SORT / _WINDOW_HALO32 (tests: 32-bit halo list), PIES_CG_CHUNK_ROWS, PIES_PCG_NEVER_EXIT (profiling: every captured CG launch works),
 * PIES_REFERENCE_TURNS (0: the reference's node-node order as one sequential chain, 1: by turns whatever the size; default: by turns from
 * 1 024 nodes on), PIES_FALLBACK_VISITS (candidate tests a pass left to the sequential loop may cost before it latches: 1e9),
 * PIES_PAIR_QUADS (0: one lane per pair in the pair order's levels) / _QUAD_BLOCKS / _QUAD_THREADS / PIES_PAIR_LOOK_WAVES (wavefronts of a level workgroup that look at frontier nodes, at most).  They take effect where the library reads them (pies_create, pies_finalize or the next graph capture).  pies_set_tuning must not
 * race with other API calls of the process (a handle reads the switches at different times of its life).  None of
 * them is read from the environment: the only environment variables the library looks at are PIES_SCHEDULE (default schedule
 * of new handles), PIES_PROFILER_SAFE (profiling runs) and the print-only PIES_PCG_DEBUG / PIES_LAYER_DEBUG. */
int pies_set_tuning(const char* name, const char* value);
/* Diagnostics of the pair order: per node the slack for the next pass, the excursion and the listed partners of the last pass. */
int pies_debug_pair_state(pies_solver_t* s, float* slack, float* excursion, uint32_t* partners, uint32_t n);
/* Pair order (PIES_COLLISION_ORDER_PAIRS): the last pass's dependency levels and listed pairs; since pies_finalize, the passes
 * that were repeated with the widest slack because a node had moved further than the pair filter allows for, and the passes in
 * which a node moved further than even that (more than 0.5 in one pass: the result may then miss visits the documented order
 * makes).  Any pointer may be NULL. */
int pies_get_collision_health(pies_solver_t* s, uint32_t* levels, uint32_t* pairs_listed, uint32_t* passes_repeated, uint32_t* passes_inexact);
/* Node-node passes (since the handle's buffers were built) that the SEQUENTIAL loop of the reference ran in place of a parallel order:
 * a pile the partner lists do not hold (more than 1 024 nodes within reach of one node, more listed pairs than reserved, more than
 * 65 535 nodes in a cell), or - reference order by turns - a pass that could not be proved exact.  Such a pass has the reference's
 * own order and result (Src/Solver.cpp:85-130 has no limit and no latch); it is slow, not wrong. */
int pies_get_collision_fallbacks(pies_solver_t* s, uint32_t* passes);

/* ---- state access --------------------------------------------------------------------------- */
int pies_count(const pies_solver_t* s, int what, uint32_t* out);
/* out: n x 3 floats for POSITION/PREV_POSITION/VELOCITY, n floats for RADIUS/INV_MASS */
int pies_read_nodes(pies_solver_t* s, int what, float* out, uint32_t n);
/* Positions into a caller array with a byte stride (Solver::Vertex::position of a std::vector<Vertex>:
 * `_vertices[i].position = node.position`, Solver.cpp:157,393).  After pies_tick this is a host-side copy. */
int pies_read_positions_strided(pies_solver_t* s, void* dst, uint64_t stride_bytes, uint32_t n);
int pies_write_nodes(pies_solver_t* s, int what, const float* in, uint32_t n);
/* node ids of a container, flattened, in the order the host added them */
int pies_get_ids(const pies_solver_t* s, int type, uint32_t* out, uint32_t capacity);
/* node ids of shape (PIES_SHAPE) or goal (PIES_GOAL) constraint `index`; ids may be NULL to query count */
int pies_get_group(const pies_solver_t* s, int type, uint32_t index, uint32_t* ids, uint32_t capacity, uint32_t* count);
/* rest data in host order: DISTANCE target (1), TET/VOLUME Qinv column-major (9), BEND angle (1) */
int pies_get_rest(const pies_solver_t* s, int type, float* out, uint32_t capacity);
/* replaces the rest data of constraints [first, first + n) of a container (same layout as pies_get_rest; TET / VOLUME: A and AtA follow
 * from the new Qinv).  The reference's factories take the rest pose from the node positions at creation (Src/Constraints.cpp:39-56,
 * 130-184, 257-310, 368-394); this is the bulk raw ingestion of SURVEY 8b ("distance (ids, rest, w) ... tet/volume (ids, Qinv,
 * params, w)") for hosts that restore a saved scene. */
int pies_set_rest(pies_solver_t* s, int type, uint32_t first, uint32_t n, const float* rest);
/* Execution order of a container under the current schedule: order[slot] = index in host order.
 * batch_offsets (may be NULL) receives n_batches+1 slot offsets; batches run one after another. */
int pies_get_order(pies_solver_t* s, int type, uint32_t* order, uint32_t capacity);
int pies_get_batches(pies_solver_t* s, int type, uint32_t* batch_offsets, uint32_t capacity, uint32_t* n_batches);

/* ---- measurement ---------------------------------------------------------------------------- */
/* Times one kernel class in isolation: a graph holding only that class's launches of one substep is
 * replayed a few times back to back between two HIP events recorded on the solver's stream (the launches form
 * one dependent chain, so time / launches is the per-launch device time including the kernel boundary).  Returns the launches
 * timed, the total milliseconds and the units (constraints or nodes) processed.  One class's working set is usually cache
 * resident in such a replay: these are launch-latency figures, not bandwidth figures.  The node state is put back. */
enum { PIES_KERNEL_PREDICT = 0, PIES_KERNEL_POSITION = 1, PIES_KERNEL_DISTANCE = 2, PIES_KERNEL_TET = 3,
       PIES_KERNEL_BEND = 4, PIES_KERNEL_FLOOR = 5, PIES_KERNEL_VELOCITY = 6, PIES_KERNEL_HASH = 7,
       PIES_KERNEL_COLLIDE = 8,
       /* projective dynamics (Solver.cpp:228-485) */
       PIES_KERNEL_PD_PREDICT = 9, PIES_KERNEL_PD_LOCAL_DISTANCE = 10, PIES_KERNEL_PD_LOCAL_TET = 11,
       PIES_KERNEL_PD_LOCAL_VOLUME = 12, PIES_KERNEL_PD_RHS = 13, PIES_KERNEL_PD_SPMV = 14,
       PIES_KERNEL_PD_CG_UPDATE = 15, PIES_KERNEL_PD_VELOCITY = 16,
       /* (round 4, graph variants with one launch per CG iteration - pies_count(PIES_PD_CG_SINGLE): PD_SPMV is k_cg1_iter, a whole
        * PCG iteration; PD_CG_UPDATE has no launches; PD_RHS is k_cg1_init when that kernel evaluates the right-hand side
        * itself - tile-resident local step, pies_count(PIES_PD_TILES) -, otherwise k_pd_rhs) */
       /* schedule EXACT: one dependency level of the whole-substep DAG (all projection kinds + floor clamps) */
       PIES_KERNEL_WAVE = 17,
       /* schedule LAYERED: the groups of one parity, LDS resident (units = algorithmic bytes of the projections and
        * per-node steps the launches execute, SURVEY 8d figures) */
       PIES_KERNEL_LAYER = 18, PIES_KERNEL_COUNT = 19 };
int pies_profile_substep(pies_solver_t* s, int kernel, uint32_t* launches, double* total_ms, uint64_t* units);
/* Times one kernel class where it runs: `substeps` whole substeps are launched eagerly (every kernel of the substep
 * runs, so the caches hold what the substep leaves in them) and each launch of the class is bracketed by two HIP events
 * on the solver's stream.  launches = bracketed launches (for PIES_KERNEL_HASH / _COLLIDE one bracket is the whole grid
 * build / resolve pass), total_ms = the sum of their event times, units as in pies_profile_substep.  For the CG classes
 * (PD_SPMV, PD_CG_UPDATE) the solves of this pass do not take the converged early exit, so every bracketed launch does
 * the full work.  bracket_overhead_ms (may be NULL): what one bracket costs by itself, measured in the same pass around one
 * and around two launches of an empty kernel (the event packets and the wait for the end-of-kernel cache write-back: several
 * microseconds, i.e. most of a bracket around a short kernel) - subtract it per launch to compare with rocprofv3's
 * kernel durations.  The node state is put back afterwards. */
int pies_profile_in_situ(pies_solver_t* s, int kernel, uint32_t substeps, uint32_t* launches, double* total_ms, uint64_t* units,
                         double* bracket_overhead_ms);
/* The tile plan of the PD strain + volume local step, computed from the host-side scene (also on PIES_DEVICE_NONE handles): tiles of
 * up to 128 element pairs on up to 128 nodes (one wavefront each).  n_tiles = 0 when the scene keeps per-(element, node) records
 * (no strain + volume pairs).  With info != NULL the plan is copied out at fixed strides per tile: info[t] = nodes | elements << 16,
 * node[128 t + k] = node of tile node k, elem[128 t + e] = host index of element slot e, local[128 t + e] = its four tile-local
 * node indices (8 bits each), nptr[132 t + k] .. nptr[132 t + k + 1] = tile node k's entries in inc[512 t + ..] (element << 2 |
 * corner).  Any array pointer but info may be NULL. */
int pies_get_pd_tile_plan(pies_solver_t* s, uint32_t* n_tiles, uint32_t* info, uint32_t* node, uint32_t* elem, uint32_t* local, uint16_t* nptr,
                          uint16_t* inc, uint32_t tile_capacity);
/* launches per substep of the captured graph, per kernel class (PIES_KERNEL_COUNT entries) */
int pies_launch_counts(pies_solver_t* s, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* PIES_HIP_H */
