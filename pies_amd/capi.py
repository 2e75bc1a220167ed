"""ctypes binding of include/pies_hip.h (the C ABI of libpies_hip.so).

Used by tests/, bench.py and __graft_entry__.py.  There is no CPU path: if the library is missing or no
gfx950 device is present every entry point raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# PIES_LIB: development tools load the diagnostic build (pies_amd/build.py --exp) instead; tests and bench.py never set it
LIB_PATH = os.environ.get("PIES_LIB") or os.path.join(HERE, "lib", "libpies_hip.so")

OK, ERR_INVALID, ERR_HIP, ERR_STATE, ERR_UNSUPPORTED = range(5)
PBD, PD = 0, 1
POSITION, DISTANCE, TET, VOLUME, BEND, SHAPE, GOAL, TRIANGLES, LINES, NODES = range(10)
SCHEDULE_EXACT, SCHEDULE_COLOURED, SCHEDULE_LAYERED = 0, 1, 2
SCHEDULE_DEFAULT = SCHEDULE_LAYERED  # PIES_SCHEDULE_DEFAULT
DEVICE_NONE = -1  # PIES_DEVICE_NONE: host-only handle (scenes and schedules, no compute)
FLAG_RELEASE_HINGE, FLAG_NODE_COLLISIONS, FLAG_TRIANGLE_COLLISIONS, FLAG_REFERENCE_COLLISION_ORDER, FLAG_COLLISION_ORDER = 0, 1, 2, 3, 4
COLLISION_ORDER_REFERENCE, COLLISION_ORDER_GROUPS, COLLISION_ORDER_PAIRS = 0, 1, 2
NODE_POSITION, NODE_PREV_POSITION, NODE_VELOCITY, NODE_RADIUS, NODE_INV_MASS = range(5)
KERNEL_NAMES = ["predict", "position", "distance", "tet", "bend", "floor", "velocity", "hash", "collide",
                "pd_predict", "pd_local_distance", "pd_local_tet", "pd_local_volume", "pd_rhs", "pd_spmv", "pd_cg_update",
                "pd_velocity", "wave", "layer"]
KERNEL_PREDICT, KERNEL_POSITION, KERNEL_DISTANCE, KERNEL_TET, KERNEL_BEND, KERNEL_FLOOR, KERNEL_VELOCITY = range(7)
KERNEL_COUNT = 19
SYSTEM_NNZ = 10
REST_SETS = 11
ROW_STENCILS = 12
PD_TILES = 13
PD_TILE_RECORDS = 14
PD_CG_SINGLE = 15
PD_WINDOW_ENTRIES, PD_WINDOW_HALO = 16, 17
NODE_PAIRS = 18  # the node-node CollisionConstraint extension container (PD)

# every symbol include/pies_hip.h declares (checked by tests/test_capi_symbols.py against the header)
SYMBOLS = [
    "pies_create", "pies_destroy", "pies_clear", "pies_last_error", "pies_abi_version", "pies_default_options",
    "pies_get_options", "pies_add_nodes", "pies_add_nodes_ex", "pies_add_position_constraints",
    "pies_add_distance_constraints", "pies_add_tet_constraints", "pies_add_volume_constraints",
    "pies_add_bend_constraints", "pies_add_triangles", "pies_create_tet_box", "pies_create_box", "pies_create_sheet",
    "pies_create_bend_sheet", "pies_set_flag", "pies_set_schedule", "pies_finalize", "pies_tick", "pies_tick_async",
    "pies_synchronize", "pies_failed", "pies_count", "pies_read_nodes", "pies_write_nodes", "pies_get_ids",
    "pies_get_rest", "pies_get_order", "pies_get_batches", "pies_profile_substep", "pies_launch_counts",
    "pies_set_pcg", "pies_get_pcg_stats", "pies_collision_pairs", "pies_add_shape_constraint",
    "pies_add_goal_constraint", "pies_set_goal_transform", "pies_add_fixed_regions", "pies_update_fixed_regions",
    "pies_add_linked_regions", "pies_create_shape_matching_box", "pies_create_shape_matching_sheet", "pies_get_group",
    "pies_get_tri_contacts", "pies_tick_begin", "pies_export_acquire", "pies_export_release",
    "pies_read_positions_strided", "pies_set_pcg_retry", "pies_get_pcg_health", "pies_profile_in_situ",
    "pies_collision_stats", "pies_get_collision_health", "pies_set_collision_rounds", "pies_set_solver", "pies_debug_pair_state", "pies_set_tuning",
    "pies_get_pd_tile_plan", "pies_get_tri_grid_stats", "pies_set_rest", "pies_get_collision_fallbacks",
    "pies_add_node_pair_constraints",
]


class PiesError(RuntimeError):
    pass


def set_tuning(name, value):
    """pies_set_tuning: process-wide tuning / diagnostic switch (value None unsets)"""
    rc = load().pies_set_tuning(name.encode(), None if value is None else str(value).encode())
    if rc != OK:
        raise PiesError("pies_set_tuning(%s): invalid name" % name)


class Options(C.Structure):
    """pies_options_t == Pies::SolverOptions field for field."""
    _fields_ = [
        ("fixedTimestepSize", C.c_float), ("timeSubsteps", C.c_uint32), ("iterations", C.c_uint32),
        ("collisionStabilizationIterations", C.c_uint32), ("collisionThresholdDistance", C.c_float),
        ("collisionThickness", C.c_float), ("gravity", C.c_float), ("damping", C.c_float),
        ("friction", C.c_float), ("staticFrictionThreshold", C.c_float), ("floorHeight", C.c_float),
        ("gridSpacing", C.c_float), ("threadCount", C.c_uint32), ("solver", C.c_int32),
    ]

    def __init__(self, **kw):
        super().__init__(0.012, 1, 4, 4, 0.1, 0.05, 10.0, 0.006, 0.01, 0.0, 0.0, 2.0, 8, PD)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


_lib = None


def load():
    """Loads libpies_hip.so.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PiesError("%s is missing: build it with `python -m pies_amd.build` (needs hipcc); "
                        "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u32, f32, i32 = C.c_void_p, C.c_uint32, C.c_float, C.c_int
    pf, pu = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    ppv = C.POINTER(C.c_void_p)
    sig = {
        "pies_create": [C.POINTER(Options), i32, ppv],
        "pies_destroy": [vp], "pies_clear": [vp],
        "pies_default_options": [C.POINTER(Options)],
        "pies_get_options": [vp, C.POINTER(Options)],
        "pies_add_nodes": [vp, u32, pf, pu],
        "pies_add_nodes_ex": [vp, u32, pf, pf, pf, pf, pu],
        "pies_add_position_constraints": [vp, u32, pu, f32],
        "pies_add_distance_constraints": [vp, u32, pu, f32],
        "pies_add_tet_constraints": [vp, u32, pu, f32, f32, f32],
        "pies_add_volume_constraints": [vp, u32, pu, f32, f32, f32],
        "pies_add_bend_constraints": [vp, u32, pu, f32],
        "pies_add_node_pair_constraints": [vp, u32, pu],
        "pies_add_triangles": [vp, u32, pu],
        "pies_create_tet_box": [vp, u32, u32, u32, pf, f32, pf, f32, f32, u32],
        "pies_create_box": [vp, u32, u32, u32, pf, f32, f32, i32, u32, u32],
        "pies_create_sheet": [vp, u32, u32, pf, f32, f32, f32],
        "pies_create_bend_sheet": [vp, u32, u32, pf, f32, f32],
        "pies_set_flag": [vp, i32, i32], "pies_set_schedule": [vp, i32],
        "pies_finalize": [vp], "pies_tick": [vp], "pies_tick_async": [vp], "pies_synchronize": [vp],
        "pies_failed": [vp, C.POINTER(i32)],
        "pies_count": [vp, i32, pu],
        "pies_read_nodes": [vp, i32, pf, u32], "pies_write_nodes": [vp, i32, pf, u32],
        "pies_get_ids": [vp, i32, pu, u32], "pies_get_rest": [vp, i32, pf, u32],
        "pies_set_rest": [vp, i32, u32, u32, pf],
        "pies_get_collision_fallbacks": [vp, pu],
        "pies_get_order": [vp, i32, pu, u32], "pies_get_batches": [vp, i32, pu, u32, pu],
        "pies_profile_substep": [vp, i32, pu, C.POINTER(C.c_double), C.POINTER(C.c_uint64)],
        "pies_launch_counts": [vp, pu],
        "pies_set_pcg": [vp, f32, u32],
        "pies_get_pcg_stats": [vp, pf, pu, pu],
        "pies_collision_pairs": [vp, C.POINTER(C.c_uint64)],
        "pies_add_shape_constraint": [vp, u32, pu, f32],
        "pies_add_goal_constraint": [vp, u32, pu, f32, pu],
        "pies_set_goal_transform": [vp, u32, pf],
        "pies_add_fixed_regions": [vp, u32, pf, f32],
        "pies_update_fixed_regions": [vp, u32, pf],
        "pies_add_linked_regions": [vp, u32, pf, f32],
        "pies_create_shape_matching_box": [vp, pf, u32, u32, u32, f32],
        "pies_create_shape_matching_sheet": [vp, u32, u32, pf, f32, f32],
        "pies_get_group": [vp, i32, u32, pu, u32, pu],
        "pies_get_tri_contacts": [vp, pu, u32, pu],
        "pies_get_tri_grid_stats": [vp, pu],
        "pies_tick_begin": [vp, C.POINTER(C.c_uint64)],
        "pies_export_acquire": [vp, C.c_uint64, C.POINTER(pf), pu],
        "pies_export_release": [vp, C.c_uint64],
        "pies_read_positions_strided": [vp, vp, C.c_uint64, u32],
        "pies_set_pcg_retry": [vp, i32],
        "pies_collision_stats": [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
        "pies_get_pcg_health": [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), pu, pu],
        "pies_get_collision_health": [vp, pu, pu, pu, pu],
        "pies_set_collision_rounds": [vp, u32],
        "pies_set_solver": [vp, i32],
        "pies_debug_pair_state": [vp, pf, pf, pu, u32],
        "pies_profile_in_situ": [vp, i32, u32, pu, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double)],
    }
    sig["pies_set_tuning"] = [C.c_char_p, C.c_char_p]
    sig["pies_get_pd_tile_plan"] = [vp, pu, pu, pu, pu, pu, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), u32]
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = None if name == "pies_default_options" else C.c_int
    L.pies_last_error.argtypes = [vp]
    L.pies_last_error.restype = C.c_char_p
    L.pies_abi_version.argtypes = []
    L.pies_abi_version.restype = C.c_int
    _lib = L
    return L


def _pf(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _pu(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


_IDS_PER = {POSITION: 1, DISTANCE: 2, TET: 4, VOLUME: 4, BEND: 4, TRIANGLES: 3, LINES: 1, NODE_PAIRS: 2}
_REST_PER = {DISTANCE: 1, TET: 9, VOLUME: 9, BEND: 1}


class Solver:
    """Object wrapper over one pies_solver_t handle (one HIP device + stream)."""

    def __init__(self, options=None, device=0, **kw):
        self._L = load()
        self.options = options if options is not None else Options(**kw)
        h = C.c_void_p()
        rc = self._L.pies_create(C.byref(self.options), device, C.byref(h))
        if rc != OK:
            raise PiesError("pies_create failed (code %d): no gfx950 HIP device %d, or runtime error" % (rc, device))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.pies_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != OK:
            raise PiesError("pies error %d: %s" % (rc, self._L.pies_last_error(self._h).decode()))

    # -- scene ---------------------------------------------------------------------------------
    def addNodes(self, pos):
        pos = _f32(pos).reshape(-1, 3)
        first = C.c_uint32()
        self._ck(self._L.pies_add_nodes(self._h, len(pos), _pf(pos), C.byref(first)))
        return first.value

    def add_nodes_raw(self, pos, vel=None, radius=None, invMass=None):
        pos = _f32(pos).reshape(-1, 3)
        n = len(pos)
        vel = None if vel is None else _f32(vel).reshape(n, 3)
        radius = None if radius is None else _f32(np.broadcast_to(radius, (n,)))
        invMass = None if invMass is None else _f32(np.broadcast_to(invMass, (n,)))
        first = C.c_uint32()
        self._ck(self._L.pies_add_nodes_ex(self._h, n, _pf(pos), None if vel is None else _pf(vel),
                                           None if radius is None else _pf(radius),
                                           None if invMass is None else _pf(invMass), C.byref(first)))
        return first.value

    def add_position(self, ids, w):
        ids = _u32(ids).reshape(-1)
        self._ck(self._L.pies_add_position_constraints(self._h, len(ids), _pu(ids), w))

    def add_distance(self, ids, w):
        ids = _u32(ids).reshape(-1, 2)
        self._ck(self._L.pies_add_distance_constraints(self._h, len(ids), _pu(ids), w))

    def add_tet(self, ids, w, minStrain=0.8, maxStrain=1.0):
        ids = _u32(ids).reshape(-1, 4)
        self._ck(self._L.pies_add_tet_constraints(self._h, len(ids), _pu(ids), w, minStrain, maxStrain))

    def add_volume(self, ids, w, compression=1.0, stretching=1.0):
        ids = _u32(ids).reshape(-1, 4)
        self._ck(self._L.pies_add_volume_constraints(self._h, len(ids), _pu(ids), w, compression, stretching))

    def add_bend(self, ids, w):
        ids = _u32(ids).reshape(-1, 4)
        self._ck(self._L.pies_add_bend_constraints(self._h, len(ids), _pu(ids), w))

    def add_node_pairs(self, ids):
        """extension: node-node CollisionConstraints (CollisionConstraint.cpp:7-65) over the listed pairs; PD only"""
        ids = _u32(ids).reshape(-1, 2)
        self._ck(self._L.pies_add_node_pair_constraints(self._h, len(ids), _pu(ids)))

    def add_triangles(self, ids):
        ids = _u32(ids).reshape(-1, 3)
        self._ck(self._L.pies_add_triangles(self._h, len(ids), _pu(ids)))

    def add_shape(self, ids, w):
        ids = _u32(ids).reshape(-1)
        self._ck(self._L.pies_add_shape_constraint(self._h, len(ids), _pu(ids), w))

    def add_goal(self, ids, w):
        ids = _u32(ids).reshape(-1)
        k = C.c_uint32()
        self._ck(self._L.pies_add_goal_constraint(self._h, len(ids), _pu(ids), w, C.byref(k)))
        return k.value

    def set_goal_transform(self, goal, m16):
        m = _f32(m16).reshape(16)
        self._ck(self._L.pies_set_goal_transform(self._h, goal, _pf(m)))

    def add_fixed_regions(self, mats, w):
        m = _f32(mats).reshape(-1, 16)
        self._ck(self._L.pies_add_fixed_regions(self._h, len(m), _pf(m), w))

    def update_fixed_regions(self, mats):
        m = _f32(mats).reshape(-1, 16)
        self._ck(self._L.pies_update_fixed_regions(self._h, len(m), _pf(m)))

    def add_linked_regions(self, mats, w):
        m = _f32(mats).reshape(-1, 16)
        self._ck(self._L.pies_add_linked_regions(self._h, len(m), _pf(m), w))

    def create_shape_matching_box(self, translation, cx, cy, cz, w):
        t = _f32(translation)
        self._ck(self._L.pies_create_shape_matching_box(self._h, _pf(t), cx, cy, cz, w))

    def create_shape_matching_sheet(self, W, H, translation=(0, 0, 0), scale=1.0, w=1.0):
        t = _f32(translation)
        self._ck(self._L.pies_create_shape_matching_sheet(self._h, W, H, _pf(t), scale, w))

    def group_ids(self, ctype, k):
        n = C.c_uint32()
        self._ck(self._L.pies_get_group(self._h, ctype, k, None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.uint32)
        self._ck(self._L.pies_get_group(self._h, ctype, k, _pu(out), out.size, C.byref(n)))
        return out

    def create_tet_box(self, W, H, D, translation=(0, 0, 0), scale=1.0, velocity=(0, 0, 0), w=1.0, mass=1.0,
                       volume=True, triangles=True):
        t, v = _f32(translation), _f32(velocity)
        self._ck(self._L.pies_create_tet_box(self._h, W, H, D, _pf(t), scale, _pf(v), w, mass,
                                             (1 if volume else 0) | (2 if triangles else 0)))

    def create_box(self, W, H, D, translation=(0, 0, 0), scale=1.0, w=1.0, existing_offset=None, triangles=True):
        t = _f32(translation)
        self._ck(self._L.pies_create_box(self._h, W, H, D, _pf(t), scale, w, 0 if existing_offset is None else 1,
                                         0 if existing_offset is None else existing_offset, 2 if triangles else 0))

    def create_sheet(self, W, H, translation=(0, 0, 0), scale=1.0, mass=1.0, w=1.0):
        t = _f32(translation)
        self._ck(self._L.pies_create_sheet(self._h, W, H, _pf(t), scale, mass, w))

    def create_bend_sheet(self, W, H, translation=(0, 0, 0), scale=1.0, w=1.0):
        t = _f32(translation)
        self._ck(self._L.pies_create_bend_sheet(self._h, W, H, _pf(t), scale, w))

    def clear(self):
        self._ck(self._L.pies_clear(self._h))

    # -- configuration -------------------------------------------------------------------------
    def set_flag(self, flag, value):
        self._ck(self._L.pies_set_flag(self._h, flag, int(value)))

    def set_schedule(self, schedule):
        self._ck(self._L.pies_set_schedule(self._h, schedule))

    def set_solver(self, solver):
        self._ck(self._L.pies_set_solver(self._h, solver))

    def set_pcg(self, rel_tol, max_iters):
        self._ck(self._L.pies_set_pcg(self._h, rel_tol, max_iters))

    def pcg_stats(self):
        """(max relative residual, max CG iterations used, number of solves) over the last tick."""
        r, it, n = C.c_float(), C.c_uint32(), C.c_uint32()
        self._ck(self._L.pies_get_pcg_stats(self._h, C.byref(r), C.byref(it), C.byref(n)))
        return r.value, it.value, n.value

    def set_pcg_retry(self, enabled):
        self._ck(self._L.pies_set_pcg_retry(self._h, int(enabled)))

    def pcg_health(self):
        """dict: short_solves / solves (kept substeps, since finalize), substeps_retried, budget (CG iterations captured)."""
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        self._ck(self._L.pies_get_pcg_health(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(short_solves=a.value, solves=b.value, substeps_retried=c.value, budget=d.value)

    def finalize(self):
        self._ck(self._L.pies_finalize(self._h))

    # -- hot path ------------------------------------------------------------------------------
    def tick(self, n=1):
        for _ in range(n):
            self._ck(self._L.pies_tick(self._h))

    def tick_async(self, n=1):
        for _ in range(n):
            self._ck(self._L.pies_tick_async(self._h))

    def synchronize(self):
        self._ck(self._L.pies_synchronize(self._h))

    def tick_begin(self):
        """Queues one tick and the asynchronous export of its positions; returns the frame id."""
        f = C.c_uint64()
        self._ck(self._L.pies_tick_begin(self._h, C.byref(f)))
        return f.value

    def export_acquire(self, frame):
        """Blocks until `frame` has landed in pinned host memory; returns an (n, 4) float32 view (x, y, z, invMass)
        that is valid until export_release(frame)."""
        p, n = C.POINTER(C.c_float)(), C.c_uint32()
        self._ck(self._L.pies_export_acquire(self._h, frame, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros((0, 4), np.float32)
        return np.ctypeslib.as_array(p, shape=(n.value, 4))

    def export_release(self, frame):
        self._ck(self._L.pies_export_release(self._h, frame))

    def read_positions_strided(self, stride_floats):
        n = self.count(NODES)
        out = np.zeros((n, stride_floats), dtype=np.float32)
        self._ck(self._L.pies_read_positions_strided(self._h, out.ctypes.data_as(C.c_void_p), 4 * stride_floats, n))
        return out

    def last_error(self):
        return self._L.pies_last_error(self._h).decode()

    @property
    def failed(self):
        f = C.c_int()
        self._ck(self._L.pies_failed(self._h, C.byref(f)))
        return bool(f.value)

    @property
    def tri_collisions(self):
        n = C.c_uint32()
        self._ck(self._L.pies_get_tri_contacts(self._h, None, 0, C.byref(n)))
        out = np.empty((n.value, 4), dtype=np.uint32)
        if n.value:
            self._ck(self._L.pies_get_tri_contacts(self._h, _pu(out), n.value, C.byref(n)))
        return out

    def tri_grid_stats(self):
        """pies_get_tri_grid_stats: the broad phase of the last PD substep's point-triangle detection."""
        out = np.zeros(8, dtype=np.uint32)
        self._ck(self._L.pies_get_tri_grid_stats(self._h, _pu(out)))
        return {"ccd_pairs": int(out[0]), "hit_pairs": int(out[1]), "longest": out[2:5].tolist(), "listed": out[5:8].tolist()}

    @property
    def collision_pairs(self):
        n = C.c_uint64()
        self._ck(self._L.pies_collision_pairs(self._h, C.byref(n)))
        return n.value

    def collision_stats(self):
        """(resolved pairs, candidates looked at) since the last call"""
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(self._L.pies_collision_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pair_state(self):
        n = self.count(NODES)
        sl, ex, dg = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint32)
        self._ck(self._L.pies_debug_pair_state(self._h, sl.ctypes.data_as(C.POINTER(C.c_float)), ex.ctypes.data_as(C.POINTER(C.c_float)),
                                               dg.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        return sl, ex, dg

    def set_collision_rounds(self, rounds):
        self._ck(self._L.pies_set_collision_rounds(self._h, rounds))

    def collision_health(self):
        """pair order: levels and listed pairs of the last pass; passes repeated / inexact since finalize"""
        v = [C.c_uint32() for _ in range(4)]
        self._ck(self._L.pies_get_collision_health(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("levels", "pairs_listed", "passes_repeated", "passes_inexact"), (x.value for x in v)))

    @property
    def collision_fallbacks(self):
        """node-node passes the reference's sequential loop ran in place of a parallel order (pies_get_collision_fallbacks)"""
        v = C.c_uint32()
        self._ck(self._L.pies_get_collision_fallbacks(self._h, C.byref(v)))
        return v.value

    # -- state ---------------------------------------------------------------------------------
    def count(self, what):
        out = C.c_uint32()
        self._ck(self._L.pies_count(self._h, what, C.byref(out)))
        return out.value

    def _read(self, what, cols):
        n = self.count(NODES)
        out = np.empty((n, cols) if cols > 1 else (n,), dtype=np.float32)
        self._ck(self._L.pies_read_nodes(self._h, what, _pf(out), n))
        return out

    def _write(self, what, a):
        a = _f32(a)
        self._ck(self._L.pies_write_nodes(self._h, what, _pf(a), self.count(NODES)))

    positions = property(lambda self: self._read(NODE_POSITION, 3))
    prev_positions = property(lambda self: self._read(NODE_PREV_POSITION, 3))
    velocities = property(lambda self: self._read(NODE_VELOCITY, 3))
    radii = property(lambda self: self._read(NODE_RADIUS, 1))
    inv_masses = property(lambda self: self._read(NODE_INV_MASS, 1))

    def set_positions(self, p):
        self._write(NODE_POSITION, p)

    def set_prev_positions(self, p):
        self._write(NODE_PREV_POSITION, p)

    def set_velocities(self, v):
        self._write(NODE_VELOCITY, v)

    def set_radii(self, r):
        self._write(NODE_RADIUS, r)

    def set_inv_masses(self, m):
        self._write(NODE_INV_MASS, m)

    def ids(self, ctype):
        n, k = self.count(ctype), _IDS_PER[ctype]
        out = np.empty((n, k) if k > 1 else (n,), dtype=np.uint32)
        self._ck(self._L.pies_get_ids(self._h, ctype, _pu(out), out.size))
        return out

    def rest(self, ctype):
        n, k = self.count(ctype), _REST_PER[ctype]
        out = np.empty((n, k) if k > 1 else (n,), dtype=np.float32)
        self._ck(self._L.pies_get_rest(self._h, ctype, _pf(out), out.size))
        return out

    def set_rest(self, ctype, rest, first=0):
        """pies_set_rest: rest data of constraints first .. of a container (DISTANCE target, TET / VOLUME Qinv column-major, BEND angle)"""
        r = np.ascontiguousarray(rest, dtype=np.float32)
        n = r.size // _REST_PER[ctype]
        self._ck(self._L.pies_set_rest(self._h, ctype, first, n, _pf(r)))

    def order(self, ctype):
        out = np.empty(self.count(ctype), dtype=np.uint32)
        self._ck(self._L.pies_get_order(self._h, ctype, _pu(out), out.size))
        return out

    def pd_tile_plan(self):
        """pies_get_pd_tile_plan: None when the scene keeps per-(element, node) records, otherwise a dict of the plan's arrays
        (info: nodes | elements << 16 per tile; node, elem, local: 128 per tile; nptr: 132 per tile; inc: 512 per tile)."""
        nt = C.c_uint32()
        self._ck(self._L.pies_get_pd_tile_plan(self._h, C.byref(nt), None, None, None, None, None, None, 0))
        n = nt.value
        if n == 0:
            return None
        info, node, elem, local = (np.empty(k * n, dtype=np.uint32) for k in (1, 128, 128, 128))
        nptr, inc = np.empty(132 * n, dtype=np.uint16), np.empty(512 * n, dtype=np.uint16)
        p16 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint16))  # noqa: E731
        self._ck(self._L.pies_get_pd_tile_plan(self._h, C.byref(nt), _pu(info), _pu(node), _pu(elem), _pu(local), p16(nptr), p16(inc), n))
        return {"info": info, "node": node.reshape(n, 128), "elem": elem.reshape(n, 128), "local": local.reshape(n, 128),
                "nptr": nptr.reshape(n, 132), "inc": inc.reshape(n, 512)}

    def batches(self, ctype):
        nb = C.c_uint32()
        self._ck(self._L.pies_get_batches(self._h, ctype, None, 0, C.byref(nb)))
        offs = np.empty(nb.value + 1, dtype=np.uint32)
        self._ck(self._L.pies_get_batches(self._h, ctype, _pu(offs), offs.size, C.byref(nb)))
        return offs

    # -- measurement ---------------------------------------------------------------------------
    def launch_counts(self):
        out = np.zeros(KERNEL_COUNT, dtype=np.uint32)
        self._ck(self._L.pies_launch_counts(self._h, _pu(out)))
        return dict(zip(KERNEL_NAMES, out.tolist()))

    def profile_in_situ(self, kernel, substeps=3):
        """Whole substeps launched eagerly with HIP events around every launch of `kernel` (all other kernels run as
        well, so caches are in the state the substep leaves them in); returns (launches, ms, units, bracket_overhead_ms):
        ms is the sum of the brackets, bracket_overhead_ms what one bracket costs around nothing (calibrated in the same
        pass with an empty kernel)."""
        n, ms, units, ov = C.c_uint32(), C.c_double(), C.c_uint64(), C.c_double()
        self._ck(self._L.pies_profile_in_situ(self._h, kernel, substeps, C.byref(n), C.byref(ms), C.byref(units), C.byref(ov)))
        return n.value, ms.value, units.value, ov.value

    def profile_substep(self, kernel):
        """One un-graphed substep with per-dispatch timing of `kernel`; returns (launches, ms, units)."""
        n, ms, units = C.c_uint32(), C.c_double(), C.c_uint64()
        self._ck(self._L.pies_profile_substep(self._h, kernel, C.byref(n), C.byref(ms), C.byref(units)))
        return n.value, ms.value, units.value
