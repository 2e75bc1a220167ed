"""ctypes binding of the CPU oracle (oracle/pies_oracle.cpp).  Test infrastructure only: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by pies_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# PIES_ORACLE_LIB: tests/test_sanitizers.py points a child process at the ASan/UBSan build (make -C oracle asan)
_SO = os.environ.get("PIES_ORACLE_LIB") or os.path.join(_ROOT, "oracle", "_build", "libpies_oracle.so")

POSITION, DISTANCE, TET, VOLUME, BEND, SHAPE, GOAL, TRIANGLES, LINES, NODES, STATICS, TRI_CONTACTS = range(12)
NODE_PAIRS = 18
FLAG_RELEASE_HINGE, FLAG_NODE_COLLISIONS, FLAG_COLLISION_RULE, FLAG_TRIANGLE_COLLISIONS, FLAG_PD_SOLVE_FP64, FLAG_SVD_PLAIN = 0, 1, 2, 3, 4, 5
PBD, PD = 0, 1


class Options(C.Structure):
    """Field-for-field Pies::SolverOptions (/root/reference/Include/Pies/Solver.h:23-38)."""
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


def build(force=False):
    """The portable build (AVX2 + FMA), or - PIES_ORACLE_NATIVE=1, set by bench.py's cpu_baseline leg - the
    -O3 -march=native one, which is always compiled on the machine that runs it."""
    native = os.environ.get("PIES_ORACLE_NATIVE") == "1"
    so = _SO.replace("libpies_oracle.so", "libpies_oracle_native.so") if native else _SO
    if force or native or not os.path.exists(so) or any(
            os.path.getmtime(os.path.join(_ROOT, "oracle", f)) > os.path.getmtime(so)
            for f in ("pies_oracle.cpp", "ora_math.h", "Makefile")):
        if native and os.path.exists(so):
            os.remove(so)
        subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")] + (["native"] if native else []))
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp, u32, f32, i32 = C.c_void_p, C.c_uint32, C.c_float, C.c_int
        pf, pu = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.ora_create.restype = vp
        L.ora_create.argtypes = [vp]
        L.ora_destroy.argtypes = [vp]
        L.ora_options_size.restype = u32
        L.ora_set_flag.argtypes = [vp, i32, i32]
        L.ora_failed.argtypes = [vp]
        L.ora_add_nodes_raw.restype = u32
        L.ora_add_nodes_raw.argtypes = [vp, u32, pf, pf, pf, pf]
        L.ora_add_nodes.restype = u32
        L.ora_add_nodes.argtypes = [vp, u32, pf]
        L.ora_add_distance.argtypes = [vp, u32, pu, f32]
        L.ora_add_position.argtypes = [vp, u32, pu, f32]
        L.ora_add_tet.argtypes = [vp, u32, pu, f32, f32, f32]
        L.ora_add_volume.argtypes = [vp, u32, pu, f32, f32, f32]
        L.ora_add_bend.argtypes = [vp, u32, pu, f32]
        L.ora_add_node_pairs.argtypes = [vp, u32, pu]
        L.ora_add_shape.argtypes = [vp, u32, pu, f32]
        L.ora_add_goal.argtypes = [vp, u32, pu, f32]
        L.ora_set_goal_transform.argtypes = [vp, u32, pf]
        L.ora_add_triangles.argtypes = [vp, u32, pu]
        L.ora_add_fixed_regions.argtypes = [vp, u32, pf, f32]
        L.ora_update_fixed_regions.argtypes = [vp, u32, pf]
        L.ora_add_linked_regions.argtypes = [vp, u32, pf, f32]
        L.ora_create_shape_matching_box.argtypes = [vp, pf, u32, u32, u32, f32]
        L.ora_create_shape_matching_sheet.argtypes = [vp, u32, u32, pf, f32, f32]
        L.ora_group_size.restype = u32
        L.ora_group_size.argtypes = [vp, i32, u32]
        L.ora_group_ids.argtypes = [vp, i32, u32, pu]
        L.ora_create_tet_box.argtypes = [vp, u32, u32, u32, pf, f32, pf, f32, f32, u32]
        L.ora_create_box.argtypes = [vp, u32, u32, u32, pf, f32, f32, i32, u32, u32]
        L.ora_create_sheet.argtypes = [vp, u32, u32, pf, f32, f32, f32]
        L.ora_create_bend_sheet.argtypes = [vp, u32, u32, pf, f32, f32]
        L.ora_permute.argtypes = [vp, i32, pu, u32]
        L.ora_set_collision_order.argtypes = [vp, pu, u32]
        L.ora_set_batches.argtypes = [vp, i32, pu, u32]
        L.ora_set_threads.argtypes = [vp, i32]
        L.ora_set_reference_threads.argtypes = [vp, i32]
        L.ora_count.restype = u32
        L.ora_count.argtypes = [vp, i32]
        L.ora_stat_collision_pairs.restype = C.c_uint64
        L.ora_stat_collision_pairs.argtypes = [vp]
        L.ora_get.argtypes = [vp, i32, pf]
        L.ora_set.argtypes = [vp, i32, pf]
        L.ora_get_ids.argtypes = [vp, i32, pu]
        L.ora_get_rest.argtypes = [vp, i32, pf]
        L.ora_tick.argtypes = [vp]
        L.ora_svd3.argtypes = [pf, pf, pf, pf]
        L.ora_project_tet.argtypes = [pf, pf, f32, f32, pf]
        L.ora_project_volume.argtypes = [pf, pf, f32, f32, pf]
        L.ora_project_distance.argtypes = [pf, f32, pf]
        L.ora_project_bend.argtypes = [pf, pf, f32, pf]
        L.ora_tet_rest.argtypes = [pf, pf, pf]
        L.ora_node_range.argtypes = [pf, f32, f32, C.POINTER(C.c_int64)]
        L.ora_point_triangle_ccd.argtypes = [pf, f32, pf]
        L.ora_point_triangle_ccd.restype = C.c_int
        L.ora_get_tri_collisions.argtypes = [vp, pu]
        assert L.ora_options_size() == C.sizeof(Options)
        _lib = L
    return _lib


def _pf(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _pu(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


_IDS_PER = {POSITION: 1, DISTANCE: 2, TET: 4, VOLUME: 4, BEND: 4, TRIANGLES: 3, LINES: 1}
_REST_PER = {DISTANCE: 1, TET: 9, VOLUME: 9, BEND: 1}


class OracleSolver:
    """Thin object wrapper; method names follow Pies::Solver where one exists."""

    def __init__(self, options=None, **kw):
        self.options = options if options is not None else Options(**kw)
        self._h = lib().ora_create(C.byref(self.options))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ora_destroy(self._h)
            self._h = None

    # -- scene --------------------------------------------------------------------------------
    def addNodes(self, pos):
        pos = _f32(pos).reshape(-1, 3)
        return lib().ora_add_nodes(self._h, len(pos), _pf(pos))

    def add_nodes_raw(self, pos, vel=None, radius=None, invMass=None):
        pos = _f32(pos).reshape(-1, 3)
        n = len(pos)
        vel = None if vel is None else _f32(vel).reshape(n, 3)
        radius = None if radius is None else _f32(np.broadcast_to(radius, (n,)))
        invMass = None if invMass is None else _f32(np.broadcast_to(invMass, (n,)))
        return lib().ora_add_nodes_raw(self._h, n, _pf(pos), None if vel is None else _pf(vel),
                                       None if radius is None else _pf(radius),
                                       None if invMass is None else _pf(invMass))

    def add_distance(self, ids, w):
        ids = _u32(ids).reshape(-1, 2)
        lib().ora_add_distance(self._h, len(ids), _pu(ids), w)

    def add_position(self, ids, w):
        ids = _u32(ids).reshape(-1)
        lib().ora_add_position(self._h, len(ids), _pu(ids), w)

    def add_tet(self, ids, w, minStrain=0.8, maxStrain=1.0):
        ids = _u32(ids).reshape(-1, 4)
        lib().ora_add_tet(self._h, len(ids), _pu(ids), w, minStrain, maxStrain)

    def add_volume(self, ids, w, compression=1.0, stretching=1.0):
        ids = _u32(ids).reshape(-1, 4)
        lib().ora_add_volume(self._h, len(ids), _pu(ids), w, compression, stretching)

    def add_bend(self, ids, w):
        ids = _u32(ids).reshape(-1, 4)
        lib().ora_add_bend(self._h, len(ids), _pu(ids), w)

    def add_node_pairs(self, ids):
        ids = _u32(ids).reshape(-1, 2)
        lib().ora_add_node_pairs(self._h, len(ids), _pu(ids))

    def add_shape(self, ids, w):
        ids = _u32(ids).reshape(-1)
        lib().ora_add_shape(self._h, len(ids), _pu(ids), w)

    def add_goal(self, ids, w):
        ids = _u32(ids).reshape(-1)
        lib().ora_add_goal(self._h, len(ids), _pu(ids), w)

    def set_goal_transform(self, goal, m16):
        m = _f32(m16).reshape(16)
        lib().ora_set_goal_transform(self._h, goal, _pf(m))

    def add_triangles(self, ids):
        ids = _u32(ids).reshape(-1, 3)
        lib().ora_add_triangles(self._h, len(ids), _pu(ids))

    def add_fixed_regions(self, mats, w):
        m = _f32(mats).reshape(-1, 16)
        lib().ora_add_fixed_regions(self._h, len(m), _pf(m), w)

    def update_fixed_regions(self, mats):
        m = _f32(mats).reshape(-1, 16)
        lib().ora_update_fixed_regions(self._h, len(m), _pf(m))

    def add_linked_regions(self, mats, w):
        m = _f32(mats).reshape(-1, 16)
        lib().ora_add_linked_regions(self._h, len(m), _pf(m), w)

    def create_shape_matching_box(self, translation, cx, cy, cz, w):
        t = _f32(translation)
        lib().ora_create_shape_matching_box(self._h, _pf(t), cx, cy, cz, w)

    def create_shape_matching_sheet(self, W, H, translation=(0, 0, 0), scale=1.0, w=1.0):
        t = _f32(translation)
        lib().ora_create_shape_matching_sheet(self._h, W, H, _pf(t), scale, w)

    def group_ids(self, ctype, k):
        out = np.empty(lib().ora_group_size(self._h, ctype, k), dtype=np.uint32)
        lib().ora_group_ids(self._h, ctype, k, _pu(out))
        return out

    def create_tet_box(self, W, H, D, translation=(0, 0, 0), scale=1.0, velocity=(0, 0, 0), w=1.0, mass=1.0,
                       volume=True, triangles=True):
        t, v = _f32(translation), _f32(velocity)
        lib().ora_create_tet_box(self._h, W, H, D, _pf(t), scale, _pf(v), w, mass, (1 if volume else 0) | (2 if triangles else 0))

    def create_box(self, W, H, D, translation=(0, 0, 0), scale=1.0, w=1.0, existing_offset=None, triangles=True):
        t = _f32(translation)
        lib().ora_create_box(self._h, W, H, D, _pf(t), scale, w, 0 if existing_offset is None else 1,
                             0 if existing_offset is None else existing_offset, 2 if triangles else 0)

    def create_sheet(self, W, H, translation=(0, 0, 0), scale=1.0, mass=1.0, w=1.0):
        t = _f32(translation)
        lib().ora_create_sheet(self._h, W, H, _pf(t), scale, mass, w)

    def create_bend_sheet(self, W, H, translation=(0, 0, 0), scale=1.0, w=1.0):
        t = _f32(translation)
        lib().ora_create_bend_sheet(self._h, W, H, _pf(t), scale, w)

    def permute(self, ctype, perm):
        perm = _u32(perm)
        lib().ora_permute(self._h, ctype, _pu(perm), len(perm))

    def set_batches(self, ctype, offsets):
        """Conflict-free batches (n+1 slot offsets) of a container, for the multi-threaded sweep (set_threads > 1)."""
        offsets = _u32(offsets)
        lib().ora_set_batches(self._h, ctype, _pu(offsets), max(0, len(offsets) - 1))

    def set_reference_threads(self, on):
        """the reference's own threading: 16 threads for the node-hash insert, threadCount for PD collision detection"""
        lib().ora_set_reference_threads(self._h, int(on))

    def set_threads(self, n):
        lib().ora_set_threads(self._h, int(n))

    def set_collision_order(self, order):
        order = _u32(order)
        lib().ora_set_collision_order(self._h, _pu(order), len(order))

    def set_flag(self, flag, value):
        lib().ora_set_flag(self._h, flag, int(value))

    # -- state --------------------------------------------------------------------------------
    def count(self, what):
        return lib().ora_count(self._h, what)

    def _get(self, what, cols):
        n = self.count(NODES)
        out = np.empty((n, cols) if cols > 1 else (n,), dtype=np.float32)
        lib().ora_get(self._h, what, _pf(out))
        return out

    positions = property(lambda self: self._get(0, 3))
    prev_positions = property(lambda self: self._get(1, 3))
    velocities = property(lambda self: self._get(2, 3))
    radii = property(lambda self: self._get(3, 1))
    inv_masses = property(lambda self: self._get(4, 1))

    def set_positions(self, p):
        p = _f32(p)
        lib().ora_set(self._h, 0, _pf(p))

    def set_prev_positions(self, p):
        p = _f32(p)
        lib().ora_set(self._h, 1, _pf(p))

    def set_velocities(self, v):
        v = _f32(v)
        lib().ora_set(self._h, 2, _pf(v))

    def set_radii(self, r):
        r = _f32(r)
        lib().ora_set(self._h, 3, _pf(r))

    def set_inv_masses(self, m):
        m = _f32(m)
        lib().ora_set(self._h, 4, _pf(m))

    def ids(self, ctype):
        n = self.count(ctype)
        k = _IDS_PER[ctype]
        out = np.empty((n, k) if k > 1 else (n,), dtype=np.uint32)
        lib().ora_get_ids(self._h, ctype, _pu(out))
        return out

    def rest(self, ctype):
        n = self.count(ctype)
        k = _REST_PER[ctype]
        out = np.empty((n, k) if k > 1 else (n,), dtype=np.float32)
        lib().ora_get_rest(self._h, ctype, _pf(out))
        return out

    def tick(self, n=1):
        for _ in range(n):
            lib().ora_tick(self._h)

    @property
    def tri_collisions(self):
        out = np.empty((self.count(TRI_CONTACTS), 4), dtype=np.uint32)
        lib().ora_get_tri_collisions(self._h, _pu(out))
        return out

    @property
    def failed(self):
        return bool(lib().ora_failed(self._h))

    @property
    def collision_pairs(self):
        return lib().ora_stat_collision_pairs(self._h)


# -- single-operation entry points ---------------------------------------------------------------
def svd3(a):
    a = _f32(a).reshape(3, 3)
    s, b, v = np.empty(3, np.float32), np.empty((3, 3), np.float32), np.empty((3, 3), np.float32)
    lib().ora_svd3(_pf(a), _pf(s), _pf(b), _pf(v))
    return s, b, v


def project_tet(x, qinv, minStrain=0.8, maxStrain=1.0):
    x, qinv = _f32(x).reshape(4, 3), _f32(qinv).reshape(9)
    out = np.empty((4, 3), np.float32)
    lib().ora_project_tet(_pf(x), _pf(qinv), minStrain, maxStrain, _pf(out))
    return out


def project_volume(x, qinv, minOmega=1.0, maxOmega=1.0):
    x, qinv = _f32(x).reshape(4, 3), _f32(qinv).reshape(9)
    out = np.empty((4, 3), np.float32)
    lib().ora_project_volume(_pf(x), _pf(qinv), minOmega, maxOmega, _pf(out))
    return out


def project_distance(x, target):
    x = _f32(x).reshape(2, 3)
    out = np.empty((2, 3), np.float32)
    lib().ora_project_distance(_pf(x), target, _pf(out))
    return out


def project_bend(x, invMass, angle):
    x, im = _f32(x).reshape(4, 3), _f32(invMass).reshape(4)
    out = np.empty((4, 3), np.float32)
    lib().ora_project_bend(_pf(x), _pf(im), angle, _pf(out))
    return out


def point_triangle_ccd(ap0, ab0, ac0, ap1, ab1, ac1, threshold):
    v = _f32(np.stack([ap0, ab0, ac0, ap1, ab1, ac1])).reshape(18)
    t = C.c_float()
    hit = lib().ora_point_triangle_ccd(_pf(v), threshold, C.byref(t))
    return bool(hit), t.value


def tet_rest(x):
    x = _f32(x).reshape(4, 3)
    q, a = np.empty(9, np.float32), np.empty((4, 4), np.float32)
    lib().ora_tet_rest(_pf(x), _pf(q), _pf(a))
    return q, a


def node_range(pos, radius, scale):
    pos = _f32(pos).reshape(3)
    out = np.empty(6, np.int64)
    lib().ora_node_range(_pf(pos), radius, scale, out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out
