"""Scene builders shared by the parity tests and bench.py: the same scene is built through the product's
C ABI and through the oracle, using each side's own generators (node numbering and constraint order
follow Src/PrimitiveUtilities.cpp in the reference)."""
import numpy as np

# synthetic lattices W x H x D (x, y, z; z is the fastest index)
L1K = (10, 10, 10)
L100K = (20, 20, 250)
L250K = (25, 25, 400)
L500K = (50, 100, 100)
L1M = (100, 100, 100)


def pbd_options(mod, iterations, **kw):
    o = dict(solver=mod.PBD, iterations=iterations, timeSubsteps=1, fixedTimestepSize=0.012, gravity=10.0,
             floorHeight=0.0)
    o.update(kw)
    return mod.Options(**o)


def build_beam(solver, dims, w_tet=0.05, w_dist=0.5, translation=(0.0, 5.0, 0.0), scale=1.0, mass=1.0,
               distance=True, tets=True, volume=False, triangles=False):
    """BASELINE configs 1/2: createTetBox-pattern tets (+ optional volume) and createBox-pattern distance
    constraints over the same lattice."""
    W, H, D = dims
    first = solver.count(9)  # NODES
    if tets:
        solver.create_tet_box(W, H, D, translation=translation, scale=scale, w=w_tet, mass=mass, volume=volume,
                              triangles=triangles)
        if distance:
            solver.create_box(W, H, D, scale=scale, w=w_dist, existing_offset=first, triangles=False)
    else:
        solver.create_box(W, H, D, translation=translation, scale=scale, w=w_dist, triangles=triangles)


def perturb(solver, seed, amplitude):
    """Deterministic position/velocity perturbation so that every constraint is active."""
    rng = np.random.default_rng(seed)
    p = solver.positions
    solver.set_positions(p + rng.uniform(-amplitude, amplitude, size=p.shape).astype(np.float32))
    solver.set_velocities(rng.uniform(-1, 1, size=p.shape).astype(np.float32))


def projections_per_substep(solver, mod, iterations):
    n = sum(solver.count(t) for t in (mod.POSITION, mod.DISTANCE, mod.TET, mod.BEND))
    return n * iterations


def delaunay_beam(dims, seed=5, jitter=0.3, min_volume=0.02, max_edge=2.6):
    """An unstructured tetrahedral beam (the BASELINE configs call for a tetgen beam; tetgen is not available, scipy's
    Delaunay triangulation of a jittered lattice stands in): returns (positions n x 3 float32, tets m x 4 uint32,
    edges k x 2 uint32).  Slivers below `min_volume` and the long flat tetrahedra Delaunay puts on the convex hull
    (an edge longer than `max_edge`; tetgen's quality bound would not produce them) are dropped; edges are the unique
    tetrahedron edges."""
    from scipy.spatial import Delaunay
    W, H, D = dims
    rng = np.random.default_rng(seed)
    p = np.stack(np.meshgrid(np.arange(W), np.arange(H), np.arange(D), indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    p += rng.uniform(-jitter, jitter, p.shape)
    tets = Delaunay(p).simplices.astype(np.uint32)
    a, b, c = p[tets[:, 1]] - p[tets[:, 0]], p[tets[:, 2]] - p[tets[:, 0]], p[tets[:, 3]] - p[tets[:, 0]]
    tets = tets[np.abs(np.einsum("ij,ij->i", np.cross(a, b), c)) / 6.0 > min_volume]
    longest = np.zeros(len(tets))
    for i, j in ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)):
        longest = np.maximum(longest, np.linalg.norm(p[tets[:, i]] - p[tets[:, j]], axis=1))
    tets = tets[longest <= max_edge]
    e = np.concatenate([tets[:, [0, 1]], tets[:, [0, 2]], tets[:, [0, 3]], tets[:, [1, 2]], tets[:, [1, 3]], tets[:, [2, 3]]])
    e.sort(axis=1)
    e = np.unique(e, axis=0).astype(np.uint32)
    return (p + [0.0, 5.0, 0.0]).astype(np.float32), tets, e


def build_unstructured(solver, mesh, w_tet=0.05, w_dist=0.5, radius=0.3):
    pos, tets, edges = mesh
    solver.add_nodes_raw(pos, radius=radius)
    solver.add_distance(edges, w_dist)
    solver.add_tet(tets, w_tet)


def boundary_triangles(pos, tets):
    """Faces that belong to exactly one tetrahedron, wound so that the normal points away from the tetrahedron's fourth node
    (what addTriMeshVolume keeps of tetgen's face list, Src/PrimitiveUtilities.cpp:243-266)."""
    tets = np.asarray(tets, dtype=np.int64)
    faces = np.concatenate([tets[:, [1, 2, 3]], tets[:, [0, 3, 2]], tets[:, [0, 1, 3]], tets[:, [0, 2, 1]]])
    opp = np.concatenate([tets[:, 0], tets[:, 1], tets[:, 2], tets[:, 3]])
    key = np.sort(faces, axis=1)
    _, first, counts = np.unique(key, axis=0, return_index=True, return_counts=True)
    keep = first[counts == 1]
    f, o = faces[keep], opp[keep]
    p = np.asarray(pos, dtype=np.float64)
    nrm = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
    flip = np.einsum("ij,ij->i", nrm, p[o] - p[f[:, 0]]) > 0
    f[flip] = f[flip][:, [0, 2, 1]]
    return f.astype(np.uint32)


def build_unstructured_pd(solver, mesh, w=1.0, pin_w=2.0, radius=0.5, triangles=True):
    """BASELINE configs[2] on an unstructured mesh: a strain and a volume constraint per tetrahedron (added pairwise, like
    createTetBox / addTriMeshVolume do), the surface faces as collision triangles, the z ~ 0 end cap pinned."""
    pos, tets, _ = mesh
    solver.add_nodes_raw(pos, radius=radius)
    solver.add_tet(tets, w)
    solver.add_volume(tets, w)
    if triangles:
        solver.add_triangles(boundary_triangles(pos, tets))
    solver.add_position(np.nonzero(np.asarray(pos)[:, 2] < 0.5)[0].astype(np.uint32), pin_w)


def loose_particles(dims, spacing=0.9, jitter=0.05, seed=1234, y0=0.5):
    W, H, D = dims
    rng = np.random.default_rng(seed)
    p = np.stack(np.meshgrid(np.arange(W), np.arange(H), np.arange(D), indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    p = p * spacing + rng.uniform(-jitter, jitter, p.shape) + [0, y0, 0]
    v = np.random.default_rng(4321).uniform(-1, 1, p.shape)
    return p.astype(np.float32), v.astype(np.float32)
