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
