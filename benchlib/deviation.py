"""How far apart are the device schedules?  Every PBD schedule is a Gauss-Seidel sweep over the same constraints with
the same per-constraint arithmetic; EXACT visits them in the reference's order (Src/Solver.cpp:58-137: containers in
insertion order, colliding nodes in ascending index), COLOURED / LAYERED in another order.  These helpers run the same
scene under two schedules and report ("report, don't gate"): max |dpos|, the centre-of-mass difference and
each result's own constraint residual norms, after a list of tick counts.  Used by tests/ and by bench.py's
`order_deviation` field."""
import numpy as np


def residuals(solver, mod, lo=0.8, hi=1.0):
    """RMS constraint violation of the current positions: distance |len - rest|; tetrahedral strain = the distance of the
    singular values of F = P Qinv from [lo, hi] (Constraints.cpp:11-37, 76-128)."""
    p = solver.positions.astype(np.float64)
    out = {}
    if solver.count(mod.DISTANCE):
        ids = solver.ids(mod.DISTANCE).astype(np.int64)
        rest = solver.rest(mod.DISTANCE).astype(np.float64)
        out["distance_rms"] = float(np.sqrt(np.mean((np.linalg.norm(p[ids[:, 1]] - p[ids[:, 0]], axis=1) - rest) ** 2)))
    if solver.count(mod.TET):
        ids = solver.ids(mod.TET).astype(np.int64)
        q = solver.rest(mod.TET).astype(np.float64).reshape(-1, 3, 3).transpose(0, 2, 1)  # column-major -> Q[n, row, col]
        P = np.stack([p[ids[:, 1]] - p[ids[:, 0]], p[ids[:, 2]] - p[ids[:, 0]], p[ids[:, 3]] - p[ids[:, 0]]], axis=2)
        sv = np.linalg.svd(P @ q, compute_uv=False)
        out["tet_strain_rms"] = float(np.sqrt(np.mean(np.sum((sv - np.clip(sv, lo, hi)) ** 2, axis=1))))
    return out


def compare(make, mod, variants, ticks=(1, 5, 10)):
    """make(variant) -> a fresh solver of the scene configured for `variant`; variants[0] is the reference order.
    Returns {variant: {ticks: {...}}} for the other variants."""
    solvers = {v: make(v) for v in variants}
    ref = variants[0]
    out = {v: {} for v in variants[1:]}
    done = 0
    for t in sorted(ticks):
        for s in solvers.values():
            s.tick(t - done)
        done = t
        pr = solvers[ref].positions.astype(np.float64)
        rr = residuals(solvers[ref], mod)
        for v in variants[1:]:
            pv = solvers[v].positions.astype(np.float64)
            entry = {
                "max_abs_dpos": float(np.abs(pv - pr).max()),
                "rms_dpos": float(np.sqrt(np.mean(np.sum((pv - pr) ** 2, axis=1)))),
                "centre_of_mass_delta": float(np.abs(pv.mean(0) - pr.mean(0)).max()),
                "extent_delta": float(np.abs((pv.max(0) - pv.min(0)) - (pr.max(0) - pr.min(0))).max()),
                "residuals": residuals(solvers[v], mod),
                "residuals_reference_order": rr,
                "finite": bool(np.isfinite(pv).all() and np.isfinite(pr).all()),
            }
            out[v]["after_%d_ticks" % t] = entry
    for s in solvers.values():
        s.close()
    return out
