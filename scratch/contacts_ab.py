import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np, bench, scenes
from pies_amd import capi
for rep in range(2):
  for v in ("1", "0"):
    capi.set_tuning("PIES_PD_CG_SINGLE_ROWS", v)
    r = bench.run_pd_contacts(0)
    c5 = bench.run_config5_share(0, with_rooflines=False)
    print("single rows", v, "pd_contacts %.1f (iters %s budget %s contacts %d)  config5 share %.1f max/median %.2f quiet %.1f" % (
        r["value"], r["pcg_max_iterations_used"], r["pcg_health"]["budget"], r["tri_contacts_last_substep"], c5["value"], c5["max_over_median_frame"], c5["value_without_tri_contacts"] or 0), flush=True)
