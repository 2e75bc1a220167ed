#!/usr/bin/env python3
"""Randomised soak of the point-triangle broad phase against the oracle (tests/test_tri_collisions_gpu.py run_soup: triangle
soup over three size classes, slots shared between distant cells, contact lists compared entry for entry):
python tools/soak_tri_soup.py [scenes] [first seed]."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle_api  # noqa: E402
from pies_amd import capi  # noqa: E402
from test_tri_collisions_gpu import run_soup  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
contacts = 0
for seed in range(first, first + n):
    contacts += run_soup(capi, oracle_api, seed)
    print("seed %d ok, %d contacts so far" % (seed, contacts), flush=True)
print("done: %d scenes" % n)
