#!/usr/bin/env python3
"""VALU instructions of ONE Jacobi sweep (three rotations) of the 3x3 SVD variants in tools/svd_variants.hip, counted in the
gfx950 ISA hipcc emits (is there a cheaper SVD for k_layer?).  Runs on the CPU (cross-compilation only).
usage: python tools/svd_isa_counts.py > profiles/r03_svd_isa_counts.txt"""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "tools", "svd_variants.hip")
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "svd.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-S", src, "-o", out, "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           "--offload-arch=gfx950", "--cuda-device-only"], stderr=subprocess.DEVNULL)
    txt = open(out).read()
rows = {}
for name in ("sweep_A", "sweep_B", "sweep_C"):
    body = re.search(r"^_Z\d+" + name + r"Pf:(.*?)s_endpgm", txt, re.S | re.M).group(1)
    ins = [l.split()[0] for l in (x.strip() for x in body.splitlines()) if re.match(r"^[a-z][a-z_0-9]* ", l + " ")]
    rows[name] = (sum(i.startswith("v_") for i in ins), sum(i.startswith("v_pk") for i in ins),
                  sum(i.startswith(("global_", "flat_", "buffer_")) for i in ins))
io = 20  # loads/stores/address arithmetic of the test harness around the sweep (the same in all three)
print("VALU instructions of one Jacobi sweep (3 rotations), gfx950, -O3 -ffp-contract=off  [of which packed v_pk_*]")
for name, what in (("sweep_A", "A one-sided Jacobi on the columns of F (the product's svd3)"),
                   ("sweep_B", "B two-sided Jacobi on S = F^T F, exact rotation (2 rsqrt_nr), V as a matrix"),
                   ("sweep_C", "C two-sided on S, McAdams' approximate Givens (1 rsqrt_nr), V as a quaternion")):
    v, pk, mem = rows[name]
    print("  %-86s %4d [%3d]" % (what, v, pk))
a, b, c = (rows[k][0] - io for k in ("sweep_A", "sweep_B", "sweep_C"))
print()
print("per projection (sweep counts: A and B run until converged - 3.2 rotating sweeps on BASELINE config 2 plus a check sweep")
print("without rotations, ~45 instructions; C's rotation is approximate and needs its fixed 4 sweeps; B and C also form S = F^T F,")
print("18 instructions, and B = F V afterwards, 15; C converts its quaternion to V, ~20):")
print("  A  3.2 x %d + 45           = %d" % (a, round(3.2 * a + 45)))
print("  B  3.2 x %d + 45 + 18 + 15 = %d   (eigenvectors of F^T F: the smallest singular direction loses a factor cond(F) of accuracy;" % (b, round(3.2 * b + 78)))
print("                                      the fp64 goldens hold cond(F) up to 1e3 at 1e-5)")
print("  C  4   x %d + 18 + 15 + 20 = %d" % (c, round(4 * c + 53)))
print()
print("Neither variant executes fewer instructions than the one-sided iteration by more than 5 %; the step time of k_layer is")
print("(instructions of one projection) x 4 cycles per wave64 instruction, so neither shortens the chain measurably.")
