#!/usr/bin/env python3
"""Static VALU / scalar / memory instruction counts per basic block of k_layer<256>'s tetrahedral colour loop, in the gfx950 ISA
hipcc emits for the product build (cross-compilation only: runs without a GPU).
usage: python tools/layer_isa.py > profiles/r06_layer_isa.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "layer.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-S", os.path.join(ROOT, "pies_amd/csrc/layer_kernels.hip"), "-o", out, "-O3", "-std=c++17",
                           "-ffp-contract=off", "-fno-fast-math"] + sys.argv[1:] + ["--offload-arch=gfx950", "--cuda-device-only", "-I", os.path.join(ROOT, "include")],
                          stderr=subprocess.DEVNULL)
    txt = open(out).read()
m = re.search(r"^_ZN4pies7k_layerILi256ELi0ELi1EEE.*?s_endpgm", txt, re.S | re.M)
lines = m.group(0).splitlines()
meta = re.search(r"_ZN4pies7k_layerILi256ELi0ELi1EEE.*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", txt, re.S)
print("k_layer<256, 0, 1>: %d lines of ISA, %s VGPRs, scratch %s bytes, occupancy %s" % (len(lines), meta.group(1), meta.group(2), meta.group(3)))
# the tetrahedral colour loop: the first loop whose body holds four 12/16-byte LDS reads followed (later) by four LDS writes and a barrier
blocks, cur = [], None
for l in lines:
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t) or re.match(r"^; %bb\.\d+:", t):
        cur = {"name": t.split(":")[0].replace("; %", ""), "valu": 0, "salu": 0, "mem": 0, "ops": [], "br": []}
        blocks.append(cur)
        continue
    if cur is None or not t or t.startswith((";", ".")):
        continue
    op = t.split()[0]
    cur["ops"].append(op)
    if op.startswith("v_"):
        cur["valu"] += 1
    elif op.startswith(("ds_", "global_", "buffer_", "flat_")):
        cur["mem"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            cur["br"].append(t.replace("\t", " "))
start = next(i for i, b in enumerate(blocks) if sum(o in ("ds_read_b96", "ds_read_b128") for o in b["ops"]) >= 4 and
             sum(o == "global_load_dwordx4" for o in b["ops"]) >= 3)  # (the next colour's records are requested at the head of the body)
end = next(i for i in range(start, len(blocks)) if "s_barrier" in blocks[i]["ops"] or (i > start and sum(o.startswith("ds_write_b") for o in blocks[i]["ops"]) >= 4))
print("tetrahedral colour loop: blocks %s .. %s" % (blocks[start]["name"], blocks[end]["name"]))
tot = 0
for b in blocks[start:end + 1]:
    tot += b["valu"]
    print("  %-12s VALU %4d  scalar %3d  memory %2d   %s" % (b["name"], b["valu"], b["salu"], b["mem"], "; ".join(b["br"])))
print("VALU instructions in the loop body (all paths): %d" % tot)
