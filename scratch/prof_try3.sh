#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PIES_PROFILER_SAFE=1
B="--no-cpu-baseline --no-exact --no-extras --no-scale"
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t_$name -- python bench.py "$@" $B > gpurun_out/t_$name.out 2> gpurun_out/t_$name.err; echo "$name rc=$? : $(grep -c . gpurun_out/t_$name.out) lines out; $(ls gpurun_out/t_$name/*/ 2>/dev/null | wc -l) files"; }
run v1 --steps 5 --warmup 2 --no-kernel-profile
run v2 --steps 40 --warmup 5 --no-kernel-profile
run v3 --steps 5 --warmup 2
run v5 --steps 12 --warmup 2 --no-kernel-profile
PIES_NO_GRAPH=1 run v4 --steps 40 --warmup 5 --no-kernel-profile
find gpurun_out -name "*kernel_trace.csv" -delete
