#!/bin/bash
# which feature makes rocprofv3 --kernel-trace crash?  every step under its own timeout
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== A: launch_bench (graph + ext launch)"; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pa -- ./scratch/launch_bench 2>&1 | grep -v "^    @" | tail -4 | cut -c1-200
echo "== B: bench eager, no profile pass"; PIES_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pb -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact --no-kernel-profile 2>&1 | grep -v "^    @" | tail -4 | cut -c1-300
echo "== C: bench graph, no profile pass"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact --no-kernel-profile 2>&1 | grep -v "^    @" | tail -4 | cut -c1-300
find gpurun_out/pa gpurun_out/pb gpurun_out/pc -name "*.csv" 2>/dev/null | head -20
