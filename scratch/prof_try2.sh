#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
env | grep -i -E "rocp|preload" | head
export PIES_PROFILER_SAFE=1
CMD="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-exact --no-extras"
echo "== kernel trace"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01_trace -- $CMD > gpurun_out/bench_trace.json 2> gpurun_out/trace.err; grep -v "^    @" gpurun_out/trace.err | tail -3 | cut -c1-200
find gpurun_out/r01_trace -name "*.csv" | head; find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
python - <<'PY'
import os
print({k:v for k,v in os.environ.items() if 'ROC' in k.upper() or 'PRELOAD' in k.upper()})
PY
