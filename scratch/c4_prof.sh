cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PIES_PROFILER_SAFE=1
mkdir -p gpurun_out/c4prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4prof -- python3 scratch/c4_probe.py 2 96 > gpurun_out/c4prof.log 2>&1; echo rc=$?
tail -5 gpurun_out/c4prof.log
find gpurun_out/c4prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/c4_kernel_stats.csv
find gpurun_out/c4prof -name "*kernel_trace.csv" -size +20M -delete
head -30 gpurun_out/c4_kernel_stats.csv | cut -c1-200
