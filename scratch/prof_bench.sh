#!/bin/bash
# Round profile of the benchmark: plain run, rocprofv3 kernel trace + stats, and two PMC passes (HBM bytes).
# rocprofv3 7.2 on this pool segfaults once roughly 20k graph kernel nodes have been traced, so the traced
# command uses --steps 12 --warmup 2; PIES_PROFILER_SAFE=1 keeps the library from creating a second graph.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r01}; mkdir -p $OUT
B="--no-cpu-baseline --no-exact --no-extras --no-scale"
echo "== plain"; timeout 300 python bench.py --steps 200 --warmup 20 $B > $OUT/bench_plain.json 2> $OUT/bench_plain.err; tail -c 600 $OUT/bench_plain.json; echo
export PIES_PROFILER_SAFE=1
echo "== kernel trace"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 12 --warmup 2 $B > $OUT/bench_traced.json 2> $OUT/trace.err; echo rc=$?
echo "== pmc fetch"; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 2 --warmup 1 $B --no-kernel-profile > /dev/null 2> $OUT/pmc_fetch.err; echo rc=$?
echo "== pmc write"; timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 2 --warmup 1 $B --no-kernel-profile > /dev/null 2> $OUT/pmc_write.err; echo rc=$?
find $OUT -name "*kernel_trace.csv" -size +8M -delete
du -sh $OUT
