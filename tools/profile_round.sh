#!/bin/bash
# Round profile, run ON the GPU box (gpurun -- 'bash tools/profile_round.sh r02'): for every workload bench.py reports -
# config 2 (headline), config 3 (PD), config 4 (node-node collisions), the 250k contact scene and the 125k / 29k-contact scene - a rocprofv3
# --kernel-trace --stats pass of WHOLE substeps (in situ) and two --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one
# pass on gfx950), plus a VALU counter pass for the dominant kernel.  Summaries are copied into profiles/ by
# tools/summarize_profile.py.  rocprofv3 7.2 on this pool crashes on large traced graphs / a second graph instantiation:
# PIES_PROFILER_SAFE=1 makes the library launch eagerly-safe sequences and synchronise every tick.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r04}; OUT=gpurun_out/$R; mkdir -p $OUT
LIST=${2:-"config2 config3 config4 contacts pdcontacts pbd1m pd1m"}   # (a gpurun call is limited to 20 minutes: the round is taken in two or three calls)
export PIES_PROFILER_SAFE=1
for W in $LIST; do
  echo "== $W trace"; timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -- python3 tools/profile_target.py $W 6 > $OUT/trace_$W.log 2>&1; echo rc=$?
  echo "== $W pmc fetch"; timeout -k 10 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$W -- python3 tools/profile_target.py $W 2 > $OUT/fetch_$W.log 2>&1; echo rc=$?
  echo "== $W pmc write"; timeout -k 10 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$W -- python3 tools/profile_target.py $W 2 > $OUT/write_$W.log 2>&1; echo rc=$?
done
if [[ " $LIST " == *" config4 "* ]]; then echo "== config4 valu"; timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/valu_config4 -- python3 tools/profile_target.py config4 2 > $OUT/valu_config4.log 2>&1; echo rc=$?; fi
if [[ " $LIST " == *" config2 "* ]]; then echo "== config2 valu"; timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/valu_config2 -- python3 tools/profile_target.py config2 2 > $OUT/valu_config2.log 2>&1; echo rc=$?; fi
find $OUT -name "*kernel_trace.csv" -size +6M -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
du -sh $OUT
