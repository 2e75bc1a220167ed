#!/bin/bash
# rocprofv3 evidence for the CG iteration with working launches (run ON the GPU box: gpurun -- 'bash tools/profile_cg.sh r05 "pd1m_streamed pd1m_work pd_unstructured"'):
# per workload a --kernel-trace --stats pass and two --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/profile_target.py; summaries
# are copied into profiles/ by tools/summarize_profile.py.  Third argument: workloads for a VALU counter pass (e.g. "config3 pd1m").
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r05}; OUT=gpurun_out/$R; mkdir -p $OUT
LIST=${2:-"pd1m_streamed pd1m_work pd_unstructured"}
export PIES_PROFILER_SAFE=1
for W in $LIST; do
  N=4; [[ $W == pd_unstructured ]] && N=6
  echo "== $W trace"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -- python3 tools/profile_target.py $W $N > $OUT/trace_$W.log 2>&1; echo rc=$?
  echo "== $W pmc fetch"; timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$W -- python3 tools/profile_target.py $W 2 > $OUT/fetch_$W.log 2>&1; echo rc=$?
  echo "== $W pmc write"; timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$W -- python3 tools/profile_target.py $W 2 > $OUT/write_$W.log 2>&1; echo rc=$?
done
# VALU issue of the PD local step (k_pd_local_tiles is bound by it, not by HBM): wave-instructions per launch
for W in ${3:-}; do
  echo "== $W valu"; timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/valu_$W -- python3 tools/profile_target.py $W 2 > $OUT/valu_$W.log 2>&1; echo rc=$?
done
find $OUT -name "*kernel_trace.csv" -size +6M -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
du -sh $OUT
