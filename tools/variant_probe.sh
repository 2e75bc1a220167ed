#!/bin/bash
# Development aid, run ON the GPU box: rebuilds pd_cg1_kernels.hip with -DPIES_ITER_VARIANT=v for each v given and runs the CG probe.
# usage: bash tools/variant_probe.sh "0 1 2 3" "pd1m pd1m_streamed unstructured"
cd $GRAFT_REPO_ROOT
OBJ=pies_amd/lib/obj
for v in $1; do
  echo "==== variant $v"
  /opt/rocm/bin/hipcc -c pies_amd/csrc/pd_cg1_kernels.hip -o $OBJ/pd_cg1_kernels.hip.o -DPIES_ITER_VARIANT=$v -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I include --offload-arch=gfx950 || exit 1
  /opt/rocm/bin/hipcc -shared -o pies_amd/lib/libpies_hip.so $OBJ/*.o --offload-arch=gfx950 -Wl,--no-undefined || exit 1
  timeout -k 10 400 python tools/probe_cg.py $2 2>&1 | grep -v "pd_local_tet"
done
