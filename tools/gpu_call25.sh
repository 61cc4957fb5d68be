#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c25
mkdir -p $OUT
cd $R
for k in 1 2; do timeout 200 python tools/placement_probe2.py 2>&1 | grep -v amdgpu > $OUT/probe2_$k.log; done
cat $OUT/probe2_1.log
