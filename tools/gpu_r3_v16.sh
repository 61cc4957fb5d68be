#!/bin/bash
# Visit 16: which part of the restructured head pays — options off one at a time (measurement build), finer stamps.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v16}
mkdir -p $OUT
cd $R
(timeout 300 python tools/ab_opts.py 0 1 2 3 2>&1 | grep -v amdgpu.ids > $OUT/ab_opts.txt); cat $OUT/ab_opts.txt
(timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps.txt); head -n 24 $OUT/phase_stamps.txt
(AB_OPTS=1 timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps_no_touch.txt); head -n 24 $OUT/phase_stamps_no_touch.txt
