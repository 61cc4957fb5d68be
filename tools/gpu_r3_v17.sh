#!/bin/bash
# Visit 17: launch constants from the launcher, inputs touched by two back-to-back LDS-DMA loads, RNG operands by plain
# loads across the barrier: options off one at a time, A/B against the previous build, stamps.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v17}
mkdir -p $OUT
cd $R
C=marlgrid_amd/csrc
(timeout 300 python tools/ab_opts.py 0 1 2 3 2>&1 | grep -v amdgpu.ids > $OUT/ab_opts.txt); cat $OUT/ab_opts.txt
(timeout 300 python tools/ab_fused.py $C/libmarlgrid_hip_ref2.so $C/libmarlgrid_hip.so 2>&1 | grep -v amdgpu.ids > $OUT/ab_fused.txt); tail -n 3 $OUT/ab_fused.txt
(timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps.txt); head -n 24 $OUT/phase_stamps.txt
