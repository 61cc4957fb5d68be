#!/bin/bash
# Visit 18: the parts of the restructured head taken out of the product build one by one (compile-time switches),
# all builds interleaved in one process against the previous build.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v18}
mkdir -p $OUT
cd $R
C=marlgrid_amd/csrc
(CHECK=0 timeout 600 python tools/ab_fused.py $C/libmarlgrid_hip_ref2.so $C/libvar_h1t1m1.so $C/libvar_h1t0m0.so $C/libvar_h0t0m0.so $C/libvar_h0t1m1.so $C/libvar_h0t1m0.so $C/libvar_h0t0m1.so 2>&1 | grep -v amdgpu.ids > $OUT/ab_head_variants.txt); tail -n 8 $OUT/ab_head_variants.txt
