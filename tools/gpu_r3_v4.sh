#!/bin/bash
# Round 3, visit 4: remapped buffers — correctness (stale translations?) and statistics by kind; the raster's
# three-phase inner loop against the previous build.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v4}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 python tools/vmm_reuse_check.py 2>&1 | grep -v amdgpu.ids > $OUT/vmm_reuse_check.txt); cat $OUT/vmm_reuse_check.txt
(cd $R && timeout 300 python tools/placement_stats.py 2>&1 | grep -v amdgpu.ids > $OUT/placement_stats.txt); cut -c1-400 $OUT/placement_stats.txt
(cd $R && timeout 300 python tools/ab_two_libs.py marlgrid_amd/csrc/libmarlgrid_hip_ref.so 9 2>&1 | grep -v amdgpu.ids > $OUT/ab_raster_vs_ref.txt); cat $OUT/ab_raster_vs_ref.txt
(cd $R && timeout 300 python tools/ab_fused.py marlgrid_amd/csrc/libmarlgrid_hip_ref.so marlgrid_amd/csrc/libmarlgrid_hip.so 2>&1 | grep -v amdgpu.ids > $OUT/ab_fused_vs_ref.txt); cat $OUT/ab_fused_vs_ref.txt
