#!/bin/bash
# Round 3, visit 2: placement — handles permuted / traded between buffers; the fused kernel's launch options A/B
# + phase stamps; strict overhead on one set of buffers.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 python tools/placement_vmm2.py 2>&1 | grep -v amdgpu.ids > $OUT/placement_vmm2.txt); cat $OUT/placement_vmm2.txt
(cd $R && timeout 400 python tools/ab_fused.py ref 0 1 257 513 2>&1 | grep -v amdgpu.ids > $OUT/ab_fused.txt); cat $OUT/ab_fused.txt
(cd $R && OPTS="0 1 0x101" timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps_opts.txt); grep -v "by look-ahead\|by XCD" $OUT/phase_stamps_opts.txt | grep -A9 "^MG_RENDER_OPT" | grep -v "mg_render_obs" | head -60
(cd $R && timeout 300 python tools/strict_overhead.py 2>&1 | grep -v amdgpu.ids > $OUT/strict_overhead.json); cat $OUT/strict_overhead.json
