#!/bin/bash
# Visit 15: the prologue restructured (inputs touched ahead, the launch's constants worked out while they travel, the
# prologue as the env loop's first thing, no wait in the middle of its round trip) and the RNG head's operands
# requested ahead: A/B against the previous build (with its 130-step identity check), stamps, the -m gpu suite.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v15}
mkdir -p $OUT
cd $R
C=marlgrid_amd/csrc
(timeout 300 python tools/ab_fused.py $C/libmarlgrid_hip_ref2.so $C/libmarlgrid_hip.so 2>&1 | grep -v amdgpu.ids > $OUT/ab_fused.txt); tail -n 4 $OUT/ab_fused.txt
(timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps.txt); head -n 36 $OUT/phase_stamps.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -n 4 $OUT/pytest.log
(B=262144 timeout 300 python tools/ab_fused.py $C/libmarlgrid_hip_ref2.so $C/libmarlgrid_hip.so 2>&1 | grep -v amdgpu.ids > $OUT/ab_fused_262144.txt); tail -n 3 $OUT/ab_fused_262144.txt
