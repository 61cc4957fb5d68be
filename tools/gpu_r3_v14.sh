#!/bin/bash
# Visit 14: the RNG head refill out of line (one copy instead of fourteen between the blocks of the step's hot path):
# A/B of the fused launch against the previous build, stamps inside step_run of a wave's first and last batch for
# both forms, reset cost, the reset / step parity tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v14}
mkdir -p $OUT
cd $R
C=marlgrid_amd/csrc
(timeout 300 python tools/ab_fused.py $C/libmarlgrid_hip_ref2.so $C/libmarlgrid_hip.so 2>&1 | grep -v amdgpu.ids > $OUT/ab_fused.txt); tail -n 4 $OUT/ab_fused.txt
(AB_LIB=libmarlgrid_hip_ab_inl.so timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps_inline.txt); head -n 26 $OUT/phase_stamps_inline.txt
(timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps_outlined.txt); head -n 26 $OUT/phase_stamps_outlined.txt
for lib in libmarlgrid_hip_ref2.so libmarlgrid_hip.so; do
MARLGRID_HIP_LIB=$R/$C/$lib timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $OUT/reset_cost.txt
import os, time, torch
from marlgrid_amd.envs import make
env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=32768, auto_reset=True, strict=False, place_obs=False)
for _ in range(3): env.reset()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): env.reset()
torch.cuda.synchronize(); print(os.path.basename(os.environ["MARLGRID_HIP_LIB"]), "reset() of 32768 envs: %.3f ms" % ((time.perf_counter() - t) / 20 * 1e3))
PY
done
timeout 900 python -m pytest tests -m gpu -x -q -k "reset or golden or respawn or place" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -n 4 $OUT/pytest.log
