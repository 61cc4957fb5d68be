#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c39
mkdir -p $OUT
cd $R
timeout 200 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "fused_step_equals" > $OUT/pytest_first.log 2>&1
rc=$?; tail -n 3 $OUT/pytest_first.log
if [ $rc -ne 0 ]; then echo "first tests failed rc=$rc: stopping"; exit 1; fi
timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu > $OUT/stamps.log; head -n 14 $OUT/stamps.log | grep -v "XCD\|depth"
timeout 120 python tools/time_kernels.py 2>&1 | grep -v amdgpu
B=4096 WL=MarlGrid-3AgentCluttered11x11-v0 timeout 120 python tools/time_kernels.py 2>&1 | grep -v amdgpu
timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/b.log 2> $OUT/b.err
python - $OUT/b.log <<'PY'
import json,sys
b=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
p=b["obs_placement"]
print("%.1f M  %.4f ms  kept %s  candidates %s" % (b["value"]/1e6, b["ms_per_step"], ["%.4f"%x for x in p["kept"]], p.get("candidates")))
PY
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; tail -n 3 $OUT/pytest.log
