#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 12 $OUT/pytest.log
for mode in "" "--unfused" "" "--unfused"; do
(cd $R && timeout 300 python bench.py --no-pmc --no-strong --no-cpu-baseline --min-seconds 0.8 $mode > $OUT/bench_quick.log 2> $OUT/bench_quick.err); python - <<PY
import json
d=json.load(open("$OUT/bench_quick.log"))
print("$mode", {k:d[k] for k in ("value","ms_per_step","kernels")}, d["roofline"]["raster_only_ms"])
PY
done
