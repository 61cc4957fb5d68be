#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 700 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 8 $OUT/pytest.log
(cd $R && timeout 300 python tools/ab_render.py 0 D1 D2 D4 D8 4 > $OUT/ab_render.log 2>&1); tail -n 7 $OUT/ab_render.log
(cd $R && B=262144 timeout 300 python tools/ab_render.py 0 D1 > $OUT/ab_render_big.log 2>&1); tail -n 3 $OUT/ab_render_big.log
(cd $R && timeout 300 python tools/bench_generic.py > $OUT/bench_generic.log 2>&1); cat $OUT/bench_generic.log | tail -n 6
(cd $R && timeout 300 python bench.py --no-pmc --no-strong --no-cpu-baseline --min-seconds 0.6 > $OUT/bench_quick.log 2> $OUT/bench_quick.err); python - <<PY
import json
d=json.load(open("$OUT/bench_quick.log"))
print({k:d[k] for k in ("value","ms_per_step","kernels")})
PY
