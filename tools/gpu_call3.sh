#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 200 tools/microbench/store_patterns4 0.25 3 > $OUT/store_patterns4.log 2>&1; echo "rc=$?" >> $OUT/store_patterns4.log)
grep "^r2\|rc=" $OUT/store_patterns4.log | cut -c1-120
(cd $R && timeout 600 python -m pytest tests -m gpu -q -x -k "batch_vs_oracle or golden_trajectory or fuzz or auto_reset" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 8 $OUT/pytest.log
(cd $R && timeout 300 python tools/bench_generic.py > $OUT/bench_generic.log 2>&1); cat $OUT/bench_generic.log | tail -n 6
(cd $R && timeout 300 python bench.py --no-pmc --no-strong --no-cpu-baseline --min-seconds 0.6 > $OUT/bench_quick.log 2> $OUT/bench_quick.err); python - <<PY
import json
d=json.load(open("$OUT/bench_quick.log"))
print({k:d[k] for k in ("value","ms_per_step","kernels","closure")})
PY
(cd $R && timeout 120 python tools/time_kernels.py > $OUT/time_kernels.log 2>&1); cat $OUT/time_kernels.log | tail -n 5
