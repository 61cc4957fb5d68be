#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c7
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 12 $OUT/pytest.log
(cd $R && timeout 300 python tools/ab_render.py 0 E 0 E > $OUT/ab_render.log 2>&1); tail -n 5 $OUT/ab_render.log
(cd $R && B=262144 timeout 300 python tools/ab_render.py 0 E > $OUT/ab_render_big.log 2>&1); tail -n 3 $OUT/ab_render_big.log
(cd $R && timeout 300 python bench.py --no-pmc --no-strong --no-cpu-baseline --min-seconds 0.8 > $OUT/bench_quick.log 2> $OUT/bench_quick.err); python - <<PY
import json
d=json.load(open("$OUT/bench_quick.log"))
print({k:d[k] for k in ("value","ms_per_step","kernels")}, d["roofline"]["raster_only_ms"])
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline --min-seconds 0.3 > $OUT/prof.log 2>&1
head -n 6 $OUT/prof/*kernel_stats.csv | cut -c1-200
