#!/bin/bash
# Round-2 GPU visit 2: parity of the assemble-and-stream raster (all tile sizes off the chunk path), its
# rate at tile 5 / 6 / 11 and vs the chunk raster at tile 8, and store-pattern microbench 3.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 15 $OUT/pytest.log
(cd $R && timeout 200 tools/microbench/store_patterns3 0.3 3 > $OUT/store_patterns3.log 2>&1; echo "rc=$?" >> $OUT/store_patterns3.log)
grep "^r2\|rc=" $OUT/store_patterns3.log | cut -c1-230
(cd $R && timeout 300 python tools/bench_generic.py > $OUT/bench_generic.log 2>&1); cat $OUT/bench_generic.log | tail -n 8
(cd $R && timeout 300 python tools/ab_render.py 0 R 4 3 > $OUT/ab_render.log 2>&1); tail -n 6 $OUT/ab_render.log
