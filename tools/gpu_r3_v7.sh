#!/bin/bash
# Round 3, visit 7: index arithmetic in 24-bit multiplies, affine view map, global (not flat) stores in the
# assemble-and-stream raster: the whole -m gpu suite, the off-path cases, SQ counters for tile 5, the bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v7}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 6 $OUT/pytest.log
(cd $R && timeout 400 python tools/ab_offpath.py 2>&1 | grep -v amdgpu.ids > $OUT/ab_offpath.jsonl); cut -c1-330 $OUT/ab_offpath.jsonl
CASES="tile5 human" bash $R/tools/pmc_offpath.sh $OUT > /dev/null 2>&1; cat $OUT/pmc_offpath.txt
(cd $R && timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/bench_driver_flags.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err)
python3 -c "
import json
d=json.load(open('$OUT/bench_driver_flags.json'))
print('bench (driver flags): value %.1f M (median block %.1f M) ms_per_step %.4f kernel_ms %.4f frac %.3f closure %s placement %s' % (d['value']/1e6, d['value_median_block']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['closure'], {k:v for k,v in (d.get('obs_placement') or {}).items() if k!='all'}))"
tail -n 2 $OUT/bench.err
