#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v31}
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "placement_search or shard_pipeline" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -n 3 $OUT/pytest.log
for i in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-pipeline --no-cpu-baseline > $OUT/bench_driver_flags_$i.json 2>> $OUT/bench.err
python3 -c "
import json
d=json.load(open('$OUT/bench_driver_flags_$i.json'))
print('driver flags $i: value %.1f M median %.1f M ms %.4f kernel_ms %.4f frac %.3f closure %.4f outliers %s placement %s' % (d['value']/1e6, d['value_median_block']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['closure']['vs_ms_per_step'], d['timing']['blocks']['outliers'], {k:v for k,v in (d.get('obs_placement') or {}).items() if k!='all'}))"
done
