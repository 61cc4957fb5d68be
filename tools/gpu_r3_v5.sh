#!/bin/bash
# Round 3, visit 5: the -m gpu suite (placement search by default, three-phase raster loop, 'prestige' colours from the
# stepping lane), the off-path instantiations under their workgroup shapes, the bench line + rocprofv3 kernel trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v5}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 6 $OUT/pytest.log
(cd $R && timeout 400 python tools/ab_offpath.py 2>&1 | grep -v amdgpu.ids > $OUT/ab_offpath.jsonl); cut -c1-330 $OUT/ab_offpath.jsonl
(cd $R && timeout 600 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err)
python3 -c "
import json
d=json.load(open('$OUT/bench.log'))
print('bench: value %.1f M (median block %.1f M) ms_per_step %.4f kernel_ms %.4f frac %.3f closure %.4f traffic %s placement %s strong %.1f M cpu %s' % (d['value']/1e6, d['value_median_block']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['closure']['vs_ms_per_step'], d['roofline']['traffic'], {k:v for k,v in (d.get('obs_placement') or {}).items() if k!='all'}, d['extra']['strong_n1']['value']/1e6, d['cpu_baseline']['value']))"
tail -n 2 $OUT/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o kt --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/prof_bench.log 2>&1
head -n 4 $OUT/prof_bench/*kernel_stats.csv | cut -c1-200; cut -c1-200 $OUT/prof_bench.log | tail -n 2
