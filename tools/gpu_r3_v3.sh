#!/bin/bash
# Round 3, visit 3: observation buffers — one set of physical handles behind many virtual ranges; the env's own
# range search; the bench line from three fresh processes; the -m gpu suite.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 python tools/placement_va.py 2>&1 | grep -v amdgpu.ids > $OUT/placement_va.txt); cat $OUT/placement_va.txt
for k in 1 2 3; do
  (cd $R && timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/bench_proc$k.json 2>> $OUT/bench_proc.err)
  python3 -c "
import json,sys
d=json.load(open('$OUT/bench_proc$k.json'))
print('process $k: value %.1f M (median block %.1f M) ms_per_step %.4f kernel_ms %.4f closure %.4f placement %s' % (d['value']/1e6, d['value_median_block']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['closure']['vs_ms_per_step'], {k:v for k,v in (d.get('obs_placement') or {}).items() if k!='all'}))"
done
(cd $R && timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 5 $OUT/pytest.log
