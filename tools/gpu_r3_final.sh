#!/bin/bash
# Round 3 evidence visit on the final kernels: the bench line (defaults: PMC traffic, strong point, CPU baseline),
# the driver's flags, the same under rocprofv3 --kernel-trace --stats, the off-path cases (placed buffers), phase
# stamps, the other BASELINE configs.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-final}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err)
python3 -c "
import json
d=json.load(open('$OUT/bench_n1.json'))
print('bench: value %.1f M (median block %.1f M) ms_per_step %.4f kernel_ms %.4f frac %.3f closure %.4f traffic %s placement %s strong %.1f M cpu %.2f M' % (d['value']/1e6, d['value_median_block']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['closure']['vs_ms_per_step'], d['roofline']['traffic'], {k:v for k,v in (d.get('obs_placement') or {}).items() if k!='all'}, d['extra']['strong_n1']['value']/1e6, d['cpu_baseline']['value']/1e6))
for p in d['extra'].get('pipelined_shards', {}).get('points', []): print('pipelined shards: %d envs as 2 x %d: %.1f M (%.4f ms) vs one env %.1f M (%.4f ms): x %.3f' % (p['envs'], p['envs_per_part'], p['value']/1e6, p['ms_per_step'], p['one_env_value']/1e6, p['one_env_ms_per_step'], p['vs_one_env']))"
tail -n 2 $OUT/bench.err
(cd $R && timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-pipeline --no-cpu-baseline > $OUT/bench_driver_flags.json 2>> $OUT/bench.err)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o kt --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-pipeline --no-cpu-baseline > $OUT/prof_bench.log 2>&1
head -n 3 $OUT/prof_bench/*kernel_stats.csv | cut -c1-200
(cd $R && timeout 400 python tools/ab_offpath.py 2>&1 | grep -v amdgpu.ids > $OUT/ab_offpath.jsonl); cut -c1-260 $OUT/ab_offpath.jsonl
(cd $R && OPTS="0 0" timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps.txt); head -n 26 $OUT/phase_stamps.txt
(cd $R && timeout 400 python tools/profile_misc.py "configs" "tile 16" "view 11" "tile 32" "mg_encode" > $OUT/misc.jsonl 2> $OUT/misc.err); cut -c1-300 $OUT/misc.jsonl
