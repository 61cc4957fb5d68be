#!/bin/bash
# The round's last visit: the -m gpu suite, smoke(), the self-launched two-rank bench (plumbing), the bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-last}
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -n 4 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -n 2
timeout 600 python bench.py --gpus 2 --oversubscribe --min-seconds 1 --no-pmc > $OUT/bench_gpus2.json 2> $OUT/bench_gpus2.err; echo "gpus2 rc=$?"
python3 -c "
import json
d=json.load(open('$OUT/bench_gpus2.json'))
print('gpus 2 (oversubscribed): n_gpus %d value %.1f M launched_by %s ranks %s' % (d['n_gpus'], d['value']/1e6, d['timing'].get('launched_by'), [(r['rank'], r['device_index']) for r in d['timing']['per_rank']]))"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"
python3 -c "
import json
d=json.load(open('$OUT/bench_driver.json'))
print('driver command: value %.1f M median %.1f M ms %.4f kernel_ms %.4f frac %.3f closure %.4f outliers %s traffic %s cpu %.2f M pipelined %.1f M placement %s' % (d['value']/1e6, d['value_median_block']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['closure']['vs_ms_per_step'], d['timing']['blocks']['outliers'], d['roofline']['traffic'], d['cpu_baseline']['value']/1e6, d.get('value_pipelined_shards', 0)/1e6, {k:v for k,v in (d.get('obs_placement') or {}).items() if k!='all'}))"
