#!/bin/bash
# Round 3, visit 11: the env step per view group (not the whole batch up front), place_obj(reject_fn=) as tables —
# suite, A/B on placed buffers against the build before (ref2: same ABI, whole-batch step), phase stamps, bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v11}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 4 $OUT/pytest.log
(cd $R && timeout 300 python tools/ab_fused.py marlgrid_amd/csrc/libmarlgrid_hip_ref2.so marlgrid_amd/csrc/libmarlgrid_hip.so 2>&1 | grep -v amdgpu.ids > $OUT/ab_fused.txt); cat $OUT/ab_fused.txt
(cd $R && timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps.txt); head -n 13 $OUT/phase_stamps.txt
(cd $R && timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/bench_driver_flags.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err)
python3 -c "
import json
d=json.load(open('$OUT/bench_driver_flags.json'))
print('bench (driver flags): value %.1f M (median block %.1f M) ms_per_step %.4f kernel_ms %.4f frac %.3f closure %.4f placement %s' % (d['value']/1e6, d['value_median_block']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['closure']['vs_ms_per_step'], {k:v for k,v in (d.get('obs_placement') or {}).items() if k!='all'}))"
(cd $R && B=262144 timeout 300 python tools/ab_fused.py marlgrid_amd/csrc/libmarlgrid_hip_ref2.so marlgrid_amd/csrc/libmarlgrid_hip.so 2>&1 | grep -v amdgpu.ids > $OUT/ab_fused_262144.txt); tail -n 2 $OUT/ab_fused_262144.txt
