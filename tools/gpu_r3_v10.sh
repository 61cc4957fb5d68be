#!/bin/bash
# Round 3, visit 10: view groups that ramp (1, 2, 4 envs) — suite, A/B on PLACED buffers against the build before
# (ref2) and round 2's kernel (ref), phase stamps, bench at the driver's flags.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v10}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 4 $OUT/pytest.log
(cd $R && timeout 300 python tools/ab_fused.py marlgrid_amd/csrc/libmarlgrid_hip_ref.so marlgrid_amd/csrc/libmarlgrid_hip_ref2.so marlgrid_amd/csrc/libmarlgrid_hip.so 2>&1 | grep -v amdgpu.ids > $OUT/ab_fused.txt); cat $OUT/ab_fused.txt
(cd $R && timeout 300 python tools/ab_two_libs.py marlgrid_amd/csrc/libmarlgrid_hip_ref2.so 9 2>&1 | grep -v amdgpu.ids > $OUT/ab_raster_vs_ref2.txt); cat $OUT/ab_raster_vs_ref2.txt
(cd $R && timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps.txt); head -n 13 $OUT/phase_stamps.txt
(cd $R && timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/bench_driver_flags.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err)
python3 -c "
import json
d=json.load(open('$OUT/bench_driver_flags.json'))
print('bench (driver flags): value %.1f M (median block %.1f M) ms_per_step %.4f kernel_ms %.4f frac %.3f closure %.4f placement %s' % (d['value']/1e6, d['value_median_block']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['closure']['vs_ms_per_step'], {k:v for k,v in (d.get('obs_placement') or {}).items() if k!='all'}))"
