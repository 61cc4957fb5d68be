#!/bin/bash
# Round 3, visit 9: views of up to four envs at a time under the chunk raster — the whole -m gpu suite, interleaved A/B against the earlier
# build (raster alone and the whole step), the bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v9}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 6 $OUT/pytest.log
(cd $R && timeout 300 python tools/ab_two_libs.py marlgrid_amd/csrc/libmarlgrid_hip_ref.so 9 2>&1 | grep -v amdgpu.ids > $OUT/ab_raster_vs_ref.txt); cat $OUT/ab_raster_vs_ref.txt
(cd $R && timeout 300 python tools/ab_fused.py marlgrid_amd/csrc/libmarlgrid_hip_ref.so marlgrid_amd/csrc/libmarlgrid_hip.so 2>&1 | grep -v amdgpu.ids > $OUT/ab_fused_vs_ref.txt); cat $OUT/ab_fused_vs_ref.txt
(cd $R && timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/bench_driver_flags.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err)
python3 -c "
import json
d=json.load(open('$OUT/bench_driver_flags.json'))
print('bench (driver flags): value %.1f M (median block %.1f M) ms_per_step %.4f kernel_ms %.4f frac %.3f closure %.4f placement %s' % (d['value']/1e6, d['value_median_block']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['closure']['vs_ms_per_step'], {k:v for k,v in (d.get('obs_placement') or {}).items() if k!='all'}))"
tail -n 2 $OUT/bench.err
CASES="tile8" bash $R/tools/pmc_offpath.sh $OUT > /dev/null 2>&1; cat $OUT/pmc_offpath.txt
