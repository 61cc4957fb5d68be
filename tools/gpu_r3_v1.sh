#!/bin/bash
# Round 3, visit 1: the -m gpu suite on ABI 3, the obs-buffer construction probe, the bench line in its new form,
# the self-launched 2-rank run, the cost of the error contract, phase stamps of the unchanged kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 5 $OUT/pytest.log
(cd $R && timeout 300 python tools/placement_vmm.py 2>&1 | grep -v amdgpu.ids > $OUT/placement_vmm.txt); cut -c1-200 $OUT/placement_vmm.txt
(cd $R && timeout 600 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err)
cut -c1-600 $OUT/bench.log; tail -n 3 $OUT/bench.err
(cd $R && timeout 300 python bench.py --gpus 2 --oversubscribe --batch-per-gpu 16384 --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/bench_gpus2_selflaunch.log 2> $OUT/bench_gpus2_selflaunch.err; echo "rc=$?" >> $OUT/bench_gpus2_selflaunch.err)
cut -c1-400 $OUT/bench_gpus2_selflaunch.log; tail -n 2 $OUT/bench_gpus2_selflaunch.err
(cd $R && timeout 120 python bench.py --gpus 2 --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/bench_gpus2_refused.log 2>&1; echo "rc=$?" >> $OUT/bench_gpus2_refused.log); tail -n 3 $OUT/bench_gpus2_refused.log
(cd $R && timeout 300 python tools/strict_overhead.py 2>&1 | grep -v amdgpu.ids > $OUT/strict_overhead.json); cat $OUT/strict_overhead.json
(cd $R && timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps_baseline.txt); head -n 14 $OUT/phase_stamps_baseline.txt
