#!/bin/bash
# Round-2 evidence run: everything profiles/r02 holds comes from one visit of this script on one MI355X box.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-final}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocm-smi --showclocks --showpower --showtemp --json > $OUT/smi_idle.json 2>&1
(cd $R && timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 4 $OUT/pytest.log
(cd $R && timeout 600 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err)
cut -c1-400 $OUT/bench.log; tail -n 2 $OUT/bench.err
(cd $R && timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/bench_driver_flags.log 2>> $OUT/bench.err)
(cd $R && timeout 300 python bench.py --unfused --no-pmc --no-strong --no-cpu-baseline > $OUT/bench_unfused.log 2>> $OUT/bench.err)
(cd $R && timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --oversubscribe --batch-per-gpu 16384 > $OUT/bench_2rank_oversub.log 2> $OUT/bench_2rank_oversub.err; echo "rc=$?" >> $OUT/bench_2rank_oversub.err)
cut -c1-300 $OUT/bench_2rank_oversub.log
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o kt --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/prof_bench.log 2>&1
head -n 5 $OUT/prof_bench/*kernel_stats.csv | cut -c1-170
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_unfused -o kt --output-format csv -- python $R/bench.py --unfused --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/prof_unfused.log 2>&1
(cd $R && timeout 400 python tools/profile_misc.py > $OUT/misc.jsonl 2> $OUT/misc.err); cut -c1-260 $OUT/misc.jsonl
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_misc -o kt --output-format csv -- python $R/tools/profile_misc.py > $OUT/prof_misc.log 2>&1
(cd $R && timeout 300 python tools/power_trace.py --seconds 4 --out $OUT/power_trace.json > $OUT/power_trace.log 2>&1); cut -c1-330 $OUT/power_trace.log
(cd $R && timeout 200 python tools/ab_render.py 0 4 3 F R > $OUT/ab_render.log 2>&1); tail -n 6 $OUT/ab_render.log
(cd $R && timeout 100 tools/microbench/store_patterns5 0.25 2 > $OUT/store_patterns5.log 2>&1); grep "^r1" $OUT/store_patterns5.log | cut -c1-110
(cd $R && timeout 120 python tools/time_kernels.py > $OUT/time_kernels.log 2>&1); cat $OUT/time_kernels.log | grep -v amdgpu
(cd $R && for k in 1 2; do timeout 100 tools/microbench/store_patterns6 6; echo "--- new process"; done > $OUT/store_patterns6.log 2>&1); grep -c "render pattern" $OUT/store_patterns6.log
(cd $R && timeout 200 python tools/placement_probe.py 2>&1 | grep -v amdgpu > $OUT/placement_probe.log); tail -n 3 $OUT/placement_probe.log
