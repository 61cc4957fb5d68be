#!/bin/bash
# Round-2 GPU visit 1: parity of the new step/reset path, the new bench line, N>1 plumbing on one GPU,
# hot-state store patterns, clock/power trace, rocprofv3 kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocm-smi --showclocks --showpower --showtemp --json > $OUT/smi_idle.json 2>&1
(cd $R && timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 4 $OUT/pytest.log
(cd $R && timeout 600 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err)
cut -c1-1500 $OUT/bench.log; tail -n 3 $OUT/bench.err
(cd $R && timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --oversubscribe --batch-per-gpu 16384 > $OUT/bench_2rank_oversub.log 2> $OUT/bench_2rank_oversub.err; echo "rc=$?" >> $OUT/bench_2rank_oversub.err)
cut -c1-600 $OUT/bench_2rank_oversub.log; tail -n 2 $OUT/bench_2rank_oversub.err
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o kt --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/prof_bench.log 2>&1
head -n 6 $OUT/prof_bench/*kernel_stats.csv | cut -c1-160
(cd $R && timeout 200 tools/microbench/store_patterns2 15 0.4 5 > $OUT/store_patterns2.log 2>&1)
grep "^hot4\|^cold" $OUT/store_patterns2.log | cut -c1-110
(cd $R && timeout 300 python tools/power_trace.py --seconds 5 --out $OUT/power_trace.json > $OUT/power_trace.log 2>&1)
cut -c1-400 $OUT/power_trace.log
(cd $R && timeout 120 python tools/time_kernels.py > $OUT/time_kernels.log 2>&1); cat $OUT/time_kernels.log
