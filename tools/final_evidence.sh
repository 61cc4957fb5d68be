#!/bin/bash
# The closing evidence of a round on ONE box, everything on the library as it is in the tree: build id, smoke, the whole GPU
# suite, the bench line (default flags and the driver's), the driver's flags under rocprofv3 --kernel-trace --stats, the SQ
# counter passes (separate --pmc runs, no trace domains), the case table with both placement modes, the step with the
# encoding.  usage: [PARTS="tests bench prof cases"] bash tools/final_evidence.sh [out_dir]      (~12 GPU-minutes)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/final}
mkdir -p $OUT
OUT=$(cd $OUT && pwd)          # (absolute: the profiler passes run from /tmp)
cd $R
PARTS=${PARTS:-tests bench prof cases}      # which sections run
python -c "from marlgrid_amd import _native as N; print(N.lib().mg_build_info().decode())" 2>/dev/null | tail -1 > $OUT/build_info.txt
cat $OUT/build_info.txt
if [[ " $PARTS " == *" tests "* ]]; then
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 1500 > $OUT/pytest_gpu.txt 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
fi
if [[ " $PARTS " == *" bench "* ]]; then
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
fi
if [[ " $PARTS " == *" prof "* ]]; then
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-strong --no-pipeline --no-encode-leg --no-pmc --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err )
find $OUT/rocprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -4 $OUT/kernel_stats.csv
bash tools/step_counters.sh $OUT/pmc > /dev/null 2>&1; cp $OUT/pmc/step_counters.txt $OUT/ 2>/dev/null
CASES="tile8 tile5" bash tools/pmc_offpath.sh $OUT/pmc > /dev/null 2>&1; cp $OUT/pmc/pmc_offpath.txt $OUT/ 2>/dev/null
fi
if [[ " $PARTS " == *" cases "* ]]; then
python tools/bench_cases.py --place=default > $OUT/bench_cases_default.jsonl 2>&1
python tools/bench_cases.py --place=thorough > $OUT/bench_cases_thorough.jsonl 2>&1
python tools/ab_encode_fused.py > $OUT/ab_encode_fused_headline.txt 2>&1
fi
rm -rf $OUT/rocprof $OUT/pmc/scA $OUT/pmc/scB $OUT/pmc/pmcA_* $OUT/pmc/pmcB_*
ls -la $OUT
