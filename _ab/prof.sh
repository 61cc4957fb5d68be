mkdir -p gpurun_out/v24
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-strong --no-pipeline > $R/gpurun_out/v24/bench_under_rocprof.json 2> $R/gpurun_out/v24/rocprof.err
find /tmp/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/v24/kernel_stats.csv
head -3 $R/gpurun_out/v24/kernel_stats.csv | cut -c1-300
python - <<PY
import json
d=json.loads(open("$R/gpurun_out/v24/bench_under_rocprof.json").read().strip().splitlines()[-1])
print(round(d["value"]/1e6,1), d["roofline"]["kernel_ms"])
PY
