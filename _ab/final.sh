set -x
mkdir -p gpurun_out/v24
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/v24/pytest_gpu.txt 2>&1; tail -3 gpurun_out/v24/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/v24/smoke.txt 2>&1; tail -1 gpurun_out/v24/smoke.txt
timeout 300 python bench.py > gpurun_out/v24/bench_default.json 2> gpurun_out/v24/bench_default.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/v24/bench_driver.json 2> gpurun_out/v24/bench_driver.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/v24/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/v24/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/v24/rocprof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/v24/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/v24/kernel_stats.csv
find gpurun_out/v24/prof -type f ! -name "*stats.csv" -delete 2>/dev/null
python - <<'PY'
import json
for f in ("default","driver","under_rocprof"):
    try:
        d=json.loads(open("gpurun_out/v24/bench_%s.json"%f).read().strip().splitlines()[-1])
        r=d["roofline"]; print(f, round(d["value"]/1e6,1), d["ms_per_step"], r["kernel_ms"], round(r["frac"],3), r["raster_only_ms"], d.get("obs_placement",{}).get("kept"), d["cpu_baseline"]["value"])
    except Exception as e: print(f, "ERR", e)
PY
head -4 gpurun_out/v24/kernel_stats.csv
