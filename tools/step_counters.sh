#!/bin/bash
# SQ instruction counters of the bench workload's kernel, raster only against step + raster (the last 6 dispatches
# of tools/profile_step_counters.py are fused, the 6 before them raster only) -> $OUT/step_counters.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -d $OUT/scA -o a --output-format csv -- python $R/tools/profile_step_counters.py > $OUT/scA.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/scB -o b --output-format csv -- python $R/tools/profile_step_counters.py > $OUT/scB.log 2>&1
python3 - $OUT <<'PY' > $OUT/step_counters.txt
import csv, glob, collections, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/sc[AB]/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    fused, raster = set(ids[-6:]), set(ids[-12:-6])
    acc = {"fused": collections.defaultdict(float), "raster": collections.defaultdict(float)}
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d in fused: acc["fused"][r["Counter_Name"]] += float(r["Counter_Value"]) / 6
        elif d in raster: acc["raster"][r["Counter_Name"]] += float(r["Counter_Value"]) / 6
    print(f.split("/")[-3] if "/sc" in f else f)
    for k in sorted(acc["fused"]):
        a, b = acc["fused"][k], acc["raster"].get(k, 0.0)
        print("   %-22s fused %13.0f  raster %13.0f  step = %12.0f per launch = %9.1f per wave (4 096 waves, one batch of 8 envs each)" % (k, a, b, a - b, (a - b) / 4096))
PY
cat $OUT/step_counters.txt
