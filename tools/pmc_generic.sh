#!/bin/bash
# SQ counter passes over the obs kernel at tile 5 and tile 8 -> gpurun_out/pmc_generic_*.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ts in 5 8; do
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -d $OUT/pmcA$ts -o a --output-format csv -- python $R/tools/profile_generic.py $ts > $OUT/pmcA$ts.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL -d $OUT/pmcB$ts -o b --output-format csv -- python $R/tools/profile_generic.py $ts > $OUT/pmcB$ts.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for f in sorted(glob.glob(out + "/pmc[AB][58]/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    print(f.split("/")[-3] if "pmc" in f.split("/")[-3] else f)
    for k, (n, v) in sorted(acc.items()):
        print("   %-28s %14.0f per launch (%d)" % (k, v / n, n))
PY
