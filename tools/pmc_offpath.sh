#!/bin/bash
# SQ counter passes (two per case: they do not fit one) over the obs kernel of the off-path cases -> $OUT/pmc_offpath.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in ${CASES:-tile5 tile11 human tile8}; do
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -d $OUT/pmcA_$c -o a --output-format csv -- python $R/tools/profile_offpath.py $c > $OUT/pmcA_$c.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR -d $OUT/pmcB_$c -o b --output-format csv -- python $R/tools/profile_offpath.py $c > $OUT/pmcB_$c.log 2>&1
done
python3 - $OUT <<'PY' > $OUT/pmc_offpath.txt
import csv, glob, collections, os, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc[AB]_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    kern = None
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            kern = r["Kernel_Name"].split("(")[0]
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    print([p for p in f.split("/") if p.startswith("pmc")][0], kern)
    for k, (n, v) in sorted(acc.items()):
        print("   %-28s %14.0f per launch (%d launches)" % (k, v / n, n))
PY
cat $OUT/pmc_offpath.txt
