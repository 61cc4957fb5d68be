#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c32
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"; do
  tag=$(echo $P | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $P -d $OUT/$tag -o pmc --output-format csv -- python $R/tools/time_kernels.py > $OUT/$tag.log 2>&1
  python - $OUT/$tag <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
if not f: print("no output"); sys.exit()
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    acc[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in acc.items():
    if "mg::" in k: print(k, {c:(sum(x)/len(x), len(x)) for c,x in v.items()})
PY
done
