#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/placement_search
mkdir -p $OUT
cd $R
for k in 1 2 3; do
  S=$(date +%s.%N); timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-strong --no-cpu-baseline > $OUT/b$k.log 2> $OUT/b$k.err
  python - $OUT/b$k.log <<'PY'
import json,sys
b=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
p=b["obs_placement"]
print("%.1f M  %.4f ms  kept %s  candidates %s  %.2f s  stopped: %s" % (b["value"]/1e6, b["ms_per_step"], ["%.4f"%x for x in p["kept"]], p.get("candidates"), p.get("seconds", -1), p.get("stopped")))
PY
  echo "wall $(echo "$(date +%s.%N) - $S" | bc) s"; tail -n 1 $OUT/b$k.err
done
