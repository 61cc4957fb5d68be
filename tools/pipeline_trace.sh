#!/bin/bash
# rocprofv3 kernel trace of tools/pipeline_trace.py -> $OUT/pipeline_trace.txt: duration of a launch and the part of it
# that ran while the previous launch was still running, for the two-stream pipeline and for the one env
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/trace}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/kt -o t --output-format csv -- python $R/tools/pipeline_trace.py > $OUT/kt.log 2>&1
python3 - $OUT <<'PY' > $OUT/pipeline_trace.txt
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 300 launches are the one env; the 600 before them the pipeline (skip each phase's first 50)
def stats(rs, name):
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
    ov, gap = [], []
    for a, b in zip(rs[:-1], rs[1:]):
        o = int(a["End_Timestamp"]) - int(b["Start_Timestamp"])
        ov.append(max(0, o)); gap.append(max(0, -o))
    span = int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])
    q = sorted({r.get("Queue_Id", "?") for r in rs})
    print("%-34s %4d launches: duration mean %.1f us; overlap with the launch before it mean %.1f us (%.0f %% of a launch); idle gap mean %.2f us; one launch per %.1f us; queues %s" %
          (name, len(rs), sum(dur) / len(dur) / 1e3, sum(ov) / len(ov) / 1e3, 100.0 * sum(ov) / sum(dur[1:]), sum(gap) / len(gap) / 1e3, span / (len(rs) - 1) / 1e3, q))
one = rows[-300:][50:]
pipe = rows[-900:-300][100:]
stats(pipe, "two envs of 16 384, two streams")
stats(one, "one env of 32 768")
PY
cat $OUT/pipeline_trace.txt
