#!/usr/bin/env python
"""gpurun_out/prof_wr + prof_rd (tools/gpu_round.sh / prof_only.sh: separate rocprofv3 --pmc WRITE_SIZE and
--pmc FETCH_SIZE passes over tools/profile_render.py) -> profiles/r01/pmc_hbm_bytes.json"""
import collections
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for name, counter in (("wr", "WRITE_SIZE"), ("rd", "FETCH_SIZE")):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(os.path.join(ROOT, "gpurun_out", "prof_%s" % name, "%s_counter_collection.csv" % name))):
        if r["Counter_Name"] == counter:
            acc.setdefault(r["Kernel_Name"].split("(")[0][:80], []).append(round(float(r["Counter_Value"]), 1))
    out["%s_KiB_per_dispatch" % counter] = acc
render = [k for k in out["WRITE_SIZE_KiB_per_dispatch"] if "render_kernel" in k][0]
w = out["WRITE_SIZE_KiB_per_dispatch"][render]
f = out["FETCH_SIZE_KiB_per_dispatch"][render]
wb = sum(w) / len(w) * 1024
fb = 2 * sum(f) / len(f) * 1024
out["render_kernel_write_bytes"] = wb
out["render_kernel_fetch_bytes_corrected_x2"] = fb
out["render_kernel_hbm_bytes_per_launch"] = wb + fb
out["algorithmic_bytes_per_launch"] = 32768 * 3 * 9481
out["notes"] = ("rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE collected in separate passes over tools/profile_render.py "
                "at the bench workload (32768 envs of MarlGrid-3AgentCluttered15x15-v0). Calibration in the same runs: "
                "1 GiB torch fill -> WRITE_SIZE 1048576 KiB (x1.0); 1 GiB copyBuffer -> FETCH_SIZE ~524300 KiB, i.e. the "
                "gfx950 half-count the MI355X guide documents (x2 correction applied to fetches).")
json.dump(out, open(os.path.join(ROOT, "profiles", "r01", "pmc_hbm_bytes.json"), "w"), indent=1)
print("render kernel: %.1f MB written + %.1f MB fetched = %.1f MB vs %.1f MB algorithmic (x%.3f)"
      % (wb / 1e6, fb / 1e6, (wb + fb) / 1e6, out["algorithmic_bytes_per_launch"] / 1e6, (wb + fb) / out["algorithmic_bytes_per_launch"]))
