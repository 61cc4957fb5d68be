#!/usr/bin/env python
"""Clock / power trace of the obs raster vs a dense fill under sustained load.

Phases of `--seconds` each, back to back in ONE process: torch fill_ of an obs-sized tensor, the
production raster, the raster's stores-only / raster-only measurement variants (from the A/B build),
then fill and the raster again.  Per phase: the average launch time of every 100-launch chunk (HIP
events on the launch stream) and rocm-smi samples (sclk / mclk / fclk / power / temperature) taken by a
background thread.  Writes JSON to --out and prints a per-phase summary."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=5.0)
ap.add_argument("--batch", type=int, default=32768)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "power_trace.json"))
args = ap.parse_args()

import torch  # noqa: E402
import smi  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B = args.batch
env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, auto_reset=True, strict=False)
env.reset()
g = torch.Generator().manual_seed(0)
for i in range(30):
    env.step(torch.randint(0, 7, (B, env.num_agents), generator=g).cuda())
torch.cuda.synchronize()
nbytes = env.obs.numel()
fill_t = torch.empty(nbytes, dtype=torch.uint8, device=env.device)
ab = None
ab_path = os.path.join(ROOT, "marlgrid_amd", "csrc", "libmarlgrid_hip_ab.so")
if os.path.exists(ab_path):
    ab = C.CDLL(ab_path)
    ab.mg_render_obs.argtypes = env._lib.mg_render_obs.argtypes
    ab.mg_render_obs.restype = C.c_int32


def launch_fill():
    fill_t.fill_(7)


def launch_render(lib=None):
    L = lib or env._lib
    N.check(L.mg_render_obs(C.byref(env._cfg), C.byref(env._state), env.obs.data_ptr(), None, None, None,
                            env._stream()))


def phase(name, fn, seconds, sampler, rows):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_end = time.perf_counter() + seconds
    chunk = []
    while time.perf_counter() < t_end:
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        chunk.append(ms)
        rows.append({"phase": name, "t": round(sampler.now(), 3), "ms": ms, "GBps": nbytes / ms / 1e6})
    return chunk


rows = []
phases = [("fill", launch_fill), ("render", launch_render)]
if ab is not None:
    for v, label in (("4", "raster stores only (A/B 4)"), ("3", "raster without phases 2-5 (A/B 3)"),
                     ("0", "render (A/B build, variant 0)")):
        def f(v=v):
            os.environ["MG_RENDER_VARIANT"] = v
            launch_render(ab)
        phases.append((label, f))
phases += [("fill again", launch_fill), ("render again", launch_render)]
summary = []
with smi.Sampler(0.2, env.device.index or 0) as sampler:
    time.sleep(0.5)
    for name, fn in phases:
        t0 = sampler.now()
        ch = phase(name, fn, args.seconds, sampler, rows)
        t1 = sampler.now()
        ss = [s for (t, s) in sampler.rows if t0 <= t <= t1 and s]
        avg = lambda k: (sum(s[k] for s in ss if k in s) / max(1, sum(1 for s in ss if k in s))) if ss else None   # noqa: E731
        summary.append({"phase": name, "first_ms": ch[0], "last_ms": ch[-1], "min_ms": min(ch), "max_ms": max(ch),
                        "GBps_first": nbytes / ch[0] / 1e6, "GBps_last": nbytes / ch[-1] / 1e6,
                        "sclk_mhz": avg("sclk_mhz"), "mclk_mhz": avg("mclk_mhz"), "fclk_mhz": avg("fclk_mhz"),
                        "power_w": avg("power_w"), "temp_junction_c": avg("temp_junction_c"),
                        "temp_hbm_c": avg("temp_hbm_c")})
        print(json.dumps(summary[-1]), flush=True)
    smi_rows = list(sampler.rows)
os.environ.pop("MG_RENDER_VARIANT", None)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump({"bytes_per_launch": nbytes, "summary": summary, "launch_chunks": rows,
           "smi": [{"t": t, **s} for t, s in smi_rows]}, open(args.out, "w"))
print("wrote", args.out)
