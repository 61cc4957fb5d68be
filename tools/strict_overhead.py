#!/usr/bin/env python
"""What the error contract costs per step: env.step() at the bench workload with strict=True (default: the
kernels raise a host-mapped flag that step() polls — no synchronize), strict=False and strict='sync' (one host
synchronize per step, the round-2 behaviour of strict=True).  Interleaved rounds, host wall time per step over
K steps bracketed by synchronize."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B, K = int(os.environ.get("BATCH", "32768")), 400
# ONE env (one set of observation buffers: their placement moves a launch by more than what is measured here),
# its `strict` attribute switched between rounds
_env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, auto_reset=True)
envs = {m: _env for m in (True, False, "sync")}
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (B, 3), generator=g).cuda() for _ in range(16)]
res = {repr(m): [] for m in envs}
for e in envs.values():
    e.reset()
    for i in range(20):
        e.step(acts[i % 16])
for rnd in range(5):
    for m, e in envs.items():
        e.strict = m
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            e.step(acts[i % 16])
        torch.cuda.synchronize()
        res[repr(m)].append((time.perf_counter() - t0) / K * 1e3)
out = {m: {"ms_per_step_rounds": v, "median": sorted(v)[len(v) // 2]} for m, v in res.items()}
out["default_vs_strict_false"] = out["True"]["median"] / out["False"]["median"]
out["sync_vs_strict_false"] = out["'sync'"]["median"] / out["False"]["median"]
out["batch"] = B
print(json.dumps(out))
