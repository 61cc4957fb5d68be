#!/usr/bin/env python
"""overlap launches.  The per-GPU batch as TWO envs of B / 2 on two streams, stepped
alternately without a synchronisation in between — half B's head (staging, step, first views) can run while half A's
waves are still storing, and the other way round across steps — against ONE env of B on one stream; interleaved
rounds in one process, placed buffers.  usage: [B=32768] [PARTS=2] two_halves.py"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B = int(os.environ.get("B", "32768"))
PARTS = int(os.environ.get("PARTS", "2"))
WL = "MarlGrid-3AgentCluttered15x15-v0"
one = make(WL, batch_size=B, auto_reset=True, strict=False)
parts = [make(WL, batch_size=B // PARTS, auto_reset=True, strict=False, seed=1337 + k * (B // PARTS)) for k in range(PARTS)]
streams = [torch.cuda.Stream() for _ in range(PARTS)]
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (B, 3), generator=g).cuda() for _ in range(16)]
acts_p = [[a[k * (B // PARTS):(k + 1) * (B // PARTS)].contiguous() for a in acts] for k in range(PARTS)]
one.reset()
for p in parts:
    p.reset()
torch.cuda.synchronize()
# the parts hold the same global env ids as the one env: same trajectories
for i in range(3):
    o, r, d, _ = one.step(acts[i])
    for k, p in enumerate(parts):
        o2, r2, d2, _ = p.step(acts_p[k][i])
        lo = k * (B // PARTS)
        assert torch.equal(o[lo:lo + B // PARTS], o2) and torch.equal(r[lo:lo + B // PARTS], r2)
print("parts identical to the one env over 3 steps", flush=True)


def run_one(n):
    for i in range(n):
        one.step(acts[i % 16])


def run_parts(n):
    for i in range(n):
        for k, p in enumerate(parts):
            with torch.cuda.stream(streams[k]):
                p.step(acts_p[k][i % 16])


def run_parts_one_stream(n):      # the control: the same parts, the same buffers, no overlap
    for i in range(n):
        for k, p in enumerate(parts):
            p.step(acts_p[k][i % 16])


def run_parts_joined(n):          # a caller that needs ALL observations of step i before it can issue step i + 1:
    main = torch.cuda.current_stream()        # the parts overlap inside a step only
    for i in range(n):
        ev0 = torch.cuda.Event()
        ev0.record(main)
        for k, p in enumerate(parts):
            streams[k].wait_event(ev0)
            with torch.cuda.stream(streams[k]):
                p.step(acts_p[k][i % 16])
            ev = torch.cuda.Event()
            ev.record(streams[k])
            main.wait_event(ev)


res = {"one": [], "parts": [], "parts, one stream": [], "parts, joined": []}
for rep in range(7):
    for name, fn in (("one", run_one), ("parts", run_parts), ("parts, one stream", run_parts_one_stream), ("parts, joined", run_parts_joined)):
        fn(20)
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn(400)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t) / 400 * 1e3)
base = statistics.median(res["one"])
for name in ("one", "parts", "parts, one stream", "parts, joined"):
    m = statistics.median(res[name])
    print("%-18s %d x %6d envs: median %.4f ms per step of all %d envs (min %.4f max %.4f)  %+.2f%%  -> %.1f M agent-steps/s" %
          (name, 1 if name == "one" else PARTS, B if name == "one" else B // PARTS, m, B, min(res[name]), max(res[name]), 100 * (m / base - 1), B * 3 / m / 1e3))
