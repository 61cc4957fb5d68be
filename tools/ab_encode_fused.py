#!/usr/bin/env python
"""What does MultiGrid.encode of the stepped batch cost when the step's own launch writes it (mg_step_render_encode)
against a second launch behind the step (mg_step_render + mg_encode), and against the step alone?  Interleaved rounds of
100 steps each, the same env, the same observation buffer; identity of the two encodings first.
usage: [TILE=5 | WL=... ] [B=...] ab_encode_fused.py"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B = int(os.environ.get("B", "32768"))
WL = os.environ.get("WL", "MarlGrid-3AgentCluttered15x15-v0")
if os.environ.get("TILE"):
    from marlgrid_amd.agents import GridAgentInterface
    from marlgrid_amd.envs import ClutteredMultiGrid
    env = ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=7, view_tile_size=int(os.environ["TILE"])) for c in ("red", "blue", "purple")],
                             grid_size=15, clutter_density=0.15, batch_size=B, strict=False, auto_reset=True)
else:
    env = make(WL, batch_size=B, auto_reset=True, strict=False)
env.reset()
n = env.num_agents
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (B, n), generator=g).cuda() for _ in range(16)]
L, cfg, st = env._lib, C.byref(env._cfg), C.byref(env._state)
prog = C.byref(env._reset_prog)
enc = torch.empty((B, env.width, env.height, 3), dtype=torch.uint8, device="cuda")
enc2 = torch.empty_like(enc)
stream = env._stream()
print(N.build_info() if hasattr(N, "build_info") else "", env.kernel_name)


def step_plain(i):
    N.check(L.mg_step_render(cfg, st, acts[i % 16].data_ptr(), 8, env.rewards.data_ptr(), prog, env.obs.data_ptr(), stream))


def step_fused(i):
    N.check(L.mg_step_render_encode(cfg, st, acts[i % 16].data_ptr(), 8, env.rewards.data_ptr(), prog, env.obs.data_ptr(), enc.data_ptr(), stream))


def step_two(i):
    N.check(L.mg_step_render(cfg, st, acts[i % 16].data_ptr(), 8, env.rewards.data_ptr(), prog, env.obs.data_ptr(), stream))
    N.check(L.mg_encode(cfg, st, None, enc2.data_ptr(), stream))


for i in range(30):
    step_fused(i)
    N.check(L.mg_encode(cfg, st, None, enc2.data_ptr(), stream))
    assert torch.equal(enc, enc2), i
print("the step's own encoding == mg_encode behind the step over 30 steps (auto-reset on)")
modes = [("step alone", step_plain), ("step + encode in the launch", step_fused), ("step, then mg_encode", step_two)]
times = {m: [] for m, _ in modes}
for rnd in range(9):
    for m, f in modes:
        for i in range(10):
            f(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100):
            f(i)
        e1.record()
        torch.cuda.synchronize()
        times[m].append(e0.elapsed_time(e1) / 100)
base = statistics.median(times["step alone"])
obs_bytes = env.obs.numel()
for m, _ in modes:
    t = statistics.median(times[m])
    print("%-30s median %.4f ms (min %.4f max %.4f)  %+.2f %% vs the step alone" % (m, t, min(times[m]), max(times[m]), 100 * (t / base - 1)))
print("bytes: observations %d + encoding %d (%+.2f %%)" % (obs_bytes, enc.numel(), 100.0 * enc.numel() / obs_bytes))
env.check_errors()
