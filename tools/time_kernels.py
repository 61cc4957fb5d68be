#!/usr/bin/env python
"""Per-kernel wall time of one env.step() at the bench workload (torch events on the launch stream)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B = int(os.environ.get("B", "32768"))
env = make(os.environ.get("WL", "MarlGrid-3AgentCluttered15x15-v0"), batch_size=B, auto_reset=True, strict=False)
env.reset()
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (B, env.num_agents), generator=g).cuda() for _ in range(16)]
for i in range(40):
    env.step(acts[i % 16])
L, cfg, st = env._lib, C.byref(env._cfg), C.byref(env._state)


def timed(fn, iters=200):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(0)
    torch.cuda.synchronize()
    a.record()
    for i in range(iters):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def k_step(i):      # action loop + the fused reset of finished episodes
    N.check(L.mg_step(cfg, st, acts[i % 16].data_ptr(), 8, env.rewards.data_ptr(), C.byref(env._reset_prog),
                      env._stream()))


def k_step_only(i):  # action loop alone (finished envs stay finished: state drifts towards all-done)
    N.check(L.mg_step(cfg, st, acts[i % 16].data_ptr(), 8, env.rewards.data_ptr(), None, env._stream()))


def k_render(i):
    N.check(L.mg_render_obs(cfg, st, env.obs.data_ptr(), None, None, None, env._stream()))


def k_fused(i):     # the same step as ONE launch: action loop + reset + raster
    N.check(L.mg_step_render(cfg, st, acts[i % 16].data_ptr(), 8, env.rewards.data_ptr(), C.byref(env._reset_prog),
                             env.obs.data_ptr(), env._stream()))


def k_all(i):
    env.step(acts[i % 16])


for name, fn in (("mg_step with fused auto-reset", k_step), ("render", k_render),
                 ("mg_step_render (one launch)", k_fused), ("env.step (python + launch)", k_all),
                 ("mg_step without reset program", k_step_only)):
    print("%-34s %.4f ms" % (name, timed(fn)))
env.check_errors()
