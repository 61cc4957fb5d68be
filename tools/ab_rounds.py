#!/usr/bin/env python
"""Measurement build: the one-launch step with 1, 2, 3, 4 x as many workgroups as are resident at a time
(MG_RENDER_OVERSUB: the later workgroups start as the first ones exit — their store-free heads run under the others'
stores), interleaved in one process, all into the same observation buffer.  usage: [B=32768] ab_rounds.py [factors ...]"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MARLGRID_HIP_LIB"] = os.path.join(ROOT, "marlgrid_amd", "csrc", "libmarlgrid_hip_ab.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

factors = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
B = int(os.environ.get("B", "32768"))
env = make(os.environ.get("WL", "MarlGrid-3AgentCluttered15x15-v0"), batch_size=B, auto_reset=True, strict=False)
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (B, env.num_agents), generator=g).cuda() for _ in range(16)]
env.reset()
env.step(acts[0])
L = env._lib


def launch(i):
    N.check(L.mg_step_render(C.byref(env._cfg), C.byref(env._state), acts[i % 16].data_ptr(), 8, env.rewards.data_ptr(),
                             C.byref(env._reset_prog), env.obs.data_ptr(), env._stream()))


res = {f: [] for f in factors}
for rep in range(9):
    for f in factors:
        os.environ["MG_RENDER_OVERSUB"] = str(f)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        launch(0)
        a.record()
        for i in range(100):
            launch(i)
        b.record()
        b.synchronize()
        res[f].append(a.elapsed_time(b) / 100)
os.environ.pop("MG_RENDER_OVERSUB")
env.check_errors()
base = statistics.median(res[factors[0]])
for f in factors:
    m = statistics.median(res[f])
    print("workgroups x %d: median %.4f ms (min %.4f max %.4f)  %+.2f%% vs x %d" % (f, m, min(res[f]), max(res[f]), 100 * (m / base - 1), factors[0]))
