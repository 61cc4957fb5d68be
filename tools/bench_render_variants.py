#!/usr/bin/env python
"""Time one obs-render variant (MG_RENDER_VARIANT) / step block size (MG_STEP_BLOCK) and print a
checksum so that variants that must be output-identical can be compared across processes."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B = int(os.environ.get("B", "32768"))
wl = os.environ.get("WL", "MarlGrid-3AgentCluttered15x15-v0")
env = make(wl, batch_size=B, auto_reset=True, strict=False)
env.reset()
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (B, env.num_agents), generator=g).cuda() for _ in range(16)]
for i in range(30):
    env.step(acts[i % 16])
torch.cuda.synchronize()
ms = C.c_float(0)
best = 1e9
for rep in range(3):
    N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), env.obs.data_ptr(), 100, C.byref(ms),
                                        env._stream()))
    best = min(best, ms.value)
chk = int(env.obs.to(torch.int64).sum().item())
vs, ts, n = env.view_size, env.tile_size, env.num_agents
alg = B * n * (vs * ts * vs * ts * 3 + vs * vs + 8 * n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    env.step(acts[i % 16])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
print("variant=%s stepblock=%s  render %.4f ms  %.1f GB/s (%.1f%% of 8 TB/s)  full step %.4f ms  %.1f M agent-steps/s  checksum %d"
      % (os.environ.get("MG_RENDER_VARIANT", "0"), os.environ.get("MG_STEP_BLOCK", "auto"), best, alg / best / 1e6,
         alg / best / 1e6 / 80.0, dt * 1e3, B * n / dt / 1e6, chk))
