#!/usr/bin/env python
"""Dense-front raster vs the alignment of the obs buffer (measurement build): times mg_render_obs into torch's own
obs tensor and into views of a scratch buffer at chosen offsets from a 2 MiB boundary."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MARLGRID_HIP_LIB", os.path.join(ROOT, "marlgrid_amd", "csrc", "libmarlgrid_hip_ab.so"))
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B = 32768
env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, auto_reset=True, strict=False)
env.reset()
if os.environ.get("MARLGRID_HIP_LIB", "").endswith("_ab.so"):      # two-kernel experiment: its view scratch
    _vs = torch.zeros((B, env.num_agents * env.view_size ** 2), dtype=torch.int16, device=env.device)
    env._lib.mg_ab_view_scratch.restype = None
    env._lib.mg_ab_view_scratch(C.c_void_p(_vs.data_ptr()))
for i in range(10):
    env.step(torch.randint(0, 7, (B, 3)).cuda())
torch.cuda.synchronize()
nbytes = env.obs.numel()
print("obs.data_ptr %% 2MiB = %d, %% 4096 = %d; ring: %s" % (env.obs.data_ptr() % (1 << 21), env.obs.data_ptr() % 4096,
      [r["obs"].data_ptr() % 4096 for r in env._ring]))
buf = torch.empty(nbytes + (4 << 20), dtype=torch.uint8, device=env.device)
base = (buf.data_ptr() + (1 << 21) - 1) // (1 << 21) * (1 << 21)
ms = C.c_float(0)
for mode in sys.argv[1:] or ["0", "6"]:
    os.environ["MG_FRONT_MODE"] = mode
    for off in (None, 0, 512, 1024, 2048, 4096):
        ptr = env.obs.data_ptr() if off is None else base + off
        N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), C.c_void_p(ptr), 30, C.byref(ms),
                                            env._stream()))
        print("mode %s  %-22s views+raster %.4f ms" % (mode, "torch obs tensor" if off is None else "2MiB + %d" % off, ms.value))
