#!/usr/bin/env python
"""Does the obs raster's HBM write throughput depend on the alignment of the output stream?
Times the tile-8 (16-byte-chunk) kernel into buffers offset by 0..112 bytes, then the size-generic
kernels."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.agents import GridAgentInterface  # noqa: E402
from marlgrid_amd.envs import ClutteredMultiGrid  # noqa: E402


def make(vs, ts, B=32768):
    env = ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=vs, view_tile_size=ts) for c in ("red", "blue", "purple")],
                             grid_size=15, clutter_density=0.15, batch_size=B, strict=False)
    env.reset()
    return env


def timed(env, ptr, iters=20):
    ms = C.c_float(0)
    N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), ptr, iters, C.byref(ms), env._stream()))
    return ms.value


if "--generic-only" in sys.argv:
    for vs, ts in ((7, 5), (7, 11)):
        env = make(vs, ts)
        ms = timed(env, env.obs.data_ptr())
        print("view %d tile %d: %.3f ms  %.0f GB/s" % (vs, ts, ms, env.obs.numel() / ms / 1e6))
        del env
    sys.exit(0)
env = make(7, 8)
nbytes = env.obs.numel()
buf = torch.empty(nbytes + 4096, dtype=torch.uint8, device=env.obs.device)
base = (buf.data_ptr() + 255) // 256 * 256
for off in (0, 16, 32, 48, 64, 80, 128, 0):
    ms = timed(env, base + off)
    print("tile 8, output offset %3d B: %.3f ms  %.0f GB/s" % (off, ms, nbytes / ms / 1e6))
del env, buf
for vs, ts in ((7, 5), (7, 6), (7, 11), (7, 12)):
    env = make(vs, ts)
    ms = timed(env, env.obs.data_ptr())
    print("view %d tile %d: %.3f ms  %.0f GB/s" % (vs, ts, ms, env.obs.numel() / ms / 1e6))
    del env
