#!/usr/bin/env python
"""Construction instead of search: the obs raster (the real kernel, mg_time_render_obs) timed into observation
buffers of the bench workload that differ only in HOW their memory was obtained —
    torch      torch.empty (the caching allocator: hipMalloc underneath)
    hipmalloc  raw hipMalloc through the library (mg_obs_alloc chunk 0)
    vmm-2M     one virtual range backed by 2 MiB physical handles (mg_obs_alloc, the product's construction)
    vmm-32M / vmm-whole   bigger handles / one handle for everything
— a dense fill of the same buffers, a page-scattered read probe (one dword from each of 2^22 random places: what
a translation miss costs shows here), and what each construction costs in time.  One line per buffer."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vmm_buffer import VmmBuffer as _LibBuffer, ab_lib  # noqa: E402  (sets MARLGRID_HIP_LIB: measurement build)
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B = int(os.environ.get("BATCH", "32768"))
env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, auto_reset=True, strict=False, place_obs=False)
env.reset()
dev = env.device
nbytes = env.obs.numel()
ms = C.c_float(0)
L = ab_lib()
g = torch.Generator(device="cpu").manual_seed(1)
idx = (torch.randint(0, nbytes // 4, (1 << 22,), generator=g)).to(dev)


def raster(ptr, iters=5):
    N.check(L.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), C.c_void_p(ptr), iters, C.byref(ms), env._stream()))
    return ms.value


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters


def probe(kind, count, chunk):
    held = []
    for i in range(count):
        t0 = time.perf_counter()
        if kind == "torch":
            t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            mem = None
        else:
            mem = _LibBuffer(L, nbytes, dev, chunk)
            if not mem.ok:
                print("%-10s %2d  allocation failed" % (kind, i))
                break
            t = mem.tensor((nbytes,))
        torch.cuda.synchronize()
        t_alloc = time.perf_counter() - t0
        r = raster(t.data_ptr())
        f = timed(lambda: t.fill_(3))
        r2 = raster(t.data_ptr())
        t32 = t[: nbytes // 4 * 4].view(torch.int32)
        s = timed(lambda: torch.index_select(t32, 0, idx), 3)
        print("%-10s %2d  va %#014x  alloc %7.1f ms  raster %.4f / %.4f ms = %5.0f GB/s  fill %.4f ms = %5.0f GB/s  "
              "scattered read %.3f ms" % (kind, i, t.data_ptr(), t_alloc * 1e3, r, r2, nbytes / min(r, r2) / 1e6, f,
                                          nbytes / f / 1e6, s), flush=True)
        held.append((t, mem))
    return held


print("obs buffer %.1f MB; free %.1f GiB" % (nbytes / 1e6, torch.cuda.mem_get_info()[0] / 2**30))
plan = [("torch", 8, None), ("hipmalloc", 8, 0), ("vmm-2M", 8, 2 << 20), ("vmm-32M", 4, 32 << 20), ("vmm-whole", 4, -1),
        ("vmm-2M", 4, 2 << 20), ("hipmalloc", 4, 0)]
keep = []
for kind, count, chunk in plan:
    keep.append(probe(kind, count, chunk))      # everything stays allocated: later rows see a fuller HBM
del keep
