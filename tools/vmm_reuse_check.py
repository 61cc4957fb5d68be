#!/usr/bin/env python
"""Correctness of remapped observation buffers (HIP virtual memory management): (1) the same five steps with
place_obs = False / 'search' (and, in round 3, 'vmm') must give identical observations — every mismatch is located (which bytes,
in units of 2 MiB); (2) a virtual range that is freed and handed to ANOTHER buffer must not alias the first one's
physical memory (stale translations), directly: write through the reused range, read both buffers back."""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from vmm_buffer import VmmBuffer as _LibBuffer, ab_lib  # noqa: E402  (sets MARLGRID_HIP_LIB: measurement build)
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B = int(os.environ.get("BATCH", "4096"))


def run(mode):
    env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, place_obs=mode)
    env.reset()
    g = torch.Generator().manual_seed(3)
    outs = []
    for _ in range(5):
        o, r, d, _ = env.step(torch.randint(0, 7, (B, 3), generator=g))
        outs.append(o.clone().cpu())
    info = getattr(env._groups[0], "placement_ms", None)
    ptrs = [t.data_ptr() for t in env._groups[0].ring]
    del env
    gc.collect()
    return outs, info, ptrs


ref, _, _ = run(False)
for mode in ("search", False):      # (round 3's place_obs="vmm" — buffers built and freed with the VMM API — failed here)
    outs, info, ptrs = run(mode)
    bad = 0
    for t, (a, b) in enumerate(zip(ref, outs)):
        ne = (a != b).reshape(-1)
        nb = int(ne.sum())
        bad += nb
        if nb:
            idx = torch.nonzero(ne).reshape(-1)
            chunks = sorted(set((idx // (2 << 20)).tolist()))
            print("  mode %r step %d: %d bytes differ, offsets %d .. %d, 2 MiB chunks %s, envs %s" % (
                mode, t, nb, int(idx[0]), int(idx[-1]), chunks[:12], sorted(set((idx // (3 * 56 * 56 * 3)).tolist()))[:12]))
    print("mode %r: %s (ring at %s; %s)" % (mode, "IDENTICAL to place_obs=False" if bad == 0 else "%d bytes differ" % bad,
                                            ["%#x" % p for p in ptrs],
                                            {k: v for k, v in (info or {}).items() if k in ("ranges_tried", "candidates", "stopped")}))

# (2) a freed range handed to another buffer
L = ab_lib()
dev = torch.device("cuda", torch.cuda.current_device())
nb = 64 << 20
a = _LibBuffer(L, nb, dev, 2 << 20)
ta = a.tensor((nb,))
ta.fill_(1)
torch.cuda.synchronize()
va1 = a.ptr
del ta
assert a.rebase()
va2 = a.ptr
a.trim()                                   # va1 goes back
b = _LibBuffer(L, nb, dev, 2 << 20)        # ... and may be handed to b
tb = b.tensor((nb,))
tb.fill_(7)
torch.cuda.synchronize()
ta = a.tensor((nb,))
print("a: range %#x -> %#x; b built at %#x (%s a's old range)" % (va1, va2, b.ptr, "REUSES" if b.ptr == va1 else "not"))
print("   a holds only 1s: %s   b holds only 7s: %s" % (bool((ta == 1).all()), bool((tb == 7).all())))
for k in range(6):                         # repeated rebase / trim cycles with a neighbour in between
    del ta, tb
    assert a.rebase()
    a.trim()
    assert b.rebase()
    b.trim()
    ta, tb = a.tensor((nb,)), b.tensor((nb,))
    ta.add_(2)
    tb.add_(2)
    torch.cuda.synchronize()
    ok = bool((ta == 1 + 2 * (k + 1)).all()) and bool((tb == 7 + 2 * (k + 1)).all())
    print("   cycle %d: a at %#x, b at %#x, contents %s" % (k, a.ptr, b.ptr, "ok" if ok else "WRONG"))
