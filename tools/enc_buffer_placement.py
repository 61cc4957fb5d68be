"""Does it matter WHERE the encoding of mg_step_render_encode is written?  Ten allocations (the allocator stirred between them), two
offsets each: whole step with the encoding against the step alone.  (profiles/r06: +3.0 ... +3.6 %: no placement class to find.)"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from marlgrid_amd import _native as N
from marlgrid_amd.envs import make
B = 32768
env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, auto_reset=True, strict=False)
env.reset()
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (B, 3), generator=g).cuda() for _ in range(16)]
L, cfg, st, prog = env._lib, C.byref(env._cfg), C.byref(env._state), C.byref(env._reset_prog)
stream = env._stream()
def run(f, reps=5):
    out = []
    for r in range(reps):
        for i in range(10): f(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(100): f(i)
        b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / 100)
    return statistics.median(out)
plain = lambda i: N.check(L.mg_step_render(cfg, st, acts[i % 16].data_ptr(), 8, env.rewards.data_ptr(), prog, env.obs.data_ptr(), stream))
base = run(plain)
print("step alone %.4f ms" % base)
keep = []
nb = B * 15 * 15 * 3
for c in range(10):
    pad = torch.empty(((c * 37 + 11) % 97 + 1) * (1 << 20), dtype=torch.uint8, device="cuda")     # stir the allocator
    big = torch.empty(nb + (1 << 21), dtype=torch.uint8, device="cuda")
    keep += [pad, big]
    for off in (0, 4096 * 3 + 64):
        ptr = big.data_ptr() + off
        f = lambda i: N.check(L.mg_step_render_encode(cfg, st, acts[i % 16].data_ptr(), 8, env.rewards.data_ptr(), prog, env.obs.data_ptr(), ptr, stream))
        t = run(f, 3)
        print("candidate %d +%6d: ptr %% 2MiB = %7d  %.4f ms  %+.2f %%" % (c, off, ptr % (1 << 21), t, 100 * (t / base - 1)))
print("step alone again %.4f ms" % run(plain))
