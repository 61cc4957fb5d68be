#!/usr/bin/env python
"""Where does a wave of the fused launch spend its time before its first store?  Measurement build: every wave
stamps wall_clock64 (100 MHz) at the phase boundaries of its first batch (mg_render.hip, MG_STAMP); this prints
the per-phase durations over all waves of one mg_step_render launch (and of one mg_render_obs launch)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
N.use_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "marlgrid_amd", "csrc",
                           os.environ.get("AB_LIB", "libmarlgrid_hip_ab.so")))      # the measurement build
from marlgrid_amd.envs import make  # noqa: E402

B = int(os.environ.get("B", "32768"))
if os.environ.get("TILE"):          # the bench scenario's shape with another view_tile_size (the instantiations off the fast path)
    from marlgrid_amd.agents import GridAgentInterface
    from marlgrid_amd.envs import ClutteredMultiGrid
    env = ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=7, view_tile_size=int(os.environ["TILE"])) for c in ("red", "blue", "purple")],
                             grid_size=15, clutter_density=0.15, batch_size=B, strict=False, auto_reset=True)
elif os.environ.get("PRESTIGE"):    # PRESTIGE=<agents>,<tile>: the goal-cycle scenario of tools/bench_cases.py ('prestige'-coloured agents)
    from marlgrid_amd.agents import GridAgentInterface
    from marlgrid_amd.envs import ClutteredGoalCycleEnv
    _n, _ts = (int(x) for x in os.environ["PRESTIGE"].split(","))
    env = ClutteredGoalCycleEnv(
        agents=[GridAgentInterface(color="prestige", view_size=7, view_tile_size=_ts, view_offset=1) for _ in range(_n)],
        grid_size=13, clutter_density=0.15, n_bonus_tiles=3, max_steps=250, respawn=True, reward_decay=False,
        initial_reward=True, penalty=-1.5, batch_size=B, strict=False, auto_reset=True)
else:
    env = make(os.environ.get("WL", "MarlGrid-3AgentCluttered15x15-v0"), batch_size=B, auto_reset=True, strict=False)
env.reset()
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (B, env.num_agents), generator=g).cuda() for _ in range(16)]
for i in range(40):
    env.step(acts[i % 16])
L, cfg, st = env._lib, C.byref(env._cfg), C.byref(env._state)
stamps = torch.zeros(65536 * 24, dtype=torch.int64, device="cuda")
names = ["tables + atlas -> LDS, barrier", "stage grids (+ step_load)", "step_run (+ records, write-back)",
         "views of the first group", "raster of the first env", "rest of the run"]


def report(title):
    torch.cuda.synchronize()
    t = stamps.view(-1, 24).cpu()
    if os.environ.get("STAMPS_OUT"):
        import numpy as np
        np.save(os.path.join(os.environ["STAMPS_OUT"], "stamps_%s_%d.npy" % (title, report.n)), t[t[:, 6] != 0].numpy())
        report.n += 1
    t = t[t[:, 6] != 0].double() / 100.0          # us
    t0 = t[:, 0].min()
    print("%s: %d waves; first entry -> last exit %.1f us; entry skew %.1f us" %
          (title, len(t), (t[:, 6].max() - t0).item(), (t[:, 0].max() - t0).item()))
    for k in range(6):
        d = t[:, k + 1] - t[:, k]
        print("   %-36s mean %7.2f  min %7.2f  max %7.2f us" % (names[k], d.mean().item(), d.min().item(), d.max().item()))
    ex = t[:, 6] - t0
    q = torch.quantile(ex, torch.tensor([0.0, 0.01, 0.1, 0.5, 0.9, 0.99, 1.0], dtype=torch.float64))
    print("   wave exit time after the first entry: min %.1f  p1 %.1f  p10 %.1f  median %.1f  p90 %.1f  p99 %.1f  max %.1f us" % tuple(q.tolist()))
    fs = t[:, 4] - t0       # about when the wave's first store is issued
    q = torch.quantile(fs, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64))
    print("   first views done after the first entry: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us" % tuple(q.tolist()))
    by_xcd = [ex[(torch.arange(len(ex)) // 16) % 8 == x].mean().item() for x in range(8)]   # workgroup b runs on XCD b % 8
    print("   mean exit time by XCD (workgroup %% 8): %s" % " ".join("%.1f" % v for v in by_xcd))
    by_wave = [ex[torch.arange(len(ex)) % 4 == w].mean().item() for w in range(4)]
    print("   mean exit time by look-ahead depth 1/2/4/8 (wave %% 4): %s" % " ".join("%.1f" % v for v in by_wave))
    if (t[:, 13] != 0).any():
        for nm, x, y in (("entry -> launch constants set up", 0, 13), ("-> env loop entered (hoisted invariants)", 13, 14),
                         ("prologue's loads issued and back", 14, 15), ("LDS stores + barrier", 15, 1)):
            d = t[:, y] - t[:, x]
            print("      head: %-40s mean %7.2f  min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us" %
                  (nm, d.mean().item(), d.min().item(), d.median().item(), torch.quantile(d, 0.9).item(), d.max().item()))
    inner = ["spawns, front cells, shuffle", "agent loop", "done / respawn / fused reset", "write-back + RNG head refill"]
    for base, which in ((8, "first batch"), (16, "last batch")):
        if not (t[:, base] != 0).any():        # inside step_run (lane 0's env)
            continue
        print("      step_run of the wave's %s: stage/step_load -> entry mean %.2f us" % (which, (t[:, base] - t[:, 2]).mean().item()) if base == 8
              else "      step_run of the wave's %s (entered %.1f us after the wave's entry, mean):" % (which, (t[:, base] - t[:, 0]).mean().item()))
        for k in range(4):
            d = t[:, base + 1 + k] - t[:, base + k]
            print("         %-30s mean %7.2f  min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us" %
                  (inner[k], d.mean().item(), d.min().item(), d.median().item(), torch.quantile(d, 0.9).item(), d.max().item()))
    d = t[:, 4] - t[:, 0]
    print("   entry -> first views done            mean %7.2f  min %7.2f  max %7.2f us" % (d.mean().item(), d.min().item(), d.max().item()))
    stamps.zero_()
    torch.cuda.synchronize()


report.n = 0
L.mg_ab_stamps.restype = C.c_int
assert L.mg_ab_stamps(C.c_void_p(stamps.data_ptr())) == 0
for rep in range(2):
    N.check(L.mg_step_render(cfg, st, acts[rep].data_ptr(), 8, env.rewards.data_ptr(), C.byref(env._reset_prog),
                             env.obs.data_ptr(), env._stream()))
    report("mg_step_render")
    N.check(L.mg_render_obs(cfg, st, env.obs.data_ptr(), None, None, None, env._stream()))
    report("mg_render_obs")
L.mg_ab_stamps(C.c_void_p(0))
env.check_errors()
