#!/usr/bin/env python
"""Every configuration that is NOT the contract line, on the product library with placed observation buffers
(place_obs="thorough": this tool owns the GPU and measures kernels, not the placement search's bounded default): the other BASELINE.json configs at their per-GPU batch, the tile / view sizes off the fast path,
the 'prestige' cases and the reference's one runnable example.  Per case one JSON line: the raster alone
(mg_render_obs, HIP events inside the library) and the whole env.step() (HIP events around 30 steps), both as a
fraction of 8 TB/s on the observation bytes, and mg_encode (MultiGrid.encode for the batch) alone against ITS algorithmic
bytes (grid + records in, 3 bytes per cell out).  usage: bench_cases.py [--place=default|thorough] [substring of a case label ...]
--place=default: the constructor's own bounded placement search (what a user gets out of the box) instead of "thorough"."""
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.agents import GridAgentInterface  # noqa: E402
from marlgrid_amd.envs import ClutteredGoalCycleEnv, ClutteredMultiGrid, make  # noqa: E402

COLS = ["red", "blue", "purple", "orange", "olive", "pink", "cyan", "yellow"]
PLACE = "thorough"
for a in list(sys.argv[1:]):
    if a.startswith("--place="):
        PLACE = a.split("=", 1)[1]
        sys.argv.remove(a)
assert PLACE in ("default", "thorough")
KW = dict(strict=False, auto_reset=True, place_obs="thorough" if PLACE == "thorough" else True)


def agents(n, vs, ts, **kw):
    return [GridAgentInterface(color=COLS[k % 8], view_size=vs, view_tile_size=ts, **kw) for k in range(n)]


def cluttered(ts, vs=7, B=32768):
    return lambda: ClutteredMultiGrid(agents=agents(3, vs, ts), grid_size=15, clutter_density=0.15, batch_size=B, **KW)


def goalcycle(n, ts, B=32768):
    return lambda: ClutteredGoalCycleEnv(
        agents=[GridAgentInterface(color="prestige", view_size=7, view_tile_size=ts, view_offset=1) for _ in range(n)],
        grid_size=13, clutter_density=0.15, n_bonus_tiles=3, max_steps=250, respawn=True, reward_decay=False,
        initial_reward=True, penalty=-1.5, batch_size=B, **KW)


cases = [
    ("configs[1] MarlGrid-3AgentCluttered11x11-v0, B = 4 096", lambda: make("MarlGrid-3AgentCluttered11x11-v0", batch_size=4096, **KW)),
    ("configs[2] MarlGrid-4AgentEmpty9x9-v0, B = 65 536", lambda: make("MarlGrid-4AgentEmpty9x9-v0", batch_size=65536, **KW)),
    ("configs[3] MarlGrid-3AgentCluttered15x15-v0, B = 32 768 (the contract line's workload)", lambda: make("MarlGrid-3AgentCluttered15x15-v0", batch_size=32768, **KW)),
    ("configs[4] 8-agent Cluttered 30x30 view 9, B = 131 072 per GPU", lambda: ClutteredMultiGrid(
        agents=agents(8, 9, 8), grid_size=30, clutter_density=0.15, batch_size=131072, **KW)),
    ("tile 5 (GridAgentInterface default)", cluttered(5)),
    ("tile 6", cluttered(6)),
    ("tile 11", cluttered(11)),
    ("tile 16", cluttered(16, B=16384)),
    ("tile 32 (atlas in global memory)", cluttered(32, B=4096)),
    ("view 11 (run-time view size), tile 8", cluttered(8, vs=11, B=16384)),
    ("view 6 (even), tile 8", cluttered(8, vs=6)),
    ("goal cycle, 3 'prestige' agents, tile 8", goalcycle(3, 8)),
    ("human_player: goal cycle, 1 'prestige' agent, tile 11", goalcycle(1, 11)),
]
only = sys.argv[1:]
for label, mk in cases:
    if only and not any(o in label for o in only):
        continue
    try:
        env = mk()
        env.reset()
        B = env.batch_size
        g = torch.Generator().manual_seed(0)
        acts = [torch.randint(0, 7, (B, env.num_agents), generator=g).to(env.device) for _ in range(8)]
        env.step(acts[0])
        ms = C.c_float(0)
        rast, step = [], []
        for rep in range(5):
            N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), env.obs.data_ptr(), 20, C.byref(ms), env._stream()))
            rast.append(ms.value)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            env.step(acts[0])
            a.record()
            for i in range(30):
                env.step(acts[i % 8])
            b.record()
            b.synchronize()
            step.append(a.elapsed_time(b) / 30)
        enc_out = torch.empty((B, env.width, env.height, 3), dtype=torch.uint8, device=env.device)
        enc = []
        for rep in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            N.check(env._lib.mg_encode(C.byref(env._cfg), C.byref(env._state), None, enc_out.data_ptr(), env._stream()))
            a.record()
            for i in range(50):
                N.check(env._lib.mg_encode(C.byref(env._cfg), C.byref(env._state), None, enc_out.data_ptr(), env._stream()))
            b.record()
            b.synchronize()
            enc.append(a.elapsed_time(b) / 50)
        del enc_out
        # the step with MultiGrid.encode of the batch written by the SAME launch (encode_in_step: mg_step_render_encode where the
        # library has it compiled in, an mg_encode launch behind the step elsewhere), interleaved with the plain step
        env.encode_in_step = True
        env.step(acts[0])
        step_enc, step_again = [], []
        for rep in range(5):
            for flag, into in ((True, step_enc), (False, step_again)):
                env.encode_in_step = flag
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                env.step(acts[0])
                a.record()
                for i in range(30):
                    env.step(acts[i % 8])
                b.record()
                b.synchronize()
                into.append(a.elapsed_time(b) / 30)
        enc_in_launch = bool(env._enc_fused) and env.fused_step and not env._hetero
        enc_bytes = B * (env.cells_stride + 8 * env.num_agents + 3 * env.width * env.height)
        env.check_errors()
        nb = env.obs.numel()
        pm = getattr(env._groups[0], "placement_ms", None) or {}
        r, s = statistics.median(rast), statistics.median(step)
        print(json.dumps({"case": label, "B": B, "n": env.num_agents, "view": env.view_size, "tile": env.tile_size, "obs_bytes": nb,
                          "raster_ms": r, "raster_frac_of_8TBps": nb / r / 1e6 / 8000, "step_ms": s,
                          "step_frac_of_8TBps": nb / s / 1e6 / 8000, "agent_steps_per_s": B * env.num_agents / s * 1e3,
                          "kernel": env.kernel_name, "place_obs": PLACE,
                          "encode_ms": statistics.median(enc), "encode_bytes": enc_bytes,
                          "encode_frac_of_8TBps": enc_bytes / statistics.median(enc) / 1e6 / 8000,
                          "step_with_encode_ms": statistics.median(step_enc), "encode_in_the_step_launch": enc_in_launch,
                          "step_with_encode_vs_step_pct": 100 * (statistics.median(step_enc) / statistics.median(step_again) - 1),
                          "obs_placement": {k: pm.get(k) for k in ("found", "kept", "candidates", "stopped", "seconds", "pinned_bytes",
                                                                   "budget_bytes", "median_ms", "candidate_bytes")}}), flush=True)
        del env
    except Exception as e:      # noqa: BLE001 — one case must not cost the others
        print(json.dumps({"case": label, "error": "%s: %s" % (type(e).__name__, e)}), flush=True)
    torch.cuda.empty_cache()
