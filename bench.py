#!/usr/bin/env python
"""bench.py — agent-steps/sec of the batched step engine on MarlGrid-3AgentCluttered15x15-v0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch-per-gpu B]

A "step" is one `env.step(actions)` over the whole per-GPU batch: action apply (mg_step), device
auto-reset of finished episodes (mg_reset with the done flags as mask) and the observation raster
(mg_render_obs), with every input already resident in HBM.  The env batch shards over GPUs with no
collective on the data path (weak scaling: 32 768 envs per GPU; 8 GPUs = BASELINE.json's 262 144).
For N > 1 launch with torch.distributed.run (one rank per GPU); rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOAD = "MarlGrid-3AgentCluttered15x15-v0"
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-per-gpu", type=int, default=32768)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default=WORKLOAD, help="exploration only; the contract line uses the default")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        dist.init_process_group(backend="nccl", device_id=dev)
    n_gpus = world if distributed else 1
    if args.gpus != n_gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d (launch with torch.distributed.run)" % (args.gpus, world),
              file=sys.stderr)

    from marlgrid_amd.envs import make
    from marlgrid_amd import _native as N
    from marlgrid_amd import sharding
    B = args.batch_per_gpu
    # the env batch shards embarrassingly: rank r owns global envs [r*B, (r+1)*B), seeds 1337 + id
    seeds = sharding.shard_seeds(1337, B * n_gpus, rank, n_gpus)
    assert len(seeds) == B
    wl = args.workload
    if wl == "Custom-8AgentCluttered30x30":     # BASELINE.json configs[4]; not a registered id upstream
        from marlgrid_amd.agents import GridAgentInterface
        from marlgrid_amd.envs import ClutteredMultiGrid
        cols = ["red", "blue", "purple", "orange", "olive", "pink", "cyan", "yellow"]
        env = ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=9, view_tile_size=8) for c in cols],
                                 grid_size=30, clutter_density=0.15, batch_size=B, device=dev, seeds=seeds,
                                 auto_reset=True, strict=False)
    else:
        env = make(wl, batch_size=B, device=dev, seeds=seeds, auto_reset=True, strict=False)
    env.reset()
    n = env.num_agents
    K, Wm = args.steps, args.warmup
    g = torch.Generator(device="cpu").manual_seed(rank)
    pool = [torch.randint(0, 7, (B, n), generator=g).to(dev) for _ in range(min(K + Wm, 64))]

    for i in range(Wm):
        env.step(pool[i % len(pool)])
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(K):
        env.step(pool[(Wm + i) % len(pool)])
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, device=dev)      # the slowest rank defines the step time
    env.check_errors()

    # dominant kernel: mg_render_obs, timed live with HIP events on the launch stream
    vs, ts = env.view_size, env.tile_size
    P = vs * ts
    alg_bytes_per_agent_step = P * P * 3 + vs * vs + 8 * n
    avg_ms = C.c_float(0)
    N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), env.obs.data_ptr(), 50,
                                        C.byref(avg_ms), env._stream()))
    render_s = avg_ms.value * 1e-3
    achieved = B * n * alg_bytes_per_agent_step / render_s / 1e9
    # HBM bytes per launch from the PMC passes (rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE, collected
    # separately and corrected as MI355X_MICROARCH.md prescribes; summary under profiles/): only valid
    # for the exact workload / batch they were collected on
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01", "pmc_hbm_bytes.json")
    if wl == WORKLOAD and B == 32768 and os.path.exists(pmc):
        try:
            t = json.load(open(pmc)).get("render_kernel_hbm_bytes_per_launch")
            traffic = float(t) if t else None
        except Exception:
            traffic = None

    out = None
    if rank == 0:
        total_agent_steps = n_gpus * B * n * K
        out = {
            "metric": "agent-steps/sec at batch B, 3AgentCluttered15x15, 1/2/4/8 MI355X",
            "value": total_agent_steps / elapsed,
            "unit": "agent-steps/s",
            "n_gpus": n_gpus, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl, "batch_per_gpu": B, "global_batch": B * n_gpus, "n_agents": n,
                       "view_size": vs, "tile_size": ts, "obs_shape": [B * n_gpus, n, P, P, 3],
                       "actions": "uniform over 7 ids, torch.randint seed=rank", "auto_reset": True,
                       "sharding": "env batch split contiguously, no collectives"},
            "roofline": {"bound": "hbm", "kernel": "mg::render_kernel<%d, %d, %d, 0>" % (vs, ts, 16 if B >= 4096 else 4), "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "kernel_ms": avg_ms.value, "algorithmic_bytes_per_agent_step": alg_bytes_per_agent_step,
                         "algorithmic_bytes_per_launch": B * n * alg_bytes_per_agent_step,
                         "traffic": traffic, "traffic_unit": "bytes per launch (PMC: WRITE_SIZE + 2*FETCH_SIZE)"},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, wl)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(budget_s, workload=WORKLOAD):
    """The parity-checked CPU oracle (a C port of the reference algorithm, OpenMP over envs) on a
    bounded sample of the same workload, on this box's host cores.  The Python reference cannot
    travel to the GPU box; its own measured speed (2 377 agent-steps/s on one Xeon core) is in
    BASELINE.md."""
    import numpy as np
    import scenarios
    from oracle import oracle as O
    Bc = 2048
    seeds = 1337 + np.arange(Bc)
    orc = O.OracleBatch(scenarios.registered(workload), seeds)
    orc.reset()
    threads = orc.max_threads()
    rng = np.random.RandomState(0)
    acts = [rng.randint(0, 7, size=(Bc, orc.n)) for _ in range(16)]
    orc.step(acts[0], auto_reset=True, reuse_obs=True)
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget_s:
        orc.step(acts[steps % 16], auto_reset=True, reuse_obs=True)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": Bc * orc.n * steps / dt, "unit": "agent-steps/s", "cores": int(threads), "kind": "port",
            "sample": "%d envs x %d steps of %s (C oracle, OpenMP, obs render included), %.1f s" % (
                Bc, steps, workload, dt),
            "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    main()
