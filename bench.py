#!/usr/bin/env python
"""bench.py — agent-steps/sec of the batched step engine on MarlGrid-3AgentCluttered15x15-v0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch-per-gpu B]

A "step" is one `env.step(actions)` over the whole per-GPU batch: ONE launch on one stream — mg_step_render:
the action loop, the reset of every env whose episode just ended and the observation raster in one kernel —
with every input already resident in HBM (`--unfused`: the two-launch form, mg_step + mg_render_obs).  The
env batch shards over GPUs with no collective on the data path (weak scaling: 32 768 envs per GPU; 8 GPUs =
BASELINE.json's 262 144).  `--gpus N` with N > 1 starts its own N ranks (torch.distributed.run, one rank per
GPU, RCCL control plane) unless it is already running under a launcher (WORLD_SIZE set); rank 0 prints ONE
JSON line.

How the line is measured (one self-consistent measurement, not a collage):
  * after W warm-up steps, BLOCKS of exactly K steps are timed, each bracketed by barrier + synchronize on
    both sides and MAX-reduced over ranks, until >= --min-seconds of timed steps (the driver's K = 20 is
    3.5 ms: far below what clocks and samplers resolve).  `ms_per_step` and `value` are the SUSTAINED rate
    over all plain blocks — total steps / total time, nothing dropped: every env of this workload runs into
    max_steps together, so one step in 100 resets the whole batch inside the launch and costs twice the
    others.  The median block (which hides those steps) is `value_median_block`; min / max / first / last
    and the blocks beyond 3x the median (`outliers`) are listed.
  * EVERY block carries two HIP events on the launch stream — before its first launch and after its last (an
    event costs ~2 us of stream time: 0.1 % of a 20-step block) — so a block's launch interval (GPU span / K:
    launch + dispatch gap) and its host-timed ms per step describe the same K launches.  `kernels` /
    `roofline.kernel_ms` report the MEDIAN block, with mean / min / max and the blocks beyond 3x the median
    (`outliers`) beside it; `closure` = that median / the median host-timed ms per step.
  * `clocks` holds rocm-smi samples before the first and after the last block; `roofline.traffic`
    is measured in this run (tools/pmc.py: rocprofv3 --pmc passes in a child process, N = 1 only).
  * `extra.strong_n1`: the full headline batch (262 144 envs) on ONE GPU, same fields.
  * `extra.step_with_encode`: the step with `MultiGrid.encode` of the batch written by its own launch, against the step alone.
  * `extra.pipelined_shards`: the per-GPU batch (and twice it) as TWO envs on two streams, stepped without a join
    (marlgrid_amd.sharding.ShardPipeline): what overlapping launches of independent shards are worth, next to the
    same number of envs as one env.  The contract line itself is ONE env on one stream.
  * `parity_after_timed`: when the timed region is over, eight envs of the rank's batch are REPLAYED on the CPU oracle
    (child process; the checker, never the thing measured) with the very actions the bench fed them — reset, the W
    warm-up steps and every timed step, auto-reset on `done` — and grid, agent records (stack order included), step
    counter, MT19937 state, the last step's rewards / done and the last observations are compared: the state the
    timed ~60 000 steps per env end in is the reference's (marlgrid/base.py:501-653 over ~600 episodes per env).
  * the interpreter's garbage is collected (and the survivors frozen) before every timed region: a full collection in
    the middle of a 20-step block is a 50 ms host stall (settle_interpreter).
The run refuses to start if an MG_* / MARLGRID_* environment variable is set, and echoes the
library's build id.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

WORKLOAD = "MarlGrid-3AgentCluttered15x15-v0"
METRIC = "agent-steps/sec at batch B, 3AgentCluttered15x15, 1/2/4/8 MI355X"
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec


class Control(object):
    """The benchmark's control plane: barriers and small gathers over ranks.  Nothing on the data path goes
    through it (one RNG per env, nothing shared: marlgrid/base.py:371-374).

    The default process group is ALWAYS gloo (TCP on 127.0.0.1: it comes up wherever torch.distributed does) and
    carries the gathers and every agreement between ranks.  With prefer="nccl" (one rank per GPU) an RCCL group is
    formed on top of it for the barriers and PROBED with one all-reduce under a timeout; the ranks then agree over
    gloo whether it worked for ALL of them.  If it did not — an exception at init, at the first collective, or a
    timeout on any rank — the barriers run over gloo too, `fallback` holds the exception text of every rank that
    failed, and the run goes on: a control plane that cannot come up must not cost the scaling curve."""

    PROBE_TIMEOUT_S = 90

    def __init__(self, prefer, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.device = device
        self.backend = None          # what the barriers run on: "nccl" | "gloo" | None (one rank)
        self.fallback = None         # why not RCCL, when RCCL was asked for
        self._pg = None
        if self.world == 1:
            return
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # gloo announces its connections on stdout ("[Gloo] Rank 0 is connected to ..."): stdout is for the ONE
        # JSON line, so the descriptor points at stderr while the group forms
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        self.backend = "gloo"
        if prefer == "nccl":
            err = self._try_rccl()
            errs = self.gather_objects(err)
            if all(e is None for e in errs):
                self.backend = "nccl"
            else:
                self._pg = None
                self.fallback = {"asked_for": "nccl", "using": "gloo",
                                 "errors_by_rank": {str(r): e for r, e in enumerate(errs) if e is not None}}

    def _try_rccl(self):
        """form the RCCL group and run one all-reduce through it; None, or the text of what went wrong"""
        import datetime
        import torch
        import torch.distributed as dist
        # a collective that cannot complete must raise HERE (after the timeout), not abort the process from
        # the watchdog thread: blocking wait turns the timeout into an exception on this thread
        os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
        os.environ.setdefault("NCCL_BLOCKING_WAIT", "1")
        try:
            hook = os.environ.get("BENCH_TEST_FAIL_NCCL")      # tests: the failure path without breaking RCCL
            if hook and (hook == "all" or int(hook) == self.rank):
                raise RuntimeError("forced by BENCH_TEST_FAIL_NCCL=%s" % hook)
            pg = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=self.PROBE_TIMEOUT_S))
            t = torch.ones(1, device=self.device)
            work = dist.all_reduce(t, group=pg, async_op=True)
            work.wait(timeout=datetime.timedelta(seconds=self.PROBE_TIMEOUT_S))
            torch.cuda.synchronize(self.device)
            if int(t.item()) != self.world:
                raise RuntimeError("RCCL all-reduce returned %r, expected %d" % (t.item(), self.world))
            self._pg = pg
            return None
        except BaseException as e:     # noqa: BLE001 — whatever it is, the line must still be printed
            if isinstance(e, (KeyboardInterrupt, SystemExit)):
                raise
            return ("%s: %s" % (type(e).__name__, e))[:600]

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            if self._pg is not None:
                dist.barrier(group=self._pg, device_ids=[self.device.index])
            else:
                dist.barrier()

    def gather(self, values):
        """every rank's tuple of floats, on every rank (list of `world` lists); over gloo, host memory"""
        if self.world == 1:
            return [[float(v) for v in values]]
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return [[float(x) for x in o.tolist()] for o in out]

    def gather_objects(self, obj):
        if self.world == 1:
            return [obj]
        import torch.distributed as dist
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def describe(self):
        if self.world == 1:
            return None
        return {"barriers": self.backend, "gathers": "gloo", "fallback": self.fallback,
                "test_hooks": {k: v for k, v in os.environ.items() if k.startswith("BENCH_TEST_")} or None}

    def close(self):
        if self.world > 1:
            import torch.distributed as dist
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception:       # noqa: BLE001 — the line is out; a teardown hiccup must not turn rc != 0
                pass


def pin_to_gpu_numa(dev_index):
    """Pin this rank's launch thread to the CPUs of its GPU's NUMA node (sysfs: the PCI device's local_cpulist).
    Eight Python launch loops at 6 k launches/s each should not migrate across sockets or share whatever the
    scheduler gives them.  Returns what was done (reported per rank); never raises."""
    info = {"pinned": False}
    try:
        import torch
        before = sorted(os.sched_getaffinity(0))
        info["cpus_before"] = len(before)
        p = torch.cuda.get_device_properties(dev_index)
        dom, bus, devn = (getattr(p, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        if bus is None:
            info["why"] = "torch reports no PCI ids for the device"
            return info, before
        bdf = "%04x:%02x:%02x.0" % (int(dom or 0), int(bus), int(devn or 0))
        info["pci"] = bdf
        base = "/sys/bus/pci/devices/%s/" % bdf
        node = int(open(base + "numa_node").read().strip())
        info["numa_node"] = node
        cpulist = open(base + "local_cpulist").read().strip()
        cpus = set()
        for part in cpulist.split(","):
            if part:
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= set(before)
        if node < 0 or not cpus or len(cpus) == len(before):
            info["why"] = "no NUMA locality to pin to (numa_node %d, %d local of %d allowed CPUs)" % (node, len(cpus), len(before))
            return info, before
        os.sched_setaffinity(0, cpus)
        info.update(pinned=True, cpus=cpulist, n_cpus=len(cpus))
        return info, before
    except Exception as e:      # noqa: BLE001
        info["why"] = "%s: %s" % (type(e).__name__, e)
        return info, None


OBS_BUFFERS_NOTE = {
    "search": "placed by the library at construction (mg_obs_place: candidate HBM allocations timed with the raster itself; "
              "<= 2 s, live candidates <= min(free / 4, 32 GiB); ms per launch in obs_placement); a rank whose search found "
              "nothing in the fast class places once more, thoroughly, before anything is timed (placement_retries)",
    "thorough": "placed by the library at construction, long search (mg_obs_place, MG_PLACE_THOROUGH | MG_PLACE_STIR)",
    False: "plain torch allocations"}


def ensure_placed(env, share=1):
    """A rank whose observation buffers are not in the fast class places them once more before anything is timed — the
    long search (larger candidates, a second pass, one big allocate-and-free: this process owns its GPU) — and says so:
    with MAX-over-ranks timing one rank's plain-speed buffers would cost the whole N-GPU point a fifth.  Returns the
    number of retries (0 or 1)."""
    import torch
    retries = 0
    for gi, pm in enumerate(getattr(env, "obs_placement", None) or []):
        if pm is not None and not pm.get("found") and pm.get("stopped") != "out of memory":
            first = {k: pm.get(k) for k in ("kept", "candidates", "stopped", "seconds")}
            free = torch.cuda.mem_get_info(env.device)[0]
            # (ranks that share a GPU — --oversubscribe — count on their part of what is free, and do not stir each other's free lists)
            env._place_obs_buffers(thorough=True, stir=share == 1, seconds=6.0, budget=min(free // (2 * share), 128 << 30), share=share)
            retries = 1
            for pm2 in env.obs_placement:
                if pm2 is not None:
                    pm2["first_attempt"] = first
            break
    return retries


def device_identity(index):
    """what distinguishes this rank's GPU from the others: PCI bus id / uuid where torch reports them"""
    import torch
    p = torch.cuda.get_device_properties(index)
    out = {"name": p.name, "cus": p.multi_processor_count}
    for k in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
        v = getattr(p, k, None)
        if v is not None:
            out[k] = str(v)
    return out


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start N ranks of this very command under
    torch.distributed.run on this node (one rank per GPU, rendezvous on 127.0.0.1) and hand their exit code
    back.  The children see WORLD_SIZE and take the normal path."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, BENCH_SELF_LAUNCHED="bench.py --gpus %d (torch.distributed.run)" % args.gpus)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def timed_blocks(step_fn, sync_fn, ctl, K, min_seconds, max_blocks, probe_ctl=None):
    """Blocks of exactly K steps, each bracketed by sync + barrier + sync on both sides; the block
    time is the MAX over ranks of each rank's own K steps (opening barrier -> its own synchronize).  Odd blocks
    are instrumented when `probe_ctl` is given.  Returns a list of dicts (one per block)."""
    blocks, total, i = [], 0.0, 0
    while True:
        instrumented = probe_ctl is not None and (probe_ctl.every_block or i % 2 == 1)
        if instrumented:
            probe_ctl.arm()
        sync_fn()
        ctl.barrier()
        sync_fn()
        t0 = time.perf_counter()
        for j in range(K):
            step_fn(i * K + j)
        sync_fn()
        own = time.perf_counter() - t0                # this rank's K steps alone (a straggler shows here)
        ctl.barrier()
        sync_fn()
        mine = time.perf_counter() - t0
        both = ctl.gather((mine, own))
        # The slowest rank's OWN K steps define the step time: from the opening barrier to this rank's own
        # synchronize.  `mine` also holds the closing barrier — a kernel launch and an all-reduce over RCCL, 30-80 us,
        # 1-2 % of a 20-step block, and nothing at N = 1 (no-op): as the block time it would bend the scaling curve
        # by something that is not the engine.  It rides along as `with_barrier_s`.
        elapsed = max(b[1] for b in both)
        b = {"elapsed_s": elapsed, "with_barrier_s": max(b[0] for b in both),
             "instrumented": instrumented and not probe_ctl.every_block,
             "per_rank_s": [x[1] for x in both]}
        if instrumented:
            b["kernels"] = probe_ctl.collect()
        blocks.append(b)
        total += elapsed
        i += 1
        # `total` comes out of an all_gather: identical on every rank, so all ranks stop together
        if (total >= min_seconds and i >= 4) or i >= max_blocks:
            return blocks


def settle_interpreter():
    """Before a timed region: one full garbage collection now, and what survives it moved out of the collector's way
    (gc.freeze).  A generation-2 collection walks every torch / ctypes / numpy object of the process: one 40-55 ms host
    stall some hundred steps into the run (profiles/r03/stall_probe_python_gc.txt) — hidden when the host runs ahead
    of the GPU, fully exposed in a K = 20 block that synchronises on both sides (it was the one `outliers` block of
    every driver-flags run).  The collector stays enabled."""
    import gc
    gc.collect()
    gc.freeze()


class Probes(object):
    """HIP events on the launch stream (torch's current stream IS the stream the C ABI launches on:
    MultiGridEnv passes torch.cuda.current_stream().cuda_stream to every call).  MultiGridEnv.step calls
    the probe before its first launch (tag 0), between mg_step and mg_render_obs when it runs as two
    launches (tag 1), and after its last launch (tag 2).  One launch per step (mg_step_render): TWO events per
    K-step block — before the block's first launch and after its last — in EVERY block (an event costs ~2 us of
    stream time: 0.1 % of a 20-step block; one per launch would add 1 % to what it measures), so a block's
    launch interval (GPU span / K: launch + dispatch gap) and its host-timed ms per step describe the same K
    launches.  Two launches per step (--unfused, A/B only): three events per step in every other block, for the
    breakdown."""

    def __init__(self, env, K):
        import torch
        self.env, self.K = env, K
        self.fused = bool(env.fused_step) and not env._hetero
        self.every_block = self.fused
        n = 2 if self.fused else 3 * K
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        self.i = self.steps = 0
        self.on = False

    def __call__(self, tag):
        if not self.on:
            return
        if self.fused:
            if tag == 0 and self.steps == 0:
                self.ev[0].record()          # the block's first launch interval begins here
            elif tag == 2:
                self.steps += 1
                if self.steps == self.K:
                    self.ev[1].record()
        else:
            self.ev[self.i].record()
            self.i += 1

    def arm(self):
        self.i, self.steps, self.on = 0, 0, True

    def collect(self):
        self.on = False
        K, ev = self.K, self.ev
        if self.fused:
            assert self.steps == K
            span = ev[0].elapsed_time(ev[1]) / K
            return {"step_render_interval_ms": span, "gpu_span_ms_per_step": span}
        assert self.i == 3 * K
        between = [ev[3 * j + 2].elapsed_time(ev[3 * j + 3]) for j in range(K - 1)]   # gap between steps
        return {"step_interval_ms": sum(ev[3 * j].elapsed_time(ev[3 * j + 1]) for j in range(K)) / K,
                "render_interval_ms": sum(ev[3 * j + 1].elapsed_time(ev[3 * j + 2]) for j in range(K)) / K,
                "between_steps_ms": sum(between) / max(1, K - 1), "gpu_span_ms_per_step": ev[0].elapsed_time(ev[3 * K - 1]) / K}


def robust(values, factor=3.0):
    """mean / median / min / max of a list, and the same mean without the values beyond `factor` x the median
    (listed as outliers: a host hiccup in one K-step block must not move the statistic, and must not vanish)"""
    med = statistics.median(values)
    kept = [v for v in values if v <= factor * med]
    return {"count": len(values), "mean": sum(values) / len(values), "median": med, "min": min(values),
            "max": max(values), "first": values[0], "last": values[-1],
            "mean_within_3x_median": sum(kept) / len(kept),
            "outliers": [{"block": i, "value": v} for i, v in enumerate(values) if v > factor * med]}


def summarise(blocks, K):
    """`plain` = the blocks the contract line is made of (all of them when every block carries its two events;
    the un-instrumented ones in the --unfused A/B form), `kernels` = the event intervals of the blocks that have
    them: MEDIAN block, with mean / min / max / outliers beside it."""
    plain = [b for b in blocks if not b["instrumented"]]
    inst = [b for b in blocks if "kernels" in b]
    ms = lambda bs: [b["elapsed_s"] / K * 1e3 for b in bs]      # noqa: E731
    out = {"plain": robust(ms(plain)), "seconds_timed": sum(b["elapsed_s"] for b in blocks),
           # the same blocks timed up to the end of the closing barrier (what round 3 reported as the block time)
           "with_barrier": robust([b.get("with_barrier_s", b["elapsed_s"]) / K * 1e3 for b in plain])}
    if inst:
        out["blocks_with_events"] = robust(ms(inst))
        ks = [b["kernels"] for b in inst]
        dom = "step_render_interval_ms" if "step_render_interval_ms" in ks[0] else "render_interval_ms"
        kern = {"dominant": dom}
        for key in ks[0]:
            r = robust([k[key] for k in ks])
            kern[key] = r["median"]
            kern[key + "_stats"] = r
        out["kernels"] = kern
        launches = sum(kern[key] for key in ("step_render_interval_ms", "step_interval_ms", "render_interval_ms",
                                             "between_steps_ms") if key in kern)
        out["closure"] = {
            "event_intervals_ms": launches,
            "vs_ms_per_step_of_the_same_blocks": launches / out["blocks_with_events"]["median"],
            "vs_ms_per_step": launches / out["plain"]["median"],
            "note": "median per-launch event interval (GPU span of a K-step block / K) / median host-timed ms per step "
                    "of the blocks that carry the events, and / that of the contract line's blocks"}
    return out


def build_env(wl, B, dev, seeds, fused=True, share=1):
    from marlgrid_amd.envs import make
    place = {"share": share} if share > 1 else True
    if wl == "Custom-8AgentCluttered30x30":     # BASELINE.json configs[4]; not a registered id upstream
        from marlgrid_amd.agents import GridAgentInterface
        from marlgrid_amd.envs import ClutteredMultiGrid
        cols = ["red", "blue", "purple", "orange", "olive", "pink", "cyan", "yellow"]
        return ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=9, view_tile_size=8) for c in cols],
                                  grid_size=30, clutter_density=0.15, batch_size=B, device=dev, seeds=seeds,
                                  auto_reset=True, fused_step=fused, place_obs=place)
    # everything else at its default — strict=True included (errors are polled, not synchronised on)
    return make(wl, batch_size=B, device=dev, seeds=seeds, auto_reset=True, fused_step=fused, place_obs=place)


def parity_ids(B):
    """the envs of a rank's batch that are replayed on the oracle after the timed region: both ends, a wave boundary,
    the middle (eight, fewer for tiny batches)"""
    return sorted({b for b in (0, 1, 63, 64, B // 2, B // 2 + 1, B - 2, B - 1) if 0 <= b < B})


def parity_snapshot(env, wl, seeds, pool, steps, last):
    """what the oracle replay needs, as host arrays: the sampled envs' seeds, their column of the action pool, and the
    state / last outputs the HIP path ended in (taken right after the timed region, before anything else touches the env)"""
    import numpy as np
    import torch
    ids = parity_ids(env.batch_size)
    ix = torch.as_tensor(ids, device=env.device)
    W, H = env.width, env.height
    return {"workload": wl, "ids": np.array(ids), "seeds": np.array([int(seeds[b]) for b in ids], dtype=np.int64),
            "pool": np.stack([a[ix].cpu().numpy() for a in pool]).astype(np.int32), "steps": int(steps),
            "grid": env.grid_state[ix][:, :W * H].reshape(len(ids), W, H).cpu().numpy(),
            "agents": env.agent_state[ix].cpu().numpy(), "step_count": env.step_count_t[ix].cpu().numpy(),
            "mt": env.mt_state[ix].cpu().numpy().view(np.uint32), "mt_pos": env.mt_pos[ix].cpu().numpy(),
            "obs": last[0][ix].cpu().numpy(), "rewards": last[1][ix].cpu().numpy(), "done": last[2][ix].cpu().numpy()}


def measure(wl, B, dev, ctl, seeds, K, Wm, min_seconds, max_blocks, action_seed, fused=True, share=1, snapshot=False):
    import torch
    env = build_env(wl, B, dev, seeds, fused, share)
    env.placement_retries = ensure_placed(env, share)
    env.reset()
    n = env.num_agents
    g = torch.Generator(device="cpu").manual_seed(action_seed)
    pool = [torch.randint(0, 7, (B, n), generator=g).to(dev) for _ in range(64)]
    probes = Probes(env, K)
    env._probe = probes
    last = [None]

    def step(i):            # global step i of the run takes pool[i % 64] (the warm-up and the timed blocks: one sequence)
        last[0] = env.step(pool[i % 64])
    for i in range(Wm):
        step(i)
    settle_interpreter()
    blocks = timed_blocks(lambda i: step(Wm + i), lambda: torch.cuda.synchronize(dev), ctl, K,
                          min_seconds, max_blocks, probes)
    env._probe = None
    env.check_errors()
    if snapshot:
        env.parity_snapshot = parity_snapshot(env, wl, seeds, pool, Wm + len(blocks) * K, last[0])
    return env, summarise(blocks, K), blocks


_PIPE_STREAMS = {}


def measure_pipeline(wl, B, parts, dev, ctl, K, Wm, min_seconds, max_blocks, action_seed, fused=True):
    """The same batch as `parts` envs on as many streams (marlgrid_amd.sharding.ShardPipeline: overlapping launches
    of independent shards), timed like the contract line: K-step blocks, barrier + synchronize on both sides."""
    import torch
    from marlgrid_amd.envs import make
    if parts not in _PIPE_STREAMS:      # the same streams for every pipeline of this process (hardware queues are few)
        _PIPE_STREAMS[parts] = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    # the public path: make(id, pipeline=P) and the sampler's own call, step_part(k, actions), part after part
    pipe = make(wl, pipeline=parts, batch_size=B, device=dev, seed=1337, auto_reset=True, fused_step=fused,
                streams=_PIPE_STREAMS[parts])
    retries = 0
    for k in range(parts):          # (a retry launches — the search's rasters, the re-render — on the part's own stream, like its steps)
        with pipe.on(k):
            retries += ensure_placed(pipe.envs[k])
    pipe.reset()
    n = pipe.envs[0].num_agents
    g = torch.Generator(device="cpu").manual_seed(action_seed)
    pool = [torch.randint(0, 7, (B, n), generator=g).to(dev) for _ in range(64)]
    pool = [[pipe.part(k, a).contiguous() for k in range(parts)] for a in pool]
    torch.cuda.synchronize(dev)
    def step(i):
        acts = pool[i % 64]
        for k in range(parts):
            pipe.step_part(k, acts[k])
    for i in range(Wm):
        step(i)
    settle_interpreter()
    blocks = timed_blocks(lambda i: step(Wm + i), lambda: torch.cuda.synchronize(dev), ctl, K, min_seconds, max_blocks, None)
    pipe.check_errors()
    placement = [{k: v for k, v in (getattr(e._groups[0], "placement_ms", None) or {}).items() if k != "all"} for e in pipe.envs]
    if placement:
        placement[0]["placement_retries"] = retries
    return n, summarise(blocks, K), placement


def raster_only_ms(env, iters=50):
    """the obs raster alone (mg_render_obs launched back to back on the launch stream, HIP events): what the
    fused launch's raster part costs without the step in front of it"""
    import ctypes as C
    from marlgrid_amd import _native as N
    ms = C.c_float(0)
    N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), env.obs.data_ptr(), iters, C.byref(ms),
                                        env._stream()))
    return ms.value


def roofline_of(env, B, summary, traffic, raster_ms=None):
    vs, ts, n = env.view_size, env.tile_size, env.num_agents
    P = vs * ts
    alg = P * P * 3 + vs * vs + 8 * n                    # bytes per agent-step
    k = summary.get("kernels")
    if not k:
        return None
    dom = k["dominant"]
    fused = dom == "step_render_interval_ms"
    ms, st = k[dom], k[dom + "_stats"]
    ach = B * n * alg / (ms * 1e-3) / 1e9
    kname = getattr(env, "kernel_name", None) or "mg::render_kernel (name not recorded)"      # the launcher's own (mg_render_kernel_name)
    return {"bound": "hbm", "kernel": kname + (" launched by mg_step_render (the env step fused in front of the raster)"
                                                  if fused else ""),
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "kernel_ms": ms, "kernel_ms_mean": st["mean"], "kernel_ms_mean_within_3x_median": st["mean_within_3x_median"],
            "kernel_ms_min": st["min"], "kernel_ms_max": st["max"], "kernel_ms_first": st["first"],
            "kernel_ms_last": st["last"], "kernel_ms_outliers": st["outliers"],
            "kernel_ms_source": "HIP events on the launch stream of the instrumented K-step blocks (event span / K: "
                                "launch + dispatch gap); the MEDIAN block (mean / min / max / outliers beside it)",
            "raster_only_ms": raster_ms,
            "raster_only_GBps": (B * n * alg / (raster_ms * 1e-3) / 1e9) if raster_ms else None,
            "algorithmic_bytes_per_agent_step": alg, "algorithmic_bytes_per_launch": B * n * alg,
            "traffic": traffic.get("step_render" if fused else "render", {}).get("hbm_bytes_per_launch") if traffic else None,
            "traffic_unit": "bytes per launch (PMC, this run: WRITE_SIZE + FETCH_SIZE, calibrated; see pmc)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-per-gpu", type=int, default=32768)
    ap.add_argument("--min-seconds", type=float, default=10.0, help="keep timing K-step blocks until this much is timed "
                    "(10 s of back-to-back blocks: long enough for clocks and any utilisation sampler to see the card busy)")
    ap.add_argument("--max-blocks", type=int, default=6000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--no-strong", action="store_true", help="skip the 262 144-env single-GPU point")
    ap.add_argument("--no-pipeline", action="store_true", help="skip extra.pipelined_shards (two envs on two streams)")
    ap.add_argument("--no-encode-leg", action="store_true", help="skip extra.step_with_encode (the step with MultiGrid.encode from its own launch)")
    ap.add_argument("--unfused", action="store_true",
                    help="env.step() as two launches (mg_step, mg_render_obs) instead of one (mg_step_render): A/B only")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="ranks map to local_rank %% device_count and the control plane runs on gloo: the N > 1 code "
                         "path on a box with fewer GPUs than ranks (plumbing check, not a scaling number)")
    ap.add_argument("--control-plane", choices=("auto", "nccl", "gloo"), default="auto",
                    help="barriers over RCCL (one rank per GPU: the default) or gloo (--oversubscribe's default); either "
                         "way the gathers run over gloo, and a RCCL group that does not come up on every rank falls "
                         "back to gloo (timing.control_plane.fallback says why)")
    ap.add_argument("--no-parity", action="store_true", help="skip parity_after_timed (the oracle replay of eight envs of the timed run)")
    ap.add_argument("--parity-replay", default=None, help=argparse.SUPPRESS)               # (parity_after_timed's child)
    ap.add_argument("--no-pin", action="store_true", help="do not pin the rank to the CPUs of its GPU's NUMA node")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="run the distributed measurement skeleton with a sleep() in place of the engine (gloo, no GPU)")
    ap.add_argument("--workload", default=WORKLOAD, help="exploration only; the contract line uses the default")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)   # (the cpu_baseline leg's child)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        return cpu_baseline_child(args.cpu_seconds, args.workload)
    if args.parity_replay:
        return parity_replay_child(args.parity_replay)

    bad = sorted(k for k in os.environ if k.startswith("MG_") or k.startswith("MARLGRID_")
                 or (k.startswith("BENCH_TEST_") and not args.selftest_cpu))      # (test hooks: the CPU skeleton only)
    if bad:
        print("bench.py: refusing to run with %s set (measurement switches must not touch the contract line)"
              % ", ".join(bad), file=sys.stderr)
        sys.exit(2)

    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))          # --gpus N alone starts its own N ranks
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    K, Wm = args.steps, args.warmup
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but the launcher started WORLD_SIZE %d ranks: refusing to print a line whose "
                  "n_gpus would not be what was asked for" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    if args.selftest_cpu:
        return selftest_cpu(args, rank, local_rank, world, K, Wm)

    import torch
    from marlgrid_amd import _native as N
    from marlgrid_amd import sharding
    import smi

    ndev = torch.cuda.device_count()
    if ndev < 1 or (world > ndev and not args.oversubscribe):
        print("bench.py: %d ranks but %d visible GPU(s): one rank per GPU is the contract (--oversubscribe maps "
              "ranks onto fewer GPUs for a plumbing check)" % (world, ndev), file=sys.stderr)
        sys.exit(3)
    shared = args.oversubscribe and world > ndev
    share = (world + ndev - 1) // ndev if shared else 1          # ranks per GPU: each counts on its part of the free memory
    dev_index = local_rank % ndev if args.oversubscribe else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    affinity, cpus_all = ({"pinned": False, "why": "--no-pin"}, None) if args.no_pin else pin_to_gpu_numa(dev_index)
    prefer = args.control_plane if args.control_plane != "auto" else ("gloo" if args.oversubscribe else "nccl")
    ctl = Control(prefer, dev)
    n_gpus = world
    B = args.batch_per_gpu
    wl = args.workload
    build_info = N.lib().mg_build_info().decode()

    # the env batch shards embarrassingly: rank r owns global envs [r*B, (r+1)*B), seeds 1337 + id
    seeds = sharding.shard_seeds(1337, B * n_gpus, rank, n_gpus)
    assert len(seeds) == B
    clocks_before = smi.sample(dev_index) if rank == 0 else None
    fused = not args.unfused
    env, summary, blocks = measure(wl, B, dev, ctl, seeds, K, Wm, args.min_seconds, args.max_blocks, rank, fused, share,
                                   snapshot=not args.no_parity)
    clocks_after = smi.sample(dev_index) if rank == 0 else None
    # every rank replays eight envs of ITS shard on the CPU oracle (a child process each, a few seconds, GPUs idle)
    parity = None if args.no_parity else parity_after_timed(env.parity_snapshot)
    parity_by_rank = ctl.gather_objects(parity)
    raster_ms = raster_only_ms(env) if rank == 0 else None
    # who ran what: every rank's device and its own K-step times (a straggler, or two ranks on one GPU, shows)
    own = robust([b["per_rank_s"][rank] / K * 1e3 for b in blocks if not b["instrumented"]])
    place = getattr(env._groups[0], "placement_ms", None) or {}
    ranks_info = ctl.gather_objects({"rank": rank, "local_rank": local_rank, "device_index": dev_index,
                                     "device": device_identity(dev_index), "affinity": affinity,
                                     # the slice of the global env batch this rank stepped: [lo, hi), seeds 1337 + id
                                     "global_envs": [seeds[0] - 1337, seeds[-1] - 1337 + 1], "ranks_on_this_gpu": share,
                                     "ms_per_step_own": own["mean"], "ms_per_step_own_median": own["median"],
                                     # which observation buffers this rank drew (ms per raster launch into each kept
                                     # buffer): the MAX over ranks is the unluckiest rank's
                                     "obs_placement": {k: place.get(k) for k in ("found", "reused", "kept", "candidates", "stopped", "seconds",
                                                                                 "pinned_bytes", "budget_bytes", "share", "first_attempt")},
                                     "placement_retries": getattr(env, "placement_retries", 0)})
    n, vs, ts = env.num_agents, env.view_size, env.tile_size
    P = vs * ts
    kname0 = env.kernel_name

    out = None
    if rank == 0:
        pl = summary["plain"]
        ms = pl["mean"]                               # sustained: total time / total steps of the plain blocks
        out = {
            "metric": METRIC,
            "value": n_gpus * B * n / (ms * 1e-3),
            "unit": "agent-steps/s",
            "n_gpus": n_gpus, "steps": K, "warmup": Wm,
            "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "value_median_block": n_gpus * B * n / (pl["median"] * 1e-3),
            "ms_per_step_median_block": pl["median"],
            "config": {"workload": wl, "batch_per_gpu": B, "global_batch": B * n_gpus, "n_agents": n,
                       "view_size": vs, "tile_size": ts, "obs_shape": [B * n_gpus, n, P, P, 3],
                       "actions": "uniform over 7 ids, torch.randint seed=rank", "auto_reset": True,
                       "strict": repr(env.strict),
                       "launches_per_step": (["mg_step_render (action loop + reset of finished episodes + obs raster)"]
                                             if fused else ["mg_step (+ fused reset of finished episodes)",
                                                            "mg_render_obs"]),
                       "sharding": "env batch split contiguously, no collectives",
                       "obs_buffers": OBS_BUFFERS_NOTE.get(env.place_obs, str(env.place_obs))},
            "timing": {"what": "K-step blocks, each bracketed by barrier + synchronize; a block's time = MAX over ranks of "
                               "the rank's own K steps (opening barrier -> its own synchronize; `with_barrier` = up to the "
                               "end of the closing barrier); value = all plain blocks' steps / their total time (every "
                               "100th step resets the whole batch in-launch: the median block hides those)",
                       "blocks": pl, "with_barrier": summary["with_barrier"],
                       "value_with_barrier": n_gpus * B * n / (summary["with_barrier"]["mean"] * 1e-3),
                       "seconds_timed": summary["seconds_timed"],
                       "blocks_with_events": summary.get("blocks_with_events"),
                       "per_rank": ranks_info,
                       "per_rank_ms_per_step_last_block": [x / K * 1e3 for x in blocks[-1]["per_rank_s"]],
                       "control_plane": ctl.describe(), "ranks_share_a_gpu": bool(shared),
                       "launched_by": os.environ.get("BENCH_SELF_LAUNCHED", "external launcher" if world > 1 else "direct")},
            "kernels": summary.get("kernels"),
            "closure": summary.get("closure"),
            "clocks": {"before": clocks_before, "after": clocks_after, "source": "rocm-smi"},
            "library": build_info,
            "obs_placement": getattr(env._groups[0], "placement_ms", None),
            # every rank's buffers, at the top of the line: a rank off the fast class shows before anyone reads a scaling curve
            "obs_placement_found_by_rank": [r["obs_placement"].get("found") for r in ranks_info],
            "placement_retries_by_rank": [r["placement_retries"] for r in ranks_info],
        }
        if not args.no_parity:
            p0 = parity_by_rank[0] or {}
            out["parity_after_timed"] = dict(p0, ok=all(bool(p and p.get("ok")) for p in parity_by_rank),
                                             ok_by_rank=[bool(p and p.get("ok")) for p in parity_by_rank],
                                             errors_by_rank=[(p or {}).get("error") for p in parity_by_rank]
                                             if any((p or {}).get("error") for p in parity_by_rank) else None)
    del env
    torch.cuda.empty_cache()

    # the dominant kernel's HBM traffic from the PMC counters, measured now (child processes under
    # rocprofv3 on this GPU; N = 1 only: the counters are per process and the workload is per GPU)
    # (every leg from here on is optional: whatever goes wrong in one of them is recorded under `errors` and the
    # contract line is printed all the same)
    errors = {}

    def leg(name, fn):
        try:
            return fn()
        except Exception as e:      # noqa: BLE001 — an optional leg must not cost the line
            import traceback
            errors[name] = "%s: %s | %s" % (type(e).__name__, e, traceback.format_exc().strip().splitlines()[-3:])
            try:
                torch.cuda.empty_cache()
            except Exception:       # noqa: BLE001
                pass
            return None

    traffic = None
    if rank == 0 and n_gpus > 1 and not args.no_pmc:
        traffic = {"skipped": "N > 1: the counters are per process and the other ranks' processes hold their GPUs; the "
                              "per-GPU workload is the N = 1 line's, whose traffic is measured there"}
    if rank == 0 and n_gpus == 1 and not args.no_pmc:
        def _pmc():
            import pmc
            return pmc.measure(wl, B)
        traffic = leg("pmc", _pmc)
    if rank == 0:
        # roofline needs the env's geometry only
        class _Geo(object):
            view_size, tile_size, num_agents, kernel_name = vs, ts, n, kname0
        out["roofline"] = roofline_of(_Geo, B, summary, traffic if n_gpus == 1 else None, raster_ms)
        out["pmc"] = traffic

    # strong-scaling N = 1 point: BASELINE.json's whole headline batch on one GPU
    def _strong():
        Bs = 262144
        seeds_s = sharding.shard_seeds(1337, Bs, 0, 1)
        env_s, sum_s, _ = measure(wl, Bs, dev, ctl, seeds_s, K, min(Wm, 5), min(args.min_seconds, 1.0), 400, 0, fused)
        raster_s = raster_only_ms(env_s, 10)
        ms_s = sum_s["plain"]["mean"]
        res = {
            "what": "the full headline batch (262 144 envs) on ONE GPU: the N = 1 point of a strong-scaling curve",
            "value": Bs * n / (ms_s * 1e-3), "unit": "agent-steps/s", "ms_per_step": ms_s,
            "global_batch": Bs, "timing": sum_s["plain"], "kernels": sum_s.get("kernels"),
            "closure": sum_s.get("closure"), "roofline": roofline_of(env_s, Bs, sum_s, None, raster_s)}
        del env_s
        torch.cuda.empty_cache()
        return res

    if n_gpus == 1 and not args.no_strong and wl == WORKLOAD:
        res = leg("strong_n1", _strong)
        if res is not None:
            out.setdefault("extra", {})["strong_n1"] = res

    # overlapping launches — the same batch as two envs on two streams, and twice the batch
    def _pipeline():
        pts = []
        for Bp in (B, 2 * B):
            ms_one = ms
            if Bp != B:
                seeds_o = sharding.shard_seeds(1337, Bp, 0, 1)
                env_o, sum_o, _ = measure(wl, Bp, dev, ctl, seeds_o, K, min(Wm, 5), min(args.min_seconds, 1.0), 400, 0, fused)
                ms_one = sum_o["plain"]["mean"]
                del env_o
                torch.cuda.empty_cache()
            n_p, sum_p, place_p = measure_pipeline(wl, Bp, 2, dev, ctl, K, min(Wm, 5), min(args.min_seconds, 1.0), 400, 0, fused)
            ms_p = sum_p["plain"]["mean"]
            pts.append({"envs": Bp, "parts": 2, "envs_per_part": Bp // 2, "ms_per_step": ms_p,
                        "value": Bp * n_p / (ms_p * 1e-3), "unit": "agent-steps/s", "timing": sum_p["plain"],
                        "one_env_ms_per_step": ms_one, "one_env_value": Bp * n_p / (ms_one * 1e-3),
                        "vs_one_env": ms_one / ms_p, "obs_placement": place_p})
            torch.cuda.empty_cache()
        return pts

    pts = leg("pipelined_shards", _pipeline) if (n_gpus == 1 and not args.no_pipeline and wl == WORKLOAD) else None
    if pts:
        out.setdefault("extra", {})["pipelined_shards"] = {
            "what": "the per-GPU batch as TWO envs on two streams — make(id, pipeline=2), stepped part after part with "
                    "step_part(k, actions) as a double-buffered sampler does, no join in between: launches of independent shards overlap (the store-free head of one under the "
                    "stores of the other); same trajectories env by env; the contract line above is ONE env, one stream",
            "points": pts}
        out["value_pipelined_shards"] = pts[0]["value"]       # the same batch as two envs on two streams (extra.pipelined_shards)

    # MultiGrid.encode of the stepped batch from the step's own launch (mg_step_render_encode; `encode_in_step=True`): what it adds
    # to a step, against a second launch (VERDICT r05 item 3's "or inside the launch": asked <= +3 %)
    def _with_encode():
        import statistics
        from marlgrid_amd.envs import make as mk
        seeds_e = sharding.shard_seeds(1337, B, 0, 1)
        env_e = mk(wl, batch_size=B, seeds=seeds_e, auto_reset=True, strict=False)
        env_e.reset()
        g = torch.Generator().manual_seed(11)
        acts = [torch.randint(0, 7, (B, n), generator=g).to(dev) for _ in range(16)]
        env_e.encode_in_step = True
        for i in range(30):                      # identity first: the launch's encoding == mg_encode behind the step
            env_e.step(acts[i % 16])
            if not torch.equal(env_e.grid_encoding, env_e.grid.encode()):
                raise RuntimeError("mg_step_render_encode differs from mg_encode at step %d" % i)
        in_launch = bool(env_e._enc_fused) and env_e.fused_step and not env_e._hetero
        enc2 = torch.empty_like(env_e.grid_encoding)
        t = {"step": [], "step_with_encode": [], "step_then_mg_encode": []}
        for rnd in range(7):
            for mode in t:
                env_e.encode_in_step = mode == "step_with_encode"
                for i in range(10):
                    env_e.step(acts[i % 16])
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(100):
                    env_e.step(acts[i % 16])
                    if mode == "step_then_mg_encode":
                        env_e._encode_into(enc2)
                e1.record()
                torch.cuda.synchronize()
                t[mode].append(e0.elapsed_time(e1) / 100)
        env_e.check_errors()
        med = {k: statistics.median(v) for k, v in t.items()}
        res = {"what": "one env.step() of the contract workload with MultiGrid.encode of the batch (base.py:196-214) written by the "
                       "step's own launch (encode_in_step=True -> mg_step_render_encode), against the step alone and against a second "
                       "launch (mg_encode) behind it; interleaved rounds of 100 steps, medians of 7; identity with mg_encode checked "
                       "over 30 steps first",
               "ms": med, "encoding_written_by_the_step_launch": in_launch,
               "step_with_encode_vs_step_pct": 100 * (med["step_with_encode"] / med["step"] - 1),
               "step_then_mg_encode_vs_step_pct": 100 * (med["step_then_mg_encode"] / med["step"] - 1),
               "encoding_bytes": int(enc2.numel()), "observation_bytes": int(env_e.obs.numel())}
        del env_e
        torch.cuda.empty_cache()
        return res

    if n_gpus == 1 and not args.no_encode_leg and wl == WORKLOAD:
        res = leg("step_with_encode", _with_encode)
        if res is not None:
            out.setdefault("extra", {})["step_with_encode"] = res

    if rank == 0:
        if not args.no_cpu_baseline:
            # rank 0, once, for every N (the other ranks wait in the closing barrier and leave the host cores alone);
            # a child process with all the CPUs this process was allowed before it pinned itself
            if cpus_all:
                try:
                    os.sched_setaffinity(0, cpus_all)
                except OSError:
                    pass
            out["cpu_baseline"] = leg("cpu_baseline", lambda: cpu_baseline(args.cpu_seconds, wl))
        if errors:
            out["errors"] = errors
        print(json.dumps(out), flush=True)
    ctl.close()


def selftest_cpu(args, rank, local_rank, world, K, Wm):
    """The distributed measurement skeleton with a sleep() in place of the engine (gloo, no GPU): process-group
    bring-up incl. the RCCL attempt and its fallback (--control-plane auto / nccl: on a box without GPUs RCCL cannot
    come up, which is exactly the failure the fallback is for), the barrier / synchronize bracket, the per-rank
    gather, MAX over ranks of each rank's own time, and ONE JSON line with every field the N > 1 contract line
    carries for any N — roofline (traffic null), cpu_baseline (rank 0, unless --no-cpu-baseline), per-rank affinity
    and observation-buffer placement."""
    prefer = args.control_plane if args.control_plane != "auto" else "gloo"
    ctl = Control(prefer, None)
    affinity = {"pinned": False, "why": "selftest: no GPU"}
    blocks = timed_blocks(lambda i: time.sleep(0.001 * (1 + rank)), lambda: None, ctl, K, args.min_seconds,
                          args.max_blocks)
    s = summarise(blocks, K)
    own = robust([b["per_rank_s"][rank] / K * 1e3 for b in blocks])
    ranks_info = ctl.gather_objects({"rank": rank, "local_rank": local_rank, "affinity": affinity,
                                     "ms_per_step_own": own["mean"],
                                     "obs_placement": {"kept": None, "candidates": 0, "stopped": "selftest", "seconds": 0.0}})
    if rank == 0:
        out = {"metric": "selftest (sleep in place of the engine)", "value": None, "n_gpus": world,
               "steps": K, "warmup": Wm, "ms_per_step": s["plain"]["mean"], "data": "selftest",
               "blocks": s["plain"], "with_barrier": s["with_barrier"],
               "launched_by": os.environ.get("BENCH_SELF_LAUNCHED", "external launcher"),
               "per_rank_ms_per_step": [x / K * 1e3 for x in blocks[-1]["per_rank_s"]],
               "timing": {"per_rank": ranks_info, "control_plane": ctl.describe()},
               "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                            "traffic": None},
               "pmc": {"skipped": "selftest"}}
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(min(args.cpu_seconds, 1.0))
            except Exception as e:      # noqa: BLE001
                out["errors"] = {"cpu_baseline": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out), flush=True)
    ctl.close()


def parity_after_timed(snap):
    """Replay the sampled envs of the timed run on the CPU oracle and compare where they ended (see the module docstring).
    The oracle is the CHECKER here — loaded in a child process only, after the timed region, never on the measured path.
    Returns {"envs", "steps", "ok", ...}; never raises (a failure of the leg itself is reported as ok = False + error)."""
    import subprocess
    import tempfile
    import numpy as np
    path = None
    try:
        fd, path = tempfile.mkstemp(suffix=".npz", prefix="mg_parity_")
        os.close(fd)
        np.savez(path, **snap)
        env = {k: v for k, v in os.environ.items()
               if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "BENCH_SELF_LAUNCHED")}
        env["OMP_NUM_THREADS"] = "1"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--parity-replay", path], env=env, capture_output=True,
                           text=True, timeout=1200)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"envs": len(snap["ids"]), "steps": snap["steps"], "ok": False,
                    "error": "replay child failed (rc %d): %s" % (r.returncode, r.stderr[-400:])}
        return json.loads(lines[-1])
    except Exception as e:      # noqa: BLE001
        return {"envs": len(snap["ids"]), "steps": snap["steps"], "ok": False, "error": "%s: %s" % (type(e).__name__, e)}
    finally:
        if path and os.path.exists(path):
            os.unlink(path)


def parity_replay_child(path):
    """the oracle side of parity_after_timed: same seeds, same actions (global step s takes pool[s % 64]), reset on `done`"""
    import numpy as np
    import canon
    import product_envs
    import scenarios
    from marlgrid_amd import seeding
    from oracle import oracle as O
    z = np.load(path)
    wl, ids, T = str(z["workload"]), z["ids"], int(z["steps"])
    spec = scenarios.registered(wl)
    t0 = time.perf_counter()
    orc = O.OracleBatch(spec, z["seeds"])
    orc.reset()
    pool = z["pool"]
    episodes = np.zeros(len(ids), np.int64)
    obs = rew = done = None
    for s in range(T):
        obs, rew, done, _ = orc.step(pool[s % 64], render=(s == T - 1), auto_reset=True, threads=1)
        episodes += done
    bad = []
    st = product_envs.canonical_arrays(spec, z["grid"], z["agents"], z["step_count"])
    for j, b in enumerate(ids):
        try:
            canon.assert_same(st[j], canon.oracle_canonical(orc.envs[j]), "env %d" % b)
        except AssertionError as e:
            bad.append("state: " + str(e)[:200])
        if not seeding.same_stream(seeding.numpy_form(z["mt"][j], z["mt_pos"][j], 16), orc.envs[j].mt_state()):
            bad.append("env %d: MT19937 state differs" % b)
    if T > 0 and "obs" in z.files:
        if not np.array_equal(z["obs"], obs):
            bad.append("last observations differ in envs %s" % [int(ids[j]) for j in range(len(ids)) if not np.array_equal(z["obs"][j], obs[j])])
        if np.abs(z["rewards"].astype(np.float64) - rew).max() > 1e-6:
            bad.append("last rewards differ")
        if not np.array_equal(z["done"].astype(bool), done):
            bad.append("last done flags differ")
    print(json.dumps({"envs": int(len(ids)), "env_ids": [int(b) for b in ids], "steps": T, "ok": not bad,
                      "episodes_per_env_min": int(episodes.min()), "compared": ["grid", "agent records + stack order", "step_count",
                      "MT19937 state (numpy form)", "last rewards (1e-6)", "last done", "last observations (bit-exact)"],
                      "oracle_seconds": time.perf_counter() - t0, "mismatches": bad[:8] or None}), flush=True)


def physical_cores(cpus):
    """how many physical cores the CPU set covers (sysfs thread_siblings_list; the set's size if sysfs is silent)"""
    groups = set()
    for c in cpus:
        try:
            groups.add(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip())
        except OSError:
            return len(cpus)
    return len(groups) or len(cpus)


def cpu_quota_cores():
    """the container's CPU bandwidth limit in cores (cgroup v2 cpu.max / v1 cfs quota), or None: a box may show 256 CPUs in
    its affinity mask and still be given the time of a few"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(budget_s, workload=WORKLOAD):
    """The parity-checked CPU oracle (a C port of the reference algorithm, OpenMP over envs) on a bounded sample of
    the same workload, on this box's host cores — in a CHILD process started with a clean environment: the bench
    process itself is pinned to its GPU's NUMA node, carries the HIP runtime's threads and, under
    torch.distributed.run, OMP_NUM_THREADS=1; none of that may shape the baseline (visit 1 of round 4: 256 OpenMP
    threads — every SMT sibling — inside the bench process ran the same sample at 47 k agent-steps/s, 40 x below
    the 128-thread number).  One thread per physical core the process may run on.  The Python reference cannot
    travel to the GPU box; its own measured speed (2 377 agent-steps/s on one Xeon core) is in BASELINE.md."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if not (k.startswith("OMP_") or k.startswith("GOMP_") or k.startswith("MKL_")
                   or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "BENCH_SELF_LAUNCHED"))}
    env.update(OMP_PROC_BIND="spread", OMP_PLACES="cores")      # a thread stays with the envs (and the obs pages) it touched first
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-seconds", str(budget_s),
                        "--workload", workload], env=env, capture_output=True, text=True, timeout=budget_s * 4 + 120)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("cpu baseline child failed (rc %d): %s" % (r.returncode, r.stderr[-400:]))
    return json.loads(lines[-1])


def cpu_baseline_child(budget_s, workload):
    """One thread first (64 envs), then one thread per physical core on batches of 64 and 256 envs PER THREAD: a thread's
    share of a step has to be long against the fork-join around it (round 4 fed 128 threads 16 envs each — 77 us of
    work per step and thread — and reported 13.5 k agent-steps/s per core where one core alone does 200 k)."""
    import numpy as np
    import scenarios
    from oracle import oracle as O
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = list(range(os.cpu_count() or 1))
    threads = max(1, physical_cores(cpus))
    spec = scenarios.registered(workload)

    def run(Bc, thr, seconds):
        t_build = time.perf_counter()
        orc = O.OracleBatch(spec, 1337 + np.arange(Bc))
        orc.reset()
        rng = np.random.RandomState(0)
        acts = [rng.randint(0, 7, size=(Bc, orc.n)) for _ in range(16)]
        orc.step(acts[0], auto_reset=True, reuse_obs=True, threads=thr)        # (first touch of the obs buffer, by the threads that own it)
        t_build = time.perf_counter() - t_build
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < seconds:
            orc.step(acts[steps % 16], auto_reset=True, reuse_obs=True, threads=thr)
            steps += 1
        dt = time.perf_counter() - t0
        v = Bc * orc.n * steps / dt
        return {"envs": Bc, "threads": thr, "steps": steps, "seconds": dt, "setup_seconds": t_build, "value": v,
                "per_thread": v / thr}

    share = max(1.0, budget_s / 6.0)
    pts = [run(64, 1, share)]
    quota = cpu_quota_cores()
    if threads > 1:
        pts.append(run(64 * threads, threads, 2 * share))
        if quota is not None and 1 < int(quota + 0.999) < threads:
            # the container is given the time of fewer cores than it may run on: as many threads as it has cores' time
            pts.append(run(64 * int(quota + 0.999), int(quota + 0.999), 2 * share))
        elif pts[-1]["setup_seconds"] < budget_s:                                # (building 256 envs per thread takes 4 x as long)
            pts.append(run(256 * threads, threads, 2 * share))
    best = max(pts[1:] or pts, key=lambda q: q["value"])        # (the box's cores: the one-thread point is reported beside it)
    print(json.dumps({
        "value": best["value"], "unit": "agent-steps/s", "cores": int(best["threads"]), "kind": "port",
        "per_thread": best["per_thread"], "one_thread": pts[0]["value"],
        "sample": "%d envs x %d steps of %s (C oracle, OpenMP over envs, obs render included), %.1f s on %d threads; "
                  "one thread alone: %d envs x %d steps, %.1f s" % (best["envs"], best["steps"], workload, best["seconds"],
                                                                    best["threads"], pts[0]["envs"], pts[0]["steps"], pts[0]["seconds"]),
        "points": pts, "host_cpus": os.cpu_count(), "cpus_allowed": len(cpus), "cpu_quota_cores": quota,
        "scaling_vs_one_thread": best["value"] / pts[0]["value"],
        "loadavg": (os.getloadavg() if hasattr(os, "getloadavg") else None),
        "omp": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES")},
        "cores_note": "cores = OpenMP threads = one per PHYSICAL core among the CPUs a clean child process may run on "
                      "(SMT siblings are not counted), bound to cores (OMP_PROC_BIND=spread, OMP_PLACES=cores); measured in a "
                      "child process with no other OMP_* / launcher variables; value = the better of the all-core points"}),
          flush=True)


if __name__ == "__main__":
    main()
