"""CPU, world_size 2, gloo: the N>1 path of bench.py that is not GPU work — shard arithmetic
(disjoint cover, seeds by global env id) and the max/sum-over-ranks reductions."""
import os
import socket
import sys

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from marlgrid_amd import sharding
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    G = 262144 + 3
    lo, hi = sharding.shard_range(G, rank, world)
    seeds = sharding.shard_seeds(1337, G, rank, world)
    t = sharding.max_over_ranks(1.0 + rank)              # slowest rank wins
    total = sharding.sum_over_ranks(hi - lo)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, lo, hi, seeds[0], seeds[-1], t, total))


def test_two_rank_sharding_and_reduction():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    G = 262144 + 3
    (r0, lo0, hi0, s0a, s0b, t0, tot0), (r1, lo1, hi1, s1a, s1b, t1, tot1) = out
    assert lo0 == 0 and hi0 == lo1 and hi1 == G                # disjoint, contiguous cover
    assert abs((hi0 - lo0) - (hi1 - lo1)) <= 1
    assert (s0a, s0b, s1a, s1b) == (1337, 1337 + hi0 - 1, 1337 + lo1, 1337 + G - 1)
    assert t0 == t1 == 2.0                                     # MAX over ranks
    assert tot0 == tot1 == G


def test_shard_range_properties():
    from marlgrid_amd.sharding import shard_range
    for G in (1, 7, 8, 262144, 1048576 + 5):
        for W in (1, 2, 4, 8):
            prev = 0
            for r in range(W):
                lo, hi = shard_range(G, r, W)
                assert lo == prev and hi >= lo
                prev = hi
            assert prev == G
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_bench_skeleton_two_ranks_gloo():
    """bench.py's own N > 1 skeleton — process-group init, the barrier / synchronize bracket around every
    K-step block, the per-rank gather, MAX over ranks, rank 0's single JSON line — launched exactly as the
    driver launches it (torch.distributed.run, 2 ranks), with a sleep in place of the engine: rank 1
    sleeps twice as long, so the slowest rank must define ms_per_step and show up as the straggler."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not (k.startswith("MG_") or k.startswith("MARLGRID_"))}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "5", "--warmup", "1", "--selftest-cpu", "--min-seconds", "0.05"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["data"] == "selftest"
    per_rank = out["per_rank_ms_per_step"]
    assert len(per_rank) == 2 and per_rank[1] > per_rank[0] >= 1.0
    assert out["ms_per_step"] >= 2.0 and out["blocks"]["count"] >= 4
    # a block's time is the slowest rank's OWN K steps; the interval that also holds the closing barrier rides along
    assert out["with_barrier"]["mean"] >= out["ms_per_step"]
    # what every N > 1 line carries: cpu_baseline (rank 0, once), roofline (traffic may be null),
    # per-rank affinity and observation-buffer placement, and what the control plane runs on
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] == "port"
    assert out["cpu_baseline"]["cores"] >= 1 and out["cpu_baseline"]["cpus_allowed"] == len(os.sched_getaffinity(0))
    if len(os.sched_getaffinity(0)) >= 4:
        assert out["cpu_baseline"]["cores"] > 1                                 # not torchrun's OMP_NUM_THREADS=1
    assert out["roofline"]["bound"] == "hbm" and "traffic" in out["roofline"]
    ranks = out["timing"]["per_rank"]
    assert [r["rank"] for r in ranks] == [0, 1]
    assert all("affinity" in r and "obs_placement" in r and "kept" in r["obs_placement"] for r in ranks)
    assert out["timing"]["control_plane"] == {"barriers": "gloo", "gathers": "gloo", "fallback": None, "test_hooks": None}
    assert out["cpu_baseline"]["one_thread"] > 0 and out["cpu_baseline"]["per_thread"] > 0 and len(out["cpu_baseline"]["points"]) >= 1


@pytest.mark.parametrize("fail", ["natural", "1"])
def test_bench_rccl_failure_falls_back_to_gloo(fail):
    """The RCCL group of the control plane does not come up — here for real (`--control-plane nccl` on a box without
    GPUs: ProcessGroupNCCL refuses), and forced on rank 1 only (BENCH_TEST_FAIL_NCCL: the ranks must AGREE on the
    fallback over gloo, or the one whose RCCL attempt got further would wait for the other forever): still one JSON
    line, rc 0, and the line says what happened."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not (k.startswith("MG_") or k.startswith("MARLGRID_"))}
    if fail != "natural":
        env["BENCH_TEST_FAIL_NCCL"] = fail
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "5", "--warmup", "1", "--selftest-cpu", "--min-seconds", "0.05",
           "--control-plane", "nccl", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    cp = json.loads(lines[0])["timing"]["control_plane"]
    assert cp["barriers"] == "gloo" and cp["fallback"]["asked_for"] == "nccl"
    errs = cp["fallback"]["errors_by_rank"]
    assert errs and all(isinstance(v, str) and v for v in errs.values())
    if fail != "natural":
        assert "BENCH_TEST_FAIL_NCCL" in errs["1"] and cp["test_hooks"] == {"BENCH_TEST_FAIL_NCCL": fail}


def test_bench_refuses_test_hooks_outside_the_cpu_skeleton():
    """BENCH_TEST_* switches exist for the CPU skeleton's tests; a measuring run must not be shaped by one"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict({k: v for k, v in os.environ.items() if not (k.startswith("MG_") or k.startswith("MARLGRID_"))},
               BENCH_TEST_FAIL_NCCL="all")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=root)
    assert r.returncode == 2 and "BENCH_TEST_FAIL_NCCL" in r.stderr and not r.stdout.strip()


def test_timed_blocks_take_the_slowest_ranks_own_time():
    """timed_blocks with a control plane whose closing barrier is slow: the block time must not contain it"""
    import time
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class SlowBarrier(object):
        world = 1

        def barrier(self):
            time.sleep(0.02)

        def gather(self, values):
            return [[float(v) for v in values]]

    blocks = bench.timed_blocks(lambda i: time.sleep(0.001), lambda: None, SlowBarrier(), 5, 0.0, 4)
    for b in blocks:
        assert 0.005 <= b["elapsed_s"] < 0.02 <= b["with_barrier_s"] - 0.005
    s = bench.summarise(blocks, 5)
    assert s["with_barrier"]["mean"] > s["plain"]["mean"] + 3.0


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start 2 ranks itself (the driver's SCALE run
    may invoke it exactly so) — here through the CPU self-test, which takes the same self-launch path — and a
    launcher whose WORLD_SIZE disagrees with --gpus must be refused, not silently measured as something else."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if not (k.startswith("MG_") or k.startswith("MARLGRID_") or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"))}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                        "--selftest-cpu", "--min-seconds", "0.05"], capture_output=True, text=True, timeout=300,
                       env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and "bench.py --gpus 2" in out["launched_by"]
    assert len(out["per_rank_ms_per_step"]) == 2
    assert "cpu_baseline" in out and "roofline" in out and len(out["timing"]["per_rank"]) == 2
    # WORLD_SIZE 1 (as a launcher would set it) but --gpus 2: refused with rc 2
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-cpu"],
                       capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0"), cwd=root)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_bench_refuses_measurement_switches():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--selftest-cpu"], capture_output=True,
                       text=True, timeout=120, env=dict(os.environ, MG_RENDER_VARIANT="4"))
    assert r.returncode == 2 and "MG_RENDER_VARIANT" in r.stderr
