"""GPU: bench.py's N > 1 launch path on hardware — the driver's own launch line with two ranks on the one GPU of the
box (`--oversubscribe`): what a round can show of SURVEY section 8(e) without an 8-GPU node.  One RNG per env and nothing
shared (marlgrid/base.py:371-374): the ranks exchange nothing but barriers and their times."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_on_one_gpu_through_the_drivers_launch_line():
    env = {k: v for k, v in os.environ.items()
           if not (k.startswith("MG_") or k.startswith("MARLGRID_") or k.startswith("BENCH_TEST_")
                   or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"))}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--oversubscribe", "--control-plane", "nccl", "--min-seconds", "1", "--no-strong", "--no-pipeline", "--cpu-seconds", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["scaling"] == "weak"
    assert out["config"]["workload"] == "MarlGrid-3AgentCluttered15x15-v0" and out["config"]["global_batch"] == 65536
    t = out["timing"]
    assert t["ranks_share_a_gpu"] is True
    # the control plane: RCCL was asked for; two ranks on one device are refused ("Duplicate GPU detected") and the ranks
    # AGREE over gloo to run their barriers there — or, should RCCL ever accept it, the barriers run on it.  Either way
    # the line says which.
    cp = t["control_plane"]
    assert cp["gathers"] == "gloo" and cp["test_hooks"] is None
    if cp["barriers"] == "gloo":
        assert cp["fallback"]["asked_for"] == "nccl" and cp["fallback"]["errors_by_rank"]
    else:
        assert cp["barriers"] == "nccl" and cp["fallback"] is None
    # a block's time is the slowest rank's OWN K steps; the interval that also holds the closing barrier rides along
    assert out["value"] >= t["value_with_barrier"] > 0
    assert abs(out["value"] - 2 * 32768 * 3 / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
    ranks = t["per_rank"]
    assert [q["rank"] for q in ranks] == [0, 1] and all(q["device_index"] == 0 for q in ranks)
    for q in ranks:
        assert "pinned" in q["affinity"] and q["ms_per_step_own"] > 0
        assert q["obs_placement"]["kept"] and len(q["obs_placement"]["kept"]) == 2 and q["placement_retries"] in (0, 1)
    assert out["obs_placement_found_by_rank"] == [q["obs_placement"]["found"] for q in ranks]
    assert out["placement_retries_by_rank"] == [q["placement_retries"] for q in ranks]
    # every N carries the CPU baseline (rank 0, once) and the roofline object (traffic: N = 1 only)
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["one_thread"] > 0
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and 0 < rf["frac"] < 1 and rf["traffic"] is None
    par = out["parity_after_timed"]
    assert par["ok"] is True and par["ok_by_rank"] == [True, True] and par["steps"] >= 25
    assert [q["global_envs"] for q in ranks] == [[0, 32768], [32768, 65536]]
    assert "errors" not in out or set(out["errors"]) <= {"pmc"}, out.get("errors")


def test_bench_eight_ranks_rehearsal_on_one_gpu():
    """The 8-rank launch the driver's SCALE run will make, rehearsed on the one GPU (`--gpus 8 --oversubscribe`, the
    driver's own torch.distributed.run line): no 1 -> 8 curve can be measured on this pool, so what the curve will depend
    on is exercised instead — eight ranks come up, agree on a control plane, shard the global env batch into eight
    distinct contiguous ranges, each places its observation buffers inside ITS share of the device's free memory (the
    default budget is worked out from free / ranks-on-this-GPU: eight searches of a quarter of what is free each would be
    twice the device), each replays eight of its envs on the oracle after the timed region, and rank 0 prints ONE line."""
    env = {k: v for k, v in os.environ.items()
           if not (k.startswith("MG_") or k.startswith("MARLGRID_") or k.startswith("BENCH_TEST_")
                   or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"))}
    Bp = 12288                                          # 347 MB of observations per buffer: above the placement threshold
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5",
           "--oversubscribe", "--batch-per-gpu", str(Bp), "--min-seconds", "0.5", "--no-strong", "--no-pipeline", "--cpu-seconds", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 8 * Bp and out["scaling"] == "weak"
    t = out["timing"]
    ranks = t["per_rank"]
    assert [q["rank"] for q in ranks] == list(range(8)) and t["ranks_share_a_gpu"] is True
    assert [q["global_envs"] for q in ranks] == [[k * Bp, (k + 1) * Bp] for k in range(8)]       # distinct, contiguous, complete
    pinned = 0
    for q in ranks:
        assert q["ranks_on_this_gpu"] == 8 and q["ms_per_step_own"] > 0
        pl = q["obs_placement"]
        assert pl["kept"] and len(pl["kept"]) == 2 and pl["share"] == 8
        assert pl["budget_bytes"] <= (288 << 30) // 8 // 2                                          # its share, not the device
        pinned += pl["pinned_bytes"]
    assert pinned <= 64 << 30
    assert len(out["obs_placement_found_by_rank"]) == 8 and len(out["placement_retries_by_rank"]) == 8
    assert abs(out["value"] - 8 * Bp * 3 / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
    par = out["parity_after_timed"]
    assert par["ok"] is True and par["ok_by_rank"] == [True] * 8 and par["envs"] == 8 and par["steps"] >= 25
    assert out["cpu_baseline"]["value"] > 0 and out["roofline"]["traffic"] is None
    assert "errors" not in out or set(out["errors"]) <= {"pmc"}, out.get("errors")
