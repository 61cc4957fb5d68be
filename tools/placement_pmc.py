#!/usr/bin/env python
"""What separates a FAST observation buffer from a SLOW one, seen from the memory system's counters
(look from the kernel side — TCC / TCP / UTCL counters — not from the allocator side).

    python tools/placement_pmc.py [OUTDIR]            # driver: one rocprofv3 --pmc pass per counter group
    python tools/placement_pmc.py --inner N           # the profiled workload (run under rocprofv3 by the driver)
    python tools/placement_pmc.py --subranges         # plain process: classes of sub-ranges of ONE big allocation
    python tools/placement_pmc.py --scan              # plain process: the raster's speed as a function of the offset in one arena

Inner workload: the bench env (32 768 envs of MarlGrid-3AgentCluttered15x15-v0, place_obs=False), N raw hipMalloc
candidates of the obs buffer's size, all kept alive; the obs raster (mg_render_obs) is launched ITERS times into each,
candidate after candidate.  One process, so every candidate is a different allocation of the same box, and the
counters of a pass can be correlated with the launch duration ACROSS candidates (the classes differ by 20-25 %).
The launch duration comes from the dispatch's own start / end timestamps when the CSV has them, else from HIP events
(mg_time_render_obs; under --pmc they carry the counter read-out of every dispatch: a constant).
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ITERS = 3

GROUPS = [
    ("ea_write_stalls", ["TCC_EA0_WRREQ_STALL_sum", "TCC_TOO_MANY_EA_WRREQS_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum",
                         "TCC_EA0_WRREQ_LEVEL_sum"]),
    ("tcc_pipeline", ["TCC_TAG_STALL_sum", "TCC_IB_STALL_sum", "TCC_BUSY_sum", "TCC_WRREQ_STALL_max"]),
    ("ea_write_volume", ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum",
                         "TCC_EA0_WRREQ_IO_CREDIT_STALL_sum"]),
    ("utcl1_hit_miss", ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_REQUEST_sum",
                        "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"]),
    ("utcl1_stalls", ["TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum", "TCP_UTCL1_LFIFO_FULL_sum",
                      "TCP_UTCL1_STALL_INFLIGHT_MAX_sum", "TCP_UTCL1_SERIALIZATION_STALL_sum"]),
    ("tcp_write_path", ["TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum",
                        "TCP_TCR_TCP_STALL_CYCLES_sum"]),
    ("per_channel", ["TCC_EA0_WRREQ", "TCC_EA0_WRREQ_STALL"]),
    ("utcl2_grbm", ["GRBM_UTCL2_BUSY", "GRBM_EA_BUSY", "GRBM_TC_BUSY", "GRBM_GUI_ACTIVE"]),     # (timed out on visit 1: last)
]


def inner(n):
    import ctypes as C
    import torch
    from marlgrid_amd import _native as N
    from marlgrid_amd.base import _LibBuffer
    from marlgrid_amd.envs import make
    env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=32768, auto_reset=True, strict=False, place_obs=False)
    env.reset()
    L, ms, nbytes, dev = N.lib(), C.c_float(0), env.obs.numel(), env.device
    keep, rows = [], []
    for i in range(n):
        m = _LibBuffer(L, nbytes, dev)
        if not m.ok:
            break
        keep.append(m)
    torch.cuda.synchronize()
    for i, m in enumerate(keep):
        N.check(L.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), C.c_void_p(m.ptr), ITERS, C.byref(ms), env._stream()))
        rows.append({"cand": i, "ptr": m.ptr, "event_ms": ms.value})
    torch.cuda.synchronize()
    print("PLACEMENT_PMC " + json.dumps({"iters": ITERS, "rows": rows}), flush=True)


def subranges():
    """Is the class a property of the whole allocation, or of where in it the buffer lies?  A few allocations of three
    buffer sizes; the raster is timed into the first, second and third third of each (2 MiB-aligned offsets), and into
    plain candidates of one buffer size for the same box's class statistics."""
    import ctypes as C
    import torch
    from marlgrid_amd import _native as N
    from marlgrid_amd.base import _LibBuffer
    from marlgrid_amd.envs import make
    env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=32768, auto_reset=True, strict=False, place_obs=False)
    env.reset()
    L, ms, nbytes, dev = N.lib(), C.c_float(0), env.obs.numel(), env.device

    def raster(ptr, iters=4):
        N.check(L.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), C.c_void_p(ptr), iters, C.byref(ms), env._stream()))
        return ms.value

    step = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    print("plain candidates (one buffer each): ptr, trailing zero bits of ptr, ms")
    keep = []
    for i in range(16):
        m = _LibBuffer(L, nbytes, dev)
        keep.append(m)
        tz = (m.ptr & -m.ptr).bit_length() - 1
        print("  cand %2d  %#x  tz %2d  %.4f ms" % (i, m.ptr, tz, raster(m.ptr)))
    print("arenas of 3 buffer sizes: ms into each third (offsets 0, %d, %d), then third 0 again" % (step, 2 * step))
    for i in range(8):
        a = _LibBuffer(L, 3 * step, dev)
        keep.append(a)
        t = [raster(a.ptr + k * step) for k in (0, 1, 2, 0)]
        print("  arena %d  %#x  %s" % (i, a.ptr, "  ".join("%.4f" % x for x in t)))
    print("ONE arena of 12 buffer sizes: ms into each twelfth")
    a = _LibBuffer(L, 12 * step, dev)
    if a.ok:
        print("  %#x  %s" % (a.ptr, "  ".join("%.4f" % raster(a.ptr + k * step) for k in range(12))))
    print("offsets inside the first arena-of-12 (first buffer size; 64 KiB ... 1 MiB shifts): is it the alignment?")
    if a.ok:
        for sh in (0, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20):
            print("  +%-8d %.4f" % (sh, raster(a.ptr + sh)))


def scan():
    """The class as a function of WHERE the buffer lies: one big arena, the raster timed into a window of one buffer
    size that slides over it — coarse (128 MiB steps), then fine around the fastest spots (2 MiB steps, then 64 KiB),
    then the coarse scan again (is a spot's speed reproducible?), then a second arena (do fast spots sit at the same
    offsets / the same address bits?)."""
    import ctypes as C
    import torch
    from marlgrid_amd import _native as N
    from marlgrid_amd.base import _LibBuffer
    from marlgrid_amd.envs import make
    B = int(os.environ.get("B", "32768"))
    env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, auto_reset=True, strict=False, place_obs=False)
    env.reset()
    L, ms, nbytes, dev = N.lib(), C.c_float(0), env.obs.numel(), env.device

    def raster(ptr, iters=2):
        N.check(L.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), C.c_void_p(ptr), iters, C.byref(ms), env._stream()))
        return ms.value

    GiB, MiB = 1 << 30, 1 << 20
    arena_bytes = int(os.environ.get("ARENA_GIB", "40")) * GiB
    for ai in range(2):
        a = _LibBuffer(L, arena_bytes, dev)
        if not a.ok:
            print("arena %d: allocation of %d GiB failed" % (ai, arena_bytes // GiB))
            break
        last = arena_bytes - nbytes
        print("== arena %d at %#x, %d GiB; buffer %d bytes" % (ai, a.ptr, arena_bytes // GiB, nbytes))
        coarse = [(off, raster(a.ptr + off)) for off in range(0, last, 128 * MiB)]
        print("coarse (128 MiB steps): offset_MiB:ms")
        print("  " + " ".join("%d:%.3f" % (o // MiB, t) for o, t in coarse))
        ts = sorted(t for _, t in coarse)
        print("  min %.4f  p10 %.4f  median %.4f  max %.4f  | under 0.172: %d of %d, under 0.182: %d" % (
            ts[0], ts[len(ts) // 10], ts[len(ts) // 2], ts[-1], sum(t < 0.172 for t in ts), len(ts), sum(t < 0.182 for t in ts)))
        best = sorted(coarse, key=lambda ot: ot[1])[:3]
        for o0, t0 in best:
            lo, hi = max(0, o0 - 160 * MiB), min(last, o0 + 160 * MiB)
            fine = [(off, raster(a.ptr + off)) for off in range(lo, hi, 2 * MiB)]
            print("fine around %d MiB (coarse %.4f), 2 MiB steps: offset_MiB:ms" % (o0 // MiB, t0))
            print("  " + " ".join("%d:%.3f" % (o // MiB, t) for o, t in fine))
            fb = min(fine, key=lambda ot: ot[1])
            lo2, hi2 = max(0, fb[0] - 2 * MiB), min(last, fb[0] + 2 * MiB)
            vf = [(off, raster(a.ptr + off)) for off in range(lo2, hi2, 64 << 10)]
            print("  64 KiB steps around %d MiB (%.4f): offset_KiB:ms" % (fb[0] // MiB, fb[1]))
            print("    " + " ".join("%d:%.3f" % ((o - fb[0]) // 1024, t) for o, t in vf))
        again = [(off, raster(a.ptr + off)) for off, _ in coarse]
        print("coarse again: max |delta| %.4f ms, mean |delta| %.4f; spots under 0.172 both times: %d" % (
            max(abs(x[1] - y[1]) for x, y in zip(coarse, again)), sum(abs(x[1] - y[1]) for x, y in zip(coarse, again)) / len(coarse),
            sum(x[1] < 0.172 and y[1] < 0.172 for x, y in zip(coarse, again))))
        # a random sample at 2 MiB granularity: the distribution of classes over positions
        import random
        r = random.Random(ai)
        smp = sorted(raster(a.ptr + r.randrange(0, last // (2 * MiB)) * 2 * MiB) for _ in range(200))
        print("200 random 2 MiB-aligned offsets: min %.4f p5 %.4f p25 %.4f median %.4f p75 %.4f max %.4f | under 0.172: %d, 0.172-0.182: %d" % (
            smp[0], smp[10], smp[50], smp[100], smp[150], smp[-1], sum(t < 0.172 for t in smp), sum(0.172 <= t < 0.182 for t in smp)))
        keep_a = a      # (the second arena is new memory)
    print("fill of the first arena's first buffer-sized window, for scale: ", end="")
    t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t.fill_(1)
    e0.record()
    for _ in range(5):
        t.fill_(2)
    e1.record()
    torch.cuda.synchronize()
    print("%.4f ms per fill" % (e0.elapsed_time(e1) / 5))


def walk():
    """Does a sequence of buffer-sized hipMallocs that are all KEPT walk the physical memory in order — so that the
    one that straddles a 32 GiB physical boundary (the fast spot of the arena scan) comes up within ~36 draws?  And can
    the straddle then be CENTRED: the straddler and its predecessor freed, one allocation of 4 buffer sizes made in
    their place, and the raster timed into a window sliding over it?"""
    import ctypes as C
    import time
    import torch
    from marlgrid_amd import _native as N
    from marlgrid_amd.base import _LibBuffer
    from marlgrid_amd.envs import make
    env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=32768, auto_reset=True, strict=False, place_obs=False)
    env.reset()
    L, ms, nbytes, dev = N.lib(), C.c_float(0), env.obs.numel(), env.device
    MiB = 1 << 20

    def raster(ptr, iters=3):
        N.check(L.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), C.c_void_p(ptr), iters, C.byref(ms), env._stream()))
        return ms.value

    t0 = time.perf_counter()
    keep, times = [], []
    found = []
    for i in range(int(os.environ.get("WALK", "90"))):
        m = _LibBuffer(L, nbytes, dev)
        if not m.ok:
            print("allocation %d failed" % i)
            break
        keep.append(m)
        times.append(raster(m.ptr))
    print("walk: %d candidates kept alive, %.2f s; index:ms (ptr delta to the previous in MiB)" % (len(keep), time.perf_counter() - t0))
    print("  " + " ".join("%d:%.3f(%+d)" % (i, t, 0 if i == 0 else (keep[i].ptr - keep[i - 1].ptr) // MiB) for i, t in enumerate(times)))
    med = sorted(times)[len(times) // 2]
    dips = [i for i, t in enumerate(times) if t < 0.97 * med]
    print("median %.4f; dips (< 0.97 x median): %s" % (med, ["%d:%.4f" % (i, times[i]) for i in dips]))
    # centre the first dip that has a predecessor
    for c in dips:
        if c == 0:
            continue
        t0 = time.perf_counter()
        a, b = keep[c - 1], keep[c]
        lo_ptr = min(a.ptr, b.ptr)
        keep[c - 1] = keep[c] = None
        del a, b
        torch.cuda.synchronize()
        D = _LibBuffer(L, 4 * nbytes, dev)
        print("dip at %d: freed %d and %d, allocated 4 x buffer at %#x (the pair was at %#x): %s" % (
            c, c - 1, c, D.ptr if D.ok else 0, lo_ptr, "same start" if D.ok and D.ptr == lo_ptr else "elsewhere"))
        if not D.ok:
            break
        step = 32 * MiB
        sc = [(off, raster(D.ptr + off, 2)) for off in range(0, 3 * nbytes, step)]
        print("  window every 32 MiB: " + " ".join("%d:%.3f" % (o // MiB, t) for o, t in sc))
        best = min(sc, key=lambda ot: ot[1])
        print("  best window at +%d MiB: %.4f ms (%.2f s for the centring)" % (best[0] // MiB, best[1], time.perf_counter() - t0))
        keep.append(D)
        break


def pearson(xs, ys):
    n = len(xs)
    if n < 3:
        return float("nan")
    mx, my = sum(xs) / n, sum(ys) / n
    sxx = sum((x - mx) ** 2 for x in xs)
    syy = sum((y - my) ** 2 for y in ys)
    if sxx <= 0 or syy <= 0:
        return float("nan")
    return sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / (sxx * syy) ** 0.5


def driver(outdir, n):
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    outdir = os.path.abspath(outdir)          # (rocprofv3 runs with cwd /tmp)
    os.makedirs(outdir, exist_ok=True)
    report = open(os.path.join(outdir, "placement_pmc.txt"), "w")

    def say(*a):
        line = " ".join(str(x) for x in a)
        print(line, flush=True)
        report.write(line + "\n")
        report.flush()

    say("# tools/placement_pmc.py: %d hipMalloc candidates per pass, %d raster launches into each; per candidate the mean" % (n, ITERS))
    say("# of the launches' counters; r = Pearson correlation of the counter with the launch duration across candidates")
    for name, counters in GROUPS:
        d = os.path.join(outdir, "pmc_" + name)
        shutil.rmtree(d, ignore_errors=True)
        cmd = [exe, "--pmc"] + counters + ["-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
                                           os.path.abspath(__file__), "--inner", str(n)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True,
                               timeout=int(os.environ.get("PASS_TIMEOUT", "150")))
        except subprocess.TimeoutExpired as e:
            say("== %s: timeout | %s" % (name, str(e.stderr or "")[-300:].replace("\n", " | ")))
            continue
        with open(os.path.join(outdir, "log_%s.txt" % name), "w") as f:
            f.write("rc %d\n--- stdout\n%s\n--- stderr\n%s\n" % (r.returncode, r.stdout[-6000:], r.stderr[-6000:]))
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        meta = [ln for ln in r.stdout.splitlines() if ln.startswith("PLACEMENT_PMC ")]
        if r.returncode != 0 or not files or not meta:
            say("== %s: failed (rc %d) %s" % (name, r.returncode, (r.stderr or "")[-300:].replace("\n", " | ")))
            continue
        meta = json.loads(meta[0][len("PLACEMENT_PMC "):])
        # dispatches of the raster in order; a dispatch = all rows with one Dispatch_Id
        disp = collections.OrderedDict()
        for row in csv.DictReader(open(files[0])):
            if "render_kernel" not in row["Kernel_Name"]:
                continue
            e = disp.setdefault(row["Dispatch_Id"], {"c": collections.defaultdict(list)})
            e["c"][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                e["ns"] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        order = sorted(disp, key=lambda k: int(k))
        # the env's reset() rendered once into its own buffer before the candidates: take the LAST n * ITERS dispatches
        need = len(meta["rows"]) * ITERS
        order = order[-need:]
        if len(order) < need:
            say("== %s: only %d raster dispatches for %d candidates" % (name, len(order), len(meta["rows"])))
            continue
        cnames = sorted({c for k in order for c in disp[k]["c"]})
        multi = {c: max(len(disp[k]["c"][c]) for k in order) for c in cnames}
        say("== %s  (%s)" % (name, ", ".join("%s x%d" % (c, multi[c]) if multi[c] > 1 else c for c in cnames)))
        table = []
        for i, rowm in enumerate(meta["rows"]):
            ks = order[i * ITERS:(i + 1) * ITERS]
            ns = [disp[k].get("ns") for k in ks]
            dur = (sum(ns) / len(ns) / 1e6) if all(v is not None for v in ns) else rowm["event_ms"]
            vals = {}
            for c in cnames:
                per = [disp[k]["c"][c] for k in ks]
                vals[c] = sum(sum(p) for p in per) / len(per)                  # summed over instances, mean over launches
                if multi[c] > 1:                                                # per-instance: imbalance max / mean
                    inst = [sum(p[j] for p in per if j < len(p)) / len(per) for j in range(multi[c])]
                    mean = sum(inst) / len(inst)
                    vals[c + ":max/mean"] = max(inst) / mean if mean else float("nan")
            table.append((rowm["cand"], rowm["ptr"], dur, rowm["event_ms"], vals))
        keys = list(table[0][4])
        say("   cand ptr                 ms(dispatch)  ms(events)  " + "  ".join(keys))
        for cand, ptr, dur, ev, vals in sorted(table, key=lambda t: t[2]):
            say("   %3d  %#016x  %.4f       %.4f     %s" % (cand, ptr, dur, ev, "  ".join("%.6g" % vals[k] for k in keys)))
        durs = [t[2] for t in table]
        for k in keys:
            xs = [t[4][k] for t in table]
            fast = [x for x, dd in zip(xs, durs) if dd <= min(durs) * 1.06]
            slow = [x for x, dd in zip(xs, durs) if dd >= max(durs) * 0.94]
            say("   r(%s, ms) = %+.3f   fastest-class mean %.6g (%d)   slowest-class mean %.6g (%d)" % (
                k, pearson(xs, durs), sum(fast) / max(1, len(fast)), len(fast), sum(slow) / max(1, len(slow)), len(slow)))
    report.close()


if __name__ == "__main__":
    if "--inner" in sys.argv:
        inner(int(sys.argv[sys.argv.index("--inner") + 1]))
    elif "--subranges" in sys.argv:
        subranges()
    elif "--scan" in sys.argv:
        scan()
    elif "--walk" in sys.argv:
        walk()
    else:
        args = [a for a in sys.argv[1:] if not a.startswith("--")]
        driver(args[0] if args else os.path.join(ROOT, "gpurun_out", "placement_pmc"), int(os.environ.get("N", "20")))
