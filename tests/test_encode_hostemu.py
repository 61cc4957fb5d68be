"""mg_encode's kernel (marlgrid_amd/csrc/mg_encode_core.h — the text its phases run per thread) compiled for the host and run
piece by piece, phase by phase, thread by thread (tests/native), against a cell-by-cell MultiGrid.encode (base.py:196-214)
written out in numpy here: every piece boundary, chunk phase and env crossing of the flat output stream, both piece sizes,
both agent-mark layouts (shared byte / second plane), vis_mask, a misaligned `out`, grids from 3x3 to 70x70 — and every LDS
offset the chunk phase forms is checked against the plane the launcher sizes.  No GPU; the GPU twin is every `encode`
comparison of tests/test_hip_parity.py (each step of each golden) + test_hip_host_contract.py::test_encode_*."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "native"))

from marlgrid_amd import _native as N  # noqa: E402


def encode_reference(grid, rec, W, H, n_obj, obj_enc, agent_type_idx, agent_colors, vis=None):
    """cell by cell: the base object's triple; on an empty cell the triple of the first (lowest-rank) placed agent there"""
    B = grid.shape[0]
    out = np.zeros((B, W * H, 3), np.uint8)
    by = lambda r, i: int((int(r) >> (8 * i)) & 0xFF)      # noqa: E731
    for b in range(B):
        base = grid[b, :W * H]
        out[b] = obj_enc[base]
        best = {}
        for k, r in enumerate(rec[b]):
            if by(r, N.AG_FLAGS) & N.AF_PLACED:
                c = by(r, N.AG_X) * H + by(r, N.AG_Y)
                if c not in best or by(r, N.AG_RANK) < best[c][0]:
                    best[c] = (by(r, N.AG_RANK), k, by(r, N.AG_DIR))
        for c, (_rk, k, d) in best.items():
            if base[c] == 0:
                out[b, c] = (agent_type_idx, agent_colors[k], d)
    if vis is not None:
        out[vis.reshape(B, W * H) == 0] = 0
    return out.reshape(B, W, H, 3)


def run_case(rng, B, W, H, n, n_obj, pc=0, with_vis=False, misalign=0, crowd=False):
    import hostemu
    L = hostemu.lib()
    cells, stride = W * H, (W * H + 15) // 16 * 16
    grid = np.full((B, stride), 0xEE, np.uint8)                 # (padding bytes are garbage in this test: never read as cells)
    g = rng.randint(0, n_obj, size=(B, cells)).astype(np.uint8)
    g[rng.rand(B, cells) < 0.55] = 0
    grid[:, :cells] = g
    rec = np.zeros((B, n), np.uint64)
    for b in range(B):
        ranks = rng.permutation(n)
        for k in range(n):
            x, y = (rng.randint(0, 2), rng.randint(0, 2)) if crowd else (rng.randint(0, W), rng.randint(0, H))
            if rng.rand() < 0.15:       # the env's last cell: where the flat stream crosses into the next env
                x, y = W - 1, H - 1
            fl = N.AF_PLACED | N.AF_ACTIVE if rng.rand() < 0.9 else 0
            rec[b, k] = x | (y << 8) | (int(rng.randint(0, 4)) << 16) | (fl << 24) | (int(ranks[k]) << 40)
            if fl and rng.rand() < 0.6:
                grid[b, x * H + y] = 0                              # most agents stand on empty cells
    tab = (N.ObjDesc * n_obj)()
    obj_enc = np.zeros((n_obj, 3), np.uint8)
    for o in range(1, n_obj):
        obj_enc[o] = (rng.randint(1, 14), rng.randint(0, 14), rng.randint(0, 4))
        tab[o].type_idx, tab[o].color_idx, tab[o].state = (int(v) for v in obj_enc[o])
        tab[o].flags = int(rng.randint(0, 256))                     # (must not leak into the triple)
    cfg = N.Config()
    cfg.B, cfg.W, cfg.H, cfg.n_agents, cfg.cells_stride, cfg.n_obj = B, W, H, n, stride, n_obj
    cfg.agent_type_idx = 13
    colors = [int(rng.randint(0, 14)) for _ in range(n)]
    for k in range(n):
        cfg.agent_color_idx[k] = colors[k]
    cfg.obj = C.cast(tab, C.c_void_p).value
    st = N.State()
    st.grid, st.agents = grid.ctypes.data, rec.ctypes.data
    vis = (rng.rand(B, W, H) < 0.7).astype(np.uint8) if with_vis else None
    buf = np.full(B * cells * 3 + 64, 0x5A, np.uint8)
    off = (-buf.ctypes.data) % 16 + misalign
    out = buf[off:off + B * cells * 3]
    # (the runs' store phase as large batches take it — a barrier, a lane per chunk of the piece — and as small ones do — a wave
    # streams its own runs out —: the launcher picks by the number of pieces, the test takes turns)
    run_case.turn = getattr(run_case, "turn", 0) + 1
    L.emu_encode_force_runs(1 + run_case.turn % 2)
    rc = L.emu_encode(C.byref(cfg), C.byref(st), None if vis is None else C.c_void_p(vis.ctypes.data), C.c_void_p(out.ctypes.data), pc)
    L.emu_encode_force_runs(0)
    assert rc == 0, "out-of-range LDS offsets: %d" % rc
    want = encode_reference(grid, rec, W, H, n_obj, obj_enc, 13, colors, vis)
    got = out.reshape(B, W, H, 3)
    if not np.array_equal(got, want):
        bad = np.argwhere((got != want).any(axis=-1))
        raise AssertionError("encode differs at (env, x, y) %s ... of %d cells; got %s want %s"
                             % (bad[:4].tolist(), len(bad), got[tuple(bad[0])], want[tuple(bad[0])]))
    assert (buf[:off] == 0x5A).all() and (buf[off + B * cells * 3:] == 0x5A).all(), "wrote outside `out`"
    # the same batch through the fused step's encode (a wave's staged batch of <= 8 envs at a time: encode_batch_mark /
    # encode_batch_chunks), walked with several envs-per-wave so that batches start at every phase of the 16-byte chunks
    if vis is None and n_obj + 4 * n <= 256:
        for per_wave in sorted({1, 3, 8, 13, 64} if B <= 64 else {8}):
            buf2 = np.full(B * cells * 3 + 64, 0x5A, np.uint8)
            off2 = (-buf2.ctypes.data) % 16 + misalign
            out2 = buf2[off2:off2 + B * cells * 3]
            rc = L.emu_encode_batch(C.byref(cfg), C.byref(st), C.c_void_p(out2.ctypes.data), per_wave, n + int(rng.randint(0, 3)))
            assert rc == 0, "fused encode: out-of-range LDS offsets: %d" % rc
            got2 = out2.reshape(B, W, H, 3)
            if not np.array_equal(got2, want):
                bad = np.argwhere((got2 != want).any(axis=-1))
                raise AssertionError("fused encode (per_wave %d) differs at (env, x, y) %s ... of %d cells; got %s want %s"
                                     % (per_wave, bad[:4].tolist(), len(bad), got2[tuple(bad[0])], want[tuple(bad[0])]))
            assert (buf2[:off2] == 0x5A).all() and (buf2[off2 + B * cells * 3:] == 0x5A).all(), "fused encode wrote outside `out`"


@pytest.mark.parametrize("W,H,n", [(15, 15, 3), (11, 11, 3), (9, 9, 4), (30, 30, 8), (3, 3, 1), (3, 5, 2), (7, 11, 3), (40, 40, 2),
                                   (70, 66, 5), (16, 16, 16), (5, 5, 12), (64, 64, 3), (65, 63, 1)])
def test_encode_kernel_on_the_host(W, H, n):
    rng = np.random.RandomState(W * 100 + H + n)
    for B in (1, 2, 7, 19, 64):
        if B * W * H > 120000:
            continue
        for pc in (1024, 4096, 8192):
            run_case(rng, B, W, H, n, n_obj=int(rng.randint(2, 40)), pc=pc)
    run_case(rng, 33, W, H, n, n_obj=12, with_vis=True)
    run_case(rng, 9, W, H, n, n_obj=12, misalign=int(rng.randint(1, 16)))
    run_case(rng, 21, W, H, n, n_obj=5, crowd=True, pc=4096)


def test_encode_agent_marks_in_their_own_plane():
    """object ids and agent codes do not fit one byte (n_obj + 4 n > 256): the second byte plane"""
    rng = np.random.RandomState(5)
    for (W, H, n, n_obj) in [(15, 15, 16, 250), (9, 9, 16, 200), (30, 30, 12, 255), (5, 7, 16, 255)]:
        for pc in (1024, 4096):
            run_case(rng, 23, W, H, n, n_obj=n_obj, pc=pc)
        run_case(rng, 11, W, H, n, n_obj=n_obj, with_vis=True, crowd=True)


def test_encode_launcher_piece_choice():
    """the launcher's own choice of piece size (0): a batch large enough for 4096-cell pieces"""
    rng = np.random.RandomState(6)
    run_case(rng, 19000, 15, 15, 3, n_obj=3)          # 4.3 M cells: 522 pieces of 8 192
    run_case(rng, 9500, 15, 15, 3, n_obj=3)           # 2.1 M cells: 522 pieces of 4 096
