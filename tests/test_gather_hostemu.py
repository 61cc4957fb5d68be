"""The gather raster's index arithmetic (marlgrid_amd/csrc/mg_gather.h — what the obs kernel runs per lane for tile
sizes 5 / 6) built for the host (tests/native) and compared with a byte-by-byte raster of the same tile map: every
alignment of the group's stream, groups of 1..8 envs, 1..3 viewers, and geometries beyond the two the product
instantiates.  Test infrastructure only: observations are rendered by the HIP kernel (tests/test_hip_parity.py)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "native"))
import hostemu  # noqa: E402

GEOMS = [(7, 5), (7, 6), (7, 7), (7, 9), (7, 10), (7, 11), (7, 12), (5, 5), (9, 5), (9, 6), (3, 5), (6, 5), (4, 6), (4, 5), (8, 5), (11, 5), (13, 5), (15, 5)]


def _naive(vs, ts, tmap, atlas, n_bands):
    """[band][pixel row of the tile][column] -> the tile row's 3 * ts bytes"""
    t = tmap.reshape(n_bands, vs)
    rows = atlas[t]                                    # [band][col][ts][seg]
    return np.ascontiguousarray(rows.transpose(0, 2, 1, 3)).reshape(-1)


@pytest.mark.parametrize("vs,ts", GEOMS)
def test_gather_raster_equals_bytewise_raster(vs, ts):
    L = hostemu.lib()
    L.emu_gather.restype = C.c_int
    rng = np.random.RandomState(vs * 100 + ts)
    seg = 3 * ts
    geom = np.zeros(8, np.int32)
    for trial in range(40):
        n_vt = int(rng.randint(1, 120))
        nv = int(rng.randint(1, 4))
        G = int(rng.choice([1, 2, 3, 4, 8]))
        phase = trial % 16
        n_dyn = int(rng.choice([0, 0, 4, 24]))            # a wave's own recoloured tiles ('prestige'): virtual tiles behind the atlas's
        atlas = rng.randint(1, 256, size=(n_vt + n_dyn, ts, seg)).astype(np.uint8)       # (no zero bytes: a byte that is not ORed in shows)
        n_bands = G * nv * vs
        tmap = rng.randint(0, n_vt + n_dyn, size=n_bands * vs).astype(np.uint16)
        want = _naive(vs, ts, tmap, atlas, n_bands)
        assert want.size == G * nv * (vs * ts) * (vs * ts) * 3
        buf = np.full(want.size + 64 + 16, 0xEE, np.uint8)
        base = buf.ctypes.data
        off = (-base) % 16 + 16 + phase                                           # the stream starts `phase` bytes behind a 16-byte boundary
        rc = L.emu_gather(vs, ts, n_vt, n_dyn, C.c_void_p(atlas.ctypes.data), C.c_void_p(tmap.ctypes.data), tmap.size,
                          C.c_void_p(base + off), C.c_uint32(want.size), C.c_void_p(geom.ctypes.data))
        assert rc == 0, "out-of-range LDS offsets formed: %d (geometry %s)" % (rc, geom.tolist())
        got = buf[off:off + want.size]
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "vs %d ts %d trial %d phase %d G %d nv %d: first bad byte %d of %d (geometry %s)" % (
            vs, ts, trial, phase, G, nv, bad[0], want.size, geom.tolist())
        assert (buf[:off] == 0xEE).all() and (buf[off + want.size:] == 0xEE).all(), "wrote outside the stream"
    assert geom[0] == seg
    if (vs, ts) == (7, 6):
        assert geom.tolist() == [18, 36, 63, 8, 3, 3, 63, 1]     # 63 lanes, three sets, tile rows constant
    if (vs, ts) == (5, 5):
        assert geom.tolist() == [15, 32, 75, 16, 3, 4, 57, 0]    # three periods in four trips of 57 lanes (one period: 38 of 64)
    if (vs, ts) == (7, 5):
        assert geom.tolist() == [15, 32, 105, 16, 1, 2, 53, 0]   # 53 + 52 lanes, bands per trip
