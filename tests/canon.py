"""Canonical (id-free) state comparison helpers shared by the parity tests."""
import numpy as np

from oracle import oracle as O


def obj_enc_table(spec):
    """object id -> WorldObj.encode() triple (objects.py:90-99) for a scenario spec."""
    t = np.zeros((len(spec["objects"]), 3), np.uint8)
    for i, o in enumerate(spec["objects"]):
        if o is not None:
            t[i] = (O.TYPE_IDX[o["type"]], O.COLOR_TO_IDX[o["color"]], o.get("state", 0))
    return t


def from_ids(spec, base_ids, pos, dir_, active, done, carrying, ordinal, step_count):
    """Build the canonical dict from id-based state (oracle or HIP)."""
    enc = obj_enc_table(spec)
    pos = np.asarray(pos).astype(np.int16)
    return dict(base_enc=enc[np.asarray(base_ids)], pos=pos, dir=np.asarray(dir_).astype(np.int8),
                active=np.asarray(active).astype(bool), done=np.asarray(done).astype(bool),
                carry_enc=enc[np.asarray(carrying)], ordinal=np.asarray(ordinal).astype(np.int8),
                step_count=int(step_count))


def oracle_canonical(orc):
    s = orc.state()
    return from_ids(orc.spec, s["base"], s["pos"], s["dir"], s["active"], s["done"], s["carrying"],
                    s["ordinal"], s["step_count"])


KEYS = ("base_enc", "pos", "dir", "active", "done", "carry_enc", "ordinal", "step_count")


def assert_same(a, b, what=""):
    for k in KEYS:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), "%s: canonical field %r differs\n%r\n%r" % (
            what, k, a[k], b[k])
