#!/usr/bin/env python
"""An ad-hoc sweep beyond the fuzz cases the suite runs: HIP batch == oracle, every step (tests/test_hip_parity.py::
test_batch_vs_oracle) for scenarios.fuzz_case(i) / fuzz_wide_case(i), i in the given ranges.  The oracle is pinned against the
LIVE reference for i < 51 (narrow) and i < 28 (wide) by the CPU suite; a mismatch found here is first re-checked there
(tests/test_oracle_vs_reference.py) before anybody blames the kernel.
usage: fuzz_sweep.py [--encode-in-step] [narrow_lo narrow_hi wide_lo wide_hi]"""
import os
import sys
import time
import traceback
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
warnings.simplefilter("ignore")
import test_hip_parity as T  # noqa: E402

ENC = "--encode-in-step" in sys.argv      # the encoding the test compares is the one the step's own launch (or the host behind it) wrote
if ENC:
    sys.argv.remove("--encode-in-step")
    import product_envs  # noqa: E402
    _build = product_envs.build

    from marlgrid_amd import base as PB  # noqa: E402
    _encode = PB.MultiGrid.encode
    used = [0]

    def encode(self, vis_mask=None):        # (the grid object is made anew by every reset: the class's method, not the instance's)
        e = self._env
        if vis_mask is None and e.encode_in_step and e.grid_encoding is not None:
            used[0] += 1
            return e.grid_encoding.clone()
        return _encode(self, vis_mask)
    PB.MultiGrid.encode = encode

    def build(name, **kw):
        return _build(name, encode_in_step=True, **kw)
    product_envs.build = build
a = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else [51, 300, 28, 100]
bad, t0, n = [], time.time(), 0
for name, lo, hi, Bf, steps in (("Fuzz-%d", a[0], a[1], lambda i: 40 + (i % 3) * 33, 70), ("FuzzW-%d", a[2], a[3], lambda i: 9 + (i % 4) * 20, 45)):
    for i in range(lo, hi):
        try:
            T.test_batch_vs_oracle(name % i, Bf(i), steps)
        except NotImplementedError as e:          # a configuration the host rejects loudly (LDS budget): not a parity matter
            print("skip", name % i, str(e)[:90], flush=True)
        except Exception as e:                    # noqa: BLE001
            bad.append(name % i)
            print("FAIL", name % i, type(e).__name__, str(e)[:300], flush=True)
            traceback.print_exc(limit=3)
        n += 1
print("%d cases in %.0f s; failures: %s%s" % (n, time.time() - t0, bad or "none", "; encodings taken from env.grid_encoding: %d" % used[0] if ENC else ""))
