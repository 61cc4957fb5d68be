"""gym <= 0.21 seeding, restated from its public behaviour.

np_random(seed): seed -> sha512(str(seed)) -> first 8 bytes -> (quirk: padded with 4 more zero
bytes) little-endian uint32 words -> big integer -> base-2**32 digit list ->
numpy.random.RandomState().seed(list)  (MT19937 init_by_array).
"""
import hashlib
import struct

import numpy as np


def _hash_seed(seed, max_bytes=8):
    digest = hashlib.sha512(str(seed).encode("utf8")).digest()[:max_bytes]
    digest += b"\0" * (4 - len(digest) % 4)          # the upstream padding quirk (always pads)
    words = struct.unpack("{}I".format(len(digest) // 4), digest)
    return sum(w << (32 * i) for i, w in enumerate(words))


def _digits(bigint):
    if bigint == 0:
        return [0]
    out = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        out.append(mod)
    return out


def np_random(seed=None):
    if seed is None:
        raise ValueError("refshim: explicit integer seeds only")
    seed = int(seed) % 2 ** 64
    rng = np.random.RandomState()
    rng.seed(_digits(_hash_seed(seed)))
    return rng, seed
