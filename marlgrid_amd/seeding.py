"""Seed -> MT19937 key words, host side.

The reference seeds each env with `gym.utils.seeding.np_random(seed)` (marlgrid/base.py:371-374).
gym is an unpinned third-party dependency that is not part of the reference tree; its <= 0.21
behaviour is restated here: sha512(str(seed)) -> first 8 bytes (+4 zero bytes of padding, an
upstream quirk) -> little-endian uint32 words -> big integer -> base-2**32 digits, which numpy's
`RandomState.seed(list)` feeds to MT19937 `init_by_array`.  The 624-word recurrence itself runs on
the device (`mg_mt_seed`); only the hashing happens here.
"""
import hashlib
import struct

import numpy as np

KEY_WORDS = 2


def seed_words(seed):
    seed = int(seed)
    if seed < 0:
        raise ValueError("Seed must be a non-negative integer")
    seed %= 2 ** 64
    digest = hashlib.sha512(str(seed).encode("utf8")).digest()[:8]
    digest += b"\0" * (4 - len(digest) % 4)
    words = struct.unpack("{}I".format(len(digest) // 4), digest)
    big = sum(w << (32 * i) for i, w in enumerate(words))
    if big == 0:
        return [0]
    out = []
    while big > 0:
        big, mod = divmod(big, 2 ** 32)
        out.append(mod)
    return out


def batch_keys(seeds):
    """seeds: iterable of ints -> (keys uint32 [B, KEY_WORDS], key_len int32 [B])"""
    seeds = [int(s) for s in seeds]
    keys = np.zeros((len(seeds), KEY_WORDS), np.uint32)
    lens = np.zeros(len(seeds), np.int32)
    for i, s in enumerate(seeds):
        w = seed_words(s)
        keys[i, :len(w)] = w
        lens[i] = len(w)
    return keys, lens
