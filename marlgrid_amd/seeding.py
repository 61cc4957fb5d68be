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


_UPPER, _LOWER, _MAG = 0x80000000, 0x7FFFFFFF, 0x9908B0DF


def numpy_form(mt, gen_pos, head):
    """Device RNG state -> numpy's (key[624], pos).

    The device keeps MT19937 in *lazy* form with a look-ahead head (csrc/mg_core.h): slots below
    `gen_pos` hold words of the current block, slots from `gen_pos` on still hold the previous
    block's, and the `head` outputs before `gen_pos` are generated but not consumed, i.e. the stream
    position is gen_pos - head.  numpy always holds one whole block:
      * position inside the current block: regenerate the tail [gen_pos, 624) the way numpy's block
        loop would have;
      * position still in the previous block (gen_pos <= head): wind the slots [0, gen_pos) back.
        x_new[i] = x_old[i+397] ^ twist(x_old[i] & UPPER | x_old[i+1] & LOWER) gives the top bit of
        x_old[i] and the low 31 bits of x_old[i+1]; the low 31 bits of x_old[0] are not recoverable —
        and never read again by MT19937 — and are returned as 0.
    """
    mt = [int(v) for v in mt]
    G = int(gen_pos)
    if G > head:
        for kk in range(G, 624):
            y = (mt[kk] & _UPPER) | (mt[(kk + 1) % 624] & _LOWER)
            mt[kk] = mt[(kk + 397) % 624] ^ (y >> 1) ^ (_MAG if y & 1 else 0)
        return np.array(mt, np.uint32), G - head
    ys = []
    for i in range(G):
        t = mt[i] ^ mt[i + 397]
        ys.append((((t ^ _MAG) << 1) | 1) & 0xFFFFFFFF if t & _UPPER else (t << 1) & 0xFFFFFFFF)
    for i in range(G):
        mt[i] = (ys[i] & _UPPER) | ((ys[i - 1] & _LOWER) if i > 0 else 0)
    return np.array(mt, np.uint32), G - head + 624


def same_stream(a, b):
    """Two numpy-form states (key, pos) describe the same MT19937 stream: equal position, equal
    words 1..623 and equal top bit of word 0 (the generator's state is 19937 bits: the low 31 bits
    of word 0 are never read)."""
    (ka, pa), (kb, pb) = a, b
    ka, kb = np.asarray(ka, np.uint32), np.asarray(kb, np.uint32)
    return int(pa) == int(pb) and np.array_equal(ka[1:], kb[1:]) and (int(ka[0]) >> 31) == (int(kb[0]) >> 31)
