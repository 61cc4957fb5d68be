"""Tile atlas construction (host side, runs once per env configuration).

The reference rasterises sprites lazily into a process-global cache
(`MultiGrid.render_tile / cache_render_obj / render_object / blend_tiles / empty_tile`,
marlgrid/base.py:225-299) on top of gym-minigrid's `fill_coords / point_in_* / rotate_fn /
downsample` (third-party, unpinned, not in the reference tree — restated here from its public
behaviour: 3x3 supersampled coverage, float64 mean-of-means, truncation).  Here every tile the
obs kernel can ever need is produced up front, for all four view orientations, as one immutable
uint8 array that is uploaded to HBM and staged into LDS by `mg_render_obs`:

    atlas[o][t] : (ts, ts, 3)   o = orientation of the viewing agent's sub-grid = (3 - dir) % 4
      t = 0                                   shadow (invisible cell, objects.py:25)
      t = 1 + obj                             object `obj` alone (obj 0: empty tile)
      t = 1 + n_obj + (slot*n_agents + k)*4+d overlappable object of `slot` (slot 0: empty cell)
                                              with agent k facing d drawn on top

The agent sprite is indexed by (colour, absolute dir); the blend and the "border if a corner is
black" rule are applied before rotation, exactly where the reference applies them.
"""
import math

import numpy as np

from .objects import COLORS

SUBDIVS = 3     # base.py:277 forces subdivs = 3


def _sample_grid(S):
    c = (np.arange(S, dtype=np.float64) + 0.5) / S
    return np.meshgrid(c, c)        # xf[y, x], yf[y, x]


def _in_triangle(x, y, a, b, c):
    """gym-minigrid point_in_triangle: barycentric test, float64."""
    a, b, c = (np.asarray(v, np.float64) for v in (a, b, c))
    v0, v1 = c - a, b - a
    v2x, v2y = x - a[0], y - a[1]
    dot00 = v0[0] * v0[0] + v0[1] * v0[1]
    dot01 = v0[0] * v1[0] + v0[1] * v1[1]
    dot02 = v0[0] * v2x + v0[1] * v2y
    dot11 = v1[0] * v1[0] + v1[1] * v1[1]
    dot12 = v1[0] * v2x + v1[1] * v2y
    inv = 1 / (dot00 * dot11 - dot01 * dot01)
    u = (dot11 * dot02 - dot01 * dot12) * inv
    v = (dot00 * dot12 - dot01 * dot02) * inv
    return (u >= 0) & (v >= 0) & ((u + v) < 1)


def _coverage(op, S):
    kind, p = op[0], op[1]
    xf, yf = _sample_grid(S)
    if kind == "rect":
        return (xf >= p[0]) & (xf <= p[1]) & (yf >= p[2]) & (yf <= p[3])
    if kind == "circle":
        return (xf - p[0]) * (xf - p[0]) + (yf - p[1]) * (yf - p[1]) <= p[2] * p[2]
    if kind == "agent":
        # objects.py:150-153: triangle rotated by theta = 0.5*pi*dir about (0.5, 0.5); rotate_fn
        # evaluates the shape at the sample rotated by -theta
        theta = p[0]
        x, y = xf - 0.5, yf - 0.5
        x2 = 0.5 + x * math.cos(-theta) - y * math.sin(-theta)
        y2 = 0.5 + y * math.cos(-theta) + x * math.sin(-theta)
        return _in_triangle(x2, y2, (0.12, 0.19), (0.87, 0.50), (0.12, 0.81))
    raise ValueError(kind)


def render_sprite(ops, ts):
    """MultiGrid.render_object (base.py:252-258): paint at 3x, downsample by 3, truncate."""
    S = ts * SUBDIVS
    img = np.zeros((S, S, 3), np.uint8)
    for op in ops:
        img[_coverage(op, S)] = np.asarray(op[2], np.uint8)
    img = img.reshape(ts, SUBDIVS, ts, SUBDIVS, 3).mean(axis=3).mean(axis=1)
    return img.astype(np.uint8)


def agent_sprite(color, direction, ts):
    return render_sprite([("agent", (0.5 * np.pi * direction,), tuple(int(v) for v in COLORS[color]))], ts)


def empty_tile(ts):
    """MultiGrid.empty_tile (base.py:245-250): faint top/right border, nothing below 11 px."""
    alpha = max(0, min(20, ts - 10))
    img = np.full((ts, ts, 3), alpha, np.uint8)
    img[1:, :-1] = 0
    return img


def blend(base, top):
    """MultiGrid.blend_tiles (base.py:260-273)."""
    alpha = top.sum(2, keepdims=True, dtype=np.uint64)
    max_alpha = alpha.max()
    if max_alpha == 0:
        return base
    return ((base * (max_alpha - alpha) + top * alpha) / max_alpha).astype(base.dtype)


def with_border(img, ts):
    """tail of MultiGrid.render_tile (base.py:296-298): uint8 wrap-around add of the empty tile
    whenever one of the four corner pixels is black."""
    corners = img[[0, 0, -1, -1], [0, -1, 0, -1]]
    if (corners == 0).all(axis=-1).any():
        return (img + empty_tile(ts)).astype(np.uint8)
    return img


def rotate_tile(t, orientation):
    """rotate_grid as applied to a tile at base.py:324."""
    o = orientation % 4
    if o == 3:
        return np.moveaxis(t[:, ::-1], 0, 1)
    if o == 1:
        return np.moveaxis(t[::-1, :], 0, 1)
    if o == 2:
        return t[::-1, ::-1]
    return t


def build_atlas(objects, agent_colors, ts, prestige_sprites=False):
    """objects: list indexed by object id (index 0 = None).  Returns
    (atlas uint8 [4][n_tiles][ts][ts][3], ovl_slot list[int] per object id, n_slots).
    prestige_sprites: append the 4 un-bordered white agent sprites (dir 0..3) that the kernels
    recolour per env for 'prestige' agents (GridAgentInterface.render_post, agents.py:92-119)."""
    n_obj, n_ag = len(objects), len(agent_colors)
    ovl_slot = [0xFF] * n_obj
    ovl_slot[0] = 0
    n_slots = 1
    for i, o in enumerate(objects):
        if o is not None and o.can_overlap():
            ovl_slot[i] = n_slots
            n_slots += 1
    n_tiles = 1 + n_obj + n_slots * n_ag * 4 + (4 if prestige_sprites else 0)
    tiles = np.zeros((n_tiles, ts, ts, 3), np.uint8)
    tiles[0] = np.asarray(COLORS["shadow"], np.uint8)
    plain = [None] * n_obj
    for i, o in enumerate(objects):
        if o is None:
            tiles[1 + i] = empty_tile(ts)
            continue
        try:
            plain[i] = render_sprite(o.sprite_ops(), ts)
        except NotImplementedError:
            plain[i] = np.zeros((ts, ts, 3), np.uint8)   # unrenderable upstream as well (Floor/Lava/..)
        tiles[1 + i] = with_border(plain[i], ts)
    sprites = [[agent_sprite(c, d, ts) for d in range(4)] for c in agent_colors]
    for i in range(n_obj):
        s = ovl_slot[i]
        if s == 0xFF:
            continue
        for k in range(n_ag):
            for d in range(4):
                t = 1 + n_obj + (s * n_ag + k) * 4 + d
                img = sprites[k][d] if i == 0 else blend(plain[i], sprites[k][d])
                tiles[t] = with_border(img, ts)
    if prestige_sprites:
        for d in range(4):
            tiles[n_tiles - 4 + d] = agent_sprite("prestige", d, ts)
    atlas = np.stack([np.stack([rotate_tile(t, o) for t in tiles]) for o in range(4)])
    return np.ascontiguousarray(atlas), ovl_slot, n_slots
