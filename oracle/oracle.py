"""ctypes front-end of the CPU ORACLE (`oracle/mg_oracle.c`).  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg import this module.
The product package `marlgrid_amd` never does (a test asserts that).

The oracle builds its own object semantics from *type names*: the class tables below restate
`marlgrid/objects.py` (predicates, encode indices, sprites) and `marlgrid/agents.py` independently
of the product's tables in `marlgrid_amd/objects.py`; the two meet only through a plain-data
"scenario spec" (object list by type name/colour/state + `_gen_grid` program).

Parity pin: checked against the real reference in `tests/test_oracle_vs_reference.py` (live, build
container only) and against the committed fixtures in `tests/golden/`.
"""
import ctypes as C
import hashlib
import math
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmgoracle.so")

MAX_AGENTS, MAX_OBJ, MAX_FILL, MAX_GEN = 32, 256, 8, 192

# objects.py:11-29
COLORS = {
    "red": (255, 0, 0), "orange": (255, 165, 0), "green": (0, 255, 0), "blue": (0, 0, 255),
    "cyan": (0, 139, 139), "purple": (112, 39, 195), "yellow": (255, 255, 0),
    "olive": (128, 128, 0), "grey": (100, 100, 100), "worst": (74, 65, 42),
    "pink": (255, 0, 189), "white": (255, 255, 255), "prestige": (255, 255, 255),
    "shadow": (35, 25, 30),
}
COLOR_TO_IDX = {k: i for i, k in enumerate(COLORS)}

# objects.py:31-43 — definition order under the RegisteredObjectType metaclass, plus
# GridAgentInterface registered when agents.py is imported
TYPE_IDX = {"WorldObj": 0, "GridAgent": 1, "BulkObj": 2, "BonusTile": 3, "Goal": 4, "Floor": 5,
            "EmptySpace": 6, "Lava": 7, "Wall": 8, "Key": 9, "Ball": 10, "Door": 11, "Box": 12,
            "GridAgentInterface": 13}

# predicates per class: (can_overlap, can_pickup, see_behind) — objects.py:75-88 + overrides
# (Goal :216-217, BonusTile :174-175, Floor :230-231, Lava :258-259, Wall :281-282, Key :292-293,
#  Ball :314-315, Box :378-379; Door depends on state :327-331; EmptySpace's `can_verlap` typo
#  :250 leaves can_overlap False)
_PRED = {
    "Wall": (0, 0, 0), "Goal": (1, 0, 1), "BonusTile": (1, 0, 1), "Floor": (1, 0, 1),
    "Lava": (1, 0, 1), "EmptySpace": (0, 0, 1), "Key": (0, 1, 1), "Ball": (0, 1, 1),
    "Box": (0, 1, 1),
}
DOOR_OPEN, DOOR_CLOSED, DOOR_LOCKED = 1, 2, 3   # objects.py:325


class FillOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("p", C.c_double * 4), ("rgb", C.c_uint8 * 3), ("pad", C.c_uint8)]


class ObjDesc(C.Structure):
    _fields_ = [("type_idx", C.c_int32), ("color_idx", C.c_int32), ("state", C.c_int32),
                ("can_overlap", C.c_int32), ("can_pickup", C.c_int32), ("see_behind", C.c_int32),
                ("reward_kind", C.c_int32), ("reward", C.c_double), ("penalty", C.c_double),
                ("bonus_id", C.c_int32), ("n_bonus", C.c_int32), ("initial_reward", C.c_int32),
                ("reset_on_mistake", C.c_int32), ("ends_episode", C.c_int32),
                ("toggle_kind", C.c_int32), ("toggle_next", C.c_int32), ("unlock_next", C.c_int32),
                ("is_key", C.c_int32), ("n_fill", C.c_int32), ("fill", FillOp * MAX_FILL)]


class GenOp(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "obj", "count", "x", "y", "w", "h", "max_tries")] + [("reject", C.c_void_p)]


class Config(C.Structure):
    _fields_ = [("W", C.c_int32), ("H", C.c_int32), ("n_agents", C.c_int32),
                ("view_size", C.c_int32), ("tile_size", C.c_int32), ("view_offset", C.c_int32),
                ("see_through_walls", C.c_int32),
                ("max_steps", C.c_int32), ("reward_decay", C.c_int32), ("ghost_mode", C.c_int32),
                ("respawn", C.c_int32),
                ("agent_color_idx", C.c_int32 * MAX_AGENTS), ("agent_rgb", (C.c_uint8 * 4) * MAX_AGENTS),
                ("agent_type_idx", C.c_int32), ("n_obj", C.c_int32), ("obj", ObjDesc * MAX_OBJ),
                ("wall_obj", C.c_int32), ("n_gen", C.c_int32 * 2), ("gen", (GenOp * MAX_GEN) * 2),
                ("spawn_delay", C.c_int32 * MAX_AGENTS), ("is_prestige", C.c_int32 * MAX_AGENTS),
                ("prestige_beta", C.c_double * MAX_AGENTS), ("prestige_scale", C.c_double * MAX_AGENTS),
                ("hide_type_mask", C.c_uint32 * MAX_AGENTS),
                ("spawn_x0", C.c_int32), ("spawn_y0", C.c_int32), ("spawn_x1", C.c_int32), ("spawn_y1", C.c_int32),
                ("spawn_max_tries", C.c_int32), ("spawn_reject", C.c_void_p)]


_lib = None


def build(force=False):
    """Compile the oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "mg_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src),
                                                   os.path.getmtime(os.path.join(_HERE, "mg_oracle.h")))):
        return _LIB_PATH
    import fcntl
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:      # (one builder at a time under pytest-xdist)
        fcntl.flock(lock, fcntl.LOCK_EX)
        if (force or not os.path.exists(_LIB_PATH)
                or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "mg_oracle.h")))):
            subprocess.check_call(["make", "-C", _HERE, "-B", "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i32p, u32p, u8p, f64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
        L.mgo_create.restype = vp
        L.mgo_create.argtypes = [C.POINTER(Config), u32p, C.c_int32]
        L.mgo_create_like.restype = vp
        L.mgo_create_like.argtypes = [vp, u32p, C.c_int32]
        L.mgo_destroy.argtypes = [vp]
        L.mgo_reset.argtypes = [vp, C.c_int32]
        L.mgo_step.argtypes = [vp, i32p, f64p, i32p, i32p]
        L.mgo_render_obs.argtypes = [vp, C.c_int32, u8p]
        L.mgo_view.argtypes = [vp, C.c_int32, u8p, i32p]
        L.mgo_encode.argtypes = [vp, u8p, u8p]
        L.mgo_tile.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, u8p]
        L.mgo_occlude.argtypes = [C.c_int32, C.c_int32, C.c_int32, u8p, u8p]
        L.mgo_get_state.argtypes = [vp, u8p, i32p, i32p]
        L.mgo_get_mt.argtypes = [vp, u32p, i32p]
        L.mgo_get_prestige.argtypes = [vp, f64p]
        L.mgo_rich_obs.restype = None
        L.mgo_rich_obs.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), f64p, C.POINTER(C.c_int32)]
        L.mgo_set_agent_dir.argtypes = [vp, C.c_int32, C.c_int32]
        L.mgo_set_carrying.argtypes = [vp, C.c_int32, C.c_int32]
        L.mgo_regen_grid.argtypes = [vp, C.c_int32]
        L.mgo_place_obj.argtypes = [vp] + [C.c_int32] * 6 + [vp, i32p]
        L.mgo_try_place_obj.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
        L.mgo_put_obj.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
        L.mgo_place_agent_at.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
        L.mgo_mt_init_by_array.argtypes = [u32p, i32p, u32p, C.c_int32]
        L.mgo_mt_next.restype = C.c_uint32
        L.mgo_mt_next.argtypes = [u32p, i32p]
        L.mgo_bounded.restype = C.c_uint32
        L.mgo_bounded.argtypes = [u32p, i32p, C.c_uint32]
        L.mgo_batch_step.argtypes = [C.POINTER(vp), C.c_int32, i32p, f64p, u8p, u8p, C.c_int32, C.c_int32]
        L.mgo_max_threads.restype = C.c_int32
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ---------------------------------------------------------------------------------------------
# seeding: gym <= 0.21 `seeding.np_random(seed)` (the reference's base.py:373), restated
# ---------------------------------------------------------------------------------------------

def seed_key(seed):
    """seed -> uint32 key words handed to MT19937 init_by_array (sha512 -> first 8 bytes ->
    zero-padded -> little-endian words -> base-2**32 digits)."""
    seed = int(seed) % 2 ** 64
    digest = hashlib.sha512(str(seed).encode("utf8")).digest()[:8]
    digest += b"\0" * (4 - len(digest) % 4)
    words = struct.unpack("{}I".format(len(digest) // 4), digest)
    big = sum(w << (32 * i) for i, w in enumerate(words))
    if big == 0:
        return np.zeros(1, np.uint32)
    out = []
    while big > 0:
        big, mod = divmod(big, 2 ** 32)
        out.append(mod)
    return np.array(out, dtype=np.uint32)


# ---------------------------------------------------------------------------------------------
# scenario spec -> Config
# ---------------------------------------------------------------------------------------------

def _rect(xmin, xmax, ymin, ymax, rgb):
    op = FillOp()
    op.kind = 0
    op.p[:] = [xmin, xmax, ymin, ymax]
    op.rgb[:] = [int(v) & 255 for v in rgb]   # numpy uint8 assignment truncates
    return op


def _circle(cx, cy, r, rgb):
    op = FillOp()
    op.kind = 2
    op.p[:] = [cx, cy, r, 0.0]
    op.rgb[:] = [int(v) & 255 for v in rgb]
    return op


def _sprite(o):
    """obj.render(img) as a list of fill ops — objects.py render methods."""
    t, c = o["type"], COLORS[o["color"]]
    if t in ("Wall", "Goal", "BonusTile"):              # objects.py:288, 226, 209
        return [_rect(0, 1, 0, 1, c)]
    if t == "Box":                                      # objects.py:387-395
        return [_rect(0.12, 0.88, 0.12, 0.88, c), _rect(0.18, 0.82, 0.18, 0.82, (0, 0, 0)),
                _rect(0.16, 0.84, 0.47, 0.53, c)]
    if t == "Door":                                     # objects.py:348-370
        st = o.get("state", 0)
        if st == DOOR_OPEN:
            return [_rect(0.88, 1.00, 0.00, 1.00, c), _rect(0.92, 0.96, 0.04, 0.96, (0, 0, 0))]
        if st == DOOR_LOCKED:
            dim = [int(0.45 * v) for v in c]            # float -> uint8 truncation on assignment
            return [_rect(0.00, 1.00, 0.00, 1.00, c), _rect(0.06, 0.94, 0.06, 0.94, dim),
                    _rect(0.52, 0.75, 0.50, 0.56, c)]
        # closed: the reference raises NameError (point_in_circle never imported, objects.py:370);
        # sprite defined here by the obviously intended public gym-minigrid drawing
        return [_rect(0.00, 1.00, 0.00, 1.00, c), _rect(0.04, 0.96, 0.04, 0.96, (0, 0, 0)),
                _rect(0.08, 0.92, 0.08, 0.92, c), _rect(0.12, 0.88, 0.12, 0.88, (0, 0, 0)),
                _circle(0.75, 0.50, 0.08, c)]
    if t == "Ball":                                     # objects.py:320-321 (crashes upstream)
        return [_circle(0.5, 0.5, 0.31, c)]
    if t == "Key":                                      # objects.py:298-310 (crashes upstream)
        return [_rect(0.50, 0.63, 0.31, 0.88, c), _rect(0.38, 0.50, 0.59, 0.66, c),
                _rect(0.38, 0.50, 0.81, 0.88, c), _circle(0.56, 0.28, 0.190, c),
                _circle(0.56, 0.28, 0.064, (0, 0, 0))]
    return []                                           # Floor / Lava / EmptySpace: unrenderable upstream


_GEN_KIND = {"wall_rect": 0, "horz_wall": 1, "vert_wall": 2, "put": 3, "place": 4}


def reject_mask(cells, W, H):
    """place_obj(reject_fn=) as data: the rejected cells ((x, y), ...) -> uint8 mask indexed x*H + y"""
    m = np.zeros(W * H, np.uint8)
    for (x, y) in cells:
        m[x * H + y] = 1
    return m


def make_config(spec):
    cfg = Config()
    cfg._keep = []                    # the reject masks the struct points at
    cfg.W, cfg.H = spec["W"], spec["H"]
    agents = spec["agents"]
    cfg.n_agents = len(agents)
    cfg.view_size, cfg.tile_size = spec["view_size"], spec["tile_size"]
    cfg.view_offset = spec.get("view_offset", 0)
    cfg.see_through_walls = int(spec.get("see_through_walls", False))
    cfg.max_steps = spec.get("max_steps", 100)
    cfg.reward_decay = int(bool(spec.get("reward_decay", True)))
    gm = spec.get("ghost_mode", True)      # bit 1: `is not False` (base.py:541), bit 2: truthy (base.py:683)
    cfg.ghost_mode = (1 if gm is not False else 0) | (2 if gm else 0)
    cfg.respawn = int(bool(spec.get("respawn", False)))
    # place_obj(agent, **agent_spawn_kwargs) (base.py:411, 505, 643): top / size clamped as base.py:692-695
    sp = spec.get("agent_spawn", {})
    top = sp.get("top", (0, 0))
    top = (max(top[0], 0), max(top[1], 0))
    size = sp.get("size") or (cfg.W, cfg.H)
    cfg.spawn_x0, cfg.spawn_y0 = top
    cfg.spawn_x1, cfg.spawn_y1 = min(top[0] + size[0], cfg.W), min(top[1] + size[1], cfg.H)
    cfg.spawn_max_tries = int(max(1, min(sp.get("max_tries", 1e5), 1e5)))
    if sp.get("reject"):              # agent_spawn_kwargs['reject_fn'], tabulated: the rejected cells
        m = reject_mask(sp["reject"], cfg.W, cfg.H)
        cfg._keep.append(m)
        cfg.spawn_reject = m.ctypes.data
    cfg.agent_type_idx = TYPE_IDX["GridAgentInterface"]
    for k, a in enumerate(agents):
        cfg.agent_color_idx[k] = COLOR_TO_IDX[a["color"]]
        cfg.agent_rgb[k][:3] = list(COLORS[a["color"]])
        cfg.spawn_delay[k] = int(a.get("spawn_delay", 0))
        cfg.is_prestige[k] = int(a["color"] == "prestige")
        beta = a.get("prestige_beta", 0.95)
        cfg.prestige_beta[k] = 0.95 if beta > 1 else beta            # agents.py:54-56
        cfg.prestige_scale[k] = a.get("prestige_scale", 2)
        for tname in a.get("hide_item_types", []):       # item.type: class name, 'Agent' for agents
            cfg.hide_type_mask[k] |= (1 << 31) if tname == "Agent" else (1 << TYPE_IDX[tname])
    objs = spec["objects"]
    assert objs[0] is None and len(objs) <= MAX_OBJ
    cfg.n_obj = len(objs)

    def find(type_, color, state):
        for i, o in enumerate(objs):
            if o and o["type"] == type_ and o["color"] == color and o.get("state", 0) == state:
                return i
        return 0

    for i, o in enumerate(objs):
        if o is None:
            continue
        d = cfg.obj[i]
        t = o["type"]
        d.type_idx, d.color_idx, d.state = TYPE_IDX[t], COLOR_TO_IDX[o["color"]], o.get("state", 0)
        if t == "Door":
            is_open = int(d.state == DOOR_OPEN)
            d.can_overlap, d.can_pickup, d.see_behind = is_open, 0, is_open
            d.toggle_kind = 1
            if d.state == DOOR_CLOSED:
                d.toggle_next = find("Door", o["color"], DOOR_OPEN)
            elif d.state == DOOR_OPEN:
                d.toggle_next = find("Door", o["color"], DOOR_CLOSED)
            d.unlock_next = find("Door", o["color"], DOOR_CLOSED)
        else:
            d.can_overlap, d.can_pickup, d.see_behind = _PRED[t]
        if t == "Box":
            d.toggle_kind = 2
        if t == "Goal":                                  # objects.py:211-220; base.py:584
            d.reward_kind, d.reward, d.ends_episode = 1, float(o.get("reward", 1)), 1
        if t == "Lava":
            d.ends_episode = 1
        if t == "BonusTile":                             # objects.py:164-206
            d.reward_kind = 2
            d.reward, d.penalty = float(o["reward"]), float(o.get("penalty", -0.1))
            d.bonus_id, d.n_bonus = int(o.get("bonus_id", 0)), int(o.get("n_bonus", 1))
            d.initial_reward = int(bool(o.get("initial_reward", True)))
            d.reset_on_mistake = int(bool(o.get("reset_on_mistake", False)))
        d.is_key = int(t == "Key")
        ops = _sprite(o)
        d.n_fill = len(ops)
        for j, op in enumerate(ops):
            d.fill[j] = op
    cfg.wall_obj = spec["wall_obj"]
    for which, name in enumerate(("gen_ctor", "gen_reset")):
        prog = spec[name]
        assert len(prog) <= MAX_GEN
        cfg.n_gen[which] = len(prog)
        for j, g in enumerate(prog):
            op = cfg.gen[which][j]
            op.kind = _GEN_KIND[g[0]]
            if g[0] == "wall_rect":
                op.x, op.y, op.w, op.h = g[1:5]
            elif g[0] == "horz_wall":
                op.x, op.y, op.w = g[1:4]
            elif g[0] == "vert_wall":
                op.x, op.y, op.h = g[1:4]
            elif g[0] == "put":
                op.obj, op.x, op.y = g[1:4]
            elif g[0] == "place":
                op.obj, op.count, op.max_tries = g[1:4]
                if len(g) >= 8:                      # sampling rectangle [x0,x1) x [y0,y1)
                    op.x, op.y, op.w, op.h = g[4], g[5], g[6] - g[4], g[7] - g[5]
                if len(g) == 9 and g[8]:             # reject_fn, tabulated: the rejected cells ((x, y), ...)
                    m = reject_mask(g[8], cfg.W, cfg.H)
                    cfg._keep.append(m)
                    op.reject = m.ctypes.data
    return cfg


_ERR = {0: None, -1: ValueError, -2: RecursionError, -3: TypeError, -4: AssertionError, -5: ValueError, -6: AttributeError}


def _raise(rc):
    if rc != 0:
        raise _ERR[rc]("oracle rc=%d" % rc)


class OracleEnv(object):
    """One reference-semantics env.  `construct=True` replays the reference constructor's
    implicit `self.reset()` (base.py:369) with the ctor-time `_gen_grid` program."""

    def __init__(self, spec, seed=1337, construct=True, _like=None):
        self.spec = spec
        self.L = lib()
        self.cfg = make_config(spec) if _like is None else _like.cfg
        key = seed_key(seed)
        if _like is None:
            self.h = self.L.mgo_create(C.byref(self.cfg), _p(key, C.c_uint32), len(key))
        else:
            self.h = self.L.mgo_create_like(_like.h, _p(key, C.c_uint32), len(key))
        self.n = self.cfg.n_agents
        self.W, self.H = self.cfg.W, self.cfg.H
        self.vs, self.ts = self.cfg.view_size, self.cfg.tile_size
        self.P = self.vs * self.ts
        if construct:
            _raise(self.L.mgo_reset(self.h, 0))

    def __del__(self):
        try:
            if self.h:
                self.L.mgo_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def reset(self):
        _raise(self.L.mgo_reset(self.h, 1))
        return self.gen_obs()

    def step(self, actions, return_order=False):
        assert len(actions) == self.n                     # base.py:508
        a = np.ascontiguousarray(actions, dtype=np.int32)
        rew = np.zeros(self.n, np.float64)
        done = C.c_int32(0)
        order = np.zeros(self.n, np.int32)
        rc = self.L.mgo_step(self.h, _p(a, C.c_int32), _p(rew, C.c_double), C.byref(done), _p(order, C.c_int32))
        _raise(rc)
        out = (self.gen_obs(), rew, bool(done.value), {})
        return out + (order,) if return_order else out

    def gen_obs(self):
        obs = np.zeros((self.n, self.P, self.P, 3), np.uint8)
        for k in range(self.n):
            self.L.mgo_render_obs(self.h, k, _p(obs[k], C.c_uint8))
        return obs

    def view(self, k):
        vis = np.zeros((self.vs, self.vs), np.uint8)
        cells = np.zeros((self.vs, self.vs), np.int32)
        self.L.mgo_view(self.h, k, _p(vis, C.c_uint8), _p(cells, C.c_int32))
        return vis.astype(bool), cells

    def encode(self, vis_mask=None):
        out = np.zeros((self.W, self.H, 3), np.uint8)
        v = None if vis_mask is None else np.ascontiguousarray(vis_mask, dtype=np.uint8)
        self.L.mgo_encode(self.h, None if v is None else _p(v, C.c_uint8), _p(out, C.c_uint8))
        return out

    def tile(self, obj, agent_k=-1, agent_dir=0):
        out = np.zeros((self.ts, self.ts, 3), np.uint8)
        self.L.mgo_tile(self.h, obj, agent_k, agent_dir, _p(out, C.c_uint8))
        return out

    def state(self):
        base = np.zeros((self.W, self.H), np.uint8)
        ag = np.zeros((self.n, 7), np.int32)
        sc = C.c_int32(0)
        self.L.mgo_get_state(self.h, _p(base, C.c_uint8), _p(ag, C.c_int32), C.byref(sc))
        return dict(base=base, pos=ag[:, 0:2].copy(), dir=ag[:, 2].copy(), active=ag[:, 3].astype(bool),
                    done=ag[:, 4].astype(bool), carrying=ag[:, 5].copy(), ordinal=ag[:, 6].copy(),
                    step_count=sc.value)

    def prestige(self):
        out = np.zeros(self.n, np.float64)
        self.L.mgo_get_prestige(self.h, _p(out, C.c_double))
        return out

    def rich_obs(self, k):
        """the non-image fields of agent k's 'rich' observation (base.py:461-471)"""
        rew, pos, ori = C.c_double(0), np.zeros(2, np.float64), C.c_int32(0)
        self.L.mgo_rich_obs(self.h, k, C.byref(rew), _p(pos, C.c_double), C.byref(ori))
        return dict(reward=rew.value, position=pos, orientation=ori.value)

    def mt_state(self):
        mt = np.zeros(624, np.uint32)
        pos = C.c_int32(0)
        self.L.mgo_get_mt(self.h, _p(mt, C.c_uint32), C.byref(pos))
        return mt, pos.value

    def set_dir(self, k, d):
        self.L.mgo_set_agent_dir(self.h, k, d)

    def set_carrying(self, k, obj):
        self.L.mgo_set_carrying(self.h, k, obj)

    def regen_grid(self):
        """fresh `_gen_grid` of the reset program with agents lifted off (test scenes only; the
        caller re-places agents with place_agent_at)."""
        _raise(self.L.mgo_regen_grid(self.h, 1))

    def put_obj(self, obj, x, y):
        _raise(self.L.mgo_put_obj(self.h, obj, x, y))

    def place_obj(self, what, region=None, max_tries=100000, reject=None):
        """live place_obj: what >= 1 object id, what < 0 agent -(what+1); reject: rejected cells ((x, y), ...).
        Returns (x, y)."""
        x0, y0, x1, y1 = region or (0, 0, self.W, self.H)
        xy = np.zeros(2, np.int32)
        m = reject_mask(reject, self.W, self.H) if reject else None
        _raise(self.L.mgo_place_obj(self.h, what, x0, y0, x1, y1, int(max_tries),
                                    None if m is None else C.c_void_p(m.ctypes.data), _p(xy, C.c_int32)))
        return int(xy[0]), int(xy[1])

    def try_place_obj(self, what, x, y):
        return bool(self.L.mgo_try_place_obj(self.h, what, int(x), int(y)))

    def place_agent_at(self, k, x, y):
        _raise(self.L.mgo_place_agent_at(self.h, k, x, y))


class OracleEnvViews(object):
    """Agents with their own view geometry (agents.py:19-35).  Nothing in the state machine depends on how
    an agent sees (views only shape observations), so one OracleEnv per distinct geometry is stepped in
    lockstep on the same seed and actions, and agent k's observation is taken from the env that was built
    with k's geometry.  Same surface as OracleEnv; observations come back as a list of n arrays."""

    def __init__(self, spec, seed=1337):
        import copy
        self.spec = spec
        views = [a.get("view") or {k: spec[k] for k in ("view_size", "tile_size", "view_offset", "see_through_walls")}
                 for a in spec["agents"]]
        self.keys = []
        for v in views:
            key = (v["view_size"], v["tile_size"], v["view_offset"], bool(v["see_through_walls"]))
            if key not in self.keys:
                self.keys.append(key)
        self.owner = [self.keys.index((v["view_size"], v["tile_size"], v["view_offset"], bool(v["see_through_walls"])))
                      for v in views]
        self.envs = []
        for key in self.keys:
            sp = copy.deepcopy(spec)
            sp.update(view_size=key[0], tile_size=key[1], view_offset=key[2], see_through_walls=key[3])
            for a in sp["agents"]:
                a.pop("view", None)
            self.envs.append(OracleEnv(sp, seed))
        self.n = self.envs[0].n

    def _obs(self):
        per = [e.gen_obs() for e in self.envs]
        return [per[self.owner[k]][k] for k in range(self.n)]

    def gen_obs(self):
        return self._obs()

    def reset(self):
        for e in self.envs:
            e.reset()
        return self._obs()

    def step(self, actions, return_order=False):
        outs = [e.step(actions, return_order=return_order) for e in self.envs]
        return (self._obs(),) + tuple(outs[0][1:])

    def __getattr__(self, name):        # state / encode / mt_state / view ...: the state machine is shared
        return getattr(self.envs[0], name)


def make_env(spec, seed=1337):
    """OracleEnv, or OracleEnvViews when the scenario's agents carry their own views"""
    return OracleEnvViews(spec, seed) if any("view" in a for a in spec["agents"]) else OracleEnv(spec, seed=seed)


class OracleBatch(object):
    """B independent oracle envs stepped with OpenMP — bench.py's cpu_baseline leg and the
    batched parity tests."""

    def __init__(self, spec, seeds, construct=True):
        self.envs = []
        for s in seeds:
            self.envs.append(OracleEnv(spec, int(s), construct=construct,
                                       _like=self.envs[0] if self.envs else None))
        e0 = self.envs[0]
        self.B, self.n, self.P = len(self.envs), e0.n, e0.P
        self.L = e0.L
        self._h = (C.c_void_p * self.B)(*[e.h for e in self.envs])

    def reset(self):
        return np.stack([e.reset() for e in self.envs])

    def gen_obs(self):
        return np.stack([e.gen_obs() for e in self.envs])

    def step(self, actions, render=True, auto_reset=False, threads=0, reuse_obs=False):
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.B, self.n)
        rew = np.zeros((self.B, self.n), np.float64)
        done = np.zeros(self.B, np.uint8)
        obs = None
        if render:
            if reuse_obs:      # benchmark leg: no 50 MB allocation + page faults per step
                if getattr(self, "_obs_buf", None) is None:
                    self._obs_buf = np.zeros((self.B, self.n, self.P, self.P, 3), np.uint8)
                obs = self._obs_buf
            else:
                obs = np.zeros((self.B, self.n, self.P, self.P, 3), np.uint8)
        rc = self.L.mgo_batch_step(self._h, self.B, _p(a, C.c_int32), _p(rew, C.c_double), _p(done, C.c_uint8),
                                   None if obs is None else _p(obs, C.c_uint8), int(auto_reset), threads)
        _raise(rc)
        return obs, rew, done.astype(bool), {}

    def max_threads(self):
        return self.L.mgo_max_threads()


def occlude(transp, agent_pos):
    """agents.py:298-343 on a (vs,vs) transparency array indexed [i, j]."""
    t = np.ascontiguousarray(transp, dtype=np.uint8)
    vs = t.shape[0]
    out = np.zeros((vs, vs), np.uint8)
    lib().mgo_occlude(vs, agent_pos[0], agent_pos[1], _p(t, C.c_uint8), _p(out, C.c_uint8))
    return out.astype(bool)
