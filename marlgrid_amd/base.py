"""Batched multi-agent gridworld: the gym-style `MultiGridEnv.reset()/step()` surface of
kandouss/marlgrid (`marlgrid/base.py:334-708`) with a leading env-batch dimension B, the state in
HBM and every per-step operation a hand-written HIP kernel (libmarlgrid_hip.so, C ABI in
include/marlgrid_hip.h).  This module is host-side plumbing only: argument handling, the
`_gen_grid` recorder, table/atlas upload and kernel launches on torch's current stream.  There is
no CPU fallback.

State layout (all torch tensors on one MI355X; struct-of-arrays over B):
    grid_state   uint8  (B, cells_stride)   object id per cell, index x*H + y  (MultiGrid.grid[i, j])
    agent_state  int64  (B, n)              packed 8-byte agent records (MG_AG_* in the C header)
    mt_state     int32  (B, 624) + mt_pos   per-env MT19937, numpy RandomState stream (lazy form)
    mt_head      int32  (B, 16)             the next 16 outputs of that stream (what a step draws from)
    step_count   int32  (B,), done uint8 (B,), error int32 (B,)
    obs          uint8  (B, n, P, P, 3),   rewards float32 (B, n)

Differences from the reference that the batch forces (documented in DESIGN.md):
  * obs / rewards / done are tensors with a leading B (the reference returns a list of n arrays,
    a float64 ndarray and a bool); obs values are identical, dtype is uint8 as the declared
    observation space says (the reference actually returns int64 arrays, base.py:305);
  * `_gen_grid` is *recorded* once per reset (walls / put_obj -> static template, place_obj ->
    ordered rejection-sampling program) and replayed on the device for every env with that env's
    own RNG, so scenario subclasses written against the reference API keep working as long as
    their layout logic does not branch on random draws in Python;
  * per-env runtime errors (the reference's exceptions) are collected in `error` and raised as the
    matching Python exception after the step (strict=True) or on `check_errors()`.
"""
import ctypes as C
import functools
import os
import threading

import numpy as np

from . import _native as N
from . import rendering, seeding, spaces
from .agents import GridAgentInterface
from .objects import COLOR_TO_IDX, OBJECT_TYPES, BonusTile, Box, Door, Goal, Key, Wall, WorldObj

TILE_PIXELS = 32
DEFAULT_PLACE_OBS = "search"
STATE_DICT_VERSION = 3              # the C ABI version that last changed the MgState layout / RNG form the tensors follow

_trace = threading.local()


def _on_device(fn):
    """Run a method with the env's GPU as the current HIP device (kernels are launched on a stream of
    that device; a mismatch only happens when one process drives several GPUs)."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        if self._dry:
            return fn(self, *args, **kwargs)
        import torch
        if torch.cuda.current_device() == self.device.index:
            return fn(self, *args, **kwargs)
        with torch.cuda.device(self.device):
            return fn(self, *args, **kwargs)
    return wrapper


class ObjectRegistry(object):
    """object kind <-> uint8 id (the reference keeps instance <-> key maps per grid,
    base.py:19-64; here ids index the device object table and are stable for the env's life)."""

    def __init__(self, max_num_objects=N.MAX_OBJ):
        self.objs = [None]
        self.key_to_id = {}
        self.max_num_objects = max_num_objects
        self.version = 0

    def __len__(self):
        return len(self.objs)

    def get_key(self, obj):
        if obj is None:
            return 0
        if not isinstance(obj, WorldObj) or obj.is_agent:
            raise ValueError("only non-agent WorldObj instances can be put in the grid")
        k = obj.key()
        if k in self.key_to_id:
            return self.key_to_id[k]
        if len(self.objs) >= self.max_num_objects:
            raise ValueError("Object registry full.")
        self.objs.append(obj)
        self.key_to_id[k] = len(self.objs) - 1
        self.version += 1
        for r in obj.related():
            self.get_key(r)
        return self.key_to_id[k]

    def obj_of_key(self, key):
        return self.objs[int(key)]

    def find(self, obj):
        return self.key_to_id.get(obj.key(), 0)

    # the rest of the reference's registry interface (base.py:32-53)
    def get_next_key(self):
        if len(self.objs) >= self.max_num_objects:
            raise ValueError("Object registry full.")
        return len(self.objs)

    def add_object(self, obj):
        return self.get_key(obj)

    def contains_object(self, obj):
        return obj is None or (isinstance(obj, WorldObj) and obj.key() in self.key_to_id)

    def contains_key(self, key):
        return 0 <= int(key) < len(self.objs)


def rotate_grid(grid, rot_k):
    """base.py:67-80 on tensors whose last two dims are (x, y): np.rot90 with the args used for images"""
    rot_k = rot_k % 4
    if rot_k == 3:
        return grid.flip(-1).transpose(-1, -2)
    if rot_k == 1:
        return grid.flip(-2).transpose(-1, -2)
    if rot_k == 2:
        return grid.flip(-1).flip(-2)
    return grid


class MultiGrid(object):
    """The grid container (base.py:83-331).  Constructed inside `_gen_grid`, where it records the
    static part of the layout; afterwards it is a view on the env's HBM grid."""

    def __init__(self, shape, obj_reg=None, orientation=0):
        env = getattr(_trace, "env", None)
        if env is None:
            raise RuntimeError("MultiGrid(...) is constructed inside MultiGridEnv._gen_grid")
        if not isinstance(shape, tuple):
            raise ValueError("Must create grid from shape tuple.")
        self.width, self.height = shape
        if self.width < 3 or self.height < 3:
            raise ValueError("Grid needs width, height >= 3")
        if (self.width, self.height) != (env.width, env.height):
            raise ValueError("_gen_grid must build a grid of the env's own (width, height)")
        self.orientation = orientation
        self._env = env
        self.obj_reg = env.obj_reg
        self._template = np.zeros((self.width, self.height), np.uint8)
        # what `_gen_grid` itself has drawn so far — the template plus the static edits made after a place_obj (which go
        # into the reset program, not the template): what get() answers while `_gen_grid` runs
        self._shadow = np.zeros((self.width, self.height), np.uint8)
        # cells a recorded random placement may have filled since `_gen_grid` last drew on them (place_obj only takes empty
        # cells of its sampling rectangle; a later static edit of a cell decides it again)
        self._maybe = np.zeros((self.width, self.height), bool)
        env._tr_begin(self)

    # ---- layout recording / live edits --------------------------------------------------------
    def set(self, i, j, obj):
        assert i >= 0 and i < self.width
        assert j >= 0 and j < self.height
        env = self._env
        if env._tracing:
            key = self.obj_reg.get_key(obj)
            if env._tr_static(("put", key, int(i), int(j)), key, [(int(i), int(j), int(i) + 1, int(j) + 1)]):
                self._template[i, j] = key
            self._shadow[i, j] = key
            self._maybe[i, j] = False
        else:
            env.put_obj(obj, i, j)

    def horz_wall(self, x, y, length=None, obj_type=Wall):
        if length is None:
            length = self.width - x
        self._wall(("horz_wall", int(x), int(y), int(length)), [(x + i, y) for i in range(length)], obj_type,
                   [(x, y, x + length, y + 1)])

    def vert_wall(self, x, y, length=None, obj_type=Wall):
        if length is None:
            length = self.height - y
        self._wall(("vert_wall", int(x), int(y), int(length)), [(x, y + j) for j in range(length)], obj_type,
                   [(x, y, x + 1, y + length)])

    def wall_rect(self, x, y, w, h, obj_type=Wall):
        cells = ([(x + i, y) for i in range(w)] + [(x + i, y + h - 1) for i in range(w)]
                 + [(x, y + j) for j in range(h)] + [(x + w - 1, y + j) for j in range(h)])
        self._wall(("wall_rect", int(x), int(y), int(w), int(h)), cells, obj_type,
                   [(x, y, x + w, y + 1), (x, y + h - 1, x + w, y + h), (x, y, x + 1, y + h), (x + w - 1, y, x + w, y + h)])

    def _wall(self, sym, cells, obj_type, rects):
        env = self._env
        # (objects are value types here: every obj_type() of a helper is the same table row.  Walls of Walls are recorded
        # as what they are; a helper drawn with another type is, before the first place_obj, its cells one by one — the
        # template does not care — and after one a handful of rectangle fills in the reset program, not an op per cell)
        if env._tracing and (obj_type is Wall or env._tr_ops):
            key = self.obj_reg.get_key(obj_type())
            for (i, j) in cells:
                assert 0 <= i < self.width and 0 <= j < self.height
            if obj_type is not Wall:
                sym = [("put", key, int(i), int(j)) for (i, j) in dict.fromkeys(cells)]
            if env._tr_static(sym, key, [tuple(int(v) for v in r) for r in rects if r[2] > r[0] and r[3] > r[1]]):
                for (i, j) in cells:
                    self._template[i, j] = key
            for (i, j) in cells:
                self._shadow[i, j] = key
                self._maybe[i, j] = False
        else:
            for (i, j) in cells:
                self.set(i, j, obj_type())

    # ---- live views -----------------------------------------------------------------------------
    @property
    def grid(self):
        """(B, W, H) uint8 object ids (live HBM view)"""
        e = self._env
        return e.grid_state[:, :e.width * e.height].view(e.batch_size, e.width, e.height)

    def get(self, i, j, env=0):
        """the non-agent object kind in cell (i, j) of env `env` (host sync; debugging aid)"""
        assert i >= 0 and i < self.width
        assert j >= 0 and j < self.height
        if self._env._tracing:      # inside `_gen_grid`: what it has drawn so far (random placements are per env: not in here)
            if self._maybe[i, j]:
                # upstream's grid.get() here sees what an earlier place_obj put into THIS env's cell; the recorder cannot —
                # the draw happens later, on the device, per env.  Layout code that branches on the answer would be recorded
                # with a layout the reference would not produce: say so instead of answering "empty" silently.
                import warnings
                warnings.warn("marlgrid_amd: grid.get(%d, %d) inside _gen_grid reads a cell an earlier random place_obj may "
                              "have filled — placements are drawn per env on the device, the recorder answers with the static "
                              "layout only (%s); a layout that branches on this value is not recorded faithfully"
                              % (i, j, "empty" if self._shadow[i, j] == 0 else "the object drawn there"), RuntimeWarning, stacklevel=2)
            return self.obj_reg.obj_of_key(int(self._shadow[i, j]))
        return self.obj_reg.obj_of_key(int(self.grid[env, i, j].item()))

    def encode(self, vis_mask=None):
        """(B, W, H, 3) uint8 — batched MultiGrid.encode (base.py:196-214)"""
        return self._env._encode(vis_mask)

    # host-side (torch) counterparts of the array helpers gen_obs_grid is written with upstream; the
    # obs kernel does its own crop / rotate / shadow cast, these are for callers and for tests
    @property
    def opacity(self):
        """(B, W, H) bool: cells one cannot see through (base.py:103-106; empty cells and agents are transparent)"""
        import torch
        e = self._env
        opaque = torch.tensor([False] + [not o.see_behind() for o in self.obj_reg.objs[1:]], device=e.device)
        return opaque[self.grid.long()]

    def rotate_left(self, k=1):
        """(B, W', H') object ids of the whole grid rotated like base.py:115-120"""
        return rotate_grid(self.grid, k)

    def slice(self, topX, topY, width, height, rot_k=0):
        """(B, w, h) object ids of a window of every env, zero-padded outside the grid and rotated
        (base.py:123-147).  topX / topY: ints or (B,) tensors (e.g. columns of `agent.get_view_exts()`)."""
        import torch
        e = self._env
        g = self.grid
        B = g.shape[0]
        tx = torch.as_tensor(topX, device=e.device).long().expand(B)
        ty = torch.as_tensor(topY, device=e.device).long().expand(B)
        xs = tx[:, None] + torch.arange(width, device=e.device)[None, :]          # (B, w)
        ys = ty[:, None] + torch.arange(height, device=e.device)[None, :]         # (B, h)
        ok = ((xs >= 0) & (xs < self.width))[:, :, None] & ((ys >= 0) & (ys < self.height))[:, None, :]
        flat = xs.clamp(0, self.width - 1)[:, :, None] * self.height + ys.clamp(0, self.height - 1)[:, None, :]
        sub = torch.gather(g.reshape(B, -1), 1, flat.reshape(B, -1)).view(B, width, height)
        return rotate_grid(torch.where(ok, sub, torch.zeros_like(sub)), rot_k)

    @classmethod
    def decode(cls, array):
        raise NotImplementedError      # as upstream (base.py:216-218)


class _HostFlag(object):
    """MgState.error_flag: one int32 in pinned host memory that the device can write (mg_host_flag_alloc)"""

    def __init__(self, lib):
        self._lib = lib
        h, d = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        N.check(lib.mg_host_flag_alloc(C.byref(h), C.byref(d)))
        self._host = h
        self.dev = C.cast(d, C.c_void_p).value

    def raised(self):
        return self._host[0] != 0

    def clear(self):
        self._host[0] = 0

    def __del__(self):
        try:
            self._lib.mg_host_flag_free(self._host)
        except Exception:       # interpreter shutdown
            pass


class _LibBuffer(object):
    """Device memory straight from the driver (mg_obs_alloc: hipMalloc outside torch's caching allocator), handed
    to torch through the CUDA array interface: the tensor keeps this object alive, and the memory goes back to
    the driver — not into a cache — when the last view of it dies."""

    def __init__(self, lib, nbytes, device):
        self._lib, self.device, self.nbytes = lib, device, int(nbytes)
        self.ptr = lib.mg_obs_alloc(self.nbytes, device.index)
        self.ok = bool(self.ptr)
        if self.ok:
            self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False),
                                             "version": 2, "strides": None}

    def tensor(self, shape):
        import torch
        t = torch.as_tensor(self, device=self.device).view(shape)
        assert t.data_ptr() == self.ptr            # shares the memory (no copy)
        return t

    def __del__(self):
        if getattr(self, "ptr", None):
            try:
                import torch
                torch.cuda.synchronize(self.device)      # no launch may still be writing it
                self._lib.mg_obs_free(self.ptr)
            except Exception:   # interpreter shutdown
                pass
            self.ptr = None


class _PlacedBuffer(object):
    """An observation buffer mg_obs_place handed out (a window of a raw driver allocation chosen for where it lies in
    HBM), as torch sees it through the CUDA array interface.  The tensor keeps this object alive; with its last view the
    buffer goes back to the library (mg_obs_release), which remembers an arena of the fast class for the next env of the
    same size in this process."""

    def __init__(self, lib, ptr, nbytes, device):
        self._lib, self.device, self.nbytes, self.ptr = lib, device, int(nbytes), int(ptr)
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False),
                                         "version": 2, "strides": None}

    def tensor(self, shape):
        import torch
        t = torch.as_tensor(self, device=self.device).view(shape)
        assert t.data_ptr() == self.ptr            # shares the memory (no copy)
        return t

    def __del__(self):
        if getattr(self, "ptr", None):
            try:
                import torch
                with torch.cuda.device(self.device):
                    torch.cuda.synchronize(self.device)      # no launch may still be writing it
                    self._lib.mg_obs_release(self.ptr)
            except Exception:   # interpreter shutdown
                pass
            self.ptr = None


def release_obs_cache(device=None):
    """Return the observation arenas this process remembers (released buffers of the fast class, kept for the next env
    of the same size: mg_obs_release) to the driver.  `device`: a torch device / index, or None for every device.
    Returns how many there were."""
    idx = -1
    if device is not None:
        import torch
        idx = torch.device(device).index if not isinstance(device, int) else device
        idx = torch.cuda.current_device() if idx is None else idx
    return N.lib().mg_obs_trim(idx)


_GENERIC_WARNED = set()


def _warn_generic_kernel(g, generic, prestige):
    """mg_render_kernel_name's bits 1 | 2: view size AND tile size are run-time values of the launch — the fully generic
    instantiation of the observation kernel (assemble-and-stream raster, run-time dividers, a 15-row shadow cast, 8- / 4-wave
    workgroups), measured 2-2.4x slower than a specialised one (profiles/r05: view 9 at 5-pixel tiles 0.3136 -> 0.1360 ms
    when it got its own).  Said once per configuration and process, naming what is specialised nearby."""
    if generic != 3:
        return
    key = (g.view_size, g.tile_size, prestige, g.kernel_name)
    if key in _GENERIC_WARNED:
        return
    _GENERIC_WARNED.add(key)
    import warnings
    if g.kernel_name.endswith(", 3>"):      # the grid-in-place variant: nothing about the view or the tiles would change it
        warnings.warn("marlgrid_amd: this grid does not fit the observation kernel's LDS next to its per-cell agent maps (grids up "
                      "to ~140 x 140, ~110 x 110 with hide_item_types, are staged there): it is read in place by the fully run-time "
                      "instantiation (%s) — correct, every size up to 255 x 255, but slower per observation byte than a staged grid"
                      % g.kernel_name, RuntimeWarning, stacklevel=4)
        return
    near = []
    if prestige:
        near.append("'prestige' agents: view_size 7 with any view_tile_size (5, 8 and 11 fastest), or view_tile_size 8 / 16 / 32 with any view")
    else:
        if g.view_size > 9:
            near.append("view_size %d with view_tile_size 8, 16 or 32%s" % (g.view_size, ", or 5" if g.view_size in (11, 13, 15) else ""))
        near.append("view_size 3 ... 9 with this view_tile_size (%d)" % g.tile_size)
        if g.view_size not in (11, 13, 15):
            near.append("view_size 11 / 13 / 15 with view_tile_size 5")
    warnings.warn("marlgrid_amd: view_size=%d, view_tile_size=%d%s takes the fully run-time instantiation of the observation "
                  "kernel (%s): correct, but 2-2.4x slower than a specialised one.  Specialised nearby: %s."
                  % (g.view_size, g.tile_size, " with 'prestige' agents" if prestige else "", g.kernel_name, "; ".join(near)),
                  RuntimeWarning, stacklevel=4)


class _ViewGroup(object):
    """agents that share one view geometry: their launch config, atlas and observation buffers"""

    def __init__(self, key, members):
        self.key, self.members = key, members            # (view_size, tile_size, view_offset, see_through), agent ids
        self.view_size, self.tile_size, self.view_offset, self.see_through_walls = key
        self.pixels = self.view_size * self.tile_size
        self.cfg = self.atlas = self.atlas_dev = None
        self.ring = []                                   # obs tensors (B, n_g, P, P, 3), one per buffer set
        self.obs = None


class MultiGridEnv(object):
    """See module docstring.  Constructor keeps the reference's kwargs (base.py:335-347) and adds
    `batch_size`, `device`, `seeds`, `auto_reset`, `strict`."""

    metadata = {}

    def __init__(self, agents=[], grid_size=None, width=None, height=None, max_steps=100,
                 reward_decay=True, seed=1337, respawn=False, ghost_mode=True, agent_spawn_kwargs={},
                 batch_size=1, device=None, seeds=None, auto_reset=False, strict=True, obs_buffers=2,
                 fused_step=True, place_obs=True, encode_in_step=False, _dry=False):
        if grid_size is not None:
            assert width is None and height is None
            width, height = grid_size, grid_size
        self.respawn = respawn
        self.window = None
        self.width, self.height = int(width), int(height)
        self.max_steps = int(max_steps)
        self.reward_decay = reward_decay
        self.agent_spawn_kwargs = agent_spawn_kwargs
        self.ghost_mode = ghost_mode
        self.batch_size = int(batch_size)
        self.auto_reset = bool(auto_reset)
        # Per-env runtime errors (the reference's exceptions).  strict=True: raised without giving up the
        # asynchronous launch queue — the kernels raise a one-word flag in host-mapped memory that every
        # step()/reset() polls (no stream synchronize), so the exception surfaces at the next call into the env
        # after the GPU got there, at the latest at check_errors().  strict="sync": raised by the very step() that
        # caused it, as upstream does (one host synchronize per step).  strict=False: only check_errors() raises.
        if strict not in (True, False, "sync"):
            raise ValueError("strict must be True, False or 'sync'")
        self.strict = strict
        self.obs_buffers = max(1, int(obs_buffers))
        self.fused_step = bool(fused_step)     # step() = one launch (mg_step_render) instead of mg_step + mg_render_obs
        # encode_in_step=True: every step() (and reset()) also leaves `grid.encode()` of the batch — (B, W, H, 3) uint8,
        # base.py:196-214 — in `self.grid_encoding`, written by the step's own launch where it can (mg_step_render_encode:
        # +2.4 % bytes instead of a second launch), by an mg_encode launch behind it otherwise
        self.encode_in_step = bool(encode_in_step)
        self.grid_encoding = None
        self._enc_fused = True                 # until the library says MG_E_UNSUPPORTED for this configuration
        # where the observation buffers live: "search" (= True) picks the fastest of a bounded set of candidate
        # allocations by timing the raster itself into each (_place_obs_buffers -> mg_obs_place: <= 2 s, candidates <=
        # min(a quarter of the free memory, 32 GiB)); "thorough": the long search (a second pass, larger candidates, one
        # big allocate-and-free: for a process that owns the GPU); False = plain torch allocations
        if place_obs is True:
            place_obs = DEFAULT_PLACE_OBS
        # ... or a dict of the search's own knobs: {"budget": bytes of live candidates, "seconds": per pass, "thorough": bool,
        # "stir": bool, "reuse": bool, "share": processes that share this device's memory — each counts on 1 / share of what
        # is free} (what `_place_obs_buffers` takes)
        self._place_kw = {}
        if isinstance(place_obs, dict):
            unknown = set(place_obs) - {"budget", "seconds", "thorough", "stir", "reuse", "min_bytes", "share"}
            if unknown:
                raise ValueError("place_obs: unknown keys %s" % sorted(unknown))
            self._place_kw = dict(place_obs)
            place_obs = "thorough" if place_obs.get("thorough") else "search"
        if place_obs not in ("search", "thorough", False):
            raise ValueError("place_obs must be 'search' (= True), 'thorough', a dict of budgets, or False")
        self.place_obs = place_obs
        self._dry = bool(_dry)
        if self.batch_size < 1:
            raise ValueError("batch_size must be >= 1")
        if self.width > 255 or self.height > 255:
            raise ValueError("grid dimensions are limited to 255")
        self.cells_stride = (self.width * self.height + 15) // 16 * 16

        self.agents = []
        for agent in agents:
            self.add_agent(agent)
        self._check_agents()
        for k, agent in enumerate(self.agents):
            agent._bind(self, k)         # agent.pos / .dir / .done / ... become batched views of this env

        self.obj_reg = ObjectRegistry()
        self._tables_version = -1
        self._settings_seen = None
        self._tracing = False
        self._prog_cache = {}
        self._spec_ctor = None
        self._spec_last = None
        self._probe = None
        self.grid = None

        if not self._dry:
            import torch
            if not torch.cuda.is_available():
                raise RuntimeError("marlgrid_amd needs an MI355X (HIP device): there is no CPU fallback")
            self.device = torch.device(device if device is not None else "cuda")
            if self.device.type != "cuda":
                raise RuntimeError("marlgrid_amd runs on HIP devices only (got %r)" % (device,))
            if self.device.index is None:
                self.device = torch.device("cuda", torch.cuda.current_device())
            self._lib = N.lib()
            self._alloc_state()
        self._seeds_arg = seeds
        self.seed(seed=seed)
        self.reset()
        self.obs_placement = []
        if not self._dry and self.place_obs in ("search", "thorough"):
            self._place_obs_buffers(**self._place_kw)
        self._spec_ctor = self._spec_last     # the constructor-time `_gen_grid` (base.py:369)
        self._retrace = True

    @classmethod
    def pipelined(cls, agents, parts=2, batch_size=1, seed=1337, device=None, streams=None, **kwargs):
        """The batch as `parts` envs of this class on `parts` streams (marlgrid_amd.sharding.ShardPipeline): what a
        double-buffered sampler steps in turn so that the launches of independent shards overlap.  Part 0 is built
        with `agents` themselves, the other parts with shallow copies (a learner's networks and buffers are shared,
        the binding to an env — agent.pos, .dir, ... — is per part)."""
        import copy
        from .sharding import ShardPipeline
        agents = list(agents)
        made = []

        def make_part(batch_size, seeds, device):
            team = agents if not made else [copy.copy(a) for a in agents]
            made.append(True)
            return cls(agents=team, batch_size=batch_size, seeds=seeds, device=device, **kwargs)
        return ShardPipeline(make_part, batch_size, parts=parts, seed=seed, device=device, streams=streams)

    # ---- configuration ----------------------------------------------------------------------------
    def add_agent(self, agent_interface):
        if isinstance(agent_interface, dict):
            self.agents.append(GridAgentInterface(**agent_interface))
        elif isinstance(agent_interface, GridAgentInterface):
            self.agents.append(agent_interface)
        else:
            raise ValueError(
                "To add an agent to a marlgrid environment, call add_agent with either a "
                "GridAgentInterface object or a dictionary that can be used to initialize one.")

    def _check_agents(self):
        n = len(self.agents)
        if n < 1 or n > N.MAX_AGENTS:
            raise ValueError("the batched engine supports 1..%d agents (got %d)" % (N.MAX_AGENTS, n))
        # Every agent carries its own view (agents.py:19-35).  Agents that share (view_size, view_tile_size,
        # view_offset, see_through_walls) form a VIEW GROUP: one (B, n_g, P, P, 3) observation tensor and one
        # raster launch per group.  One group — every shipped scenario — is the fast path (one tensor, the
        # env step fused into the raster launch); with several, reset()/step() return the per-agent list the
        # reference returns.
        self._groups = []
        for k, a in enumerate(self.agents):
            if a.allow_negative_prestige:
                raise NotImplementedError("allow_negative_prestige=True raises AttributeError upstream "
                                          "(agents.py:147-148) and is not supported")
            if a.view_size < 3:
                raise ValueError("Grid needs width, height >= 3")      # what upstream's first observation raises (base.py:97-99)
            if a.view_size > N.MAX_VIEW:
                raise NotImplementedError("view_size must be <= %d" % N.MAX_VIEW)
            if not (0 <= a.view_offset < a.view_size):
                raise ValueError("view_offset out of range")
            key = (a.view_size, a.view_tile_size, a.view_offset, bool(a.see_through_walls))
            for g in self._groups:
                if g.key == key:
                    g.members.append(k)
                    break
            else:
                self._groups.append(_ViewGroup(key, [k]))
        self._hetero = len(self._groups) > 1
        self._prestige = [a.color == "prestige" for a in self.agents]
        self._all_image = all(a.observation_style == "image" for a in self.agents)
        # the first group's view parameters double as "the env's" (what uniform envs have always exposed)
        self.view_size, self.tile_size, self.view_offset, self.see_through_walls = self._groups[0].key
        self.obs_pixels = self.view_size * self.tile_size

    @property
    def num_agents(self):
        return len(self.agents)

    @property
    def kernel_name(self):
        """the instantiation of the observation kernel a step launches (first view group), as the launcher names it
        (mg_render_kernel_name) and as rocprofv3 prints it"""
        self._sync_tables()
        return self._groups[0].kernel_name

    @property
    def action_space(self):
        return spaces.Tuple([agent.action_space for agent in self.agents])

    @property
    def observation_space(self):
        return spaces.Tuple([agent.observation_space for agent in self.agents])

    # ---- device state -------------------------------------------------------------------------------
    def _alloc_state(self):
        import torch
        B, n, dev = self.batch_size, self.num_agents, self.device
        P = self.obs_pixels
        with torch.cuda.device(dev):
            self.grid_state = torch.zeros((B, self.cells_stride), dtype=torch.uint8, device=dev)
            self.agent_state = torch.zeros((B, n), dtype=torch.int64, device=dev)
            self.mt_state = torch.zeros((B, N.MT_N), dtype=torch.int32, device=dev)
            self.mt_pos = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.mt_head = torch.zeros((B, N.MT_HEAD), dtype=torch.int32, device=dev)
            self.step_count_t = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.error_t = torch.zeros((B,), dtype=torch.int32, device=dev)
            # agent.prestige (agents.py:141-153): only needed when some agent's colour is 'prestige'
            self.prestige_t = torch.zeros((B, n), dtype=torch.float64, device=dev) if any(self._prestige) else None
            # step() outputs rotate through `obs_buffers` buffer sets (default 2), so what one step
            # returned stays intact while the next step is computed — the README loop
            # `save_step(obs, act, next_obs, rew, done)` sees two different observations.
            for g in self._groups:
                g.shape = (B, len(g.members), g.pixels, g.pixels, 3)
                g.ring = [torch.zeros(g.shape, dtype=torch.uint8, device=dev) for _ in range(self.obs_buffers)]
                g.obs = g.ring[0]
            self._ring = [dict(obs=self._groups[0].ring[i],
                               rewards=torch.zeros((B, n), dtype=torch.float32, device=dev),
                               done=torch.zeros((B,), dtype=torch.uint8, device=dev))
                          for i in range(self.obs_buffers)]
            self._ring_i = 0
            self.obs, self.rewards, self.done_t = (self._ring[0][k] for k in ("obs", "rewards", "done"))
            self.done_b = self.done_t.view(torch.bool)      # the same bytes, as the bool tensor step() returns
            # one word of pinned host memory the kernels raise when they record an error in error_t: polled by
            # step()/reset() instead of a nonzero() + .item() round trip through the stream (mapped while the
            # env's device is current: the device pointer is this device's view of the word)
            self._flag = _HostFlag(self._lib)
        self._launch_streams = {}       # streams this env has launched on since the last check_errors()
        self._state = N.State(self.grid_state.data_ptr(), self.agent_state.data_ptr(), self.mt_state.data_ptr(),
                              self.mt_pos.data_ptr(), self.step_count_t.data_ptr(), self.done_t.data_ptr(),
                              self.error_t.data_ptr(),
                              self.prestige_t.data_ptr() if self.prestige_t is not None else None,
                              self.mt_head.data_ptr(), self._flag.dev)

    @_on_device
    def _place_obs_buffers(self, budget=0, seconds=0.0, thorough=None, stir=None, reuse=True, min_bytes=0, gain=0.0, max_candidates=0,
                           slow_alloc=0.0, stir_cap=0, iters=0, share=1, fast_rate=0.0):
        """place_obs="search" (default) / "thorough": choose WHERE in HBM the observation buffers live — mg_obs_place
        (include/marlgrid_hip.h; marlgrid_amd/csrc/mg_place_obs.hip) does the work, this is its caller.

        What is measured (MI355X, profiles/r04/README.md section 1, profiles/r05/README.md): the raster's write pattern —
        thousands of concurrent sequential streams — runs at 5.3 TB/s into a buffer that lies inside one of the driver's
        physical blocks and at the speed of a dense fill (6.9 TB/s) into one that straddles the boundary between two blocks:
        0.153 instead of 0.193 ms for the bench workload's 925 MB.  The library constructs candidates on such boundaries
        (one hipMalloc of 3 P bytes, P = the power of two >= half the buffer; the buffer = the window centred on the
        2 P | P junction), times the raster into each and keeps the fastest `obs_buffers` once they run 12 % under the median.

        What it costs (the defaults are meant to be survivable for whoever shares the GPU): candidates alive at any time
        <= min(a quarter of the free memory, 32 GiB) — for buffers of several GB raised to the kept buffers plus three
        candidates, while that is within half of the free memory (a 16 GB buffer's candidates are 24 GiB each: round 5's
        flat 32 GiB could not hold one) —; 2 s at most (0.33 s per GiB of candidate above 6 GiB, 12 s at most); a kept buffer
        pins its whole candidate — 1.5 .. 3 x its size, outside torch's allocator (`env.obs_placement[g]["pinned_bytes"]`);
        buffers under 256 MiB are plain torch allocations; at most MG_PLACE_MAX (8) buffers of a ring are placed (the
        rest stay torch allocations); `share` = k: k processes share this device, each counts on 1 / k of what is free.  "thorough" adds what round 4 found necessary on memory nobody had allocated before: larger block
        pairs (a kept buffer may then pin up to 12 x its size), a second pass, one big allocate-and-free (half of the
        free memory, 64 GiB at most) that mixes the driver's free lists, and larger budgets (candidates up to half of the
        free memory / 128 GiB, 6 s per pass).  A released buffer of the fast class is
        remembered by the library: the next env of the same size in this process takes it without a search
        (`release_obs_cache()` gives them back).  Without a candidate in the fast class the best seen is kept and
        `found` is False: try again later, or with place_obs="thorough"."""
        import torch
        if thorough is None:
            thorough = self.place_obs == "thorough"
        if stir is None:
            stir = thorough
        flags = (N.PLACE_THOROUGH if thorough else 0) | (N.PLACE_STIR if stir else 0) | (0 if reuse else N.PLACE_NO_REUSE)
        share = max(1, int(share))
        if thorough and not budget:         # the long search is for a process that owns its GPU: half of what is free, 6 s per pass
            budget = min(torch.cuda.mem_get_info(self.device)[0] // (2 * share), 128 << 30)
        if thorough and not seconds:
            seconds = 6.0
        tn = N.PlaceTuning(gain=gain, slow_alloc_s_per_gib=slow_alloc, min_bytes=min_bytes, stir_bytes=stir_cap,
                           max_candidates=max_candidates, iters=iters, share=share, fast_rate=fast_rate)
        threshold = min_bytes or (256 << 20)
        any_replaced = False
        previous = list(getattr(self, "obs_placement", None) or [])
        self.obs_placement = []
        for gi, g in enumerate(self._groups):
            nbytes = g.ring[0].numel()
            if nbytes < threshold:
                self.obs_placement.append(None)
                continue
            # (a call places at most MG_PLACE_MAX buffers: a longer ring keeps torch allocations for the rest — the ring
            # rotates through all of them, `kept` says which are placed)
            keep = min(len(g.ring), N.PLACE_MAX)
            out = (C.c_void_p * keep)()
            st = N.PlaceStats()
            rc = self._lib.mg_obs_place(C.byref(g.cfg), C.byref(self._state), keep, int(budget), float(seconds), flags, C.byref(tn),
                                        out, C.byref(st), self._stream())
            if rc == N.E_NOMEM:
                # nothing was handed out: the ring stays what it was — torch allocations, or the buffers an earlier call
                # placed (whose record then stays too: they are still pinned)
                was = previous[gi] if gi < len(previous) else None
                failed = {"found": False, "stopped": N.PLACE_STOP.get(st.stopped, str(st.stopped)) or "out of memory", "kept": [],
                          "candidates": st.candidates, "seconds": st.seconds, "buffer_bytes": nbytes, "pinned_bytes": 0}
                if was and was.get("pinned_bytes"):
                    g.placement_ms = dict(was, retry_failed=failed)
                else:
                    g.placement_ms = failed
                self.obs_placement.append(g.placement_ms)
                continue
            N.check(rc)
            g.ring = [_PlacedBuffer(self._lib, out[i], nbytes, self.device).tensor(g.shape) for i in range(keep)] + list(g.ring[keep:])
            for t in g.ring[:keep]:
                t.zero_()
            g.obs = g.ring[self._ring_i]
            any_replaced = True
            g.placement_ms = {"found": bool(st.found), "reused": st.reused, "kept": [st.kept_ms[i] for i in range(keep)],
                              "candidates": st.candidates, "windows_measured": st.windows, "passes": st.passes,
                              "stopped": N.PLACE_STOP.get(st.stopped, str(st.stopped)), "seconds": st.seconds,
                              "median_ms": st.median_ms, "all": [st.all_ms[i] for i in range(min(st.candidates, N.PLACE_ALL))],
                              "buffer_bytes": st.buffer_bytes, "candidate_bytes": st.candidate_bytes,
                              "window_offset": ((2 * (st.candidate_bytes // 3) - nbytes // 2) & ~4095) if st.candidate_bytes else None,
                              "kept_window_offsets": [st.window_offset[i] for i in range(keep)],
                              "pinned_bytes": st.pinned_bytes, "budget_bytes": st.budget_bytes, "block_pair_level": st.level,
                              "plain_stage": bool(st.plain_stage), "stirred": ({"bytes": st.stirred_bytes} if st.stirred_bytes else None),
                              "placed_buffers": keep, "ring_buffers": len(g.ring), "share": share,
                              "alloc_ms_per_GiB": 1e3 * st.alloc_seconds / max(st.alloc_bytes, 1) * (1 << 30)}
            self.obs_placement.append(g.placement_ms)
            if not st.found and not thorough and st.stopped != 7:       # (7: not bound by HBM writes — nothing to find, nothing to say)
                import warnings
                warnings.warn("marlgrid_amd: the bounded placement search found no observation buffer in the fast class for %d-byte "
                              "buffers (%d candidates, stopped: %s): if this configuration's raster is bound by HBM writes it may run up to "
                              "20 %% below its best — place_obs='thorough' searches with larger budgets (env.obs_placement has the "
                              "numbers; a configuration that is not HBM-bound has no fast class to find)"
                              % (nbytes, st.candidates, g.placement_ms["stopped"]), RuntimeWarning, stacklevel=3)
        for i, r in enumerate(self._ring):
            r["obs"] = self._groups[0].ring[i]
        self.obs = self._ring[self._ring_i]["obs"]
        # (the torch-allocated ring tensors that were replaced are unreferenced now; their blocks stay in torch's caching
        # allocator for the caller's next allocations — this package does not empty a cache it does not own)
        if any_replaced:
            self._render()                  # the current observation, into the buffer that is current now

    def _stream(self):
        """torch's current stream of the env's device, as the C ABI takes it — and remembered: check_errors()
        has to wait for the launches on EVERY stream this env was driven on (a ShardPipeline part is stepped
        under its own stream and checked from wherever the caller happens to be)."""
        import torch
        s = torch.cuda.current_stream(self.device)
        self._launch_streams[s.cuda_stream] = s
        return C.c_void_p(s.cuda_stream)

    @_on_device
    def seed(self, seed=1337):
        """Seed every env's RNG (base.py:371-374).  Env b gets `seed + b` unless explicit per-env
        `seeds` were given to the constructor."""
        if self._seeds_arg is not None:
            sa = self._seeds_arg
            if hasattr(sa, "reshape"):                 # ndarray / tensor
                sa = sa.reshape(-1).tolist()
            seeds = [int(s) for s in sa]               # python ints: seeds may exceed int64
            if len(seeds) != self.batch_size:
                raise ValueError("seeds must have batch_size entries")
        else:
            seeds = [int(seed) + b for b in range(self.batch_size)]
        self.seeds = seeds
        if not self._dry:
            import torch
            keys, lens = seeding.batch_keys(seeds)
            k = torch.from_numpy(keys.view(np.int32)).to(self.device)
            kl = torch.from_numpy(lens).to(self.device)
            N.check(self._lib.mg_mt_seed(self.batch_size, k.data_ptr(), kl.data_ptr(), self.mt_state.data_ptr(),
                                         self.mt_pos.data_ptr(), self.mt_head.data_ptr(), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()    # k / kl die here
        return [seed]

    # ---- `_gen_grid` recording ----------------------------------------------------------------------
    def _gen_grid(self, width, height):
        raise NotImplementedError("subclasses build self.grid here (see marlgrid_amd/envs)")

    def _tr_begin(self, grid):
        if not self._tracing:
            raise RuntimeError("MultiGrid(...) is constructed inside _gen_grid")
        self._tr_grid = grid
        self._tr_sym = []
        self._tr_ops = []
        self._tr_late = {}          # index into _tr_ops -> the symbolic form of a static edit recorded as op(s)

    def _tr_static(self, sym, key, rects):
        """A static layout edit of `_gen_grid` (grid.set / put_obj / the wall helpers).  Before the first random
        place_obj it goes into the template (returns True: the caller writes the template); AFTER one — upstream's
        `_gen_grid` is free Python, envs/cluttered.py:25-36 could as well put its goal last — it is recorded in the
        ordered program as fill ops (max_tries 0: write `key` into every cell of a rectangle, replacing what a
        placement put there, base.py:655-662) and replayed per env between the placements (returns False)."""
        if not self._tr_ops:
            self._tr_sym.extend(sym if isinstance(sym, list) else [sym])
            return True
        first = None                # the op this edit's first rectangle went into: where its symbolic form is listed
        for (x0, y0, x1, y1) in rects:
            # a fill that continues the one before it (same object, same rows or columns, adjacent) is the same op, larger
            last = self._tr_ops[-1]
            merged = None
            if last[2] == 0 and last[0] == int(key) and last[7] is None:
                lx0, ly0, lx1, ly1 = last[3:7]
                if (ly0, ly1) == (y0, y1) and (lx1 == x0 or x1 == lx0):
                    merged = (int(key), 1, 0, min(lx0, x0), y0, max(lx1, x1), y1, None)
                elif (lx0, lx1) == (x0, x1) and (ly1 == y0 or y1 == ly0):
                    merged = (int(key), 1, 0, x0, min(ly0, y0), x1, max(ly1, y1), None)
            if merged is not None:
                self._tr_ops[-1] = merged
            else:
                self._tr_ops.append((int(key), 1, 0, x0, y0, x1, y1, None))
            if first is None:
                first = len(self._tr_ops) - 1
        if first is not None:
            self._tr_late.setdefault(first, []).extend(sym if isinstance(sym, list) else [sym])
        return False

    def _trace_gen_grid(self):
        self._tracing = True
        self._tr_grid = None
        _trace.env = self
        try:
            self._gen_grid(self.width, self.height)
        finally:
            _trace.env = None
            self._tracing = False
        g = self._tr_grid
        if g is None or self.grid is not g:
            raise RuntimeError("_gen_grid must assign self.grid = MultiGrid((width, height))")
        if len(self._tr_ops) > N.MAX_GEN:
            fills = sum(1 for op in self._tr_ops if op[2] == 0)
            raise NotImplementedError("_gen_grid records %d reset-program ops — %d groups of random placements and %d rectangle "
                                      "fills for static edits made after the first place_obj — and a device program holds at most "
                                      "%d (MG_MAX_GEN: a sanity bound — every env replays the whole program at every reset): draw the "
                                      "static layout before the first place_obj (it then costs nothing)"
                                      % (len(self._tr_ops), len(self._tr_ops) - fills, fills, N.MAX_GEN))
        self._spec_last = dict(sym=list(self._tr_sym), ops=list(self._tr_ops), late=dict(self._tr_late))
        return g._template, list(self._tr_ops)

    @_on_device
    def put_obj(self, obj, i, j, env_mask=None):
        """Put an object at a specific position, replacing what is there (base.py:655-662) — an agent standing
        there included: as upstream it is in no cell afterwards (nobody sees it; it keeps its position, turns
        and looks), and its next successful forward move raises what upstream raises (MG_AF_EVICTED)."""
        if self._tracing:
            self._tr_grid.set(i, j, obj)
            return True
        assert 0 <= i < self.width and 0 <= j < self.height
        key = self.obj_reg.get_key(obj)
        if not self._dry:
            self._sync_tables()
            N.check(self._lib.mg_put_obj(C.byref(self._cfg), C.byref(self._state), key, int(i), int(j),
                                         self._mask_ptr(env_mask), self._stream()))
        return True

    def _table_dev(self, table):
        """a (W, H) uint8 table as the device layout the kernels index (x*H + y, cells_stride bytes)"""
        import torch
        t = np.zeros(self.cells_stride, np.uint8)
        t[:self.width * self.height] = np.asarray(table, np.uint8).reshape(-1)
        return torch.from_numpy(t).to(self.device)

    def _place_region(self, top, size):
        # sampling rectangle, clamped exactly like base.py:692-695
        top = (0, 0) if top is None else (max(int(top[0]), 0), max(int(top[1]), 0))
        if size is None:
            size = (self.width, self.height)
        x1, y1 = min(top[0] + int(size[0]), self.width), min(top[1] + int(size[1]), self.height)
        if x1 <= top[0] or y1 <= top[1]:
            raise ValueError("place_obj: empty sampling rectangle")
        return top[0], top[1], x1, y1

    def _what(self, obj):
        if isinstance(obj, GridAgentInterface):
            return -(self.agents.index(obj) + 1)
        return self.obj_reg.get_key(obj)

    @_on_device
    def _place_live(self, obj, top, size, max_tries, env_mask, reject_fn=None):
        """place_obj on the live grids: per-env rejection sampling on each env's RNG.  Returns the
        chosen positions (B, 2) int32 (-1, -1 where it failed: RecursionError on check_errors())."""
        import torch
        what = self._what(obj)
        self._sync_tables()
        x0, y0, x1, y1 = self._place_region(top, size)
        pos = torch.empty((self.batch_size, 2), dtype=torch.int32, device=self.device)
        rej = self._reject_table(reject_fn, (x0, y0, x1, y1))
        rej_dev = None if rej is None else self._table_dev(rej)
        N.check(self._lib.mg_place(C.byref(self._cfg), C.byref(self._state), what, x0, y0, x1, y1,
                                   int(max(1, min(max_tries, 1e5))), None, self._mask_ptr(env_mask),
                                   None if rej_dev is None else C.c_void_p(rej_dev.data_ptr()),
                                   pos.data_ptr(), None, self._stream()))
        if self.strict:
            self.check_errors()
        return pos

    def _reject_table(self, reject_fn, region):
        """place_obj(reject_fn=) as data: the reference calls `reject_fn(pos)` with the drawn position and nothing
        else (base.py:700-701) — a function of the position alone is a table.  Evaluated once over the sampling
        rectangle (the only positions that can be drawn): (W, H) uint8, 1 = rejected; None without a callback.
        (A callback that looks at anything else — per-env state, a counter — cannot be batched and is not
        supported: it would see one call per cell here, not one per draw.)"""
        if reject_fn is None:
            return None
        x0, y0, x1, y1 = region
        t = np.zeros((self.width, self.height), np.uint8)
        for x in range(x0, x1):
            for y in range(y0, y1):
                t[x, y] = 1 if reject_fn(np.array([x, y])) else 0
        return t

    def place_obj(self, obj, top=None, size=None, reject_fn=None, max_tries=1e5, env_mask=None):
        """Rejection-sample a free cell for `obj` (base.py:690-708).  Inside `_gen_grid` this records
        one placement; the draw happens on the device, per env.  `reject_fn(pos)` is tabulated over the
        sampling rectangle once (see `_reject_table`)."""
        if not self._tracing:
            return self._place_live(obj, top, size, max_tries, env_mask, reject_fn)
        if isinstance(obj, GridAgentInterface):
            raise NotImplementedError("inside _gen_grid agents are placed by reset() itself")
        max_tries = int(max(1, min(max_tries, 1e5)))
        key = self.obj_reg.get_key(obj)
        region = self._place_region(top, size)
        rej = self._reject_table(reject_fn, region)
        op = (key, 1, max_tries) + region + (None if rej is None else rej.tobytes(),)
        x0, y0, x1, y1 = region
        may = self._tr_grid._shadow[x0:x1, y0:y1] == 0          # (only empty cells accept a placement, base.py:672-679)
        if rej is not None:
            may &= rej[x0:x1, y0:y1] == 0
        self._tr_grid._maybe[x0:x1, y0:y1] |= may
        if self._tr_ops and self._tr_ops[-1][0] == key and self._tr_ops[-1][2:] == op[2:] and self._tr_ops[-1][2] > 0:
            self._tr_ops[-1] = (key, self._tr_ops[-1][1] + 1) + op[2:]
        else:
            self._tr_ops.append(op)
        return None

    @_on_device
    def try_place_obj(self, obj, pos, env_mask=None):
        """Try to place `obj` (an object, or one of this env's agents) at `pos` — one (x, y) for every
        env or a (B, 2) tensor — and return a (B,) bool tensor saying where it worked (base.py:664-688)."""
        import torch
        if self._tracing:
            raise NotImplementedError("inside _gen_grid use put_obj / place_obj")
        what = self._what(obj)
        self._sync_tables()
        p = torch.as_tensor(pos, dtype=torch.int32, device=self.device)
        if p.dim() == 1:
            p = p.expand(self.batch_size, 2)
        p = p.contiguous()
        ok = torch.zeros((self.batch_size,), dtype=torch.uint8, device=self.device)
        N.check(self._lib.mg_place(C.byref(self._cfg), C.byref(self._state), what, 0, 0, self.width, self.height, 1,
                                   p.data_ptr(), self._mask_ptr(env_mask), None, None, ok.data_ptr(), self._stream()))
        return ok.bool()

    def place_agents(self, top=None, size=None, rand_dir=True, max_tries=1000):
        pass    # deprecated no-op upstream as well (base.py:710-712)

    # ---- tables: object descriptors + atlas -----------------------------------------------------------
    def _obj_table(self, tile_size=None):
        tile_size = self.tile_size if tile_size is None else tile_size
        objs = self.obj_reg.objs
        tab = (N.ObjDesc * len(objs))()
        for i, o in enumerate(objs):
            if o is None:
                continue
            d = tab[i]
            t, c, s = o.encode()
            d.type_idx, d.color_idx, d.state = t, c, s
            f = 0
            f |= N.OF_CAN_OVERLAP if o.can_overlap() else 0
            f |= N.OF_CAN_PICKUP if o.can_pickup() else 0
            f |= N.OF_SEE_BEHIND if o.see_behind() else 0
            f |= N.OF_ENDS_EPISODE if o.ends_episode else 0
            f |= N.OF_IS_KEY if isinstance(o, Key) else 0
            f |= N.OF_IS_BOX if isinstance(o, Box) else 0
            if isinstance(o, Door):
                f |= N.OF_IS_DOOR
                if o.state == Door.LOCKED:
                    f |= N.OF_DOOR_LOCKED
                d.unlock_next = self.obj_reg.find(Door(o.color, Door.CLOSED))
                d.toggle_next = self.obj_reg.find(Door(o.color, Door.OPEN if o.state == Door.CLOSED else Door.CLOSED))
            d.flags = f
            try:    # "a corner of the plain tile is black" (the border rule of render_tile, base.py:296-298)
                plain = rendering.render_sprite(o.sprite_ops(), tile_size)
            except NotImplementedError:
                plain = np.zeros((tile_size, tile_size, 3), np.uint8)
            d.flags2 = int((plain[[0, 0, -1, -1], [0, -1, 0, -1]] == 0).all(axis=-1).any())
            if isinstance(o, Goal):
                d.reward_kind, d.reward = 1, float(o.reward)
            elif isinstance(o, BonusTile):
                d.reward_kind, d.reward, d.penalty = 2, float(o.reward), float(o.penalty)
                d.bonus_id, d.n_bonus = o.bonus_id, o.n_bonus
                d.bonus_flags = (1 if o.initial_reward else 0) | (2 if o.reset_on_mistake else 0)
        return tab

    def _host_tables(self, group=None):
        """Everything the kernels need that is derived on the host from the object registry and the agent
        interfaces, for one view group (default: the first): (cfg without device pointers, object table
        bytes, atlas bytes, atlas array).  Pure host work (no device access) — the upload is `_sync_tables`."""
        grp = self._groups[0] if group is None else group
        objs = self.obj_reg.objs
        atlas, ovl_slot, n_slots = rendering.build_atlas(objs, [a.color for a in self.agents], grp.tile_size,
                                                         prestige_sprites=any(self._prestige))
        tab = self._obj_table(grp.tile_size)
        for i in range(len(objs)):
            tab[i].ovl_slot = ovl_slot[i]
        raw = np.frombuffer(bytes(tab), dtype=np.uint8).copy()
        flat = atlas.reshape(-1)
        pad = (-flat.size) % 16
        flat = np.concatenate([flat, np.zeros(pad + 16, np.uint8)])
        cfg = N.Config()
        cfg.B, cfg.W, cfg.H, cfg.n_agents = self.batch_size, self.width, self.height, self.num_agents
        cfg.view_size, cfg.tile_size = grp.view_size, grp.tile_size
        cfg.view_offset, cfg.see_through_walls = grp.view_offset, int(grp.see_through_walls)
        if self._hetero:                       # this group's agents only are rendered with this geometry
            cfg.n_view = len(grp.members)
            for i, k in enumerate(grp.members):
                cfg.view_agent[i] = k
        cfg.cells_stride = self.cells_stride
        cfg.n_obj, cfg.n_ovl_slots, cfg.n_tiles = len(objs), n_slots, atlas.shape[1]
        cfg.agent_type_idx = OBJECT_TYPES.index(GridAgentInterface)
        for k, a in enumerate(self.agents):
            cfg.agent_color_idx[k] = COLOR_TO_IDX[a.color]
        # 'prestige' agents (agents.py:92-119, 141-153): per-env recoloured sprites
        for k, a in enumerate(self.agents):
            if self._prestige[k]:
                cfg.prestige_mask |= 1 << k
        if cfg.prestige_mask:
            cfg.prestige_sprite_tile = atlas.shape[1] - 4
            for d in range(4):      # max alpha of the (white) sprite per dir = max of blend_tiles' alpha map
                cfg.prestige_amax[d] = int(atlas[0, atlas.shape[1] - 4 + d][..., 0].max())
        # hide_item_types (base.py:441-449): per object id the agents that hide its type (bit k = agent k; a device table,
        # uploaded by _sync_tables), and the agents that hide 'Agent'
        hide_by = np.zeros(len(objs), np.uint32)
        for k, a in enumerate(self.agents):
            for i, o in enumerate(objs):
                if o is not None and o.type in a.hide_item_types:
                    hide_by[i] |= np.uint32(1 << k)
            if "Agent" in a.hide_item_types:
                cfg.hide_agent_mask |= 1 << k
        grp.hide_by = hide_by if hide_by.any() else None
        cfg.any_hide = int(cfg.hide_agent_mask != 0 or bool(hide_by.any()))
        self._refresh_cfg(cfg)
        return cfg, raw, flat, atlas

    def _refresh_cfg(self, cfg):
        """The scalar settings the reference reads on every step (max_steps, reward_decay, ghost_mode,
        respawn, agent_spawn_kwargs, the agents' spawn_delay / prestige parameters): copied into the
        launch config before every launch, so changing the attribute between steps takes effect on the
        next step, as upstream."""
        cfg.max_steps, cfg.reward_decay = int(self.max_steps), int(bool(self.reward_decay))
        # upstream tests `ghost_mode is False` when moving (base.py:541) but `not ghost_mode` when placing
        # (base.py:683): they differ for falsy non-False values such as 0 or None
        cfg.ghost_mode = (1 if self.ghost_mode is not False else 0) | (2 if self.ghost_mode else 0)
        cfg.respawn = int(bool(self.respawn))
        # place_obj(agent, **agent_spawn_kwargs) (base.py:411, 505, 643)
        kw = dict(self.agent_spawn_kwargs or {})
        reject_fn = kw.pop("reject_fn", None)
        top, size, max_tries = kw.pop("top", None), kw.pop("size", None), kw.pop("max_tries", 1e5)
        if kw:
            raise TypeError("place_obj() got an unexpected keyword argument %r" % sorted(kw)[0])
        cfg.spawn_x0, cfg.spawn_y0, cfg.spawn_x1, cfg.spawn_y1 = self._place_region(top, size)
        # agent_spawn_kwargs['reject_fn'] (base.py:411, 505, 643 -> :700-701): tabulated like place_obj's
        self._spawn_reject = self._reject_table(reject_fn, (cfg.spawn_x0, cfg.spawn_y0, cfg.spawn_x1, cfg.spawn_y1))
        if self._spawn_reject is not None and not self._dry:
            if getattr(self, "_spawn_reject_key", None) != self._spawn_reject.tobytes():
                self._spawn_reject_dev = self._table_dev(self._spawn_reject)
                self._spawn_reject_key = self._spawn_reject.tobytes()
            cfg.spawn_reject = self._spawn_reject_dev.data_ptr()
        else:
            cfg.spawn_reject = None
        cfg.spawn_max_tries = int(max(1, min(max_tries, 1e5)))
        for k, a in enumerate(self.agents):
            if a.spawn_delay < 0:
                raise ValueError("spawn_delay must be >= 0")
            cfg.spawn_delay[k] = int(a.spawn_delay)
            cfg.prestige_beta[k], cfg.prestige_scale[k] = float(a.prestige_beta), float(a.prestige_scale)
        cfg.any_spawn_delay = int(any(a.spawn_delay != 0 for a in self.agents))

    def _settings_key(self):
        """the attributes _refresh_cfg reads, as one comparable value (what step() checks instead of re-deriving
        every launch config on every step)"""
        kw = self.agent_spawn_kwargs

        def plain(v):       # array-likes by value (an ndarray has no truth value to compare with), callables by identity
            if callable(v):
                return ("fn", id(v))
            if isinstance(v, (list, tuple)) or hasattr(v, "tolist"):
                return tuple(np.asarray(v).reshape(-1).tolist())
            return v
        # the DERIVED values where upstream distinguishes more than == does: `ghost_mode is False` (base.py:541)
        # against `not ghost_mode` (base.py:683) — False and 0 compare equal and configure different engines
        return (int(self.max_steps), bool(self.reward_decay), self.ghost_mode is not False, bool(self.ghost_mode),
                bool(self.respawn), bool(self.auto_reset),
                tuple(sorted((k, plain(v)) for k, v in kw.items())) if kw else (),
                tuple((a.spawn_delay, a.prestige_beta, a.prestige_scale) for a in self.agents))

    def _sync_tables(self):
        """Make the launch configs current: rebuild and upload the object table / atlas of every view group
        when a new object kind was registered since the last launch, refresh the scalar settings always."""
        if self._dry:
            return
        if self._tables_version == self.obj_reg.version:
            key = self._settings_key()
            if key != self._settings_seen:
                for g in self._groups:
                    self._refresh_cfg(g.cfg)
                self._settings_seen = key
            return
        import torch
        for g in self._groups:
            cfg, raw, flat, atlas = self._host_tables(g)
            # the obs kernel keeps 4 waves of per-env scratch (and the atlas, when it fits) in one workgroup's
            # LDS: ask the library, which owns that layout, before anything is uploaded
            need = N.lib().mg_render_obs_lds_bytes(C.byref(cfg))
            if need < 0 or need > 160 * 1024:
                raise NotImplementedError(
                    "this configuration needs %d KiB of LDS per workgroup for 4 waves of per-env scratch; the obs "
                    "kernel has 160 KiB — with 'prestige' agents reduce their number or the tile size (every one of them has "
                    "its four recoloured sprites there); otherwise the number of agents / the view size" % (need // 1024))
            g.obj_dev = torch.from_numpy(raw).to(self.device)
            # the gather instantiations (raster 2: 5-, 6-, 7-, 9- ... 12-pixel tiles; examples/human_player.py's shape) take the atlas
            # in the raster's own LDS layout when the host has it ready — behind the plain one, `atlas_gather_off` bytes on —: building
            # it per workgroup was 5 us of human_player's 130 us launch (20 instructions per dword, 45 dwords per thread), 1 - 1.6 %
            # of the others'
            cfg.atlas_gather_off = 0
            name0 = N.render_kernel_name(cfg)[0]
            if name0.endswith(", 2>") and not os.environ.get("MG_NO_GATHER_ATLAS"):
                ts, seg = g.tile_size, 3 * g.tile_size
                rs = (16 + seg + 3) // 4 * 4
                rows = np.ascontiguousarray(flat[:4 * cfg.n_tiles * ts * seg]).reshape(-1, seg)
                padded = np.zeros((rows.shape[0], rs), np.uint8)
                padded[:, 16:16 + seg] = rows
                pb = np.concatenate([padded.reshape(-1), np.zeros(32, np.uint8)])
                pb = np.concatenate([pb, np.zeros((-pb.size) % 16, np.uint8)])
                off = (flat.size + 15) // 16 * 16
                flat = np.concatenate([flat, np.zeros(off - flat.size, np.uint8), pb])
                cfg.atlas_gather_off = off
            g.atlas_dev = torch.from_numpy(flat).to(self.device)
            g.atlas = atlas
            cfg.obj, cfg.atlas = g.obj_dev.data_ptr(), g.atlas_dev.data_ptr()
            g.hide_dev = None if g.hide_by is None else torch.from_numpy(g.hide_by.view(np.int32)).to(self.device)
            cfg.hide_by_obj = None if g.hide_dev is None else g.hide_dev.data_ptr()
            g.cfg = cfg
            # which instantiation of the observation kernel the launcher picks for this group (what rocprofv3 will call the
            # launch) — and a warning, once per configuration, when it is the fully generic one
            g.kernel_name, generic = N.render_kernel_name(cfg)
            _warn_generic_kernel(g, generic, bool(cfg.prestige_mask))
        g0 = self._groups[0]
        self._cfg, self.atlas, self._obj_dev, self._atlas_dev = g0.cfg, g0.atlas, g0.obj_dev, g0.atlas_dev
        self._tables_version = self.obj_reg.version
        self._settings_seen = self._settings_key()

    def _program(self, template, ops):
        import torch
        key = (template.tobytes(), tuple(ops))
        prog = self._prog_cache.get(key)
        if prog is not None:
            return prog
        t = np.zeros(self.cells_stride, np.uint8)
        t[:self.width * self.height] = template.reshape(-1)
        prog = N.GenProgram()
        # the struct carries a raw device pointer: the tensor it points at is kept on the struct
        # itself, so it lives exactly as long as any holder of the program (cache, `_reset_prog`)
        prog._template_dev = torch.from_numpy(t).to(self.device)
        prog.template_grid = prog._template_dev.data_ptr()
        prog.n_ops = len(ops)
        tables = []                   # place_obj(reject_fn=) tables, one row of cells_stride bytes each
        host_ops = (N.GenOp * max(1, len(ops)))()
        for i, (obj, count, max_tries, x0, y0, x1, y1, rej) in enumerate(ops):
            # (the library cannot look into device memory from the host: what the kernels rely on is checked here)
            assert (1 if max_tries > 0 else 0) <= obj < len(self.obj_reg.objs) and count >= 0 and max_tries >= 0
            assert 0 <= x0 < x1 <= self.width and 0 <= y0 < y1 <= self.height
            o = host_ops[i]
            o.obj, o.count, o.max_tries, o.x0, o.y0, o.x1, o.y1 = obj, count, max_tries, x0, y0, x1, y1
            o.reject = -1
            if rej is not None:
                row = np.zeros(self.cells_stride, np.uint8)
                row[:self.width * self.height] = np.frombuffer(rej, np.uint8)
                o.reject = len(tables)
                tables.append(row)
        # the ops live in device memory (MgGenProgram::ops), kept alive on the struct like the template
        prog._ops_dev = torch.from_numpy(np.frombuffer(bytes(host_ops), np.uint8).copy()).to(self.device)
        prog.ops = prog._ops_dev.data_ptr()
        prog.n_reject = len(tables)
        if tables:
            prog._reject_dev = torch.from_numpy(np.stack(tables)).to(self.device)
            prog.reject = prog._reject_dev.data_ptr()
        self._prog_cache[key] = prog
        return prog

    def _mask_ptr(self, env_mask):
        if env_mask is None:
            return None
        import torch
        m = torch.as_tensor(env_mask, device=self.device)
        if m.shape != (self.batch_size,):
            raise ValueError("env_mask must have shape (batch_size,)")
        self._mask_keep = m.to(torch.uint8).contiguous()
        return C.c_void_p(self._mask_keep.data_ptr())

    # ---- the gym surface -----------------------------------------------------------------------------
    @_on_device
    def reset(self, env_mask=None, **kwargs):
        """Start a new episode in every env (or those selected by `env_mask`) and return the
        observation tensor (B, n, P, P, 3) — base.py:402-416."""
        template, ops = self._trace_gen_grid()
        if self._dry:
            self._dry_trace = (template, ops)     # host-only instance: the recorded layout, nothing runs
            return None
        self._sync_tables()
        prog = self._program(template, ops)
        self._reset_prog, self._retrace = prog, False
        N.check(self._lib.mg_reset(C.byref(self._cfg), C.byref(self._state), C.byref(prog),
                                   self._mask_ptr(env_mask), self._stream()))
        self._render()
        if self.encode_in_step:
            self._encode_into(self._encoding_buffer())
        if self.strict:
            self.check_errors()
        return self._package_obs()

    @_on_device
    def step(self, actions):
        """actions: (B, n) integer tensor (device preferred) or array-like -> (obs, rewards, done, {})
        with obs (B, n, P, P, 3) uint8, rewards (B, n) float32, done (B,) bool — base.py:501-653."""
        import torch
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions))
        if actions.dim() == 1 and self.batch_size == 1:
            actions = actions.unsqueeze(0)
        assert actions.dim() == 2 and actions.shape[1] == len(self.agents) and actions.shape[0] == self.batch_size, \
            "actions must have shape (batch_size, n_agents)"                            # base.py:508
        if actions.dtype not in (torch.int64, torch.int32, torch.uint8):
            actions = actions.to(torch.int64)
        if actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(self.device).contiguous()
        if self.strict and self._flag.raised():
            self.check_errors()   # an earlier launch recorded an error: raise it now (this is the only host sync)
        self._sync_tables()       # re-derives the launch configs only if a setting or the object registry changed
        if self.obs_buffers > 1:
            self._ring_i = (self._ring_i + 1) % self.obs_buffers
            r = self._ring[self._ring_i]
            self.obs, self.rewards, self.done_t = r["obs"], r["rewards"], r["done"]
            for g in self._groups:
                g.obs = g.ring[self._ring_i]
            self.done_b = self.done_t.view(torch.bool)
            self._state.done = self.done_t.data_ptr()
        stream = self._stream()
        prog = None
        if self.auto_reset:
            # envs that finish in this step start their next episode inside the same launch, before the
            # obs is rendered (their returned obs is the first obs of the new episode; `done` still
            # reports the end): the lane that computes an env's done flag runs its reset — no host sync,
            # no second launch.
            if self._retrace:
                # first step after construction: subclass constructors finish configuring the
                # scenario after the base constructor's reset (cluttered.py:13-20)
                template, ops = self._trace_gen_grid()
                self._sync_tables()
                self._reset_prog = self._program(template, ops)
                self._retrace = False
            prog = C.byref(self._reset_prog)
        probe = self._probe          # bench.py: records an event on the launch stream around each launch
        if probe is not None:
            probe(0)
        # obs / rewards / done are views of the current buffer set (see `obs_buffers`)
        done = self.done_b
        if self.fused_step and not self._hetero:
            # the whole step — action loop, reset of finished episodes, observation raster — is ONE launch:
            # the wave that renders an env steps it first
            enc_rc = N.E_UNSUPPORTED
            if self.encode_in_step and self._enc_fused:
                enc_rc = self._lib.mg_step_render_encode(C.byref(self._cfg), C.byref(self._state), actions.data_ptr(),
                                                         actions.element_size(), self.rewards.data_ptr(), prog,
                                                         self.obs.data_ptr(), self._encoding_buffer().data_ptr(), stream)
                if enc_rc == N.E_UNSUPPORTED:      # (nothing was launched: more than 256 object ids + agent marks, or a grid read in place)
                    self._enc_fused = False
                else:
                    N.check(enc_rc)
            if enc_rc == N.E_UNSUPPORTED:
                N.check(self._lib.mg_step_render(C.byref(self._cfg), C.byref(self._state), actions.data_ptr(),
                                                 actions.element_size(), self.rewards.data_ptr(), prog,
                                                 self.obs.data_ptr(), stream))
                if self.encode_in_step:
                    self._encode_into(self._encoding_buffer())
        else:
            N.check(self._lib.mg_step(C.byref(self._cfg), C.byref(self._state), actions.data_ptr(),
                                      actions.element_size(), self.rewards.data_ptr(), prog, stream))
            if probe is not None:
                probe(1)
            for g in self._groups:       # one raster launch per view group (one group unless the agents' views differ)
                N.check(self._lib.mg_render_obs(C.byref(g.cfg), C.byref(self._state), g.obs.data_ptr(), None,
                                                None, None, stream))
            if self.encode_in_step:
                self._encode_into(self._encoding_buffer())
        if probe is not None:
            probe(2)
        if self.strict == "sync":
            self.check_errors()
        return self._package_obs(), self.rewards, done, {}

    def _render(self, debug=False, group=None):
        import torch
        self._sync_tables()
        if debug:
            g = self._groups[0] if group is None else group
            B, nv, vs = self.batch_size, len(g.members), g.view_size
            cells = torch.zeros((B, nv, vs, vs), dtype=torch.uint8, device=self.device)
            shown = torch.zeros_like(cells)
            vis = torch.zeros_like(cells)
            N.check(self._lib.mg_render_obs(C.byref(g.cfg), C.byref(self._state), g.obs.data_ptr(),
                                            cells.data_ptr(), shown.data_ptr(), vis.data_ptr(), self._stream()))
            return cells, shown, vis
        for g in self._groups:
            N.check(self._lib.mg_render_obs(C.byref(g.cfg), C.byref(self._state), g.obs.data_ptr(), None, None,
                                            None, self._stream()))
        return None

    @_on_device
    def gen_obs(self):
        """(B, n, P, P, 3) uint8, or the per-agent list for 'rich' agents (base.py:473-474)"""
        self._render()
        return self._package_obs()

    def _package_obs(self):
        """What reset()/step() hand back: the (B, n, P, P, 3) tensor when every agent uses the 'image'
        style (the fast path), else — like the reference's list of per-agent observations — a list
        with one entry per agent: its (B, P, P, 3) view or, for 'rich' agents, a dict of batched
        fields (base.py:459-471)."""
        if self._all_image and not self._hetero:
            return self.obs
        return [self._agent_obs(k) for k in range(self.num_agents)]

    @_on_device
    def gen_agent_obs(self, agent):
        """agent: a GridAgentInterface of this env or its index -> (B, P, P, 3) (base.py:453-471)"""
        k = agent if isinstance(agent, int) else self.agents.index(agent)
        self._render()
        return self._agent_obs(k)

    def _view_slot(self, k):
        """(view group, index inside it) of agent k"""
        for g in self._groups:
            if k in g.members:
                return g, g.members.index(k)
        raise IndexError(k)

    def _agent_obs(self, k):
        a = self.agents[k]
        g, slot = self._view_slot(k)
        pov = g.obs[:, slot]
        if a.observation_style == "image":
            return pov
        import torch
        ret = {"pov": pov}
        if a.observe_rewards:
            ret["reward"] = torch.zeros(self.batch_size, device=self.device)   # always 0 upstream (base.py:464,519)
        if a.observe_position:
            # np.array(pos) / np.array([W, H], dtype=float), `pos is None -> (0, 0)`: float64 like upstream
            wh = torch.tensor([self.width, self.height], dtype=torch.float64, device=self.device)
            pos = self.agent_pos[:, k].to(torch.float64)
            placed = (self.agent_flags[:, k] & N.AF_PLACED) != 0
            ret["position"] = torch.where(placed[:, None], pos, torch.zeros_like(pos)) / wh
        if a.observe_orientation:
            ret["orientation"] = self.agent_dir[:, k]
        return ret

    @_on_device
    def gen_obs_grid(self, agent):
        """(view cells (B, vs, vs) object ids indexed [i, j], visibility mask (B, vs, vs) bool) for one
        agent — base.py:418-451"""
        k = agent if isinstance(agent, int) else self.agents.index(agent)
        g, slot = self._view_slot(k)
        cells, _shown, vis = self._render(debug=True, group=g)
        return cells[:, slot], vis[:, slot].bool()

    @_on_device
    def _encode(self, vis_mask=None):
        import torch
        self._sync_tables()
        out = torch.empty((self.batch_size, self.width, self.height, 3), dtype=torch.uint8, device=self.device)
        vm = None
        if vis_mask is not None:
            vm = torch.as_tensor(vis_mask, device=self.device).to(torch.uint8).expand(
                self.batch_size, self.width, self.height).contiguous()
        return self._encode_into(out, vm)

    def _encode_into(self, out, vm=None):
        N.check(self._lib.mg_encode(C.byref(self._cfg), C.byref(self._state),
                                    None if vm is None else C.c_void_p(vm.data_ptr()), out.data_ptr(), self._stream()))
        return out

    def _encoding_buffer(self):
        """`grid_encoding` (encode_in_step): ONE tensor for the env's life — every step overwrites it"""
        if self.grid_encoding is None:
            import torch
            self.grid_encoding = torch.empty((self.batch_size, self.width, self.height, 3), dtype=torch.uint8, device=self.device)
        return self.grid_encoding

    @_on_device
    def check_errors(self):
        """Raise the reference's exception for the first env that hit one.  Host sync: waits for the launches
        queued so far, then reads ONE word (the error flag the kernels raise); the per-env codes are only
        fetched when it is set."""
        import torch
        torch.cuda.current_stream(self.device).synchronize()
        for s in list(self._launch_streams.values()):      # every stream a launch of this env went to
            s.synchronize()
        self._launch_streams.clear()
        if not self._flag.raised():
            return
        self._flag.clear()
        bad = torch.nonzero(self.error_t)
        if bad.numel():
            b = int(bad[0].item())
            code = int(self.error_t[b].item())
            self.error_t.zero_()
            msgs = {N.ERR_VALUE: "Environment can't handle action (env %d)." % b,
                    N.ERR_RECURSION: "Rejection sampling failed in place_obj (env %d)." % b,
                    N.ERR_TYPE: "toggle() takes 1 positional argument but 3 were given (Box, env %d)" % b,
                    N.ERR_ASSERT: "grid access out of bounds, or an agent left a cell it is not in (env %d)" % b,
                    N.ERR_ATTRIBUTE: "'NoneType' object has no attribute 'can_overlap' (env %d)" % b}
            raise N.ERR_EXC[code](msgs[code])

    def check_agent_position_integrity(self, title=""):
        """base.py:479-499 checks that every agent is in the grid exactly once (stack handling is
        error-prone there).  The packed records cannot express an agent in two places; what can be
        checked is that they are consistent: placed agents are inside the grid on a cell an agent
        can stand on, stack ranks are a permutation, active implies placed.  Raises AssertionError
        (the reference drops into pdb)."""
        import torch
        pos, fl, rk = self.agent_pos, self.agent_flags, self.agent_rank
        placed = (fl & N.AF_PLACED) != 0
        ok = ~placed | ((pos[..., 0] < self.width) & (pos[..., 1] < self.height))
        cell = (pos[..., 0] * self.height + pos[..., 1]).clamp(max=self.width * self.height - 1)
        base = torch.gather(self.grid_state[:, :self.width * self.height].long(), 1, cell)
        overlap = torch.tensor([True] + [bool(o.can_overlap()) for o in self.obj_reg.objs[1:]], device=self.device)
        ok &= ~placed | overlap[base]
        ok &= ((fl & N.AF_ACTIVE) == 0) | placed
        ok &= (rk.sort(dim=1).values == torch.arange(self.num_agents, device=self.device)).all(dim=1, keepdim=True)
        if not bool(ok.all()):
            bad = torch.nonzero(~ok.all(dim=1))[:8, 0].tolist()
            raise AssertionError("%s > Failed integrity test! envs %s" % (title, bad))

    # ---- agent state views (per-env state lives in HBM, not on the agent objects) -------------------------
    def _rec_byte(self, i):
        return ((self.agent_state >> (8 * i)) & 0xFF)

    @property
    def agent_pos(self):
        """(B, n, 2) int64"""
        import torch
        return torch.stack([self._rec_byte(N.AG_X), self._rec_byte(N.AG_Y)], dim=-1)

    @property
    def agent_dir(self):
        return self._rec_byte(N.AG_DIR)

    @property
    def agent_flags(self):
        return self._rec_byte(N.AG_FLAGS)

    @property
    def agent_active(self):
        return (self.agent_flags & N.AF_ACTIVE) != 0

    @property
    def agent_done(self):
        return (self.agent_flags & N.AF_DONE) != 0

    @property
    def agent_carrying(self):
        return self._rec_byte(N.AG_CARRY)

    @property
    def agent_rank(self):
        return self._rec_byte(N.AG_RANK)

    @property
    def step_count(self):
        return self.step_count_t

    def set_agent(self, k, x=None, y=None, dir=None, carrying=None, env_mask=None):
        """Test / scenario-building helper: overwrite fields of agent k's record in the selected envs.
        Moving an agent this way gives it the highest arrival rank (as a fresh placement would)."""
        import torch
        m = torch.ones(self.batch_size, dtype=torch.bool, device=self.device) if env_mask is None else \
            torch.as_tensor(env_mask, device=self.device).bool()
        rec = self.agent_state

        def setb(col, i, v):
            return (col & ~(0xFF << (8 * i))) | (int(v) << (8 * i))
        if x is not None or y is not None:
            n = self.num_agents
            old = self._rec_byte(N.AG_RANK)[:, k:k + 1]
            rk = self._rec_byte(N.AG_RANK)
            rk = torch.where(rk > old, rk - 1, rk)
            rk[:, k] = n - 1
            new = (rec & ~(0xFF << (8 * N.AG_RANK))) | (rk << (8 * N.AG_RANK))
            rec[m] = new[m]
        col = rec[:, k].clone()
        if x is not None:
            col = setb(col, N.AG_X, x)
        if y is not None:
            col = setb(col, N.AG_Y, y)
        if dir is not None:
            col = setb(col, N.AG_DIR, dir % 4)
        if carrying is not None:
            col = setb(col, N.AG_CARRY, self.obj_reg.get_key(carrying) if isinstance(carrying, WorldObj) else carrying)
        rec[:, k] = torch.where(m, col, rec[:, k])

    # ---- RNG state exchange with numpy (tests / checkpoints) -------------------------------------------
    def numpy_rng_state(self, b=0):
        """env b's generator as `np.random.RandomState.get_state()` would report it: (key[624], pos).
        The device keeps the *lazy* form plus a look-ahead head (see mg_core.h): words below `mt_pos`
        are already regenerated, the head holds the 16 outputs before it.  numpy regenerates whole
        blocks, so the not-yet-regenerated tail is advanced here; when the head reaches across a block
        boundary the words regenerated ahead of numpy's position are wound back (the twist is
        invertible except for the low 31 bits of word 0, which MT19937 never reads: reported as 0)."""
        mt = [int(v) for v in self.mt_state[b].cpu().numpy().view(np.uint32)]
        G = int(self.mt_pos[b].item())
        return seeding.numpy_form(mt, G, N.MT_HEAD)

    _STATE_KEYS = ("grid_state", "agent_state", "mt_state", "mt_pos", "mt_head", "step_count_t", "done_t", "error_t")

    def state_dict(self):
        """The env batch's whole state as tensors (a checkpoint).  `version` names the layout: the packed agent
        records and the RNG in its lazy form with the look-ahead head (mt_pos counts the words generated INTO the
        head, not the words consumed) — an older checkpoint is a different RNG stream position, not this one."""
        import torch
        sd = {k: getattr(self, k).clone() for k in self._STATE_KEYS}
        if self.prestige_t is not None:
            sd["prestige_t"] = self.prestige_t.clone()
        sd["version"] = torch.tensor(STATE_DICT_VERSION)
        return sd

    def load_state_dict(self, sd):
        want = set(self._STATE_KEYS) | {"version"} | ({"prestige_t"} if self.prestige_t is not None else set())
        if set(sd.keys()) != want:
            raise KeyError("load_state_dict: expected exactly the keys %s, got %s (a checkpoint without "
                           "'version' / 'mt_head' predates the look-ahead RNG form and cannot be resumed)"
                           % (sorted(want), sorted(sd.keys())))
        if int(sd["version"]) != STATE_DICT_VERSION:
            raise ValueError("load_state_dict: checkpoint version %d, this engine reads %d"
                             % (int(sd["version"]), STATE_DICT_VERSION))
        for k in want - {"version"}:
            if tuple(sd[k].shape) != tuple(getattr(self, k).shape):
                raise ValueError("load_state_dict: %s has shape %s, expected %s" % (k, tuple(sd[k].shape),
                                                                                    tuple(getattr(self, k).shape)))
        for k in want - {"version"}:
            getattr(self, k).copy_(sd[k])
        if bool((self.error_t != 0).any()):
            self._flag._host[0] = 1          # the restored batch carries recorded errors (whatever `strict` is:
                                             # check_errors() looks at the flag first)

    # ---- plain-data description (parity tests hand this to the oracle) -----------------------------------
    def scenario_spec(self):
        def ospec(o):
            if o is None:
                return None
            d = dict(type=o.__class__.__name__, color=o.color, state=o.state)
            if isinstance(o, Goal):
                d["reward"] = o.reward
            if isinstance(o, BonusTile):
                d.update(reward=o.reward, penalty=o.penalty, bonus_id=o.bonus_id, n_bonus=o.n_bonus,
                         initial_reward=o.initial_reward, reset_on_mistake=o.reset_on_mistake)
            return d

        def prog(p):
            out = list(p["sym"])
            late = p.get("late", {})
            for i, (k, c, t, x0, y0, x1, y1, rej) in enumerate(p["ops"]):
                if t == 0:               # a static edit after a placement: its symbolic form, once
                    out.extend(late.get(i, []))
                    continue
                full = (x0, y0, x1, y1) == (0, 0, self.width, self.height)
                if rej is not None:      # reject_fn, tabulated: the rejected cells of the sampling rectangle
                    cells = np.argwhere(np.frombuffer(rej, np.uint8).reshape(self.width, self.height))
                    out.append(("place", k, c, t, x0, y0, x1, y1, tuple((int(x), int(y)) for x, y in cells)))
                else:
                    out.append(("place", k, c, t) if full else ("place", k, c, t, x0, y0, x1, y1))
            return out
        def aspec(a):
            d = dict(color=a.color)
            if a.spawn_delay:
                d["spawn_delay"] = a.spawn_delay
            if a.hide_item_types:
                d["hide_item_types"] = list(a.hide_item_types)
            if a.color == "prestige":
                d.update(prestige_beta=a.prestige_beta, prestige_scale=a.prestige_scale)
            if self._hetero:
                d["view"] = dict(view_size=a.view_size, tile_size=a.view_tile_size, view_offset=a.view_offset,
                                 see_through_walls=bool(a.see_through_walls))
            if a.observation_style == "rich":
                d["rich"] = {k: True for k in ("observe_rewards", "observe_position", "observe_orientation")
                             if getattr(a, k)}
            return d
        extra = {}
        if self.agent_spawn_kwargs:
            extra["agent_spawn"] = {k: (tuple(v) if k in ("top", "size") else v)
                                    for k, v in self.agent_spawn_kwargs.items() if k != "reject_fn"}
            if self.agent_spawn_kwargs.get("reject_fn") is not None:
                x0, y0, x1, y1 = self._place_region(self.agent_spawn_kwargs.get("top"), self.agent_spawn_kwargs.get("size"))
                t = self._reject_table(self.agent_spawn_kwargs["reject_fn"], (x0, y0, x1, y1))
                extra["agent_spawn"]["reject"] = tuple((int(x), int(y)) for x, y in np.argwhere(t))
        return dict(W=self.width, H=self.height, agents=[aspec(a) for a in self.agents], **extra,
                    view_size=self.view_size, tile_size=self.tile_size, view_offset=self.view_offset,
                    see_through_walls=self.see_through_walls, max_steps=self.max_steps,
                    reward_decay=bool(self.reward_decay), ghost_mode=self.ghost_mode,
                    respawn=bool(self.respawn), objects=[ospec(o) for o in self.obj_reg.objs],
                    wall_obj=self.obj_reg.find(Wall()), gen_ctor=prog(self._spec_ctor or self._spec_last),
                    gen_reset=prog(self._spec_last))

    @_on_device
    def render(self, mode="rgb_array", close=False, highlight=True, tile_size=TILE_PIXELS, show_agent_views=True,
               max_agents_per_col=3, agent_col_width_frac=0.3, agent_col_padding_px=2, pad_grey=100, env_ids=None):
        """Whole-grid human view (base.py:714-795): every cell at `tile_size` pixels, cells visible to
        some active agent highlighted, and (show_agent_views) the agents' own observations stacked in
        side columns.  Returns a uint8 tensor (H_px, W_px, 3) for env 0, or (K, H_px, W_px, 3) for
        `env_ids`.  There is no window: mode='human' returns the image as well."""
        import torch
        if close:
            return None
        if tile_size % 4 != 0 or not (4 <= tile_size <= 64):
            raise NotImplementedError("render(tile_size=) must be a multiple of 4 in [4, 64]")
        single = env_ids is None
        ids = torch.as_tensor([0] if single else env_ids, dtype=torch.int32).reshape(-1)
        K = int(ids.numel())
        if K and (int(ids.min()) < 0 or int(ids.max()) >= self.batch_size):
            raise IndexError("render(env_ids=): env index out of range [0, %d)" % self.batch_size)
        ids = ids.to(self.device)
        self._sync_tables()
        key = (self.obj_reg.version, tile_size)
        if getattr(self, "_frame_atlas_key", None) != key:
            fa, _, _ = rendering.build_atlas(self.obj_reg.objs, [a.color for a in self.agents], tile_size,
                                             prestige_sprites=any(self._prestige))
            self._frame_atlas = torch.from_numpy(np.ascontiguousarray(fa[0])).to(self.device)
            self._frame_atlas_key = key
            self._frame_amax = 0
            if any(self._prestige):
                for d in range(4):
                    self._frame_amax |= int(fa[0, fa.shape[1] - 4 + d][..., 0].max()) << (8 * d)
        Hp, Wp = self.height * tile_size, self.width * tile_size
        # (the frame kernel keeps two bytes per cell in one workgroup's LDS — the cell's tile index and its highlight bit; the grid
        # is read where it lives —: every grid up to 255 x 255 fits, next to the recoloured sprites of up to ~10 'prestige' agents)
        lds = 2 * ((self.width * self.height + 7) // 8 * 8) + 1024 + 4 * self.num_agents * self.view_size + (
            self.num_agents * tile_size * tile_size * 3 if any(self._prestige) else 0)
        if lds > 160 * 1024:
            raise NotImplementedError("render(): the whole-grid image of a %d x %d grid with %d 'prestige' agents at %d-pixel tiles "
                                      "needs more than the 160 KiB of LDS of a workgroup; use a smaller tile_size"
                                      % (self.width, self.height, self.num_agents, tile_size))
        img = torch.empty((K, Hp, Wp, 3), dtype=torch.uint8, device=self.device)
        N.check(self._lib.mg_render_frame(C.byref(self._cfg), C.byref(self._state), ids.data_ptr(), K,
                                          self._frame_atlas.data_ptr(), tile_size, int(bool(highlight)),
                                          self._frame_amax, img.data_ptr(), self._stream()))
        if show_agent_views:
            # side columns (base.py:764-786): views enlarged by an integer factor, max_agents_per_col
            # per column, centred on a grey background.  (The reference mixes shape[0] / shape[1]
            # here; kept as written — images are square in every shipped scenario.)
            tpw = int(Hp * agent_col_width_frac - 2 * agent_col_padding_px)
            tph = (Wp - 2 * agent_col_padding_px) // max_agents_per_col
            self._render()
            cols = []
            for c0 in range(0, self.num_agents, max_agents_per_col):
                col = torch.full((K, Hp, tpw + 2 * agent_col_padding_px, 3), pad_grey, dtype=torch.uint8,
                                 device=self.device)
                for j, k in enumerate(range(c0, min(c0 + max_agents_per_col, self.num_agents))):
                    g, slot = self._view_slot(k)                            # each view at its own integer zoom
                    P = g.pixels
                    f = int(min(tpw / P, tph / P))
                    view = g.obs[ids.long(), slot]                          # (K, P, P, 3)
                    view = view.repeat_interleave(f, dim=1).repeat_interleave(f, dim=2) if f > 0 else view[:, :0, :0]
                    vh = vw = P * f
                    o0 = (tph - vw) // 2 + agent_col_padding_px + j * tph
                    o1 = (tpw - vh) // 2 + agent_col_padding_px
                    col[:, o0:o0 + vh, o1:o1 + vw] = view
                cols.append(col)
            img = torch.cat([img] + cols, dim=2)
        return img[0] if single else img

    def __str__(self):
        return "<%s B=%d %dx%d n_agents=%d>" % (self.__class__.__name__, self.batch_size, self.width,
                                                self.height, self.num_agents)
