"""ctypes binding of libmarlgrid_hip.so (C ABI: include/marlgrid_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or does not load, importing
this module raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`marlgrid_amd/csrc/build.sh`.
"""
import ctypes as C
import os

MAX_AGENTS, MAX_OBJ, MAX_GEN, MAX_VIEW, KEY_WORDS, MT_N, MT_HEAD = 32, 256, 1024, 31, 2, 624, 16
ABI_VERSION = 6

OK = 0
ERR_VALUE, ERR_RECURSION, ERR_TYPE, ERR_ASSERT, ERR_ATTRIBUTE = 1, 2, 3, 4, 5
ERR_EXC = {ERR_VALUE: ValueError, ERR_RECURSION: RecursionError, ERR_TYPE: TypeError,
           ERR_ASSERT: AssertionError, ERR_ATTRIBUTE: AttributeError}

AG_X, AG_Y, AG_DIR, AG_FLAGS, AG_CARRY, AG_RANK, AG_BONUS = range(7)
AF_ACTIVE, AF_DONE, AF_PLACED, AF_EVICTED = 1, 2, 4, 8
OF_CAN_OVERLAP, OF_CAN_PICKUP, OF_SEE_BEHIND, OF_ENDS_EPISODE = 1, 2, 4, 8
OF_IS_KEY, OF_IS_DOOR, OF_IS_BOX, OF_DOOR_LOCKED = 16, 32, 64, 128


class ObjDesc(C.Structure):
    _fields_ = [("type_idx", C.c_uint8), ("color_idx", C.c_uint8), ("state", C.c_uint8), ("flags", C.c_uint8),
                ("reward_kind", C.c_uint8), ("toggle_next", C.c_uint8), ("unlock_next", C.c_uint8),
                ("ovl_slot", C.c_uint8),
                ("bonus_id", C.c_uint8), ("n_bonus", C.c_uint8), ("bonus_flags", C.c_uint8), ("flags2", C.c_uint8),
                ("pad1", C.c_uint32), ("reward", C.c_double), ("penalty", C.c_double)]


assert C.sizeof(ObjDesc) == 32


class Config(C.Structure):
    _fields_ = [("B", C.c_int32), ("W", C.c_int32), ("H", C.c_int32), ("n_agents", C.c_int32),
                ("view_size", C.c_int32), ("tile_size", C.c_int32), ("view_offset", C.c_int32),
                ("see_through_walls", C.c_int32),
                ("max_steps", C.c_int32), ("reward_decay", C.c_int32), ("ghost_mode", C.c_int32),
                ("respawn", C.c_int32),
                ("cells_stride", C.c_int32), ("n_obj", C.c_int32), ("n_ovl_slots", C.c_int32),
                ("n_tiles", C.c_int32), ("agent_type_idx", C.c_int32), ("atlas_gather_off", C.c_int32),
                ("spawn_x0", C.c_int32), ("spawn_y0", C.c_int32), ("spawn_x1", C.c_int32), ("spawn_y1", C.c_int32),
                ("spawn_max_tries", C.c_int32),
                ("n_view", C.c_int32), ("view_agent", C.c_uint8 * MAX_AGENTS),
                ("agent_color_idx", C.c_uint8 * MAX_AGENTS),
                ("any_spawn_delay", C.c_int32), ("spawn_delay", C.c_int32 * MAX_AGENTS),
                ("prestige_mask", C.c_uint32), ("prestige_amax", C.c_uint8 * 4), ("prestige_sprite_tile", C.c_int32),
                ("prestige_beta", C.c_double * MAX_AGENTS), ("prestige_scale", C.c_double * MAX_AGENTS),
                ("any_hide", C.c_int32), ("hide_agent_mask", C.c_uint32), ("hide_by_obj", C.c_void_p),
                ("obj", C.c_void_p), ("atlas", C.c_void_p), ("spawn_reject", C.c_void_p)]


class State(C.Structure):
    _fields_ = [("grid", C.c_void_p), ("agents", C.c_void_p), ("mt", C.c_void_p), ("mt_pos", C.c_void_p),
                ("step_count", C.c_void_p), ("done", C.c_void_p), ("error", C.c_void_p), ("prestige", C.c_void_p),
                ("mt_head", C.c_void_p), ("error_flag", C.c_void_p)]


class GenOp(C.Structure):
    _fields_ = [("obj", C.c_int32), ("count", C.c_int32), ("max_tries", C.c_int32),
                ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32), ("reject", C.c_int32)]


class GenProgram(C.Structure):
    _fields_ = [("template_grid", C.c_void_p), ("n_ops", C.c_int32), ("ops", C.c_void_p),      # ops: DEVICE memory, GenOp [n_ops]
                ("reject", C.c_void_p), ("n_reject", C.c_int32)]


PLACE_MAX, PLACE_ALL = 8, 136
PLACE_STIR, PLACE_THOROUGH, PLACE_NO_REUSE = 1, 2, 4
PLACE_STOP = {0: "", 1: "found", 2: "cap", 3: "time", 4: "memory", 5: "out of memory", 6: "small", 7: "not HBM-bound"}
E_ARG, E_UNSUPPORTED, E_LAUNCH = -100, -101, -102
E_NOMEM = -103


class PlaceTuning(C.Structure):
    _fields_ = [("gain", C.c_double), ("slow_alloc_s_per_gib", C.c_double), ("min_bytes", C.c_uint64),
                ("stir_bytes", C.c_uint64), ("max_candidates", C.c_int32), ("iters", C.c_int32),
                ("share", C.c_int32), ("reserved1", C.c_int32), ("fast_rate", C.c_double)]


class PlaceStats(C.Structure):
    _fields_ = [("found", C.c_int32), ("reused", C.c_int32), ("candidates", C.c_int32), ("windows", C.c_int32),
                ("passes", C.c_int32), ("level", C.c_int32), ("plain_stage", C.c_int32), ("stopped", C.c_int32),
                ("kept_ms", C.c_float * PLACE_MAX), ("median_ms", C.c_float), ("reserved0", C.c_float),
                ("seconds", C.c_double), ("alloc_seconds", C.c_double),
                ("buffer_bytes", C.c_uint64), ("candidate_bytes", C.c_uint64), ("alloc_bytes", C.c_uint64),
                ("pinned_bytes", C.c_uint64), ("stirred_bytes", C.c_uint64), ("budget_bytes", C.c_uint64),
                ("window_offset", C.c_uint64 * PLACE_MAX), ("arena_bytes", C.c_uint64 * PLACE_MAX),
                ("all_ms", C.c_float * PLACE_ALL)]


LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmarlgrid_hip.so")

# every symbol include/marlgrid_hip.h declares
SYMBOLS = ["mg_abi_version", "mg_struct_sizes", "mg_host_flag_alloc", "mg_host_flag_free", "mg_obs_alloc", "mg_obs_free", "mg_obs_place", "mg_obs_release", "mg_obs_trim", "mg_build_info", "mg_error_string", "mg_mt_seed", "mg_reset", "mg_step", "mg_step_render", "mg_step_render_encode",
           "mg_render_obs",
           "mg_encode", "mg_put_obj", "mg_place", "mg_render_frame", "mg_time_render_obs",
           "mg_render_obs_lds_bytes", "mg_render_kernel_name"]

_lib = None
_path = LIB_PATH


def use_library(path):
    """tools/ only: bind another build of the same ABI (the measurement build libmarlgrid_hip_ab.so) instead of
    the product library — by an explicit call before the first lib(), never through the environment: the
    package loads exactly one library, the one next to it, unless a measurement script says otherwise in code."""
    global _path
    if _lib is not None:
        raise RuntimeError("marlgrid_amd: the library is already loaded (%s)" % _path)
    _path = os.path.abspath(path)


def lib():
    """Load the HIP library (after torch, so that the HIP runtime torch ships is the one bound).  Reads no
    environment variable."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  — loads libamdhip64 first; our .so resolves against the same runtime
    path = _path
    if path == LIB_PATH and not os.path.exists(LIB_PATH):
        # not a fallback: build the one and only implementation if the toolchain is at hand
        import shutil
        import subprocess
        if shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
            rc = subprocess.call(["bash", os.path.join(os.path.dirname(LIB_PATH), "build.sh")])
            if rc != 0:
                raise ImportError("marlgrid_amd: building %s failed (build.sh exit code %d)" % (LIB_PATH, rc))
    if not os.path.exists(path):
        raise ImportError("marlgrid_amd: %s is missing — build it (marlgrid_amd/csrc/build.sh); there is "
                          "no CPU fallback" % path)
    L = C.CDLL(path)
    vp, i32 = C.c_void_p, C.c_int32
    L.mg_abi_version.restype = i32
    if L.mg_abi_version() != ABI_VERSION:
        raise ImportError("marlgrid_amd: %s has ABI version %d, this package needs %d — rebuild it"
                          % (path, L.mg_abi_version(), ABI_VERSION))
    # the struct mirrors above against the library's own sizeof: a layout change that forgot this file (or
    # the ABI version) fails here, at load time, instead of shifting every later field of a launch config
    sizes = (i32 * 7)()
    L.mg_struct_sizes.argtypes = [C.POINTER(i32)]
    L.mg_struct_sizes.restype = i32
    if L.mg_struct_sizes(sizes) != 7:
        raise ImportError("marlgrid_amd: %s: mg_struct_sizes failed" % path)
    mine = [C.sizeof(t) for t in (Config, State, ObjDesc, GenOp, GenProgram, PlaceTuning, PlaceStats)]
    if list(sizes) != mine:
        raise ImportError("marlgrid_amd: struct layouts differ between %s %r and marlgrid_amd/_native.py %r "
                          "(MgConfig, MgState, MgObjDesc, MgGenOp, MgGenProgram, MgPlaceTuning, MgPlaceStats)" % (path, list(sizes), mine))
    L.mg_host_flag_alloc.argtypes = [C.POINTER(C.POINTER(i32)), C.POINTER(C.POINTER(i32))]
    L.mg_host_flag_free.argtypes = [C.POINTER(i32)]
    L.mg_obs_alloc.argtypes = [C.c_uint64, i32]
    L.mg_obs_alloc.restype = vp
    L.mg_obs_free.argtypes = [vp]
    L.mg_obs_place.argtypes = [C.POINTER(Config), C.POINTER(State), i32, C.c_uint64, C.c_double, i32, C.POINTER(PlaceTuning),
                               C.POINTER(vp), C.POINTER(PlaceStats), vp]
    L.mg_obs_release.argtypes = [vp]
    L.mg_obs_trim.argtypes = [i32]
    L.mg_error_string.restype = C.c_char_p
    L.mg_error_string.argtypes = [i32]
    L.mg_build_info.restype = C.c_char_p
    L.mg_mt_seed.argtypes = [i32, vp, vp, vp, vp, vp, vp]
    L.mg_reset.argtypes = [C.POINTER(Config), C.POINTER(State), C.POINTER(GenProgram), vp, vp]
    L.mg_step.argtypes = [C.POINTER(Config), C.POINTER(State), vp, i32, vp, C.POINTER(GenProgram), vp]
    L.mg_step_render.argtypes = [C.POINTER(Config), C.POINTER(State), vp, i32, vp, C.POINTER(GenProgram), vp, vp]
    L.mg_step_render_encode.argtypes = [C.POINTER(Config), C.POINTER(State), vp, i32, vp, C.POINTER(GenProgram), vp, vp, vp]
    L.mg_render_obs.argtypes = [C.POINTER(Config), C.POINTER(State), vp, vp, vp, vp, vp]
    L.mg_encode.argtypes = [C.POINTER(Config), C.POINTER(State), vp, vp, vp]
    L.mg_put_obj.argtypes = [C.POINTER(Config), C.POINTER(State), i32, i32, i32, vp, vp]
    L.mg_place.argtypes = [C.POINTER(Config), C.POINTER(State), i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    L.mg_render_frame.argtypes = [C.POINTER(Config), C.POINTER(State), vp, i32, vp, i32, i32, C.c_uint32, vp, vp]
    L.mg_render_kernel_name.argtypes = [C.POINTER(Config), C.c_char_p, i32]
    L.mg_time_render_obs.argtypes = [C.POINTER(Config), C.POINTER(State), vp, i32, C.POINTER(C.c_float), vp]
    for f in SYMBOLS:
        getattr(L, f)
        if f not in ("mg_error_string", "mg_build_info", "mg_obs_alloc"):
            getattr(L, f).restype = i32
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RuntimeError("libmarlgrid_hip: %s (%d)" % (lib().mg_error_string(rc).decode(), rc))


def render_kernel_name(cfg):
    """(name, generic bits) of the observation-kernel instantiation the launcher picks for `cfg` (mg_render_kernel_name)"""
    buf = C.create_string_buffer(96)
    rc = lib().mg_render_kernel_name(C.byref(cfg), buf, len(buf))
    if rc < 0:
        check(rc)
    return buf.value.decode(), rc
