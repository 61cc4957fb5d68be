/*
 * marlgrid_hip.h — C ABI of the MI355X-native batched MarlGrid step engine
 * (libmarlgrid_hip.so, built from marlgrid_amd/csrc by hipcc --offload-arch=gfx950).
 *
 * The reference (kandouss/marlgrid v0.0.5) is pure Python with no FFI; the boundary this library
 * sits behind is the Python class contract of `MultiGridEnv` (marlgrid/base.py:334-708).  Each
 * entry point below names the reference method(s) it replaces for a BATCH of B independent envs:
 *
 *   mg_mt_seed      MultiGridEnv.seed -> gym.utils.seeding.np_random -> RandomState.seed(list)
 *                                                                       (base.py:371-374)
 *   mg_reset        MultiGridEnv.reset + _gen_grid + place_obj/try_place_obj
 *                                       (base.py:402-416, 664-708; envs/empty.py:9-16,
 *                                        envs/cluttered.py:25-36, envs/goalcycle.py:30-51)
 *   mg_step         MultiGridEnv.step action loop, rewards, done  (base.py:501-649); with a reset
 *                   program also the reset() of every env whose episode just ended (auto-reset)
 *   mg_step_render  the two above + mg_render_obs in one launch: the whole MultiGridEnv.step (base.py:501-653)
 *   mg_render_obs   MultiGridEnv.gen_obs / gen_agent_obs / gen_obs_grid, MultiGrid.slice,
 *                   MultiGrid.opacity, GridAgentInterface.process_vis / occlude_mask,
 *                   MultiGrid.render / render_tile / blend_tiles
 *                                       (base.py:418-474, 123-147, 103-106, 275-331;
 *                                        agents.py:233-266, 290-343)
 *   mg_encode       MultiGrid.encode    (base.py:196-214; objects.py:90-99)
 *   mg_put_obj      MultiGridEnv.put_obj (base.py:655-662)
 *   mg_place        MultiGridEnv.place_obj / try_place_obj outside _gen_grid (base.py:664-708)
 *   mg_render_frame MultiGridEnv.render's whole-grid image: MultiGrid.render(top_agent=None) +
 *                   visibility highlight         (base.py:714-759, 301-331)
 *   mg_render_kernel_name  (no reference counterpart: names the launch for profiles and warns of generic instantiations)
 *   mg_obs_place    the observation arrays MultiGridEnv's constructor / gen_obs allocate (base.py:334-347, 453-474),
 *                   for a batch: where in HBM they lie (construction time; mg_obs_release, mg_obs_trim)
 *
 * Conventions: plain pointers and sizes only (no torch types).  Every buffer is owned by the
 * caller and lives in device memory (HBM) unless marked HOST; kernels never allocate.  All calls
 * are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default stream) and
 * return 0 or a negative MG_E_* code for argument errors detected on the host.  Per-env runtime
 * errors (the reference's exceptions) are recorded in MgState.error[b] (first error sticks) and
 * surfaced by the host wrapper as the matching Python exception.  Re-entrant: no mutable globals (the one
 * exception, behind a mutex: the record of placed observation buffers, mg_obs_place below).
 *
 * HBM layout (struct-of-arrays over the env batch, sized for 288 GB):
 *   grid        uint8  [B][cells_stride]      object id per cell, index x*H + y (= MultiGrid.grid[i,j]);
 *                                            cells_stride = W*H rounded up to 16
 *   agents      uint64 [B][n_agents]          packed record, see MG_AG_* below
 *   mt          uint32 [B][624] + mt_pos[B]   per-env MT19937 (numpy RandomState stream), *lazy*
 *                                            form: word mt_pos is the next one to regenerate
 *   mt_head     uint32 [B][16]                the 16 outputs generated last and not consumed yet
 *                                            (tempered), contiguous per env: what a step draws from
 *   step_count  int32  [B];  done uint8 [B];  error int32 [B]
 *   error_flag  int32  [1]  HOST-mapped (mg_host_flag_alloc) or device: becomes non-zero when any
 *                                            error[b] is set — a host polls this one word instead of
 *                                            scanning error[B] after every launch
 *   obs         uint8  [B][n_agents][P][P][3] with P = view_size*tile_size
 */
#ifndef MARLGRID_HIP_H
#define MARLGRID_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_ABI_VERSION 6
#define MG_MAX_AGENTS 32  /* agents per env (the reference: any number, base.py:335-369; register_marl_env asserts <= 6) */
#define MG_MAX_OBJ 256    /* object kinds incl. id 0 = None: the ids are uint8, as the reference's registry keys are (base.py:25,91) */
#define MG_MAX_GEN 1024   /* ops of a reset program (device memory: a sanity bound, not a buffer size) */
#define MG_MAX_VIEW 31    /* view_size (agents.py:19-35: any; a view row is a 32-bit mask here) */
#define MG_KEY_WORDS 2
#define MG_MT_N 624
#define MG_MT_HEAD 16

/* host-side argument errors (return values) */
#define MG_OK 0
#define MG_E_ARG (-100)
#define MG_E_UNSUPPORTED (-101)
#define MG_E_LAUNCH (-102)
#define MG_E_NOMEM (-103)   /* mg_obs_place: the device could not hold the buffers */

/* per-env runtime errors (MgState.error), mirroring the reference's exceptions */
#define MG_ERR_VALUE 1     /* ValueError: unknown action               base.py:619-620 */
#define MG_ERR_RECURSION 2 /* RecursionError: rejection sampling failed base.py:705-706 */
#define MG_ERR_TYPE 3      /* TypeError: Box.toggle arity               objects.py:381-382 */
#define MG_ERR_ASSERT 4    /* AssertionError: grid.get out of bounds    base.py:154-156 */
#define MG_ERR_ATTRIBUTE 5 /* AttributeError: None.can_overlap() — an agent whose cell put_obj(None) emptied moves on
                            * (base.py:555-558; see MG_AF_EVICTED) */

/* packed agent record (uint64, little endian bytes) */
#define MG_AG_X 0      /* byte 0: x */
#define MG_AG_Y 1      /* byte 1: y */
#define MG_AG_DIR 2    /* byte 2: dir 0..3 (agent.state % 4, objects.py:132-142); survives reset */
#define MG_AG_FLAGS 3  /* byte 3: MG_AF_* */
#define MG_AG_CARRY 4  /* byte 4: carried object id (0 = None) */
#define MG_AG_RANK 5   /* byte 5: arrival rank among the env's agents (stack order = rank order) */
#define MG_AG_BONUS 6  /* byte 6: bonus_state, 0xFF = None (agents.py:169) */
#define MG_AF_ACTIVE 1
#define MG_AF_DONE 2
#define MG_AF_PLACED 4
#define MG_AF_EVICTED 8 /* put_obj replaced the cell the agent stood on (base.py:655-662: `grid.set` drops whatever was
                         * there): the agent is in no cell any more — nobody sees it, nothing stands on it — but keeps
                         * its position, turns and looks; its next successful forward move raises what upstream's
                         * "remove agent from old cell" raises (base.py:555-559): AssertionError on a solid object,
                         * ValueError (list.remove) on an overlappable one or another agent, AttributeError on None */

/* object descriptor flags */
#define MG_OF_CAN_OVERLAP 1
#define MG_OF_CAN_PICKUP 2
#define MG_OF_SEE_BEHIND 4
#define MG_OF_ENDS_EPISODE 8 /* isinstance(fwd_cell, (Lava, Goal))  base.py:584 */
#define MG_OF_IS_KEY 16
#define MG_OF_IS_DOOR 32
#define MG_OF_IS_BOX 64
#define MG_OF_DOOR_LOCKED 128

/* one entry per object id (id 0 = None); 32 bytes */
typedef struct MgObjDesc {
    uint8_t type_idx, color_idx, state, flags; /* WorldObj.encode + predicates, objects.py:75-99 */
    uint8_t reward_kind;                       /* 0 none, 1 Goal (constant), 2 BonusTile */
    uint8_t toggle_next;                       /* Door closed<->open successor id */
    uint8_t unlock_next;                       /* Door locked->closed successor id */
    uint8_t ovl_slot;                          /* atlas slot for "object + agent on top", 0xFF none */
    uint8_t bonus_id, n_bonus, bonus_flags;    /* bonus_flags: 1 initial_reward, 2 reset_on_mistake */
    uint8_t flags2;                            /* 1: a corner pixel of the plain tile is black (base.py:297) */
    uint32_t pad1;
    double reward;                             /* Goal.reward / BonusTile.reward */
    double penalty;                            /* BonusTile.penalty */
} MgObjDesc;

typedef struct MgConfig {
    int32_t B, W, H, n_agents;
    int32_t view_size, tile_size, view_offset, see_through_walls; /* agents.py:19-35 (uniform) */
    int32_t max_steps, reward_decay, ghost_mode, respawn;         /* base.py:341-346.  ghost_mode bits:
                                                                   * 1 = moves may enter occupied cells (`ghost_mode is
                                                                   * not False`, base.py:541), 2 = placements may land on
                                                                   * occupied cells (`bool(ghost_mode)`, base.py:683) */
    int32_t cells_stride;                                         /* bytes per env in `grid` */
    int32_t n_obj;                                                /* valid object ids: 0..n_obj-1 */
    int32_t n_ovl_slots;                                          /* slot 0 = empty cell */
    int32_t n_tiles;                                              /* >= 1 + n_obj + n_ovl_slots*n_agents*4 */
    int32_t agent_type_idx;                                       /* 13 */
    int32_t atlas_gather_off;                                     /* 0, or the byte offset from `atlas` (a multiple of 16) of a copy of
                                                                   * the atlas in the gather raster's LDS layout — per tile row 16 zero
                                                                   * bytes, its 3 * tile_size bytes, zeros up to a multiple of 4; 32 zero
                                                                   * bytes behind the last row; the whole rounded up to 16 —: the
                                                                   * gather instantiations (mg_render_kernel_name: <.., 2>) then copy it as
                                                                   * it is instead of building it per workgroup (was: reserved) */
    int32_t spawn_x0, spawn_y0, spawn_x1, spawn_y1;               /* agent_spawn_kwargs top / size clamped to the
                                                                   * grid like base.py:692-695: agents are placed
                                                                   * in [x0,x1) x [y0,y1) (base.py:411, 505, 643) */
    int32_t spawn_max_tries;                                      /* agent_spawn_kwargs max_tries, 1..100000 */
    int32_t n_view;                                               /* 0: every agent is rendered with the view
                                                                   * parameters above.  k > 0: only the k agents
                                                                   * view_agent[0..k) are (obs [B][k][P][P][3]) — an env
                                                                   * whose agents differ in view size / tile size /
                                                                   * offset (agents.py:19-35) is rendered group by group,
                                                                   * one mg_render_obs per group with that group's
                                                                   * view parameters and atlas */
    uint8_t view_agent[MG_MAX_AGENTS];
    uint8_t agent_color_idx[MG_MAX_AGENTS];
    int32_t any_spawn_delay;                                      /* 1 if some spawn_delay != 0 */
    int32_t spawn_delay[MG_MAX_AGENTS];                           /* agents.py:34; base.py:409-412, 503-506 */
    uint32_t prestige_mask;                                       /* bit k: agent k's colour is 'prestige' */
    uint8_t prestige_amax[4];                                     /* max sprite alpha per dir at tile_size */
    int32_t prestige_sprite_tile;                                 /* atlas tile index of the 4 un-bordered white
                                                                   * agent sprites (dir 0..3) appended to the atlas */
    double prestige_beta[MG_MAX_AGENTS], prestige_scale[MG_MAX_AGENTS];   /* agents.py:31-32, 141-153 */
    int32_t any_hide;                                             /* 1 if any mask below is non-zero */
    uint32_t hide_agent_mask;                                     /* bit k: agent k hides type 'Agent' */
    const uint32_t* hide_by_obj; /* device uint32 [n_obj] or NULL (nobody hides an object): bit k of entry o = agent k hides
                             * object id o (hide_item_types, base.py:441-449) */
    const MgObjDesc* obj;   /* device, [n_obj] */
    const uint8_t* atlas;   /* device, [4 orientations][n_tiles][tile_size*tile_size*3], pre-rotated.
                             * tile 0 = shadow; 1+o = object o alone (o=0: empty tile);
                             * 1+n_obj+(slot*n_agents+k)*4+d = slot's object with agent k facing d */
    const uint8_t* spawn_reject; /* device [cells_stride] or NULL: agent_spawn_kwargs['reject_fn'] (base.py:411, 505,
                             * 643 -> :700-701) tabulated over the grid, index x*H + y, != 0 = rejected: a rejected
                             * draw is a spent try, exactly like a cell that does not accept the agent */
} MgConfig;

typedef struct MgState {
    uint8_t* grid;
    uint64_t* agents;
    uint32_t* mt;
    int32_t* mt_pos;
    int32_t* step_count;
    uint8_t* done;
    int32_t* error;
    double* prestige;     /* [B][n_agents] agent.prestige (agents.py:141-153); NULL unless prestige_mask != 0 */
    uint32_t* mt_head;    /* [B][MG_MT_HEAD] */
    int32_t* error_flag;  /* [1] or NULL.  A kernel that records a per-env error in error[b] also ORs 1 into this word
                           * with a system-scope atomic, so the word may live in host-mapped memory
                           * (mg_host_flag_alloc) and be polled by the host without a stream synchronize: the
                           * reference raises inside step() (base.py:619-620); a batched host learns of an error
                           * without giving up its asynchronous launch queue.  Never cleared by the library. */
} MgState;

/* `_gen_grid` as data: a static template (walls / put_obj results) + ordered random placements */
typedef struct MgGenOp {
    int32_t obj, count, max_tries; /* max_tries >= 1: place_obj(obj, max_tries) x count (rejection sampling, below).
                                   * max_tries == 0 (ABI 4): a STATIC edit that `_gen_grid` makes after a random placement
                                   * (put_obj / grid.set / a wall helper, base.py:655-662, 160-176): `obj` (0 = None) is
                                   * written into every cell of [x0,x1) x [y0,y1), replacing what is there; no RNG draw.
                                   * Static edits BEFORE the first placement are part of template_grid. */
    int32_t x0, y0, x1, y1;       /* sampling rectangle [x0,x1) x [y0,y1): place_obj(top=, size=) clamped
                                   * to the grid (base.py:692-695); the whole grid by default */
    int32_t reject;               /* place_obj(reject_fn=) (base.py:690, 700-701): the callback tabulated once over the
                                   * grid — row `reject` of MgGenProgram.reject, -1 = none.  A per-draw Python callback
                                   * cannot run on the device; a function of the position alone is a table. */
} MgGenOp;
typedef struct MgGenProgram {
    const uint8_t* template_grid; /* device, [cells_stride] */
    int32_t n_ops;
    const MgGenOp* ops;           /* DEVICE, [n_ops]: placements and late static edits, in `_gen_grid` order — upstream's `_gen_grid`
                                   * is free Python of any length (marlgrid/envs); the program lives in device memory, not in the
                                   * launch arguments.  The library cannot look into it from the host: the caller hands over ops
                                   * with 0 <= obj < n_obj (>= 1 for placements), a non-empty rectangle inside the grid, count >= 0,
                                   * max_tries >= 0 and reject in [-1, n_reject) */
    const uint8_t* reject;        /* device, [n_reject][cells_stride], index x*H + y, != 0 = rejected; NULL if no op
                                   * has a reject table */
    int32_t n_reject;
} MgGenProgram;

int32_t mg_abi_version(void);
/* sizeof of the structs of this header as THIS build sees them: out[0..6] = MgConfig, MgState, MgObjDesc, MgGenOp,
 * MgGenProgram, MgPlaceTuning, MgPlaceStats.  A binding compares them with its own mirror of the structs at load time
 * (the layouts have no other self-description; MG_ABI_VERSION changes whenever one of them does).  Returns 7. */
int32_t mg_struct_sizes(int32_t out[7]);
/* "<library> gfx950 abi<N> <source id>": which build answered (bench.py echoes it) */
const char* mg_build_info(void);
const char* mg_error_string(int32_t code);

/* keys: device uint32 [B][MG_KEY_WORDS] (sha512-derived words, host computed), key_len: device
 * int32 [B] (1 or 2).  Writes mt [B][624], mt_head [B][16] (the stream's first 16 outputs) and
 * mt_pos [B] (= 16: the next word to regenerate). */
int32_t mg_mt_seed(int32_t B, const uint32_t* keys, const int32_t* key_len, uint32_t* mt,
                   int32_t* mt_pos, uint32_t* mt_head, void* stream);

/* env_mask: device uint8 [B] or NULL (= all); only envs with mask != 0 are reset. */
int32_t mg_reset(const MgConfig* cfg, const MgState* st, const MgGenProgram* prog,
                 const uint8_t* env_mask, void* stream);

/* actions: device [B][n_agents], element size `action_bytes` in {1,4,8} (little-endian ints).
 * rewards: device float32 [B][n_agents].  Sets st->done[b].
 * auto_reset: NULL, or the reset program: an env whose episode ends in this step (done[b] = 1) starts
 * its next episode inside the same launch, exactly as mg_reset masked by the done flags would
 * (done[b] keeps reporting the end; step_count[b] = 0 afterwards). */
int32_t mg_step(const MgConfig* cfg, const MgState* st, const void* actions, int32_t action_bytes,
                float* rewards, const MgGenProgram* auto_reset, void* stream);

/* MultiGridEnv.step as ONE launch: mg_step (arguments as above) and mg_render_obs fused — the wave that
 * renders an env steps it first (one lane per env), so the whole base.py:501-653 step is a single kernel and
 * nothing separates the action loop from the observation raster.  Results are identical to mg_step
 * followed by mg_render_obs. */
int32_t mg_step_render(const MgConfig* cfg, const MgState* st, const void* actions, int32_t action_bytes,
                       float* rewards, const MgGenProgram* auto_reset, uint8_t* obs, void* stream);

/* mg_step_render that also writes MultiGrid.encode (base.py:196-214) of the stepped batch — what mg_encode(cfg, st,
 * NULL, encode_out) would write right after it — from the same launch: the wave that has stepped and drawn its envs
 * still holds their grids and records.  encode_out: device uint8 [B][W][H][3] (any alignment).  +2.4 % bytes on
 * MarlGrid-3AgentCluttered15x15-v0 instead of a second launch.  Compiled into instantiations of the step kernel of their
 * own: view sizes 7, 9 and those above 9 with 8-pixel tiles, view 7 with 5-pixel tiles.  MG_E_UNSUPPORTED —
 * nothing launched, call mg_step_render and mg_encode — for every other shape, and when n_obj + 4 n_agents > 256 (object
 * ids and agent marks do not share a byte), with 'prestige' agents, or when the grid or the atlas does not fit LDS. */
int32_t mg_step_render_encode(const MgConfig* cfg, const MgState* st, const void* actions, int32_t action_bytes,
                              float* rewards, const MgGenProgram* auto_reset, uint8_t* obs, uint8_t* encode_out,
                              void* stream);

/* obs: device uint8 [B][n][P][P][3].  Optional debug outputs (NULL to skip):
 * view_cells uint8 [B][n][vs][vs] (object id of the rotated sub-grid, index [i][j]),
 * view_agent uint8 [B][n][vs][vs] (shown agent index or 0xFF), vis_mask uint8 [B][n][vs][vs]. */
int32_t mg_render_obs(const MgConfig* cfg, const MgState* st, uint8_t* obs, uint8_t* view_cells,
                      uint8_t* view_agent, uint8_t* vis_mask, void* stream);

/* out: device uint8 [B][W][H][3]; vis_mask: device uint8 [B][W][H] or NULL. */
int32_t mg_encode(const MgConfig* cfg, const MgState* st, const uint8_t* vis_mask, uint8_t* out,
                  void* stream);

/* env_mask as in mg_reset. Replaces whatever is in the cell (base.py:655-662) — agents standing there included: they
 * leave the grid (MG_AF_EVICTED). */
int32_t mg_put_obj(const MgConfig* cfg, const MgState* st, int32_t obj, int32_t x, int32_t y,
                   const uint8_t* env_mask, void* stream);

/* Live placement (outside the reset program).  what >= 1: object id; what < 0: agent -(what+1), which
 * is lifted off the grid first and re-seated with the highest arrival rank (as a respawn does).
 * fixed_pos == NULL: rejection sampling in [x0,x1) x [y0,y1) with at most max_tries draws per env on the
 * env's RNG (place_obj, RecursionError recorded on failure); fixed_pos: device int32 [B][2], one attempt
 * at that cell (try_place_obj).  out_pos: device int32 [B][2] or NULL; out_ok: device uint8 [B] or NULL. */
int32_t mg_place(const MgConfig* cfg, const MgState* st, int32_t what, int32_t x0, int32_t y0, int32_t x1,
                 int32_t y1, int32_t max_tries, const int32_t* fixed_pos, const uint8_t* env_mask,
                 const uint8_t* reject, int32_t* out_pos, uint8_t* out_ok, void* stream);
/* reject: device uint8 [cells_stride] or NULL — place_obj(reject_fn=) tabulated (see MgGenOp.reject). */

/* Whole-grid human view of selected envs (caller-side format, not on the step path).
 * env_ids: device int32 [n_envs]; frame_atlas: device uint8 [n_tiles][ts][ts][3] (orientation 0,
 * same tile numbering as MgConfig.atlas) rendered at frame_tile_size (a multiple of 4, normally
 * TILE_PIXELS = 32); out: device uint8 [n_envs][H*ts][W*ts][3]. */
int32_t mg_render_frame(const MgConfig* cfg, const MgState* st, const int32_t* env_ids, int32_t n_envs,
                        const uint8_t* frame_atlas, int32_t frame_tile_size, int32_t highlight,
                        uint32_t frame_amax, uint8_t* out, void* stream);
/* frame_amax: the four per-dir max sprite alphas at frame_tile_size packed little-endian (only used
 * when prestige_mask != 0). */

/* LDS bytes per workgroup mg_render_obs needs for `cfg` in its smallest shape (4-wave workgroups, the
 * atlas read in place when it does not fit next to the per-env scratch).  More than 163840 (160 KiB,
 * gfx950): the launch would fail with MG_E_LAUNCH — hosts check this when they build the config
 * (no device access; obj / atlas may still be NULL). */
int32_t mg_render_obs_lds_bytes(const MgConfig* cfg);

/* Which instantiation of the observation kernel `cfg` gets from the launcher (mg_render_obs and mg_step_render launch the
 * same one), as rocprofv3 prints it: "mg::render_kernel<VS, TS, WPB, V, RM>" — view size, tile size (0: read from cfg at
 * run time), waves per workgroup, variant (0 plain, 8 atlas read in place, 9 'prestige' recolouring, 12 both), raster (0 by
 * tile size: 16-byte chunks at 8 / 16 / 32, assemble-and-stream otherwise; 2 gather).  No device access, nothing is launched
 * (obj / atlas may be NULL).  Returns a bit mask >= 0: 1 = the view size, 2 = the tile size is a run-time value — 3 is the
 * fully generic instantiation, 2-2.4x slower than a specialised one (hosts warn) — or MG_E_ARG / MG_E_LAUNCH (does not fit LDS). */
int32_t mg_render_kernel_name(const MgConfig* cfg, char* out, int32_t cap);

/* timing helper for bench.py: average duration (ms) of `iters` back-to-back mg_render_obs
 * launches on `stream`, bracketed by HIP events recorded on that same stream. */
int32_t mg_time_render_obs(const MgConfig* cfg, const MgState* st, uint8_t* obs, int32_t iters,
                           float* avg_ms, void* stream);

/* ---- memory helpers (optional: callers may bring any device-accessible memory) -------------------------------
 *
 * mg_host_flag_alloc: one int32 in pinned, coherent host memory, mapped into the device's address space:
 * *host is what the CPU reads, *dev is what goes into MgState.error_flag.  Zero-initialised. */
int32_t mg_host_flag_alloc(int32_t** host, int32_t** dev);
int32_t mg_host_flag_free(int32_t* host);

/* mg_obs_alloc / mg_obs_free: device memory straight from the driver (hipMalloc on `device`), outside any caching
 * allocator.  NULL on failure.  Nothing on the step path allocates: these are construction-time helpers, and any other
 * device memory is as valid an `obs` argument. */
void* mg_obs_alloc(uint64_t bytes, int32_t device);
int32_t mg_obs_free(void* ptr);

/* ---- where the observation buffers live ---------------------------------------------------------------------------
 *
 * MultiGridEnv's constructor allocates the observation arrays it returns (base.py:334-347, 453-474); for a batch they are
 * the one large output of a step, and WHERE in HBM they lie decides a fifth of the raster's speed (measured, MI355X,
 * profiles/r04/README.md section 1 + profiles/r05/README.md): the raster writes thousands of concurrent sequential
 * streams, and that pattern runs at 5.3 TB/s into a buffer that lies inside one of the driver's physical blocks, at the
 * speed of a dense fill (6.9 TB/s) into one that straddles the boundary between two blocks half and half — one valley per
 * allocation, at the block boundary, half a buffer wide on either side.  Nothing else moves it (stride, phase and number
 * of the streams, gaps, who writes what, address translation, L2 counters: all the same).
 *
 * mg_obs_place CONSTRUCTS `n_buffers` buffers on such boundaries for the launch configuration `cfg` (obs of
 * B * n_view-or-n_agents * P * P * 3 bytes each): a candidate is one hipMalloc of 3 P' bytes (P' = the power of two >= half
 * the buffer — the driver builds it from a 2 P' block and a P' block), the buffer the window centred on the junction;
 * candidates are timed with the raster itself (the state `st` as it is: HIP events around `iters` mg_render_obs launches on
 * `stream`, blocking) and kept alive until `n_buffers` of them run 12 % under the median candidate, then all the others
 * go back to the driver.  Buffers under 256 MiB are plain allocations (the effect needs thousands of streams), and so are
 * the buffers of a configuration whose raster writes them at under 3.5 TB/s (it is bound by something else than HBM).
 *   budget_bytes  bytes of candidates alive at any time; 0 = min(a quarter of the free memory, 32 GiB), raised to the kept
 *                 buffers + three first-level candidates while that is within half of the free memory (large buffers: a
 *                 16 GB buffer's candidates are 24 GiB each)
 *   seconds       time limit of a pass (not applied before the plain baseline allocations are in); <= 0 = 2 s, more for
 *                 candidates above 6 GiB (0.33 s per GiB of candidate, 12 s at most)
 *   flags         MG_PLACE_THOROUGH: larger block pairs (candidates of 6 P' and 12 P' bytes — a kept buffer then pins up
 *                 to 12x its size) and a second pass when the first found nothing; MG_PLACE_STIR: when nothing was found
 *                 and allocations were slow (memory nobody had before is cleared as it is handed out, front to back, all in
 *                 one block), one allocate-and-free of half the free memory (64 GiB at most) to mix the driver's free lists;
 *                 MG_PLACE_NO_REUSE: do not take released arenas (below)
 *   tuning        NULL, or overrides of the search's constants (tests force its later stages with them)
 *   out           [n_buffers] device pointers, 4 KiB-aligned, owned by the library: give each back with mg_obs_release
 *   stats         NULL, or what happened: found (0: no candidate was in the fast class — the best seen were kept, the caller
 *                 may try again later or with MG_PLACE_THOROUGH), kept_ms, pinned_bytes (what the kept buffers' allocations
 *                 hold: 1.5 .. 3 x the buffers at the default level) ...
 * Returns MG_OK, MG_E_ARG, MG_E_NOMEM (nothing handed out) or MG_E_LAUNCH.
 *
 * mg_obs_release gives a placed buffer back.  An arena of the fast class is REMEMBERED (per process and device; at most
 * min(8 GiB, a sixteenth of the device) of them): the next mg_obs_place of the same buffer size on that device takes it,
 * checks it with one measurement and does not search (stats->reused).  mg_obs_trim(device) (-1: every device) returns
 * the remembered arenas to the driver and reports how many there were.  This record is the library's only process state
 * (a mutex guards it); everything else in this header is re-entrant. */
#define MG_PLACE_MAX 8      /* buffers per call */
#define MG_PLACE_ALL 136    /* candidates recorded in MgPlaceStats.all_ms */
#define MG_PLACE_STIR 1
#define MG_PLACE_THOROUGH 2
#define MG_PLACE_NO_REUSE 4
#define MG_PLACE_STOP_FOUND 1   /* MgPlaceStats.stopped: the kept set is `gain` under the median candidate — or takes the raster's bytes
                                 * at 5.9 TB/s and more, the rate of the fast class (buffers of several GB: wherever they lie) */
#define MG_PLACE_STOP_CAP 2     /* max_candidates measured */
#define MG_PLACE_STOP_TIME 3
#define MG_PLACE_STOP_MEMORY 4  /* the budget does not hold another candidate (after the losers went back, up to six times) */
#define MG_PLACE_STOP_OOM 5     /* hipMalloc failed */
#define MG_PLACE_STOP_SMALL 6   /* buffers under min_bytes: plain allocations */
#define MG_PLACE_STOP_UNBOUND 7 /* the raster writes the plain buffers at under 3.5 TB/s: not bound by HBM writes, nothing to place */
typedef struct MgPlaceTuning {  /* 0 = the default of each */
    double gain;                /* 0.12 */
    double slow_alloc_s_per_gib;/* MG_PLACE_STIR only when allocations took at least this long: 0.02; < 0: always */
    uint64_t min_bytes;         /* 256 MiB */
    uint64_t stir_bytes;        /* cap of the allocate-and-free: 64 GiB */
    int32_t max_candidates;     /* per pass: 192 (the first MG_PLACE_ALL are recorded in MgPlaceStats.all_ms) */
    int32_t iters;              /* raster launches per measurement: 3 */
    int32_t share;              /* processes that share this device's memory (ranks of an oversubscribed launch): the default
                                 * budget is worked out from 1 / share of what is free.  0 / 1: this process counts on all of it */
    int32_t reserved1;
    double fast_rate;           /* bytes per second at which a kept set counts as found whatever the median says: 5.9e12; < 0: never */
} MgPlaceTuning;
typedef struct MgPlaceStats {
    int32_t found, reused, candidates, windows;   /* windows: positions measured (>= candidates) */
    int32_t passes, level, plain_stage, stopped;  /* level: the largest block pair tried (candidates of 3 P' << level) */
    float kept_ms[MG_PLACE_MAX];                  /* the raster into each kept buffer */
    float median_ms, reserved0;                   /* ... and into the median candidate */
    double seconds, alloc_seconds;                /* the whole call; inside hipMalloc */
    uint64_t buffer_bytes, candidate_bytes, alloc_bytes, pinned_bytes, stirred_bytes, budget_bytes;
    uint64_t window_offset[MG_PLACE_MAX];         /* of each kept buffer inside its allocation */
    uint64_t arena_bytes[MG_PLACE_MAX];           /* ... and that allocation's size */
    float all_ms[MG_PLACE_ALL];                   /* the first MG_PLACE_ALL candidates measured, in order (the first n_buffers: plain allocations) */
} MgPlaceStats;
int32_t mg_obs_place(const MgConfig* cfg, const MgState* st, int32_t n_buffers, uint64_t budget_bytes, double seconds,
                     int32_t flags, const MgPlaceTuning* tuning, void** out, MgPlaceStats* stats, void* stream);
int32_t mg_obs_release(void* ptr);
int32_t mg_obs_trim(int32_t device);

#ifdef __cplusplus
}
#endif
#endif
