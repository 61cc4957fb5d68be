/*
 * mg_oracle.h — CPU ORACLE for the batched MarlGrid hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (kandouss/marlgrid v0.0.5,
 * `marlgrid/base.py`, `marlgrid/agents.py`, `marlgrid/objects.py`, the `marlgrid/envs` scenario files), kept
 * deliberately close to the reference's own data model (a cell holds an object; objects carry an
 * ordered `.agents` list) so that it is an independent check of the product's flat rank-based HIP
 * state model.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
 * load this library; the product package `marlgrid_amd/` never does.
 *
 * Parity pin: validated against the real reference executed in the build container (through
 * test-only shims for its absent third-party imports) — live in `tests/test_oracle_vs_reference.py`
 * and frozen in the `tests/golden` npz fixtures (generator: `tests/golden/make_golden.py`).  Two inputs are
 * "parity unpinned" because they come from unpinned third-party packages that are not part of
 * /root/reference: gym-minigrid's sprite primitives and gym's seed->MT19937
 * hashing; both are restated from their public behaviour.
 */
#ifndef MG_ORACLE_H
#define MG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGO_MAX_AGENTS 32
#define MGO_MAX_OBJ 256
#define MGO_MAX_FILL 8
#define MGO_MAX_GEN 192
#define MGO_MAX_VIEW 31
#define MGO_AGENT_BASE 1000 /* cell value >= this: the cell object is agent (value - base) */

/* error codes (mirroring the reference's exceptions) */
#define MGO_OK 0
#define MGO_ERR_VALUE (-1)     /* ValueError: unknown action              base.py:619-620 */
#define MGO_ERR_RECURSION (-2) /* RecursionError: placement failed        base.py:705-706 */
#define MGO_ERR_TYPE (-3)      /* TypeError: Box.toggle arity             objects.py:381-382 */
#define MGO_ERR_ASSERT (-4)    /* AssertionError: grid.get out of bounds  base.py:154-156 */
#define MGO_ERR_STACK (-5)     /* ValueError("?!?!?!")                    base.py:568-569 */
#define MGO_ERR_ATTRIBUTE (-6) /* AttributeError: None.can_overlap() — an agent whose cell put_obj(None) emptied moves
                                * on (base.py:555-558) */

/* sprite fill op: restates one `fill_coords(img, fn, color)` call of objects.py render methods */
enum { MGO_FILL_RECT = 0, MGO_FILL_TRI_ROT = 1, MGO_FILL_CIRCLE = 2 };
typedef struct {
    int32_t kind;
    double p[4];    /* RECT: xmin,xmax,ymin,ymax; CIRCLE: cx,cy,r; TRI_ROT: theta (triangle fixed) */
    uint8_t rgb[3]; /* already truncated to uint8 as numpy assignment does */
    uint8_t pad;
} MgoFillOp;

/* object descriptor: the predicates / encode of one WorldObj instance kind (objects.py) */
typedef struct {
    int32_t type_idx, color_idx, state;               /* WorldObj.encode       objects.py:90-99 */
    int32_t can_overlap, can_pickup, see_behind;      /* objects.py:75-88 + overrides           */
    int32_t reward_kind;                              /* 0 none, 1 Goal, 2 BonusTile            */
    double reward, penalty;                           /* objects.py:165-172, 212-220            */
    int32_t bonus_id, n_bonus, initial_reward, reset_on_mistake;
    int32_t ends_episode;                             /* isinstance(.., (Lava, Goal)) base.py:584 */
    int32_t toggle_kind;                              /* 0 WorldObj.toggle, 1 Door, 2 Box       */
    int32_t toggle_next;                              /* Door: closed->open, open->closed id    */
    int32_t unlock_next;                              /* Door: locked->closed id (needs key)    */
    int32_t is_key;                                   /* isinstance(carrying, Key)              */
    int32_t n_fill;
    MgoFillOp fill[MGO_MAX_FILL];                     /* obj.render(img)                        */
} MgoObjDesc;

/* `_gen_grid` program (envs/empty.py, envs/cluttered.py, envs/goalcycle.py, envs/viz_test.py) */
enum { MGO_GEN_WALL_RECT = 0, MGO_GEN_HORZ_WALL = 1, MGO_GEN_VERT_WALL = 2, MGO_GEN_PUT = 3,
       MGO_GEN_PLACE = 4 };
typedef struct {
    int32_t kind, obj, count, x, y, w, h, max_tries;
    const uint8_t* reject;   /* place_obj(reject_fn=): the callback tabulated over the grid, index x*H + y, != 0 =
                              * rejected (base.py:700-701: a rejected draw is a spent try); NULL: none.  The
                              * caller keeps it alive. */
} MgoGenOp;

typedef struct {
    int32_t W, H, n_agents;
    int32_t view_size, tile_size, view_offset, see_through_walls; /* agents.py:19-35 (uniform) */
    int32_t max_steps, reward_decay, ghost_mode, respawn;         /* base.py:341-346           */
    int32_t agent_color_idx[MGO_MAX_AGENTS];
    uint8_t agent_rgb[MGO_MAX_AGENTS][4];
    int32_t agent_type_idx;                                       /* 13: GridAgentInterface    */
    int32_t n_obj;                                                /* ids 1..n_obj-1 (0 = None) */
    MgoObjDesc obj[MGO_MAX_OBJ];
    int32_t wall_obj;                                             /* id used by wall_rect      */
    int32_t n_gen[2];                                             /* [0] ctor-time, [1] reset  */
    MgoGenOp gen[2][MGO_MAX_GEN];
    int32_t spawn_delay[MGO_MAX_AGENTS];                          /* agents.py:34, base.py:409-412, 503-506 */
    int32_t is_prestige[MGO_MAX_AGENTS];                          /* color == 'prestige' (agents.py:99) */
    double prestige_beta[MGO_MAX_AGENTS], prestige_scale[MGO_MAX_AGENTS]; /* agents.py:31-32,54-56 */
    uint32_t hide_type_mask[MGO_MAX_AGENTS];                      /* hide_item_types: bit t = type_idx t,
                                                                   * bit 31 = 'Agent' (base.py:441-449) */
    int32_t spawn_x0, spawn_y0, spawn_x1, spawn_y1;               /* place_obj(agent, **agent_spawn_kwargs):
                                                                   * top / size clamped as base.py:692-695
                                                                   * (base.py:411, 505, 643) */
    int32_t spawn_max_tries;                                      /* agent_spawn_kwargs max_tries (<= 1e5) */
    const uint8_t* spawn_reject;                                  /* agent_spawn_kwargs reject_fn, tabulated like
                                                                   * MgoGenOp.reject; NULL: none */
} MgoConfig;

typedef struct MgoEnv MgoEnv;

/* ---- MT19937 (numpy.random.RandomState legacy stream) ---- */
void mgo_mt_init_by_array(uint32_t* mt, int32_t* pos, const uint32_t* key, int32_t key_len);
uint32_t mgo_mt_next(uint32_t* mt, int32_t* pos);
uint32_t mgo_bounded(uint32_t* mt, int32_t* pos, uint32_t max);

/* ---- single env ---- */
MgoEnv* mgo_create(const MgoConfig* cfg, const uint32_t* seed_key, int32_t key_len);
MgoEnv* mgo_create_like(const MgoEnv* src, const uint32_t* seed_key, int32_t key_len);
void mgo_destroy(MgoEnv* e);
int32_t mgo_reset(MgoEnv* e, int32_t which_gen); /* base.py:402-416 (without gen_obs) */
/* base.py:501-649. actions[n]; rewards[n] float64; order_out[n] (shuffled order, may be NULL) */
int32_t mgo_step(MgoEnv* e, const int32_t* actions, double* rewards, int32_t* episode_done,
                 int32_t* order_out);
/* base.py:453-474 'image' style: obs for agent k, (P,P,3) uint8 */
void mgo_render_obs(const MgoEnv* e, int32_t k, uint8_t* out);
/* agents.py:298-343 on the agent's view: vis (vs*vs, index i*vs+j) and view cell top-codes */
void mgo_view(const MgoEnv* e, int32_t k, uint8_t* vis, int32_t* cells);
/* base.py:196-214 encode(vis_mask) -> (W,H,3) uint8; vis may be NULL */
void mgo_encode(const MgoEnv* e, const uint8_t* vis, uint8_t* out);
/* sprite atlas access (base.py:225-299): tile for object id `obj` (0 = empty) shown with agent
 * `agent_k` facing `agent_dir` on top (agent_k < 0: none), un-rotated, (ts,ts,3) uint8 */
void mgo_tile(const MgoEnv* e, int32_t obj, int32_t agent_k, int32_t agent_dir, uint8_t* out);
/* occlusion alone (agents.py:298-343): transp/vis are (vs*vs), index i*vs+j */
void mgo_occlude(int32_t vs, int32_t ax, int32_t ay, const uint8_t* transp, uint8_t* vis);

/* canonical state dump. base: (W*H) object id of the non-agent object in each cell (0 if none);
 * agents: per agent x,y,dir,active,done,carrying,ordinal (position in its cell's stack) */
void mgo_get_state(const MgoEnv* e, uint8_t* base, int32_t* agents7, int32_t* step_count);
void mgo_get_prestige(const MgoEnv* e, double* out);
void mgo_get_mt(const MgoEnv* e, uint32_t* mt624, int32_t* pos);
void mgo_set_agent_dir(MgoEnv* e, int32_t k, int32_t dir);
void mgo_set_carrying(MgoEnv* e, int32_t k, int32_t obj);
/* test helper: overwrite a cell with a non-agent object id (env.put_obj, base.py:655-662) */
int32_t mgo_put_obj(MgoEnv* e, int32_t obj, int32_t x, int32_t y);
/* test helper: teleport an (already placed) agent; re-seats stacks like a fresh placement */
/* reward / position / orientation of agent k's 'rich' observation (base.py:461-471) */
void mgo_rich_obs(const MgoEnv* e, int32_t k, double* reward, double* position2, int32_t* orientation);
int32_t mgo_place_obj(MgoEnv* e, int32_t what, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t max_tries,
                      const uint8_t* reject, int32_t* out_xy);
int32_t mgo_try_place_obj(MgoEnv* e, int32_t what, int32_t x, int32_t y);
int32_t mgo_regen_grid(MgoEnv* e, int32_t which_gen);
int32_t mgo_place_agent_at(MgoEnv* e, int32_t k, int32_t x, int32_t y);

/* ---- batch (OpenMP over envs): cpu_baseline leg of bench.py ---- */
int32_t mgo_batch_step(MgoEnv** envs, int32_t B, const int32_t* actions, double* rewards,
                       uint8_t* done, uint8_t* obs_or_null, int32_t auto_reset, int32_t threads);
int32_t mgo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
