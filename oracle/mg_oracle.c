/*
 * mg_oracle.c — CPU ORACLE (test infrastructure only; see mg_oracle.h for the contract).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * Build: `make -C oracle` (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include "mg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* MT19937 — numpy.random.RandomState legacy stream (the reference's `self.np_random`,
 * marlgrid/base.py:371-374 via gym.utils.seeding.np_random).                                  */
/* ------------------------------------------------------------------------------------------ */

#define MT_N 624
#define MT_M 397

static void mt_init_genrand(uint32_t* mt, uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < MT_N; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

/* RandomState.seed(list) -> init_by_array; leaves pos = 624 so the first draw regenerates. */
void mgo_mt_init_by_array(uint32_t* mt, int32_t* pos, const uint32_t* key, int32_t key_len) {
    mt_init_genrand(mt, 19650218u);
    int i = 1, j = 0;
    int k = MT_N > key_len ? MT_N : key_len;
    for (; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
        if (j >= key_len) j = 0;
    }
    for (k = MT_N - 1; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
    }
    mt[0] = 0x80000000u;
    *pos = MT_N;
}

static void mt_regen(uint32_t* mt) {
    int kk;
    uint32_t y;
    for (kk = 0; kk < MT_N - MT_M; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < MT_N - 1; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

uint32_t mgo_mt_next(uint32_t* mt, int32_t* pos) {
    if (*pos >= MT_N) { mt_regen(mt); *pos = 0; }
    uint32_t y = mt[(*pos)++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* numpy's masked-rejection bounded draw (legacy RandomState.randint with array bounds and
 * RandomState.shuffle's random_interval): smallest 2^k-1 >= max, redraw 32-bit words until
 * (w & mask) <= max; max == 0 consumes nothing.                                              */
uint32_t mgo_bounded(uint32_t* mt, int32_t* pos, uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = mgo_mt_next(mt, pos) & mask; } while (v > max);
    return v;
}

/* ------------------------------------------------------------------------------------------ */
/* shared (per-config) data: config + sprite tables                                            */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    MgoConfig cfg;
    int refs;
    int tile_bytes;            /* ts*ts*3 */
    uint8_t* obj_tile;         /* [n_obj][tile_bytes]: cache_render_obj(obj) (obj 0: empty_tile) */
    uint8_t* agent_tile;       /* [n_agents][4][tile_bytes]: cache_render_obj(agent) by abs dir  */
    uint8_t* empty_tile;       /* alias of obj_tile[0] */
} MgoShared;

struct MgoEnv {
    MgoShared* sh;
    /* the grid: one int per cell, index x*H + y like MultiGrid.grid[i, j] (base.py:91) */
    int32_t* cell;             /* 0 None | 1..n_obj-1 object | MGO_AGENT_BASE+k agent k */
    int32_t* cell_agents;      /* [W*H][MAX_AGENTS]: `.agents` of the non-agent object in the cell */
    int32_t* cell_nagents;     /* [W*H] */
    int32_t ag_agents[MGO_MAX_AGENTS][MGO_MAX_AGENTS]; /* agent.agents (objects.py:52) */
    int32_t ag_nagents[MGO_MAX_AGENTS];
    int32_t ax[MGO_MAX_AGENTS], ay[MGO_MAX_AGENTS];    /* agent.pos (-1,-1 = None) */
    int32_t adir[MGO_MAX_AGENTS];                      /* agent.state % 4 (objects.py:132-142) */
    int32_t aactive[MGO_MAX_AGENTS], adone[MGO_MAX_AGENTS];
    int32_t acarry[MGO_MAX_AGENTS];                    /* carried object id, 0 = None */
    int32_t abonus[MGO_MAX_AGENTS];                    /* agent.bonus_state, -1 = None */
    double aprestige[MGO_MAX_AGENTS];                  /* agent.prestige (agents.py:141-153,168) */
    int32_t step_count;
    uint32_t mt[MT_N];
    int32_t mt_pos;
};

/* ---- sprite generation: MultiGrid.render_object (base.py:252-258) over gym-minigrid's
 * fill_coords / point_in_rect / point_in_triangle / rotate_fn / downsample ------------------ */

static int in_triangle(double x, double y) {
    /* point_in_triangle((0.12,0.19),(0.87,0.50),(0.12,0.81)) — objects.py:151 */
    const double a0 = 0.12, a1 = 0.19, b0 = 0.87, b1 = 0.50, c0 = 0.12, c1 = 0.81;
    double v00 = c0 - a0, v01 = c1 - a1;
    double v10 = b0 - a0, v11 = b1 - a1;
    double v20 = x - a0, v21 = y - a1;
    double dot00 = v00 * v00 + v01 * v01;
    double dot01 = v00 * v10 + v01 * v11;
    double dot02 = v00 * v20 + v01 * v21;
    double dot11 = v10 * v10 + v11 * v11;
    double dot12 = v10 * v20 + v11 * v21;
    double inv = 1 / (dot00 * dot11 - dot01 * dot01);
    double u = (dot11 * dot02 - dot01 * dot12) * inv;
    double v = (dot00 * dot12 - dot01 * dot02) * inv;
    return (u >= 0) && (v >= 0) && (u + v) < 1;
}

static int fill_hit(const MgoFillOp* op, double xf, double yf) {
    switch (op->kind) {
    case MGO_FILL_RECT:
        return xf >= op->p[0] && xf <= op->p[1] && yf >= op->p[2] && yf <= op->p[3];
    case MGO_FILL_CIRCLE:
        return (xf - op->p[0]) * (xf - op->p[0]) + (yf - op->p[1]) * (yf - op->p[1]) <= op->p[2] * op->p[2];
    case MGO_FILL_TRI_ROT: {
        /* rotate_fn(tri_fn, cx=0.5, cy=0.5, theta) — objects.py:152 */
        double th = op->p[0];
        double x = xf - 0.5, y = yf - 0.5;
        double x2 = 0.5 + x * cos(-th) - y * sin(-th);
        double y2 = 0.5 + y * cos(-th) + x * sin(-th);
        return in_triangle(x2, y2);
    }
    }
    return 0;
}

static void render_object(int ts, const MgoFillOp* ops, int n_ops, uint8_t* out) {
    int S = ts * 3;
    uint8_t* canvas = (uint8_t*)calloc((size_t)S * S * 3, 1);
    for (int o = 0; o < n_ops; o++)
        for (int y = 0; y < S; y++)
            for (int x = 0; x < S; x++) {
                double yf = (y + 0.5) / S, xf = (x + 0.5) / S;
                if (fill_hit(&ops[o], xf, yf)) memcpy(canvas + ((size_t)y * S + x) * 3, ops[o].rgb, 3);
            }
    /* downsample(img, 3): mean over axis 3 then axis 1 (float64), then astype(uint8) */
    for (int r = 0; r < ts; r++)
        for (int c = 0; c < ts; c++)
            for (int ch = 0; ch < 3; ch++) {
                double rows[3];
                for (int sr = 0; sr < 3; sr++) {
                    double s = 0;
                    for (int sc = 0; sc < 3; sc++)
                        s += (double)canvas[((size_t)(r * 3 + sr) * S + (c * 3 + sc)) * 3 + ch];
                    rows[sr] = s / 3.0;
                }
                double m = ((rows[0] + rows[1]) + rows[2]) / 3.0;
                out[(r * ts + c) * 3 + ch] = (uint8_t)m;
            }
    free(canvas);
}

/* MultiGrid.empty_tile — base.py:245-250 */
static void empty_tile(int ts, uint8_t* out) {
    int alpha = ts - 10;
    if (alpha > 20) alpha = 20;
    if (alpha < 0) alpha = 0;
    memset(out, alpha, (size_t)ts * ts * 3);
    for (int r = 1; r < ts; r++)
        for (int c = 0; c < ts - 1; c++) memset(out + (r * ts + c) * 3, 0, 3);
}

static MgoShared* shared_create(const MgoConfig* cfg) {
    MgoShared* sh = (MgoShared*)calloc(1, sizeof(MgoShared));
    sh->cfg = *cfg;
    int ts = cfg->tile_size;
    sh->tile_bytes = ts * ts * 3;
    sh->obj_tile = (uint8_t*)calloc((size_t)cfg->n_obj * sh->tile_bytes, 1);
    sh->agent_tile = (uint8_t*)calloc((size_t)cfg->n_agents * 4 * sh->tile_bytes, 1);
    empty_tile(ts, sh->obj_tile);
    sh->empty_tile = sh->obj_tile;
    for (int o = 1; o < cfg->n_obj; o++)
        render_object(ts, cfg->obj[o].fill, cfg->obj[o].n_fill, sh->obj_tile + (size_t)o * sh->tile_bytes);
    for (int k = 0; k < cfg->n_agents; k++)
        for (int d = 0; d < 4; d++) {
            MgoFillOp op;
            memset(&op, 0, sizeof op);
            op.kind = MGO_FILL_TRI_ROT;
            op.p[0] = (0.5 * M_PI) * d; /* theta=0.5*np.pi*(self.dir) — objects.py:152 */
            memcpy(op.rgb, cfg->agent_rgb[k], 3);
            render_object(ts, &op, 1, sh->agent_tile + ((size_t)k * 4 + d) * sh->tile_bytes);
        }
    return sh;
}

static void shared_release(MgoShared* sh) {
    if (--sh->refs > 0) return;
    free(sh->obj_tile);
    free(sh->agent_tile);
    free(sh);
}

/* MultiGrid.blend_tiles — base.py:260-273 */
static void blend_tiles(int ts, const uint8_t* base, const uint8_t* ag, uint8_t* out) {
    int npx = ts * ts;
    uint64_t max_alpha = 0;
    for (int p = 0; p < npx; p++) {
        uint64_t a = (uint64_t)ag[p * 3] + ag[p * 3 + 1] + ag[p * 3 + 2];
        if (a > max_alpha) max_alpha = a;
    }
    if (max_alpha == 0) { memcpy(out, base, (size_t)npx * 3); return; }
    for (int p = 0; p < npx; p++) {
        uint64_t a = (uint64_t)ag[p * 3] + ag[p * 3 + 1] + ag[p * 3 + 2];
        for (int ch = 0; ch < 3; ch++) {
            double v = (double)((uint64_t)base[p * 3 + ch] * (max_alpha - a) + (uint64_t)ag[p * 3 + ch] * a) /
                       (double)max_alpha;
            out[p * 3 + ch] = (uint8_t)v;
        }
    }
}

/* the tail of MultiGrid.render_tile — base.py:296-298 (uint8 wrap-around add) */
static void add_border_if_black_corner(const MgoShared* sh, uint8_t* img) {
    int ts = sh->cfg.tile_size;
    int corners[4] = {0, ts - 1, (ts - 1) * ts, (ts - 1) * ts + ts - 1};
    int any = 0;
    for (int c = 0; c < 4; c++) {
        const uint8_t* p = img + corners[c] * 3;
        if (p[0] == 0 && p[1] == 0 && p[2] == 0) any = 1;
    }
    if (!any) return;
    for (int b = 0; b < sh->tile_bytes; b++) img[b] = (uint8_t)(img[b] + sh->empty_tile[b]);
}

/* MultiGrid.render_tile (base.py:275-299), expressed on (object id, shown agent) */
/* GridAgentInterface.render_post — agents.py:92-119: an ACTIVE agent whose colour is 'prestige' is
 * recoloured between red (prestige 0) and blue by tanh(prestige / prestige_scale) */
static const uint8_t* agent_sprite(const MgoEnv* e, int k, int dir, uint8_t* scratch) {
    const MgoShared* sh = e->sh;
    const uint8_t* tile = sh->agent_tile + ((size_t)k * 4 + dir) * sh->tile_bytes;
    if (!sh->cfg.is_prestige[k] || !e->aactive[k]) return tile;
    double ps = tanh(e->aprestige[k] / sh->cfg.prestige_scale[k]);
    /* (ps*blue + (1.-ps)*red).astype(int) */
    long col[3] = {(long)(ps * 0.0 + (1. - ps) * 255.0), (long)(ps * 0.0 + (1. - ps) * 0.0), (long)(ps * 255.0 + (1. - ps) * 0.0)};
    int npx = sh->cfg.tile_size * sh->cfg.tile_size;
    for (int p = 0; p < npx; p++) {
        long alpha = tile[p * 3];                       /* tile[...,0].astype(uint16) */
        for (int ch = 0; ch < 3; ch++) scratch[p * 3 + ch] = (uint8_t)((alpha * col[ch]) >> 8);
    }
    return scratch;
}

void mgo_tile(const MgoEnv* e, int32_t obj, int32_t agent_k, int32_t agent_dir, uint8_t* out) {
    const MgoShared* sh = e->sh;
    int ts = sh->cfg.tile_size;
    uint8_t recol[64 * 64 * 3];
    if (obj == 0 && agent_k < 0) { memcpy(out, sh->empty_tile, sh->tile_bytes); return; }
    if (obj == 0) {
        memcpy(out, agent_sprite(e, agent_k, agent_dir, recol), sh->tile_bytes);
    } else {
        const uint8_t* base = sh->obj_tile + (size_t)obj * sh->tile_bytes;
        if (agent_k >= 0)
            blend_tiles(ts, base, agent_sprite(e, agent_k, agent_dir, recol), out);
        else
            memcpy(out, base, sh->tile_bytes);
    }
    add_border_if_black_corner(sh, out);
}

/* ------------------------------------------------------------------------------------------ */
/* env lifecycle                                                                               */
/* ------------------------------------------------------------------------------------------ */

static MgoEnv* env_alloc(MgoShared* sh, const uint32_t* key, int32_t key_len) {
    MgoEnv* e = (MgoEnv*)calloc(1, sizeof(MgoEnv));
    int nc = sh->cfg.W * sh->cfg.H;
    e->sh = sh;
    sh->refs++;
    e->cell = (int32_t*)calloc(nc, sizeof(int32_t));
    e->cell_agents = (int32_t*)calloc((size_t)nc * MGO_MAX_AGENTS, sizeof(int32_t));
    e->cell_nagents = (int32_t*)calloc(nc, sizeof(int32_t));
    for (int k = 0; k < MGO_MAX_AGENTS; k++) { e->ax[k] = e->ay[k] = -1; e->abonus[k] = -1; }
    /* MultiGridEnv.seed — base.py:371-374 */
    mgo_mt_init_by_array(e->mt, &e->mt_pos, key, key_len);
    return e;
}

MgoEnv* mgo_create(const MgoConfig* cfg, const uint32_t* seed_key, int32_t key_len) {
    return env_alloc(shared_create(cfg), seed_key, key_len);
}

/* a further env sharing `src`'s config and sprite tables (batches) */
MgoEnv* mgo_create_like(const MgoEnv* src, const uint32_t* seed_key, int32_t key_len) {
    return env_alloc(src->sh, seed_key, key_len);
}

void mgo_destroy(MgoEnv* e) {
    if (!e) return;
    shared_release(e->sh);
    free(e->cell);
    free(e->cell_agents);
    free(e->cell_nagents);
    free(e);
}

static inline int cidx(const MgoEnv* e, int x, int y) { return x * e->sh->cfg.H + y; }
static inline int in_grid(const MgoEnv* e, int x, int y) {
    return x >= 0 && x < e->sh->cfg.W && y >= 0 && y < e->sh->cfg.H;
}
static inline int is_agent_val(int v) { return v >= MGO_AGENT_BASE; }

/* obj.can_overlap() for a cell value (objects.py:75-76,147-148 + subclass overrides) */
static int val_can_overlap(const MgoEnv* e, int v) {
    if (is_agent_val(v)) return 1;
    return e->sh->cfg.obj[v].can_overlap;
}

/* MultiGridEnv.try_place_obj — base.py:664-688.  `val` is a cell value (object id or agent). */
static int try_place(MgoEnv* e, int val, int x, int y) {
    int c = cidx(e, x, y);
    int g = e->cell[c];
    int obj_is_agent = is_agent_val(val);
    if (g == 0) {
        e->cell[c] = val;
        if (!obj_is_agent) e->cell_nagents[c] = 0; /* a freshly constructed object has no agents */
        if (obj_is_agent) { e->ax[val - MGO_AGENT_BASE] = x; e->ay[val - MGO_AGENT_BASE] = y; }
        return 1;
    }
    if (!(val_can_overlap(e, g) && obj_is_agent)) return 0;
    int g_is_agent = is_agent_val(g);
    int g_n = g_is_agent ? e->ag_nagents[g - MGO_AGENT_BASE] : e->cell_nagents[c];
    if (!(e->sh->cfg.ghost_mode & 2) && (g_is_agent || g_n > 0)) return 0;   /* `not self.ghost_mode` */
    int k = val - MGO_AGENT_BASE;
    if (g_is_agent) e->ag_agents[g - MGO_AGENT_BASE][e->ag_nagents[g - MGO_AGENT_BASE]++] = k;
    else e->cell_agents[c * MGO_MAX_AGENTS + e->cell_nagents[c]++] = k;
    e->ax[k] = x; e->ay[k] = y;
    return 1;
}

/* MultiGridEnv.place_obj — base.py:690-708.  reject: reject_fn tabulated over the grid (x*H + y) or NULL */
static int place_obj_in(MgoEnv* e, int val, double max_tries_in, int x0, int y0, int x1, int y1, const uint8_t* reject,
                        int* ox, int* oy);
static int place_obj(MgoEnv* e, int val, double max_tries_in, const uint8_t* reject) {
    return place_obj_in(e, val, max_tries_in, 0, 0, e->sh->cfg.W, e->sh->cfg.H, reject, 0, 0);
}
/* self.place_obj(agent, **self.agent_spawn_kwargs) — base.py:411, 505, 643 */
static int place_agent_spawn(MgoEnv* e, int k) {
    const MgoConfig* cfg = &e->sh->cfg;
    return place_obj_in(e, MGO_AGENT_BASE + k, (double)cfg->spawn_max_tries, cfg->spawn_x0, cfg->spawn_y0,
                        cfg->spawn_x1, cfg->spawn_y1, cfg->spawn_reject, 0, 0);
}
/* top / size already clamped to [x0,x1) x [y0,y1) as base.py:692-695 does */
static int place_obj_in(MgoEnv* e, int val, double max_tries_in, int x0, int y0, int x1, int y1, const uint8_t* reject,
                        int* ox, int* oy) {
    double mt = max_tries_in < 1e5 ? max_tries_in : 1e5;
    if (mt < 1) mt = 1;
    long max_tries = (long)mt;
    for (long t = 0; t < max_tries; t++) {
        /* np_random.randint(top, bottom): element 0 then element 1; low + bounded(high-low-1) */
        int x = x0 + (int)mgo_bounded(e->mt, &e->mt_pos, (uint32_t)(x1 - x0 - 1));
        int y = y0 + (int)mgo_bounded(e->mt, &e->mt_pos, (uint32_t)(y1 - y0 - 1));
        if (reject && reject[x * e->sh->cfg.H + y]) continue;     /* reject_fn(pos): base.py:700-701 */
        if (try_place(e, val, x, y)) { if (ox) { *ox = x; *oy = y; } return MGO_OK; }
    }
    if (ox) { *ox = -1; *oy = -1; }
    return MGO_ERR_RECURSION;
}

static void grid_set_wall(MgoEnv* e, int x, int y) {
    int c = cidx(e, x, y);
    e->cell[c] = e->sh->cfg.wall_obj;
    e->cell_nagents[c] = 0;
}

int32_t mgo_put_obj(MgoEnv* e, int32_t obj, int32_t x, int32_t y) {
    if (!in_grid(e, x, y)) return MGO_ERR_ASSERT;
    int c = cidx(e, x, y);
    e->cell[c] = obj; /* put_obj replaces whatever is there — base.py:655-662 */
    e->cell_nagents[c] = 0;
    return MGO_OK;
}

/* `_gen_grid`: envs/empty.py:9-16, envs/cluttered.py:25-36, envs/goalcycle.py:30-51,
 * envs/viz_test.py:9-15; wall helpers base.py:160-176 */
static int gen_grid(MgoEnv* e, int which) {
    const MgoConfig* cfg = &e->sh->cfg;
    int nc = cfg->W * cfg->H;
    memset(e->cell, 0, nc * sizeof(int32_t));          /* MultiGrid((W,H)) — base.py:87-101 */
    memset(e->cell_nagents, 0, nc * sizeof(int32_t));
    for (int g = 0; g < cfg->n_gen[which]; g++) {
        const MgoGenOp* op = &cfg->gen[which][g];
        switch (op->kind) {
        case MGO_GEN_WALL_RECT:
            for (int i = 0; i < op->w; i++) { grid_set_wall(e, op->x + i, op->y); grid_set_wall(e, op->x + i, op->y + op->h - 1); }
            for (int j = 0; j < op->h; j++) { grid_set_wall(e, op->x, op->y + j); grid_set_wall(e, op->x + op->w - 1, op->y + j); }
            break;
        case MGO_GEN_HORZ_WALL:
            for (int i = 0; i < op->w; i++) grid_set_wall(e, op->x + i, op->y);
            break;
        case MGO_GEN_VERT_WALL:
            for (int j = 0; j < op->h; j++) grid_set_wall(e, op->x, op->y + j);
            break;
        case MGO_GEN_PUT:
            mgo_put_obj(e, op->obj, op->x, op->y);
            break;
        case MGO_GEN_PLACE:
            for (int n = 0; n < op->count; n++) {
                int rc = (op->w > 0) ? place_obj_in(e, op->obj, (double)op->max_tries, op->x, op->y, op->x + op->w, op->y + op->h, op->reject, 0, 0)
                                     : place_obj(e, op->obj, (double)op->max_tries, op->reject);
                if (rc != MGO_OK) return rc;
            }
            break;
        }
    }
    return MGO_OK;
}

/* MultiGridEnv.reset — base.py:402-416 (gen_obs is a separate call here) */
int32_t mgo_reset(MgoEnv* e, int32_t which_gen) {
    const MgoConfig* cfg = &e->sh->cfg;
    for (int k = 0; k < cfg->n_agents; k++) {
        e->ag_nagents[k] = 0;                      /* agent.agents = [] */
        e->adone[k] = 0; e->aactive[k] = 0;        /* agent.reset(new_episode=True) agents.py:161-170 */
        e->ax[k] = e->ay[k] = -1;
        e->acarry[k] = 0;
        e->abonus[k] = -1;
        e->aprestige[k] = 0.0;                     /* new_episode=True: agents.py:167-168 */
        /* agent.state (dir) is NOT touched by reset */
    }
    int rc = gen_grid(e, which_gen);
    if (rc != MGO_OK) return rc;
    for (int k = 0; k < cfg->n_agents; k++) {      /* base.py:409-412 */
        if (cfg->spawn_delay[k] != 0) continue;
        rc = place_agent_spawn(e, k);
        if (rc != MGO_OK) return rc;
        e->aactive[k] = 1;                         /* agent.activate() */
    }
    e->step_count = 0;
    return MGO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* step — MultiGridEnv.step, base.py:501-649                                                   */
/* ------------------------------------------------------------------------------------------ */

/* list.remove(v): 0 if v is not in the list (Python raises ValueError) */
static int list_remove(int32_t* lst, int32_t* n, int v) {
    for (int i = 0; i < *n; i++)
        if (lst[i] == v) {
            for (int j = i; j < *n - 1; j++) lst[j] = lst[j + 1];
            (*n)--;
            return 1;
        }
    return 0;
}

/* BonusTile.get_reward — objects.py:180-206 */
static double bonus_reward(MgoEnv* e, const MgoObjDesc* o, int k) {
    int first_bonus = 0;
    double rew;
    if (e->abonus[k] < 0) {
        e->abonus[k] = ((o->bonus_id - 1) % o->n_bonus + o->n_bonus) % o->n_bonus; /* python % */
        first_bonus = 1;
    }
    if (e->abonus[k] == o->bonus_id) rew = -fabs(o->penalty);
    else if ((e->abonus[k] + 1) % o->n_bonus == o->bonus_id) { e->abonus[k] = o->bonus_id; rew = o->reward; }
    else rew = -fabs(o->penalty);
    if (o->reset_on_mistake) e->abonus[k] = o->bonus_id;
    if (first_bonus && !o->initial_reward) return 0;
    return rew;
}

int32_t mgo_step(MgoEnv* e, const int32_t* actions, double* rewards, int32_t* episode_done,
                 int32_t* order_out) {
    const MgoConfig* cfg = &e->sh->cfg;
    int n = cfg->n_agents;
    static const int DX[4] = {1, 0, -1, 0}, DY[4] = {0, 1, 0, -1}; /* agents.py:183 */
    int rc = MGO_OK;

    /* spawn agents if it's time — base.py:503-506 (before step_count is incremented) */
    for (int k = 0; k < n; k++)
        if (!e->aactive[k] && !e->adone[k] && e->step_count >= cfg->spawn_delay[k]) {
            int prc = place_agent_spawn(e, k);
            if (prc != MGO_OK) rc = prc;
            e->aactive[k] = 1;
        }

    for (int k = 0; k < n; k++) rewards[k] = 0.0;     /* base.py:510 */
    e->step_count += 1;                               /* base.py:512 */

    /* iter_order = arange(n); np_random.shuffle(iter_order) — base.py:514-516
     * legacy shuffle: for i in reversed(range(1, n)): j = random_interval(i); swap */
    int order[MGO_MAX_AGENTS];
    for (int k = 0; k < n; k++) order[k] = k;
    for (int i = n - 1; i >= 1; i--) {
        int j = (int)mgo_bounded(e->mt, &e->mt_pos, (uint32_t)i);
        int t = order[i]; order[i] = order[j]; order[j] = t;
    }
    if (order_out) for (int k = 0; k < n; k++) order_out[k] = order[k];

    for (int oi = 0; oi < n; oi++) {
        int k = order[oi];
        int action = actions[k];
        if (!e->aactive[k]) continue;                 /* base.py:521 */
        int me = MGO_AGENT_BASE + k;
        int cx = e->ax[k], cy = e->ay[k];
        int fx = cx + DX[e->adir[k]], fy = cy + DY[e->adir[k]];
        if (!in_grid(e, cx, cy) || !in_grid(e, fx, fy)) { rc = MGO_ERR_ASSERT; continue; }
        int cc = cidx(e, cx, cy), fc = cidx(e, fx, fy);
        int cur_cell = e->cell[cc];
        int fwd_cell = e->cell[fc];

        if (action == 0) {                            /* left — base.py:530-531 */
            e->adir[k] = (e->adir[k] + 3) % 4;
        } else if (action == 1) {                     /* right — :534-535 */
            e->adir[k] = (e->adir[k] + 1) % 4;
        } else if (action == 2) {                     /* forward — :538-585 */
            int can_move = (fwd_cell == 0) || val_can_overlap(e, fwd_cell);
            if (!(cfg->ghost_mode & 1) && is_agent_val(fwd_cell)) can_move = 0;   /* `self.ghost_mode is False` */
            if (can_move) {
                /* add agent to new cell — :547-552 */
                if (fwd_cell == 0) e->cell[fc] = me;
                else if (is_agent_val(fwd_cell)) {
                    int f = fwd_cell - MGO_AGENT_BASE;
                    e->ag_agents[f][e->ag_nagents[f]++] = k;
                } else e->cell_agents[fc * MGO_MAX_AGENTS + e->cell_nagents[fc]++] = k;
                e->ax[k] = fx; e->ay[k] = fy;
                /* remove agent from old cell — :555-559.  After put_obj replaced the cell the agent stood on
                 * (:655-662) the agent is in no list any more: `assert cur_cell.can_overlap()` fails on a solid
                 * object, raises AttributeError on None, and `cur_cell.agents.remove(agent)` raises ValueError */
                if (cur_cell == me) e->cell[cc] = 0;
                else if (cur_cell == 0) { rc = MGO_ERR_ATTRIBUTE; continue; }
                else if (!val_can_overlap(e, cur_cell)) { rc = MGO_ERR_ASSERT; continue; }
                else if (is_agent_val(cur_cell)) {
                    int c = cur_cell - MGO_AGENT_BASE;
                    if (!list_remove(e->ag_agents[c], &e->ag_nagents[c], k)) { rc = MGO_ERR_VALUE; continue; }
                } else if (!list_remove(&e->cell_agents[cc * MGO_MAX_AGENTS], &e->cell_nagents[cc], k)) { rc = MGO_ERR_VALUE; continue; }
                /* add agent's agents to old cell — :562-569 */
                for (int li = 0; li < e->ag_nagents[k]; li++) {
                    int lb = e->ag_agents[k][li];
                    int cur_obj = e->cell[cc];
                    if (cur_obj == 0) e->cell[cc] = MGO_AGENT_BASE + lb;
                    else if (val_can_overlap(e, cur_obj)) {
                        if (is_agent_val(cur_obj)) {
                            int c = cur_obj - MGO_AGENT_BASE;
                            e->ag_agents[c][e->ag_nagents[c]++] = lb;
                        } else e->cell_agents[cc * MGO_MAX_AGENTS + e->cell_nagents[cc]++] = lb;
                    } else rc = MGO_ERR_STACK;
                }
                e->ag_nagents[k] = 0;                 /* :572 */
                /* rewards — :576-581 */
                if (fwd_cell != 0 && !is_agent_val(fwd_cell) && cfg->obj[fwd_cell].reward_kind != 0) {
                    const MgoObjDesc* o = &cfg->obj[fwd_cell];
                    double rwd = (o->reward_kind == 1) ? o->reward : bonus_reward(e, o, k);
                    if (cfg->reward_decay) rwd *= (1.0 - 0.9 * ((double)e->step_count / (double)cfg->max_steps));
                    rewards[k] += rwd;
                    /* agent.reward(rwd) — agents.py:146-153 (allow_negative_prestige=False) */
                    if (rwd >= 0) e->aprestige[k] += rwd; else e->aprestige[k] = 0;
                }
                if (fwd_cell != 0 && !is_agent_val(fwd_cell) && cfg->obj[fwd_cell].ends_episode)
                    e->adone[k] = 1;                  /* :584-585 */
            }
        } else if (action == 3) {                     /* pickup — :590-597 */
            if (fwd_cell != 0 && !is_agent_val(fwd_cell) && cfg->obj[fwd_cell].can_pickup) {
                if (e->acarry[k] == 0) {
                    e->acarry[k] = fwd_cell;
                    e->cell[fc] = 0;
                }
            }
        } else if (action == 4) {                     /* drop — :600-606 */
            if (fwd_cell == 0 && e->acarry[k] != 0) {
                e->cell[fc] = e->acarry[k];
                e->cell_nagents[fc] = 0;
                e->acarry[k] = 0;
            }
        } else if (action == 5) {                     /* toggle — :609-613 */
            if (fwd_cell != 0 && !is_agent_val(fwd_cell)) {
                const MgoObjDesc* o = &cfg->obj[fwd_cell];
                if (o->toggle_kind == 2) rc = MGO_ERR_TYPE;         /* Box.toggle arity */
                else if (o->toggle_kind == 1) {                     /* Door.toggle objects.py:333-346 */
                    if (o->state == 3) {                            /* locked */
                        int c = e->acarry[k];
                        if (c != 0 && cfg->obj[c].is_key && cfg->obj[c].color_idx == o->color_idx)
                            e->cell[fc] = o->unlock_next;
                    } else e->cell[fc] = o->toggle_next;
                }
            }
        } else if (action == 6) {                     /* done — :616-617 */
        } else {
            rc = MGO_ERR_VALUE;                       /* :619-620 */
        }
        e->aprestige[k] *= cfg->prestige_beta[k];     /* agent.on_step — base.py:622, agents.py:141-144 */
    }

    /* done agents: respawn or deactivate — base.py:627-646 */
    for (int k = 0; k < n; k++) {
        if (!e->adone[k]) continue;
        if (cfg->respawn) {
            int c = cidx(e, e->ax[k], e->ay[k]);
            int resting = e->cell[c];
            if (resting == MGO_AGENT_BASE + k) {
                if (e->ag_nagents[k] > 0) {
                    int first = e->ag_agents[k][0];
                    e->cell[c] = MGO_AGENT_BASE + first;
                    for (int li = 1; li < e->ag_nagents[k]; li++)
                        e->ag_agents[first][e->ag_nagents[first]++] = e->ag_agents[k][li];
                    /* reference quirk: agent.agents is NOT cleared on this branch (:632-634) */
                } else e->cell[c] = 0;
            } else {
                if (is_agent_val(resting)) {
                    int r = resting - MGO_AGENT_BASE;
                    list_remove(e->ag_agents[r], &e->ag_nagents[r], k);
                    for (int li = 0; li < e->ag_nagents[k]; li++) e->ag_agents[r][e->ag_nagents[r]++] = e->ag_agents[k][li];
                } else {
                    list_remove(&e->cell_agents[c * MGO_MAX_AGENTS], &e->cell_nagents[c], k);
                    for (int li = 0; li < e->ag_nagents[k]; li++)
                        e->cell_agents[c * MGO_MAX_AGENTS + e->cell_nagents[c]++] = e->ag_agents[k][li];
                }
                e->ag_nagents[k] = 0;
            }
            /* agent.reset(new_episode=False) — agents.py:161-166 */
            e->adone[k] = 0; e->aactive[k] = 0; e->ax[k] = e->ay[k] = -1; e->acarry[k] = 0;
            int prc = place_agent_spawn(e, k);
            if (prc != MGO_OK) rc = prc;
            e->aactive[k] = 1;
        } else {
            e->aactive[k] = 0;                        /* agent.deactivate() */
        }
    }

    int all_done = 1;
    for (int k = 0; k < n; k++) all_done &= (e->adone[k] != 0);
    *episode_done = (e->step_count >= cfg->max_steps) || all_done; /* :649 */
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* observations                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* occlude_mask — agents.py:298-343, literally, with the canonical reading of the unchecked
 * out-of-range accesses (reads False, writes dropped). grid/mask index [i, j] -> i*vs + j.   */
void mgo_occlude(int32_t vs, int32_t ax, int32_t ay, const uint8_t* grid, uint8_t* mask) {
    int width = vs, height = vs;
#define MK(i, j) (((i) >= 0 && (i) < width && (j) >= 0 && (j) < height) ? mask[(i) * height + (j)] : 0)
#define GR(i, j) (((i) >= 0 && (i) < width && (j) >= 0 && (j) < height) ? grid[(i) * height + (j)] : 0)
#define SET(i, j) do { if ((i) >= 0 && (i) < width && (j) >= 0 && (j) < height) mask[(i) * height + (j)] = 1; } while (0)
    memset(mask, 0, (size_t)vs * vs);
    SET(ax, ay);
    for (int j = ay + 1; j > 0; j--) {
        for (int i = ax; i < width; i++)
            if (MK(i, j) && GR(i, j)) {
                if (i < width - 1) SET(i + 1, j);
                if (j > 0) { SET(i, j - 1); if (i < width - 1) SET(i + 1, j - 1); }
            }
        for (int i = ax + 1; i > 0; i--)
            if (MK(i, j) && GR(i, j)) {
                if (i > 0) SET(i - 1, j);
                if (j > 0) { SET(i, j - 1); if (i > 0) SET(i - 1, j - 1); }
            }
    }
    for (int j = ay; j < height; j++) {
        for (int i = ax; i < width; i++)
            if (MK(i, j) && GR(i, j)) {
                if (i < width - 1) SET(i + 1, j);
                if (j < height - 1) { SET(i, j + 1); if (i < width - 1) SET(i + 1, j + 1); }
            }
        for (int i = ax + 1; i > 0; i--)
            if (MK(i, j) && GR(i, j)) {
                if (i > 0) SET(i - 1, j);
                if (j < height - 1) { SET(i, j + 1); if (i > 0) SET(i - 1, j + 1); }
            }
    }
#undef MK
#undef GR
#undef SET
}

/* rotate_grid — base.py:67-80, on a (vs,vs) int array indexed [i*vs + j] */
static void rotate_grid_i32(int vs, const int32_t* g, int rot_k, int32_t* out) {
    rot_k = ((rot_k % 4) + 4) % 4;
    for (int i = 0; i < vs; i++)
        for (int j = 0; j < vs; j++) {
            int v;
            if (rot_k == 3) v = g[j * vs + (vs - 1 - i)];        /* moveaxis(grid[:, ::-1], 0, 1) */
            else if (rot_k == 1) v = g[(vs - 1 - j) * vs + i];   /* moveaxis(grid[::-1, :], 0, 1) */
            else if (rot_k == 2) v = g[(vs - 1 - i) * vs + (vs - 1 - j)];
            else v = g[i * vs + j];
            out[i * vs + j] = v;
        }
}

/* gen_obs_grid — base.py:418-451: get_view_exts (agents.py:237-266), MultiGrid.slice
 * (base.py:123-147), opacity (base.py:103-106), process_vis (agents.py:290-295).
 * cells: the rotated sub-grid's cell values [i*vs+j]; vis: visibility mask [i*vs+j].        */
void mgo_view(const MgoEnv* e, int32_t k, uint8_t* vis, int32_t* cells) {
    const MgoConfig* cfg = &e->sh->cfg;
    int vs = cfg->view_size, off = cfg->view_offset;
    if (!e->aactive[k]) {                                /* base.py:420-425 */
        memset(vis, 0, (size_t)vs * vs);
        for (int c = 0; c < vs * vs; c++) cells[c] = 0;
        return;
    }
    int dir = e->adir[k], px = e->ax[k], py = e->ay[k];
    int topX, topY;
    if (dir == 0) { topX = px - off; topY = py - vs / 2; }
    else if (dir == 1) { topX = px - vs / 2; topY = py - off; }
    else if (dir == 2) { topX = px - vs + 1 + off; topY = py - vs / 2; }
    else { topX = px - vs / 2; topY = py - vs + 1 + off; }
    int32_t sub[MGO_MAX_VIEW * MGO_MAX_VIEW];
    for (int i = 0; i < vs; i++)
        for (int j = 0; j < vs; j++) {
            int x = topX + i, y = topY + j;
            sub[i * vs + j] = in_grid(e, x, y) ? e->cell[cidx(e, x, y)] : 0; /* zero padding */
        }
    rotate_grid_i32(vs, sub, dir + 1, cells);
    if (cfg->see_through_walls) memset(vis, 1, (size_t)vs * vs);
    else {
        uint8_t transp[MGO_MAX_VIEW * MGO_MAX_VIEW];
        for (int c = 0; c < vs * vs; c++) {
            int v = cells[c];
            transp[c] = (v == 0 || is_agent_val(v)) ? 1 : (uint8_t)cfg->obj[v].see_behind;
        }
        mgo_occlude(vs, vs / 2, vs - 1 - off, transp, vis);  /* get_view_pos agents.py:233-234 */
    }
    /* hide_item_types — base.py:441-449: after the visibility pass, a cell object whose type is
     * hidden (and that is not the viewer itself) is replaced by its first stacked agent, or None */
    uint32_t hm = cfg->hide_type_mask[k];
    if (hm) {
        int rot_k = (dir + 1) % 4;
        for (int i = 0; i < vs; i++)
            for (int j = 0; j < vs; j++) {
                int v = cells[i * vs + j];
                if (v == 0 || v == MGO_AGENT_BASE + k) continue;
                int hidden = is_agent_val(v) ? (int)((hm >> 31) & 1u) : (int)((hm >> cfg->obj[v].type_idx) & 1u);
                if (!hidden) continue;
                int repl = 0;
                if (is_agent_val(v)) {
                    int x = v - MGO_AGENT_BASE;
                    if (e->ag_nagents[x] > 0) repl = MGO_AGENT_BASE + e->ag_agents[x][0];
                } else {
                    int si, sj;   /* the world cell behind this view cell (inverse of rotate_grid) */
                    if (rot_k == 3) { si = j; sj = vs - 1 - i; }
                    else if (rot_k == 1) { si = vs - 1 - j; sj = i; }
                    else if (rot_k == 2) { si = vs - 1 - i; sj = vs - 1 - j; }
                    else { si = i; sj = j; }
                    int wc = cidx(e, topX + si, topY + sj);
                    if (e->cell_nagents[wc] > 0) repl = MGO_AGENT_BASE + e->cell_agents[wc * MGO_MAX_AGENTS];
                }
                cells[i * vs + j] = repl;
            }
    }
}

/* rotate_grid on a (ts,ts,3) tile — base.py:67-80 as used at base.py:324 */
static void blit_rotated(int ts, const uint8_t* tile, int rot_k, uint8_t* img, int P, int row0, int col0) {
    for (int r = 0; r < ts; r++)
        for (int c = 0; c < ts; c++) {
            const uint8_t* src;
            if (rot_k == 3) src = tile + (c * ts + (ts - 1 - r)) * 3;
            else if (rot_k == 1) src = tile + ((ts - 1 - c) * ts + r) * 3;
            else if (rot_k == 2) src = tile + ((ts - 1 - r) * ts + (ts - 1 - c)) * 3;
            else src = tile + (r * ts + c) * 3;
            memcpy(img + ((size_t)(row0 + r) * P + (col0 + c)) * 3, src, 3);
        }
}

static int list_has(const int32_t* lst, int n, int v) {
    for (int i = 0; i < n; i++) if (lst[i] == v) return 1;
    return 0;
}

/* gen_agent_obs ('image' style) — base.py:453-460 + MultiGrid.render base.py:301-331 +
 * render_tile base.py:275-299 */
void mgo_render_obs(const MgoEnv* e, int32_t k, uint8_t* out) {
    const MgoConfig* cfg = &e->sh->cfg;
    int vs = cfg->view_size, ts = cfg->tile_size, P = vs * ts;
    uint8_t vis[MGO_MAX_VIEW * MGO_MAX_VIEW];
    int32_t cells[MGO_MAX_VIEW * MGO_MAX_VIEW];
    uint8_t tile[64 * 64 * 3];
    mgo_view(e, k, vis, cells);
    int orientation = (((0 - (e->adir[k] + 1)) % 4) + 4) % 4;   /* base.py:130 */
    static const uint8_t SHADOW[3] = {35, 25, 30};              /* objects.py:25 */
    for (int p = 0; p < P * P; p++) memcpy(out + (size_t)p * 3, SHADOW, 3);
    /* the sub-grid shares the env's object registry, so the `.agents` lists of a cell's object
     * are the live ones; find them through the world cell that maps to this view cell. */
    for (int j = 0; j < vs; j++)
        for (int i = 0; i < vs; i++) {
            if (!vis[i * vs + j]) continue;
            int v = cells[i * vs + j];
            if (v == 0) { mgo_tile(e, 0, -1, 0, tile); }
            else if (is_agent_val(v)) {
                int x = v - MGO_AGENT_BASE;
                /* stack of agents that includes the viewer: render the viewer (:282-284) */
                int show = list_has(e->ag_agents[x], e->ag_nagents[x], k) ? k : x;
                mgo_tile(e, 0, show, e->adir[show], tile);
            } else {
                /* non-agent object; its agents list lives with its world cell */
                int found = -1;
                /* locate world cell: invert the crop/rotate by search (oracle: clarity over speed) */
                {
                    int dir = e->adir[k], px = e->ax[k], py = e->ay[k], off = cfg->view_offset;
                    int topX, topY;
                    if (dir == 0) { topX = px - off; topY = py - vs / 2; }
                    else if (dir == 1) { topX = px - vs / 2; topY = py - off; }
                    else if (dir == 2) { topX = px - vs + 1 + off; topY = py - vs / 2; }
                    else { topX = px - vs / 2; topY = py - vs + 1 + off; }
                    int rot_k = (dir + 1) % 4, si, sj;
                    if (rot_k == 3) { si = j; sj = vs - 1 - i; }
                    else if (rot_k == 1) { si = vs - 1 - j; sj = i; }
                    else if (rot_k == 2) { si = vs - 1 - i; sj = vs - 1 - j; }
                    else { si = i; sj = j; }
                    found = cidx(e, topX + si, topY + sj);
                }
                int na = e->cell_nagents[found];
                if (na > 0) {
                    const int32_t* lst = &e->cell_agents[found * MGO_MAX_AGENTS];
                    int show = list_has(lst, na, k) ? k : lst[0];   /* :289-294 */
                    mgo_tile(e, v, show, e->adir[show], tile);
                } else mgo_tile(e, v, -1, 0, tile);
            }
            blit_rotated(ts, tile, orientation, out, P, j * ts, i * ts);   /* :319-324 */
        }
}

/* MultiGrid.encode — base.py:196-214; WorldObj.encode objects.py:90-99 */
void mgo_encode(const MgoEnv* e, const uint8_t* vis, uint8_t* out) {
    const MgoConfig* cfg = &e->sh->cfg;
    for (int i = 0; i < cfg->W; i++)
        for (int j = 0; j < cfg->H; j++) {
            uint8_t* o = out + ((size_t)i * cfg->H + j) * 3;
            o[0] = o[1] = o[2] = 0;
            if (vis && !vis[i * cfg->H + j]) continue;
            int v = e->cell[cidx(e, i, j)];
            if (v == 0) continue;
            if (is_agent_val(v)) {
                int k = v - MGO_AGENT_BASE;
                o[0] = (uint8_t)cfg->agent_type_idx; o[1] = (uint8_t)cfg->agent_color_idx[k]; o[2] = (uint8_t)e->adir[k];
            } else {
                o[0] = (uint8_t)cfg->obj[v].type_idx; o[1] = (uint8_t)cfg->obj[v].color_idx; o[2] = (uint8_t)cfg->obj[v].state;
            }
        }
}

/* ------------------------------------------------------------------------------------------ */
/* state access                                                                                */
/* ------------------------------------------------------------------------------------------ */

void mgo_get_state(const MgoEnv* e, uint8_t* base, int32_t* agents7, int32_t* step_count) {
    const MgoConfig* cfg = &e->sh->cfg;
    int nc = cfg->W * cfg->H;
    for (int c = 0; c < nc; c++) base[c] = is_agent_val(e->cell[c]) ? 0 : (uint8_t)e->cell[c];
    for (int k = 0; k < cfg->n_agents; k++) {
        int32_t* a = agents7 + k * 7;
        a[0] = e->ax[k]; a[1] = e->ay[k]; a[2] = e->adir[k]; a[3] = e->aactive[k]; a[4] = e->adone[k];
        a[5] = e->acarry[k];
        int ord = -1;
        if (e->ax[k] >= 0) {
            int c = cidx(e, e->ax[k], e->ay[k]);
            int v = e->cell[c];
            if (v == MGO_AGENT_BASE + k) ord = 0;
            else if (is_agent_val(v)) {
                int t = v - MGO_AGENT_BASE;
                for (int i = 0; i < e->ag_nagents[t]; i++) if (e->ag_agents[t][i] == k) ord = 1 + i;
            } else {
                for (int i = 0; i < e->cell_nagents[c]; i++) if (e->cell_agents[c * MGO_MAX_AGENTS + i] == k) ord = i;
            }
        }
        a[6] = ord;
    }
    *step_count = e->step_count;
}

void mgo_get_prestige(const MgoEnv* e, double* out) {
    for (int k = 0; k < e->sh->cfg.n_agents; k++) out[k] = e->aprestige[k];
}

/* the non-image fields of a 'rich' observation — base.py:461-471: reward is `agent.step_reward`, which
 * step() only ever sets to 0 (base.py:519) and which does not exist before the first step (getattr
 * default 0); position is pos / (W, H) in float64 with `pos is None -> (0, 0)`; orientation is dir. */
void mgo_rich_obs(const MgoEnv* e, int32_t k, double* reward, double* position2, int32_t* orientation) {
    const MgoConfig* cfg = &e->sh->cfg;
    const int placed = e->ax[k] >= 0;
    *reward = 0.0;
    position2[0] = (double)(placed ? e->ax[k] : 0) / (double)cfg->W;
    position2[1] = (double)(placed ? e->ay[k] : 0) / (double)cfg->H;
    *orientation = e->adir[k];
}

void mgo_get_mt(const MgoEnv* e, uint32_t* mt624, int32_t* pos) {
    memcpy(mt624, e->mt, sizeof(e->mt));
    *pos = e->mt_pos;
}

void mgo_set_agent_dir(MgoEnv* e, int32_t k, int32_t dir) { e->adir[k] = ((dir % 4) + 4) % 4; }
void mgo_set_carrying(MgoEnv* e, int32_t k, int32_t obj) { e->acarry[k] = obj; }

/* test helper: a fresh `_gen_grid` with every agent lifted off the grid (pos None, lists empty);
 * dir / active / done / carrying are kept.  The caller re-seats agents with mgo_place_agent_at. */
int32_t mgo_regen_grid(MgoEnv* e, int32_t which_gen) {
    for (int k = 0; k < e->sh->cfg.n_agents; k++) { e->ag_nagents[k] = 0; e->ax[k] = e->ay[k] = -1; }
    return gen_grid(e, which_gen);
}

static void lift_agent(MgoEnv* e, int k);

/* live place_obj / try_place_obj (base.py:664-708).  what >= 1: object id; what < 0: agent -(what+1),
 * lifted off the grid first (its stack re-seated like a move-out) and activated when it lands. */
int32_t mgo_place_obj(MgoEnv* e, int32_t what, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t max_tries,
                      const uint8_t* reject, int32_t* out_xy) {
    int val = what > 0 ? what : MGO_AGENT_BASE + (-(what + 1));
    if (what < 0) { lift_agent(e, -(what + 1)); e->aactive[-(what + 1)] = 0; }
    int ox, oy;
    int rc = place_obj_in(e, val, (double)max_tries, x0, y0, x1, y1, reject, &ox, &oy);
    if (rc == MGO_OK && what < 0) e->aactive[-(what + 1)] = 1;
    if (out_xy) { out_xy[0] = ox; out_xy[1] = oy; }
    return rc;
}

int32_t mgo_try_place_obj(MgoEnv* e, int32_t what, int32_t x, int32_t y) {
    int val = what > 0 ? what : MGO_AGENT_BASE + (-(what + 1));
    if (what < 0) { lift_agent(e, -(what + 1)); e->aactive[-(what + 1)] = 0; }
    if (!in_grid(e, x, y)) return 0;
    int ok = try_place(e, val, x, y);
    if (ok && what < 0) e->aactive[-(what + 1)] = 1;
    return ok;
}

static void lift_agent(MgoEnv* e, int k) {
    int me = MGO_AGENT_BASE + k;
    if (e->ax[k] >= 0) {
        /* take the agent out of its current cell, re-seating its stack like a move-out */
        int c = cidx(e, e->ax[k], e->ay[k]);
        int v = e->cell[c];
        if (v == me) {
            e->cell[c] = 0;
            for (int li = 0; li < e->ag_nagents[k]; li++) {
                int lb = e->ag_agents[k][li];
                if (e->cell[c] == 0) e->cell[c] = MGO_AGENT_BASE + lb;
                else { int t = e->cell[c] - MGO_AGENT_BASE; e->ag_agents[t][e->ag_nagents[t]++] = lb; }
            }
            e->ag_nagents[k] = 0;
        } else if (is_agent_val(v)) list_remove(e->ag_agents[v - MGO_AGENT_BASE], &e->ag_nagents[v - MGO_AGENT_BASE], k);
        else list_remove(&e->cell_agents[c * MGO_MAX_AGENTS], &e->cell_nagents[c], k);
        e->ax[k] = e->ay[k] = -1;
    }
}

int32_t mgo_place_agent_at(MgoEnv* e, int32_t k, int32_t x, int32_t y) {
    if (!in_grid(e, x, y)) return MGO_ERR_ASSERT;
    lift_agent(e, k);
    return try_place(e, MGO_AGENT_BASE + k, x, y) ? MGO_OK : MGO_ERR_VALUE;
}

/* ------------------------------------------------------------------------------------------ */
/* batch stepping for the cpu_baseline leg                                                     */
/* ------------------------------------------------------------------------------------------ */

int32_t mgo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int32_t mgo_batch_step(MgoEnv** envs, int32_t B, const int32_t* actions, double* rewards,
                       uint8_t* done, uint8_t* obs_or_null, int32_t auto_reset, int32_t threads) {
    int worst = 0;
    if (B <= 0) return 0;
    int n = envs[0]->sh->cfg.n_agents;
    int P = envs[0]->sh->cfg.view_size * envs[0]->sh->cfg.tile_size;
    size_t img = (size_t)P * P * 3;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static) reduction(min : worst)
#endif
    for (int b = 0; b < B; b++) {
        int32_t d = 0;
        int rc = mgo_step(envs[b], actions + (size_t)b * n, rewards + (size_t)b * n, &d, NULL);
        done[b] = (uint8_t)d;
        if (d && auto_reset) { int r2 = mgo_reset(envs[b], 1); if (r2 < rc) rc = r2; }
        if (obs_or_null)
            for (int k = 0; k < n; k++) mgo_render_obs(envs[b], k, obs_or_null + ((size_t)b * n + k) * img);
        if (rc < worst) worst = rc;
    }
    return worst;
}
