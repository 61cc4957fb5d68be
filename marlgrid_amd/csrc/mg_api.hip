// mg_api.hip — the C ABI of libmarlgrid_hip.so (include/marlgrid_hip.h): argument checks and
// launches only; every buffer is caller-owned device memory, every call is asynchronous on the
// caller's stream.
#include <stdio.h>

#include "mg_device.h"
#include "mg_launch.h"

namespace {

int check_cfg(const MgConfig* c) {
    if (!c) return MG_E_ARG;
    if (c->B < 0 || c->W < 3 || c->H < 3 || c->W > 255 || c->H > 255) return MG_E_ARG;
    if (c->n_agents < 1 || c->n_agents > MG_MAX_AGENTS) return MG_E_UNSUPPORTED;
    if (c->view_size < 1 || c->view_size > MG_MAX_VIEW) return MG_E_UNSUPPORTED;   // (even sizes included: agents.py:233-266 is written with view_size // 2)
    if (c->tile_size < 1 || c->tile_size > 64) return MG_E_UNSUPPORTED;
    if (c->view_offset < 0 || c->view_offset >= c->view_size) return MG_E_ARG;
    if (c->cells_stride < c->W * c->H || (c->cells_stride & 15)) return MG_E_ARG;
    if (c->n_obj < 1 || c->n_obj > MG_MAX_OBJ) return MG_E_UNSUPPORTED;
    if (c->n_tiles < 1 + c->n_obj + c->n_ovl_slots * c->n_agents * 4) return MG_E_ARG;
    if (c->prestige_mask && (c->prestige_sprite_tile < 0 || c->prestige_sprite_tile + 4 > c->n_tiles)) return MG_E_ARG;
    if (!c->obj || !c->atlas) return MG_E_ARG;
    if (c->max_steps < 1) return MG_E_ARG;
    // the viewer subset of a launch (agents that differ in view geometry are rendered group by group)
    if (c->n_view < 0 || c->n_view > c->n_agents) return MG_E_ARG;
    for (int i = 0; i < c->n_view; i++)
        if (c->view_agent[i] >= c->n_agents) return MG_E_ARG;
    if (c->spawn_x0 < 0 || c->spawn_y0 < 0 || c->spawn_x1 > c->W || c->spawn_y1 > c->H || c->spawn_x1 <= c->spawn_x0 ||
        c->spawn_y1 <= c->spawn_y0 || c->spawn_max_tries < 1 || c->spawn_max_tries > 100000)
        return MG_E_ARG;
    return MG_OK;
}

int check_prog(const MgConfig* cfg, const MgGenProgram* prog) {
    // (the ops themselves are device memory — MgGenProgram::ops —: what they must satisfy is the caller's to guarantee)
    (void)cfg;
    if (!prog || !prog->template_grid || prog->n_ops < 0 || prog->n_ops > MG_MAX_GEN) return MG_E_ARG;
    if (prog->n_ops > 0 && !prog->ops) return MG_E_ARG;
    if (prog->n_reject < 0 || (prog->n_reject > 0 && !prog->reject)) return MG_E_ARG;
    return MG_OK;
}

int check_state(const MgState* s) {
    if (!s || !s->grid || !s->agents || !s->mt || !s->mt_pos || !s->mt_head || !s->step_count || !s->done || !s->error)
        return MG_E_ARG;
    return MG_OK;
}

int check_both(const MgConfig* c, const MgState* s) {
    int e = check_cfg(c);
    if (e) return e;
    e = check_state(s);
    if (e) return e;
    if (c->prestige_mask && !s->prestige) return MG_E_ARG;
    return MG_OK;
}

int rc(hipError_t e) { return e == hipSuccess ? MG_OK : MG_E_LAUNCH; }

}  // namespace

extern "C" {

int32_t mg_abi_version(void) { return MG_ABI_VERSION; }

int32_t mg_struct_sizes(int32_t out[7]) {
    if (!out) return MG_E_ARG;
    out[0] = (int32_t)sizeof(MgConfig);
    out[1] = (int32_t)sizeof(MgState);
    out[2] = (int32_t)sizeof(MgObjDesc);
    out[3] = (int32_t)sizeof(MgGenOp);
    out[4] = (int32_t)sizeof(MgGenProgram);
    out[5] = (int32_t)sizeof(MgPlaceTuning);
    out[6] = (int32_t)sizeof(MgPlaceStats);
    return 7;
}

#define MG_STR2(x) #x
#define MG_STR(x) MG_STR2(x)
#ifndef MG_BUILD_ID
#define MG_BUILD_ID "unknown"
#endif
const char* mg_build_info(void) {
#if defined(MG_AB_VARIANTS)
    return "libmarlgrid_hip_ab gfx950 abi" MG_STR(MG_ABI_VERSION) " " MG_BUILD_ID " (measurement variants: tools/ only)";
#else
    return "libmarlgrid_hip gfx950 abi" MG_STR(MG_ABI_VERSION) " " MG_BUILD_ID;
#endif
}

const char* mg_error_string(int32_t code) {
    switch (code) {
    case MG_OK: return "ok";
    case MG_E_ARG: return "invalid argument";
    case MG_E_UNSUPPORTED: return "unsupported configuration";
    case MG_E_LAUNCH: return "HIP launch failed";
    case MG_E_NOMEM: return "out of device memory";
    case MG_ERR_VALUE: return "ValueError: environment can't handle action";
    case MG_ERR_RECURSION: return "RecursionError: rejection sampling failed in place_obj";
    case MG_ERR_TYPE: return "TypeError: toggle() arity (Box)";
    case MG_ERR_ASSERT: return "AssertionError: grid access out of bounds";
    case MG_ERR_ATTRIBUTE: return "AttributeError: the cell under an agent was emptied by put_obj(None)";
    default: return "unknown";
    }
}

int32_t mg_mt_seed(int32_t B, const uint32_t* keys, const int32_t* key_len, uint32_t* mt, int32_t* mt_pos,
                   uint32_t* mt_head, void* stream) {
    if (B < 0 || !keys || !key_len || !mt || !mt_pos || !mt_head) return MG_E_ARG;
    return rc(mg::launch_mt_seed(B, keys, key_len, mt, mt_pos, mt_head, (hipStream_t)stream));
}

int32_t mg_reset(const MgConfig* cfg, const MgState* st, const MgGenProgram* prog, const uint8_t* env_mask,
                 void* stream) {
    int e = check_both(cfg, st);
    if (e) return e;
    e = check_prog(cfg, prog);
    if (e) return e;
    return rc(mg::launch_reset(*cfg, *st, *prog, env_mask, (hipStream_t)stream));
}

int32_t mg_step(const MgConfig* cfg, const MgState* st, const void* actions, int32_t action_bytes, float* rewards,
                const MgGenProgram* auto_reset, void* stream) {
    int e = check_both(cfg, st);
    if (e) return e;
    if (!actions || !rewards) return MG_E_ARG;
    if (action_bytes != 1 && action_bytes != 4 && action_bytes != 8) return MG_E_ARG;
    if (auto_reset && (e = check_prog(cfg, auto_reset))) return e;
    return rc(mg::launch_step(*cfg, *st, actions, action_bytes, rewards, auto_reset, (hipStream_t)stream));
}

int32_t mg_render_obs(const MgConfig* cfg, const MgState* st, uint8_t* obs, uint8_t* view_cells,
                      uint8_t* view_agent, uint8_t* vis_mask, void* stream) {
    int e = check_both(cfg, st);
    if (e) return e;
    if (!obs) return MG_E_ARG;
    return rc(mg::launch_render(*cfg, *st, obs, view_cells, view_agent, vis_mask, (hipStream_t)stream));
}

static int32_t step_render(const MgConfig* cfg, const MgState* st, const void* actions, int32_t action_bytes, float* rewards,
                           const MgGenProgram* auto_reset, uint8_t* obs, uint8_t* encode_out, void* stream) {
    int e = check_both(cfg, st);
    if (e) return e;
    if (!actions || !rewards || !obs) return MG_E_ARG;
    if (action_bytes != 1 && action_bytes != 4 && action_bytes != 8) return MG_E_ARG;
    if (cfg->n_view != 0) return MG_E_ARG;   // one launch steps AND renders every agent: view groups take mg_step + mg_render_obs
    if (auto_reset && (e = check_prog(cfg, auto_reset))) return e;
    if (encode_out && !mg::render_can_encode(*cfg)) return MG_E_UNSUPPORTED;
    mg::FusedStep fs;
    fs.actions = actions;
    fs.rewards = rewards;
    fs.action_bytes = action_bytes;
    fs.enabled = 1;
    fs.has_prog = auto_reset ? 1 : 0;
    if (auto_reset) fs.prog = *auto_reset;
    else { fs.prog.template_grid = nullptr; fs.prog.n_ops = 0; fs.prog.ops = nullptr; fs.prog.reject = nullptr; fs.prog.n_reject = 0; }
    fs.encode_out = encode_out;
    const uint32_t cells = (uint32_t)(cfg->W * cfg->H), n = (uint32_t)cfg->n_agents;
    fs.enc_m_cells = (uint32_t)((0x100000000ull + cells - 1) / cells);
    fs.enc_m_n = (uint32_t)((0x100000000ull + n - 1) / n);
    fs.enc_ne = 0;        // (launch_render fills it in for the instantiations that use it)
    return rc(mg::launch_render(*cfg, *st, obs, nullptr, nullptr, nullptr, (hipStream_t)stream, &fs));
}

int32_t mg_step_render(const MgConfig* cfg, const MgState* st, const void* actions, int32_t action_bytes, float* rewards,
                       const MgGenProgram* auto_reset, uint8_t* obs, void* stream) {
    return step_render(cfg, st, actions, action_bytes, rewards, auto_reset, obs, nullptr, stream);
}

int32_t mg_step_render_encode(const MgConfig* cfg, const MgState* st, const void* actions, int32_t action_bytes, float* rewards,
                              const MgGenProgram* auto_reset, uint8_t* obs, uint8_t* encode_out, void* stream) {
    if (!encode_out) return MG_E_ARG;
    return step_render(cfg, st, actions, action_bytes, rewards, auto_reset, obs, encode_out, stream);
}

int32_t mg_encode(const MgConfig* cfg, const MgState* st, const uint8_t* vis_mask, uint8_t* out, void* stream) {
    int e = check_cfg(cfg);
    if (e) return e;
    e = check_state(st);
    if (e) return e;
    if (!out) return MG_E_ARG;
    return rc(mg::launch_encode(*cfg, *st, vis_mask, out, (hipStream_t)stream));
}

int32_t mg_put_obj(const MgConfig* cfg, const MgState* st, int32_t obj, int32_t x, int32_t y,
                   const uint8_t* env_mask, void* stream) {
    int e = check_cfg(cfg);
    if (e) return e;
    e = check_state(st);
    if (e) return e;
    if (obj < 0 || obj >= cfg->n_obj || x < 0 || x >= cfg->W || y < 0 || y >= cfg->H) return MG_E_ARG;
    return rc(mg::launch_put_obj(*cfg, *st, obj, x, y, env_mask, (hipStream_t)stream));
}

int32_t mg_place(const MgConfig* cfg, const MgState* st, int32_t what, int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                 int32_t max_tries, const int32_t* fixed_pos, const uint8_t* env_mask, const uint8_t* reject,
                 int32_t* out_pos, uint8_t* out_ok, void* stream) {
    int e = check_cfg(cfg);
    if (e) return e;
    e = check_state(st);
    if (e) return e;
    if (what == 0 || what >= cfg->n_obj || -(what + 1) >= cfg->n_agents) return MG_E_ARG;
    if (!fixed_pos && (x0 < 0 || y0 < 0 || x1 > cfg->W || y1 > cfg->H || x1 <= x0 || y1 <= y0 || max_tries < 1))
        return MG_E_ARG;
    return rc(mg::launch_place(*cfg, *st, what, x0, y0, x1, y1, max_tries, fixed_pos, env_mask, reject, out_pos, out_ok,
                               (hipStream_t)stream));
}

int32_t mg_render_frame(const MgConfig* cfg, const MgState* st, const int32_t* env_ids, int32_t n_envs,
                        const uint8_t* frame_atlas, int32_t frame_tile_size, int32_t highlight, uint32_t frame_amax,
                        uint8_t* out, void* stream) {
    int e = check_cfg(cfg);
    if (e) return e;
    e = check_state(st);
    if (e) return e;
    if (!env_ids || n_envs < 0 || !frame_atlas || !out) return MG_E_ARG;
    if (frame_tile_size < 4 || frame_tile_size > 64 || (frame_tile_size & 3)) return MG_E_UNSUPPORTED;
    if (cfg->prestige_mask && !st->prestige) return MG_E_ARG;
    return rc(mg::launch_frame(*cfg, *st, env_ids, n_envs, frame_atlas, frame_tile_size, highlight, frame_amax, out,
                               (hipStream_t)stream));
}

int32_t mg_render_obs_lds_bytes(const MgConfig* cfg) {
    if (!cfg || cfg->n_agents < 1 || cfg->n_agents > MG_MAX_AGENTS || cfg->view_size < 1 || cfg->view_size > MG_MAX_VIEW ||
        cfg->tile_size < 1 || cfg->tile_size > 64 || cfg->cells_stride < 0 || cfg->n_tiles < 0)
        return MG_E_ARG;
    return mg::render_min_lds_bytes(*cfg);
}

int32_t mg_render_kernel_name(const MgConfig* cfg, char* out, int32_t cap) {
    if (!cfg || !out || cap < 1 || cfg->B < 1 || cfg->n_agents < 1 || cfg->n_agents > MG_MAX_AGENTS || cfg->view_size < 1 ||
        cfg->view_size > MG_MAX_VIEW || cfg->tile_size < 1 || cfg->tile_size > 64 || cfg->cells_stride < 0 || cfg->n_tiles < 0)
        return MG_E_ARG;
    mg::RenderPick pick = {0, 0, 0, 0, 0, 0};
    MgState none = {};
    const hipError_t e = mg::launch_render(*cfg, none, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &pick);
    if (e != hipSuccess) { out[0] = 0; return MG_E_LAUNCH; }      // (the configuration does not fit LDS: mg_render_obs_lds_bytes)
    snprintf(out, (size_t)cap, "mg::render_kernel<%d, %d, %d, %d, %d>", pick.vs, pick.ts, pick.wpb, pick.v, pick.rm);
    return (pick.vs == 0 ? 1 : 0) | (pick.ts == 0 ? 2 : 0);
}

int32_t mg_time_render_obs(const MgConfig* cfg, const MgState* st, uint8_t* obs, int32_t iters, float* avg_ms,
                           void* stream) {
    int e = check_cfg(cfg);
    if (e) return e;
    e = check_state(st);
    if (e) return e;
    if (!obs || !avg_ms || iters < 1) return MG_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t t0, t1;
    if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess) return MG_E_LAUNCH;
    hipError_t err = mg::launch_render(*cfg, *st, obs, nullptr, nullptr, nullptr, s);   // warm
    (void)hipEventRecord(t0, s);
    for (int i = 0; i < iters && err == hipSuccess; i++) err = mg::launch_render(*cfg, *st, obs, nullptr, nullptr, nullptr, s);
    (void)hipEventRecord(t1, s);
    (void)hipEventSynchronize(t1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, t0, t1);
    *avg_ms = ms / (float)iters;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return rc(err);
}

int32_t mg_host_flag_alloc(int32_t** host, int32_t** dev) {
    if (!host || !dev) return MG_E_ARG;
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return MG_E_LAUNCH;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return MG_E_LAUNCH; }
    *static_cast<volatile int32_t*>(h) = 0;
    *host = static_cast<int32_t*>(h);
    *dev = static_cast<int32_t*>(d);
    return MG_OK;
}

int32_t mg_host_flag_free(int32_t* host) {
    if (!host) return MG_OK;
    return rc(hipHostFree(host));
}

void* mg_obs_alloc(uint64_t bytes, int32_t device) {
    if (bytes == 0) return nullptr;
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != device && hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    void* p = nullptr;
    const hipError_t e = hipMalloc(&p, (size_t)bytes);
    if (cur != device) (void)hipSetDevice(cur);
    if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

int32_t mg_obs_free(void* ptr) { return ptr ? rc(hipFree(ptr)) : MG_OK; }

}  // extern "C"
