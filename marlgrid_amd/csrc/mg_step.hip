// mg_step.hip — batched MultiGridEnv.step action loop (marlgrid/base.py:501-649).
//
// Per env the reference is strictly sequential: agents act in a freshly shuffled order
// (base.py:514-516) and each action sees the grid left by the previous one, so the only parallel
// axis is the env batch.  One LANE per env (64 envs per wave, 256 per workgroup): a wave-per-env
// mapping would idle 63 of 64 lanes through ~100 scalar instructions.  The env's agent records are
// staged in LDS as [agent][lane] (conflict-free 8-byte columns) so they can be indexed dynamically
// by the shuffled order; the grid is touched in place in HBM (<= 2 cells per agent).
//
// Flat state model that reproduces the reference's object graph: a cell holds a
// base object id (0 = none) and any number of agents; the ordered stack the reference keeps in
// `obj.agents` lists (append on entry base.py:547-552, remove on exit :555-559, re-seat left-behind
// agents in order :562-569) is exactly "agents in this cell sorted by arrival", carried as a rank
// permutation: a successful move gives the mover the highest rank.
#include "mg_device.h"
#include <stdlib.h>

#include "mg_launch.h"

namespace mg {

template <typename ActT, int BS>
__global__ __launch_bounds__(BS) void step_kernel(MgConfig cfg, MgState st, const ActT* __restrict__ actions,
                                                  float* __restrict__ rewards) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_rec[];                      // [n][BS]
    uint8_t* s_order = reinterpret_cast<uint8_t*>(s_rec + (size_t)cfg.n_agents * BS);      // [n][BS]
    uint8_t* s_act = s_order + (size_t)cfg.n_agents * BS;                                   // [n][BS] action (0xFF: invalid)
    uint8_t* s_fb = s_act + (size_t)cfg.n_agents * BS;                                      // [n][BS] front-cell base id
    uint8_t* s_oflags = s_fb + (size_t)cfg.n_agents * BS;                                   // [MG_MAX_OBJ]
    const int tid = threadIdx.x;
    const int b = blockIdx.x * BS + tid;
    const bool live = b < cfg.B;
    const int n = cfg.n_agents, W = cfg.W, H = cfg.H;

    // ---- round trip 1: everything whose address is known up front ----
    if (tid < MG_MAX_OBJ) s_oflags[tid] = (tid < cfg.n_obj && tid > 0) ? cfg.obj[tid].flags : 0;
    int pos0 = 0, sc0 = 0;
    if (live) {
        for (int k = 0; k < n; k++) s_rec[k * BS + tid] = st.agents[(size_t)b * n + k];
        for (int k = 0; k < n; k++) {
            const long long a = (long long)actions[(size_t)b * n + k];
            s_act[k * BS + tid] = (a >= 0 && a <= 6) ? (uint8_t)a : (uint8_t)0xFF;
        }
        pos0 = st.mt_pos[b];
        sc0 = st.step_count[b];
    }
    __syncthreads();
    if (!live) return;
    uint8_t* g = st.grid + (size_t)b * cfg.cells_stride;
    uint32_t* mtw = st.mt + (size_t)b * MG_MT_N;

    Mt mt{mtw, pos0};
    int err = 0;
    // place_obj(agent) (base.py:690-708 over try_place_obj :664-688) for an agent that is off the
    // grid: rejection-sample a cell whose object can be overlapped (or that is empty) and, without
    // ghost_mode, holds no agent; arrival gives the highest rank.  Used by late spawns and respawns.
    auto place_agent = [&](int k) {
        uint64_t r = s_rec[k * BS + tid];
        bool ok = false;
        for (int t = 0; t < 100000; t++) {                    // place_obj default max_tries = 1e5
            const int x = (int)mt.bounded((uint32_t)(W - 1));
            const int y = (int)mt.bounded((uint32_t)(H - 1));
            const uint32_t base = g[x * H + y];
            const uint32_t xy = (uint32_t)x | ((uint32_t)y << 8);
            int cnt = 0;
            for (int j = 0; j < n; j++) {
                const uint64_t rj = s_rec[j * BS + tid];
                cnt += ((rec_byte(rj, MG_AG_FLAGS) & MG_AF_PLACED) && rec_xy(rj) == xy) ? 1 : 0;
            }
            if ((base == 0 || (s_oflags[base] & MG_OF_CAN_OVERLAP)) && (cnt == 0 || (cfg.ghost_mode & 2))) {
                const uint32_t old_rank = rec_byte(r, MG_AG_RANK);
                for (int j = 0; j < n; j++) {
                    const uint64_t rj = s_rec[j * BS + tid];
                    const uint32_t rk = rec_byte(rj, MG_AG_RANK);
                    if (rk > old_rank) s_rec[j * BS + tid] = rec_set(rj, MG_AG_RANK, rk - 1);
                }
                r = rec_set(r, MG_AG_RANK, (uint32_t)(n - 1));
                r = rec_set(r, MG_AG_X, (uint32_t)x);
                r = rec_set(r, MG_AG_Y, (uint32_t)y);
                r = rec_set(r, MG_AG_FLAGS, MG_AF_ACTIVE | MG_AF_PLACED);
                ok = true;
                break;
            }
        }
        if (!ok) err = err ? err : MG_ERR_RECURSION;
        s_rec[k * BS + tid] = r;
    };

    // late spawns (base.py:503-506), before step_count is incremented and before the shuffle: any agent
    // that is neither active nor done (spawn_delay not reached at reset, or lifted off the grid by a
    // failed live placement) is placed as soon as step_count >= its spawn_delay
    {
        for (int k = 0; k < n; k++) {
            const uint32_t f = rec_byte(s_rec[k * BS + tid], MG_AG_FLAGS);
            if (!(f & (MG_AF_ACTIVE | MG_AF_DONE)) && sc0 >= cfg.spawn_delay[k]) place_agent(k);
        }
    }

    // ---- round trip 2: every agent's front cell.  An agent's position
    // and heading are only ever changed by its own action, so its front cell is known before the
    // loop; the cell's *content* can only be changed by a pickup / drop / toggle earlier in this
    // step (grid_dirty), in which case it is re-read. ----
    for (int k = 0; k < n; k++) {
        const uint64_t r = s_rec[k * BS + tid];
        const int dir = (int)rec_byte(r, MG_AG_DIR);
        const int fx = (int)rec_byte(r, MG_AG_X) + dir_dx(dir), fy = (int)rec_byte(r, MG_AG_Y) + dir_dy(dir);
        const bool ok = (rec_byte(r, MG_AG_FLAGS) & MG_AF_ACTIVE) && fx >= 0 && fx < W && fy >= 0 && fy < H;
        s_fb[k * BS + tid] = ok ? g[fx * H + fy] : (uint8_t)0;
    }
    bool grid_dirty = false;

    const int step_count = sc0 + 1;   // base.py:512
    // reward decay factor, float64 like the reference (base.py:579)
    const double decay = cfg.reward_decay ? (1.0 - 0.9 * ((double)step_count / (double)cfg.max_steps)) : 1.0;

    // iter_order = arange(n); np_random.shuffle(iter_order)  (base.py:514-516): legacy Fisher-Yates
    // over numpy's masked-rejection bounded draws
    for (int k = 0; k < n; k++) s_order[k * BS + tid] = (uint8_t)k;
    for (int i = n - 1; i >= 1; i--) {
        const int j = (int)mt.bounded((uint32_t)i);
        uint8_t t = s_order[i * BS + tid];
        s_order[i * BS + tid] = s_order[j * BS + tid];
        s_order[j * BS + tid] = t;
    }

    for (int oi = 0; oi < n; oi++) {
        const int k = s_order[oi * BS + tid];
        float rew = 0.0f;
        bool rewarded = false;      // agent.reward(rwd) was called (prestige bookkeeping)
        double rwd_applied = 0.0;
        uint64_t r = s_rec[k * BS + tid];
        const uint32_t flags = rec_byte(r, MG_AG_FLAGS);
        if (flags & MG_AF_ACTIVE) {   // base.py:521
            const int action = (int)s_act[k * BS + tid];
            const int cx = (int)rec_byte(r, MG_AG_X), cy = (int)rec_byte(r, MG_AG_Y);
            const int dir = (int)rec_byte(r, MG_AG_DIR);
            const int fx = cx + dir_dx(dir), fy = cy + dir_dy(dir);   // agent.front_pos agents.py:194-198
            if (fx < 0 || fx >= W || fy < 0 || fy >= H) {
                err = err ? err : MG_ERR_ASSERT;   // grid.get asserts (base.py:154-156)
            } else {
                const int fcell = fx * H + fy;
                const uint32_t fbase = grid_dirty ? (uint32_t)g[fcell] : (uint32_t)s_fb[k * BS + tid];
                const uint32_t fxy = (uint32_t)fx | ((uint32_t)fy << 8);
                const uint32_t fflags = s_oflags[fbase];
                if (action == 0) {                                   // left  base.py:530-531
                    r = rec_set(r, MG_AG_DIR, (uint32_t)((dir + 3) & 3));
                } else if (action == 1) {                            // right :534-535
                    r = rec_set(r, MG_AG_DIR, (uint32_t)((dir + 1) & 3));
                } else if (action == 2 || action == 4) {             // forward :538-585 / drop :600-606
                    int agents_there = 0;
                    for (int j = 0; j < n; j++) {
                        uint64_t rj = s_rec[j * BS + tid];
                        agents_there += ((rec_byte(rj, MG_AG_FLAGS) & MG_AF_PLACED) && rec_xy(rj) == fxy) ? 1 : 0;
                    }
                    if (action == 2) {
                        // fwd_cell is None, or it can_overlap(); the top object is the base object if
                        // there is one, else the first agent standing there (agents overlap)
                        bool can_move = fbase ? (fflags & MG_OF_CAN_OVERLAP) != 0 : true;
                        if (!(cfg.ghost_mode & 1) && fbase == 0 && agents_there > 0) can_move = false;  // :541-542
                        if (can_move) {
                            // arrival: highest rank; everyone above the old rank slides down
                            const uint32_t old_rank = rec_byte(r, MG_AG_RANK);
                            for (int j = 0; j < n; j++) {
                                uint64_t rj = s_rec[j * BS + tid];
                                uint32_t rk = rec_byte(rj, MG_AG_RANK);
                                if (rk > old_rank) s_rec[j * BS + tid] = rec_set(rj, MG_AG_RANK, rk - 1);
                            }
                            r = rec_set(r, MG_AG_RANK, (uint32_t)(n - 1));
                            r = rec_set(r, MG_AG_X, (uint32_t)fx);
                            r = rec_set(r, MG_AG_Y, (uint32_t)fy);
                            if (fbase) {
                                const MgObjDesc od = cfg.obj[fbase];
                                if (od.reward_kind) {                 // hasattr(fwd_cell,'get_reward') :576-581
                                    double rwd;
                                    if (od.reward_kind == 1) {
                                        rwd = od.reward;              // Goal.get_reward objects.py:219-220
                                    } else {                          // BonusTile.get_reward objects.py:180-206
                                        int bs = (int)rec_byte(r, MG_AG_BONUS);
                                        bool first_bonus = false;
                                        int nb = od.n_bonus ? od.n_bonus : 1;
                                        if (bs == 0xFF) { bs = ((int)od.bonus_id - 1 + nb) % nb; first_bonus = true; }
                                        if (bs == od.bonus_id) rwd = -fabs(od.penalty);
                                        else if ((bs + 1) % nb == od.bonus_id) { bs = od.bonus_id; rwd = od.reward; }
                                        else rwd = -fabs(od.penalty);
                                        if (od.bonus_flags & 2) bs = od.bonus_id;
                                        if (first_bonus && !(od.bonus_flags & 1)) rwd = 0.0;
                                        r = rec_set(r, MG_AG_BONUS, (uint32_t)bs);
                                    }
                                    rwd_applied = rwd * decay;
                                    rewarded = true;
                                    rew = (float)rwd_applied;
                                }
                                if (fflags & MG_OF_ENDS_EPISODE) r = rec_set(r, MG_AG_FLAGS, flags | MG_AF_DONE);  // :584-585
                            }
                        }
                    } else {
                        // drop: `if not fwd_cell and agent.carrying`
                        const uint32_t carry = rec_byte(r, MG_AG_CARRY);
                        if (fbase == 0 && agents_there == 0 && carry) {
                            g[fcell] = (uint8_t)carry;
                            grid_dirty = true;
                            r = rec_set(r, MG_AG_CARRY, 0);
                        }
                    }
                } else if (action == 3) {                            // pickup :590-597
                    if (fbase && (fflags & MG_OF_CAN_PICKUP) && rec_byte(r, MG_AG_CARRY) == 0) {
                        r = rec_set(r, MG_AG_CARRY, fbase);
                        g[fcell] = 0;
                        grid_dirty = true;
                    }
                } else if (action == 5) {                            // toggle :609-613
                    if (fbase) {
                        if (fflags & MG_OF_IS_BOX) {
                            err = err ? err : MG_ERR_TYPE;           // Box.toggle arity objects.py:381-382
                        } else if (fflags & MG_OF_IS_DOOR) {         // Door.toggle objects.py:333-346
                            const MgObjDesc od = cfg.obj[fbase];
                            if (fflags & MG_OF_DOOR_LOCKED) {
                                const uint32_t carry = rec_byte(r, MG_AG_CARRY);
                                if (carry) {
                                    const MgObjDesc cd = cfg.obj[carry];
                                    if ((cd.flags & MG_OF_IS_KEY) && cd.color_idx == od.color_idx)
                                        { g[fcell] = od.unlock_next; grid_dirty = true; }
                                }
                            } else {
                                g[fcell] = od.toggle_next;
                                grid_dirty = true;
                            }
                        }
                    }
                } else if (action == 6) {                            // done :616-617
                } else {
                    err = err ? err : MG_ERR_VALUE;                  // :619-620
                }
            }
            s_rec[k * BS + tid] = r;
            if (cfg.prestige_mask) {
                // agent.reward(rwd) then agent.on_step(): agents.py:141-153 (allow_negative_prestige=False)
                double* pp = st.prestige + (size_t)b * n + k;
                double p = *pp;
                if (rewarded) p = (rwd_applied >= 0) ? p + rwd_applied : 0.0;
                *pp = p * cfg.prestige_beta[k];
            }
        }
        rewards[(size_t)b * n + k] = rew;
    }

    // done agents (base.py:627-646), in index order: without respawn they are deactivated but stay
    // where they are; with respawn they leave their cell (an agent only ever becomes done on a Goal /
    // Lava, i.e. inside that object's stack, so nothing is left behind), drop what they carry
    // (agent.reset(new_episode=False), agents.py:161-166) and are re-placed by rejection sampling
    // among the agents currently on the grid.  Then episode done (base.py:649).
    bool all_done = true;
    for (int k = 0; k < n; k++) {
        uint64_t r = s_rec[k * BS + tid];
        const uint32_t f = rec_byte(r, MG_AG_FLAGS);
        if (f & MG_AF_DONE) {
            if (cfg.respawn) {
                r = rec_set(r, MG_AG_FLAGS, 0);
                r = rec_set(r, MG_AG_CARRY, 0);
                s_rec[k * BS + tid] = r;                      // off the grid while sampling
                place_agent(k);
                r = s_rec[k * BS + tid];
                all_done = false;
            } else {
                r = rec_set(r, MG_AG_FLAGS, f & ~MG_AF_ACTIVE);
            }
            s_rec[k * BS + tid] = r;
        } else all_done = false;
    }
    for (int k = 0; k < n; k++) st.agents[(size_t)b * n + k] = s_rec[k * BS + tid];
    st.step_count[b] = step_count;
    st.mt_pos[b] = mt.pos;
    st.done[b] = (uint8_t)((step_count >= cfg.max_steps) || all_done);
    if (err && st.error[b] == 0) st.error[b] = err;
}

template <int BS>
static hipError_t launch_step_bs(const MgConfig& cfg, const MgState& st, const void* actions, int action_bytes,
                                 float* rewards, hipStream_t s) {
    dim3 grid((cfg.B + BS - 1) / BS), block(BS);
    size_t lds = (size_t)cfg.n_agents * BS * (sizeof(uint64_t) + 3) + MG_MAX_OBJ;
    if (action_bytes == 8)
        hipLaunchKernelGGL((step_kernel<int64_t, BS>), grid, block, lds, s, cfg, st, (const int64_t*)actions, rewards);
    else if (action_bytes == 4)
        hipLaunchKernelGGL((step_kernel<int32_t, BS>), grid, block, lds, s, cfg, st, (const int32_t*)actions, rewards);
    else if (action_bytes == 1)
        hipLaunchKernelGGL((step_kernel<uint8_t, BS>), grid, block, lds, s, cfg, st, (const uint8_t*)actions, rewards);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_step(const MgConfig& cfg, const MgState& st, const void* actions, int action_bytes,
                       float* rewards, hipStream_t s) {
    if (cfg.B <= 0) return hipSuccess;
    // One lane per env: spread the envs over as many CUs as possible with single-wave workgroups
    // until the batch alone fills the chip several times over.
    const int forced = getenv("MG_STEP_BLOCK") ? atoi(getenv("MG_STEP_BLOCK")) : 0;
    const int bs = forced ? forced : (cfg.B > 256 * 8 * 64 ? 256 : 64);
    if (bs == 256) return launch_step_bs<256>(cfg, st, actions, action_bytes, rewards, s);
    return launch_step_bs<64>(cfg, st, actions, action_bytes, rewards, s);
}

}  // namespace mg
