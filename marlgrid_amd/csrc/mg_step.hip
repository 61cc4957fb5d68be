// mg_step.hip — batched MultiGridEnv.step action loop (marlgrid/base.py:501-649), with the reset of
// finished episodes (base.py:402-416) fused into its tail when the caller passes a reset program.
//
// Per env the reference is strictly sequential, so the only parallel axis is the env batch: one
// LANE per env (a wave-per-env mapping would idle 63 of 64 lanes through ~100 scalar instructions).
// The per-env body is mg::step_load / mg::step_run (mg_core.h).  What this file adds is the CDNA4
// shape of it:
//   * every per-env array the body indexes dynamically — agent records, the shuffled order, the
//     look-ahead RNG words — lives in LDS as [item][lane] columns (conflict-free), and the object
//     table is staged once per workgroup, so after the first memory round trip (all of it contiguous
//     across the batch: records, actions, RNG head, counters) the action loop only touches HBM for
//     the <= 2 grid cells an agent looks at;
//   * the RNG state proper (2.5 KB per env, three scattered words per draw) is only touched at the
//     very end, when the look-ahead head is topped up with all loads in flight at once;
//   * a finished env resets in the same lane right away (the done flag is computed by the lane that
//     would run the reset): no second launch, no second pass over the records.
#include "mg_device.h"
#include "mg_launch.h"

#if defined(MG_AB_VARIANTS)
#include <stdlib.h>
#endif

namespace mg {

template <int BS>
__global__ __launch_bounds__(BS) void step_kernel(MgConfig cfg, MgState st, MgGenProgram prog, int has_prog,
                                                  const void* __restrict__ actions, int action_bytes,
                                                  float* __restrict__ rewards) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_mem[];
    const int n = cfg.n_agents;
    StepScratch sc;
    sc.rec = s_mem;                                                          // [n][BS] u64
    MgObjDesc* s_obj = reinterpret_cast<MgObjDesc*>(s_mem + (size_t)n * BS);   // [MG_MAX_OBJ] 32 B each
    sc.head = reinterpret_cast<uint32_t*>(s_obj + MG_MAX_OBJ);               // [MG_MT_HEAD][BS] u32
    sc.act = reinterpret_cast<uint8_t*>(sc.head + MG_MT_HEAD * BS);          // [n][BS]
    sc.fb = sc.act + (size_t)n * BS;                                         // [n][BS]
    sc.ord = sc.fb + (size_t)n * BS;                                         // [n][BS] iter_order (more than 16 agents)
    uint8_t* s_oflags = sc.ord + (size_t)n * BS;                             // [MG_MAX_OBJ]
    sc.obj = s_obj;
    sc.oflags = s_oflags;
    sc.S = BS;
    const int tid = threadIdx.x;
    sc.col = tid;
    const int b = blockIdx.x * BS + tid;
    const bool live = b < cfg.B;

    // object table -> LDS (16-byte pieces; id 0 = None reads as all-zero flags)
    {
        const uint4* src = reinterpret_cast<const uint4*>(cfg.obj);
        uint4* dst = reinterpret_cast<uint4*>(s_obj);
        for (int i = tid; i < cfg.n_obj * 2; i += BS) dst[i] = src[i];
        for (int i = tid; i < MG_MAX_OBJ; i += BS) s_oflags[i] = (i > 0 && i < cfg.n_obj) ? cfg.obj[i].flags : 0;
    }
    StepEnv env{0, 0};
    if (live) env = step_load(cfg, st, actions, action_bytes, b, sc);
    __syncthreads();
    if (!live) return;
    step_run(cfg, st, prog, has_prog != 0, rewards, b, env, sc, st.grid + (size_t)b * cfg.cells_stride);
}

template <int BS>
static hipError_t launch_step_bs(const MgConfig& cfg, const MgState& st, const void* actions, int action_bytes,
                                 float* rewards, const MgGenProgram* prog, hipStream_t s) {
    dim3 grid((cfg.B + BS - 1) / BS), block(BS);
    const size_t lds = (size_t)cfg.n_agents * BS * (sizeof(uint64_t) + 4) + MG_MAX_OBJ * sizeof(MgObjDesc) +
                       (size_t)MG_MT_HEAD * BS * sizeof(uint32_t) + MG_MAX_OBJ;
    MgGenProgram none;
    none.template_grid = nullptr;
    none.n_ops = 0;
    none.ops = nullptr;
    none.reject = nullptr;
    none.n_reject = 0;
    const MgGenProgram& p = prog ? *prog : none;
    const int has = prog ? 1 : 0;
    if (action_bytes != 1 && action_bytes != 4 && action_bytes != 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL((step_kernel<BS>), grid, block, lds, s, cfg, st, p, has, actions, action_bytes, rewards);
    return hipGetLastError();
}

hipError_t launch_step(const MgConfig& cfg, const MgState& st, const void* actions, int action_bytes,
                       float* rewards, const MgGenProgram* prog, hipStream_t s) {
    if (cfg.B <= 0) return hipSuccess;
    // One lane per env: spread the envs over as many CUs as possible with single-wave workgroups
    // until the batch alone fills the chip several times over.
    int bs = cfg.B > 256 * 8 * 64 ? 256 : 64;
    // (256-lane workgroups of many agents would need more than the 64 KiB of LDS a launch gets without asking)
    if ((size_t)cfg.n_agents * 256 * (sizeof(uint64_t) + 4) + MG_MAX_OBJ * (sizeof(MgObjDesc) + 1) + (size_t)MG_MT_HEAD * 256 * 4 > 64 * 1024) bs = 64;
#if defined(MG_AB_VARIANTS)
    if (const char* f = getenv("MG_STEP_BLOCK")) { const int v = atoi(f); if (v == 64 || v == 256) bs = v; }
#endif
    if (bs == 256) return launch_step_bs<256>(cfg, st, actions, action_bytes, rewards, prog, s);
    return launch_step_bs<64>(cfg, st, actions, action_bytes, rewards, prog, s);
}

}  // namespace mg
