// mg_reset.hip — batched MultiGridEnv.reset (marlgrid/base.py:402-416): `_gen_grid` as a static
// template + ordered rejection-sampled placements (base.py:664-708; envs/cluttered.py:25-36,
// envs/empty.py:9-16, envs/goalcycle.py:30-51), then agent placement in index order.
//
// One lane per env (64 envs per wave): the work per env is a short, data-dependent rejection
// loop over that env's own RNG, so the batch is the only parallel axis.  The template copy is
// 16-byte vector traffic; placements are byte stores into the env's own grid slice.  The per-env
// bodies are mg::reset_run / mg::place_run (mg_core.h); the auto-reset of finished episodes does not
// come through here — it runs in the tail of the step kernel (mg_step.hip).
#include "mg_device.h"
#include "mg_launch.h"

namespace mg {

// object flags -> LDS once per workgroup (the bodies test can_overlap per rejection-sampling draw)
__device__ __forceinline__ void stage_oflags(const MgConfig& cfg, uint8_t* s_oflags) {
    for (int i = threadIdx.x; i < MG_MAX_OBJ; i += kBlock) s_oflags[i] = (i > 0 && i < cfg.n_obj) ? cfg.obj[i].flags : 0;
    __syncthreads();
}

__global__ __launch_bounds__(kBlock) void reset_kernel(MgConfig cfg, MgState st, MgGenProgram prog,
                                                       const uint8_t* __restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_rec[];  // [n][kBlock]
    uint8_t* s_oflags = reinterpret_cast<uint8_t*>(s_rec + (size_t)cfg.n_agents * kBlock);
    stage_oflags(cfg, s_oflags);
    const int tid = threadIdx.x;
    const int b = blockIdx.x * kBlock + tid;
    if (b >= cfg.B) return;
    if (mask && !mask[b]) return;
    // a caller that passes the done flags themselves as the mask keeps them readable as step()'s
    // return value (the next mg_step overwrites them); any other reset clears them
    reset_run(cfg, st, prog, s_oflags, b, mask != st.done, s_rec, kBlock, tid);
}

__global__ __launch_bounds__(kBlock) void put_obj_kernel(MgConfig cfg, MgState st, int obj, int x, int y,
                                                         const uint8_t* __restrict__ mask) {
    int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= cfg.B) return;
    if (mask && !mask[b]) return;
    st.grid[(size_t)b * cfg.cells_stride + x * cfg.H + y] = (uint8_t)obj;
    // `grid.set(i, j, obj)` replaces the cell's object: an agent that WAS the cell's object, or stood in the replaced
    // object's `.agents`, is in no cell any more (base.py:655-662)
    for (int k = 0; k < cfg.n_agents; k++) {
        const uint64_t r = st.agents[(size_t)b * cfg.n_agents + k];
        const uint32_t f = rec_byte(r, MG_AG_FLAGS);
        if ((f & MG_AF_PLACED) && (int)rec_byte(r, MG_AG_X) == x && (int)rec_byte(r, MG_AG_Y) == y)
            st.agents[(size_t)b * cfg.n_agents + k] = rec_set(r, MG_AG_FLAGS, (f & ~MG_AF_PLACED) | MG_AF_EVICTED);
    }
}

// MultiGridEnv.place_obj / try_place_obj on a live grid (base.py:664-708): lane per env.
__global__ __launch_bounds__(kBlock) void place_kernel(MgConfig cfg, MgState st, int what, int x0, int y0, int x1,
                                                       int y1, int max_tries, const int32_t* __restrict__ fixed_pos,
                                                       const uint8_t* __restrict__ mask, const uint8_t* __restrict__ reject,
                                                       int32_t* __restrict__ out_pos, uint8_t* __restrict__ out_ok) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_rec[];  // [n][kBlock]
    uint8_t* s_oflags = reinterpret_cast<uint8_t*>(s_rec + (size_t)cfg.n_agents * kBlock);
    stage_oflags(cfg, s_oflags);
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= cfg.B) return;
    if (mask && !mask[b]) return;
    place_run(cfg, st, s_oflags, b, what, x0, y0, x1, y1, max_tries, fixed_pos, reject, out_pos, out_ok, s_rec, kBlock,
              (int)threadIdx.x);
}

hipError_t launch_place(const MgConfig& cfg, const MgState& st, int what, int x0, int y0, int x1, int y1, int max_tries,
                        const int32_t* fixed_pos, const uint8_t* mask, const uint8_t* reject, int32_t* out_pos,
                        uint8_t* out_ok, hipStream_t s) {
    if (cfg.B <= 0) return hipSuccess;
    const size_t lds = (size_t)cfg.n_agents * kBlock * sizeof(uint64_t) + MG_MAX_OBJ;
    hipLaunchKernelGGL(place_kernel, dim3((cfg.B + kBlock - 1) / kBlock), dim3(kBlock), lds, s, cfg, st, what, x0, y0,
                       x1, y1, max_tries, fixed_pos, mask, reject, out_pos, out_ok);
    return hipGetLastError();
}

hipError_t launch_reset(const MgConfig& cfg, const MgState& st, const MgGenProgram& prog, const uint8_t* mask,
                        hipStream_t s) {
    if (cfg.B <= 0) return hipSuccess;
    size_t lds = (size_t)cfg.n_agents * kBlock * sizeof(uint64_t) + MG_MAX_OBJ;
    hipLaunchKernelGGL(reset_kernel, dim3((cfg.B + kBlock - 1) / kBlock), dim3(kBlock), lds, s, cfg, st, prog, mask);
    return hipGetLastError();
}

hipError_t launch_put_obj(const MgConfig& cfg, const MgState& st, int obj, int x, int y, const uint8_t* mask,
                          hipStream_t s) {
    if (cfg.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(put_obj_kernel, dim3((cfg.B + kBlock - 1) / kBlock), dim3(kBlock), 0, s, cfg, st, obj, x, y,
                       mask);
    return hipGetLastError();
}

}  // namespace mg
