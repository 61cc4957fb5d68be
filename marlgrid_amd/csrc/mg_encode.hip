// mg_encode.hip — batched MultiGrid.encode (marlgrid/base.py:196-214): per cell the
// (type_idx, colour_idx, state) triple of the cell's *top object* (WorldObj.encode,
// objects.py:90-99); agents stacked on another object are not reflected, an agent that is the
// cell object encodes as (13, colour, dir).
//
// One lane per cell; a wave covers 64 consecutive cells of the flattened [B][W*H] space, so the
// grid read and the 3-byte-per-cell write are both contiguous across the wave.
#include "mg_device.h"
#include "mg_launch.h"

namespace mg {

__global__ __launch_bounds__(kBlock) void encode_kernel(MgConfig cfg, MgState st, const uint8_t* __restrict__ vis,
                                                        uint8_t* __restrict__ out) {
    const int cells = cfg.W * cfg.H;
    const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (long long)cfg.B * cells) return;
    const int b = (int)(idx / cells), c = (int)(idx - (long long)b * cells);
    uint32_t e0 = 0, e1 = 0, e2 = 0;
    if (!vis || vis[idx]) {
        const uint32_t base = st.grid[(size_t)b * cfg.cells_stride + c];
        if (base) {
            const MgObjDesc od = cfg.obj[base];
            e0 = od.type_idx; e1 = od.color_idx; e2 = od.state;
        } else {
            const uint32_t xy = (uint32_t)(c / cfg.H) | ((uint32_t)(c % cfg.H) << 8);
            uint32_t best = 0xFFFF;
            for (int k = 0; k < cfg.n_agents; k++) {
                const uint64_t r = st.agents[(size_t)b * cfg.n_agents + k];
                if ((rec_byte(r, MG_AG_FLAGS) & MG_AF_PLACED) && rec_xy(r) == xy && rec_byte(r, MG_AG_RANK) < best) {
                    best = rec_byte(r, MG_AG_RANK);
                    e0 = (uint32_t)cfg.agent_type_idx; e1 = cfg.agent_color_idx[k]; e2 = rec_byte(r, MG_AG_DIR);
                }
            }
        }
    }
    uint8_t* o = out + (size_t)idx * 3;
    o[0] = (uint8_t)e0; o[1] = (uint8_t)e1; o[2] = (uint8_t)e2;
}

hipError_t launch_encode(const MgConfig& cfg, const MgState& st, const uint8_t* vis_mask, uint8_t* out,
                         hipStream_t s) {
    if (cfg.B <= 0) return hipSuccess;
    long long total = (long long)cfg.B * cfg.W * cfg.H;
    hipLaunchKernelGGL(encode_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, cfg, st,
                       vis_mask, out);
    return hipGetLastError();
}

}  // namespace mg
