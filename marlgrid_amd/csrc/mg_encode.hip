// mg_encode.hip — batched MultiGrid.encode (marlgrid/base.py:196-214): per cell the (type_idx, colour_idx, state) triple
// of the cell's *top object* (WorldObj.encode, objects.py:90-99); agents stacked on another object are not reflected, an
// agent that is the cell object encodes as (13, colour, dir).  out: uint8 [B][W][H][3].
//
// Roofline: HBM.  Algorithmic bytes per env = cells_stride + 8 n (in) + 3 W H (out): 939 B at 15x15 with three agents.
//
// The output of a batch is ONE flat byte stream (3 bytes per cell, cells of consecutive envs back to back — an env's
// 675 bytes are not a multiple of anything), so the kernel is written from the output's side: a PIECE is PC consecutive
// cells of the flat [B * W*H] cell space (PC a multiple of 16: a piece's 3 PC bytes are whole aligned 16-byte chunks) and
// belongs to one workgroup of PC / 16 threads:
//   A. the piece's slice of `grid` — one contiguous run of HBM, the 15 padding bytes between envs included — goes to LDS
//      as it is (aligned dwords, every load in flight before the first wait), the (type, colour, state) table of the
//      object kinds (one dword per kind) next to it;
//   B. one lane per (env, agent) of the envs the piece touches: an agent that is the FIRST of its cell (lowest arrival
//      rank: `obj.agents[0]` / the cell object, base.py:547-552) marks the cell — in the grid bytes themselves where object
//      ids and agent codes share a byte (n_obj + 4 n <= 256), in a second byte plane otherwise;
//   C. a lane per 16-byte chunk of output: the six cells the chunk spans (the second half of them possibly in the next
//      env: two aligned LDS windows merged by a 64-bit shift), six table look-ups, the 18 bytes packed with v_perm_b32
//      and cut at the chunk's phase with v_alignbyte, one global_store_dwordx4.
// Nothing per cell but a table look-up; no division per cell (one per chunk); every store 16 bytes and aligned.
// (Round 2's kernel — a lane per cell, a 64-bit divide, a 32-byte descriptor gather and n record loads per empty cell,
// three byte stores — ran at 0.11 of the HBM roofline: profiles/r02/README.md.)
#include "mg_device.h"
#include "mg_encode_core.h"
#include "mg_launch.h"
#if defined(MG_AB_VARIANTS)
#include <stdlib.h>
#endif

namespace mg {

template <int PC>
__global__ __launch_bounds__(PC / 16) void encode_kernel(MgConfig cfg, MgState st, const uint8_t* __restrict__ vis,
                                                         uint8_t* __restrict__ out, EncodeLaunch lc) {
    constexpr int T = PC / 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
#if defined(MG_AB_VARIANTS)
    if (lc.aligned == 2) return;        // measurement build (MG_ENCODE_EMPTY): what a launch of this shape costs when it does nothing
#endif
    const EncodePiece P = encode_piece(cfg, lc, (long long)blockIdx.x, PC);
    encode_stage(cfg, st, lc, P, smem, tid, T);
    __syncthreads();
    encode_agents(cfg, st, lc, P, smem, tid, T);
    __syncthreads();
    // (by runs of 16 cells where nothing stands in the way; the chunk form finishes the batch's last cells — and does everything
    // with a vis_mask, agent marks in their own plane, an `out` that is not 16-byte aligned, grids of fewer than 16 cells)
    int q_first = 0;
    if (lc.runs && !vis) {
        q_first = encode_runs(cfg, lc, P, smem, tid, T);
        if (lc.runs == 2) wave_lds_sync();
        else __syncthreads();
        encode_runs_store(lc, P, out, smem, tid, T);
    }
    encode_chunks(cfg, lc, P, vis, out, smem, tid, T, PC, q_first);
}

hipError_t launch_encode(const MgConfig& cfg, const MgState& st, const uint8_t* vis_mask, uint8_t* out,
                         hipStream_t s) {
    if (cfg.B <= 0) return hipSuccess;
    int PC = 0;
#if defined(MG_AB_VARIANTS)
    if (const char* f = getenv("MG_ENCODE_PC")) PC = atoi(f);      // measurement build: 1024 / 2048 / 4096 / 8192 / 16384
#endif
    EncodeLaunch lc = encode_launch(cfg, out, PC);
#if defined(MG_AB_VARIANTS)
    if (const char* f = getenv("MG_ENCODE_EMPTY")) { if (atoi(f)) lc.aligned = 2; }
#endif
    const size_t lds = kEncTab + (size_t)lc.nraw * (lc.two ? 2 : 1) + (lc.runs ? 3 * (size_t)PC : 0);      // (+ the piece's output image: encode_runs)
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const unsigned pieces = (unsigned)((lc.total + PC - 1) / PC);
#define MG_ENCODE_LAUNCH(N)                                                                                              \
    do {                                                                                                                 \
        if (lds > 64 * 1024) {                                                                                           \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_kernel<N>),                         \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
            if (e != hipSuccess) return e;                                                                               \
        }                                                                                                                \
        hipLaunchKernelGGL((encode_kernel<N>), dim3(pieces), dim3(N / 16), lds, s, cfg, st, vis_mask, out, lc);          \
    } while (0)
    if (PC == 8192) MG_ENCODE_LAUNCH(8192);
    else if (PC == 4096) MG_ENCODE_LAUNCH(4096);
#if defined(MG_AB_VARIANTS)
    else if (PC == 2048) MG_ENCODE_LAUNCH(2048);
    else if (PC == 16384) MG_ENCODE_LAUNCH(16384);
#endif
    else MG_ENCODE_LAUNCH(1024);
#undef MG_ENCODE_LAUNCH
    return hipGetLastError();
}

}  // namespace mg
