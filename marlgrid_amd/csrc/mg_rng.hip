// mg_rng.hip — per-env MT19937 seeding (replaces MultiGridEnv.seed -> gym seeding.np_random ->
// numpy RandomState.seed(list) = init_by_array; marlgrid/base.py:371-374).
//
// One lane per env: init_by_array is a 624-long sequential recurrence, so the parallelism is
// across the batch.  Runs once per env lifetime (B * 2.5 KB of HBM written).  Body: mg::mt_seed_env
// (mg_core.h), which also generates the stream's first look-ahead head.
#include "mg_device.h"
#include "mg_launch.h"

namespace mg {

__global__ __launch_bounds__(kBlock) void mt_seed_kernel(int B, const uint32_t* __restrict__ keys,
                                                         const int32_t* __restrict__ key_len,
                                                         uint32_t* __restrict__ mt_all, int32_t* __restrict__ mt_pos,
                                                         uint32_t* __restrict__ mt_head) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    uint32_t key[MG_KEY_WORDS];
#pragma unroll
    for (int i = 0; i < MG_KEY_WORDS; i++) key[i] = keys[(size_t)b * MG_KEY_WORDS + i];
    uint32_t head[MG_MT_HEAD];
    int32_t pos;
    mt_seed_env(key, key_len[b], mt_all + (size_t)b * MG_MT_N, &pos, head);
    mt_pos[b] = pos;
#pragma unroll
    for (int i = 0; i < MG_MT_HEAD; i++) mt_head[(size_t)b * MG_MT_HEAD + i] = head[i];
}

hipError_t launch_mt_seed(int B, const uint32_t* keys, const int32_t* key_len, uint32_t* mt, int32_t* mt_pos,
                          uint32_t* mt_head, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(mt_seed_kernel, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0, s, B, keys, key_len, mt,
                       mt_pos, mt_head);
    return hipGetLastError();
}

}  // namespace mg
