// mg_rng.hip — per-env MT19937 seeding (replaces MultiGridEnv.seed -> gym seeding.np_random ->
// numpy RandomState.seed(list) = init_by_array; marlgrid/base.py:371-374).
//
// One lane per env: init_by_array is a 624-long sequential recurrence, so the parallelism is
// across the batch.  Runs once per env lifetime (B * 2.5 KB of HBM written).
#include "mg_device.h"
#include "mg_launch.h"

namespace mg {

__global__ __launch_bounds__(kBlock) void mt_seed_kernel(int B, const uint32_t* __restrict__ keys,
                                                         const int32_t* __restrict__ key_len,
                                                         uint32_t* __restrict__ mt_all, int32_t* __restrict__ mt_pos) {
    int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    uint32_t* mt = mt_all + (size_t)b * MG_MT_N;
    uint32_t key[MG_KEY_WORDS];
#pragma unroll
    for (int i = 0; i < MG_KEY_WORDS; i++) key[i] = keys[(size_t)b * MG_KEY_WORDS + i];
    int klen = key_len[b];
    if (klen < 1) klen = 1;
    if (klen > MG_KEY_WORDS) klen = MG_KEY_WORDS;

    // init_genrand(19650218)
    uint32_t prev = 19650218u;
    mt[0] = prev;
    for (int i = 1; i < MG_MT_N; i++) {
        prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i;
        mt[i] = prev;
    }
    // init_by_array
    int i = 1, j = 0;
    prev = mt[0];
    for (int k = MG_MT_N; k; k--) {   // max(N, key_len) == N for key_len <= 2
        uint32_t kj = (j == 0) ? key[0] : key[1];
        uint32_t v = (mt[i] ^ ((prev ^ (prev >> 30)) * 1664525u)) + kj + (uint32_t)j;
        mt[i] = v;
        prev = v;
        i++; j++;
        if (i >= MG_MT_N) { mt[0] = prev; i = 1; }
        if (j >= klen) j = 0;
    }
    for (int k = MG_MT_N - 1; k; k--) {
        uint32_t v = (mt[i] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
        mt[i] = v;
        prev = v;
        i++;
        if (i >= MG_MT_N) { mt[0] = prev; i = 1; }
    }
    mt[0] = 0x80000000u;
    mt_pos[b] = 0;   // lazy form of numpy's pos == 624
}

hipError_t launch_mt_seed(int B, const uint32_t* keys, const int32_t* key_len, uint32_t* mt, int32_t* mt_pos,
                          hipStream_t s) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(mt_seed_kernel, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0, s, B, keys, key_len, mt,
                       mt_pos);
    return hipGetLastError();
}

}  // namespace mg
