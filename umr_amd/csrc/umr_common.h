// umr_common.h -- shared device helpers for libumr_hip.so (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "../../include/umr_hip.h"

#define UMR_WAVE 64

static inline int umr_launch_status() { return hipGetLastError() == hipSuccess ? UMR_OK : UMR_ERR_LAUNCH; }

// 64-lane butterfly sum; every lane ends with the total (ds_bpermute based, order fixed -> deterministic)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, UMR_WAVE);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, UMR_WAVE));
    return v;
}

// block-wide sum for blocks of up to 1024 threads; result valid in thread 0 (and all of wave 0)
__device__ __forceinline__ float block_sum(float v, float *smem /* >= 16 floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    float r = (threadIdx.x < nw) ? smem[threadIdx.x] : 0.f;
    if (wave == 0) r = wave_sum(r);
    return r;
}
