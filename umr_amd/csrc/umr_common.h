// umr_common.h -- shared device helpers for libumr_hip.so (gfx950 / CDNA4 only, wave64).
#pragma once
#ifndef UMR_HOST_SHIM          // tests/host_kernel/device_shim.h compiles this source for the host (test infrastructure)
#include <hip/hip_runtime.h>
#endif
#include <stddef.h>
#include "../../include/umr_hip.h"

#define UMR_WAVE 64

// Kernel launch, `kernel<<<grid, block, lds, stream>>>(args...)`, spelt as a macro: the tests' wave64 emulation build
// (tests/host_kernel/wave_emu.h defines UMR_LAUNCH before this header is read) runs the very same launch sequences of the
// entry points on the CPU.  A template kernel's name goes in parentheses.
#ifndef UMR_LAUNCH
#define UMR_LAUNCH(kernel, grid, block, lds, stream, ...) kernel<<<(grid), (block), (lds), (stream)>>>(__VA_ARGS__)
#endif
// Lanes of ONE wave handing data to each other through LDS without a barrier rely on the wave executing in lockstep (its LDS
// operations complete in order).  The emulation build runs a wave's lanes one after the other between cross-lane operations
// and needs such a point marked; on the device the marker emits no instruction but pins the order for the COMPILER (a
// wavefront-scope release fence + a scheduling barrier: without it nothing would stop a later compiler from moving the loads
// of other lanes' slots above the conditional stores that fill them).
#ifdef UMR_HOST_SHIM
#define UMR_WAVE_LDS_HANDOVER() umr_host_wave_fence()
#else
#define UMR_WAVE_LDS_HANDOVER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

static inline int umr_launch_status() { return hipGetLastError() == hipSuccess ? UMR_OK : UMR_ERR_LAUNCH; }

// 64-lane butterfly sum; every lane ends with the total (ds_bpermute based, order fixed -> deterministic)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, UMR_WAVE);
    return v;
}

// Same total with DPP adds (no LDS crossbar traffic, 7 VALU instead of ~30 VALU + 6 ds_bpermute): butterfly inside each
// 16-lane row, row_bcast:15 / row_bcast:31 across rows, total read from lane 63 into an SGPR.  REQUIRES all 64 lanes
// active at the call (v_readlane ignores exec); summation order differs from wave_sum (both are deterministic).
__device__ __forceinline__ float wave_sum_full(float v) {
#define UMR_DPP_ADD(ctrl, rowmask) \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rowmask, 0xf, false))
    UMR_DPP_ADD(0xB1, 0xf);    // quad_perm [1,0,3,2]
    UMR_DPP_ADD(0x4E, 0xf);    // quad_perm [2,3,0,1]
    UMR_DPP_ADD(0x124, 0xf);   // row_ror:4
    UMR_DPP_ADD(0x128, 0xf);   // row_ror:8   -> every lane holds its row's sum
    UMR_DPP_ADD(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    UMR_DPP_ADD(0x143, 0xc);   // row_bcast:31 into rows 2 and 3 -> lanes 48..63 hold the total
#undef UMR_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Three sums at once, step by step: a DPP read needs two wait states after the VALU write of its source, so ONE chain costs an
// s_nop between its steps; three interleaved chains fill those slots with each other's work (the forward's p2f accumulators:
// 3 x (12 VALU + 7 s_nop) -> 36 VALU + 0).  Same lanes, same order of additions per value as wave_sum_full.
__device__ __forceinline__ void wave_sum_full3(float &a, float &b, float &c) {
#define UMR_DPP_ADD3(ctrl, rowmask)                                                                          \
    { const float ta = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), ctrl, rowmask, 0xf, false)); \
      const float tb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), ctrl, rowmask, 0xf, false)); \
      const float tc = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), ctrl, rowmask, 0xf, false)); \
      a += ta; b += tb; c += tc; }
    UMR_DPP_ADD3(0xB1, 0xf);
    UMR_DPP_ADD3(0x4E, 0xf);
    UMR_DPP_ADD3(0x124, 0xf);
    UMR_DPP_ADD3(0x128, 0xf);
    UMR_DPP_ADD3(0x142, 0xa);
    UMR_DPP_ADD3(0x143, 0xc);
#undef UMR_DPP_ADD3
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
    c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
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

// Zero-fill as a KERNEL, not hipMemsetAsync: memset nodes captured into a HIP graph from these entry points replayed
// correctly once and then not at all on ROCm 7.0/7.2 (second replay of a captured training step: the IoU sums and the
// flatten loss kept accumulating) -- kernel nodes replay reliably, and an eager launch costs the same ~2 us.
static __global__ void umr_k_zero(unsigned *__restrict__ p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
#ifndef UMR_HOST_SHIM
static inline bool umr_zero_async(void *p, size_t bytes, hipStream_t st) {   // bytes: multiple of 4, p 4-byte aligned
    const size_t words = bytes / 4;
    if (!words) return true;
    const unsigned blocks = (unsigned)((words + 255) / 256 < 2048 ? (words + 255) / 256 : 2048);
    umr_k_zero<<<blocks, 256, 0, st>>>((unsigned *)p, words);
    return hipGetLastError() == hipSuccess;
}
#elif defined(UMR_HOST_EMU)   // tests/host_kernel/wave_emu.h: the emulation build zero-fills through the same kernel
static inline bool umr_zero_async(void *p, size_t bytes, hipStream_t st) {
    const size_t words = bytes / 4;
    if (!words) return true;
    const unsigned blocks = (unsigned)((words + 255) / 256 < 2048 ? (words + 255) / 256 : 2048);
    UMR_LAUNCH(umr_k_zero, blocks, 256, 0, st, (unsigned *)p, words);
    return true;
}
#endif

// ---- non-finite tripwire (debug builds only: -DUMR_TRAP=1, tools/nan/bench_trap.py) -----------------------------------
// Kernels report the FIRST non-finite value they read or write without adding a launch or a host synchronisation: one
// 64-bit word per translation unit holds min over reports of (device wall clock << 8 | site id); umr_debug_trap() returns
// the earliest site of all translation units.  Compiled out of the product build (UMR_TRAP undefined).
#ifndef UMR_TRAP
#define UMR_TRAP 0
#endif
#if UMR_TRAP
static __device__ unsigned long long g_umr_trap = ~0ull;
__device__ __forceinline__ bool umr_bad(float x) { return !(fabsf(x) <= 3.0e38f); }   // NaN or inf
__device__ __forceinline__ void umr_trap(bool nonfinite, unsigned site) {
    if (nonfinite) atomicMin(&g_umr_trap, ((unsigned long long)wall_clock64() << 8) | (unsigned long long)(site & 0xffu));
}
#define UMR_TRAP_IF(cond, site) umr_trap((cond), (site))
// ... and WHERE: a second word, min over reports of (device clock << 24 | 24 bits the site packs -- the raster backward:
// launch size class (N > 32) << 23 | mesh of the launch (mod 128) << 16 | face (16 bits: the face-major path takes F <= 65535)), so the offending (view, face) can be replayed on the CPU
static __device__ unsigned long long g_umr_trap_info = ~0ull;
__device__ __forceinline__ void umr_trap_at(bool nonfinite, unsigned site, unsigned info) {
    if (nonfinite) {
        const unsigned long long c = (unsigned long long)wall_clock64();
        atomicMin(&g_umr_trap, (c << 8) | (unsigned long long)(site & 0xffu));
        atomicMin(&g_umr_trap_info, (c << 24) | (unsigned long long)(info & 0xffffffu));
    }
}
#define UMR_TRAP_AT(cond, site, info) umr_trap_at((cond), (site), (info))
#define UMR_TRAP_INFO_ACCESSOR(name)                                                                         \
    extern "C" unsigned long long name(void) {                                                               \
        unsigned long long v = ~0ull;                                                                        \
        (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_umr_trap_info), sizeof(v), 0, hipMemcpyDeviceToHost);     \
        return v;                                                                                            \
    }
#define UMR_TRAP_ACCESSOR(name)                                                                              \
    extern "C" unsigned long long name(int reset) {                                                          \
        unsigned long long v = ~0ull;                                                                        \
        (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_umr_trap), sizeof(v), 0, hipMemcpyDeviceToHost);          \
        if (reset) { const unsigned long long z = ~0ull;                                                     \
                     (void)hipMemcpyToSymbol(HIP_SYMBOL(g_umr_trap), &z, sizeof(z), 0, hipMemcpyHostToDevice); } \
        return v;                                                                                            \
    }
#else
#define UMR_TRAP_IF(cond, site) ((void)0)
#define UMR_TRAP_AT(cond, site, info) ((void)0)
#define UMR_TRAP_INFO_ACCESSOR(name)
#define UMR_TRAP_ACCESSOR(name)
__device__ __forceinline__ bool umr_bad(float) { return false; }
#endif
// site ids: 1 raster face setup: non-finite vertex in | 2 raster forward: non-finite pixel out | 3 raster backward: non-finite
// incoming gradient | 4 raster backward: non-finite vertex gradient out, 5: texel gradient out (+ 0x40 silhouette variant, + 0x80 hard) | 10 projection: vertex / camera in | 11 projection backward:
// incoming gradient | 12 projection backward: gradient out | 20 IoU loss out | 21 IoU backward incoming | 22 grid-sample out |
// 23 grid-sample backward incoming | 24 grid-sample backward out | 25 Laplacian in | 26 flatten in | 27 up-sampling in |
// 28 up-sampling backward incoming | 30 PNet head: feature in | 31 PNet head: value out | 32 PNet backward: gradient out |
// 33 perceptual prologue in | 34 perceptual prologue backward incoming | 40 distance transform out
