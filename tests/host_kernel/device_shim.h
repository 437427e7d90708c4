// device_shim.h -- TEST INFRASTRUCTURE.  Lets the PRODUCT's own device code (umr_amd/csrc/raster_core.h: k_face_setup,
// eval_pair, clip_depth, ...) compile for the host with clang++, one "thread" at a time, so the shipped source itself -- not a
// restatement of it -- can be fuzzed against the oracle without a GPU (tests/test_kernel_source_on_host.py).  A wave is one
// lane here: wave votes degenerate to the lane's own predicate, cross-lane shuffles to the identity.  Nothing under umr_amd/
// includes this file.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define UMR_HOST_SHIM 1
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct uint2 { unsigned x, y; };
struct dim3 { unsigned x, y, z; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
typedef void *hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline int hipGetLastError() { return 0; }

static thread_local dim3 threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0}, blockDim = {1, 1, 1}, gridDim = {1, 1, 1};
static inline void __syncthreads() {}
static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }
static inline bool __any(bool p) { return p; }
static inline bool __all(bool p) { return p; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
template <class T> static inline T __shfl_xor(T v, int, int = 64) { return v; }
template <class T> static inline T __shfl_up(T v, int, int = 64) { return v; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
#define __expf(x) expf(x)
static inline float __frsqrt_rn(float x) { return 1.0f / sqrtf(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline int __builtin_amdgcn_readlane(int v, int) { return v; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline int __builtin_amdgcn_update_dpp(int old, int, int, int, int, bool) { return old; }   // no other lane: keep `old`
static inline unsigned long long wall_clock64() { return 0; }
template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p += v; return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < *p) *p = v; return o; }
template <class T> static inline T min(T a, T b) { return a < b ? a : b; }
template <class T> static inline T max(T a, T b) { return a > b ? a : b; }
