// wave_emu.h -- TEST INFRASTRUCTURE.  A wave64 SIMT emulator for the host: lets the product's raster translation unit
// (umr_amd/csrc/raster.hip with all its kernels, unmodified apart from the UMR_HOST_SHIM hooks the source itself carries) compile
// with clang++ for x86-64 and RUN on the CPU, workgroup by workgroup, lane by lane, so that whole kernels -- binning, tile
// filters, visit loops, culling passes, reductions -- and the C-ABI entry points above them can be checked against the goldens
// and the oracle, and fuzzed on degenerate scenes, without a GPU (tests/test_raster_library_on_host.py).  Nothing under umr_amd/
// includes this file; the product has no CPU path.
//
// Execution model.  Every lane of a workgroup is a fibre (its own stack, a 7-register context switch).  A lane runs until it
// reaches a cross-lane operation, deposits its operand and parks; when no lane of the workgroup can run any more the scheduler
// completes the operations that are ready and releases their lanes:
//   * a workgroup barrier (__syncthreads) when every lane that has not returned is parked at it;
//   * a UNIFORM wave operation (__ballot, __any, v_readlane, v_readfirstlane, __shfl_*, the DPP steps of wave_sum_full) when every
//     lane of the wave that has not returned is parked at the same operation -- these sit in wave-uniform control flow in the
//     kernels; anything else is reported as a deadlock with the lanes' positions;
//   * a SUBSET wave operation -- one the kernels execute under lane divergence, where the hardware's EXEC mask makes exactly
//     the lanes that reach it take part: __all (the dead-sub-tile vote of the face-major backward's visit) and the DPP steps of
//     texel_accumulate (emu_dpp_subset) -- with the lanes parked at it, and before any uniform operation other lanes of the wave
//     wait at: the divergent lanes have to catch up to the reconvergence point first, which is the order the hardware runs
//     them in.  Priority: DPP subset > __all > uniform (later in the visit body first).
// Inactive lanes read by a DPP operand yield `old` (bound_ctrl = 0), as on the hardware.  LDS is `static thread_local` storage
// (garbage at workgroup start, like the real thing: the dynamic part is filled with NaNs), atomics are plain read-modify-writes
// (one OS thread runs a launch).
#pragma once
#if !defined(__x86_64__)
#error "wave_emu.h: the fibre switch is written for x86-64"
#endif
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <functional>
#include <vector>

#define UMR_HOST_SHIM 1
#define UMR_HOST_EMU 1
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define FWD_WPE_ATTR   // occupancy attributes of the kernels (amdgpu_waves_per_eu): nothing to say on the host
#define BWD_WPE_ATTR

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct uint2 { unsigned x, y; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
typedef void *hipStream_t;
typedef void *hipEvent_t;
typedef int hipError_t;
#define hipSuccess 0
static inline int hipGetLastError() { return 0; }
static inline int hipDeviceSynchronize() { return 0; }
static inline int hipEventCreate(hipEvent_t *e) { *e = nullptr; return 0; }
static inline int hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline int hipEventSynchronize(hipEvent_t) { return 0; }
static inline int hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
static inline int hipEventDestroy(hipEvent_t) { return 0; }

static thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(".text\n"
    ".weak emu_switch\n"            // one definition per translation unit of libumr_host.so; the linker keeps one
    ".type emu_switch,@function\n"
    "emu_switch:\n"
    "    pushq %rbp\n    pushq %rbx\n    pushq %r12\n    pushq %r13\n    pushq %r14\n    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    popq %r15\n    popq %r14\n    popq %r13\n    popq %r12\n    popq %rbx\n    popq %rbp\n"
    "    ret\n"
    ".size emu_switch,.-emu_switch\n");

namespace emu {

enum Kind { K_NONE = 0, K_BARRIER, K_BALLOT, K_ANY, K_ALL, K_SHFL, K_READLANE, K_READFIRST, K_DPP, K_DPP_SUBSET, K_FENCE };
enum State { S_RUN = 0, S_WAVE, S_BLOCK, S_DONE };
constexpr int MAX_THREADS = 1024;
constexpr size_t STACK_BYTES = 256 * 1024;

struct Lane {
    void *sp;
    int state, kind, prio;
};

struct Block {
    int nthreads = 0, cur = -1;
    Lane lane[MAX_THREADS];
    uint32_t val[MAX_THREADS];            // operand a parked lane deposited
    uint32_t snap[MAX_THREADS];           // operands of the released group (indexed by thread id)
    uint64_t active[MAX_THREADS / 64];    // lanes of the released group, per wave
    void *sched_sp = nullptr;
    char *stacks = nullptr;
    const std::function<void()> *body = nullptr;
    std::vector<float> dyn_lds;
    long switches = 0, collectives = 0;
};

static thread_local Block *g_block = nullptr;

static inline void die(const char *msg) {
    Block *b = g_block;
    fprintf(stderr, "wave_emu: %s (block %u,%u)\n", msg, blockIdx.x, blockIdx.y);
    if (b)
        for (int t = 0; t < b->nthreads; ++t)
            if (b->lane[t].state != S_DONE && (t % 64 == 0 || b->lane[t].kind != b->lane[t - 1].kind || b->lane[t].state != b->lane[t - 1].state))
                fprintf(stderr, "  thread %d: state %d kind %d prio %d\n", t, b->lane[t].state, b->lane[t].kind, b->lane[t].prio);
    abort();
}

static void lane_entry() {
    Block *b = g_block;
    (*b->body)();
    b->lane[b->cur].state = S_DONE;
    void *dummy;
    emu_switch(&dummy, b->sched_sp);
    abort();   // a finished lane is never resumed
}

// park the running lane at a cross-lane operation (operand deposited); returns once the scheduler has released it
static inline void park(int state, int kind, int prio, uint32_t operand) {
    Block *b = g_block;
    const int t = b->cur;
    b->val[t] = operand;
    b->lane[t].state = state; b->lane[t].kind = kind; b->lane[t].prio = prio;
    ++b->switches;
    emu_switch(&b->lane[t].sp, b->sched_sp);
}

static inline bool resolve(Block *b) {
    bool released = false;
    const int nw = (b->nthreads + 63) / 64;
    int live = 0, at_barrier = 0;
    for (int t = 0; t < b->nthreads; ++t) {
        live += b->lane[t].state != S_DONE;
        at_barrier += b->lane[t].state == S_BLOCK;
    }
    if (live && at_barrier == live) {
        for (int t = 0; t < b->nthreads; ++t)
            if (b->lane[t].state == S_BLOCK) b->lane[t].state = S_RUN;
        return true;
    }
    for (int w = 0; w < nw; ++w) {
        const int t0 = w * 64, t1 = t0 + 64 < b->nthreads ? t0 + 64 : b->nthreads;
        int wl = 0, ww = 0, maxp = -1;
        for (int t = t0; t < t1; ++t) {
            wl += b->lane[t].state != S_DONE;
            if (b->lane[t].state == S_WAVE) { ++ww; if (b->lane[t].prio > maxp) maxp = b->lane[t].prio; }
        }
        if (!ww) continue;
        if (maxp == 0 && ww != wl) continue;     // a uniform operation waits for every lane of the wave that has not returned
        int kind = -1;
        uint64_t act = 0;
        for (int t = t0; t < t1; ++t)
            if (b->lane[t].state == S_WAVE && b->lane[t].prio == maxp) {
                if (kind < 0) kind = b->lane[t].kind;
                else if (kind != b->lane[t].kind) die("lanes of one wave parked at different operations of the same class");
                act |= 1ull << (t - t0);
            }
        for (int t = t0; t < t1; ++t) b->snap[t] = b->val[t];
        b->active[w] = act;
        for (int t = t0; t < t1; ++t)
            if ((act >> (t - t0)) & 1) b->lane[t].state = S_RUN;
        ++b->collectives;
        released = true;
    }
    return released;
}

static inline void run_block(Block *b) {
    g_block = b;
    for (int t = 0; t < b->nthreads; ++t) {
        char *top = b->stacks + (size_t)(t + 1) * STACK_BYTES;
        void **sp = (void **)top;
        *--sp = nullptr;                    // return address slot of lane_entry's frame (never used)
        *--sp = (void *)&lane_entry;        // popped by emu_switch's `ret`
        for (int k = 0; k < 6; ++k) *--sp = nullptr;
        b->lane[t].sp = sp;
        b->lane[t].state = S_RUN; b->lane[t].kind = K_NONE; b->lane[t].prio = 0;
    }
    for (;;) {
        int done = 0;
        for (int t = 0; t < b->nthreads; ++t) {
            if (b->lane[t].state == S_RUN) {
                b->cur = t;
                threadIdx = dim3(t, 0, 0);
                emu_switch(&b->sched_sp, b->lane[t].sp);
            }
            done += b->lane[t].state == S_DONE;
        }
        if (done == b->nthreads) break;
        if (!resolve(b)) die("deadlock: no cross-lane operation can complete");
    }
    g_block = nullptr;
}

struct Stats { long blocks = 0, switches = 0, collectives = 0; };
static Stats g_stats;

static inline Block *the_block() {   // one per OS thread, fibre stacks mapped lazily
    static thread_local Block *b = nullptr;
    if (!b) {
        b = new Block();
        b->stacks = (char *)mmap(nullptr, (size_t)MAX_THREADS * STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (b->stacks == (char *)MAP_FAILED) { perror("mmap"); abort(); }
    }
    return b;
}

template <class F>
static inline void launch(dim3 grid, dim3 block, size_t lds_bytes, F &&fn) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > MAX_THREADS || block.y != 1 || block.z != 1) { fprintf(stderr, "wave_emu: unsupported block shape\n"); abort(); }
    Block *b = the_block();
    const std::function<void()> body(fn);
    b->body = &body;
    b->nthreads = nthreads;
    b->dyn_lds.assign(lds_bytes / 4 + 4, NAN);
    blockDim = block; gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = dim3(bx, by, bz);
                if (lds_bytes) std::fill(b->dyn_lds.begin(), b->dyn_lds.end(), NAN);
                run_block(b);
                ++g_stats.blocks;
            }
    g_stats.switches += b->switches; g_stats.collectives += b->collectives;
    b->switches = b->collectives = 0;
}

struct Snap { const uint32_t *v; uint64_t active; int lane; };   // v[l] = operand of lane l of this wave
static inline Snap collect(int kind, int prio, uint32_t operand) {
    park(S_WAVE, kind, prio, operand);
    Block *b = g_block;
    const int t = b->cur, w = t >> 6;
    return Snap{b->snap + w * 64, b->active[w], t & 63};
}

}  // namespace emu

static inline void *umr_host_dynamic_lds() { return emu::g_block->dyn_lds.data(); }
static inline int umr_host_lane() { return emu::g_block->cur & 63; }

#define UMR_LAUNCH(kernel, grid, block, lds, stream, ...) emu::launch(dim3(grid), dim3(block), (size_t)(lds), [=]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { emu::park(emu::S_BLOCK, emu::K_BARRIER, 0, 0); }
static inline void umr_host_wave_fence() { (void)emu::collect(emu::K_FENCE, 0, 0); }   // UMR_WAVE_LDS_HANDOVER (umr_common.h)
static inline unsigned long long __ballot(bool p) {
    const emu::Snap s = emu::collect(emu::K_BALLOT, 0, p ? 1u : 0u);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (((s.active >> l) & 1) && s.v[l]) m |= 1ull << l;
    return m;
}
static inline bool __any(bool p) {
    const emu::Snap s = emu::collect(emu::K_ANY, 0, p ? 1u : 0u);
    for (int l = 0; l < 64; ++l) if (((s.active >> l) & 1) && s.v[l]) return true;
    return false;
}
static inline bool __all(bool p) {   // executed under divergence in the kernels: the lanes that reach it vote
    const emu::Snap s = emu::collect(emu::K_ALL, 1, p ? 1u : 0u);
    for (int l = 0; l < 64; ++l) if (((s.active >> l) & 1) && !s.v[l]) return false;
    return true;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
template <class T> static inline uint32_t emu_bits(T v) { static_assert(sizeof(T) == 4, "32-bit operands only"); uint32_t u; memcpy(&u, &v, 4); return u; }
template <class T> static inline T emu_from_bits(uint32_t u) { T v; memcpy(&v, &u, 4); return v; }
template <class T> static inline T __shfl_xor(T v, int o, int = 64) {
    const emu::Snap s = emu::collect(emu::K_SHFL, 0, emu_bits(v));
    const int src = s.lane ^ o;
    return ((s.active >> src) & 1) ? emu_from_bits<T>(s.v[src]) : v;
}
template <class T> static inline T __shfl_up(T v, int o, int = 64) {
    const emu::Snap s = emu::collect(emu::K_SHFL, 0, emu_bits(v));
    const int src = s.lane - o;
    return (src >= 0 && ((s.active >> src) & 1)) ? emu_from_bits<T>(s.v[src]) : v;
}
static inline int __builtin_amdgcn_readlane(int v, int l) {
    const emu::Snap s = emu::collect(emu::K_READLANE, 0, (uint32_t)v);
    return (int)s.v[l & 63];       // v_readlane ignores EXEC: the register of lane l as it is
}
static inline int __builtin_amdgcn_readfirstlane(int v) {
    const emu::Snap s = emu::collect(emu::K_READFIRST, 0, (uint32_t)v);
    return (int)s.v[__builtin_ctzll(s.active)];
}
// DPP source lane of `lane` for the controls the kernels use; -1 = no valid source (the destination keeps `old`)
static inline int emu_dpp_src(int lane, int ctrl) {
    const int row = lane & ~15, r = lane & 15;
    if (ctrl >= 0x000 && ctrl <= 0x0FF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);   // quad_perm
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const int s = r + (ctrl & 15); return s < 16 ? row + s : -1; }   // row_shl:n
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int s = r - (ctrl & 15); return s >= 0 ? row + s : -1; }   // row_shr:n
    if (ctrl >= 0x121 && ctrl <= 0x12F) return row + ((r - (ctrl & 15)) & 15);                            // row_ror:n
    if (ctrl == 0x142) return row >= 16 ? row - 1 : -1;                                                    // row_bcast:15
    if (ctrl == 0x143) return lane >= 32 ? 31 : -1;                                                        // row_bcast:31
    fprintf(stderr, "wave_emu: DPP control 0x%x not modelled\n", ctrl);
    abort();
}
static inline int emu_dpp(int kind, int prio, int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const emu::Snap s = emu::collect(kind, prio, (uint32_t)src);
    const int lane = s.lane;
    if (!((row_mask >> (lane >> 4)) & 1) || !((bank_mask >> ((lane & 15) >> 2)) & 1)) return old;
    const int sl = emu_dpp_src(lane, ctrl);
    if (sl < 0 || !((s.active >> sl) & 1)) return bound_ctrl ? 0 : old;
    return (int)s.v[sl];
}
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    return emu_dpp(emu::K_DPP, 0, old, src, ctrl, row_mask, bank_mask, bound_ctrl);
}
// the DPP steps of texel_accumulate (raster_backward.h), executed by the lanes that contribute to the face in this visit
static inline int emu_dpp_subset(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    return emu_dpp(emu::K_DPP_SUBSET, 2, old, src, ctrl, row_mask, bank_mask, bound_ctrl);
}
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned base) {
    const int lane = umr_host_lane();
    return base + (unsigned)__builtin_popcount(lane >= 32 ? mask : (mask & ((1u << lane) - 1u)));
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned base) {
    const int lane = umr_host_lane();
    return base + (lane > 32 ? (unsigned)__builtin_popcount(mask & ((1u << (lane - 32)) - 1u)) : 0u);
}
#define __expf(x) expf(x)
static inline float __frsqrt_rn(float x) { return 1.0f / sqrtf(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline unsigned long long wall_clock64() { return 0; }
template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p += v; return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < *p) *p = v; return o; }
template <class T> static inline T min(T a, T b) { return a < b ? a : b; }
template <class T> static inline T max(T a, T b) { return a > b ? a : b; }
