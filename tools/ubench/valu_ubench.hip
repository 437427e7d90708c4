// valu_ubench.hip -- issue-rate microbenchmark for the instruction mix of the raster kernels (gfx950).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_ubench.hip -o /tmp/valu_ubench && /tmp/valu_ubench
// For each instruction class: cycles per wave-instruction per SIMD at 1/2/4/8 resident waves per SIMD (s_memtime
// ticks and wall clock), so kernel edits are priced with measured issue costs instead of datasheet rates.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

#define REP8(X) X X X X X X X X
constexpr int ITERS = 512;   // loop trips; each trip issues 64 measured instructions per wave

// Eight independent chains (a0..a7) so dependent-issue latency never limits; BODY uses %0..%7 accumulators,
// %8,%9 vector sources, %10 scalar source.
#define KERNEL(NAME, BODY)                                                                                   \
    __global__ void NAME(float *out, unsigned long long *ticks, float s0) {                                   \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
              a7 = a0 + 7;                                                                                     \
        float b = 1.0001f + 1e-7f * threadIdx.x, c = 0.5f;                                                    \
        unsigned long long t0 = __builtin_readcyclecounter();                                                  \
        for (int i = 0; i < ITERS; ++i) {                                                                      \
            asm volatile(REP8(BODY)                                                                            \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                         : "v"(b), "v"(c), "s"(s0) : "vcc", "scc", "s20", "s21", "s22", "s23");                      \
        }                                                                                                      \
        unsigned long long t1 = __builtin_readcyclecounter();                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                   \
        if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;           \
    }

#define I8(OP) OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7)

#define OP_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define OP_FMAC(r) "v_fmac_f32 " #r ", %8, %9\n"
#define OP_MUL(r) "v_mul_f32 " #r ", " #r ", %8\n"
#define OP_ADD(r) "v_add_f32 " #r ", " #r ", %8\n"
#define OP_MULS(r) "v_mul_f32 " #r ", %10, " #r "\n"
#define OP_FMAS(r) "v_fma_f32 " #r ", " #r ", %10, %9\n"
#define OP_MAX(r) "v_max_f32 " #r ", " #r ", %8\n"
#define OP_MED3(r) "v_med3_f32 " #r ", " #r ", %8, %9\n"
#define OP_CMP(r) "v_cmp_gt_f32 vcc, " #r ", %8\n"
#define OP_CMPS(r) "v_cmp_gt_f32 s[20:21], " #r ", %8\n"
#define OP_CND(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc\n"
#define OP_CNDS(r) "v_cndmask_b32 " #r ", " #r ", %8, s[22:23]\n"
#define OP_RCP(r) "v_rcp_f32 " #r ", " #r "\n"
#define OP_EXP(r) "v_exp_f32 " #r ", " #r "\n"
#define OP_RSQ(r) "v_rsq_f32 " #r ", " #r "\n"
#define OP_SQRT(r) "v_sqrt_f32 " #r ", " #r "\n"
#define OP_ANDB(r) "v_and_b32 " #r ", " #r ", %8\n"
#define OP_LSHL(r) "v_lshlrev_b32 " #r ", 1, " #r "\n"
#define OP_ADDU(r) "v_add_u32 " #r ", " #r ", %8\n"
#define OP_CVTFI(r) "v_cvt_i32_f32 " #r ", " #r "\n"
#define OP_CVTIF(r) "v_cvt_f32_i32 " #r ", " #r "\n"
#define OP_BFE(r) "v_bfe_u32 " #r ", " #r ", 2, 5\n"
#define OP_MOVDPP(r) "v_mov_b32_dpp " #r ", " #r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_ADDDPP(r) "v_add_f32_dpp " #r ", " #r ", " #r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_ADDDPPR(r) "v_add_f32_dpp " #r ", " #r ", " #r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_READLANE(r) "v_readlane_b32 s20, " #r ", 3\n"
#define OP_READFL(r) "v_readfirstlane_b32 s20, " #r "\n"
#define OP_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", 3, " #r "\n"
#define OP_SUBREV(r) "v_subrev_f32 " #r ", %8, " #r "\n"
#define OP_MOV(r) "v_mov_b32 " #r ", %8\n"
// VALU + one SALU per VALU (does the scalar unit steal VALU issue slots of the same wave / of other waves?)
#define OP_FMA_SALU(r) "v_fma_f32 " #r ", " #r ", %8, %9\n s_and_b32 s20, s20, s21\n"
#define OP_FMA_SALU2(r) "v_fma_f32 " #r ", " #r ", %8, %9\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_add_u32 s22, s22, 1\n"
#define OP_CMP_SAND(r) "v_cmp_gt_f32 s[20:21], " #r ", %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n"

KERNEL(k_fma, I8(OP_FMA))
KERNEL(k_fmac, I8(OP_FMAC))
KERNEL(k_mul, I8(OP_MUL))
KERNEL(k_add, I8(OP_ADD))
KERNEL(k_mul_sgpr, I8(OP_MULS))
KERNEL(k_fma_sgpr, I8(OP_FMAS))
KERNEL(k_max, I8(OP_MAX))
KERNEL(k_med3, I8(OP_MED3))
KERNEL(k_cmp_vcc, I8(OP_CMP))
KERNEL(k_cmp_sgpr, I8(OP_CMPS))
KERNEL(k_cndmask_vcc, I8(OP_CND))
KERNEL(k_cndmask_sgpr, I8(OP_CNDS))
KERNEL(k_rcp, I8(OP_RCP))
KERNEL(k_exp, I8(OP_EXP))
KERNEL(k_rsq, I8(OP_RSQ))
KERNEL(k_sqrt, I8(OP_SQRT))
KERNEL(k_and, I8(OP_ANDB))
KERNEL(k_lshl, I8(OP_LSHL))
KERNEL(k_add_u32, I8(OP_ADDU))
KERNEL(k_cvt_i32_f32, I8(OP_CVTFI))
KERNEL(k_cvt_f32_i32, I8(OP_CVTIF))
KERNEL(k_bfe, I8(OP_BFE))
KERNEL(k_mov_dpp_quad, I8(OP_MOVDPP))
KERNEL(k_add_dpp_quad, I8(OP_ADDDPP))
KERNEL(k_add_dpp_rowshr, I8(OP_ADDDPPR))
KERNEL(k_readlane, I8(OP_READLANE))
KERNEL(k_readfirstlane, I8(OP_READFL))
KERNEL(k_mad_u32_u24, I8(OP_MAD24))
KERNEL(k_subrev, I8(OP_SUBREV))
KERNEL(k_mov, I8(OP_MOV))
KERNEL(k_fma_plus_1salu, I8(OP_FMA_SALU))
KERNEL(k_fma_plus_2salu, I8(OP_FMA_SALU2))
KERNEL(k_cmp_sgpr_plus_sand, I8(OP_CMP_SAND))

// packed fp32: 64-bit register pairs
__global__ void k_pk_fma(float *out, unsigned long long *ticks, float s0) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {1.f * threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
       a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
#define OP_PKFMA(r) "v_pk_fma_f32 " #r ", " #r ", %8, %9\n"
        asm volatile(REP8(I8(OP_PKFMA))
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s0;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
__global__ void k_pk_mul(float *out, unsigned long long *ticks, float s0) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {1.f * threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
       a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
#define OP_PKMUL(r) "v_pk_mul_f32 " #r ", " #r ", %8\n"
        asm volatile(REP8(I8(OP_PKMUL))
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s0;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// dependent chain: latency of back-to-back dependent v_fma / v_mul+v_add
__global__ void k_dep_fma(float *out, unsigned long long *ticks, float s0) {
    float a0 = threadIdx.x;
    float b = 1.0001f, c = 0.5f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
#define OP_D(r) "v_fma_f32 %0, %0, %1, %2\n"
        asm volatile(REP8(REP8(OP_D(0))) : "+v"(a0) : "v"(b), "v"(c));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + s0;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// LDS float atomics: 64 ds_add_f32 per trip, distinct addresses per lane (conflict-free) vs 4-lane same address
template <int SAME>
__global__ void k_ds_add(float *out, unsigned long long *ticks, float s0) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = threadIdx.x; j < 64 * (int)(blockDim.x >> 6); j += blockDim.x) lds[j] = 0.f;
    __syncthreads();
    float *p = lds + wave * 64 + (SAME ? (lane / SAME) * SAME : lane);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) __hip_atomic_fetch_add(p, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = lds[threadIdx.x % (64 * (blockDim.x >> 6))];
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// L1/L2-resident global loads: 64 dword loads per trip from a 4 KB window per wave
__global__ void k_gload(float *out, unsigned long long *ticks, float s0) {
    const float *src = out + (1 << 22);   // separate region, pre-zeroed
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) acc += __builtin_nontemporal_load(src + ((k * 64 + lane + i) & 1023));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + s0;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*kern_t)(float *, unsigned long long *, float);
struct Entry { const char *name; kern_t fn; int lds_per_wave; double instr_per_trip; };

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int CUS = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, CUS, prop.clockRate);
    float *out;
    unsigned long long *ticks;
    CHECK(hipMalloc(&out, sizeof(float) * ((1 << 22) + 4096)));
    CHECK(hipMemset(out, 0, sizeof(float) * ((1 << 22) + 4096)));
    CHECK(hipMalloc(&ticks, sizeof(unsigned long long) * 65536));
    std::vector<Entry> es = {
        {"v_fma_f32", k_fma, 0, 64}, {"v_fmac_f32", k_fmac, 0, 64}, {"v_mul_f32", k_mul, 0, 64}, {"v_add_f32", k_add, 0, 64},
        {"v_mul_f32 (sgpr src)", k_mul_sgpr, 0, 64}, {"v_fma_f32 (sgpr src)", k_fma_sgpr, 0, 64},
        {"v_subrev_f32", k_subrev, 0, 64}, {"v_mov_b32", k_mov, 0, 64},
        {"v_max_f32", k_max, 0, 64}, {"v_med3_f32", k_med3, 0, 64},
        {"v_cmp_gt_f32 vcc", k_cmp_vcc, 0, 64}, {"v_cmp_gt_f32 sgpr", k_cmp_sgpr, 0, 64},
        {"v_cndmask vcc", k_cndmask_vcc, 0, 64}, {"v_cndmask sgpr", k_cndmask_sgpr, 0, 64},
        {"v_rcp_f32", k_rcp, 0, 64}, {"v_exp_f32", k_exp, 0, 64}, {"v_rsq_f32", k_rsq, 0, 64}, {"v_sqrt_f32", k_sqrt, 0, 64},
        {"v_and_b32", k_and, 0, 64}, {"v_lshlrev_b32", k_lshl, 0, 64}, {"v_add_u32", k_add_u32, 0, 64},
        {"v_cvt_i32_f32", k_cvt_i32_f32, 0, 64}, {"v_cvt_f32_i32", k_cvt_f32_i32, 0, 64}, {"v_bfe_u32", k_bfe, 0, 64},
        {"v_mad_u32_u24", k_mad_u32_u24, 0, 64},
        {"v_mov_b32_dpp quad", k_mov_dpp_quad, 0, 64}, {"v_add_f32_dpp quad", k_add_dpp_quad, 0, 64},
        {"v_add_f32_dpp row_shr", k_add_dpp_rowshr, 0, 64},
        {"v_readlane_b32", k_readlane, 0, 64}, {"v_readfirstlane_b32", k_readfirstlane, 0, 64},
        {"v_pk_fma_f32", k_pk_fma, 0, 64}, {"v_pk_mul_f32", k_pk_mul, 0, 64},
        {"dependent v_fma chain", k_dep_fma, 0, 64},
        {"ds_add_f32 distinct addr", k_ds_add<0>, 256, 64}, {"ds_add_f32 4 lanes/addr", k_ds_add<4>, 256, 64},
        {"ds_add_f32 16 lanes/addr", k_ds_add<16>, 256, 64},
        {"global_load_dword L1-hit", k_gload, 0, 64},
        {"v_fma + 1 s_and_b32 (per VALU)", k_fma_plus_1salu, 0, 64}, {"v_fma + 2 SALU (per VALU)", k_fma_plus_2salu, 0, 64},
        {"v_cmp sgpr + s_and_b64 (per pair)", k_cmp_sgpr_plus_sand, 0, 64},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-36s | waves/SIMD: cycles per wave-instruction per SIMD by s_memtime [and by wall clock @2.4GHz]\n", "instruction");
    const int WPS[4] = {1, 2, 4, 8};
    for (auto &e : es) {
        printf("%-36s |", e.name);
        for (int wi = 0; wi < 4; ++wi) {
            const int wps = WPS[wi];
            // wps waves per SIMD: blocks of 256*min(wps,4) threads, (wps > 4 ? 2 : 1) blocks per CU
            const int bt = 256 * (wps > 4 ? 4 : wps), bpc = wps > 4 ? wps / 4 : 1;
            const int blocks = CUS * bpc;
            const size_t lds = (size_t)e.lds_per_wave * (bt / 64);
            e.fn<<<blocks, bt, lds>>>(out, ticks, 1.0f);   // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            e.fn<<<blocks, bt, lds>>>(out, ticks, 1.0f);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const int nw = blocks * bt / 64;
            std::vector<unsigned long long> h(nw);
            CHECK(hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
            double avg = 0;
            for (auto t : h) avg += (double)t;
            avg /= nw;
            const double ipw = e.instr_per_trip * ITERS;          // instructions per wave
            // per SIMD: wps waves share the SIMD; cycles per instruction = wave's elapsed ticks / (ipw * wps)
            const double cpi_tick = avg / (ipw * wps);
            const double cpi_wall = (ms * 1e-3 * 2.4e9) / (ipw * wps);
            printf(" %d: %6.2f [%6.2f] |", wps, cpi_tick, cpi_wall);
        }
        printf("\n");
    }
    return 0;
}
