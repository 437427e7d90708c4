// valu_ubench2.hip -- round-4 additions to valu_ubench.hip (gfx950): what does a select cost, by the register its lane mask
// comes from and by its encoding; do three-operand min / max and packed fp32 with scalar sources pay; what do the two memory
// round trips of a raster visit (scalar record load, per-lane edge-block load) cost in latency.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_ubench2.hip -o /tmp/valu_ubench2 && /tmp/valu_ubench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
#define REP8(X) X X X X X X X X
constexpr int ITERS = 512;

// PRE runs once before the loop (sets up s[20:23] / vcc); BODY: %0..%7 accumulators, %8 %9 vector sources, %10 scalar source
#define KERNEL(NAME, PRE, BODY)                                                                                \
    __global__ void NAME(float *out, unsigned long long *ticks, float s0) {                                   \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
              a7 = a0 + 7;                                                                                     \
        float b = 1.0001f + 1e-7f * threadIdx.x, c = 0.5f;                                                    \
        asm volatile(PRE ::: "vcc", "s20", "s21", "s22", "s23");                                               \
        unsigned long long t0 = __builtin_readcyclecounter();                                                  \
        for (int i = 0; i < ITERS; ++i) {                                                                      \
            asm volatile(REP8(BODY)                                                                            \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                         : "v"(b), "v"(c), "s"(s0) : "vcc", "scc", "s20", "s21", "s22", "s23");               \
        }                                                                                                      \
        unsigned long long t1 = __builtin_readcyclecounter();                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                   \
        if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;           \
    }
#define I8(OP) OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7)
#define SETUP "s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333\n s_mov_b64 s[22:23], 0x0f0f\n"

#define OP_CND32(r) "v_cndmask_b32_e32 " #r ", " #r ", %8, vcc\n"
#define OP_CND64V(r) "v_cndmask_b32_e64 " #r ", " #r ", %8, vcc\n"
#define OP_CND64S(r) "v_cndmask_b32_e64 " #r ", " #r ", %8, s[22:23]\n"
#define OP_CND32_0(r) "v_cndmask_b32_e32 " #r ", 0, " #r ", vcc\n"
// compare + select pairs, the select consuming the compare's mask
#define OP_PAIR_VCC32(r) "v_cmp_gt_f32_e32 vcc, %8, " #r "\n v_cndmask_b32_e32 " #r ", " #r ", %9, vcc\n"
#define OP_PAIR_VCC64(r) "v_cmp_gt_f32_e32 vcc, %8, " #r "\n v_cndmask_b32_e64 " #r ", " #r ", %9, vcc\n"
#define OP_PAIR_SGPR(r) "v_cmp_gt_f32_e64 s[20:21], %8, " #r "\n s_nop 1\n v_cndmask_b32_e64 " #r ", " #r ", %9, s[20:21]\n"
// the same with an independent VALU between compare and select (what a scheduler can do instead of the s_nop)
#define OP_PAIR_SGPR_FILL(r) "v_cmp_gt_f32_e64 s[20:21], %8, " #r "\n v_mul_f32 " #r ", " #r ", %8\n v_cndmask_b32_e64 " #r ", " #r ", %9, s[20:21]\n"
#define OP_PAIR_VCC_FILL(r) "v_cmp_gt_f32_e32 vcc, %8, " #r "\n v_mul_f32 " #r ", " #r ", %8\n v_cndmask_b32_e32 " #r ", " #r ", %9, vcc\n"
#define OP_MIN3(r) "v_min3_f32 " #r ", " #r ", %8, %9\n"
#define OP_MAX3(r) "v_max3_f32 " #r ", " #r ", %8, %9\n"
#define OP_CLASS(r) "v_cmp_class_f32_e64 s[20:21], " #r ", %8\n"
#define OP_MAXCLAMP(r) "v_max_f32_e64 " #r ", " #r ", " #r " clamp\n"
#define OP_MULCLAMP(r) "v_mul_f32_e64 " #r ", " #r ", %8 clamp\n"
#define OP_ADDC(r) "v_add_f32_e32 " #r ", 1.0, " #r "\n"
#define OP_MULLIT(r) "v_mul_f32_e32 " #r ", 0x3fb8aa3b, " #r "\n"
#define OP_SAND(r) "s_and_b64 s[20:21], s[20:21], s[22:23]\n"
#define OP_SFF1(r) "s_ff1_i32_b64 s20, s[22:23]\n"
#define OP_CMP_SAND2(r) "v_cmp_gt_f32_e64 s[20:21], " #r ", %8\n v_cmp_lt_f32_e64 s[22:23], " #r ", %9\n s_and_b64 s[22:23], s[22:23], s[20:21]\n"
#define OP_CMPX(r) "v_cmpx_gt_f32_e32 " #r ", %8\n s_mov_b64 exec, -1\n"

KERNEL(k_cnd_e32_vcc, SETUP, I8(OP_CND32))
KERNEL(k_cnd_e64_vcc, SETUP, I8(OP_CND64V))
KERNEL(k_cnd_e64_sgpr, SETUP, I8(OP_CND64S))
KERNEL(k_cnd_e32_vcc_const0, SETUP, I8(OP_CND32_0))
KERNEL(k_pair_vcc32, SETUP, I8(OP_PAIR_VCC32))
KERNEL(k_pair_vcc64, SETUP, I8(OP_PAIR_VCC64))
KERNEL(k_pair_sgpr, SETUP, I8(OP_PAIR_SGPR))
KERNEL(k_pair_sgpr_fill, SETUP, I8(OP_PAIR_SGPR_FILL))
KERNEL(k_pair_vcc_fill, SETUP, I8(OP_PAIR_VCC_FILL))
KERNEL(k_min3, SETUP, I8(OP_MIN3))
KERNEL(k_max3, SETUP, I8(OP_MAX3))
KERNEL(k_cmp_class, SETUP, I8(OP_CLASS))
KERNEL(k_max_clamp, SETUP, I8(OP_MAXCLAMP))
KERNEL(k_mul_clamp, SETUP, I8(OP_MULCLAMP))
KERNEL(k_add_inline_const, SETUP, I8(OP_ADDC))
KERNEL(k_mul_literal, SETUP, I8(OP_MULLIT))
KERNEL(k_s_and_b64, SETUP, I8(OP_SAND))
KERNEL(k_s_ff1, SETUP, I8(OP_SFF1))
KERNEL(k_cmp2_sand, SETUP, I8(OP_CMP_SAND2))

// packed fp32 with a scalar register pair as one source (the face record lives in SGPRs)
#define PK_KERNEL(NAME, OPSTR)                                                                                 \
    __global__ void NAME(float *out, unsigned long long *ticks, float s0) {                                   \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                  \
        f2 a0 = {1.f * threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, \
           a6 = a0 + 6.f, a7 = a0 + 7.f;                                                                       \
        f2 b = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};                                                          \
        asm volatile("s_mov_b32 s20, 0x3f800001\n s_mov_b32 s21, 0x3f800002\n" ::: "s20", "s21");              \
        unsigned long long t0 = __builtin_readcyclecounter();                                                  \
        for (int i = 0; i < ITERS; ++i) {                                                                      \
            asm volatile(REP8(I8(OPSTR))                                                                       \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                         : "v"(b), "v"(c) : "s20", "s21");                                                     \
        }                                                                                                      \
        unsigned long long t1 = __builtin_readcyclecounter();                                                  \
        f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s0;                                           \
        if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;           \
    }
#define OP_PKMULS(r) "v_pk_mul_f32 " #r ", " #r ", s[20:21]\n"
#define OP_PKMULV(r) "v_pk_mul_f32 " #r ", " #r ", %8\n"
#define OP_PKADDV(r) "v_pk_add_f32 " #r ", " #r ", %8\n"
#define OP_PKFMAS(r) "v_pk_fma_f32 " #r ", " #r ", s[20:21], %9\n"
PK_KERNEL(k_pk_mul_sgpr, OP_PKMULS)
PK_KERNEL(k_pk_mul_vgpr, OP_PKMULV)
PK_KERNEL(k_pk_add_vgpr, OP_PKADDV)
PK_KERNEL(k_pk_fma_sgpr, OP_PKFMAS)

// ---- latencies: a dependent chain of loads, one wave per SIMD, cycles per load ---------------------------------------------
// scalar: s_load_dwordx16 whose address depends on the previous load's data (all zeros -> same 256-byte record / a walk)
__global__ void k_lat_sload(float *out, unsigned long long *ticks, float s0) {
    const unsigned *src = (const unsigned *)(out + (1 << 22));   // zero-filled region
    unsigned off = 0, acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        typedef unsigned v16u __attribute__((ext_vector_type(16)));
        const v16u v = *(const __attribute__((address_space(4))) v16u *)(src + off + (i & 15) * 64);
        off = __builtin_amdgcn_readfirstlane(v[0]);     // 0: the next address depends on this load
        acc += v[3];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + s0;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = (t1 - t0) * 64 / 1;   // scaled: 1 "instr" per trip
}
// vector: global_load_dwordx4 per lane, address depends on the previous load (L1-resident 4 KB window per wave)
template <int STRIDE_KB>
__global__ void k_lat_vload(float *out, unsigned long long *ticks, float s0) {
    const char *src = (const char *)(out + (1 << 22));
    const int lane = threadIdx.x & 63;
    unsigned off = (unsigned)lane * 16u;
    float acc = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        const float4 v = *(const float4 *)(src + off + (unsigned)((i * STRIDE_KB * 1024) & ((1 << 22) - 1)));
        off = (unsigned)lane * 16u + (unsigned)__float_as_int(v.x);   // 0
        acc += v.y;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + s0;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = (t1 - t0) * 64;
}

typedef void (*kern_t)(float *, unsigned long long *, float);
struct Entry { const char *name; kern_t fn; double instr_per_trip; int max_wps; };

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int CUS = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, CUS, prop.clockRate);
    float *out;
    unsigned long long *ticks;
    CHECK(hipMalloc(&out, sizeof(float) * ((1 << 22) + (1 << 21))));
    CHECK(hipMemset(out, 0, sizeof(float) * ((1 << 22) + (1 << 21))));
    CHECK(hipMalloc(&ticks, sizeof(unsigned long long) * 65536));
    std::vector<Entry> es = {
        {"v_cndmask_e32 vcc", k_cnd_e32_vcc, 64, 8}, {"v_cndmask_e64 vcc", k_cnd_e64_vcc, 64, 8},
        {"v_cndmask_e64 sgpr", k_cnd_e64_sgpr, 64, 8}, {"v_cndmask_e32 0,v,vcc", k_cnd_e32_vcc_const0, 64, 8},
        {"cmp->vcc + cnd_e32 vcc (pair)", k_pair_vcc32, 64, 8}, {"cmp->vcc + cnd_e64 vcc (pair)", k_pair_vcc64, 64, 8},
        {"cmp->sgpr + s_nop 1 + cnd_e64 (pair)", k_pair_sgpr, 64, 8},
        {"cmp->sgpr + v_mul + cnd_e64 (triple)", k_pair_sgpr_fill, 64, 8},
        {"cmp->vcc + v_mul + cnd_e32 (triple)", k_pair_vcc_fill, 64, 8},
        {"v_min3_f32", k_min3, 64, 8}, {"v_max3_f32", k_max3, 64, 8}, {"v_cmp_class_f32 sgpr", k_cmp_class, 64, 8},
        {"v_max_f32 clamp", k_max_clamp, 64, 8}, {"v_mul_f32 clamp (e64)", k_mul_clamp, 64, 8},
        {"v_add_f32 1.0 (inline const)", k_add_inline_const, 64, 8}, {"v_mul_f32 literal", k_mul_literal, 64, 8},
        {"s_and_b64", k_s_and_b64, 64, 8}, {"s_ff1_i32_b64", k_s_ff1, 64, 8},
        {"2 cmp->sgpr + s_and (triple)", k_cmp2_sand, 64, 8},
        {"v_pk_mul_f32 sgpr pair", k_pk_mul_sgpr, 64, 8}, {"v_pk_mul_f32 vgpr", k_pk_mul_vgpr, 64, 8},
        {"v_pk_add_f32 vgpr", k_pk_add_vgpr, 64, 8}, {"v_pk_fma_f32 sgpr pair", k_pk_fma_sgpr, 64, 8},
        {"LATENCY s_load_dwordx16 dependent", k_lat_sload, 64, 1},
        {"LATENCY global_load_dwordx4 dep, same 1 KB (L1)", k_lat_vload<0>, 64, 1},
        {"LATENCY global_load_dwordx4 dep, stride 64 KB (L2)", k_lat_vload<64>, 64, 1},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-52s | waves/SIMD: cycles per wave-instruction per SIMD by s_memtime [and by wall clock @2.4GHz]\n", "instruction");
    const int WPS[4] = {1, 2, 4, 8};
    for (auto &e : es) {
        printf("%-52s |", e.name);
        for (int wi = 0; wi < 4; ++wi) {
            const int wps = WPS[wi];
            if (wps > e.max_wps) break;
            const int bt = 256 * (wps > 4 ? 4 : wps), bpc = wps > 4 ? wps / 4 : 1;
            const int blocks = CUS * bpc;
            e.fn<<<blocks, bt, 0>>>(out, ticks, 1.0f);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            e.fn<<<blocks, bt, 0>>>(out, ticks, 1.0f);
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
            const double ipw = e.instr_per_trip * ITERS;
            printf(" %d: %6.2f [%6.2f] |", wps, avg / (ipw * wps), (ms * 1e-3 * 2.4e9) / (ipw * wps));
        }
        printf("\n");
    }
    return 0;
}
