// raster_backward.h -- backward kernels (included by raster.hip only): k_raster_backward_fm, the face-major kernel that
// runs in production, and k_raster_backward, the pixel-major variant kept for A/B runs and very large TS.
#pragma once
#include "raster_core.h"
#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------------
template <int RGB>
__global__ __launch_bounds__(BLK_THREADS) void k_raster_backward(const RasterArgs A) {
    __shared__ int s_list[LIST_CAP];
    __shared__ int s_wcnt[BLK_THREADS / 64];
    Tile t;
    tile_setup(t, A);
    const int F = A.F, IS = A.IS, TS = A.TS;
    const size_t npix = (size_t)IS * IS;
    const size_t pn = (size_t)t.row * IS + t.xi;
    const float4 *__restrict__ bbox_n = A.bbox + (size_t)t.n * F;
    const float *__restrict__ rec_n = A.rec + (size_t)t.n * F * REC;
    const float *__restrict__ tex_n = A.textures + (size_t)(t.n / A.tex_group) * F * TS * 3;

    float ssum = 1.f, smax = 0.f, oc0 = 0.f, oc1 = 0.f, oc2 = 0.f, oa = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    if (t.valid) {
        const float *ag = A.aggrs + (size_t)t.n * 2 * npix + pn;
        ssum = ag[0]; smax = ag[npix];
        const float *sc = A.soft_colors + (size_t)t.n * 4 * npix + pn;
        oc0 = sc[0]; oc1 = sc[npix]; oc2 = sc[2 * npix]; oa = sc[3 * npix];
        if (A.grad_pooled) {  // avg_pool2d backward fused: every pixel of a 2x2 cell sees g/4
            const int H = IS >> 1;
            const float *gp = A.grad_colors + ((size_t)t.n * 4 * H + (t.row >> 1)) * H + (t.xi >> 1);
            const size_t hp = (size_t)H * H;
            g0 = 0.25f * gp[0]; g1 = 0.25f * gp[hp]; g2 = 0.25f * gp[2 * hp]; g3 = 0.25f * gp[3 * hp];
        } else {
            const float *gp = A.grad_colors + (size_t)t.n * 4 * npix + pn;
            g0 = gp[0]; g1 = gp[npix]; g2 = gp[2 * npix]; g3 = gp[3 * npix];
        }
    }

    const int *sb_ids;
    const int ncand = superblock_list(A, t, sb_ids);      // the super-block's pre-binned faces (or all F)
    for (int f0 = 0; f0 < ncand; f0 += LIST_CAP) {
        const int f1 = min(ncand, f0 + LIST_CAP);
        if (f0 > 0) __syncthreads();
        const int count = build_list(s_list, s_wcnt, bbox_n, sb_ids, f0, f1, t);
        if (!t.wave_on) continue;
        for (int base = 0; base < count; base += 64) {
            const int li = base + t.lane;
            const int fcand = li < count ? s_list[li] : -1;
            bool hit = false;
            if (fcand >= 0) {
                const float4 bb = bbox_n[fcand];
                hit = !(t.wxlo > bb.y || t.wxhi < bb.x || t.wylo > bb.w || t.wyhi < bb.z);
                if (hit) {  // one lane per candidate face: exact-ish tile/triangle test
                    const float4 *q = (const float4 *)(rec_n + (size_t)fcand * REC + R_INV);
                    hit = tile_may_hit(q[0], q[1], q[2], 0.5f * (t.wxlo + t.wxhi), 0.5f * (t.wylo + t.wyhi),
                                       0.5f * (t.wxhi - t.wxlo), 0.5f * (t.wyhi - t.wylo), A.thr);
                }
            }
            unsigned long long m = __ballot(hit);
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const int f = __builtin_amdgcn_readlane(fcand, b);
                Face fc;
                load_face(fc, rec_n + (size_t)f * REC);
                float gv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                float gt0 = 0.f, gt1 = 0.f, gt2 = 0.f;  // texture gradient of this lane (at texel tix)
                int tix = 0;
                bool contrib = false;
                Pair p;
                if (t.valid && eval_pair(p, fc, t.xp, t.yp, A.threshold, A.nis)) {
                    float c_xy = g3 * ((1.f - oa) * __builtin_amdgcn_rcpf(fmaxf(1.f - p.frag, 1e-6f)));  // :584
                    float q0, q1, q2;
                    const float zp = clip_depth(q0, q1, q2, p, fc);
                    if (!(zp < A.near_ || zp > A.far_)) {  // :592 -- drops the alpha term as well
                        contrib = true;
                        if (RGB == 0) {
                            if ((float)f == smax) {  // :596
                                tix = texel_index(q0, q1, A.R);
                                gt0 = g0; gt1 = g1; gt2 = g2;
                            }
                        } else if (fc.front() || A.double_side) {
                            const float zn = div_r(A.far_ - zp, A.far_ - A.near_, A.r_range);
                            const float ps = p.frag * __expf((zn - smax) * A.inv_gamma) * __builtin_amdgcn_rcpf(ssum);  // :608
                            tix = texel_index(q0, q1, A.R);
                            const float *tx = tex_n + ((size_t)f * TS + tix) * 3;
                            gt0 = ps * g0; gt1 = ps * g1; gt2 = ps * g2;
                            float c_rgb = g0 * (tx[0] - oc0);
                            c_rgb += g1 * (tx[1] - oc1);
                            c_rgb += g2 * (tx[2] - oc2);
                            c_rgb *= ps;
                            c_xy += c_rgb * __builtin_amdgcn_rcpf(p.frag);
                            const float c_z = -(c_rgb * A.inv_gamma * A.r_range) * zp * zp;  // :624
                            gv[2] = c_z * q0 * fc.g<R_RZ0>() * fc.g<R_RZ0>();
                            gv[5] = c_z * q1 * fc.g<R_RZ1>() * fc.g<R_RZ1>();
                            gv[8] = c_z * q2 * fc.g<R_RZ2>() * fc.g<R_RZ2>();
                        }
                        c_xy *= p.frag * (1.f - p.frag) * (-A.nis);  // :632
                        const float k2 = 2.f * p.sign * c_xy;        // :640
                        const float b0 = k2 * p.b0, b1 = k2 * p.b1, b2 = k2 * p.b2;
                        gv[0] = b0 * p.dx; gv[1] = b0 * p.dy;
                        gv[3] = b1 * p.dx; gv[4] = b1 * p.dy;
                        gv[6] = b2 * p.dx; gv[7] = b2 * p.dy;
                    }
                }
                if (!__any(contrib)) continue;
                if (A.need_gf) {
                    float mine = 0.f;
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const float s = wave_sum_full(gv[k]);
                        if (t.lane == k) mine = s;
                    }
                    if (t.lane < 9) atomicAdd(A.grad_faces + ((size_t)t.n * F + f) * 9 + t.lane, mine);
                }
                if (A.need_gt) {
                    float *gtf = A.grad_textures + ((size_t)t.n * F + f) * TS * 3;
                    if (TS == 1) {
                        const float s0 = wave_sum_full(gt0), s1 = wave_sum_full(gt1), s2 = wave_sum_full(gt2);
                        if (t.lane < 3) atomicAdd(gtf + t.lane, t.lane == 0 ? s0 : (t.lane == 1 ? s1 : s2));
                    } else if (gt0 != 0.f || gt1 != 0.f || gt2 != 0.f) {
                        atomicAdd(gtf + tix * 3 + 0, gt0);
                        atomicAdd(gtf + tix * 3 + 1, gt1);
                        atomicAdd(gtf + tix * 3 + 2, gt2);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Start order of the face-major backward's waves.  A wave owns one face for its whole life and faces differ ~50x in
// work (a front face under a texel-gradient launch walks ~13 visits of ~900 issue cycles, a back face is culled after one
// pass), so with waves started in index order the launch ends on a few heavy faces running alone on otherwise empty SIMDs:
// N = 16 meshes cost 8.6 us per mesh against 6.2 us at N = 128.  This kernel sorts, per XCD (each XCD keeps its contiguous
// eighth of every mesh's faces -- the L2 locality of the saved state) and per group of `G` meshes, the faces by DESCENDING
// estimated work (counting sort, 1024 buckets), so the heavy faces of all meshes of the group start first and the launch
// drains on cheap ones.  Work estimate: 4x4 sub-tiles under the dilated bbox; / 8 for faces the state cull will remove --
// mode 1 (texel gradients only): back faces; mode 2 (silhouette): faces whose corners and centroid all sit on alpha == 1.
// The order changes no result (every face is still reduced by one wave in its own fixed order).
#define ORDER_KEYS 1024
#define ORDER_MAX_ENTRIES 16384
#define ORDER_THREADS 1024
__global__ __launch_bounds__(ORDER_THREADS) void k_face_order(const unsigned short *__restrict__ cost, const float *__restrict__ rec,
                                                              const float *__restrict__ alpha, int *__restrict__ order, int N,
                                                              int F, int IS, int G, int mode, int split) {
    __shared__ int s_hist[ORDER_KEYS];
    __shared__ int s_wsum[ORDER_THREADS / 64];
    __shared__ unsigned short s_key[ORDER_MAX_ENTRIES];
    const int xcd = blockIdx.x, g = blockIdx.y, per = F >> 3;
    const int m0 = g * G, gl = min(G, N - m0), E = gl * per;
    for (int k = threadIdx.x; k < ORDER_KEYS; k += ORDER_THREADS) s_hist[k] = 0;
    __syncthreads();
    const float h = 0.5f * IS;
    for (int e = threadIdx.x; e < E; e += ORDER_THREADS) {
        const int ml = e / per, f = fm_owned_face(xcd, e % per, per, split);
        const size_t fi = (size_t)(m0 + ml) * F + f;
        const unsigned c = cost[fi];                       // k_face_setup: sub-tiles under the bbox | front << 15
        int nt = (int)(c & 0x7fffu);
        if (mode == 1 && !(c & 0x8000u)) nt >>= 3;
        if (mode == 2 && nt > 0) {
            const float *r = rec + fi * REC;
            const float *ap = alpha + (size_t)(m0 + ml) * IS * IS;
            const float cx[4] = {r[R_X0], r[R_X1], r[R_X2], (r[R_X0] + r[R_X1] + r[R_X2]) * (1.f / 3.f)};
            const float cy[4] = {r[R_Y0], r[R_Y1], r[R_Y2], (r[R_Y0] + r[R_Y1] + r[R_Y2]) * (1.f / 3.f)};
            bool opaque = true;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int xi = min(max((int)(cx[c4] * h + h), 0), IS - 1), yi = min(max((int)(cy[c4] * h + h), 0), IS - 1);
                opaque &= ap[(size_t)(IS - 1 - yi) * IS + xi] == 1.f;
            }
            if (opaque) nt >>= 3;
        }
        const int key = min(nt, ORDER_KEYS - 1);
        s_key[e] = (unsigned short)key;
        atomicAdd(&s_hist[key], 1);
    }
    __syncthreads();
    // exclusive prefix over DESCENDING keys: thread t owns key 1023 - t
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kb = ORDER_KEYS - 1 - (int)threadIdx.x;
    const int mine = s_hist[kb];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
    __syncthreads();
    s_hist[kb] = base;
    __syncthreads();
    int *out = order + ((size_t)g * 8 + xcd) * ((size_t)G * per);
    for (int e = threadIdx.x; e < E; e += ORDER_THREADS) {
        const int pos = atomicAdd(&s_hist[s_key[e]], 1);
        out[pos] = ((e / per) << 16) | fm_owned_face(xcd, e % per, per, split);
    }
}

// ------------------------------------------------------------------------------------------------
// Face-major backward.  Given the saved per-pixel forward state, every (pixel, face) contribution to
// the gradient is independent (:531-655), so the loop nest can be turned inside out: ONE WAVEFRONT PER
// FACE walks the 8x8 pixel tiles under that face's dilated bounding box, accumulates the 9 vertex
// gradients in registers (texel gradients in a per-wave LDS array) across all tiles, reduces across the
// 64 lanes ONCE and stores.  No global atomics, no per-(tile, face) reduction, deterministic results; the
// face record lives in SGPRs for the whole walk.  Per-pixel state is re-read once per overlapping face
// (~6x, L1/L2 hits: consecutive faces of a subdivided mesh are spatial neighbours and share a workgroup).
#ifndef FM_WAVES
#define FM_WAVES 1   // faces (wavefronts) per workgroup of the face-major backward: 1 = finest scheduling granularity,
                     // no straggler waves holding a CU slot (measured 1 < 2 < 4 < 8 in time)
#endif
#ifndef FM_RELOAD_PER_TILE
#define FM_RELOAD_PER_TILE 1
#endif
#ifndef FM_TW
#define FM_TW 4   // sub-tile width / height in pixels (8x8 = one tile per wave visit, 4x4 = four)
#define FM_TH 4
#endif
#define FM_NQ (64 / (FM_TW * FM_TH))
#ifndef FM_VCONST
#define FM_VCONST 1
#endif
#ifndef FM_TEXMERGE
#define FM_TEXMERGE 2   // DPP pre-merge steps before the LDS texel atomics: 0 none, 1 = x^1, 2 = x^1 then x^2
                       // (a third, vertical step measured slower)
#endif
#ifndef FM_VREC
#define FM_VREC 1         // 1: the face's inverse barycentric matrix and corner coordinates as VGPR operands (FaceV)
#endif
#ifndef FM_FMA_ACC
#define FM_FMA_ACC 1      // 1: gradient accumulators updated with explicit fused multiply-adds (the summation order of the
#endif                    // reference's atomics is not defined either, so these sums are not pinned to a rounding sequence)
#if FM_FMA_ACC
#define FM_ACC(acc, a, b) acc = fmaf(a, b, acc)
#else
#define FM_ACC(acc, a, b) acc += (a) * (b)
#endif
#ifndef FM_SLOTS
#define FM_SLOTS 1        // 1: sub-tile hand-out through LDS slots for the silhouette variant (A/B: -DFM_SLOTS=0 = v_readlane
#endif                    // rounds everywhere, round 2's form; 2 = slots for every variant).  Measured on MI355X, us per launch
                          // N = 16 / 128, readlane -> slots: silhouette 82.6 -> 73.7 / 490 -> 451 (-8..11 %; -14 % at F = 5120,
                          // IS = 1024); colour variants the other way -- texel-only 121 -> 125 / 812 -> 826, vertex + texel
                          // 206 -> 241 / 1321 -> 1571: they sit at the 72-VGPR budget of 7 waves, the slot's 4 values spill
#ifndef FM_STATE_CULL
#define FM_STATE_CULL 1   // sub-tile skips from the saved forward state inside the culling pass (A/B: -DFM_STATE_CULL=0)
#endif
#ifndef FM_TEXCOPY
#define FM_TEXCOPY 4   // private copies of a wave's LDS texel accumulators (power of two): neighbouring pixels share a
#endif                 // texel, and same-address ds_add_f32 from one wave serialise -- spread them over copies
#define FM_TEX_STRIDE(TS) (((TS) * 3) | 1)   // odd stride: copy c of a texel lands in another bank
// Texel-gradient accumulation of the face-major backward (TS > 1): 3 ds_add_f32 per visit into the wave's LDS
// accumulators.  Neighbouring pixels mostly fall into the same texel, and the LDS atomic pipe -- shared by every wave
// of the CU -- saturates (SQ_WAIT_INST_LDS 21 % of wave time, the kernel 35 % slower than without the atomics).
// So horizontally adjacent lanes holding the same texel are first summed with DPP quad permutes (the partner's value
// is read only when the partner is active at this point: bound_ctrl off -> `old`), and only the surviving lane of
// each run issues the atomics.  Deterministic; only the summation order differs from lane-by-lane atomics.
__device__ __forceinline__ float dpp_f(float old, float v, const int ctrl_sel) {
    // ctrl_sel: 0 -> quad_perm [1,0,3,2] (x^1), 1 -> quad_perm [2,3,0,1] (x^2), 2 -> row_shl:4, 3 -> row_shr:4
    const int o = __float_as_int(old), i = __float_as_int(v);
    int r;
    if (ctrl_sel == 0) r = __builtin_amdgcn_update_dpp(o, i, 0xB1, 0xf, 0xf, false);
    else if (ctrl_sel == 1) r = __builtin_amdgcn_update_dpp(o, i, 0x4E, 0xf, 0xf, false);
    else if (ctrl_sel == 2) r = __builtin_amdgcn_update_dpp(o, i, 0x104, 0xf, 0xf, false);
    else r = __builtin_amdgcn_update_dpp(o, i, 0x114, 0xf, 0xf, false);
    return __int_as_float(r);
}
__device__ __forceinline__ int dpp_i(int old, int v, const int ctrl_sel) {
    if (ctrl_sel == 0) return __builtin_amdgcn_update_dpp(old, v, 0xB1, 0xf, 0xf, false);
    if (ctrl_sel == 1) return __builtin_amdgcn_update_dpp(old, v, 0x4E, 0xf, 0xf, false);
    if (ctrl_sel == 2) return __builtin_amdgcn_update_dpp(old, v, 0x104, 0xf, 0xf, false);
    return __builtin_amdgcn_update_dpp(old, v, 0x114, 0xf, 0xf, false);
}
__device__ __forceinline__ void texel_accumulate(float *my_tex, int tix, float a, float b, float c, int lane) {
#if FM_TEXMERGE >= 1
#pragma unroll
    for (int step = 0; step < (FM_TEXMERGE >= 2 ? 2 : 1); ++step) {
        // keeper = the lane of the pair with bit `step` clear.  The per-lane constants live in VGPRs (as 64-bit lane
        // masks they were SGPR spills, restored with v_readlane every visit); multiplying the partner's value by the
        // 0/1 weight lets the backend fuse the DPP read into one v_fmac_f32_dpp per channel.
        const bool keep = (lane & (1 << step)) == 0;
        const float keepf = keep ? 1.f : 0.f;
        const int dropm = keep ? 0 : -1;
        const int t = dpp_i(-1, tix, step);                     // partner's texel, -1 if it is not here
        const bool same = t == tix;
        const float w = same ? keepf : 0.f;
        a = fmaf(dpp_f(0.f, a, step), w, a);
        b = fmaf(dpp_f(0.f, b, step), w, b);
        c = fmaf(dpp_f(0.f, c, step), w, c);
        tix |= same ? dropm : 0;                                // merged into the partner: nothing left to add
    }
#endif
    if (tix >= 0) {
        atomicAdd(&my_tex[tix * 3], a);
        atomicAdd(&my_tex[tix * 3 + 1], b);
        atomicAdd(&my_tex[tix * 3 + 2], c);
    }
}
template <int RGB, bool NEED_GF, bool NEED_GT, bool COMMON>  // RGB 2 = silhouette only (soft_colors / grads are alpha planes)
// COMMON = the production case (gradient arrives 2x2-pooled, power-of-two image, double-sided faces) as compile-time
// facts: the wave-uniform flags otherwise live as 64-bit lane masks in SGPRs that spill (v_readlane per visit)
#ifndef BWD_WPE
#define BWD_WPE 7
#endif
#define BWD_WPE_ATTR __attribute__((amdgpu_waves_per_eu(BWD_WPE, BWD_WPE)))
__global__ __launch_bounds__(FM_WAVES * 64) BWD_WPE_ATTR void k_raster_backward_fm(const RasterArgs A) {
    extern __shared__ __attribute__((aligned(16))) float s_tex[];  // [FM_WAVES][FM_TEXCOPY][FM_TEX_STRIDE(TS)]
    constexpr bool SLOTS = FM_SLOTS == 2 || (FM_SLOTS == 1 && RGB == 2);
    // sub-tile hand-out: the lane that owns a wanted candidate of the culling pass writes the sub-tile's origin (pixel-centre
    // coordinates and byte offsets into the full / pooled planes) into slot [its rank among the wanted]; visit v hands slot
    // 4 v + g to lane group g with one 16-byte LDS read (a broadcast within the group) -- it replaced four rounds of
    // s_ff1 / v_readlane / v_cndmask and ~20 half-rate instructions of per-lane coordinate arithmetic per visit
    __shared__ float4 s_slot[SLOTS ? FM_WAVES : 1][SLOTS ? 64 : 1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: uniform
    const int F = A.F, IS = A.IS, TS = A.TS;
    const bool pooled = COMMON ? true : (A.grad_pooled != 0);
    const bool two_sided = COMMON ? true : (A.double_side != 0);
    // Wave-uniform float constants of the visit body, held in VGPRs on purpose: with the 32-float face record in SGPRs
    // the scalar file is full, and every constant the allocator spills comes back as a v_readlane (VALU) per visit.
    // As VALU operands they are as cheap from a VGPR.  (#define FM_VCONST 0 keeps them scalar for A/B.)
#if FM_VCONST
#define FM_V(x) ({ float v_; asm volatile("v_mov_b32 %0, %1" : "=v"(v_) : "s"(x)); v_; })
#else
#define FM_V(x) (x)
#endif
    const float c_near = FM_V(A.near_), c_far = FM_V(A.far_), c_rr = FM_V(A.r_range), c_ig = FM_V(A.inv_gamma);
    const float c_thr2 = FM_V(A.threshold), c_nis = FM_V(A.nis);
    // XCD-aware: hardware XCD = blockIdx % 8.  Each XCD owns a fixed contiguous EIGHTH of every mesh's faces
    // (index-neighbouring faces of a subdivided mesh are spatial neighbours), so the per-pixel state its waves
    // re-read (~6x) covers 1/8 of the screen and stays in that XCD's 4 MB L2, and all 8 XCDs share every mesh
    // (balance at small N).  Measured fabric reads: 47 MB/mesh round-robin -> ~20 MB/mesh (11.8 MB algorithmic).
    const int fblocks = (F + FM_WAVES - 1) / FM_WAVES;   // blocks per mesh (grid = N * fblocks)
    int nb = blockIdx.x / fblocks, fb = blockIdx.x % fblocks;
    if (fblocks % 8 == 0 && (A.N * fblocks) % 8 == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = fblocks >> 3;
        nb = slot / per;
        fb = __builtin_amdgcn_readfirstlane(fm_owned_face(xcd, slot % per, per, A.fm_split));   // (uniform; the division hides it)
    }
    int fidx = fb * FM_WAVES + wave;
    if (FM_WAVES == 1 && A.order) {   // cost-ordered start (k_face_order): same XCD ownership, heavy faces first
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, gsz = A.order_group * (F >> 3);
        const int g = slot / gsz;
        const int e = A.order[((size_t)g * 8 + xcd) * gsz + slot % gsz];
        nb = g * A.order_group + (e >> 16);
        fidx = e & 0xffff;
    }
    const bool live = fidx < F;
    const int n = nb, f = live ? fidx : 0;
    const size_t npix = (size_t)IS * IS;
    // wave-uniform bases of this mesh's per-pixel planes; every per-pixel load below is base + 32-bit byte offset
    const int H2 = IS >> 1;
    const unsigned pst = (unsigned)(npix * sizeof(float));                      // plane stride in bytes
    const unsigned gps = pooled ? (unsigned)((size_t)H2 * H2 * sizeof(float)) : pst;
    const int cplanes = RGB == 2 ? 1 : 4;
    const char *sc_n = (const char *)(A.soft_colors + (size_t)n * cplanes * npix);
    const char *ag_n = (const char *)(A.aggrs + (size_t)n * 2 * npix);
    const char *gc_n = (const char *)(A.grad_colors + (size_t)n * cplanes * (pooled ? (size_t)H2 * H2 : npix));
    float *wave_tex = s_tex + (size_t)wave * FM_TEXCOPY * FM_TEX_STRIDE(TS);
    // this lane's copy: horizontally and vertically adjacent pixels of a 4x4 / 8x8 tile get different copies
    float *my_tex = wave_tex + ((lane ^ (lane >> 2) ^ (lane >> 4)) & (FM_TEXCOPY - 1)) * FM_TEX_STRIDE(TS);
    if (NEED_GT && TS > 1)
        for (int j = lane; j < FM_TEXCOPY * FM_TEX_STRIDE(TS); j += 64) wave_tex[j] = 0.f;
    float gv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gt0 = 0.f, gt1 = 0.f, gt2 = 0.f;  // TS == 1 texel gradient
    bool visited = false;                   // wave-uniform: some sub-tile survived the culling pass
    if (live) {
        // VGPR-resident operands where the register budget of 7 waves / SIMD has room for them (silhouette and
        // texel-gradient-only variants: 56-60 VGPRs with them; the full variant would spill)
        constexpr bool VREC = FM_VREC != 0 && (RGB == 2 || !NEED_GF);
        typename std::conditional<VREC, FaceV, Face>::type fc;
        load_face(fc, A.rec + ((size_t)n * F + f) * REC);
        if constexpr (VREC) fc.fill();
        const float *__restrict__ tex_f = A.textures + ((size_t)(n / A.tex_group) * F + f) * TS * 3;
        // pixel-index window of the dilated bbox, widened by one pixel; the exact per-pixel reject of the
        // reference (:536) still runs inside eval_pair, so the window only has to be conservative.
        // xp(i) = (2i + 1 - IS)/IS  <=>  i = (xp*IS + IS - 1)/2
        const float h = 0.5f * IS;
        int x0 = (int)floorf(fc.template g<R_XLO>() * h + h - 0.5f) - 1, x1 = (int)ceilf(fc.template g<R_XHI>() * h + h - 0.5f) + 1;
        int yi0 = (int)floorf(fc.template g<R_YLO>() * h + h - 0.5f) - 1, yi1 = (int)ceilf(fc.template g<R_YHI>() * h + h - 0.5f) + 1;
        // NaN / inf bounds: comparisons below fail safe to the full image (the reference would visit all pixels)
        if (!(fc.template g<R_XLO>() == fc.template g<R_XLO>() && fc.template g<R_XHI>() == fc.template g<R_XHI>() && fc.template g<R_YLO>() == fc.template g<R_YLO>() && fc.template g<R_YHI>() == fc.template g<R_YHI>())) { x0 = 0; x1 = IS - 1; yi0 = 0; yi1 = IS - 1; }
        x0 = max(x0, 0); x1 = min(x1, IS - 1); yi0 = max(yi0, 0); yi1 = min(yi1, IS - 1);
        const int r0 = IS - 1 - yi1, r1 = IS - 1 - yi0;  // row = IS-1-yi
        if (x0 <= x1 && r0 <= r1) {
            // Sub-tiles of FM_TW x FM_TH pixels, FM_NQ = 64 / (FM_TW * FM_TH) of them per wave visit: the face is
            // wave-uniform here, so the 64 lanes need not form ONE tile -- each group of FM_TW*FM_TH lanes takes its own
            // needed sub-tile of this face.  4x4 sub-tiles fill 74 % of their lanes with contributing pixels against
            // 54 % for one 8x8 tile (CPU simulation of the culling, 1280-face sphere at IS = 512).
            const int tx0 = x0 / FM_TW, tx1 = x1 / FM_TW, ty0 = r0 / FM_TH, ty1 = r1 / FM_TH;
            const bool pow2 = COMMON ? true : ((IS & (IS - 1)) == 0);
            const float inv_is = 1.f / (float)IS;
            const int ntx = tx1 - tx0 + 1, ntiles = ntx * (ty1 - ty0 + 1);
            const float4 i0 = make_float4(fc.template g<R_INV + 0>(), fc.template g<R_INV + 1>(), fc.template g<R_INV + 2>(), fc.template g<R_INV + 3>());
            const float4 i1 = make_float4(fc.template g<R_INV + 4>(), fc.template g<R_INV + 5>(), fc.template g<R_INV + 6>(), fc.template g<R_INV + 7>());
            const float4 i2 = make_float4(fc.template g<R_INV + 8>(), fc.template g<R_K0>(), fc.template g<R_K1>(), fc.template g<R_K2>());
            const int sub = lane / (FM_TW * FM_TH), sl = lane % (FM_TW * FM_TH);   // sub-tile slot of this lane, lane in it
            // (slots) this lane's place inside a sub-tile, as the increments the slot's origin takes (exact: see the visit)
            const bool fastxy = pow2 && IS % FM_TW == 0 && IS % FM_TH == 0;   // exact incremental pixel centres, no ragged sub-tile
            const int lx = sl % FM_TW, ly = sl / FM_TW;
            const float lxf = (float)(2 * lx) * inv_is, lyf = (float)(2 * ly) * inv_is;
            const unsigned lo_pn = (unsigned)(ly * IS + lx) * 4u, lo_gp = pooled ? (unsigned)((ly >> 1) * H2 + (lx >> 1)) * 4u : lo_pn;
#ifdef FM_NO_CULL            // time-split experiment (tools/r3/split.sh): per-face set-up and reductions only
            for (int tb = ntiles; tb < ntiles; tb += 64) {
#else
            for (int tb = 0; tb < ntiles; tb += 64) {
#endif
                // one lane per sub-tile: drop those no pixel of which can survive (conservative), then walk the rest
                const int ti = tb + lane;
                bool want = false;
                int tpk = 0;   // packed (tx, ty) of this lane's candidate
                if (ti < ntiles) {
                    const int ttx = tx0 + ti % ntx, tty = ty0 + ti / ntx;
                    tpk = ttx | (tty << 16);
                    const int px0 = ttx * FM_TW, px1 = min(px0 + FM_TW - 1, IS - 1), pr0 = tty * FM_TH, pr1 = min(pr0 + FM_TH - 1, IS - 1);
                    const float cxl = ndc_coord_fast(px0, IS, inv_is, pow2), cxh = ndc_coord_fast(px1, IS, inv_is, pow2);
                    const float cyh = ndc_coord_fast(IS - 1 - pr0, IS, inv_is, pow2), cyl = ndc_coord_fast(IS - 1 - pr1, IS, inv_is, pow2);
                    want = tile_may_hit(i0, i1, i2, 0.5f * (cxl + cxh), 0.5f * (cyl + cyh), 0.5f * (cxh - cxl),
                                        0.5f * (cyh - cyl), A.thr);
#if FM_STATE_CULL
                    // Exact sub-tile skips from the saved forward state, decided HERE by the one lane that owns the
                    // candidate (64 candidates per pass) instead of by a whole wave visit that finds its four sub-tiles dead:
                    //  * silhouette: every pixel of the sub-tile has alpha == 1.0f -> g (1 - alpha) finite = 0 (:584);
                    //  * texel gradients only, soft-max: even the face's nearest depth is >= 89 gamma behind the soft-max
                    //    maximum of every pixel -> p = D exp(<-89) / S = 0.0f (:608); hard mode: the face wins no pixel (:596).
                    // Inside the silhouette that removes most of the back-facing half of the mesh before any visit.
                    if (FM_TW == 4 && FM_TH == 4 && (RGB == 2 || !NEED_GF) && want && px0 + 3 < IS && pr0 + 3 < IS && (IS & 3) == 0) {
                        const char *plane = RGB == 2 ? sc_n : ag_n + pst;           // alpha | soft-max maximum (hard: face id)
                        const unsigned o0 = (unsigned)(pr0 * IS + px0) * 4u, rs = (unsigned)IS * 4u;
                        const float4 q0 = ld_u4(plane, o0), q1 = ld_u4(plane, o0 + rs), q2 = ld_u4(plane, o0 + 2u * rs),
                                     q3 = ld_u4(plane, o0 + 3u * rs);
                        if (RGB == 2) {
                            want = !((q0.x == 1.f) & (q0.y == 1.f) & (q0.z == 1.f) & (q0.w == 1.f) & (q1.x == 1.f) & (q1.y == 1.f) &
                                     (q1.z == 1.f) & (q1.w == 1.f) & (q2.x == 1.f) & (q2.y == 1.f) & (q2.z == 1.f) & (q2.w == 1.f) &
                                     (q3.x == 1.f) & (q3.y == 1.f) & (q3.z == 1.f) & (q3.w == 1.f));
                        } else if (RGB == 1) {
                            const float mn = fminf(fminf(fminf(fminf(q0.x, q0.y), fminf(q0.z, q0.w)), fminf(fminf(q1.x, q1.y), fminf(q1.z, q1.w))),
                                                   fminf(fminf(fminf(q2.x, q2.y), fminf(q2.z, q2.w)), fminf(fminf(q3.x, q3.y), fminf(q3.z, q3.w))));
                            const float zmin_c = fminf(fminf(fc.template g<R_Z0>(), fc.template g<R_Z1>()), fc.template g<R_Z2>());
                            // same expression as the per-pixel test below; monotone in the maximum, so the sub-tile's smallest
                            // maximum decides for all 16 pixels.  fminf drops NaN operands, so a NaN in the saved state is
                            // looked for explicitly (the sum of the 16 values is NaN iff one of them is, or +inf - inf):
                            // such a sub-tile is visited, as the per-pixel test below would have it
                            const float sm = ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w)) +
                                             (((q2.x + q2.y) + (q2.z + q2.w)) + ((q3.x + q3.y) + (q3.z + q3.w)));
                            want = !(((c_far - zmin_c) * c_rr - mn) * c_ig < -89.f) || !(sm == sm);
                        } else {
                            const float ff = (float)f;
                            want = (q0.x == ff) | (q0.y == ff) | (q0.z == ff) | (q0.w == ff) | (q1.x == ff) | (q1.y == ff) | (q1.z == ff) |
                                   (q1.w == ff) | (q2.x == ff) | (q2.y == ff) | (q2.z == ff) | (q2.w == ff) | (q3.x == ff) | (q3.y == ff) |
                                   (q3.z == ff) | (q3.w == ff);
                        }
                    }
#endif
                }
                unsigned long long tm = __ballot(want);
                visited |= tm != 0;
#ifdef FM_NO_VISIT          // time-split experiment (tools/r3/split.sh): per-face set-up + culling pass only
                tm = 0;
#endif
                const int nv = __popcll(tm);
                float4 sd_next = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (SLOTS) {
                  if (want) {
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(tm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)tm, 0u));
                    const int px0 = (tpk & 0xffff) * FM_TW, pr0 = (tpk >> 16) * FM_TH;
                    const unsigned o_pn = (unsigned)(pr0 * IS + px0) * 4u;
                    const unsigned o_gp = pooled ? (unsigned)((pr0 >> 1) * H2 + (px0 >> 1)) * 4u : o_pn;
                    // power-of-two image: the origin's pixel-centre coordinates (exact floats); otherwise the packed tile index
                    // and the visit evaluates the fp64 expression per lane as before
                    s_slot[wave][rank] = make_float4(fastxy ? ndc_coord_fast(px0, IS, inv_is, true) : __int_as_float(tpk),
                                                     fastxy ? ndc_coord_fast(IS - 1 - pr0, IS, inv_is, true) : 0.f,
                                                     __int_as_float((int)o_pn), __int_as_float((int)o_gp));
                  }
                  // the slot of visit v + 1 is read at the top of visit v: LDS operations of a wave complete in order, so a read
                  // issued behind a visit's texel atomics (ds_add_f32, ~12 cycles a lane) would stall the next visit on them
                  sd_next = s_slot[wave][sub < nv ? sub : 0];
                }
                for (int v0 = 0; v0 < nv; v0 += FM_NQ) {
                    // next FM_NQ wanted sub-tiles, one per lane group (groups past the last one idle this visit)
                    int mine = -1;
                    float4 sd = sd_next;
                    if constexpr (SLOTS) {
                        mine = v0 + sub < nv ? v0 + sub : -1;
                        sd_next = s_slot[wave][v0 + FM_NQ + sub < nv ? v0 + FM_NQ + sub : 0];
                    } else {
#pragma unroll
                        for (int qq = 0; qq < FM_NQ; ++qq) {
                            if (tm) {
                                const int tbit = __builtin_ctzll(tm);
                                tm &= tm - 1;
                                const int e = __builtin_amdgcn_readlane(tpk, tbit);
                                if (sub == qq) mine = e;
                            }
                        }
                    }
                    if (FM_RELOAD_PER_TILE && NEED_GF && RGB != 2) {
                        // re-fetch the record from the scalar cache every visit: keeps the 32 constants loop-VARIANT so the
                        // compiler cannot hoist 30+ SGPR->VGPR copies out of the tile loop.  Only for the variants that
                        // also carry the 9 vertex-gradient accumulators and the colour path (measured 4-6 % faster with
                        // the re-fetch there, 1-5 % slower for the texel-only and silhouette kernels)
                        const float *rp = A.rec + ((size_t)n * F + f) * REC;
                        asm volatile("" : "+s"(rp));
                        load_face(fc, rp);
                    }
                    if (mine < 0) continue;
                    float xp, yp;
                    unsigned pn4, gp4;                                                          // byte offsets in a full / pooled plane
                    if constexpr (SLOTS) {
                      if (fastxy) {
                        // (2 (x0 + lx) + 1 - IS) / IS = origin + 2 lx / IS: every term and the sum are small integers over a power
                        // of two -- exact in fp32, the same bits as ndc_coord_fast per pixel (IS % 4 == 0: no ragged sub-tile)
                        xp = sd.x + lxf; yp = sd.y - lyf;
                      } else {
                        const int tp = __float_as_int(sd.x);
                        const int row = (tp >> 16) * FM_TH + ly, xi = (tp & 0xffff) * FM_TW + lx;
                        if (xi >= IS || row >= IS) continue;
                        yp = ndc_coord_fast(IS - 1 - row, IS, inv_is, pow2); xp = ndc_coord_fast(xi, IS, inv_is, pow2);
                      }
                      pn4 = (unsigned)__float_as_int(sd.z) + lo_pn;
                      gp4 = (unsigned)__float_as_int(sd.w) + lo_gp;
                    } else {
                        const int row = (mine >> 16) * FM_TH + sl / FM_TW;
                        const int xi = (mine & 0xffff) * FM_TW + sl % FM_TW;
                        if (xi >= IS || row >= IS) continue;
                        yp = ndc_coord_fast(IS - 1 - row, IS, inv_is, pow2);
                        xp = ndc_coord_fast(xi, IS, inv_is, pow2);
                        pn4 = (unsigned)(row * IS + xi) * 4u;
                        gp4 = pooled ? (unsigned)((row >> 1) * H2 + (xi >> 1)) * 4u : pn4;
                    }
                    // Exact tile skips from the saved forward state, before any geometry:
                    //  * alpha term: a pixel with alpha == 1.0f exactly contributes g*(1-alpha)*finite = 0 (:584);
                    //  * colour term: p = D*exp((zn - max)/gamma)/S (:608) is 0.0f when even the face's nearest depth
                    //    is >= 89 gamma behind the pixel's soft-max maximum (hard mode: the face is not the winner).
                    {
                        bool dead;
                        if (RGB == 2) {
                            dead = ld_u(sc_n, pn4) == 1.f;
                        } else {
                            dead = false;
                            if (!NEED_GF) {   // (with vertex gradients both terms must vanish: too rare to pay for)
                                const float smx = ld_u(ag_n, pn4 + pst);
                                const float zmin_f = fminf(fminf(fc.template g<R_Z0>(), fc.template g<R_Z1>()), fc.template g<R_Z2>());
                                dead = RGB == 0 ? (float)f != smx
                                                : ((c_far - zmin_f) * c_rr - smx) * c_ig < -89.f;
                            }
                        }
                        if ((RGB == 2 || !NEED_GF) && __all(dead)) continue;
                    }
                    Pair p;
                    if (!eval_pair(p, fc, xp, yp, c_thr2, c_nis)) continue;
                    if (RGB == 2) {  // silhouette: d alpha only (:584, :632-642); soft_colors/grad are [N,IS,IS] | [N,H,H]
                        if (!fc.depth_in_range()) {
                            float u0, u1, u2;
                            const float zq = clip_depth(u0, u1, u2, p, fc);
                            if (zq < c_near || zq > c_far) continue;  // :592
                        }
                        const float ga = (pooled ? 0.25f : 1.f) * ld_u(gc_n, gp4);
                        UMR_TRAP_IF(umr_bad(ga), 3);
                        const float oa = ld_u(sc_n, pn4);
                        float c_a = ga * ((1.f - oa) * __builtin_amdgcn_rcpf(fmaxf(1.f - p.frag, 1e-6f)));
                        c_a *= p.frag * (1.f - p.frag) * (-c_nis);
                        const float k2a = 2.f * p.sign * c_a;
                        const float a0 = k2a * p.b0, a1 = k2a * p.b1, a2 = k2a * p.b2;
                        FM_ACC(gv[0], a0, p.dx); FM_ACC(gv[1], a0, p.dy);
                        FM_ACC(gv[3], a1, p.dx); FM_ACC(gv[4], a1, p.dy);
                        FM_ACC(gv[6], a2, p.dx); FM_ACC(gv[7], a2, p.dy);
                        continue;
                    }
                    const float gscale = pooled ? 0.25f : 1.f;   // 2x2 mean pool: each fine pixel gets a quarter
                    const float g0 = gscale * ld_u(gc_n, gp4), g1 = gscale * ld_u(gc_n, gp4 + gps),
                                g2 = gscale * ld_u(gc_n, gp4 + 2 * gps);
                    const float g3 = NEED_GF ? gscale * ld_u(gc_n, gp4 + 3 * gps) : 0.f;
                    UMR_TRAP_IF(umr_bad(g0) | umr_bad(g1) | umr_bad(g2) | umr_bad(g3), 3);
                    const float ssum = ld_u(ag_n, pn4), smax = ld_u(ag_n, pn4 + pst);
                    float c_xy = 0.f;
                    if (NEED_GF) c_xy = g3 * ((1.f - ld_u(sc_n, pn4 + 3 * pst)) * __builtin_amdgcn_rcpf(fmaxf(1.f - p.frag, 1e-6f)));  // :584
                    float q0, q1, q2;
                    const float zp = clip_depth(q0, q1, q2, p, fc);
                    if (zp < c_near || zp > c_far) continue;  // :592
                    float gz0 = 0.f, gz1 = 0.f, gz2 = 0.f;
                    if (RGB == 0) {
                        if (NEED_GT && (float)f == smax) {  // :596
                            const int tix = texel_index(q0, q1, A.R);
                            if (TS == 1) { gt0 += g0; gt1 += g1; gt2 += g2; }
                            else texel_accumulate(my_tex, tix, g0, g1, g2, lane);
                        }
                    } else if (two_sided || fc.front()) {
                        const float zn = div_r(c_far - zp, c_far - c_near, c_rr);
                        const float ps = p.frag * __expf((zn - smax) * c_ig) * __builtin_amdgcn_rcpf(ssum);  // :608
                        const int tix = texel_index(q0, q1, A.R);
                        if (NEED_GT) {
                            if (TS == 1) { FM_ACC(gt0, ps, g0); FM_ACC(gt1, ps, g1); FM_ACC(gt2, ps, g2); }
                            else texel_accumulate(my_tex, tix, ps * g0, ps * g1, ps * g2, lane);
                        }
                        if (NEED_GF) {
                            const char *tx = (const char *)tex_f;
                            const unsigned t12 = (unsigned)tix * 12u;
                            float c_rgb = g0 * (ld_u(tx, t12) - ld_u(sc_n, pn4));
                            c_rgb += g1 * (ld_u(tx, t12 + 4) - ld_u(sc_n, pn4 + pst));
                            c_rgb += g2 * (ld_u(tx, t12 + 8) - ld_u(sc_n, pn4 + 2 * pst));
                            c_rgb *= ps;
                            c_xy += c_rgb * __builtin_amdgcn_rcpf(p.frag);
                            const float c_z = -(c_rgb * c_ig * c_rr) * zp * zp;  // :624
                            gz0 = c_z * q0 * fc.template g<R_RZ0>() * fc.template g<R_RZ0>();
                            gz1 = c_z * q1 * fc.template g<R_RZ1>() * fc.template g<R_RZ1>();
                            gz2 = c_z * q2 * fc.template g<R_RZ2>() * fc.template g<R_RZ2>();
                        }
                    }
                    if (NEED_GF) {
                        c_xy *= p.frag * (1.f - p.frag) * (-c_nis);  // :632
                        const float k2 = 2.f * p.sign * c_xy;        // :640
                        const float b0 = k2 * p.b0, b1 = k2 * p.b1, b2 = k2 * p.b2;
                        FM_ACC(gv[0], b0, p.dx); FM_ACC(gv[1], b0, p.dy); gv[2] += gz0;
                        FM_ACC(gv[3], b1, p.dx); FM_ACC(gv[4], b1, p.dy); gv[5] += gz1;
                        FM_ACC(gv[6], b2, p.dx); FM_ACC(gv[7], b2, p.dy); gv[8] += gz2;
                    }
                }
            }
        }
    }
#ifndef FM_SKIP_EMPTY
#define FM_SKIP_EMPTY 1   // a face none of whose sub-tiles survived the culling pass (half the mesh under a texel-gradient
#endif                    // launch) adds exact zeros: skip its lane reductions, LDS read-out and read-modify-write stores
    if (FM_SKIP_EMPTY && FM_WAVES == 1 && !visited) return;
    if (NEED_GF) {
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float sv = wave_sum_full(gv[k]);
            if (lane == k) mine = sv;
        }
        UMR_TRAP_IF(umr_bad(mine), 4);
        if (live && lane < 9) A.grad_faces[((size_t)n * F + f) * 9 + lane] += mine;
    }
    if (NEED_GT) {
        if (TS == 1) {
            const float s0 = wave_sum_full(gt0), s1 = wave_sum_full(gt1), s2 = wave_sum_full(gt2);
            if (live && lane < 3) A.grad_textures[((size_t)n * F + f) * 3 + lane] += lane == 0 ? s0 : (lane == 1 ? s1 : s2);
        } else {
            __syncthreads();  // every wave arrives exactly once; orders the LDS atomics before the read-out
            if (live) {
                float *dst = A.grad_textures + ((size_t)n * F + f) * TS * 3;
                for (int j = lane; j < TS * 3; j += 64) {
                    float acc = wave_tex[j];
#pragma unroll
                    for (int c = 1; c < FM_TEXCOPY; ++c) acc += wave_tex[c * FM_TEX_STRIDE(TS) + j];
                    UMR_TRAP_IF(umr_bad(acc), 4);
                    dst[j] += acc;
                }
            }
        }
    }
}

template <int RGB, bool COMMON>
void launch_backward_fm2(const RasterArgs &A, hipStream_t st) {
    const int blocks = A.N * ((A.F + FM_WAVES - 1) / FM_WAVES);
    const size_t lds = (A.need_gt && A.TS > 1) ? (size_t)FM_WAVES * FM_TEXCOPY * FM_TEX_STRIDE(A.TS) * sizeof(float) : 0;
    if (RGB == 2) k_raster_backward_fm<2, true, false, COMMON><<<blocks, FM_WAVES * 64, 0, st>>>(A);
    else if (A.need_gf && A.need_gt) k_raster_backward_fm<RGB, true, true, COMMON><<<blocks, FM_WAVES * 64, lds, st>>>(A);
    else if (A.need_gf) k_raster_backward_fm<RGB, true, false, COMMON><<<blocks, FM_WAVES * 64, lds, st>>>(A);
    else k_raster_backward_fm<RGB, false, true, COMMON><<<blocks, FM_WAVES * 64, lds, st>>>(A);
}
template <int RGB>
void launch_backward_fm(const RasterArgs &A, hipStream_t st) {
    const bool common = A.grad_pooled && A.double_side && (A.IS & (A.IS - 1)) == 0;
    if (common) launch_backward_fm2<RGB, true>(A, st);
    else launch_backward_fm2<RGB, false>(A, st);
}

// Mode ids of the reference binding (functional/soft_rasterize.py:21-24).  *general = anything but the combination UMR
// instantiates (euclidean distance, 'prod' alpha, surface textures), which has the specialised kernels.
bool modes_ok(int func_id_dist, int func_id_rgb, int func_id_alpha, int texture_sample_type, int TS, int *R, bool *general) {
    if (func_id_dist < 0 || func_id_dist > 2 || func_id_alpha < 0 || func_id_alpha > 2) return false;
    if (func_id_rgb != 0 && func_id_rgb != 1) return false;
    if (texture_sample_type != 0 && texture_sample_type != 1) return false;
    *general = !(func_id_dist == 2 && func_id_alpha == 2 && texture_sample_type == 0);
    if (texture_sample_type == 1) {   // vertex colours: the reference reads w[j] for j < texture_size (:215)
        *R = 1;
        return TS == 3;
    }
    int r = 1;
    while (r * r < TS) ++r;
    if (r * r != TS) return false;
    *R = r;
    return true;
}

}  // namespace
