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
                    const float4 *q = (const float4 *)(rec_n + (size_t)fcand * REC + R_I0);
                    hit = tile_may_hit(q[0], q[1], q[2], 0.5f * (t.wxlo + t.wxhi), 0.5f * (t.wylo + t.wyhi),
                                       0.5f * (t.wxhi - t.wxlo), 0.5f * (t.wyhi - t.wylo), A.thr + rec_n[(size_t)fcand * REC + R_CULL]);
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
                if (eval_pair(p, fc, t.xp, t.yp, A.threshold, A.nis, A.amb_thr, t.valid) && t.valid) {
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
                            const float ex_ = (zn - smax) * A.inv_gamma;
                            const float ps = p.frag * __expf(ex_ > 0.f ? 0.f : ex_) * __builtin_amdgcn_rcpf(ssum);  // :608 (NaN-preserving clamp: raster_backward_fm.h)
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
// Work items of the face-major backward and their start order.  A wave owns one item for its whole life and faces differ
// enormously in work: on a regular mesh a front face under a texel-gradient launch walks ~13 visits of ~900 issue cycles while
// a back face is culled after one pass (50x), and the meshes a training step really renders (profiles/scenes/live_s1_*.npz:
// step ~57 of a GAN-driven trajectory) carry a few faces of 1000 - 1500 candidate sub-tiles against a median of 63 -- ONE such
// wave then runs for the whole launch on an otherwise empty chip (one-pass backward: 212 - 273 us on the live scenes against
// 137 us on the regular SURVEY 8d scene, with fewer instructions issued).  This kernel, per XCD (each XCD keeps its contiguous
// share of every mesh's faces -- the L2 locality of the saved state) and per group of `G` meshes,
//   1. estimates every face's work: 4x4 sub-tiles under the dilated bbox; / 8 for faces the state cull will remove -- at their
//      corners and centroid alpha == 1 (mode 2: silhouette) or, for back faces, a nearer face's soft-max weight (mode 1: colour);
//   2. SPLITS a face whose estimate exceeds T into parts = ceil(estimate / T) items (<= SPLIT_MAX_PARTS, <= its culling passes),
//      each a contiguous share of the face's culling passes.  T = the smallest of T0 x {1, 1.25, 1.5, 2, 3, 4, 8, 16} whose extra
//      items fit the list's budget X (so the launch grid and the slab pool have fixed sizes);
//   3. sorts the ITEMS by descending work (counting sort, 1024 buckets; a part's key = the face's estimate / parts), so the
//      heavy items of all meshes of the group start first and the launch drains on cheap ones (`sorted` 0, an A/B switch, keeps
//      the index order with the parts of a split face side by side); the list is padded with 0xffffffff to its fixed length;
//   4. lists the split faces (split[list][0] = their number, then {item word, first slab} each) for k_split_reduce, which runs
//      after the main kernel and adds each split face's partial sums in part order.
// Neither order nor split changes WHICH pairs contribute, and a split face's sum is formed in a fixed order: results do not
// depend on scheduling.
#define ORDER_KEYS 1024
#define ORDER_MAX_ENTRIES 16384
#define ORDER_THREADS 1024
#define SPLIT_MAX_PARTS 32
#define SPLIT_LEVELS 8          // thresholds tried: T0 x {1, 1.25, 1.5, 2, 3, 4, 8, 16}, then "no split"
__device__ __forceinline__ int split_threshold(int T0, int k) {
    const int num[SPLIT_LEVELS] = {4, 5, 6, 8, 12, 16, 32, 64};
    return max((max(T0, 1) * num[k]) >> 2, 1);
}
// parts of a face of estimate nt at threshold T: ceil(nt / T) in [1, lim], in float arithmetic (exact for the quotients <= 32 that
// matter; an integer division is ~40 instructions and the level search makes eight per face).  Every pass of k_face_order calls
// THIS function, so the item count and the lists agree.
__device__ __forceinline__ int split_parts(int nt, int T, int lim) {
    return min(max((int)ceilf((float)nt / (float)T), 1), lim);
}
#define SPLIT_UNIT 128          // floats per slab unit (a part of the BASELINE variants -- 9 + 3 x 36 sums -- takes one)
#ifndef SPLIT_T0
#define SPLIT_T0 192            // estimated work (sub-tiles) one item may carry before its face is split: 64 ... 192 time the same on the
#endif                          // frozen training scenes (107 / 108 us), 256 costs 8 - 12 %; the larger, the fewer faces of a regular mesh split
struct OrderArgs {
    const float4 *bbox; const float *rec; uint2 *order; uint2 *split;   // split: [lists][X + 1]
    unsigned *est;                            // [N * F] k_face_estimate -> k_face_order
    const float *alpha;                       // silhouette launch: the alpha plane [N, IS, IS]
    const float *state, *aggrs;               // colour launches: the packed saved state, or aggrs_info [N, 2, IS, IS]
    float far_, r_range, inv_gamma;
    int N, F, IS, G, mode, sorted, run_split, rotate, stride, X, X_alloc, slabs_per_list, units_per_part, T0;   // X <= X_alloc: this launch's budget of extra items
};
// Step 1 of k_face_order's list, one thread per (mesh, face) over the whole chip (inside k_face_order -- one 1 024-thread workgroup
// per XCD -- it was 20 of that kernel's 29 us at N = 16): est[n * F + f] = estimate (15 bits) | start order at an eighth << 15 |
// most parts the face can have << 16.
__global__ __launch_bounds__(256) void k_face_estimate(const OrderArgs O) {
    const int i = blockIdx.x * 256 + (int)threadIdx.x, F = O.F, IS = O.IS;
    if (i >= O.N * F) return;
    const size_t fi = (size_t)i;
    const size_t n_ = fi / F;
    const float h = 0.5f * IS;
    // work estimate: 4x4 sub-tiles under the dilated bbox (the window the wave will walk; NaN bounds: the whole image)
    const float4 bb = O.bbox[fi];
    int raw = 0x7fff;
    if (bb.x == bb.x && bb.y == bb.y && bb.z == bb.z && bb.w == bb.w) {
        const int px0 = max((int)floorf(bb.x * h + h - 0.5f) - 1, 0), px1 = min((int)ceilf(bb.y * h + h - 0.5f) + 1, IS - 1);
        const int py0 = max((int)floorf(bb.z * h + h - 0.5f) - 1, 0), py1 = min((int)ceilf(bb.w * h + h - 0.5f) + 1, IS - 1);
        raw = (px0 <= px1 && py0 <= py1) ? min(((px1 >> 2) - (px0 >> 2) + 1) * ((py1 >> 2) - (py0 >> 2) + 1), 0x7fff) : 0;
    }
    const unsigned c = (__float_as_int(O.rec[fi * REC + R_FLAGS]) & 32) ? 0x8000u : 0u;      // front-facing (k_face_setup)
    int nt = raw;
    bool order_eighth = false;
    // Faces the state cull will (mostly) remove, judged at the face's corners and centroid with the kernels' own tests:
    //  * silhouette launch (mode 2): alpha == 1.0f at all five -- g (1 - alpha) = 0 there;
    //  * colour launches (mode 1), back faces only: the face is depth-dead at all five -- even its nearest depth lies >= 89
    //    gamma behind the pixel's soft-max maximum, i.e. a nearer face covers it and its weight is 0.0f.
    // A back face that sticks OUT of the silhouette (a spike of a half-trained mesh) is its own nearest surface: it is
    // walked in full.  (Round 5 discounted every back face; on the captured training scenes such faces were the launch's tail.)
    if (nt > 0 && (O.mode == 2 || (O.mode == 1 && !(c & 0x8000u))) && (O.alpha || O.state || O.aggrs)) {
        const float *r = O.rec + fi * REC;
        const float cx[4] = {r[R_X0], r[R_X1], r[R_X2], (r[R_X0] + r[R_X1] + r[R_X2]) * (1.f / 3.f)};
        const float cy[4] = {r[R_Y0], r[R_Y1], r[R_Y2], (r[R_Y0] + r[R_Y1] + r[R_Y2]) * (1.f / 3.f)};
        const float zq = (O.far_ - fminf(fminf(r[R_Z0], r[R_Z1]), r[R_Z2])) * O.r_range;
        bool removed = true;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int xi = min(max((int)(cx[c4] * h + h), 0), IS - 1), row = IS - 1 - min(max((int)(cy[c4] * h + h), 0), IS - 1);
            const size_t pn = (size_t)row * IS + xi;
            if (O.mode == 2) removed &= O.alpha[n_ * IS * IS + pn] == 1.f;
            else {
                // the pixel's soft-max maximum: packed state, [16 + 4 (y & 3) + (x & 3)] of its 4x4 tile's record, or plane 1 of aggrs_info
                const float smax = O.state ? O.state[(n_ * (IS >> 2) * (IS >> 2) + (size_t)(row >> 2) * (IS >> 2) + (xi >> 2)) * STATE_REC +
                                                     STATE_O_MAX + (row & 3) * 4 + (xi & 3)]
                                           : O.aggrs[(n_ * 2 + 1) * IS * IS + pn];
                removed &= (zq - smax) * O.inv_gamma < -89.f;
            }
        }
        if (removed) nt >>= 3;
        // (a back face that is NOT removed at all five probes: split by its full estimate -- it may be a spike --, but started
        // where round 5's rule puts it, an eighth: on regular meshes such faces sit near the silhouette's rim and the quad-level
        // state cull still removes most of them; measured on the SURVEY 8d scene, one-pass backward: 142 -> 137 us)
        else if (O.mode == 1) order_eighth = true;
    }
    // a face cannot have more parts than culling passes (64 candidate sub-tiles each)
    const int lim = max(1, min(SPLIT_MAX_PARTS, (raw + 63) >> 6));
    O.est[fi] = (unsigned)nt | (order_eighth ? 0x8000u : 0u) | ((unsigned)lim << 16);
}

__global__ __launch_bounds__(ORDER_THREADS) void k_face_order(const OrderArgs O) {
    __shared__ int s_hist[ORDER_KEYS];
    __shared__ int s_wsum[ORDER_THREADS / 64];
    __shared__ unsigned short s_key[ORDER_MAX_ENTRIES];
    __shared__ unsigned char s_np[ORDER_MAX_ENTRIES];
    __shared__ int s_ex[SPLIT_LEVELS];
    __shared__ int s_slab, s_level, s_nsplit;
    const int xcd = blockIdx.x, g = blockIdx.y, F = O.F, per = F >> 3;
    const int m0 = g * O.G, gl = min(O.G, O.N - m0), E = gl * per;
    for (int k = threadIdx.x; k < ORDER_KEYS; k += ORDER_THREADS) s_hist[k] = 0;
    if (threadIdx.x < SPLIT_LEVELS) s_ex[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_slab = 0; s_nsplit = 0; }
    __syncthreads();
    int ex[SPLIT_LEVELS];
#pragma unroll
    for (int k = 0; k < SPLIT_LEVELS; ++k) ex[k] = 0;
    for (int e = threadIdx.x; e < E; e += ORDER_THREADS) {
        const int ml = e / per, f = fm_owned_face((xcd + (m0 + ml) * O.rotate) & 7, e % per, per, O.run_split);
        const unsigned v = O.est[(size_t)(m0 + ml) * F + f];      // k_face_estimate: estimate (15 bits) | start at an eighth << 15 | most parts << 16
        const int nt = (int)(v & 0x7fffu), lim = (int)(v >> 16);
        s_key[e] = (unsigned short)(v & 0xffffu);
        s_np[e] = (unsigned char)lim;
#pragma unroll
        for (int k = 0; k < SPLIT_LEVELS; ++k) ex[k] += split_parts(nt, split_threshold(O.T0, k), lim) - 1;
    }
#pragma unroll
    for (int k = 0; k < SPLIT_LEVELS; ++k) {
        int v = ex[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_ex[k], v);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int k = 0;
        while (k < SPLIT_LEVELS && s_ex[k] > O.X) ++k;      // SPLIT_LEVELS: nothing fits -- one item per face
        s_level = O.T0 > 0 ? k : SPLIT_LEVELS;
    }
    __syncthreads();
    const int level = s_level;
    for (int e = threadIdx.x; e < E; e += ORDER_THREADS) {
        const int nt = s_key[e] & 0x7fff, lim = s_np[e];
        int np = 1;
        if (level < SPLIT_LEVELS) np = split_parts(nt, split_threshold(O.T0, level), lim);
        const int nt_order = (s_key[e] & 0x8000) ? nt >> 3 : nt;
        // sorted: heavy items first (a part's key = its share of the face's estimate); else INDEX order, bucket by bucket (A/B)
        const int key = O.sorted ? min((nt_order + np - 1) / np, ORDER_KEYS - 1) : ORDER_KEYS - 1 - (int)(((long long)e * ORDER_KEYS) / E);
        s_key[e] = (unsigned short)key;
        s_np[e] = (unsigned char)np;
        atomicAdd(&s_hist[key], np);
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
    const size_t list = (size_t)g * 8 + xcd;
    uint2 *out = O.order + list * (size_t)O.stride;
    const unsigned slab0 = (unsigned)(list * (size_t)O.slabs_per_list);
    uint2 *split_list = O.split + list * (size_t)(O.X_alloc + 1);
    for (int e = threadIdx.x; e < E; e += ORDER_THREADS) {
        const int np = s_np[e];
        const int pos = atomicAdd(&s_hist[s_key[e]], np);
        const unsigned code = ((unsigned)(np - 1) << 21) | ((unsigned)(e / per) << 16) | (unsigned)fm_owned_face((xcd + (m0 + e / per) * O.rotate) & 7, e % per, per, O.run_split);
        unsigned slab = 0;
        if (np > 1) {
            slab = slab0 + (unsigned)atomicAdd(&s_slab, np);
            split_list[1 + atomicAdd(&s_nsplit, 1)] = make_uint2(code, slab);
        }
        for (int p = 0; p < np; ++p) out[pos + p] = make_uint2(code | ((unsigned)p << 26), slab);
    }
    const int items = E + (level < SPLIT_LEVELS ? s_ex[level] : 0);
    for (int i = items + (int)threadIdx.x; i < O.stride; i += ORDER_THREADS) out[i] = make_uint2(0xffffffffu, 0u);
    __syncthreads();
    if (threadIdx.x == 0) split_list[0] = make_uint2((unsigned)s_nsplit, 0u);
}

// Second half of a split face (k_face_order): one wave per split face adds the partial sums its items left in their slabs, in
// part order, and adds the total to the face's gradient -- the launch boundary is what orders the items' stores before these
// loads.  ONE workgroup of SPLIT_REDUCE_WAVES waves per list; wave w takes entries w, w + SPLIT_REDUCE_WAVES, ...  (What the shape
// costs, measured: 32 one-wave workgroups per list 12 us whether or not anything was split -- 0.8 ms per train_s2 step with its
// 11 x 128 lists; one 4-wave workgroup per list 3 us empty but 44 us on a captured training scene with ~100 split faces per list:
// a face is a chain of ~5 dependent memory round trips.)
#define SPLIT_REDUCE_WAVES 16
struct SplitReduceArgs {
    const uint2 *split; const float *slab; float *grad_faces; float *grad_textures;
    int F, TS, G, X, slab_stride, need_gf, need_gt;
};
__global__ __launch_bounds__(SPLIT_REDUCE_WAVES * 64) void k_split_reduce(const SplitReduceArgs R) {
    const int list = blockIdx.x, g = list >> 3, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint2 *sp = R.split + (size_t)list * (R.X + 1);
    const int count = (int)sp[0].x;
    for (int i = wave; i < count; i += SPLIT_REDUCE_WAVES) {
        const uint2 e = sp[1 + i];
        const int nparts = (int)((e.x >> 21) & 31u) + 1, f = (int)(e.x & 0xffffu), n = g * R.G + (int)((e.x >> 16) & 31u);
        const float *s0 = R.slab + (size_t)e.y * R.slab_stride;
        if (R.need_gf && lane < 9) {
            float acc = 0.f;
            for (int q = 0; q < nparts; ++q) acc += s0[(size_t)q * R.slab_stride + lane];
            UMR_TRAP_AT(umr_bad(acc), 4, ((unsigned)(n & 127) << 16) | ((unsigned)f & 0xffffu));
            R.grad_faces[((size_t)n * R.F + f) * 9 + lane] += acc;
        }
        if (R.need_gt) {
            float *dst = R.grad_textures + ((size_t)n * R.F + f) * R.TS * 3;
            for (int j = lane; j < R.TS * 3; j += 64) {
                float acc = 0.f;
                for (int q = 0; q < nparts; ++q) acc += s0[(size_t)q * R.slab_stride + 16 + j];
                UMR_TRAP_AT(umr_bad(acc), 5, ((unsigned)(n & 127) << 16) | ((unsigned)f & 0xffffu));
                dst[j] += acc;
            }
        }
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
                       // (a third step across quads -- row_shl:4 -- measured slower in round 5 and was removed)
#endif
#if FM_TEXMERGE > 2
#error "FM_TEXMERGE: 0, 1 or 2"
#endif
#ifndef FM_VREC
#define FM_VREC 1         // 1: the face's barycentric rows and corner coordinates as VGPR operands (FaceV)
#endif
#ifndef FM_FMA_ACC
#define FM_FMA_ACC 1      // 1: gradient accumulators updated with explicit fused multiply-adds (the summation order of the
#endif                    // reference's atomics is not defined either, so these sums are not pinned to a rounding sequence)
#if FM_FMA_ACC
#define FM_ACC(acc, a, b) acc = fmaf(a, b, acc)
#else
#define FM_ACC(acc, a, b) acc += (a) * (b)
#endif
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
#ifdef UMR_HOST_SHIM   // tests/host_kernel/wave_emu.h: these DPP reads run under lane divergence (only the lanes that
#define UMR_DPP_DIVERGENT emu_dpp_subset           // contribute to the face in this visit reach them)
#else
#define UMR_DPP_DIVERGENT __builtin_amdgcn_update_dpp
#endif
__device__ __forceinline__ float dpp_f(float old, float v, const int ctrl_sel) {
    // ctrl_sel: 0 -> quad_perm [1,0,3,2] (x^1), 1 -> quad_perm [2,3,0,1] (x^2)
    const int o = __float_as_int(old), i = __float_as_int(v);
    int r;
    if (ctrl_sel == 0) r = UMR_DPP_DIVERGENT(o, i, 0xB1, 0xf, 0xf, false);
    else r = UMR_DPP_DIVERGENT(o, i, 0x4E, 0xf, 0xf, false);
    return __int_as_float(r);
}
__device__ __forceinline__ int dpp_i(int old, int v, const int ctrl_sel) {
    if (ctrl_sel == 0) return UMR_DPP_DIVERGENT(old, v, 0xB1, 0xf, 0xf, false);
    return UMR_DPP_DIVERGENT(old, v, 0x4E, 0xf, 0xf, false);
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
#ifdef FM_NO_TEXACC      // time-split experiment: no LDS accumulation (results wrong)
    if (tix == -12345) {
#else
    if (tix >= 0) {
#endif
        atomicAdd(&my_tex[tix * 3], a);
        atomicAdd(&my_tex[tix * 3 + 1], b);
        atomicAdd(&my_tex[tix * 3 + 2], c);
    }
}
#define FM_ALPHA_GEOM 0
#define FM_PACKED 0
#define FM_QUADS 0
#ifndef FM_AG_VREC
#define FM_AG_VREC 0
#endif
#ifndef FM_AGP_SLOT16
#define FM_AGP_SLOT16 1   // packed-state one-pass kernel (COMMON): 16-byte quad slots {x, y, state offset, gradient offset} as floats
#endif
#ifndef FM_DEAD_EAGER
#define FM_DEAD_EAGER 1   // packed-state one-pass kernel: the visit's two dead-test words are loaded together
#endif
#define FM_KERNEL_NAME k_raster_backward_fm
#include "raster_backward_fm.h"
#undef FM_KERNEL_NAME
#undef BWD_WPE              // (BWD_WPE_ATTR expands it at the kernel's declaration)
#define BWD_WPE 6            // vertex + texel variant: 84 VGPRs and no SGPR spill traffic in the visit (213 -> 201 us at N = 16)
#define FM_KERNEL_NAME k_raster_backward_fm_w6
#include "raster_backward_fm.h"
#undef FM_KERNEL_NAME
#undef BWD_WPE
#define BWD_WPE 7
#undef FM_QUADS
#define FM_QUADS 1
#define FM_KERNEL_NAME k_raster_backward_fm_quads
#include "raster_backward_fm.h"
#undef FM_KERNEL_NAME
#undef FM_ALPHA_GEOM
#define FM_ALPHA_GEOM 1
#ifndef FM_AG_WPE
#define FM_AG_WPE 6       // waves / SIMD of the one-pass kernels: 6 leaves no VGPR spill and half the SGPR spills (planar state: 167.6 -> 158.6 us on the fixed
                          // scene; packed state 142 either way -- its 6.2 KB of LDS per wave admit 25 waves per CU anyway); 5: 152, 8: 160
#endif
#ifndef FM_AG_VREC
#define FM_AG_VREC 0      // 1: VGPR copies of the barycentric rows / corners in the one-pass kernels as well (spills at 7 waves)
#endif
#undef BWD_WPE
#define BWD_WPE FM_AG_WPE
#define FM_KERNEL_NAME k_raster_backward_fm_ag
#include "raster_backward_fm.h"
#undef FM_KERNEL_NAME
#undef FM_PACKED
#define FM_PACKED 1
#define FM_KERNEL_NAME k_raster_backward_fm_agp
#include "raster_backward_fm.h"
#undef FM_KERNEL_NAME
#undef FM_PACKED
#undef FM_ALPHA_GEOM
#undef FM_QUADS
#undef BWD_WPE
#define BWD_WPE 7
#ifndef FM_QUADS_MASK
#define FM_QUADS_MASK 1   // which variants take the quad hand-out: bit 0 silhouette, bit 1 texel-gradient-only, bit 2 vertex + texel
#endif                    // (measured: header of raster_backward_fm.h)
#ifndef FM_FULL_W6
#define FM_FULL_W6 1      // vertex + texel variant at 6 waves / SIMD
#endif

template <int RGB, bool COMMON>
void launch_backward_fm2(const RasterArgs &A, hipStream_t st) {
    const int blocks = A.fm_blocks;
    const size_t lds = (A.need_gt && A.TS > 1) ? (size_t)FM_WAVES * FM_TEXCOPY * FM_TEX_STRIDE(A.TS) * sizeof(float) : 0;
    if constexpr (RGB == 2) {
        if constexpr ((FM_QUADS_MASK & 1) != 0) UMR_LAUNCH((k_raster_backward_fm_quads<2, true, false, COMMON>), blocks, FM_WAVES * 64, 0, st, A);
        else UMR_LAUNCH((k_raster_backward_fm<2, true, false, COMMON>), blocks, FM_WAVES * 64, 0, st, A);
    } else if (A.need_gf && A.need_gt) {
        if constexpr ((FM_QUADS_MASK & 4) != 0) UMR_LAUNCH((k_raster_backward_fm_quads<RGB, true, true, COMMON>), blocks, FM_WAVES * 64, lds, st, A);
        else if constexpr (FM_FULL_W6 != 0) UMR_LAUNCH((k_raster_backward_fm_w6<RGB, true, true, COMMON>), blocks, FM_WAVES * 64, lds, st, A);
        else UMR_LAUNCH((k_raster_backward_fm<RGB, true, true, COMMON>), blocks, FM_WAVES * 64, lds, st, A);
    } else if (A.need_gf) {
        UMR_LAUNCH((k_raster_backward_fm<RGB, true, false, COMMON>), blocks, FM_WAVES * 64, lds, st, A);
    } else {
        if constexpr ((FM_QUADS_MASK & 2) != 0) UMR_LAUNCH((k_raster_backward_fm_quads<RGB, false, true, COMMON>), blocks, FM_WAVES * 64, lds, st, A);
        else UMR_LAUNCH((k_raster_backward_fm<RGB, false, true, COMMON>), blocks, FM_WAVES * 64, lds, st, A);
    }
}
// d alpha -> vertices and d rgb -> texels of a soft-max render in one pass (UMR_BWD_ALPHA_GEOMETRY)
void launch_backward_fm_ag(const RasterArgs &A, hipStream_t st) {
    const int blocks = A.fm_blocks;
    const size_t lds = A.TS > 1 ? (size_t)FM_WAVES * FM_TEXCOPY * FM_TEX_STRIDE(A.TS) * sizeof(float) : 0;
    const bool common = A.grad_pooled && A.double_side && (A.IS & (A.IS - 1)) == 0;
    if (A.state) {    // packed saved state (UMR_BWD_PACKED_STATE)
        if (common) UMR_LAUNCH((k_raster_backward_fm_agp<1, true, true, true>), blocks, FM_WAVES * 64, lds, st, A);
        else UMR_LAUNCH((k_raster_backward_fm_agp<1, true, true, false>), blocks, FM_WAVES * 64, lds, st, A);
    } else if (common) UMR_LAUNCH((k_raster_backward_fm_ag<1, true, true, true>), blocks, FM_WAVES * 64, lds, st, A);
    else UMR_LAUNCH((k_raster_backward_fm_ag<1, true, true, false>), blocks, FM_WAVES * 64, lds, st, A);
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
