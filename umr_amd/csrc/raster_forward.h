// raster_forward.h -- k_raster_forward: pixel-major forward (included by raster.hip only).
#pragma once
#include "raster_core.h"

namespace {

// ------------------------------------------------------------------------------------------------
template <int RGB, bool P2F, bool TWO_SIDED, bool VIS = false>  // VIS (RGB == 1): also the hard z-buffer planes.  0 = hard z-buffer colour (:408-416), 1 = soft-max over depth (:417-437),
                    // 2 = silhouette only: alpha plane, no depth / colour / p2f (soft_colors is then [N,IS,IS]),
                    // 3 = visibility only: the hard z-buffer's (depth, face id) planes, nothing else
// Register budget for 7 waves per SIMD: the default allocation (106 SGPRs) admits 6; the kernels are VALU-issue bound
// with every wave stalled ~50 % of its life, so the seventh wave pays (measured: 5 < 6 < 7 ~ 8 waves, -3..6 % time).
// (the forward variants without p2f accumulators fit 8 waves and gain another 2-5 %; with p2f 8 is slower)
#ifndef FWD_WPE_ATTR
#define FWD_WPE_ATTR __attribute__((amdgpu_waves_per_eu(P2F ? 7 : 8, P2F ? 7 : 8)))
#endif
__global__ __launch_bounds__(BLK_THREADS) FWD_WPE_ATTR void k_raster_forward(const RasterArgs A) {
    __shared__ int s_list[LIST_CAP];
    __shared__ int s_wcnt[BLK_THREADS / 64];
    Tile t;
    tile_setup(t, A);
    const int F = A.F, IS = A.IS;
    const size_t npix = (size_t)IS * IS;
    const size_t pn = (size_t)t.row * IS + t.xi;
    const float4 *__restrict__ bbox_n = A.bbox + (size_t)t.n * F;
    const float *__restrict__ rec_n = A.rec + (size_t)t.n * F * REC;
    const float *__restrict__ tex_n = A.textures + (size_t)(t.n / A.tex_group) * F * A.TS * 3;

    float alpha = 1.f;
    float ssum = __expf(A.eps / A.gamma), smax = A.eps;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, gx = 0.f, gy = 0.f;
    float depth_min = 10000000.f;
    int face_min = -1;
    if (t.valid && RGB < 2) {
        if (A.bg_arg) { c0 = A.bg0; c1 = A.bg1; c2 = A.bg2; }
        else {
            const float *sc = A.soft_colors + (size_t)t.n * 4 * npix + pn;
            c0 = sc[0]; c1 = sc[npix]; c2 = sc[2 * npix];
        }
        if (RGB == 1) {
            c0 *= ssum; c1 *= ssum; c2 *= ssum;
            if (P2F) { gx = A.grid[pn * 2]; gy = A.grid[pn * 2 + 1]; }
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
#ifdef FWD_NO_VISIT         // time-split experiment (HISTORY.md 4.1): binning and tile filter only
            m = 0;
#endif
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const int f = __builtin_amdgcn_readlane(fcand, b);
                Face fc;
                load_face(fc, rec_n + (size_t)f * REC);
                float wgt = 0.f;  // this lane's p2f weight for face f
                if (RGB == 3) {
                    // z-buffer winner only (:408-411): needs the bbox test, the barycentrics, the depth -- no distance.
                    // A pixel inside [0,1]^3 is never rejected by the distance threshold (inside: sign > 0; on the
                    // boundary: d = 0), except the defined-as-skip k = -1 case (no w <= 0 yet some w >= 1).
                    float w0, w1, w2;
                    fc.bary(w0, w1, w2, t.xp, t.yp);
                    const bool inb = !((t.xp > fc.g<R_XHI>()) | (t.xp < fc.g<R_XLO>()) | (t.yp > fc.g<R_YHI>()) | (t.yp < fc.g<R_YLO>()));
                    const bool incl = (w0 <= 1) & (w0 >= 0) & (w1 <= 1) & (w1 >= 0) & (w2 <= 1) & (w2 >= 0);
                    const bool strict = (w0 > 0) & (w1 > 0) & (w2 > 0) & (w0 < 1) & (w1 < 1) & (w2 < 1);
                    const bool cand = inb & incl & t.valid & (strict | (w0 <= 0) | (w1 <= 0) | (w2 <= 0)) &
                                      (TWO_SIDED | fc.front());
                    if (__any(cand)) {
                        Pair pw; pw.w0 = w0; pw.w1 = w1; pw.w2 = w2;
                        float q0, q1, q2;
                        const float zp = clip_depth(q0, q1, q2, pw, fc);
                        if (cand & !(zp < A.near_ || zp > A.far_) & (zp < depth_min)) { depth_min = zp; face_min = f; }
                    }
                    continue;
                }
                Pair p;
                const bool live = eval_pair(p, fc, t.xp, t.yp, A.threshold, A.nis, A.amb_thr, t.valid) & t.valid;
                if (RGB == 2) {
                    alpha *= live ? 1.f - p.frag : 1.f;
                    continue;
                }
                if (live) {
                    alpha *= 1.f - p.frag;  // 'prod' alpha (:396), BEFORE the depth-range test
                    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
                    const float zp = clip_depth(q0, q1, q2, p, fc);
                    if (!(zp < A.near_ || zp > A.far_)) {
                        if (VIS) {   // the z-buffer winner of the hard render (:408-411) of the same faces, on the side
                            const bool incl = p.w0 <= 1 && p.w0 >= 0 && p.w1 <= 1 && p.w1 >= 0 && p.w2 <= 1 && p.w2 >= 0;
                            if (zp < depth_min && incl && (TWO_SIDED || fc.front())) { depth_min = zp; face_min = f; }
                        }
                        if (RGB == 0) {
                            const bool inside = p.w0 <= 1 && p.w0 >= 0 && p.w1 <= 1 && p.w1 >= 0 && p.w2 <= 1 && p.w2 >= 0;
                            if (zp < depth_min && inside && (TWO_SIDED || fc.front())) {
                                depth_min = zp;
                                face_min = f;
                                const char *tf = (const char *)(tex_n + (size_t)f * A.TS * 3);   // uniform per face
                                const unsigned t12 = (unsigned)texel_index(q0, q1, A.R) * 12u;
                                c0 = ld_u(tf, t12); c1 = ld_u(tf, t12 + 4); c2 = ld_u(tf, t12 + 8);
                            }
                        } else if (TWO_SIDED || fc.front()) {
                            const float zn = div_r(A.far_ - zp, A.far_ - A.near_, A.r_range);
                            float rescale = 1.f;
                            if (zn > smax) {
                                rescale = __expf((smax - zn) * A.inv_gamma);
                                smax = zn;
                            }
                            const float ez = __expf((zn - smax) * A.inv_gamma);
                            ssum = rescale * ssum + ez * p.frag;
                            wgt = ez * p.frag;
                            const char *tf = (const char *)(tex_n + (size_t)f * A.TS * 3);       // uniform per face
                            const unsigned t12 = (unsigned)texel_index(q0, q1, A.R) * 12u;
                            c0 = rescale * c0 + wgt * ld_u(tf, t12);
                            c1 = rescale * c1 + wgt * ld_u(tf, t12 + 4);
                            c2 = rescale * c2 + wgt * ld_u(tf, t12 + 8);
                        }
                    }
                }
                if (RGB == 1 && P2F) {  // :427-430, reduced over the 8x8 tile first
                    if (__any(wgt != 0.f)) {
                        float sx = wgt * gx, sy = wgt * gy, sw = wgt;
                        wave_sum_full3(sx, sy, sw);
                        if (t.lane < 4) {
                            const size_t o = ((size_t)t.n * F + f) * 2;
                            float *dst = t.lane < 2 ? A.p2f_info + o + t.lane : A.p2f_sum + o + (t.lane - 2);
                            atomicAdd(dst, t.lane == 0 ? sx : (t.lane == 1 ? sy : sw));
                        }
                    }
                }
            }
        }
    }

    if (!t.wave_on) return;
    if (RGB == 3) {
        if (t.valid) {
            float *ag = A.aggrs + (size_t)t.n * 2 * npix + pn;
            ag[0] = depth_min;
            ag[npix] = (float)face_min;
        }
        return;
    }
    const float o3 = 1.f - alpha;
    UMR_TRAP_IF(t.valid && umr_bad(o3), 2);
    if (RGB == 2) {
        if (t.valid) A.soft_colors[(size_t)t.n * npix + pn] = o3;
        if (A.pooled) {
            const int H = IS >> 1;
            float sv = o3 + __shfl_xor(o3, 1, 64);
            sv += __shfl_xor(sv, 8, 64);
            if (t.valid && !(t.lane & 1) && !(t.lane & 8))
                A.pooled[((size_t)t.n * H + (t.row >> 1)) * H + (t.xi >> 1)] = 0.25f * sv;
        }
        return;
    }
    // epilogue (:442-475)
    float o0, o1, o2;
    if (RGB == 0) { o0 = c0; o1 = c1; o2 = c2; }
    else { o0 = c0 / ssum; o1 = c1 / ssum; o2 = c2 / ssum; }
    UMR_TRAP_IF(t.valid && (umr_bad(o0) | umr_bad(o1) | umr_bad(o2) | umr_bad(ssum) | umr_bad(smax)), 2);
    if (RGB == 1 && A.state) {
        // packed saved state (RasterArgs::state; IS is a multiple of 8 here, so every lane of the wave holds a pixel): nothing
        // else is written at full resolution -- the caller consumes the pooled image, the backward this buffer
        const int tpr = IS >> 2;
        float *rec = A.state + (((size_t)t.n * tpr + (t.row >> 2)) * tpr + (t.xi >> 2)) * STATE_REC;
        const int i = (t.row & 3) * 4 + (t.xi & 3);
        rec[i] = __builtin_amdgcn_rcpf(ssum);
        rec[STATE_O_MAX + i] = smax;
        rec[STATE_O_ALPHA + i] = o3;
        // quad summaries for the backward's culling lanes: the quad's lanes are this one, ^1 (x) and ^8 (y)
        float mn = fminf(smax, __shfl_xor(smax, 1, 64)), sm = smax + __shfl_xor(smax, 1, 64);
        mn = fminf(mn, __shfl_xor(mn, 8, 64)); sm += __shfl_xor(sm, 8, 64);
        float op = o3 == 1.f ? 1.f : 0.f;
        op = fminf(op, __shfl_xor(op, 1, 64)); op = fminf(op, __shfl_xor(op, 8, 64));
        if (!(t.lane & 1) && !(t.lane & 8)) {
            const int q = ((t.row & 3) >> 1) * 2 + ((t.xi & 3) >> 1);
            rec[STATE_O_QMIN + q] = sm == sm ? mn : sm;     // fminf drops NaNs: a NaN anywhere in the quad keeps it visited
            rec[STATE_O_QOPAQUE + q] = op;
        }
    } else if (t.valid) {
        float *sc = A.soft_colors + (size_t)t.n * 4 * npix + pn;
        if (RGB == 1 || face_min != -1 || A.bg_arg) { sc[0] = o0; sc[npix] = o1; sc[2 * npix] = o2; }
        sc[3 * npix] = o3;
        float *ag = A.aggrs + (size_t)t.n * 2 * npix + pn;
        ag[0] = RGB == 0 ? depth_min : ssum;
        ag[npix] = RGB == 0 ? (float)face_min : smax;
    }
    if (VIS && t.valid) {
        if (A.vis_ids_only) A.vis[(size_t)t.n * npix + pn] = (float)face_min;
        else {
            float *vg = A.vis + (size_t)t.n * 2 * npix + pn;
            vg[0] = depth_min;
            vg[npix] = (float)face_min;
        }
    }
    if (A.pooled) {  // fused anti-aliasing 2x2 average (rasterizer.py:52-53); IS is even here
        float v[4] = {o0, o1, o2, o3};
        const int H = IS >> 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s = v[k] + __shfl_xor(v[k], 1, 64);
            s += __shfl_xor(s, 8, 64);
            if (t.valid && !(t.lane & 1) && !(t.lane & 8))
                A.pooled[(((size_t)t.n * 4 + k) * H + (t.row >> 1)) * H + (t.xi >> 1)] = 0.25f * s;
        }
    }
}

}  // namespace
