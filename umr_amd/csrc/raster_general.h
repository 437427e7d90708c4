// raster_general.h -- the rasterizer modes UMR itself never selects but the reference's extension accepts
// (functional/soft_rasterize.py:21-24 -> soft_rasterize_cuda_kernel.cu:154-218, :363-398, :442-451, :577-587, :634-642):
// distance 'hard' (0) / 'barycentric' (1) besides 'euclidean' (2), alpha 'hard' (0) / 'sum' (1) besides 'prod' (2), vertex
// textures (1, texture_size 3) besides surface textures (0).  One forward and one backward kernel with the modes as
// wave-uniform run-time switches: the same execution plan as the specialised kernels (16x16 pixel block per workgroup,
// per-mesh + per-block binning on the dilated bbox -- the reference's check_border (:33-38) precedes every mode, so the
// bins are exact for all of them -- wave-uniform face records from scalar loads, wave-reduced atomics), without their
// mode-specific culling and without the face-major backward.  Included by raster.hip only.
#pragma once
#include "raster_core.h"

namespace {

struct GenFrag {       // one (pixel, face) pair in any distance mode
    Pair p;            // w0..w2 always; b*, dx, dy, sign only for the euclidean mode
    float frag, dis;   // soft fragment D; signed squared distance (modes 1, 2)
    bool live;         // false where the reference `continue`s before touching the pixel (:355, :367, :371, :382)
};

__device__ __forceinline__ void gen_fragment(GenFrag &g, const Face &fc, const RasterArgs &A, float xp, float yp, bool valid) {
    if (A.dist_mode == 2) {
        g.live = eval_pair(g.p, fc, xp, yp, A.threshold, A.nis, A.amb_thr) & valid;
        g.frag = g.p.frag;
        g.dis = g.p.dx * g.p.dx + g.p.dy * g.p.dy;
        return;
    }
    const bool inb = !((xp > fc.g<R_XHI>()) | (xp < fc.g<R_XLO>()) | (yp > fc.g<R_YHI>()) | (yp < fc.g<R_YLO>()));
    const float w0 = (fc.inv<0>() * xp + fc.inv<1>() * yp) + fc.inv<2>();   // :25-29, the reference's operation order
    const float w1 = (fc.inv<3>() * xp + fc.inv<4>() * yp) + fc.inv<5>();
    const float w2 = (fc.inv<6>() * xp + fc.inv<7>() * yp) + fc.inv<8>();
    g.p.w0 = w0; g.p.w1 = w1; g.p.w2 = w2;
    g.p.b0 = g.p.b1 = g.p.b2 = g.p.dx = g.p.dy = g.p.sign = 0.f;
    if (A.dist_mode == 0) {   // :365-367
        g.frag = 1.f; g.dis = 0.f;
        g.live = inb & valid & (w0 <= 1) & (w0 >= 0) & (w1 <= 1) & (w1 >= 0) & (w2 <= 1) & (w2 >= 0);
        return;
    }
    float d = w0 > w1 ? (w1 > w2 ? w2 : w1) : (w0 > w2 ? w2 : w0);   // :154-157
    d = d > 0 ? d * d : -(d * d);
    g.dis = d;
    g.live = inb & valid & !(-d >= A.threshold);                     // :371
    g.frag = __builtin_amdgcn_rcpf(1.f + __expf(d * A.nis));         // 1 / (1 + exp(-dis / sigma)) (:372)
}

// colour of face `tf` at clipped barycentrics (q0, q1, q2), channel c (:178-195)
__device__ __forceinline__ float gen_sample(const char *tf, const RasterArgs &A, float q0, float q1, float q2, int c, unsigned t12) {
    if (!A.tex_vertex) return ld_u(tf, t12 + 4u * c);
    const float *v = (const float *)tf;   // wave-uniform: scalar loads
    return (q0 * v[c] + q1 * v[3 + c]) + q2 * v[6 + c];
}

__global__ __launch_bounds__(BLK_THREADS) void k_raster_forward_general(const RasterArgs A) {
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
    const bool softmax = A.rgb_mode == 1, two_sided = A.double_side != 0;

    float alpha = A.alpha_mode == 2 ? 1.f : 0.f;   // :335-336
    float ssum = __expf(A.eps / A.gamma), smax = A.eps;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, gx = 0.f, gy = 0.f;
    float depth_min = 10000000.f;
    int face_min = -1;
    if (t.valid) {
        if (A.bg_arg) { c0 = A.bg0; c1 = A.bg1; c2 = A.bg2; }
        else {
            const float *sc = A.soft_colors + (size_t)t.n * 4 * npix + pn;
            c0 = sc[0]; c1 = sc[npix]; c2 = sc[2 * npix];
        }
        if (softmax) {
            c0 *= ssum; c1 *= ssum; c2 *= ssum;
            if (A.with_p2f) { gx = A.grid[pn * 2]; gy = A.grid[pn * 2 + 1]; }
        }
    }

    const int *sb_ids;
    const int ncand = superblock_list(A, t, sb_ids);
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
            }
            unsigned long long m = __ballot(hit);
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const int f = __builtin_amdgcn_readlane(fcand, b);
                Face fc;
                load_face(fc, rec_n + (size_t)f * REC);
                GenFrag g;
                gen_fragment(g, fc, A, t.xp, t.yp, t.valid);
                float wgt = 0.f;
                if (g.live) {
                    if (A.alpha_mode == 0) alpha = g.frag > 0.5f ? 1.f : alpha;   // :390-391
                    else if (A.alpha_mode == 1) alpha += g.frag;                   // :393
                    else alpha *= 1.f - g.frag;                                    // :396
                    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
                    const float zp = clip_depth(q0, q1, q2, g.p, fc);
                    if (!(zp < A.near_ || zp > A.far_)) {
                        const char *tf = (const char *)(tex_n + (size_t)f * A.TS * 3);
                        const unsigned t12 = A.tex_vertex ? 0u : (unsigned)texel_index(q0, q1, A.R) * 12u;
                        if (!softmax) {   // :408-416
                            const bool inside = g.p.w0 <= 1 && g.p.w0 >= 0 && g.p.w1 <= 1 && g.p.w1 >= 0 && g.p.w2 <= 1 && g.p.w2 >= 0;
                            if (zp < depth_min && inside && (two_sided || fc.front())) {
                                depth_min = zp;
                                face_min = f;
                                c0 = gen_sample(tf, A, q0, q1, q2, 0, t12);
                                c1 = gen_sample(tf, A, q0, q1, q2, 1, t12);
                                c2 = gen_sample(tf, A, q0, q1, q2, 2, t12);
                            }
                        } else if (two_sided || fc.front()) {   // :417-436
                            const float zn = div_r(A.far_ - zp, A.far_ - A.near_, A.r_range);
                            float rescale = 1.f;
                            if (zn > smax) {
                                rescale = __expf((smax - zn) * A.inv_gamma);
                                smax = zn;
                            }
                            const float ez = __expf((zn - smax) * A.inv_gamma);
                            ssum = rescale * ssum + ez * g.frag;
                            wgt = ez * g.frag;
                            c0 = rescale * c0 + wgt * gen_sample(tf, A, q0, q1, q2, 0, t12);
                            c1 = rescale * c1 + wgt * gen_sample(tf, A, q0, q1, q2, 1, t12);
                            c2 = rescale * c2 + wgt * gen_sample(tf, A, q0, q1, q2, 2, t12);
                        }
                    }
                }
                if (softmax && A.with_p2f && __any(wgt != 0.f)) {   // :427-430, reduced over the tile first
                    const float sx = wave_sum_full(wgt * gx), sy = wave_sum_full(wgt * gy), sw = wave_sum_full(wgt);
                    if (t.lane < 4) {
                        const size_t o = ((size_t)t.n * F + f) * 2;
                        float *dst = t.lane < 2 ? A.p2f_info + o + t.lane : A.p2f_sum + o + (t.lane - 2);
                        atomicAdd(dst, t.lane == 0 ? sx : (t.lane == 1 ? sy : sw));
                    }
                }
            }
        }
    }

    if (!t.wave_on || !t.valid) return;
    float *sc = A.soft_colors + (size_t)t.n * 4 * npix + pn;
    sc[3 * npix] = A.alpha_mode == 0 ? alpha : (A.alpha_mode == 1 ? alpha / (float)F : 1.f - alpha);   // :442-451
    float *ag = A.aggrs + (size_t)t.n * 2 * npix + pn;
    if (!softmax) {
        if (face_min != -1 || A.bg_arg) { sc[0] = c0; sc[npix] = c1; sc[2 * npix] = c2; }
        ag[0] = depth_min;
        ag[npix] = (float)face_min;
    } else {
        sc[0] = c0 / ssum; sc[npix] = c1 / ssum; sc[2 * npix] = c2 / ssum;
        ag[0] = ssum;
        ag[npix] = smax;
    }
}

// Pixel-major backward for every mode (:480-656): lane = pixel, the wave walks the faces binned to its tile; the nine
// vertex gradients (and the nine vertex-colour gradients) of a visit are summed over the 64 lanes before the atomics,
// surface texel gradients go out per lane.  Summation order over pixels is as undefined as with the reference's atomics.
__global__ __launch_bounds__(BLK_THREADS) void k_raster_backward_general(const RasterArgs A) {
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
    const bool softmax = A.rgb_mode == 1, two_sided = A.double_side != 0;

    float ssum = 1.f, smax = 0.f, oc0 = 0.f, oc1 = 0.f, oc2 = 0.f, oa = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    if (t.valid) {
        const float *ag = A.aggrs + (size_t)t.n * 2 * npix + pn;
        ssum = ag[0]; smax = ag[npix];
        const float *sc = A.soft_colors + (size_t)t.n * 4 * npix + pn;
        oc0 = sc[0]; oc1 = sc[npix]; oc2 = sc[2 * npix]; oa = sc[3 * npix];
        const float *gp = A.grad_colors + (size_t)t.n * 4 * npix + pn;
        g0 = gp[0]; g1 = gp[npix]; g2 = gp[2 * npix]; g3 = gp[3 * npix];
    }

    const int *sb_ids;
    const int ncand = superblock_list(A, t, sb_ids);
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
            }
            unsigned long long m = __ballot(hit);
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const int f = __builtin_amdgcn_readlane(fcand, b);
                Face fc;
                load_face(fc, rec_n + (size_t)f * REC);
                GenFrag g;
                gen_fragment(g, fc, A, t.xp, t.yp, t.valid);
                float gv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                float gt[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // surface: [0..2] at texel tix; vertex: [j*3+c]
                int tix = 0;
                bool contrib = false;
                if (g.live) {
                    float c_xy = g3;                                                   // :577; 'hard' alpha adds it as is
                    if (A.alpha_mode == 1) c_xy = g3 / (float)F;                       // :582
                    else if (A.alpha_mode == 2) c_xy = g3 * ((1.f - oa) * __builtin_amdgcn_rcpf(fmaxf(1.f - g.frag, 1e-6f)));  // :584
                    float q0, q1, q2;
                    const float zp = clip_depth(q0, q1, q2, g.p, fc);
                    if (!(zp < A.near_ || zp > A.far_)) {   // :592 -- drops the alpha term as well
                        contrib = true;
                        const char *tf = (const char *)(tex_n + (size_t)f * TS * 3);
                        tix = A.tex_vertex ? 0 : texel_index(q0, q1, A.R);
                        const unsigned t12 = (unsigned)tix * 12u;
                        float ps = 0.f;
                        bool tex_on = false;
                        if (!softmax) {
                            tex_on = (float)f == smax;    // :596
                            ps = 1.f;
                        } else if (two_sided || fc.front()) {
                            tex_on = true;
                            const float zn = div_r(A.far_ - zp, A.far_ - A.near_, A.r_range);
                            ps = g.frag * __expf((zn - smax) * A.inv_gamma) * __builtin_amdgcn_rcpf(ssum);   // :608
                            float c_rgb = g0 * (gen_sample(tf, A, q0, q1, q2, 0, t12) - oc0);
                            c_rgb += g1 * (gen_sample(tf, A, q0, q1, q2, 1, t12) - oc1);
                            c_rgb += g2 * (gen_sample(tf, A, q0, q1, q2, 2, t12) - oc2);
                            c_rgb *= ps;
                            c_xy += c_rgb * __builtin_amdgcn_rcpf(g.frag);
                            const float c_z = -(c_rgb * A.inv_gamma * A.r_range) * zp * zp;   // :624
                            gv[2] = c_z * q0 * fc.g<R_RZ0>() * fc.g<R_RZ0>();
                            gv[5] = c_z * q1 * fc.g<R_RZ1>() * fc.g<R_RZ1>();
                            gv[8] = c_z * q2 * fc.g<R_RZ2>() * fc.g<R_RZ2>();
                        }
                        if (tex_on) {
                            if (!A.tex_vertex) { gt[0] = ps * g0; gt[1] = ps * g1; gt[2] = ps * g2; }
                            else {   // :215: grad_texture[3 j + c] += p * (w_j * g_c)
                                const float q[3] = {q0, q1, q2}, gc[3] = {g0, g1, g2};
#pragma unroll
                                for (int j = 0; j < 3; ++j)
#pragma unroll
                                    for (int c = 0; c < 3; ++c) gt[j * 3 + c] = ps * (q[j] * gc[c]);
                            }
                        }
                        c_xy *= g.frag * (1.f - g.frag) * (-A.nis);   // :632
                        if (A.dist_mode == 2) {                       // :637-642
                            const float k2 = 2.f * g.p.sign * c_xy;
                            const float b0 = k2 * g.p.b0, b1 = k2 * g.p.b1, b2 = k2 * g.p.b2;
                            gv[0] = b0 * g.p.dx; gv[1] = b0 * g.p.dy;
                            gv[3] = b1 * g.p.dx; gv[4] = b1 * g.p.dy;
                            gv[6] = b2 * g.p.dx; gv[7] = b2 * g.p.dy;
                        } else if (A.dist_mode == 1) {                // :160-175 with t = the unclipped barycentrics (:553)
                            const float w0 = g.p.w0, w1 = g.p.w1, w2 = g.p.w2;
                            const int pm = w0 > w1 ? (w1 > w2 ? 2 : 1) : (w0 > w2 ? 2 : 0);
                            const float ax = pm == 0 ? fc.inv<0>() : (pm == 1 ? fc.inv<3>() : fc.inv<6>());
                            const float ay = pm == 0 ? fc.inv<1>() : (pm == 1 ? fc.inv<4>() : fc.inv<7>());
                            const float s2 = 2.f * sqrtf(fabsf(g.dis));
                            const float iv[9] = {fc.inv<0>(), fc.inv<1>(), fc.inv<2>(), fc.inv<3>(), fc.inv<4>(), fc.inv<5>(),
                                                 fc.inv<6>(), fc.inv<7>(), fc.inv<8>()};
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const float kx = ((-ax * iv[3 * k]) * t.xp + (-ax * iv[3 * k + 1]) * t.yp) + (-ax * iv[3 * k + 2]);
                                const float ky = ((-ay * iv[3 * k]) * t.xp + (-ay * iv[3 * k + 1]) * t.yp) + (-ay * iv[3 * k + 2]);
                                gv[3 * k] = (kx * c_xy) * s2;
                                gv[3 * k + 1] = (ky * c_xy) * s2;
                            }
                        }
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
                    if (A.tex_vertex) {
                        float mine = 0.f;
#pragma unroll
                        for (int k = 0; k < 9; ++k) {
                            const float s = wave_sum_full(gt[k]);
                            if (t.lane == k) mine = s;
                        }
                        if (t.lane < 9) atomicAdd(gtf + t.lane, mine);
                    } else if (gt[0] != 0.f || gt[1] != 0.f || gt[2] != 0.f) {
                        atomicAdd(gtf + tix * 3 + 0, gt[0]);
                        atomicAdd(gtf + tix * 3 + 1, gt[1]);
                        atomicAdd(gtf + tix * 3 + 2, gt[2]);
                    }
                }
            }
        }
    }
}

}  // namespace
