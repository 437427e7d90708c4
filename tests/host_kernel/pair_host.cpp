// pair_host.cpp -- TEST INFRASTRUCTURE: the product's per-face set-up (face_setup_one, the body of k_face_setup) + eval_pair + clip_depth (raster_core.h, unmodified source)
// compiled for the host through device_shim.h and exported for ctypes.  One lane at a time.
#include "device_shim.h"
#include "../../umr_amd/csrc/raster_core.h"
#include "../../umr_amd/csrc/raster_general.h"

namespace {
// ---- a well-conditioned closest-point evaluation (TEST INSTRUMENT, not product code) -------------------------------------
// Round 3 built this as a 'lean' geometry for the face-major backward and measured it on the MI355X (HISTORY.md 4.7): rejected,
// because what it differs by from eval_pair -- the reference formulation's own rounding noise -- decides which rim pixels
// contribute.  It stays here to MEASURE that noise on the kernel source (test_reference_order_geometry_carries_rounding_noise).
// The reference obtains the closest boundary point through barycentrics of O(1) homogeneous products (:63-152): region
// tables, one edge parameter from differences of `face_sym`, the offset as sum_k (t_k - w_k) p_k.  Mathematically that IS
// the Euclidean closest point of the triangle's boundary (the obtuse-corner override exists to make it so), so for a
// well-conditioned face the same point follows from the three clamped edge projections directly: per edge e = (A, A + E)
//   t = clamp(<P - A, E> / |E|^2, 0, 1),  q = (P - A) - t E = P - Q_e,  d2_e = |q|^2,
// 9 full-rate VALU each on operands relative to the face's own vertices (no cancellation of O(1) terms), and the nearest
// of the three is the reference's point; inside the triangle the clamp never acts (the foot on the nearest edge's line
// lies on the edge), so one formula serves both branches.  The FORWARD keeps the reference's operation order -- outside
// the silhouette colours are ratios of weights ~1e-9 and have to carry the reference's own rounding noise to agree within
// 1e-4 -- but gradients are sums of such terms and are held to a relative tolerance; what differs is the reference's
// rounding noise in d^2 (~1e-7 absolute in the offset: up to ~1e-3 relative in D at the rim of the 3.9 px band, ~1e-5
// near the edge where the weight is).
struct LeanFace {   // per lane (VGPRs; the face is wave-uniform, the copies make every operand a full-rate VGPR source)
    float ax[3], ay[3];   // edge e starts at vertex e ...
    float ex[3], ey[3];   // ... and runs to vertex e + 1
    float rl[3];          // 1 / |E_e|^2
    float orient;         // +1 | -1: sign that makes the edge functions positive inside
};

template <class FaceT>
__device__ __forceinline__ void lean_setup(LeanFace &L, const FaceT &fc) {
    float x[3], y[3];
#ifdef UMR_HOST_SHIM
    x[0] = fc.template g<R_X0>(); y[0] = fc.template g<R_Y0>(); x[1] = fc.template g<R_X1>(); y[1] = fc.template g<R_Y1>();
    x[2] = fc.template g<R_X2>(); y[2] = fc.template g<R_Y2>();
#else
#define UMR_VMOV(dst, src) asm volatile("v_mov_b32 %0, %1" : "=v"(dst) : "s"(src))
    UMR_VMOV(x[0], fc.template g<R_X0>()); UMR_VMOV(y[0], fc.template g<R_Y0>()); UMR_VMOV(x[1], fc.template g<R_X1>());
    UMR_VMOV(y[1], fc.template g<R_Y1>()); UMR_VMOV(x[2], fc.template g<R_X2>()); UMR_VMOV(y[2], fc.template g<R_Y2>());
#undef UMR_VMOV
#endif
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int b = (e + 1) % 3;
        L.ax[e] = x[e]; L.ay[e] = y[e];
        L.ex[e] = x[b] - x[e]; L.ey[e] = y[b] - y[e];
        L.rl[e] = 1.f / (L.ex[e] * L.ex[e] + L.ey[e] * L.ey[e]);   // >= 1e-4 for flagged faces
    }
    // edge function of edge 0 at vertex 2 = twice the signed area
    L.orient = (L.ex[0] * (y[2] - y[0]) - L.ey[0] * (x[2] - x[0])) > 0.f ? 1.f : -1.f;
}

struct LeanSeg { float qx[3], qy[3], t[3], d2[3]; };   // per edge: P - Q_e, parameter of Q_e, squared distance

template <bool EDGE_FN>
__device__ __forceinline__ float lean_segments(LeanSeg &s, const LeanFace &L, float xp, float yp) {
    // returns (EDGE_FN) the smallest oriented edge function: > 0 <=> the pixel centre is strictly inside
    float cmin = 0.f;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const float px = xp - L.ax[e], py = yp - L.ay[e];
        const float u = fmaf(px, L.ex[e], py * L.ey[e]);
        const float t = fminf(fmaxf(u * L.rl[e], 0.f), 1.f);
        const float qx = fmaf(-t, L.ex[e], px), qy = fmaf(-t, L.ey[e], py);
        s.qx[e] = qx; s.qy[e] = qy; s.t[e] = t;
        s.d2[e] = fmaf(qx, qx, qy * qy);
        if (EDGE_FN) {
            const float c = fmaf(L.ex[e], py, -(L.ey[e] * px)) * L.orient;
            cmin = e == 0 ? c : fminf(cmin, c);
        }
    }
    return cmin;
}

}  // namespace

extern "C" {

static float g_thin_h = THIN_FACE_H;   // k_face_setup's thin-face threshold (the product default; host_set_thin_h for A/B)
void host_set_thin_h(float h) { g_thin_h = h; }

// faces [n,9] -> per (face, pixel): live flag, soft fragment D, unclipped barycentrics, dx, dy, clipped depth zp
int host_pairs(const float *faces, int n, const float *xp, const float *yp, int npix, float thr, float threshold, float nis,
               float near_, float far_, unsigned char *live, float *frag, float *dxy, float *zp_out) {
    float *rec = new float[(size_t)n * REC];
    float4 *bbox = new float4[n];
    blockDim.x = 1;
    for (int i = 0; i < n; ++i) {
        blockIdx.x = (unsigned)i; threadIdx.x = 0;
        face_setup_one(i, faces, nullptr, bbox, rec + (size_t)i * REC, thr, near_, far_, g_thin_h);
    }
    for (int i = 0; i < n; ++i) {
        Face fc;
        load_face(fc, rec + (size_t)i * REC);
        for (int p = 0; p < npix; ++p) {
            Pair pr;
            const bool ok = eval_pair(pr, fc, xp[p], yp[p], threshold, nis);
            const size_t o = (size_t)i * npix + p;
            live[o] = ok ? 1 : 0;
            frag[o] = pr.frag;
            dxy[2 * o] = pr.dx; dxy[2 * o + 1] = pr.dy;
            float q0, q1, q2;
            zp_out[o] = ok ? clip_depth(q0, q1, q2, pr, fc) : 0.f;
        }
    }
    delete[] rec; delete[] bbox;
    return 0;
}

// texel of the surface texture a covered pixel samples (clip_depth + texel_index, :54-59, :180-189), -1 where the pair is
// rejected or the pixel is not inside [0,1]^3 (the hard z-buffer's condition, :409)
int host_texels(const float *faces, int n, const float *xp, const float *yp, int npix, float thr, float threshold, float nis,
                int R, int *tix_out) {
    float *rec = new float[(size_t)n * REC];
    float4 *bbox = new float4[n];
    blockDim.x = 1;
    for (int i = 0; i < n; ++i) {
        blockIdx.x = (unsigned)i; threadIdx.x = 0;
        face_setup_one(i, faces, nullptr, bbox, rec + (size_t)i * REC, thr, 1.f, 100.f, g_thin_h);
    }
    for (int i = 0; i < n; ++i) {
        Face fc;
        load_face(fc, rec + (size_t)i * REC);
        for (int p = 0; p < npix; ++p) {
            Pair pr;
            const bool ok = eval_pair(pr, fc, xp[p], yp[p], threshold, nis);
            int t = -1;
            if (ok) {
                float q0, q1, q2;
                const float zp = clip_depth(q0, q1, q2, pr, fc);
                const bool incl = pr.w0 <= 1 && pr.w0 >= 0 && pr.w1 <= 1 && pr.w1 >= 0 && pr.w2 <= 1 && pr.w2 >= 0;
                if (incl && !(zp < 1.f || zp > 100.f)) t = texel_index(q0, q1, R);
            }
            tix_out[(size_t)i * npix + p] = t;
        }
    }
    delete[] rec; delete[] bbox;
    return 0;
}

// soft fragment of the general-mode kernels (raster_general.h gen_fragment) for dist_mode 0 (hard) / 1 (barycentric) / 2
int host_general_frag(const float *faces, int n, const float *xp, const float *yp, int npix, float thr, float threshold, float nis,
                      int dist_mode, unsigned char *live, float *frag) {
    float *rec = new float[(size_t)n * REC];
    float4 *bbox = new float4[n];
    blockDim.x = 1;
    for (int i = 0; i < n; ++i) {
        blockIdx.x = (unsigned)i; threadIdx.x = 0;
        face_setup_one(i, faces, nullptr, bbox, rec + (size_t)i * REC, thr, 1.f, 100.f, g_thin_h);
    }
    RasterArgs A = {};
    A.threshold = threshold; A.nis = nis; A.thr = thr; A.dist_mode = dist_mode;
    for (int i = 0; i < n; ++i) {
        Face fc;
        load_face(fc, rec + (size_t)i * REC);
        for (int p = 0; p < npix; ++p) {
            GenFrag g;
            gen_fragment(g, fc, A, xp[p], yp[p], true);
            live[(size_t)i * npix + p] = g.live ? 1 : 0;
            frag[(size_t)i * npix + p] = g.frag;
        }
    }
    delete[] rec; delete[] bbox;
    return 0;
}

// pixel-centre coordinate both ways: the fp64 expression of the reference (:325-326) and the fp32 shortcut for power-of-two
// images; also tile_setup's per-thread pixel coordinates for one (block, thread) of a launch
int host_ndc(int IS, float *ref_out, float *fast_out) {
    const bool pow2 = (IS & (IS - 1)) == 0;
    const float inv_is = 1.f / (float)IS;
    for (int i = 0; i < IS; ++i) { ref_out[i] = ndc_coord(i, IS); fast_out[i] = ndc_coord_fast(i, IS, inv_is, pow2); }
    return 0;
}
int host_tile_setup(int N, int IS, int no_xcd_remap, unsigned block, unsigned thread, int *out5) {
    RasterArgs A = {};
    A.N = N; A.IS = IS; A.tiles_x = (IS + BLK_W - 1) / BLK_W; A.tiles_y = (IS + BLK_H - 1) / BLK_H; A.no_xcd_remap = no_xcd_remap;
    blockIdx.x = block; threadIdx.x = thread; blockDim.x = BLK_THREADS;
    Tile t;
    tile_setup(t, A);
    out5[0] = t.n; out5[1] = t.xi; out5[2] = t.row; out5[3] = t.valid ? 1 : 0; out5[4] = t.wave_on ? 1 : 0;
    return 0;
}

int host_fm_owned_face(int xcd, int j, int per, int split) { return fm_owned_face(xcd, j, per, split); }

// the conservative tile-vs-dilated-triangle test both raster directions cull with: 1 = "some pixel of the tile may survive"
// tiles: [ntiles,4] = (cx, cy, hx, hy) centre and half extent of the tile's pixel-centre rectangle
int host_tile_may_hit(const float *faces, int n, const float *tiles, int ntiles, float thr, unsigned char *out) {
    float *rec = new float[(size_t)n * REC];
    float4 *bbox = new float4[n];
    blockDim.x = 1;
    for (int i = 0; i < n; ++i) {
        blockIdx.x = (unsigned)i; threadIdx.x = 0;
        face_setup_one(i, faces, nullptr, bbox, rec + (size_t)i * REC, thr, 1.f, 100.f, g_thin_h);
    }
    for (int i = 0; i < n; ++i) {
        const float4 *q = (const float4 *)(rec + (size_t)i * REC + R_I0);
        const float4 bb = bbox[i];
        for (int t = 0; t < ntiles; ++t) {
            const float cx = tiles[4 * t], cy = tiles[4 * t + 1], hx = tiles[4 * t + 2], hy = tiles[4 * t + 3];
            bool hit = !(cx - hx > bb.y || cx + hx < bb.x || cy - hy > bb.w || cy + hy < bb.z);   // the kernels' bbox test first
            if (hit) hit = tile_may_hit(q[0], q[1], q[2], cx, cy, hx, hy, thr + rec[(size_t)i * REC + R_CULL]);   // as the kernels call it
            out[(size_t)i * ntiles + t] = hit ? 1 : 0;
        }
    }
    delete[] rec; delete[] bbox;
    return 0;
}

// The two raster directions cull at different granularity -- the forward per 8x8 wave tile, the face-major backward per 4x4
// sub-tile -- and then decide per pixel with eval_pair.  For every face: all pixels under its dilated bbox (+ 2 pixels); counts
//   out[0] pixels eval_pair includes                     out[1] ... whose 8x8 tile the cull drops (the forward never sees them)
//   out[2] ... whose 4x4 sub-tile the cull drops         out[3] ... dropped at 8x8 but KEPT at 4x4: the backward includes a pair
// the forward did not -- where no nearer face covers the pixel the saved soft-max maximum is then far below the pair's depth and
// exp((zn - max) / gamma) overflows (round 3's non-finite training runs at configs[3], HISTORY.md 10).
//   out[4] ... kept at 4x4 whose 2x2 QUAD the backward's quad refinement drops (subtile_quads_may_hit; power-of-two images)
// noise_scale multiplies the per-face widening R_CULL of the band: 0 = the band at the exact threshold (round 3's early builds).
// first[0..3]: (face, xi, row, -) of the first out[3] case.
int host_cull_granularity(const float *faces, int n, int IS, float thr, float threshold, float nis, float noise_scale, long *out, int *first) {
    float *rec = new float[(size_t)n * REC];
    float4 *bbox = new float4[n];
    blockDim.x = 1;
    for (int i = 0; i < n; ++i) {
        blockIdx.x = (unsigned)i; threadIdx.x = 0;
        face_setup_one(i, faces, nullptr, bbox, rec + (size_t)i * REC, thr, 1.f, 100.f, g_thin_h);
    }
    const bool pow2 = (IS & (IS - 1)) == 0;
    const float inv_is = 1.f / (float)IS, h = 0.5f * IS;
    out[0] = out[1] = out[2] = out[3] = out[4] = 0;
    first[0] = -1;
    auto hit = [&](const float4 *q, const float4 bb, float band, int px0, int pr0, int T) {
        const int px1 = min(px0 + T - 1, IS - 1), pr1 = min(pr0 + T - 1, IS - 1);
        const float xl = ndc_coord_fast(px0, IS, inv_is, pow2), xh = ndc_coord_fast(px1, IS, inv_is, pow2);
        const float yh = ndc_coord_fast(IS - 1 - pr0, IS, inv_is, pow2), yl = ndc_coord_fast(IS - 1 - pr1, IS, inv_is, pow2);
        if (xl > bb.y || xh < bb.x || yl > bb.w || yh < bb.z) return false;                      // the kernels' bbox test first
        return tile_may_hit(q[0], q[1], q[2], 0.5f * (xl + xh), 0.5f * (yl + yh), 0.5f * (xh - xl), 0.5f * (yh - yl), band);
    };
    for (int i = 0; i < n; ++i) {
        Face fc;
        load_face(fc, rec + (size_t)i * REC);
        const float4 *q = (const float4 *)(rec + (size_t)i * REC + R_I0);
        const float4 bb = bbox[i];
        if (!(bb.x == bb.x && bb.y == bb.y && bb.z == bb.z && bb.w == bb.w)) continue;
        const float band = thr + noise_scale * rec[(size_t)i * REC + R_CULL];
        const int x0 = max((int)floorf(bb.x * h + h - 0.5f) - 2, 0), x1 = min((int)ceilf(bb.y * h + h - 0.5f) + 2, IS - 1);
        const int yi0 = max((int)floorf(bb.z * h + h - 0.5f) - 2, 0), yi1 = min((int)ceilf(bb.w * h + h - 0.5f) + 2, IS - 1);
        const int r0 = IS - 1 - yi1, r1 = IS - 1 - yi0;
        for (int ty = r0 / 8; ty <= r1 / 8; ++ty)
            for (int tx = x0 / 8; tx <= x1 / 8; ++tx) {
                const bool h8 = hit(q, bb, band, tx * 8, ty * 8, 8);
                for (int sy = 0; sy < 2; ++sy)
                    for (int sx = 0; sx < 2; ++sx) {
                        const int px0 = tx * 8 + sx * 4, pr0 = ty * 8 + sy * 4;
                        if (px0 >= IS || pr0 >= IS) continue;
                        const bool h4 = hit(q, bb, band, px0, pr0, 4);
                        unsigned qm = 15u;
                        if (h4 && pow2 && IS >= 4) {
                            const float xl = ndc_coord_fast(px0, IS, inv_is, true), xh = ndc_coord_fast(px0 + 3, IS, inv_is, true);
                            const float yh = ndc_coord_fast(IS - 1 - pr0, IS, inv_is, true), yl = ndc_coord_fast(IS - 1 - (pr0 + 3), IS, inv_is, true);
                            qm = subtile_quads_may_hit(q[0], q[1], q[2], 0.5f * (xl + xh), 0.5f * (yl + yh), 2.f * inv_is, band);
                        }
                        for (int row = pr0; row < min(pr0 + 4, IS); ++row)
                            for (int xi = px0; xi < min(px0 + 4, IS); ++xi) {
                                Pair pr;
                                if (!eval_pair(pr, fc, ndc_coord_fast(xi, IS, inv_is, pow2), ndc_coord_fast(IS - 1 - row, IS, inv_is, pow2), threshold, nis)) continue;
                                ++out[0];
                                out[1] += !h8; out[2] += !h4;
                                if (h4 && !((qm >> ((((row - pr0) >> 1) << 1) | ((xi - px0) >> 1))) & 1u)) ++out[4];
                                if (!h8 && h4) {
                                    if (!out[3]) { first[0] = i; first[1] = xi; first[2] = row; }
                                    ++out[3];
                                }
                            }
                    }
            }
    }
    delete[] rec; delete[] bbox;
    return 0;
}

// Replay of ONE face of a soft-max render through the arithmetic of both raster directions (tools/r4/replay_nan.py): for every pixel
// of face `f`'s window the forward state (running soft-max maximum and sum, :417-437) over ALL faces in ascending order -- a face
// contributes at a pixel when its 8x8 wave tile passes the forward's cull AND eval_pair includes the pixel AND the depth is in
// range --, then the backward's weight of face f at that pixel, ps = D exp((zn - max) / gamma) / sum (:608), where its 4x4 sub-tile
// passes the backward's cull.  noise_scale as in host_cull_granularity.  out[p * 8 ...] = (xi, row, D_f, zn_f, max, sum, ps,
// code) with code = 1 forward included f here | 2 backward includes f here; returns the number of window pixels (<= cap).
int host_replay_face(const float *faces, int n, int f, int IS, float thr, float threshold, float nis, float gamma, float near_, float far_,
                     float eps, float noise_scale, float amb_thr, float *out, int cap) {
    float *rec = new float[(size_t)n * REC];
    float4 *bbox = new float4[n];
    blockDim.x = 1;
    for (int i = 0; i < n; ++i) {
        blockIdx.x = (unsigned)i; threadIdx.x = 0;
        face_setup_one(i, faces, nullptr, bbox, rec + (size_t)i * REC, thr, near_, far_, g_thin_h);
    }
    const bool pow2 = (IS & (IS - 1)) == 0;
    const float inv_is = 1.f / (float)IS, h = 0.5f * IS, ig = 1.f / gamma, rr = 1.f / (far_ - near_);
    auto hit = [&](int i, int px0, int pr0, int T) {
        const float4 *q = (const float4 *)(rec + (size_t)i * REC + R_I0);
        const float4 bb = bbox[i];
        const float band = thr + noise_scale * rec[(size_t)i * REC + R_CULL];
        const int px1 = min(px0 + T - 1, IS - 1), pr1 = min(pr0 + T - 1, IS - 1);
        const float xl = ndc_coord_fast(px0, IS, inv_is, pow2), xh = ndc_coord_fast(px1, IS, inv_is, pow2);
        const float yh = ndc_coord_fast(IS - 1 - pr0, IS, inv_is, pow2), yl = ndc_coord_fast(IS - 1 - pr1, IS, inv_is, pow2);
        if (xl > bb.y || xh < bb.x || yl > bb.w || yh < bb.z) return false;
        return tile_may_hit(q[0], q[1], q[2], 0.5f * (xl + xh), 0.5f * (yl + yh), 0.5f * (xh - xl), 0.5f * (yh - yl), band);
    };
    const float4 bf = bbox[f];
    const int x0 = max((int)floorf(bf.x * h + h - 0.5f) - 2, 0), x1 = min((int)ceilf(bf.y * h + h - 0.5f) + 2, IS - 1);
    const int yi0 = max((int)floorf(bf.z * h + h - 0.5f) - 2, 0), yi1 = min((int)ceilf(bf.w * h + h - 0.5f) + 2, IS - 1);
    int np = 0;
    for (int row = IS - 1 - yi1; row <= IS - 1 - yi0 && np < cap; ++row)
        for (int xi = x0; xi <= x1 && np < cap; ++xi) {
            const float xp = ndc_coord_fast(xi, IS, inv_is, pow2), yp = ndc_coord_fast(IS - 1 - row, IS, inv_is, pow2);
            float ssum = expf(eps / gamma), smax = eps, Df = 0.f, znf = 0.f;
            int code = 0;
            for (int i = 0; i < n; ++i) {
                const float4 bb = bbox[i];
                if (xp > bb.y || xp < bb.x || yp > bb.w || yp < bb.z) continue;          // (block / tile bbox filters: implied by this one)
                if (!hit(i, xi & ~7, row & ~7, 8)) continue;                              // the forward's 8x8 wave tile
                Face fc;
                load_face(fc, rec + (size_t)i * REC);
                Pair pr;
                if (!eval_pair(pr, fc, xp, yp, threshold, nis, amb_thr)) continue;
                float q0, q1, q2;
                const float zp = clip_depth(q0, q1, q2, pr, fc);
                if (zp < near_ || zp > far_) continue;
                const float zn = div_r(far_ - zp, far_ - near_, rr);
                float rescale = 1.f;
                if (zn > smax) { rescale = expf((smax - zn) * ig); smax = zn; }
                ssum = rescale * ssum + expf((zn - smax) * ig) * pr.frag;
                if (i == f) code |= 1;
            }
            float ps = 0.f;
            {
                Face fc;
                load_face(fc, rec + (size_t)f * REC);
                Pair pr;
                if (hit(f, xi & ~3, row & ~3, 4) && eval_pair(pr, fc, xp, yp, threshold, nis, amb_thr)) {
                    float q0, q1, q2;
                    const float zp = clip_depth(q0, q1, q2, pr, fc);
                    if (!(zp < near_ || zp > far_)) {
                        znf = div_r(far_ - zp, far_ - near_, rr); Df = pr.frag;
                        ps = pr.frag * expf((znf - smax) * ig) * (1.0f / ssum);
                        code |= 2;
                    }
                }
            }
            float *o = out + (size_t)np * 8;
            o[0] = (float)xi; o[1] = (float)row; o[2] = Df; o[3] = znf; o[4] = smax; o[5] = ssum; o[6] = ps; o[7] = (float)code;
            ++np;
        }
    delete[] rec; delete[] bbox;
    return np;
}

// per (face, pixel): the closest boundary point by three clamped edge projections on operands relative to the face's own
// vertices (agrees with a float64 evaluation to ~1e-9): live (inside | d2 < threshold), soft fragment, P - Q
int host_accurate_pairs(const float *faces, int n, const float *xp, const float *yp, int npix, float thr, float threshold, float nis,
                        unsigned char *live, float *frag, float *qxy) {
    float *rec = new float[(size_t)n * REC];
    float4 *bbox = new float4[n];
    blockDim.x = 1;
    for (int i = 0; i < n; ++i) {
        blockIdx.x = (unsigned)i; threadIdx.x = 0;
        face_setup_one(i, faces, nullptr, bbox, rec + (size_t)i * REC, thr, 1.f, 100.f, g_thin_h);
    }
    for (int i = 0; i < n; ++i) {
        Face fc;
        load_face(fc, rec + (size_t)i * REC);
        LeanFace L;
        lean_setup(L, fc);
        for (int p = 0; p < npix; ++p) {
            LeanSeg sg;
            const bool inside = lean_segments<true>(sg, L, xp[p], yp[p]) > 0.f;
            const float dmin = fminf(fminf(sg.d2[0], sg.d2[1]), sg.d2[2]);
            const int e = sg.d2[0] <= fminf(sg.d2[1], sg.d2[2]) ? 0 : (sg.d2[1] <= sg.d2[2] ? 1 : 2);
            const size_t o = (size_t)i * npix + p;
            live[o] = (inside || dmin < threshold) ? 1 : 0;
            frag[o] = 1.0f / (1.f + expf((inside ? dmin : -dmin) * nis));
            qxy[2 * o] = sg.qx[e]; qxy[2 * o + 1] = sg.qy[e];
        }
    }
    delete[] rec; delete[] bbox;
    return 0;
}

}
