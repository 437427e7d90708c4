// raster.hip -- soft rasterizer forward / backward for MI355X (gfx950, wave64).
//
// Behaviour follows the reference kernels
//   external/SoftRas/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu:223-282 (face preprocessing),
//   :286-476 (forward), :480-656 (backward)
// but the execution plan is CDNA4-native rather than "one thread per pixel looping over all faces":
//
//   * a 256-thread workgroup owns a 16x16 pixel block; each of its 4 wavefronts owns one 8x8 tile
//     (lane = pixel), so a face's per-wave data is wave-uniform and is fetched with scalar loads;
//   * faces are binned per block IN the kernel: the block scans the compact [N,F] bbox array
//     (coalesced float4 loads), ballots, and appends surviving face ids to an LDS list in
//     ASCENDING index order (the forward's online soft-max and the p2f weights depend on visit
//     order, reference :421-430);  each wave then filters the LDS list against its own 8x8 tile
//     with one lane per candidate face + ballot and walks the set bits;
//   * the forward's p2f accumulators are reduced across the wavefront (DPP) before touching memory: one atomic
//     per (tile, face, component) instead of the reference's one per (pixel, face, component);
//   * the backward is FACE-major (k_raster_backward_fm): one wavefront per face walks the face's 4x4-pixel
//     sub-tiles four at a time, keeps the 9 vertex gradients in registers and the texel gradients in LDS, and
//     writes each face's result once -- no global atomics at all.
//
// Numerics: fp32 throughout, IEEE division, no FMA contraction (-ffp-contract=off) so that the
// branch-deciding quantities (barycentrics, distances, depth) round like the reference's
// scalar_t=float code; the reference's stray double sub-expressions are evaluated in float
// (differences <= 1 ulp, covered by the 1e-4 parity tolerance).
#include "umr_common.h"
#include <mutex>
#include <string>
#include <vector>

#define REC 64         // floats per preprocessed face record
#define LIST_CAP 2048  // LDS face list capacity (faces are processed in super-chunks of this many)
#ifndef BLK_W
#define BLK_W 16   // measured on MI355X (N=128, F=1280, IS=512, soft-max forward): 16x16 1.55 ms, 32x8 / 16x8 1.64,
#define BLK_H 16   // 32x16 1.81, 32x32 2.08, 8x8 2.28 -- 4 waves share one binning pass and still schedule finely
#endif
#define BLK_WX (BLK_W / 8)                          // 8x8 wave tiles across / in the workgroup
#define BLK_THREADS (BLK_WX * (BLK_H / 8) * 64)

namespace {

// Record layout (floats) written by k_face_setup: 256 bytes per face.
//   [0,32)  wave-uniform part, fetched with 2 x s_load_dwordx16 into SGPRs
//   [32,56) three 32-byte edge blocks {a0, a1, a2, a[v1], den, RN(1/den), -, -}; a lane reads ONLY the block
//           of its nearest edge (per-lane address, 2 x global_load_dwordx4, L1-resident)
enum { R_XLO = 0, R_XHI = 1, R_YLO = 2, R_YHI = 3, R_X0 = 4, R_Y0 = 5, R_X1 = 6, R_Y1 = 7, R_X2 = 8, R_Y2 = 9,
       R_Z0 = 10, R_Z1 = 11, R_Z2 = 12, R_RZ0 = 13, R_RZ1 = 14, R_RZ2 = 15,
       R_INV = 16, R_K0 = 25, R_K1 = 26, R_K2 = 27, R_FLAGS = 28, R_FRONT = 29, R_OX = 30, R_OY = 31,
       R_EDGE = 32 };

struct RasterArgs {
    const float4 *bbox;   // [N*F] (xlo, xhi, ylo, yhi) = bbox dilated by sqrt(threshold)
    const float *rec;     // [N*F*REC]
    const float *textures;
    const float *grid;
    float *aggrs;
    float *p2f_info;
    float *p2f_sum;
    float *soft_colors;
    float *pooled;
    // backward only
    const float *grad_colors;
    float *grad_faces;
    float *grad_textures;
    int N, F, IS, TS, R;
    float near_, far_, eps, sigma, threshold, gamma;
    float thr;        // sqrt(threshold)
    float nis;        // -1/sigma
    float r_range;    // RN(1/(far-near))
    float inv_gamma;
    int double_side, with_p2f, grad_pooled, need_gf, need_gt;
    int tiles_x, tiles_y;
    int tex_group;    // K >= 1: mesh n samples textures[n / K] (K views share one texture set)
    int bg_arg;       // background passed by value: soft_colors arrives uninitialised
    float bg0, bg1, bg2;
};

// ---- per-face preprocessing (:223-282) + packed record for the raster kernels ----------------
__global__ void k_face_setup(const float *__restrict__ faces, float *__restrict__ faces_info,
                             float4 *__restrict__ bbox, float *__restrict__ rec, int total, float thr,
                             float near_, float far_) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float *f = faces + (size_t)i * 9;
    const float x0 = f[0], y0 = f[1], z0 = f[2], x1 = f[3], y1 = f[4], z1 = f[5], x2 = f[6], y2 = f[7], z2 = f[8];
    float adj[9] = {y1 - y2, x2 - x1, x1 * y2 - x2 * y1,
                    y2 - y0, x0 - x2, x2 * y0 - x0 * y2,
                    y0 - y1, x1 - x0, x0 * y1 - x1 * y0};
    float det = x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0);
    det = det > 0 ? fmaxf(det, 1e-10f) : fminf(det, -1e-10f);
    float inv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) inv[k] = adj[k] / det;
    const float px[3] = {x0, x1, x2}, py[3] = {y0, y1, y2};
    float sym[9];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) sym[j * 3 + k] = px[j] * px[k] + py[j] * py[k] + 1.f;
    int obt = -1;
#pragma unroll
    for (int k = 2; k >= 0; --k) {  // first obtuse corner wins (:273-281)
        const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        if ((px[k1] - px[k]) * (px[k2] - px[k]) + (py[k1] - py[k]) * (py[k2] - py[k]) < 0) obt = k;
    }
    if (faces_info) {
        float *fi = faces_info + (size_t)i * 27;
#pragma unroll
        for (int k = 0; k < 9; ++k) { fi[k] = inv[k]; fi[9 + k] = sym[k]; }
        fi[18] = obt == 0 ? 1.f : 0.f; fi[19] = obt == 1 ? 1.f : 0.f; fi[20] = obt == 2 ? 1.f : 0.f;
    }
    const float xlo = fminf(fminf(x0, x1), x2) - thr, xhi = fmaxf(fmaxf(x0, x1), x2) + thr;
    const float ylo = fminf(fminf(y0, y1), y2) - thr, yhi = fmaxf(fmaxf(y0, y1), y2) + thr;
    bbox[i] = make_float4(xlo, xhi, ylo, yhi);
    // ---- packed record: three 64-byte lines, fetched by the raster kernels with 3 x s_load_dwordx16 ----
    float *r = rec + (size_t)i * REC;
    r[R_XLO] = xlo; r[R_XHI] = xhi; r[R_YLO] = ylo; r[R_YHI] = yhi;
    r[R_X0] = x0; r[R_Y0] = y0; r[R_X1] = x1; r[R_Y1] = y1; r[R_X2] = x2; r[R_Y2] = y2;
    r[R_Z0] = z0; r[R_Z1] = z1; r[R_Z2] = z2;
    r[R_RZ0] = 1.f / z0; r[R_RZ1] = 1.f / z1; r[R_RZ2] = 1.f / z2;  // correctly rounded (Markstein division)
#pragma unroll
    for (int k = 0; k < 9; ++k) r[R_INV + k] = inv[k];
    // squared height of corner c over its opposite edge: inside the triangle the squared distance to that
    // edge's line is w_c^2 K_c -- used only to PICK the nearest edge (:99), the distance itself is then
    // evaluated with the reference's own formula
    const float det_raw = x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int a = (c + 1) % 3, b = (c + 2) % 3;
        const float ex = px[a] - px[b], ey = py[a] - py[b];
        r[R_K0 + c] = det_raw * det_raw / fmaxf(ex * ex + ey * ey, 1e-30f);
    }
    // depth chain may use reciprocal-multiply division only when every z is an ordinary positive number
    const bool sane = z0 > 1e-20f && z1 > 1e-20f && z2 > 1e-20f && z0 < 1e20f && z1 < 1e20f && z2 < 1e20f;
    // bit 3: every vertex depth strictly inside (near, far) => the interpolated depth (a convex combination of the
    // 1/z_k with positive clipped weights) can never be rejected by the depth-range test (:404, :592)
    const float zmin = fminf(fminf(z0, z1), z2), zmax = fmaxf(fmaxf(z0, z1), z2);
    const bool inrange = sane && zmin > near_ * 1.0001f && zmax < far_ * 0.9999f;
    // bits 0-1: obtuse corner + 1 (0 = none); bit 2: slow division path
    r[R_FLAGS] = __int_as_float((obt + 1) | (sane ? 0 : 4) | (inrange ? 8 : 0));
    r[R_FRONT] = ((y2 - y0) * (x1 - x0) < (y1 - y0) * (x2 - x0)) ? 1.f : 0.f;  // :42-44
    // vector of the obtuse-corner override test (:116,:119,:122): corner k -> p_{k+2} - p_k
    const int ob = obt < 0 ? 0 : obt;
    r[R_OX] = px[(ob + 2) % 3] - px[ob];
    r[R_OY] = py[(ob + 2) % 3] - py[ob];
    // edge e = (e, e+1): a_e[j] = sym[e][j] - sym[e+1][j] (:82-84,:133-135); den_e = a_e[e] - a_e[e+1]
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int e1 = (e + 1) % 3;
        float a[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) a[j] = sym[3 * e + j] - sym[3 * e1 + j];
        const float den = a[e] - a[e1];
        float *eb = r + R_EDGE + 8 * e;
        eb[0] = a[0]; eb[1] = a[1]; eb[2] = a[2]; eb[3] = a[e1];
        eb[4] = den; eb[5] = 1.f / den; eb[6] = 0.f; eb[7] = 0.f;
    }
#pragma unroll
    for (int k = R_EDGE + 24; k < REC; ++k) r[k] = 0.f;
}

__device__ __forceinline__ float ndc_coord(int i, int IS) {  // (2i + 1 - IS) / IS, evaluated in double (:325-326)
    return (float)((2.0 * i + 1.0 - IS) / IS);
}

// Same value without fp64 when IS is a power of two (every BASELINE config): 2i+1-IS is an exact small
// integer and the division is an exponent shift, so float arithmetic is exact -- provably identical bits.
__device__ __forceinline__ float ndc_coord_fast(int i, int IS, float inv_is, bool pow2) {
    return pow2 ? (float)(2 * i + 1 - IS) * inv_is : ndc_coord(i, IS);
}

// The record address is wave-uniform (face id comes from v_readlane / the wave id); reading it through the
// constant address space makes the backend emit s_load_dwordx16 (scalar cache, SGPR operands) instead of
// 64-lane broadcast vector loads.  Three explicit 64-byte vector loads issue back to back and are waited for
// once.  Safe: the records are written by k_face_setup in an EARLIER launch.
typedef float v16f __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) v16f cv16f_t;

// Load through a wave-uniform base pointer plus a per-lane 32-bit BYTE offset: the form the backend turns into
// `global_load_dword v, v_off, s[base:base+1]` (scalar base, no 64-bit per-lane address arithmetic).
__device__ __forceinline__ float ld_u(const char *base, unsigned byte_off) {
    return *(const float *)(base + byte_off);
}
__device__ __forceinline__ float4 ld_u4(const char *base, unsigned byte_off) {
    return *(const float4 *)(base + byte_off);
}

struct Face {  // wave-uniform: 32 SGPRs + the record's address
    v16f qa, qb;
    const char *edges;    // three 32-byte edge blocks (global memory, read per lane: uniform base + k * 32)
    template <int I> __device__ __forceinline__ float g() const {
        if constexpr (I < 16) return qa[I];
        else return qb[I - 16];
    }
    __device__ __forceinline__ int obt() const { return (__float_as_int(g<R_FLAGS>()) & 3) - 1; }
    __device__ __forceinline__ bool front() const { return g<R_FRONT>() != 0.f; }
    __device__ __forceinline__ bool slow() const { return (__float_as_int(g<R_FLAGS>()) & 4) != 0; }
    __device__ __forceinline__ bool depth_in_range() const { return (__float_as_int(g<R_FLAGS>()) & 8) != 0; }
};

__device__ __forceinline__ void load_face(Face &fc, const float *rg) {
    cv16f_t *r = (cv16f_t *)rg;
    fc.qa = r[0]; fc.qb = r[1];
    fc.edges = (const char *)(rg + R_EDGE);
}

struct Pair {  // per-lane result of the pixel/face geometry
    float w0, w1, w2;   // unclipped barycentrics
    float b0, b1, b2;   // barycentrics of the closest boundary point (the reference's t + w, :640)
    float dx, dy, sign, frag;
};

// RN(1/b) for ordinary b: v_rcp_f32 (1 ulp) + one Newton step
__device__ __forceinline__ float rcp_nr(float b) {
    const float r = __builtin_amdgcn_rcpf(b);
    return fmaf(fmaf(-b, r, 1.f), r, r);
}
// a/b given r ~ RN(1/b): Markstein's correction -> correctly rounded quotient for ordinary operands
__device__ __forceinline__ float div_r(float a, float b, float r) {
    const float q = a * r;
    return fmaf(fmaf(-b, q, a), r, q);
}

// bbox reject (:355), barycentric (:25-29), euclidean distance (:63-152), threshold reject (:382),
// sigmoid (:383).  Returns false when the reference would `continue` before touching the pixel.
__device__ __forceinline__ bool eval_pair(Pair &p, const Face &fc, float xp, float yp, float threshold,
                                          float neg_inv_sigma) {
    // Written branch-free (predicates + selects): per-lane divergence would otherwise cost ~80 scalar
    // exec-mask instructions per face visit.  Dead lanes compute garbage that the returned predicate masks.
    const bool inb = !((xp > fc.g<R_XHI>()) | (xp < fc.g<R_XLO>()) | (yp > fc.g<R_YHI>()) | (yp < fc.g<R_YLO>()));
    // barycentrics in the reference's operation order (no FMA): they decide inside/outside, feed the depth
    // chain and -- through cancellation -- carry ~1e-6 of rounding noise that has to match the reference's
    const float w0 = (fc.g<R_INV + 0>() * xp + fc.g<R_INV + 1>() * yp) + fc.g<R_INV + 2>();
    const float w1 = (fc.g<R_INV + 3>() * xp + fc.g<R_INV + 4>() * yp) + fc.g<R_INV + 5>();
    const float w2 = (fc.g<R_INV + 6>() * xp + fc.g<R_INV + 7>() * yp) + fc.g<R_INV + 8>();
    p.w0 = w0; p.w1 = w1; p.w2 = w2;
    const bool inside = (w0 > 0) & (w1 > 0) & (w2 > 0) & (w0 < 1) & (w1 < 1) & (w2 < 1);
    // inside: nearest edge LINE, first minimum in the reference's order k = 0,1,2 (:78-107); edge k is opposite
    // corner k+2 and its squared distance is w_c^2 K_c
    const float m0 = w2 * w2 * fc.g<R_K2>(), m1 = w0 * w0 * fc.g<R_K0>(), m2 = w1 * w1 * fc.g<R_K1>();
    const bool c1 = m1 < m0;
    const float best = c1 ? m1 : m0;
    const int kin = (m2 < best) ? 2 : (c1 ? 1 : 0);
    // outside: region selection (:112-126), lowest priority first so the highest-priority match is applied last
    const int ob = fc.obt();
    // obtuse corner's coordinates: a wave-uniform pick among SGPRs, written with masks so that it stays three scalar
    // and/or ops -- as nested selects the compiler turns it into a dynamically indexed vector read, which on gfx9
    // means copying 16 SGPRs to VGPRs (8 v_mov_b64 + s_set_gpr_idx) on every face visit
    const int mk0 = -(int)(ob == 0), mk1 = -(int)(ob == 1), mk2 = -(int)(ob == 2);
    const float cx = __int_as_float((__float_as_int(fc.g<R_X0>()) & mk0) | (__float_as_int(fc.g<R_X1>()) & mk1) |
                                    (__float_as_int(fc.g<R_X2>()) & mk2));
    const float cy = __int_as_float((__float_as_int(fc.g<R_Y0>()) & mk0) | (__float_as_int(fc.g<R_Y1>()) & mk1) |
                                    (__float_as_int(fc.g<R_Y2>()) & mk2));
    const bool ovr = (xp - cx) * fc.g<R_OX>() + (yp - cy) * fc.g<R_OY>() > 0;
    // Region code m = n0 | n1 << 1 | n2 << 2 with n_k = (w_k <= 0); the reference's if-chain (:112-126) is a table of
    // m -- single flag: opposite edge (n0 -> 1, n1 -> 2, n2 -> 0); two flags: the vertex region between them
    // ({n0,n1} -> 2, {n2,n0} -> 1, {n1,n2} -> 0; all three -- degenerate faces only -- ends like {n1,n2}); none: -1 --
    // plus ONE wave-uniform exception: in the vertex region of the flagged obtuse corner `ob` the other edge is taken
    // when the pixel lies on its far side (`ovr`).  Entries are stored +1 in 2 bits each.
    const int m = min((w0 <= 0 ? 1 : 0) | (w1 <= 0 ? 2 : 0) | (w2 <= 0 ? 4 : 0), 6);
    constexpr unsigned KOUT_LUT = (0u << 0) | (2u << 2) | (3u << 4) | (3u << 6) | (1u << 8) | (2u << 10) | (1u << 12);
    const int m_ob = ob == 0 ? 6 : (ob == 1 ? 5 : (ob == 2 ? 3 : -1));   // two-flag code of the obtuse corner's region
    const int k_ob = ob == 0 ? 2 : (ob == 1 ? 0 : 1);                     // the edge its override selects
    int kout = (int)((KOUT_LUT >> (2 * m)) & 3u) - 1;
    kout = ((m == m_ob) & ovr) ? k_ob : kout;
    const int ksel = inside ? kin : kout;
    const bool kvalid = ksel >= 0;  // k = -1: reference UB (index -1); defined here and in the oracle as "skip"
    const int k = max(ksel, 0);
    // t[v0] = (w . a - a[v1]) / (a[v0] - a[v1]) in the reference's operation order (:86,:137); IEEE-exact
    // quotient through Markstein's correction.  Far from the silhouette the soft-max renormalises weights
    // D ~ exp(-d^2/sigma) ~ 1e-9, amplifying rounding noise in d^2 ~20x: parity there needs the reference's
    // own noise, i.e. its own arithmetic, not just the same formula.
    const unsigned ko = (unsigned)k * 32u;
    const float4 ea = ld_u4(fc.edges, ko), eb = ld_u4(fc.edges, ko + 16u);  // {a0,a1,a2,a[v1]}, {den, 1/den, -, -}
    const float tv = div_r(((w0 * ea.x + w1 * ea.y) + w2 * ea.z) - ea.w, eb.x, eb.y);
    const bool k0 = k == 0, k1 = k == 1;
    const float tb = 1.f - tv;
    const float ba = inside ? tv : fminf(fmaxf(tv, 0.f), 1.f);  // unclamped inside (:86-88), clamped outside (:142-145)
    const float bb = inside ? tb : fminf(fmaxf(tb, 0.f), 1.f);
    const float b0 = k0 ? ba : (k1 ? 0.f : bb);
    const float b1 = k0 ? bb : (k1 ? ba : 0.f);
    const float b2 = k0 ? 0.f : (k1 ? bb : ba);
    const float t0 = b0 - w0, t1 = b1 - w1, t2 = b2 - w2;
    const float dx = (t0 * fc.g<R_X0>() + t1 * fc.g<R_X1>()) + t2 * fc.g<R_X2>();  // :95-96, :148-149
    const float dy = (t0 * fc.g<R_Y0>() + t1 * fc.g<R_Y1>()) + t2 * fc.g<R_Y2>();
    const float dis = dx * dx + dy * dy;
    p.b0 = b0; p.b1 = b1; p.b2 = b2; p.dx = dx; p.dy = dy;
    p.sign = inside ? 1.f : -1.f;
    // 1 / (1 + exp(-sign * dis / sigma))
    const float e = __expf((inside ? dis : -dis) * neg_inv_sigma);
    p.frag = __builtin_amdgcn_rcpf(1.f + e);
    return inb & kvalid & (inside | !(dis >= threshold));  // rejects of :355, :382
}

// barycentric_clip (:54-59) + perspective-correct depth (:403).  The soft-max weights are exp(zn/gamma) with
// gamma = 1e-4: one ulp of zn moves a weight by ~7e-4, so this chain reproduces the reference's IEEE
// divisions to the last bit (Markstein-corrected reciprocal multiplies; plain IEEE when z is degenerate).
__device__ __forceinline__ float clip_depth(float &c0, float &c1, float &c2, const Pair &p, const Face &fc) {
    c0 = fmaxf(fminf(p.w0, 1.f - 1e-5f), 1e-5f);
    c1 = fmaxf(fminf(p.w1, 1.f - 1e-5f), 1e-5f);
    c2 = fmaxf(fminf(p.w2, 1.f - 1e-5f), 1e-5f);
    const float s = fmaxf(c0 + c1 + c2, 1e-5f);
    if (fc.slow()) {
        c0 /= s; c1 /= s; c2 /= s;
        return 1.f / (c0 / fc.g<R_Z0>() + c1 / fc.g<R_Z1>() + c2 / fc.g<R_Z2>());
    }
    const float rs = rcp_nr(s);
    c0 = div_r(c0, s, rs); c1 = div_r(c1, s, rs); c2 = div_r(c2, s, rs);
    const float x = (div_r(c0, fc.g<R_Z0>(), fc.g<R_RZ0>()) + div_r(c1, fc.g<R_Z1>(), fc.g<R_RZ1>())) + div_r(c2, fc.g<R_Z2>(), fc.g<R_RZ2>());
    const float rx = rcp_nr(x);
    return fmaf(fmaf(-x, rx, 1.f), rx, rx);
}

__device__ __forceinline__ int texel_index(float c0, float c1, int R) {  // :180-189
    if (R == 1) return 0;
    const int wx = (int)(c0 * R), wy = (int)(c1 * R);
    if ((c0 + c1) * R - wx - wy <= 1) return wy * R + wx;
    return (R - 1 - wy) * R + (R - 1 - wx);
}


// Conservative "can any pixel centre of this tile survive the reference's rejects?" test.  A pixel whose
// perpendicular distance to the outer side of ONE edge line exceeds sqrt(threshold) is outside the triangle and
// farther than the threshold from it (whatever edge the reference's region logic picks, its clamped closest
// point is at least that far), so it is rejected at :382.  w_c is affine in the pixel position, so its maximum
// over the tile is w_c(centre) + hx |dw_c/dx| + hy |dw_c/dy|; signed distance = w_c * h_c with h_c^2 = K_c.
// NaN / degenerate faces (K_c = 0) never cull.  1e-3 (in barycentric units) absorbs rounding.
__device__ __forceinline__ bool tile_may_hit(const float4 i0, const float4 i1, const float4 i2, float cx, float cy,
                                             float hx, float hy, float thr) {
    // i0 = inv[0..3], i1 = inv[4..7], i2 = {inv[8], K0, K1, K2}
    const float w0 = fmaf(i0.x, cx, fmaf(i0.y, cy, i0.z)) + (hx * fabsf(i0.x) + hy * fabsf(i0.y));
    const float w1 = fmaf(i0.w, cx, fmaf(i1.x, cy, i1.y)) + (hx * fabsf(i0.w) + hy * fabsf(i1.x));
    const float w2 = fmaf(i1.z, cx, fmaf(i1.w, cy, i2.x)) + (hx * fabsf(i1.z) + hy * fabsf(i1.w));
    const bool out = w0 < -(thr * __frsqrt_rn(i2.y)) - 1e-3f || w1 < -(thr * __frsqrt_rn(i2.z)) - 1e-3f ||
                     w2 < -(thr * __frsqrt_rn(i2.w)) - 1e-3f;
    return !out;
}

// XCD-aware work mapping: hardware places workgroup b on XCD b % 8; give each XCD a contiguous
// run of (mesh, tile) work items so one mesh's face records stay in one L2.
__device__ __forceinline__ int xcd_remap(int b, int total) {
    return (total % 8 == 0) ? (b % 8) * (total / 8) + b / 8 : b;
}

struct Tile {
    int n, lane, wave, xi, row;
    bool valid, wave_on;
    float xp, yp;
    float bxlo, bxhi, bylo, byhi;  // block bounds (pixel centres)
    float wxlo, wxhi, wylo, wyhi;  // wave tile bounds
};

__device__ __forceinline__ void tile_setup(Tile &t, const RasterArgs &A) {
    const int total = A.N * A.tiles_x * A.tiles_y;
    int wid = xcd_remap(blockIdx.x, total);
    const int bx = wid % A.tiles_x; wid /= A.tiles_x;
    const int by = wid % A.tiles_y;
    t.n = wid / A.tiles_y;
    t.lane = threadIdx.x & 63;
    t.wave = threadIdx.x >> 6;
    const int IS = A.IS;
    const int px0 = bx * BLK_W + (t.wave % BLK_WX) * 8, py0 = by * BLK_H + (t.wave / BLK_WX) * 8;
    t.xi = px0 + (t.lane & 7);
    t.row = py0 + (t.lane >> 3);
    t.valid = t.xi < IS && t.row < IS;
    t.wave_on = px0 < IS && py0 < IS;
    // pixel centres: exact float path when IS is a power of two (12 fp64 divisions per thread otherwise -- they
    // were 40 % of the silhouette kernel's VALU instructions)
    const bool pow2 = (IS & (IS - 1)) == 0;
    const float inv_is = 1.f / (float)IS;
    t.xp = ndc_coord_fast(t.xi, IS, inv_is, pow2);
    t.yp = ndc_coord_fast(IS - 1 - t.row, IS, inv_is, pow2);
    t.bxlo = ndc_coord_fast(bx * BLK_W, IS, inv_is, pow2);
    t.bxhi = ndc_coord_fast(min(bx * BLK_W + BLK_W - 1, IS - 1), IS, inv_is, pow2);
    t.byhi = ndc_coord_fast(IS - 1 - by * BLK_H, IS, inv_is, pow2);
    t.bylo = ndc_coord_fast(IS - 1 - min(by * BLK_H + BLK_H - 1, IS - 1), IS, inv_is, pow2);
    t.wxlo = ndc_coord_fast(px0, IS, inv_is, pow2);
    t.wxhi = ndc_coord_fast(min(px0 + 7, IS - 1), IS, inv_is, pow2);
    t.wyhi = ndc_coord_fast(IS - 1 - py0, IS, inv_is, pow2);
    t.wylo = ndc_coord_fast(IS - 1 - min(py0 + 7, IS - 1), IS, inv_is, pow2);
}

// Block-level binning of faces [f0, f1) into the LDS list, ascending order.  Returns the count.
__device__ __forceinline__ int build_list(int *s_list, int *s_wcnt, const float4 *__restrict__ bbox_n, int f0,
                                          int f1, const Tile &t) {
    int count = 0;
    for (int c = f0; c < f1; c += BLK_THREADS) {
        const int f = c + (int)threadIdx.x;
        bool pass = false;
        if (f < f1) {
            const float4 bb = bbox_n[f];
            // same predicate as the per-pixel reject, applied to the block's extreme pixel centres
            pass = !(t.bxlo > bb.y || t.bxhi < bb.x || t.bylo > bb.w || t.byhi < bb.z);
        }
        const unsigned long long m = __ballot(pass);
        if (t.lane == 0) s_wcnt[t.wave] = __popcll(m);
        __syncthreads();
        int base = count, tot = 0;
#pragma unroll
        for (int w = 0; w < BLK_THREADS / 64; ++w) {
            const int cw = s_wcnt[w];
            if (w < t.wave) base += cw;
            tot += cw;
        }
        if (pass) s_list[base + __popcll(m & ((1ull << t.lane) - 1ull))] = f;
        count += tot;
        __syncthreads();
    }
    return count;
}

// ------------------------------------------------------------------------------------------------
template <int RGB, bool P2F, bool TWO_SIDED>  // 0 = hard z-buffer colour (:408-416), 1 = soft-max over depth (:417-437),
                    // 2 = silhouette only: alpha plane, no depth / colour / p2f (soft_colors is then [N,IS,IS]),
                    // 3 = visibility only: the hard z-buffer's (depth, face id) planes, nothing else
// Register budget for 7 waves per SIMD: the default allocation (106 SGPRs) admits 6; the kernels are VALU-issue bound
// with every wave stalled ~50 % of its life, so the seventh wave pays (measured: 5 < 6 < 7 ~ 8 waves, -3..6 % time).
// (the forward variants without p2f accumulators fit 8 waves and gain another 2-5 %; with p2f 8 is slower)
#define FWD_WPE_ATTR __attribute__((amdgpu_waves_per_eu(P2F ? 7 : 8, P2F ? 7 : 8)))
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

    for (int f0 = 0; f0 < F; f0 += LIST_CAP) {
        const int f1 = min(F, f0 + LIST_CAP);
        if (f0 > 0) __syncthreads();
        const int count = build_list(s_list, s_wcnt, bbox_n, f0, f1, t);
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
                float wgt = 0.f;  // this lane's p2f weight for face f
                if (RGB == 3) {
                    // z-buffer winner only (:408-411): needs the bbox test, the barycentrics, the depth -- no distance.
                    // A pixel inside [0,1]^3 is never rejected by the distance threshold (inside: sign > 0; on the
                    // boundary: d = 0), except the defined-as-skip k = -1 case (no w <= 0 yet some w >= 1).
                    const float w0 = (fc.g<R_INV + 0>() * t.xp + fc.g<R_INV + 1>() * t.yp) + fc.g<R_INV + 2>();
                    const float w1 = (fc.g<R_INV + 3>() * t.xp + fc.g<R_INV + 4>() * t.yp) + fc.g<R_INV + 5>();
                    const float w2 = (fc.g<R_INV + 6>() * t.xp + fc.g<R_INV + 7>() * t.yp) + fc.g<R_INV + 8>();
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
                const bool live = eval_pair(p, fc, t.xp, t.yp, A.threshold, A.nis) & t.valid;
                if (RGB == 2) {
                    alpha *= live ? 1.f - p.frag : 1.f;
                    continue;
                }
                if (live) {
                    alpha *= 1.f - p.frag;  // 'prod' alpha (:396), BEFORE the depth-range test
                    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
                    const float zp = clip_depth(q0, q1, q2, p, fc);
                    if (!(zp < A.near_ || zp > A.far_)) {
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
    if (t.valid) {
        float *sc = A.soft_colors + (size_t)t.n * 4 * npix + pn;
        if (RGB == 1 || face_min != -1 || A.bg_arg) { sc[0] = o0; sc[npix] = o1; sc[2 * npix] = o2; }
        sc[3 * npix] = o3;
        float *ag = A.aggrs + (size_t)t.n * 2 * npix + pn;
        ag[0] = RGB == 0 ? depth_min : ssum;
        ag[npix] = RGB == 0 ? (float)face_min : smax;
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

    for (int f0 = 0; f0 < F; f0 += LIST_CAP) {
        const int f1 = min(F, f0 + LIST_CAP);
        if (f0 > 0) __syncthreads();
        const int count = build_list(s_list, s_wcnt, bbox_n, f0, f1, t);
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
#ifndef FM_TEXMERGE
#define FM_TEXMERGE 2   // DPP pre-merge steps before the LDS texel atomics: 0 none, 1 = x^1, 2 = x^1 then x^2
                       // (a third, vertical step measured slower)
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
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: uniform
    const int F = A.F, IS = A.IS, TS = A.TS;
    const bool pooled = COMMON ? true : (A.grad_pooled != 0);
    const bool two_sided = COMMON ? true : (A.double_side != 0);
    // XCD-aware: hardware XCD = blockIdx % 8.  Each XCD owns a fixed contiguous EIGHTH of every mesh's faces
    // (index-neighbouring faces of a subdivided mesh are spatial neighbours), so the per-pixel state its waves
    // re-read (~6x) covers 1/8 of the screen and stays in that XCD's 4 MB L2, and all 8 XCDs share every mesh
    // (balance at small N).  Measured fabric reads: 47 MB/mesh round-robin -> ~20 MB/mesh (11.8 MB algorithmic).
    const int fblocks = (F + FM_WAVES - 1) / FM_WAVES;   // blocks per mesh (grid = N * fblocks)
    int nb = blockIdx.x / fblocks, fb = blockIdx.x % fblocks;
    if (fblocks % 8 == 0 && (A.N * fblocks) % 8 == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = fblocks >> 3;
        nb = slot / per;
        fb = xcd * per + slot % per;
    }
    const int fidx = fb * FM_WAVES + wave;
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
    if (live) {
        Face fc;
        load_face(fc, A.rec + ((size_t)n * F + f) * REC);
        const float *__restrict__ tex_f = A.textures + ((size_t)(n / A.tex_group) * F + f) * TS * 3;
        // pixel-index window of the dilated bbox, widened by one pixel; the exact per-pixel reject of the
        // reference (:536) still runs inside eval_pair, so the window only has to be conservative.
        // xp(i) = (2i + 1 - IS)/IS  <=>  i = (xp*IS + IS - 1)/2
        const float h = 0.5f * IS;
        int x0 = (int)floorf(fc.g<R_XLO>() * h + h - 0.5f) - 1, x1 = (int)ceilf(fc.g<R_XHI>() * h + h - 0.5f) + 1;
        int yi0 = (int)floorf(fc.g<R_YLO>() * h + h - 0.5f) - 1, yi1 = (int)ceilf(fc.g<R_YHI>() * h + h - 0.5f) + 1;
        // NaN / inf bounds: comparisons below fail safe to the full image (the reference would visit all pixels)
        if (!(fc.g<R_XLO>() == fc.g<R_XLO>() && fc.g<R_XHI>() == fc.g<R_XHI>() && fc.g<R_YLO>() == fc.g<R_YLO>() && fc.g<R_YHI>() == fc.g<R_YHI>())) { x0 = 0; x1 = IS - 1; yi0 = 0; yi1 = IS - 1; }
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
            const float4 i0 = make_float4(fc.g<R_INV + 0>(), fc.g<R_INV + 1>(), fc.g<R_INV + 2>(), fc.g<R_INV + 3>());
            const float4 i1 = make_float4(fc.g<R_INV + 4>(), fc.g<R_INV + 5>(), fc.g<R_INV + 6>(), fc.g<R_INV + 7>());
            const float4 i2 = make_float4(fc.g<R_INV + 8>(), fc.g<R_K0>(), fc.g<R_K1>(), fc.g<R_K2>());
            const int sub = lane / (FM_TW * FM_TH), sl = lane % (FM_TW * FM_TH);   // sub-tile slot of this lane, lane in it
            for (int tb = 0; tb < ntiles; tb += 64) {
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
                }
                unsigned long long tm = __ballot(want);
                while (tm) {
                    // next FM_NQ wanted sub-tiles, one per lane group (groups past the last one idle this visit)
                    int mine = -1;
#pragma unroll
                    for (int qq = 0; qq < FM_NQ; ++qq) {
                        if (tm) {
                            const int tbit = __builtin_ctzll(tm);
                            tm &= tm - 1;
                            const int e = __builtin_amdgcn_readlane(tpk, tbit);
                            if (sub == qq) mine = e;
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
                    const int row = (mine >> 16) * FM_TH + sl / FM_TW;
                    const int xi = (mine & 0xffff) * FM_TW + sl % FM_TW;
                    if (xi >= IS || row >= IS) continue;
                    const float yp = ndc_coord_fast(IS - 1 - row, IS, inv_is, pow2);
                    const float xp = ndc_coord_fast(xi, IS, inv_is, pow2);
                    const unsigned pn4 = (unsigned)(row * IS + xi) * 4u;                       // byte offset in a full plane
                    const unsigned gp4 = pooled ? (unsigned)((row >> 1) * H2 + (xi >> 1)) * 4u : pn4;
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
                                const float zmin_f = fminf(fminf(fc.g<R_Z0>(), fc.g<R_Z1>()), fc.g<R_Z2>());
                                dead = RGB == 0 ? (float)f != smx
                                                : ((A.far_ - zmin_f) * A.r_range - smx) * A.inv_gamma < -89.f;
                            }
                        }
                        if ((RGB == 2 || !NEED_GF) && __all(dead)) continue;
                    }
                    Pair p;
                    if (!eval_pair(p, fc, xp, yp, A.threshold, A.nis)) continue;
                    if (RGB == 2) {  // silhouette: d alpha only (:584, :632-642); soft_colors/grad are [N,IS,IS] | [N,H,H]
                        if (!fc.depth_in_range()) {
                            float u0, u1, u2;
                            const float zq = clip_depth(u0, u1, u2, p, fc);
                            if (zq < A.near_ || zq > A.far_) continue;  // :592
                        }
                        const float ga = (pooled ? 0.25f : 1.f) * ld_u(gc_n, gp4);
                        const float oa = ld_u(sc_n, pn4);
                        float c_a = ga * ((1.f - oa) * __builtin_amdgcn_rcpf(fmaxf(1.f - p.frag, 1e-6f)));
                        c_a *= p.frag * (1.f - p.frag) * (-A.nis);
                        const float k2a = 2.f * p.sign * c_a;
                        const float a0 = k2a * p.b0, a1 = k2a * p.b1, a2 = k2a * p.b2;
                        gv[0] += a0 * p.dx; gv[1] += a0 * p.dy;
                        gv[3] += a1 * p.dx; gv[4] += a1 * p.dy;
                        gv[6] += a2 * p.dx; gv[7] += a2 * p.dy;
                        continue;
                    }
                    const float gscale = pooled ? 0.25f : 1.f;   // 2x2 mean pool: each fine pixel gets a quarter
                    const float g0 = gscale * ld_u(gc_n, gp4), g1 = gscale * ld_u(gc_n, gp4 + gps),
                                g2 = gscale * ld_u(gc_n, gp4 + 2 * gps);
                    const float g3 = NEED_GF ? gscale * ld_u(gc_n, gp4 + 3 * gps) : 0.f;
                    const float ssum = ld_u(ag_n, pn4), smax = ld_u(ag_n, pn4 + pst);
                    float c_xy = 0.f;
                    if (NEED_GF) c_xy = g3 * ((1.f - ld_u(sc_n, pn4 + 3 * pst)) * __builtin_amdgcn_rcpf(fmaxf(1.f - p.frag, 1e-6f)));  // :584
                    float q0, q1, q2;
                    const float zp = clip_depth(q0, q1, q2, p, fc);
                    if (zp < A.near_ || zp > A.far_) continue;  // :592
                    float gz0 = 0.f, gz1 = 0.f, gz2 = 0.f;
                    if (RGB == 0) {
                        if (NEED_GT && (float)f == smax) {  // :596
                            const int tix = texel_index(q0, q1, A.R);
                            if (TS == 1) { gt0 += g0; gt1 += g1; gt2 += g2; }
                            else texel_accumulate(my_tex, tix, g0, g1, g2, lane);
                        }
                    } else if (two_sided || fc.front()) {
                        const float zn = div_r(A.far_ - zp, A.far_ - A.near_, A.r_range);
                        const float ps = p.frag * __expf((zn - smax) * A.inv_gamma) * __builtin_amdgcn_rcpf(ssum);  // :608
                        const int tix = texel_index(q0, q1, A.R);
                        if (NEED_GT) {
                            if (TS == 1) { gt0 += ps * g0; gt1 += ps * g1; gt2 += ps * g2; }
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
                            const float c_z = -(c_rgb * A.inv_gamma * A.r_range) * zp * zp;  // :624
                            gz0 = c_z * q0 * fc.g<R_RZ0>() * fc.g<R_RZ0>();
                            gz1 = c_z * q1 * fc.g<R_RZ1>() * fc.g<R_RZ1>();
                            gz2 = c_z * q2 * fc.g<R_RZ2>() * fc.g<R_RZ2>();
                        }
                    }
                    if (NEED_GF) {
                        c_xy *= p.frag * (1.f - p.frag) * (-A.nis);  // :632
                        const float k2 = 2.f * p.sign * c_xy;        // :640
                        const float b0 = k2 * p.b0, b1 = k2 * p.b1, b2 = k2 * p.b2;
                        gv[0] += b0 * p.dx; gv[1] += b0 * p.dy; gv[2] += gz0;
                        gv[3] += b1 * p.dx; gv[4] += b1 * p.dy; gv[5] += gz1;
                        gv[6] += b2 * p.dx; gv[7] += b2 * p.dy; gv[8] += gz2;
                    }
                }
            }
        }
    }
    if (NEED_GF) {
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float sv = wave_sum_full(gv[k]);
            if (lane == k) mine = sv;
        }
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

bool modes_ok(int func_id_dist, int func_id_rgb, int func_id_alpha, int texture_sample_type, int TS, int *R) {
    if (func_id_dist != 2 || func_id_alpha != 2 || texture_sample_type != 0) return false;
    if (func_id_rgb != 0 && func_id_rgb != 1) return false;
    int r = 1;
    while (r * r < TS) ++r;
    if (r * r != TS) return false;
    *R = r;
    return true;
}

size_t ws_bbox_bytes(int N, int F) { return (((size_t)N * F * sizeof(float4)) + 255) & ~(size_t)255; }

// ---- optional per-kernel timing with library-owned HIP events (umr_profile_*) -------------------
struct ProfRec { hipEvent_t e0, e1; double bytes; int which; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
bool g_prof_on = false;
bool g_bwd_pixel_major = false;  // umr_debug_set("bwd_pixel_major", 1)

struct ProfScope {  // brackets exactly one kernel launch on `st`
    bool on; hipEvent_t e0, e1; hipStream_t st; int which; double bytes;
    ProfScope(hipStream_t s, int w, double b) : on(g_prof_on), st(s), which(w), bytes(b) {
        if (on) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, st); }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(e1, st);
            std::lock_guard<std::mutex> lk(g_prof_mu);
            g_prof.push_back({e0, e1, bytes, which});
        }
    }
};

}  // namespace

extern "C" {

const char *umr_version(void) { return "umr_hip 0.1 gfx950"; }

int umr_debug_set(const char *key, int value) {
    if (!key) return UMR_ERR_ARG;
    if (std::string(key) == "bwd_pixel_major") { g_bwd_pixel_major = value != 0; return UMR_OK; }
    return UMR_ERR_ARG;
}

int umr_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return UMR_OK;
}

int umr_profile_collect(int which, double *total_ms, long *launches, double *total_bytes) {
    if (!total_ms || !launches || !total_bytes) return UMR_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    *total_ms = 0.0; *launches = 0; *total_bytes = 0.0;
    std::vector<ProfRec> keep;
    for (const ProfRec &r : g_prof) {
        if (r.which != which) { keep.push_back(r); continue; }
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            *total_ms += ms; *launches += 1; *total_bytes += r.bytes;
        }
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    g_prof.swap(keep);
    return UMR_OK;
}

size_t umr_raster_workspace_bytes(int N, int F) {
    if (N <= 0 || F <= 0) return 0;
    return ws_bbox_bytes(N, F) + (size_t)N * F * REC * sizeof(float);
}

int umr_raster_forward(const float *faces, const float *textures, float *faces_info, float *aggrs_info,
                       const float *grid, float *p2f_info, float *p2f_sum, float *soft_colors,
                       float *pooled_out, int N, int F, int TS, int image_size, float near_, float far_,
                       float eps, float sigma_val, int func_id_dist, float dist_eps, float gamma_val,
                       int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side,
                       int flags, const float *background, void *workspace, size_t workspace_bytes,
                       void *stream) {
    int R = 0;
    const bool alpha_only = (flags & UMR_RASTER_ALPHA_ONLY) != 0;
    const bool ids_only = (flags & UMR_RASTER_FACE_ID_ONLY) != 0;
    const int tex_group = ((flags >> 8) & 0xff) ? ((flags >> 8) & 0xff) : 1;
    if (N > 0 && N % tex_group) return UMR_ERR_ARG;
    if (alpha_only && ids_only) return UMR_ERR_ARG;
    if (ids_only && (func_id_rgb != 0 || !aggrs_info)) return UMR_ERR_ARG;
    if (!faces || (!soft_colors && !ids_only) || !workspace) return UMR_ERR_ARG;
    if (!alpha_only && !ids_only && (!textures || !aggrs_info)) return UMR_ERR_ARG;
    if (N <= 0 || F <= 0 || TS <= 0 || image_size <= 0) return UMR_ERR_ARG;
    if (!modes_ok(func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type, TS, &R)) return UMR_ERR_ARG;
    if (workspace_bytes < umr_raster_workspace_bytes(N, F)) return UMR_ERR_ARG;
    const int with_p2f = func_id_rgb == 1 && !alpha_only && !(flags & UMR_RASTER_NO_P2F);
    if (with_p2f && (!grid || !p2f_info || !p2f_sum)) return UMR_ERR_ARG;
    if (pooled_out && (image_size & 1)) return UMR_ERR_ARG;
    if ((long long)N * ((image_size + BLK_W - 1) / BLK_W) * ((image_size + BLK_H - 1) / BLK_H) > 0x7fffffffLL)
        return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    RasterArgs A = {};
    A.bbox = (const float4 *)workspace;
    A.rec = (const float *)((char *)workspace + ws_bbox_bytes(N, F));
    A.textures = textures; A.grid = grid; A.aggrs = aggrs_info; A.p2f_info = p2f_info; A.p2f_sum = p2f_sum;
    A.soft_colors = soft_colors; A.pooled = pooled_out;
    A.N = N; A.F = F; A.IS = image_size; A.TS = TS; A.R = R;
    A.near_ = near_; A.far_ = far_; A.eps = eps; A.sigma = sigma_val;
    A.threshold = dist_eps * sigma_val;  // :332
    A.gamma = gamma_val; A.double_side = double_side; A.with_p2f = with_p2f; A.tex_group = tex_group;
    A.thr = sqrtf(A.threshold); A.nis = -1.f / sigma_val; A.r_range = 1.f / (far_ - near_); A.inv_gamma = 1.f / gamma_val;
    A.tiles_x = (image_size + BLK_W - 1) / BLK_W;
    A.tiles_y = (image_size + BLK_H - 1) / BLK_H;
    if (background) { A.bg_arg = 1; A.bg0 = background[0]; A.bg1 = background[1]; A.bg2 = background[2]; }
    const int total = N * F;
    k_face_setup<<<(total + 255) / 256, 256, 0, st>>>(faces, faces_info, (float4 *)workspace, (float *)A.rec, total,
                                                      sqrtf(A.threshold), near_, far_);
    const int blocks = N * A.tiles_x * A.tiles_y;
    {
        // algorithmic bytes of one forward launch: 24 IS^2 + F (36 + 12 TS + 16) per mesh
        // (silhouette-only launches are accounted separately, id 2: 4 IS^2 + 36 F)
        ProfScope ps(st, (alpha_only || ids_only) ? 2 : 0,
                     alpha_only ? (double)N * (4.0 * image_size * image_size + 36.0 * F)
                     : ids_only ? (double)N * (8.0 * image_size * image_size + 36.0 * F)
                                : (double)N * (24.0 * image_size * image_size + (double)F * (36.0 + 12.0 * TS + 16.0)));
        // p2f accumulation and face culling are compile-time: as run-time flags they cost SGPRs in every variant
        if (ids_only) {
            if (double_side) k_raster_forward<3, false, true><<<blocks, BLK_THREADS, 0, st>>>(A);
            else k_raster_forward<3, false, false><<<blocks, BLK_THREADS, 0, st>>>(A);
        } else if (alpha_only) {
            k_raster_forward<2, false, true><<<blocks, BLK_THREADS, 0, st>>>(A);   // alpha does not look at the side
        } else if (func_id_rgb == 0) {
            if (double_side) k_raster_forward<0, false, true><<<blocks, BLK_THREADS, 0, st>>>(A);
            else k_raster_forward<0, false, false><<<blocks, BLK_THREADS, 0, st>>>(A);
        } else if (with_p2f) {
            if (double_side) k_raster_forward<1, true, true><<<blocks, BLK_THREADS, 0, st>>>(A);
            else k_raster_forward<1, true, false><<<blocks, BLK_THREADS, 0, st>>>(A);
        } else {
            if (double_side) k_raster_forward<1, false, true><<<blocks, BLK_THREADS, 0, st>>>(A);
            else k_raster_forward<1, false, false><<<blocks, BLK_THREADS, 0, st>>>(A);
        }
    }
    return umr_launch_status();
}

int umr_raster_backward(const float *faces, const float *textures, const float *soft_colors,
                        const float *faces_info, const float *aggrs_info, float *grad_faces,
                        float *grad_textures, const float *grad_soft_colors, int grad_is_pooled,
                        int need_grad_faces, int need_grad_textures, int N, int F, int TS, int image_size,
                        float near_, float far_, float eps, float sigma_val, int func_id_dist,
                        float dist_eps, float gamma_val, int func_id_rgb, int func_id_alpha,
                        int texture_sample_type, int double_side, void *workspace, size_t workspace_bytes,
                        void *stream) {
    (void)faces_info;  // recomputed into the workspace (bit-identical: same kernel, same input)
    int R = 0;
    const bool alpha_only = (grad_is_pooled & UMR_BWD_ALPHA_ONLY) != 0;
    const int tex_group = ((grad_is_pooled >> 8) & 0xff) ? ((grad_is_pooled >> 8) & 0xff) : 1;
    grad_is_pooled &= UMR_BWD_GRAD_POOLED;
    if (N > 0 && N % tex_group) return UMR_ERR_ARG;
    if (!faces || !soft_colors || !grad_soft_colors || !workspace) return UMR_ERR_ARG;
    if (!alpha_only && (!textures || !aggrs_info)) return UMR_ERR_ARG;
    if (alpha_only && (need_grad_textures || !need_grad_faces)) return UMR_ERR_ARG;
    if ((need_grad_faces && !grad_faces) || (need_grad_textures && !grad_textures)) return UMR_ERR_ARG;
    if (N <= 0 || F <= 0 || TS <= 0 || image_size <= 0) return UMR_ERR_ARG;
    if (!modes_ok(func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type, TS, &R)) return UMR_ERR_ARG;
    if (workspace_bytes < umr_raster_workspace_bytes(N, F)) return UMR_ERR_ARG;
    if (grad_is_pooled && (image_size & 1)) return UMR_ERR_ARG;
    if (!need_grad_faces && !need_grad_textures) return UMR_OK;
    hipStream_t st = (hipStream_t)stream;
    RasterArgs A = {};
    A.bbox = (const float4 *)workspace;
    A.rec = (const float *)((char *)workspace + ws_bbox_bytes(N, F));
    A.textures = textures; A.aggrs = (float *)aggrs_info; A.soft_colors = (float *)soft_colors;
    A.grad_colors = grad_soft_colors; A.grad_faces = grad_faces; A.grad_textures = grad_textures;
    A.N = N; A.F = F; A.IS = image_size; A.TS = TS; A.R = R;
    A.near_ = near_; A.far_ = far_; A.eps = eps; A.sigma = sigma_val;
    A.threshold = dist_eps * sigma_val;
    A.gamma = gamma_val; A.double_side = double_side;
    A.thr = sqrtf(A.threshold); A.nis = -1.f / sigma_val; A.r_range = 1.f / (far_ - near_); A.inv_gamma = 1.f / gamma_val;
    A.grad_pooled = grad_is_pooled; A.need_gf = need_grad_faces; A.need_gt = need_grad_textures;
    A.tex_group = tex_group;
    A.tiles_x = (image_size + BLK_W - 1) / BLK_W;
    A.tiles_y = (image_size + BLK_H - 1) / BLK_H;
    const int total = N * F;
    k_face_setup<<<(total + 255) / 256, 256, 0, st>>>(faces, nullptr, (float4 *)workspace, (float *)A.rec, total,
                                                      sqrtf(A.threshold), near_, far_);
    const int blocks = N * A.tiles_x * A.tiles_y;
    {
        // algorithmic bytes of one backward launch: per mesh (40 | 28 when the gradient
        // arrives 2x2-pooled: 4 instead of 16 B/pixel) IS^2 + F (180 + 24 TS)
        const double px = grad_is_pooled ? 28.0 : 40.0;
        // (silhouette-only launches, id 3: alpha + its gradient per pixel, faces + grad_faces per face)
        ProfScope ps(st, alpha_only ? 3 : 1,
                     alpha_only ? (double)N * ((grad_is_pooled ? 5.0 : 8.0) * image_size * image_size + 72.0 * F)
                                : (double)N * (px * image_size * image_size + (double)F * (180.0 + 24.0 * TS)));
        const bool lds_ok = (size_t)FM_WAVES * FM_TEXCOPY * FM_TEX_STRIDE(TS) * sizeof(float) <= 48 * 1024;
        if (alpha_only) launch_backward_fm<2>(A, st);
        else if (g_bwd_pixel_major || !lds_ok) {  // pixel-major variant (global atomics); kept for A/B and huge TS
            if (func_id_rgb == 0) k_raster_backward<0><<<blocks, BLK_THREADS, 0, st>>>(A);
            else k_raster_backward<1><<<blocks, BLK_THREADS, 0, st>>>(A);
        } else if (func_id_rgb == 0) launch_backward_fm<0>(A, st);
        else launch_backward_fm<1>(A, st);
    }
    return umr_launch_status();
}

}  // extern "C"
