// perceptual.hip -- the distance head of the perceptual texture loss and the reductions of the part-matching loss,
// gfx950.  Both are byte-moving reductions over small feature / image planes (HBM/L2-bound, no MFMA on purpose); the
// dense AlexNet convolutions in front of the head stay on MIOpen.
//
//   PNet head:  external/PerceptualSimilarity/models/networks_basic.py:42-64 (sum over taps of 1 - cos_sim) with
//               util/util.py:71-83 (normalise over channels with eps = 1e-10, dot, mean over x then y).
//   part loss:  nnutils/loss_utils.py:399-440 (loss_type 'mse') + nnutils/scops_utils.py:12-54 (soft centroids).
#include "umr_common.h"

namespace {

constexpr int COS_MAX_TAPS = UMR_COS_MAX_TAPS;

struct CosTapDev { const float *f0, *f1; float *g0, *g1; float *stats; int C, P; };
struct CosArgs { CosTapDev tap[COS_MAX_TAPS]; int ntaps, N, chunks; float eps; int bwd_start[COS_MAX_TAPS + 1]; };

// Forward: a 256-thread block owns 64 consecutive pixels of sample n, tap t; lane = pixel (the channel loop reads
// f[n, c, p]: consecutive lanes consecutive p = coalesced 256-byte rows), wave w = channels w, w+4, w+8, ... so that the
// small late taps (15 x 15 pixels, 256-384 channels) still spread over 4 x 4 x N x 5 waves.  Per pixel:
// s00 = sum f0^2, s11 = sum f1^2, s01 = sum f0 f1 (partial sums of the 4 waves combined through LDS in a fixed order);
// cos = s01 / ((sqrt(s00) + eps)(sqrt(s11) + eps)).  stats[n, p] = (sqrt(s00), sqrt(s11), s01) for the backward.
__global__ __launch_bounds__(256) void k_cos_forward(const CosArgs A, float *__restrict__ partial) {
    __shared__ float s_part[4][3][64];
    const int t = blockIdx.z, n = blockIdx.y;
    const CosTapDev T = A.tap[t];
    const int C = T.C, P = T.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    float s00 = 0.f, s11 = 0.f, s01 = 0.f;
    if (p < P) {
        const float *a = T.f0 + (size_t)n * C * P + p, *b = T.f1 + (size_t)n * C * P + p;
        int c = wave;
        for (; c + 12 < C; c += 16) {       // four independent loads in flight per stream
            const float x0 = a[(size_t)c * P], x1 = a[(size_t)(c + 4) * P], x2 = a[(size_t)(c + 8) * P], x3 = a[(size_t)(c + 12) * P];
            const float y0 = b[(size_t)c * P], y1 = b[(size_t)(c + 4) * P], y2 = b[(size_t)(c + 8) * P], y3 = b[(size_t)(c + 12) * P];
            s00 = fmaf(x0, x0, s00); s00 = fmaf(x1, x1, s00); s00 = fmaf(x2, x2, s00); s00 = fmaf(x3, x3, s00);
            s11 = fmaf(y0, y0, s11); s11 = fmaf(y1, y1, s11); s11 = fmaf(y2, y2, s11); s11 = fmaf(y3, y3, s11);
            s01 = fmaf(x0, y0, s01); s01 = fmaf(x1, y1, s01); s01 = fmaf(x2, y2, s01); s01 = fmaf(x3, y3, s01);
        }
        for (; c < C; c += 4) {
            const float x = a[(size_t)c * P], y = b[(size_t)c * P];
            s00 = fmaf(x, x, s00); s11 = fmaf(y, y, s11); s01 = fmaf(x, y, s01);
        }
        UMR_TRAP_IF(umr_bad(s00) | umr_bad(s11), 30);
    }
    s_part[wave][0][lane] = s00; s_part[wave][1][lane] = s11; s_part[wave][2][lane] = s01;
    __syncthreads();
    if (wave == 0) {
        float cosv = 0.f;
        if (p < P) {
            s00 = ((s_part[0][0][lane] + s_part[1][0][lane]) + s_part[2][0][lane]) + s_part[3][0][lane];
            s11 = ((s_part[0][1][lane] + s_part[1][1][lane]) + s_part[2][1][lane]) + s_part[3][1][lane];
            s01 = ((s_part[0][2][lane] + s_part[1][2][lane]) + s_part[2][2][lane]) + s_part[3][2][lane];
            const float r0 = sqrtf(s00), r1 = sqrtf(s11);
            cosv = s01 / ((r0 + A.eps) * (r1 + A.eps));
            if (T.stats) {
                float *st = T.stats + ((size_t)n * P + p) * 3;
                st[0] = r0; st[1] = r1; st[2] = s01;
            }
        }
        const float s = wave_sum(cosv);
        if (lane == 0) partial[((size_t)t * A.N + n) * A.chunks + blockIdx.x] = s;
    }
}

// val[n] = sum_t (1 - (sum over chunks) / P_t): fixed summation order -> deterministic
__global__ void k_cos_finalize(const CosArgs A, const float *__restrict__ partial, float *__restrict__ val) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= A.N) return;
    float v = 0.f;
    for (int t = 0; t < A.ntaps; ++t) {
        float s = 0.f;
        const int used = (A.tap[t].P + 63) / 64;
        for (int k = 0; k < used; ++k) s += partial[((size_t)t * A.N + n) * A.chunks + k];
        v += 1.f - s / (float)A.tap[t].P;
    }
    UMR_TRAP_IF(umr_bad(v), 31);
    val[n] = v;
}

// d val[n] / d f1[c] at pixel p = -(1/P) [ f0[c] i0 i1 - s01 i0 i1^2 f1[c] / r1 ],  i = 1 / (r + eps)  (and symmetrically
// for f0).  A zero feature vector (r = 0) gets the derivative of its norm defined as 0: torch's sqrt backward gives
// 0 * inf = NaN there -- a defined deviation, listed in oracle/README.md.
// Pure element-wise map over [C, P] with three per-pixel coefficients.  A block owns 64 pixels x COS_BWD_CH channels of
// (sample, tap): lane = pixel (coalesced 256-byte rows), wave w = channels w, w+4, ... of the block's channel slab, all of
// a wave's loads issued before the first use.  (One block per 64 pixels looping over ALL channels -- 96 dependent
// load/store rounds for the 384-channel tap on 4 x N blocks -- ran 94 us per B = 16 step against ~30 us of HBM time.)
constexpr int COS_BWD_CH = 32;
__global__ __launch_bounds__(256) void k_cos_backward(const CosArgs A, const float *__restrict__ gval) {
    const int n = blockIdx.y;
    int t = 0;                                                      // blocks of all taps are laid out back to back
    while (t + 1 < A.ntaps && (int)blockIdx.x >= A.bwd_start[t + 1]) ++t;
    const CosTapDev T = A.tap[t];
    const int C = T.C, P = T.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pchunks = (P + 63) / 64, local = (int)blockIdx.x - A.bwd_start[t];
    const int p = (local % pchunks) * 64 + lane;
    const int c0 = (local / pchunks) * COS_BWD_CH;
    if (p >= P) return;
    const float g = -gval[n] / (float)P;
    const float *st = T.stats + ((size_t)n * P + p) * 3;
    const float r0 = st[0], r1 = st[1], s01 = st[2];
    const float i0 = 1.f / (r0 + A.eps), i1 = 1.f / (r1 + A.eps);
    const float k = g * i0 * i1;
    const float m0 = r0 > 0.f ? s01 * i0 / r0 : 0.f, m1 = r1 > 0.f ? s01 * i1 / r1 : 0.f;
    const size_t base = (size_t)n * C * P + p;
    const float *a = T.f0 + base, *b = T.f1 + base;
    float *g0 = T.g0 ? T.g0 + base : nullptr, *g1 = T.g1 ? T.g1 + base : nullptr;
    constexpr int PER = COS_BWD_CH / 4;
    float x[PER], y[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = c0 + wave + 4 * j;
        x[j] = c < C ? a[(size_t)c * P] : 0.f;
        y[j] = c < C ? b[(size_t)c * P] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = c0 + wave + 4 * j;
        if (c < C) {
            UMR_TRAP_IF(umr_bad(k * (y[j] - m0 * x[j])), 32);
            if (g0) g0[(size_t)c * P] = k * (y[j] - m0 * x[j]);
            if (g1) g1[(size_t)c * P] = k * (x[j] - m1 * y[j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ perceptual prologue
// Input side of the perceptual texture term, three reference steps in one pass per image stack:
//   loss_utils.py:141-146   x1 = img * mask            (mask [B,H,W] broadcast over the channels)
//   perceptual_loss.py:52-54  x2 = 2 * x1 - 1
//   networks_basic.py:45-46   y  = (x2 - shift_c) / scale_c
// each rounded to fp32 on its own, in that order (no contraction) -> the values torch's five element-wise kernels produce.
// Backward (autograd of the same chain): g2 = g_y / scale_c; g1 = 2 g2; g_img = g1 * mask; g_mask = sum_c g1 * img_c.
__global__ __launch_bounds__(256) void k_pp_forward(const float *__restrict__ img, const float *__restrict__ mask,
                                                    float *__restrict__ out, int C, unsigned HW, float3 shift, float3 scale) {
    const unsigned p = blockIdx.x * 256u + threadIdx.x;
    if (p >= HW) return;
    const size_t b = blockIdx.y;
    const float m = mask[b * HW + p];
    const float sh[3] = {shift.x, shift.y, shift.z}, sc[3] = {scale.x, scale.y, scale.z};
    for (int c = 0; c < C; ++c) {
        const size_t i = (b * C + c) * HW + p;
        const float x1 = img[i] * m;
        UMR_TRAP_IF(umr_bad(x1), 33);
        const float x2 = 2.f * x1 - 1.f;
        out[i] = (x2 - sh[c % 3]) / sc[c % 3];
    }
}

__global__ __launch_bounds__(256) void k_pp_backward(const float *__restrict__ gout, const float *__restrict__ img,
                                                     const float *__restrict__ mask, float *__restrict__ gimg,
                                                     float *__restrict__ gmask, int C, unsigned HW, float3 scale) {
    const unsigned p = blockIdx.x * 256u + threadIdx.x;
    if (p >= HW) return;
    const size_t b = blockIdx.y;
    const float m = mask[b * HW + p];
    const float sc[3] = {scale.x, scale.y, scale.z};
    float gm = 0.f;
    for (int c = 0; c < C; ++c) {
        const size_t i = (b * C + c) * HW + p;
        const float g1 = 2.f * (gout[i] / sc[c % 3]);
        UMR_TRAP_IF(umr_bad(g1), 34);
        if (gimg) gimg[i] = g1 * m;
        gm += g1 * img[i];
    }
    if (gmask) gmask[b * HW + p] = gm;
}

// ------------------------------------------------------------------------------------------------ part matching
// proj plane c (c = 1..4) of sample b: planes 1-3 = channels 0-2 of render A, plane 4 = channel 0 of render B (both
// [B,4,H,W], the pooled raster output); plane 0 is the constant background 0.1 (loss_utils.py:372-373, 399).
struct PartArgs {
    const float *pa, *pb;          // renders [B,4,H,W]
    const float *parts;            // [B,5,H,W]
    float *partial;                // [B, chunks, PART_NP]
    float *stats;                  // [B, PART_NS]
    float *partial2;               // [B, chunks, 5]: l_eqv partial, D_1..4
    float *ga, *gb;                // gradients [B,4,H,W] (zero-initialised by the caller)
    const float *g_eqv, *g_lm;     // upstream per-sample gradients [B]
    int B, H, W, chunks;
    float w[5];
    float eps_c;                   // 1e-3 of get_centers
    float bg;                      // 0.1
};
constexpr int PART_NP = 40;   // per-chunk partials: proj S/SX/SY (12) | parts S/SX/SY (12) | max proj 1..4 (4) | argmax (4) | max part 0..4 (5) | pad
constexpr int PART_NS = 40;   // per-sample: cp[4][2] (8) | cq[4][2] (8) | S_c (4) | mp[5] (5) | mq[5] (5) | argmax[4] (4) | unclamped[4] (4) | pad

__device__ __forceinline__ void softmax5(const float x[5], float sm[5]) {
    const float m = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), x[4]);
    float e[5], s = 0.f;
#pragma unroll
    for (int c = 0; c < 5; ++c) { e[c] = expf(x[c] - m); s += e[c]; }
#pragma unroll
    for (int c = 0; c < 5; ++c) sm[c] = e[c] / s;
}

__device__ __forceinline__ void load_proj(const PartArgs &A, int b, int p, float x[5]) {
    const size_t hw = (size_t)A.H * A.W;
    x[0] = A.bg;
    x[1] = A.pa[((size_t)b * 4 + 0) * hw + p];
    x[2] = A.pa[((size_t)b * 4 + 1) * hw + p];
    x[3] = A.pa[((size_t)b * 4 + 2) * hw + p];
    x[4] = A.pb[((size_t)b * 4 + 0) * hw + p];
}

// pass 1: soft-centroid sums of softmax(proj)[1:], softmax(parts)[1:] and the per-plane maxima, per pixel chunk
__global__ __launch_bounds__(256) void k_part_pass1(const PartArgs A) {
    __shared__ float smem[16];
    __shared__ float s_mx[4][4];
    __shared__ int s_ix[4][4];
    const int b = blockIdx.y, HW = A.H * A.W;
    const size_t hw = (size_t)HW;
    float sp[12], sq[12], mxp[4], mxq[5];
    int ixp[4];
#pragma unroll
    for (int k = 0; k < 12; ++k) { sp[k] = 0.f; sq[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { mxp[k] = -INFINITY; ixp[k] = 0x7fffffff; }
#pragma unroll
    for (int k = 0; k < 5; ++k) mxq[k] = -INFINITY;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
        // get_coordinate_tensors(h, w) is called as (x_max = h, y_max = w): x = col / H * 2 - 1, y = row / W * 2 - 1
        const int row = p / A.W, col = p - row * A.W;
        const float xm = (float)col / (float)A.H * 2.f - 1.f, ym = (float)row / (float)A.W * 2.f - 1.f;
        float x[5], q[5], sm[5];
        load_proj(A, b, p, x);
        softmax5(x, sm);
#pragma unroll
        for (int c = 1; c < 5; ++c) {
            const float v = sm[c] + A.eps_c;
            sp[(c - 1) * 3] += v; sp[(c - 1) * 3 + 1] += v * xm; sp[(c - 1) * 3 + 2] += v * ym;
            if (x[c] > mxp[c - 1]) { mxp[c - 1] = x[c]; ixp[c - 1] = p; }   // first maximum of this thread's ascending walk
        }
#pragma unroll
        for (int c = 0; c < 5; ++c) q[c] = A.parts[((size_t)b * 5 + c) * hw + p];
        softmax5(q, sm);
#pragma unroll
        for (int c = 1; c < 5; ++c) {
            const float v = sm[c] + A.eps_c;
            sq[(c - 1) * 3] += v; sq[(c - 1) * 3 + 1] += v * xm; sq[(c - 1) * 3 + 2] += v * ym;
        }
#pragma unroll
        for (int c = 0; c < 5; ++c) mxq[c] = fmaxf(mxq[c], q[c]);
    }
    float *out = A.partial + ((size_t)b * A.chunks + blockIdx.x) * PART_NP;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float r = block_sum(sp[k], smem);
        if (threadIdx.x == 0) out[k] = r;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float r = block_sum(sq[k], smem);
        if (threadIdx.x == 0) out[12 + k] = r;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // (max, smallest index attaining it): deterministic tie rule = lowest pixel index
        float m = mxp[k]; int ix = ixp[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(m, o, 64); const int oi = __shfl_xor(ix, o, 64);
            if (om > m || (om == m && oi < ix)) { m = om; ix = oi; }
        }
        if (lane == 0) { s_mx[k][wave] = m; s_ix[k][wave] = ix; }
    }
    float mq[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) mq[k] = wave_max(mxq[k]);
    __syncthreads();
    if (lane == 0) smem[wave] = 0.f;
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int k = 0; k < 4; ++k) {
            float m = s_mx[k][0]; int ix = s_ix[k][0];
            for (int w = 1; w < nw; ++w)
                if (s_mx[k][w] > m || (s_mx[k][w] == m && s_ix[k][w] < ix)) { m = s_mx[k][w]; ix = s_ix[k][w]; }
            out[24 + k] = m; out[28 + k] = __int_as_float(ix);
        }
    }
    __shared__ float s_q[5][4];
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 5; ++k) s_q[k][wave] = mq[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int k = 0; k < 5; ++k) {
            float m = s_q[k][0];
            for (int w = 1; w < nw; ++w) m = fmaxf(m, s_q[k][w]);
            out[32 + k] = m;
        }
    }
}

// per sample: centroids, clamped maxima, landmark term l_lm[b] = sum_{c,xy} (cp - cq)^2
__global__ void k_part_stats(const PartArgs A, float *__restrict__ l_lm) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.B) return;
    float sp[12], sq[12], mp[4], mq[5];
    int ix[4];
    for (int k = 0; k < 12; ++k) { sp[k] = 0.f; sq[k] = 0.f; }
    for (int k = 0; k < 4; ++k) { mp[k] = -INFINITY; ix[k] = 0x7fffffff; }
    for (int k = 0; k < 5; ++k) mq[k] = -INFINITY;
    for (int ch = 0; ch < A.chunks; ++ch) {
        const float *in = A.partial + ((size_t)b * A.chunks + ch) * PART_NP;
        for (int k = 0; k < 12; ++k) { sp[k] += in[k]; sq[k] += in[12 + k]; }
        for (int k = 0; k < 4; ++k) {
            const float m = in[24 + k]; const int i = __float_as_int(in[28 + k]);
            if (m > mp[k] || (m == mp[k] && i < ix[k])) { mp[k] = m; ix[k] = i; }
        }
        for (int k = 0; k < 5; ++k) mq[k] = fmaxf(mq[k], in[32 + k]);
    }
    float *st = A.stats + (size_t)b * PART_NS;
    float lm = 0.f;
    for (int c = 0; c < 4; ++c) {
        const float cpx = sp[c * 3 + 1] / sp[c * 3], cpy = sp[c * 3 + 2] / sp[c * 3];
        const float cqx = sq[c * 3 + 1] / sq[c * 3], cqy = sq[c * 3 + 2] / sq[c * 3];
        st[c * 2] = cpx; st[c * 2 + 1] = cpy; st[8 + c * 2] = cqx; st[8 + c * 2 + 1] = cqy;
        st[16 + c] = sp[c * 3];
        lm += (cpx - cqx) * (cpx - cqx) + (cpy - cqy) * (cpy - cqy);
    }
    // max_proj[max_proj < 1e-5] = 1e-5 (:418-419): the replaced entries carry no gradient
    st[20] = fmaxf(A.bg, 1e-5f);
    for (int c = 0; c < 4; ++c) {
        st[21 + c] = mp[c] < 1e-5f ? 1e-5f : mp[c];
        st[30 + c] = __int_as_float(ix[c]);
        st[34 + c] = mp[c] < 1e-5f ? 0.f : 1.f;
    }
    for (int c = 0; c < 5; ++c) st[25 + c] = mq[c] < 1e-5f ? 1e-5f : mq[c];
    l_lm[b] = lm;
}

// pass 2: l_eqv partial = sum_p sum_c w_c (p_c / mp_c - q_c / mq_c)^2 and D_c = sum_p 2 w_c (pn_c - qn_c) p_c
__global__ __launch_bounds__(256) void k_part_pass2(const PartArgs A) {
    __shared__ float smem[16];
    const int b = blockIdx.y, HW = A.H * A.W;
    const size_t hw = (size_t)HW;
    const float *st = A.stats + (size_t)b * PART_NS;
    float imp[5], imq[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) { imp[c] = st[20 + c]; imq[c] = st[25 + c]; }
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
        float x[5];
        load_proj(A, b, p, x);
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const float q = A.parts[((size_t)b * 5 + c) * hw + p];
            const float d = x[c] / imp[c] - q / imq[c];
            acc[0] += A.w[c] * d * d;
            if (c > 0) acc[c] += 2.f * A.w[c] * d * x[c];
        }
    }
    float *out = A.partial2 + ((size_t)b * A.chunks + blockIdx.x) * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float r = block_sum(acc[k], smem);
        if (threadIdx.x == 0) out[k] = r;
    }
}

__global__ void k_part_sum2(const PartArgs A, float *__restrict__ l_eqv) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.B) return;
    float s = 0.f;
    for (int ch = 0; ch < A.chunks; ++ch) s += A.partial2[((size_t)b * A.chunks + ch) * 5];
    l_eqv[b] = s;
}

// gradient wrt the four rendered part planes
__global__ __launch_bounds__(256) void k_part_backward(const PartArgs A) {
    const int b = blockIdx.y, HW = A.H * A.W;
    const size_t hw = (size_t)HW;
    const float *st = A.stats + (size_t)b * PART_NS;
    const float ge = A.g_eqv[b], gl = A.g_lm[b];
    float mp[5], mq[5], hx[4], hy[4], cx[4], cy[4], isum[4], dmax[4];
    int amax[4];
#pragma unroll
    for (int c = 0; c < 5; ++c) { mp[c] = st[20 + c]; mq[c] = st[25 + c]; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cx[c] = st[c * 2]; cy[c] = st[c * 2 + 1];
        hx[c] = 2.f * gl * (cx[c] - st[8 + c * 2]); hy[c] = 2.f * gl * (cy[c] - st[8 + c * 2 + 1]);
        isum[c] = 1.f / st[16 + c];
        amax[c] = __float_as_int(st[30 + c]);
        float D = 0.f;
        for (int ch = 0; ch < A.chunks; ++ch) D += A.partial2[((size_t)b * A.chunks + ch) * 5 + 1 + c];
        dmax[c] = st[34 + c] != 0.f ? -ge * D / (mp[c + 1] * mp[c + 1]) : 0.f;   // d/d max_proj, routed to the arg-max pixel
    }
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
        const int row = p / A.W, col = p - row * A.W;
        const float xm = (float)col / (float)A.H * 2.f - 1.f, ym = (float)row / (float)A.W * 2.f - 1.f;
        float x[5], sm[5], h[5];
        load_proj(A, b, p, x);
        softmax5(x, sm);
        h[0] = 0.f;
        float dot = 0.f;
#pragma unroll
        for (int c = 1; c < 5; ++c) {
            h[c] = (hx[c - 1] * (xm - cx[c - 1]) + hy[c - 1] * (ym - cy[c - 1])) * isum[c - 1];
            dot += sm[c] * h[c];
        }
        float g[5];
#pragma unroll
        for (int c = 1; c < 5; ++c) {
            const float q = A.parts[((size_t)b * 5 + c) * hw + p];
            const float d = x[c] / mp[c] - q / mq[c];
            g[c] = sm[c] * (h[c] - dot) + ge * A.w[c] * 2.f * d / mp[c] + (p == amax[c - 1] ? dmax[c - 1] : 0.f);
        }
        A.ga[((size_t)b * 4 + 0) * hw + p] = g[1];
        A.ga[((size_t)b * 4 + 1) * hw + p] = g[2];
        A.ga[((size_t)b * 4 + 2) * hw + p] = g[3];
        A.gb[((size_t)b * 4 + 0) * hw + p] = g[4];
    }
}

int part_chunks(int HW) { return max(1, min(64, (HW + 1023) / 1024)); }

}  // namespace

UMR_TRAP_ACCESSOR(umr_trap_read_perceptual)

extern "C" {

size_t umr_cos_sim_workspace_bytes(int ntaps, int N, const int *P) {
    if (ntaps <= 0 || ntaps > COS_MAX_TAPS || N <= 0 || !P) return 0;
    size_t fl = (size_t)ntaps * N * UMR_COS_CHUNKS;
    for (int t = 0; t < ntaps; ++t) fl += (size_t)N * P[t] * 3;
    return fl * sizeof(float);
}

int umr_cos_sim_forward(int ntaps, const float *const *f0, const float *const *f1, const int *C, const int *P, int N,
                        float eps, float *val, void *workspace, size_t workspace_bytes, void *stream) {
    if (ntaps <= 0 || ntaps > COS_MAX_TAPS || !f0 || !f1 || !C || !P || N <= 0 || !val || !workspace) return UMR_ERR_ARG;
    if (workspace_bytes < umr_cos_sim_workspace_bytes(ntaps, N, P)) return UMR_ERR_ARG;
    CosArgs A = {};
    A.ntaps = ntaps; A.N = N; A.chunks = UMR_COS_CHUNKS; A.eps = eps;
    float *ws = (float *)workspace;
    float *partial = ws;
    ws += (size_t)ntaps * N * UMR_COS_CHUNKS;
    for (int t = 0; t < ntaps; ++t) {
        if (!f0[t] || !f1[t] || C[t] <= 0 || P[t] <= 0) return UMR_ERR_ARG;
        A.tap[t].f0 = f0[t]; A.tap[t].f1 = f1[t]; A.tap[t].C = C[t]; A.tap[t].P = P[t]; A.tap[t].stats = ws;
        ws += (size_t)N * P[t] * 3;
    }
    hipStream_t st = (hipStream_t)stream;
    int maxp = 0;
    for (int t = 0; t < ntaps; ++t) maxp = P[t] > maxp ? P[t] : maxp;
    if ((maxp + 63) / 64 > UMR_COS_CHUNKS) return UMR_ERR_ARG;     // feature maps of up to 64 * UMR_COS_CHUNKS pixels
    UMR_LAUNCH(k_cos_forward, dim3((maxp + 63) / 64, N, ntaps), 256, 0, st, A, partial);
    UMR_LAUNCH(k_cos_finalize, (N + 63) / 64, 64, 0, st, A, partial, val);
    return umr_launch_status();
}

int umr_cos_sim_backward(int ntaps, const float *const *f0, const float *const *f1, float *const *g0, float *const *g1,
                         const int *C, const int *P, int N, float eps, const float *grad_val, const void *workspace,
                         size_t workspace_bytes, void *stream) {
    if (ntaps <= 0 || ntaps > COS_MAX_TAPS || !f0 || !f1 || !C || !P || N <= 0 || !grad_val || !workspace) return UMR_ERR_ARG;
    if (workspace_bytes < umr_cos_sim_workspace_bytes(ntaps, N, P)) return UMR_ERR_ARG;
    CosArgs A = {};
    A.ntaps = ntaps; A.N = N; A.chunks = UMR_COS_CHUNKS; A.eps = eps;
    float *ws = (float *)workspace + (size_t)ntaps * N * UMR_COS_CHUNKS;
    for (int t = 0; t < ntaps; ++t) {
        if (!f0[t] || !f1[t] || C[t] <= 0 || P[t] <= 0) return UMR_ERR_ARG;
        A.tap[t].f0 = f0[t]; A.tap[t].f1 = f1[t]; A.tap[t].C = C[t]; A.tap[t].P = P[t]; A.tap[t].stats = ws;
        A.tap[t].g0 = g0 ? g0[t] : nullptr; A.tap[t].g1 = g1 ? g1[t] : nullptr;
        ws += (size_t)N * P[t] * 3;
    }
    int maxp = 0;
    for (int t = 0; t < ntaps; ++t) maxp = P[t] > maxp ? P[t] : maxp;
    if ((maxp + 63) / 64 > UMR_COS_CHUNKS) return UMR_ERR_ARG;
    A.bwd_start[0] = 0;
    for (int t = 0; t < ntaps; ++t)
        A.bwd_start[t + 1] = A.bwd_start[t] + ((P[t] + 63) / 64) * ((C[t] + COS_BWD_CH - 1) / COS_BWD_CH);
    UMR_LAUNCH(k_cos_backward, dim3(A.bwd_start[ntaps], N), 256, 0, (hipStream_t)stream, A, grad_val);
    return umr_launch_status();
}

int umr_perceptual_prologue_forward(const float *img, const float *mask, float *out, int B, int C, long HW,
                                    const float *shift3, const float *scale3, void *stream) {
    if (!img || !mask || !out || !shift3 || !scale3 || B <= 0 || C <= 0 || C > 3 || HW <= 0 || HW > 0x7fffffffL || B > 65535)
        return UMR_ERR_ARG;
    UMR_LAUNCH(k_pp_forward, dim3((unsigned)((HW + 255) / 256), (unsigned)B), 256, 0, (hipStream_t)stream,
        img, mask, out, C, (unsigned)HW, make_float3(shift3[0], shift3[1], shift3[2]), make_float3(scale3[0], scale3[1], scale3[2]));
    return umr_launch_status();
}

int umr_perceptual_prologue_backward(const float *grad_out, const float *img, const float *mask, float *grad_img,
                                     float *grad_mask, int B, int C, long HW, const float *scale3, void *stream) {
    if (!grad_out || !img || !mask || !scale3 || B <= 0 || C <= 0 || C > 3 || HW <= 0 || HW > 0x7fffffffL || B > 65535)
        return UMR_ERR_ARG;
    if (!grad_img && !grad_mask) return UMR_OK;
    UMR_LAUNCH(k_pp_backward, dim3((unsigned)((HW + 255) / 256), (unsigned)B), 256, 0, (hipStream_t)stream,
        grad_out, img, mask, grad_img, grad_mask, C, (unsigned)HW, make_float3(scale3[0], scale3[1], scale3[2]));
    return umr_launch_status();
}

size_t umr_part_match_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int ch = part_chunks(H * W);
    return ((size_t)B * ch * PART_NP + (size_t)B * PART_NS + (size_t)B * ch * 5) * sizeof(float);
}

static int part_args(PartArgs &A, const float *pa, const float *pb, const float *parts, int B, int H, int W,
                     const float *weights5, float bg, float eps_c, void *workspace, size_t workspace_bytes) {
    if (!pa || !pb || !parts || !weights5 || !workspace || B <= 0 || H <= 0 || W <= 0) return UMR_ERR_ARG;
    if ((long long)H * W > 0x3fffffffLL || workspace_bytes < umr_part_match_workspace_bytes(B, H, W)) return UMR_ERR_ARG;
    A = PartArgs{};
    A.pa = pa; A.pb = pb; A.parts = parts; A.B = B; A.H = H; A.W = W; A.chunks = part_chunks(H * W);
    for (int c = 0; c < 5; ++c) A.w[c] = weights5[c];
    A.bg = bg; A.eps_c = eps_c;
    float *ws = (float *)workspace;
    A.partial = ws; ws += (size_t)B * A.chunks * PART_NP;
    A.stats = ws; ws += (size_t)B * PART_NS;
    A.partial2 = ws;
    return UMR_OK;
}

int umr_part_match_forward(const float *render_a, const float *render_b, const float *part_segs, int B, int H, int W,
                           const float *weights5, float background, float center_eps, float *l_eqv, float *l_lm,
                           void *workspace, size_t workspace_bytes, void *stream) {
    PartArgs A;
    if (!l_eqv || !l_lm) return UMR_ERR_ARG;
    const int rc = part_args(A, render_a, render_b, part_segs, B, H, W, weights5, background, center_eps, workspace,
                             workspace_bytes);
    if (rc != UMR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    UMR_LAUNCH(k_part_pass1, dim3(A.chunks, B), 256, 0, st, A);
    UMR_LAUNCH(k_part_stats, (B + 63) / 64, 64, 0, st, A, l_lm);
    UMR_LAUNCH(k_part_pass2, dim3(A.chunks, B), 256, 0, st, A);
    UMR_LAUNCH(k_part_sum2, (B + 63) / 64, 64, 0, st, A, l_eqv);
    return umr_launch_status();
}

int umr_part_match_backward(const float *render_a, const float *render_b, const float *part_segs, int B, int H, int W,
                            const float *weights5, float background, float center_eps, const float *grad_l_eqv,
                            const float *grad_l_lm, float *grad_render_a, float *grad_render_b, const void *workspace,
                            size_t workspace_bytes, void *stream) {
    PartArgs A;
    if (!grad_l_eqv || !grad_l_lm || !grad_render_a || !grad_render_b) return UMR_ERR_ARG;
    const int rc = part_args(A, render_a, render_b, part_segs, B, H, W, weights5, background, center_eps,
                             (void *)workspace, workspace_bytes);
    if (rc != UMR_OK) return rc;
    A.g_eqv = grad_l_eqv; A.g_lm = grad_l_lm; A.ga = grad_render_a; A.gb = grad_render_b;
    UMR_LAUNCH(k_part_backward, dim3(A.chunks, B), 256, 0, (hipStream_t)stream, A);
    return umr_launch_status();
}

}  // extern "C"
