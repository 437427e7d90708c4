// regs.hip -- the small regularisers and the masked L1 image term of the render-and-compare loop, gfx950.
//   deform_l2reg         nnutils/loss_utils.py:118-123   mean over rows of ||V_row||_2
//   sym_reg              nnutils/loss_utils.py:125-126   mean |verts[:, :, 1]|
//   texture_loss_masks   nnutils/loss_utils.py:103-116   L1(img_pred * mask_pred, img_gt * mask_gt), mean or per sample
// Byte-moving reductions (HBM / launch-latency bound): two stages, no float atomics -- per-block partial sums, then one thread
// adds them in index order, so every value is bit-reproducible run to run.  The backward kernels are element-wise maps.
#include "umr_common.h"

namespace {

#define REG_PER_BLOCK 4096   // elements (rows) per partial-sum block

__host__ __device__ inline int reg_blocks(long n) { return (int)((n + REG_PER_BLOCK - 1) / REG_PER_BLOCK); }

// mode 0: rows of `width` floats -> sum of row L2 norms; mode 1: sum of |x[i * stride + offset]|
template <int MODE>
__global__ __launch_bounds__(256) void k_reg_partial(const float *__restrict__ x, float *__restrict__ partial, long rows, int width,
                                                     int offset) {
    __shared__ float smem[16];
    const long start = (long)blockIdx.x * REG_PER_BLOCK, end = min(rows, start + REG_PER_BLOCK);
    float s = 0.f;
    for (long i = start + threadIdx.x; i < end; i += blockDim.x) {
        const float *r = x + (size_t)i * width;
        if (MODE == 0) {
            float q = 0.f;
            for (int k = 0; k < width; ++k) q += r[k] * r[k];
            s += sqrtf(q);
        } else {
            s += fabsf(r[offset]);
        }
    }
    const float t = block_sum(s, smem);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void k_reg_finalize(const float *__restrict__ partial, float *__restrict__ out, int nb, float scale) {
    if (threadIdx.x || blockIdx.x) return;
    float s = 0.f;
    for (int k = 0; k < nb; ++k) s += partial[k];
    out[0] = s * scale;
}

template <int MODE>
__global__ void k_reg_backward(const float *__restrict__ x, const float *__restrict__ g, float *__restrict__ gx, long rows, int width,
                               int offset, float scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const float *r = x + (size_t)i * width;
    float *o = gx + (size_t)i * width;
    const float gs = g[0] * scale;
    if (MODE == 0) {
        float q = 0.f;
        for (int k = 0; k < width; ++k) q += r[k] * r[k];
        const float nrm = sqrtf(q);
        const float c = nrm > 0.f ? gs / nrm : 0.f;       // torch's norm backward: zero (sub)gradient at the origin
        for (int k = 0; k < width; ++k) o[k] = c * r[k];
    } else {
        for (int k = 0; k < width; ++k) o[k] = 0.f;
        const float v = r[offset];
        o[offset] = v > 0.f ? gs : (v < 0.f ? -gs : 0.f);   // sign(0) = 0, as torch.abs backward
    }
}

// ---- masked L1 (loss_utils.py:103-116): d = ip * mp - ig * mg over [B, C, HW]; per-sample partial sums ------------------
__global__ __launch_bounds__(256) void k_ml1_partial(const float *__restrict__ ip, const float *__restrict__ ig,
                                                     const float *__restrict__ mg, const float *__restrict__ mp,
                                                     float *__restrict__ partial, int C, long HW) {
    __shared__ float smem[16];
    const int b = blockIdx.y, nb = gridDim.x;
    const long n = (long)C * HW, start = (long)blockIdx.x * REG_PER_BLOCK, end = min(n, start + REG_PER_BLOCK);
    float s = 0.f;
    for (long i = start + threadIdx.x; i < end; i += blockDim.x) {
        const long p = i % HW;
        const size_t e = (size_t)b * n + i, m = (size_t)b * HW + p;
        s += fabsf(ip[e] * mp[m] - ig[e] * mg[m]);
    }
    const float t = block_sum(s, smem);
    if (threadIdx.x == 0) partial[(size_t)b * nb + blockIdx.x] = t;
}

__global__ void k_ml1_finalize(const float *__restrict__ partial, float *__restrict__ per_sample, int B, int nb, float scale) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int k = 0; k < nb; ++k) s += partial[(size_t)b * nb + k];
    per_sample[b] = s * scale;
}

// grad_ip[b,c,p] = g[b] scale sign(d) mp[b,p];  grad_mp[b,p] = sum_c g[b] scale sign(d) ip[b,c,p]   (one thread per pixel)
__global__ void k_ml1_backward(const float *__restrict__ ip, const float *__restrict__ ig, const float *__restrict__ mg,
                               const float *__restrict__ mp, const float *__restrict__ g, float *__restrict__ gip,
                               float *__restrict__ gmp, int C, long HW, float scale) {
    const int b = blockIdx.y;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float gs = g[b] * scale, vmp = mp[(size_t)b * HW + p], vmg = mg[(size_t)b * HW + p];
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
        const size_t e = ((size_t)b * C + c) * HW + p;
        const float d = ip[e] * vmp - ig[e] * vmg;
        const float sg = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
        if (gip) gip[e] = sg * vmp;
        acc += sg * ip[e];
    }
    if (gmp) gmp[(size_t)b * HW + p] = acc;
}

}  // namespace

extern "C" {

/* partial-sum scratch a caller passes to the three forward entry points, in floats */
long umr_reg_scratch_floats(long elements, int batch) { return (long)reg_blocks(elements) * (batch > 0 ? batch : 1); }

int umr_row_norm_mean_forward(const float *x, float *out, float *scratch, size_t scratch_bytes, long rows, int width, void *stream) {
    if (!x || !out || !scratch || rows <= 0 || width <= 0) return UMR_ERR_ARG;
    const int nb = reg_blocks(rows);
    if (scratch_bytes < (size_t)nb * sizeof(float)) return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    UMR_LAUNCH((k_reg_partial<0>), nb, 256, 0, st, x, scratch, rows, width, 0);
    UMR_LAUNCH(k_reg_finalize, 1, 64, 0, st, scratch, out, nb, 1.f / (float)rows);
    return umr_launch_status();
}

int umr_row_norm_mean_backward(const float *x, const float *grad_out, float *grad_x, long rows, int width, void *stream) {
    if (!x || !grad_out || !grad_x || rows <= 0 || width <= 0) return UMR_ERR_ARG;
    UMR_LAUNCH((k_reg_backward<0>), (unsigned)((rows + 255) / 256), 256, 0, (hipStream_t)stream, x, grad_out, grad_x, rows, width, 0, 1.f / (float)rows);
    return umr_launch_status();
}

int umr_abs_mean_forward(const float *x, float *out, float *scratch, size_t scratch_bytes, long rows, int width, int column, void *stream) {
    if (!x || !out || !scratch || rows <= 0 || width <= 0 || column < 0 || column >= width) return UMR_ERR_ARG;
    const int nb = reg_blocks(rows);
    if (scratch_bytes < (size_t)nb * sizeof(float)) return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    UMR_LAUNCH((k_reg_partial<1>), nb, 256, 0, st, x, scratch, rows, width, column);
    UMR_LAUNCH(k_reg_finalize, 1, 64, 0, st, scratch, out, nb, 1.f / (float)rows);
    return umr_launch_status();
}

int umr_abs_mean_backward(const float *x, const float *grad_out, float *grad_x, long rows, int width, int column, void *stream) {
    if (!x || !grad_out || !grad_x || rows <= 0 || width <= 0 || column < 0 || column >= width) return UMR_ERR_ARG;
    UMR_LAUNCH((k_reg_backward<1>), (unsigned)((rows + 255) / 256), 256, 0, (hipStream_t)stream, x, grad_out, grad_x, rows, width, column,
                                                                                       1.f / (float)rows);
    return umr_launch_status();
}

int umr_masked_l1_forward(const float *img_pred, const float *img_gt, const float *mask_gt, const float *mask_pred, float *per_sample,
                          float *scratch, size_t scratch_bytes, int B, int C, long HW, void *stream) {
    if (!img_pred || !img_gt || !mask_gt || !mask_pred || !per_sample || !scratch || B <= 0 || C <= 0 || HW <= 0) return UMR_ERR_ARG;
    const int nb = reg_blocks((long)C * HW);
    if (scratch_bytes < (size_t)nb * B * sizeof(float)) return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    UMR_LAUNCH(k_ml1_partial, dim3(nb, B), 256, 0, st, img_pred, img_gt, mask_gt, mask_pred, scratch, C, HW);
    UMR_LAUNCH(k_ml1_finalize, (B + 63) / 64, 64, 0, st, scratch, per_sample, B, nb, 1.f / ((float)C * (float)HW));
    return umr_launch_status();
}

int umr_masked_l1_backward(const float *img_pred, const float *img_gt, const float *mask_gt, const float *mask_pred,
                           const float *grad_per_sample, float *grad_img_pred, float *grad_mask_pred, int B, int C, long HW,
                           void *stream) {
    if (!img_pred || !img_gt || !mask_gt || !mask_pred || !grad_per_sample || B <= 0 || C <= 0 || HW <= 0) return UMR_ERR_ARG;
    if (!grad_img_pred && !grad_mask_pred) return UMR_OK;
    UMR_LAUNCH(k_ml1_backward, dim3((unsigned)((HW + 255) / 256), B), 256, 0, (hipStream_t)stream,
        img_pred, img_gt, mask_gt, mask_pred, grad_per_sample, grad_img_pred, grad_mask_pred, C, HW, 1.f / ((float)C * (float)HW));
    return umr_launch_status();
}

}  // extern "C"
