// edt.hip -- barrier distance transform on the GPU.
//
// Replaces utils/image.py:130-141 (compute_dt_barrier), which the reference runs on the HOST for every image of
// every step (experiments/train_s1.py:171-174, train_s2.py:196): two scipy exact Euclidean distance transforms,
//   dist = sigmoid(k * (EDT(1 - mask) - EDT(mask)) / max(H, W)).
// Exact EDT by separability: pass 1 = nearest feature along each column (integer), pass 2 = lower envelope over
// the row, d^2(x, y) = min_x' (x - x')^2 + g(x', y)^2, evaluated by brute force (W <= 1024; 256 for UMR) in
// INTEGER arithmetic -- squared distances are bit-exact with scipy; only the final sqrt/sigmoid is float.
#include "umr_common.h"

namespace {

#define EDT_INF 30000   // > any in-image distance; EDT_INF^2 + W^2 < 2^31

// one thread per column; features of EDT(1-mask) are mask != 0, features of EDT(mask) are mask == 0.
// The sweep is serial in y, so rows are fetched in batches of 16 independent (coalesced-across-columns) loads:
// one memory latency per 16 rows instead of one per row (314 us -> tens of us for 16 x 256^2).
#define EDT_BATCH 16
__global__ void k_edt_columns(const float *__restrict__ mask, int *__restrict__ g, int H, int W) {
    const int b = blockIdx.y, x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    const float *__restrict__ m = mask + (size_t)b * H * W + x;
    int *__restrict__ go = g + (size_t)b * 2 * H * W + x;
    int *__restrict__ gi = go + (size_t)H * W;
    int dofg = EDT_INF, dobg = EDT_INF;  // distance to the last foreground / background pixel seen
    for (int y0 = 0; y0 < H; y0 += EDT_BATCH) {
        float mv[EDT_BATCH];
#pragma unroll
        for (int j = 0; j < EDT_BATCH; ++j) mv[j] = (y0 + j < H) ? m[(size_t)(y0 + j) * W] : 0.f;
#pragma unroll
        for (int j = 0; j < EDT_BATCH; ++j) {
            if (y0 + j < H) {
                const bool fg = mv[j] != 0.f;
                dofg = fg ? 0 : min(dofg + 1, EDT_INF);
                dobg = fg ? min(dobg + 1, EDT_INF) : 0;
                go[(size_t)(y0 + j) * W] = dofg;
                gi[(size_t)(y0 + j) * W] = dobg;
            }
        }
    }
    dofg = dobg = EDT_INF;
    for (int y1 = H - 1; y1 >= 0; y1 -= EDT_BATCH) {
        float mv[EDT_BATCH];
        int vo[EDT_BATCH], vi[EDT_BATCH];
#pragma unroll
        for (int j = 0; j < EDT_BATCH; ++j) {
            const int y = y1 - j;
            mv[j] = y >= 0 ? m[(size_t)y * W] : 0.f;
            vo[j] = y >= 0 ? go[(size_t)y * W] : 0;
            vi[j] = y >= 0 ? gi[(size_t)y * W] : 0;
        }
#pragma unroll
        for (int j = 0; j < EDT_BATCH; ++j) {
            const int y = y1 - j;
            if (y >= 0) {
                const bool fg = mv[j] != 0.f;
                dofg = fg ? 0 : min(dofg + 1, EDT_INF);
                dobg = fg ? min(dobg + 1, EDT_INF) : 0;
                go[(size_t)y * W] = min(vo[j], dofg);
                gi[(size_t)y * W] = min(vi[j], dobg);
            }
        }
    }
}

// one block per image row; LDS holds the row's squared column distances for both transforms
__global__ void k_edt_rows(const int *__restrict__ g, float *__restrict__ out, int *__restrict__ sq_out,
                           int *__restrict__ sq_in, int H, int W, float k, float inv_max) {
    extern __shared__ int s_g[];  // [2][W]
    const int b = blockIdx.y, y = blockIdx.x;
    const int *go = g + ((size_t)b * 2 * H + y) * W, *gi = go + (size_t)H * W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const int a = go[x], c = gi[x];
        s_g[x] = a * a;
        s_g[W + x] = c * c;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        int bo = 0x7fffffff, bi = 0x7fffffff;
        for (int xp = 0; xp < W; ++xp) {
            const int d = (x - xp) * (x - xp);
            bo = min(bo, d + s_g[xp]);
            bi = min(bi, d + s_g[W + xp]);
        }
        // no feature anywhere: scipy's distance_transform_edt then measures to a virtual feature at (-1, -1)
        // (observed with scipy 1.15; undefined input for the reference: an all-foreground / all-background mask)
        const int virt = (x + 1) * (x + 1) + (y + 1) * (y + 1);
        if (bo >= EDT_INF * EDT_INF) bo = virt;
        if (bi >= EDT_INF * EDT_INF) bi = virt;
        const size_t o = ((size_t)b * H + y) * W + x;
        if (sq_out) sq_out[o] = bo;
        if (sq_in) sq_in[o] = bi;
        const float diff = (sqrtf((float)bo) - sqrtf((float)bi)) * inv_max;
        out[o] = 1.f / (1.f + expf(-k * diff));
    }
}

}  // namespace

extern "C" {

size_t umr_dt_barrier_workspace_bytes(int B, int H, int W) {
    return (B > 0 && H > 0 && W > 0) ? (size_t)B * 2 * H * W * sizeof(int) : 0;
}

int umr_dt_barrier(const float *mask, float *out, int *sq_out, int *sq_in, int B, int H, int W, float k,
                   void *workspace, size_t workspace_bytes, void *stream) {
    if (!mask || !out || !workspace || B <= 0 || H <= 0 || W <= 0 || W > 8192 || H >= EDT_INF || W >= EDT_INF)
        return UMR_ERR_ARG;
    if (workspace_bytes < umr_dt_barrier_workspace_bytes(B, H, W)) return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 g1((W + 63) / 64, B);   // 64-thread blocks: spread the (few) columns over more CUs
    k_edt_columns<<<g1, 64, 0, st>>>(mask, (int *)workspace, H, W);
    dim3 g2(H, B);
    const int threads = W >= 256 ? 256 : ((W + 63) / 64) * 64;
    k_edt_rows<<<g2, threads, (size_t)2 * W * sizeof(int), st>>>((const int *)workspace, out, sq_out, sq_in, H, W, k,
                                                              1.f / (float)(H > W ? H : W));
    return umr_launch_status();
}

}  // extern "C"
