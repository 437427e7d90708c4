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

// Column pass: for every pixel the distance (in rows) to the nearest foreground / background pixel of its column.
// Features of EDT(1-mask) are mask != 0, features of EDT(mask) are mask == 0.  The sweep along a column is a serial
// dependence, and an image batch has few columns (16 x 256 = 4096), so one thread per column leaves the chip idle and
// pays H dependent steps (measured 230-310 us for 16 x 256^2).  Here a column is cut into `nseg` segments of <= 16 rows:
// a workgroup = 16 columns x nseg segments; every thread summarises its segment (first / last feature row of each
// kind), the summaries meet in LDS, each thread derives the carry from the segments above / below, then sweeps its own
// rows down and up.  Same integer results as the serial sweep.
#define EDT_COLS 16
#define EDT_BIG (2 * EDT_INF)
__global__ void k_edt_columns(const float *__restrict__ mask, int *__restrict__ g, int H, int W, int nseg) {
#ifdef UMR_HOST_SHIM   // tests/host_kernel/wave_emu.h: the launch's dynamic LDS
    int *s_sum = (int *)umr_host_dynamic_lds();
#else
    extern __shared__ int s_sum[];  // [4][nseg][EDT_COLS]: last fg, first fg, last bg, first bg
#endif
    const int b = blockIdx.y, c = threadIdx.x % EDT_COLS, seg = threadIdx.x / EDT_COLS;
    const int x = blockIdx.x * EDT_COLS + c;
    const bool on = x < W;
    const int L = (H + nseg - 1) / nseg, y0 = seg * L, y1 = min(H, y0 + L);
    const float *__restrict__ m = mask + (size_t)b * H * W + (on ? x : 0);
    int *__restrict__ go = g + (size_t)b * 2 * H * W + (on ? x : 0);
    int *__restrict__ gi = go + (size_t)H * W;
    int lfg = -EDT_INF, ffg = EDT_BIG, lbg = -EDT_INF, fbg = EDT_BIG;
    if (on) {
#pragma unroll 4
        for (int y = y0; y < y1; ++y) {
            const bool fg = m[(size_t)y * W] != 0.f;
            lfg = fg ? y : lfg; ffg = fg ? min(ffg, y) : ffg;
            lbg = fg ? lbg : y; fbg = fg ? fbg : min(fbg, y);
        }
    }
    const int plane = nseg * EDT_COLS, slot = seg * EDT_COLS + c;
    s_sum[slot] = lfg; s_sum[plane + slot] = ffg; s_sum[2 * plane + slot] = lbg; s_sum[3 * plane + slot] = fbg;
    __syncthreads();
    if (!on) return;
    int last_fg = -EDT_INF, last_bg = -EDT_INF, next_fg = EDT_BIG, next_bg = EDT_BIG;
    for (int s2 = 0; s2 < seg; ++s2) {
        last_fg = max(last_fg, s_sum[s2 * EDT_COLS + c]);
        last_bg = max(last_bg, s_sum[2 * plane + s2 * EDT_COLS + c]);
    }
    for (int s2 = seg + 1; s2 < nseg; ++s2) {
        next_fg = min(next_fg, s_sum[plane + s2 * EDT_COLS + c]);
        next_bg = min(next_bg, s_sum[3 * plane + s2 * EDT_COLS + c]);
    }
#pragma unroll 4
    for (int y = y0; y < y1; ++y) {      // downward: distance to the last feature at or above
        const bool fg = m[(size_t)y * W] != 0.f;
        last_fg = fg ? y : last_fg;
        last_bg = fg ? last_bg : y;
        go[(size_t)y * W] = min(y - last_fg, EDT_INF);
        gi[(size_t)y * W] = min(y - last_bg, EDT_INF);
    }
#pragma unroll 4
    for (int y = y1 - 1; y >= y0; --y) {  // upward: nearest of above / below
        const bool fg = m[(size_t)y * W] != 0.f;
        next_fg = fg ? y : next_fg;
        next_bg = fg ? next_bg : y;
        go[(size_t)y * W] = min(go[(size_t)y * W], min(next_fg - y, EDT_INF));
        gi[(size_t)y * W] = min(gi[(size_t)y * W], min(next_bg - y, EDT_INF));
    }
}

// one block per image row; LDS holds the row's squared column distances for both transforms
__global__ void k_edt_rows(const int *__restrict__ g, float *__restrict__ out, int *__restrict__ sq_out,
                           int *__restrict__ sq_in, int H, int W, float k, float inv_max) {
#ifdef UMR_HOST_SHIM   // tests/host_kernel/wave_emu.h: the launch's dynamic LDS
    int *s_g = (int *)umr_host_dynamic_lds();
#else
    extern __shared__ int s_g[];  // [2][W]
#endif
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
        UMR_TRAP_IF(umr_bad(1.f / (1.f + expf(-k * diff))), 40);
        out[o] = 1.f / (1.f + expf(-k * diff));
    }
}

}  // namespace

UMR_TRAP_ACCESSOR(umr_trap_read_edt)

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
    const int nseg = max(1, min(64, (H + 15) / 16));   // <= 16 rows per thread up to H = 1024
    dim3 g1((W + EDT_COLS - 1) / EDT_COLS, B);
    UMR_LAUNCH(k_edt_columns, g1, EDT_COLS * nseg, (size_t)4 * nseg * EDT_COLS * sizeof(int), st, mask, (int *)workspace, H, W, nseg);
    dim3 g2(H, B);
    const int threads = W >= 256 ? 256 : ((W + 63) / 64) * 64;
    UMR_LAUNCH(k_edt_rows, g2, threads, (size_t)2 * W * sizeof(int), st, (const int *)workspace, out, sq_out, sq_in, H, W, k,
                                                              1.f / (float)(H > W ? H : W));
    return umr_launch_status();
}

}  // extern "C"
