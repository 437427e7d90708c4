// eval.hip -- keypoint-transfer evaluation of BASELINE config 5 on the device (experiments/test_kp.py:125-193, 253-258,
// 317-323; utils/kp_utils.py:9-69), gfx950.  The reference draws K heat maps on the host per pair (python loop, numpy),
// round-trips them to the GPU, grid-samples them, takes means and arg-maxes with torch ops, and accumulates the PCK in
// numpy; here a pair is three launches per mapping mode and the heat maps never exist: a heat-map pixel is a look-up in the
// (6 sigma + 1)^2 Gaussian patch (computed on the host in float64 exactly as kp_utils.draw_labelmap does, rounded to float
// as its assignment into the float tensor does), clipped the way the reference clips it.  Index work: arg-max / arg-min ties
// go to the FIRST index (torch.max / torch.min on dim); PCK counters are integers.
#include "umr_common.h"

namespace {

__device__ __forceinline__ bool in_img(int x, int y, int W, int H) { return x >= 0 && x < W && y >= 0 && y < H; }

// kp_utils.create_grid (:9-21): affine_grid of the identity with torch 1.1.0's align_corners=True semantics = linspace(-1, 1, S)
// along each axis; torch evaluates linspace from both ends (start + step i below the middle, end - step (S-1-i) above it)
__device__ __forceinline__ float grid_coord(int i, int S) {
    const float step = 2.f / (float)(S - 1);
    return i < S / 2 ? -1.f + step * (float)i : 1.f - step * (float)(S - 1 - i);
}

struct HeatMap {   // one keypoint's heat map, kp_utils.draw_labelmap (:42-69)
    int ulx, uly, brx, bry;   // patch corner (int() = truncation toward zero, :46-47) and clipped end (:63-64)
    bool empty;               // no part of the Gaussian in bounds (:48-50)
};

__device__ __forceinline__ HeatMap heat_map(float kx, float ky, int size_img, int sigma) {
    // test_kp.py:145: kp_src = (kp_src[:, 0:2] + 1) / 2.0 * 256 (float32 tensor arithmetic; the 256 is hard-coded there)
    const float px = (kx + 1.f) / 2.f * 256.f, py = (ky + 1.f) / 2.f * 256.f;
    const float s3 = (float)(3 * sigma);
    HeatMap h;
    h.ulx = (int)(px - s3); h.uly = (int)(py - s3);
    const int bx = (int)((px + s3) + 1.f), by = (int)((py + s3) + 1.f);
    h.empty = h.ulx >= size_img || h.uly >= size_img || bx < 0 || by < 0;
    h.brx = min(bx, size_img); h.bry = min(by, size_img);
    return h;
}

__device__ __forceinline__ float heat_value(const HeatMap &h, const float *__restrict__ patch, int psize, int x, int y) {
    // img[img_y0:img_y1, img_x0:img_x1] = g[g_y0:g_y1, g_x0:g_x1] (:68): pixel (x, y) holds g[y - uly][x - ulx] inside the
    // clipped window [max(0, ul), min(br, size)), 0 elsewhere
    if (h.empty || x < max(h.ulx, 0) || x >= h.brx || y < max(h.uly, 0) || y >= h.bry) return 0.f;
    const int gx = x - h.ulx, gy = y - h.uly;
    if (gx >= psize || gy >= psize) return 0.f;     // (cannot happen for in-range keypoints; NaN / huge coordinates)
    return patch[gy * psize + gx];
}

// Stage 1 of the flow mode, one thread per face: the mean over the face's T^2 texels of
//   * the coordinate grid sampled at the TARGET flow (:136-138)  -> p2face [F,2]: the image point a face lands on
//   * every keypoint heat map sampled at the SOURCE flow (:150-151) -> score [K,F]
// grid_sample = bilinear, zeros padding, align_corners=True (torch 1.1.0 semantics); the arithmetic of k_grid_sample_fwd.
#define EVAL_MAX_K 32
__global__ __launch_bounds__(128) void k_kp_face_scores(const float *__restrict__ kp_src, int kp_stride, const float *__restrict__ flow_src,
                                                        const float *__restrict__ flow_tgt, const float *__restrict__ patch,
                                                        float *__restrict__ score, float *__restrict__ p2face, int K, int F, int TT,
                                                        int S, int sigma) {
    __shared__ HeatMap s_h[EVAL_MAX_K];
    const int pair = blockIdx.y, psize = 6 * sigma + 1;
    if (threadIdx.x < K) {
        const float *kp = kp_src + ((size_t)pair * K + threadIdx.x) * kp_stride;
        s_h[threadIdx.x] = heat_map(kp[0], kp[1], S, sigma);
    }
    __syncthreads();
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float *fs = flow_src + ((size_t)pair * F + f) * TT * 2, *ft = flow_tgt + ((size_t)pair * F + f) * TT * 2;
    float acc[EVAL_MAX_K];
#pragma unroll
    for (int k = 0; k < EVAL_MAX_K; ++k) acc[k] = 0.f;
    float gx = 0.f, gy = 0.f;
    for (int t = 0; t < TT; ++t) {
        {   // coordinate grid at the target flow
            const float ix = ((ft[2 * t] + 1.f) / 2.f) * (S - 1), iy = ((ft[2 * t + 1] + 1.f) / 2.f) * (S - 1);
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
            const float nw = ((fx + 1.f) - ix) * ((fy + 1.f) - iy), ne = (ix - fx) * ((fy + 1.f) - iy);
            const float sw = ((fx + 1.f) - ix) * (iy - fy), se = (ix - fx) * (iy - fy);
            const float cx0 = grid_coord(x0, S), cx1 = grid_coord(x1, S), cy0 = grid_coord(y0, S), cy1 = grid_coord(y1, S);
            float vx = 0.f, vy = 0.f;
            if (in_img(x0, y0, S, S)) { vx += cx0 * nw; vy += cy0 * nw; }
            if (in_img(x1, y0, S, S)) { vx += cx1 * ne; vy += cy0 * ne; }
            if (in_img(x0, y1, S, S)) { vx += cx0 * sw; vy += cy1 * sw; }
            if (in_img(x1, y1, S, S)) { vx += cx1 * se; vy += cy1 * se; }
            gx += vx; gy += vy;
        }
        const float ix = ((fs[2 * t] + 1.f) / 2.f) * (S - 1), iy = ((fs[2 * t + 1] + 1.f) / 2.f) * (S - 1);
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        const float nw = ((fx + 1.f) - ix) * ((fy + 1.f) - iy), ne = (ix - fx) * ((fy + 1.f) - iy);
        const float sw = ((fx + 1.f) - ix) * (iy - fy), se = (ix - fx) * (iy - fy);
        const bool bnw = in_img(x0, y0, S, S), bne = in_img(x1, y0, S, S), bsw = in_img(x0, y1, S, S), bse = in_img(x1, y1, S, S);
#pragma unroll
        for (int k = 0; k < EVAL_MAX_K; ++k) {      // (unrolled with a guard: acc[] stays in registers)
            if (k < K) {
                const HeatMap h = s_h[k];
                // skip when all four corners lie outside the clipped patch window (their values are 0)
                if (!(h.empty || x1 < max(h.ulx, 0) || x0 >= h.brx || y1 < max(h.uly, 0) || y0 >= h.bry)) {
                    float v = 0.f;
                    if (bnw) v += heat_value(h, patch, psize, x0, y0) * nw;
                    if (bne) v += heat_value(h, patch, psize, x1, y0) * ne;
                    if (bsw) v += heat_value(h, patch, psize, x0, y1) * sw;
                    if (bse) v += heat_value(h, patch, psize, x1, y1) * se;
                    acc[k] += v;
                }
            }
        }
    }
    const float inv = 1.f / (float)TT;
    p2face[((size_t)pair * F + f) * 2] = gx * inv;
    p2face[((size_t)pair * F + f) * 2 + 1] = gy * inv;
#pragma unroll
    for (int k = 0; k < EVAL_MAX_K; ++k)
        if (k < K) score[((size_t)pair * K + k) * F + f] = acc[k] * inv;
}

// (value, index) reduction, larger value wins, smaller index on ties: torch.max(dim) returns the first maximal index
__device__ __forceinline__ void better(float &v, int &i, float v2, int i2) {
    if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}

// Stage 2: one block per (pair, keypoint): arg-max of its F scores (:152), k2k = p2face[idx] (:155), and -- when a ground
// truth is given -- the PCK counters of test_kp.py:253-258, 317-323: err = |k2k - gt| (1 + 2 pf) / 2, visible keypoints only
__global__ __launch_bounds__(256) void k_kp_pick(const float *__restrict__ score, const float *__restrict__ p2face, int *__restrict__ face_idx,
                                                 float *__restrict__ k2k, const float *__restrict__ kp_gt, int gt_stride,
                                                 const float *__restrict__ vis, int *__restrict__ counters, int K, int F,
                                                 float err_scale, float thr_a, float thr_b) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const int k = blockIdx.x, pair = blockIdx.y;
    const float *sc = score + ((size_t)pair * K + k) * F;
    float v = -INFINITY;
    int idx = 0x7fffffff;
    for (int f = threadIdx.x; f < F; f += blockDim.x) better(v, idx, sc[f], f);
    s_v[threadIdx.x] = v; s_i[threadIdx.x] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) better(s_v[threadIdx.x], s_i[threadIdx.x], s_v[threadIdx.x + o], s_i[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // every score NaN (a diverged model's flow): `better` never picks a NaN, the index stays at its sentinel -- face 0 then
        // (torch.max of an all-NaN row returns the first index), instead of an out-of-bounds gather
        const int best = s_i[0] == 0x7fffffff ? 0 : s_i[0];
        face_idx[(size_t)pair * K + k] = best;
        const float x = p2face[((size_t)pair * F + best) * 2], y = p2face[((size_t)pair * F + best) * 2 + 1];
        k2k[((size_t)pair * K + k) * 2] = x; k2k[((size_t)pair * K + k) * 2 + 1] = y;
        if (kp_gt && counters) {
            const float *g = kp_gt + ((size_t)pair * K + k) * gt_stride;
            const float ex = x - g[0], ey = y - g[1];
            const float err = sqrtf(ex * ex + ey * ey) * err_scale;
            const bool seen = vis[(size_t)pair * K + k] != 0.f;
            if (seen) {
                atomicAdd(&counters[k], 1);
                if (err < thr_a) atomicAdd(&counters[K + k], 1);
                if (err < thr_b) atomicAdd(&counters[2 * K + k], 1);
            }
        }
    }
}

// Cam mode (:160-193).  For every projected template vertex the nearest FOREGROUND pixel of the target mask under the
// reference's expansion P = |a|^2 + |b|^2 - 2 a.b (chamfer_python.py:56-63) with a = the pixel's grid coordinate; the
// reference compacts the foreground pixels first (torch.nonzero, :177) and indexes the compacted list -- raster order is
// kept by the compaction, so the first minimum over the masked full grid is the same pixel.  One block per vertex.
__global__ __launch_bounds__(256) void k_nearest_fg_pixel(const float *__restrict__ verts2d, const float *__restrict__ mask, int *__restrict__ pix,
                                                          int V, int S) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const int v = blockIdx.x, pair = blockIdx.y;
    const float bx = verts2d[((size_t)pair * V + v) * 2], by = verts2d[((size_t)pair * V + v) * 2 + 1];
    const float bb = bx * bx + by * by;
    const float *m = mask + (size_t)pair * S * S;
    float best = -INFINITY;      // arg-MIN as arg-max of the negated value: `better` keeps the first index on ties
    int idx = 0x7fffffff;
    for (int p = threadIdx.x; p < S * S; p += blockDim.x) {
        if (m[p] == 0.f) continue;
        const float ax = grid_coord(p % S, S), ay = grid_coord(p / S, S);
        const float P = (ax * ax + ay * ay) + bb - 2.f * (ax * bx + ay * by);
        better(best, idx, -P, p);
    }
    s_v[threadIdx.x] = best; s_i[threadIdx.x] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) better(s_v[threadIdx.x], s_i[threadIdx.x], s_v[threadIdx.x + o], s_i[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) pix[(size_t)pair * V + v] = s_i[0] == 0x7fffffff ? -1 : s_i[0];
}

// keypoint -> nearest projected vertex under the source camera (:188) -> that vertex's foreground pixel (:192) -> its grid
// coordinate; one thread per (pair, keypoint); PCK counters as in k_kp_pick
__global__ void k_kp_cam_pick(const float *__restrict__ kp_src, int kp_stride, const float *__restrict__ verts_src, const int *__restrict__ pix,
                              int *__restrict__ vert_idx, float *__restrict__ k2k, const float *__restrict__ kp_gt, int gt_stride,
                              const float *__restrict__ vis, int *__restrict__ counters, int K, int V, int S, float err_scale,
                              float thr_a, float thr_b, int pairs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pairs * K) return;
    const int pair = i / K, k = i % K;
    const float *kp = kp_src + (size_t)i * kp_stride;
    const float ax = kp[0], ay = kp[1], aa = ax * ax + ay * ay;
    float best = INFINITY;
    int bi = 0;
    for (int v = 0; v < V; ++v) {
        const float bx = verts_src[((size_t)pair * V + v) * 2], by = verts_src[((size_t)pair * V + v) * 2 + 1];
        const float P = aa + (bx * bx + by * by) - 2.f * (ax * bx + ay * by);
        if (P < best) { best = P; bi = v; }
    }
    vert_idx[i] = bi;
    const int p = pix[(size_t)pair * V + bi];
    const float x = p >= 0 ? grid_coord(p % S, S) : 0.f, y = p >= 0 ? grid_coord(p / S, S) : 0.f;
    k2k[(size_t)i * 2] = x; k2k[(size_t)i * 2 + 1] = y;
    if (kp_gt && counters) {
        const float *g = kp_gt + (size_t)i * gt_stride;
        const float ex = x - g[0], ey = y - g[1];
        const float err = sqrtf(ex * ex + ey * ey) * err_scale;
        if (vis[i] != 0.f) {
            atomicAdd(&counters[k], 1);
            if (err < thr_a) atomicAdd(&counters[K + k], 1);
            if (err < thr_b) atomicAdd(&counters[2 * K + k], 1);
        }
    }
}

}  // namespace

extern "C" {

size_t umr_kp_flow_workspace_bytes(int pairs, int K, int F) {
    if (pairs <= 0 || K <= 0 || F <= 0) return 0;
    return ((size_t)pairs * K * F + (size_t)pairs * F * 2) * sizeof(float);
}

int umr_kp_flow_transfer(const float *kp_src, int kp_stride, const float *flow_src, const float *flow_tgt, const float *patch,
                         int *face_idx, float *k2k, const float *kp_gt, int gt_stride, const float *vis, int *counters, int pairs,
                         int K, int F, int TT, int image_size, int sigma, float padding_frac, float thr_a, float thr_b,
                         void *workspace, size_t workspace_bytes, void *stream) {
    if (!kp_src || !flow_src || !flow_tgt || !patch || !face_idx || !k2k || !workspace) return UMR_ERR_ARG;
    if (pairs <= 0 || K <= 0 || K > EVAL_MAX_K || F <= 0 || TT <= 0 || image_size < 2 || sigma <= 0 || kp_stride < 2) return UMR_ERR_ARG;
    if ((kp_gt != nullptr) != (counters != nullptr) || (kp_gt && (!vis || gt_stride < 2))) return UMR_ERR_ARG;
    if (workspace_bytes < umr_kp_flow_workspace_bytes(pairs, K, F)) return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    float *score = (float *)workspace, *p2face = score + (size_t)pairs * K * F;
    UMR_LAUNCH(k_kp_face_scores, dim3((F + 127) / 128, pairs), 128, 0, st, kp_src, kp_stride, flow_src, flow_tgt, patch, score, p2face, K, F, TT,
                                                                    image_size, sigma);
    UMR_LAUNCH(k_kp_pick, dim3(K, pairs), 256, 0, st, score, p2face, face_idx, k2k, kp_gt, gt_stride, vis, counters, K, F,
                                              (1.f + 2.f * padding_frac) / 2.f, thr_a, thr_b);
    return umr_launch_status();
}

int umr_kp_cam_transfer(const float *kp_src, int kp_stride, const float *verts_src, const float *verts_tgt, const float *mask_tgt,
                        int *vert_idx, int *pixel_of_vertex, float *k2k, const float *kp_gt, int gt_stride, const float *vis,
                        int *counters, int pairs, int K, int V, int image_size, float padding_frac, float thr_a, float thr_b,
                        void *stream) {
    if (!kp_src || !verts_src || !verts_tgt || !mask_tgt || !vert_idx || !pixel_of_vertex || !k2k) return UMR_ERR_ARG;
    if (pairs <= 0 || K <= 0 || V <= 0 || image_size < 2 || kp_stride < 2) return UMR_ERR_ARG;
    if ((kp_gt != nullptr) != (counters != nullptr) || (kp_gt && (!vis || gt_stride < 2))) return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    UMR_LAUNCH(k_nearest_fg_pixel, dim3(V, pairs), 256, 0, st, verts_tgt, mask_tgt, pixel_of_vertex, V, image_size);
    UMR_LAUNCH(k_kp_cam_pick, (pairs * K + 127) / 128, 128, 0, st, kp_src, kp_stride, verts_src, pixel_of_vertex, vert_idx, k2k, kp_gt, gt_stride,
                                                           vis, counters, K, V, image_size, (1.f + 2.f * padding_frac) / 2.f, thr_a,
                                                           thr_b, pairs);
    return umr_launch_status();
}

}  // extern "C"
