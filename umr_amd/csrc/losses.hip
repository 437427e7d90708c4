// losses.hip -- geometric / image loss kernels around the rasterizer, gfx950 (HBM-bound reductions
// and gathers; no MFMA on purpose -- these are byte-moving ops).  Reference per kernel below.
#include "umr_common.h"

namespace {

// ------------------------------------------------------------------ neg_iou (loss_utils.py:41-48)
// Two stages, no atomics: block k of image n writes its partial (intersection, union) sums into that image's row of
// `sums`, k_iou_finalize adds them in index order -> the loss is bit-reproducible run to run (float atomics were not).
// Row layout (IOU_STRIDE(P) floats): [I, U + 1e-6, I_0 .. I_{nb-1}, U_0 .. U_{nb-1}].
#define IOU_PER_BLOCK 2048
__host__ __device__ inline int iou_blocks(long P) { return (int)((P + IOU_PER_BLOCK - 1) / IOU_PER_BLOCK); }
__host__ __device__ inline long iou_stride(long P) { return 2 + 2L * iou_blocks(P); }

__global__ __launch_bounds__(256) void k_iou_partial(const float *__restrict__ predict, long pstride,
                                                     const float *__restrict__ target, float *__restrict__ sums, long P) {
    __shared__ float smem[16];
    const int n = blockIdx.y, nb = gridDim.x;
    const long start = (long)blockIdx.x * IOU_PER_BLOCK;
    const long end = min(P, start + IOU_PER_BLOCK);
    const float *p = predict + (size_t)n * pstride;
    const float *t = target + (size_t)n * P;
    float si = 0.f, su = 0.f;
    for (long i = start + threadIdx.x; i < end; i += blockDim.x) {
        const float a = p[i], b = t[i];
        const float ab = a * b;
        si += ab;
        su += a + b - ab;
    }
    const float ri = block_sum(si, smem);
    const float ru = block_sum(su, smem);
    if (threadIdx.x == 0) {
        float *row = sums + (size_t)n * iou_stride(P);
        row[2 + blockIdx.x] = ri;
        row[2 + nb + blockIdx.x] = ru;
    }
}

__global__ void k_iou_finalize(float *__restrict__ sums, float *__restrict__ loss, int N, long P) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float *row = sums + (size_t)n * iou_stride(P);
    const int nb = iou_blocks(P);
    float I = 0.f, U = 0.f;
    for (int k = 0; k < nb; ++k) { I += row[2 + k]; U += row[2 + nb + k]; }
    U += 1e-6f;
    row[0] = I; row[1] = U;
    UMR_TRAP_IF(umr_bad(I) | umr_bad(U), 20);
    loss[n] = 1.f - I / U;
}

__global__ void k_iou_backward(const float *__restrict__ predict, long pstride, const float *__restrict__ target,
                               const float *__restrict__ sums, const float *__restrict__ grad_loss,
                               float *__restrict__ grad_predict, long gstride, long P) {
    const int n = blockIdx.y;
    const float *row = sums + (size_t)n * iou_stride(P);
    const float I = row[0], U = row[1], g = grad_loss[n];
    UMR_TRAP_IF(umr_bad(g), 21);
    const float inv_u = 1.f / U, r = I * inv_u * inv_u;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long)gridDim.x * blockDim.x) {
        const float t = target[(size_t)n * P + i];
        // d/dp [1 - I/U] = -(t U - I (1 - t)) / U^2
        grad_predict[(size_t)n * gstride + i] += g * (r * (1.f - t) - t * inv_u);
        (void)predict;
    }
    (void)pstride;
}

// ------------------------------------------------------------------ chamfer (chamfer_python.py:43-64)
// one thread per query point; targets staged through LDS in tiles of 256 points.
template <int D>
__global__ __launch_bounds__(256) void k_chamfer_nn(const float *__restrict__ q, const float *__restrict__ t,
                                                    float *__restrict__ dist, int *__restrict__ idx, int nq, int nt) {
    __shared__ float s_t[256 * 3];
    __shared__ float s_tt[256];
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float *qb = q + (size_t)b * nq * D;
    const float *tb = t + (size_t)b * nt * D;
    float x[D];
    float xx = 0.f;
    if (i < nq) {
#pragma unroll
        for (int d = 0; d < D; ++d) { x[d] = qb[(size_t)i * D + d]; xx += x[d] * x[d]; }
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = 0.f;
    }
    float best = INFINITY;
    int besti = 0;
    for (int j0 = 0; j0 < nt; j0 += 256) {
        const int j = j0 + threadIdx.x;
        __syncthreads();
        if (j < nt) {
            float yy = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) { const float y = tb[(size_t)j * D + d]; s_t[threadIdx.x * D + d] = y; yy += y * y; }
            s_tt[threadIdx.x] = yy;
        }
        __syncthreads();
        const int lim = min(256, nt - j0);
        for (int k = 0; k < lim; ++k) {
            float zz = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) zz += x[d] * s_t[k * D + d];
            const float Pij = xx + s_tt[k] - 2.f * zz;  // rx^T + ry - 2 zz (:63)
            if (Pij < best) { best = Pij; besti = j0 + k; }
        }
    }
    if (i < nq) { dist[(size_t)b * nq + i] = best; idx[(size_t)b * nq + i] = besti; }
}

template <int D>
__global__ void k_chamfer_bwd(const float *__restrict__ q, const float *__restrict__ t, const int *__restrict__ idx,
                              const float *__restrict__ g, float *__restrict__ grad_q, float *__restrict__ grad_t,
                              int nq, int nt) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float gi = g[(size_t)b * nq + i];
    const int j = idx[(size_t)b * nq + i];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float diff = 2.f * (q[((size_t)b * nq + i) * D + d] - t[((size_t)b * nt + j) * D + d]);
        atomicAdd(&grad_q[((size_t)b * nq + i) * D + d], gi * diff);
        atomicAdd(&grad_t[((size_t)b * nt + j) * D + d], -gi * diff);
    }
}

// ------------------------------------------------------------------ grid_sample (geom_utils.py:55, loss_utils.py:64)
// bilinear, zero padding, align_corners=True.  One thread per (b, p); channels looped.
__device__ __forceinline__ bool in_bounds(int x, int y, int W, int H) { return x >= 0 && x < W && y >= 0 && y < H; }

__global__ void k_grid_sample_fwd(const float *__restrict__ image, const float *__restrict__ grid,
                                  float *__restrict__ out, int C, int H, int W, long P) {
    const int b = blockIdx.y;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float *g = grid + ((size_t)b * P + p) * 2;
    const float ix = ((g[0] + 1.f) / 2.f) * (W - 1), iy = ((g[1] + 1.f) / 2.f) * (H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float nw = ((fx + 1.f) - ix) * ((fy + 1.f) - iy), ne = (ix - fx) * ((fy + 1.f) - iy);
    const float sw = ((fx + 1.f) - ix) * (iy - fy), se = (ix - fx) * (iy - fy);
    const bool bnw = in_bounds(x0, y0, W, H), bne = in_bounds(x1, y0, W, H);
    const bool bsw = in_bounds(x0, y1, W, H), bse = in_bounds(x1, y1, W, H);
    const float *img = image + (size_t)b * C * H * W;
    float *o = out + ((size_t)b * P + p) * C;
    for (int c = 0; c < C; ++c) {
        const float *ic = img + (size_t)c * H * W;
        float v = 0.f;
        if (bnw) v += ic[(size_t)y0 * W + x0] * nw;
        if (bne) v += ic[(size_t)y0 * W + x1] * ne;
        if (bsw) v += ic[(size_t)y1 * W + x0] * sw;
        if (bse) v += ic[(size_t)y1 * W + x1] * se;
        UMR_TRAP_IF(umr_bad(v), 22);
        o[c] = v;
    }
}

__global__ void k_grid_sample_bwd(const float *__restrict__ image, const float *__restrict__ grid,
                                  const float *__restrict__ grad_out, float *__restrict__ grad_grid,
                                  float *__restrict__ grad_image, int C, int H, int W, long P) {
    const int b = blockIdx.y;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float *g = grid + ((size_t)b * P + p) * 2;
    const float ix = ((g[0] + 1.f) / 2.f) * (W - 1), iy = ((g[1] + 1.f) / 2.f) * (H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float ex = (fx + 1.f) - ix, wx = ix - fx, ey = (fy + 1.f) - iy, wy = iy - fy;
    const bool bnw = in_bounds(x0, y0, W, H), bne = in_bounds(x1, y0, W, H);
    const bool bsw = in_bounds(x0, y1, W, H), bse = in_bounds(x1, y1, W, H);
    const float *img = image + (size_t)b * C * H * W;
    float *gi = grad_image ? grad_image + (size_t)b * C * H * W : nullptr;
    const float *go = grad_out + ((size_t)b * P + p) * C;
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < C; ++c) {
        const float gv = go[c];
        UMR_TRAP_IF(umr_bad(gv), 23);
        const size_t co = (size_t)c * H * W;
        if (bnw) {
            const float v = img[co + (size_t)y0 * W + x0];
            gix -= v * ey * gv; giy -= v * ex * gv;
            if (gi) atomicAdd(&gi[co + (size_t)y0 * W + x0], ex * ey * gv);
        }
        if (bne) {
            const float v = img[co + (size_t)y0 * W + x1];
            gix += v * ey * gv; giy -= v * wx * gv;
            if (gi) atomicAdd(&gi[co + (size_t)y0 * W + x1], wx * ey * gv);
        }
        if (bsw) {
            const float v = img[co + (size_t)y1 * W + x0];
            gix -= v * wy * gv; giy += v * ex * gv;
            if (gi) atomicAdd(&gi[co + (size_t)y1 * W + x0], ex * wy * gv);
        }
        if (bse) {
            const float v = img[co + (size_t)y1 * W + x1];
            gix += v * wy * gv; giy += v * wx * gv;
            if (gi) atomicAdd(&gi[co + (size_t)y1 * W + x1], wx * wy * gv);
        }
    }
    UMR_TRAP_IF(umr_bad(gix) | umr_bad(giy), 24);
    if (grad_grid) {
        float *gg = grad_grid + ((size_t)b * P + p) * 2;
        gg[0] = gix * ((W - 1) / 2.f);
        gg[1] = giy * ((H - 1) / 2.f);
    }
}

// ------------------------------------------------------------------ LaplacianLoss (losses.py:29-37)
__global__ __launch_bounds__(256) void k_laplacian_fwd(const float *__restrict__ x, const int *__restrict__ off,
                                                       const int *__restrict__ nbr, float *__restrict__ lap,
                                                       float *__restrict__ loss, int V) {
    __shared__ float smem[16];
    const int b = blockIdx.x;
    const float *xb = x + (size_t)b * V * 3;
    float acc = 0.f;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        const int s = off[v], e = off[v + 1];
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int k = s; k < e; ++k) { const float *u = xb + (size_t)nbr[k] * 3; sx += u[0]; sy += u[1]; sz += u[2]; }
        const float inv = e > s ? 1.f / (float)(e - s) : 0.f;
        const float lx = xb[v * 3] - sx * inv, ly = xb[v * 3 + 1] - sy * inv, lz = xb[v * 3 + 2] - sz * inv;
        UMR_TRAP_IF(umr_bad(lx) | umr_bad(ly) | umr_bad(lz), 25);
        float *l = lap + ((size_t)b * V + v) * 3;
        l[0] = lx; l[1] = ly; l[2] = lz;
        acc += lx * lx + ly * ly + lz * lz;
    }
    const float r = block_sum(acc, smem);
    if (threadIdx.x == 0) loss[b] = r;
}

__global__ void k_laplacian_bwd(const float *__restrict__ lap, const int *__restrict__ off,
                                const int *__restrict__ nbr, const float *__restrict__ grad_loss,
                                float *__restrict__ grad_x, int V) {
    const int b = blockIdx.y;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float *lb = lap + (size_t)b * V * 3;
    const int s = off[v], e = off[v + 1];
    float ax = lb[v * 3], ay = lb[v * 3 + 1], az = lb[v * 3 + 2];
    for (int k = s; k < e; ++k) {  // L^T: -1/deg(u) for every neighbour u of v
        const int u = nbr[k];
        const float inv = 1.f / (float)(off[u + 1] - off[u]);
        ax -= lb[u * 3] * inv; ay -= lb[u * 3 + 1] * inv; az -= lb[u * 3 + 2] * inv;
    }
    const float g = 2.f * grad_loss[b];
    float *o = grad_x + ((size_t)b * V + v) * 3;
    o[0] += g * ax; o[1] += g * ay; o[2] += g * az;
}

// ------------------------------------------------------------------ FlattenLoss (losses.py:72-114)
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float *p) { V3 v = {p[0], p[1], p[2]}; return v; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { V3 v = {a.x - b.x, a.y - b.y, a.z - b.z}; return v; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { V3 v = {a.x + b.x, a.y + b.y, a.z + b.z}; return v; }
__device__ __forceinline__ V3 mul(V3 a, float s) { V3 v = {a.x * s, a.y * s, a.z * s}; return v; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct FlatSide { V3 b, cb; float bl1, ab, cosv, sinv, cbl1, dc; };

__device__ __forceinline__ FlatSide flat_side(V3 a, V3 b, float al2, float al1, float eps) {
    FlatSide s;
    s.b = b;
    const float bl2 = dot(b, b);
    s.bl1 = sqrtf(bl2 + eps);
    s.ab = dot(a, b);
    s.dc = al1 * s.bl1 + eps;
    s.cosv = s.ab / s.dc;
    s.sinv = sqrtf(1.f - s.cosv * s.cosv + eps);
    s.cb = sub(b, mul(a, s.ab / (al2 + eps)));
    s.cbl1 = s.bl1 * s.sinv;
    return s;
}

// reverse mode through one side; accumulates into g_a, g_al2, g_al1 and returns g_b
__device__ __forceinline__ V3 flat_side_bwd(const FlatSide &s, V3 a, float al2, float al1, float eps, V3 g_cb,
                                            float g_cbl1, V3 &g_a, float &g_al2, float &g_al1) {
    const float k = s.ab / (al2 + eps);
    V3 g_b = g_cb;
    g_a = sub(g_a, mul(g_cb, k));
    const float g_k = -dot(a, g_cb);
    float g_ab = g_k / (al2 + eps);
    g_al2 += -g_k * s.ab / ((al2 + eps) * (al2 + eps));
    float g_bl1 = g_cbl1 * s.sinv;
    const float g_sin = g_cbl1 * s.bl1;
    const float g_cos = -g_sin * s.cosv / s.sinv;
    g_ab += g_cos / s.dc;
    const float g_dc = -g_cos * s.ab / (s.dc * s.dc);
    g_al1 += g_dc * s.bl1;
    g_bl1 += g_dc * al1;
    g_b = add(g_b, mul(s.b, g_bl1 / s.bl1));  // bl1 = sqrt(b.b + eps): d/db = b / bl1
    g_a = add(g_a, mul(s.b, g_ab));
    g_b = add(g_b, mul(a, g_ab));
    return g_b;
}

template <bool BWD>
__global__ __launch_bounds__(256) void k_flatten(const float *__restrict__ x, const int *__restrict__ quads,
                                                 float *__restrict__ loss, const float *__restrict__ grad_loss,
                                                 float *__restrict__ grad_x, int V, int E) {
    __shared__ float smem[16];
    const float eps = 1e-6f;
    const int b = blockIdx.y;
    const float *xb = x + (size_t)b * V * 3;
    float acc = 0.f;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const int i0 = quads[e * 4], i1 = quads[e * 4 + 1], i2 = quads[e * 4 + 2], i3 = quads[e * 4 + 3];
        const V3 v0 = ld3(xb + (size_t)i0 * 3);
        const V3 a = sub(ld3(xb + (size_t)i1 * 3), v0);
        const float al2 = dot(a, a), al1 = sqrtf(al2 + eps);
        const FlatSide s1 = flat_side(a, sub(ld3(xb + (size_t)i2 * 3), v0), al2, al1, eps);
        const FlatSide s2 = flat_side(a, sub(ld3(xb + (size_t)i3 * 3), v0), al2, al1, eps);
        const float nn = dot(s1.cb, s2.cb), dd = s1.cbl1 * s2.cbl1 + eps;
        const float cosd = nn / dd;
        UMR_TRAP_IF(umr_bad(cosd), 26);
        if (!BWD) {
            acc += (cosd + 1.f) * (cosd + 1.f);
        } else {
            const float g_cosd = 2.f * (cosd + 1.f) * grad_loss[b];
            const float g_nn = g_cosd / dd, g_dd = -g_cosd * nn / (dd * dd);
            V3 g_a = {0.f, 0.f, 0.f};
            float g_al2 = 0.f, g_al1 = 0.f;
            const V3 g_b1 = flat_side_bwd(s1, a, al2, al1, eps, mul(s2.cb, g_nn), g_dd * s2.cbl1, g_a, g_al2, g_al1);
            const V3 g_b2 = flat_side_bwd(s2, a, al2, al1, eps, mul(s1.cb, g_nn), g_dd * s1.cbl1, g_a, g_al2, g_al1);
            g_al2 += g_al1 / (2.f * al1);
            g_a = add(g_a, mul(a, 2.f * g_al2));
            float *gb = grad_x + (size_t)b * V * 3;
            const V3 g0 = {-(g_a.x + g_b1.x + g_b2.x), -(g_a.y + g_b1.y + g_b2.y), -(g_a.z + g_b1.z + g_b2.z)};
            atomicAdd(gb + (size_t)i0 * 3, g0.x); atomicAdd(gb + (size_t)i0 * 3 + 1, g0.y); atomicAdd(gb + (size_t)i0 * 3 + 2, g0.z);
            atomicAdd(gb + (size_t)i1 * 3, g_a.x); atomicAdd(gb + (size_t)i1 * 3 + 1, g_a.y); atomicAdd(gb + (size_t)i1 * 3 + 2, g_a.z);
            atomicAdd(gb + (size_t)i2 * 3, g_b1.x); atomicAdd(gb + (size_t)i2 * 3 + 1, g_b1.y); atomicAdd(gb + (size_t)i2 * 3 + 2, g_b1.z);
            atomicAdd(gb + (size_t)i3 * 3, g_b2.x); atomicAdd(gb + (size_t)i3 * 3 + 1, g_b2.y); atomicAdd(gb + (size_t)i3 * 3 + 2, g_b2.z);
        }
    }
    if (!BWD) {
        const float r = block_sum(acc, smem);
        if (threadIdx.x == 0) atomicAdd(&loss[b], r);
    }
}

// ------------------------------------------------------------------ TexCycle visibility (loss_utils.py:173-179)
__global__ void k_visible_mask(const float *__restrict__ ids, float *__restrict__ mask, long P, int F) {
    const int b = blockIdx.y;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long)gridDim.x * blockDim.x) {
        int f = (int)ids[(size_t)b * P + i];
        if (f < 0) f += F;  // python negative indexing: background id -1 marks the last face
        if (f >= 0 && f < F) mask[(size_t)b * F + f] = 1.f;
    }
}

}  // namespace

UMR_TRAP_ACCESSOR(umr_trap_read_losses)

extern "C" {

long umr_neg_iou_sums_stride(long P) { return P > 0 ? iou_stride(P) : 0; }

int umr_neg_iou_forward(const float *predict, long predict_stride, const float *target, float *loss,
                        float *sums, size_t sums_bytes, int N, long P, void *stream) {
    if (!predict || !target || !loss || !sums || N <= 0 || P <= 0 || predict_stride < P) return UMR_ERR_ARG;
    if (sums_bytes < (size_t)N * (size_t)iou_stride(P) * sizeof(float)) return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)iou_blocks(P), (unsigned)N);
    UMR_LAUNCH(k_iou_partial, grid, 256, 0, st, predict, predict_stride, target, sums, P);
    UMR_LAUNCH(k_iou_finalize, (N + 63) / 64, 64, 0, st, sums, loss, N, P);
    return umr_launch_status();
}

int umr_neg_iou_backward(const float *predict, long predict_stride, const float *target, const float *sums,
                         const float *grad_loss, float *grad_predict, long grad_stride, int N, long P,
                         void *stream) {
    if (!predict || !target || !sums || !grad_loss || !grad_predict || N <= 0 || P <= 0) return UMR_ERR_ARG;
    dim3 grid((unsigned)min((long)1024, (P + 255) / 256), (unsigned)N);
    UMR_LAUNCH(k_iou_backward, grid, 256, 0, (hipStream_t)stream, predict, predict_stride, target, sums, grad_loss, grad_predict,
                                                         grad_stride, P);
    return umr_launch_status();
}

int umr_chamfer_forward(const float *a, const float *b, float *dist1, float *dist2, int *idx1, int *idx2,
                        int B, int n, int m, int D, void *stream) {
    if (!a || !b || !dist1 || !dist2 || !idx1 || !idx2 || B <= 0 || n <= 0 || m <= 0 || (D != 2 && D != 3))
        return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 g1((n + 255) / 256, B), g2((m + 255) / 256, B);
    if (D == 2) {
        UMR_LAUNCH((k_chamfer_nn<2>), g1, 256, 0, st, a, b, dist1, idx1, n, m);
        UMR_LAUNCH((k_chamfer_nn<2>), g2, 256, 0, st, b, a, dist2, idx2, m, n);
    } else {
        UMR_LAUNCH((k_chamfer_nn<3>), g1, 256, 0, st, a, b, dist1, idx1, n, m);
        UMR_LAUNCH((k_chamfer_nn<3>), g2, 256, 0, st, b, a, dist2, idx2, m, n);
    }
    return umr_launch_status();
}

int umr_chamfer_backward(const float *a, const float *b, const int *idx1, const int *idx2, const float *g1,
                         const float *g2, float *grad_a, float *grad_b, int B, int n, int m, int D,
                         void *stream) {
    if (!a || !b || !idx1 || !idx2 || !g1 || !g2 || !grad_a || !grad_b || B <= 0 || n <= 0 || m <= 0 ||
        (D != 2 && D != 3))
        return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (!umr_zero_async(grad_a, (size_t)B * n * D * sizeof(float), st)) return UMR_ERR_LAUNCH;
    if (!umr_zero_async(grad_b, (size_t)B * m * D * sizeof(float), st)) return UMR_ERR_LAUNCH;
    dim3 ga((n + 255) / 256, B), gb((m + 255) / 256, B);
    if (D == 2) {
        UMR_LAUNCH((k_chamfer_bwd<2>), ga, 256, 0, st, a, b, idx1, g1, grad_a, grad_b, n, m);
        UMR_LAUNCH((k_chamfer_bwd<2>), gb, 256, 0, st, b, a, idx2, g2, grad_b, grad_a, m, n);
    } else {
        UMR_LAUNCH((k_chamfer_bwd<3>), ga, 256, 0, st, a, b, idx1, g1, grad_a, grad_b, n, m);
        UMR_LAUNCH((k_chamfer_bwd<3>), gb, 256, 0, st, b, a, idx2, g2, grad_b, grad_a, m, n);
    }
    return umr_launch_status();
}

int umr_grid_sample_forward(const float *image, const float *grid, float *out, int B, int C, int H, int W,
                            long P, void *stream) {
    if (!image || !grid || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || P <= 0) return UMR_ERR_ARG;
    dim3 g((unsigned)((P + 255) / 256), (unsigned)B);
    UMR_LAUNCH(k_grid_sample_fwd, g, 256, 0, (hipStream_t)stream, image, grid, out, C, H, W, P);
    return umr_launch_status();
}

int umr_grid_sample_backward(const float *image, const float *grid, const float *grad_out, float *grad_grid,
                             float *grad_image, int B, int C, int H, int W, long P, void *stream) {
    if (!image || !grid || !grad_out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || P <= 0) return UMR_ERR_ARG;
    if (!grad_grid && !grad_image) return UMR_OK;
    dim3 g((unsigned)((P + 255) / 256), (unsigned)B);
    UMR_LAUNCH(k_grid_sample_bwd, g, 256, 0, (hipStream_t)stream, image, grid, grad_out, grad_grid, grad_image, C, H, W, P);
    return umr_launch_status();
}

int umr_laplacian_forward(const float *x, const int *nbr_off, const int *nbr_idx, float *lap, float *loss,
                          int B, int V, void *stream) {
    if (!x || !nbr_off || !nbr_idx || !lap || !loss || B <= 0 || V <= 0) return UMR_ERR_ARG;
    UMR_LAUNCH(k_laplacian_fwd, B, 256, 0, (hipStream_t)stream, x, nbr_off, nbr_idx, lap, loss, V);
    return umr_launch_status();
}

int umr_laplacian_backward(const float *lap, const int *nbr_off, const int *nbr_idx, const float *grad_loss,
                           float *grad_x, int B, int V, void *stream) {
    if (!lap || !nbr_off || !nbr_idx || !grad_loss || !grad_x || B <= 0 || V <= 0) return UMR_ERR_ARG;
    dim3 g((V + 255) / 256, B);
    UMR_LAUNCH(k_laplacian_bwd, g, 256, 0, (hipStream_t)stream, lap, nbr_off, nbr_idx, grad_loss, grad_x, V);
    return umr_launch_status();
}

int umr_flatten_forward(const float *x, const int *quads, float *loss, int B, int V, int E, void *stream) {
    if (!x || !quads || !loss || B <= 0 || V <= 0 || E <= 0) return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (!umr_zero_async(loss, (size_t)B * sizeof(float), st)) return UMR_ERR_LAUNCH;
    dim3 g(1, B);   // one block per mesh (grid-stride over the edges): a single add into loss[b] -> bit-reproducible
    UMR_LAUNCH((k_flatten<false>), g, 256, 0, st, x, quads, loss, nullptr, nullptr, V, E);
    return umr_launch_status();
}

int umr_flatten_backward(const float *x, const int *quads, const float *grad_loss, float *grad_x, int B, int V,
                         int E, void *stream) {
    if (!x || !quads || !grad_loss || !grad_x || B <= 0 || V <= 0 || E <= 0) return UMR_ERR_ARG;
    dim3 g((E + 255) / 256, B);
    UMR_LAUNCH((k_flatten<true>), g, 256, 0, (hipStream_t)stream, x, quads, nullptr, grad_loss, grad_x, V, E);
    return umr_launch_status();
}

int umr_visible_face_mask(const float *face_ids, float *mask, int B, long P, int F, void *stream) {
    if (!face_ids || !mask || B <= 0 || P <= 0 || F <= 0) return UMR_ERR_ARG;
    dim3 g((unsigned)min((long)512, (P + 255) / 256), (unsigned)B);
    UMR_LAUNCH(k_visible_mask, g, 256, 0, (hipStream_t)stream, face_ids, mask, P, F);
    return umr_launch_status();
}

}  // extern "C"

// ------------------------------------------------------------------ 2x bilinear up-sampling (network decoder)
// nn.Upsample(scale_factor=2, mode='bilinear') of the texture-flow decoder (nnutils/net_blocks.py upconv2d,
// cub_mesh.py:141): PyTorch-ROCm's generic kernel takes ~440 us per call at these shapes (13 % of the training
// step); the fixed 2x case is a 4-tap stencil.  align_corners=False: src = dst/2 - 0.25 clamped at 0.
namespace {
__device__ __forceinline__ void up2_taps(int o, int n, int &i0, int &i1, float &w1) {
    const float src = fmaxf(0.5f * (float)o - 0.25f, 0.f);
    i0 = (int)src;
    i1 = min(i0 + 1, n - 1);
    w1 = src - (float)i0;
}

// grid (ceil(OH * OW / 256), planes): 32-bit index arithmetic, one udiv (a shift when the width is a power of two).  The
// first version used one flat 64-bit index -- two 64-bit div/mod pairs per element, ~300 instructions around 4 loads.
__global__ void k_upsample2x_fwd(const float *__restrict__ in, float *__restrict__ out, int H, int W, int wshift) {
    const unsigned OW = 2u * W, OH = 2u * H;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= OW * OH) return;
    const unsigned oy = wshift >= 0 ? i >> wshift : i / OW, ox = i - oy * OW;
    int x0, x1, y0, y1; float wx, wy;
    up2_taps((int)ox, W, x0, x1, wx);
    up2_taps((int)oy, H, y0, y1, wy);
    const float *s = in + (size_t)blockIdx.y * H * W;
    const float a = s[y0 * W + x0], b = s[y0 * W + x1], c = s[y1 * W + x0], d = s[y1 * W + x1];
    UMR_TRAP_IF(umr_bad(a), 27);
    out[(size_t)blockIdx.y * OH * OW + i] = (1.f - wy) * ((1.f - wx) * a + wx * b) + wy * ((1.f - wx) * c + wx * d);
}

// gather form of the transpose: every input pixel sums its (<= 4x4) output contributions -> deterministic, no atomics
__global__ void k_upsample2x_bwd(const float *__restrict__ gout, float *__restrict__ gin, int H, int W, int wshift) {
    const int OW = 2 * W, OH = 2 * H;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (unsigned)(W * H)) return;
    const int y = wshift >= 0 ? (int)(i >> wshift) : (int)(i / (unsigned)W), x = (int)i - y * W;
    const float *g = gout + (size_t)blockIdx.y * OH * OW;
    float acc = 0.f;
    for (int oy = max(2 * y - 1, 0); oy <= min(2 * y + 2, OH - 1); ++oy) {
        int y0, y1; float wy;
        up2_taps(oy, H, y0, y1, wy);
        const float cy = (y0 == y ? 1.f - wy : 0.f) + (y1 == y ? wy : 0.f);
        if (cy == 0.f) continue;
        for (int ox = max(2 * x - 1, 0); ox <= min(2 * x + 2, OW - 1); ++ox) {
            int x0, x1; float wx;
            up2_taps(ox, W, x0, x1, wx);
            const float cx = (x0 == x ? 1.f - wx : 0.f) + (x1 == x ? wx : 0.f);
            acc += cy * cx * g[oy * OW + ox];
        }
    }
    UMR_TRAP_IF(umr_bad(acc), 28);
    gin[(size_t)blockIdx.y * H * W + i] = acc;
}
}  // namespace

extern "C" {
static int pow2_shift(int v) { return (v > 0 && (v & (v - 1)) == 0) ? __builtin_ctz(v) : -1; }

int umr_upsample2x_bilinear_forward(const float *in, float *out, long planes, int H, int W, void *stream) {
    if (!in || !out || planes <= 0 || H <= 0 || W <= 0 || (long)H * W > (1L << 27)) return UMR_ERR_ARG;
    const unsigned per = 4u * H * W;
    for (long p0 = 0; p0 < planes; p0 += 65535) {          // grid.y limit
        const long np = planes - p0 < 65535 ? planes - p0 : 65535;
        UMR_LAUNCH(k_upsample2x_fwd, dim3((per + 255) / 256, (unsigned)np), 256, 0, (hipStream_t)stream,
            in + (size_t)p0 * H * W, out + (size_t)p0 * per, H, W, pow2_shift(2 * W));
    }
    return umr_launch_status();
}

int umr_upsample2x_bilinear_backward(const float *grad_out, float *grad_in, long planes, int H, int W, void *stream) {
    if (!grad_out || !grad_in || planes <= 0 || H <= 0 || W <= 0 || (long)H * W > (1L << 27)) return UMR_ERR_ARG;
    const unsigned per = (unsigned)H * W;
    for (long p0 = 0; p0 < planes; p0 += 65535) {
        const long np = planes - p0 < 65535 ? planes - p0 : 65535;
        UMR_LAUNCH(k_upsample2x_bwd, dim3((per + 255) / 256, (unsigned)np), 256, 0, (hipStream_t)stream,
            grad_out + (size_t)p0 * 4 * per, grad_in + (size_t)p0 * per, H, W, pow2_shift(W));
    }
    return umr_launch_status();
}
}
