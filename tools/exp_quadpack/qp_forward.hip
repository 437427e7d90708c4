// qp_forward.hip -- EXPERIMENT (round 3; not part of libumr_hip.so): "forward lane packing" of VERDICT r2 item 2 as a kernel.
//
// The product's pixel-major forward gives a wave one 8x8 tile and visits every face that touches the tile with all 64 lanes:
// 52 % of the lanes of a visit hold a pixel the face contributes to.  Here a workgroup still owns a 16x16
// block, but the unit of work is the 4x4 QUADRANT: the block's candidate faces are binned per quadrant (16 ascending lists in
// LDS), the 16 quadrants are dealt to the 16 lane groups of the block's four waves by descending list length (each wave gets
// four quadrants of similar length), and in visit i every 16-lane group evaluates the i-th face of ITS OWN quadrant's list --
// four different faces per wave instruction.  Pixels still see their faces in ascending index order, so results are
// bit-identical (checked on the wave64 emulator against the product kernel: tools/exp_quadpack/check_and_time.py).  The price:
// the face record is no longer wave-uniform -- it comes from memory into VGPRs per lane (7 x 16 B) instead of 2 scalar loads
// into SGPRs -- and the lane-to-pixel map is decided at run time.  CPU count (tools/sim/sim_q.py): 20 081 -> 16 085 visits per
// mesh on the bench scene (-20 %).  Silhouette variant only (k_raster_forward<2>'s job): alpha = 1 - prod(1 - D_f).
#include "../../umr_amd/csrc/raster_core.h"

#ifndef QP_LDS_REC
#define QP_LDS_REC 0    // 1: the sub-chunk's face records are staged in LDS once per block and read from there per visit
#endif                  //    (7 x ds_read_b128 instead of 7 x global_load_dwordx4 per lane and visit)
#if QP_LDS_REC
#define QP_CAP 128      // faces per sub-chunk of the block list = capacity of a quadrant list (LDS: 128 x 112 B of records)
#else
#define QP_CAP 256
#endif

namespace {

struct FaceL {   // per-lane copy of the record fields the silhouette path reads
    float r[32];
    const char *edges;
    template <int I> __device__ __forceinline__ float g() const { return r[I]; }
    template <int I> __device__ __forceinline__ float inv() const { return r[R_INV + I]; }
    template <int I> __device__ __forceinline__ float xy() const { return r[R_X0 + I]; }
    __device__ __forceinline__ int obt() const { return (__float_as_int(r[R_FLAGS]) & 3) - 1; }
    __device__ __forceinline__ bool ill_conditioned() const { return (__float_as_int(r[R_FLAGS]) & 16) != 0; }
};

__device__ __forceinline__ void load_face_lane(FaceL &fc, const float *rg, const float4 *staged = nullptr) {
    const float4 *q = (const float4 *)rg;
    float4 a, b, c, e, f, g, h;
    if (staged) { a = staged[0]; b = staged[1]; c = staged[2]; e = staged[3]; f = staged[4]; g = staged[5]; h = staged[6]; }
    else { a = q[0]; b = q[1]; c = q[2]; e = q[4]; f = q[5]; g = q[6]; h = q[7]; }
    fc.r[0] = a.x; fc.r[1] = a.y; fc.r[2] = a.z; fc.r[3] = a.w;
    fc.r[4] = b.x; fc.r[5] = b.y; fc.r[6] = b.z; fc.r[7] = b.w;
    fc.r[8] = c.x; fc.r[9] = c.y;
    fc.r[16] = e.x; fc.r[17] = e.y; fc.r[18] = e.z; fc.r[19] = e.w;
    fc.r[20] = f.x; fc.r[21] = f.y; fc.r[22] = f.z; fc.r[23] = f.w;
    fc.r[24] = g.x; fc.r[25] = g.y; fc.r[26] = g.z; fc.r[27] = g.w;
    fc.r[28] = h.x; fc.r[29] = h.y; fc.r[30] = h.z; fc.r[31] = h.w;
    fc.edges = (const char *)(rg + R_EDGE);
}

#ifndef QP_WPE
#define QP_WPE 6
#endif
#ifdef UMR_HOST_SHIM
#define QP_ATTR
#else
#define QP_ATTR __attribute__((amdgpu_waves_per_eu(QP_WPE, QP_WPE)))
#endif

__global__ __launch_bounds__(BLK_THREADS) QP_ATTR void k_sil_forward_qp(const RasterArgs A) {
    __shared__ int s_list[LIST_CAP];
    __shared__ int s_wcnt[BLK_THREADS / 64];
    __shared__ int q_list[16][QP_CAP];
    __shared__ int q_cnt[16];
    __shared__ int q_perm[16];
#if QP_LDS_REC
    __shared__ float4 s_rec[QP_CAP][7];
#endif
    Tile t;
    tile_setup(t, A);
    const int F = A.F, IS = A.IS;
    const size_t npix = (size_t)IS * IS;
    const float4 *__restrict__ bbox_n = A.bbox + (size_t)t.n * F;
    const float *__restrict__ rec_n = A.rec + (size_t)t.n * F * REC;
    const bool pow2 = (IS & (IS - 1)) == 0;
    const float inv_is = 1.f / (float)IS;
    const int tid = threadIdx.x, grp = tid >> 4, j = tid & 15;

    // binning role of this thread: quadrant `grp` of the block, j-th face of a group of 16
    const int bqx0 = t.bx0 + 4 * (grp & 3), bqy0 = t.by0 + 4 * (grp >> 2);
    const float qxl = ndc_coord_fast(min(bqx0, IS - 1), IS, inv_is, pow2), qxh = ndc_coord_fast(min(bqx0 + 3, IS - 1), IS, inv_is, pow2);
    const float qyh = ndc_coord_fast(IS - 1 - min(bqy0, IS - 1), IS, inv_is, pow2), qyl = ndc_coord_fast(IS - 1 - min(bqy0 + 3, IS - 1), IS, inv_is, pow2);
    const bool q_on = bqx0 < IS && bqy0 < IS;

    float alpha = 1.f;
    int my_q = 0, px = 0, row = 0;   // visiting role (decided after the first sub-chunk is binned)
    bool valid = false, have_perm = false;

    const int *sb_ids;
    const int ncand = superblock_list(A, t, sb_ids);
    for (int f0 = 0; f0 < ncand; f0 += LIST_CAP) {
        const int f1 = min(ncand, f0 + LIST_CAP);
        if (f0 > 0) __syncthreads();
        const int count = build_list(s_list, s_wcnt, bbox_n, sb_ids, f0, f1, t);
        for (int c0 = 0; c0 < count; c0 += QP_CAP) {
            const int c1 = min(count, c0 + QP_CAP);
            if (c0 > 0) __syncthreads();           // the previous sub-chunk's lists are still being walked
#if QP_LDS_REC
            for (int e = tid; e < (c1 - c0) * 7; e += BLK_THREADS) {      // stage the records of this sub-chunk's faces
                const int fi = e / 7, part = e - fi * 7;
                s_rec[fi][part] = ((const float4 *)(rec_n + (size_t)s_list[c0 + fi] * REC))[part < 3 ? part : part + 1];
            }
#endif
            // ---- per-quadrant lists of this sub-chunk, ascending ----
            int qn = 0;
            for (int b = c0; b < c1; b += 16) {
                const int li = b + j;
                bool hit = false;
                int f = 0;
                if (li < c1 && q_on) {
                    f = s_list[li];
                    const float4 bb = bbox_n[f];
                    hit = !(qxl > bb.y || qxh < bb.x || qyl > bb.w || qyh < bb.z);
                    if (hit) {
                        const float4 *q = (const float4 *)(rec_n + (size_t)f * REC + R_INV);
                        hit = tile_may_hit(q[0], q[1], q[2], 0.5f * (qxl + qxh), 0.5f * (qyl + qyh), 0.5f * (qxh - qxl), 0.5f * (qyh - qyl),
                                           A.thr + rec_n[(size_t)f * REC + R_CULL]);
                    }
                }
                const unsigned long long m = __ballot(hit);
                const unsigned m16 = (unsigned)(m >> (16 * ((tid >> 4) & 3))) & 0xffffu;
#if QP_LDS_REC
                if (hit) q_list[grp][qn + __builtin_popcount(m16 & ((1u << j) - 1u))] = li - c0;     // index into s_rec
#else
                if (hit) q_list[grp][qn + __builtin_popcount(m16 & ((1u << j) - 1u))] = f;
#endif
                qn += __builtin_popcount(m16);
            }
            if (j == 0) q_cnt[grp] = qn;
            __syncthreads();
            if (!have_perm) {   // deal the quadrants: descending list length, four of similar length per wave
                if (tid < 16) {
                    const int mine = q_cnt[tid];
                    int rank = 0;
#pragma unroll
                    for (int o = 0; o < 16; ++o) {
                        const int c = q_cnt[o];
                        rank += (c > mine) | ((c == mine) & (o < tid));
                    }
                    q_perm[rank] = tid;
                }
                __syncthreads();
                my_q = q_perm[grp];
                px = t.bx0 + 4 * (my_q & 3) + (tid & 3);
                row = t.by0 + 4 * (my_q >> 2) + ((tid >> 2) & 3);
                valid = px < IS && row < IS;
                have_perm = true;
            }
            const float xp = ndc_coord_fast(min(px, IS - 1), IS, inv_is, pow2), yp = ndc_coord_fast(IS - 1 - min(row, IS - 1), IS, inv_is, pow2);
            const int n_mine = q_cnt[my_q];
            const int w4 = (tid >> 6) * 4;
            const int nmax = max(max(q_cnt[q_perm[w4]], q_cnt[q_perm[w4 + 1]]), max(q_cnt[q_perm[w4 + 2]], q_cnt[q_perm[w4 + 3]]));
            const int *mylist = q_list[my_q];
            for (int i = 0; i < nmax; ++i) {
                const bool act = i < n_mine;
                const int f = mylist[act ? i : 0];
                FaceL fc;
#if QP_LDS_REC
                const int fi = act && n_mine > 0 ? f : 0;
                load_face_lane(fc, rec_n + (size_t)s_list[c0 + fi] * REC, s_rec[fi]);
#else
                load_face_lane(fc, rec_n + (size_t)(act && n_mine > 0 ? f : 0) * REC);
#endif
                Pair p;
                const bool live = eval_pair(p, fc, xp, yp, A.threshold, A.nis) & valid & act;
                alpha *= live ? 1.f - p.frag : 1.f;
            }
        }
    }
    if (!have_perm) {   // no candidate face at all: any dealing will do
        px = t.bx0 + 4 * (grp & 3) + (tid & 3);
        row = t.by0 + 4 * (grp >> 2) + ((tid >> 2) & 3);
        valid = px < IS && row < IS;
    }
    const float o3 = 1.f - alpha;
    if (valid) A.soft_colors[(size_t)t.n * npix + (size_t)row * IS + px] = o3;
    if (A.pooled) {
        const int H = IS >> 1;
        float sv = o3 + __shfl_xor(o3, 1, 64);
        sv += __shfl_xor(sv, 4, 64);
        if (valid && !(tid & 1) && !(tid & 4)) A.pooled[((size_t)t.n * H + (row >> 1)) * H + (px >> 1)] = 0.25f * sv;
    }
}

size_t x_bbox_bytes(int N, int F) { return (((size_t)N * F * sizeof(float4)) + 255) & ~(size_t)255; }
size_t x_rec_bytes(int N, int F) { return (size_t)N * F * REC * sizeof(float); }
size_t x_sbcount_bytes(int N) { return (((size_t)N * SB_SLOTS * sizeof(int)) + 255) & ~(size_t)255; }
int x_sb_cap(int F) { return F < SB_CAP ? F : SB_CAP; }

}  // namespace

extern "C" {

size_t umr_exp_qp_workspace_bytes(int N, int F) {
    return x_bbox_bytes(N, F) + x_rec_bytes(N, F) + x_sbcount_bytes(N) + (size_t)N * SB_SLOTS * x_sb_cap(F) * sizeof(int);
}

// silhouette forward, quadrant-packed: alpha [N,IS,IS] (+ pooled [N,IS/2,IS/2] or NULL); same scalars as umr_raster_forward
int umr_exp_sil_forward_qp(const float *faces, float *alpha, float *pooled, int N, int F, int image_size, float near_, float far_,
                           float sigma_val, float dist_eps, void *workspace, size_t workspace_bytes, void *stream) {
    if (!faces || !alpha || !workspace || N <= 0 || F <= 0 || image_size <= 0) return UMR_ERR_ARG;
    if (workspace_bytes < umr_exp_qp_workspace_bytes(N, F) || (pooled && (image_size & 1))) return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    RasterArgs A = {};
    A.bbox = (const float4 *)workspace;
    A.rec = (const float *)((char *)workspace + x_bbox_bytes(N, F));
    A.soft_colors = alpha; A.pooled = pooled;
    A.N = N; A.F = F; A.IS = image_size; A.TS = 1; A.R = 1;
    A.near_ = near_; A.far_ = far_; A.sigma = sigma_val;
    A.threshold = dist_eps * sigma_val;
    A.thr = sqrtf(A.threshold); A.nis = -1.f / sigma_val;
    A.tiles_x = (image_size + BLK_W - 1) / BLK_W;
    A.tiles_y = (image_size + BLK_H - 1) / BLK_H;
    A.no_xcd_remap = 2;
    A.tex_group = 1;
    const int total = N * F;
    UMR_LAUNCH(k_face_setup, (total + 63) / 64, 64, 0, st, faces, nullptr, (float4 *)workspace, (float *)A.rec, total, A.thr, near_, far_,
               nullptr, 0, THIN_FACE_H);
    char *p = (char *)workspace + x_bbox_bytes(N, F) + x_rec_bytes(N, F);
    int *cnt = (int *)p, *lst = (int *)(p + x_sbcount_bytes(N));
    const int sixteenth = (((image_size + 15) / 16) + 15) & ~15;
    A.sb_size = sixteenth > 64 ? sixteenth : 64;
    A.sb_nx = (image_size + A.sb_size - 1) / A.sb_size;
    A.sb_cap = x_sb_cap(F);
    UMR_LAUNCH(k_superblock_bin, dim3(A.sb_nx * A.sb_nx, N), 256, 0, st, A.bbox, cnt, lst, F, image_size, A.sb_size, A.sb_nx, A.sb_cap);
    A.sb_count = cnt; A.sb_list = lst;
    UMR_LAUNCH(k_sil_forward_qp, N * A.tiles_x * A.tiles_y, BLK_THREADS, 0, st, A);
    return umr_launch_status();
}

}  // extern "C"
