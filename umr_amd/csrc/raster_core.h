// raster_core.h -- shared device code of the soft rasterizer (included by raster.hip only): record layout, per-face
// preprocessing, the pixel/face geometry core (eval_pair), depth / texel helpers, tile culling and the forward's
// in-kernel binning.  See raster.hip for the execution plan and the reference line numbers.
#pragma once
#include "umr_common.h"

#define REC 64         // floats per preprocessed face record
#define LIST_CAP 2048  // LDS face list capacity (faces are processed in super-chunks of this many)
#ifndef BLK_W
#define BLK_W 16   // measured on MI355X (N=128, F=1280, IS=512, soft-max forward): 16x16 1.55 ms, 32x8 / 16x8 1.64,
#define BLK_H 16   // 32x16 1.81, 32x32 2.08, 8x8 2.28 -- 4 waves share one binning pass and still schedule finely
#endif
#ifndef THIN_FACE_H
#define THIN_FACE_H 1.6e-2f     // screen units: default of k_face_setup's thin_h (umr_debug_set("thin_face_h_1e6", ..) overrides)
#endif
#ifndef TILE_CULL_NOISE
#define TILE_CULL_NOISE 5e-6f   // constant of the per-face widening of the tile cull (k_face_setup -> R_CULL, tile_may_hit)
#endif
#define SB_SLOTS 256  // super-block slots per mesh in the workspace layout (<= 16 x 16 super-blocks of >= 64^2 pixels)
#define SB_CAP 1024   // list capacity per slot (entries); a slot that more faces touch is scanned in full (superblock_list)
#ifndef UMR_EDGE_SKIP
#define UMR_EDGE_SKIP 1     // eval_pair's doubt path: edge lines no lane in doubt can pick are not evaluated (A/B: -DUMR_EDGE_SKIP=0)
#endif
#ifndef UMR_REGION_VOTE
#define UMR_REGION_VOTE 0   // 1: eval_pair votes per visit and runs a specialised body when its lanes all lie inside / all outside
#endif                      // the face.  Measured on MI355X (profiles/r04_ab_region_vote.jsonl): the bodies are 15-25 % shorter, a
                            // visit is uniform too rarely at 8x8 / 4x4 granularity and the vote costs two VALU + a branch each
                            // way -- +-1 % on every kernel, silhouette forward 2 % slower.  Off; the bodies stay for a caller
                            // that KNOWS the side (REGION 1 / 2 of eval_pair_region).
#define BLK_WX (BLK_W / 8)                          // 8x8 wave tiles across / in the workgroup
#define BLK_THREADS (BLK_WX * (BLK_H / 8) * 64)

namespace {

// Record layout (floats) written by k_face_setup: 256 bytes per face.
//   [0,32)  wave-uniform part, fetched with 2 x s_load_dwordx16 into SGPRs.  Quantities that enter the geometry as 2-vectors
//           sit in EVEN-ALIGNED PAIRS so that they are operands of packed fp32 instructions straight from the scalar file:
//           on gfx950 a v_mul_f32 with an SGPR source issues in 4.2 cycles per wave, a v_pk_mul_f32 with an SGPR PAIR source in
//           4.3 -- two products for the price of one (tools/ubench/valu_ubench2.hip, profiles/r04_valu_ubench2.log)
//   [32,36) reciprocal depths (third scalar load, only in the kernels that interpolate depth)
//   [40,64) three 32-byte edge blocks {a0, a1, a2, a[v1], den, RN(1/den), -, -}; a lane reads ONLY the block
//           of its nearest edge (per-lane address, 2 x global_load_dwordx4, L1-resident)
enum { R_XLO = 0, R_XHI = 1, R_YLO = 2, R_YHI = 3,
       R_X0 = 4, R_Y0 = 5, R_X1 = 6, R_Y1 = 7, R_X2 = 8, R_Y2 = 9,       // corners as (x, y) pairs
       R_CX = 10, R_CY = 11,      // the obtuse corner (0, 0 without one): operand of the override test (:116,:119,:122)
       R_OX = 12, R_OY = 13,      // ... and the vector it is tested against: corner k -> p_{k+2} - p_k
       R_FLAGS = 14,              // bits 0-1 obtuse corner + 1 | 2 slow division | 3 depth always in range | 4 ill-conditioned / thin
                                  // | 5 front-facing
       R_CULL = 15,               // how far beyond sqrt(threshold) the reference still includes pixels (tile_may_hit)
       // inverse barycentric matrix, rows 0 and 1 interleaved: (inv0, inv3) (inv1, inv4) (inv2, inv5) -> (w0, w1) in three packed ops
       R_I0 = 16, R_I3 = 17, R_I1 = 18, R_I4 = 19, R_I2 = 20, R_I5 = 21, R_I6 = 22, R_I7 = 23, R_I8 = 24,
       R_K2 = 25, R_K0 = 26, R_K1 = 27,                                  // squared heights; (K0, K1) is a pair
       R_Z0 = 28, R_Z1 = 29, R_Z2 = 30,
       R_LUT = 31,                // the face's region table (eval_pair): 16 entries of 2 bits
       R_RZ0 = 32, R_RZ1 = 33, R_RZ2 = 34,
       R_EDGE = 40 };
// position of inv[i] (row-major 3 x 3, the reference's face_inv) in the record
__host__ __device__ constexpr int r_inv(int i) { return i == 0 ? R_I0 : i == 1 ? R_I1 : i == 2 ? R_I2 : i == 3 ? R_I3 : i == 4 ? R_I4 : i == 5 ? R_I5 : R_I6 + (i - 6); }

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
    // per-mesh coarse bins written by k_superblock_bin: the image is cut into <= 16 x 16 super-blocks of sb_size^2 pixels
    // (sb_size >= 64, a multiple of the 16-pixel workgroup block); sb_list[(n * sb_slots + sb) * sb_cap ...] holds, ascending,
    // the first sb_cap faces whose dilated bbox touches super-block sb, sb_count their TRUE number (> sb_cap: the list is
    // incomplete and the workgroup scans all F faces).  NULL = scan all F faces per workgroup.
    const int *sb_count;
    const int *sb_list;
    int sb_size, sb_nx, sb_cap, sb_slots;   // sb_slots = sb_nx^2: slots per mesh in sb_count / sb_list
    // face-major backward: the work items of the launch in start order (k_face_order).  order[(g * 8 + xcd) * order_stride + i] =
    // the i-th wave XCD `xcd` starts within mesh group g: .x = part << 26 | (parts - 1) << 21 | (mesh - g * order_group) << 16 |
    // face (0xffffffff: padding, the wave exits), .y = first slab of the face's partial sums (parts > 1).  A face whose estimated
    // work exceeds the split threshold is `parts` items -- each walks a contiguous share of the culling passes under the face's
    // bounding box and leaves its partial sums in slab .y + part; k_split_reduce (the next launch) adds them up IN PART ORDER and
    // stores the face's gradient: no wave owns more than a bounded share of a heavy face, results stay deterministic, no float
    // atomics.  NULL = one wave per face in index order.
    const uint2 *order;
    int order_group, order_stride;
    int fm_blocks;                  // workgroups of the face-major launch: N x F, or the lists' total length
    float *slab;                    // [slabs][slab_stride]: vertex gradients at [0, 9), texel gradients at [16, 16 + 3 TS)
    int slab_stride;
    int fm_split;      // runs of faces per XCD and mesh in the face-major backward (fm_owned_face); 0 / 1 = one
    int tex_group;    // K >= 1: mesh n samples textures[n / K] (K views share one texture set)
    float amb_thr;    // eval_pair: 0, or 20 sigma with umr_debug_set("exact_edges", 1)
    // read by the general-mode kernels only (raster_general.h): the reference's func_id_dist / func_id_alpha /
    // func_id_rgb and texture_sample_type
    int dist_mode, alpha_mode, rgb_mode, tex_vertex;
    float *vis;         // k_raster_forward<1, .., VIS>: hard z-buffer planes [N,2,IS,IS] = (nearest depth, its face id | -1),
                        // written next to the soft-max render of the same faces (umr_raster_forward_vis)
    int no_xcd_remap;   // A/B switch (umr_debug_set("xcd_remap", 0)): pixel-major work items in plain blockIdx order
    // forward: start order of the 16x16-pixel workgroup blocks (k_block_order).  block_order[(g * 8 + xcd) * block_group * per_mesh + i]
    // = (mesh - g * block_group) << 16 | block row << 8 | block column of the i-th workgroup XCD `xcd` starts within mesh group g
    // (per_mesh = tiles_y / 8 * tiles_x: XCD x keeps block rows x, x + 8, ... of every mesh); NULL = row by row, mesh by mesh.
    const int *block_order;
    int block_group;
    int bg_arg;       // background passed by value: soft_colors arrives uninitialised
    float bg0, bg1, bg2;
    // UMR_RASTER_PACKED_STATE / UMR_BWD_PACKED_STATE: the soft-max render's saved state as ONE tiled buffer instead of the planes
    // soft_colors[:, 3] + aggrs_info -- per mesh (IS/4)^2 records of 64 floats (256 B), one per 4x4 pixel tile, row-major over
    // tiles; in a record, pixel (x, y) of the tile sits at i = 4 y + x:
    //   [0,16)  v_rcp_f32 of the soft-max sum  (the backward uses the sum only through that reciprocal, :608)
    //   [16,32) soft-max maximum          [32,48) alpha
    //   [48,52) per 2x2 quad q = 2 (y / 2) + x / 2: smallest maximum of its four pixels (NaN if any of them is NaN)
    //   [52,56) per quad: 1.0f when all four alphas are exactly 1.0f, else 0.0f          [56,64) unused
    // What a backward wave touches of a face's neighbourhood is then whole records (a quad's state: 3 x 16 B of one record; a
    // culling lane's sub-tile: 32 B) instead of 8-byte pieces of 6 - 8 rows of three planes.
    float *state;
    int vis_ids_only;   // k_raster_forward<.., VIS>: `vis` is [N,IS,IS], the face-id plane alone (UMR_RASTER_VIS_IDS_ONLY)
};
#define STATE_REC 64          // floats per 4x4-tile record of the packed state
#define STATE_O_MAX 16
#define STATE_O_ALPHA 32
#define STATE_O_QMIN 48
#define STATE_O_QOPAQUE 52

// ---- per-face preprocessing (:223-282) + packed record for the raster kernels ----------------
// (one face: everything but the record goes straight to memory; the record's 64 floats go to `r`, which the kernel below
// points at a row of LDS)
__device__ __forceinline__ void face_setup_one(int i, const float *__restrict__ faces, float *__restrict__ faces_info,
                                               float4 *__restrict__ bbox, float *r, float thr, float near_, float far_,
                                               float thin_h) {
    const float *f = faces + (size_t)i * 9;
    const float x0 = f[0], y0 = f[1], z0 = f[2], x1 = f[3], y1 = f[4], z1 = f[5], x2 = f[6], y2 = f[7], z2 = f[8];
    UMR_TRAP_IF(umr_bad(x0) | umr_bad(y0) | umr_bad(z0) | umr_bad(x1) | umr_bad(y1) | umr_bad(z1) | umr_bad(x2) | umr_bad(y2) | umr_bad(z2), 1);
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
    // ---- packed record ----
#pragma unroll
    for (int k = 0; k < REC; ++k) r[k] = 0.f;
    r[R_XLO] = xlo; r[R_XHI] = xhi; r[R_YLO] = ylo; r[R_YHI] = yhi;
    r[R_X0] = x0; r[R_Y0] = y0; r[R_X1] = x1; r[R_Y1] = y1; r[R_X2] = x2; r[R_Y2] = y2;
    r[R_Z0] = z0; r[R_Z1] = z1; r[R_Z2] = z2;
    r[R_RZ0] = 1.f / z0; r[R_RZ1] = 1.f / z1; r[R_RZ2] = 1.f / z2;  // correctly rounded (Markstein division)
#pragma unroll
    for (int k = 0; k < 9; ++k) r[r_inv(k)] = inv[k];
    // squared height of corner c over its opposite edge: inside the triangle the squared distance to that
    // edge's line is w_c^2 K_c -- used only to PICK the nearest edge (:99), the distance itself is then
    // evaluated with the reference's own formula
    const float det_raw = x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0);
    float kh[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int a = (c + 1) % 3, b = (c + 2) % 3;
        const float ex = px[a] - px[b], ey = py[a] - py[b];
        kh[c] = det_raw * det_raw / fmaxf(ex * ex + ey * ey, 1e-30f);
    }
    r[R_K0] = kh[0]; r[R_K1] = kh[1]; r[R_K2] = kh[2];
    // depth chain may use reciprocal-multiply division only when every z is an ordinary positive number
    const bool sane = z0 > 1e-20f && z1 > 1e-20f && z2 > 1e-20f && z0 < 1e20f && z1 < 1e20f && z2 < 1e20f;
    // bit 3: every vertex depth strictly inside (near, far) => the interpolated depth (a convex combination of the
    // 1/z_k with positive clipped weights) can never be rejected by the depth-range test (:404, :592)
    const float zmin = fminf(fminf(z0, z1), z2), zmax = fmaxf(fmaxf(z0, z1), z2);
    const bool inrange = sane && zmin > near_ * 1.0001f && zmax < far_ * 0.9999f;
    const bool front = (y2 - y0) * (x1 - x0) < (y1 - y0) * (x2 - x0);  // :42-44
    // The reference's region if-chain (:112-126) as a table of the region code m = n0 | n1 << 1 | n2 << 2 (n_k = w_k <= 0), entries
    // v0 + 1 in 2 bits: no flag -> -1 (the reference's out-of-bounds case, defined as "skip"); one flag: the opposite edge (n0 -> 1,
    // n1 -> 2, n2 -> 0); two flags: the vertex region between them ({n0,n1} -> 2, {n2,n0} -> 1, {n1,n2} -> 0); all three
    // (degenerate faces only) ends like {n1,n2}.  Entries 8 + m: the same with the override of the obtuse corner applied
    // (:116,:119,:122: in ITS vertex region the other edge is taken when the pixel lies beyond it) -- corner 0: region {n1,n2} = 6
    // and 7 -> edge 2; corner 1: {n2,n0} = 5 -> edge 0; corner 2: {n0,n1} = 3 -> edge 1.  eval_pair indexes with m + 8 * override.
    unsigned lut = 0;
    {
        const int base[8] = {-1, 1, 2, 2, 0, 1, 0, 0};
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            int v = base[m];
            lut |= (unsigned)(v + 1) << (2 * m);
            if (obt == 0 && (m == 6 || m == 7)) v = 2;
            if (obt == 1 && m == 5) v = 0;
            if (obt == 2 && m == 3) v = 1;
            lut |= (unsigned)(v + 1) << (2 * (8 + m));
        }
    }
    r[R_LUT] = __int_as_float((int)lut);
    int flags = (obt + 1) | (sane ? 0 : 4) | (inrange ? 8 : 0) | (front ? 32 : 0);
    const int ob = obt < 0 ? 0 : obt;
    r[R_CX] = obt < 0 ? 0.f : px[ob]; r[R_CY] = obt < 0 ? 0.f : py[ob];
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
        // bit 4: some edge is shorter than ~3e-3 screen units (0.8 px at IS = 512): its `den` -- a squared length obtained
        // by cancellation of O(1) terms -- is rounding noise, possibly exactly 0.  eval_pair then evaluates the inside
        // branch the reference's way (all three edge lines, smallest computed distance).  NaN den: flagged as well.
        if (!(fabsf(den) >= 1e-5f)) flags |= 16;
    }
    // ... and so does a THIN face (a height below thin_h screen units).  Inside a triangle the reference keeps the edge LINE with
    // the smallest COMPUTED distance (:78-107); eval_pair picks the line by its true distance w_c^2 K_c first and evaluates
    // only that one the reference's way.  The two agree unless two line distances tie within the rounding noise of the
    // reference's formulation (~1e-5 .. 1e-4 screen units, growing as edges get shorter) -- a zone of that width around the
    // bisectors, i.e. a fraction noise / height of a face's interior: 3e-5 of the inside pixels of a face of 0.05 units, 1 %
    // at 0.01, a quarter at 1e-3 (measured on this source compiled for the host, tests/host_kernel), and there the soft
    // fragment differs by up to 0.1 and the gradient goes to another pair of vertices.  Flagged faces take the reference's
    // route whenever a lane is inside; the cost is theirs alone.
    if (!(fminf(fminf(kh[0], kh[1]), kh[2]) >= thin_h * thin_h)) flags |= 16;
    r[R_FLAGS] = __int_as_float(flags);
    // Widening of the tile cull (tile_may_hit).  Outside the triangle the reference computes the closest point through an edge
    // parameter t = (w . a - a[v1]) / den whose operands are differences of O(1 + |p|^2) products (:82-84, :133-137): its
    // rounding error, times |w| ~ threshold distance / height, divided by den = |edge|^2, times |edge| again for the point.  So
    // pixels up to ~ c (1 + |p|^2) (1 + thr / h_min) / L_min BEYOND sqrt(threshold) in exact geometry still pass the reference's
    // own reject (:382) with D ~ 1e-10 -- invisible in alpha, but such a fragment moves the running soft-max maximum (and with
    // it the p2f weights of later faces) and, outside the silhouette where every weight is that small, colour and texel
    // gradients by O(1).  c measured on this source compiled for the host (tools/reference_noise.py cull): 1.8e-6 for faces on
    // the screen, 4.8e-6 up to 2.5 screen half-widths out; 5e-6 here.  0.04 px for a BASELINE face at IS = 512 (h 0.07, L 0.08),
    // a pixel for a sliver of 0.001 units; a degenerate face (h = 0 or L = 0) gets inf and is culled by its box alone.
    float k_min = fminf(fminf(kh[0], kh[1]), kh[2]), l2_min = 3.0e38f, p2_max = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int a = (c + 1) % 3;
        const float ex = px[a] - px[c], ey = py[a] - py[c];
        l2_min = fminf(l2_min, ex * ex + ey * ey);
        p2_max = fmaxf(p2_max, px[c] * px[c] + py[c] * py[c]);
    }
    r[R_CULL] = TILE_CULL_NOISE * (1.f + p2_max) * (1.f + thr / sqrtf(k_min)) / sqrtf(l2_min);
}

// One thread per face, 64 faces per workgroup.  A thread that wrote its 256-byte record itself issued 64 dword stores 256 bytes
// apart from its neighbours' (7.7 us per 20 480 faces, four such launches per train_s1 step); the records are staged in LDS
// instead (row stride 65 floats: thread i's k-th word lands in bank (i + k) % 64) and the workgroup writes them out as 64 rows of
// 256 contiguous bytes.
#define FACE_SETUP_THREADS 64
__global__ __launch_bounds__(FACE_SETUP_THREADS) void k_face_setup(const float *__restrict__ faces, float *__restrict__ faces_info,
                             float4 *__restrict__ bbox, float *__restrict__ rec, int total, float thr,
                             float near_, float far_, float thin_h = 0.f) {
    __shared__ float s_rec[FACE_SETUP_THREADS][REC + 1];
    const int base = blockIdx.x * FACE_SETUP_THREADS, i = base + (int)threadIdx.x;
    if (i < total) face_setup_one(i, faces, faces_info, bbox, s_rec[threadIdx.x], thr, near_, far_, thin_h);
    __syncthreads();
    const int rows = min(FACE_SETUP_THREADS, total - base);
    float *out = rec + (size_t)base * REC;
    for (int row = 0; row < rows; ++row) out[(size_t)row * REC + threadIdx.x] = s_rec[row][threadIdx.x];
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
// ... plus a compile-time displacement, added in the 64-bit address domain so that it becomes the instruction's immediate offset
// (`pn4 + 64` in 32 bits may wrap as far as the compiler knows: it then spends a v_add_u32 per load)
template <unsigned IMM> __device__ __forceinline__ float ld_ui(const char *base, unsigned byte_off) {
    return *(const float *)(base + ((size_t)byte_off + IMM));
}

// Wave votes over the lanes that reach them (EXEC), straight on the lane mask: HIP's __all / __any take an int and cost a
// select and a compare per vote.  The emulation build (tests/host_kernel) maps them to its subset vote.
#ifdef UMR_HOST_SHIM
__device__ __forceinline__ bool wave_all(bool p) { return __all(p); }
__device__ __forceinline__ bool wave_any(bool p) { return !__all(!p); }
#define UMR_KEEP_BRANCH() ((void)0)
#else
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
#define UMR_KEEP_BRANCH() asm volatile("" ::: "memory")   // a rare wave-uniform path stays a branch (no if-conversion into selects)
#endif

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4f cv4f_t;
__device__ __forceinline__ v2f splat2(float a) { return (v2f){a, a}; }

struct Face {  // wave-uniform: 32 (+ 4) SGPRs + the record's address
    v16f qa, qb;
    v4f qc;               // reciprocal depths: loaded only where a kernel reads them (an unused load is dropped by the compiler)
    const char *edges;    // three 32-byte edge blocks (global memory, read per lane: uniform base + k * 32)
    template <int I> __device__ __forceinline__ float g() const {
        if constexpr (I < 16) return qa[I];
        else if constexpr (I < 32) return qb[I - 16];
        else return qc[I - 32];
    }
    // an even-aligned pair (I, I + 1) of the record as a 2-vector: an SGPR pair, i.e. one source operand of a packed fp32 op
    template <int I> __device__ __forceinline__ v2f g2() const {
        static_assert((I & 1) == 0 && I < 32, "pairs are even-aligned and live in the first 32 floats");
        if constexpr (I < 16) return __builtin_shufflevector(qa, qa, I, I + 1);
        else return __builtin_shufflevector(qb, qb, I - 16, I - 15);
    }
    template <int I> __device__ __forceinline__ float inv() const { return g<r_inv(I)>(); }   // the reference's face_inv[I]
    // barycentrics (:25-29) and the offset sum (:95-96, :148-149) in the reference's operation order, rounded once per
    // operation; here two rows / both components at a time as packed ops on SGPR pairs
    __device__ __forceinline__ void bary(float &w0, float &w1, float &w2, float xp, float yp) const {
        const v2f w01 = (g2<R_I0>() * splat2(xp) + g2<R_I1>() * splat2(yp)) + g2<R_I2>();
        w0 = w01.x; w1 = w01.y;
        w2 = (g<R_I6>() * xp + g<R_I7>() * yp) + g<R_I8>();
    }
    __device__ __forceinline__ void offset(float &dx, float &dy, float t0, float t1, float t2) const {
        const v2f d = (splat2(t0) * g2<R_X0>() + splat2(t1) * g2<R_X1>()) + splat2(t2) * g2<R_X2>();
        dx = d.x; dy = d.y;
    }
    __device__ __forceinline__ int flags() const { return __float_as_int(g<R_FLAGS>()); }
    __device__ __forceinline__ int obt() const { return (flags() & 3) - 1; }
    __device__ __forceinline__ bool front() const { return (flags() & 32) != 0; }
    __device__ __forceinline__ bool slow() const { return (flags() & 4) != 0; }
    __device__ __forceinline__ bool depth_in_range() const { return (flags() & 8) != 0; }
    __device__ __forceinline__ bool ill_conditioned() const { return (flags() & 16) != 0; }
    __device__ __forceinline__ unsigned region_lut() const { return (unsigned)__float_as_int(g<R_LUT>()); }
};

// + 15 VGPRs per lane, filled once per face by the face-major backward (a wave owns one face for all its visits): the operands
// of the barycentric rows and of the offset sums as VGPR copies.  On gfx950 a VALU instruction with an SGPR source issues at half
// the rate of the same instruction on VGPRs (v_mul_f32 2.6 vs 4.2 cycles per wave-instruction); plain full-rate ops on these copies
// measured faster in the backward than packed ops on them (dependent packed ops want a wait state in between).
struct FaceV : Face {
    float vinv[9], vxy[6];
    __device__ __forceinline__ void bary(float &w0, float &w1, float &w2, float xp, float yp) const {
        w0 = (vinv[0] * xp + vinv[1] * yp) + vinv[2];
        w1 = (vinv[3] * xp + vinv[4] * yp) + vinv[5];
        w2 = (vinv[6] * xp + vinv[7] * yp) + vinv[8];
    }
    __device__ __forceinline__ void offset(float &dx, float &dy, float t0, float t1, float t2) const {
        dx = (t0 * vxy[0] + t1 * vxy[2]) + t2 * vxy[4];
        dy = (t0 * vxy[1] + t1 * vxy[3]) + t2 * vxy[5];
    }
    __device__ __forceinline__ void fill() {
#ifdef UMR_HOST_SHIM   // tests/host_kernel: the same copies without the instruction
#define UMR_VMOV(dst, src) dst = (src)
#else
#define UMR_VMOV(dst, src) asm volatile("v_mov_b32 %0, %1" : "=v"(dst) : "s"(src))
#endif
        UMR_VMOV(vinv[0], inv<0>()); UMR_VMOV(vinv[1], inv<1>()); UMR_VMOV(vinv[2], inv<2>());
        UMR_VMOV(vinv[3], inv<3>()); UMR_VMOV(vinv[4], inv<4>()); UMR_VMOV(vinv[5], inv<5>());
        UMR_VMOV(vinv[6], inv<6>()); UMR_VMOV(vinv[7], inv<7>()); UMR_VMOV(vinv[8], inv<8>());
        UMR_VMOV(vxy[0], g<R_X0>()); UMR_VMOV(vxy[1], g<R_Y0>()); UMR_VMOV(vxy[2], g<R_X1>());
        UMR_VMOV(vxy[3], g<R_Y1>()); UMR_VMOV(vxy[4], g<R_X2>()); UMR_VMOV(vxy[5], g<R_Y2>());
#undef UMR_VMOV
    }
};

__device__ __forceinline__ void load_face(Face &fc, const float *rg) {
    cv16f_t *r = (cv16f_t *)rg;
    fc.qa = r[0]; fc.qb = r[1];
    fc.qc = *(cv4f_t *)(rg + R_RZ0);
    fc.edges = (const char *)(rg + R_EDGE);
}

// the 32 wave-uniform floats again (face-major backward: keeps them loop-VARIANT, see FM_RELOAD_PER_TILE)
__device__ __forceinline__ void reload_face(Face &fc, const float *rg) {
    cv16f_t *r = (cv16f_t *)rg;
    fc.qa = r[0]; fc.qb = r[1];
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

__device__ __forceinline__ float fmed3_(float a, float b, float c) {   // v_med3_f32
#ifdef UMR_HOST_SHIM
    return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
#else
    return __builtin_amdgcn_fmed3f(a, b, c);
#endif
}

// bbox reject (:355), barycentric (:25-29), euclidean distance (:63-152), threshold reject (:382),
// sigmoid (:383).  Returns false when the reference would `continue` before touching the pixel.
//
// Arithmetic: every product and sum is the reference's, in the reference's order, rounded once each (-ffp-contract=off); where
// two of them are independent and their scalar operands sit in an aligned pair of the record they are issued as ONE packed
// instruction (v_pk_mul_f32 / v_pk_add_f32 are per-component IEEE operations: the bits are those of the two scalar ops).
//
// REGION (compile time; raster kernels pick it per visit with one wave vote, see eval_pair_voted):
//   0  lanes on both sides of the boundary: everything below
//   1  every ACTIVE lane is inside the triangle: the outside branch (:110-151) is dead -- no region table, no override test, no
//      clamps, no threshold reject -- the result for each lane is what REGION 0 gives it
//   2  every ACTIVE lane is outside: the inside branch (:68-109) is dead -- no nearest-line pick, no doubt path
template <int REGION, class FaceT>
__device__ __forceinline__ bool eval_pair_region(Pair &p, const FaceT &fc, float xp, float yp, float w0, float w1, float w2, bool inside,
                                                 float threshold, float neg_inv_sigma, float amb_thr) {
    // Written branch-free (predicates + selects): per-lane divergence would otherwise cost ~80 scalar
    // exec-mask instructions per face visit.  Dead lanes compute garbage that the returned predicate masks.
    const bool inb = !((xp > fc.template g<R_XHI>()) | (xp < fc.template g<R_XLO>()) | (yp > fc.template g<R_YHI>()) | (yp < fc.template g<R_YLO>()));
    const v2f w01 = (v2f){w0, w1};
    int ksel;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    if (REGION != 2) {
        // inside: nearest edge LINE, first minimum in the reference's order k = 0,1,2 (:78-107); edge k is opposite
        // corner k+2 and its squared distance is w_c^2 K_c
        const v2f m12 = (w01 * w01) * fc.template g2<R_K0>();
        m0 = w2 * w2 * fc.template g<R_K2>(); m1 = m12.x; m2 = m12.y;
        const bool c1 = m1 < m0;
        const float best = c1 ? m1 : m0;
        ksel = (m2 < best) ? 2 : (c1 ? 1 : 0);
    }
    if (REGION != 1) {
        // outside: region selection (:112-126) through the face's table (k_face_setup): index = region code m = n0 | n1 << 1 |
        // n2 << 2 with n_k = (w_k <= 0), + 8 when the pixel lies beyond the obtuse corner's far edge (the override test)
        const v2f pr = ((v2f){xp, yp} - fc.template g2<R_CX>()) * fc.template g2<R_OX>();   // (xp - cx) ox, (yp - cy) oy
        const bool ovr = pr.x + pr.y > 0;
        const int m = ((w0 <= 0 ? 1 : 0) | (w1 <= 0 ? 2 : 0)) | ((w2 <= 0 ? 4 : 0) | (ovr ? 8 : 0));
        const int kout = (int)((fc.region_lut() >> (2 * m)) & 3u) - 1;
        ksel = REGION == 2 ? kout : (inside ? ksel : kout);
    }
    const bool kvalid = ksel >= 0;  // k = -1: reference UB (index -1); defined here and in the oracle as "skip"
    // t[v0] = (w . a - a[v1]) / (a[v0] - a[v1]) in the reference's operation order (:86,:137); IEEE-exact
    // quotient through Markstein's correction.  Far from the silhouette the soft-max renormalises weights
    // D ~ exp(-d^2/sigma) ~ 1e-9, amplifying rounding noise in d^2 ~20x: parity there needs the reference's
    // own noise, i.e. its own arithmetic, not just the same formula.
    auto edge_param = [&](int kk) {   // {a0,a1,a2,a[v1]}, {den, 1/den, -, -} of edge kk -> t[v0] = num / den (:86,:137)
        const unsigned ko = (unsigned)kk * 32u;
        const float4 ea = ld_u4(fc.edges, ko), eb = ld_u4(fc.edges, ko + 16u);
        const float num = ((w0 * ea.x + w1 * ea.y) + w2 * ea.z) - ea.w;
        // den = 0 (see below): the IEEE quotient is +-inf (NaN for 0 / 0); Markstein's correction would turn inf into NaN
        return fabsf(eb.y) <= 3.0e38f ? div_r(num, eb.x, eb.y) : num * eb.y;
    };
    int k = REGION == 1 ? ksel : max(ksel, 0);
    float tv = edge_param(k);
    // An edge whose screen-space length is below ~3e-4 (an edge seen end-on) loses its squared length in the cancellation of
    // :82-84 -- den comes out as exactly 0 about every second time, as noise otherwise -- and t[v0] is inf, NaN or garbage.
    // Outside the triangle the clamp below absorbs that exactly as in the reference.  INSIDE, the reference evaluates all
    // three edge lines and keeps the smallest COMPUTED distance with `dis < dis_min` (:78-107) -- false for NaN and inf, so a
    // collapsed edge is never the winner, and among garbage distances the smallest garbage wins.  Picking the edge
    // beforehand by its true line distance (above) is equivalent only while the three `den` are well conditioned; faces
    // flagged by k_face_setup (wave-uniform) take the reference's own route here.  With no usable edge the reference ends
    // with dis_x = dis_y = 0 (:72-73,:105-106).
    // amb_thr > 0 (umr_debug_set("exact_edges", 1); RasterArgs::amb_thr = 20 sigma): ANY face takes this route for the lanes
    // where the choice of the edge can matter and can be in doubt -- inside, with the SECOND nearest edge line closer than
    // sqrt(20 sigma).  Elsewhere the fast pick is provably the reference's result: beyond 17.4 sigma the fragment is 1.0f
    // whichever line is taken (e^-17.4 < 2^-25; its gradient factor 1 - D is then 0.0f), and a nearest line inside 17.4 sigma with
    // the next one beyond 20 sigma leaves a gap of 2.6 sigma ~ 0.1 px at the band's edge, two orders above the noise of the
    // computed distances of faces that are not flagged thin.
    bool no_edge = false;
    if (REGION != 2 && (fc.ill_conditioned() | (amb_thr > 0.f))) {          // wave-uniform: the default build pays nothing for unflagged faces
        const bool doubt = inside & (fc.ill_conditioned() | (fmed3_(m0, m1, m2) < amb_thr));
        if (doubt) {
            float dmin = 100000000.f;
            int kb = -1;
            float tbest = 0.f;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                // An edge line that is >= amb_thr away (true distance) from EVERY lane in doubt cannot win in any of them (a lane in
                // doubt has its two nearest lines inside amb_thr; a third one beyond it loses by more than the noise of the computed
                // distances, or all three lie beyond 17.4 sigma and the fragment is 1.0f whichever wins -- the argument above):
                // skipped for the whole wave.  Flagged faces evaluate all three.  (Typically one of the three is skipped: -5 % time)
                const float mk = kk == 0 ? m0 : (kk == 1 ? m1 : m2);
                if (UMR_EDGE_SKIP && !wave_any(fc.ill_conditioned() | (mk < amb_thr))) continue;
                const float tk = edge_param(kk), uk = 1.f - tk;
                const float c0 = kk == 0 ? tk : (kk == 1 ? 0.f : uk), c1 = kk == 0 ? uk : (kk == 1 ? tk : 0.f),
                            c2 = kk == 0 ? 0.f : (kk == 1 ? uk : tk);
                const float s0 = c0 - w0, s1 = c1 - w1, s2 = c2 - w2;
                float ex, ey;
                fc.offset(ex, ey, s0, s1, s2);
                const float dk = ex * ex + ey * ey;
                if (dk < dmin) { dmin = dk; kb = kk; tbest = tk; }
            }
            no_edge = kb < 0;
            k = no_edge ? 0 : kb;
            tv = no_edge ? 0.f : tbest;
        }
    }
    const bool k0 = k == 0, k1 = k == 1;
    const float tb = 1.f - tv;
    float ba, bb;
    if (REGION == 1) { ba = tv; bb = tb; }                                            // unclamped inside (:86-88)
    else if (REGION == 2) { ba = fminf(fmaxf(tv, 0.f), 1.f); bb = fminf(fmaxf(tb, 0.f), 1.f); }   // clamped outside (:142-145)
    else { ba = inside ? tv : fminf(fmaxf(tv, 0.f), 1.f); bb = inside ? tb : fminf(fmaxf(tb, 0.f), 1.f); }
    float b0 = k0 ? ba : (k1 ? 0.f : bb);
    float b1 = k0 ? bb : (k1 ? ba : 0.f);
    float b2 = k0 ? 0.f : (k1 ? bb : ba);
    if (REGION != 2 && wave_any(no_edge)) {   // (degenerate faces only) closest point := the pixel itself -> dis_x = dis_y = 0
        UMR_KEEP_BRANCH();
        b0 = no_edge ? w0 : b0; b1 = no_edge ? w1 : b1; b2 = no_edge ? w2 : b2;
    }
    const float t0 = b0 - w0, t1 = b1 - w1, t2 = b2 - w2;
    float dx, dy;
    fc.offset(dx, dy, t0, t1, t2);  // :95-96, :148-149
    const float dis = dx * dx + dy * dy;
    // the gradient's barycentrics are the reference's `t[k] + w[k]` (:640) with t[k] = b_k - w_k already rounded: for an
    // ordinary face that is b_k to an ulp, for a degenerate one (|w| ~ 1e9) whatever survives the cancellation -- like there
    p.b0 = t0 + w0; p.b1 = t1 + w1; p.b2 = t2 + w2; p.dx = dx; p.dy = dy;
    const bool in_ = REGION == 1 ? true : (REGION == 2 ? false : inside);
    p.sign = in_ ? 1.f : -1.f;
    // 1 / (1 + exp(-sign * dis / sigma))
    const float e = __expf((in_ ? dis : -dis) * neg_inv_sigma);
    p.frag = __builtin_amdgcn_rcpf(1.f + e);
    if (REGION == 1) return inb;
    return inb & kvalid & (in_ | !(dis >= threshold));  // rejects of :355, :382
}

// `active`: the lanes whose result the caller will use (a valid pixel of the tile / sub-tile); the others may hold anything.
// One wave vote picks the specialised body when all active lanes lie on one side of the face's boundary.
template <class FaceT>
__device__ __forceinline__ bool eval_pair(Pair &p, const FaceT &fc, float xp, float yp, float threshold,
                                          float neg_inv_sigma, float amb_thr = 0.f, bool active = true) {
    // barycentrics in the reference's operation order (no FMA): they decide inside/outside, feed the depth
    // chain and -- through cancellation -- carry ~1e-6 of rounding noise that has to match the reference's
    float w0, w1, w2;
    fc.bary(w0, w1, w2, xp, yp);
    p.w0 = w0; p.w1 = w1; p.w2 = w2;
    // 0 < w_k < 1 for all k (:68-69) on the bit patterns: a float in (0, 1) is an integer in [1, 0x3f7fffff]; zeros, negatives
    // (sign bit), 1.0 and above, infinities and NaNs of either sign all fall outside after the wrapping decrement.  Three
    // full-rate subtractions, one three-operand max and one compare instead of six compares and five mask ANDs.
    const unsigned u0 = (unsigned)__float_as_int(w0) - 1u, u1 = (unsigned)__float_as_int(w1) - 1u, u2 = (unsigned)__float_as_int(w2) - 1u;
    const bool inside = max(max(u0, u1), u2) < 0x3f7fffffu;
#if UMR_REGION_VOTE
    if (wave_all(inside | !active)) return eval_pair_region<1>(p, fc, xp, yp, w0, w1, w2, true, threshold, neg_inv_sigma, amb_thr);
    if (!wave_any(inside & active)) return eval_pair_region<2>(p, fc, xp, yp, w0, w1, w2, false, threshold, neg_inv_sigma, amb_thr);
#endif
    return eval_pair_region<0>(p, fc, xp, yp, w0, w1, w2, inside, threshold, neg_inv_sigma, amb_thr);
}

// barycentric_clip (:54-59) + perspective-correct depth (:403).  The soft-max weights are exp(zn/gamma) with
// gamma = 1e-4: one ulp of zn moves a weight by ~7e-4, so this chain reproduces the reference's IEEE
// divisions to the last bit (Markstein-corrected reciprocal multiplies; plain IEEE when z is degenerate).
__device__ __forceinline__ float clip_depth(float &c0, float &c1, float &c2, const Pair &p, const Face &fc) {
    // clamp to [1e-5, 1 - 1e-5] (:56-57) as ONE v_med3_f32 each: fminf / fmaxf cost a canonicalising v_max_f32 x, x in front
    // (three half-rate instructions per weight).  Same value for every non-NaN weight; a NaN weight never gets here (no region
    // flag of a NaN is set: table entry 0, the pair is skipped; the visibility-only kernel tests 0 <= w <= 1 first)
    c0 = fmed3_(p.w0, 1e-5f, 1.f - 1e-5f);
    c1 = fmed3_(p.w1, 1e-5f, 1.f - 1e-5f);
    c2 = fmed3_(p.w2, 1e-5f, 1.f - 1e-5f);
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
__device__ __forceinline__ bool tile_may_hit(const float4 q0, const float4 q1, const float4 q2, float cx, float cy,
                                             float hx, float hy, float thr) {
    // the record's floats [R_I0, R_I0 + 12): q0 = {inv0, inv3, inv1, inv4}, q1 = {inv2, inv5, inv6, inv7}, q2 = {inv8, K2, K0, K1}
    const float w0 = fmaf(q0.x, cx, fmaf(q0.z, cy, q1.x)) + (hx * fabsf(q0.x) + hy * fabsf(q0.z));
    const float w1 = fmaf(q0.y, cx, fmaf(q0.w, cy, q1.y)) + (hx * fabsf(q0.y) + hy * fabsf(q0.w));
    const float w2 = fmaf(q1.z, cx, fmaf(q1.w, cy, q2.x)) + (hx * fabsf(q1.z) + hy * fabsf(q1.w));
    // `thr` = sqrt(threshold) + the face's R_CULL: the reference decides the threshold reject (:382) on ITS computed distance, so
    // the band has to be wider than the exact one by that distance's rounding noise (k_face_setup).
    const bool out = w0 < -(thr * __frsqrt_rn(q2.z)) - 1e-3f || w1 < -(thr * __frsqrt_rn(q2.w)) - 1e-3f ||
                     w2 < -(thr * __frsqrt_rn(q2.y)) - 1e-3f;
    // A face one of whose heights is below ~3e-5 screen units (a sliver, a needle, a face seen edge-on) is not culled beyond
    // its bounding box: its barycentrics carry rounding noise of the size of this very test, and what the reference's
    // arithmetic makes of such a face (soft fragments up to 0.5 along its line) follows that noise, not the geometry.
    // (Found by fuzzing this source on the host, tests/test_kernel_source_on_host.py.)
    const bool sliver = !((q2.z >= 1e-9f) & (q2.w >= 1e-9f) & (q2.y >= 1e-9f));
    return sliver | !out;
}

// tile_may_hit for a 4x4 sub-tile AND its four 2x2 quads (power-of-two image: no ragged sub-tile): returns bit q set when quad
// (q & 1, q >> 1) -- columns 2 (q & 1) .. +1, rows 2 (q >> 1) .. +1 of the sub-tile -- may hold a surviving pixel centre, 0 when the
// sub-tile as a whole cannot.  w_k is affine, so a quad's bound is the sub-tile centre's w_k moved by one pixel each way plus the
// half-pixel extent: three adds per (edge, quad) instead of a fresh evaluation.  `px` = one pixel in screen units (2 / IS).
__device__ __forceinline__ unsigned subtile_quads_may_hit(const float4 q0, const float4 q1, const float4 q2, float cx, float cy,
                                                         float px, float thr) {
    const float t0 = -(thr * __frsqrt_rn(q2.z)) - 1e-3f, t1 = -(thr * __frsqrt_rn(q2.w)) - 1e-3f, t2 = -(thr * __frsqrt_rn(q2.y)) - 1e-3f;
    const bool sliver = !((q2.z >= 1e-9f) & (q2.w >= 1e-9f) & (q2.y >= 1e-9f));
    const float c0 = fmaf(q0.x, cx, fmaf(q0.z, cy, q1.x)), c1 = fmaf(q0.y, cx, fmaf(q0.w, cy, q1.y)), c2 = fmaf(q1.z, cx, fmaf(q1.w, cy, q2.x));
    // whole sub-tile: half extents 1.5 pixels
    const float a0 = px * q0.x, b0 = px * q0.z, a1 = px * q0.y, b1 = px * q0.w, a2 = px * q1.z, b2 = px * q1.w;
    const float e0 = fabsf(a0) + fabsf(b0), e1 = fabsf(a1) + fabsf(b1), e2 = fabsf(a2) + fabsf(b2);
    const bool out = fmaf(1.5f, e0, c0) < t0 || fmaf(1.5f, e1, c1) < t1 || fmaf(1.5f, e2, c2) < t2;
    if (sliver) return 15u;
    if (out) return 0u;
    // quads: centre one pixel off the sub-tile's centre each way (rows grow downwards: row pair 0 is +y), half extent 0.5 pixel
    const float m0 = fmaf(0.5f, e0, c0), m1 = fmaf(0.5f, e1, c1), m2 = fmaf(0.5f, e2, c2);
    unsigned m = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float sx = (q & 1) ? 1.f : -1.f, sy = (q >> 1) ? -1.f : 1.f;
        const bool o = (m0 + sx * a0) + sy * b0 < t0 || (m1 + sx * a1) + sy * b1 < t1 || (m2 + sx * a2) + sy * b2 < t2;
        m |= o ? 0u : 1u << q;
    }
    return m;
}

// XCD-aware work mapping: hardware places workgroup b on XCD b % 8; give each XCD a contiguous
// run of (mesh, tile) work items so one mesh's face records stay in one L2.
__device__ __forceinline__ int xcd_remap(int b, int total) {
    return (total % 8 == 0) ? (b % 8) * (total / 8) + b / 8 : b;
}

// Face ownership of the face-major backward: XCD `xcd` owns, of every mesh, `split` contiguous runs of per / split faces
// (run q of XCD x = chunk q * 8 + x of the mesh's 8 * split chunks); j = 0 .. per-1 enumerates them.  Contiguous index runs
// are spatial patches of a subdivided mesh, so the saved state an XCD's waves re-read stays in its L2; but with one run
// per XCD whole patches are front- or back-facing and the XCDs' loads differ 0.3x .. 1.8x per mesh.  Measured (N = 16,
// split 1 -> 4): texel gradients only 122.9 -> 119.0 us, silhouette 77.7 -> 74.1 us, vertex + texel gradients 204 -> 228 us
// (28 B of state per pixel: locality wins); no gain at N = 128.  raster.hip picks 4 for the two light variants at N <= 16.
__device__ __forceinline__ int fm_owned_face(int xcd, int j, int per, int split) {
    if (split > 1) {
        const int c = per / split, q = j / c;
        return (q * 8 + xcd) * c + (j - q * c);
    }
    return xcd * per + j;
}
struct Tile {
    int n, lane, wave, xi, row;
    int bx0, by0;                  // pixel origin of the workgroup's block
    bool valid, wave_on;
    float xp, yp;
    float bxlo, bxhi, bylo, byhi;  // block bounds (pixel centres)
    float wxlo, wxhi, wylo, wyhi;  // wave tile bounds
};

__device__ __forceinline__ void tile_setup(Tile &t, const RasterArgs &A) {
    const int total = A.N * A.tiles_x * A.tiles_y;
    int bx, by;
    if (A.block_order) {          // (k_block_order: only with the row-interleaved mapping, tiles_y % 8 == 0)
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, gsz = A.block_group * (A.tiles_y >> 3) * A.tiles_x;
        const int g = slot / gsz;
        const int e = A.block_order[((size_t)g * 8 + xcd) * gsz + slot % gsz];
        t.n = g * A.block_group + (e >> 16);
        by = (e >> 8) & 255;
        bx = e & 255;
    } else if (A.no_xcd_remap == 2 && A.tiles_y % 8 == 0) {
        // XCD x (= blockIdx % 8) takes block rows x, x + 8, x + 16, ... of EVERY mesh: the dense middle rows of each mesh are
        // spread over all eight XCDs (balance at small N) while horizontally adjacent blocks -- which share most of their
        // faces' records -- stay in one L2
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_mesh = (A.tiles_y >> 3) * A.tiles_x;
        t.n = slot / per_mesh;
        const int rem = slot - t.n * per_mesh;
        by = (rem / A.tiles_x) * 8 + xcd;
        bx = rem % A.tiles_x;
    } else {
        int wid = A.no_xcd_remap == 1 ? (int)blockIdx.x : xcd_remap(blockIdx.x, total);   // (row interleave impossible: contiguous runs)
        bx = wid % A.tiles_x; wid /= A.tiles_x;
        by = wid % A.tiles_y;
        t.n = wid / A.tiles_y;
    }
    t.lane = threadIdx.x & 63;
    t.wave = threadIdx.x >> 6;
    t.bx0 = bx * BLK_W; t.by0 = by * BLK_H;
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

// Block-level binning into the LDS list, ascending order; returns the count.  Candidates are entries [i0, i1) of `ids`
// (the super-block's pre-binned face list, ascending) or, with ids == NULL, the faces i0 .. i1-1 themselves.
__device__ __forceinline__ int build_list(int *s_list, int *s_wcnt, const float4 *__restrict__ bbox_n,
                                          const int *__restrict__ ids, int i0, int i1, const Tile &t) {
    int count = 0;
    for (int c = i0; c < i1; c += BLK_THREADS) {
        const int i = c + (int)threadIdx.x;
        bool pass = false;
        int f = i;
        if (i < i1) {
            if (ids) f = ids[i];
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

// Super-block of the workgroup's 16x16 block and that super-block's candidate list.
__device__ __forceinline__ int superblock_list(const RasterArgs &A, const Tile &t, const int *&ids) {
    if (!A.sb_list) { ids = nullptr; return A.F; }
    const int sb = (t.by0 / A.sb_size) * A.sb_nx + (t.bx0 / A.sb_size);
    const int cnt = A.sb_count[t.n * A.sb_slots + sb];
    if (cnt > A.sb_cap) { ids = nullptr; return A.F; }          // overflowed slot (degenerate scene): full scan
    ids = A.sb_list + ((size_t)t.n * A.sb_slots + sb) * A.sb_cap;
    return cnt;
}

// Coarse binning, once per mesh instead of once per 16x16 workgroup: block (sb, n) scans the mesh's F dilated bounding
// boxes (coalesced float4) against the pixel-centre bounds of super-block sb and writes the survivors, ascending
// (ballot + prefix compaction), to sb_list.  The predicate is the workgroup's own (build_list) on a rectangle that
// contains every workgroup block of the super-block, so no face a workgroup needs is ever missing (NaN boxes pass both).
__global__ __launch_bounds__(256) void k_superblock_bin(const float4 *__restrict__ bbox, int *__restrict__ sb_count,
                                                        int *__restrict__ sb_list, int F, int IS, int sb_size, int sb_nx, int sb_cap) {
    __shared__ int s_w[4];
    const int sb = blockIdx.x, n = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sbx = sb % sb_nx, sby = sb / sb_nx;
    int *out = sb_list + ((size_t)n * gridDim.x + sb) * sb_cap;     // gridDim.x = slots per mesh
    const bool pow2 = (IS & (IS - 1)) == 0;
    const float inv_is = 1.f / (float)IS;
    const int px0 = sbx * sb_size, px1 = min(px0 + sb_size - 1, IS - 1), pr0 = sby * sb_size, pr1 = min(pr0 + sb_size - 1, IS - 1);
    const float xlo = ndc_coord_fast(px0, IS, inv_is, pow2), xhi = ndc_coord_fast(px1, IS, inv_is, pow2);
    const float yhi = ndc_coord_fast(IS - 1 - pr0, IS, inv_is, pow2), ylo = ndc_coord_fast(IS - 1 - pr1, IS, inv_is, pow2);
    const float4 *bbox_n = bbox + (size_t)n * F;
    int count = 0;
    for (int c = 0; c < F; c += 256) {
        const int f = c + (int)threadIdx.x;
        bool pass = false;
        if (f < F) {
            const float4 bb = bbox_n[f];
            pass = !(xlo > bb.y || xhi < bb.x || ylo > bb.w || yhi < bb.z);
        }
        const unsigned long long m = __ballot(pass);
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int base = count, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int cw = s_w[w];
            if (w < wave) base += cw;
            tot += cw;
        }
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (pass && pos < sb_cap) out[pos] = f;
        count += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) sb_count[n * gridDim.x + sb] = count;
}

// Start order of the forward's workgroup blocks.  A 16x16 block's work is its visits, and those follow the number of faces
// over it: on a regular mesh blocks differ ~3x, on the half-trained meshes a training step renders (profiles/scenes/) a few
// blocks sit under hundreds of stacked faces and, started late, are the launch's tail (VALU busy 0.72 against 0.82 on the regular
// scene).  Per XCD -- which keeps its block rows x, x + 8, ... of every mesh: the records' L2 locality -- and per group of G meshes
// the blocks are sorted by DESCENDING face count of their 64x64 super-block (k_superblock_bin's sb_count; counting sort, 1024
// buckets): heavy blocks start first, the launch drains on empty ones.  The order changes no result (a block's pixels are its own).
#define BLOCK_ORDER_KEYS 1024
#define BLOCK_ORDER_MAX_ENTRIES 16384
#define BLOCK_ORDER_THREADS 1024
__global__ __launch_bounds__(BLOCK_ORDER_THREADS) void k_block_order(const int *__restrict__ sb_count, int *__restrict__ order, int N,
                                                                      int tiles_x, int tiles_y, int sb_size, int sb_nx, int sb_slots, int G) {
    __shared__ int s_hist[BLOCK_ORDER_KEYS];
    __shared__ int s_wsum[BLOCK_ORDER_THREADS / 64];
    __shared__ unsigned short s_key[BLOCK_ORDER_MAX_ENTRIES];
    const int xcd = blockIdx.x, g = blockIdx.y, per_mesh = (tiles_y >> 3) * tiles_x;
    const int m0 = g * G, gl = min(G, N - m0), E = gl * per_mesh;
    for (int k = threadIdx.x; k < BLOCK_ORDER_KEYS; k += BLOCK_ORDER_THREADS) s_hist[k] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += BLOCK_ORDER_THREADS) {
        const int ml = e / per_mesh, rem = e - ml * per_mesh;
        const int by = (rem / tiles_x) * 8 + xcd, bx = rem % tiles_x;
        const int sb = ((by * BLK_H) / sb_size) * sb_nx + (bx * BLK_W) / sb_size;
        const int key = min(sb_count[(m0 + ml) * sb_slots + sb], BLOCK_ORDER_KEYS - 1);
        s_key[e] = (unsigned short)key;
        atomicAdd(&s_hist[key], 1);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kb = BLOCK_ORDER_KEYS - 1 - (int)threadIdx.x;      // exclusive prefix over DESCENDING keys: thread t owns key 1023 - t
    const int mine = s_hist[kb];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
    __syncthreads();
    s_hist[kb] = base;
    __syncthreads();
    int *out = order + ((size_t)g * 8 + xcd) * ((size_t)G * per_mesh);
    for (int e = threadIdx.x; e < E; e += BLOCK_ORDER_THREADS) {
        const int ml = e / per_mesh, rem = e - ml * per_mesh;
        const int pos = atomicAdd(&s_hist[s_key[e]], 1);
        out[pos] = (ml << 16) | (((rem / tiles_x) * 8 + xcd) << 8) | (rem % tiles_x);
    }
}

}  // namespace
