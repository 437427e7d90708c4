/*
 * umr_hip.h -- C ABI of libumr_hip.so: the MI355X (gfx950) render-and-compare hot path of UMR.
 *
 * This is the drop-in boundary.  Every entry point replaces one interface of
 * the reference (cited per function, paths relative to the reference tree) and is what a maintainer
 * binds from Python (ctypes) in place of the reference's pybind/CUDA module or
 * torch op chain.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; fp32 data, int32 indices; every tensor dense, row-major
 *   - the CALLER owns and pre-allocates every buffer (as the reference binding does,
 *     functional/soft_rasterize.py:47-55,95-97); nothing is allocated or freed inside
 *   - work is enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream);
 *     no host synchronisation inside; re-entrant; callable from several host threads
 *   - return value: 0 = enqueued; UMR_ERR_ARG = rejected arguments / unsupported mode (nothing
 *     enqueued); UMR_ERR_LAUNCH = hipGetLastError() reported a launch failure (the reference only
 *     printf()s those, soft_rasterize_cuda_kernel.cu:700-702)
 *   - scratch memory is passed in by the caller: size from the matching *_workspace_bytes()
 */
#ifndef UMR_HIP_H
#define UMR_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UMR_OK 0
#define UMR_ERR_ARG (-1)
#define UMR_ERR_LAUNCH (-2)

/* library / build identification: returns e.g. "umr_hip 0.1 gfx950" */
const char *umr_version(void);
/* hash of the sources this binary was compiled from (umr_amd/build.py:source_hash); "unknown" for ad-hoc builds */
const char *umr_build_id(void);

/* Optional kernel timing for benchmarks.  While enabled, every raster main-kernel launch is bracketed by
 * library-owned HIP events recorded on the launch stream.  umr_profile_collect(which) waits for the
 * recorded events of `which` (0 = k_raster_forward, 1 = k_raster_backward), returns their summed
 * duration, the launch count and the summed ALGORITHMIC bytes (SURVEY.md section 8d formulas), and
 * forgets them. */
int umr_profile_enable(int on);
/* A/B switches for benchmarking kernel variants ("bwd_pixel_major": 1 selects the tile-binned
 * pixel-major backward with global atomics instead of the default face-major one; "superblock_bins",
 * "xcd_remap", "face_order" (0 one wave per face in index order, 1 work-item lists in cost order), "face_order_group",
 * "face_split" (estimated work -- 4x4 sub-tiles -- beyond which a face is split into several work items; 0 never,
 * negative = default 192), "face_split_budget", "fm_runs", "fm_rotate", "block_order" (0: the forward starts its
 * workgroup blocks row by row instead of heavy blocks first): see raster.hip), and two switches that trade time for
 * exactness (HISTORY.md 4.4).  Inside a triangle the reference keeps the edge line with the smallest COMPUTED
 * distance (:78-107); the kernels pick it by its true distance unless the face is thin.
 *   "exact_edges" (default 1): inside pixels whose SECOND nearest edge line is closer than sqrt(20 sigma) evaluate
 *       all three lines the reference's way -- the nearest-edge choice then never differs from the reference's
 *       (MI355X, BASELINE size: every one of 2.1 M colour values within 8e-7, every vertex-gradient value
 *       within 6e-7 of the largest).  0 = pick by true distance everywhere but in thin faces: 8..15 % less
 *       kernel time, for isolated pixels off by up to ~1e-4 in alpha and faces off by ~1 % in their gradient.
 *   "thin_face_h_1e6" = h * 1e6: faces with a height below h screen units do so for every inside pixel.
 *       Default 16000 (h = 0.016, +1..2 %); 1000000000 = every face (+15..33 %); negative = default. */
int umr_debug_set(const char *key, int value);
int umr_profile_collect(int which, double *total_ms, long *launches, double *total_bytes);

/* ---------------------------------------------------------------------------------------------
 * Soft rasterizer.  Replaces the pybind module soft_renderer.cuda.soft_rasterize
 *   forward_soft_rasterize   external/SoftRas/soft_renderer/cuda/soft_rasterize_cuda.cpp:62-97
 *   backward_soft_rasterize  external/SoftRas/soft_renderer/cuda/soft_rasterize_cuda.cpp:100-138
 * (kernels soft_rasterize_cuda_kernel.cu:223-656).  Same buffers, same scalars, same in-place
 * contract: soft_colors arrives filled with (background rgb, 1), every other output zero-filled.
 *
 *   faces        [N,F,9]   screen-space face vertices (x0,y0,z0,x1,...)          in
 *   textures     [N,F,TS,3]                                                     in
 *   faces_info   [N,F,27]  inv(9) sym(9) obt(3) unused(6)                        out (fwd) / unused (bwd)
 *   aggrs_info   [N,2,IS,IS] softmax: (sum, max); hard: (depth_min, face id|-1)  out (fwd) / in (bwd)
 *   grid         [IS,IS,2] pixel -> normalised coordinate map (p2f weights)      in  (softmax only)
 *   p2f_info     [N,F,2], p2f_sum [N,F,2]  raw accumulators (un-normalised)      out (softmax only)
 *   soft_colors  [N,4,IS,IS]                                                     in/out
 *   pooled_out   [N,4,IS/2,IS/2] or NULL: fused 2x2 average pool of soft_colors
 *                (anti-aliasing, rasterizer.py:52-53); requires even IS            out, optional
 *
 * Modes: every id the reference binding accepts (functional/soft_rasterize.py:21-24) -- func_id_dist {0 hard,
 * 1 barycentric, 2 euclidean}, func_id_alpha {0 hard, 1 sum, 2 prod}, func_id_rgb {0 hard, 1 softmax},
 * texture_sample_type {0 surface: TS a perfect square; 1 vertex: TS == 3, the reference reads w[j] for j < TS,
 * soft_rasterize_cuda_kernel.cu:215}.  The combination UMR instantiates (nnutils/smr.py:53-66: euclidean, prod,
 * surface) runs the specialised kernels; any other runs the general-mode kernels (csrc/raster_general.h), for which the
 * fused extras (pooled_out, UMR_RASTER_ALPHA_ONLY / FACE_ID_ONLY, UMR_BWD_GRAD_POOLED / ALPHA_ONLY) are not offered.
 * Anything else returns UMR_ERR_ARG.
 * dist_eps is the value the reference binding receives: log(1/eps_dist - 1).
 *
 * background: NULL = the reference contract above (soft_colors pre-filled by the caller).  A HOST pointer to
 * 3 floats instead lets soft_colors / aggrs_info arrive uninitialised: the kernel starts every pixel from that
 * colour and writes all six planes (saves the caller's 0.8 GB of fills per N=128 call,
 * functional/soft_rasterize.py:48-55).  faces_info may be NULL when the caller does not want it.
 *
 * flags: bit 0 (UMR_RASTER_NO_P2F) skips the p2f_info/p2f_sum accumulation (callers that discard
 * them, e.g. MultiMaskLoss, nnutils/loss_utils.py:265).
 * -------------------------------------------------------------------------------------------*/
#define UMR_RASTER_NO_P2F 1
/* flags bit 1: silhouette only.  For renders of which only the alpha channel is consumed (the mask render of
 * MultiMaskLoss / train_s1.py:199-200, the GAN-view render train_s1.py:235-236): the kernel evaluates the soft
 * coverage only -- alpha = 1 - prod(1 - D_f) does not depend on depth, colour or the depth-range test (:396
 * precedes :404) -- and `soft_colors` / `pooled_out` are then ALPHA PLANES [N,IS,IS] / [N,IS/2,IS/2];
 * textures, aggrs_info, grid, p2f_* may be NULL.  Bit-identical alpha to the full kernel. */
#define UMR_RASTER_ALPHA_ONLY 2
/* flags bit 2: visibility only (hard mode): writes aggrs_info = (nearest depth, its face id | -1) and nothing
 * else -- what TexCycle consumes from the hard renderer (nnutils/loss_utils.py:327-328, train_s1.py:223-224).
 * soft_colors, textures, grid, p2f_* may be NULL.  Bit-identical planes to the full hard kernel. */
#define UMR_RASTER_FACE_ID_ONLY 4
/* flags bits 8-23: texture group G (0 = 1).  `textures` is then [N/G,F,TS,3] and view n samples textures[n / G]:
 * K camera hypotheses of one image share one texture set, so the reference's textures.repeat(K) (70 MB at N=128,
 * nnutils/loss_utils.py:303-306) is folded into indexing.  umr_raster_backward takes the same field in bits 8-23 of
 * `grad_is_pooled`; grad_textures stays PER VIEW [N,F,TS,3] (the caller sums the G views, as autograd does for repeat). */
#define UMR_RASTER_TEX_GROUP(G) (((G) & 0xffff) << 8)
/* flags bit 3: packed saved state (soft-max colour with UMR's modes; needs pooled_out, a by-value `background` and an image size
 * that is a multiple of 8).  For a render whose caller consumes the POOLED image and whose backward is the one-pass
 * UMR_BWD_ALPHA_GEOMETRY | UMR_BWD_PACKED_STATE call: `aggrs_info` is then a buffer of umr_raster_state_bytes(N, image_size)
 * bytes that receives the render's saved state in the backward's own layout -- per mesh (IS/4)^2 records of 64 floats, one per
 * 4x4 pixel tile (row-major over tiles; pixel (x, y) of a tile at i = 4 y + x): [0,16) v_rcp_f32 of the soft-max sum (:608 uses
 * the sum only through its reciprocal), [16,32) soft-max maximum, [32,48) alpha, [48,52) per 2x2 quad the smallest maximum (NaN if
 * one of its pixels holds NaN), [52,56) per quad 1.0f iff all four alphas are exactly 1.0f, [56,64) unused -- and NOTHING is
 * written at full resolution besides it: soft_colors is not touched (may be NULL).  Same arithmetic, same pooled image, same
 * p2f as the planar call; the forward writes 20 B per pixel instead of 28 (+ 4 | 8 for visibility), the backward's waves read
 * whole 256-byte records instead of 8-byte pieces of 6 - 8 rows of three planes. */
#define UMR_RASTER_PACKED_STATE 8
/* flags bit 4 (umr_raster_forward_vis): `visibility` is [N,IS,IS], the face-id plane alone -- TexCycle reads nothing else
 * (nnutils/loss_utils.py:327-328, train_s1.py:223-224). */
#define UMR_RASTER_VIS_IDS_ONLY 16
/* umr_raster_backward `grad_is_pooled` is a bit field: */
#define UMR_BWD_GRAD_POOLED 1   /* gradient arrives at the 2x2-pooled resolution */
#define UMR_BWD_ALPHA_ONLY 2    /* soft_colors and grad_soft_colors are alpha planes (see above); exact when the rgb
                                   gradient is zero, which is what "only alpha is consumed" means; needs need_grad_faces */
#define UMR_BWD_ALPHA_GEOMETRY 4 /* soft-max render whose rgb channels were rendered from DETACHED geometry (train_s1.py:217,
                                   loss_utils.py:313) and whose alpha channel is the mask render of the same views (:199, :265):
                                   grad_faces receives the alpha term only, grad_textures the rgb term -- what
                                   UMR_BWD_ALPHA_ONLY on the alpha plane plus a texel-only call return, from one pass over the
                                   (pixel, face) pairs; needs need_grad_faces and need_grad_textures, func_id_rgb 1, and
                                   TS <= 1023 texels per face (the face-major kernels' LDS accumulators): anything else returns
                                   UMR_ERR_ARG with nothing enqueued -- the flag is never silently dropped */
#define UMR_BWD_PACKED_STATE 8   /* with UMR_BWD_ALPHA_GEOMETRY: `aggrs_info` is the packed saved state a UMR_RASTER_PACKED_STATE
                                   forward wrote (see there); soft_colors is not read (may be NULL).  Same gradients, bit for bit,
                                   as the planar call on the same render */
#define UMR_BWD_REUSE_WORKSPACE 16 /* `workspace` is the buffer the umr_raster_forward[_vis] call of the SAME faces, N, F, image_size
                                   and scalars filled, untouched since: its face records and bounding boxes are read, not rebuilt
                                   (one small launch less per backward; the ABI stays stateless without the flag) */

/* Bytes of caller-provided scratch one raster call needs (bounding boxes, face records, per-mesh coarse bins; the forward's block
 * start order; the backward's work-item lists, work estimates, split-face lists and the slabs of the split faces' partial sums).  umr_raster_workspace_bytes(N, F) is valid for EVERY image size (coarse bins sized for their 256-slot worst
 * case: 1 MB per mesh at F >= 1024); umr_raster_workspace_bytes_for(N, F, image_size) is the exact amount for that size (64 slots
 * per mesh up to 512^2: a quarter of it) -- a caller that knows the size it is about to render may allocate this instead. */
size_t umr_raster_workspace_bytes(int N, int F);
size_t umr_raster_workspace_bytes_for(int N, int F, int image_size);
/* Bytes of the packed saved state of UMR_RASTER_PACKED_STATE (16 per pixel); 0 when image_size is not a multiple of 8. */
size_t umr_raster_state_bytes(int N, int image_size);

int umr_raster_forward(const float *faces, const float *textures, float *faces_info, float *aggrs_info,
                       const float *grid, float *p2f_info, float *p2f_sum, float *soft_colors,
                       float *pooled_out, int N, int F, int TS, int image_size, float near_, float far_,
                       float eps, float sigma_val, int func_id_dist, float dist_eps, float gamma_val,
                       int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side,
                       int flags, const float *background, void *workspace, size_t workspace_bytes,
                       void *stream);

/* umr_raster_forward plus one more output: visibility [N,2,IS,IS] = (nearest depth, its face id | -1), the aggrs_info planes
 * a func_id_rgb = 0 ('hard') render of the SAME faces would write (soft_rasterize_cuda_kernel.cu:408-411, :463-464),
 * bit-identical to UMR_RASTER_FACE_ID_ONLY.  train_s1 renders the textured soft-max image (train_s1.py:217) and, from the
 * same mesh and camera, the hard render of which only the face-id plane is read (:223-224): here the second falls out of
 * the first's visits.  Soft-max colour with UMR's modes only; visibility == NULL is umr_raster_forward. */
int umr_raster_forward_vis(const float *faces, const float *textures, float *faces_info, float *aggrs_info,
                           const float *grid, float *p2f_info, float *p2f_sum, float *soft_colors,
                           float *pooled_out, int N, int F, int TS, int image_size, float near_, float far_,
                           float eps, float sigma_val, int func_id_dist, float dist_eps, float gamma_val,
                           int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side,
                           int flags, const float *background, void *workspace, size_t workspace_bytes,
                           void *stream, float *visibility);

/*   grad_faces    [N,F,9], grad_textures [N,F,TS,3]   zero-filled by the caller; contributions are ADDED
 *   grad_soft_colors [N,4,IS,IS], or -- when grad_is_pooled != 0 -- the gradient of the 2x2-pooled
 *                 image [N,4,IS/2,IS/2] (the avg_pool2d backward is fused: each pixel sees g/4)
 *   need_grad_faces / need_grad_textures: 0 skips that output (pointer may then be NULL); the
 *                 reference always computes both (functional/soft_rasterize.py:95-106). */
int umr_raster_backward(const float *faces, const float *textures, const float *soft_colors,
                        const float *faces_info, const float *aggrs_info, float *grad_faces,
                        float *grad_textures, const float *grad_soft_colors, int grad_is_pooled,
                        int need_grad_faces, int need_grad_textures, int N, int F, int TS, int image_size,
                        float near_, float far_, float eps, float sigma_val, int func_id_dist,
                        float dist_eps, float gamma_val, int func_id_rgb, int func_id_alpha,
                        int texture_sample_type, int double_side, void *workspace, size_t workspace_bytes,
                        void *stream);

/* ---------------------------------------------------------------------------------------------
 * Camera projection + face gather.  Replaces the torch op chain of one SoftRenderer.forward:
 *   nnutils/geom_utils.py:74-91,119-165  orthographic_proj_withz / quat_rotate / hamilton_product
 *   nnutils/smr.py:36                    y flip
 *   external/SoftRas/soft_renderer/functional/face_vertices.py:4-22   gather
 *   external/SoftRas/soft_renderer/functional/look_at.py:48-60 + orthogonal.py:13-16
 *                                        (eye (0,0,eye_z), at 0, up y => z -= eye_z; scale 1)
 *   verts [N/G,V,3], cams [N,7] = (s, tx, ty, qw, qx, qy, qz), faces_idx [N/G,F,3] int32
 *   face_pre  [N,F,9] or NULL: projected + flipped, BEFORE look_at (what Lighting sees, mesh.py:111-118)
 *   face_out  [N,F,9]: after look_at/orthogonal -- the rasterizer's `faces`
 *   mesh_group G >= 1: view n renders mesh n / G.  G = 1 is the reference layout; G = K folds the x K repeats of
 *   vertices and faces that MultiMaskLoss / MultiTextureLoss materialise (nnutils/loss_utils.py:260-262, 303-306:
 *   one mesh, K camera hypotheses) into indexing.
 * -------------------------------------------------------------------------------------------*/
int umr_project_faces_forward(const float *verts, const float *cams, const int *faces_idx, float *face_pre,
                              float *face_out, int N, int V, int F, float offset_z, float eye_z, int mesh_group,
                              void *stream);

/* backward: grad_face_out [N,F,9] (and optional grad_face_pre, may be NULL) ->
 *   grad_verts [N,V,3] PER VIEW (ADDED into; caller zero-fills and, for mesh_group > 1, sums the G views of a mesh)
 *   and grad_cams [N,7] (overwritten).
 *   workspace: umr_project_workspace_bytes(N, V) bytes. */
size_t umr_project_workspace_bytes(int N, int V);
int umr_project_faces_backward(const float *grad_face_out, const float *grad_face_pre, const float *verts,
                               const float *cams, const int *faces_idx, float *grad_verts, float *grad_cams,
                               int N, int V, int F, int mesh_group, void *workspace, size_t workspace_bytes,
                               void *stream);

/* The same projection with the per-face surface light of sr.Lighting folded in (external/SoftRas/soft_renderer/
 * lighting.py:50-57 = ambient_lighting.py:17 + directional_lighting.py:26-27 on mesh.py:111-118's face normals):
 *   light_out [N,F,3] = ambient * color + directional * color * relu(n . direction),
 *   n = normalize(cross(p2 - p1, p0 - p1), eps 1e-6) on the pre-look_at corners (light_out may be NULL = skip).
 * color / direction: HOST float[3] (NULL = (1,1,1) / (0,1,0), the renderer.py:58-60 defaults).  The caller multiplies
 * textures [N,F,TS,3] by light_out[:, :, None, :] (lighting.py:56).
 * backward: grad_light [N,F,3] (may be NULL) is chained through the normal into the corner gradients; it needs
 *   face_out as written by the forward. */
int umr_project_faces_lit_forward(const float *verts, const float *cams, const int *faces_idx, float *face_pre,
                                  float *face_out, float *light_out, int N, int V, int F, float offset_z, float eye_z,
                                  int mesh_group, float light_ambient, float light_directional, const float *light_color3,
                                  const float *light_direction3, void *stream);
int umr_project_faces_lit_backward(const float *grad_face_out, const float *grad_face_pre, const float *grad_light,
                                   const float *face_out, const float *verts, const float *cams, const int *faces_idx,
                                   float *grad_verts, float *grad_cams, int N, int V, int F, int mesh_group,
                                   float light_directional, const float *light_color3, const float *light_direction3,
                                   void *workspace, size_t workspace_bytes, void *stream);

/* Camera rotated about the y axis by angle_deg [B] degrees: geom_utils.rotate_cam(cam, angle, axis=[0,1,0])
 * (nnutils/geom_utils.py:167-193, a per-sample numpy / cv2.Rodrigues / quaternion_from_matrix round trip through the
 * host in the reference; call sites experiments/train_s1.py:233, train_s2.py:257).  cam, out [B,7]; forward only
 * (both call sites detach the camera).  out = [s, tx, ty, q'] with q' = q_y(angle) (x) q, unit norm, q'_w >= 0. */
int umr_rotate_cam_y(const float *cam, const float *angle_deg, float *out, int B, void *stream);
/* ... and about any axis (HOST pointer to 3 floats; the reference's cv2.Rodrigues(rad_angle * axis) rotates by |rad_angle * axis|
 * about axis / |axis|, so a non-unit axis scales the angle). */
int umr_rotate_cam_axis(const float *cam, const float *angle_deg, const float *axis3, float *out, int B, void *stream);

/* vertices only, no flip, no look_at:
 *   out_dim 2: SoftRenderer.project_points / orthographic_proj (nnutils/smr.py:76-78, geom_utils.py:60-72)
 *              out [N,V,2] = (s R(q) X)[:2] + t
 *   out_dim 3: orthographic_proj_withz (geom_utils.py:74-91): out [N,V,3], z = s (R X)_z + offset_z
 * backward: grad_verts [N,V,3] ADDED into (may be NULL), grad_cams [N,7] overwritten. */
int umr_project_points_forward(const float *verts, const float *cams, float *out, int N, int V, int out_dim,
                               float offset_z, void *stream);
int umr_project_points_backward(const float *grad_out, const float *verts, const float *cams, float *grad_verts,
                                float *grad_cams, int N, int V, int out_dim, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Silhouette IoU.  Replaces nnutils/loss_utils.py:41-48 (neg_iou_loss, avg=False form):
 *   loss[n] = 1 - sum(p*t) / (sum(p + t - p*t) + 1e-6)      predict/target [N,P], loss [N]
 * predict may be a strided channel view: element (n,i) at predict[n*predict_stride + i].
 * backward: grad_predict (same striding, ADDED into) = grad_loss[n] * d loss[n] / d p.
 * sums [N, umr_neg_iou_sums_stride(P)]: forward scratch (per-block partial sums, added in a fixed order: no float
 * atomics, bit-reproducible) whose first two floats per row, (intersect, union + 1e-6), backward reads.
 * sums_bytes: size of the caller's scratch, checked against N * umr_neg_iou_sums_stride(P) * 4 (the row grew from 2 floats
 * in version 0.1: a caller built against the old contract is refused instead of being written past).
 * -------------------------------------------------------------------------------------------*/
long umr_neg_iou_sums_stride(long P);
int umr_neg_iou_forward(const float *predict, long predict_stride, const float *target, float *loss,
                        float *sums, size_t sums_bytes, int N, long P, void *stream);
int umr_neg_iou_backward(const float *predict, long predict_stride, const float *target, const float *sums,
                         const float *grad_loss, float *grad_predict, long grad_stride, int N, long P,
                         void *stream);

/* ---------------------------------------------------------------------------------------------
 * Chamfer distance.  Replaces nnutils/chamfer_python.py:43-64 (distChamfer) without the [B,n,m]
 * matrix:  P_ij = |a_i|^2 + |b_j|^2 - 2 a_i.b_j  (same expansion as the reference, so values agree
 * to rounding); dist1[i] = min_j, dist2[j] = min_i, idx int32, first index wins ties.
 *   a [B,n,D], b [B,m,D], D in {2,3}
 * backward: grad_a, grad_b overwritten with the gradient of (g1 . dist1 + g2 . dist2).
 * -------------------------------------------------------------------------------------------*/
int umr_chamfer_forward(const float *a, const float *b, float *dist1, float *dist2, int *idx1, int *idx2,
                        int B, int n, int m, int D, void *stream);
int umr_chamfer_backward(const float *a, const float *b, const int *idx1, const int *idx2, const float *g1,
                         const float *g2, float *grad_a, float *grad_b, int B, int n, int m, int D,
                         void *stream);

/* ---------------------------------------------------------------------------------------------
 * Texture sampling.  Replaces F.grid_sample as called by nnutils/geom_utils.py:41-59
 * (sample_textures) and nnutils/loss_utils.py:60-64 (texture_dt_loss): bilinear, zero padding,
 * torch-1.1.0 coordinate convention (= align_corners=True).
 *   image [B,C,H,W], grid [B,P,2] (x,y in [-1,1]) -> out [B,P,C]  (channel-LAST: the layout
 *   sample_textures permutes to, geom_utils.py:59)
 * backward: grad_out [B,P,C] -> grad_grid [B,P,2] (overwritten, may be NULL),
 *           grad_image [B,C,H,W] (ADDED into, may be NULL)
 * -------------------------------------------------------------------------------------------*/
int umr_grid_sample_forward(const float *image, const float *grid, float *out, int B, int C, int H, int W,
                            long P, void *stream);
int umr_grid_sample_backward(const float *image, const float *grid, const float *grad_out, float *grad_grid,
                             float *grad_image, int B, int C, int H, int W, long P, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Mesh regularisers.  Replace external/SoftRas/soft_renderer/losses.py
 *   LaplacianLoss.forward :29-37 -- sum_v |x_v - mean_{u in nbr(v)} x_u|^2 per mesh, from a CSR
 *     neighbour list instead of the dense row-normalised [V,V] matrix (identical rows: the
 *     reference's matrix has 1 on the diagonal and -1/deg on neighbours)
 *   FlattenLoss.forward :72-114   -- sum_edges (cos_dihedral + 1)^2, eps = 1e-6 guards kept
 *   x [B,V,3]; nbr_off [V+1], nbr_idx [nnz] int32;  quads [E,4] int32 = (v0,v1,v2,v3)
 *   loss [B]; backward ADDS grad_loss[b] * dloss/dx into grad_x [B,V,3].
 * -------------------------------------------------------------------------------------------*/
int umr_laplacian_forward(const float *x, const int *nbr_off, const int *nbr_idx, float *lap, float *loss,
                          int B, int V, void *stream);
int umr_laplacian_backward(const float *lap, const int *nbr_off, const int *nbr_idx, const float *grad_loss,
                           float *grad_x, int B, int V, void *stream);
int umr_flatten_forward(const float *x, const int *quads, float *loss, int B, int V, int E, void *stream);
int umr_flatten_backward(const float *x, const int *quads, const float *grad_loss, float *grad_x, int B, int V,
                         int E, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Texture-cycle visibility mask.  Replaces the per-sample torch.unique + CPU mask + H2D loop of
 * nnutils/loss_utils.py:173-179 (TexCycle.forward):
 *   face_ids [B,P] float (hard renderer's aggrs_info[:,1], -1 = background) -> mask [B,F] (0/1)
 *   id -1 marks the LAST face (python negative indexing -- reference behaviour, kept).
 *   mask must arrive zero-filled.
 * -------------------------------------------------------------------------------------------*/
int umr_visible_face_mask(const float *face_ids, float *mask, int B, long P, int F, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Perceptual distance head.  Replaces the torch chain of
 *   external/PerceptualSimilarity/models/networks_basic.py:42-64 (PNet.forward after the feature taps: sum over taps of
 *   1 - cos_sim) and external/PerceptualSimilarity/util/util.py:71-83 (normalize_tensor with eps 1e-10, channel dot
 *   product, mean over x then y), reached from nnutils/loss_utils.py:128-150 (PerceptualTextureLoss) through
 *   nnutils/perceptual_loss.py:38-57.  The dense convolutions producing the taps stay on MIOpen.
 *     ntaps <= UMR_COS_MAX_TAPS feature pairs; tap t: f0[t], f1[t] are [N, C[t], P[t]] (P = X*Y, NCHW contiguous)
 *     val [N] = sum_t ( 1 - mean_p  <f0,f1>_p / ((|f0|_p + eps)(|f1|_p + eps)) )          (overwritten)
 *   backward: g0[t] / g1[t] (either array, or single entries, may be NULL) are overwritten with
 *     grad_val[n] * d val[n] / d f; a zero feature vector gets d|f|/df := 0 (torch: NaN).
 *   f0/f1/g0/g1/C/P are HOST arrays of length ntaps (device pointers inside).
 *   workspace: umr_cos_sim_workspace_bytes(); forward writes per-pixel statistics there that backward reads.
 * -------------------------------------------------------------------------------------------*/
#define UMR_COS_MAX_TAPS 8
#define UMR_COS_CHUNKS 1024   /* 64-pixel chunks per feature map: maps of up to 65536 pixels (256 x 256) */
size_t umr_cos_sim_workspace_bytes(int ntaps, int N, const int *P);

/* Input side of the perceptual texture term in one pass per image stack (img [B,C<=3,H,W], mask [B,H,W], HW = H*W):
 *   out = ((2 * (img * mask) - 1) - shift_c) / scale_c
 * = nnutils/loss_utils.py:141-146 (image * mask), nnutils/perceptual_loss.py:52-54 (2 x - 1) and
 * external/PerceptualSimilarity/models/networks_basic.py:45-46 ((x - shift) / scale), every step rounded to fp32 in the
 * reference's order.  shift3 / scale3: HOST pointers to 3 floats.  backward: grad_img [B,C,H,W] and / or grad_mask [B,H,W]
 * (either may be NULL), fully written. */
int umr_perceptual_prologue_forward(const float *img, const float *mask, float *out, int B, int C, long HW,
                                    const float *shift3, const float *scale3, void *stream);
int umr_perceptual_prologue_backward(const float *grad_out, const float *img, const float *mask, float *grad_img,
                                     float *grad_mask, int B, int C, long HW, const float *scale3, void *stream);
int umr_cos_sim_forward(int ntaps, const float *const *f0, const float *const *f1, const int *C, const int *P, int N,
                        float eps, float *val, void *workspace, size_t workspace_bytes, void *stream);
int umr_cos_sim_backward(int ntaps, const float *const *f0, const float *const *f1, float *const *g0, float *const *g1,
                         const int *C, const int *P, int N, float eps, const float *grad_val, const void *workspace,
                         size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Part-matching reductions.  Replaces everything after the part renders in nnutils/loss_utils.py:399-440
 * (part_matching_loss.forward, loss_type 'mse') including nnutils/scops_utils.py:12-54 (batch_get_centers):
 *   proj = [background (const), p1, p2, p3, p4] with p1..p3 = channels 0..2 of render_a, p4 = channel 0 of render_b
 *          (both [B,4,H,W], the renderer's output; the reference renders each part three times over, :357-397)
 *   l_lm[b]  = sum_{c=1..4} | centroid(softmax(proj)_c) - centroid(softmax(part_segs)_c) |^2      (centroids of the
 *              maps + center_eps, normalised to sum 1, coordinates col/H*2-1, row/W*2-1 as scops_utils.py:12-16)
 *   l_eqv[b] = sum_{c=0..4} sum_pixels w_c (proj_c / max(max_p proj_c, 1e-5) - part_c / max(max_p part_c, 1e-5))^2
 * The caller forms (mean(l_eqv)/(5 H W) + mean(l_lm)/8) / 4 (avg=True) or the cam-prob weighted sums (avg=False).
 * backward: grad_render_a / grad_render_b [B,4,H,W] must arrive zero-filled; planes 0..2 of a and plane 0 of b are
 *   overwritten with d(sum_b g_eqv[b] l_eqv[b] + g_lm[b] l_lm[b])/d render, the max term routed to the first arg-max.
 * workspace: umr_part_match_workspace_bytes(B, H, W); forward leaves the statistics there that backward reads.
 * -------------------------------------------------------------------------------------------*/
size_t umr_part_match_workspace_bytes(int B, int H, int W);
int umr_part_match_forward(const float *render_a, const float *render_b, const float *part_segs, int B, int H, int W,
                           const float *weights5, float background, float center_eps, float *l_eqv, float *l_lm,
                           void *workspace, size_t workspace_bytes, void *stream);
int umr_part_match_backward(const float *render_a, const float *render_b, const float *part_segs, int B, int H, int W,
                            const float *weights5, float background, float center_eps, const float *grad_l_eqv,
                            const float *grad_l_lm, float *grad_render_a, float *grad_render_b, const void *workspace,
                            size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Barrier distance transform.  Replaces utils/image.py:130-141 (compute_dt_barrier), two scipy EDTs per image per
 * step on the host (experiments/train_s2.py:196):
 *   out = sigmoid(k * (EDT(1 - mask) - EDT(mask)) / max(H, W)),   mask [B,H,W] (non-zero = foreground)
 * sq_out / sq_in (optional, int32 [B,H,W]) receive the exact squared distances of the two transforms.
 * workspace: umr_dt_barrier_workspace_bytes(B, H, W).
 * -------------------------------------------------------------------------------------------*/
size_t umr_dt_barrier_workspace_bytes(int B, int H, int W);
int umr_dt_barrier(const float *mask, float *out, int *sq_out, int *sq_in, int B, int H, int W, float k,
                   void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * 2x bilinear up-sampling of the texture-flow decoder (nn.Upsample(scale_factor=2, mode='bilinear'),
 * nnutils/net_blocks.py upconv2d; align_corners=False).  in [planes,H,W] -> out [planes,2H,2W]; backward is the
 * exact transpose in gather form (deterministic).  grad_in is overwritten.
 * -------------------------------------------------------------------------------------------*/
int umr_upsample2x_bilinear_forward(const float *in, float *out, long planes, int H, int W, void *stream);
int umr_upsample2x_bilinear_backward(const float *grad_out, float *grad_in, long planes, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Texture-atlas bake for textured OBJ dumps: create_texture_image
 * (external/SoftRas/soft_renderer/cuda/create_texture_image_cuda.cpp:20-31 -> create_texture_image_cuda_kernel.cu:10-70)
 * plus the host prologue/epilogue of functional/save_obj.py:9-35, 50-53 (atlas triangle layout, vt normalisation,
 * clip/x255/uint8/vertical flip).  textures [F, res_in^2, 3]; the atlas is [height, width, 3] with
 * (height, width) from umr_texture_atlas_shape (tile grid of save_obj.py:11-12, cells of res_out^2 pixels).
 * Any of the three outputs may be NULL:  image = float atlas, NOT flipped (what the reference kernel returns);
 * image_u8 = the bytes save_obj hands to imsave (clipped, x255, truncated, rows reversed);
 * uv [F,3,2] = vt coordinates (corners / (width-1, height-1)).  eps = 1e-5 in the reference (save_obj.py:26).
 * -------------------------------------------------------------------------------------------*/
int umr_texture_atlas_shape(int F, int res_out, int *height, int *width);
int umr_texture_atlas(const float *textures, float *image, unsigned char *image_u8, float *uv, int F, int res_in,
                      int res_out, float eps, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Small regularisers and the masked L1 image term (csrc/regs.hip).  Replace the torch op chains of
 *   deform_l2reg        nnutils/loss_utils.py:118-123   out[0] = mean over rows of ||x_row||_2,  x [rows, width]
 *   sym_reg             nnutils/loss_utils.py:125-126   out[0] = mean |x[row, column]|           (verts [B*V,3], column 1)
 *   texture_loss_masks  nnutils/loss_utils.py:103-116   per_sample[b] = mean_{c,p} |img_pred mask_pred - img_gt mask_gt|
 *                       (img [B,C,HW], masks [B,HW]); the avg=True form is the mean of per_sample
 * Two-stage sums in a fixed order (no float atomics): `scratch` holds the per-block partials,
 * umr_reg_scratch_floats(elements, batch) floats (elements = rows, or C*HW with batch = B).
 * backward: grad_x / grad_img_pred / grad_mask_pred are OVERWRITTEN (either masked-L1 gradient may be NULL); the norm's
 * gradient at a zero row and sign(0) are 0, as in torch.
 * -------------------------------------------------------------------------------------------*/
long umr_reg_scratch_floats(long elements, int batch);
int umr_row_norm_mean_forward(const float *x, float *out, float *scratch, size_t scratch_bytes, long rows, int width, void *stream);
int umr_row_norm_mean_backward(const float *x, const float *grad_out, float *grad_x, long rows, int width, void *stream);
int umr_abs_mean_forward(const float *x, float *out, float *scratch, size_t scratch_bytes, long rows, int width, int column,
                         void *stream);
int umr_abs_mean_backward(const float *x, const float *grad_out, float *grad_x, long rows, int width, int column, void *stream);
int umr_masked_l1_forward(const float *img_pred, const float *img_gt, const float *mask_gt, const float *mask_pred,
                          float *per_sample, float *scratch, size_t scratch_bytes, int B, int C, long HW, void *stream);
int umr_masked_l1_backward(const float *img_pred, const float *img_gt, const float *mask_gt, const float *mask_pred,
                           const float *grad_per_sample, float *grad_img_pred, float *grad_mask_pred, int B, int C, long HW,
                           void *stream);

/* ---------------------------------------------------------------------------------------------
 * Keypoint-transfer evaluation (csrc/eval.hip).  Replaces the per-pair host loops of experiments/test_kp.py:
 *   flow mode  :125-158  keypoint -> face (arg-max over faces of the keypoint heat map, utils/kp_utils.py:42-69, sampled at
 *                        the source texture flow) -> image point (mean of the coordinate grid sampled at the target flow)
 *   cam mode   :160-193  keypoint -> nearest projected template vertex (source camera) -> nearest foreground pixel of the
 *                        target mask to that vertex (target camera)
 *   PCK        :253-258, :317-323  integer counters [3,K] = (visible, err < thr_a, err < thr_b) per keypoint, ADDED into
 * One call maps `pairs` independent (source, target) entries (a test pair is two entries, one per direction).
 *   kp_src [pairs,K,kp_stride>=2] in [-1,1]; flow_* [pairs,F,TT,2]; patch [(6 sigma+1)^2] = the Gaussian of draw_labelmap
 *   (caller computes it in float64 as the reference does); face_idx / vert_idx [pairs,K] int32; k2k [pairs,K,2];
 *   kp_gt [pairs,K,gt_stride] + vis [pairs,K] + counters, all three or none; verts_src / verts_tgt [pairs,V,2] = template
 *   vertices projected with the two cameras (umr_project_points_forward); mask_tgt [pairs,S,S] (non-zero = foreground);
 *   pixel_of_vertex [pairs,V] int32 scratch/out (raster index of each vertex's foreground pixel, -1 for an empty mask).
 * Ties go to the first index, as torch.max / torch.min do.
 * -------------------------------------------------------------------------------------------*/
size_t umr_kp_flow_workspace_bytes(int pairs, int K, int F);
int umr_kp_flow_transfer(const float *kp_src, int kp_stride, const float *flow_src, const float *flow_tgt, const float *patch,
                         int *face_idx, float *k2k, const float *kp_gt, int gt_stride, const float *vis, int *counters, int pairs,
                         int K, int F, int TT, int image_size, int sigma, float padding_frac, float thr_a, float thr_b,
                         void *workspace, size_t workspace_bytes, void *stream);
int umr_kp_cam_transfer(const float *kp_src, int kp_stride, const float *verts_src, const float *verts_tgt, const float *mask_tgt,
                        int *vert_idx, int *pixel_of_vertex, float *k2k, const float *kp_gt, int gt_stride, const float *vis,
                        int *counters, int pairs, int K, int V, int image_size, float padding_frac, float thr_a, float thr_b,
                        void *stream);

#ifdef __cplusplus
}
#endif
#endif /* UMR_HIP_H */
