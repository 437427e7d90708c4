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
#include "raster_core.h"
#include "raster_forward.h"
#include "raster_backward.h"
#include "raster_general.h"
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

namespace {

size_t ws_bbox_bytes(int N, int F) { return (((size_t)N * F * sizeof(float4)) + 255) & ~(size_t)255; }
size_t ws_rec_bytes(int N, int F) { return (size_t)N * F * REC * sizeof(float); }
// the coarse bins are laid out for the super-block slots a mesh actually has at this image size (sb_slots_for: 64 up to 512^2, 256
// at 1024^2 and beyond); umr_raster_workspace_bytes(N, F) -- the size-independent query -- asks for the 256-slot worst case
size_t ws_sbcount_bytes(int N, int slots) { return (((size_t)N * slots * sizeof(int)) + 255) & ~(size_t)255; }
int sb_cap_for(int F) { return F < SB_CAP ? F : SB_CAP; }
size_t ws_sblist_bytes(int N, int F, int slots) { return (size_t)N * slots * sb_cap_for(F) * sizeof(int); }
// Work-item lists of the face-major backward (k_face_order): per (mesh group, XCD) a list of gsz = G x F / 8 faces' items plus
// up to order_extra(gsz) more from split faces, the slab pool of the split faces' partial sums (2 x extra units of SPLIT_UNIT
// floats per list: extras + split faces <= 2 x extras) and the list of the split faces.  Sized for the worst group size G (the
// "face_order_group" debug switch can only shrink G).
int order_extra(int gsz) { return gsz / 4 + 16; }
int order_group_for(int N, int F) { return std::max(1, std::min(N <= 16 ? 16 : 8, ORDER_MAX_ENTRIES / std::max(1, F / 8))); }
struct OrderLayout { int G, groups, stride, extra, slabs_per_list; size_t order_bytes, ctr_bytes, slab_bytes, est_bytes; };
OrderLayout order_layout(int N, int F, int G) {
    OrderLayout L;
    L.G = G; L.groups = (N + G - 1) / G;
    const int gsz = G * (F / 8);
    L.extra = order_extra(gsz); L.stride = gsz + L.extra; L.slabs_per_list = 2 * L.extra;
    const size_t lists = (size_t)L.groups * 8;
    L.order_bytes = (lists * L.stride * sizeof(uint2) + 255) & ~(size_t)255;
    L.ctr_bytes = (lists * (L.extra + 1) * sizeof(uint2) + 255) & ~(size_t)255;          // the split-face lists (k_split_reduce)
    L.slab_bytes = lists * L.slabs_per_list * SPLIT_UNIT * sizeof(float);
    L.est_bytes = ((size_t)N * F * sizeof(unsigned) + 255) & ~(size_t)255;                 // k_face_estimate -> k_face_order
    return L;
}
size_t ws_order_bytes(int N, int F) {
    size_t worst = 0;
    for (int G = 1; G <= order_group_for(N, F); ++G) {
        const OrderLayout L = order_layout(N, F, G);
        worst = std::max(worst, L.order_bytes + L.ctr_bytes + L.slab_bytes + L.est_bytes);
    }
    return worst;
}
int g_block_order = 1;           // umr_debug_set("block_order", 0 | 1): the forward starts its 16x16 workgroup blocks in descending order of the face
                                 // count of their super-block (k_block_order) instead of row by row
int g_fm_runs = 0;               // umr_debug_set("fm_runs", r): runs of faces per XCD and mesh (fm_owned_face) for every face-major launch; 0 = automatic
int g_fm_rotate = -1;            // umr_debug_set("fm_rotate", 0 | 1): mesh m's runs go to XCD (x + m) % 8 instead of x (item lists only); -1 = automatic
int g_split_budget_div = 16;     // umr_debug_set("face_split_budget", d >= 4): a list may hold gsz / d + 16 extra items (<= the gsz / 4 + 16 the workspace
                                 // is sized for); the lists are padded to that length, i.e. the launch carries that many workgroups that exit at once
int g_face_split = SPLIT_T0;     // umr_debug_set("face_split", T): estimated work (4x4 sub-tiles) beyond which k_face_order splits a face
                                 // into several work items; 0 = never (one wave per face, round 5's form)
int g_xcd_remap = 2;             // umr_debug_set("xcd_remap", v): work mapping of the pixel-major kernels.  2 (default): XCD x takes
                                 // block rows x, x + 8, ... of every mesh; 1: each XCD a contiguous run of (mesh, block) items
                                 // (forward 3-7 % slower than 2 at N = 16: two whole meshes per XCD balance worse); 0: plain
                                 // blockIdx order (12-25 % slower: one mesh's records then live in all eight L2s)
int g_face_order_group = 0;      // umr_debug_set("face_order_group", G): meshes per start-order group (0 = automatic)
int g_face_order = 1;            // umr_debug_set("face_order", v): 0 = every face-major backward starts one wave per face in index order,
                                 // 1 = work-item lists in cost order (k_face_order: heavy faces split, heavy items first), 3 = work-item
                                 // lists in index order (A/B)
float g_thin_face_h = THIN_FACE_H;   // umr_debug_set("thin_face_h_1e6", h * 1e6): faces with a height below h screen units evaluate
                                     // inside pixels the reference's way (k_face_setup, bit 4 of the record's flags)
bool g_exact_edges = true;           // umr_debug_set("exact_edges", 0 | 1): eval_pair's amb_thr = 20 sigma (see there).  On by default:
                                     // the nearest-edge choice inside a face is then the reference's in every pixel; 0 trades that
                                     // for 8-15 % of the raster kernels' time (HISTORY.md 4.4)
bool g_superblocks = true;       // umr_debug_set("superblock_bins", 0): every workgroup scans all F faces (A/B)

// super-block edge: 64 pixels, or a sixteenth of the image rounded up to whole 16-pixel workgroup blocks when that is larger
// (<= 16 x 16 super-blocks).  64^2 keeps a workgroup's candidate list at ~100 faces for the BASELINE meshes at IS = 512 and
// at IS = 1024 alike (a fixed 8 x 8 grid made them 128^2 and the lists 3x as long there: 80.9 us per mesh forward at
// N = 32, F = 5120, IS = 1024 in round 2).
void superblock_geometry(int IS, int *size, int *nx) {
    const int sixteenth = (((IS + 15) / 16) + 15) & ~15;
    *size = sixteenth > 64 ? sixteenth : 64;
    *nx = (IS + *size - 1) / *size;
}
int sb_slots_for(int IS) {   // super-block slots per mesh: sb_nx^2 (<= SB_SLOTS)
    if (IS <= 0) return SB_SLOTS;
    int size, nx;
    superblock_geometry(IS, &size, &nx);
    return nx * nx;
}
size_t ws_bins_offset(int N, int F) { return ws_bbox_bytes(N, F) + ws_rec_bytes(N, F); }
size_t ws_order_offset(int N, int F, int IS) { return ws_bins_offset(N, F) + ws_sbcount_bytes(N, sb_slots_for(IS)) + ws_sblist_bytes(N, F, sb_slots_for(IS)); }

// workspace pointers + the per-mesh coarse binning pass (after k_face_setup on the same stream)
void setup_bins(RasterArgs &A, void *workspace, int N, int F, int IS, hipStream_t st) {
    A.sb_count = nullptr; A.sb_list = nullptr;
    if (!g_superblocks) return;
    char *p = (char *)workspace + ws_bins_offset(N, F);
    superblock_geometry(IS, &A.sb_size, &A.sb_nx);
    A.sb_slots = A.sb_nx * A.sb_nx;
    int *cnt = (int *)p, *lst = (int *)(p + ws_sbcount_bytes(N, A.sb_slots));
    A.sb_cap = sb_cap_for(F);
    UMR_LAUNCH(k_superblock_bin, dim3(A.sb_slots, N), 256, 0, st, A.bbox, cnt, lst, F, IS, A.sb_size, A.sb_nx, A.sb_cap);
    A.sb_count = cnt; A.sb_list = lst;
}

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

UMR_TRAP_ACCESSOR(umr_trap_read_raster)
UMR_TRAP_INFO_ACCESSOR(umr_debug_trap_where)
#if UMR_TRAP
extern "C" unsigned long long umr_trap_read_geometry(int), umr_trap_read_losses(int), umr_trap_read_perceptual(int), umr_trap_read_edt(int);
// earliest non-finite report of the whole library: site id (0 = none), *when = its device wall-clock stamp
extern "C" int umr_debug_trap(int reset, unsigned long long *when) {
    (void)hipDeviceSynchronize();
    unsigned long long best = ~0ull;
    const unsigned long long v[5] = {umr_trap_read_raster(reset), umr_trap_read_geometry(reset), umr_trap_read_losses(reset),
                                     umr_trap_read_perceptual(reset), umr_trap_read_edt(reset)};
    for (int i = 0; i < 5; ++i) best = v[i] < best ? v[i] : best;
    if (when) *when = best >> 8;
    return best == ~0ull ? 0 : (int)(best & 0xff);
}
#endif

extern "C" {

const char *umr_version(void) { return "umr_hip 0.7 gfx950"; }

#ifndef UMR_SRC_HASH
#define UMR_SRC_HASH "unknown"
#endif
const char *umr_build_id(void) { return UMR_SRC_HASH; }

int umr_debug_set(const char *key, int value) {
    if (!key) return UMR_ERR_ARG;
    if (std::string(key) == "bwd_pixel_major") { g_bwd_pixel_major = value != 0; return UMR_OK; }
    if (std::string(key) == "superblock_bins") { g_superblocks = value != 0; return UMR_OK; }
    if (std::string(key) == "xcd_remap") { g_xcd_remap = value; return UMR_OK; }   // 0 off, 1 contiguous runs, 2 row-interleaved
    if (std::string(key) == "face_order") { g_face_order = value; return UMR_OK; }
    if (std::string(key) == "exact_edges") { g_exact_edges = value != 0; return UMR_OK; }
    if (std::string(key) == "thin_face_h_1e6") { g_thin_face_h = value < 0 ? THIN_FACE_H : 1e-6f * (float)value; return UMR_OK; }
    if (std::string(key) == "block_order") { g_block_order = value != 0; return UMR_OK; }
    if (std::string(key) == "fm_runs") { g_fm_runs = std::max(0, value); return UMR_OK; }
    if (std::string(key) == "fm_rotate") { g_fm_rotate = value < 0 ? -1 : (value != 0); return UMR_OK; }
    if (std::string(key) == "face_split_budget") { g_split_budget_div = std::max(4, value); return UMR_OK; }
    if (std::string(key) == "face_split") { g_face_split = value < 0 ? SPLIT_T0 : std::min(1 << 14, value); return UMR_OK; }   // (< 0: the default)
    if (std::string(key) == "face_order_group") { g_face_order_group = std::max(0, std::min(16, value)); return UMR_OK; }
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

size_t umr_raster_workspace_bytes_for(int N, int F, int image_size) {
    if (N <= 0 || F <= 0) return 0;
    return ws_order_offset(N, F, image_size) + ws_order_bytes(N, F);
}

size_t umr_raster_workspace_bytes(int N, int F) { return umr_raster_workspace_bytes_for(N, F, 0); }

size_t umr_raster_state_bytes(int N, int image_size) {   // the packed saved state: 16 B per pixel (RasterArgs::state)
    if (N <= 0 || image_size <= 0 || (image_size & 7) || image_size > 8192) return 0;
    return (size_t)N * image_size * image_size * (STATE_REC / 16) * sizeof(float);
}

int umr_raster_forward(const float *faces, const float *textures, float *faces_info, float *aggrs_info,
                       const float *grid, float *p2f_info, float *p2f_sum, float *soft_colors,
                       float *pooled_out, int N, int F, int TS, int image_size, float near_, float far_,
                       float eps, float sigma_val, int func_id_dist, float dist_eps, float gamma_val,
                       int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side,
                       int flags, const float *background, void *workspace, size_t workspace_bytes,
                       void *stream) {
    return umr_raster_forward_vis(faces, textures, faces_info, aggrs_info, grid, p2f_info, p2f_sum, soft_colors, pooled_out,
                                  N, F, TS, image_size, near_, far_, eps, sigma_val, func_id_dist, dist_eps, gamma_val,
                                  func_id_rgb, func_id_alpha, texture_sample_type, double_side, flags, background, workspace,
                                  workspace_bytes, stream, nullptr);
}

int umr_raster_forward_vis(const float *faces, const float *textures, float *faces_info, float *aggrs_info,
                           const float *grid, float *p2f_info, float *p2f_sum, float *soft_colors,
                           float *pooled_out, int N, int F, int TS, int image_size, float near_, float far_,
                           float eps, float sigma_val, int func_id_dist, float dist_eps, float gamma_val,
                           int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side,
                           int flags, const float *background, void *workspace, size_t workspace_bytes,
                           void *stream, float *visibility) {
    int R = 0;
    const bool alpha_only = (flags & UMR_RASTER_ALPHA_ONLY) != 0;
    const bool ids_only = (flags & UMR_RASTER_FACE_ID_ONLY) != 0;
    // packed saved state: `aggrs_info` is the tiled buffer of umr_raster_state_bytes(N, image_size) bytes, soft_colors is not
    // written (may be NULL); the image leaves through pooled_out alone
    const bool packed = (flags & UMR_RASTER_PACKED_STATE) != 0;
    const bool vis_ids = (flags & UMR_RASTER_VIS_IDS_ONLY) != 0;
    const int tex_group = ((flags >> 8) & 0xffff) ? ((flags >> 8) & 0xffff) : 1;
    if (N > 0 && N % tex_group) return UMR_ERR_ARG;
    if (alpha_only && ids_only) return UMR_ERR_ARG;
    if (ids_only && (func_id_rgb != 0 || !aggrs_info)) return UMR_ERR_ARG;
    // (the packed state is addressed with 32-bit byte offsets per mesh: 16 B x image_size^2 < 2^32)
    if (packed && (alpha_only || ids_only || func_id_rgb != 1 || !pooled_out || !background || !aggrs_info || (image_size & 7) || image_size > 8192))
        return UMR_ERR_ARG;
    if (vis_ids && !visibility) return UMR_ERR_ARG;
    if (!faces || (!soft_colors && !ids_only && !packed) || !workspace) return UMR_ERR_ARG;
    if (!alpha_only && !ids_only && (!textures || !aggrs_info)) return UMR_ERR_ARG;
    if (N <= 0 || F <= 0 || TS <= 0 || image_size <= 0) return UMR_ERR_ARG;
    bool general = false;
    if (!modes_ok(func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type, TS, &R, &general)) return UMR_ERR_ARG;
    if (general && (alpha_only || ids_only || pooled_out || packed)) return UMR_ERR_ARG;   // fused variants exist for UMR's modes only
    if (visibility && (general || alpha_only || ids_only || func_id_rgb != 1)) return UMR_ERR_ARG;
    if (workspace_bytes < umr_raster_workspace_bytes_for(N, F, image_size)) return UMR_ERR_ARG;
    const int with_p2f = func_id_rgb == 1 && !alpha_only && !(flags & UMR_RASTER_NO_P2F);
    if (with_p2f && (!grid || !p2f_info || !p2f_sum)) return UMR_ERR_ARG;
    if (pooled_out && (image_size & 1)) return UMR_ERR_ARG;
    if ((long long)N * ((image_size + BLK_W - 1) / BLK_W) * ((image_size + BLK_H - 1) / BLK_H) > 0x7fffffffLL)
        return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    RasterArgs A = {};
    A.bbox = (const float4 *)workspace;
    A.rec = (const float *)((char *)workspace + ws_bbox_bytes(N, F));
    A.textures = textures; A.grid = grid; A.aggrs = packed ? nullptr : aggrs_info; A.p2f_info = p2f_info; A.p2f_sum = p2f_sum;
    A.soft_colors = packed ? nullptr : soft_colors; A.pooled = pooled_out; A.vis = visibility;
    A.state = packed ? aggrs_info : nullptr; A.vis_ids_only = vis_ids;
    A.N = N; A.F = F; A.IS = image_size; A.TS = TS; A.R = R;
    A.near_ = near_; A.far_ = far_; A.eps = eps; A.sigma = sigma_val;
    A.threshold = dist_eps * sigma_val;  // :332
    A.gamma = gamma_val; A.double_side = double_side; A.with_p2f = with_p2f; A.tex_group = tex_group;
    A.amb_thr = g_exact_edges ? 20.f * sigma_val : 0.f;
    A.thr = sqrtf(A.threshold); A.nis = -1.f / sigma_val; A.r_range = 1.f / (far_ - near_); A.inv_gamma = 1.f / gamma_val;
    A.tiles_x = (image_size + BLK_W - 1) / BLK_W;
    A.tiles_y = (image_size + BLK_H - 1) / BLK_H;
    if (background) { A.bg_arg = 1; A.bg0 = background[0]; A.bg1 = background[1]; A.bg2 = background[2]; }
    A.dist_mode = func_id_dist; A.alpha_mode = func_id_alpha; A.rgb_mode = func_id_rgb; A.tex_vertex = texture_sample_type;
    A.no_xcd_remap = g_xcd_remap == 1 ? 0 : (g_xcd_remap == 0 ? 1 : 2);
    const int total = N * F;
    UMR_LAUNCH(k_face_setup, (total + 63) / 64, 64, 0, st, faces, faces_info, (float4 *)workspace, (float *)A.rec, total,
                                                      sqrtf(A.threshold), near_, far_, g_thin_face_h);
    setup_bins(A, workspace, N, F, image_size, st);
    const int blocks = N * A.tiles_x * A.tiles_y;
    // start the heavy workgroup blocks first (k_block_order): the list lives in the workspace's work-item region, which belongs to
    // the backward call that follows (a backward that reuses this workspace reads the records and boxes only)
    if (g_block_order && !general && A.sb_count && A.no_xcd_remap == 2 && A.tiles_y % 8 == 0 && A.tiles_x <= 256 && A.tiles_y <= 256) {
        const int per_mesh = (A.tiles_y >> 3) * A.tiles_x;
        const int G = std::max(1, std::min(std::min(N, 65535), BLOCK_ORDER_MAX_ENTRIES / per_mesh));
        const int groups = (N + G - 1) / G;
        if (per_mesh <= BLOCK_ORDER_MAX_ENTRIES && (size_t)groups * 8 * G * per_mesh * sizeof(int) <= ws_order_bytes(N, F)) {
            int *border = (int *)((char *)workspace + ws_order_offset(N, F, image_size));
            UMR_LAUNCH(k_block_order, dim3(8, groups), BLOCK_ORDER_THREADS, 0, st, A.sb_count, border, N, A.tiles_x, A.tiles_y, A.sb_size, A.sb_nx,
                       A.sb_slots, G);
            A.block_order = border; A.block_group = G;
        }
    }
    {
        // algorithmic bytes of one forward launch: 24 IS^2 + F (36 + 12 TS + 16) per mesh
        // (silhouette-only launches are accounted separately, id 2: 4 IS^2 + 36 F)
        ProfScope ps(st, (alpha_only || ids_only) ? 2 : 0,
                     alpha_only ? (double)N * (4.0 * image_size * image_size + 36.0 * F)
                     : ids_only ? (double)N * (8.0 * image_size * image_size + 36.0 * F)
                     // packed state: 16 B / pixel of state + the pooled image (4 planes at a quarter of the pixels = 4 B / pixel)
                     // [+ the id plane]; the planar form: 24 IS^2 (SURVEY 8d's op boundary; its fused pool / visibility planes
                     // are not counted there)
                     : packed ? (double)N * ((20.0 + (visibility ? 4.0 : 0.0)) * image_size * image_size + (double)F * (36.0 + 12.0 * TS + 16.0))
                                : (double)N * (24.0 * image_size * image_size + (double)F * (36.0 + 12.0 * TS + 16.0)));
        // p2f accumulation and face culling are compile-time: as run-time flags they cost SGPRs in every variant
        if (general) {
            UMR_LAUNCH(k_raster_forward_general, blocks, BLK_THREADS, 0, st, A);
        } else if (ids_only) {
            if (double_side) UMR_LAUNCH((k_raster_forward<3, false, true>), blocks, BLK_THREADS, 0, st, A);
            else UMR_LAUNCH((k_raster_forward<3, false, false>), blocks, BLK_THREADS, 0, st, A);
        } else if (alpha_only) {
            UMR_LAUNCH((k_raster_forward<2, false, true>), blocks, BLK_THREADS, 0, st, A);   // alpha does not look at the side
        } else if (func_id_rgb == 0) {
            if (double_side) UMR_LAUNCH((k_raster_forward<0, false, true>), blocks, BLK_THREADS, 0, st, A);
            else UMR_LAUNCH((k_raster_forward<0, false, false>), blocks, BLK_THREADS, 0, st, A);
        } else if (visibility) {
            if (with_p2f) {
                if (double_side) UMR_LAUNCH((k_raster_forward<1, true, true, true>), blocks, BLK_THREADS, 0, st, A);
                else UMR_LAUNCH((k_raster_forward<1, true, false, true>), blocks, BLK_THREADS, 0, st, A);
            } else {
                if (double_side) UMR_LAUNCH((k_raster_forward<1, false, true, true>), blocks, BLK_THREADS, 0, st, A);
                else UMR_LAUNCH((k_raster_forward<1, false, false, true>), blocks, BLK_THREADS, 0, st, A);
            }
        } else if (with_p2f) {
            if (double_side) UMR_LAUNCH((k_raster_forward<1, true, true>), blocks, BLK_THREADS, 0, st, A);
            else UMR_LAUNCH((k_raster_forward<1, true, false>), blocks, BLK_THREADS, 0, st, A);
        } else {
            if (double_side) UMR_LAUNCH((k_raster_forward<1, false, true>), blocks, BLK_THREADS, 0, st, A);
            else UMR_LAUNCH((k_raster_forward<1, false, false>), blocks, BLK_THREADS, 0, st, A);
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
    // the rgb gradient reaches the texels only, the alpha gradient the geometry: the shared mask / texture render of
    // train_s1 / train_s2 (silhouette backward on the render's alpha plane + texel-only backward) in ONE pass over the pairs
    const bool alpha_geom = (grad_is_pooled & UMR_BWD_ALPHA_GEOMETRY) != 0;
    const bool packed = (grad_is_pooled & UMR_BWD_PACKED_STATE) != 0;    // aggrs_info = the forward's packed saved state
    const bool reuse_ws = (grad_is_pooled & UMR_BWD_REUSE_WORKSPACE) != 0;
    const int tex_group = ((grad_is_pooled >> 8) & 0xffff) ? ((grad_is_pooled >> 8) & 0xffff) : 1;
    grad_is_pooled &= UMR_BWD_GRAD_POOLED;
    if (N > 0 && N % tex_group) return UMR_ERR_ARG;
    if (!faces || (!soft_colors && !packed) || !grad_soft_colors || !workspace) return UMR_ERR_ARG;
    if (packed && (!alpha_geom || !aggrs_info || (image_size & 7) || image_size > 8192)) return UMR_ERR_ARG;   // the one-pass kernel reads it, no other
    if (!alpha_only && (!textures || !aggrs_info)) return UMR_ERR_ARG;
    if (alpha_only && (need_grad_textures || !need_grad_faces)) return UMR_ERR_ARG;
    if ((need_grad_faces && !grad_faces) || (need_grad_textures && !grad_textures)) return UMR_ERR_ARG;
    if (N <= 0 || F <= 0 || TS <= 0 || image_size <= 0) return UMR_ERR_ARG;
    bool general = false;
    if (!modes_ok(func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type, TS, &R, &general)) return UMR_ERR_ARG;
    if (general && (alpha_only || grad_is_pooled)) return UMR_ERR_ARG;
    if (workspace_bytes < umr_raster_workspace_bytes_for(N, F, image_size)) return UMR_ERR_ARG;
    if (grad_is_pooled && (image_size & 1)) return UMR_ERR_ARG;
    if (!need_grad_faces && !need_grad_textures) return UMR_OK;
    if (alpha_geom && (alpha_only || !need_grad_faces || !need_grad_textures || func_id_rgb != 1 || general)) return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    RasterArgs A = {};
    A.bbox = (const float4 *)workspace;
    A.rec = (const float *)((char *)workspace + ws_bbox_bytes(N, F));
    A.textures = textures; A.aggrs = packed ? nullptr : (float *)aggrs_info; A.soft_colors = packed ? nullptr : (float *)soft_colors;
    A.state = packed ? (float *)aggrs_info : nullptr;
    A.grad_colors = grad_soft_colors; A.grad_faces = grad_faces; A.grad_textures = grad_textures;
    A.N = N; A.F = F; A.IS = image_size; A.TS = TS; A.R = R;
    A.near_ = near_; A.far_ = far_; A.eps = eps; A.sigma = sigma_val;
    A.threshold = dist_eps * sigma_val;
    A.gamma = gamma_val; A.double_side = double_side;
    A.thr = sqrtf(A.threshold); A.nis = -1.f / sigma_val; A.r_range = 1.f / (far_ - near_); A.inv_gamma = 1.f / gamma_val;
    A.grad_pooled = grad_is_pooled; A.need_gf = need_grad_faces; A.need_gt = need_grad_textures;
    A.tex_group = tex_group;
    A.amb_thr = g_exact_edges ? 20.f * sigma_val : 0.f;
    A.dist_mode = func_id_dist; A.alpha_mode = func_id_alpha; A.rgb_mode = func_id_rgb; A.tex_vertex = texture_sample_type;
    A.tiles_x = (image_size + BLK_W - 1) / BLK_W;
    A.tiles_y = (image_size + BLK_H - 1) / BLK_H;
    const int total = N * F;
    const int blocks = N * A.tiles_x * A.tiles_y;
    const bool lds_ok = (size_t)FM_WAVES * FM_TEXCOPY * FM_TEX_STRIDE(TS) * sizeof(float) <= 48 * 1024;
    const bool face_major = !general && (alpha_only || !(g_bwd_pixel_major || !lds_ok));
    // only the face-major kernels route the two gradients of UMR_BWD_ALPHA_GEOMETRY; the pixel-major pair (umr_debug_set
    // "bwd_pixel_major", or TS beyond the LDS accumulators) would send the rgb gradient into grad_faces: rejected, nothing
    // enqueued (soft_rasterize_cuda.cpp:122-129 raises on what it cannot do, it never returns other data)
    if (alpha_geom && !face_major) return UMR_ERR_ARG;
    // Work-item lists in cost order (k_face_order), every variant.  Round 3 (one wave per face, regular SURVEY 8d scene) had found the
    // cost order slower for the variants that read 28 B of state per pixel (205 -> 263 us) and kept them in index order; with heavy
    // faces split (round 6) it wins there too: vertex-gradient backward 160 / 176 -> 135 / 139 us on the frozen training scenes,
    // 187 -> 182 on the 8d scene (profiles/r06_scene_ab_block_order.jsonl).  One group when the launch has <= 16 meshes, groups of 8 otherwise.
    const int order_mode = alpha_only ? 2 : (((!need_grad_faces || alpha_geom) && func_id_rgb == 1) ? 1 : 0);
    const bool sorted_items = g_face_order != 3;
    const bool ordered = face_major && g_face_order >= 1 && FM_WAVES == 1 &&
                         F % 8 == 0 && F <= 0xffff && F / 8 <= ORDER_MAX_ENTRIES;
    int G = order_group_for(N, F);
    if (g_face_order_group) G = std::min(G, g_face_order_group);
    const OrderLayout OL = order_layout(N, F, G);
    char *op = (char *)workspace + ws_order_offset(N, F, image_size);
    uint2 *order = (uint2 *)op;
    uint2 *split_list = (uint2 *)(op + OL.order_bytes);
    float *slab = (float *)(op + OL.order_bytes + OL.ctr_bytes);
    // the face records and bounding boxes: rebuilt here (the ABI is stateless), unless the caller hands back the workspace its
    // forward call of the SAME faces / N / F / image_size / scalars filled (UMR_BWD_REUSE_WORKSPACE)
    if (!reuse_ws)
        UMR_LAUNCH(k_face_setup, (total + 63) / 64, 64, 0, st, faces, nullptr, (float4 *)workspace, (float *)A.rec, total,
                                                       sqrtf(A.threshold), near_, far_, g_thin_face_h);
    // light variants at small N: four runs of faces per XCD instead of one (fm_owned_face)
    A.fm_split = (face_major && N <= 16 && (alpha_only || !need_grad_faces || alpha_geom) && F % 32 == 0) ? 4 : 1;
    // the one-pass kernel: eight runs, and mesh m's runs rotated to XCD (x + m) % 8 -- a batch of similar poses then loads the XCDs
    // alike (frozen training scenes 121 / 116 -> 117 / 109 us, SURVEY 8d scene 142 -> 140; the silhouette variant loses 3 %: not taken)
    int rotate = g_fm_rotate;
    if (face_major && alpha_geom && N <= 16 && F % 64 == 0 && g_fm_rotate < 0) { A.fm_split = 8; rotate = 1; }
    if (rotate < 0) rotate = 0;
    if (g_fm_runs > 0 && face_major && (F / 8) % g_fm_runs == 0) A.fm_split = g_fm_runs;
    int fm_blocks = N * ((F + FM_WAVES - 1) / FM_WAVES);
    if (ordered) {
        // a part's partial sums: 9 vertex gradients at [0, 9), 3 TS texel gradients from 16 -- in whole slab units
        const int part_floats = need_grad_textures ? 16 + 3 * TS : 16;
        const int units = (part_floats + SPLIT_UNIT - 1) / SPLIT_UNIT;
        OrderArgs O = {};
        O.bbox = A.bbox; O.rec = A.rec; O.order = order; O.split = split_list;
        O.alpha = alpha_only ? soft_colors : nullptr; O.state = packed ? aggrs_info : nullptr;
        O.aggrs = (!alpha_only && !packed && func_id_rgb == 1) ? aggrs_info : nullptr;
        O.far_ = far_; O.r_range = A.r_range; O.inv_gamma = A.inv_gamma;
        O.N = N; O.F = F; O.IS = image_size; O.G = G; O.mode = order_mode; O.sorted = sorted_items ? 1 : 0; O.run_split = A.fm_split; O.rotate = rotate;
        const int gsz = G * (F / 8);
        const int extra_run = std::min(OL.extra, g_face_split > 0 ? gsz / g_split_budget_div + 16 : 0);   // (no split: no padding)
        const int stride_run = gsz + extra_run;
        O.stride = stride_run; O.units_per_part = units;
        O.X_alloc = OL.extra;
        O.X = extra_run / units;                       // extra items this launch's lists may hold (its parts take `units` slab units each)
        O.slabs_per_list = OL.slabs_per_list / units;
        O.T0 = g_face_split;
        O.est = (unsigned *)(op + OL.order_bytes + OL.ctr_bytes + OL.slab_bytes);
        UMR_LAUNCH(k_face_estimate, (total + 255) / 256, 256, 0, st, O);
        UMR_LAUNCH(k_face_order, dim3(8, OL.groups), ORDER_THREADS, 0, st, O);
        A.order = order; A.order_group = G; A.order_stride = stride_run;
        A.slab = slab; A.slab_stride = units * SPLIT_UNIT;
        fm_blocks = OL.groups * 8 * stride_run;
    }
    A.fm_blocks = fm_blocks;
    {
        // Algorithmic bytes of one backward launch, per VARIANT, by SURVEY.md 8d's rule (every op-boundary buffer the variant
        // touches, once; fp32): per pixel -- the gradient planes it reads (4 B per plane per COARSE pixel when the gradient
        // arrives 2x2-pooled, i.e. 1 B per plane per raster pixel), soft_colors only where the colour / alpha terms need it,
        // aggrs_info (8); per face -- faces (36) + faces_info (108) read, textures read only for vertex gradients of a
        // soft-max render, grad_faces (36) / grad_textures (12 TS) written only when requested.
        //   full (vertex + texel grads): (40 | 28) IS^2 + F (180 + 24 TS)          = SURVEY 8d's formula
        //   vertex grads only:           (40 | 28) IS^2 + F (180 + 12 TS)
        //   texel grads only:            (20 | 11) IS^2 + F (144 + 12 TS)          (3 colour-gradient planes + aggrs)
        //   silhouette (id 3):           ( 8 |  5) IS^2 + 72 F
        //   alpha -> geometry + rgb -> texels (one pass): texel-only's + the alpha plane (4) + its gradient plane (4 | 1)
        //                                (28 | 16) IS^2 + F (180 + 12 TS)          (grad_faces written as well)
        const double is2 = (double)image_size * image_size;
        const bool gp = grad_is_pooled != 0;
        double per_mesh;
        if (alpha_only) per_mesh = (gp ? 5.0 : 8.0) * is2 + 72.0 * F;
        else if (alpha_geom) per_mesh = (gp ? 16.0 : 28.0) * is2 + (double)F * (180.0 + 12.0 * TS);
        else if (need_grad_faces && need_grad_textures) per_mesh = (gp ? 28.0 : 40.0) * is2 + (double)F * (180.0 + 24.0 * TS);
        else if (need_grad_faces) per_mesh = (gp ? 28.0 : 40.0) * is2 + (double)F * (180.0 + 12.0 * TS);
        else per_mesh = (gp ? 11.0 : 20.0) * is2 + (double)F * (144.0 + 12.0 * TS);
        ProfScope ps(st, alpha_only ? 3 : 1, (double)N * per_mesh);
        if (general) {
            setup_bins(A, workspace, N, F, image_size, st);
            UMR_LAUNCH(k_raster_backward_general, blocks, BLK_THREADS, 0, st, A);
        } else if (alpha_only) launch_backward_fm<2>(A, st);
        else if (g_bwd_pixel_major || !lds_ok) {  // pixel-major variant (global atomics); kept for A/B and huge TS
            setup_bins(A, workspace, N, F, image_size, st);
            if (func_id_rgb == 0) UMR_LAUNCH((k_raster_backward<0>), blocks, BLK_THREADS, 0, st, A);
            else UMR_LAUNCH((k_raster_backward<1>), blocks, BLK_THREADS, 0, st, A);
        } else if (alpha_geom) launch_backward_fm_ag(A, st);
        else if (func_id_rgb == 0) launch_backward_fm<0>(A, st);
        else launch_backward_fm<1>(A, st);
    }
    if (A.order && g_face_split > 0) {     // the split faces' partial sums -> their gradients, in part order
        SplitReduceArgs R = {};
        R.split = split_list; R.slab = slab; R.grad_faces = grad_faces; R.grad_textures = grad_textures;
        R.F = F; R.TS = TS; R.G = G; R.X = OL.extra; R.slab_stride = A.slab_stride;
        R.need_gf = need_grad_faces; R.need_gt = need_grad_textures;
        UMR_LAUNCH(k_split_reduce, OL.groups * 8, SPLIT_REDUCE_WAVES * 64, 0, st, R);
    }
    return umr_launch_status();
}

}  // extern "C"
