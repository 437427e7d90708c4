// host_exec_shim.cpp -- runs the reference's rasterizer DEVICE code on host cores.
//
// TEST INFRASTRUCTURE (see oracle/README.md).  This translation unit contains no
// reference source.  oracle/ref_shim/build_ref.sh extracts the device-code span
// (the anonymous namespace, lines 22-659) of
//   /root/reference/external/SoftRas/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu
// into a scratch build directory OUTSIDE the repository and this file #includes
// it after neutralising the CUDA execution-model keywords.  The arithmetic that
// runs is therefore the reference's own, unmodified; only "who calls the kernel
// body for which thread index" is supplied here (a serial loop replaces <<<>>>).
//
// The product is oracle/_ref/libsoftras_ref.so (git-ignored, never shipped in
// umr_amd/).  It is used to (1) validate oracle/softras_oracle.c and (2) generate
// tests/golden/*.npz.  Built with -ftrivial-auto-var-init=zero because the
// reference's backward_sample_texture (:199-218) returns an uninitialised local
// for non-selected texels; zero is the only value for which its own atomicAdd of
// "nothing" is a no-op.
#include <cmath>
#include <cstdint>

#define __global__
#define __device__
#define __restrict__
#define __forceinline__ inline

struct idx3 { unsigned x, y, z; };
static thread_local idx3 blockIdx, blockDim, threadIdx;

// serial stand-in for the device atomic: one host thread executes all "threads"
template <typename T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }

// CUDA's overload set for mixed float/double min/max (promotion to double)
static inline float  min(float a, float b)   { return fminf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double min(float a, double b)  { return fmin((double)a, b); }
static inline double min(double a, float b)  { return fmin(a, (double)b); }
static inline float  max(float a, float b)   { return fmaxf(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
static inline double max(float a, double b)  { return fmax((double)a, b); }
static inline double max(double a, float b)  { return fmax(a, (double)b); }
using std::exp;
using std::pow;
using std::sqrt;

#include "kernels_body.inc"   // extracted at build time, lives only in the scratch dir
#include "atlas_body.inc"     // create_texture_image_cuda_kernel.cu:8-72, same treatment

extern "C" {

// argument order of cuda/soft_rasterize_cuda.cpp:62-82
int ref_forward_soft_rasterize(const float *faces, const float *textures, float *faces_info,
                               float *aggrs_info, float *grid, float *p2f_info, float *p2f_sum,
                               float *soft_colors, int batch_size, int num_faces, int image_size,
                               int texture_size, float near, float far, float eps, float sigma_val,
                               int func_id_dist, float dist_eps, float gamma_val, int func_id_rgb,
                               int func_id_alpha, int texture_sample_type, int double_side) {
    const int texture_res = int(std::sqrt((double)texture_size));
    blockDim = {1, 1, 1};
    threadIdx = {0, 0, 0};
    for (long i = 0; i < (long)batch_size * num_faces; ++i) {
        blockIdx.x = (unsigned)i;
        forward_soft_rasterize_inv_cuda_kernel<float>(faces, faces_info, batch_size, num_faces, image_size);
    }
    for (long i = 0; i < (long)batch_size * image_size * image_size; ++i) {
        blockIdx.x = (unsigned)i;
        forward_soft_rasterize_cuda_kernel<float>(faces, textures, faces_info, aggrs_info, grid, p2f_info,
                                                  p2f_sum, soft_colors, batch_size, num_faces, image_size,
                                                  texture_size, texture_res, near, far, eps, sigma_val,
                                                  func_id_dist, dist_eps, gamma_val, func_id_rgb,
                                                  func_id_alpha, texture_sample_type, double_side != 0);
    }
    return 0;
}

// argument order of cuda/soft_rasterize_cuda.cpp:100-120
int ref_backward_soft_rasterize(const float *faces, const float *textures, const float *soft_colors,
                                const float *faces_info, const float *aggrs_info, float *grad_faces,
                                float *grad_textures, float *grad_soft_colors, int batch_size,
                                int num_faces, int image_size, int texture_size, float near, float far,
                                float eps, float sigma_val, int func_id_dist, float dist_eps,
                                float gamma_val, int func_id_rgb, int func_id_alpha,
                                int texture_sample_type, int double_side) {
    const int texture_res = int(std::sqrt((double)texture_size));
    blockDim = {1, 1, 1};
    threadIdx = {0, 0, 0};
    for (long i = 0; i < (long)batch_size * image_size * image_size; ++i) {
        blockIdx.x = (unsigned)i;
        backward_soft_rasterize_cuda_kernel<float>(faces, textures, soft_colors, faces_info, aggrs_info,
                                                   grad_faces, grad_textures, grad_soft_colors, batch_size,
                                                   num_faces, image_size, texture_size, texture_res, near,
                                                   far, eps, sigma_val, func_id_dist, dist_eps, gamma_val,
                                                   func_id_rgb, func_id_alpha, texture_sample_type,
                                                   double_side != 0);
    }
    return 0;
}

// launcher of cuda/create_texture_image_cuda_kernel.cu:74-107: one "thread" per atlas pixel, rounded up to
// whole 1024-thread blocks exactly as the reference launches it (the kernel has no i < pixels guard of its own)
int ref_create_texture_image(const float *faces_uv, const float *textures, float *image, long image_numel,
                             int num_faces, int texture_res_in, int texture_res_out, int tile_width, float eps) {
    const long threads = 1024, blocks = (image_numel / 3 - 1) / threads + 1;
    blockDim = {1, 1, 1};
    threadIdx = {0, 0, 0};
    for (long i = 0; i < blocks * threads; ++i) {
        blockIdx.x = (unsigned)i;
        create_texture_image_cuda_kernel<float>(faces_uv, textures, image, (size_t)image_numel, (size_t)num_faces,
                                                (size_t)texture_res_in, (size_t)texture_res_out,
                                                (size_t)tile_width, eps);
    }
    return 0;
}
}
