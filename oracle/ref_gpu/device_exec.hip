// device_exec.hip -- runs the reference's rasterizer DEVICE code natively on the MI355X (gfx950).
//
// TEST INFRASTRUCTURE (see oracle/README.md).  This translation unit contains no reference source.
// oracle/ref_gpu/build_ref_gpu.sh extracts the device-code span (the anonymous namespace, lines 22-659) of
//   /root/reference/external/SoftRas/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu
// into a scratch directory OUTSIDE the repository and this file #includes it.  Nothing is re-defined for it:
// `__global__`, `__device__`, `blockIdx`, `atomicAdd(float*, float)` and the mixed float/double `min` / `max`
// overloads are HIP's own (hip_runtime.h) -- the CUDA execution-model vocabulary the reference is written in IS the
// HIP vocabulary, so hipcc compiles the text as it stands.  That makes this a second execution of the same device
// text, independent of oracle/ref_shim (host loop + builder-written stand-ins): GPU threads, hardware float atomics,
// the vendor's overloads.  The kernels are launched with the reference's own geometry (512 threads per block,
// soft_rasterize_cuda_kernel.cu:687-689, :706, :767).
//
// The ATen host launchers below line 659 are not compiled (they need the reference's torch-1.1 build); the
// extern "C" entry points here take raw device pointers in the argument order of cuda/soft_rasterize_cuda.cpp:62-120.
// Product: oracle/_ref/libsoftras_ref_gfx950.so (git-ignored; travels to the GPU box with the snapshot).  Used only by
// tests/test_gpu_reference_device_code.py.
#include <hip/hip_runtime.h>
#include <cmath>

#include "kernels_body.inc"   // extracted at build time, lives only in the scratch dir

extern "C" {

int refgpu_forward_soft_rasterize(const float *faces, const float *textures, float *faces_info, float *aggrs_info,
                                  float *grid, float *p2f_info, float *p2f_sum, float *soft_colors, int batch_size,
                                  int num_faces, int image_size, int texture_size, float near, float far, float eps,
                                  float sigma_val, int func_id_dist, float dist_eps, float gamma_val, int func_id_rgb,
                                  int func_id_alpha, int texture_sample_type, int double_side, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const int texture_res = int(sqrt((double)texture_size));
    const int threads = 512;
    const dim3 blocks_1((batch_size * num_faces - 1) / threads + 1);
    forward_soft_rasterize_inv_cuda_kernel<float><<<blocks_1, threads, 0, st>>>(faces, faces_info, batch_size, num_faces,
                                                                               image_size);
    const dim3 blocks_2((batch_size * image_size * image_size - 1) / threads + 1);
    forward_soft_rasterize_cuda_kernel<float><<<blocks_2, threads, 0, st>>>(
        faces, textures, faces_info, aggrs_info, grid, p2f_info, p2f_sum, soft_colors, batch_size, num_faces, image_size,
        texture_size, texture_res, near, far, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
        texture_sample_type, double_side != 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int refgpu_backward_soft_rasterize(const float *faces, const float *textures, const float *soft_colors,
                                   const float *faces_info, const float *aggrs_info, float *grad_faces, float *grad_textures,
                                   float *grad_soft_colors, int batch_size, int num_faces, int image_size, int texture_size,
                                   float near, float far, float eps, float sigma_val, int func_id_dist, float dist_eps,
                                   float gamma_val, int func_id_rgb, int func_id_alpha, int texture_sample_type,
                                   int double_side, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const int texture_res = int(sqrt((double)texture_size));
    const int threads = 512;
    const dim3 blocks((batch_size * image_size * image_size - 1) / threads + 1);
    backward_soft_rasterize_cuda_kernel<float><<<blocks, threads, 0, st>>>(
        faces, textures, soft_colors, faces_info, aggrs_info, grad_faces, grad_textures, grad_soft_colors, batch_size,
        num_faces, image_size, texture_size, texture_res, near, far, eps, sigma_val, func_id_dist, dist_eps, gamma_val,
        func_id_rgb, func_id_alpha, texture_sample_type, double_side != 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"
