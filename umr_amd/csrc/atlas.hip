// atlas.hip -- texture-atlas bake for textured OBJ dumps.
//
// Replaces external/SoftRas/soft_renderer/cuda/create_texture_image_cuda_kernel.cu:10-70 together with the host
// prologue/epilogue of functional/save_obj.py:9-35, 50-53 around it: the per-face atlas triangles are generated in
// the kernel instead of being built on the host and uploaded, and the clip -> x255 -> uint8 -> vertical flip that
// the reference does in numpy after a float32 device-to-host copy is an optional fused output, so a dump moves
// H*W*3 bytes over PCIe instead of 4x that.  Face fn owns atlas cell (fn % tile_width, fn / tile_width) of
// res_out^2 pixels; its triangle corners (pixel units, save_obj.py:17-22) are
//   p0 = (col*res + res/2, row*res + 1), p1 = (col*res + 1, (row+1)*res - 2), p2 = ((col+1)*res - 2, (row+1)*res - 2).
// Pure gather, one thread per pixel, HBM-write bound (12 B + 3 B per pixel); arithmetic order and float/double
// promotions follow the reference expression by expression (bit-exact image).
#include "umr_common.h"

namespace {

__global__ void k_texture_atlas(const float *__restrict__ textures, float *__restrict__ image,
                                unsigned char *__restrict__ image_u8, int F, int R, int res, int tile_w, int H,
                                float eps) {
    const int W = tile_w * res;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int x = i % W, y = i / W;
    const int col = x / res, row = y / res, fn = col + row * tile_w;
    float rgb[3] = {1.f, 1.f, 1.f};  // cells past the last face keep the reference's ones-filled canvas
    if (fn < F) {
        const float fc = (float)col, fr = (float)row, fres = (float)res;
        const float p0x = fc * fres + fres / 2, p0y = fr * fres + 1;
        const float p1x = fc * fres + 1, p1y = (fr + 1) * fres - 1 - 1;
        const float p2x = (fc + 1) * fres - 1 - 1, p2y = p1y;
        float inv[9] = {p1y - p2y, p2x - p1x, p1x * p2y - p2x * p1y,
                        p2y - p0y, p0x - p2x, p2x * p0y - p0x * p2y,
                        p0y - p1y, p1x - p0x, p0x * p1y - p1x * p0y};
        const float den = p2x * (p0y - p1y) + p0x * (p1y - p2y) + p1x * (p2y - p0y);
        float w[3], wsum = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a = inv[3 * k] / (den + eps), b = inv[3 * k + 1] / (den + eps), c = inv[3 * k + 2] / (den + eps);
            w[k] = a * (float)x + b * (float)y + c;
            w[k] = (float)fmax(fmin((double)w[k], 1.), 0.);
            wsum += w[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) w[k] /= (wsum + eps);
        const int wx = (int)(w[0] * R), wy = (int)(w[1] * R);
        const int t = ((w[0] + w[1]) * R - wx - wy <= 1) ? (wy * R + wx) : ((R - 1 - wy) * R + (R - 1 - wx));
        const float *__restrict__ tex = textures + ((size_t)fn * R * R + t) * 3;
        rgb[0] = tex[0]; rgb[1] = tex[1]; rgb[2] = tex[2];
    }
    if (image) {
        float *o = image + (size_t)i * 3;
        o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2];
    }
    if (image_u8) {  // save_obj.py:51-52 + the [::-1] of :33 -- row y lands at H-1-y
        unsigned char *o = image_u8 + ((size_t)(H - 1 - y) * W + x) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = (unsigned char)(fminf(fmaxf(rgb[k], 0.f), 1.f) * 255.f);
    }
}

// vt coordinates: the corners above divided by (W-1, H-1) (save_obj.py:28-29); one thread per face
__global__ void k_atlas_uv(float *__restrict__ uv, int F, int res, int tile_w, int H) {
    const int fn = blockIdx.x * blockDim.x + threadIdx.x;
    if (fn >= F) return;
    const float fc = (float)(fn % tile_w), fr = (float)(fn / tile_w), fres = (float)res;
    const float sx = (float)(tile_w * res - 1), sy = (float)(H - 1);
    float *o = uv + (size_t)fn * 6;
    o[0] = (fc * fres + fres / 2) / sx;      o[1] = (fr * fres + 1) / sy;
    o[2] = (fc * fres + 1) / sx;             o[3] = ((fr + 1) * fres - 1 - 1) / sy;
    o[4] = ((fc + 1) * fres - 1 - 1) / sx;   o[5] = o[3];
}

}  // namespace

extern "C" {

int umr_texture_atlas_shape(int F, int res_out, int *height, int *width) {
    if (F <= 0 || res_out < 2 || !height || !width) return -1;
    const int tile_w = (int)sqrt((double)F - 1.) + 1;              // save_obj.py:11-12
    const int tile_h = (int)(((double)F - 1.) / tile_w) + 1;
    *height = tile_h * res_out;
    *width = tile_w * res_out;
    return 0;
}

int umr_texture_atlas(const float *textures, float *image, unsigned char *image_u8, float *uv, int F, int res_in,
                      int res_out, float eps, void *stream) {
    int H, W;
    if (umr_texture_atlas_shape(F, res_out, &H, &W) || !textures || res_in <= 0 || (!image && !image_u8 && !uv))
        return -1;
    hipStream_t s = (hipStream_t)stream;
    const int tile_w = W / res_out;
    if (image || image_u8) {
        const int n = H * W;
        UMR_LAUNCH(k_texture_atlas, (n + 255) / 256, 256, 0, s, textures, image, image_u8, F, res_in, res_out, tile_w, H, eps);
    }
    if (uv) UMR_LAUNCH(k_atlas_uv, (F + 255) / 256, 256, 0, s, uv, F, res_out, tile_w, H);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
}
