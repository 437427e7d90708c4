// pair_host.cpp -- TEST INFRASTRUCTURE: the product's k_face_setup + eval_pair + clip_depth (raster_core.h, unmodified source)
// compiled for the host through device_shim.h and exported for ctypes.  One lane at a time.
#include "device_shim.h"
#include "../../umr_amd/csrc/raster_core.h"

extern "C" {

// faces [n,9] -> per (face, pixel): live flag, soft fragment D, unclipped barycentrics, dx, dy, clipped depth zp
int host_pairs(const float *faces, int n, const float *xp, const float *yp, int npix, float thr, float threshold, float nis,
               float near_, float far_, unsigned char *live, float *frag, float *dxy, float *zp_out) {
    float *rec = new float[(size_t)n * REC];
    float4 *bbox = new float4[n];
    blockDim.x = 1;
    for (int i = 0; i < n; ++i) {
        blockIdx.x = (unsigned)i; threadIdx.x = 0;
        k_face_setup(faces, nullptr, bbox, rec, n, thr, near_, far_, nullptr, 0);
    }
    for (int i = 0; i < n; ++i) {
        Face fc;
        load_face(fc, rec + (size_t)i * REC);
        for (int p = 0; p < npix; ++p) {
            Pair pr;
            const bool ok = eval_pair(pr, fc, xp[p], yp[p], threshold, nis);
            const size_t o = (size_t)i * npix + p;
            live[o] = ok ? 1 : 0;
            frag[o] = pr.frag;
            dxy[2 * o] = pr.dx; dxy[2 * o + 1] = pr.dy;
            float q0, q1, q2;
            zp_out[o] = ok ? clip_depth(q0, q1, q2, pr, fc) : 0.f;
        }
    }
    delete[] rec; delete[] bbox;
    return 0;
}

// the conservative tile-vs-dilated-triangle test both raster directions cull with: 1 = "some pixel of the tile may survive"
// tiles: [ntiles,4] = (cx, cy, hx, hy) centre and half extent of the tile's pixel-centre rectangle
int host_tile_may_hit(const float *faces, int n, const float *tiles, int ntiles, float thr, unsigned char *out) {
    float *rec = new float[(size_t)n * REC];
    float4 *bbox = new float4[n];
    blockDim.x = 1;
    for (int i = 0; i < n; ++i) {
        blockIdx.x = (unsigned)i; threadIdx.x = 0;
        k_face_setup(faces, nullptr, bbox, rec, n, thr, 1.f, 100.f, nullptr, 0);
    }
    for (int i = 0; i < n; ++i) {
        const float4 *q = (const float4 *)(rec + (size_t)i * REC + R_INV);
        const float4 bb = bbox[i];
        for (int t = 0; t < ntiles; ++t) {
            const float cx = tiles[4 * t], cy = tiles[4 * t + 1], hx = tiles[4 * t + 2], hy = tiles[4 * t + 3];
            bool hit = !(cx - hx > bb.y || cx + hx < bb.x || cy - hy > bb.w || cy + hy < bb.z);   // the kernels' bbox test first
            if (hit) hit = tile_may_hit(q[0], q[1], q[2], cx, cy, hx, hy, thr);
            out[(size_t)i * ntiles + t] = hit ? 1 : 0;
        }
    }
    delete[] rec; delete[] bbox;
    return 0;
}

}
