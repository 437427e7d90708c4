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

}
