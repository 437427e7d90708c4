// geometry.hip -- camera projection + face gather (forward and analytic backward), gfx950.
//
// One kernel replaces the ~70 elementwise launches of the reference's op chain
//   nnutils/geom_utils.py:74-91 (orthographic_proj_withz), :147-165 (quat_rotate), :119-144
//   (hamilton_product x2), nnutils/smr.py:36 (y flip),
//   external/SoftRas/soft_renderer/functional/face_vertices.py:19-22 (gather),
//   functional/look_at.py:48-60 (eye on the z axis => R = I, z -= eye_z), orthogonal.py:13-16 (x1).
// The forward evaluates the two Hamilton products in the reference's operation order (zero terms
// dropped: 0*a and +0 are exact) so projected coordinates match the torch chain to rounding.
#include "umr_common.h"

namespace {

struct Cam { float s, tx, ty, w, a, b, c; };

__device__ __forceinline__ Cam load_cam(const float *__restrict__ cams, int n) {
    const float *p = cams + (size_t)n * 7;
    Cam cm = {p[0], p[1], p[2], p[3], p[4], p[5], p[6]};
    return cm;
}

// r = q (0,X) conj(q), q NOT normalised (geom_utils.py:160-164)
__device__ __forceinline__ void quat_rot(const Cam &q, float x, float y, float z, float &r1, float &r2, float &r3) {
    // p = (0,X) * conj(q)
    const float p0 = (x * q.a + y * q.b) + z * q.c;
    const float p1 = (x * q.w - y * q.c) + z * q.b;
    const float p2 = (x * q.c + y * q.w) - z * q.a;
    const float p3 = (y * q.a - x * q.b) + z * q.w;
    // r = q * p
    r1 = ((q.w * p1 + q.a * p0) + q.b * p3) - q.c * p2;
    r2 = ((q.w * p2 - q.a * p3) + q.b * p0) + q.c * p1;
    r3 = ((q.w * p3 + q.a * p2) - q.b * p1) + q.c * p0;
}

__global__ void k_project_faces(const float *__restrict__ verts, const float *__restrict__ cams,
                                const int *__restrict__ faces_idx, float *__restrict__ face_pre,
                                float *__restrict__ face_out, int N, int V, int F, float offset_z, float eye_z,
                                int group) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (n, f, corner)
    if (i >= N * F * 3) return;
    const int n = i / (F * 3), m = n / group;              // view n renders mesh m = n / group (K views per mesh)
    const Cam cm = load_cam(cams, n);
    const int vi = faces_idx[(size_t)m * F * 3 + (i - n * F * 3)];
    const float *v = verts + ((size_t)m * V + vi) * 3;
    float r1, r2, r3;
    quat_rot(cm, v[0], v[1], v[2], r1, r2, r3);
    const float X = cm.s * r1 + cm.tx;
    const float Y = -(cm.s * r2 + cm.ty);
    const float Z = cm.s * r3 + offset_z;
    if (face_pre) { float *o = face_pre + (size_t)i * 3; o[0] = X; o[1] = Y; o[2] = Z; }
    float *o = face_out + (size_t)i * 3;
    o[0] = X; o[1] = Y; o[2] = Z - eye_z;
}

__global__ void k_project_points(const float *__restrict__ verts, const float *__restrict__ cams,
                                 float *__restrict__ out, int N, int V, int out_dim, float offset_z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * V) return;
    const Cam cm = load_cam(cams, i / V);
    const float *v = verts + (size_t)i * 3;
    float r1, r2, r3;
    quat_rot(cm, v[0], v[1], v[2], r1, r2, r3);
    out[(size_t)i * out_dim] = cm.s * r1 + cm.tx;
    out[(size_t)i * out_dim + 1] = cm.s * r2 + cm.ty;
    if (out_dim == 3) out[(size_t)i * 3 + 2] = cm.s * r3 + offset_z;
}

// scatter-add face-corner gradients onto projected vertices: gproj[n, v, :] (zeroed beforehand)
__global__ void k_scatter_face_grads(const float *__restrict__ g_out, const float *__restrict__ g_pre,
                                     const int *__restrict__ faces_idx, float *__restrict__ gproj, int N, int V,
                                     int F, int group) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (n, f, corner)
    if (i >= N * F * 3) return;
    const int n = i / (F * 3);
    const int vi = faces_idx[(size_t)(n / group) * F * 3 + (i - n * F * 3)];
    float gx = g_out[(size_t)i * 3], gy = g_out[(size_t)i * 3 + 1], gz = g_out[(size_t)i * 3 + 2];
    if (g_pre) { gx += g_pre[(size_t)i * 3]; gy += g_pre[(size_t)i * 3 + 1]; gz += g_pre[(size_t)i * 3 + 2]; }
    float *d = gproj + ((size_t)n * V + vi) * 3;
    atomicAdd(d, gx); atomicAdd(d + 1, gy); atomicAdd(d + 2, gz);
}

// geom_utils.rotate_cam (nnutils/geom_utils.py:167-193) for the y axis: new_q = q_y(angle) (x) q, renormalised, w >= 0
// representative (what transformations.quaternion_from_matrix(isprecise=True) returns).  One thread per camera; the
// torch formulation was ~20 elementwise launches for [B,7] values.
__global__ void k_rotate_cam_y(const float *__restrict__ cam, const float *__restrict__ angle_deg,
                               float *__restrict__ out, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const float *c = cam + (size_t)i * 7;
    const float half = angle_deg[i] * 0.008726646259971648f;   // pi / 360
    const float rw = cosf(half), ry = sinf(half);
    const float qw = c[3], qx = c[4], qy = c[5], qz = c[6];
    float nw = rw * qw - ry * qy, nx = rw * qx + ry * qz, ny = rw * qy + ry * qw, nz = rw * qz - ry * qx;
    const float nrm = fmaxf(sqrtf(nw * nw + nx * nx + ny * ny + nz * nz), 1e-12f);
    nw /= nrm; nx /= nrm; ny /= nrm; nz /= nrm;
    if (nw < 0.f) { nw = -nw; nx = -nx; ny = -ny; nz = -nz; }
    float *o = out + (size_t)i * 7;
    o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = nw; o[4] = nx; o[5] = ny; o[6] = nz;
}

// One block per mesh.  With M(q) = (w^2-|u|^2) I + 2 u u^T + 2 w [u]x  and  P = s M X + t:
//   dL/dX = s M^T g,  dL/ds = sum g.(M X),  dL/dt = sum g_xy,
//   dL/dw = 2 s sum [ w (g.X) + g.(u x X) ],
//   dL/du = 2 s sum [ -(g.X) u + (g.u) X + (u.X) g + w (X x g) ]
// MODE 0: g from gproj [N,V,3] with the renderer's y flip; MODE 1: g from grad_out [N,V,2];
// MODE 2: g from grad_out [N,V,3] (no flip).
template <int MODE>
__global__ __launch_bounds__(256) void k_project_backward(const float *__restrict__ gsrc,
                                                          const float *__restrict__ verts,
                                                          const float *__restrict__ cams,
                                                          float *__restrict__ grad_verts,
                                                          float *__restrict__ grad_cams, int V, int group) {
    __shared__ float smem[16];
    const int n = blockIdx.x;
    verts += (size_t)(n / group) * V * 3 - (size_t)n * V * 3;   // view n reads mesh n / group; gradients stay per view
    const Cam q = load_cam(cams, n);
    const float uu = q.a * q.a + q.b * q.b + q.c * q.c;
    const float d = q.w * q.w - uu;
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        const float *X = verts + ((size_t)n * V + v) * 3;
        const float x = X[0], y = X[1], z = X[2];
        float gx, gy, gz;
        if (MODE == 0) {
            const float *g = gsrc + ((size_t)n * V + v) * 3;
            gx = g[0]; gy = -g[1]; gz = g[2];
        } else if (MODE == 1) {
            const float *g = gsrc + ((size_t)n * V + v) * 2;
            gx = g[0]; gy = g[1]; gz = 0.f;
        } else {
            const float *g = gsrc + ((size_t)n * V + v) * 3;
            gx = g[0]; gy = g[1]; gz = g[2];
        }
        const float ug = q.a * gx + q.b * gy + q.c * gz;
        const float ux = q.a * x + q.b * y + q.c * z;
        const float gX = gx * x + gy * y + gz * z;
        // u x g, u x X, X x g
        const float cgx = q.b * gz - q.c * gy, cgy = q.c * gx - q.a * gz, cgz = q.a * gy - q.b * gx;
        const float cxx = q.b * z - q.c * y, cxy = q.c * x - q.a * z, cxz = q.a * y - q.b * x;
        const float xgx = y * gz - z * gy, xgy = z * gx - x * gz, xgz = x * gy - y * gx;
        if (grad_verts) {
            float *o = grad_verts + ((size_t)n * V + v) * 3;
            o[0] += q.s * (d * gx + 2.f * ug * q.a - 2.f * q.w * cgx);
            o[1] += q.s * (d * gy + 2.f * ug * q.b - 2.f * q.w * cgy);
            o[2] += q.s * (d * gz + 2.f * ug * q.c - 2.f * q.w * cgz);
        }
        // r = M X
        const float rx = d * x + 2.f * ux * q.a + 2.f * q.w * cxx;
        const float ry = d * y + 2.f * ux * q.b + 2.f * q.w * cxy;
        const float rz = d * z + 2.f * ux * q.c + 2.f * q.w * cxz;
        acc[0] += gx * rx + gy * ry + gz * rz;
        acc[1] += gx;
        acc[2] += gy;
        acc[3] += q.w * gX + (gx * cxx + gy * cxy + gz * cxz);
        acc[4] += -gX * q.a + ug * x + ux * gx + q.w * xgx;
        acc[5] += -gX * q.b + ug * y + ux * gy + q.w * xgy;
        acc[6] += -gX * q.c + ug * z + ux * gz + q.w * xgz;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const float s = block_sum(acc[k], smem);
        if (threadIdx.x == 0) grad_cams[(size_t)n * 7 + k] = k < 3 ? s : 2.f * q.s * s;
    }
}

}  // namespace

extern "C" {

int umr_project_faces_forward(const float *verts, const float *cams, const int *faces_idx, float *face_pre,
                              float *face_out, int N, int V, int F, float offset_z, float eye_z, int mesh_group,
                              void *stream) {
    if (!verts || !cams || !faces_idx || !face_out || N <= 0 || V <= 0 || F <= 0) return UMR_ERR_ARG;
    if (mesh_group < 1 || N % mesh_group) return UMR_ERR_ARG;
    if ((long long)N * F * 3 > 0x7fffffffLL) return UMR_ERR_ARG;
    const int total = N * F * 3;
    k_project_faces<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(verts, cams, faces_idx, face_pre, face_out,
                                                                        N, V, F, offset_z, eye_z, mesh_group);
    return umr_launch_status();
}

size_t umr_project_workspace_bytes(int N, int V) { return N > 0 && V > 0 ? (size_t)N * V * 3 * sizeof(float) : 0; }

int umr_project_faces_backward(const float *grad_face_out, const float *grad_face_pre, const float *verts,
                               const float *cams, const int *faces_idx, float *grad_verts, float *grad_cams,
                               int N, int V, int F, int mesh_group, void *workspace, size_t workspace_bytes,
                               void *stream) {
    if (!grad_face_out || !verts || !cams || !faces_idx || !grad_cams || !workspace) return UMR_ERR_ARG;
    if (N <= 0 || V <= 0 || F <= 0 || (long long)N * F * 3 > 0x7fffffffLL) return UMR_ERR_ARG;
    if (mesh_group < 1 || N % mesh_group) return UMR_ERR_ARG;
    if (workspace_bytes < umr_project_workspace_bytes(N, V)) return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, umr_project_workspace_bytes(N, V), st) != hipSuccess) return UMR_ERR_LAUNCH;
    const int total = N * F * 3;
    k_scatter_face_grads<<<(total + 255) / 256, 256, 0, st>>>(grad_face_out, grad_face_pre, faces_idx,
                                                              (float *)workspace, N, V, F, mesh_group);
    k_project_backward<0><<<N, 256, 0, st>>>((const float *)workspace, verts, cams, grad_verts, grad_cams, V, mesh_group);
    return umr_launch_status();
}

int umr_project_points_forward(const float *verts, const float *cams, float *out, int N, int V, int out_dim,
                               float offset_z, void *stream) {
    if (!verts || !cams || !out || N <= 0 || V <= 0 || (long long)N * V > 0x7fffffffLL) return UMR_ERR_ARG;
    if (out_dim != 2 && out_dim != 3) return UMR_ERR_ARG;
    const int total = N * V;
    k_project_points<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(verts, cams, out, N, V, out_dim, offset_z);
    return umr_launch_status();
}

int umr_project_points_backward(const float *grad_out, const float *verts, const float *cams, float *grad_verts,
                                float *grad_cams, int N, int V, int out_dim, void *stream) {
    if (!grad_out || !verts || !cams || !grad_cams || N <= 0 || V <= 0) return UMR_ERR_ARG;
    if (out_dim == 2)
        k_project_backward<1><<<N, 256, 0, (hipStream_t)stream>>>(grad_out, verts, cams, grad_verts, grad_cams, V, 1);
    else if (out_dim == 3)
        k_project_backward<2><<<N, 256, 0, (hipStream_t)stream>>>(grad_out, verts, cams, grad_verts, grad_cams, V, 1);
    else
        return UMR_ERR_ARG;
    return umr_launch_status();
}

int umr_rotate_cam_y(const float *cam, const float *angle_deg, float *out, int B, void *stream) {
    if (!cam || !angle_deg || !out || B <= 0) return UMR_ERR_ARG;
    k_rotate_cam_y<<<(B + 63) / 64, 64, 0, (hipStream_t)stream>>>(cam, angle_deg, out, B);
    return umr_launch_status();
}

}  // extern "C"
