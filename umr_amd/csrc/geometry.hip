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

struct Light { float ambient, directional, cr, cg, cb, dx, dy, dz; };

// One thread per (view, face): rotates the three corners, writes the 9 + 9 coordinates and -- when asked -- the face's
// surface light (external/SoftRas/soft_renderer/lighting.py:50-57, functional/ambient_lighting.py:17,
// directional_lighting.py:26-27 on mesh.py:111-118's normals):
//   n = normalize(cross(p2 - p1, p0 - p1), eps 1e-6);  light = ambient c + directional c relu(n . d)
// evaluated on the PRE look_at coordinates like the reference (lighting precedes the transform, renderer.py:89-91; the
// look_at step only shifts z, so the edge vectors are the same).
__global__ void k_project_faces(const float *__restrict__ verts, const float *__restrict__ cams,
                                const int *__restrict__ faces_idx, float *__restrict__ face_pre,
                                float *__restrict__ face_out, float *__restrict__ light_out, int N, int V, int F,
                                float offset_z, float eye_z, int group, const Light L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (n, f)
    if (i >= N * F) return;
    const int n = i / F, m = n / group;                    // view n renders mesh m = n / group (K views per mesh)
    const Cam cm = load_cam(cams, n);
    const int *fi = faces_idx + ((size_t)m * F + (i - n * F)) * 3;
    float P[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *v = verts + ((size_t)m * V + fi[c]) * 3;
        UMR_TRAP_IF(umr_bad(v[0]) | umr_bad(v[1]) | umr_bad(v[2]) | umr_bad(cm.s) | umr_bad(cm.tx) | umr_bad(cm.w), 10);
        float r1, r2, r3;
        quat_rot(cm, v[0], v[1], v[2], r1, r2, r3);
        P[c * 3] = cm.s * r1 + cm.tx;
        P[c * 3 + 1] = -(cm.s * r2 + cm.ty);
        P[c * 3 + 2] = cm.s * r3 + offset_z;
    }
    if (face_pre) {
        float *o = face_pre + (size_t)i * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = P[k];
    }
    float *o = face_out + (size_t)i * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = (k % 3 == 2) ? P[k] - eye_z : P[k];
    if (light_out) {
        const float ax = P[0] - P[3], ay = P[1] - P[4], az = P[2] - P[5];      // v10
        const float bx = P[6] - P[3], by = P[7] - P[4], bz = P[8] - P[5];      // v12
        const float nx = by * az - bz * ay, ny = bz * ax - bx * az, nz = bx * ay - by * ax;   // cross(v12, v10)
        const float nrm = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-6f);
        const float cosine = fmaxf((nx / nrm) * L.dx + (ny / nrm) * L.dy + (nz / nrm) * L.dz, 0.f);
        float *lo = light_out + (size_t)i * 3;
        lo[0] = L.ambient * L.cr + L.directional * (L.cr * cosine);
        lo[1] = L.ambient * L.cg + L.directional * (L.cg * cosine);
        lo[2] = L.ambient * L.cb + L.directional * (L.cb * cosine);
    }
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

// scatter-add face-corner gradients onto projected vertices: gproj[n, v, :] (zeroed beforehand).  One thread per
// (view, face); the gradient of the surface light (when given) is chained through normalize / cross here, using the
// projected corners saved by the forward.
__global__ void k_scatter_face_grads(const float *__restrict__ g_out, const float *__restrict__ g_pre,
                                     const float *__restrict__ g_light, const float *__restrict__ face_out,
                                     const int *__restrict__ faces_idx, float *__restrict__ gproj, int N, int V,
                                     int F, int group, const Light L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (n, f)
    if (i >= N * F) return;
    const int n = i / F;
    const int *fi = faces_idx + ((size_t)(n / group) * F + (i - n * F)) * 3;
    float g[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = g_out[(size_t)i * 9 + k] + (g_pre ? g_pre[(size_t)i * 9 + k] : 0.f);
    UMR_TRAP_IF(umr_bad(g[0]) | umr_bad(g[1]) | umr_bad(g[2]) | umr_bad(g[3]) | umr_bad(g[4]) | umr_bad(g[5]) | umr_bad(g[6]) | umr_bad(g[7]) | umr_bad(g[8]), 11);
    if (g_light) {
        const float *P = face_out + (size_t)i * 9;
        const float ax = P[0] - P[3], ay = P[1] - P[4], az = P[2] - P[5];
        const float bx = P[6] - P[3], by = P[7] - P[4], bz = P[8] - P[5];
        const float nx = by * az - bz * ay, ny = bz * ax - bx * az, nz = bx * ay - by * ax;
        const float len = sqrtf(nx * nx + ny * ny + nz * nz);
        const float nrm = fmaxf(len, 1e-6f), inv = 1.f / nrm;
        const float ux = nx * inv, uy = ny * inv, uz = nz * inv;
        const float cosine = ux * L.dx + uy * L.dy + uz * L.dz;
        const float *gl = g_light + (size_t)i * 3;
        const float gc = cosine > 0.f ? L.directional * (L.cr * gl[0] + L.cg * gl[1] + L.cb * gl[2]) : 0.f;
        // d cos / d n: (d - u (u.d)) / |n| while |n| > eps, d / eps below it (F.normalize clamps the denominator)
        const float k = len > 1e-6f ? cosine : 0.f;
        const float gnx = gc * (L.dx - ux * k) * inv, gny = gc * (L.dy - uy * k) * inv, gnz = gc * (L.dz - uz * k) * inv;
        // n = b x a  =>  db = a x gn,  da = gn x b
        const float dbx = ay * gnz - az * gny, dby = az * gnx - ax * gnz, dbz = ax * gny - ay * gnx;
        const float dax = gny * bz - gnz * by, day = gnz * bx - gnx * bz, daz = gnx * by - gny * bx;
        g[0] += dax; g[1] += day; g[2] += daz;
        g[6] += dbx; g[7] += dby; g[8] += dbz;
        g[3] -= dax + dbx; g[4] -= day + dby; g[5] -= daz + dbz;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float *d = gproj + ((size_t)n * V + fi[c]) * 3;
        atomicAdd(d, g[c * 3]); atomicAdd(d + 1, g[c * 3 + 1]); atomicAdd(d + 2, g[c * 3 + 2]);
    }
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

// The general form (any axis; the reference's rotate_by = cv2.Rodrigues(rad_angle * axis): rotation by |rad_angle * axis| about
// axis / |axis|): new_q = q_axis (x) q, same normalisation and sign convention.
__global__ void k_rotate_cam_axis(const float *__restrict__ cam, const float *__restrict__ angle_deg, float ax, float ay, float az,
                                  float *__restrict__ out, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const float *c = cam + (size_t)i * 7;
    const float len = sqrtf(ax * ax + ay * ay + az * az);
    const float half = angle_deg[i] * 0.008726646259971648f * len;   // pi / 360 * |axis|
    const float inv = len > 0.f ? 1.f / len : 0.f;
    const float rw = cosf(half), s = sinf(half), rx = s * ax * inv, ry = s * ay * inv, rz = s * az * inv;
    const float qw = c[3], qx = c[4], qy = c[5], qz = c[6];
    // Hamilton product r (x) q
    float nw = rw * qw - rx * qx - ry * qy - rz * qz;
    float nx = rw * qx + rx * qw + ry * qz - rz * qy;
    float ny = rw * qy - rx * qz + ry * qw + rz * qx;
    float nz = rw * qz + rx * qy - ry * qx + rz * qw;
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
        UMR_TRAP_IF(threadIdx.x == 0 && umr_bad(s), 12);
        if (threadIdx.x == 0) grad_cams[(size_t)n * 7 + k] = k < 3 ? s : 2.f * q.s * s;
    }
}

}  // namespace

UMR_TRAP_ACCESSOR(umr_trap_read_geometry)

extern "C" {

static Light make_light(float ambient, float directional, const float *color3, const float *direction3) {
    Light L = {ambient, directional, 1.f, 1.f, 1.f, 0.f, 1.f, 0.f};
    if (color3) { L.cr = color3[0]; L.cg = color3[1]; L.cb = color3[2]; }
    if (direction3) { L.dx = direction3[0]; L.dy = direction3[1]; L.dz = direction3[2]; }
    return L;
}

int umr_project_faces_lit_forward(const float *verts, const float *cams, const int *faces_idx, float *face_pre,
                                  float *face_out, float *light_out, int N, int V, int F, float offset_z, float eye_z,
                                  int mesh_group, float light_ambient, float light_directional, const float *light_color3,
                                  const float *light_direction3, void *stream) {
    if (!verts || !cams || !faces_idx || !face_out || N <= 0 || V <= 0 || F <= 0) return UMR_ERR_ARG;
    if (mesh_group < 1 || N % mesh_group) return UMR_ERR_ARG;
    if ((long long)N * F * 3 > 0x7fffffffLL) return UMR_ERR_ARG;
    const int total = N * F;
    UMR_LAUNCH(k_project_faces, (total + 255) / 256, 256, 0, (hipStream_t)stream,
        verts, cams, faces_idx, face_pre, face_out, light_out, N, V, F, offset_z, eye_z, mesh_group,
        make_light(light_ambient, light_directional, light_color3, light_direction3));
    return umr_launch_status();
}

int umr_project_faces_forward(const float *verts, const float *cams, const int *faces_idx, float *face_pre,
                              float *face_out, int N, int V, int F, float offset_z, float eye_z, int mesh_group,
                              void *stream) {
    return umr_project_faces_lit_forward(verts, cams, faces_idx, face_pre, face_out, nullptr, N, V, F, offset_z, eye_z,
                                         mesh_group, 0.f, 0.f, nullptr, nullptr, stream);
}

size_t umr_project_workspace_bytes(int N, int V) { return N > 0 && V > 0 ? (size_t)N * V * 3 * sizeof(float) : 0; }

int umr_project_faces_lit_backward(const float *grad_face_out, const float *grad_face_pre, const float *grad_light,
                                   const float *face_out, const float *verts, const float *cams, const int *faces_idx,
                                   float *grad_verts, float *grad_cams, int N, int V, int F, int mesh_group,
                                   float light_directional, const float *light_color3, const float *light_direction3,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    if (!grad_face_out || !verts || !cams || !faces_idx || !grad_cams || !workspace) return UMR_ERR_ARG;
    if (grad_light && !face_out) return UMR_ERR_ARG;
    if (N <= 0 || V <= 0 || F <= 0 || (long long)N * F * 3 > 0x7fffffffLL) return UMR_ERR_ARG;
    if (mesh_group < 1 || N % mesh_group) return UMR_ERR_ARG;
    if (workspace_bytes < umr_project_workspace_bytes(N, V)) return UMR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (!umr_zero_async(workspace, umr_project_workspace_bytes(N, V), st)) return UMR_ERR_LAUNCH;
    const int total = N * F;
    UMR_LAUNCH(k_scatter_face_grads, (total + 255) / 256, 256, 0, st, grad_face_out, grad_face_pre, grad_light, face_out, faces_idx,
                                                              (float *)workspace, N, V, F, mesh_group,
                                                              make_light(0.f, light_directional, light_color3, light_direction3));
    UMR_LAUNCH((k_project_backward<0>), N, 256, 0, st, (const float *)workspace, verts, cams, grad_verts, grad_cams, V, mesh_group);
    return umr_launch_status();
}

int umr_project_faces_backward(const float *grad_face_out, const float *grad_face_pre, const float *verts,
                               const float *cams, const int *faces_idx, float *grad_verts, float *grad_cams,
                               int N, int V, int F, int mesh_group, void *workspace, size_t workspace_bytes,
                               void *stream) {
    return umr_project_faces_lit_backward(grad_face_out, grad_face_pre, nullptr, nullptr, verts, cams, faces_idx, grad_verts,
                                          grad_cams, N, V, F, mesh_group, 0.f, nullptr, nullptr, workspace, workspace_bytes,
                                          stream);
}

int umr_project_points_forward(const float *verts, const float *cams, float *out, int N, int V, int out_dim,
                               float offset_z, void *stream) {
    if (!verts || !cams || !out || N <= 0 || V <= 0 || (long long)N * V > 0x7fffffffLL) return UMR_ERR_ARG;
    if (out_dim != 2 && out_dim != 3) return UMR_ERR_ARG;
    const int total = N * V;
    UMR_LAUNCH(k_project_points, (total + 255) / 256, 256, 0, (hipStream_t)stream, verts, cams, out, N, V, out_dim, offset_z);
    return umr_launch_status();
}

int umr_project_points_backward(const float *grad_out, const float *verts, const float *cams, float *grad_verts,
                                float *grad_cams, int N, int V, int out_dim, void *stream) {
    if (!grad_out || !verts || !cams || !grad_cams || N <= 0 || V <= 0) return UMR_ERR_ARG;
    if (out_dim == 2)
        UMR_LAUNCH((k_project_backward<1>), N, 256, 0, (hipStream_t)stream, grad_out, verts, cams, grad_verts, grad_cams, V, 1);
    else if (out_dim == 3)
        UMR_LAUNCH((k_project_backward<2>), N, 256, 0, (hipStream_t)stream, grad_out, verts, cams, grad_verts, grad_cams, V, 1);
    else
        return UMR_ERR_ARG;
    return umr_launch_status();
}

int umr_rotate_cam_axis(const float *cam, const float *angle_deg, const float *axis3, float *out, int B, void *stream) {
    if (!cam || !angle_deg || !axis3 || !out || B <= 0) return UMR_ERR_ARG;
    UMR_LAUNCH(k_rotate_cam_axis, (B + 63) / 64, 64, 0, (hipStream_t)stream, cam, angle_deg, axis3[0], axis3[1], axis3[2], out, B);
    return umr_launch_status();
}

int umr_rotate_cam_y(const float *cam, const float *angle_deg, float *out, int B, void *stream) {
    if (!cam || !angle_deg || !out || B <= 0) return UMR_ERR_ARG;
    UMR_LAUNCH(k_rotate_cam_y, (B + 63) / 64, 64, 0, (hipStream_t)stream, cam, angle_deg, out, B);
    return umr_launch_status();
}

}  // extern "C"
