"""Geometry helpers with the reference's names and argument meaning (nnutils/geom_utils.py),
backed by HIP kernels (umr_amd/csrc/geometry.hip, losses.hip)."""

from . import functional as UF


def orthographic_proj_withz(X, cam, offset_z=0.):
    """nnutils/geom_utils.py:74-91.  X [B,N,3], cam [B,7] = [s, tx, ty, quat] -> [B,N,3]."""
    return UF.project_points(X, cam, 3, float(offset_z))


def orthographic_proj(X, cam):
    """nnutils/geom_utils.py:60-72 -> [B,N,2]."""
    return UF.project_points(X, cam, 2, 0.0)


def sample_textures(texture_flow, images):
    """nnutils/geom_utils.py:41-59.  texture_flow [B,F,T,T,2] in [-1,1], images [B,C,H,W] ->
    [B,F,T,T,C] (bilinear, zero padding, torch-1.1.0 / align_corners=True coordinates)."""
    B, F, T = texture_flow.shape[0], texture_flow.shape[1], texture_flow.shape[-2]
    C = images.shape[1]
    out = UF.grid_sample_cl(images, texture_flow.reshape(B, F * T * T, 2))
    return out.view(B, F, T, T, C)


def rotate_cam(cam, angle=90, axis=[0, 1, 0], extra_elev=False):
    """nnutils/geom_utils.py:167-193: every camera [B,7] rotated by angle[b] degrees about `axis` (the reference's per-sample
    numpy / cv2.Rodrigues / quaternion_from_matrix round trip through the host, one kernel here).  `angle`: scalar, sequence
    or tensor [B].  Forward only, as in the reference (it rebuilds the camera from numpy values: no gradient)."""
    import ctypes
    import torch
    from . import _lib
    c = cam.detach().to(torch.float32).contiguous()
    B = c.shape[0]
    a = torch.as_tensor(angle, dtype=torch.float32, device=c.device).reshape(-1)
    if a.numel() == 1:
        a = a.expand(B)
    a = a.contiguous()
    out = torch.empty_like(c)
    ax = (ctypes.c_float * 3)(*[float(v) for v in axis])
    _lib.check(_lib.lib().umr_rotate_cam_axis(_lib.ptr(c), _lib.ptr(a), ax, _lib.ptr(out), B, _lib.stream_ptr(c.device)),
               "umr_rotate_cam_axis")
    return out
