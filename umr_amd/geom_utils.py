"""Geometry helpers with the reference's names and argument meaning (nnutils/geom_utils.py),
backed by HIP kernels (umr_amd/csrc/geometry.hip, losses.hip)."""

from . import functional as UF


def orthographic_proj_withz(X, cam, offset_z=0.):
    """nnutils/geom_utils.py:74-91.  X [B,N,3], cam [B,7] = [s, tx, ty, quat] -> [B,N,3]."""
    return UF.ProjectPointsFunction.apply(X, cam, 3, float(offset_z))


def orthographic_proj(X, cam):
    """nnutils/geom_utils.py:60-72 -> [B,N,2]."""
    return UF.ProjectPointsFunction.apply(X, cam, 2, 0.0)


def sample_textures(texture_flow, images):
    """nnutils/geom_utils.py:41-59.  texture_flow [B,F,T,T,2] in [-1,1], images [B,C,H,W] ->
    [B,F,T,T,C] (bilinear, zero padding, torch-1.1.0 / align_corners=True coordinates)."""
    B, F, T = texture_flow.shape[0], texture_flow.shape[1], texture_flow.shape[-2]
    C = images.shape[1]
    out = UF.GridSampleCLFunction.apply(images, texture_flow.reshape(B, F * T * T, 2))
    return out.view(B, F, T, T, C)
