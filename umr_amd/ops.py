"""The C-ABI rasterizer entry points as registered PyTorch custom ops (`torch.ops.umr.*`).

The reference exposes its rasterizer to PyTorch as an extension module with two functions
(external/SoftRas/soft_renderer/cuda/soft_rasterize_cuda.cpp:141-144: forward_soft_rasterize, backward_soft_rasterize)
and wraps them in an autograd.Function (functional/soft_rasterize.py:9-108).  Here the same two calls -- umr_raster_forward /
umr_raster_backward of libumr_hip.so -- are registered through torch.library with

  * a device implementation (ctypes call, enqueued on the current HIP stream, no host synchronisation),
  * a fake (meta) implementation, so shapes / dtypes propagate under FakeTensor tracing (torch.compile, export,
    torch.library.opcheck) without touching the GPU,
  * an autograd formula (umr::soft_rasterize's backward is umr::soft_rasterize_backward),

so the hot-path step is visible to PyTorch's graph machinery as ordinary operators and can be captured in a HIP graph
(tests/test_gpu_round2.py::test_hot_path_step_replays_from_a_hip_graph).  umr_amd.functional.soft_rasterize and the
SoftRenderer route through these ops.

  umr::soft_rasterize(face_vertices[N,F,3,3], textures[N/G,F,TS,3], image_size, background[3], near, far, fill_back, eps,
                      sigma_val, dist_eps, gamma_val, modes (pack_modes: 0 hard / 1 soft-max colour with UMR's euclidean +
                      prod + surface modes; the other ids of the reference binding in the upper bits), pool, need_p2f,
                      want_visibility (soft-max colour with UMR's modes: also the hard render's z-buffer planes))
        -> (image [N,4,S,S] (S = image_size or image_size/2 when pool), p2f [N,F,2], aggrs_info [N,2,IS,IS],
            soft_colors [N,4,IS,IS] (saved state; == image when not pool), visibility [N,2,IS,IS] | empty)
  umr::soft_rasterize_backward(face_vertices, textures, soft_colors, aggrs_info, grad_image, <same scalars>,
                               need_grad_faces, need_grad_textures) -> (grad_face_vertices, grad_textures)
  umr::soft_rasterize_alpha_geometry(<same arguments>, lean) / umr::soft_rasterize_alpha_geometry_backward(..., lean): the one
        render the training steps make for the mask AND the texture term (below); lean = its saved state as ONE packed buffer
        in the one-pass backward's layout (UMR_RASTER_PACKED_STATE / UMR_BWD_PACKED_STATE), returned in the aggrs_info slot
  umr::silhouette(face_vertices, image_size, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, pool)
        -> (alpha_out [N,S,S], alpha [N,IS,IS] (saved state))
  umr::silhouette_backward(face_vertices, alpha, grad_alpha_out, <same scalars>) -> grad_face_vertices
"""
import ctypes
import math
from typing import List, Tuple

import torch
from torch.library import custom_op

from . import _lib
from ._lib import ptr

BWD_ALPHA_GEOMETRY = 4      # UMR_BWD_ALPHA_GEOMETRY (include/umr_hip.h)
BWD_PACKED_STATE = 8        # UMR_BWD_PACKED_STATE
BWD_REUSE_WORKSPACE = 16    # UMR_BWD_REUSE_WORKSPACE
RASTER_PACKED_STATE, RASTER_VIS_IDS_ONLY = 8, 16   # UMR_RASTER_PACKED_STATE, UMR_RASTER_VIS_IDS_ONLY
ONE_PASS_MAX_TS = 1023      # texels per face the face-major backward's LDS accumulators take (4 copies x (3 TS | 1) floats <= 48 KB)


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def pack_modes(func_id_rgb, func_id_dist=2, func_id_alpha=2, texture_sample_type=0):
    """The four mode ids of the reference binding (functional/soft_rasterize.py:21-24) in one op argument.  Distance and
    alpha are stored relative to UMR's own choice (euclidean = 2, prod = 2), so a plain 0 / 1 means "hard / soft-max colour
    with UMR's modes"."""
    return int(func_id_rgb) | ((int(func_id_dist) ^ 2) << 4) | ((int(func_id_alpha) ^ 2) << 8) | (int(texture_sample_type) << 12)


def unpack_modes(modes):
    return modes & 0xf, ((modes >> 4) & 0xf) ^ 2, ((modes >> 8) & 0xf) ^ 2, (modes >> 12) & 0xf   # rgb, dist, alpha, texture


def _scalars(image_size, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, modes):
    # the 12 scalars of soft_rasterize_cuda.cpp:71-82
    rgb, dist, alpha, tex = unpack_modes(int(modes))
    return (int(image_size), float(near), float(far), float(eps), float(sigma_val), dist, float(math.log(1. / dist_eps - 1.)),
            float(gamma_val), rgb, alpha, tex, 1 if fill_back else 0)


def lean_state_ok(image_size, TS, modes, pool):
    """Can the shared render keep its saved state packed (UMR_RASTER_PACKED_STATE)?  Needs the fused pool, an image size that is
    a multiple of 8, UMR's own modes and a texel count the one-pass backward takes."""
    return bool(pool) and int(image_size) % 8 == 0 and int(modes) == 1 and int(TS) <= ONE_PASS_MAX_TS


def _raster_forward(face_vertices, textures, image_size, background, near, far, fill_back, eps, sigma_val, dist_eps,
                    gamma_val, modes, pool, need_p2f, want_visibility, lean=False):
    """Body of umr::soft_rasterize and umr::soft_rasterize_alpha_geometry: one umr_raster_forward_vis call.
    lean (alpha_geometry only): the saved state is ONE packed buffer in the one-pass backward's layout, returned in the
    aggrs_info slot; no full-resolution image or soft-max plane is written, and visibility is the id plane [N,IS,IS] alone."""
    from .functional import standard_grid
    L = _lib.lib()
    dev = face_vertices.device
    fv, tex = _f32c(face_vertices), _f32c(textures)
    N, F = fv.shape[:2]
    if fv.dim() != 4 or fv.shape[2:] != (3, 3) or tex.dim() != 4 or tex.shape[0] < 1 or N % tex.shape[0] \
            or N // tex.shape[0] > 65535 or tex.shape[1] != F or tex.shape[3] != 3:
        raise RuntimeError("soft_rasterize: face_vertices must be [N,F,3,3] and textures [N or N/G,F,TS,3]; got "
                           "%s and %s" % (tuple(fv.shape), tuple(tex.shape)))   # kernels index textures by (n//G, f)
    G = N // tex.shape[0]     # G views share one texture set (the reference repeats textures x K, loss_utils.py:303-306)
    TS, IS = tex.shape[2], int(image_size)
    # the reference fills 0.8 GB of buffers per N=128 call (functional/soft_rasterize.py:47-55); here the kernel takes the
    # background colour by value and writes every plane, so nothing is pre-filled
    if lean:
        if not lean_state_ok(IS, TS, modes, pool):
            raise RuntimeError("soft_rasterize_alpha_geometry(lean): needs pool, image_size %% 8 == 0, UMR's modes, TS <= %d" % ONE_PASS_MAX_TS)
        # the packed state's size is the LIBRARY's to say (16 B / pixel today): a layout change on the C side then fails here,
        # not as a write past the tensor
        state_bytes = L.umr_raster_state_bytes(N, IS)
        if state_bytes == 0 or state_bytes % (4 * N):
            raise RuntimeError("soft_rasterize_alpha_geometry(lean): umr_raster_state_bytes(%d, %d) = %d" % (N, IS, state_bytes))
        aggrs_info = torch.empty(N, state_bytes // (4 * N), device=dev, dtype=torch.float32)
        soft_colors = None
    else:
        aggrs_info = torch.empty(N, 2, IS, IS, device=dev, dtype=torch.float32)
        soft_colors = torch.empty(N, 4, IS, IS, device=dev, dtype=torch.float32)
    p2f_acc = torch.zeros(2, N, F, 2, device=dev, dtype=torch.float32)
    bg = (ctypes.c_float * 3)(float(background[0]), float(background[1]), float(background[2]))
    pooled = torch.empty(N, 4, IS // 2, IS // 2, device=dev, dtype=torch.float32) if pool else None
    with_p2f = need_p2f and (modes & 0xf) == 1
    grid = standard_grid(IS, dev) if with_p2f else None
    ws_bytes = L.umr_raster_workspace_bytes_for(N, F, int(image_size))
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    sc = _scalars(IS, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, modes)
    # the hard render's (depth, face id) planes of the same faces, from the same visits (umr_raster_forward_vis)
    vis = (torch.empty((N, IS, IS) if lean else (N, 2, IS, IS), device=dev, dtype=torch.float32)) if want_visibility else None
    flags = (0 if need_p2f else 1) | (G << 8)
    if lean:
        flags |= RASTER_PACKED_STATE | (RASTER_VIS_IDS_ONLY if want_visibility else 0)
    if getattr(_lib, "TAP", None) is not None:
        _lib.TAP("raster_forward", dict(face_vertices=fv, textures=tex, lean=lean, image_size=IS))
    rc = L.umr_raster_forward_vis(ptr(fv), ptr(tex), None, ptr(aggrs_info), ptr(grid), ptr(p2f_acc[0]), ptr(p2f_acc[1]),
                                  ptr(soft_colors), ptr(pooled), N, F, TS, *sc, flags, bg, ptr(ws),
                                  ws_bytes, _lib.stream_ptr(dev), ptr(vis))
    _lib.check(rc, "umr_raster_forward_vis")
    p2f = p2f_acc[0] / p2f_acc[1].clamp_min(1e-12)  # functional/soft_rasterize.py:73
    if lean:    # 4th slot: the raster workspace this call filled (face records, bounding boxes) -- the backward reads it instead of
        # rebuilding it (UMR_BWD_REUSE_WORKSPACE)
        return pooled, p2f, aggrs_info, ws, (vis if want_visibility else pooled.new_empty(0))
    # custom-op outputs may not alias each other: without the fused pool the image IS the saved state, returned once more
    # as an empty placeholder in the 4th slot
    return ((pooled if pool else soft_colors), p2f, aggrs_info, (soft_colors if pool else soft_colors.new_empty(0)),
            (vis if want_visibility else soft_colors.new_empty(0)))


@custom_op("umr::soft_rasterize", mutates_args=(), device_types="cuda")
def soft_rasterize_op(face_vertices: torch.Tensor, textures: torch.Tensor, image_size: int, background: List[float],
                      near: float, far: float, fill_back: bool, eps: float, sigma_val: float, dist_eps: float,
                      gamma_val: float, modes: int, pool: bool, need_p2f: bool, want_visibility: bool
                      ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    return _raster_forward(face_vertices, textures, image_size, background, near, far, fill_back, eps, sigma_val, dist_eps,
                           gamma_val, modes, pool, need_p2f, want_visibility)


def _raster_fake(face_vertices, textures, image_size, background, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, modes,
                 pool, need_p2f, want_visibility, lean=False):
    N, F = face_vertices.shape[:2]
    IS = int(image_size)
    S = IS // 2 if pool else IS
    f = lambda *s: face_vertices.new_empty(s, dtype=torch.float32)
    if lean:
        ws_bytes = _lib.lib().umr_raster_workspace_bytes_for(N, F, IS)       # (a host-side size query: no device work)
        return (f(N, 4, S, S), f(N, F, 2), f(N, IS * IS * 4), face_vertices.new_empty((ws_bytes,), dtype=torch.uint8),
                (f(N, IS, IS) if want_visibility else f(0)))
    return (f(N, 4, S, S), f(N, F, 2), f(N, 2, IS, IS), (f(N, 4, IS, IS) if pool else f(0)),
            (f(N, 2, IS, IS) if want_visibility else f(0)))


soft_rasterize_op.register_fake(_raster_fake)


@custom_op("umr::soft_rasterize_backward", mutates_args=(), device_types="cuda")
def soft_rasterize_backward_op(face_vertices: torch.Tensor, textures: torch.Tensor, soft_colors: torch.Tensor,
                               aggrs_info: torch.Tensor, grad_image: torch.Tensor, image_size: int, near: float, far: float,
                               fill_back: bool, eps: float, sigma_val: float, dist_eps: float, gamma_val: float,
                               modes: int, pool: bool, need_grad_faces: bool, need_grad_textures: bool
                               ) -> Tuple[torch.Tensor, torch.Tensor]:
    L = _lib.lib()
    fv, tex = _f32c(face_vertices), _f32c(textures)
    dev = fv.device
    N, F = fv.shape[:2]
    TS = tex.shape[2]
    G = N // tex.shape[0]
    grad_faces = torch.zeros(N, F, 9, device=dev, dtype=torch.float32) if need_grad_faces else None
    grad_textures = torch.zeros(N, F, TS, 3, device=dev, dtype=torch.float32) if need_grad_textures else None   # per view
    g = grad_image.to(torch.float32).contiguous()
    ws_bytes = L.umr_raster_workspace_bytes_for(N, F, int(image_size))
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    sc = _scalars(image_size, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, modes)
    rc = L.umr_raster_backward(ptr(fv), ptr(tex), ptr(soft_colors), None, ptr(aggrs_info), ptr(grad_faces),
                               ptr(grad_textures), ptr(g), (1 if pool else 0) | (G << 8), 1 if need_grad_faces else 0,
                               1 if need_grad_textures else 0, N, F, TS, *sc, ptr(ws), ws_bytes, _lib.stream_ptr(dev))
    _lib.check(rc, "umr_raster_backward")
    gf = grad_faces.view(N, F, 3, 3) if need_grad_faces else fv.new_empty(0)
    if need_grad_textures and G > 1:
        grad_textures = grad_textures.view(N // G, G, F, TS, 3).sum(1)   # autograd of the reference's repeat
    return gf, (grad_textures if need_grad_textures else fv.new_empty(0))


@soft_rasterize_backward_op.register_fake
def _(face_vertices, textures, soft_colors, aggrs_info, grad_image, image_size, near, far, fill_back, eps, sigma_val,
      dist_eps, gamma_val, modes, pool, need_grad_faces, need_grad_textures):
    f = lambda *s: face_vertices.new_empty(s, dtype=torch.float32)
    return (f(*face_vertices.shape) if need_grad_faces else f(0)), (f(*textures.shape) if need_grad_textures else f(0))


def _raster_setup(ctx, inputs, output):
    (fv, tex, image_size, background, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, modes, pool, need_p2f,
     want_visibility) = inputs
    image, p2f, aggrs, saved, vis = output
    ctx.save_for_backward(fv, tex, (saved if pool else image), aggrs)
    ctx.cfg = (image_size, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, modes, pool)


def _raster_backward(ctx, g_image, g_p2f, g_aggrs, g_saved, g_vis):
    # grad_p2f_info / grad_aggrs_info are ignored, as in the reference (functional/soft_rasterize.py:78): p2f is forward-only
    fv, tex, soft_colors, aggrs = ctx.saved_tensors
    need_gf, need_gt = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    if not (need_gf or need_gt):
        return (None,) * 15
    gf, gt = torch.ops.umr.soft_rasterize_backward(fv, tex, soft_colors, aggrs, g_image, *ctx.cfg, need_gf, need_gt)
    return (gf if need_gf else None, gt if need_gt else None) + (None,) * 13


soft_rasterize_op.register_autograd(_raster_backward, setup_context=_raster_setup)


# ------------------------------------------------------------------------------------------------ silhouette
@custom_op("umr::silhouette", mutates_args=(), device_types="cuda")
def silhouette_op(face_vertices: torch.Tensor, image_size: int, near: float, far: float, fill_back: bool, eps: float,
                  sigma_val: float, dist_eps: float, gamma_val: float, pool: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    L = _lib.lib()
    fv = _f32c(face_vertices)
    dev = fv.device
    N, F = fv.shape[:2]
    IS = int(image_size)
    alpha = torch.empty(N, IS, IS, device=dev, dtype=torch.float32)
    pooled = torch.empty(N, IS // 2, IS // 2, device=dev, dtype=torch.float32) if pool else None
    ws_bytes = L.umr_raster_workspace_bytes_for(N, F, int(image_size))
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    sc = _scalars(IS, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, 1)
    if getattr(_lib, "TAP", None) is not None:
        _lib.TAP("silhouette_forward", dict(face_vertices=fv, image_size=IS))
    rc = L.umr_raster_forward(ptr(fv), None, None, None, None, None, None, ptr(alpha), ptr(pooled), N, F, 1, *sc, 2 | 1, None,
                              ptr(ws), ws_bytes, _lib.stream_ptr(dev))
    _lib.check(rc, "umr_raster_forward(alpha only)")
    return (pooled if pool else alpha), (alpha if pool else alpha.new_empty(0))


@silhouette_op.register_fake
def _(face_vertices, image_size, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, pool):
    N, IS = face_vertices.shape[0], int(image_size)
    S = IS // 2 if pool else IS
    f = lambda *s: face_vertices.new_empty(s, dtype=torch.float32)
    return f(N, S, S), (f(N, IS, IS) if pool else f(0))


@custom_op("umr::silhouette_backward", mutates_args=(), device_types="cuda")
def silhouette_backward_op(face_vertices: torch.Tensor, alpha: torch.Tensor, grad_alpha: torch.Tensor, image_size: int,
                           near: float, far: float, fill_back: bool, eps: float, sigma_val: float, dist_eps: float,
                           gamma_val: float, pool: bool) -> torch.Tensor:
    L = _lib.lib()
    fv = _f32c(face_vertices)
    dev = fv.device
    N, F = fv.shape[:2]
    grad_faces = torch.zeros(N, F, 9, device=dev, dtype=torch.float32)
    g = grad_alpha.to(torch.float32).contiguous()
    ws_bytes = L.umr_raster_workspace_bytes_for(N, F, int(image_size))
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    sc = _scalars(image_size, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, 1)
    rc = L.umr_raster_backward(ptr(fv), None, ptr(alpha), None, None, ptr(grad_faces), None, ptr(g), 2 | (1 if pool else 0),
                               1, 0, N, F, 1, *sc, ptr(ws), ws_bytes, _lib.stream_ptr(dev))
    _lib.check(rc, "umr_raster_backward(alpha only)")
    return grad_faces.view(N, F, 3, 3)


@silhouette_backward_op.register_fake
def _(face_vertices, alpha, grad_alpha, image_size, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, pool):
    return face_vertices.new_empty(face_vertices.shape, dtype=torch.float32)


def _sil_setup(ctx, inputs, output):
    fv = inputs[0]
    out, saved = output
    ctx.save_for_backward(fv, saved if inputs[-1] else out)
    ctx.cfg = tuple(inputs[1:])


def _sil_backward(ctx, g_out, g_saved):
    fv, alpha = ctx.saved_tensors
    if not ctx.needs_input_grad[0]:
        return (None,) * 10
    return (torch.ops.umr.silhouette_backward(fv, alpha, g_out, *ctx.cfg),) + (None,) * 9


silhouette_op.register_autograd(_sil_backward, setup_context=_sil_setup)


# ------------------------------------------------------------------------- one render for the mask AND the texture term
# train_s1.py:199 / :217 and loss_utils.py:265 / :313 render the SAME meshes from the SAME cameras twice: once for the mask
# (only alpha is read; gradients go to vertices and cameras) and once textured with vertices and cameras DETACHED (gradients go
# to the texels only).  Alpha depends on neither textures nor lighting, and the two kernels compute it with the same arithmetic
# in the same face order, so the alpha channel of the textured render IS the mask render, bit for bit
# (tests/test_gpu_round2.py::test_raster_flags_and_fused_pool).  This operator is that one render with the reference's gradient
# routing: d(alpha) -> face_vertices through the silhouette backward, d(rgb) -> textures through the texel-gradient backward; the
# colour channels' dependence on the geometry is cut, exactly as `.detach()` cuts it in the reference.
@custom_op("umr::soft_rasterize_alpha_geometry", mutates_args=(), device_types="cuda")
def soft_rasterize_alpha_geometry_op(face_vertices: torch.Tensor, textures: torch.Tensor, image_size: int, background: List[float],
                                     near: float, far: float, fill_back: bool, eps: float, sigma_val: float, dist_eps: float,
                                     gamma_val: float, modes: int, pool: bool, need_p2f: bool, want_visibility: bool,
                                     lean: bool = False
                                     ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """lean: for callers that consume the pooled image, p2f and the visible-face ids only (the training steps): outputs 3 - 5 are
    then (packed saved state [N, 4 IS^2] -- opaque, the backward's input --, the call's raster workspace (uint8, opaque: the
    backward reads its face records instead of rebuilding them), face ids [N,IS,IS])."""
    if int(modes) != 1:
        raise RuntimeError("soft_rasterize_alpha_geometry: soft-max colour with UMR's own modes only")
    return _raster_forward(face_vertices, textures, image_size, background, near, far, fill_back, eps, sigma_val, dist_eps,
                           gamma_val, modes, pool, need_p2f, want_visibility, lean)


soft_rasterize_alpha_geometry_op.register_fake(_raster_fake)


@custom_op("umr::soft_rasterize_alpha_geometry_backward", mutates_args=(), device_types="cuda")
def soft_rasterize_alpha_geometry_backward_op(face_vertices: torch.Tensor, textures: torch.Tensor, soft_colors: torch.Tensor,
                                              aggrs_info: torch.Tensor, grad_image: torch.Tensor, image_size: int, near: float,
                                              far: float, fill_back: bool, eps: float, sigma_val: float, dist_eps: float,
                                              gamma_val: float, modes: int, pool: bool, lean: bool = False
                                              ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Both gradients of the shared render from ONE pass over its (pixel, face) pairs (UMR_BWD_ALPHA_GEOMETRY): what
    umr::silhouette_backward on the alpha plane and the texel-only umr::soft_rasterize_backward return."""
    L = _lib.lib()
    fv, tex = _f32c(face_vertices), _f32c(textures)
    dev = fv.device
    N, F = fv.shape[:2]
    TS = tex.shape[2]
    if TS > ONE_PASS_MAX_TS:
        raise RuntimeError("soft_rasterize_alpha_geometry_backward: %d texels per face; the one-pass kernel takes at most %d "
                           "(use the silhouette + texel-only backward pair)" % (TS, ONE_PASS_MAX_TS))
    G = N // tex.shape[0]
    grad_faces = torch.zeros(N, F, 9, device=dev, dtype=torch.float32)
    grad_textures = torch.zeros(N, F, TS, 3, device=dev, dtype=torch.float32)   # per view
    g = grad_image.to(torch.float32).contiguous()
    ws_bytes = L.umr_raster_workspace_bytes_for(N, F, int(image_size))
    # lean: `soft_colors` carries the forward's workspace (its face records and bounding boxes are read, not rebuilt)
    reuse = bool(lean) and soft_colors.dtype == torch.uint8 and soft_colors.numel() >= ws_bytes and soft_colors.is_contiguous()
    ws = soft_colors if reuse else torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    sc = _scalars(image_size, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, modes)
    if getattr(_lib, "TAP", None) is not None:
        _lib.TAP("raster_backward_alpha_geometry", dict(face_vertices=fv, textures=tex, grad_image=g, lean=lean, image_size=int(image_size)))
    # lean: aggrs_info is the forward's packed saved state, soft_colors an empty placeholder
    rc = L.umr_raster_backward(ptr(fv), ptr(tex), None if lean else ptr(soft_colors), None, ptr(aggrs_info), ptr(grad_faces),
                               ptr(grad_textures), ptr(g),
                               (1 if pool else 0) | BWD_ALPHA_GEOMETRY | (BWD_PACKED_STATE if lean else 0) |
                               (BWD_REUSE_WORKSPACE if reuse else 0) | (G << 8), 1, 1, N, F, TS,
                               *sc, ptr(ws), ws_bytes, _lib.stream_ptr(dev))
    _lib.check(rc, "umr_raster_backward(alpha geometry)")
    if G > 1:
        grad_textures = grad_textures.view(N // G, G, F, TS, 3).sum(1)   # autograd of the reference's repeat
    return grad_faces.view(N, F, 3, 3), grad_textures


@soft_rasterize_alpha_geometry_backward_op.register_fake
def _(face_vertices, textures, soft_colors, aggrs_info, grad_image, image_size, near, far, fill_back, eps, sigma_val, dist_eps,
      gamma_val, modes, pool, lean=False):
    return (face_vertices.new_empty(face_vertices.shape, dtype=torch.float32), textures.new_empty(textures.shape, dtype=torch.float32))


def _raster_ag_setup(ctx, inputs, output):
    _raster_setup(ctx, inputs[:15], output)
    ctx.lean = bool(inputs[15]) if len(inputs) > 15 else False


def _raster_ag_backward(ctx, g_image, g_p2f, g_aggrs, g_saved, g_vis):
    fv, tex, soft_colors, aggrs = ctx.saved_tensors
    image_size, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, modes, pool = ctx.cfg
    need_gf, need_gt = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    gf = gt = None
    if ctx.lean:
        # the packed state feeds the one-pass kernel only; a gradient nobody asked for is computed and dropped (the training
        # steps ask for both)
        if not (need_gf or need_gt):
            return (None,) * 16
        gf, gt = torch.ops.umr.soft_rasterize_alpha_geometry_backward(fv, tex, soft_colors, aggrs, g_image, *ctx.cfg, True)
        return (gf if need_gf else None, gt if need_gt else None) + (None,) * 14
    # ONE pass for both gradients where the face-major kernels run: their per-wave LDS texel accumulators hold TS <= 1023 texels
    # (beyond that umr_raster_backward takes its pixel-major route, which library 0.5 does not specialise for this flag)
    if need_gf and need_gt and tex.shape[2] <= ONE_PASS_MAX_TS:
        gf, gt = torch.ops.umr.soft_rasterize_alpha_geometry_backward(fv, tex, soft_colors, aggrs, g_image, *ctx.cfg)
        return (gf, gt) + (None,) * 14
    if need_gf:      # alpha -> geometry: the mask render's backward on this render's alpha plane
        gf = torch.ops.umr.silhouette_backward(fv, soft_colors[:, 3].contiguous(), g_image[:, 3].contiguous(), image_size, near, far,
                                               fill_back, eps, sigma_val, dist_eps, gamma_val, pool)
    if need_gt:      # rgb -> texels only
        _, gt = torch.ops.umr.soft_rasterize_backward(fv, tex, soft_colors, aggrs, g_image, *ctx.cfg, False, True)
    return (gf, gt) + (None,) * 14


soft_rasterize_alpha_geometry_op.register_autograd(_raster_ag_backward, setup_context=_raster_ag_setup)
