"""torch.autograd front-ends over the C ABI (include/umr_hip.h).

PyTorch is plumbing here: device memory (caching allocator), the current HIP stream and autograd
bookkeeping.  All arithmetic happens in libumr_hip.so; there is no CPU or eager-torch fallback.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import ptr

_FUNC_DIST = {'hard': 0, 'barycentric': 1, 'euclidean': 2}
_FUNC_RGB = {'hard': 0, 'softmax': 1}
_FUNC_ALPHA = {'hard': 0, 'sum': 1, 'prod': 2}
_FUNC_SAMPLE = {'surface': 0, 'vertex': 1}

_grid_cache = {}


def standard_grid(image_size, device):
    """The `grid` the reference builds every call (functional/soft_rasterize.py:57-62): identity
    affine_grid under torch-1.1.0 semantics (= align_corners=True).  Cached per (size, device)."""
    key = (int(image_size), str(device))
    g = _grid_cache.get(key)
    if g is None:
        theta = torch.tensor([[1, 0, 0], [0, 1, 0]], dtype=torch.float)
        g = torch.nn.functional.affine_grid(theta.unsqueeze(0), (1, 1, image_size, image_size), align_corners=True)
        g = g.view(image_size, image_size, 2).contiguous().to(device)
        _grid_cache[key] = g
    return g


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


class KernelCtx:
    """What the *Kernel classes below hand from forward() to backward(): the tensors to keep, which inputs want a gradient,
    and a few plain attributes.  The registered operators (umr_amd/ops_losses.py) build one per call -- forward of
    torch.ops.umr.<name>, backward of torch.ops.umr.<name>_backward -- from their own saved tensors; this is an ordinary class of
    this package, not torch's autograd context."""

    def __init__(self, needs=(), saved=(), **attrs):
        self.needs_input_grad = tuple(needs)
        self.saved_tensors = tuple(saved)
        self.__dict__.update(attrs)

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass


def _check_raster_shapes(face_vertices, textures):
    fv, tex = face_vertices, textures
    N, F = fv.shape[:2] if fv.dim() >= 2 else (0, 0)
    if fv.dim() != 4 or fv.shape[2:] != (3, 3) or tex.dim() != 4 or tex.shape[0] < 1 or N % tex.shape[0] \
            or N // tex.shape[0] > 65535 or tex.shape[1] != F or tex.shape[3] != 3:
        raise RuntimeError("soft_rasterize: face_vertices must be [N,F,3,3] and textures [N or N/G,F,TS,3]; got "
                           "%s and %s" % (tuple(fv.shape), tuple(tex.shape)))   # kernels index textures by (n//G, f)


class SoftRasterizeFunction:
    """Drop-in for external/SoftRas/soft_renderer/functional/soft_rasterize.py:9-108 (`.apply(...)` with the same
    positional arguments).  The autograd formula lives on the registered operator torch.ops.umr.soft_rasterize
    (umr_amd/ops.py); this class keeps the reference's call form.

    Extra (trailing) arguments drive the fused MI355X paths and default to the reference behaviour:
      pool:       also return/consume the 2x2 average-pooled image (anti-aliasing fused into the kernels;
                  the first output is then [N,4,IS/2,IS/2] and its gradient is consumed at that size)
      need_p2f:   False skips the p2f accumulators (returned as zeros)
      want_visibility: a 4th output [N,2,IS,IS] = the aggrs_info of the 'hard' render of the same faces (nearest depth, its
                  face id | -1), produced by the same kernel visits (soft-max colour only)
      detach_rgb_geometry: the colour channels see face_vertices DETACHED (their gradient reaches the textures only) while the
                  alpha channel keeps its gradient to face_vertices: ONE render where the reference renders the same views
                  twice, for the mask and -- with detached vertices -- for the texture term (torch.ops.umr.
                  soft_rasterize_alpha_geometry)
      lean_state: (with detach_rgb_geometry and pool) the caller consumes the pooled image, p2f and the visible-face ids ONLY:
                  returns (image, p2f, None[, face ids [N,IS,IS]]) -- no aggrs_info, and of the visibility planes the id plane alone.
                  The render then keeps its saved state as one packed buffer in the one-pass backward's layout and writes nothing
                  else at full resolution (UMR_RASTER_PACKED_STATE); where that form does not apply (image size not a multiple
                  of 8, > 1023 texels per face) the planar call runs and the same values come back
    """

    @staticmethod
    def apply(face_vertices, textures, image_size=256, background_color=[0, 0, 0], near=1, far=100,
              fill_back=True, eps=1e-3, sigma_val=1e-5, dist_func='euclidean', dist_eps=1e-4,
              gamma_val=1e-4, aggr_func_rgb='softmax', aggr_func_alpha='prod', texture_type='surface',
              pool=False, need_p2f=True, want_visibility=False, detach_rgb_geometry=False, lean_state=False):
        from . import ops  # noqa: F401  (registers torch.ops.umr.*)
        _check_raster_shapes(face_vertices, textures)
        modes = ops.pack_modes(_FUNC_RGB[aggr_func_rgb], _FUNC_DIST[dist_func], _FUNC_ALPHA[aggr_func_alpha],
                               _FUNC_SAMPLE[texture_type])
        if modes > 1 and pool:
            raise RuntimeError("soft_rasterize: the fused 2x2 pool exists for UMR's own modes only (dist_func='euclidean', "
                               "aggr_func_alpha='prod', texture_type='surface')")
        if _FUNC_SAMPLE[texture_type] == 1 and textures.shape[2] != 3:
            raise RuntimeError("soft_rasterize: texture_type='vertex' needs textures [N,F,3,3]; got %s" % (tuple(textures.shape),))
        if not _lib.on_device(face_vertices):
            raise RuntimeError("umr_amd: expected a GPU tensor, got %s (no CPU path exists)" % face_vertices.device)
        if want_visibility and modes != 1:
            raise RuntimeError("soft_rasterize: want_visibility needs aggr_func_rgb='softmax' with UMR's own modes")
        if detach_rgb_geometry and modes != 1:
            raise RuntimeError("soft_rasterize: detach_rgb_geometry needs aggr_func_rgb='softmax' with UMR's own modes")
        args = (face_vertices, textures, int(image_size), [float(c) for c in background_color], float(near), float(far),
                bool(fill_back), float(eps), float(sigma_val), float(dist_eps), float(gamma_val), modes, bool(pool), bool(need_p2f),
                bool(want_visibility))
        if lean_state:
            if not (detach_rgb_geometry and pool):
                raise RuntimeError("soft_rasterize: lean_state goes with detach_rgb_geometry and pool")
            packed = ops.lean_state_ok(image_size, textures.shape[2], modes, pool)
            image, p2f, _, _, vis = torch.ops.umr.soft_rasterize_alpha_geometry(*args, packed)
            if want_visibility and not packed:
                vis = vis[:, 1]
            return (image, p2f, None, vis) if want_visibility else (image, p2f, None)
        if detach_rgb_geometry:
            image, p2f, aggrs, _, vis = torch.ops.umr.soft_rasterize_alpha_geometry(*args, False)
        else:
            image, p2f, aggrs, _, vis = torch.ops.umr.soft_rasterize(*args)
        return (image, p2f, aggrs, vis) if want_visibility else (image, p2f, aggrs)


class SilhouetteFunction:
    """Alpha channel of the soft render only (UMR_RASTER_ALPHA_ONLY): face_vertices [N,F,3,3] -> alpha [N,S,S]
    with S = image_size (or image_size/2 when `pool`).  Bit-identical to channel 3 of SoftRasterizeFunction; the
    backward is the reference's with a zero rgb gradient (mask / GAN-view renders).
    Registered operator: torch.ops.umr.silhouette."""

    @staticmethod
    def apply(face_vertices, image_size, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, pool):
        from . import ops  # noqa: F401
        if not _lib.on_device(face_vertices):
            raise RuntimeError("umr_amd: expected a GPU tensor, got %s (no CPU path exists)" % face_vertices.device)
        out, _ = torch.ops.umr.silhouette(face_vertices, int(image_size), float(near), float(far), bool(fill_back), float(eps),
                                          float(sigma_val), float(dist_eps), float(gamma_val), bool(pool))
        return out


def visibility(face_vertices, image_size, near=1., far=100., fill_back=True, eps=1e-3, sigma_val=1e-5, dist_eps=1e-10,
               gamma_val=1e-4):
    """Hard z-buffer planes only (UMR_RASTER_FACE_ID_ONLY): -> aggrs_info [N,2,IS,IS] = (nearest depth, face id | -1),
    bit-identical to the third output of soft_rasterize(..., aggr_func_rgb='hard').  Forward only (no gradient flows
    through these planes in the reference either)."""
    L = _lib.lib()
    fv = _f32c(face_vertices)
    dev = fv.device
    N, F = fv.shape[:2]
    IS = int(image_size)
    aggrs = torch.empty(N, 2, IS, IS, device=dev, dtype=torch.float32)
    ws_bytes = L.umr_raster_workspace_bytes_for(N, F, int(image_size))
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    rc = L.umr_raster_forward(ptr(fv), None, None, ptr(aggrs), None, None, None, None, None, N, F, 1, IS, float(near),
                              float(far), float(eps), float(sigma_val), 2, float(math.log(1. / dist_eps - 1.)),
                              float(gamma_val), 0, 2, 0, 1 if fill_back else 0, 4 | 1, None, ptr(ws), ws_bytes,
                              _lib.stream_ptr(dev))
    _lib.check(rc, "umr_raster_forward(visibility only)")
    return aggrs


def silhouette(face_vertices, image_size, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, pool):
    """Alpha channel of the soft render only -> [N,S,S]; see SilhouetteFunction."""
    return SilhouetteFunction.apply(face_vertices, image_size, near, far, fill_back, eps, sigma_val, dist_eps, gamma_val, pool)


def soft_rasterize(face_vertices, textures, image_size=256, background_color=[0, 0, 0], near=1, far=100,
                   fill_back=True, eps=1e-3, sigma_val=1e-5, dist_func='euclidean', dist_eps=1e-4,
                   gamma_val=1e-4, aggr_func_rgb='softmax', aggr_func_alpha='prod', texture_type='surface',
                   pool=False, need_p2f=True, want_visibility=False, detach_rgb_geometry=False, lean_state=False):
    """Same signature and return as soft_renderer.functional.soft_rasterize
    (functional/soft_rasterize.py:111-125): (soft_colors [N,4,IS,IS], p2f_info [N,F,2], aggrs_info)."""
    if not _lib.on_device(face_vertices):
        # the reference's guard (:117-118) is dead code; ours is real
        raise TypeError('Rasterize module supports only GPU (ROCm) tensors')
    return SoftRasterizeFunction.apply(face_vertices, textures, image_size, background_color, near, far,
                                       fill_back, eps, sigma_val, dist_func, dist_eps, gamma_val,
                                       aggr_func_rgb, aggr_func_alpha, texture_type, pool, need_p2f, want_visibility,
                                       detach_rgb_geometry, lean_state)


class ProjectFacesKernel:
    """verts [N/G,V,3], cams [N,7], faces [N/G,F,3] int32 -> (face_pre [N,F,3,3], face_out [N,F,3,3], light [N,F,3]).
    Fuses geom_utils.orthographic_proj_withz + smr.Render's y flip + face_vertices + LookAt/orthogonal and, when
    `light` = (ambient, directional, color3, direction3) is given, sr.Lighting's per-face surface light
    (lighting.py:50-57); face_pre / light come back empty when not asked for.
    G = N_cams / N_meshes >= 1 camera hypotheses per mesh: view n renders mesh n // G, i.e. the layout of
    `vs.unsqueeze(1).repeat(1, K, 1, 1).view(B*K, V, 3)` (nnutils/loss_utils.py:260-262) without the copies."""

    @staticmethod
    def forward(ctx, verts, cams, faces_idx, offset_z, eye_z, want_pre, light=None):
        L = _lib.lib()
        dev = verts.device
        v, c = _f32c(verts), _f32c(cams)
        M, V = v.shape[:2]
        N = c.shape[0]
        F = faces_idx.shape[1]
        if N % M or faces_idx.shape[0] != M:
            raise RuntimeError("project_faces: %d cameras for %d meshes / %d face sets" % (N, M, faces_idx.shape[0]))
        G = N // M
        face_out = torch.empty(N, F, 3, 3, device=dev, dtype=torch.float32)
        face_pre = torch.empty(N, F, 3, 3, device=dev, dtype=torch.float32) if want_pre else None
        light_out = torch.empty(N, F, 3, device=dev, dtype=torch.float32) if light is not None else None
        amb, dirn, col, dvec = light if light is not None else (0., 0., (1., 1., 1.), (0., 1., 0.))
        ctx.light = (float(dirn), (ctypes.c_float * 3)(*[float(x) for x in col]),
                     (ctypes.c_float * 3)(*[float(x) for x in dvec])) if light is not None else None
        rc = L.umr_project_faces_lit_forward(ptr(v), ptr(c), ptr(faces_idx), ptr(face_pre), ptr(face_out), ptr(light_out),
                                             N, V, F, float(offset_z), float(eye_z), G, float(amb), float(dirn),
                                             (ctypes.c_float * 3)(*[float(x) for x in col]),
                                             (ctypes.c_float * 3)(*[float(x) for x in dvec]), _lib.stream_ptr(dev))
        _lib.check(rc, "umr_project_faces_lit_forward")
        if light is not None:
            ctx.save_for_backward(v, c, faces_idx, face_out)
        else:
            ctx.save_for_backward(v, c, faces_idx)
        ctx.want_pre = want_pre
        empty = face_out.new_empty(0)
        return (face_pre if want_pre else empty), face_out, (light_out if light is not None else empty)

    @staticmethod
    def backward(ctx, g_pre, g_out, g_light):
        L = _lib.lib()
        v, c, faces_idx = ctx.saved_tensors[:3]
        face_out = ctx.saved_tensors[3] if ctx.light is not None else None
        dev = v.device
        M, V = v.shape[:2]
        N = c.shape[0]
        G = N // M
        F = faces_idx.shape[1]
        g_out = (g_out if g_out is not None else torch.zeros(N, F, 3, 3, device=dev)).to(torch.float32).contiguous()
        g_pre = g_pre.to(torch.float32).contiguous() if (ctx.want_pre and g_pre is not None) else None
        g_light = g_light.to(torch.float32).contiguous() if (ctx.light is not None and g_light is not None) else None
        grad_verts = torch.zeros(N, V, 3, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        grad_cams = torch.empty_like(c)
        ws_bytes = L.umr_project_workspace_bytes(N, V)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        dirn, col, dvec = ctx.light if ctx.light is not None else (0., None, None)
        rc = L.umr_project_faces_lit_backward(ptr(g_out), ptr(g_pre), ptr(g_light), ptr(face_out), ptr(v), ptr(c),
                                              ptr(faces_idx), ptr(grad_verts), ptr(grad_cams), N, V, F, G, dirn, col, dvec,
                                              ptr(ws), ws_bytes, _lib.stream_ptr(dev))
        _lib.check(rc, "umr_project_faces_lit_backward")
        if grad_verts is not None and G > 1:
            grad_verts = grad_verts.view(M, G, V, 3).sum(1)      # what autograd does for the reference's repeat
        return grad_verts, (grad_cams if ctx.needs_input_grad[1] else None), None, None, None, None, None


class ProjectPointsKernel:
    """verts [N,V,3], cams [N,7] -> [N,V,out_dim] (2: xy; 3: xy + z with offset_z).  No y flip."""

    @staticmethod
    def forward(ctx, verts, cams, out_dim=2, offset_z=0.0):
        L = _lib.lib()
        v, c = _f32c(verts), _f32c(cams)
        N, V = v.shape[:2]
        out = torch.empty(N, V, out_dim, device=v.device, dtype=torch.float32)
        _lib.check(L.umr_project_points_forward(ptr(v), ptr(c), ptr(out), N, V, int(out_dim), float(offset_z),
                                                _lib.stream_ptr(v.device)), "umr_project_points_forward")
        ctx.save_for_backward(v, c)
        ctx.out_dim = int(out_dim)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        v, c = ctx.saved_tensors
        N, V = v.shape[:2]
        g = g.to(torch.float32).contiguous()
        grad_verts = torch.zeros_like(v) if ctx.needs_input_grad[0] else None
        grad_cams = torch.empty_like(c)
        _lib.check(L.umr_project_points_backward(ptr(g), ptr(v), ptr(c), ptr(grad_verts), ptr(grad_cams), N, V,
                                                 ctx.out_dim, _lib.stream_ptr(v.device)),
                   "umr_project_points_backward")
        return grad_verts, (grad_cams if ctx.needs_input_grad[1] else None), None, None


class NegIoUKernel:
    """loss[n] = 1 - sum(p t) / (sum(p + t - p t) + 1e-6)   (nnutils/loss_utils.py:41-48, avg=False)."""

    @staticmethod
    def forward(ctx, predict, target):
        L = _lib.lib()
        p = _f32c(predict).view(predict.shape[0], -1)
        t = _f32c(target).view(target.shape[0], -1)
        N, P = p.shape
        loss = torch.empty(N, device=p.device, dtype=torch.float32)
        sums = torch.empty(N, L.umr_neg_iou_sums_stride(P), device=p.device, dtype=torch.float32)
        _lib.check(L.umr_neg_iou_forward(ptr(p), P, ptr(t), ptr(loss), ptr(sums), sums.numel() * 4, N, P, _lib.stream_ptr(p.device)),
                   "umr_neg_iou_forward")
        ctx.save_for_backward(p, t, sums)
        ctx.shape = predict.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        p, t, sums = ctx.saved_tensors
        N, P = p.shape
        gp = torch.zeros_like(p)
        g = g.to(torch.float32).contiguous()
        _lib.check(L.umr_neg_iou_backward(ptr(p), P, ptr(t), ptr(sums), ptr(g), ptr(gp), P, N, P,
                                          _lib.stream_ptr(p.device)), "umr_neg_iou_backward")
        return gp.view(ctx.shape), None


class ChamferKernel:
    @staticmethod
    def forward(ctx, a, b):
        L = _lib.lib()
        a_, b_ = _f32c(a), _f32c(b)
        B, n, D = a_.shape
        m = b_.shape[1]
        dev = a_.device
        d1 = torch.empty(B, n, device=dev, dtype=torch.float32)
        d2 = torch.empty(B, m, device=dev, dtype=torch.float32)
        i1 = torch.empty(B, n, device=dev, dtype=torch.int32)
        i2 = torch.empty(B, m, device=dev, dtype=torch.int32)
        _lib.check(L.umr_chamfer_forward(ptr(a_), ptr(b_), ptr(d1), ptr(d2), ptr(i1), ptr(i2), B, n, m, D,
                                         _lib.stream_ptr(dev)), "umr_chamfer_forward")
        ctx.save_for_backward(a_, b_, i1, i2)
        ctx.mark_non_differentiable(i1, i2)
        return d1, d2, i1, i2

    @staticmethod
    def backward(ctx, g1, g2, _gi1, _gi2):
        L = _lib.lib()
        a_, b_, i1, i2 = ctx.saved_tensors
        B, n, D = a_.shape
        m = b_.shape[1]
        g1 = (g1 if g1 is not None else torch.zeros(B, n, device=a_.device)).to(torch.float32).contiguous()
        g2 = (g2 if g2 is not None else torch.zeros(B, m, device=a_.device)).to(torch.float32).contiguous()
        ga, gb = torch.empty_like(a_), torch.empty_like(b_)
        _lib.check(L.umr_chamfer_backward(ptr(a_), ptr(b_), ptr(i1), ptr(i2), ptr(g1), ptr(g2), ptr(ga), ptr(gb),
                                          B, n, m, D, _lib.stream_ptr(a_.device)), "umr_chamfer_backward")
        return ga, gb


class GridSampleCLKernel:
    """image [B,C,H,W], grid [B,P,2] -> out [B,P,C]; bilinear / zeros / align_corners=True."""

    @staticmethod
    def forward(ctx, image, grid):
        L = _lib.lib()
        img, g = _f32c(image), _f32c(grid)
        B, C, H, W = img.shape
        P = g.shape[1]
        out = torch.empty(B, P, C, device=img.device, dtype=torch.float32)
        _lib.check(L.umr_grid_sample_forward(ptr(img), ptr(g), ptr(out), B, C, H, W, P, _lib.stream_ptr(img.device)),
                   "umr_grid_sample_forward")
        ctx.save_for_backward(img, g)
        return out

    @staticmethod
    def backward(ctx, go):
        L = _lib.lib()
        img, g = ctx.saved_tensors
        B, C, H, W = img.shape
        P = g.shape[1]
        go = go.to(torch.float32).contiguous()
        gi = torch.zeros_like(img) if ctx.needs_input_grad[0] else None
        gg = torch.empty_like(g) if ctx.needs_input_grad[1] else None
        _lib.check(L.umr_grid_sample_backward(ptr(img), ptr(g), ptr(go), ptr(gg), ptr(gi), B, C, H, W, P,
                                              _lib.stream_ptr(img.device)), "umr_grid_sample_backward")
        return gi, gg


class LaplacianKernel:
    @staticmethod
    def forward(ctx, x, nbr_off, nbr_idx):
        L = _lib.lib()
        x_ = _f32c(x)
        B, V = x_.shape[:2]
        lap = torch.empty_like(x_)
        loss = torch.empty(B, device=x_.device, dtype=torch.float32)
        _lib.check(L.umr_laplacian_forward(ptr(x_), ptr(nbr_off), ptr(nbr_idx), ptr(lap), ptr(loss), B, V,
                                           _lib.stream_ptr(x_.device)), "umr_laplacian_forward")
        ctx.save_for_backward(lap, nbr_off, nbr_idx)
        return loss

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        lap, nbr_off, nbr_idx = ctx.saved_tensors
        B, V = lap.shape[:2]
        gx = torch.zeros_like(lap)
        g = g.to(torch.float32).contiguous()
        _lib.check(L.umr_laplacian_backward(ptr(lap), ptr(nbr_off), ptr(nbr_idx), ptr(g), ptr(gx), B, V,
                                            _lib.stream_ptr(lap.device)), "umr_laplacian_backward")
        return gx, None, None


class FlattenKernel:
    @staticmethod
    def forward(ctx, x, quads):
        L = _lib.lib()
        x_ = _f32c(x)
        B, V = x_.shape[:2]
        E = quads.shape[0]
        loss = torch.empty(B, device=x_.device, dtype=torch.float32)
        _lib.check(L.umr_flatten_forward(ptr(x_), ptr(quads), ptr(loss), B, V, E, _lib.stream_ptr(x_.device)),
                   "umr_flatten_forward")
        ctx.save_for_backward(x_, quads)
        return loss

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        x_, quads = ctx.saved_tensors
        B, V = x_.shape[:2]
        gx = torch.zeros_like(x_)
        g = g.to(torch.float32).contiguous()
        _lib.check(L.umr_flatten_backward(ptr(x_), ptr(quads), ptr(g), ptr(gx), B, V, quads.shape[0],
                                          _lib.stream_ptr(x_.device)), "umr_flatten_backward")
        return gx, None


def visible_face_mask(face_ids, num_faces):
    """face_ids [B,P] float (hard renderer face-id plane) -> [B,F] 0/1 mask (loss_utils.py:173-179)."""
    L = _lib.lib()
    ids = _f32c(face_ids)
    B, P = ids.shape
    mask = torch.zeros(B, num_faces, device=ids.device, dtype=torch.float32)
    _lib.check(L.umr_visible_face_mask(ptr(ids), ptr(mask), B, P, num_faces, _lib.stream_ptr(ids.device)),
               "umr_visible_face_mask")
    return mask


class Upsample2xBilinearKernel:
    """[B,C,H,W] -> [B,C,2H,2W], == F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)."""

    @staticmethod
    def forward(ctx, x):
        L = _lib.lib()
        x_ = _f32c(x)
        B, C, H, W = x_.shape
        out = torch.empty(B, C, 2 * H, 2 * W, device=x_.device, dtype=torch.float32)
        _lib.check(L.umr_upsample2x_bilinear_forward(ptr(x_), ptr(out), B * C, H, W, _lib.stream_ptr(x_.device)),
                   "umr_upsample2x_bilinear_forward")
        ctx.shape = (B, C, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        B, C, H, W = ctx.shape
        g = g.to(torch.float32).contiguous()
        gi = torch.empty(B, C, H, W, device=g.device, dtype=torch.float32)
        _lib.check(L.umr_upsample2x_bilinear_backward(ptr(g), ptr(gi), B * C, H, W, _lib.stream_ptr(g.device)),
                   "umr_upsample2x_bilinear_backward")
        return gi


class PerceptualPrologueKernel:
    """img [B,C<=3,H,W], mask [B,H,W] -> ((2 (img * mask) - 1) - shift_c) / scale_c: the input side of the perceptual texture
    term (loss_utils.py:141-146, perceptual_loss.py:52-54, networks_basic.py:45-46) in one launch each way instead of
    five element-wise kernels forward and as many backward.  shift / scale: 3 python floats each."""

    @staticmethod
    def forward(ctx, img, mask, shift, scale):
        L = _lib.lib()
        x, m = _f32c(img), _f32c(mask)
        B, C, H, W = x.shape
        if C > 3 or tuple(m.shape) not in ((B, H, W), (B, 1, H, W)):   # the kernel reads mask[b * H * W + p]: no broadcasting
            raise ValueError("perceptual prologue: img %s needs C <= 3 and a mask of shape [B,H,W], got %s"
                             % (tuple(x.shape), tuple(m.shape)))
        out = torch.empty_like(x)
        sh, sc = (ctypes.c_float * 3)(*[float(v) for v in shift]), (ctypes.c_float * 3)(*[float(v) for v in scale])
        _lib.check(L.umr_perceptual_prologue_forward(ptr(x), ptr(m), ptr(out), B, C, H * W, sh, sc, _lib.stream_ptr(x.device)),
                   "umr_perceptual_prologue_forward")
        ctx.save_for_backward(x, m)
        ctx.scale = [float(v) for v in scale]
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        x, m = ctx.saved_tensors
        B, C, H, W = x.shape
        g = g.to(torch.float32).contiguous()
        gi = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gm = torch.empty_like(m) if ctx.needs_input_grad[1] else None
        sc = (ctypes.c_float * 3)(*ctx.scale)
        _lib.check(L.umr_perceptual_prologue_backward(ptr(g), ptr(x), ptr(m), ptr(gi), ptr(gm), B, C, H * W, sc,
                                                      _lib.stream_ptr(x.device)), "umr_perceptual_prologue_backward")
        return gi, gm, None, None


class CosSimDistanceKernel:
    """PNet head (networks_basic.py:42-64 + util/util.py:71-83): apply(eps, *feats0, *feats1) with 2 T feature maps
    [N,C_t,X_t,Y_t] -> val [N] = sum_t (1 - mean_xy cos(f0_t, f1_t)).  One launch for all taps each way."""

    @staticmethod
    def forward(ctx, eps, *feats):
        L = _lib.lib()
        T = len(feats) // 2
        f0 = [_f32c(f) for f in feats[:T]]
        f1 = [_f32c(f) for f in feats[T:]]
        N = f0[0].shape[0]
        for a, b in zip(f0, f1):
            if a.shape != b.shape or a.dim() != 4 or a.shape[0] != N:
                raise RuntimeError("cos_sim_distance: feature pairs must be [N,C,X,Y] of equal shapes")
        dev = f0[0].device
        C = (ctypes.c_int * T)(*[a.shape[1] for a in f0])
        P = (ctypes.c_int * T)(*[a.shape[2] * a.shape[3] for a in f0])
        ws_bytes = L.umr_cos_sim_workspace_bytes(T, N, P)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        val = torch.empty(N, device=dev, dtype=torch.float32)
        p0 = (ctypes.c_void_p * T)(*[ptr(a) for a in f0])
        p1 = (ctypes.c_void_p * T)(*[ptr(a) for a in f1])
        _lib.check(L.umr_cos_sim_forward(T, p0, p1, C, P, N, float(eps), ptr(val), ptr(ws), ws_bytes,
                                         _lib.stream_ptr(dev)), "umr_cos_sim_forward")
        ctx.save_for_backward(ws, *f0, *f1)
        ctx.eps, ctx.T = float(eps), T
        return val

    @staticmethod
    def backward(ctx, gval):
        L = _lib.lib()
        T = ctx.T
        ws, feats = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        f0, f1 = feats[:T], feats[T:]
        N = f0[0].shape[0]
        dev = ws.device
        need = ctx.needs_input_grad[1:]
        g0 = [torch.empty_like(a) if need[t] else None for t, a in enumerate(f0)]
        g1 = [torch.empty_like(a) if need[T + t] else None for t, a in enumerate(f1)]
        if not any(need):
            return (None,) * (1 + 2 * T)
        C = (ctypes.c_int * T)(*[a.shape[1] for a in f0])
        P = (ctypes.c_int * T)(*[a.shape[2] * a.shape[3] for a in f0])
        tab = lambda ts: (ctypes.c_void_p * T)(*[ptr(t) for t in ts])
        gv = gval.to(torch.float32).contiguous()
        _lib.check(L.umr_cos_sim_backward(T, tab(f0), tab(f1), tab(g0), tab(g1), C, P, N, ctx.eps, ptr(gv), ptr(ws),
                                          ws.numel(), _lib.stream_ptr(dev)), "umr_cos_sim_backward")
        return (None,) + tuple(g0) + tuple(g1)


class PartMatchKernel:
    """Reductions of part_matching_loss (nnutils/loss_utils.py:399-440, scops_utils.py:12-54):
    apply(render_a [B,4,H,W], render_b [B,4,H,W], part_segs [B,5,H,W], weights5 (python floats), background, eps)
    -> (l_eqv [B], l_lm [B]); see include/umr_hip.h for the exact definition.  Gradients flow to the two renders."""

    @staticmethod
    def forward(ctx, render_a, render_b, part_segs, weights5, background, center_eps):
        L = _lib.lib()
        a, b, q = _f32c(render_a), _f32c(render_b), _f32c(part_segs)
        B, _, H, W = a.shape
        if a.shape != b.shape or a.shape[1] != 4 or q.shape != (B, 5, H, W):
            raise RuntimeError("part_match: renders must be [B,4,H,W] and part_segs [B,5,H,W]")
        dev = a.device
        w = (ctypes.c_float * 5)(*[float(x) for x in weights5])
        ws_bytes = L.umr_part_match_workspace_bytes(B, H, W)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        out = torch.empty(2, B, device=dev, dtype=torch.float32)
        _lib.check(L.umr_part_match_forward(ptr(a), ptr(b), ptr(q), B, H, W, w, float(background), float(center_eps),
                                            ptr(out[0]), ptr(out[1]), ptr(ws), ws_bytes, _lib.stream_ptr(dev)),
                   "umr_part_match_forward")
        ctx.save_for_backward(a, b, q, ws)
        ctx.cfg = ([float(x) for x in weights5], float(background), float(center_eps))
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_eqv, g_lm):
        L = _lib.lib()
        a, b, q, ws = ctx.saved_tensors
        B, _, H, W = a.shape
        dev = a.device
        w5, bg, eps = ctx.cfg
        w = (ctypes.c_float * 5)(*w5)
        ge = (g_eqv if g_eqv is not None else torch.zeros(B, device=dev)).to(torch.float32).contiguous()
        gl = (g_lm if g_lm is not None else torch.zeros(B, device=dev)).to(torch.float32).contiguous()
        ga, gb = torch.zeros_like(a), torch.zeros_like(b)
        _lib.check(L.umr_part_match_backward(ptr(a), ptr(b), ptr(q), B, H, W, w, bg, eps, ptr(ge), ptr(gl), ptr(ga), ptr(gb),
                                             ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "umr_part_match_backward")
        return ga, gb, None, None, None, None


class RowNormMeanKernel:
    """mean over rows of ||x_row||_2 (deform_l2reg, nnutils/loss_utils.py:118-123): x [..., W] -> scalar."""

    @staticmethod
    def forward(ctx, x):
        L = _lib.lib()
        x2 = _f32c(x).view(-1, x.shape[-1])
        rows, width = x2.shape
        out = torch.empty((), device=x2.device, dtype=torch.float32)
        scratch = torch.empty(L.umr_reg_scratch_floats(rows, 1), device=x2.device, dtype=torch.float32)
        _lib.check(L.umr_row_norm_mean_forward(ptr(x2), ptr(out), ptr(scratch), scratch.numel() * 4, rows, width,
                                               _lib.stream_ptr(x2.device)), "umr_row_norm_mean_forward")
        ctx.save_for_backward(x2)
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        x2, = ctx.saved_tensors
        gx = torch.empty_like(x2)
        g = g.to(torch.float32).contiguous()
        _lib.check(L.umr_row_norm_mean_backward(ptr(x2), ptr(g), ptr(gx), x2.shape[0], x2.shape[1], _lib.stream_ptr(x2.device)),
                   "umr_row_norm_mean_backward")
        return gx.view(ctx.shape)


class AbsColumnMeanKernel:
    """mean |x[..., column]| (sym_reg, nnutils/loss_utils.py:125-126: column 1 of verts [B,V,3]) -> scalar."""

    @staticmethod
    def forward(ctx, x, column):
        L = _lib.lib()
        x2 = _f32c(x).view(-1, x.shape[-1])
        rows, width = x2.shape
        out = torch.empty((), device=x2.device, dtype=torch.float32)
        scratch = torch.empty(L.umr_reg_scratch_floats(rows, 1), device=x2.device, dtype=torch.float32)
        _lib.check(L.umr_abs_mean_forward(ptr(x2), ptr(out), ptr(scratch), scratch.numel() * 4, rows, width, int(column),
                                          _lib.stream_ptr(x2.device)), "umr_abs_mean_forward")
        ctx.save_for_backward(x2)
        ctx.shape, ctx.column = x.shape, int(column)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        x2, = ctx.saved_tensors
        gx = torch.empty_like(x2)
        g = g.to(torch.float32).contiguous()
        _lib.check(L.umr_abs_mean_backward(ptr(x2), ptr(g), ptr(gx), x2.shape[0], x2.shape[1], ctx.column,
                                           _lib.stream_ptr(x2.device)), "umr_abs_mean_backward")
        return gx.view(ctx.shape), None


class MaskedL1Kernel:
    """per_sample[b] = mean_{c,p} |img_pred mask_pred - img_gt mask_gt| (texture_loss_masks, nnutils/loss_utils.py:103-116).
    img_* [B,C,H,W], mask_* [B,H,W]; gradients to img_pred and mask_pred."""

    @staticmethod
    def forward(ctx, img_pred, img_gt, mask_gt, mask_pred):
        L = _lib.lib()
        ip, ig, mg, mp = _f32c(img_pred), _f32c(img_gt), _f32c(mask_gt), _f32c(mask_pred)
        B, C, H, W = ip.shape
        if ig.shape != ip.shape or mg.numel() != B * H * W or mp.numel() != B * H * W:
            raise ValueError("masked L1: img %s / %s need masks [B,H,W], got %s / %s"
                             % (tuple(ip.shape), tuple(ig.shape), tuple(mg.shape), tuple(mp.shape)))
        per = torch.empty(B, device=ip.device, dtype=torch.float32)
        scratch = torch.empty(L.umr_reg_scratch_floats(C * H * W, B), device=ip.device, dtype=torch.float32)
        _lib.check(L.umr_masked_l1_forward(ptr(ip), ptr(ig), ptr(mg), ptr(mp), ptr(per), ptr(scratch), scratch.numel() * 4, B, C,
                                           H * W, _lib.stream_ptr(ip.device)), "umr_masked_l1_forward")
        ctx.save_for_backward(ip, ig, mg, mp)
        ctx.mp_shape = mask_pred.shape
        return per

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        ip, ig, mg, mp = ctx.saved_tensors
        B, C, H, W = ip.shape
        g = g.to(torch.float32).contiguous()
        gip = torch.empty_like(ip) if ctx.needs_input_grad[0] else None
        gmp = torch.empty_like(mp) if ctx.needs_input_grad[3] else None
        _lib.check(L.umr_masked_l1_backward(ptr(ip), ptr(ig), ptr(mg), ptr(mp), ptr(g), ptr(gip), ptr(gmp), B, C, H * W,
                                            _lib.stream_ptr(ip.device)), "umr_masked_l1_backward")
        return gip, None, None, (gmp.view(ctx.mp_shape) if gmp is not None else None)


# ------------------------------------------------------------------------------------------------------------------------
# The ONE route into the kernels above: the registered operators torch.ops.umr.* (umr_amd/ops_losses.py; the rasterizer's are in
# umr_amd/ops.py).  The functions below are what the loss / geometry modules call; the *Function classes keep the
# `Function.apply(...)` call form of rounds 1-3 for tests and tools and are thin forwards to the same operators.
def _ops(*tensors):
    """torch.ops.umr, after refusing host tensors with this package's own message (the dispatcher's "no kernel for the CPU
    backend" says the same less readably): there is no CPU path."""
    from . import ops_losses  # noqa: F401  (registers torch.ops.umr.*)
    for t in tensors:
        if torch.is_tensor(t) and not _lib.on_device(t):
            raise RuntimeError("umr_amd: expected a GPU tensor, got %s (no CPU path exists)" % t.device)
    return torch.ops.umr


def _light_list(light):
    if light is None:
        return []
    amb, dirn, col, dvec = light
    return [float(amb), float(dirn)] + [float(x) for x in col] + [float(x) for x in dvec]


def project_faces(verts, cams, faces_idx, offset_z, eye_z, want_pre=False, light=None):
    """-> (face_pre [N,F,3,3] | empty, face_out [N,F,3,3], light [N,F,3] | empty); see ProjectFacesKernel."""
    return _ops(verts, cams).project_faces_lit(verts, cams, faces_idx, float(offset_z), float(eye_z), bool(want_pre), _light_list(light))


def project_points(verts, cams, out_dim=2, offset_z=0.0):
    return _ops(verts, cams).project_points(verts, cams, int(out_dim), float(offset_z))


def neg_iou(predict, target):
    return _ops(predict, target).neg_iou(predict, target)[0]


def chamfer(a, b):
    return _ops(a, b).chamfer(a, b)


def grid_sample_cl(image, grid):
    return _ops(image, grid).grid_sample_cl(image, grid)


def laplacian(x, nbr_off, nbr_idx):
    return _ops(x).laplacian(x, nbr_off, nbr_idx)[0]


def flatten(x, quads):
    return _ops(x).flatten(x, quads)


def upsample2x_bilinear(x):
    return _ops(x).upsample2x_bilinear(x)


def perceptual_prologue(img, mask, shift, scale):
    return _ops(img, mask).perceptual_prologue(img, mask, [float(v) for v in shift], [float(v) for v in scale])


def cos_sim_distance(eps, feats0, feats1):
    return _ops(*feats0, *feats1).cos_sim(list(feats0), list(feats1), float(eps))[0]


def part_match(render_a, render_b, part_segs, weights5, background, center_eps):
    e, l, _ = _ops(render_a, render_b, part_segs).part_match(render_a, render_b, part_segs, [float(w) for w in weights5], float(background), float(center_eps))
    return e, l


def row_norm_mean(x):
    return _ops(x).row_norm_mean(x)


def abs_column_mean(x, column):
    return _ops(x).abs_column_mean(x, int(column))


def masked_l1(img_pred, img_gt, mask_gt, mask_pred):
    return _ops(img_pred, img_gt, mask_gt, mask_pred).masked_l1(img_pred, img_gt, mask_gt, mask_pred)


class _Apply:
    def __init__(self, fn):
        self.apply = fn


ProjectFacesFunction = _Apply(project_faces)
ProjectPointsFunction = _Apply(project_points)
NegIoUFunction = _Apply(neg_iou)
ChamferFunction = _Apply(chamfer)
GridSampleCLFunction = _Apply(grid_sample_cl)
LaplacianFunction = _Apply(laplacian)
FlattenFunction = _Apply(flatten)
Upsample2xBilinearFunction = _Apply(upsample2x_bilinear)
PerceptualPrologueFunction = _Apply(perceptual_prologue)
CosSimDistanceFunction = _Apply(lambda eps, *feats: cos_sim_distance(eps, feats[:len(feats) // 2], feats[len(feats) // 2:]))
PartMatchFunction = _Apply(part_match)
RowNormMeanFunction = _Apply(row_norm_mean)
AbsColumnMeanFunction = _Apply(abs_column_mean)
MaskedL1Function = _Apply(masked_l1)
