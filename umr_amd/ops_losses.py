"""The remaining C-ABI entry points of the render-and-compare path as registered PyTorch operators (`torch.ops.umr.*`).

umr_amd/ops.py registers the rasterizer (the reference's own extension module, soft_rasterize_cuda.cpp:141-144).  The
north_star asks for ALL kernels of the path as custom ops; this module registers the geometry and loss kernels the same way:
a schema, a device implementation, a fake (meta) kernel for FakeTensor tracing / torch.library.opcheck, and an autograd
formula whose backward is itself a registered operator.  The implementations are the ctypes calls of umr_amd/functional.py
(the *Kernel classes there; these operators are the ONE route from the loss / geometry modules into them -- umr_amd.functional's
call wrappers, and the `XFunction.apply` forms kept for tests, forward here).

  umr::project_points(verts[N,V,3], cams[N,7], int out_dim, float offset_z) -> [N,V,out_dim]     geom_utils.py:60-91
  umr::neg_iou(predict[N,...], target[N,...]) -> (loss[N], sums)                                  loss_utils.py:41-48
  umr::chamfer(a[B,n,D], b[B,m,D]) -> (d1[B,n], d2[B,m], i1 int32, i2 int32)                      chamfer_python.py:43-64
  umr::grid_sample_cl(image[B,C,H,W], grid[B,P,2]) -> [B,P,C]                                     geom_utils.py:41-59
  umr::laplacian(x[B,V,3], nbr_off int32, nbr_idx int32) -> (loss[B], lap[B,V,3])                 soft_renderer/losses.py:6-37
  umr::flatten(x[B,V,3], quads[E,4] int32) -> loss[B]                                             soft_renderer/losses.py:39-114
  umr::cos_sim(Tensor[] feats0, Tensor[] feats1, float eps) -> (val[N], workspace)                networks_basic.py:42-64
  umr::part_match(render_a, render_b, part_segs, float[] weights5, float background, float eps) -> (l_eqv, l_lm, workspace)
  umr::dt_barrier(mask[B,H,W], float k) -> [B,H,W]                                                utils/image.py:130-141
  umr::row_norm_mean(x[...,W]) -> scalar, umr::abs_column_mean(x[...,W], int column) -> scalar    loss_utils.py:118-126
  umr::masked_l1(img_pred, img_gt, mask_gt, mask_pred) -> per_sample[B]                           loss_utils.py:103-116
  umr::project_faces(verts, cams, faces int32, float offset_z, float eye_z) -> face_vertices[N,F,3,3]   smr.py:36-44 fused
  umr::project_faces_lit(verts, cams, faces, offset_z, eye_z, bool want_pre, float[] light) -> (face_pre, face_vertices, light[N,F,3])
        the same with the pre-look_at faces and / or sr.Lighting's per-face surface light (lighting.py:50-57); light = [] or
        [ambient, directional, colour x 3, direction x 3]
  umr::upsample2x_bilinear(x[B,C,H,W]) -> [B,C,2H,2W]                                             cub_mesh.py:150-157 (texture decoder)
  umr::perceptual_prologue(img, mask, float[] shift, float[] scale) -> Tensor                     loss_utils.py:141-146 input side
each with umr::<name>_backward.
"""
from typing import List

import torch
from torch.library import Library, register_autograd, register_fake

from . import functional as UF

_LIB = Library("umr", "FRAGMENT")


_Ctx = UF.KernelCtx     # the kernels' own forward -> backward context (umr_amd/functional.py)


def _f32(t, *shape):
    return t.new_empty(shape, dtype=torch.float32)


IMPLS = []      # (name, implementation): tests/host_raster.py::emulated_product registers them for host tensors as well


def _define(name, schema, impl, fake):
    _LIB.define(name + schema)
    _LIB.impl(name, impl, "CUDA")
    IMPLS.append((name, impl))
    register_fake("umr::" + name)(fake)


# ---------------------------------------------------------------------------------------------- project_points
_define("project_points", "(Tensor verts, Tensor cams, int out_dim, float offset_z) -> Tensor",
        lambda verts, cams, out_dim, offset_z: UF.ProjectPointsKernel.forward(_Ctx(), verts, cams, out_dim, offset_z),
        lambda verts, cams, out_dim, offset_z: _f32(verts, verts.shape[0], verts.shape[1], out_dim))


def _pp_bwd(grad, verts, cams, out_dim, need_verts):
    ctx = _Ctx((need_verts, True), out_dim=out_dim)
    ctx.saved_tensors = (UF._f32c(verts), UF._f32c(cams))
    gv, gc, _, _ = UF.ProjectPointsKernel.backward(ctx, grad)
    return (gv if gv is not None else verts.new_empty(0)), gc


_define("project_points_backward", "(Tensor grad, Tensor verts, Tensor cams, int out_dim, bool need_verts) -> (Tensor, Tensor)", _pp_bwd,
        lambda grad, verts, cams, out_dim, need_verts: ((_f32(verts, *verts.shape) if need_verts else _f32(verts, 0)), _f32(cams, *cams.shape)))


def _pp_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.out_dim = inputs[2]


def _pp_autograd(ctx, g):
    v, c = ctx.saved_tensors
    gv, gc = torch.ops.umr.project_points_backward(g, v, c, ctx.out_dim, ctx.needs_input_grad[0])
    return (gv if ctx.needs_input_grad[0] else None), (gc if ctx.needs_input_grad[1] else None), None, None


register_autograd("umr::project_points", _pp_autograd, setup_context=_pp_setup)


# ---------------------------------------------------------------------------------------------- neg_iou
def _iou_fwd(predict, target):
    ctx = _Ctx()
    loss = UF.NegIoUKernel.forward(ctx, predict, target)
    return loss, ctx.saved_tensors[2]


def _iou_fake(predict, target):
    N = predict.shape[0]
    P = predict.numel() // max(N, 1)
    return _f32(predict, N), _f32(predict, N, 2 + 2 * ((P + 2047) // 2048))


_define("neg_iou", "(Tensor predict, Tensor target) -> (Tensor, Tensor)", _iou_fwd, _iou_fake)


def _iou_bwd(grad, predict, target, sums):
    ctx = _Ctx(shape=predict.shape)
    ctx.saved_tensors = (UF._f32c(predict).view(predict.shape[0], -1), UF._f32c(target).view(target.shape[0], -1), sums)
    return UF.NegIoUKernel.backward(ctx, grad)[0]


_define("neg_iou_backward", "(Tensor grad, Tensor predict, Tensor target, Tensor sums) -> Tensor", _iou_bwd,
        lambda grad, predict, target, sums: _f32(predict, *predict.shape))
register_autograd("umr::neg_iou", lambda ctx, g, _gs: (torch.ops.umr.neg_iou_backward(g, *ctx.saved_tensors), None),
                  setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0], inputs[1], output[1]))


# ---------------------------------------------------------------------------------------------- chamfer
_define("chamfer", "(Tensor a, Tensor b) -> (Tensor, Tensor, Tensor, Tensor)",
        lambda a, b: UF.ChamferKernel.forward(_Ctx(), a, b),
        lambda a, b: (_f32(a, a.shape[0], a.shape[1]), _f32(a, b.shape[0], b.shape[1]),
                      a.new_empty((a.shape[0], a.shape[1]), dtype=torch.int32), a.new_empty((b.shape[0], b.shape[1]), dtype=torch.int32)))


def _ch_bwd(g1, g2, a, b, i1, i2):
    ctx = _Ctx()
    ctx.saved_tensors = (UF._f32c(a), UF._f32c(b), i1, i2)
    return UF.ChamferKernel.backward(ctx, g1, g2, None, None)


_define("chamfer_backward", "(Tensor g1, Tensor g2, Tensor a, Tensor b, Tensor i1, Tensor i2) -> (Tensor, Tensor)", _ch_bwd,
        lambda g1, g2, a, b, i1, i2: (_f32(a, *a.shape), _f32(b, *b.shape)))


def _ch_autograd(ctx, g1, g2, _a, _b):
    a, b, i1, i2 = ctx.saved_tensors
    g1 = g1 if g1 is not None else a.new_zeros(a.shape[:2])
    g2 = g2 if g2 is not None else b.new_zeros(b.shape[:2])
    return torch.ops.umr.chamfer_backward(g1, g2, a, b, i1, i2)


register_autograd("umr::chamfer", _ch_autograd,
                  setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0], inputs[1], output[2], output[3]))


# ---------------------------------------------------------------------------------------------- grid_sample (channels last)
_define("grid_sample_cl", "(Tensor image, Tensor grid) -> Tensor",
        lambda image, grid: UF.GridSampleCLKernel.forward(_Ctx(), image, grid),
        lambda image, grid: _f32(image, image.shape[0], grid.shape[1], image.shape[1]))


def _gs_bwd(grad, image, grid, need_image, need_grid):
    ctx = _Ctx((need_image, need_grid))
    ctx.saved_tensors = (UF._f32c(image), UF._f32c(grid))
    gi, gg = UF.GridSampleCLKernel.backward(ctx, grad)
    return (gi if gi is not None else image.new_empty(0)), (gg if gg is not None else image.new_empty(0))


_define("grid_sample_cl_backward", "(Tensor grad, Tensor image, Tensor grid, bool need_image, bool need_grid) -> (Tensor, Tensor)", _gs_bwd,
        lambda grad, image, grid, need_image, need_grid: ((_f32(image, *image.shape) if need_image else _f32(image, 0)),
                                                           (_f32(grid, *grid.shape) if need_grid else _f32(image, 0))))


def _gs_autograd(ctx, g):
    image, grid = ctx.saved_tensors
    ni, ng = ctx.needs_input_grad
    gi, gg = torch.ops.umr.grid_sample_cl_backward(g, image, grid, ni, ng)
    return (gi if ni else None), (gg if ng else None)


register_autograd("umr::grid_sample_cl", _gs_autograd, setup_context=lambda ctx, inputs, output: ctx.save_for_backward(*inputs))


# ---------------------------------------------------------------------------------------------- laplacian / flatten
def _lap_fwd(x, nbr_off, nbr_idx):
    ctx = _Ctx()
    loss = UF.LaplacianKernel.forward(ctx, x, nbr_off, nbr_idx)
    return loss, ctx.saved_tensors[0]


_define("laplacian", "(Tensor x, Tensor nbr_off, Tensor nbr_idx) -> (Tensor, Tensor)", _lap_fwd,
        lambda x, nbr_off, nbr_idx: (_f32(x, x.shape[0]), _f32(x, *x.shape)))


def _lap_bwd(grad, lap, nbr_off, nbr_idx):
    ctx = _Ctx()
    ctx.saved_tensors = (lap, nbr_off, nbr_idx)
    return UF.LaplacianKernel.backward(ctx, grad)[0]


_define("laplacian_backward", "(Tensor grad, Tensor lap, Tensor nbr_off, Tensor nbr_idx) -> Tensor", _lap_bwd,
        lambda grad, lap, nbr_off, nbr_idx: _f32(lap, *lap.shape))
register_autograd("umr::laplacian", lambda ctx, g, _gl: (torch.ops.umr.laplacian_backward(g, *ctx.saved_tensors), None, None),
                  setup_context=lambda ctx, inputs, output: ctx.save_for_backward(output[1], inputs[1], inputs[2]))

_define("flatten", "(Tensor x, Tensor quads) -> Tensor",
        lambda x, quads: UF.FlattenKernel.forward(_Ctx(), x, quads), lambda x, quads: _f32(x, x.shape[0]))


def _fl_bwd(grad, x, quads):
    ctx = _Ctx()
    ctx.saved_tensors = (UF._f32c(x), quads)
    return UF.FlattenKernel.backward(ctx, grad)[0]


_define("flatten_backward", "(Tensor grad, Tensor x, Tensor quads) -> Tensor", _fl_bwd, lambda grad, x, quads: _f32(x, *x.shape))
register_autograd("umr::flatten", lambda ctx, g: (torch.ops.umr.flatten_backward(g, *ctx.saved_tensors), None),
                  setup_context=lambda ctx, inputs, output: ctx.save_for_backward(*inputs))


# ---------------------------------------------------------------------------------------------- PNet head (cos_sim)
def _cos_fwd(feats0: List[torch.Tensor], feats1: List[torch.Tensor], eps: float):
    ctx = _Ctx()
    val = UF.CosSimDistanceKernel.forward(ctx, eps, *feats0, *feats1)
    return val, ctx.saved_tensors[0]


def _cos_ws_bytes(feats0):
    # umr_cos_sim_workspace_bytes: three per-pixel coefficients per tap and image (include/umr_hip.h); the fake kernel only needs
    # a size of the right dtype -- ask the library (a host call, no GPU work)
    import ctypes
    from . import _lib
    T, N = len(feats0), feats0[0].shape[0]
    P = (ctypes.c_int * T)(*[int(a.shape[2] * a.shape[3]) for a in feats0])
    return int(_lib.lib().umr_cos_sim_workspace_bytes(T, N, P))


_define("cos_sim", "(Tensor[] feats0, Tensor[] feats1, float eps) -> (Tensor, Tensor)", _cos_fwd,
        lambda feats0, feats1, eps: (_f32(feats0[0], feats0[0].shape[0]),
                                     feats0[0].new_empty((_cos_ws_bytes(feats0),), dtype=torch.uint8)))


def _cos_bwd(grad, feats0: List[torch.Tensor], feats1: List[torch.Tensor], ws, eps: float, need: List[bool]):
    T = len(feats0)
    ctx = _Ctx((False,) + tuple(need), eps=float(eps), T=T)
    ctx.saved_tensors = (ws,) + tuple(UF._f32c(f) for f in feats0) + tuple(UF._f32c(f) for f in feats1)
    out = UF.CosSimDistanceKernel.backward(ctx, grad)[1:]
    return [g if g is not None else grad.new_empty(0) for g in out]


_define("cos_sim_backward", "(Tensor grad, Tensor[] feats0, Tensor[] feats1, Tensor ws, float eps, bool[] need) -> Tensor[]", _cos_bwd,
        lambda grad, feats0, feats1, ws, eps, need: [(_f32(f, *f.shape) if n else _f32(f, 0))
                                                     for f, n in zip(list(feats0) + list(feats1), need)])


def _cos_setup(ctx, inputs, output):
    f0, f1, eps = inputs
    ctx.T, ctx.eps = len(f0), eps
    ctx.save_for_backward(output[1], *f0, *f1)


def _cos_autograd(ctx, g, _gws):
    T = ctx.T
    ws, feats = ctx.saved_tensors[0], ctx.saved_tensors[1:]
    need = [bool(t.requires_grad) for t in feats]
    if not any(need):
        return None, None, None
    grads = torch.ops.umr.cos_sim_backward(g, list(feats[:T]), list(feats[T:]), ws, ctx.eps, need)
    grads = [gr if n else None for gr, n in zip(grads, need)]
    return grads[:T], grads[T:], None


register_autograd("umr::cos_sim", _cos_autograd, setup_context=_cos_setup)


# ---------------------------------------------------------------------------------------------- part_match
def _pm_fwd(render_a, render_b, part_segs, weights5: List[float], background: float, center_eps: float):
    ctx = _Ctx()
    e, l = UF.PartMatchKernel.forward(ctx, render_a, render_b, part_segs, weights5, background, center_eps)
    return e.clone(), l.clone(), ctx.saved_tensors[3]     # (the two results are rows of one buffer: outputs may not alias)


def _pm_ws_bytes(a):
    from . import _lib
    return int(_lib.lib().umr_part_match_workspace_bytes(a.shape[0], a.shape[2], a.shape[3]))


_define("part_match", "(Tensor render_a, Tensor render_b, Tensor part_segs, float[] weights5, float background, float center_eps) -> (Tensor, Tensor, Tensor)",
        _pm_fwd, lambda a, b, q, w, bg, eps: (_f32(a, a.shape[0]), _f32(a, a.shape[0]), a.new_empty((_pm_ws_bytes(a),), dtype=torch.uint8)))


def _pm_bwd(g_eqv, g_lm, render_a, render_b, part_segs, ws, weights5: List[float], background: float, center_eps: float):
    ctx = _Ctx(cfg=([float(x) for x in weights5], float(background), float(center_eps)))
    ctx.saved_tensors = (UF._f32c(render_a), UF._f32c(render_b), UF._f32c(part_segs), ws)
    ga, gb = UF.PartMatchKernel.backward(ctx, g_eqv, g_lm)[:2]
    return ga, gb


_define("part_match_backward", "(Tensor g_eqv, Tensor g_lm, Tensor render_a, Tensor render_b, Tensor part_segs, Tensor ws, float[] weights5, "
        "float background, float center_eps) -> (Tensor, Tensor)", _pm_bwd,
        lambda ge, gl, a, b, q, ws, w, bg, eps: (_f32(a, *a.shape), _f32(b, *b.shape)))


def _pm_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], inputs[2], output[2])
    ctx.cfg = (inputs[3], inputs[4], inputs[5])


def _pm_autograd(ctx, ge, gl, _gws):
    a, b, q, ws = ctx.saved_tensors
    ge = ge if ge is not None else a.new_zeros(a.shape[0])
    gl = gl if gl is not None else a.new_zeros(a.shape[0])
    ga, gb = torch.ops.umr.part_match_backward(ge, gl, a, b, q, ws, *ctx.cfg)
    return ga, gb, None, None, None, None


register_autograd("umr::part_match", _pm_autograd, setup_context=_pm_setup)


# ---------------------------------------------------------------------------------------------- dt_barrier (forward only)
def _dt_fwd(mask, k: float):
    from .image_utils import compute_dt_barrier
    return compute_dt_barrier(mask, k)


_define("dt_barrier", "(Tensor mask, float k) -> Tensor", _dt_fwd, lambda mask, k: _f32(mask, *mask.shape))


# ---------------------------------------------------------------------------------------------- small regularisers, masked L1
_define("row_norm_mean", "(Tensor x) -> Tensor", lambda x: UF.RowNormMeanKernel.forward(_Ctx(), x), lambda x: _f32(x))


def _rn_bwd(grad, x):
    ctx = _Ctx(shape=x.shape)
    ctx.saved_tensors = (UF._f32c(x).view(-1, x.shape[-1]),)
    return UF.RowNormMeanKernel.backward(ctx, grad)


_define("row_norm_mean_backward", "(Tensor grad, Tensor x) -> Tensor", _rn_bwd, lambda grad, x: _f32(x, *x.shape))
register_autograd("umr::row_norm_mean", lambda ctx, g: torch.ops.umr.row_norm_mean_backward(g, *ctx.saved_tensors),
                  setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]))

_define("abs_column_mean", "(Tensor x, int column) -> Tensor", lambda x, column: UF.AbsColumnMeanKernel.forward(_Ctx(), x, column),
        lambda x, column: _f32(x))


def _ac_bwd(grad, x, column):
    ctx = _Ctx(shape=x.shape, column=int(column))
    ctx.saved_tensors = (UF._f32c(x).view(-1, x.shape[-1]),)
    return UF.AbsColumnMeanKernel.backward(ctx, grad)[0]


_define("abs_column_mean_backward", "(Tensor grad, Tensor x, int column) -> Tensor", _ac_bwd, lambda grad, x, column: _f32(x, *x.shape))


def _ac_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])
    ctx.column = inputs[1]


register_autograd("umr::abs_column_mean", lambda ctx, g: (torch.ops.umr.abs_column_mean_backward(g, ctx.saved_tensors[0], ctx.column), None),
                  setup_context=_ac_setup)

_define("masked_l1", "(Tensor img_pred, Tensor img_gt, Tensor mask_gt, Tensor mask_pred) -> Tensor",
        lambda ip, ig, mg, mp: UF.MaskedL1Kernel.forward(_Ctx(), ip, ig, mg, mp), lambda ip, ig, mg, mp: _f32(ip, ip.shape[0]))


def _ml_bwd(grad, ip, ig, mg, mp, need_img, need_mask):
    ctx = _Ctx((need_img, False, False, need_mask), mp_shape=mp.shape)
    ctx.saved_tensors = (UF._f32c(ip), UF._f32c(ig), UF._f32c(mg), UF._f32c(mp))
    gi, _, _, gm = UF.MaskedL1Kernel.backward(ctx, grad)
    return (gi if gi is not None else ip.new_empty(0)), (gm if gm is not None else ip.new_empty(0))


_define("masked_l1_backward", "(Tensor grad, Tensor img_pred, Tensor img_gt, Tensor mask_gt, Tensor mask_pred, bool need_img, bool need_mask) "
        "-> (Tensor, Tensor)", _ml_bwd,
        lambda grad, ip, ig, mg, mp, ni, nm: ((_f32(ip, *ip.shape) if ni else _f32(ip, 0)), (_f32(mp, *mp.shape) if nm else _f32(ip, 0))))


def _ml_autograd(ctx, g):
    ni, nm = ctx.needs_input_grad[0], ctx.needs_input_grad[3]
    gi, gm = torch.ops.umr.masked_l1_backward(g, *ctx.saved_tensors, ni, nm)
    return (gi if ni else None), None, None, (gm if nm else None)


register_autograd("umr::masked_l1", _ml_autograd, setup_context=lambda ctx, inputs, output: ctx.save_for_backward(*inputs))


# ---------------------------------------------------------------------------------------------- project_faces (ambient light)
def _pf_fwd(verts, cams, faces_idx, offset_z: float, eye_z: float):
    return UF.ProjectFacesKernel.forward(_Ctx(), verts, cams, faces_idx, offset_z, eye_z, False, None)[1]


_define("project_faces", "(Tensor verts, Tensor cams, Tensor faces_idx, float offset_z, float eye_z) -> Tensor", _pf_fwd,
        lambda verts, cams, faces_idx, offset_z, eye_z: _f32(verts, cams.shape[0], faces_idx.shape[1], 3, 3))


def _pf_bwd(grad, verts, cams, faces_idx, need_verts):
    ctx = _Ctx((need_verts, True), light=None, want_pre=False)
    ctx.saved_tensors = (UF._f32c(verts), UF._f32c(cams), faces_idx)
    gv, gc = UF.ProjectFacesKernel.backward(ctx, None, grad, None)[:2]
    return (gv if gv is not None else verts.new_empty(0)), gc


_define("project_faces_backward", "(Tensor grad, Tensor verts, Tensor cams, Tensor faces_idx, bool need_verts) -> (Tensor, Tensor)", _pf_bwd,
        lambda grad, verts, cams, faces_idx, need_verts: ((_f32(verts, *verts.shape) if need_verts else _f32(verts, 0)), _f32(cams, *cams.shape)))


def _pf_autograd(ctx, g):
    v, c, f = ctx.saved_tensors
    gv, gc = torch.ops.umr.project_faces_backward(g, v, c, f, ctx.needs_input_grad[0])
    return (gv if ctx.needs_input_grad[0] else None), (gc if ctx.needs_input_grad[1] else None), None, None, None


register_autograd("umr::project_faces", _pf_autograd, setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0], inputs[1], inputs[2]))


# ---------------------------------------------------------------------------------------------- project_faces_lit (full form)
def _light_tuple(light):
    return None if len(light) == 0 else (light[0], light[1], tuple(light[2:5]), tuple(light[5:8]))


def _pfl_fwd(verts, cams, faces_idx, offset_z: float, eye_z: float, want_pre: bool, light: List[float]):
    return UF.ProjectFacesKernel.forward(_Ctx(), verts, cams, faces_idx, offset_z, eye_z, want_pre, _light_tuple(light))


def _pfl_fake(verts, cams, faces_idx, offset_z, eye_z, want_pre, light):
    N, F = cams.shape[0], faces_idx.shape[1]
    return ((_f32(verts, N, F, 3, 3) if want_pre else _f32(verts, 0)), _f32(verts, N, F, 3, 3),
            (_f32(verts, N, F, 3) if len(light) else _f32(verts, 0)))


_define("project_faces_lit", "(Tensor verts, Tensor cams, Tensor faces_idx, float offset_z, float eye_z, bool want_pre, float[] light) "
        "-> (Tensor, Tensor, Tensor)", _pfl_fwd, _pfl_fake)


def _pfl_bwd(g_pre, g_out, g_light, face_out, verts, cams, faces_idx, want_pre: bool, light: List[float], need_verts: bool):
    import ctypes
    lt = _light_tuple(light)
    ctx = _Ctx((need_verts, True), want_pre=want_pre,
               light=((float(lt[1]), (ctypes.c_float * 3)(*[float(x) for x in lt[2]]), (ctypes.c_float * 3)(*[float(x) for x in lt[3]]))
                      if lt is not None else None))
    ctx.saved_tensors = (UF._f32c(verts), UF._f32c(cams), faces_idx) + ((face_out,) if lt is not None else ())
    gv, gc = UF.ProjectFacesKernel.backward(ctx, (g_pre if want_pre else None), g_out, (g_light if lt is not None else None))[:2]
    return (gv if gv is not None else verts.new_empty(0)), gc


_define("project_faces_lit_backward", "(Tensor g_pre, Tensor g_out, Tensor g_light, Tensor face_out, Tensor verts, Tensor cams, Tensor faces_idx, "
        "bool want_pre, float[] light, bool need_verts) -> (Tensor, Tensor)", _pfl_bwd,
        lambda g_pre, g_out, g_light, face_out, verts, cams, faces_idx, want_pre, light, need_verts:
        ((_f32(verts, *verts.shape) if need_verts else _f32(verts, 0)), _f32(cams, *cams.shape)))


def _pfl_setup(ctx, inputs, output):
    verts, cams, faces_idx, _, _, want_pre, light = inputs
    ctx.want_pre, ctx.light = want_pre, list(light)
    ctx.save_for_backward(verts, cams, faces_idx, output[1])


def _pfl_autograd(ctx, g_pre, g_out, g_light):
    v, c, f, face_out = ctx.saved_tensors
    N, F = c.shape[0], f.shape[1]
    z = lambda g, *shape: g if g is not None else face_out.new_zeros(shape)
    gv, gc = torch.ops.umr.project_faces_lit_backward(z(g_pre, N, F, 3, 3) if ctx.want_pre else face_out.new_empty(0), z(g_out, N, F, 3, 3),
                                                       z(g_light, N, F, 3) if ctx.light else face_out.new_empty(0), face_out, v, c, f,
                                                       ctx.want_pre, ctx.light, ctx.needs_input_grad[0])
    return (gv if ctx.needs_input_grad[0] else None), (gc if ctx.needs_input_grad[1] else None), None, None, None, None, None


register_autograd("umr::project_faces_lit", _pfl_autograd, setup_context=_pfl_setup)


# ---------------------------------------------------------------------------------------------- upsample2x_bilinear
_define("upsample2x_bilinear", "(Tensor x) -> Tensor", lambda x: UF.Upsample2xBilinearKernel.forward(_Ctx(), x),
        lambda x: _f32(x, x.shape[0], x.shape[1], 2 * x.shape[2], 2 * x.shape[3]))
_define("upsample2x_bilinear_backward", "(Tensor grad) -> Tensor",
        lambda grad: UF.Upsample2xBilinearKernel.backward(_Ctx(shape=(grad.shape[0], grad.shape[1], grad.shape[2] // 2, grad.shape[3] // 2)), grad),
        lambda grad: _f32(grad, grad.shape[0], grad.shape[1], grad.shape[2] // 2, grad.shape[3] // 2))
register_autograd("umr::upsample2x_bilinear", lambda ctx, g: torch.ops.umr.upsample2x_bilinear_backward(g),
                  setup_context=lambda ctx, inputs, output: None)


# ---------------------------------------------------------------------------------------------- perceptual_prologue
_define("perceptual_prologue", "(Tensor img, Tensor mask, float[] shift, float[] scale) -> Tensor",
        lambda img, mask, shift, scale: UF.PerceptualPrologueKernel.forward(_Ctx(), img, mask, shift, scale),
        lambda img, mask, shift, scale: _f32(img, *img.shape))


def _ppl_bwd(grad, img, mask, scale: List[float], need_img: bool, need_mask: bool):
    ctx = _Ctx((need_img, need_mask), scale=[float(v) for v in scale])
    ctx.saved_tensors = (UF._f32c(img), UF._f32c(mask))
    gi, gm = UF.PerceptualPrologueKernel.backward(ctx, grad)[:2]
    return (gi if gi is not None else img.new_empty(0)), (gm if gm is not None else img.new_empty(0))


_define("perceptual_prologue_backward", "(Tensor grad, Tensor img, Tensor mask, float[] scale, bool need_img, bool need_mask) -> (Tensor, Tensor)",
        _ppl_bwd, lambda grad, img, mask, scale, need_img, need_mask: ((_f32(img, *img.shape) if need_img else _f32(img, 0)),
                                                                      (_f32(mask, *mask.shape) if need_mask else _f32(img, 0))))


def _ppl_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.scale = list(inputs[3])


def _ppl_autograd(ctx, g):
    img, mask = ctx.saved_tensors
    ni, nm = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    gi, gm = torch.ops.umr.perceptual_prologue_backward(g, img, mask, ctx.scale, ni, nm)
    return (gi if ni else None), (gm.view(mask.shape) if nm else None), None, None


register_autograd("umr::perceptual_prologue", _ppl_autograd, setup_context=_ppl_setup)

ALL_OPS = ("project_points", "neg_iou", "chamfer", "grid_sample_cl", "laplacian", "flatten", "cos_sim", "part_match", "dt_barrier",
           "row_norm_mean", "abs_column_mean", "masked_l1", "project_faces", "project_faces_lit", "upsample2x_bilinear",
           "perceptual_prologue")
