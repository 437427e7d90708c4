"""Round-5 GPU tests (MI355X, through the C ABI): the one-pass backward of the shared mask / texture render against the oracle
at full size, and the C ABI's refusal of UMR_BWD_ALPHA_GEOMETRY where the face-major kernels do not run."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import scene, assert_close_frac, t2n  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
UMR_BWD_GRAD_POOLED, UMR_BWD_ALPHA_GEOMETRY = 1, 4


def _forward_cabi(fv, tex, IS, pooled, flags=0):
    """umr_raster_forward with the reference's buffer contract (soft_colors pre-filled with (background, 1), the rest zeroed)."""
    from umr_amd import _lib
    from umr_amd.functional import standard_grid
    L, p = _lib.lib(), _lib.ptr
    N, F = fv.shape[:2]
    TS = tex.shape[2]
    st = dict(faces_info=torch.zeros(N, F, 27, device=DEV), aggrs=torch.zeros(N, 2, IS, IS, device=DEV),
              p2f_info=torch.zeros(N, F, 2, device=DEV), p2f_sum=torch.zeros(N, F, 2, device=DEV),
              sc=torch.cat((torch.zeros(N, 3, IS, IS, device=DEV), torch.ones(N, 1, IS, IS, device=DEV)), 1).contiguous(),
              pool=torch.empty(N, 4, IS // 2, IS // 2, device=DEV) if pooled else None)
    wsb = L.umr_raster_workspace_bytes(N, F)
    st["ws"], st["wsb"] = torch.empty(wsb, dtype=torch.uint8, device=DEV), wsb
    st["scal"] = (1.0, 100.0, 1e-3, 1e-5, 2, float(math.log(1e10 - 1.)), 1e-4, 1, 2, 0, 1)
    grid = standard_grid(IS, torch.device(DEV))
    rc = L.umr_raster_forward(p(fv), p(tex), p(st["faces_info"]), p(st["aggrs"]), p(grid), p(st["p2f_info"]), p(st["p2f_sum"]),
                              p(st["sc"]), p(st["pool"]), N, F, TS, IS, *st["scal"], flags, None, p(st["ws"]), wsb,
                              _lib.stream_ptr(torch.device(DEV)))
    assert rc == 0
    return st


@pytest.mark.parametrize("pooled", [True, False])
def test_alpha_geometry_backward_vs_oracle(oracle_built, pooled):
    """UMR_BWD_ALPHA_GEOMETRY -- the backward the timed step lives on -- against the ORACLE, directly, through the C ABI at
    BASELINE size (2 x 1280 faces x 512^2, TS 36).  The reference's backward (soft_rasterize_cuda_kernel.cu:480-656) is linear in
    the upstream gradient, so what the flag promises is two oracle calls:
        grad_faces    = backward with upstream (0, 0, 0, g_alpha)   (the alpha term alone reaches the geometry: the mask render's)
        grad_textures = backward with upstream (g_r, g_g, g_b, 0)   (the rgb term reaches the texels: the detached textured render's)
    every element, at the bounds of test_full_size_vs_oracle.  pooled: the gradient arrives at the 2x2-pooled resolution (the
    step's form, `COMMON` kernel); else at full resolution (the general instantiation)."""
    from oracle import softras, torch_ref
    from umr_amd import _lib
    L, p = _lib.lib(), _lib.ptr
    verts, faces, cams, gen = scene(2, 3, seed=31)
    proj = torch_ref.orthographic_proj_withz(verts, cams, 5.) * torch.tensor([1., -1., 1.])
    fv = torch_ref.face_vertices(torch_ref.look_at_ortho(proj), faces).contiguous()
    N, F, IS, TS = 2, fv.shape[1], 512, 36
    tex = torch.rand(N, F, TS, 3, generator=gen)
    S = IS // 2 if pooled else IS
    g = torch.randn(N, 4, S, S, generator=gen)
    # upstream gradient at the raster resolution, as the oracle takes it: avg_pool2d's backward hands every fine pixel g / 4
    g_full = (0.25 * g).repeat_interleave(2, 2).repeat_interleave(2, 3).contiguous() if pooled else g
    cfg = dict(near=1., far=100., eps=1e-3, sigma_val=1e-5, dist_eps_log=float(math.log(1e10 - 1.)), gamma_val=1e-4,
               func_id_rgb=1, double_side=True)
    nt = softras.max_threads()
    o = softras.raster_forward(fv.numpy(), tex.numpy(), IS, backend="port", n_threads=nt, **cfg)
    g_a, g_rgb = g_full.clone(), g_full.clone()
    g_a[:, :3] = 0
    g_rgb[:, 3] = 0
    args = (o["faces"], o["textures"], o["soft_colors"], o["faces_info"], o["aggrs_info"])
    gf_ref, _ = softras.raster_backward(*args, g_a.numpy(), IS, backend="port", n_threads=nt, **cfg)
    _, gt_ref = softras.raster_backward(*args, g_rgb.numpy(), IS, backend="port", n_threads=nt, **cfg)

    fvd, texd = fv.to(DEV).reshape(N, F, 9).contiguous(), tex.to(DEV)
    st = _forward_cabi(fvd, texd, IS, pooled)
    gd = g.to(DEV)
    gf = torch.zeros(N, F, 9, device=DEV)
    gt = torch.zeros(N, F, TS, 3, device=DEV)
    rc = L.umr_raster_backward(p(fvd), p(texd), p(st["sc"]), p(st["faces_info"]), p(st["aggrs"]), p(gf), p(gt), p(gd),
                               (UMR_BWD_GRAD_POOLED if pooled else 0) | UMR_BWD_ALPHA_GEOMETRY, 1, 1, N, F, TS, IS, *st["scal"],
                               p(st["ws"]), st["wsb"], _lib.stream_ptr(torch.device(DEV)))
    assert rc == 0
    torch.cuda.synchronize()
    assert_close_frac(t2n(st["sc"]), o["soft_colors"], atol=1e-4, frac=1.0, max_outlier=1e-5, name="soft_colors")
    sf = np.abs(gf_ref).max()
    assert sf > 0 and np.abs(gt_ref).max() > 0
    assert_close_frac(t2n(gf).reshape(gf_ref.shape), gf_ref, atol=1e-5 * sf, rtol=1e-4, frac=1.0, name="grad_faces (alpha term)")
    stx = np.abs(gt_ref).max()
    assert_close_frac(t2n(gt), gt_ref, atol=3e-6 * stx, rtol=1e-4, frac=1.0, name="grad_textures (rgb term)")
    # nothing of the depth term (z gradients come from the rgb term only, :624-627) in grad_faces
    assert float(gf.view(N, F, 3, 3)[..., 2].abs().max()) == 0.0


def test_alpha_geometry_flag_is_refused_off_the_face_major_route():
    """The library never drops UMR_BWD_ALPHA_GEOMETRY silently (VERDICT r4 weak #6 / ADVICE r4): with the pixel-major A/B switch
    on, or more texels per face than the face-major kernels' LDS accumulators take, umr_raster_backward returns UMR_ERR_ARG and
    leaves both gradient buffers untouched; the same call on the face-major route succeeds."""
    from umr_amd import _lib
    from umr_amd import functional as UF
    L, p = _lib.lib(), _lib.ptr
    verts, faces, cams, gen = scene(2, 1, seed=3)
    _, fv, _ = UF.ProjectFacesFunction.apply(verts.to(DEV), cams.to(DEV), faces.int().to(DEV), 5.0, -2.732, False)
    N, F, IS = 2, fv.shape[1], 64
    fvd = fv.detach().reshape(N, F, 9).contiguous()

    def run(TS):
        tex = torch.rand(N, F, TS, 3, generator=gen).to(DEV)
        st = _forward_cabi(fvd, tex, IS, True)
        g = torch.randn(N, 4, IS // 2, IS // 2, generator=gen).to(DEV)
        gf, gt = torch.zeros(N, F, 9, device=DEV), torch.zeros(N, F, TS, 3, device=DEV)
        rc = L.umr_raster_backward(p(fvd), p(tex), p(st["sc"]), None, p(st["aggrs"]), p(gf), p(gt), p(g),
                                   UMR_BWD_GRAD_POOLED | UMR_BWD_ALPHA_GEOMETRY, 1, 1, N, F, TS, IS, *st["scal"], p(st["ws"]),
                                   st["wsb"], _lib.stream_ptr(torch.device(DEV)))
        torch.cuda.synchronize()
        return rc, float(gf.abs().sum()), float(gt.abs().sum())

    rc, a, b = run(36)
    assert rc == 0 and a > 0 and b > 0
    _lib.debug_set("bwd_pixel_major", 1)
    try:
        assert run(36) == (-1, 0.0, 0.0)
    finally:
        _lib.debug_set("bwd_pixel_major", 0)
    assert run(1024) == (-1, 0.0, 0.0)          # 32 x 32 texels per face: beyond the one-pass kernel's budget
