"""GPU parity tests added in round 3 (run with -m gpu): the lean geometry of the face-major backward against the
reference-order geometry and against the oracle, BASELINE config 4's shape (512^2 render = IS 1024, 5120 faces) through the
train_s2 render-and-compare module, regularisers / eval kernels / rotate_cam goldens.

Every comparison is HIP (through the C ABI) against golden vectors written by the reference itself, the CPU oracle on the same
seeded inputs, or -- where a kernel has two formulations -- the reference-order formulation of the same library; tolerances are
written at each check and the measured figures are appended to gpurun_out/parity_measured.jsonl (helpers.assert_close_frac)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import scene, assert_close_frac, t2n

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RARGS = ([0.1, 0.2, 0.3], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4)


def _backward_set(fv0, tex0, IS, w, wa, pool):
    """Gradients of the three production variants of the face-major backward: vertex + texel, texel only, silhouette."""
    from umr_amd import functional as UF
    res = {}
    fv = fv0.detach().clone().requires_grad_(True); tex = tex0.clone().requires_grad_(True)
    sc, _, _ = UF.soft_rasterize(fv, tex, IS, *RARGS, 'softmax', pool=pool)
    (sc * w).sum().backward()
    res["full_gf"], res["full_gt"] = fv.grad.clone(), tex.grad.clone()
    tex = tex0.clone().requires_grad_(True)
    sc, _, _ = UF.soft_rasterize(fv0.detach(), tex, IS, *RARGS, 'softmax', pool=pool)
    (sc * w).sum().backward()
    res["texel_only_gt"] = tex.grad.clone()
    fv = fv0.detach().clone().requires_grad_(True)
    sc, _, _ = UF.soft_rasterize(fv, tex0.detach(), IS, *RARGS, 'softmax', pool=pool)
    (sc * w).sum().backward()
    res["vertex_only_gf"] = fv.grad.clone()
    fv = fv0.detach().clone().requires_grad_(True)
    a = UF.SilhouetteFunction.apply(fv, IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, pool)
    (a * wa).sum().backward()
    res["silhouette_gf"] = fv.grad.clone()
    return res


@pytest.mark.parametrize("cfg", [(2, 3, 512, 36, True), (2, 4, 1024, 36, True), (3, 2, 200, 4, False), (2, 3, 256, 1, True)])
def test_lean_backward_vs_reference_order_backward(cfg):
    """Faces flagged well-conditioned take the lean geometry in the face-major backward (raster_core.h lean_segments: the
    closest boundary point from three clamped edge projections); umr_debug_set("bwd_lean", 0) sends every face down the
    reference-order eval_pair.  Same closest point, different rounding: the reference's formulation carries ~1e-6 .. 1e-5 of
    offset noise (tests/test_kernel_source_on_host.py), so the two gradient sets agree to ~1e-4 of the gradient's scale, not
    to the bit.  BASELINE mesh sizes (642 / 2562 vertices) at IS 512 / 1024, a ragged image, TS = 1."""
    from umr_amd import _lib, functional as UF
    n, sub, IS, ts, pool = cfg
    verts, faces, cams, gen = scene(n, sub, seed=IS + 11)
    _, fv0, _ = UF.ProjectFacesFunction.apply(verts.to(DEV), cams.to(DEV), faces.int().to(DEV), 5.0, -2.732, False)
    tex0 = torch.rand(n, faces.shape[1], ts, 3, generator=gen).to(DEV)
    H = IS // 2 if pool else IS
    w = torch.randn(n, 4, H, H, generator=gen).to(DEV)
    wa = torch.randn(n, H, H, generator=gen).to(DEV)
    out = {}
    for lean in (1, 0):
        _lib.debug_set("bwd_lean", lean)
        try:
            out[lean] = _backward_set(fv0, tex0, IS, w, wa, pool)
        finally:
            _lib.debug_set("bwd_lean", 1)
    for k in out[1]:
        a, b = t2n(out[1][k]), t2n(out[0][k])
        scale = float(np.abs(b).max())
        assert np.isfinite(a).all() and scale > 0, k
        # >= 99.9 % of the elements within 2e-4 of the gradient's scale + 2e-3 relative; no element off by more than 1 % of scale
        assert_close_frac(a, b, atol=2e-4 * scale, rtol=2e-3, frac=0.999, max_outlier=1e-2 * scale,
                          name="lean vs reference-order backward %s N=%d F=%d IS=%d TS=%d" % (k, n, faces.shape[1], IS, ts))


def test_lean_backward_vs_oracle_full_size(oracle_built):
    """The lean backward against the CPU oracle (the reference's algorithm, reference operation order) at 2 x 1280 faces x
    512^2, TS = 36: vertex and texel gradients of the textured soft-max render and the silhouette's vertex gradients."""
    from oracle import softras
    from umr_amd import functional as UF
    n, sub, IS, ts = 2, 3, 512, 36
    verts, faces, cams, gen = scene(n, sub, seed=77)
    _, fv0, _ = UF.ProjectFacesFunction.apply(verts.to(DEV), cams.to(DEV), faces.int().to(DEV), 5.0, -2.732, False)
    tex0 = torch.rand(n, faces.shape[1], ts, 3, generator=gen).to(DEV)
    w = torch.randn(n, 4, IS, IS, generator=gen).to(DEV)
    fv = fv0.detach().clone().requires_grad_(True); tex = tex0.clone().requires_grad_(True)
    sc, _, _ = UF.soft_rasterize(fv, tex, IS, *RARGS, 'softmax')
    (sc * w).sum().backward()
    cfg = dict(near=1., far=100., eps=1e-3, sigma_val=1e-5, dist_eps_log=float(np.log(1. / 1e-10 - 1.)), gamma_val=1e-4,
               func_id_rgb=1, double_side=True)
    nt = softras.max_threads()
    ref = softras.raster_forward(t2n(fv0).reshape(n, -1, 9), t2n(tex0), IS, background=(0.1, 0.2, 0.3), backend="port", n_threads=nt, **cfg)
    gf, gt = softras.raster_backward(ref["faces"], ref["textures"], ref["soft_colors"], ref["faces_info"], ref["aggrs_info"],
                                     t2n(w), IS, backend="port", n_threads=nt, **cfg)
    gref = {"grad_faces": gf, "grad_textures": gt}
    for name, got, want in (("grad_faces", fv.grad, gref["grad_faces"]), ("grad_textures", tex.grad, gref["grad_textures"])):
        a, b = t2n(got).reshape(want.shape), want
        scale = float(np.abs(b).max())
        assert_close_frac(a, b, atol=2e-4 * scale, rtol=5e-3, frac=0.998, max_outlier=2e-2 * scale,
                          name="lean backward vs oracle " + name)
