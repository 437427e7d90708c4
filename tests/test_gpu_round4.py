"""Round-4 GPU tests (MI355X, through the C ABI): the shared mask / texture render, the configs[3] soak, the 10 000-pair
evaluation at its stated size."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import scene  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None


def test_alpha_geometry_render_routes_gradients_like_the_two_renders():
    """umr::soft_rasterize_alpha_geometry = ONE textured render whose alpha channel keeps its gradient to the geometry while the
    colour channels see it detached -- against the two renders the reference makes of the same views (mask render with gradients
    to the vertices, train_s1.py:199; textured render of detached vertices, :217): same image bits, the vertex gradient of the
    silhouette render, the texel gradient of the textured one.  opcheck: schema, fake kernel, autograd registration."""
    from torch.library import opcheck
    from umr_amd import ops  # noqa: F401
    from umr_amd import functional as UF
    verts, faces, cams, gen = scene(4, 2, seed=11)
    _, fv, _ = UF.ProjectFacesFunction.apply(verts.to(DEV), cams.to(DEV), faces.int().to(DEV), 5.0, -2.732, False)
    F = faces.shape[1]
    tex = torch.rand(4, F, 36, 3, generator=gen).to(DEV)
    IS = 128
    cfg = (IS, [0., 0., 0.], 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, 1, True, True, True)
    fv1, tex1 = fv.detach().clone().requires_grad_(True), tex.clone().requires_grad_(True)
    opcheck(torch.ops.umr.soft_rasterize_alpha_geometry.default, (fv1, tex1) + cfg,
            test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    img, p2f, aggr, saved, vis = torch.ops.umr.soft_rasterize_alpha_geometry(fv1, tex1, *cfg)
    g = torch.randn(img.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    (img * g).sum().backward()
    # the reference's two renders
    fv2, tex2 = fv.detach().clone().requires_grad_(True), tex.clone().requires_grad_(True)
    alpha = UF.SilhouetteFunction.apply(fv2, IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, True)
    img2, p2f2, aggr2, vis2 = UF.soft_rasterize(fv2.detach(), tex2, IS, [0, 0, 0], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4,
                                                'softmax', 'prod', 'surface', pool=True, need_p2f=True, want_visibility=True)
    ((alpha * g[:, 3]).sum() + (img2[:, :3] * g[:, :3]).sum()).backward()
    assert torch.equal(img[:, 3], alpha) and torch.equal(img[:, :3], img2[:, :3]) and torch.equal(vis, vis2)
    assert torch.equal(fv1.grad, fv2.grad), float((fv1.grad - fv2.grad).abs().max())
    assert torch.equal(tex1.grad, tex2.grad)
    # and nothing of the colour gradient reaches the geometry
    fv3 = fv.detach().clone().requires_grad_(True)
    img3 = torch.ops.umr.soft_rasterize_alpha_geometry(fv3, tex, *cfg)[0]
    (img3[:, :3] * g[:, :3]).sum().backward()
    assert float(fv3.grad.abs().max()) == 0.0


def test_shared_mask_render_step_equals_the_two_render_step():
    """RenderCompareS1 / RenderCompareS2 with and without share_mask_render at the bench shape (256^2, 1280 faces): every term
    equal to the bit, every gradient equal to rounding (the same per-view contributions, accumulated in another order)."""
    from umr_amd.synthetic import make_s1_inputs
    from umr_amd.train_step import RenderCompareS1
    tv, faces, outputs, batch = make_s1_inputs(2, 256, 3, seed=21, device=DEV)
    res = []
    for share in (True, False):
        rc = RenderCompareS1(tv.to(DEV), faces.to(DEV), 256, share_mask_render=share).to(DEV)
        leaves = [outputs[k].detach().clone().requires_grad_(True) for k in ("delta_v", "cam", "tex_flow")]
        out = dict(outputs, delta_v=leaves[0], cam=leaves[1], tex_flow=leaves[2])
        out["pred_vs"] = outputs["mean_shape"][None] + leaves[0]
        total, terms = rc(out, batch)
        total.backward()
        res.append(({k: float(v) for k, v in terms.items()}, [l.grad.clone() for l in leaves]))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert float((a - b).abs().max()) <= 5e-6 * float(b.abs().max())
