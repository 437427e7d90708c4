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
    silhouette render and the texel gradient of the textured one to rounding.  opcheck: schema, fake kernel, autograd registration."""
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
    # gradients: ONE backward pass over the pairs (UMR_BWD_ALPHA_GEOMETRY) against the silhouette backward + the texel-only
    # backward of the two renders -- the same per-pair terms summed in another order (measured 8e-8 / 1.5e-7 of the largest)
    assert float((fv1.grad - fv2.grad).abs().max()) <= 2e-6 * float(fv2.grad.abs().max())
    assert float((tex1.grad - tex2.grad).abs().max()) <= 2e-6 * float(tex2.grad.abs().max())
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


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_config4_training_soak_stays_finite(seed):
    """BASELINE configs[3] (train_s2 at 512^2 images = 1024^2 render, 2562 vertices / 5120 faces, K = 8): bench.py's own training
    step -- MeshNet, all 12 + 19 raster launches per image, every loss, Adam -- for 70 optimizer steps per seed at bs 4 (840
    image-steps over the three seeds, different network initialisations and data).  Every loss term of every step must be
    finite; on a failure the first offending step and term are reported (the terms are kept on the device, one read at the end).
    Round 3's early builds ended ~2 in 1000 steps of this shape on a non-finite loss (HISTORY.md 5); see tools/nan/nan_hunt.sh for
    the instrumented hunt."""
    import argparse
    from umr_amd.model import build_training_step_s2
    os.environ["UMR_WATCH_TERMS"] = "1"
    try:
        torch.manual_seed(1000 + seed)
        args = argparse.Namespace(batch=4, image_size=512, subdivide=4, workload="s2", epoch=0, graph=0, data_seed=500 + 10 * seed)
        step = build_training_step_s2(args, DEV, 1)
    finally:
        os.environ.pop("UMR_WATCH_TERMS", None)
    losses = [step() for _ in range(70)]
    torch.cuda.synchronize()
    names = list(step.watch[0][0]) + ["|delta_v|", "|cams|", "|flow|"]
    vals = torch.stack([v for _, v in step.watch]).cpu().numpy()            # [steps, terms + 3]
    bad = np.argwhere(~np.isfinite(vals))
    assert bad.size == 0, "first non-finite: step %d, %s; row %s" % (bad[0][0], names[bad[0][1]], dict(zip(names, vals[bad[0][0]].tolist())))
    assert all(math.isfinite(float(l)) for l in losses)
    params = torch.cat([p.detach().reshape(-1) for p in step.model.parameters()])
    assert bool(torch.isfinite(params).all())


def _eval_batch(g, P, K, F, T, S):
    kps = torch.rand(P, 2, K, 3, generator=g) * 1.8 - 0.9
    kps[..., 2] = (kps[..., 2] > -0.7).float()
    flows = (torch.rand(P, 2, F, 1, 1, 2, generator=g) * 1.6 - 0.8 + 0.08 * (torch.rand(P, 2, F, T, T, 2, generator=g) - 0.5)).clamp(-1, 1)
    cams = torch.cat([0.6 + 0.3 * torch.rand(P, 2, 1, generator=g), 0.2 * torch.rand(P, 2, 2, generator=g) - 0.1,
                      torch.nn.functional.normalize(torch.randn(P, 2, 4, generator=g), dim=2)], 2)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, S), torch.linspace(-1, 1, S), indexing="ij")
    r = 0.25 + 0.2 * torch.rand(P, 2, 1, 1, generator=g)
    masks = ((xx[None, None] ** 2 + yy[None, None] ** 2) < r).float()
    return kps, flows, cams, masks


def test_keypoint_transfer_evaluation_at_10k_pairs(oracle_built):
    """BASELINE configs[4] at its stated size: 10 000 image pairs x 2 directions x 15 keypoints (test_kp.py:195-324) through the
    device evaluation path (csrc/eval.hip), 40 batches of 250 pairs of fresh synthetic flows / cameras / masks, PCK accumulated
    in the device counters.  Checked: (i) 100 pair-directions per mode against the CPU restatement of test_kp.py:125-193
    (oracle.torch_ref.map_kp_flow / map_kp_cam, pinned on tests/golden/eval_kp.npz written by the reference's modules): the
    arg-max FACE / nearest VERTEX pick through its transferred point, exact; (ii) on all 20 000 pair-directions the device
    counters against the PCK of test_kp.py:253-258 recomputed on the host from the returned points: visible counts equal,
    threshold counts equal up to points within 1e-6 of a threshold."""
    from oracle import torch_ref
    from umr_amd import eval_utils as EU
    from umr_amd.synthetic import template
    PAIRS, BATCH, K, T, S = 10000, 250, 15, 6, 256
    tv, faces = template(3)
    F = faces.shape[0]
    mean_shape_c = tv * 0.9
    mean_shape = mean_shape_c.to(DEV)
    g = torch.Generator().manual_seed(123)
    cnt = {m: EU.PCKCounters(K, DEV) for m in ("flow", "cam")}
    host = {m: np.zeros((3, K), np.int64) for m in ("flow", "cam")}
    near = {m: 0 for m in ("flow", "cam")}
    for b in range(PAIRS // BATCH):
        kps, flows, cams, masks = _eval_batch(g, BATCH, K, F, T, S)
        vis = kps[:, 0, :, 2] * kps[:, 1, :, 2]
        src = torch.cat([kps[:, 0], kps[:, 1]]); gt = torch.cat([kps[:, 1], kps[:, 0]]); v2 = torch.cat([vis, vis])
        fsrc, ftgt = torch.cat([flows[:, 0], flows[:, 1]]), torch.cat([flows[:, 1], flows[:, 0]])
        csrc, ctgt, mtgt = torch.cat([cams[:, 0], cams[:, 1]]), torch.cat([cams[:, 1], cams[:, 0]]), torch.cat([masks[:, 1], masks[:, 0]])
        out = {}
        out["flow"] = EU.map_kp_flow_batch(src.to(DEV), fsrc.to(DEV), ftgt.to(DEV), S, 3, kp_gt=gt.to(DEV), vis=v2.to(DEV), counters=cnt["flow"])
        out["cam"] = EU.map_kp_cam_batch(src.to(DEV), csrc.to(DEV), ctgt.to(DEV), mtgt.to(DEV), mean_shape, S, kp_gt=gt.to(DEV), vis=v2.to(DEV),
                                         counters=cnt["cam"])
        for m in ("flow", "cam"):
            k2k = out[m][0].cpu().numpy().astype(np.float32)
            # test_kp.py:253-258: err = |k2k - gt| * (1 + 2 * padding_frac) / 2, visible keypoints only, thresholds 0.1 / 0.15
            d = k2k - gt[:, :, :2].numpy()
            err = np.sqrt((d * d).sum(-1, dtype=np.float32), dtype=np.float32) * np.float32((1 + 2 * 0.05) / 2)
            seen = v2.numpy() != 0
            host[m][0] += seen.sum(0); host[m][1] += (seen & (err < 0.1)).sum(0); host[m][2] += (seen & (err < 0.15)).sum(0)
            near[m] += int((seen & ((np.abs(err - 0.1) < 1e-6) | (np.abs(err - 0.15) < 1e-6))).sum())
        if b == 0:      # (i) the first 50 pairs, both directions, against the restatement, pair by pair
            for p in list(range(50)) + list(range(BATCH, BATCH + 50)):
                rf = torch_ref.map_kp_flow(src[p], fsrc[p], ftgt[p], S, 3).numpy()
                np.testing.assert_allclose(out["flow"][0][p].cpu().numpy(), rf, atol=2e-6, err_msg="flow pair %d" % p)
                rc = torch_ref.map_kp_cam(src[p], csrc[p], ctgt[p], mtgt[p], mean_shape_c, S).numpy()
                np.testing.assert_allclose(out["cam"][0][p].cpu().numpy(), rc, atol=2e-6, err_msg="cam pair %d" % p)
    torch.cuda.synchronize()
    for m in ("flow", "cam"):
        c = cnt[m].counts.cpu().numpy().astype(np.int64)
        assert c[0].sum() > 100000 and (c[1] <= c[2]).all() and (c[2] <= c[0]).all()
        np.testing.assert_array_equal(c[0], host[m][0])                              # visible keypoints: exact
        assert np.abs(c[1:] - host[m][1:]).sum() <= near[m], (m, c, host[m], near[m])  # threshold counts: up to ties at a threshold
        p1, p15 = cnt[m].pck()
        assert 0.0 <= p1 <= p15 <= 1.0
