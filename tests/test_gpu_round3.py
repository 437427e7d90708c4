"""GPU parity tests added in round 3 (run with -m gpu): BASELINE config 4's shape (512^2 render = IS 1024, 5120 faces)
through the train_s2 render-and-compare module and the loss kernels at H = 512, regulariser / eval kernels / rotate_cam
goldens, capped super-block bins.

Every comparison is HIP (through the C ABI) against golden vectors written by the reference itself, the CPU oracle on the same
seeded inputs, tolerances are
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




def test_train_s2_step_at_config4_shape_vs_oracle(oracle_built):
    """BASELINE configs[3]: train_s2 at 512x512 renders (IS = 1024) of the 2562-vertex / 5120-face mesh (subdivide 4,
    experiments/train_s2.py:62, utils/mesh.py:37-41), K = 8 camera hypotheses, AlexNet perceptual texture term; B = 2
    images per step instead of 16 per GPU.  Every term of RenderCompareS2 and every gradient against the CPU restatement of
    train_s2.py:201-316 on the host cores: 20 raster forwards + 19 backwards at N = 8 | 1, F = 5120, IS = 1024, the IoU /
    part / chamfer / texture-sampling kernels at H = 512."""
    from oracle import softras, torch_ref
    from oracle.train_step_ref import RenderCompareS2Ref
    from umr_amd.synthetic import make_s2_inputs
    from umr_amd.train_step import RenderCompareS2
    K, H, B = 8, 512, 2
    nt = softras.max_threads()
    tv, faces, out_g, batch_g, ex = make_s2_inputs(B, K, H, 4, seed=41, device=DEV)
    assert tv.shape[0] == 2562 and faces.shape[0] == 5120
    torch.manual_seed(29)
    step = RenderCompareS2(tv.to(DEV), faces.to(DEV), ex["part_vertex_ids"], ex["uv_img"], ex["uv_sampler"], H, K,
                           texture_loss_type="perceptual").to(DEV)
    total, terms = step(out_g, batch_g)
    total.backward()
    tv, faces, out_c, batch_c, ex = make_s2_inputs(B, K, H, 4, seed=41, device="cpu")
    ptl = torch_ref.PerceptualTextureLoss(step.texture_loss_fn.pnet.state_dict())
    ref_total, ref_terms = RenderCompareS2Ref(tv, faces, ex["part_vertex_ids"], ex["uv_img"], ex["uv_sampler"], H, K,
                                              n_threads=nt, texture_loss=ptl)(out_c, batch_c)
    ref_total.backward()
    for k in ref_terms:
        assert abs(float(terms[k]) - float(ref_terms[k])) <= 3e-4 * max(1.0, abs(float(ref_terms[k]))), (k, float(terms[k]), float(ref_terms[k]))
    assert abs(float(total) - float(ref_total)) <= 3e-4 * max(1.0, abs(float(ref_total)))
    for k in ("delta_v", "cam_hypotheses", "cam_probs", "tex_flow"):
        r = out_c[k].grad.numpy()
        # every element; measured at B = 2 (round 4; a million-pixel sum per gradient value, float32 on both sides in different
        # orders): max error 3.8e-4 (vertices), 3.5e-4 (camera hypotheses), 2.5e-7 (probabilities), 4.7e-3 (texture flow) of the
        # largest gradient; B = 1 (round 3): 4.4e-5, 4.6e-6, 1.5e-7, 3.5e-4
        assert_close_frac(t2n(out_g[k].grad), r, atol={"tex_flow": 2e-2, "cam_probs": 1e-5}.get(k, 2e-3) * np.abs(r).max(), frac=1.0,
                          name="s2_cfg4_grad_" + k)


def test_loss_kernels_at_config4_resolution(oracle_built):
    """The image-space kernels at BASELINE configs[3]'s resolution (H = 512): barrier distance transform vs scipy (integer
    exact), silhouette IoU and its gradient, texture sampling and the texture-dt term vs the torch-CPU restatement."""
    from oracle import torch_ref
    from umr_amd import geom_utils, loss_utils
    from umr_amd.image_utils import compute_dt_barrier
    g = torch.Generator().manual_seed(5)
    H, B, F = 512, 2, 5120
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, H), indexing="ij")
    masks = torch.stack([((xx - 0.1) ** 2 + (yy + 0.05) ** 2 < 0.45).float(), ((xx.abs() < 0.6) & (yy.abs() < 0.3)).float()])
    dt, so, si = compute_dt_barrier(masks.to(DEV), return_squared=True)
    for b in range(B):                                                   # utils/image.py:130-141; scipy is the oracle
        ref, d_out, d_in = torch_ref.compute_dt_barrier(masks[b].numpy())
        assert np.array_equal(t2n(so[b]), np.rint(d_out ** 2).astype(np.int32))
        assert np.array_equal(t2n(si[b]), np.rint(d_in ** 2).astype(np.int32))
        np.testing.assert_allclose(t2n(dt[b]), ref, atol=1e-6)
    pred = torch.rand(B, H, H, generator=g)
    pg = pred.to(DEV).requires_grad_(True); pc = pred.clone().requires_grad_(True)
    lg = loss_utils.neg_iou_loss(pg, masks.to(DEV)); lc = torch_ref.neg_iou_loss(pc, masks)
    np.testing.assert_allclose(float(lg), float(lc), rtol=2e-6)
    lg.backward(); lc.backward()
    np.testing.assert_allclose(t2n(pg.grad), pc.grad.numpy(), atol=1e-12, rtol=2e-5)
    flow = torch.rand(B, F, 6, 6, 2, generator=g) * 2 - 1
    imgs = torch.rand(B, 3, H, H, generator=g)
    fg = flow.to(DEV).requires_grad_(True); fc = flow.clone().requires_grad_(True)
    tg = geom_utils.sample_textures(fg, imgs.to(DEV)); tc = torch_ref.sample_textures(fc, imgs)
    np.testing.assert_allclose(t2n(tg), tc.detach().numpy(), atol=2e-6)
    w = torch.rand(tc.shape, generator=g)
    (tg * w.to(DEV)).sum().backward(); (tc * w).sum().backward()
    r = fc.grad.numpy()
    assert_close_frac(t2n(fg.grad), r, atol=1e-5 * np.abs(r).max(), rtol=1e-4, frac=0.9999, name="cfg4_sample_textures_grad_flow")
    dts = torch.rand(B, 1, H, H, generator=g)
    fg2 = flow.to(DEV).requires_grad_(True); fc2 = flow.clone().requires_grad_(True)
    dg = loss_utils.texture_dt_loss(fg2, dts.to(DEV)); dc = torch_ref.texture_dt_loss(fc2, dts)
    np.testing.assert_allclose(float(dg), float(dc), rtol=1e-5)


def test_small_regularisers_and_masked_l1_vs_reference_goldens():
    """deform_l2reg / sym_reg (nnutils/loss_utils.py:118-126) and texture_loss_masks (:103-116) as HIP kernels (csrc/regs.hip)
    against values and gradients written by the reference's own functions (oracle/gen_golden.py, gen_golden_r3.py)."""
    from umr_amd import loss_utils as LU
    g = load_golden("loss_small_regs.npz")
    v = torch.from_numpy(g["v"]).to(DEV).requires_grad_(True)
    d, s = LU.deform_l2reg(v), LU.sym_reg(v)
    np.testing.assert_allclose(float(d), float(g["deform"]), rtol=2e-7)
    np.testing.assert_allclose(float(s), float(g["sym"]), rtol=2e-7)
    (d + 2 * s).backward()
    np.testing.assert_allclose(t2n(v.grad), g["grad_v"], atol=2e-9, rtol=1e-6)
    z = torch.zeros(2, 5, 3, device=DEV, requires_grad=True)              # zero rows / zero ordinates: zero (sub)gradient
    (LU.deform_l2reg(z) + LU.sym_reg(z)).backward()
    assert float(z.grad.abs().max()) == 0.0
    big = torch.randn(16, 642, 3, device=DEV, requires_grad=True)          # the train_s1 size: several partial-sum blocks
    ref = big.detach().cpu().clone().requires_grad_(True)
    (LU.deform_l2reg(big) + LU.sym_reg(big)).backward()
    (ref.view(-1, 3).norm(p=2, dim=1).mean() + ref[:, :, 1].abs().mean()).backward()
    np.testing.assert_allclose(t2n(big.grad), ref.grad.numpy(), atol=1e-9, rtol=1e-5)

    g = load_golden("loss_masked_l1.npz")
    ip = torch.from_numpy(g["img_pred"]).to(DEV).requires_grad_(True)
    mp = torch.from_numpy(g["mask_pred"]).to(DEV).requires_grad_(True)
    ig, mg = torch.from_numpy(g["img_gt"]).to(DEV), torch.from_numpy(g["mask_gt"]).to(DEV)
    la = LU.texture_loss_masks(ip, ig, mg, mp, avg=True)
    np.testing.assert_allclose(float(la), float(g["loss_avg"]), rtol=5e-7)
    la.backward()
    np.testing.assert_allclose(t2n(ip.grad), g["grad_img_pred_avg"], atol=1e-10, rtol=1e-6)
    np.testing.assert_allclose(t2n(mp.grad), g["grad_mask_pred_avg"], atol=1e-10, rtol=2e-6)
    ip.grad = None; mp.grad = None
    lp = LU.texture_loss_masks(ip, ig, mg, mp, avg=False)
    np.testing.assert_allclose(t2n(lp), g["loss_per_sample"], rtol=5e-7)
    (lp * torch.from_numpy(g["w"]).to(DEV)).sum().backward()
    np.testing.assert_allclose(t2n(ip.grad), g["grad_img_pred_w"], atol=1e-10, rtol=1e-6)
    np.testing.assert_allclose(t2n(mp.grad), g["grad_mask_pred_w"], atol=1e-10, rtol=2e-6)


def test_rotate_cam_vs_reference_golden():
    """geom_utils.rotate_cam (nnutils/geom_utils.py:167-193) for four axes and per-sample angles against the reference's own
    function run through the imported utils/transformations.py (golden rotate_cam.npz; cv2.Rodrigues restated in the
    generator).  The reference goes quaternion -> float64 matrix -> product -> quaternion_from_matrix(isprecise=True); the
    kernel multiplies quaternions: same rotation, same w >= 0 representative -- except where w ~ 0 (a half turn), where the
    representative's sign is rounding noise on both sides: compared up to sign there."""
    from umr_amd import geom_utils as GU
    from umr_amd.train_step import rotate_cam_y
    g = load_golden("rotate_cam.npz")
    cam, angles = torch.from_numpy(g["cam"]).to(DEV), torch.from_numpy(g["angles"]).to(DEV)
    for name in ("y", "x", "z", "d"):
        got = t2n(GU.rotate_cam(cam, angles, axis=[float(v) for v in g["axis_" + name]]))
        want = g["new_cam_" + name]
        np.testing.assert_array_equal(got[:, :3], want[:, :3])                       # scale / translation pass through
        flip = np.where((np.abs(want[:, 3]) < 1e-6) & ((got[:, 4:] * want[:, 4:]).sum(1) < 0), -1.0, 1.0)[:, None]
        np.testing.assert_allclose(got[:, 3:] * flip, want[:, 3:], atol=2e-6)
        assert (got[:, 3] >= 0).all()
    np.testing.assert_allclose(t2n(rotate_cam_y(cam, angles)), t2n(GU.rotate_cam(cam, angles, axis=[0, 1, 0])), atol=1e-7)


def test_keypoint_transfer_vs_reference_golden():
    """The evaluation kernels of csrc/eval.hip against golden vectors written by the reference's own utils/kp_utils.py
    (create_grid, draw_labelmap), nnutils/chamfer_python.py and nnutils/smr.py driven exactly as test_kp.py:125-193 drives
    them (oracle/gen_golden_r3.py): flow mode -- per-face heat-map responses, the arg-max FACE of every keypoint (exact),
    the transferred points; cam mode -- nearest VERTEX of every keypoint (exact), the transferred points; the PCK of
    test_kp.py:253-258, 317-323 from the device counters."""
    from umr_amd import eval_utils as EU
    g = load_golden("eval_kp.npz")
    kps = torch.from_numpy(g["kps"]).to(DEV)                     # [P,2,K,3]
    flows = torch.from_numpy(g["flows"].astype(np.float32)).to(DEV)
    cams, masks = torch.from_numpy(g["cams"]).to(DEV), torch.from_numpy(g["masks"].astype(np.float32)).to(DEV)
    mean_shape = torch.from_numpy(g["mean_shape"]).to(DEV)
    S, sigma = int(g["image_size"]), int(g["sigma"])
    P, K = kps.shape[0], kps.shape[2]
    vis = (kps[:, 0, :, 2] * kps[:, 1, :, 2])
    # entries: direction 1 -> 2 of every pair, then 2 -> 1; ground truth = the target image's keypoints
    src = torch.cat([kps[:, 0], kps[:, 1]]); gt = torch.cat([kps[:, 1], kps[:, 0]]); v2 = torch.cat([vis, vis])
    cnt = EU.PCKCounters(K, DEV)
    k2k, face = EU.map_kp_flow_batch(src, torch.cat([flows[:, 0], flows[:, 1]]), torch.cat([flows[:, 1], flows[:, 0]]), S, sigma,
                                     kp_gt=gt, vis=v2, counters=cnt)
    np.testing.assert_array_equal(t2n(face[:P]), g["flow_face_12"])
    np.testing.assert_array_equal(t2n(face[P:]), g["flow_face_21"])
    np.testing.assert_allclose(t2n(k2k[:P]), g["flow_k1_to_k2"], atol=1e-6)
    np.testing.assert_allclose(t2n(k2k[P:]), g["flow_k2_to_k1"], atol=1e-6)
    p1, p15 = cnt.pck()
    # (the fixture stores float32)
    assert abs(p1 - float(g["pck1"])) < 1e-7 and abs(p15 - float(g["pck15"])) < 1e-7, (p1, p15, float(g["pck1"]), float(g["pck15"]))
    # cam mode
    k2c, vert = EU.map_kp_cam_batch(src, torch.cat([cams[:, 0], cams[:, 1]]), torch.cat([cams[:, 1], cams[:, 0]]),
                                    torch.cat([masks[:, 1], masks[:, 0]]), mean_shape, S)
    np.testing.assert_array_equal(t2n(vert[:P]), g["cam_vert_12"])
    np.testing.assert_array_equal(t2n(vert[P:]), g["cam_vert_21"])
    np.testing.assert_allclose(t2n(k2c[:P]), g["cam_k1_to_k2"], atol=1e-6)
    np.testing.assert_allclose(t2n(k2c[P:]), g["cam_k2_to_k1"], atol=1e-6)


def test_loss_and_geometry_operators_opcheck_and_match_the_function_path():
    """torch.ops.umr.* of umr_amd/ops_losses.py on the device: torch.library.opcheck (schema, fake kernel against the real one,
    autograd registration) on real inputs, and values / gradients identical to the autograd.Function route the loss modules
    use (same kernels, same stream: bit-equal)."""
    from torch.library import opcheck
    from umr_amd import functional as UF, ops_losses  # noqa: F401
    from umr_amd.loss_utils import LaplacianLoss, FlattenLoss
    g = torch.Generator().manual_seed(9)
    utils = ("test_schema", "test_faketensor", "test_autograd_registration")
    verts, faces, cams, _ = scene(2, 1, seed=6)
    v = verts.to(DEV).requires_grad_(True); c = cams.to(DEV).requires_grad_(True); fi = faces.int().to(DEV)
    opcheck(torch.ops.umr.project_faces.default, (v, c, fi, 5.0, -2.732), test_utils=utils)
    opcheck(torch.ops.umr.project_points.default, (v, c, 2, 0.0), test_utils=utils)
    fo = torch.ops.umr.project_faces(v, c, fi, 5.0, -2.732)
    w = torch.rand(fo.shape, generator=g).to(DEV)
    (fo * w).sum().backward()
    gv, gc = v.grad.clone(), c.grad.clone(); v.grad = None; c.grad = None
    _, fo2, _ = UF.ProjectFacesFunction.apply(v, c, fi, 5.0, -2.732, False)
    (fo2 * w).sum().backward()
    assert torch.equal(fo, fo2)
    # backward: the face gradients are scattered onto the vertices with float atomics, and the camera gradient is reduced
    # from those sums: equal up to summation order
    assert float((gv - v.grad).abs().max()) <= 1e-5 * float(gv.abs().max())
    assert float((gc - c.grad).abs().max()) <= 1e-5 * float(gc.abs().max())
    p, t = torch.rand(3, 32, 32, generator=g).to(DEV).requires_grad_(True), (torch.rand(3, 32, 32, generator=g) > 0.5).float().to(DEV)
    opcheck(torch.ops.umr.neg_iou.default, (p, t), test_utils=utils)
    l1 = torch.ops.umr.neg_iou(p, t)[0]; l1.sum().backward(); g1 = p.grad.clone(); p.grad = None
    l2 = UF.NegIoUFunction.apply(p, t); l2.sum().backward()
    assert torch.equal(l1, l2) and torch.equal(g1, p.grad)
    a, b = torch.rand(2, 9, 2, generator=g).to(DEV).requires_grad_(True), torch.rand(2, 13, 2, generator=g).to(DEV).requires_grad_(True)
    opcheck(torch.ops.umr.chamfer.default, (a, b), test_utils=utils)
    d1, d2, i1, i2 = torch.ops.umr.chamfer(a, b); (d1.sum() + 2 * d2.sum()).backward(); ga = a.grad.clone(); a.grad = None; b.grad = None
    e1, e2, j1, j2 = UF.ChamferFunction.apply(a, b); (e1.sum() + 2 * e2.sum()).backward()
    assert torch.equal(d1, e1) and torch.equal(i2, j2) and float((ga - a.grad).abs().max()) <= 1e-6
    img, grid = torch.rand(2, 3, 16, 16, generator=g).to(DEV).requires_grad_(True), (torch.rand(2, 20, 2, generator=g) * 2 - 1).to(DEV).requires_grad_(True)
    opcheck(torch.ops.umr.grid_sample_cl.default, (img, grid), test_utils=utils)
    assert torch.equal(torch.ops.umr.grid_sample_cl(img, grid), UF.GridSampleCLFunction.apply(img, grid))
    lap, fl = LaplacianLoss(verts[0], faces[0].int()).to(DEV), FlattenLoss(faces[0].int()).to(DEV)
    x = verts.to(DEV).requires_grad_(True)
    opcheck(torch.ops.umr.flatten.default, (x, fl.quads), test_utils=utils)
    assert torch.equal(torch.ops.umr.flatten(x, fl.quads), fl(x))
    opcheck(torch.ops.umr.laplacian.default, (x, lap.nbr_off, lap.nbr_idx), test_utils=utils)
    assert torch.equal(torch.ops.umr.laplacian(x, lap.nbr_off, lap.nbr_idx)[0], lap(x))
    f0 = [torch.rand(2, 8, 6, 6, generator=g).to(DEV).requires_grad_(True), torch.rand(2, 4, 3, 3, generator=g).to(DEV).requires_grad_(True)]
    f1 = [torch.rand(2, 8, 6, 6, generator=g).to(DEV), torch.rand(2, 4, 3, 3, generator=g).to(DEV)]
    opcheck(torch.ops.umr.cos_sim.default, (f0, f1, 1e-10), test_utils=utils)
    val = torch.ops.umr.cos_sim(f0, f1, 1e-10)[0]; val.sum().backward(); gc0 = f0[0].grad.clone(); f0[0].grad = None; f0[1].grad = None
    val2 = UF.CosSimDistanceFunction.apply(1e-10, *f0, *f1); val2.sum().backward()
    assert torch.equal(val, val2) and torch.equal(gc0, f0[0].grad)
    ra, rb = torch.rand(2, 4, 16, 16, generator=g).to(DEV).requires_grad_(True), torch.rand(2, 4, 16, 16, generator=g).to(DEV).requires_grad_(True)
    q = torch.randn(2, 5, 16, 16, generator=g).to(DEV)
    opcheck(torch.ops.umr.part_match.default, (ra, rb, q, [0., 5., 0., 0., 5.], 0.1, 1e-3), test_utils=utils)
    e, l, _ = torch.ops.umr.part_match(ra, rb, q, [0., 5., 0., 0., 5.], 0.1, 1e-3)
    e2, l2 = UF.PartMatchFunction.apply(ra, rb, q, [0., 5., 0., 0., 5.], 0.1, 1e-3)
    assert torch.equal(e, e2) and torch.equal(l, l2)
    m = (torch.rand(2, 32, 32, generator=g) > 0.6).float().to(DEV)
    opcheck(torch.ops.umr.dt_barrier.default, (m, 50.0), test_utils=("test_schema", "test_faketensor"))
    xr = torch.randn(3, 7, 3, generator=g).to(DEV).requires_grad_(True)
    opcheck(torch.ops.umr.row_norm_mean.default, (xr,), test_utils=utils)
    opcheck(torch.ops.umr.abs_column_mean.default, (xr, 1), test_utils=utils)
    ip, ig = torch.rand(2, 3, 8, 8, generator=g).to(DEV).requires_grad_(True), torch.rand(2, 3, 8, 8, generator=g).to(DEV)
    mg, mp = (torch.rand(2, 8, 8, generator=g) > 0.5).float().to(DEV), torch.rand(2, 8, 8, generator=g).to(DEV).requires_grad_(True)
    opcheck(torch.ops.umr.masked_l1.default, (ip, ig, mg, mp), test_utils=utils)
    assert torch.equal(torch.ops.umr.masked_l1(ip, ig, mg, mp), UF.MaskedL1Function.apply(ip, ig, mg, mp))


def test_nearest_edge_choice_is_the_references_at_full_size(oracle_built):
    """Inside a triangle the reference keeps the edge line with the smallest COMPUTED distance (:78-107).  The default build
    (umr_debug_set("exact_edges", 1)) evaluates all three lines the reference's way wherever the choice can matter and be in doubt
    (HISTORY.md 4.4), so at BASELINE size (2 x 1280 faces x 512^2) the render and its gradients agree with the oracle in EVERY
    element; so does the brute-force switch "thin_face_h_1e6" = 1e9 (every inside lane); "exact_edges" = 0 (the fast pick, 8-15 %
    less kernel time) holds the same bounds with isolated outliers."""
    import math
    from oracle import softras, torch_ref
    from umr_amd import _lib, functional as UF
    verts, faces, cams, gen = scene(2, 3, seed=5)
    proj = torch_ref.orthographic_proj_withz(verts, cams, 5.) * torch.tensor([1., -1., 1.])
    fv = torch_ref.face_vertices(torch_ref.look_at_ortho(proj), faces).contiguous()
    tex = torch.rand(2, 1280, 36, 3, generator=gen)
    gsc = torch.randn(2, 4, 512, 512, generator=gen)
    cfg = dict(near=1., far=100., eps=1e-3, sigma_val=1e-5, dist_eps_log=float(math.log(1e10 - 1.)), gamma_val=1e-4, func_id_rgb=1,
               double_side=True)
    nt = softras.max_threads()
    o = softras.raster_forward(fv.numpy(), tex.numpy(), 512, backend="port", n_threads=nt, **cfg)
    gf, gt = softras.raster_backward(o["faces"], o["textures"], o["soft_colors"], o["faces_info"], o["aggrs_info"], gsc.numpy(), 512,
                                     backend="port", n_threads=nt, **cfg)
    sf, st = np.abs(gf).max(), np.abs(gt).max()
    for mode, sets in (("default", ()), ("every_inside_lane", (("thin_face_h_1e6", 1000000000),)), ("fast_pick", (("exact_edges", 0),))):
        for k, v in sets:
            _lib.debug_set(k, v)
        try:
            fvd, texd = fv.to(DEV).requires_grad_(True), tex.to(DEV).requires_grad_(True)
            sc, _, _ = UF.soft_rasterize(fvd, texd, 512, [0, 0, 0], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, 'softmax', 'prod', 'surface')
            sc.backward(gsc.to(DEV))
            sc, g_f, g_t = t2n(sc), t2n(fvd.grad).reshape(gf.shape), t2n(texd.grad)
        finally:
            _lib.debug_set("thin_face_h_1e6", -1)
            _lib.debug_set("exact_edges", 1)
        if mode != "fast_pick":
            # measured on the MI355X: alpha max |err| 2.4e-7, colours 8.0e-7 (2.1 M values), vertex gradients 2.8e-3 absolute at a
            # scale of 4924 = 5.7e-7 of scale (23 040 values), texel gradients 6.1e-6 at a scale of 22 -- bounds ~10x that, EVERY element
            assert_close_frac(sc, o["soft_colors"], atol=1e-4, frac=1.0, max_outlier=1e-5, name=mode + "_soft_colors")
            assert_close_frac(g_f, gf, atol=1e-5 * sf, rtol=1e-4, frac=1.0, name=mode + "_grad_faces")
            assert_close_frac(g_t, gt, atol=3e-6 * st, rtol=1e-4, frac=1.0, name=mode + "_grad_textures")
        else:
            # measured: alpha max 3.0e-5, and 5 of the 23 040 gradient values off by up to 1.4 % of scale (nearest-edge ties inside
            # faces of 4 - 16 px)
            assert_close_frac(sc[:, 3], o["soft_colors"][:, 3], atol=1e-4, frac=1.0, name=mode + "_alpha")
            assert_close_frac(g_f, gf, atol=1e-4 * sf, rtol=5e-3, frac=0.997, max_outlier=6e-2 * sf, name=mode + "_grad_faces")
