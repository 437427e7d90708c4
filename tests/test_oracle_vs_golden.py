"""CPU: the C restatement (oracle/softras_oracle.c) and the torch restatement (oracle/torch_ref.py)
against golden vectors produced by the reference itself (oracle/gen_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, close_frac

RASTER = ["raster_softmax_ts36.npz", "raster_softmax_ts1.npz", "raster_hard_ts1.npz", "raster_hard_ts4.npz"]


def _cfg(g):
    return dict(near=float(g["near"]), far=float(g["far"]), eps=float(g["eps"]), sigma_val=float(g["sigma_val"]),
                dist_eps_log=float(g["dist_eps_log"]), gamma_val=float(g["gamma_val"]),
                func_id_rgb=int(g["func_id_rgb"]), double_side=bool(g["double_side"]))


@pytest.mark.parametrize("name", RASTER)
@pytest.mark.parametrize("threads", [1, 4])
def test_raster_port_matches_reference_golden(oracle_built, name, threads):
    from oracle import softras
    g = load_golden(name)
    o = softras.raster_forward(g["faces"], g["textures"], int(g["image_size"]), background=tuple(g["background"]),
                               backend="port", n_threads=threads, **_cfg(g))
    # the restatement reproduces the reference's promotion pattern -> expect (near) bit equality
    np.testing.assert_allclose(o["faces_info"], g["faces_info"], rtol=0, atol=0)
    np.testing.assert_allclose(o["soft_colors"], g["soft_colors"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(o["aggrs_info"], g["aggrs_info"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(o["p2f_info"], g["p2f_info"], rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(o["p2f_sum"], g["p2f_sum"], rtol=2e-5, atol=1e-4)
    gf, gt = softras.raster_backward(g["faces"], g["textures"], g["soft_colors"], g["faces_info"], g["aggrs_info"],
                                     g["grad_soft_colors"], int(g["image_size"]), backend="port",
                                     n_threads=threads, **_cfg(g))
    scale = np.abs(g["grad_faces"]).max()
    np.testing.assert_allclose(gf, g["grad_faces"], rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(gt, g["grad_textures"], rtol=1e-4, atol=1e-5)


def test_raster_port_single_thread_is_bit_exact(oracle_built):
    from oracle import softras
    g = load_golden("raster_softmax_ts36.npz")
    o = softras.raster_forward(g["faces"], g["textures"], int(g["image_size"]), background=tuple(g["background"]),
                               backend="port", n_threads=1, **_cfg(g))
    for k in ("faces_info", "soft_colors", "aggrs_info", "p2f_info", "p2f_sum"):
        assert np.array_equal(o[k], g[k]), k
    gf, gt = softras.raster_backward(g["faces"], g["textures"], g["soft_colors"], g["faces_info"], g["aggrs_info"],
                                     g["grad_soft_colors"], int(g["image_size"]), backend="port", n_threads=1,
                                     **_cfg(g))
    assert np.array_equal(gf, g["grad_faces"])
    assert np.array_equal(gt, g["grad_textures"])


def test_unsupported_modes_rejected(oracle_built):
    """vertex textures index w[j] for j < texture_size (:215): only texture_size == 3 is defined; unknown ids refused"""
    from oracle import softras
    g = load_golden("raster_softmax_ts1.npz")
    with pytest.raises(RuntimeError):
        softras.raster_forward(g["faces"], g["textures"], 16, texture_sample_type=1, **_cfg(g))
    with pytest.raises(RuntimeError):
        softras.raster_forward(g["faces"], g["textures"], 16, func_id_dist=3, **_cfg(g))


def mode_cases():
    g = load_golden("raster_modes.npz")
    return [str(c) for c in g["cases"]]


@pytest.mark.parametrize("case", mode_cases())
def test_raster_port_other_modes_bit_exact(oracle_built, case):
    """The mode ids UMR does not select (hard / barycentric distance, hard / sum alpha, vertex textures,
    soft_rasterize_cuda_kernel.cu:154-218, :365-398, :444-447, :577-583, :634-636): the single-thread restatement
    reproduces the reference kernels' outputs and gradients to the bit."""
    from oracle import softras
    g = load_golden("raster_modes.npz")
    fd, fa, fr, tt = [int(v) for v in g[case + "/modes"]]
    cfg = dict(near=float(g["near"]), far=float(g["far"]), eps=float(g["eps"]), sigma_val=float(g[case + "/sigma_val"]),
               dist_eps_log=float(g["dist_eps_log"]), gamma_val=float(g["gamma_val"]), func_id_rgb=fr, double_side=True,
               func_id_dist=fd, func_id_alpha=fa, texture_sample_type=tt)
    IS = int(g["image_size"])
    o = softras.raster_forward(g["faces"], g[case + "/textures"], IS, background=tuple(g["background"]), backend="port",
                               n_threads=1, **cfg)
    for k in ("soft_colors", "aggrs_info", "p2f_info", "p2f_sum"):
        assert np.array_equal(o[k], g[case + "/" + k]), k
    gf, gt = softras.raster_backward(g["faces"], g[case + "/textures"], o["soft_colors"], o["faces_info"], o["aggrs_info"],
                                     g["grad_soft_colors"], IS, backend="port", n_threads=1, **cfg)
    assert np.array_equal(gf, g[case + "/grad_faces"]) and np.array_equal(gt, g[case + "/grad_textures"])


@pytest.mark.parametrize("name", ["smr_mask_default_light.npz", "smr_tex_ambient.npz", "smr_tex_default_light.npz",
                                  "smr_hard_default_light.npz"])
def test_smr_renderer_restatement(oracle_built, name):
    from oracle import torch_ref
    g = load_golden(name)
    verts = torch.from_numpy(g["verts"]).requires_grad_(True)
    cams = torch.from_numpy(g["cams"]).requires_grad_(True)
    faces = torch.from_numpy(g["faces"])
    tex = torch.from_numpy(g["textures"]).requires_grad_(True) if "textures" in g else None
    r = torch_ref.SoftRenderer(int(g["img_size"]), str(g["render_type"]))
    if bool(g["ambient_only"]):
        r.ambient_light_only()
    imgs, p2f, aggr = r.forward(verts, faces, cams, tex)
    np.testing.assert_allclose(imgs.detach().numpy(), g["imgs"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(p2f.detach().numpy(), g["p2f"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(aggr.detach().numpy(), g["aggr"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(r.project_points(verts, cams).detach().numpy(), g["proj_points"], atol=1e-6)
    imgs.backward(torch.from_numpy(g["grad_imgs"]))
    for got, key in ((verts.grad, "grad_verts"), (cams.grad, "grad_cams")):
        ref = g[key]
        np.testing.assert_allclose(got.numpy(), ref, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(ref).max()))
    if tex is not None:
        np.testing.assert_allclose(tex.grad.numpy(), g["grad_textures"], rtol=1e-4, atol=1e-6)


def test_multimask_loss_restatement(oracle_built):
    from oracle import torch_ref
    g = load_golden("loss_multimask.npz")
    verts = torch.from_numpy(g["verts"]).requires_grad_(True)
    cams = torch.from_numpy(g["cams_all_hypo"]).requires_grad_(True)
    probs = torch.from_numpy(g["cam_probs"]).requires_grad_(True)
    r = torch_ref.SoftRenderer(int(g["image_size"]), "softmax")
    loss, masks = torch_ref.multi_mask_loss(r, verts, torch.from_numpy(g["faces"]), cams, probs,
                                            torch.from_numpy(g["masks_gt"]), int(g["num_hypo_cams"]),
                                            int(g["image_size"]))
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    np.testing.assert_allclose(masks.detach().numpy(), g["mask_all_hypo"], atol=2e-6)
    loss.backward()
    np.testing.assert_allclose(verts.grad.numpy(), g["grad_verts"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(cams.grad.numpy(), g["grad_cams"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(probs.grad.numpy(), g["grad_probs"], rtol=1e-5, atol=1e-7)


def test_small_losses_restatement():
    from oracle import torch_ref
    g = load_golden("loss_neg_iou.npz")
    p = torch.from_numpy(g["predict"]).requires_grad_(True)
    t = torch.from_numpy(g["target"])
    l1, l2 = torch_ref.neg_iou_loss(p, t), torch_ref.neg_iou_loss(p, t, avg=False)
    (l1 + (l2 * torch.tensor([1., 2., 3.])).sum()).backward()
    np.testing.assert_allclose(l1.item(), g["loss_avg"], atol=1e-7)
    np.testing.assert_allclose(l2.detach().numpy(), g["loss_per"], atol=1e-7)
    np.testing.assert_allclose(p.grad.numpy(), g["grad_predict"], atol=1e-8, rtol=1e-5)

    g = load_golden("loss_texture_sampling.npz")
    flow = torch.from_numpy(g["flow"]).requires_grad_(True)
    images = torch.from_numpy(g["images"]).requires_grad_(True)
    tex = torch_ref.sample_textures(flow, images)
    np.testing.assert_allclose(tex.detach().numpy(), g["tex"], atol=1e-6)
    tex.backward(torch.from_numpy(g["grad_tex"]))
    np.testing.assert_allclose(flow.grad.numpy(), g["grad_flow_from_tex"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(images.grad.numpy(), g["grad_images"], atol=1e-5, rtol=1e-5)
    flow.grad = None
    dt = torch_ref.texture_dt_loss(flow, torch.from_numpy(g["dts"]))
    dt.backward()
    np.testing.assert_allclose(dt.item(), g["dt_loss"], atol=1e-7)
    np.testing.assert_allclose(flow.grad.numpy(), g["grad_flow_from_dt"], atol=1e-8, rtol=1e-4)

    g = load_golden("loss_texcycle.npz")
    flow = torch.from_numpy(g["flow"]).requires_grad_(True)
    l, avg10 = torch_ref.tex_cycle(flow, torch.from_numpy(g["p2f_soft"]), torch.from_numpy(g["face_ids"]))
    l.backward()
    np.testing.assert_allclose(l.item(), g["loss"], atol=1e-8)
    np.testing.assert_allclose(flow.grad.numpy(), g["grad_flow"], atol=1e-9, rtol=1e-5)
    lh, _ = torch_ref.tex_cycle(flow, torch.from_numpy(g["p2f_hard"]), torch.from_numpy(g["face_ids"]))
    np.testing.assert_allclose(lh.item(), g["loss_hard_target"], atol=1e-8)
    assert np.abs(g["p2f_hard"]).sum() == 0  # hard renderer never writes p2f
    assert g["face_ids"].min() == -1         # quirk 2: background id -1 indexes the last face

    g = load_golden("loss_small_regs.npz")
    v = torch.from_numpy(g["v"]).requires_grad_(True)
    (torch_ref.deform_l2reg(v) + 2 * torch_ref.sym_reg(v)).backward()
    np.testing.assert_allclose(torch_ref.deform_l2reg(v).item(), g["deform"], atol=1e-6)
    np.testing.assert_allclose(torch_ref.sym_reg(v).item(), g["sym"], atol=1e-6)
    np.testing.assert_allclose(v.grad.numpy(), g["grad_v"], atol=1e-7)


def test_chamfer_restatement():
    from oracle import torch_ref
    g = load_golden("chamfer.npz")
    for i in range(int(g["n_cases"])):
        a = torch.from_numpy(g["a%d" % i]).requires_grad_(True)
        b = torch.from_numpy(g["b%d" % i]).requires_grad_(True)
        d1, d2, i1, i2 = torch_ref.dist_chamfer(a, b)
        np.testing.assert_allclose(d1.detach().numpy(), g["d1_%d" % i], atol=1e-6)
        np.testing.assert_allclose(d2.detach().numpy(), g["d2_%d" % i], atol=1e-6)
        assert np.array_equal(i1.numpy(), g["i1_%d" % i]) and np.array_equal(i2.numpy(), g["i2_%d" % i])
        (d1.sum() + 0.5 * d2.sum()).backward()
        np.testing.assert_allclose(a.grad.numpy(), g["ga%d" % i], atol=1e-5)
        np.testing.assert_allclose(b.grad.numpy(), g["gb%d" % i], atol=1e-5)


def test_mesh_regs_restatement():
    from oracle import torch_ref
    g = load_golden("mesh_regs.npz")
    vt, ft = torch.from_numpy(g["verts0"]), torch.from_numpy(g["faces"]).int()
    lap, flat = torch_ref.LaplacianLoss(vt, ft), torch_ref.FlattenLoss(ft)
    assert np.array_equal(lap.laplacian.numpy(), g["lap_matrix"])
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    ll, fl = lap(x), flat(x)
    np.testing.assert_allclose(ll.detach().numpy(), g["laplacian"], rtol=1e-5)
    np.testing.assert_allclose(fl.detach().numpy(), g["flatten"], rtol=1e-5)
    (ll.sum() + fl.sum()).backward()
    np.testing.assert_allclose(x.grad.numpy(), g["grad_x"], rtol=1e-3, atol=1e-4)


def test_parts_and_cossim_restatement():
    from oracle import torch_ref
    g = load_golden("parts_and_cossim.npz")
    pm = torch.from_numpy(g["part_maps"]).requires_grad_(True)
    cen = torch_ref.batch_get_centers(pm[:, 1:])
    np.testing.assert_allclose(cen.detach().numpy(), g["centers"], atol=1e-6)
    cen.backward(torch.ones_like(cen))
    np.testing.assert_allclose(pm.grad.numpy(), g["grad_part_maps"], atol=1e-7, rtol=1e-4)
    f0 = [torch.from_numpy(g["f0_0"]), torch.from_numpy(g["f0_1"])]
    f1 = [torch.from_numpy(g["f1_0"]), torch.from_numpy(g["f1_1"])]
    np.testing.assert_allclose(torch_ref.cos_sim_distance(f0, f1).numpy(), g["cos_dist"], atol=1e-6)


def test_part_matching_and_cos_sim_gradients_restatement():
    """oracle restatements of part_matching_loss's reductions (loss_utils.py:399-440) and of the PNet head
    (networks_basic.py:50-58 + util.py:71-83) against values AND gradients produced by the reference's own code
    (oracle/gen_golden.py --only part_cos)."""
    from oracle import torch_ref
    g = load_golden("part_loss_and_cos_grads.npz")
    planes = torch.from_numpy(g["planes"]).requires_grad_(True)
    parts = torch.from_numpy(g["part_segs"])
    loss = torch_ref.part_matching_core([planes[:, k:k + 1] for k in range(4)], parts)
    np.testing.assert_allclose(float(loss), float(g["loss_avg"]), rtol=1e-6)
    loss.backward()
    np.testing.assert_allclose(planes.grad.numpy(), g["grad_avg"], atol=1e-9, rtol=1e-4)
    f0 = [torch.from_numpy(g["cf0_%d" % k]).requires_grad_(True) for k in range(3)]
    f1 = [torch.from_numpy(g["cf1_%d" % k]).requires_grad_(True) for k in range(3)]
    val = torch_ref.cos_sim_distance(f0, f1)
    np.testing.assert_allclose(val.detach().numpy(), g["cos_val"], atol=1e-6)
    (val * torch.from_numpy(g["cos_gv"])).sum().backward()
    for k in range(3):
        np.testing.assert_allclose(f0[k].grad.numpy(), g["cg0_%d" % k], atol=1e-7, rtol=1e-4)
        np.testing.assert_allclose(f1[k].grad.numpy(), g["cg1_%d" % k], atol=1e-7, rtol=1e-4)


def test_perceptual_texture_loss_restatement_consistency():
    """oracle.torch_ref.PerceptualTextureLoss (the CPU side of the step-level parity tests) against a second formulation:
    nn.Module AlexNet taps (umr_amd.perceptual.AlexNetFeatures evaluated on CPU tensors by torch itself -- no HIP code
    involved) + the golden-pinned cos_sim_distance.  Pins the weight-key mapping and the shift / scale / 2x-1 prologue."""
    from oracle import torch_ref
    from umr_amd.perceptual import PNet
    torch.manual_seed(3)
    net = PNet()
    ptl = torch_ref.PerceptualTextureLoss(net.state_dict())
    gen = torch.Generator().manual_seed(5)
    pred, gt = torch.rand(2, 3, 64, 64, generator=gen), torch.rand(2, 3, 64, 64, generator=gen)
    m_gt, m_pr = (torch.rand(2, 64, 64, generator=gen) > 0.4).float(), torch.rand(2, 64, 64, generator=gen)
    got = ptl(pred, gt, m_gt, m_pr, avg=False)
    with torch.no_grad():
        in0, in1 = 2 * (gt * m_gt[:, None]) - 1, 2 * (pred * m_pr[:, None]) - 1
        f0 = net.net((in0 - net.shift) / net.scale)
        f1 = net.net((in1 - net.shift) / net.scale)
    np.testing.assert_allclose(got.numpy(), torch_ref.cos_sim_distance(f0, f1).numpy(), atol=1e-6)
    assert got.shape == (2,) and float(got.min()) > 0


def test_texture_atlas_and_obj_text_vs_reference_golden(oracle_built):
    """save_obj golden produced by the reference's own functional/save_obj.py + its atlas kernel body."""
    from oracle import softras as S
    g = load_golden("save_obj.npz")
    img, uv = S.create_texture_image(g["textures"], 16)
    assert np.array_equal((img.clip(0, 1) * 255).astype("uint8"), g["png"])
    a7, uv7 = S.create_texture_image(g["tex7"], 8)              # ragged grid: 7 faces in 3x3 cells, 2 cells stay 1.0
    assert np.array_equal(a7, g["atlas7"]) and np.array_equal(uv7, g["uv7"])
    assert S.obj_text("bird.obj", g["verts"], g["faces"], uv) == g["obj_textured"].tobytes().decode()
    assert S.obj_text("plain.obj", g["verts"], g["faces"]) == g["obj_plain"].tobytes().decode()
    if S.have_backend("ref"):
        b, _ = S.create_texture_image(g["textures"], 16, backend="ref")
        assert np.array_equal(img, b)


def test_reference_algorithm_is_immune_to_collapsed_edges(oracle_built):
    """An edge seen end-on: `den` of soft_rasterize_cuda_kernel.cu:86 is exactly 0 (fp32 cancellation in :82-84), the edge
    parameter inf / NaN.  The reference's inside branch never selects such an edge (`dis < dis_min` is false for NaN / inf),
    its outside branch clamps: the restatement -- and the reference kernels themselves where they were built -- render the
    crafted faces of tests/test_gpu_zz_collapsed_edges.py without a single non-finite value.  (The GPU test holds the
    product kernels to the same.)"""
    import os
    from oracle import softras
    from test_gpu_zz_collapsed_edges import collapsed_edge_faces, _den_collapsed
    fv, _ = collapsed_edge_faces(64, 64)
    assert all(_den_collapsed(f[0, :2], f[1, :2], f[2, :2]) == 0 for f in fv)
    tex = np.random.default_rng(1).random((1, 64, 4, 3), dtype=np.float32)
    cfg = dict(near=1., far=100., eps=1e-3, sigma_val=1e-5, dist_eps_log=float(np.log(1. / 1e-10 - 1.)), gamma_val=1e-4,
               func_id_rgb=1, double_side=True)
    backends = ["port"] + (["ref"] if os.path.exists(os.path.join(os.path.dirname(softras.__file__), "_ref", "libsoftras_ref.so")) else [])
    outs = []
    for be in backends:
        o = softras.raster_forward(fv.reshape(1, 64, 9), tex, 64, background=(0.1, 0.2, 0.3), backend=be, **cfg)
        assert np.isfinite(o["soft_colors"]).all() and np.isfinite(o["aggrs_info"]).all(), be
        gsc = np.ones((1, 4, 64, 64), np.float32)
        gf, gt = softras.raster_backward(o["faces"], o["textures"], o["soft_colors"], o["faces_info"], o["aggrs_info"], gsc, 64,
                                         backend=be, **cfg)
        outs.append(o["soft_colors"])
        assert np.isfinite(gt).all(), be      # (grad_faces may legitimately be huge for needles; texel gradients are weights)
    if len(outs) == 2:
        assert np.array_equal(outs[0], outs[1])


def test_eval_restatement_vs_reference_golden():
    """oracle/torch_ref.map_kp_flow / map_kp_cam (restatements of experiments/test_kp.py:125-193) pinned on the golden written
    by the reference's own kp_utils / chamfer_python / smr (oracle/gen_golden_r3.py): transferred keypoints of every pair in
    both directions, flow and cam mode."""
    from oracle import torch_ref
    g = load_golden("eval_kp.npz")
    kps = torch.from_numpy(g["kps"]); flows = torch.from_numpy(g["flows"].astype(np.float32))
    cams = torch.from_numpy(g["cams"]); masks = torch.from_numpy(g["masks"].astype(np.float32)); ms = torch.from_numpy(g["mean_shape"])
    S, sigma = int(g["image_size"]), int(g["sigma"])
    for p in range(kps.shape[0]):
        a = torch_ref.map_kp_flow(kps[p, 0], flows[p, 0], flows[p, 1], S, sigma)
        b = torch_ref.map_kp_flow(kps[p, 1], flows[p, 1], flows[p, 0], S, sigma)
        np.testing.assert_allclose(a.numpy(), g["flow_k1_to_k2"][p], atol=1e-6)
        np.testing.assert_allclose(b.numpy(), g["flow_k2_to_k1"][p], atol=1e-6)
        c = torch_ref.map_kp_cam(kps[p, 0], cams[p, 0], cams[p, 1], masks[p, 1], ms, S)
        d = torch_ref.map_kp_cam(kps[p, 1], cams[p, 1], cams[p, 0], masks[p, 0], ms, S)
        np.testing.assert_allclose(c.numpy(), g["cam_k1_to_k2"][p], atol=1e-6)
        np.testing.assert_allclose(d.numpy(), g["cam_k2_to_k1"][p], atol=1e-6)
