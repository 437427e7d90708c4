"""CPU tests (no GPU): the `-m gpu` parity tests themselves -- tests/test_gpu_parity.py, test_gpu_round2.py, test_gpu_round3.py,
test_gpu_round5.py, test_gpu_zz_collapsed_edges.py, unmodified, with their MI355X bounds -- executed on HOST tensors over libumr_host.so: every
translation unit of the product library (umr_amd/csrc/*.hip) compiled for x86-64 on the wave64 emulator of
tests/host_kernel/wave_emu.h, driven by the product's own Python layer (functional.py, smr.py, loss_utils.py, train_step.py,
eval_utils.py ...) through tests/host_raster.py::emulated_product.  So the reference's goldens, the oracle comparisons and the
whole train_s1 / train_s2 steps check the kernels' SOURCE on every CPU run; the MI355X run of the same tests then adds what only
the hardware can: its own exp / rcp rounding, real atomics and wave scheduling, and time.

Left to the GPU box: timing, HIP-graph capture, RCCL, the reference's device code (oracle/ref_gpu), the torch operator
registrations' opcheck, and the shapes too large for an emulator (config 4, bench shape; the longer identity tests are covered
at smaller size by tests/test_raster_library_on_host.py)."""
import importlib
import inspect
import pathlib

import pytest

import host_raster as HR

pytestmark = pytest.mark.skipif(not HR.available(), reason="clang++ of the ROCm toolchain not present")

GPU_TESTS = {
    "test_gpu_parity": [
        "test_raster_cabi_vs_reference_golden", "test_raster_flags_and_fused_pool", "test_raster_rejects_undefined_modes",
        "test_smr_softrenderer_vs_reference_golden", "test_full_size_vs_oracle", "test_losses_vs_reference_goldens",
        "test_projection_gradients_vs_torch_autograd", "test_empty_scene_and_offscreen_mesh", "test_train_s1_step_vs_oracle",
        "test_backward_variants_agree", "test_face_major_backward_non_pow2_and_determinism", "test_train_s2_step_vs_oracle",
        "test_dt_barrier_vs_scipy", "test_upsample2x_matches_torch", "test_eval_metrics_vs_reference_restatement",
        "test_front_face_culling_and_depth_range_vs_oracle", "test_degenerate_faces_do_not_poison_the_image",
        "test_visibility_only_kernel_matches_hard_render", "test_texture_atlas_and_textured_obj_vs_reference_golden",
        "test_camera_hypothesis_groups_equal_explicit_repeats", "test_rotate_cam_y_kernel_vs_quaternion_product"],
    "test_gpu_round2": [
        "test_cos_sim_head_vs_reference_golden", "test_part_match_reductions_vs_reference_golden",
        "test_directional_light_folded_into_projection_vs_oracle", "test_perceptual_texture_loss_vs_oracle",
        "test_multi_texture_loss_perceptual_branch_vs_oracle", "test_train_s1_step_with_perceptual_term_and_epoch_gating",
        "test_baseline_config1_shape_single_image"],
    "test_gpu_round3": [
        "test_small_regularisers_and_masked_l1_vs_reference_goldens", "test_rotate_cam_vs_reference_golden",
        "test_keypoint_transfer_vs_reference_golden"],
    "test_gpu_zz_collapsed_edges": ["test_collapsed_edges_stay_finite_and_follow_the_reference"],
    "test_gpu_round5": ["test_alpha_geometry_backward_vs_oracle", "test_alpha_geometry_flag_is_refused_off_the_face_major_route",
                        "test_packed_state_equals_the_planar_state", "test_lean_shared_render_step_equals_the_planar_one"],
}


def _cases():
    out = []
    for mod_name, names in GPU_TESTS.items():
        mod = importlib.import_module(mod_name)
        for name in names:
            fn = getattr(mod, name)
            combos = [{}]
            for mark in [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]:
                keys = [k.strip() for k in mark.args[0].split(",")]
                new = []
                for c in combos:
                    for v in mark.args[1]:
                        v = v.values if hasattr(v, "values") else v
                        new.append(dict(c, **dict(zip(keys, v if len(keys) > 1 else (v,)))))
                combos = new
            for kw in combos:
                tag = "-".join(str(v) for v in kw.values())
                out.append(pytest.param(mod_name, name, kw, id="%s::%s%s" % (mod_name, name, "[%s]" % tag if tag else "")))
    return out


def test_the_product_still_refuses_host_tensors_outside_the_emulation():
    """emulated_product is a test harness, not a CPU path of the product: outside it the Python layer raises on a host tensor.
    (Runs before the module's emulation fixture is set up.)"""
    import torch
    from umr_amd import _lib, functional as UF
    t = torch.zeros(2, 4, 3)
    assert not _lib.on_device(t)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _lib.ptr(t)
    with pytest.raises((RuntimeError, TypeError)):
        UF.soft_rasterize(torch.zeros(1, 4, 3, 3), torch.zeros(1, 4, 1, 3), 16)


@pytest.fixture(scope="module")
def emulated():
    HR.lib(HR.build())
    with HR.emulated_product() as e:
        yield e


@pytest.mark.parametrize("mod_name,name,kw", _cases())
def test_gpu_test_on_the_emulated_library(emulated, oracle_built, tmp_path, mod_name, name, kw):
    mod = importlib.import_module(mod_name)
    fn = getattr(mod, name)
    params = inspect.signature(fn).parameters
    kw = dict(kw)
    if "oracle_built" in params:
        kw["oracle_built"] = oracle_built
    if "tmp_path" in params:
        kw["tmp_path"] = pathlib.Path(tmp_path)
    saved = mod.DEV
    mod.DEV = "cpu"
    try:
        fn(**kw)
    finally:
        mod.DEV = saved


def test_shared_mask_render_equals_the_two_renders_on_the_emulator():
    """RenderCompareS1 / S2 with share_mask_render (the mask render = the alpha channel of the textured render of the same views,
    SoftRenderer.forward(detach_rgb_geometry=True)) against the two renders the reference makes (train_s1.py:199 + :217,
    loss_utils.py:265 + :313): the alpha planes are the same bits, each gradient comes from the same kernel on the same
    inputs -- every loss term is EQUAL, not close; the gradients are sums of the same per-view contributions accumulated in
    another order (one 2B-view launch against two B-view launches feeding the projection's backward): equal to rounding."""
    import torch
    from host_raster import emulated_product
    from umr_amd.synthetic import make_s1_inputs
    from umr_amd.train_step import RenderCompareS1
    B, H = 2, 32
    with emulated_product():
        tv, faces, outputs, batch = make_s1_inputs(B, H, 1, seed=5, device="cpu")
        res = []
        for share in (True, False):
            rc = RenderCompareS1(tv, faces, H, share_mask_render=share)
            leaves = [outputs[k].detach().clone().requires_grad_(True) for k in ("delta_v", "cam", "tex_flow")]
            out = dict(outputs, delta_v=leaves[0], cam=leaves[1], tex_flow=leaves[2])
            out["pred_vs"] = outputs["mean_shape"][None] + leaves[0]
            total, terms = rc(out, batch)
            total.backward()
            res.append((float(total), {k: float(v) for k, v in terms.items()}, [l.grad.clone() for l in leaves]))
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
    for a, b in zip(res[0][2], res[1][2]):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())


def test_shared_render_backward_beyond_the_one_pass_kernels_texel_budget():
    """umr::soft_rasterize_alpha_geometry with 32 x 32 texels per face: the face-major kernels' LDS accumulators hold at most 1023
    texels, so the autograd formula takes the two-launch form (silhouette backward on the alpha plane + texel-only backward) and
    the one-pass operator refuses; gradients still route as the reference's two renders do."""
    import torch
    from host_raster import emulated_product
    from helpers import scene
    from umr_amd import functional as UF, ops
    with emulated_product():
        verts, faces, cams, gen = scene(2, 1, seed=3)
        _, fv, _ = UF.project_faces(verts, cams, faces.int(), 5.0, -2.732)
        fv = fv.detach()
        tex = torch.rand(2, faces.shape[1], 1024, 3, generator=gen)
        IS = 32
        args = (IS, [0, 0, 0], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, 'softmax', 'prod', 'surface')
        g = torch.randn(2, 4, IS, IS, generator=gen)
        fv1, tex1 = fv.clone().requires_grad_(True), tex.clone().requires_grad_(True)
        img = UF.soft_rasterize(fv1, tex1, *args, detach_rgb_geometry=True)[0]
        (img * g).sum().backward()
        fv2, tex2 = fv.clone().requires_grad_(True), tex.clone().requires_grad_(True)
        alpha = UF.silhouette(fv2, IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, False)
        img2 = UF.soft_rasterize(fv2.detach(), tex2, *args)[0]
        ((alpha * g[:, 3]).sum() + (img2[:, :3] * g[:, :3]).sum()).backward()
        assert torch.equal(img[:, 3], alpha) and float(fv2.grad.abs().max()) > 0 and float(tex2.grad.abs().max()) > 0
        assert torch.equal(fv1.grad, fv2.grad) and torch.equal(tex1.grad, tex2.grad)        # the same two kernels on the same inputs
        with pytest.raises(RuntimeError, match="texels per face"):
            torch.ops.umr.soft_rasterize_alpha_geometry_backward(fv, tex, img.detach().contiguous(), torch.zeros(2, 2, IS, IS), g,
                                                                 IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, ops.pack_modes(1), False)
