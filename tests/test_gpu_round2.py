"""GPU parity tests added in round 2 (run with -m gpu): fused loss heads, lighting, the perceptual texture term, whole
training steps at the BENCH shape, BASELINE config 1's shape, the 1-rank RCCL path, and build provenance.

Every comparison is HIP (through the C ABI) against either golden vectors written by the reference itself or the CPU
oracle on the same seeded inputs; tolerances are written at each check and the measured figures are appended to
gpurun_out/parity_measured.jsonl (helpers.assert_close_frac)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import scene, assert_close_frac, t2n

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_was_built_from_these_sources():
    """The shipped libumr_hip.so carries a hash of the sources it was compiled from (umr_build_id); a stale .so --
    csrc/ edited after the last build -- fails here instead of silently testing old kernels."""
    from umr_amd import _lib, build
    assert _lib.build_id() == build.source_hash(), "libumr_hip.so is stale: run `python -m umr_amd.build`"


def test_cos_sim_head_vs_reference_golden():
    """PNet head (networks_basic.py:50-58 + util.py:71-83), one HIP launch per direction for all taps: values and
    gradients wrt BOTH feature stacks against the reference's autograd (golden)."""
    from umr_amd.perceptual import cos_sim_distance
    g = load_golden("part_loss_and_cos_grads.npz")
    f0 = [torch.from_numpy(g["cf0_%d" % k]).to(DEV).requires_grad_(True) for k in range(3)]
    f1 = [torch.from_numpy(g["cf1_%d" % k]).to(DEV).requires_grad_(True) for k in range(3)]
    val = cos_sim_distance(f0, f1)
    np.testing.assert_allclose(t2n(val), g["cos_val"], atol=1e-6)
    (val * torch.from_numpy(g["cos_gv"]).to(DEV)).sum().backward()
    for k in range(3):
        np.testing.assert_allclose(t2n(f0[k].grad), g["cg0_%d" % k], atol=1e-7, rtol=1e-4)
        np.testing.assert_allclose(t2n(f1[k].grad), g["cg1_%d" % k], atol=1e-7, rtol=1e-4)
    g2 = load_golden("parts_and_cossim.npz")       # round-1 golden (values only)
    a = [torch.from_numpy(g2["f0_0"]).to(DEV), torch.from_numpy(g2["f0_1"]).to(DEV)]
    b = [torch.from_numpy(g2["f1_0"]).to(DEV), torch.from_numpy(g2["f1_1"]).to(DEV)]
    np.testing.assert_allclose(t2n(cos_sim_distance(a, b)), g2["cos_dist"], atol=1e-6)
    # only one side needs a gradient (the ground-truth image's features do not): the other table entry is NULL
    f1b = [t.detach().clone().requires_grad_(True) for t in f1]
    cos_sim_distance([t.detach() for t in f0], f1b).sum().backward()
    assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in f1b)
    # a zero feature vector: defined as zero norm-gradient (torch gives NaN there), everything stays finite
    z0 = [torch.zeros(1, 4, 2, 2, device=DEV, requires_grad=True)]
    z1 = [torch.rand(1, 4, 2, 2, device=DEV, requires_grad=True)]
    v = cos_sim_distance(z0, z1)
    v.sum().backward()
    assert abs(float(v) - 1.0) < 1e-6 and torch.isfinite(z0[0].grad).all() and torch.isfinite(z1[0].grad).all()


def test_part_match_reductions_vs_reference_golden():
    """Everything after the renders in part_matching_loss.forward (loss_utils.py:399-440 incl. scops_utils centroids),
    fused: loss and gradient wrt the four part planes for avg=True and for the cam_probs-weighted path, against the
    reference's own forward (golden), incl. the max < 1e-5 branch (sample 1's third part is invisible)."""
    from umr_amd.functional import PartMatchFunction
    g = load_golden("part_loss_and_cos_grads.npz")
    planes = torch.from_numpy(g["planes"])
    B, _, H, W = planes.shape
    parts = torch.from_numpy(g["part_segs"]).to(DEV)

    def run(weights):
        ra = torch.rand(B, 4, H, W)
        rb = torch.rand(B, 4, H, W)            # unrelated channels must be ignored (and get zero gradient)
        ra[:, 0:3] = planes[:, 0:3]
        rb[:, 0] = planes[:, 3]
        ra, rb = ra.to(DEV).requires_grad_(True), rb.to(DEV).requires_grad_(True)
        l_eqv, l_lm = PartMatchFunction.apply(ra, rb, parts, (0., 5., 0., 0., 5.), 0.1, 1e-3)
        if weights is None:
            loss = (l_eqv.sum() / (B * 5 * H * W) + l_lm.sum() / (B * 8)) / 4.0
        else:
            w = weights.to(DEV)
            loss = (((l_eqv / (5 * H * W)).view(w.shape) * w).sum(1).mean() + ((l_lm / 8).view(w.shape) * w).sum(1).mean()) / 4.0
        loss.backward()
        grad = torch.cat((ra.grad[:, 0:3], rb.grad[:, 0:1]), 1)
        assert float(ra.grad[:, 3].abs().max()) == 0 and float(rb.grad[:, 1:].abs().max()) == 0
        return float(loss), t2n(grad)

    loss, grad = run(None)
    assert abs(loss - float(g["loss_avg"])) <= 1e-6 * max(1.0, abs(float(g["loss_avg"])))
    np.testing.assert_allclose(grad, g["grad_avg"], atol=1e-9, rtol=2e-4)
    loss, grad = run(torch.from_numpy(g["cam_probs"]))
    assert abs(loss - float(g["loss_weighted"])) <= 1e-6
    np.testing.assert_allclose(grad, g["grad_weighted"], atol=1e-9, rtol=2e-4)


def test_directional_light_folded_into_projection_vs_oracle(oracle_built):
    """sr.Lighting's surface light (lighting.py:50-57) comes out of the projection kernel: a textured render with the
    default light (ambient 0.8 + directional 0.5 along +y, smr.py:63) and its gradients wrt vertices, cameras AND
    textures -- the vertex gradient includes the path through the face normals -- against the CPU oracle."""
    from oracle import torch_ref
    from umr_amd.smr import SoftRenderer
    verts, faces, cams, gen = scene(2, 2, seed=21)
    tex = torch.rand(2, faces.shape[1], 4, 3, generator=gen)
    gimg = torch.randn(2, 4, 64, 64, generator=gen)
    vc, cc, tc = verts.clone().requires_grad_(True), cams.clone().requires_grad_(True), tex.clone().requires_grad_(True)
    ref = torch_ref.SoftRenderer(64, "softmax", n_threads=8)
    ri, _, _ = ref(vc, faces, cc, tc)
    (ri * gimg).sum().backward()
    vg, cg, tg = (verts.clone().to(DEV).requires_grad_(True), cams.clone().to(DEV).requires_grad_(True),
                  tex.clone().to(DEV).requires_grad_(True))     # clones: .to() of a tensor already on DEV returns the tensor itself
    r = SoftRenderer(64, "softmax")
    assert r.light_intensity_directional == 0.5 and r.light_intensity_ambient == 0.8
    img, _, _ = r(vg, faces.to(DEV), cg, tg)
    (img * gimg.to(DEV)).sum().backward()
    assert_close_frac(t2n(img), ri.detach().numpy(), atol=1e-4, frac=0.999, max_outlier=0.05, name="lit_image")
    for name, a, b in (("verts", vg, vc), ("cams", cg, cc), ("tex", tg, tc)):
        ref_g = b.grad.numpy()
        assert_close_frac(t2n(a.grad), ref_g, atol=3e-5 * np.abs(ref_g).max(), rtol=1e-3, frac=1.0, name="lit_grad_" + name)
    # the light term really is in the vertex gradient: with the directional part off the gradient differs
    v2 = verts.clone().to(DEV).requires_grad_(True)
    r2 = SoftRenderer(64, "softmax")
    r2.light_intensity_directional = 0
    img2, _, _ = r2(v2, faces.to(DEV), cams.to(DEV), tex.to(DEV))
    (img2 * gimg.to(DEV)).sum().backward()
    assert float((v2.grad - vg.grad).abs().max()) > 1e-3 * float(vg.grad.abs().max())


def test_perceptual_texture_loss_vs_oracle():
    """PerceptualTextureLoss (loss_utils.py:128-150): AlexNet taps on MIOpen + the HIP distance head, against the oracle's
    F.conv2d restatement with the SAME weights; value and gradients wrt the predicted image and the predicted mask."""
    from oracle import torch_ref
    from umr_amd.perceptual import PerceptualTextureLoss
    torch.manual_seed(11)
    ptl = PerceptualTextureLoss(DEV)
    ref = torch_ref.PerceptualTextureLoss(ptl.perceptual_loss.model.state_dict())
    gen = torch.Generator().manual_seed(2)
    pred, gt = torch.rand(3, 3, 128, 128, generator=gen), torch.rand(3, 3, 128, 128, generator=gen)
    m_gt, m_pr = (torch.rand(3, 128, 128, generator=gen) > 0.4).float(), torch.rand(3, 128, 128, generator=gen)
    pc, mc = pred.clone().requires_grad_(True), m_pr.clone().requires_grad_(True)
    rv = ref(pc, gt, m_gt, mc, avg=False)
    w = torch.tensor([0.2, 0.5, 0.3])
    (rv * w).sum().backward()
    pg, mg = pred.to(DEV).requires_grad_(True), m_pr.to(DEV).requires_grad_(True)
    v = ptl(pg, gt.to(DEV), m_gt.to(DEV), mg, avg=False)
    (v * w.to(DEV)).sum().backward()
    np.testing.assert_allclose(t2n(v), rv.detach().numpy(), atol=1e-4, rtol=1e-4)     # fp32 convolutions, different
    for name, a, b in (("img", pg, pc), ("mask", mg, mc)):                           # summation order (MIOpen vs CPU)
        rg = b.grad.numpy()
        assert_close_frac(t2n(a.grad), rg, atol=2e-5 * np.abs(rg).max(), rtol=1e-3, frac=1.0, name="perc_grad_" + name)
    assert abs(float(ptl(pred.to(DEV), gt.to(DEV), m_gt.to(DEV))) - float(ref(pred, gt, m_gt))) < 1e-4   # mask_pred=None branch


def test_multi_texture_loss_perceptual_branch_vs_oracle(oracle_built):
    """MultiTextureLoss with the reference's texture term (loss_utils.py:291-292, 313-321): K hypothesis renders sharing
    one texture set -> perceptual distance weighted by camera probabilities, against the oracle (explicit xK repeats)."""
    from oracle import torch_ref
    from umr_amd.loss_utils import MultiTextureLoss
    from umr_amd.synthetic import make_s2_inputs
    K, H = 2, 64
    tv, faces, out_c, batch_c, _ = make_s2_inputs(2, K, H, 2, seed=9, device="cpu")
    tv, faces, out_g, batch_g, _ = make_s2_inputs(2, K, H, 2, seed=9, device=DEV)
    torch.manual_seed(4)
    mtl = MultiTextureLoss(0, K, H, "softmax", "perceptual").to(DEV)
    ptl = torch_ref.PerceptualTextureLoss(mtl.pnet.state_dict())
    B = 2
    F = faces.shape[0]

    def textures(out, batch):
        return torch_ref.sample_textures(out["tex_flow"], batch["imgs"]).reshape(B, F, -1, 3) if out["tex_flow"].device.type == "cpu" \
            else None
    # oracle side
    tex_c = textures(out_c, batch_c)
    rep = lambda x: x.unsqueeze(1).repeat(1, K, *([1] * (x.dim() - 1))).view(-1, *x.shape[1:])
    fc = faces[None].expand(B, -1, -1)
    r = torch_ref.SoftRenderer(H, "softmax", n_threads=8)
    r.ambient_light_only()
    mask_r = torch_ref.SoftRenderer(H, "softmax", n_threads=8)
    pv_c = out_c["pred_vs"].detach()
    masks_pred_c = mask_r(rep(pv_c), rep(fc), out_c["cam_hypotheses"].detach().view(-1, 7))[0][:, 3]
    rgba, _, _ = r(rep(pv_c), rep(fc), out_c["cam_hypotheses"].detach().view(-1, 7), rep(tex_c))
    tl = ptl(rgba[:, :3], rep(batch_c["imgs"]), rep(batch_c["masks"]), masks_pred_c, avg=False)
    ref_loss = (tl.view(B, -1) * out_c["cam_probs"].detach()).sum(1).mean()
    ref_loss.backward()
    # HIP side
    from umr_amd import geom_utils
    from umr_amd.smr import SoftRenderer
    fg = faces.to(DEV)[None].expand(B, -1, -1)
    tex_g = geom_utils.sample_textures(out_g["tex_flow"], batch_g["imgs"]).reshape(B, F, -1, 3)
    mr = SoftRenderer(H, "softmax")
    masks_pred_g = mr(out_g["pred_vs"].detach(), fg, out_g["cam_hypotheses"].detach().view(-1, 7))[0][:, 3]
    tex_loss, _, _, _ = mtl(out_g["pred_vs"].detach(), fg, out_g["cam_hypotheses"].detach(), out_g["cam_probs"].detach(),
                            out_g["cam"].detach(), batch_g["imgs"], batch_g["masks"], masks_pred_g, tex_g, out_g["tex_flow"],
                            batch_g["dts_barrier"])
    tex_loss.backward()
    assert abs(float(tex_loss) - float(ref_loss)) <= 2e-4 * max(1.0, abs(float(ref_loss)))
    rg = out_c["tex_flow"].grad.numpy()
    assert_close_frac(t2n(out_g["tex_flow"].grad), rg, atol=2e-5 * np.abs(rg).max(), rtol=1e-3, frac=0.999, name="mtl_grad_flow")


def _s1_pair(B, H, subdiv, seed, epoch):
    from oracle import softras, torch_ref
    from oracle.train_step_ref import RenderCompareS1Ref
    from umr_amd.perceptual import PerceptualTextureLoss
    from umr_amd.synthetic import make_s1_inputs
    from umr_amd.train_step import RenderCompareS1
    torch.manual_seed(17)
    ptl = PerceptualTextureLoss(DEV)
    nt = softras.max_threads()
    tv, faces, out_c, batch_c = make_s1_inputs(B, H, subdiv, seed=seed, device="cpu")
    ref_total, ref_terms = RenderCompareS1Ref(tv, faces, H, n_threads=nt, epoch=epoch, texture_loss=torch_ref.PerceptualTextureLoss(
        ptl.perceptual_loss.model.state_dict()))(out_c, batch_c)
    ref_total.backward()
    tv, faces, out_g, batch_g = make_s1_inputs(B, H, subdiv, seed=seed, device=DEV)
    step = RenderCompareS1(tv.to(DEV), faces.to(DEV), H, texture_loss=ptl, epoch=epoch).to(DEV)
    total, terms = step(out_g, batch_g)
    total.backward()
    return ref_total, ref_terms, out_c, total, terms, out_g


@pytest.mark.parametrize("epoch", [0, 6])
def test_train_s1_step_with_perceptual_term_and_epoch_gating(oracle_built, epoch):
    """train_s1's own step: PerceptualTextureLoss (train_s1.py:150) and the epoch gating of the symmetry / deformation
    regularisers (:250-255: ori only while epoch < 3, deform only once epoch > 5), small shape, both gate states."""
    ref_total, ref_terms, out_c, total, terms, out_g = _s1_pair(2, 64, 2, seed=3, epoch=epoch)
    for k in ref_terms:
        assert abs(float(terms[k]) - float(ref_terms[k])) <= 2e-4 * max(1.0, abs(float(ref_terms[k]))), k
    assert abs(float(total) - float(ref_total)) <= 3e-4 * max(1.0, abs(float(ref_total)))
    w = 5.0 * float(ref_terms["deform"]) - 0.4 * float(ref_terms["ori"])          # what the gate moves in the total
    assert abs(w) > 1e-3
    for k in ("delta_v", "cam", "tex_flow"):
        r = out_c[k].grad.numpy()
        # measured: camera / texture-flow gradients in every element; 2 of 972 vertex-gradient values up to 1 % of scale off
        # (rotated camera of the adversarial term: float64 numpy in the restatement, float32 on the device)
        sc_ = np.abs(r).max()
        assert_close_frac(t2n(out_g[k].grad), r, atol=1e-3 * sc_, rtol=2e-2, frac=(0.997 if k == "delta_v" else 1.0),
                          max_outlier=3e-2 * sc_, name="s1_e%d_grad_%s" % (epoch, k))


def test_train_s1_step_at_bench_shape_vs_oracle(oracle_built):
    """The step bench.py times (BASELINE configs[1]: 256x256 images = 512x512 raster, 642-vertex / 1280-face mesh,
    perceptual texture term), B = 2 instead of 16, every term and every gradient against the CPU oracle on host cores."""
    ref_total, ref_terms, out_c, total, terms, out_g = _s1_pair(2, 256, 3, seed=31, epoch=0)
    for k in ref_terms:
        assert abs(float(terms[k]) - float(ref_terms[k])) <= 2e-4 * max(1.0, abs(float(ref_terms[k]))), (k, float(terms[k]), float(ref_terms[k]))
    assert abs(float(total) - float(ref_total)) <= 3e-4 * max(1.0, abs(float(ref_total)))
    for k in ("delta_v", "cam", "tex_flow"):
        r = out_c[k].grad.numpy()
        # every element; measured max error 7e-5 (vertices), 7e-8 (camera), 9e-7 (texture flow) of the largest gradient
        assert_close_frac(t2n(out_g[k].grad), r, atol={"delta_v": 5e-4, "cam": 1e-5, "tex_flow": 1e-5}[k] * np.abs(r).max(), frac=1.0,
                          name="s1_bench_grad_" + k)


def test_train_s2_step_at_bench_shape_vs_oracle(oracle_built):
    """train_s2 at the bench shape (256x256, 1280 faces, K = 8 hypotheses, AlexNet perceptual texture term), B = 2."""
    from oracle import softras, torch_ref
    from oracle.train_step_ref import RenderCompareS2Ref
    from umr_amd.synthetic import make_s2_inputs
    from umr_amd.train_step import RenderCompareS2
    K, H = 8, 256
    nt = softras.max_threads()
    tv, faces, out_g, batch_g, ex = make_s2_inputs(2, K, H, 3, seed=37, device=DEV)
    torch.manual_seed(23)
    step = RenderCompareS2(tv.to(DEV), faces.to(DEV), ex["part_vertex_ids"], ex["uv_img"], ex["uv_sampler"], H, K,
                           texture_loss_type="perceptual").to(DEV)
    total, terms = step(out_g, batch_g)
    total.backward()
    tv, faces, out_c, batch_c, ex = make_s2_inputs(2, K, H, 3, seed=37, device="cpu")
    ptl = torch_ref.PerceptualTextureLoss(step.texture_loss_fn.pnet.state_dict())
    ref_total, ref_terms = RenderCompareS2Ref(tv, faces, ex["part_vertex_ids"], ex["uv_img"], ex["uv_sampler"], H, K,
                                              n_threads=nt, texture_loss=ptl)(out_c, batch_c)
    ref_total.backward()
    for k in ref_terms:
        assert abs(float(terms[k]) - float(ref_terms[k])) <= 3e-4 * max(1.0, abs(float(ref_terms[k]))), (k, float(terms[k]), float(ref_terms[k]))
    for k in ("delta_v", "cam_hypotheses", "cam_probs", "tex_flow"):
        r = out_c[k].grad.numpy()
        # every element; measured max error 4.5e-5 (vertices), 3e-7 (cameras, probabilities), 6e-7 (texture flow) of scale
        assert_close_frac(t2n(out_g[k].grad), r, atol={"delta_v": 5e-4}.get(k, 1e-5) * np.abs(r).max(), frac=1.0,
                          name="s2_bench_grad_" + k)


def test_baseline_config1_shape_single_image(oracle_built):
    """BASELINE configs[0] (demo.py): ONE 256x256 image, 642-vertex icosphere, soft render + chamfer.  The reference runs it
    on its CPU plumbing path; there is deliberately no CPU backend in the product (HISTORY.md section 7), so the shape is
    covered here on the HIP path against the oracle: image, p2f, and distChamfer of the projected vertices."""
    from oracle import torch_ref
    from umr_amd.chamfer_python import distChamfer
    from umr_amd.smr import SoftRenderer
    verts, faces, cams, gen = scene(1, 3, seed=51)
    assert verts.shape == (1, 642, 3) and faces.shape == (1, 1280, 3)
    tex = torch.rand(1, 1280, 36, 3, generator=gen)
    r = SoftRenderer(256, "softmax")
    r.ambient_light_only()
    img, p2f, aggr = r(verts.to(DEV), faces.to(DEV), cams.to(DEV), tex.to(DEV))
    ref = torch_ref.SoftRenderer(256, "softmax", n_threads=16)
    ref.ambient_light_only()
    ri, rp, ra = ref(verts, faces, cams, tex)
    assert img.shape == (1, 4, 256, 256) and aggr.shape == (1, 2, 512, 512)
    assert_close_frac(t2n(img), ri.numpy(), atol=1e-4, frac=1.0, max_outlier=1e-5, name="cfg1_image")   # measured max 2.4e-7
    assert_close_frac(t2n(p2f), rp.numpy(), atol=1e-5, frac=1.0, name="cfg1_p2f")                       # measured max 7.2e-7
    pts = torch.rand(1, 300, 2, generator=gen) * 2 - 1
    v2d = r.project_points(verts.to(DEV), cams.to(DEV))
    d1, d2, i1, i2 = distChamfer(v2d, pts.to(DEV))
    rd1, rd2, ri1, ri2 = torch_ref.dist_chamfer(torch_ref.orthographic_proj_withz(verts, cams)[:, :, :2], pts)
    np.testing.assert_allclose(t2n(d1), rd1.numpy(), atol=1e-6)
    np.testing.assert_allclose(t2n(d2), rd2.numpy(), atol=1e-6)
    np.testing.assert_array_equal(t2n(i1), ri1.numpy())       # arg-mins: index work, exact (first minimum, the reference's strict <)
    np.testing.assert_array_equal(t2n(i2), ri2.numpy())


def test_one_rank_rccl_training_step():
    """The data-parallel path on real RCCL with one rank: NCCL(=RCCL) process group on 127.0.0.1; the step's gradient exchange
    in both forms -- parallel.BucketedGradSync (the default: gradients as views of one flat buffer, 64 MB buckets all-reduced from
    the gradient hooks) and torch DDP (64 MB buckets, no buffer broadcast) --, one full train_s1 step each; gradients are identical
    to the plain model's (all-reduce over one rank = identity) and the optimiser moves the weights."""
    import argparse
    import socket
    import torch.distributed as dist
    from umr_amd.model import build_training_step
    from umr_amd.synthetic import template
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device(DEV)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        args = argparse.Namespace(batch=4, image_size=64, subdivide=2, epoch=0)
        tv, faces = template(2)
        grads = {}
        for wrapped in ("buckets", "ddp", False):
            torch.manual_seed(5)
            args.grad_sync = wrapped or "buckets"
            step = build_training_step(tv, faces, args, dev, 2 if wrapped else 1)   # world 2: the data-parallel form
            assert (step.sync is not None) == (wrapped == "buckets")
            step.model.eval()       # BatchNorm on running statistics: with 4 samples at 64x64 the batch statistics of the
                                    # 1x1 bottleneck amplify summation-order noise (float atomics) into O(1) differences
            before = [p.detach().clone() for p in step.model.parameters() if p.requires_grad]
            torch.manual_seed(6)
            loss = step()
            assert torch.isfinite(loss)
            grads[wrapped] = torch.cat([p.grad.flatten() for p in step.model.parameters() if p.grad is not None]).clone()
            moved = sum(float((a - b.detach()).abs().sum()) for a, b in
                        zip(before, [p for p in step.model.parameters() if p.requires_grad]))
            assert moved > 0
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)
        assert float(t.sum()) == 4.0
        for form in ("buckets", "ddp"):
            assert grads[form].shape == grads[False].shape
            rel = float((grads[form] - grads[False]).abs().max() / grads[False].abs().max())
            assert rel < 1e-3, (form, rel)          # same seeds, same kernels; atomics in the projection scatter reorder a few sums
    finally:
        dist.destroy_process_group()


def test_bench_json_contract_small():
    """bench.py end to end on the GPU at a tiny configuration: ONE JSON line with the contract's fields, per-kernel
    roofline entries, honest workload text (no all-reduce claimed at dp1) and a CPU baseline."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2",
                          "--image-size", "64", "--subdivide", "2", "--model", "0", "--cpu-sample", "1", "--profile-steps", "1"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["dtype"] == "f32" and j["vs_baseline"] is None and j["scaling"] == "weak"
    assert "all-reduce" not in j["config"]["workload"] and "perceptual" in j["config"]["workload"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["launches"] >= 1 and 0 < rf["frac"] < 1 and rf["forward_kernel"]["launches"] >= 1
    assert rf["silhouette_forward"]["launches"] >= 1 and rf["silhouette_backward"]["launches"] >= 1
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0


def test_registered_custom_ops_schema_and_fake_kernels():
    """torch.ops.umr.{soft_rasterize, soft_rasterize_backward, silhouette, silhouette_backward}: torch.library.opcheck
    (schema, fake-tensor propagation against the real kernels, autograd registration) on real inputs."""
    from torch.library import opcheck
    from umr_amd import ops  # noqa: F401
    from umr_amd import functional as UF
    verts, faces, cams, gen = scene(2, 1, seed=4)
    _, fv, _ = UF.ProjectFacesFunction.apply(verts.to(DEV), cams.to(DEV), faces.int().to(DEV), 5.0, -2.732, False)
    tex = torch.rand(1, faces.shape[1], 4, 3, generator=gen).to(DEV)           # one texture set shared by both views
    args = (fv.detach().requires_grad_(True), tex.requires_grad_(True), 32, [0., 0., 0.], 1., 100., True, 1e-3, 1e-5, 1e-10,
            1e-4, 1, True, True, True)
    opcheck(torch.ops.umr.soft_rasterize.default, args, test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    opcheck(torch.ops.umr.silhouette.default, (fv.detach().requires_grad_(True), 32, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, True),
            test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    img, p2f, aggr, saved, vis = torch.ops.umr.soft_rasterize(*args)
    assert img.shape == (2, 4, 16, 16) and saved.shape == (2, 4, 32, 32) and aggr.shape == (2, 2, 32, 32) and vis.shape == aggr.shape
    img.sum().backward()
    assert args[0].grad.shape == fv.shape and args[1].grad.shape == tex.shape     # texture gradient summed over the group


def test_hot_path_step_replays_from_a_hip_graph():
    """The whole render-and-compare step (--model 0 path of bench.py: every raster / loss kernel forward AND backward, the
    AlexNet perceptual term included) captured once into a HIP graph and replayed: same losses and gradients as the eager
    step, also after the inputs are changed in place.  (Float-atomic sums -- p2f, the projection scatter -- are not
    bit-reproducible even eager-to-eager, hence 1e-5 relative instead of bit equality; the deterministic terms are equal to
    the bit.)"""
    from umr_amd.perceptual import PerceptualTextureLoss
    from umr_amd.synthetic import make_s1_inputs
    from umr_amd.train_step import RenderCompareS1
    torch.manual_seed(9)
    tv, faces, outputs, batch = make_s1_inputs(4, 128, 2, seed=3, device=DEV)
    rc = RenderCompareS1(tv.to(DEV), faces.to(DEV), 128, texture_loss=PerceptualTextureLoss(DEV)).to(DEV)
    leaves = [outputs["delta_v"], outputs["cam"], outputs["tex_flow"]]

    def step():
        for l in leaves:
            l.grad = None
        outputs["pred_vs"] = outputs["mean_shape"][None] + outputs["delta_v"]
        total, terms = rc(outputs, batch)
        total.backward()
        return total, terms

    def snapshot(total, terms):
        torch.cuda.synchronize()
        return float(total), {k: float(v) for k, v in terms.items()}, [l.grad.detach().clone() for l in leaves]

    # everything up to and including the capture on ONE side stream (first eager step, warm-up, capture): no node of the
    # captured autograd pass is then tied to the default stream.  Nothing allocated by the warm-up may die INSIDE the
    # capture: the step's non-leaf tensors and the gradients are dropped first (the hygiene torch.cuda.make_graphed_callables
    # applies too; a block of another stream freed during capture takes hipStreamEndCapture down on this stack).
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager = snapshot(*step())
        for _ in range(3):
            step()
        outputs["pred_vs"] = None
        for l in leaves:
            l.grad = None
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            static_total, static_terms = step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for l in leaves:
        l.grad.zero_()
    graph.replay()
    replay = snapshot(static_total, static_terms)

    def same(a, b):
        assert abs(a[0] - b[0]) <= 1e-5 * abs(b[0]), (a[0], b[0])
        for k in b[1]:
            assert abs(a[1][k] - b[1][k]) <= 1e-5 * max(1.0, abs(b[1][k])), k
        for k in ("mask", "triangle", "flatten", "tex_dt"):            # no float atomics on these paths
            assert a[1][k] == b[1][k], k
        for x, y in zip(a[2], b[2]):
            assert float((x - y).abs().max()) <= 2e-4 * float(y.abs().max())
    same(replay, eager)
    # new inputs, written in place into the captured buffers
    with torch.no_grad():
        outputs["cam"][:, 0] *= 0.9
        outputs["delta_v"].mul_(0.5)
        batch["masks"].copy_(batch["masks"].roll(1, 0))
    for _ in range(12):                                   # back-to-back replays, no host sync in between
        graph.replay()
    replay2 = snapshot(static_total, static_terms)
    eager2 = snapshot(*step())
    same(replay2, eager2)
    assert abs(eager2[0] - eager[0]) > 1e-3            # the inputs really changed


def test_superblock_bins_do_not_change_results():
    """Per-mesh coarse binning (k_superblock_bin) only narrows the candidate list a workgroup scans: forward outputs are
    bit-identical with the bins off (every workgroup scanning all F faces), for power-of-two, ragged and tiny images."""
    from umr_amd import _lib, functional as UF
    for (n, sub, IS, ts, rgb) in ((2, 3, 512, 4, "softmax"), (2, 2, 200, 1, "hard"), (1, 1, 24, 1, "softmax"), (3, 2, 136, 9, "softmax")):
        verts, faces, cams, gen = scene(n, sub, seed=IS)
        _, fv, _ = UF.ProjectFacesFunction.apply(verts.to(DEV), cams.to(DEV), faces.int().to(DEV), 5.0, -2.732, False)
        tex = torch.rand(n, faces.shape[1], ts, 3, generator=gen).to(DEV)
        outs = []
        for on in (1, 0):
            _lib.debug_set("superblock_bins", on)
            try:
                sc, p2f, aggr = UF.soft_rasterize(fv, tex, IS, [0.1, 0.2, 0.3], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, rgb)
                a = UF.SilhouetteFunction.apply(fv, IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, IS % 2 == 0)
                vis = UF.visibility(fv, IS)
                outs.append((sc.clone(), aggr.clone(), a.clone(), vis.clone(), p2f.clone()))
            finally:
                _lib.debug_set("superblock_bins", 1)
        for x, y in list(zip(outs[0], outs[1]))[:4]:
            assert torch.equal(x, y)
        assert float((outs[0][4] - outs[1][4]).abs().max()) <= 1e-5      # p2f: float atomics, order not fixed


def test_face_start_order_does_not_change_results():
    """k_face_order only decides WHEN the face-major backward starts the wave of a face (heavy faces first); every face is
    still reduced by one wave in its own order, so all gradients are bit-identical with the ordering off -- textured
    (vertex + texel, texel only), hard colour, silhouette; several mesh groups (N > 16), ragged N, image sizes."""
    from umr_amd import _lib, functional as UF
    for (n, sub, IS, ts, rgb) in ((18, 2, 128, 4, "softmax"), (3, 3, 256, 36, "softmax"), (2, 2, 200, 1, "hard"), (1, 1, 24, 1, "softmax")):
        verts, faces, cams, gen = scene(n, sub, seed=IS + 1)
        _, fv0, _ = UF.ProjectFacesFunction.apply(verts.to(DEV), cams.to(DEV), faces.int().to(DEV), 5.0, -2.732, False)
        tex0 = torch.rand(n, faces.shape[1], ts, 3, generator=gen).to(DEV)
        w = torch.rand(n, 4, IS, IS, generator=gen).to(DEV)
        wa = torch.rand(n, IS // 2 if IS % 2 == 0 else IS, IS // 2 if IS % 2 == 0 else IS, generator=gen).to(DEV)
        outs = []
        for on in (1, 0):                                                    # 1: cost-ordered work-item lists (every variant)
            _lib.debug_set("face_order", on)
            _lib.debug_set("face_split", 0)                                  # (order only: a split face sums in another order, round 6's tests)
            try:
                res = []
                fv = fv0.detach().clone().requires_grad_(True); tex = tex0.clone().requires_grad_(True)
                sc, _, _ = UF.soft_rasterize(fv, tex, IS, [0.1, 0.2, 0.3], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, rgb)
                (sc * w).sum().backward()
                res += [fv.grad.clone(), tex.grad.clone()]
                tex = tex0.clone().requires_grad_(True)                      # texel gradients only
                sc, _, _ = UF.soft_rasterize(fv0.detach(), tex, IS, [0.1, 0.2, 0.3], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, rgb)
                (sc * w).sum().backward()
                res.append(tex.grad.clone())
                fv = fv0.detach().clone().requires_grad_(True)               # silhouette
                a = UF.SilhouetteFunction.apply(fv, IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, IS % 2 == 0)
                (a * wa).sum().backward()
                res.append(fv.grad.clone())
                outs.append(res)
            finally:
                _lib.debug_set("face_order", 1)
                _lib.debug_set("face_split", -1)
        for x, y in zip(outs[0], outs[1]):
            assert torch.isfinite(x).all() and float(x.abs().sum()) > 0
            assert torch.equal(x, y)


def test_visibility_planes_from_the_textured_render_equal_the_hard_render():
    """train_s1.py:217-224 renders the textured soft-max image and, for the same mesh and camera, the hard image of which
    only aggrs_info's face-id plane is read.  umr_raster_forward_vis produces those planes inside the soft-max render's own
    visits: bit-identical to the hard render's aggrs_info and to the visibility-only kernel, the soft-max outputs
    unchanged -- power-of-two, ragged and tiny images, one-sided faces, shared texture sets."""
    from umr_amd import functional as UF
    for (n, sub, IS, ts, fill_back, G, pool) in ((3, 3, 512, 36, True, 1, True), (2, 2, 200, 1, False, 1, False),
                                                 (1, 1, 24, 4, True, 1, True), (4, 2, 136, 9, True, 2, True)):
        verts, faces, cams, gen = scene(n, sub, seed=IS + 5)
        _, fv, _ = UF.ProjectFacesFunction.apply(verts.to(DEV), cams.to(DEV), faces.int().to(DEV), 5.0, -2.732, False)
        tex = torch.rand(n // G, faces.shape[1], ts, 3, generator=gen).to(DEV)
        args = (IS, [0.1, 0.2, 0.3], 1, 100, fill_back, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4)
        sc, p2f, aggr, vis = UF.soft_rasterize(fv, tex, *args, 'softmax', pool=pool, want_visibility=True)
        sc0, p2f0, aggr0 = UF.soft_rasterize(fv, tex, *args, 'softmax', pool=pool)
        assert torch.equal(sc, sc0) and torch.equal(aggr, aggr0)
        _, _, hard = UF.soft_rasterize(fv, tex.repeat_interleave(G, 0), *args, 'hard')
        assert torch.equal(vis, hard)
        assert torch.equal(vis, UF.visibility(fv, IS, 1., 100., fill_back, 1e-3, 1e-5, 1e-10, 1e-4))
        assert float((vis[:, 1] >= 0).float().mean()) > 0.05            # the meshes are on screen
