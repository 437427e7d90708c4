#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF, in this container only.

TEST INFRASTRUCTURE -- run by hand (`python oracle/gen_golden.py`) where /root/reference exists;
its outputs (small .npz files of inputs + expected outputs) are committed, the reference is not.

How the reference is made to run on a GPU-less host:
  * its CUDA kernel bodies are compiled for the host by oracle/ref_shim/build_ref.sh into
    oracle/_ref/libsoftras_ref.so; a tiny in-memory module with the pybind names
    (cuda/soft_rasterize_cuda.cpp:141-144) forwards torch CPU tensors to it via ctypes;
  * the SoftRas Python stack (external/SoftRas/soft_renderer) and the UMR wrappers
    (nnutils/{smr,loss_utils,geom_utils,chamfer_python,scops_utils}.py) are imported UNMODIFIED
    from /root/reference; absent third-party modules that they import but never use on this
    path (cv2, torchvision, neural_renderer, skimage, scipy.misc) are registered as empty stubs;
  * `.cuda()` is patched to identity and `affine_grid`/`grid_sample` defaults to the torch-1.1.0
    behaviour the reference pins (requirements.txt:8) = align_corners=True.
Nothing is written under /root/reference (bytecode writing is disabled first).
"""
import sys
sys.dont_write_bytecode = True
import ctypes
import importlib.util
import os
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from umr_amd.mesh import create_sphere  # noqa: E402  (input construction only)

_Fp = ctypes.POINTER(ctypes.c_float)


def _p(t):
    assert t.dtype == torch.float32 and t.is_contiguous()
    return ctypes.cast(t.data_ptr(), _Fp)


def install_reference():
    lib = ctypes.CDLL(os.path.join(HERE, "_ref", "libsoftras_ref.so"))

    def forward_soft_rasterize(faces, textures, faces_info, aggrs_info, grid, p2f_info, p2f_sum, soft_colors,
                               image_size, near, far, eps, sigma_val, func_id_dist, dist_eps, gamma_val,
                               func_id_rgb, func_id_alpha, texture_sample_type, double_side):
        n, f = faces.shape[:2]
        lib.ref_forward_soft_rasterize(
            _p(faces), _p(textures), _p(faces_info), _p(aggrs_info), _p(grid), _p(p2f_info), _p(p2f_sum),
            _p(soft_colors), n, f, int(image_size), int(textures.shape[2]), ctypes.c_float(near),
            ctypes.c_float(far), ctypes.c_float(eps), ctypes.c_float(sigma_val), int(func_id_dist),
            ctypes.c_float(dist_eps), ctypes.c_float(gamma_val), int(func_id_rgb), int(func_id_alpha),
            int(texture_sample_type), int(bool(double_side)))
        return faces_info, aggrs_info, p2f_info, p2f_sum, soft_colors

    def backward_soft_rasterize(faces, textures, soft_colors, faces_info, aggrs_info, grad_faces, grad_textures,
                                grad_soft_colors, image_size, near, far, eps, sigma_val, func_id_dist, dist_eps,
                                gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side):
        n, f = faces.shape[:2]
        lib.ref_backward_soft_rasterize(
            _p(faces), _p(textures), _p(soft_colors), _p(faces_info), _p(aggrs_info), _p(grad_faces),
            _p(grad_textures), _p(grad_soft_colors), n, f, int(image_size), int(textures.shape[2]),
            ctypes.c_float(near), ctypes.c_float(far), ctypes.c_float(eps), ctypes.c_float(sigma_val),
            int(func_id_dist), ctypes.c_float(dist_eps), ctypes.c_float(gamma_val), int(func_id_rgb),
            int(func_id_alpha), int(texture_sample_type), int(bool(double_side)))
        return grad_faces, grad_textures

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    cuda_pkg = stub("soft_renderer.cuda")
    cuda_pkg.__path__ = []
    stub("soft_renderer.cuda.soft_rasterize", forward_soft_rasterize=forward_soft_rasterize,
         backward_soft_rasterize=backward_soft_rasterize)
    for n in ("load_textures", "create_texture_image", "voxelization"):
        stub("soft_renderer.cuda." + n)
    sk = stub("skimage"); sk.__path__ = []
    stub("skimage.io", imread=None, imsave=None)
    stub("cv2")
    stub("neural_renderer")
    tv = stub("torchvision"); tv.__path__ = []
    stub("torchvision.utils")
    import scipy
    scipy.misc = stub("scipy.misc")

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    _ag, _gs = F.affine_grid, F.grid_sample
    F.affine_grid = lambda theta, size, align_corners=True: _ag(theta, size, align_corners=align_corners)
    F.grid_sample = lambda i, g, mode='bilinear', padding_mode='zeros', align_corners=True: \
        _gs(i, g, mode=mode, padding_mode=padding_mode, align_corners=align_corners)
    torch.nn.functional.affine_grid, torch.nn.functional.grid_sample = F.affine_grid, F.grid_sample

    sys.path.insert(0, os.path.join(REF, "external", "SoftRas"))
    umr = types.ModuleType("UMR"); umr.__path__ = [REF]; sys.modules["UMR"] = umr
    for sub in ("nnutils", "utils"):
        m = types.ModuleType("UMR." + sub); m.__path__ = [os.path.join(REF, sub)]; sys.modules["UMR." + sub] = m
    import soft_renderer as sr
    from UMR.nnutils import smr, loss_utils, geom_utils, chamfer_python, scops_utils
    spec = importlib.util.spec_from_file_location(
        "ps_util", os.path.join(REF, "external", "PerceptualSimilarity", "util", "util.py"))
    return lib, sr, smr, loss_utils, geom_utils, chamfer_python, scops_utils, spec


def scene(n_meshes, subdiv, seed, scale=(0.6, 0.9)):
    """Seeded synthetic scene: perturbed icosphere + random cameras."""
    g = torch.Generator().manual_seed(seed)
    v, f = create_sphere(subdiv)
    verts = torch.from_numpy(v).float()[None].repeat(n_meshes, 1, 1)
    verts = verts + 0.05 * torch.randn(verts.shape, generator=g)
    faces = torch.from_numpy(f).long()[None].repeat(n_meshes, 1, 1)
    s = scale[0] + (scale[1] - scale[0]) * torch.rand(n_meshes, 1, generator=g)
    t = -0.1 + 0.2 * torch.rand(n_meshes, 2, generator=g)
    q = torch.randn(n_meshes, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    return verts, faces, torch.cat([s, t, q], 1), g


def np_(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def gen_save_obj(lib):
    """(vii) functional/save_obj.py run unmodified: the atlas bake goes through the reference's own kernel body
    (ref_create_texture_image), `imsave` is captured instead of encoded, and LongTensor / int is given its
    torch-1.1 meaning (integer division; save_obj.py:17 relies on it) for the duration of the call."""
    import tempfile

    def create_texture_image(vertices, textures, image, eps):
        vertices, textures = vertices.contiguous(), textures.contiguous()
        nf, r_in = textures.shape[0], int(textures.shape[1] ** 0.5)
        tile_width = int((nf - 1) ** 0.5) + 1                 # create_texture_image_cuda_kernel.cu:82
        lib.ref_create_texture_image(_p(vertices), _p(textures), _p(image), ctypes.c_long(image.numel()), nf, r_in,
                                     image.shape[1] // tile_width, tile_width, ctypes.c_float(eps))
        return image

    sys.modules["soft_renderer.cuda.create_texture_image"].create_texture_image = create_texture_image
    captured = {}
    sys.modules["skimage.io"].imsave = lambda fn, arr: captured.update(png=np.array(arr))
    spec = importlib.util.spec_from_file_location(
        "ref_save_obj", os.path.join(REF, "external", "SoftRas", "soft_renderer", "functional", "save_obj.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    g = torch.Generator().manual_seed(23)
    v, f = create_sphere(1)                                   # 42 verts / 80 faces
    verts = torch.from_numpy(v).float() + 0.05 * torch.randn(v.shape, generator=g)
    faces = torch.from_numpy(f).long()
    tex = torch.rand(faces.shape[0], 36, 3, generator=g) * 1.2 - 0.1      # some values outside [0,1]: exercises clip
    tex7 = torch.rand(7, 4, 3, generator=g)                   # ragged atlas: 7 faces in a 3x3 grid, R=2, 8 px cells
    true_div = torch.Tensor.__truediv__
    torch.Tensor.__truediv__ = lambda a, b: (torch.floor_divide(a, b) if (not a.is_floating_point()
                                             and isinstance(b, int)) else true_div(a, b))
    try:
        with tempfile.TemporaryDirectory() as d:
            mod.save_obj(os.path.join(d, "bird.obj"), verts, faces, textures=tex, texture_res=16)
            obj_tex, mtl = open(os.path.join(d, "bird.obj")).read(), open(os.path.join(d, "bird.mtl")).read()
            mod.save_obj(os.path.join(d, "plain.obj"), verts, faces)
            obj_plain = open(os.path.join(d, "plain.obj")).read()
        img8, uv8 = mod.create_texture_image(tex7, texture_res=8)
    finally:
        torch.Tensor.__truediv__ = true_div
    np.savez_compressed(os.path.join(OUT, "save_obj.npz"), verts=np_(verts), faces=np_(faces), textures=np_(tex),
                        png=captured["png"], obj_textured=np.frombuffer(obj_tex.encode(), np.uint8),
                        mtl=np.frombuffer(mtl.encode(), np.uint8),
                        obj_plain=np.frombuffer(obj_plain.encode(), np.uint8),
                        tex7=np_(tex7), atlas7=np.ascontiguousarray(img8), uv7=uv8)
    print("save_obj golden:", captured["png"].shape, len(obj_tex), "bytes of OBJ")


def gen_part_loss_and_cos_grads(loss_utils, ps_spec):
    """(viii) tests/golden/part_loss_and_cos_grads.npz, run by `python oracle/gen_golden.py --only part_cos`:
      * part_matching_loss.forward (nnutils/loss_utils.py:380-440) UNBOUND on a stand-in `self` that carries only what
        forward reads (renderer -> queued images, the stex buffers it passes through, proj, weights, loss_type): the class
        constructor needs scipy.misc.imread and the undistributed SCOPS template, forward itself is plain torch.  The
        "renders" are random images so every reduction after them is exercised, with one part invisible in one sample
        (max < 1e-5 branch, :418-419); values and gradients wrt the rgb-mean planes for avg=True and the cam_probs path;
      * util.cos_sim (external/PerceptualSimilarity/util/util.py:71-83) summed as PNet.forward does
        (networks_basic.py:50-58): values and autograd gradients wrt both feature stacks."""
    torch.Tensor.get_device = lambda self: "cpu"            # :411 `.to(cam_probs.get_device())` on a GPU-less host
    g = torch.Generator().manual_seed(20260926)
    B, H = 4, 32
    imgs = [torch.rand(B, 4, H, H, generator=g) * torch.rand(B, 1, 1, 1, generator=g) for _ in range(4)]
    imgs[2][1, 0:3] = 0.0                                     # part 3 invisible in sample 1
    imgs = [i.requires_grad_(True) for i in imgs]
    part_segs = (torch.randn(B, 5, H, H, generator=g) * 2).contiguous()
    part_segs[3, 2] = -1.0                                    # a constant negative plane: max_part < 1e-5 branch

    class _Self:
        pass

    def run(avg, cam_probs):
        st = _Self()
        q = list(imgs)
        st.renderer = lambda verts, faces, cams, tex: (q.pop(0), None, None)
        st.stex1 = st.stex2 = st.stex3 = st.stex4 = torch.zeros(B, 1)
        st.proj = torch.full((B, 1, H, H), 0.1)
        st.weights = torch.tensor([0, 5.0, 0.0, 0.0, 5.0]).view(1, 5, 1, 1)
        st.loss_type = 'mse'
        for i in imgs:
            i.grad = None
        loss, projs = loss_utils.part_matching_loss.forward(st, torch.zeros(B, 1), None, None, part_segs, cam_probs, avg)
        loss.backward()
        # gradient wrt the rgb-mean plane = sum of the gradients of its three channels
        return loss.detach(), torch.stack([i.grad[:, 0:3].sum(1) for i in imgs], 1), projs

    loss_avg, grad_avg, projs = run(True, None)
    probs = torch.softmax(torch.randn(2, 2, generator=g), 1)
    loss_w, grad_w, _ = run(False, probs)
    planes = torch.cat([p.detach() for p in projs], 1)        # [B,4,H,H] rgb means, the quantity the loss consumes
    ps_util_src = open(ps_spec.origin).read()
    ns = {"torch": torch, "np": np}
    start = ps_util_src.index("def normalize_tensor"); end = ps_util_src.index("# Converts a Tensor into a Numpy array")
    exec(compile(ps_util_src[start:end], ps_spec.origin, "exec"), ns)
    shapes = ((8, 7, 7), (16, 5, 5), (12, 3, 6))
    f0 = [torch.randn(3, *sh, generator=g).relu().requires_grad_(True) for sh in shapes]
    f1 = [torch.randn(3, *sh, generator=g).relu().requires_grad_(True) for sh in shapes]
    val = 0
    for kk in range(len(f0)):
        cur = (1. - ns["cos_sim"](f0[kk], f1[kk]))
        val = 1. * cur if kk == 0 else val + cur               # networks_basic.py:53-58
    gv = torch.rand(3, generator=g) + 0.5
    (val * gv).sum().backward()
    out = dict(planes=np_(planes), part_segs=np_(part_segs), loss_avg=np_(loss_avg), grad_avg=np_(grad_avg),
               cam_probs=np_(probs), loss_weighted=np_(loss_w), grad_weighted=np_(grad_w), cos_val=np_(val), cos_gv=np_(gv))
    for k in range(len(f0)):
        out.update({"cf0_%d" % k: np_(f0[k]), "cf1_%d" % k: np_(f1[k]), "cg0_%d" % k: np_(f0[k].grad),
                    "cg1_%d" % k: np_(f1[k].grad)})
    np.savez_compressed(os.path.join(OUT, "part_loss_and_cos_grads.npz"), **out)
    print("part_loss_and_cos_grads.npz: loss_avg %.6f loss_weighted %.6f cos %s" % (float(loss_avg), float(loss_w), np_(val)))


MODE_CASES = (  # tag, func_id_dist, func_id_alpha, func_id_rgb, texture_sample_type, texture_size, sigma_val
    ("bary_sum_softmax_vertex", 1, 1, 1, 1, 3, 1e-4),
    ("hard_hard_hard_surface", 0, 0, 0, 0, 4, 1e-5),
    ("euclid_sum_hard_vertex", 2, 1, 0, 1, 3, 1e-5),
    ("bary_prod_softmax_surface", 1, 2, 1, 0, 4, 1e-4),
    ("hard_prod_softmax_surface", 0, 2, 1, 0, 1, 1e-5),
    ("euclid_hard_softmax_surface", 2, 0, 1, 0, 9, 1e-5),
    ("bary_hard_hard_vertex", 1, 0, 0, 1, 3, 1e-4),
)


def gen_raster_modes(sr, geom_utils, softras):
    """The mode ids UMR never selects but the binding accepts (functional/soft_rasterize.py:21-24): hard / barycentric
    distance, hard / sum alpha, vertex textures -- forward outputs and gradients straight from the reference kernels."""
    verts, faces, cams, g = scene(2, 1, seed=57)
    proj = geom_utils.orthographic_proj_withz(verts, cams, offset_z=5.)
    proj[:, :, 1] *= -1
    proj[:, :, 2] += 2.732
    fv = np_(sr.functional.face_vertices(proj, faces.int()).contiguous())
    IS = 32
    gsc = np_(torch.randn(2, 4, IS, IS, generator=g))
    d = dict(faces=fv, image_size=IS, background=np.float32([0.1, 0.2, 0.3]), grad_soft_colors=gsc,
             cases=np.array([c[0] for c in MODE_CASES]), near=1., far=100., eps=1e-3,
             dist_eps_log=float(np.log(1. / 1e-10 - 1.)), gamma_val=1e-4, double_side=True)
    for tag, fd, fa, fr, tt, ts, sigma in MODE_CASES:
        tex = np_(torch.rand(2, faces.shape[1], ts, 3, generator=g))
        cfg = dict(near=1., far=100., eps=1e-3, sigma_val=sigma, dist_eps_log=d["dist_eps_log"], gamma_val=1e-4,
                   func_id_rgb=fr, double_side=True, func_id_dist=fd, func_id_alpha=fa, texture_sample_type=tt)
        o = softras.raster_forward(fv, tex, IS, background=(0.1, 0.2, 0.3), backend="ref", **cfg)
        gf, gt = softras.raster_backward(o["faces"], o["textures"], o["soft_colors"], o["faces_info"], o["aggrs_info"],
                                         gsc, IS, backend="ref", **cfg)
        assert np.isfinite(gf).all() and np.isfinite(gt).all()
        d.update({tag + "/modes": np.int32([fd, fa, fr, tt]), tag + "/sigma_val": np.float32(sigma), tag + "/textures": tex,
                  tag + "/soft_colors": o["soft_colors"], tag + "/aggrs_info": o["aggrs_info"], tag + "/p2f_info": o["p2f_info"],
                  tag + "/p2f_sum": o["p2f_sum"], tag + "/grad_faces": gf, tag + "/grad_textures": gt})
        print("raster_modes/%s" % tag, "alpha mean %.4f" % o["soft_colors"][:, 3].mean(), "|gf| %.3e |gt| %.3e" % (np.abs(gf).sum(), np.abs(gt).sum()))
    np.savez_compressed(os.path.join(OUT, "raster_modes.npz"), **d)


def main():
    os.makedirs(OUT, exist_ok=True)
    lib, sr, smr, loss_utils, geom_utils, chamfer_python, scops_utils, ps_spec = install_reference()
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "part_cos":
        gen_part_loss_and_cos_grads(loss_utils, ps_spec)
        return
    sys.path.insert(0, ROOT)
    from oracle import softras  # only for its ctypes helper on the ref .so
    if sys.argv[1:] == ["save_obj"]:
        return gen_save_obj(lib)
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "modes":
        return gen_raster_modes(sr, geom_utils, softras)

    # (i) kernel-level: faces in screen space straight into the reference kernels -------------------------
    for tag, ts, rgb in (("softmax_ts36", 36, 1), ("softmax_ts1", 1, 1), ("hard_ts1", 1, 0), ("hard_ts4", 4, 0)):
        verts, faces, cams, g = scene(2, 1, seed=11)
        proj = geom_utils.orthographic_proj_withz(verts, cams, offset_z=5.)
        proj[:, :, 1] *= -1
        proj[:, :, 2] += 2.732
        fv = sr.functional.face_vertices(proj, faces.int()).contiguous()
        tex = torch.rand(2, faces.shape[1], ts, 3, generator=g)
        IS = 64
        cfg = dict(near=1., far=100., eps=1e-3, sigma_val=1e-5, dist_eps_log=float(np.log(1. / 1e-10 - 1.)),
                   gamma_val=1e-4, func_id_rgb=rgb, double_side=True)
        o = softras.raster_forward(np_(fv), np_(tex), IS, background=(0.1, 0.2, 0.3), backend="ref", **cfg)
        gsc = torch.randn(2, 4, IS, IS, generator=g)
        gf, gt = softras.raster_backward(o["faces"], o["textures"], o["soft_colors"], o["faces_info"],
                                         o["aggrs_info"], np_(gsc), IS, backend="ref", **cfg)
        np.savez_compressed(os.path.join(OUT, "raster_%s.npz" % tag), faces=o["faces"], textures=o["textures"],
                            image_size=IS, background=np.float32([0.1, 0.2, 0.3]), func_id_rgb=rgb,
                            faces_info=o["faces_info"], aggrs_info=o["aggrs_info"], p2f_info=o["p2f_info"],
                            p2f_sum=o["p2f_sum"], soft_colors=o["soft_colors"], grad_soft_colors=np_(gsc),
                            grad_faces=gf, grad_textures=gt, **{k: v for k, v in cfg.items() if k != "func_id_rgb"})
        print("raster_%s" % tag, "alpha mean %.4f" % o["soft_colors"][:, 3].mean(), "|gf| %.3e" % np.abs(gf).sum())

    # (ii) nnutils.smr.SoftRenderer.forward end to end (both light settings, with and without textures) ---
    for tag, ambient_only, with_tex, rtype in (("mask_default_light", False, False, "softmax"),
                                               ("tex_ambient", True, True, "softmax"),
                                               ("tex_default_light", False, True, "softmax"),
                                               ("hard_default_light", False, False, "hard")):
        verts, faces, cams, g = scene(2, 1, seed=23)
        verts.requires_grad_(True); cams.requires_grad_(True)
        tex = torch.rand(2, faces.shape[1], 36, 3, generator=g).requires_grad_(True) if with_tex else None
        r = smr.SoftRenderer(32, rtype)
        if ambient_only:
            r.ambient_light_only()
        imgs, p2f, aggr = r.forward(verts, faces, cams, tex) if with_tex else r.forward(verts, faces, cams)
        gi = torch.randn(imgs.shape, generator=g)
        imgs.backward(gi)
        d = dict(verts=np_(verts), faces=np_(faces), cams=np_(cams), img_size=32, render_type=rtype,
                 ambient_only=ambient_only, imgs=np_(imgs), p2f=np_(p2f), aggr=np_(aggr), grad_imgs=np_(gi),
                 grad_verts=np_(verts.grad), grad_cams=np_(cams.grad),
                 proj_points=np_(r.project_points(verts, cams)))
        if with_tex:
            d.update(textures=np_(tex), grad_textures=np_(tex.grad))
        np.savez_compressed(os.path.join(OUT, "smr_%s.npz" % tag), **d)
        print("smr_%s" % tag, "img mean %.4f" % imgs.mean().item())

    # (iii) losses around the renderer ----------------------------------------------------------------------
    K = 2
    verts, faces, cams, g = scene(2, 1, seed=31)
    cams_h = torch.stack([scene(2, 1, seed=40 + k)[2] for k in range(K)], 1).requires_grad_(True)  # [B,K,7]
    verts.requires_grad_(True)
    probs = torch.softmax(torch.randn(2, K, generator=g), 1).requires_grad_(True)
    masks_gt = (torch.rand(2, 32, 32, generator=g) > 0.5).float()
    mml = loss_utils.MultiMaskLoss(32, "softmax", K)
    loss, mask_all = mml.forward(verts, faces, cams_h, probs, masks_gt)
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "loss_multimask.npz"), verts=np_(verts), faces=np_(faces),
                        cams_all_hypo=np_(cams_h), cam_probs=np_(probs), masks_gt=np_(masks_gt), loss=np_(loss),
                        mask_all_hypo=np_(mask_all), grad_verts=np_(verts.grad), grad_cams=np_(cams_h.grad),
                        grad_probs=np_(probs.grad), image_size=32, num_hypo_cams=K)
    print("loss_multimask %.6f" % loss.item())

    p = torch.rand(3, 16, 16, generator=g).requires_grad_(True)
    t = (torch.rand(3, 16, 16, generator=g) > 0.4).float()
    l1 = loss_utils.neg_iou_loss(p, t); l2 = loss_utils.neg_iou_loss(p, t, avg=False)
    (l1 + (l2 * torch.tensor([1., 2., 3.])).sum()).backward()
    np.savez_compressed(os.path.join(OUT, "loss_neg_iou.npz"), predict=np_(p), target=np_(t), loss_avg=np_(l1),
                        loss_per=np_(l2), grad_predict=np_(p.grad))

    B, Fn, T = 2, 80, 6
    flow = (torch.rand(B, Fn, T, T, 2, generator=g) * 2.4 - 1.2).requires_grad_(True)  # some out of range
    images = torch.rand(B, 3, 24, 24, generator=g).requires_grad_(True)
    dts = torch.rand(B, 1, 24, 24, generator=g)
    tex = geom_utils.sample_textures(flow, images)
    gtex = torch.randn(tex.shape, generator=g)
    tex.backward(gtex)
    gflow_tex, gimg = flow.grad.clone(), images.grad.clone()
    flow.grad = None
    dtl = loss_utils.texture_dt_loss(flow, dts)
    dtl.backward()
    np.savez_compressed(os.path.join(OUT, "loss_texture_sampling.npz"), flow=np_(flow), images=np_(images),
                        dts=np_(dts), tex=np_(tex), grad_tex=np_(gtex), grad_flow_from_tex=np_(gflow_tex),
                        grad_images=np_(gimg), dt_loss=np_(dtl), grad_flow_from_dt=np_(flow.grad))

    # TexCycle on real renderer outputs: softmax p2f (train_s1.py:217-226) and hard face-id plane
    verts, faces, cams, g = scene(2, 1, seed=57)
    texr = smr.SoftRenderer(32, "softmax"); texr.ambient_light_only()
    _, p2f_soft, _ = texr.forward(verts, faces, cams, torch.rand(2, 80, 36, 3, generator=g))
    _, p2f_hard, aggr_hard = smr.SoftRenderer(32, "hard").forward(verts, faces, cams)
    ids = aggr_hard[:, 1].reshape(2, -1)
    flow = (torch.rand(2, 80, 6, 6, 2, generator=g) * 2 - 1).requires_grad_(True)
    tc = loss_utils.TexCycle()
    lc, avg10 = tc.forward(flow, p2f_soft.detach(), ids.detach())
    lc.backward()
    lc_hard, _ = tc.forward(flow, p2f_hard.detach(), ids.detach())
    np.savez_compressed(os.path.join(OUT, "loss_texcycle.npz"), verts=np_(verts), faces=np_(faces), cams=np_(cams),
                        flow=np_(flow), p2f_soft=np_(p2f_soft), p2f_hard=np_(p2f_hard), face_ids=np_(ids),
                        loss=np_(lc), loss_hard_target=np_(lc_hard), grad_flow=np_(flow.grad), avg_flow10=np_(avg10))
    print("loss_texcycle %.6f hard-p2f abs sum %.3e ids min %d" % (lc.item(), p2f_hard.abs().sum().item(), int(ids.min())))

    dv = torch.randn(2, 30, 3, generator=g).requires_grad_(True)
    (loss_utils.deform_l2reg(dv) + 2 * loss_utils.sym_reg(dv)).backward()
    np.savez_compressed(os.path.join(OUT, "loss_small_regs.npz"), v=np_(dv), deform=np_(loss_utils.deform_l2reg(dv)),
                        sym=np_(loss_utils.sym_reg(dv)), grad_v=np_(dv.grad))

    # (iv) chamfer ------------------------------------------------------------------------------------------
    ch = {}
    for i, (b, n, m, d) in enumerate(((3, 17, 10, 2), (2, 40, 30, 2), (1, 700, 42, 2), (2, 9, 13, 3))):
        a = torch.randn(b, n, d, generator=g).requires_grad_(True)
        bb = torch.randn(b, m, d, generator=g).requires_grad_(True)
        d1, d2, i1, i2 = chamfer_python.distChamfer(a, bb)
        (d1.sum() + 0.5 * d2.sum()).backward()
        ch.update({"a%d" % i: np_(a), "b%d" % i: np_(bb), "d1_%d" % i: np_(d1), "d2_%d" % i: np_(d2),
                   "i1_%d" % i: np_(i1), "i2_%d" % i: np_(i2), "ga%d" % i: np_(a.grad), "gb%d" % i: np_(bb.grad)})
    np.savez_compressed(os.path.join(OUT, "chamfer.npz"), n_cases=4, **ch)

    # (v) mesh regularisers on icosphere(1) -----------------------------------------------------------------
    v, f = create_sphere(1)
    vt, ft = torch.from_numpy(v).float(), torch.from_numpy(f).int()
    lap, flat = sr.LaplacianLoss(vt, ft), sr.FlattenLoss(ft)
    x = (vt[None].repeat(2, 1, 1) + 0.1 * torch.randn(2, 42, 3, generator=g)).requires_grad_(True)
    ll, fl = lap(x), flat(x)
    (ll.sum() + fl.sum()).backward()
    np.savez_compressed(os.path.join(OUT, "mesh_regs.npz"), verts0=v.astype(np.float32), faces=f, x=np_(x),
                        laplacian=np_(ll), flatten=np_(fl), grad_x=np_(x.grad), lap_matrix=np_(lap.laplacian))

    # (vi) SCOPS centroids + PNet cosine head ---------------------------------------------------------------
    pm = torch.softmax(torch.randn(2, 5, 24, 24, generator=g), 1).requires_grad_(True)
    cen = scops_utils.batch_get_centers(pm[:, 1:])
    cen.backward(torch.ones_like(cen))
    ps_util_src = open(ps_spec.origin).read()
    ns = {"torch": torch, "np": np}
    # only the two pure-torch helpers are needed; the module's top-level imports (matplotlib, skimage...) are absent
    start = ps_util_src.index("def normalize_tensor"); end = ps_util_src.index("# Converts a Tensor into a Numpy array")
    exec(compile(ps_util_src[start:end], ps_spec.origin, "exec"), ns)
    f0 = [torch.randn(2, c, s, s, generator=g) for c, s in ((8, 7), (16, 5))]
    f1 = [torch.randn(2, c, s, s, generator=g) for c, s in ((8, 7), (16, 5))]
    val = sum((1. - ns["cos_sim"](a, b)) for a, b in zip(f0, f1))
    np.savez_compressed(os.path.join(OUT, "parts_and_cossim.npz"), part_maps=np_(pm), centers=np_(cen),
                        grad_part_maps=np_(pm.grad), f0_0=np_(f0[0]), f0_1=np_(f0[1]), f1_0=np_(f1[0]),
                        f1_1=np_(f1[1]), cos_dist=np_(val))
    gen_save_obj(lib)
    print("done ->", OUT)


if __name__ == "__main__":
    main()
