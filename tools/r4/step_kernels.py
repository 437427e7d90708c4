"""GPU box: the four raster launches of one train_s1 step (bs 16) back to back, a few times, for rocprofv3 PMC passes and for
HIP-event timing (library-owned events, umr_profile_*):
  textured soft-max forward with visibility planes + fused pool (N = 16, TS = 36), its texel-gradient-only backward,
  silhouette forward / backward over 2 x 16 views.
usage: step_kernels.py [iters] [scale_lo scale_hi] [N] [IS subdiv]      (scale = camera scale range of tests.helpers.scene; bench.py's
networks start near 1.0: the mesh fills the frame; IS 1024 subdiv 4 = BASELINE configs[3]'s raster shape)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import scene  # noqa: E402
from umr_amd import _lib, functional as UF  # noqa: E402


def run(iters=3, scale=(0.6, 0.9), N=16, IS=512, subdiv=3, TS=36, timed=True):
    dev = torch.device("cuda:0")
    verts, faces, cams, gen = scene(2 * N, subdiv, seed=0, scale=scale)
    _, fv2, _ = UF.ProjectFacesFunction.apply(verts.to(dev), cams.to(dev), faces.int().to(dev), 5.0, -2.732, False)
    fv2 = fv2.detach()
    F = faces.shape[1]
    fv_tex = fv2[:N].clone()
    tex = torch.rand(N, F, TS, 3, generator=gen).to(dev).requires_grad_(True)
    fv_sil = fv2.clone().requires_grad_(True)
    H = IS // 2
    g_tex = torch.randn(N, 4, H, H, generator=gen).to(dev)
    g_sil = torch.randn(2 * N, H, H, generator=gen).to(dev)
    args = (IS, [0, 0, 0], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, 'softmax', 'prod', 'surface', True, True, True)
    out = {}
    for phase in ("warm", "timed"):
        if phase == "timed" and timed:
            _lib.profile_enable(True)
            for i in range(4):
                _lib.profile_collect(i)
        for _ in range(2 if phase == "warm" else iters):
            tex.grad = None; fv_sil.grad = None
            sc = UF.soft_rasterize(fv_tex, tex, *args)[0]
            a = UF.SilhouetteFunction.apply(fv_sil, IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, True)
            sc.backward(g_tex)
            a.backward(g_sil)
        torch.cuda.synchronize()
    if timed:
        for name, i in (("tex_fwd", 0), ("tex_bwd_texel_only", 1), ("sil_fwd", 2), ("sil_bwd", 3)):
            ms, n, _ = _lib.profile_collect(i)
            out[name] = round(ms * 1e3 / max(n, 1), 1)
        # the vertex + texel variant (train_s2's textured renders): same scene, geometry gradients wanted as well
        fv_full = fv_tex.clone().requires_grad_(True)
        for phase in ("warm", "timed"):
            if phase == "timed":
                _lib.profile_collect(0); _lib.profile_collect(1)
            for _ in range(2 if phase == "warm" else iters):
                tex.grad = None; fv_full.grad = None
                UF.soft_rasterize(fv_full, tex, *args)[0].backward(g_tex)
            torch.cuda.synchronize()
        ms, n, _ = _lib.profile_collect(1)
        out["tex_bwd_full"] = round(ms * 1e3 / max(n, 1), 1)
        out["fp_full_vertex_grad"] = float(fv_full.grad.double().abs().sum())
        if os.environ.get("UMR_AG") == "1":
            # experiment (tools/exp_quads): ONE backward pass for the shared mask / texture render (grad flag 4: alpha gradient ->
            # geometry, rgb gradient -> texels) against the two launches the product makes for it
            from umr_amd import ops as O
            L = _lib.lib()
            modes = O.pack_modes(1)
            outs = O._raster_forward(fv_tex, tex.detach(), IS, [0., 0., 0.], 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, modes, True, True, True)
            soft_colors, aggrs = outs[3], outs[2]
            cfg = (IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, modes, True)
            for phase in ("warm", "timed"):
                if phase == "timed":
                    _lib.profile_collect(1); _lib.profile_collect(3)
                for _ in range(2 if phase == "warm" else iters):
                    _, gt_ref = torch.ops.umr.soft_rasterize_backward(fv_tex, tex.detach(), soft_colors, aggrs, g_tex, *cfg, False, True)
                    gf_ref = torch.ops.umr.silhouette_backward(fv_tex, soft_colors[:, 3].contiguous(), g_tex[:, 3].contiguous(), IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, True)
                torch.cuda.synchronize()
            ms, n, _ = _lib.profile_collect(1); out["ag_ref_texel_only"] = round(ms * 1e3 / max(n, 1), 1)
            ms, n, _ = _lib.profile_collect(3); out["ag_ref_sil_alpha_N16"] = round(ms * 1e3 / max(n, 1), 1)
            N_, F_ = fv_tex.shape[:2]
            ws_bytes = L.umr_raster_workspace_bytes_for(N_, F_, IS)
            ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
            scal = O._scalars(*cfg[:-1])
            for phase in ("warm", "timed"):
                if phase == "timed":
                    _lib.profile_collect(1)
                for _ in range(2 if phase == "warm" else iters):
                    gf = torch.zeros(N_, F_, 9, device=dev); gt = torch.zeros(N_, F_, TS, 3, device=dev)
                    rc = L.umr_raster_backward(O.ptr(fv_tex.contiguous()), O.ptr(tex.detach()), O.ptr(soft_colors), None, O.ptr(aggrs), O.ptr(gf), O.ptr(gt),
                                               O.ptr(g_tex), 1 | 4, 1, 1, N_, F_, TS, *scal, O.ptr(ws), ws_bytes, _lib.stream_ptr(dev))
                    assert rc == 0, rc
                torch.cuda.synchronize()
            ms, n, _ = _lib.profile_collect(1); out["ag_bwd"] = round(ms * 1e3 / max(n, 1), 1)
            out["ag_err_vertex"] = float((gf.view_as(gf_ref) - gf_ref).abs().max() / gf_ref.abs().max())
            out["ag_err_texel"] = float((gt - gt_ref).abs().max() / gt_ref.abs().max())
        _lib.profile_enable(False)
    # gradient fingerprints: an experimental build must reproduce them to summation-order rounding
    for name, t in (("texel_grad", tex.grad), ("sil_vertex_grad", fv_sil.grad)):
        d = t.double()
        out["fp_" + name] = [float(d.sum()), float(d.abs().sum()), float(d.abs().max()), float((d * torch.arange(d.numel(), device=dev).view(d.shape).remainder(97)).sum())]
    return out


if os.environ.get("UMR_LIB_FILE"):      # an experimental build (tools/r4/build_asm.py, build_variants.py)
    _lib.LIB_PATH = os.path.abspath(os.environ["UMR_LIB_FILE"])

if __name__ == "__main__":
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    sc = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.6, 0.9)
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    IS_, sub_ = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (512, 3)
    for kv in os.environ.get("UMR_DEBUG_SET", "").split(","):      # e.g. UMR_DEBUG_SET=exact_edges=0,face_order=0
        if "=" in kv:
            _lib.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
    print(json.dumps({"lib": os.path.basename(_lib.LIB_PATH), "set": os.environ.get("UMR_DEBUG_SET", ""), "scale": sc, "N": n, "IS": IS_, "subdiv": sub_, "us_per_launch": run(it, sc, n, IS_, sub_)}), flush=True)
