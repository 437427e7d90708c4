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
    g_tex = torch.randn(N, 4, H, H, device=dev)
    g_sil = torch.randn(2 * N, H, H, device=dev)
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
        _lib.profile_enable(False)
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
