"""Emulator check of an experimental backward build against the product's: the same forward state, every backward variant of both
libraries, gradients compared (sum order differs: lanes own other pixels).  usage: compare_host.py <base.so> <exp.so> [n_meshes] [IS subdiv]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_raster as HR  # noqa: E402
from helpers import scene  # noqa: E402
from oracle import torch_ref as TR  # noqa: E402
from umr_amd._lib import SIGNATURES  # noqa: E402


def load(so):
    L = ctypes.CDLL(so)
    for name, (argtypes, restype) in SIGNATURES.items():
        if hasattr(L, name):
            getattr(L, name).argtypes, getattr(L, name).restype = argtypes, restype
    return L


def main():
    base, exp = load(sys.argv[1]), load(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    IS, sub = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (512, 3)
    verts, faces, cams, g = scene(8, sub, seed=0)
    verts, faces, cams = verts[-n:], faces[-n:], cams[-n:]
    pv = TR.orthographic_proj_withz(verts, cams, offset_z=5.) * torch.tensor([1., -1., 1.])
    fv = np.ascontiguousarray(TR.face_vertices(TR.look_at_ortho(pv), faces).numpy(), np.float32)
    TS = 36
    DEL = float(np.float32(np.log(1. / 1e-10 - 1.)))
    rng = np.random.default_rng(1)
    tex = rng.random((n, faces.shape[1], TS, 3), dtype=np.float32)
    worst = 0.0

    def cmp(tag, a, b):
        nonlocal worst
        for name, x, y in zip(("grad_faces", "grad_textures"), a, b):
            if x is None:
                continue
            sc = np.abs(x).max()
            err = np.abs(x - y).max() / max(sc, 1e-30)
            worst = max(worst, err)
            print("%-44s %-14s scale %.3e  max|diff|/scale %.2e  equal bits %.4f" % (tag, name, sc, err, (x == y).mean()))

    for rgb_mode, label in ((1, "softmax"), (0, "hard")):
        out = HR.forward(fv, tex, IS, pooled=True, dist_eps_log=DEL, func_id_rgb=rgb_mode, L=base)
        for pooled in (True, False):
            g_rgb = rng.standard_normal((n, 4, IS // 2, IS // 2) if pooled else (n, 4, IS, IS)).astype(np.float32)
            fl = HR.BWD_GRAD_POOLED if pooled else 0
            for gf, gt in ((False, True), (True, True), (True, False)):
                r = [HR.backward(fv, tex, out["soft_colors"], out["aggrs_info"], g_rgb, IS, need_gf=gf, need_gt=gt, grad_flags=fl,
                                 dist_eps_log=DEL, func_id_rgb=rgb_mode, L=L) for L in (base, exp)]
                cmp("%s pooled=%d need_gf=%d need_gt=%d" % (label, pooled, gf, gt), r[0], r[1])
    outa = HR.forward(fv, None, IS, flags=HR.ALPHA_ONLY | HR.NO_P2F, pooled=True, dist_eps_log=DEL, L=base)
    for pooled in (True, False):
        g_a = rng.standard_normal((n, IS // 2, IS // 2) if pooled else (n, IS, IS)).astype(np.float32)
        fl = (HR.BWD_GRAD_POOLED if pooled else 0) | HR.BWD_ALPHA_ONLY
        r = [HR.backward(fv, None, outa["soft_colors"], None, g_a, IS, need_gf=True, need_gt=False, grad_flags=fl, dist_eps_log=DEL, L=L)
             for L in (base, exp)]
        cmp("silhouette pooled=%d" % pooled, r[0], r[1])
    # alpha-geometry pass of the experimental library (grad flag 4) against the product's silhouette backward on the render's
    # alpha plane + texel-only backward
    if os.environ.get("AG", "1") == "1":
        out = HR.forward(fv, tex, IS, pooled=True, dist_eps_log=DEL, L=base)
        for pooled in (True, False):
            g = rng.standard_normal((n, 4, IS // 2, IS // 2) if pooled else (n, 4, IS, IS)).astype(np.float32)
            fl = HR.BWD_GRAD_POOLED if pooled else 0
            _, gt_ref = HR.backward(fv, tex, out["soft_colors"], out["aggrs_info"], g, IS, need_gf=False, need_gt=True, grad_flags=fl, dist_eps_log=DEL, L=base)
            gf_ref, _ = HR.backward(fv, None, np.ascontiguousarray(out["soft_colors"][:, 3]), None, np.ascontiguousarray(g[:, 3]), IS, need_gf=True,
                                    need_gt=False, grad_flags=fl | HR.BWD_ALPHA_ONLY, dist_eps_log=DEL, L=base)
            gf, gt = HR.backward(fv, tex, out["soft_colors"], out["aggrs_info"], g, IS, need_gf=True, need_gt=True, grad_flags=fl | 4, dist_eps_log=DEL, L=exp)
            cmp("alpha-geometry pooled=%d" % pooled, (gf_ref, gt_ref), (gf, gt))
    print("worst max|diff|/scale: %.2e" % worst)
    return 0 if worst < 2e-5 else 1


if __name__ == "__main__":
    sys.exit(main())
