"""CPU: replay the render a tripwire run blamed (tools/r4/bench_trap.py -> gpurun_out/nan/repro_<pid>.pt) on the wave64 emulation
of the library built with the SAME settings (round 3's early ones by default: no reference-noise widening of the tile cull, no
thin-face route, exact_edges off), one view, and report what the raster backward makes of the offending face: which pixels
contribute, the saved forward state there, and where the non-finite value comes from.

usage: replay_nan.py repro.pt [--new]      (--new: the current settings, to show the same face is clean there)"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_raster as HR  # noqa: E402
from oracle import torch_ref as TR  # noqa: E402

CFG = dict(near=1., far=100., eps=1e-3, sigma_val=1e-5, dist_eps_log=float(math.log(1e10 - 1.)), gamma_val=1e-4, double_side=True, func_id_rgb=1)


def view_faces(d, step_rec, n, big):
    verts, faces = step_rec["pred_vs"], d["faces"].long()
    B = verts.shape[0]
    if big:                                   # the B*K-view launches: view n = image n // K, hypothesis n % K
        K = step_rec["cam_hypotheses"].shape[1]
        b, cam = n // K, step_rec["cam_hypotheses"][n // K, n % K]
    else:
        b, cam = n, step_rec["cam"][n]
    proj = TR.orthographic_proj_withz(verts[b:b + 1], cam[None], 5.) * torch.tensor([1., -1., 1.])
    return TR.face_vertices(TR.look_at_ortho(proj), faces[None]).reshape(1, -1, 9).numpy().astype(np.float32), b


def main():
    d = torch.load(sys.argv[1], weights_only=False)
    new = "--new" in sys.argv
    info, site = d["where"], d["site"]
    big, n, f = info >> 20, (info >> 13) & 127, info & 0x1fff
    print("site 0x%x, launch with N %s 32, view %d, face %d, first non-finite step %s" % (site, ">" if big else "<=", n, f, d["first_bad_step"]))
    flags = [] if new else ["-DTILE_CULL_NOISE=0.f", "-DTHIN_FACE_H=0.f"]
    L = HR.lib(HR.build(extra_flags=flags, out=os.path.join(HR.SRC_DIR, "libumr_host_%s.so" % ("new" if new else "old")))) if flags else HR.lib()
    L.umr_debug_set(b"exact_edges", 1 if new else 0)
    IS = 1024
    for s, rec in d["ring"]:
        fv, b = view_faces(d, rec, n, big)
        F = fv.shape[1]
        rng = np.random.default_rng(0)
        tex = rng.uniform(0, 1, (1, F, 36, 3)).astype(np.float32)
        o = HR.forward(fv, tex, IS, L=L, pooled=True, background_by_value=True, flags=HR.NO_P2F, **CFG)
        g = rng.standard_normal((1, 4, IS // 2, IS // 2)).astype(np.float32)
        gf, gt = HR.backward(fv, tex, o["soft_colors"], o["aggrs_info"], g, IS, need_gf=False, need_gt=True, grad_flags=HR.BWD_GRAD_POOLED, L=L, **CFG)
        ga = HR.backward(fv, None, o["soft_colors"][:, 3].copy(), None, g[:, 3].copy(), IS, need_gf=True, need_gt=False,
                         grad_flags=HR.BWD_GRAD_POOLED | HR.BWD_ALPHA_ONLY, L=L, **CFG)[0]
        bad_t = np.argwhere(~np.isfinite(gt).all(axis=(2, 3))[0]).ravel()
        bad_a = np.argwhere(~np.isfinite(ga).all(axis=2)[0]).ravel()
        print("step %d image %d: forward finite %s; texel-gradient faces non-finite %s; silhouette-gradient faces non-finite %s"
              % (s, b, bool(np.isfinite(o["soft_colors"]).all() and np.isfinite(o["aggrs_info"]).all()), bad_t[:10], bad_a[:10]))
        for ff in list(bad_t[:3]) + ([f] if f < F else []):
            p = fv[0, ff].reshape(3, 3)
            print("  face %d: corners (x, y, z) %s" % (ff, np.array2string(p, precision=6)))
            xs = (p[:, 0] * IS + IS - 1) / 2; ys = IS - 1 - (p[:, 1] * IS + IS - 1) / 2
            x0, x1, y0, y1 = int(max(xs.min() - 6, 0)), int(min(xs.max() + 6, IS - 1)), int(max(ys.min() - 6, 0)), int(min(ys.max() + 6, IS - 1))
            smax, ssum = o["aggrs_info"][0, 1, y0:y1 + 1, x0:x1 + 1], o["aggrs_info"][0, 0, y0:y1 + 1, x0:x1 + 1]
            print("    window rows %d-%d cols %d-%d: soft-max maximum min %.6g (eps = background only), sum min %.4g max %.4g"
                  % (y0, y1, x0, x1, smax.min(), ssum.min(), ssum.max()))


if __name__ == "__main__":
    main()
