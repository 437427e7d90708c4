"""GPU box: the backward kernels of THIS tree's libumr_hip.so against a reference build of the library (the round-4 kernels that the
GPU suite, the oracle and the reference's device code pinned), both loaded in one process and driven through the C ABI on the same
device buffers: every backward variant x image shapes (power of two / ragged / odd), pooled and full-resolution gradients,
one- and two-sided, soft-max and hard colour, TS 1 / 9 / 36, texture groups, the configs[3] shape -- and the one-pass
alpha-geometry backward against the reference build's silhouette + texel-only pair.  Differences are summation order only.
usage: differential_gpu.py <reference.so> [tolerance=3e-6]      exit status 0 = all within tolerance (of the largest element)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers import scene  # noqa: E402
from umr_amd import _lib, functional as UF, ops as O  # noqa: E402


def load(path):
    L = ctypes.CDLL(path)
    for name, (argtypes, restype) in _lib.SIGNATURES.items():
        if hasattr(L, name):
            getattr(L, name).argtypes, getattr(L, name).restype = argtypes, restype
    L.umr_build_id.restype = ctypes.c_char_p
    return L


def main():
    ref = load(os.path.abspath(sys.argv[1]))
    new = _lib.lib()
    tol = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-6
    print("reference build %s, this build %s" % (ref.umr_build_id().decode(), _lib.build_id()))
    dev = torch.device("cuda:0")
    worst, n_cases = 0.0, 0
    p = O.ptr

    def backward(L, fv, tex, sc, ag, g, IS, flags, need_gf, need_gt, rgb, two_sided, K=1):
        N, F = fv.shape[:2]
        TS = tex.shape[2] if tex is not None else 1
        gf = torch.zeros(N, F, 9, device=dev) if need_gf else None
        gt = torch.zeros(N, F, TS, 3, device=dev) if need_gt else None
        wsb = L.umr_raster_workspace_bytes_for(N, F, IS)
        ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
        scal = O._scalars(IS, 1., 100., two_sided, 1e-3, 1e-5, 1e-10, 1e-4, O.pack_modes(rgb))
        rc = L.umr_raster_backward(p(fv), p(tex), p(sc), None, p(ag), p(gf), p(gt), p(g), flags | ((K << 8) if K > 1 else 0), int(need_gf), int(need_gt),
                                   N, F, TS, *scal, p(ws), wsb, None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        return gf, gt

    def check(tag, a, b):
        nonlocal worst, n_cases
        for name, x, y in zip(("vertices", "texels"), a, b):
            if x is None:
                continue
            s = float(y.abs().max())
            e = float((x - y).abs().max()) / max(s, 1e-30)
            ok = e <= tol and bool(torch.isfinite(x).all())
            worst = max(worst, e); n_cases += 1
            print("%-86s %-8s scale %.2e  diff %.1e %s" % (tag, name, s, e, "" if ok else "  <-- FAIL"))

    shapes = [(4, 2, 64, 9), (3, 2, 50, 4), (2, 2, 33, 1), (2, 3, 256, 36), (16, 3, 512, 36), (2, 4, 1024, 36), (5, 1, 16, 1), (1, 2, 8, 4)]
    for (N, sub, IS, TS) in shapes:
        verts, faces, cams, gen = scene(N, sub, seed=100 + IS)
        _, fv, _ = UF.project_faces(verts.to(dev), cams.to(dev), faces.int().to(dev), 5.0, -2.732)
        fv = fv.detach().contiguous()
        F = faces.shape[1]
        for rgb in (1, 0):
            for two_sided in (True, False):
                tex = torch.rand(N, F, TS, 3, generator=gen).to(dev)
                outs = O._raster_forward(fv, tex, IS, [0., 0., 0.], 1., 100., two_sided, 1e-3, 1e-5, 1e-10, 1e-4, O.pack_modes(rgb), False, True, False)
                sc, ag = outs[0].contiguous(), outs[2].contiguous()
                for pooled in ((True, False) if IS % 2 == 0 else (False,)):
                    H = IS // 2 if pooled else IS
                    g = torch.randn(N, 4, H, H, generator=gen).to(dev)
                    fl = 1 if pooled else 0
                    tag = "N %d F %d IS %d TS %d %s %s %s" % (N, F, IS, TS, "soft-max" if rgb else "hard", "two-sided" if two_sided else "front only", "pooled" if pooled else "full-res")
                    for gfw, gtw in ((True, True), (True, False), (False, True)):
                        check(tag + " gf%d gt%d" % (gfw, gtw), backward(new, fv, tex, sc, ag, g, IS, fl, gfw, gtw, rgb, two_sided),
                              backward(ref, fv, tex, sc, ag, g, IS, fl, gfw, gtw, rgb, two_sided))
                    if rgb == 1:
                        alpha, ga = sc[:, 3].contiguous(), g[:, 3].contiguous()
                        s_new = backward(new, fv, None, alpha, None, ga, IS, fl | 2, True, False, 1, two_sided)
                        s_ref = backward(ref, fv, None, alpha, None, ga, IS, fl | 2, True, False, 1, two_sided)
                        check(tag + " silhouette", s_new, s_ref)
                        t_ref = backward(ref, fv, tex, sc, ag, g, IS, fl, False, True, 1, two_sided)
                        one = backward(new, fv, tex, sc, ag, g, IS, fl | O.BWD_ALPHA_GEOMETRY, True, True, 1, two_sided)
                        check(tag + " one-pass vs silhouette + texel-only", one, (s_ref[0], t_ref[1]))
        if N % 2 == 0:     # two views per texture set
            tex2 = torch.rand(N // 2, F, TS, 3, generator=gen).to(dev)
            outs = O._raster_forward(fv, tex2, IS, [0., 0., 0.], 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, O.pack_modes(1), False, True, False)
            sc, ag = outs[0].contiguous(), outs[2].contiguous()
            g = torch.randn(N, 4, IS, IS, generator=gen).to(dev)
            t_ref = backward(ref, fv, tex2, sc, ag, g, IS, 0, False, True, 1, True, K=2)
            s_ref = backward(ref, fv, None, sc[:, 3].contiguous(), None, g[:, 3].contiguous(), IS, 2, True, False, 1, True)
            one = backward(new, fv, tex2, sc, ag, g, IS, O.BWD_ALPHA_GEOMETRY, True, True, 1, True, K=2)
            check("N %d F %d IS %d TS %d two views per texture set, one-pass" % (N, F, IS, TS), one, (s_ref[0], t_ref[1]))
    print("%d comparisons, worst difference %.2e of the largest element (tolerance %.1e)" % (n_cases, worst, tol))
    return 0 if worst <= tol else 1


if __name__ == "__main__":
    sys.exit(main())
