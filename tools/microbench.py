"""Kernel micro-benchmark (GPU box): times umr_raster_forward/backward with HIP events at the shapes of
SURVEY.md section 8d and prints algorithmic GB/s per kernel."""
import json
import sys

import torch

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import scene  # noqa: E402
from umr_amd import functional as UF  # noqa: E402


def bench(N, subdiv, IS, TS, rgb="softmax", iters=10, need_gf=True, need_gt=True, pool=False, need_p2f=True):
    dev = torch.device("cuda:0")
    verts, faces, cams, gen = scene(N, subdiv, seed=0)
    _, fv, _ = UF.ProjectFacesFunction.apply(verts.to(dev), cams.to(dev), faces.int().to(dev), 5.0, -2.732, False)
    F = faces.shape[1]
    tex = torch.rand(N, F, TS, 3, generator=gen).to(dev)
    fv = fv.detach().requires_grad_(need_gf)
    tex = tex.requires_grad_(need_gt)
    H = IS // 2 if pool else IS
    g = torch.randn(N, 4, H, H, device=dev)
    args = (IS, [0, 0, 0], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, rgb, 'prod', 'surface', pool, need_p2f)
    tfs, tbs = [], []
    for it in range(iters + 2):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        fv.grad = None; tex.grad = None
        e0.record()
        sc, _, _ = UF.soft_rasterize(fv, tex, *args)
        e1.record()
        if need_gf or need_gt:
            sc.backward(g)
        e2.record()
        torch.cuda.synchronize()
        if it >= 2:
            tfs.append(e0.elapsed_time(e1)); tbs.append(e1.elapsed_time(e2))
    tf, tb = sorted(tfs)[len(tfs) // 2], sorted(tbs)[len(tbs) // 2]   # median: an allocator growth inside one iteration
                                                                       # (hipMalloc) would otherwise skew the mean
    fwd_bytes = N * (24 * IS * IS + F * (36 + 12 * TS + 16))
    bwd_bytes = N * (40 * IS * IS + F * (180 + 24 * TS))
    return dict(N=N, F=F, IS=IS, TS=TS, rgb=rgb, pool=pool, fwd_ms=round(tf, 4), bwd_ms=round(tb, 4),
                fwd_GBs=round(fwd_bytes / tf / 1e6, 1), bwd_GBs=round(bwd_bytes / tb / 1e6, 1),
                fwd_us_per_mesh=round(tf * 1e3 / N, 2), bwd_us_per_mesh=round(tb * 1e3 / N, 2))


if __name__ == "__main__" and "--alpha" not in sys.argv:
    print(torch.cuda.get_device_name(0))
    for cfg in [(16, 3, 512, 1), (16, 3, 512, 36), (128, 3, 512, 1), (128, 3, 512, 36)]:
        print(json.dumps(bench(*cfg)), flush=True)
    print("no-p2f + fused pool:", json.dumps(bench(128, 3, 512, 1, pool=True, need_p2f=False)), flush=True)
    print("no-p2f + fused pool:", json.dumps(bench(128, 3, 512, 36, pool=True, need_p2f=False, need_gf=False)), flush=True)
    print(json.dumps(bench(16, 3, 512, 1, rgb="hard", need_gf=False, need_gt=False)), flush=True)
    print(json.dumps(bench(32, 4, 1024, 36)), flush=True)


def bench_alpha(N, subdiv, IS, iters=10, pool=True):
    dev = torch.device("cuda:0")
    verts, faces, cams, gen = scene(N, subdiv, seed=0)
    _, fv, _ = UF.ProjectFacesFunction.apply(verts.to(dev), cams.to(dev), faces.int().to(dev), 5.0, -2.732, False)
    fv = fv.detach().requires_grad_(True)
    H = IS // 2 if pool else IS
    g = torch.randn(N, H, H, device=dev)
    tfs, tbs = [], []
    for it in range(iters + 2):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        fv.grad = None
        e0.record()
        a = UF.SilhouetteFunction.apply(fv, IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, pool)
        e1.record()
        a.backward(g)
        e2.record()
        torch.cuda.synchronize()
        if it >= 2:
            tfs.append(e0.elapsed_time(e1)); tbs.append(e1.elapsed_time(e2))
    tf, tb = sorted(tfs)[len(tfs) // 2], sorted(tbs)[len(tbs) // 2]
    return dict(kind="alpha_only", N=N, IS=IS, fwd_us_per_mesh=round(tf * 1e3 / N, 2), bwd_us_per_mesh=round(tb * 1e3 / N, 2))


if __name__ == "__main__" and "--alpha" in sys.argv:
    print(json.dumps(bench_alpha(16, 3, 512)), flush=True)
    print(json.dumps(bench_alpha(128, 3, 512)), flush=True)
