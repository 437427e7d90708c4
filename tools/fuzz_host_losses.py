"""Fuzz the geometry / loss kernels of the library on the CPU (wave64 emulation, tests/host_raster.py::emulated_product) against
plain torch / the oracle's restatements, values AND gradients, on random RAGGED shapes (sizes that are not multiples of the
kernels' block, wave or tile sizes; single elements; empty masks): camera projection, silhouette IoU, chamfer (2-D / 3-D),
texture sampling (grid_sample through channel-last flows), texture-dt loss, masked-L1 texture loss, deformation / symmetry
regularisers, Laplacian and flatten losses on spheres of every subdivision, 2x bilinear up-sampling, the barrier distance
transform, the PNet cosine-distance head.

    python tools/fuzz_host_losses.py --cases 200 --seed 0
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_raster as HR          # noqa: E402
from oracle import torch_ref as TR  # noqa: E402


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    s = float(b.abs().max())
    return float((a - b).abs().max()) / max(s, 1e-6) if a.numel() else 0.0      # (1e-6: a gradient that is mathematically zero is rounding noise on both sides)


def grads(fn, inputs, seed):
    xs = [x.clone().requires_grad_(True) for x in inputs]
    out = fn(*xs)
    outs = out if isinstance(out, (tuple, list)) else (out,)
    g = torch.Generator().manual_seed(seed)
    tot = sum((o * torch.randn(o.shape, generator=g)).sum() for o in outs if torch.is_tensor(o) and o.is_floating_point())
    tot.backward()
    return [o.detach() for o in outs if torch.is_tensor(o)], [x.grad if x.grad is not None else torch.zeros_like(x) for x in xs]


def case_projection(rng, g):
    from umr_amd import functional as UF
    N, V = int(rng.integers(1, 5)), int(rng.integers(1, 700))
    v = torch.randn(N, V, 3, generator=g)
    cam = torch.cat([0.5 + torch.rand(N, 1, generator=g), 0.3 * torch.randn(N, 2, generator=g), torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=1)], 1)
    od = int(rng.choice([2, 3]))
    a = grads(lambda v_, c_: UF.ProjectPointsFunction.apply(v_, c_, od, 5.0 if od == 3 else 0.0), [v, cam], 1)
    b = grads(lambda v_, c_: (TR.orthographic_proj_withz(v_, c_, 5.0) if od == 3 else TR.orthographic_proj_withz(v_, c_, 0.0)[..., :2]), [v, cam], 1)
    return a, b, "N%d V%d out%d" % (N, V, od)


def case_iou(rng, g):
    from umr_amd import loss_utils as LU
    N, H, W = int(rng.integers(1, 6)), int(rng.integers(1, 90)), int(rng.integers(1, 90))
    p = torch.rand(N, H, W, generator=g)
    t = (torch.rand(N, H, W, generator=g) > float(rng.uniform(0, 1))).float()
    avg = bool(rng.integers(0, 2))
    return grads(lambda p_: LU.neg_iou_loss(p_, t, avg), [p], 2), grads(lambda p_: TR.neg_iou_loss(p_, t, avg), [p], 2), "N%d %dx%d avg%d" % (N, H, W, avg)


def case_chamfer(rng, g):
    from umr_amd.chamfer_python import distChamfer
    B, n, m, d = int(rng.integers(1, 4)), int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.choice([2, 3]))
    a, b = torch.randn(B, n, d, generator=g), torch.randn(B, m, d, generator=g)

    def f(mod):
        def run(a_, b_):
            d1, d2, i1, i2 = mod(a_, b_)
            return d1, d2, i1.double(), i2.double()
        return run
    A, Bq = grads(f(distChamfer), [a, b], 3), grads(f(TR.dist_chamfer), [a, b], 3)
    return A, Bq, "B%d n%d m%d d%d" % (B, n, m, d)


def case_sample_textures(rng, g):
    from umr_amd.geom_utils import sample_textures
    B, F, T, H, W = int(rng.integers(1, 3)), int(rng.integers(1, 200)), int(rng.integers(1, 7)), int(rng.integers(2, 70)), int(rng.integers(2, 70))
    flow = torch.rand(B, F, T, T, 2, generator=g) * 2.4 - 1.2          # partly outside [-1, 1]: zero padding
    img = torch.rand(B, 3, H, W, generator=g)
    return grads(sample_textures, [flow, img], 4), grads(TR.sample_textures, [flow, img], 4), "B%d F%d T%d %dx%d" % (B, F, T, H, W)


def case_texture_dt(rng, g):
    from umr_amd import loss_utils as LU
    B, F, T, H = int(rng.integers(1, 3)), int(rng.integers(1, 150)), int(rng.integers(1, 7)), int(rng.integers(2, 70))
    flow = torch.rand(B, F, T, T, 2, generator=g) * 2.2 - 1.1
    dt = torch.rand(B, 1, H, H, generator=g)
    return grads(lambda f_: LU.texture_dt_loss(f_, dt), [flow], 5), grads(lambda f_: TR.texture_dt_loss(f_, dt), [flow], 5), "B%d F%d T%d H%d" % (B, F, T, H)


def case_masked_l1(rng, g):
    from umr_amd import loss_utils as LU
    B, H, W = int(rng.integers(1, 5)), int(rng.integers(1, 80)), int(rng.integers(1, 80))
    ip, ig = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g)
    mg = (torch.rand(B, H, W, generator=g) > float(rng.uniform(0, 1))).float()
    mp = torch.rand(B, H, W, generator=g)
    avg = bool(rng.integers(0, 2))
    return (grads(lambda a_, b_: LU.texture_loss_masks(a_, ig, mg, b_, avg), [ip, mp], 6),
            grads(lambda a_, b_: TR.texture_loss_masks(a_, ig, mg, b_, avg), [ip, mp], 6), "B%d %dx%d avg%d" % (B, H, W, avg))


def case_regs(rng, g):
    from umr_amd import loss_utils as LU
    B, V = int(rng.integers(1, 6)), int(rng.integers(1, 3000))
    x = torch.randn(B, V, 3, generator=g)
    return (grads(lambda x_: (LU.deform_l2reg(x_), LU.sym_reg(x_)), [x], 7), grads(lambda x_: (TR.deform_l2reg(x_), TR.sym_reg(x_)), [x], 7), "B%d V%d" % (B, V))


def case_mesh_losses(rng, g):
    from umr_amd import loss_utils as LU
    from umr_amd.mesh import create_sphere
    sub, B = int(rng.integers(0, 4)), int(rng.integers(1, 4))
    v, f = create_sphere(sub)
    vt, ft = torch.from_numpy(v).float(), torch.from_numpy(f).long()
    x = vt[None].repeat(B, 1, 1) + 0.1 * torch.randn(B, vt.shape[0], 3, generator=g)
    avg = bool(rng.integers(0, 2))
    lap_p, lap_r = LU.LaplacianLoss(vt, ft, avg), TR.LaplacianLoss(vt, ft, avg)
    fl_p, fl_r = LU.FlattenLoss(ft, avg), TR.FlattenLoss(ft, avg)
    return grads(lambda x_: (lap_p(x_), fl_p(x_)), [x], 8), grads(lambda x_: (lap_r(x_), fl_r(x_)), [x], 8), "subdiv%d B%d avg%d" % (sub, B, avg)


def case_upsample(rng, g):
    from umr_amd import functional as UF
    B, C, H, W = int(rng.integers(1, 3)), int(rng.integers(1, 9)), int(rng.integers(1, 40)), int(rng.integers(1, 40))
    x = torch.randn(B, C, H, W, generator=g)
    return (grads(UF.Upsample2xBilinearFunction.apply, [x], 9),
            grads(lambda x_: torch.nn.functional.interpolate(x_, scale_factor=2, mode="bilinear", align_corners=False), [x], 9), "%dx%dx%dx%d" % (B, C, H, W))


def case_dt(rng, g):
    from umr_amd.image_utils import compute_dt_barrier
    B, H, W = int(rng.integers(1, 3)), int(rng.integers(1, 130)), int(rng.integers(1, 130))
    m = (torch.rand(B, H, W, generator=g) > float(rng.choice([0.03, 0.3, 0.7, 0.97]))).float()
    if H * W >= 2:      # a mask with no foreground or no background is undefined input for the reference (scipy measures to a
        m[:, 0, 0] = 1.0  # virtual feature whose place is an artefact of its implementation): at least one pixel of each
        m[:, -1, -1] = 0.0
    a = compute_dt_barrier(m)
    b = torch.stack([torch.from_numpy(np.asarray(TR.compute_dt_barrier(m[i].numpy())[0])).float() for i in range(B)])
    return ([a], []), ([b], []), "B%d %dx%d fg%.2f" % (B, H, W, float(m.mean()))


def case_cos(rng, g):
    from umr_amd import functional as UF
    N, taps = int(rng.integers(1, 4)), int(rng.integers(1, 6))
    # (C >= 2: with one channel the cosine is +-1 identically and its gradient is rounding noise around zero on both sides)
    f0 = [torch.randn(N, int(rng.integers(2, 70)), int(rng.integers(1, 20)), int(rng.integers(1, 20)), generator=g) for _ in range(taps)]
    f1 = [torch.randn(f.shape, generator=g) for f in f0]

    def prod(*fs):
        return UF.CosSimDistanceFunction.apply(1e-10, *fs)
    def ref(*fs):
        k = len(fs) // 2
        return TR.cos_sim_distance(list(fs[:k]), list(fs[k:]))
    return grads(prod, f0 + f1, 10), grads(ref, f0 + f1, 10), "N%d taps%d" % (N, taps)


CASES = dict(projection=case_projection, iou=case_iou, chamfer=case_chamfer, sample_textures=case_sample_textures, texture_dt=case_texture_dt,
             masked_l1=case_masked_l1, regs=case_regs, mesh_losses=case_mesh_losses, upsample=case_upsample, dt_barrier=case_dt, cos_head=case_cos)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=50)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(2)
    HR.lib(HR.build())
    names = [n for n in CASES if not a.only or n in a.only.split(",")]
    stats = {n: dict(cases=0, value_dev_max=0.0, grad_dev_max=0.0, nonfinite=0, flagged=0) for n in names}
    t0 = time.time()
    with HR.emulated_product():
        for i in range(a.cases):
            name = names[i % len(names)]
            rng = np.random.default_rng([a.seed, i])
            g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
            try:
                (po, pg), (ro, rg), desc = CASES[name](rng, g)
            except Exception as e:
                print("case %d %s: EXCEPTION %s: %s" % (i, name, type(e).__name__, str(e)[:300]), flush=True)
                stats[name]["flagged"] += 1
                continue
            st = stats[name]
            st["cases"] += 1
            fin = all(bool(torch.isfinite(t).all()) for t in po + pg)
            finr = all(bool(torch.isfinite(t).all()) for t in ro + rg)
            if finr and not fin:
                st["nonfinite"] += 1
            dv = max([rel(x, y) for x, y in zip(po, ro)] + [0.0])
            dg = max([rel(x, y) for x, y in zip(pg, rg)] + [0.0])
            if finr:
                st["value_dev_max"], st["grad_dev_max"] = max(st["value_dev_max"], dv), max(st["grad_dev_max"], dg)
            if (finr and not fin) or dv > 1e-4 or dg > 1e-3:
                st["flagged"] += 1
                print("case %d %s (%s; rng [%d, %d]): value dev %.3g, grad dev %.3g%s" % (i, name, desc, a.seed, i, dv, dg, "" if fin else " NON-FINITE"), flush=True)
    res = dict(seed=a.seed, cases=a.cases, seconds=round(time.time() - t0, 1), kernels=stats)
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
