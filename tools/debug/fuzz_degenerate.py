"""GPU box: does the raster forward / backward stay finite on degenerate but FINITE faces?  (The tripwire build named the
forward as the first producer of a non-finite value in two diverged bench runs, with finite vertices going in.)
One face per mesh (F = 1), thousands of meshes, IS = 64: every pixel of a mesh's image is that face's doing."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from umr_amd import functional as UF

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(int(os.environ.get("SEED", 0)))
IS = 64
px = (2 * torch.arange(IS) + 1 - IS).float() / IS          # pixel-centre coordinates
def rnd(n, *s): return torch.rand(n, *s, generator=g) * 2 - 1
cases = {}
n = 4000
a, b = rnd(n, 2), rnd(n, 2)
t = torch.rand(n, 1, generator=g)
cases["collinear"] = torch.stack([a, b, a + t * (b - a)], 1)                       # third vertex ON the segment
cases["collinear_outside"] = torch.stack([a, b, a + (1 + t) * (b - a)], 1)
cases["two_equal"] = torch.stack([a, a, b], 1)
cases["two_equal_b"] = torch.stack([a, b, b], 1)
cases["two_equal_c"] = torch.stack([a, b, a], 1)
cases["all_equal"] = torch.stack([a, a, a], 1)
cases["tiny"] = torch.stack([a, a + 1e-7 * rnd(n, 2), a + 1e-7 * rnd(n, 2)], 1)
cases["sliver"] = torch.stack([a, b, a + t * (b - a) + 1e-8 * rnd(n, 2)], 1)
pc = px[torch.randint(0, IS, (n, 2), generator=g)]
cases["vertex_on_pixel_centre"] = torch.stack([pc, b, rnd(n, 2)], 1)
cases["edge_through_pixel_centres"] = torch.stack([pc, px[torch.randint(0, IS, (n, 2), generator=g)], rnd(n, 2)], 1)
cases["axis_aligned"] = torch.stack([pc, torch.stack([pc[:, 0], b[:, 1]], 1), torch.stack([b[:, 0], pc[:, 1]], 1)], 1)
cases["huge"] = torch.stack([a * 50, b * 50, rnd(n, 2) * 50], 1)
cases["random"] = torch.stack([a, b, rnd(n, 2)], 1)
report = {}
for name, xy in cases.items():
    for zmode in ("equal", "random"):
        z = torch.full((n, 3, 1), 7.7) if zmode == "equal" else 7.7 + rnd(n, 3, 1)
        fv = torch.cat([xy, z], 2).view(n, 1, 3, 3).to(dev)
        tex = torch.rand(n, 1, 36, 3, generator=g).to(dev).requires_grad_(True)
        fvg = fv.clone().requires_grad_(True)
        sc, p2f, aggr = UF.soft_rasterize(fvg, tex, IS, [0.1, 0.2, 0.3], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, 'softmax')
        sc.sum().backward()
        al = UF.SilhouetteFunction.apply(fv, IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, False)
        bad = {"soft_colors": int((~torch.isfinite(sc)).flatten(1).any(1).sum()), "aggrs": int((~torch.isfinite(aggr)).flatten(1).any(1).sum()),
               "p2f": int((~torch.isfinite(p2f)).flatten(1).any(1).sum()), "alpha": int((~torch.isfinite(al)).flatten(1).any(1).sum()),
               "grad_faces": int((~torch.isfinite(fvg.grad)).flatten(1).any(1).sum()), "grad_tex": int((~torch.isfinite(tex.grad)).flatten(1).any(1).sum())}
        if any(bad.values()):
            which = (~torch.isfinite(sc)).flatten(1).any(1) | (~torch.isfinite(al)).flatten(1).any(1) | (~torch.isfinite(aggr)).flatten(1).any(1)
            ex = fv[which][:2].flatten(1).tolist() if bool(which.any()) else []
            report[name + "/" + zmode] = dict(bad, meshes=n, examples=ex)
print(json.dumps(report if report else {"all finite": True, "cases": list(cases)}, indent=None))
