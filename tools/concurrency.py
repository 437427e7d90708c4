"""GPU box: what running the two independent raster launches of each direction of a train_s1 step on TWO streams would buy --
the shared render (textured, packed state, N = 16) and the unseen-view silhouette (N = 16) forward, and their backwards, timed as
pairs on one stream and on two (wall clock over `iters` pairs, fixed SURVEY 8d scene).  One JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import scene  # noqa: E402
from umr_amd import functional as UF  # noqa: E402

dev = torch.device("cuda:0")
N, IS, TS, iters = 16, 512, 36, 50
verts, faces, cams, gen = scene(2 * N, 3, seed=0)
_, fv, _ = UF.project_faces(verts.to(dev), cams.to(dev), faces.int().to(dev), 5.0, -2.732)
fv = fv.detach()
tex = torch.rand(N, faces.shape[1], TS, 3, generator=gen).to(dev).requires_grad_(True)
fa, fb = fv[:N].clone().requires_grad_(True), fv[N:].clone().requires_grad_(True)
g_tex, g_sil = torch.randn(N, 4, IS // 2, IS // 2, generator=gen).to(dev), torch.randn(N, IS // 2, IS // 2, generator=gen).to(dev)
args = (IS, [0, 0, 0], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, 'softmax', 'prod', 'surface')
side = torch.cuda.Stream()
cur = torch.cuda.current_stream()


def fwd_pair(two):
    if two:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            a = UF.silhouette(fb, IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, True)
        sc = UF.soft_rasterize(fa, tex, *args, pool=True, need_p2f=True, want_visibility=True, detach_rgb_geometry=True, lean_state=True)[0]
        cur.wait_stream(side)
    else:
        sc = UF.soft_rasterize(fa, tex, *args, pool=True, need_p2f=True, want_visibility=True, detach_rgb_geometry=True, lean_state=True)[0]
        a = UF.silhouette(fb, IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, True)
    return sc, a


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / iters


out = {}
for two in (False, True, False, True):
    out.setdefault("fwd_pair_us_%s" % ("two_streams" if two else "one_stream"), []).append(round(timed(lambda: fwd_pair(two)), 1))
# backward pairs: build the graphs once per timing call (forward outside the clock is impossible with autograd; time fwd + bwd and subtract)
for two in (False, True, False, True):
    def both():
        sc, a = fwd_pair(two)
        tex.grad = None; fa.grad = None; fb.grad = None
        if two:     # autograd runs each backward on its forward's stream
            torch.autograd.backward([sc, a], [g_tex, g_sil])
        else:
            torch.autograd.backward([sc, a], [g_tex, g_sil])
    out.setdefault("fwd_bwd_pairs_us_%s" % ("two_streams" if two else "one_stream"), []).append(round(timed(both), 1))
print(json.dumps(out), flush=True)
