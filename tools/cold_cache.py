"""GPU box: the step's raster launches with COLD caches -- inside the training step the network's kernels run between a render's
forward and its backward and between consecutive renders, so the saved state / gradient planes come from HBM, not from the
256 MB Infinity Cache as in a back-to-back microbenchmark.  Fixed SURVEY 8d scene; before every raster launch a 1 GiB buffer is
overwritten (evicts L2 and the Infinity Cache).  Library-owned HIP events, us per launch, warm vs cold.  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import scene  # noqa: E402
from umr_amd import _lib, functional as UF  # noqa: E402

if os.environ.get("UMR_LIB_FILE"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UMR_LIB_FILE"])
dev = torch.device("cuda:0")
N, IS, TS, iters = 16, 512, 36, 10
verts, faces, cams, gen = scene(2 * N, 3, seed=0)
_, fv, _ = UF.project_faces(verts.to(dev), cams.to(dev), faces.int().to(dev), 5.0, -2.732)
fv = fv.detach()
tex = torch.rand(N, faces.shape[1], TS, 3, generator=gen).to(dev).requires_grad_(True)
fa, fb = fv[:N].clone().requires_grad_(True), fv[N:].clone().requires_grad_(True)
g_tex, g_sil = torch.randn(N, 4, IS // 2, IS // 2, generator=gen).to(dev), torch.randn(N, IS // 2, IS // 2, generator=gen).to(dev)
args = (IS, [0, 0, 0], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, 'softmax', 'prod', 'surface')
trash = torch.empty(256 * 1024 * 1024, device=dev)
out = {}
for cold in (False, True, False, True):
    flush = (lambda: trash.fill_(1.0)) if cold else (lambda: None)
    for phase in range(2):
        if phase:
            _lib.profile_enable(True)
            for k in range(4):
                _lib.profile_collect(k)
        for _ in range(iters if phase else 2):
            tex.grad = None; fa.grad = None; fb.grad = None
            flush(); sc = UF.soft_rasterize(fa, tex, *args, pool=True, need_p2f=True, want_visibility=True, detach_rgb_geometry=True, lean_state=True)[0]
            flush(); a = UF.silhouette(fb, IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, True)
            flush(); sc.backward(g_tex)
            flush(); a.backward(g_sil)
        torch.cuda.synchronize()
    _lib.profile_enable(False)
    for name, k in (("fwd_packed", 0), ("agp_bwd", 1), ("sil_fwd16", 2), ("sil_bwd16", 3)):
        ms, n, _ = _lib.profile_collect(k)
        out.setdefault(("cold_" if cold else "warm_") + name, []).append(round(1e3 * ms / max(n, 1), 1))
print(json.dumps({"lib": os.path.basename(_lib.LIB_PATH), "set": os.environ.get("UMR_DEBUG_SET", ""), "us": out}), flush=True)
