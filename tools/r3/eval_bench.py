"""GPU box: BASELINE configs[4] -- "test_kp (10k pairs)" -- as a timed synthetic run of the device evaluation path
(umr_amd/eval_utils.py on csrc/eval.hip): 10 000 image pairs x 2 directions x 15 keypoints, 1280-face / 642-vertex model,
256^2 images; flow mode (texture flows) and cam mode (cameras + target masks), PCK accumulated in device counters.
The CUB test pairs and the trained network are not distributed: flows / cameras / masks / keypoints are synthetic, so the
PCK values mean nothing -- the pairs/s do.  One JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from umr_amd import eval_utils as EU  # noqa: E402
from umr_amd.synthetic import template  # noqa: E402

dev = torch.device("cuda:0")
PAIRS, BATCH, K, T, S = int(os.environ.get("PAIRS", 10000)), 250, 15, 6, 256
tv, faces = template(3)
F = faces.shape[0]
mean_shape = (tv * 0.9).to(dev)
g = torch.Generator().manual_seed(0)


def batch_inputs():
    kps = torch.rand(BATCH, 2, K, 3, generator=g) * 1.8 - 0.9
    kps[..., 2] = (kps[..., 2] > -0.7).float()
    flows = (torch.rand(BATCH, 2, F, 1, 1, 2, generator=g) * 1.6 - 0.8 + 0.08 * (torch.rand(BATCH, 2, F, T, T, 2, generator=g) - 0.5)).clamp(-1, 1)
    cams = torch.cat([0.6 + 0.3 * torch.rand(BATCH, 2, 1, generator=g), 0.2 * torch.rand(BATCH, 2, 2, generator=g) - 0.1,
                      torch.nn.functional.normalize(torch.randn(BATCH, 2, 4, generator=g), dim=2)], 2)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, S), torch.linspace(-1, 1, S), indexing="ij")
    r = 0.25 + 0.2 * torch.rand(BATCH, 2, 1, 1, generator=g)
    masks = ((xx[None, None] ** 2 + yy[None, None] ** 2) < r).float()
    return [t.to(dev) for t in (kps, flows, cams, masks)]


def run(mode, data, cnt):
    kps, flows, cams, masks = data
    vis = kps[:, 0, :, 2] * kps[:, 1, :, 2]
    src = torch.cat([kps[:, 0], kps[:, 1]]); gt = torch.cat([kps[:, 1], kps[:, 0]]); v2 = torch.cat([vis, vis])
    if mode == "flow":
        EU.map_kp_flow_batch(src, torch.cat([flows[:, 0], flows[:, 1]]), torch.cat([flows[:, 1], flows[:, 0]]), S, 3, kp_gt=gt, vis=v2,
                             counters=cnt)
    else:
        EU.map_kp_cam_batch(src, torch.cat([cams[:, 0], cams[:, 1]]), torch.cat([cams[:, 1], cams[:, 0]]),
                            torch.cat([masks[:, 1], masks[:, 0]]), mean_shape, S, kp_gt=gt, vis=v2, counters=cnt)


data = batch_inputs()            # one resident batch, re-used: the timed region is the device path (inputs in HBM)
out = {"pairs": PAIRS, "batch_pairs": BATCH, "keypoints": K, "faces": F, "vertices": int(tv.shape[0]), "image_size": S}
for mode in ("flow", "cam"):
    cnt = EU.PCKCounters(K, dev)
    run(mode, data, cnt)
    torch.cuda.synchronize()
    cnt = EU.PCKCounters(K, dev)
    t0 = time.perf_counter()
    for _ in range(PAIRS // BATCH):
        run(mode, data, cnt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p1, p15 = cnt.pck()
    out[mode] = {"seconds": round(dt, 4), "pairs_per_s": round(PAIRS / dt, 1), "us_per_pair": round(1e6 * dt / PAIRS, 2),
                 "visible_keypoints_counted": int(cnt.counts[0].sum()), "pck1_synthetic": round(p1, 4), "pck15_synthetic": round(p15, 4)}
print(json.dumps(out), flush=True)
