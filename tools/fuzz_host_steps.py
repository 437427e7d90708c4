"""Fuzz whole render-and-compare steps on the CPU: umr_amd's RenderCompareS1 / RenderCompareS2 (the product's Python layer and every
kernel behind it, on the wave64 emulation of the library -- tests/host_raster.py::emulated_product) against the oracle's
restatement of experiments/train_s1.py:177-265 / train_s2.py:201-316, on randomised synthetic batches: sizes, camera scale
(meshes partly or wholly off screen, sub-pixel meshes), deformation amplitude, empty / full masks.  Reports non-finite terms or
gradients on either side and the largest deviations.

    python tools/fuzz_host_steps.py --steps 200 --seed 0 [--out file.json]
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
import host_raster as HR   # noqa: E402


def perturb(rng, out, batch, kind):
    with torch.no_grad():
        if kind == "big":          # camera scale x2..4: most of the mesh off screen
            for k in ("cam", "cam_hypotheses"):
                if k in out:
                    out[k][..., 0] *= float(rng.uniform(2, 4))
        elif kind == "small":      # sub-pixel .. few-pixel meshes
            for k in ("cam", "cam_hypotheses"):
                if k in out:
                    out[k][..., 0] *= float(10 ** rng.uniform(-2.5, -0.7))
        elif kind == "wild":       # large deformation: folded, self-intersecting, needle faces
            out["delta_v"] += float(rng.uniform(0.2, 1.0)) * torch.from_numpy(rng.standard_normal(tuple(out["delta_v"].shape)).astype(np.float32))
        elif kind == "empty_mask":
            batch["masks"].zero_()
        elif kind == "full_mask":
            batch["masks"].fill_(1.0)
        elif kind == "shift":
            for k in ("cam", "cam_hypotheses"):
                if k in out:
                    out[k][..., 1:3] += torch.from_numpy(rng.uniform(-1.5, 1.5, tuple(out[k][..., 1:3].shape)).astype(np.float32))


def clone_inputs(out, batch):
    o = {k: (v.detach().clone().requires_grad_(v.requires_grad) if torch.is_tensor(v) else v) for k, v in out.items()}
    b = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    return o, b


def run(rng, stats):
    from oracle.train_step_ref import RenderCompareS1Ref, RenderCompareS2Ref
    from umr_amd.synthetic import make_s1_inputs, make_s2_inputs
    from umr_amd.train_step import RenderCompareS1, RenderCompareS2
    stage = int(rng.integers(1, 3))
    H = int(rng.choice([32, 48, 64, 96]))
    sub = int(rng.integers(1, 3))
    B = 2
    kind = str(rng.choice(["plain", "big", "small", "wild", "empty_mask", "full_mask", "shift"]))
    seed = int(rng.integers(0, 1 << 30))
    if stage == 1:
        tv, faces, out, batch = make_s1_inputs(B, H, sub, seed=seed, device="cpu")
        ex = None
    else:
        K = int(rng.integers(2, 4))
        tv, faces, out, batch, ex = make_s2_inputs(B, K, H, sub, seed=seed, device="cpu")
    perturb(rng, out, batch, kind)
    out_r, batch_r = clone_inputs(out, batch)
    out_p, batch_p = clone_inputs(out, batch)
    if stage == 1:
        ref = RenderCompareS1Ref(tv, faces, H, n_threads=1)
        prod = RenderCompareS1(tv, faces, H)
    else:
        ref = RenderCompareS2Ref(tv, faces, ex["part_vertex_ids"], ex["uv_img"], ex["uv_sampler"], H, K, n_threads=1)
        prod = RenderCompareS2(tv, faces, ex["part_vertex_ids"], ex["uv_img"], ex["uv_sampler"], H, K, texture_loss_type="l1")
    rt, rterms = ref(out_r, batch_r)
    rt.backward()
    pt, pterms = prod(out_p, batch_p)
    pt.backward()
    key = "s%d/%s" % (stage, kind)
    rec = stats.setdefault(key, dict(steps=0, nonfinite_ref=0, nonfinite_product_only=0, term_dev_max=0.0, grad_dev_max=0.0, flagged=0))
    rec["steps"] += 1
    msgs = []
    fr = all(np.isfinite(float(v)) for v in rterms.values()) and all(
        torch.isfinite(v.grad).all() for v in out_r.values() if torch.is_tensor(v) and v.grad is not None)
    fp = all(np.isfinite(float(v)) for v in pterms.values()) and all(
        torch.isfinite(v.grad).all() for v in out_p.values() if torch.is_tensor(v) and v.grad is not None)
    if not fr:
        rec["nonfinite_ref"] += 1
    if fr and not fp:
        rec["nonfinite_product_only"] += 1
        msgs.append("NON-FINITE in the product only")
    if fr and fp:
        for k in rterms:
            d = abs(float(pterms[k]) - float(rterms[k])) / max(1.0, abs(float(rterms[k])))
            rec["term_dev_max"] = max(rec["term_dev_max"], d)
            if d > 2e-3:
                msgs.append("term %s %.6g vs %.6g" % (k, float(pterms[k]), float(rterms[k])))
        for k, v in out_r.items():
            if torch.is_tensor(v) and v.grad is not None and out_p[k].grad is not None:
                s = float(v.grad.abs().max())
                if s > 0:
                    d = float((out_p[k].grad - v.grad).abs().max()) / s
                    rec["grad_dev_max"] = max(rec["grad_dev_max"], d)
                    if d > 5e-2:
                        msgs.append("grad %s dev %.3g of max" % (k, d))
    if msgs:
        rec["flagged"] += 1
    return key, H, sub, seed, msgs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(1)
    HR.lib(HR.build())
    stats = {}
    t0 = time.time()
    with HR.emulated_product():
        for i in range(a.steps):
            rng = np.random.default_rng([a.seed, i])
            try:
                key, H, sub, seed, msgs = run(rng, stats)
            except Exception as e:       # an argument the product or the restatement refuses, a crash of either
                print("step %d: EXCEPTION %s: %s" % (i, type(e).__name__, str(e)[:300]), flush=True)
                continue
            if msgs:
                print("step %d (%s, H=%d, subdivide %d, input seed %d, rng [%d, %d]): %s" % (i, key, H, sub, seed, a.seed, i, "; ".join(msgs)), flush=True)
    res = dict(seed=a.seed, steps=a.steps, seconds=round(time.time() - t0, 1), classes=stats)
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
