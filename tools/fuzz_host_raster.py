"""Fuzz the product's raster library on the CPU (tests/host_raster.py: raster.hip compiled on the wave64 emulator) against the
oracle: random and degenerate scenes, forward + every backward variant.  Reports non-finite outputs, pixels whose set of
contributing faces differs from the reference's (visible as a different running soft-max maximum), and the error statistics.

    python tools/fuzz_host_raster.py --scenes 200 --seed 0 [--classes dense,needles,...] [--is 64] [--lib path.so]
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_raster as HR          # noqa: E402
from oracle import softras       # noqa: E402

CFG = dict(near=1., far=100., eps=1e-3, sigma_val=1e-5, dist_eps_log=float(math.log(1e10 - 1.)), gamma_val=1e-4, double_side=True,
           func_id_rgb=1)


def sphere_scene(rng, n, subdiv, scale):
    import torch
    from oracle import torch_ref as TR
    from umr_amd.mesh import create_sphere
    v, f = create_sphere(subdiv)
    verts = torch.from_numpy(v).float()[None].repeat(n, 1, 1)
    verts = verts + float(rng.uniform(0.0, 0.08)) * torch.from_numpy(rng.standard_normal(verts.shape).astype(np.float32))
    faces = torch.from_numpy(f).long()[None].repeat(n, 1, 1)
    s = torch.from_numpy(rng.uniform(scale[0], scale[1], (n, 1)).astype(np.float32))
    t = torch.from_numpy(rng.uniform(-0.3, 0.3, (n, 2)).astype(np.float32))
    q = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32))
    q = q / q.norm(dim=1, keepdim=True)
    cams = torch.cat([s, t, q], 1)
    proj = TR.orthographic_proj_withz(verts, cams, 5.) * torch.tensor([1., -1., 1.])
    return TR.face_vertices(TR.look_at_ortho(proj), faces).reshape(n, -1, 9).numpy().astype(np.float32)


def random_faces(rng, n, F, size, z=(3.0, 8.0)):
    c = rng.uniform(-1.0, 1.0, (n, F, 1, 2))
    p = c + size * rng.standard_normal((n, F, 3, 2))
    zz = rng.uniform(z[0], z[1], (n, F, 3, 1))
    return np.concatenate([p, zz], -1).reshape(n, F, 9).astype(np.float32)


def make_scene(rng, kind, IS):
    n = 2
    if kind == "sphere":
        return sphere_scene(rng, n, int(rng.integers(1, 3)), (0.4, 1.1))
    if kind == "dense":          # many faces of a few pixels: slivers and near-threshold pairs abound
        return sphere_scene(rng, n, 3, (0.3, 0.9))
    if kind == "soup":
        return random_faces(rng, n, 96, float(rng.uniform(0.02, 0.3)))
    if kind == "tiny":
        return random_faces(rng, n, 128, float(rng.uniform(0.2, 3.0)) / IS)
    if kind == "needles":        # one short edge, or nearly collinear vertices
        f = random_faces(rng, n, 96, 0.15).reshape(n, 96, 3, 3)
        w = rng.integers(0, 2, (n, 96, 1))
        eps = 10.0 ** rng.uniform(-7, -2, (n, 96, 1))
        a, b = f[:, :, 0, :2], f[:, :, 1, :2]
        f[:, :, 2, :2] = np.where(w == 0, a + eps * rng.standard_normal((n, 96, 2)),      # collapsed edge
                                  a + (b - a) * rng.uniform(-0.5, 1.5, (n, 96, 1)) + eps * rng.standard_normal((n, 96, 2)))  # collinear
        return f.reshape(n, 96, 9).astype(np.float32)
    if kind == "degenerate":     # repeated vertices, zero-area faces, faces on pixel centres, off-screen / huge faces
        f = random_faces(rng, n, 64, 0.2).reshape(n, 64, 3, 3)
        k = rng.integers(0, 5, (n, 64))
        f[k == 0, 1] = f[k == 0, 0]
        f[k == 1, 2, :2] = 0.5 * (f[k == 1, 0, :2] + f[k == 1, 1, :2])
        grid = (2 * rng.integers(0, IS, (n, 64, 3, 2)) + 1 - IS) / IS
        f[k == 2, :, :2] = grid[k == 2]
        f[k == 3, :, :2] *= 20.0
        f[k == 4, :, 2] = rng.choice([0.5, 1.0, 100.0, 150.0], (int((k == 4).sum()), 3))   # depth at / beyond near and far
        return f.reshape(n, 64, 9).astype(np.float32)
    raise ValueError(kind)


def run_one(rng, kind, IS, L, stats):
    faces = make_scene(rng, kind, IS)
    n, F = faces.shape[:2]
    TS = int(rng.choice([1, 4, 9]))
    tex = rng.uniform(0, 1, (n, F, TS, 3)).astype(np.float32)
    gsc = rng.standard_normal((n, 4, IS, IS)).astype(np.float32)
    cfg = dict(CFG, double_side=bool(rng.integers(0, 4) > 0))
    ref = softras.raster_forward(faces, tex, IS, background=(0.1, 0.2, 0.3), n_threads=1, **cfg)
    o = HR.forward(faces, tex, IS, background=(0.1, 0.2, 0.3), L=L, **cfg)
    rec = stats.setdefault(kind, dict(scenes=0, pixels=0, nonfinite_ref=0, nonfinite_host_only=0, membership_pixels=0,
                                      alpha_err_max=0.0, rgb_err_gt_1e4=0, gf_err_max=0.0, gt_err_max=0.0, gfa_err_max=0.0,
                                      gf_bad=0, gt_bad=0))
    rec["scenes"] += 1
    rec["pixels"] += n * IS * IS
    fin_r = np.isfinite(ref["soft_colors"]).all() and np.isfinite(ref["aggrs_info"]).all()
    fin_h = np.isfinite(o["soft_colors"]).all() and np.isfinite(o["aggrs_info"]).all()
    rec["nonfinite_ref"] += int(not fin_r)
    rec["nonfinite_host_only"] += int(fin_r and not fin_h)
    with np.errstate(invalid="ignore"):
        memb = np.abs(o["aggrs_info"][:, 1] - ref["aggrs_info"][:, 1]) > 2e-6
        rec["membership_pixels"] += int(memb.sum())
        rec["alpha_err_max"] = max(rec["alpha_err_max"], float(np.nanmax(np.abs(o["soft_colors"][:, 3] - ref["soft_colors"][:, 3]))))
        rec["rgb_err_gt_1e4"] += int((np.abs(o["soft_colors"][:, :3] - ref["soft_colors"][:, :3]) > 1e-4).sum())
    # backward on the REFERENCE's saved state (so that a membership difference of the forward does not propagate)
    rgf, rgt = softras.raster_backward(faces, tex, ref["soft_colors"], ref["faces_info"], ref["aggrs_info"], gsc, IS, n_threads=1, **cfg)
    gf, gt = HR.backward(faces, tex, ref["soft_colors"], ref["aggrs_info"], gsc, IS, L=L, **cfg)
    _, gt1 = HR.backward(faces, tex, ref["soft_colors"], ref["aggrs_info"], gsc, IS, need_gf=False, L=L, **cfg)
    ga = gsc.copy()
    ga[:, :3] = 0
    rgfa, _ = softras.raster_backward(faces, tex, ref["soft_colors"], ref["faces_info"], ref["aggrs_info"], ga, IS, n_threads=1, **cfg)
    gfa, _ = HR.backward(faces, None, np.ascontiguousarray(ref["soft_colors"][:, 3]), None, np.ascontiguousarray(ga[:, 3]), IS,
                         need_gt=False, grad_flags=HR.BWD_ALPHA_ONLY, L=L, **cfg)
    # the one-pass backward of a shared mask / texture render: its vertex gradient is the alpha term alone, its texel gradient
    # the full one
    gf1p, gt1p = HR.backward(faces, tex, ref["soft_colors"], ref["aggrs_info"], gsc, IS, grad_flags=HR.BWD_ALPHA_GEOMETRY, L=L, **cfg)
    # ... and the same from the PACKED saved state (UMR_BWD_PACKED_STATE: what the training steps run), built here from the
    # reference's planes by the numpy restatement of the layout
    state = HR.pack_state(ref["aggrs_info"][:, 0], ref["aggrs_info"][:, 1], ref["soft_colors"][:, 3])
    gf1k, gt1k = HR.backward(faces, tex, None, state, gsc, IS, grad_flags=HR.BWD_ALPHA_GEOMETRY | HR.BWD_PACKED_STATE, L=L, **cfg)
    bad = []
    for name, a, r in (("gf", gf, rgf), ("gt", gt, rgt), ("gt", gt1, rgt), ("gfa", gfa, rgfa), ("gfa", gf1p, rgfa), ("gt", gt1p, rgt),
                       ("gfa", gf1k, rgfa), ("gt", gt1k, rgt)):
        fr = np.isfinite(r).all()
        if fr and not np.isfinite(a).all():
            rec["nonfinite_host_only"] += 1
            bad.append(name + ":nonfinite")
            continue
        if not fr:
            rec["nonfinite_ref"] += 1
            continue
        sc = max(float(np.abs(r).max()), 1e-30)
        e = float(np.abs(a - r).max()) / sc
        rec[name + "_err_max"] = max(rec[name + "_err_max"], e)
        if e > 1e-3:
            rec["gf_bad" if name != "gt" else "gt_bad"] += 1
            bad.append("%s:%.2e" % (name, e))
    return bad, int(memb.sum())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=50)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--is", dest="IS", type=int, default=64)
    ap.add_argument("--classes", default="sphere,dense,soup,tiny,needles,degenerate")
    ap.add_argument("--lib", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--exact-edges", type=int, default=1, help="umr_debug_set(\"exact_edges\", v) before the run")
    ap.add_argument("--thin-h-1e6", type=int, default=-1, help="umr_debug_set(\"thin_face_h_1e6\", v); negative = the default")
    a = ap.parse_args()
    L = HR.lib(a.lib or HR.build())
    L.umr_debug_set(b"exact_edges", a.exact_edges)
    L.umr_debug_set(b"thin_face_h_1e6", a.thin_h_1e6)
    stats = {}
    t0 = time.time()
    kinds = a.classes.split(",")
    for i in range(a.scenes):
        kind = kinds[i % len(kinds)]
        rng = np.random.default_rng([a.seed, i])
        bad, memb = run_one(rng, kind, a.IS, L, stats)
        if bad:
            print("scene %d (%s, seed [%d, %d]): %s" % (i, kind, a.seed, i, " ".join(bad)), flush=True)
    out = dict(seed=a.seed, scenes=a.scenes, image_size=a.IS, seconds=round(time.time() - t0, 1), emulator=HR.stats(L), classes=stats)
    print(json.dumps(out, indent=1))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(out, fh, indent=1)
