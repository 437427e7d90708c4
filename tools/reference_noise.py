"""CPU measurements behind HISTORY.md 4.4 (thin faces) and 4.1 (tile cull): the rounding noise of the reference's closest-point
formulation, measured on the product's own eval_pair (tests/host_kernel/pair_host.cpp = raster_core.h compiled for the host; bit
for bit the reference's arithmetic, tests/test_kernel_source_on_host.py).

  edge    inside pixels: how often 'nearest edge line by its true distance w_c^2 K_c, then the reference's formula for that edge'
          (eval_pair's fast route) ends at another closest point than the reference's 'smallest COMPUTED distance of the three'
          (the route of flagged faces), by the face's smallest height
  cull    outside pixels the reference still includes (computed distance below the threshold) although their exact distance is
          beyond it: the largest exact excess, and the slack the tile cull needs in its own units, by the face's smallest height

    python tools/reference_noise.py edge|cull [faces]
"""
import ctypes
import math
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HK = os.path.join(ROOT, "tests", "host_kernel")
f32 = np.float32
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


def load():
    so = os.path.join(HK, "libpair_host.so")
    srcs = [os.path.join(HK, "pair_host.cpp"), os.path.join(HK, "device_shim.h")] + [os.path.join(ROOT, "umr_amd", "csrc", f) for f in
                                                                                     ("raster_core.h", "umr_common.h", "raster_general.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-Wno-unused-function", "-Wno-unknown-attributes", srcs[0], "-o", so])
    h = ctypes.CDLL(so)
    h.host_pairs.argtypes = [P, I, P, P, I, F, F, F, F, F, P, P, P, P]
    h.host_pairs.restype = I
    h.host_set_thin_h.argtypes = [F]
    return h


sigma, del_ = 1e-5, math.log(1e10 - 1)
threshold = f32(f32(del_) * f32(sigma))
thr, nis = f32(np.sqrt(threshold)), f32(-1.0 / f32(sigma))
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731


def pairs(h, face, xp, yp):
    M = len(xp)
    live, frag, dxy, zp = np.zeros((1, M), np.uint8), np.zeros((1, M), f32), np.zeros((1, M, 2), f32), np.zeros((1, M), f32)
    assert h.host_pairs(p(face), 1, p(xp), p(yp), M, float(thr), float(threshold), float(nis), 1.0, 100.0, p(live), p(frag), p(dxy), p(zp)) == 0
    return live[0] != 0, frag[0], dxy[0]


def random_face(rng):
    L = 10 ** rng.uniform(-2, -0.5)
    hh = min(10 ** rng.uniform(-5.0, -0.7), L)
    ang = rng.uniform(0, 2 * np.pi)
    c = rng.uniform(-0.9, 0.9, 2)
    u = np.array([np.cos(ang), np.sin(ang)])
    v = np.array([-u[1], u[0]])
    tri = np.array([c, c + L * u, c + rng.uniform(0.0, 1.0) * L * u + hh * v * rng.choice([-1, 1])])[rng.permutation(3)]
    face = np.concatenate([tri, rng.uniform(3, 8, (3, 1))], 1).astype(f32).reshape(1, 9)
    t = face.reshape(3, 3)[:, :2].astype(np.float64)
    x, y = t[:, 0], t[:, 1]
    det = x[2] * (y[0] - y[1]) + x[0] * (y[1] - y[2]) + x[1] * (y[2] - y[0])
    hs = [abs(det) / max(np.hypot(x[(k + 1) % 3] - x[(k + 2) % 3], y[(k + 1) % 3] - y[(k + 2) % 3]), 1e-300) for k in range(3)]
    return face, t, det, hs


def seg_d2(px, py, a, b):
    ex, ey = b[0] - a[0], b[1] - a[1]
    t = np.clip(((px - a[0]) * ex + (py - a[1]) * ey) / (ex * ex + ey * ey + 1e-300), 0, 1)
    return (a[0] + t * ex - px) ** 2 + (a[1] + t * ey - py) ** 2


def edge(h, nf):
    rng = np.random.default_rng(1)
    hb = np.array([1e-4, 3e-4, 1e-3, 2e-3, 4e-3, 8e-3, 1.6e-2, 3.2e-2, 6.4e-2, 0.2])
    res = {}
    for _ in range(nf):
        face, t, det, hs = random_face(rng)
        M = 3000
        pts = rng.dirichlet([1, 1, 1], M) @ t
        xp, yp = np.ascontiguousarray(pts[:, 0], f32), np.ascontiguousarray(pts[:, 1], f32)
        h.host_set_thin_h(0.0)
        _, fa, da = pairs(h, face, xp, yp)
        h.host_set_thin_h(1e9)
        _, fb, db = pairs(h, face, xp, yp)
        k = np.searchsorted(hb, min(hs)) - 1
        if 0 <= k < len(hb) - 1:
            r = res.setdefault(k, [0, 0, 0, 0.0, 0])
            r[0] += M
            r[1] += int((np.abs(da - db).max(-1) > 1e-6).sum())
            r[2] += int((np.abs(fa - fb) > 1e-4).sum())
            r[3] = max(r[3], float(np.abs(fa - fb).max()))
            r[4] += 1
    for k in sorted(res):
        r = res[k]
        print("smallest height [%.1e, %.1e): %4d faces, %8d inside pixels; another closest point: %.2e of them; |dD| > 1e-4: %.2e; max |dD| %.2e"
              % (hb[k], hb[k + 1], r[4], r[0], r[1] / r[0], r[2] / r[0], r[3]))


def cull(h, nf):
    rng = np.random.default_rng(0)
    bins = np.logspace(-5.5, -0.5, 11)
    worst, worst_w, cnt = np.zeros(len(bins) - 1), np.zeros(len(bins) - 1), np.zeros(len(bins) - 1, int)
    h.host_set_thin_h(1.6e-2)
    for _ in range(nf):
        face, t, det, hs = random_face(rng)
        M = 4000
        e = rng.integers(0, 3, M)
        s = rng.uniform(-0.2, 1.2, M)
        A, B, C = t[e], t[(e + 1) % 3], t[(e + 2) % 3]
        base = A + (B - A) * np.clip(s, 0, 1)[:, None]
        ed = B - A
        nrm = np.stack([-ed[:, 1], ed[:, 0]], 1)
        nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
        sgn = np.sign(((base - C) * nrm).sum(1))
        sgn[sgn == 0] = 1
        d = nrm * sgn[:, None] + rng.uniform(-1, 1, M)[:, None] * (ed / np.maximum(np.linalg.norm(ed, axis=1, keepdims=True), 1e-300)) * ((s < 0) | (s > 1))[:, None]
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-300)
        pts = base + d * (float(thr) * (1 + rng.uniform(-0.01, 0.3, M)))[:, None]
        xp, yp = np.ascontiguousarray(pts[:, 0], f32), np.ascontiguousarray(pts[:, 1], f32)
        live, _, _ = pairs(h, face, xp, yp)
        X, Y = xp.astype(np.float64), yp.astype(np.float64)
        d2 = np.minimum(np.minimum(seg_d2(X, Y, t[0], t[1]), seg_d2(X, Y, t[1], t[2])), seg_d2(X, Y, t[2], t[0]))
        cr = lambda a, b: (b[0] - a[0]) * (Y - a[1]) - (b[1] - a[1]) * (X - a[0])   # noqa: E731
        c0, c1, c2 = cr(t[0], t[1]), cr(t[1], t[2]), cr(t[2], t[0])
        ins = ((c0 > 0) & (c1 > 0) & (c2 > 0)) | ((c0 < 0) & (c1 < 0) & (c2 < 0))
        lv = live & ~ins
        if not lv.any():
            continue
        x, y = t[:, 0], t[:, 1]
        inv = np.array([[y[1] - y[2], x[2] - x[1], x[1] * y[2] - x[2] * y[1]], [y[2] - y[0], x[0] - x[2], x[2] * y[0] - x[0] * y[2]],
                        [y[0] - y[1], x[1] - x[0], x[0] * y[1] - x[1] * y[0]]]) / det
        w = inv[:, 0:1] * X[lv][None] + inv[:, 1:2] * Y[lv][None] + inv[:, 2:3]
        need = np.max(-(float(thr) / np.array(hs))[:, None] - w, axis=0)
        k = np.searchsorted(bins, min(hs)) - 1
        if 0 <= k < len(worst):
            worst[k] = max(worst[k], (np.sqrt(d2[lv]) / float(thr) - 1.0).max())
            worst_w[k] = max(worst_w[k], need.max())
            cnt[k] += 1
    for k in range(len(worst)):
        print("smallest height [%.1e, %.1e): %4d faces; included pixels lie up to %.3e x threshold distance beyond it; slack the cull "
              "needs %.3e barycentric units = %.2e / h^2" % (bins[k], bins[k + 1], cnt[k], worst[k], worst_w[k], worst_w[k] * bins[k] * bins[k + 1]))


if __name__ == "__main__":
    h = load()
    {"edge": edge, "cull": cull}[sys.argv[1]](h, int(sys.argv[2]) if len(sys.argv) > 2 else 3000)
