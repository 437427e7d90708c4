"""A numpy (fp32, same operation order) model of the product's pixel/face geometry -- k_face_setup's record and eval_pair of
umr_amd/csrc/raster_core.h -- fuzzed on the CPU against the oracle (the reference's algorithm): one face per mesh, so a
mesh's alpha plane is that face's soft fragment D per pixel (0 where the pair is rejected).  Finds classes of faces on which
the product's pick-the-edge-first formulation and the reference's evaluate-all-three formulation disagree, or where the
former is not finite -- without a GPU.  (Test infrastructure; the product never imports it.  It was the design sketch for the
ill-conditioned-face rule now in eval_pair -- `exact3_below=1e-5` is the shipped behaviour; since then the kernel source
itself is fuzzed on the host, tests/test_kernel_source_on_host.py.)

usage: python tests/eval_pair_model.py [meshes_per_case]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
f32 = np.float32
BIG = f32(3.0e38)


def face_setup(fv):
    """fv [n,3,3] float32 -> dict of per-face arrays (k_face_setup)"""
    x = fv[:, :, 0].astype(f32); y = fv[:, :, 1].astype(f32)
    x0, x1, x2 = x[:, 0], x[:, 1], x[:, 2]; y0, y1, y2 = y[:, 0], y[:, 1], y[:, 2]
    adj = np.stack([y1 - y2, x2 - x1, x1 * y2 - x2 * y1, y2 - y0, x0 - x2, x2 * y0 - x0 * y2, y0 - y1, x1 - x0, x0 * y1 - x1 * y0], 1).astype(f32)
    det_raw = (x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0)).astype(f32)
    det = np.where(det_raw > 0, np.maximum(det_raw, f32(1e-10)), np.minimum(det_raw, f32(-1e-10))).astype(f32)
    inv = (adj / det[:, None]).astype(f32)
    sym = ((x[:, :, None] * x[:, None, :] + y[:, :, None] * y[:, None, :]).astype(f32) + f32(1)).astype(f32)   # [n,3,3]
    obt = np.full(len(fv), -1)
    for k in (2, 1, 0):
        k1, k2 = (k + 1) % 3, (k + 2) % 3
        c = ((x[:, k1] - x[:, k]) * (x[:, k2] - x[:, k]) + (y[:, k1] - y[:, k]) * (y[:, k2] - y[:, k])).astype(f32) < 0
        obt = np.where(c, k, obt)
    K = np.zeros((len(fv), 3), f32)
    for c in range(3):
        a, b = (c + 1) % 3, (c + 2) % 3
        ex, ey = x[:, a] - x[:, b], y[:, a] - y[:, b]
        K[:, c] = (det_raw * det_raw / np.maximum((ex * ex + ey * ey).astype(f32), f32(1e-30))).astype(f32)
    edges = np.zeros((len(fv), 3, 6), f32)     # a0 a1 a2 a[v1] den rden
    with np.errstate(divide="ignore", invalid="ignore"):
        for e in range(3):
            e1 = (e + 1) % 3
            a = (sym[:, e, :] - sym[:, e1, :]).astype(f32)
            den = (a[:, e] - a[:, e1]).astype(f32)
            edges[:, e, 0:3] = a; edges[:, e, 3] = a[:, e1]; edges[:, e, 4] = den; edges[:, e, 5] = (f32(1) / den).astype(f32)
    ob = np.where(obt < 0, 0, obt)
    ox = x[np.arange(len(fv)), (ob + 2) % 3] - x[np.arange(len(fv)), ob]
    oy = y[np.arange(len(fv)), (ob + 2) % 3] - y[np.arange(len(fv)), ob]
    return dict(x=x, y=y, inv=inv, K=K, edges=edges, obt=obt, ox=ox.astype(f32), oy=oy.astype(f32))


def eval_pair(rec, xp, yp, thr, threshold, nis, fallback=True, exact3_below=0.0):
    """rec: face_setup of n faces; xp, yp [P] pixel centres -> (live [n,P] bool, frag [n,P] f32, nonfinite_tv [n,P] bool)"""
    n = len(rec["x"]); X = xp[None, :].astype(f32); Y = yp[None, :].astype(f32)
    x, y, inv, K = rec["x"], rec["y"], rec["inv"], rec["K"]
    xlo = (np.minimum(np.minimum(x[:, 0], x[:, 1]), x[:, 2]) - thr)[:, None]; xhi = (np.maximum(np.maximum(x[:, 0], x[:, 1]), x[:, 2]) + thr)[:, None]
    ylo = (np.minimum(np.minimum(y[:, 0], y[:, 1]), y[:, 2]) - thr)[:, None]; yhi = (np.maximum(np.maximum(y[:, 0], y[:, 1]), y[:, 2]) + thr)[:, None]
    inb = ~((X > xhi) | (X < xlo) | (Y > yhi) | (Y < ylo))
    w = [((inv[:, 3 * c, None] * X).astype(f32) + (inv[:, 3 * c + 1, None] * Y).astype(f32)).astype(f32) + inv[:, 3 * c + 2, None] for c in range(3)]
    w = [a.astype(f32) for a in w]
    inside = (w[0] > 0) & (w[1] > 0) & (w[2] > 0) & (w[0] < 1) & (w[1] < 1) & (w[2] < 1)
    m = [(w[2] * w[2] * K[:, 2, None]).astype(f32), (w[0] * w[0] * K[:, 0, None]).astype(f32), (w[1] * w[1] * K[:, 1, None]).astype(f32)]
    c1 = m[1] < m[0]
    best = np.where(c1, m[1], m[0])
    kin = np.where(m[2] < best, 2, np.where(c1, 1, 0))
    ob = rec["obt"][:, None]
    cx = np.where(ob == 0, x[:, 0, None], np.where(ob == 1, x[:, 1, None], np.where(ob == 2, x[:, 2, None], f32(0))))
    cy = np.where(ob == 0, y[:, 0, None], np.where(ob == 1, y[:, 1, None], np.where(ob == 2, y[:, 2, None], f32(0))))
    ovr = (((X - cx).astype(f32) * rec["ox"][:, None]).astype(f32) + ((Y - cy).astype(f32) * rec["oy"][:, None]).astype(f32)).astype(f32) > 0
    code = np.minimum((w[0] <= 0) * 1 | (w[1] <= 0) * 2 | (w[2] <= 0) * 4, 6)
    lut = np.array([0, 2, 3, 3, 1, 2, 1]) - 1            # KOUT_LUT entries - 1, index = region code
    kout = lut[code]
    m_ob = np.where(ob == 0, 6, np.where(ob == 1, 5, np.where(ob == 2, 3, -1)))
    k_ob = np.where(ob == 0, 2, np.where(ob == 1, 0, 1))
    kout = np.where((code == m_ob) & ovr, k_ob, kout)
    ksel = np.where(inside, kin, kout)
    kvalid = ksel >= 0
    k = np.maximum(ksel, 0)
    E = rec["edges"]
    idx = np.arange(n)[:, None]

    def edge_param(kk):
        ea = E[idx, kk]                                  # [n,P,6]
        num = (((w[0] * ea[..., 0]).astype(f32) + (w[1] * ea[..., 1]).astype(f32)).astype(f32) + (w[2] * ea[..., 2]).astype(f32)).astype(f32) - ea[..., 3]
        with np.errstate(all="ignore"):
            return (num.astype(f32) / ea[..., 4]).astype(f32)   # IEEE quotient (div_r is correctly rounded for ordinary operands)
    with np.errstate(all="ignore"):
        tv = edge_param(k)
        bad = inside & ~(np.abs(tv) <= BIG)
        nonfinite = bad.copy()
        no_edge = np.zeros_like(bad)
        if fallback and bad.any():
            M = np.stack(m, -1)
            ka = np.where(k == 0, 1, 0); kb = np.where(k == 2, 1, 2)
            ma = np.take_along_axis(M, ka[..., None], -1)[..., 0]; mb = np.take_along_axis(M, kb[..., None], -1)[..., 0]
            sw = mb < ma
            ka, kb = np.where(sw, kb, ka), np.where(sw, ka, kb)
            ta = edge_param(ka); oka = np.abs(ta) <= BIG
            tb_ = edge_param(kb); okb = np.abs(tb_) <= BIG
            k = np.where(bad & oka, ka, np.where(bad & ~oka & okb, kb, k))
            tv = np.where(bad & oka, ta, np.where(bad & ~oka & okb, tb_, np.where(bad, f32(0), tv))).astype(f32)
            no_edge = bad & ~oka & ~okb
        if exact3_below > 0:
            # faces with an ill-conditioned edge (|den| below the bound): the reference's own inside evaluation -- all three
            # edge lines, smallest computed distance with `<` in the order k = 0, 1, 2 (:78-107)
            flagged = (np.abs(E[:, :, 4]).min(1) < exact3_below)[:, None] & inside
            if flagged.any():
                dmin = np.full(tv.shape, f32(1e8)); kbest = np.full(tv.shape, -1); tbest = np.zeros(tv.shape, f32)
                for kk in range(3):
                    tk = edge_param(np.full(tv.shape, kk))
                    bk = [np.where(kk == 0, tk, np.where(kk == 1, f32(0), f32(1) - tk)), np.where(kk == 0, f32(1) - tk, np.where(kk == 1, tk, f32(0))),
                          np.where(kk == 0, f32(0), np.where(kk == 1, f32(1) - tk, tk))]
                    tt = [(bk[i].astype(f32) - w[i]).astype(f32) for i in range(3)]
                    ddx = (((tt[0] * x[:, 0, None]).astype(f32) + (tt[1] * x[:, 1, None]).astype(f32)).astype(f32) + (tt[2] * x[:, 2, None]).astype(f32)).astype(f32)
                    ddy = (((tt[0] * y[:, 0, None]).astype(f32) + (tt[1] * y[:, 1, None]).astype(f32)).astype(f32) + (tt[2] * y[:, 2, None]).astype(f32)).astype(f32)
                    dk = ((ddx * ddx).astype(f32) + (ddy * ddy).astype(f32)).astype(f32)
                    better = dk < dmin
                    dmin = np.where(better, dk, dmin); kbest = np.where(better, kk, kbest); tbest = np.where(better, tk, tbest)
                k = np.where(flagged & (kbest >= 0), kbest, k)
                tv = np.where(flagged & (kbest >= 0), tbest, tv).astype(f32)
                no_edge = np.where(flagged, kbest < 0, no_edge)
                tv = np.where(no_edge, f32(0), tv).astype(f32)
        tb = (f32(1) - tv).astype(f32)
        ba = np.where(inside, tv, np.minimum(np.fmax(tv, f32(0)), f32(1)))
        bb = np.where(inside, tb, np.minimum(np.fmax(tb, f32(0)), f32(1)))
        b = [np.where(k == 0, ba, np.where(k == 1, f32(0), bb)), np.where(k == 0, bb, np.where(k == 1, ba, f32(0))),
             np.where(k == 0, f32(0), np.where(k == 1, bb, ba))]
        b = [np.where(no_edge, w[i], b[i]).astype(f32) for i in range(3)]
        t = [(b[i] - w[i]).astype(f32) for i in range(3)]
        dx = (((t[0] * x[:, 0, None]).astype(f32) + (t[1] * x[:, 1, None]).astype(f32)).astype(f32) + (t[2] * x[:, 2, None]).astype(f32)).astype(f32)
        dy = (((t[0] * y[:, 0, None]).astype(f32) + (t[1] * y[:, 1, None]).astype(f32)).astype(f32) + (t[2] * y[:, 2, None]).astype(f32)).astype(f32)
        dis = ((dx * dx).astype(f32) + (dy * dy).astype(f32)).astype(f32)
        e = np.exp((np.where(inside, dis, -dis) * nis).astype(np.float64)).astype(f32)
        frag = (f32(1) / (f32(1) + e)).astype(f32)
        live = inb & kvalid & (inside | ~(dis >= threshold))
    return live, frag, nonfinite


def fuzz_cases(n, IS, rng):
    px = ((2 * np.arange(IS) + 1 - IS) / IS).astype(f32)
    r = lambda *s: rng.uniform(-1, 1, s).astype(f32)
    a, b = r(n, 2), r(n, 2); t = rng.uniform(0, 1, (n, 1)).astype(f32)
    pc = px[rng.integers(0, IS, (n, 2))]
    cases = {
        "random": np.stack([a, b, r(n, 2)], 1),
        "small": np.stack([a, a + 0.05 * r(n, 2), a + 0.05 * r(n, 2)], 1),
        "collapsed_edge_1e-4": np.stack([a, a + 1e-4 * r(n, 2), b], 1),
        "collapsed_edge_1e-5": np.stack([a, a + 1e-5 * r(n, 2), b], 1),
        "collapsed_edge_last": np.stack([b, a, a + 3e-5 * r(n, 2)], 1),
        "needle_on_pixel": np.stack([pc + np.array([-2e-5, -1e-5], f32), pc + np.array([2e-5, -1e-5], f32), pc + np.stack([0.01 * r(n), 0.03 + 0.05 * np.abs(r(n))], 1)], 1),
        "collinear": np.stack([a, b, a + t * (b - a)], 1),
        "sliver": np.stack([a, b, a + t * (b - a) + 1e-6 * r(n, 2)], 1),
        "two_equal": np.stack([a, a, b], 1),
        "tiny": np.stack([a, a + 1e-6 * r(n, 2), a + 1e-6 * r(n, 2)], 1),
        "vertex_on_pixel": np.stack([pc, b, r(n, 2)], 1),
        "obtuse": np.stack([a, a + np.array([0.3, 0.0], f32) + 0.01 * r(n, 2), a + np.array([0.15, 0.01], f32) * (1 + 0.5 * r(n, 1))], 1),
    }
    return {k: np.concatenate([v.astype(f32), np.full((n, 3, 1), 7.7, f32)], 2) for k, v in cases.items()}, px


def main():
    from oracle import softras
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    IS = 48
    rng = np.random.default_rng(0)
    cases, px = fuzz_cases(n, IS, rng)
    sigma = f32(1e-5); dist_eps_log = f32(np.log(1. / 1e-10 - 1.))
    threshold = f32(dist_eps_log * sigma); thr = f32(np.sqrt(threshold)); nis = f32(-1.0 / sigma)
    xi, yi = np.meshgrid(np.arange(IS), np.arange(IS))          # image row r <-> yi = IS-1-r
    xp = px[xi.ravel()]; yp = px[(IS - 1 - yi).ravel()]
    cfg = dict(near=1., far=100., eps=1e-3, sigma_val=float(sigma), dist_eps_log=float(dist_eps_log), gamma_val=1e-4, func_id_rgb=1, double_side=True)
    print("%-22s %10s %10s %12s %12s %14s" % ("case", "pairs", "live(ref)", ">1e-4 (old)", ">1e-4 (new)", "non-finite old/new"))
    for name, fv in cases.items():
        tex = np.ones((n, 1, 1, 3), f32)
        ref = softras.raster_forward(fv.reshape(n, 1, 9), tex, IS, background=(0, 0, 0), backend="port", n_threads=8, **cfg)
        ref_a = ref["soft_colors"][:, 3].reshape(n, -1)
        rec = face_setup(fv)
        out = {}
        for fb in (False, True):
            live, frag, nonfin = eval_pair(rec, xp, yp, thr, threshold, nis, fallback=fb)
            a = np.where(live, frag, f32(0))
            bad = ~np.isfinite(a)
            err = np.abs(np.where(bad, 1.0, a.astype(np.float64)) - ref_a)
            out[fb] = (int((err > 1e-4).sum()), int(bad.sum()))
        print("%-22s %10d %10d %12d %12d %9d / %d" % (name, ref_a.size, int((ref_a > 0).sum()), out[False][0], out[True][0], out[False][1], out[True][1]))


if __name__ == "__main__":
    main()
