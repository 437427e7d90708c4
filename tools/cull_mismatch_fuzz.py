"""CPU (wave64 emulation of the library): the mechanism behind round 3's non-finite training runs at configs[3], isolated.
Scenes of thin, isolated faces (every pixel sees at most one face -> the saved soft-max maximum is the background's eps wherever
the forward did not include the face); forward + texel-gradient backward through the library built (a) with the tile-cull band
at the exact threshold (-DTILE_CULL_NOISE=0.f: round 3's early builds) and (b) as shipped (band widened by the reference's
distance noise).  Counts faces with a non-finite texel gradient.
usage: cull_mismatch_fuzz.py [scenes=40] [IS=1024] [seed=0]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_raster as HR  # noqa: E402

CFG = dict(near=1., far=100., eps=1e-3, sigma_val=1e-5, dist_eps_log=float(math.log(1e10 - 1.)), gamma_val=1e-4, double_side=True, func_id_rgb=1)


def thin_faces(rng, F):
    """F needles on a jittered grid (no two within reach of each other): length 0.02-0.12, height 2e-5 .. 3e-3 screen units."""
    g = int(math.ceil(math.sqrt(F)))
    cells = [(i, j) for i in range(g) for j in range(g)][:F]
    fv = np.zeros((1, F, 3, 3), np.float32)
    for k, (i, j) in enumerate(cells):
        c = np.array([-0.9 + 1.8 * (i + 0.5) / g, -0.9 + 1.8 * (j + 0.5) / g]) + rng.uniform(-0.02, 0.02, 2)
        L, h, a = rng.uniform(0.02, 0.12), 10 ** rng.uniform(-4.7, -2.5), rng.uniform(0, math.pi)
        u, v = np.array([math.cos(a), math.sin(a)]), np.array([-math.sin(a), math.cos(a)])
        t = rng.uniform(0.1, 0.9)
        p = [c - 0.5 * L * u, c + 0.5 * L * u, c + (t - 0.5) * L * u + h * v]
        if rng.uniform() < 0.5:
            p[1], p[2] = p[2], p[1]
        for q in range(3):
            fv[0, k, q, :2] = p[q]
            fv[0, k, q, 2] = rng.uniform(3.0, 6.0)
    return fv.reshape(1, F, 9)


def main():
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    IS = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    libs = {"band = exact threshold (-DTILE_CULL_NOISE=0.f)": HR.lib(HR.build(extra_flags=["-DTILE_CULL_NOISE=0.f"], out=os.path.join(os.environ.get("TMPDIR", "/tmp"), "libumr_host_nocullnoise.so"))),
            "as shipped": HR.lib()}
    bad = {k: 0 for k in libs}
    faces_total = 0
    for s in range(scenes):
        F = 64
        fv = thin_faces(rng, F)
        tex = rng.uniform(0, 1, (1, F, 36, 3)).astype(np.float32)
        g = rng.standard_normal((1, 4, IS // 2, IS // 2)).astype(np.float32)
        faces_total += F
        for name, L in libs.items():
            o = HR.forward(fv, tex, IS, L=L, pooled=True, background_by_value=True, flags=HR.NO_P2F, **CFG)
            _, gt = HR.backward(fv, tex, o["soft_colors"], o["aggrs_info"], g, IS, need_gf=False, need_gt=True, grad_flags=HR.BWD_GRAD_POOLED, L=L, **CFG)
            nb = int((~np.isfinite(gt).all(axis=(2, 3))).sum())
            if nb:
                ff = np.argwhere(~np.isfinite(gt).all(axis=(2, 3))[0]).ravel()
                print("scene %d, %s: %d face(s) with a non-finite texel gradient: %s; forward finite: %s" % (
                    s, name, nb, ff[:6], bool(np.isfinite(o["soft_colors"]).all())), flush=True)
            bad[name] += nb
    print({"scenes": scenes, "faces": faces_total, "image_size": IS, "faces_with_nonfinite_texel_gradient": bad})


if __name__ == "__main__":
    main()
