"""How many of the textured forward's wave visits could stop after the alpha product: the product's forward source (scratch copy)
with counters, on the emulator, fixed SURVEY 8d scene.  A visit is 'colour-dead' when every live lane's soft-max weight of the face
is exactly 0 (the face's nearest depth >= 89 gamma behind the lane's running maximum), 'z-dead' when the face cannot win the
z-buffer plane of any live lane either.  usage: python tools/forward_census.py [n_meshes=2]"""
import ctypes
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NAMES = ["visits(wave)", "visits with a live lane", "... all live lanes colour-dead", "... and z-dead too", "live lanes", "colour-dead live lanes"]
INJECT = [
    ("                if (live) {\n                    alpha *= 1.f - p.frag;",
     "                if (RGB == 1) { const float zm_ = fminf(fminf(fc.g<R_Z0>(), fc.g<R_Z1>()), fc.g<R_Z2>());\n"
     "                    const bool cd_ = !live || ((A.far_ - zm_) * A.r_range - smax) * A.inv_gamma < -89.f;\n"
     "                    const bool vd_ = !live || zm_ >= depth_min;\n"
     "                    const bool any_ = wave_any(live), acd_ = wave_all(cd_), avd_ = wave_all(cd_ && vd_);\n"
     "                    if (t.lane == 0) { CNT(0); if (any_) { CNT(1); if (acd_) CNT(2); if (avd_) CNT(3); } }\n"
     "                    if (live) { CNT(4); if (cd_) CNT(5); } }\n"
     "                if (live) {\n                    alpha *= 1.f - p.frag;"),
]
HEAD = ('extern "C" { long g_census[16]; }\n#define CNT(i) __atomic_fetch_add(&g_census[i], 1L, __ATOMIC_RELAXED)\n')


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    import host_raster as HR
    from helpers import scene
    from oracle import torch_ref as TR
    from umr_amd._lib import SIGNATURES
    tmp = tempfile.mkdtemp(prefix="umr_fcensus_")
    csrc = os.path.join(tmp, "csrc")
    shutil.copytree(os.path.join(ROOT, "umr_amd", "csrc"), csrc)
    p = os.path.join(csrc, "raster_forward.h")
    s = open(p).read()
    for a, b in INJECT:
        assert s.count(a) == 1, a
        s = s.replace(a, b)
    open(p, "w").write(s.replace("#pragma once\n", "#pragma once\n" + HEAD))
    so = os.path.join(tmp, "libfcensus.so")
    subprocess.check_call([HR.CLANG, "-O1", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", "-Wno-unknown-attributes",
                           "-Wno-ignored-attributes", '-DUMR_SRC_HASH="census"', "-DUMR_TU_STATS", "-I" + os.path.join(ROOT, "include"), "-I" + HR.SRC_DIR,
                           '-DUMR_TU="%s"' % os.path.join(csrc, "raster.hip"), "-shared", os.path.join(HR.SRC_DIR, "host_tu.cpp"), "-o", so])
    L = ctypes.CDLL(so)
    for name, (argtypes, restype) in SIGNATURES.items():
        if hasattr(L, name):
            getattr(L, name).argtypes, getattr(L, name).restype = argtypes, restype
    census = (ctypes.c_long * 16).in_dll(L, "g_census")
    for scale in ((0.6, 0.9), (0.95, 1.05)):
        verts, faces, cams, g = scene(32, 3, seed=0, scale=scale)
        pv = TR.orthographic_proj_withz(verts[:n], cams[:n], offset_z=5.) * torch.tensor([1., -1., 1.])
        fv = np.ascontiguousarray(TR.face_vertices(TR.look_at_ortho(pv), faces[:n]).numpy(), np.float32)
        tex = np.random.default_rng(0).random((n, faces.shape[1], 36, 3), dtype=np.float32)
        for i in range(16):
            census[i] = 0
        HR.forward(fv, tex, 512, pooled=True, visibility=True, dist_eps_log=float(np.float32(np.log(1. / 1e-10 - 1.))), L=L)
        print("scale %s, textured forward with p2f + visibility planes, per mesh:" % (scale,))
        for k, name in enumerate(NAMES):
            print("  %-36s %10.0f" % (name, census[k] / n))


if __name__ == "__main__":
    main()
