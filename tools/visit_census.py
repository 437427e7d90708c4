"""Where the face-major backward's lane slots go, counted on the CPU: the product's kernel source (umr_amd/csrc, copied to a
scratch directory -- the product tree is not touched) with event counters injected at the stations of a visit, compiled for the
wave64 emulator and run on meshes of the fixed SURVEY 8d scene (bench.py:fixed_scene_kernel_times).  Prints, per variant,
lanes at each station per mesh:  candidates culled -> wanted sub-tiles -> lanes handed a pixel -> not state-dead -> inside the
band (eval_pair) -> depth in range -> non-zero weight.  Usage: python tools/visit_census.py [n_meshes=2]"""
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

STATIONS = ["live_face_lanes", "visited_face_lanes", "cull_candidates", "pass_band_test", "wanted_after_state_cull",
            "lanes_at_visit_top", "lanes_with_subtile", "lanes_after_wave_dead_vote", "lanes_in_band(eval_pair)",
            "lanes_depth_in_range", "lanes_nonzero_weight", "lanes_not_dead_individually", "visits(wave)", "cull_passes(wave)"]

INJECT = [
    ("                if (ti < tb_end) {\n", "                if (ti < tb_end) { CNT(2);\n"),
    ("                                        0.5f * (cyh - cyl), thr_cull);\n", "                                        0.5f * (cyh - cyl), thr_cull); if (want) CNT(3);\n"),
    ("                unsigned long long tm = __ballot(want);\n", "                unsigned long long tm = __ballot(want); if (want) CNT(4); if (lane == 0) CNT(13);\n"),
    ("                    if (mine < 0) continue;\n", "                    CNT(5); if (lane == 0) { CNT(12); CFACE(n, f); } if (mine < 0) continue; CNT(6);\n"),
    ("                        if ((RGB == 2 || !NEED_GF || AG) && __all(dead)) continue;\n",
     "                        if (!dead) CNT(11); if ((RGB == 2 || !NEED_GF || AG) && __all(dead)) continue; CNT(7);\n"),
    ("                    if (!eval_pair(p, fc, xp, yp, c_thr2, c_nis, A.amb_thr)) continue;\n",
     "                    if (!eval_pair(p, fc, xp, yp, c_thr2, c_nis, A.amb_thr)) continue; CNT(8); CREC(n, f, pn4 >> 2);\n"),
    ("                    if (zp < c_near || zp > c_far) continue;  // :592\n", "                    if (zp < c_near || zp > c_far) continue; CNT(9);\n"),
    ("                        const int tix = texel_index(q0, q1, A.R);\n                        if (NEED_GT) {\n",
     "                        const int tix = texel_index(q0, q1, A.R); if (ps != 0.f) CNT(10);\n                        if (NEED_GT) {\n"),
    ("    if (live) {\n        // VGPR-resident operands", "    if (live) { CNT(0);\n        // VGPR-resident operands"),
    ("    if (FM_SKIP_EMPTY && FM_WAVES == 1 && !visited) return;\n", "    if (FM_SKIP_EMPTY && FM_WAVES == 1 && !visited) return; CNT(1);\n"),
]
HEAD = ('#ifndef UMR_CENSUS\n#define UMR_CENSUS\nextern "C" { long g_census[16]; long g_nrec; unsigned long long g_rec[1 << 23]; long g_facevis[1 << 16]; }\n'
        '#define CFACE(n, f) __atomic_fetch_add(&g_facevis[((n) * 2048 + (f)) & 0xffff], 1L, __ATOMIC_RELAXED)\n'
        '#define CNT(i) __atomic_fetch_add(&g_census[i], 1L, __ATOMIC_RELAXED)\n'
        '#define CREC(n, f, pix) do { long i_ = __atomic_fetch_add(&g_nrec, 1L, __ATOMIC_RELAXED); if (i_ < (1 << 23)) g_rec[i_] = '
        '((unsigned long long)(n) << 48) | ((unsigned long long)(f) << 32) | (unsigned)(pix); } while (0)\n#endif\n')


def build():
    import host_raster as HR
    tmp = tempfile.mkdtemp(prefix="umr_census_")
    csrc = os.path.join(tmp, "csrc")
    shutil.copytree(os.environ.get("CENSUS_CSRC", os.path.join(ROOT, "umr_amd", "csrc")), csrc)     # CENSUS_CSRC: an experimental copy
    p = os.path.join(csrc, "raster_backward_fm.h")
    s = open(p).read()
    for a, b in INJECT:
        assert s.count(a) == 1, a
        s = s.replace(a, b)
    open(p, "w").write(HEAD + s)
    out = os.path.join(tmp, "libcensus.so")
    flags = ["-O1", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", "-Wno-unknown-attributes", "-Wno-ignored-attributes",
             '-DUMR_SRC_HASH="census"', "-DUMR_TU_STATS", "-I" + os.path.join(ROOT, "include"), "-I" + HR.SRC_DIR,
             '-DUMR_TU="%s"' % os.path.join(csrc, "raster.hip")] + os.environ.get("CENSUS_FLAGS", "").split()
    subprocess.check_call([HR.CLANG] + flags + ["-shared", os.path.join(HR.SRC_DIR, "host_tu.cpp"), "-o", out])
    return out


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(argv[0]) if argv else 2
    scene_file = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--scene=")), None)      # a frozen capture (profiles/scenes/*.npz)
    meshes = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--meshes=")), None)         # "12,4": which of its views
    import host_raster as HR
    from helpers import scene
    from oracle import torch_ref as TR
    so = build()
    from umr_amd._lib import SIGNATURES
    L = ctypes.CDLL(so)                      # the raster translation unit only: prototypes of the symbols it has
    for name, (argtypes, restype) in SIGNATURES.items():
        if hasattr(L, name):
            getattr(L, name).argtypes, getattr(L, name).restype = argtypes, restype
    census = (ctypes.c_long * 16).in_dll(L, "g_census")
    facevis = (ctypes.c_long * (1 << 16)).in_dll(L, "g_facevis")
    verts, faces, cams, g = scene(32, 3, seed=0)
    if scene_file:
        z = np.load(scene_file)
        pick = [int(m) for m in meshes.split(",")] if meshes else list(range(n))
        fv = np.ascontiguousarray(z["fv_shared"][pick], np.float32)
        n = len(pick)
        print("scene %s, views %s" % (scene_file, pick))
    else:
        pv = TR.orthographic_proj_withz(verts[:n], cams[:n], offset_z=5.) * torch.tensor([1., -1., 1.])
        fv = np.ascontiguousarray(TR.face_vertices(TR.look_at_ortho(pv), faces[:n]).numpy(), np.float32)
    IS, TS = 512, 36
    DEL = float(np.float32(np.log(1. / 1e-10 - 1.)))
    rng = np.random.default_rng(0)
    tex = rng.random((n, faces.shape[1], TS, 3), dtype=np.float32)

    nrec = ctypes.c_long.in_dll(L, "g_nrec")
    rec = (ctypes.c_ulonglong * (1 << 23)).in_dll(L, "g_rec")

    def shapes():
        """in-band (pixel, face) pairs of the launch -> lanes a hand-out in w x h pieces would have to visit (pieces with at least
        one in-band pixel: the lower bound a conservative piece test approaches)"""
        k = min(nrec.value, 1 << 23)
        r = np.frombuffer(rec, dtype=np.uint64, count=k).copy()
        nrec.value = 0
        nf = r >> np.uint64(32)
        pix = (r & np.uint64(0xffffffff)).astype(np.int64)
        x, y = pix % IS, pix // IS
        res = {}
        for w, h in ((4, 4), (4, 2), (2, 4), (2, 2), (8, 1), (4, 1), (2, 1), (1, 1)):
            key = (nf.astype(np.int64) << 32) | ((y // h) << 16) | (x // w)
            res["%dx%d" % (w, h)] = len(np.unique(key)) * w * h / n
        print("  in-band pairs %.0f / mesh; lanes to visit by piece shape: %s" % (k / n, {a: int(b) for a, b in res.items()}))

    def report(tag):
        c = [census[i] / n for i in range(14)]
        print("== %s (per mesh)" % tag)
        for name, v in zip(STATIONS, c):
            print("  %-32s %12.0f" % (name, v))
        for i in range(16):
            census[i] = 0
        shapes()
        # wave visits per face (work items of a split face add up to their face): what ONE wave per face would have to walk
        fvis = np.frombuffer(facevis, dtype=np.int64, count=1 << 16).copy().reshape(32, 2048)[:n, :fv.shape[1]]
        ctypes.memset(facevis, 0, ctypes.sizeof(facevis))
        v = fvis.ravel()
        print("  wave visits per face: mean %.1f  median %.0f  p99 %.0f  max %d; faces never visited %.0f %%; the 10 heaviest carry %.1f %% of all visits"
              % (v.mean(), np.median(v), np.percentile(v, 99), v.max(), 100.0 * (v == 0).mean(), 100.0 * np.sort(v)[-10:].sum() / max(v.sum(), 1)))
        return c

    out = HR.forward(fv, tex, IS, pooled=True, dist_eps_log=DEL, L=L)
    g_rgb = rng.standard_normal((n, 4, IS // 2, IS // 2)).astype(np.float32)
    HR.backward(fv, tex, out["soft_colors"], out["aggrs_info"], g_rgb, IS, need_gf=False, need_gt=True, grad_flags=HR.BWD_GRAD_POOLED, dist_eps_log=DEL, L=L)
    report("texel-gradient backward <1, false, true>")
    HR.backward(fv, tex, out["soft_colors"], out["aggrs_info"], g_rgb, IS, need_gf=True, need_gt=True, grad_flags=HR.BWD_GRAD_POOLED, dist_eps_log=DEL, L=L)
    report("vertex + texel backward <1, true, true>")
    HR.backward(fv, tex, out["soft_colors"], out["aggrs_info"], g_rgb, IS, need_gf=True, need_gt=True,
                grad_flags=HR.BWD_GRAD_POOLED | HR.BWD_ALPHA_GEOMETRY, dist_eps_log=DEL, L=L)
    report("one-pass backward of the shared render (alpha -> vertices, rgb -> texels)")
    outa = HR.forward(fv, None, IS, flags=HR.ALPHA_ONLY | HR.NO_P2F, pooled=True, dist_eps_log=DEL, L=L)
    g_a = rng.standard_normal((n, IS // 2, IS // 2)).astype(np.float32)
    HR.backward(fv, None, outa["soft_colors"], None, g_a, IS, need_gf=True, need_gt=False,
                grad_flags=HR.BWD_GRAD_POOLED | HR.BWD_ALPHA_ONLY, dist_eps_log=DEL, L=L)
    report("silhouette backward <2, true, false>")


if __name__ == "__main__":
    main()
