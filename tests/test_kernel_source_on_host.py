"""CPU: the product's OWN pair-geometry source (umr_amd/csrc/raster_core.h: k_face_setup, eval_pair, clip_depth -- the file
the GPU library is built from, unmodified) compiled for the host through tests/host_kernel/device_shim.h, one lane at a
time, and fuzzed against the oracle: one face per mesh, so the oracle's alpha plane is that face's soft fragment per pixel.
Every class must agree with the reference's render -- ordinary faces as well as needles, sub-pixel faces and faces with an
edge seen end-on (the class that produced round 2's sporadic NaN: their `den` is rounding noise, often exactly 0, and
eval_pair takes the reference's own evaluate-all-three route for them)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HK = os.path.join(ROOT, "tests", "host_kernel")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
f32 = np.float32


@pytest.fixture(scope="module")
def host_lib():
    if not os.path.exists(CLANG):
        pytest.skip("clang++ of the ROCm toolchain not present")
    so = os.path.join(HK, "libpair_host.so")
    srcs = [os.path.join(HK, "pair_host.cpp"), os.path.join(HK, "device_shim.h"),
            os.path.join(ROOT, "umr_amd", "csrc", "raster_core.h"), os.path.join(ROOT, "umr_amd", "csrc", "umr_common.h"),
            os.path.join(ROOT, "umr_amd", "csrc", "raster_general.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call([CLANG, "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-function",
                               "-Wno-unknown-attributes", srcs[0], "-o", so])
    h = ctypes.CDLL(so)
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    h.host_pairs.argtypes = [P, I, P, P, I, F, F, F, F, F, P, P, P, P]
    h.host_pairs.restype = I
    h.host_tile_may_hit.argtypes = [P, I, P, I, F, P]
    h.host_tile_may_hit.restype = I
    h.host_texels.argtypes = [P, I, P, P, I, F, F, F, I, P]
    h.host_texels.restype = I
    h.host_general_frag.argtypes = [P, I, P, P, I, F, F, F, I, P, P]
    h.host_general_frag.restype = I
    h.host_ndc.argtypes = [I, P, P]
    h.host_ndc.restype = I
    h.host_tile_setup.argtypes = [I, I, I, ctypes.c_uint, ctypes.c_uint, P]
    h.host_tile_setup.restype = I
    h.host_fm_owned_face.argtypes = [I, I, I, I]
    h.host_fm_owned_face.restype = I
    h.host_replay_face.argtypes = [P, I, I, I, F, F, F, F, F, F, F, F, F, P, I]
    h.host_replay_face.restype = I
    h.host_cull_granularity.argtypes = [P, I, I, F, F, F, F, P, P]
    h.host_cull_granularity.restype = I
    return h


def _pixels(IS):
    px = ((2 * np.arange(IS) + 1 - IS) / IS).astype(f32)
    xi, ri = np.meshgrid(np.arange(IS), np.arange(IS))
    return np.ascontiguousarray(px[xi.ravel()]), np.ascontiguousarray(px[(IS - 1 - ri).ravel()])


def _alpha_from_kernel_source(h, fv, IS, sigma, dist_eps_log):
    n = len(fv)
    px = ((2 * np.arange(IS) + 1 - IS) / IS).astype(f32)
    xi, ri = np.meshgrid(np.arange(IS), np.arange(IS))
    xp = np.ascontiguousarray(px[xi.ravel()]); yp = np.ascontiguousarray(px[(IS - 1 - ri).ravel()])
    threshold = f32(f32(dist_eps_log) * f32(sigma))
    thr, nis = f32(np.sqrt(threshold)), f32(-1.0 / f32(sigma))
    faces = np.ascontiguousarray(fv.reshape(n, 9), f32)
    npix = IS * IS
    live = np.zeros((n, npix), np.uint8); frag = np.zeros((n, npix), f32); dxy = np.zeros((n, npix, 2), f32); zp = np.zeros((n, npix), f32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert h.host_pairs(p(faces), n, p(xp), p(yp), npix, float(thr), float(threshold), float(nis), 1.0, 100.0, p(live), p(frag), p(dxy),
                        p(zp)) == 0
    return np.where(live != 0, frag, f32(0)), zp


def test_kernel_pair_geometry_on_host_vs_oracle(host_lib, oracle_built):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from eval_pair_model import fuzz_cases
    from oracle import softras
    IS, n = 48, 160
    sigma, del_ = 1e-5, float(np.log(1. / 1e-10 - 1.))
    cases, px = fuzz_cases(n, IS, np.random.default_rng(3))
    rng = np.random.default_rng(7)
    r = lambda *s_: rng.uniform(-1, 1, s_).astype(f32)
    pc = px[rng.integers(0, IS, (n, 2))]
    th = rng.uniform(0, 2 * np.pi, n).astype(f32)
    u, v = np.stack([np.cos(th), np.sin(th)], 1).astype(f32), np.stack([-np.sin(th), np.cos(th)], 1).astype(f32)
    z = np.full((n, 3, 1), 7.7, f32)
    cases["needle_any_direction"] = np.concatenate([np.stack([pc - 2e-5 * u - 1e-5 * v, pc + 2e-5 * u - 1e-5 * v,
                                                              pc + (0.03 + 0.05 * np.abs(r(n, 1))) * v + 0.01 * r(n, 1) * u], 1).astype(f32), z], 2)
    cases["needle_wide"] = np.concatenate([np.stack([pc - 2e-4 * u - 1e-4 * v, pc + 2e-4 * u - 1e-4 * v,
                                                     pc + (0.03 + 0.05 * np.abs(r(n, 1))) * v], 1).astype(f32), z], 2)
    cases["sub_pixel_right_angle"] = np.concatenate([np.stack([pc, pc + np.array([1e-3, 0], f32), pc + np.array([0, 1e-3], f32)], 1).astype(f32), z], 2)
    cases["two_edges_collapsed"] = np.concatenate([np.stack([pc, pc + 2e-5 * u, pc + 3e-5 * v], 1).astype(f32), z], 2)
    cfg = dict(near=1., far=100., eps=1e-3, sigma_val=sigma, dist_eps_log=del_, gamma_val=1e-4, func_id_rgb=1, double_side=True)
    report = {}
    for name, fv in cases.items():
        ref = softras.raster_forward(fv.reshape(n, 1, 9), np.ones((n, 1, 1, 3), f32), IS, background=(0, 0, 0), backend="port",
                                     n_threads=8, **cfg)
        ra = ref["soft_colors"][:, 3].reshape(n, -1)
        alpha, zp = _alpha_from_kernel_source(host_lib, fv, IS, sigma, del_)
        assert np.isfinite(alpha).all() and np.isfinite(zp).all(), name + ": the kernel source produced a non-finite value"
        err = np.abs(alpha.astype(np.float64) - ra)
        report[name] = (int((err > 1e-4).sum()), int((ra > 0).sum()))
    for name, (bad, live) in report.items():   # every class, needles and sub-pixel faces included: the reference's render up to
        assert bad <= 2, (name, bad, live)     # a tie or two between two nearest edges


def test_tile_culling_of_the_kernel_source_is_conservative(host_lib):
    """The bbox + tile_may_hit test that forward and backward cull with (raster_core.h) must not drop a tile in which some
    pixel contributes: for 12 classes of faces, every 8x8 and 4x4 tile of a 48^2 image that holds a pixel with a soft fragment
    above 1e-6 must pass the test -- checked with the kernel's own source on the host.  (Below 1e-6: on needles the
    reference's COMPUTED distance can be ~30 % short of the geometric one -- its arithmetic is ill conditioned there -- so a
    pixel 1.2 thresholds away from the needle still gets D ~ 1e-7 from the reference while the geometric cull, rightly,
    drops its tile; parity is unaffected at the 1e-4 the renders are held to.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from eval_pair_model import fuzz_cases
    IS, n = 48, 120
    sigma, del_ = 1e-5, float(np.log(1. / 1e-10 - 1.))
    thr = float(np.sqrt(f32(f32(del_) * f32(sigma))))
    cases, px = fuzz_cases(n, IS, np.random.default_rng(11))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for T in (8, 4):
        nt = IS // T
        lo = px[np.arange(nt) * T]; hi = px[np.arange(nt) * T + T - 1]
        cxs, hxs = 0.5 * (lo + hi), 0.5 * (hi - lo)
        # tile (ty, tx): image rows ty*T.. -> y index IS-1-row: centres px[IS-1-row]
        ylo = px[IS - 1 - (np.arange(nt) * T + T - 1)]; yhi = px[IS - 1 - np.arange(nt) * T]
        cys, hys = 0.5 * (ylo + yhi), 0.5 * (yhi - ylo)
        tiles = np.ascontiguousarray(np.stack([np.tile(cxs, nt), np.repeat(cys, nt), np.tile(hxs, nt), np.repeat(hys, nt)], 1), f32)
        for name, fv in cases.items():
            alpha, _ = _alpha_from_kernel_source(host_lib, fv, IS, sigma, del_)
            live = (alpha > 1e-6).reshape(n, nt, T, nt, T).any(axis=(2, 4)).reshape(n, nt * nt)     # [n, ty*nt+tx]
            out = np.zeros((n, nt * nt), np.uint8)
            assert host_lib.host_tile_may_hit(p(np.ascontiguousarray(fv.reshape(n, 9), f32)), n, p(tiles), nt * nt, thr, p(out)) == 0
            dropped = live & (out == 0)
            assert not dropped.any(), "%s, %dx%d tiles: %d needed tiles culled" % (name, T, T, int(dropped.sum()))


def test_texel_lookup_of_the_kernel_source_vs_oracle(host_lib, oracle_built):
    """clip_depth + texel_index of the kernel source: with a 'hard' render of ONE face whose 36 texels carry their own index
    as colour, the oracle's image names the texel every covered pixel samples (soft_rasterize_cuda_kernel.cu:54-59,
    :180-189, :408-414) -- the host-compiled kernel source must name the same one."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from eval_pair_model import fuzz_cases
    from oracle import softras
    IS, n, R = 48, 200, 6
    sigma, del_ = 1e-5, float(np.log(1. / 1e-10 - 1.))
    threshold = f32(f32(del_) * f32(sigma)); thr, nis = float(np.sqrt(threshold)), float(f32(-1.0 / f32(sigma)))
    cases, _ = fuzz_cases(n, IS, np.random.default_rng(21))
    xp, yp = _pixels(IS)
    tex = np.zeros((n, 1, R * R, 3), f32)
    tex[:, 0, :, 0] = np.arange(R * R, dtype=f32)[None] + 1            # channel 0 = texel index + 1 (background stays 0)
    cfg = dict(near=1., far=100., eps=1e-3, sigma_val=sigma, dist_eps_log=del_, gamma_val=1e-4, func_id_rgb=0, double_side=True)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for name in ("random", "small", "vertex_on_pixel", "obtuse", "sliver", "collapsed_edge_1e-4"):
        fv = cases[name].copy()
        fv[:, :, 2] = 7.7 + np.random.default_rng(1).uniform(-1, 1, (n, 3)).astype(f32)        # perspective-correct weights matter
        ref = softras.raster_forward(fv.reshape(n, 1, 9), tex, IS, background=(0, 0, 0), backend="port", n_threads=8, **cfg)
        ref_tix = np.rint(ref["soft_colors"][:, 0].reshape(n, -1)).astype(np.int64) - 1          # -1 = not covered
        got = np.zeros((n, IS * IS), np.int32)
        assert host_lib.host_texels(p(np.ascontiguousarray(fv.reshape(n, 9), f32)), n, p(xp), p(yp), IS * IS, thr, float(threshold), nis, R,
                                    p(got)) == 0
        diff = int((got != ref_tix).sum())
        assert diff <= 2, (name, diff, int((ref_tix >= 0).sum()))        # (a tie on a texel boundary at most)


@pytest.mark.parametrize("dist_mode", [0, 1])
def test_general_mode_fragments_of_the_kernel_source_vs_oracle(host_lib, oracle_built, dist_mode):
    """gen_fragment of raster_general.h (hard / barycentric distance, :154-157, :365-372) on the host against the oracle's
    alpha plane of one-face meshes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from eval_pair_model import fuzz_cases
    from oracle import softras
    IS, n = 48, 200
    sigma, del_ = (1e-5 if dist_mode == 0 else 1e-4), float(np.log(1. / 1e-10 - 1.))
    threshold = f32(f32(del_) * f32(sigma)); thr, nis = float(np.sqrt(threshold)), float(f32(-1.0 / f32(sigma)))
    cases, _ = fuzz_cases(n, IS, np.random.default_rng(31))
    xp, yp = _pixels(IS)
    cfg = dict(near=1., far=100., eps=1e-3, sigma_val=sigma, dist_eps_log=del_, gamma_val=1e-4, func_id_rgb=1, double_side=True,
               func_id_dist=dist_mode, func_id_alpha=2, texture_sample_type=0)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for name in ("random", "small", "vertex_on_pixel", "obtuse", "two_equal", "collinear"):
        fv = cases[name]
        ref = softras.raster_forward(fv.reshape(n, 1, 9), np.ones((n, 1, 1, 3), f32), IS, background=(0, 0, 0), backend="port",
                                     n_threads=8, **cfg)
        ra = ref["soft_colors"][:, 3].reshape(n, -1)
        live = np.zeros((n, IS * IS), np.uint8); frag = np.zeros((n, IS * IS), f32)
        assert host_lib.host_general_frag(p(np.ascontiguousarray(fv.reshape(n, 9), f32)), n, p(xp), p(yp), IS * IS, thr, float(threshold),
                                          nis, dist_mode, p(live), p(frag)) == 0
        alpha = np.where(live != 0, frag, f32(0))
        assert np.isfinite(alpha).all(), name
        bad = int((np.abs(alpha.astype(np.float64) - ra) > 1e-4).sum())
        assert bad <= 2, (name, bad, int((ra > 0).sum()))


def test_pixel_coordinates_and_work_mapping_of_the_kernel_source(host_lib):
    """(a) the fp32 pixel-centre shortcut equals the reference's fp64 expression (:325-326) to the bit, for power-of-two and
    other image sizes; (b) every work mapping of the pixel-major kernels (plain, contiguous per XCD, row-interleaved per XCD)
    sends the launch's workgroups x threads onto every (mesh, pixel) exactly once -- tile_setup of the kernel source, run
    for all blocks and threads of a launch."""
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for IS in (16, 64, 512, 1024, 24, 136, 200):
        a, b = np.zeros(IS, f32), np.zeros(IS, f32)
        host_lib.host_ndc(IS, p(a), p(b))
        expect = ((2.0 * np.arange(IS) + 1.0 - IS) / IS).astype(f32)
        assert np.array_equal(a, expect) and np.array_equal(a.view(np.uint32), b.view(np.uint32)), IS
    out = np.zeros(5, np.int32)
    for (N, IS) in ((3, 128), (2, 136), (8, 64), (1, 24)):
        tiles = ((IS + 15) // 16) ** 2
        for mode in (0, 1, 2):
            seen = np.zeros((N, IS, IS), np.int32)
            for block in range(N * tiles):
                for thread in range(256):
                    host_lib.host_tile_setup(N, IS, mode, block, thread, p(out))
                    if out[3]:
                        seen[out[0], out[2], out[1]] += 1
            assert (seen == 1).all(), (N, IS, mode, int((seen != 1).sum()))
    # (c) face ownership of the face-major backward: for every split the 8 XCDs' runs partition the mesh's faces
    for F in (1280, 5120, 320, 64):
        per = F // 8
        for split in (1, 2, 4):
            if per % split:
                continue
            owned = sorted(host_lib.host_fm_owned_face(x, j, per, split) for x in range(8) for j in range(per))
            assert owned == list(range(F)), (F, split)


def test_depth_in_range_flag_of_the_kernel_source_is_sound(host_lib):
    """k_face_setup's flag "every vertex depth strictly inside (near, far)" lets the silhouette backward skip the
    perspective-correct depth and its range test (:592).  Sound only if the interpolated depth of such a face can never
    leave [near, far]: for faces the flag covers (vertex depths inside (near * 1.0001, far * 0.9999)), every live pixel's
    clip_depth of the kernel source must lie in range -- needles and slivers included."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from eval_pair_model import fuzz_cases
    IS, n = 48, 150
    sigma, del_ = 1e-5, float(np.log(1. / 1e-10 - 1.))
    near, far = 1.0, 100.0
    rng = np.random.default_rng(41)
    cases, _ = fuzz_cases(n, IS, rng)
    for name, fv in cases.items():
        fv = fv.copy()
        fv[:, :, 2] = np.exp(rng.uniform(np.log(near * 1.001), np.log(far * 0.999), (n, 3))).astype(f32)   # flag set for all
        alpha, zp = _alpha_from_kernel_source(host_lib, fv, IS, sigma, del_)
        lv = alpha > 0
        assert lv.any() or name == "tiny"
        z = zp[lv]
        assert np.isfinite(z).all() and (z >= near).all() and (z <= far).all(), (name, float(z.min()), float(z.max()))
        zmin, zmax = fv[:, :, 2].min(1), fv[:, :, 2].max(1)
        inside = (zp >= zmin[:, None] * (1 - 1e-5)) & (zp <= zmax[:, None] * (1 + 1e-5))     # a convex combination of the 1/z_k
        assert inside[lv].all(), name


def test_reference_order_geometry_carries_rounding_noise(host_lib):
    """Why every raster kernel keeps the reference's operation order.  The reference rebuilds the pixel from barycentrics
    times ABSOLUTE vertex coordinates (soft_rasterize_cuda_kernel.cu:63-152); a well-conditioned evaluation of the same closest
    boundary point (three clamped edge projections relative to the face's own vertices, tests/host_kernel) agrees with
    float64 to 1e-9 -- and differs from eval_pair (= the reference, bit for bit) by up to ~1e-3 in the soft fragment D for
    faces of the BASELINE meshes' size.  Round 3 measured that formulation as a faster backward on the MI355X: 1e-3 of D noise
    decides which pixels at the rim of the distance band contribute, and outside the silhouette those pixels carry soft-max
    weights of O(1) -- gradients off by more than their scale on ~1 % of the faces, inf where the forward had rejected the
    only face of a pixel (HISTORY.md 4.7).  This test pins the size of that noise on the kernel source: if it ever vanished
    (e.g. a reformulated eval_pair), the 1e-4 parity with the reference's render would have vanished with it."""
    host_lib.host_accurate_pairs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                             ctypes.c_float, ctypes.c_float] + [ctypes.c_void_p] * 3
    host_lib.host_accurate_pairs.restype = ctypes.c_int
    IS, n = 48, 200
    sigma, del_ = 1e-5, float(np.log(1. / 1e-10 - 1.))
    threshold = f32(f32(del_) * f32(sigma)); thr, nis = float(np.sqrt(threshold)), float(f32(-1.0 / f32(sigma)))
    rng = np.random.default_rng(51)
    # faces shaped like the meshes of the BASELINE configs: near-equilateral, edges 0.03 .. 0.2, anywhere on screen
    c = rng.uniform(-0.8, 0.8, (n, 1, 2)); ang = rng.uniform(0, 2 * np.pi, (n, 1)) + np.array([0, 2.1, 4.2])[None] + rng.normal(0, 0.25, (n, 3))
    rad = rng.uniform(0.02, 0.12, (n, 1)) * rng.uniform(0.7, 1.3, (n, 3))
    fv = np.concatenate([c + np.stack([rad * np.cos(ang), rad * np.sin(ang)], 2), np.full((n, 3, 1), 5.0)], 2).astype(f32)
    xp, yp = _pixels(IS)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    faces = np.ascontiguousarray(fv.reshape(n, 9), f32)
    npix = IS * IS
    live = np.zeros((n, npix), np.uint8); frag = np.zeros((n, npix), f32); dxy = np.zeros((n, npix, 2), f32); zp = np.zeros((n, npix), f32)
    assert host_lib.host_pairs(p(faces), n, p(xp), p(yp), npix, thr, float(threshold), nis, 1.0, 100.0, p(live), p(frag), p(dxy), p(zp)) == 0
    alive = np.zeros((n, npix), np.uint8); afrag = np.zeros((n, npix), f32); q = np.zeros((n, npix, 2), f32)
    assert host_lib.host_accurate_pairs(p(faces), n, p(xp), p(yp), npix, thr, float(threshold), nis, p(alive), p(afrag), p(q)) == 0
    # the accurate evaluation against float64 geometry
    X, Y = fv[:, :, 0].astype(np.float64), fv[:, :, 1].astype(np.float64)
    best = np.full((n, npix), np.inf)
    for e in range(3):
        ax, ay = X[:, e, None], Y[:, e, None]; ex, ey = X[:, (e + 1) % 3, None] - ax, Y[:, (e + 1) % 3, None] - ay
        t = np.clip(((xp[None] - ax) * ex + (yp[None] - ay) * ey) / (ex * ex + ey * ey), 0, 1)
        best = np.minimum(best, (xp[None] - ax - t * ex) ** 2 + (yp[None] - ay - t * ey) ** 2)
    d2a = (q.astype(np.float64) ** 2).sum(2)
    band = best < 4e-4
    assert np.abs(np.sqrt(d2a) - np.sqrt(best))[band].max() <= 2e-8                     # 5e-6 px at IS = 512
    both = (live != 0) & (alive != 0)
    dD = np.abs(frag.astype(np.float64) - afrag)[both]
    off = np.abs(dxy.astype(np.float64) + q).max(axis=2)[both]                         # eval_pair's (dx, dy) = Q - P; q = P - Q
    print("[reference-order noise] |dD| max %.2e p99 %.2e   |d offset| max %.2e p99 %.2e" % (dD.max(), np.percentile(dD, 99), off.max(),
                                                                                            np.percentile(off, 99)))
    assert 1e-4 <= dD.max() <= 1e-2 and np.percentile(dD, 99) <= 1e-3        # measured: 1.5e-3 / 2.3e-4
    assert 1e-6 <= off.max() <= 1e-4                                         # measured: 2.4e-5 (6e-3 px at IS = 512)
    # and the contributing SETS differ exactly where the weight sits on the 1e-10 threshold
    only = (live != 0) ^ (alive != 0)
    on_boundary = d2a < 1e-13
    w = np.where(live != 0, frag, 0).astype(np.float64); wa = np.where(alive != 0, afrag, 0).astype(np.float64)
    assert ((w < 2e-10) & (wa < 2e-10) | on_boundary)[only].all()


def test_tile_culls_of_both_directions_agree_where_a_pixel_is_included(host_lib):
    """The cause of round 3's non-finite training runs at BASELINE configs[3] (HISTORY.md 10), on the geometry two tripwire runs
    on the MI355X captured (tests/golden/nan_cfg4_faces.npz: the faces around the offender of the blamed view, written by
    tools/nan/make_nan_fixture.py).  The forward culls per 8x8 wave tile, the face-major backward per 4x4 sub-tile, then both decide
    per pixel with eval_pair.  With the cull band at the EXACT threshold (round 3's early builds: noise_scale 0) a thin face's
    corner pixel of a tile -- included by the reference's own noisy distance with the smallest possible fragment, 1.03e-10 -- was
    dropped by the 8x8 test and kept by the 4x4 test (the two evaluate the same maximum through different tile centres and round
    differently): the backward then weighs a pair the forward never saw, the pixel's saved soft-max maximum is the background's
    eps, and exp((zn - max) / gamma) = exp(9300) overflows -- an infinite texel / vertex gradient, NaN parameters one step later.
    With the band widened by the reference's distance noise (R_CULL, the shipped library) such pixels lie far inside the band:
    both directions include the same pairs and every weight is a soft-max weight <= 1."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "nan_cfg4_faces.npz"))
    sigma, gamma = 1e-5, 1e-4
    threshold = float(np.log(1e10 - 1.0) * sigma)
    thr = float(np.sqrt(f32(threshold)))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for c in range(2):
        fv, off, IS = np.ascontiguousarray(g["faces_%d" % c], f32), int(g["offender_%d" % c]), int(g["meta_%d" % c][5])
        res = {}
        for noise in (0.0, 1.0):
            out = np.zeros((20000, 8), f32)
            n = host_lib.host_replay_face(p(fv), fv.shape[0], off, IS, thr, threshold, -1.0 / sigma, gamma, 1.0, 100.0, 1e-3, noise,
                                          20.0 * sigma, p(out), out.shape[0])
            o = out[:n]
            code = o[:, 7].astype(int)
            fwd, bwd = (code & 1) > 0, (code & 2) > 0
            res[noise] = (int((bwd & ~fwd).sum()), int((fwd & ~bwd).sum()), o[bwd, 6], o[bwd & ~fwd])
        only_b, only_f, ps, rows = res[0.0]
        assert only_b == 1 and not np.isfinite(ps).all(), "the captured mechanism no longer reproduces with the exact band"
        assert abs(rows[0][2] - 1.03e-10) < 3e-12 and rows[0][4] == f32(1e-3) and int(rows[0][0]) % 8 in (0, 7) and int(rows[0][1]) % 8 in (0, 7)
        only_b, only_f, ps, _ = res[1.0]
        assert only_b == 0 and only_f == 0 and np.isfinite(ps).all() and ps.max() <= 1.0 + 1e-6 and len(ps) > 500
    # and as a count over random thin faces: with the shipped band no included pixel is dropped by either cull
    rng = np.random.default_rng(7)
    n = 4000
    c = rng.uniform(-0.95, 0.95, (n, 2)); L = rng.uniform(0.01, 0.12, n); hh = 10 ** rng.uniform(-5.0, -1.5, n); a = rng.uniform(0, np.pi, n)
    u, v = np.stack([np.cos(a), np.sin(a)], 1), np.stack([-np.sin(a), np.cos(a)], 1)
    fv = np.zeros((n, 3, 3), f32)
    fv[:, 0, :2] = c - 0.5 * L[:, None] * u; fv[:, 1, :2] = c + 0.5 * L[:, None] * u
    fv[:, 2, :2] = c + ((rng.uniform(0.05, 0.95, n) - 0.5) * L)[:, None] * u + hh[:, None] * v
    fv[:, :, 2] = rng.uniform(3, 6, (n, 3))
    fv = np.ascontiguousarray(fv.reshape(n, 9))
    cnt, first = (ctypes.c_long * 5)(), (ctypes.c_int * 4)()
    host_lib.host_cull_granularity(p(fv), n, 1024, thr, threshold, -1.0 / sigma, 1.0, cnt, first)
    # ... nor by the backward's refinement of a surviving 4x4 sub-tile into 2x2 quads (cnt[4]; the quad hand-out of round 4)
    assert cnt[0] > 1000000 and cnt[1] == 0 and cnt[2] == 0 and cnt[3] == 0 and cnt[4] == 0, list(cnt)
    host_lib.host_cull_granularity(p(fv), n, 1024, thr, threshold, -1.0 / sigma, 0.0, cnt, first)
    assert cnt[1] > 0 and cnt[2] > 0      # with the exact band both culls DO drop pixels the reference's arithmetic includes
    # the same count on ordinary faces (the bench's proportions): no included pixel is lost at any granularity
    n2 = 3000
    c2 = rng.uniform(-0.9, 0.9, (n2, 1, 2)); fv2 = np.zeros((n2, 3, 3), f32)
    fv2[:, :, :2] = c2 + rng.uniform(-0.06, 0.06, (n2, 3, 2)); fv2[:, :, 2] = rng.uniform(3, 6, (n2, 3))
    fv2 = np.ascontiguousarray(fv2.reshape(n2, 9))
    for IS in (512, 64):
        host_lib.host_cull_granularity(p(fv2), n2, IS, thr, threshold, -1.0 / sigma, 1.0, cnt, first)
        assert cnt[0] > 10000 and cnt[1] == 0 and cnt[2] == 0 and cnt[3] == 0 and cnt[4] == 0, (IS, list(cnt))
