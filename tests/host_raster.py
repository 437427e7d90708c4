"""TEST INFRASTRUCTURE: the product's raster translation unit (umr_amd/csrc/raster.hip -- every kernel, the launch sequences, the
C-ABI entry points) compiled for x86-64 on the wave64 emulator of tests/host_kernel/wave_emu.h, with numpy front-ends that
mirror how umr_amd/functional.py drives libumr_hip.so.  Host pointers in place of device pointers, same prototypes
(umr_amd._lib.SIGNATURES).  Used by tests/test_raster_library_on_host.py and tools/fuzz_host_raster.py; never by the product."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SRC_DIR = os.path.join(HERE, "host_kernel")
SO = os.path.join(SRC_DIR, "libraster_host.so")

NO_P2F, ALPHA_ONLY, FACE_ID_ONLY = 1, 2, 4      # UMR_RASTER_* (include/umr_hip.h)
BWD_GRAD_POOLED, BWD_ALPHA_ONLY = 1, 2          # UMR_BWD_*


def available():
    return os.path.exists(CLANG)


def build(extra_flags=(), out=SO):
    csrc = os.path.join(ROOT, "umr_amd", "csrc")
    deps = [os.path.join(SRC_DIR, f) for f in ("raster_host.cpp", "wave_emu.h")] + \
           [os.path.join(csrc, f) for f in os.listdir(csrc) if f.startswith("raster") or f == "umr_common.h"] + \
           [os.path.join(ROOT, "include", "umr_hip.h")]
    if extra_flags or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([CLANG, "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-function",
                               "-Wno-unknown-attributes", "-Wno-ignored-attributes", '-DUMR_SRC_HASH="host-emulation"'] +
                              list(extra_flags) + [os.path.join(SRC_DIR, "raster_host.cpp"), "-o", out])
    return out


_LIB = {}


def lib(path=None):
    path = path or build()
    if path not in _LIB:
        from umr_amd._lib import SIGNATURES
        L = ctypes.CDLL(path)
        for name in ("umr_raster_workspace_bytes", "umr_raster_forward", "umr_raster_forward_vis", "umr_raster_backward",
                     "umr_debug_set", "umr_version", "umr_build_id"):
            if name in SIGNATURES:
                getattr(L, name).argtypes, getattr(L, name).restype = SIGNATURES[name]
        L.umr_host_emu_stats.argtypes = [ctypes.POINTER(ctypes.c_long)] * 3
        _LIB[path] = L
    return _LIB[path]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def standard_grid(IS):
    """affine_grid(identity), align_corners=True (functional/soft_rasterize.py:57-62 under torch 1.1): linspace(-1, 1, IS)."""
    l = np.linspace(-1.0, 1.0, IS, dtype=np.float64).astype(np.float32) if IS > 1 else np.zeros(1, np.float32)
    gx, gy = np.meshgrid(l, l)
    return np.ascontiguousarray(np.stack([gx, gy], -1), np.float32)


def _scalars(near, far, eps, sigma_val, dist_eps_log, gamma_val, func_id_rgb, double_side, func_id_dist=2, func_id_alpha=2, tex_type=0):
    return (float(near), float(far), float(eps), float(sigma_val), int(func_id_dist), float(dist_eps_log), float(gamma_val),
            int(func_id_rgb), int(func_id_alpha), int(tex_type), int(bool(double_side)))


def forward(faces, textures, image_size, background=(0, 0, 0), near=1.0, far=100.0, eps=1e-3, sigma_val=1e-5,
            dist_eps_log=None, gamma_val=1e-4, func_id_rgb=1, double_side=True, flags=0, pooled=False, visibility=False,
            background_by_value=False, tex_group=1, L=None, **modes):
    """umr_raster_forward[_vis] on host arrays.  Returns dict of the output buffers (+ 'ws' = the workspace, reusable)."""
    L = L or lib()
    faces = np.ascontiguousarray(faces, np.float32).reshape(faces.shape[0], faces.shape[1], 9)
    N, F = faces.shape[:2]
    IS = int(image_size)
    tex = None if textures is None else np.ascontiguousarray(textures, np.float32)
    TS = 1 if tex is None else tex.shape[2]
    alpha_only = bool(flags & ALPHA_ONLY)
    fi = np.zeros((N, F, 27), np.float32)
    ag = np.zeros((N, 2, IS, IS), np.float32)
    pi, ps = np.zeros((N, F, 2), np.float32), np.zeros((N, F, 2), np.float32)
    if alpha_only:
        sc = np.full((N, IS, IS), np.nan, np.float32)
    else:
        sc = np.full((N, 4, IS, IS), np.nan, np.float32) if background_by_value else np.ones((N, 4, IS, IS), np.float32)
        if not background_by_value:
            for k in range(3):
                sc[:, k] *= np.float32(background[k])
    pool = (np.full((N, IS // 2, IS // 2) if alpha_only else (N, 4, IS // 2, IS // 2), np.nan, np.float32)) if pooled else None
    vis = np.full((N, 2, IS, IS), np.nan, np.float32) if visibility else None
    grid = standard_grid(IS)
    wsb = L.umr_raster_workspace_bytes(N, F)
    ws = np.zeros(wsb + 64, np.uint8)
    bg = (ctypes.c_float * 3)(*[float(b) for b in background]) if background_by_value else None
    scal = _scalars(near, far, eps, sigma_val, dist_eps_log, gamma_val, func_id_rgb, double_side, **modes)
    fl = int(flags) | ((tex_group & 0xffff) << 8 if tex_group > 1 else 0)
    args = (_p(faces), _p(tex), _p(fi), _p(ag), _p(grid), _p(pi), _p(ps), _p(sc), _p(pool), N, F, TS, IS) + scal + \
           (fl, bg, _p(ws), wsb, None)
    rc = L.umr_raster_forward_vis(*args, _p(vis)) if visibility else L.umr_raster_forward(*args)
    if rc != 0:
        raise RuntimeError("umr_raster_forward (host emulation) rc=%d" % rc)
    return dict(faces=faces, textures=tex, faces_info=fi, aggrs_info=ag, p2f_info=pi, p2f_sum=ps, soft_colors=sc, pooled=pool,
                visibility=vis, ws=ws)


def backward(faces, textures, soft_colors, aggrs_info, grad_soft_colors, image_size, near=1.0, far=100.0, eps=1e-3,
             sigma_val=1e-5, dist_eps_log=None, gamma_val=1e-4, func_id_rgb=1, double_side=True, need_gf=True, need_gt=True,
             grad_flags=0, tex_group=1, L=None, **modes):
    """umr_raster_backward on host arrays -> (grad_faces | None, grad_textures | None)."""
    L = L or lib()
    faces = np.ascontiguousarray(faces, np.float32).reshape(faces.shape[0], faces.shape[1], 9)
    N, F = faces.shape[:2]
    IS = int(image_size)
    tex = None if textures is None else np.ascontiguousarray(textures, np.float32)
    TS = 1 if tex is None else tex.shape[2]
    sc = np.ascontiguousarray(soft_colors, np.float32)
    ag = None if aggrs_info is None else np.ascontiguousarray(aggrs_info, np.float32)
    g = np.ascontiguousarray(grad_soft_colors, np.float32)
    gf = np.zeros((N, F, 9), np.float32) if need_gf else None
    gt = np.zeros((N, F, TS, 3), np.float32) if need_gt else None
    wsb = L.umr_raster_workspace_bytes(N, F)
    ws = np.zeros(wsb + 64, np.uint8)
    scal = _scalars(near, far, eps, sigma_val, dist_eps_log, gamma_val, func_id_rgb, double_side, **modes)
    fl = int(grad_flags) | ((tex_group & 0xffff) << 8 if tex_group > 1 else 0)
    rc = L.umr_raster_backward(_p(faces), _p(tex), _p(sc), None, _p(ag), _p(gf), _p(gt), _p(g), fl, int(need_gf), int(need_gt),
                               N, F, TS, IS, *scal, _p(ws), wsb, None)
    if rc != 0:
        raise RuntimeError("umr_raster_backward (host emulation) rc=%d" % rc)
    return gf, gt


def stats(L=None):
    L = L or lib()
    a, b, c = ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
    L.umr_host_emu_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    return dict(workgroups=a.value, lane_switches=b.value, wave_operations=c.value)
