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
SO = os.path.join(SRC_DIR, "libumr_host.so")
TUS = ["raster", "geometry", "losses", "perceptual", "edt", "atlas", "regs", "eval"]     # umr_amd/build.py SOURCES

NO_P2F, ALPHA_ONLY, FACE_ID_ONLY = 1, 2, 4      # UMR_RASTER_* (include/umr_hip.h)
BWD_GRAD_POOLED, BWD_ALPHA_ONLY, BWD_ALPHA_GEOMETRY, BWD_PACKED_STATE, BWD_REUSE_WORKSPACE = 1, 2, 4, 8, 16          # UMR_BWD_*


def available():
    return os.path.exists(CLANG)


def build(extra_flags=(), out=SO):
    """Every translation unit of libumr_hip.so (umr_amd/csrc/*.hip) compiled for x86-64 through host_kernel/host_tu.cpp and
    linked into one library with the C ABI of include/umr_hip.h."""
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(ROOT, "umr_amd", "csrc")
    deps = [os.path.join(SRC_DIR, f) for f in ("host_tu.cpp", "wave_emu.h")] + \
           [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))] + [os.path.join(ROOT, "include", "umr_hip.h")]
    if not (extra_flags or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps)):
        return out
    objdir = out + ".objs"
    os.makedirs(objdir, exist_ok=True)
    flags = ["-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", "-Wno-unknown-attributes",
             "-Wno-ignored-attributes", '-DUMR_SRC_HASH="host-emulation"'] + list(extra_flags)

    def one(tu):
        obj = os.path.join(objdir, tu + ".o")
        subprocess.check_call([CLANG] + flags + (["-DUMR_TU_STATS"] if tu == "raster" else []) +
                              ['-DUMR_TU="../../umr_amd/csrc/%s.hip"' % tu, "-c", os.path.join(SRC_DIR, "host_tu.cpp"), "-o", obj])
        return obj
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(one, TUS))
    subprocess.check_call([CLANG, "-shared", "-o", out] + objs)
    return out


_LIB = {}


def lib(path=None):
    path = path or build()
    if path not in _LIB:
        from umr_amd._lib import SIGNATURES
        L = ctypes.CDLL(path)
        L.umr_version.restype = ctypes.c_char_p
        L.umr_build_id.restype = ctypes.c_char_p
        for name, (argtypes, restype) in SIGNATURES.items():      # the product's own prototype table
            getattr(L, name).argtypes, getattr(L, name).restype = argtypes, restype
        L.umr_host_emu_stats.argtypes = [ctypes.POINTER(ctypes.c_long)] * 3
        _LIB[path] = L
    return _LIB[path]


class emulated_product:
    """Context manager for TESTS: umr_amd's Python layer (functional.py, smr.py, loss_utils.py, train_step.py, ...) on HOST
    tensors over the emulated library.  The product itself refuses host tensors and a missing libumr_hip.so (umr_amd/_lib.py:
    no CPU path); for the duration of a test the accessors that enforce that are replaced and the raster operators get a
    host kernel -- their own Python bodies, which then call into libumr_host.so.  Nothing under umr_amd/ knows about this."""
    _ops_registered = False

    def __enter__(self):
        import sys
        import torch
        from umr_amd import _lib, ops
        self._saved = (_lib._lib, _lib.ptr, _lib.stream_ptr, _lib.on_device, torch.cuda.synchronize)
        orig_ptr = _lib.ptr
        _lib._lib = lib()

        def ptr(t):
            if t is None:
                return None
            if t.is_cuda or not t.is_contiguous():
                raise RuntimeError("emulated_product: expected a contiguous host tensor")
            return t.data_ptr()
        _lib.ptr = ptr
        _lib.on_device = lambda t: True
        _lib.stream_ptr = lambda device=None: None
        torch.cuda.synchronize = lambda *a, **k: None
        self._by_name = []           # modules that did `from ._lib import ptr`
        for name, mod in list(sys.modules.items()):
            if name.startswith("umr_amd.") and getattr(mod, "ptr", None) is orig_ptr:
                mod.ptr = ptr
                self._by_name.append(mod)
        self._orig_ptr = orig_ptr
        if not emulated_product._ops_registered:
            for op in (ops.soft_rasterize_op, ops.soft_rasterize_backward_op, ops.silhouette_op, ops.silhouette_backward_op,
                       ops.soft_rasterize_alpha_geometry_op, ops.soft_rasterize_alpha_geometry_backward_op):
                op.register_kernel("cpu")(op._init_fn)
            from umr_amd import ops_losses          # the geometry / loss operators: the same implementations for host tensors
            for name, impl in ops_losses.IMPLS:
                ops_losses._LIB.impl(name, impl, "CPU")
            emulated_product._ops_registered = True
        return self

    def __exit__(self, *exc):
        import torch
        from umr_amd import _lib
        _lib._lib, _lib.ptr, _lib.stream_ptr, _lib.on_device, torch.cuda.synchronize = self._saved
        for mod in self._by_name:
            mod.ptr = self._orig_ptr
        return False


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
             grad_flags=0, tex_group=1, L=None, workspace=None, **modes):
    """umr_raster_backward on host arrays -> (grad_faces | None, grad_textures | None).  workspace: the 'ws' of the forward call of
    the same faces (for UMR_BWD_REUSE_WORKSPACE); default a fresh one."""
    L = L or lib()
    faces = np.ascontiguousarray(faces, np.float32).reshape(faces.shape[0], faces.shape[1], 9)
    N, F = faces.shape[:2]
    IS = int(image_size)
    tex = None if textures is None else np.ascontiguousarray(textures, np.float32)
    TS = 1 if tex is None else tex.shape[2]
    sc = None if soft_colors is None else np.ascontiguousarray(soft_colors, np.float32)     # (None: UMR_BWD_PACKED_STATE)
    ag = None if aggrs_info is None else np.ascontiguousarray(aggrs_info, np.float32)
    g = np.ascontiguousarray(grad_soft_colors, np.float32)
    gf = np.zeros((N, F, 9), np.float32) if need_gf else None
    gt = np.zeros((N, F, TS, 3), np.float32) if need_gt else None
    wsb = L.umr_raster_workspace_bytes(N, F)
    ws = np.zeros(wsb + 64, np.uint8) if workspace is None else workspace
    scal = _scalars(near, far, eps, sigma_val, dist_eps_log, gamma_val, func_id_rgb, double_side, **modes)
    fl = int(grad_flags) | ((tex_group & 0xffff) << 8 if tex_group > 1 else 0)
    rc = L.umr_raster_backward(_p(faces), _p(tex), _p(sc), None, _p(ag), _p(gf), _p(gt), _p(g), fl, int(need_gf), int(need_gt),
                               N, F, TS, IS, *scal, _p(ws), wsb, None)
    if rc != 0:
        raise RuntimeError("umr_raster_backward (host emulation) rc=%d" % rc)
    return gf, gt


def pack_state(ssum, smax, alpha):
    """The packed saved state of UMR_RASTER_PACKED_STATE (include/umr_hip.h) built from the planes [N,IS,IS] a planar forward -- or
    the oracle -- wrote: per 4x4 tile a record of 64 floats = [1 / sum][maximum][alpha][per 2x2 quad: smallest maximum, NaN if one
    of its pixels is NaN][per quad: 1.0 iff all four alphas == 1.0][8 unused].  An independent restatement of the layout, in numpy."""
    ssum, smax, alpha = (np.asarray(a, np.float32) for a in (ssum, smax, alpha))
    N, IS = ssum.shape[0], ssum.shape[1]
    T = IS // 4
    tiles = lambda p: p.reshape(N, T, 4, T, 4).transpose(0, 1, 3, 2, 4).reshape(N, T, T, 16)
    quads = lambda p: p.reshape(N, T, 2, 2, T, 2, 2).transpose(0, 1, 4, 2, 5, 3, 6).reshape(N, T, T, 4, 4)
    rec = np.zeros((N, T, T, 64), np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        rec[..., 0:16] = tiles(np.float32(1.0) / ssum)
    rec[..., 16:32] = tiles(smax)
    rec[..., 32:48] = tiles(alpha)
    q = quads(smax)
    with np.errstate(invalid="ignore"):
        rec[..., 48:52] = np.where(np.isnan(q).any(-1), np.float32(np.nan), np.nanmin(np.where(np.isnan(q), np.float32(np.inf), q), -1))
    rec[..., 52:56] = (quads(alpha) == 1).all(-1).astype(np.float32)
    return rec.reshape(N, -1)


def stats(L=None):
    L = L or lib()
    a, b, c = ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
    L.umr_host_emu_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    return dict(workgroups=a.value, lane_switches=b.value, wave_operations=c.value)
