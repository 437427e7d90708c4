"""EXPERIMENT harness for tools/exp_quadpack/qp_forward.hip (quadrant-packed silhouette forward).

  python tools/exp_quadpack/check_and_time.py check      CPU: the experiment kernel on the wave64 emulator against the product's
                                                          k_raster_forward<2> (emulated as well): alpha planes bit-identical
  python tools/exp_quadpack/check_and_time.py time       GPU box: builds the experiment with hipcc, checks it against the product
                                                          library on the device, and times both (HIP events, kernel + set-up)
"""
import ctypes
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SCAL = dict(near=1., far=100., sigma_val=1e-5, dist_eps=float(math.log(1e10 - 1.)))
P, I, F_, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t


def proto(L):
    L.umr_exp_qp_workspace_bytes.argtypes, L.umr_exp_qp_workspace_bytes.restype = [I, I], Z
    L.umr_exp_sil_forward_qp.argtypes = [P, P, P, I, I, I, F_, F_, F_, F_, P, Z, P]
    L.umr_exp_sil_forward_qp.restype = I
    return L


def check():
    import host_raster as HR
    from test_raster_library_on_host import _scene_faces, CFG
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    for lds in (0, 1):
        so = "/tmp/libqp_host_%d.so" % lds
        subprocess.check_call([HR.CLANG, "-DQP_LDS_REC=%d" % lds, "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-Wno-unused-function", "-Wno-unknown-attributes", "-Wno-ignored-attributes", "-include",
                               os.path.join(ROOT, "tests", "host_kernel", "wave_emu.h"), "-x", "c++", os.path.join(HERE, "qp_forward.hip"), "-o", so])
        L = proto(ctypes.CDLL(so))
        for IS, subdiv, n in ((64, 1, 2), (96, 2, 2), (100, 2, 1), (128, 3, 2), (256, 3, 1)):
            faces, _ = _scene_faces(n, subdiv, seed=7 + IS)
            ref = HR.forward(faces, None, IS, flags=HR.ALPHA_ONLY, pooled=IS % 2 == 0, **dict(CFG, func_id_rgb=1))
            N, F = faces.shape[:2]
            alpha = np.full((N, IS, IS), np.nan, np.float32)
            pooled = np.full((N, IS // 2, IS // 2), np.nan, np.float32) if IS % 2 == 0 else None
            wsb = L.umr_exp_qp_workspace_bytes(N, F)
            ws = np.zeros(wsb + 64, np.uint8)
            rc = L.umr_exp_sil_forward_qp(p(faces), p(alpha), p(pooled), N, F, IS, SCAL["near"], SCAL["far"], SCAL["sigma_val"], SCAL["dist_eps"],
                                          p(ws), wsb, None)
            assert rc == 0
            same = np.array_equal(alpha, ref["soft_colors"]) and (pooled is None or np.array_equal(pooled, ref["pooled"]))
            print("records from %s: IS %d, %d faces x %d: alpha%s bit-identical to the product kernel: %s"
                  % ("LDS" if lds else "global memory", IS, F, N, " and pooled" if pooled is not None else "", same))
            assert same


def time_gpu():
    import torch
    from umr_amd import _lib
    so = os.path.join(ROOT, "gpurun_out", "libqp.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    out = {}
    VARIANTS = [(0, 6), (0, 8), (1, 5), (1, 6)]     # (records staged in LDS, waves per SIMD asked for)
    for lds, wpe in VARIANTS:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
                               "-ffp-contract=off", "-fno-slp-vectorize", "-DQP_WPE=%d" % wpe, "-DQP_LDS_REC=%d" % lds,
                               os.path.join(HERE, "qp_forward.hip"), "-o", so.replace(".so", "_%d_%d.so" % (lds, wpe))])
    from tests.helpers import scene
    from umr_amd import functional as UF
    dev = torch.device("cuda:0")
    Lp = _lib.lib()
    for tag, N, subdiv, IS in (("n16_is512", 16, 3, 512), ("n32_is512", 32, 3, 512), ("n128_is512", 128, 3, 512), ("n32_f5120_is1024", 32, 4, 1024)):
        verts, faces_i, cams, _ = scene(N, subdiv, seed=100)
        _, fv, _ = UF.ProjectFacesFunction.apply(verts.to(dev), cams.to(dev), faces_i.int().to(dev), 5.0, -2.732, False)
        faces = fv.reshape(N, -1, 9).contiguous()
        Fn = faces.shape[1]
        a_ref = torch.empty(N, IS, IS, device=dev)
        p_ref = torch.empty(N, IS // 2, IS // 2, device=dev)
        wsb = Lp.umr_raster_workspace_bytes(N, Fn)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        st = _lib.stream_ptr(dev)
        ptr = _lib.ptr

        def run_product():
            _lib.check(Lp.umr_raster_forward(ptr(faces), None, None, None, None, None, None, ptr(a_ref), ptr(p_ref), N, Fn, 1, IS, 1., 100., 1e-3,
                                             1e-5, 2, SCAL["dist_eps"], 1e-4, 1, 2, 0, 1, 2 | 1, None, ptr(ws), wsb, st), "product")

        def timed(fn, iters=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / iters
        rec = {"product_us": round(timed(run_product), 1)}
        for lds, wpe in VARIANTS:
            Lx = proto(ctypes.CDLL(so.replace(".so", "_%d_%d.so" % (lds, wpe))))
            a_x = torch.full((N, IS, IS), float("nan"), device=dev)
            p_x = torch.full((N, IS // 2, IS // 2), float("nan"), device=dev)
            wsx = torch.empty(Lx.umr_exp_qp_workspace_bytes(N, Fn), dtype=torch.uint8, device=dev)

            def run_exp():
                rc = Lx.umr_exp_sil_forward_qp(ptr(faces), ptr(a_x), ptr(p_x), N, Fn, IS, 1., 100., 1e-5, SCAL["dist_eps"], ptr(wsx), wsx.numel(), st)
                assert rc == 0
            run_exp(); run_product()
            torch.cuda.synchronize()
            key = "quadpack_%s_wpe%d" % ("lds" if lds else "global", wpe)
            rec[key + "_identical"] = bool(torch.equal(a_x, a_ref) and torch.equal(p_x, p_ref))
            rec[key + "_us"] = round(timed(run_exp), 1)
        out[tag] = rec
        print(tag, json.dumps(rec), flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "quadpack.json"), "w"), indent=1)


if __name__ == "__main__":
    {"check": check, "time": time_gpu}[sys.argv[1]]()
