"""CPU: replay the face a tripwire run blamed (tools/nan/bench_trap.py -> gpurun_out/nan/repro_<pid>.pt) through the arithmetic of
both raster directions, compiled for the host from the kernel source (tests/host_kernel/pair_host.cpp::host_replay_face): the
forward's saved soft-max state at every pixel of the face's window and the weight the backward gives the face there.
usage: replay_nan.py repro.pt [noise_scale=0] [exact_edges=0] [thin_h=0]      (defaults: round 3's early settings)"""
import ctypes
import math
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import torch_ref as TR  # noqa: E402

HK = os.path.join(ROOT, "tests", "host_kernel")


def pair_lib(thin_h):
    so = "/tmp/libpair_host_replay_%s.so" % ("thin" if thin_h else "nothin")
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-function",
                           "-Wno-unknown-attributes"] + ([] if thin_h else ["-DTHIN_FACE_H=0.f"]) + [os.path.join(HK, "pair_host.cpp"), "-o", so])
    h = ctypes.CDLL(so)
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    h.host_replay_face.argtypes = [P, I, I, I, F, F, F, F, F, F, F, F, F, P, I]
    h.host_replay_face.restype = I
    return h


def view_faces(d, rec, n, big):
    verts, faces = rec["pred_vs"], d["faces"].long()
    if big:
        K = rec["cam_hypotheses"].shape[1]
        b, cam = n // K, rec["cam_hypotheses"][n // K, n % K]
    else:
        b, cam = n, rec["cam"][n]
    proj = TR.orthographic_proj_withz(verts[b:b + 1], cam[None], 5.) * torch.tensor([1., -1., 1.])
    return TR.face_vertices(TR.look_at_ortho(proj), faces[None]).reshape(-1, 9).numpy().astype(np.float32), b


def main():
    d = torch.load(sys.argv[1], weights_only=False)
    noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    exact = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    thin = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    info = d["where"]
    big, n, f = info >> 23, (info >> 16) & 127, info & 0xffff
    step = d["first_bad_step"] - 1
    rec = dict(d["ring"])[step]
    fv, b = view_faces(d, rec, n, big)
    sigma, gamma, IS = 1e-5, 1e-4, 1024
    threshold = float(math.log(1e10 - 1.) * sigma)
    thr = float(np.sqrt(np.float32(threshold)))
    h = pair_lib(thin)
    cap = 200000
    out = np.zeros((cap, 8), np.float32)
    npix = h.host_replay_face(fv.ctypes.data_as(ctypes.c_void_p), fv.shape[0], f, IS, thr, threshold, -1.0 / sigma, gamma, 1.0, 100.0, 1e-3,
                              noise, (20.0 * sigma if exact else 0.0), out.ctypes.data_as(ctypes.c_void_p), cap)
    o = out[:npix]
    p = fv[f].reshape(3, 3)
    e = [np.linalg.norm(p[(k + 1) % 3, :2] - p[k, :2]) for k in range(3)]
    area2 = abs((p[1, 0] - p[0, 0]) * (p[2, 1] - p[0, 1]) - (p[1, 1] - p[0, 1]) * (p[2, 0] - p[0, 0]))
    print("site 0x%x step %d image %d view %d face %d | settings: cull noise x%g, exact_edges %d, thin route %d" % (d["site"], step, b, n, f, noise, exact, thin))
    print("face corners (x, y, z):", np.array2string(p, precision=6).replace("\n", " "))
    print("edge lengths %s px, smallest height %.4g px" % (np.round(np.array(e) * IS / 2, 3), area2 / max(e) * IS / 2))
    code = o[:, 7].astype(int)
    fwd, bwd = (code & 1) > 0, (code & 2) > 0
    print("window %d pixels: forward included the face at %d, backward includes it at %d, backward-only %d, forward-only %d"
          % (npix, fwd.sum(), bwd.sum(), (bwd & ~fwd).sum(), (fwd & ~bwd).sum()))
    bad = bwd & ~np.isfinite(o[:, 6])
    print("non-finite backward weights: %d; largest finite weight %.4g" % (bad.sum(), np.nanmax(np.where(np.isfinite(o[:, 6]), o[:, 6], 0))))
    for r in o[bad | (bwd & ~fwd)][:8]:
        print("   pixel (%d, %d): D %.4g zn %.6f | saved max %.6f sum %.4g -> (zn - max) / gamma = %.1f, ps %s, forward had the face: %s"
              % (r[0], r[1], r[2], r[3], r[4], r[5], (r[3] - r[4]) / gamma, r[6], bool(int(r[7]) & 1)))


if __name__ == "__main__":
    main()
