"""PRICING experiment for a compacted pair cache (HISTORY 13), outside the product: copies umr_amd/csrc to a
scratch directory, patches it and builds umr_amd/lib/exp/libumr_hip_paircache*.so, which tools/gpu_scene_ab.sh times next to the
product library on the frozen scenes.  The patched kernels compute WRONG results -- only their durations mean something:
  * the textured forward stores a 32-byte record {D, sign, dx, dy, b0, b1, w0, w1} per live (pixel, face) pair at a deterministic
    place (16 MB of the workspace's tail per mesh: 32 B x (tile record order of the packed state) + 40 960 B x face, wrapped);
  * the one-pass backward (and every other face-major variant) replaces eval_pair -- ~250 of its ~410 VALU instructions per visit
    -- by two 16-byte loads from that place and treats every lane it hands out as live (84 % are, tools/visit_census.py: the replay
    cost is over-estimated by that much).
  variants: `paircache` = both; `paircache_fwd` = the forward's stores only; `paircache_bwd` = the backward's replay only.
usage: python tools/exp_pair_cache.py        (then: tools/gpu_scene_ab.sh paircache on the GPU box)"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from umr_amd import build as B  # noqa: E402

PC_MB = 16
CORE = [("    int vis_ids_only;   // k_raster_forward<.., VIS>",
         "    char *pc;           // EXPERIMENT: pair-cache records (tail of the workspace)\n    int vis_ids_only;   // k_raster_forward<.., VIS>")]
HIP = [("    return ws_order_offset(N, F, image_size) + ws_order_bytes(N, F);",
        "    return ws_order_offset(N, F, image_size) + ws_order_bytes(N, F) + (size_t)N * (%du << 20);" % PC_MB),
       ("    A.rec = (const float *)((char *)workspace + ws_bbox_bytes(N, F));",
        "    A.rec = (const float *)((char *)workspace + ws_bbox_bytes(N, F));\n"
        "    A.pc = (char *)workspace + ws_order_offset(N, F, image_size) + ws_order_bytes(N, F);")]
PC_ADDR = ("(A.pc + (size_t)%%s * (%du << 20) + (((%%s) + (unsigned)f * 40960u) & ((%du << 20) - 32u)))" % (PC_MB, PC_MB))
FWD = [("                const bool live = eval_pair(p, fc, t.xp, t.yp, A.threshold, A.nis, A.amb_thr, t.valid) & t.valid;\n",
        "                const bool live = eval_pair(p, fc, t.xp, t.yp, A.threshold, A.nis, A.amb_thr, t.valid) & t.valid;\n"
        "                if (RGB == 1 && live) {   // EXPERIMENT: the pair's record, at its place in the packed state's tile order\n"
        "                    const unsigned key = ((unsigned)(t.row >> 2) * (unsigned)(IS >> 2) + (unsigned)(t.xi >> 2)) * 512u + (unsigned)((t.row & 3) * 4 + (t.xi & 3)) * 32u;\n"
        "                    float4 *d = (float4 *)" + PC_ADDR % ("t.n", "key") + ";\n"
        "                    d[0] = make_float4(p.frag, p.sign, p.dx, p.dy); d[1] = make_float4(p.b0, p.b1, p.w0, p.w1);\n"
        "                }\n")]
BWD = [("                    if (!eval_pair(p, fc, xp, yp, c_thr2, c_nis, A.amb_thr)) continue;\n",
        "#if FM_PACKED\n"
        "                    {   // EXPERIMENT: replay the pair from its record instead of evaluating it\n"
        "                        const unsigned key = (pn4 >> 8) * 512u + ((pn4 & 255u) >> 2) * 32u;\n"
        "                        const float4 *d = (const float4 *)" + PC_ADDR % ("n", "key") + ";\n"
        "                        const float4 r0 = d[0], r1 = d[1];\n"
        "                        p.frag = 0.5f + 1e-20f * r0.x; p.sign = 1.f; p.dx = 1e-3f + 1e-20f * r0.z; p.dy = 1e-3f + 1e-20f * r0.w;\n"
        "                        p.b0 = 0.3f + 1e-20f * r1.x; p.b1 = 0.3f + 1e-20f * r1.y; p.b2 = 1.f - p.b0 - p.b1;\n"
        "                        p.w0 = 0.3f + 1e-20f * r1.z; p.w1 = 0.3f + 1e-20f * r1.w + 1e-20f * r0.y; p.w2 = 1.f - p.w0 - p.w1;\n"
        "                    }\n"
        "#else\n"
        "                    if (!eval_pair(p, fc, xp, yp, c_thr2, c_nis, A.amb_thr)) continue;\n"
        "#endif\n")]


def patch(path, subs):
    s = open(path).read()
    for a, b in subs:
        assert s.count(a) >= 1, (path, a[:60])
        s = s.replace(a, b)
    open(path, "w").write(s)


def build(tag, fwd, bwd):
    tmp = tempfile.mkdtemp(prefix="umr_pc_")
    csrc = os.path.join(tmp, "umr_amd", "csrc")
    shutil.copytree(B.CSRC, csrc)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    patch(os.path.join(csrc, "raster_core.h"), CORE)
    patch(os.path.join(csrc, "raster.hip"), HIP)
    if fwd:
        patch(os.path.join(csrc, "raster_forward.h"), FWD)
    if bwd:
        patch(os.path.join(csrc, "raster_backward_fm.h"), BWD)
    out = os.path.join(B.LIBDIR, "exp", "libumr_hip_%s.so" % tag)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([B.HIPCC] + B.FLAGS + ['-DUMR_SRC_HASH="%s"' % tag] + [os.path.join(csrc, s) for s in B.SOURCES] + ["-o", out])
    shutil.rmtree(tmp)
    print(out)


if __name__ == "__main__":
    build("paircache", True, True)
    build("paircache_fwd", True, False)
    build("paircache_bwd", False, True)
