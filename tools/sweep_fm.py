"""GPU box: raster kernel timings (library-owned HIP events around the main kernels, umr_profile_*) at the bench's
launch sizes for a compile-time variant of the library (built by tools/build_variant.py <tag> -DFLAG...; UMR_LIB_VARIANT=<tag>).
One JSON line tagged argv[1]."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import bench, bench_alpha  # noqa: E402
from umr_amd import _lib  # noqa: E402

if os.environ.get("UMR_LIB_VARIANT"):   # a library built by tools/build_variant.py
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", os.environ["UMR_LIB_VARIANT"], "libumr_hip.so")


def timed(fn, *a, ids=(0, 1), **k):
    fn(*a, **dict(k, iters=2))                  # module load, allocator warm-up
    _lib.profile_enable(True)
    for i in range(4):
        _lib.profile_collect(i)                  # reset
    fn(*a, **k)
    (fms, fn_, _), (bms, bn, _) = _lib.profile_collect(ids[0]), _lib.profile_collect(ids[1])
    _lib.profile_enable(False)
    return round(fms * 1e3 / max(fn_, 1), 1), round(bms * 1e3 / max(bn, 1), 1)   # us per launch: fwd, bwd


tag = sys.argv[1] if len(sys.argv) > 1 else "default"
if os.environ.get("UMR_SB") == "0":
    _lib.debug_set("superblock_bins", 0)
    tag += " [no super-block bins]"
if os.environ.get("UMR_FO") == "0":
    _lib.debug_set("face_order", 0)
    tag += " [index-order backward]"
if os.environ.get("UMR_XCD"):
    _lib.debug_set("xcd_remap", int(os.environ["UMR_XCD"]))
    tag += " [xcd_remap %s]" % os.environ["UMR_XCD"]
if os.environ.get("UMR_FOG"):
    _lib.debug_set("face_order_group", int(os.environ["UMR_FOG"]))
    tag += " [order group %s]" % os.environ["UMR_FOG"]
if os.environ.get("UMR_THIN"):     # thin-face threshold of k_face_setup in 1e-6 screen units (0 = off, 1000000000 = every face)
    _lib.debug_set("thin_face_h_1e6", int(os.environ["UMR_THIN"]))
    tag += " [thin_face_h %se-6]" % os.environ["UMR_THIN"]
if os.environ.get("UMR_EXACT"):    # per-lane doubt criterion of eval_pair (umr_debug_set("exact_edges"))
    _lib.debug_set("exact_edges", int(os.environ["UMR_EXACT"]))
    tag += " [exact_edges %s]" % os.environ["UMR_EXACT"]
out = {"tag": tag}
out["n16_ts36_texonly_pooled"] = timed(bench, 16, 3, 512, 36, pool=True, need_p2f=False, need_gf=False, iters=20)
out["n16_ts36_p2f"] = timed(bench, 16, 3, 512, 36, iters=20)
out["n16_ts1"] = timed(bench, 16, 3, 512, 1, iters=20)
out["n128_ts36_texonly_pooled"] = timed(bench, 128, 3, 512, 36, pool=True, need_p2f=False, need_gf=False)
out["n128_ts36"] = timed(bench, 128, 3, 512, 36)
out["n128_ts1"] = timed(bench, 128, 3, 512, 1)
if os.environ.get("UMR_CFG4", "1") != "0":      # BASELINE config 4's raster shape: 5120 faces at IS = 1024
    out["n32_f5120_is1024_ts36"] = timed(bench, 32, 4, 1024, 36, iters=5)
    out["n32_f5120_is1024_ts36_texonly_pooled"] = timed(bench, 32, 4, 1024, 36, pool=True, need_p2f=False, need_gf=False, iters=5)
    out["alpha_n32_f5120_is1024"] = timed(bench_alpha, 32, 4, 1024, ids=(2, 3), iters=5)
out["alpha_n16"] = timed(bench_alpha, 16, 3, 512, ids=(2, 3))
out["alpha_n128"] = timed(bench_alpha, 128, 3, 512, ids=(2, 3))
print(json.dumps(out), flush=True)
