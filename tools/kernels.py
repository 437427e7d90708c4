"""GPU box: the raster launches of one train_s1 step on the FIXED SURVEY 8d scene (bench.py's fixed_scene_kernel_times: library-owned
HIP events, us per launch), for the product library or an experimental build, with umr_debug_set keys -- one JSON line per run.
usage: kernels.py [iters]     env UMR_LIB_FILE=<lib.so>  UMR_DEBUG_SET=key=v,key=v  UMR_SCALE="0.95 1.05" (camera scale range;
default 0.6 0.9; bench.py's networks start near 1.0: the mesh fills the frame)  UMR_N=128 (views per launch; default 16)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umr_amd import _lib  # noqa: E402

if os.environ.get("UMR_LIB_FILE"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UMR_LIB_FILE"])
import bench  # noqa: E402

if __name__ == "__main__":
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for kv in os.environ.get("UMR_DEBUG_SET", "").split(","):
        if "=" in kv:
            _lib.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
    sc = tuple(float(x) for x in os.environ.get("UMR_SCALE", "0.6 0.9").split())
    n = int(os.environ.get("UMR_N", "16"))          # views per launch (16: a train_s1 step; 128: train_s2's hypothesis render)
    r = bench.fixed_scene_kernel_times(torch.device("cuda:0"), it, sc, n)
    r = {k: v for k, v in r.items() if not k.startswith("_")}
    print(json.dumps({"lib": os.path.basename(_lib.LIB_PATH), "build": _lib.build_id()[:12], "set": os.environ.get("UMR_DEBUG_SET", ""),
                      "scale": sc, "N": n, "us_per_launch": r}), flush=True)
