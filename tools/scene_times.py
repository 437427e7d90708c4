"""GPU box: the raster launches of one train_s1 step on FROZEN captures of the training step's own geometry (bench.py --capture-scene ->
profiles/scenes/*.npz), library-owned HIP events, us per launch -- one JSON line per scene.
usage: scene_times.py [--real] [--iters K] scene.npz ...     (no files: every profiles/scenes/*.npz + the SURVEY 8d scene)
  --real   also time with the capture's own texels / upstream gradient where the file holds them (the committed scenes hold the
           geometry only: values steer no branch; profiles/r06_scene_capture.jsonl is the comparison)
env UMR_LIB_FILE=<lib.so>  UMR_DEBUG_SET=key=v,..."""
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
    argv = sys.argv[1:]
    real = "--real" in argv
    iters = int(argv[argv.index("--iters") + 1]) if "--iters" in argv else 20
    files = [a for a in argv if a.endswith(".npz")]
    dev = torch.device("cuda:0")
    scenes = [(os.path.basename(f)[:-4], f) for f in files] or bench.frozen_scenes()
    base = {"lib": os.path.basename(_lib.LIB_PATH), "build": _lib.build_id()[:12], "set": os.environ.get("UMR_DEBUG_SET", "")}
    for name, path in scenes:
        for rv in ([False, True] if real else [False]):
            r = bench.frozen_scene_kernel_times(dev, path, iters, real_values=rv)
            r = {k: v for k, v in r.items() if not k.startswith("_")}
            print(json.dumps(dict(base, scene=name, values="captured" if rv else "seeded noise", us_per_launch=r)), flush=True)
    if not files:
        r = bench.fixed_scene_kernel_times(dev, iters)
        r = {k: v for k, v in r.items() if not k.startswith("_")}
        print(json.dumps(dict(base, scene="survey_8d", values="seeded noise", us_per_launch=r)), flush=True)
