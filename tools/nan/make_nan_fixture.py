"""From the tripwire captures of tools/nan/nan_hunt.sh (gpurun_out/nan/repro_<pid>.pt: geometry of the step whose raster backward wrote
the first non-finite value, + the offending (view, face)) to tests/golden/nan_cfg4_faces.npz: per case the projected faces of that
view whose dilated bounding box meets the offending face's window, the offender's index among them, image size.
usage: make_nan_fixture.py repro_a.pt repro_b.pt ..."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "r4"))
from replay_nan import view_faces  # noqa: E402

out = {}
thr = float(np.sqrt(np.float32(math.log(1e10 - 1.) * 1e-5)))
for c, fn in enumerate(sys.argv[1:]):
    d = torch.load(fn, weights_only=False)
    info = d["where"]
    big, n, f = info >> 23, (info >> 16) & 127, info & 0xffff
    rec = dict(d["ring"])[d["first_bad_step"] - 1]
    fv, b = view_faces(d, rec, n, big)
    x, y = fv[:, 0::3], fv[:, 1::3]
    pad = thr + 6.0 / 1024
    lo = np.array([x[f].min() - pad, y[f].min() - pad]); hi = np.array([x[f].max() + pad, y[f].max() + pad])
    near = (x.max(1) + thr >= lo[0]) & (x.min(1) - thr <= hi[0]) & (y.max(1) + thr >= lo[1]) & (y.min(1) - thr <= hi[1])
    idx = np.nonzero(near)[0]
    out["faces_%d" % c] = fv[idx]
    out["offender_%d" % c] = np.int32(np.nonzero(idx == f)[0][0])
    out["meta_%d" % c] = np.array([d["site"], d["first_bad_step"] - 1, b, n, f, 1024], np.int32)   # site, step, image, view, face, IS
    print(fn, "-> %d faces around face %d of view %d" % (len(idx), f, n))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "nan_cfg4_faces.npz"), **out)
