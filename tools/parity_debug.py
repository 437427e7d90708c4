"""Debug aid (GPU box): where does the HIP forward differ from a raster golden?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
import test_gpu_parity as T

name = sys.argv[1] if len(sys.argv) > 1 else "raster_softmax_ts36.npz"
g = load_golden(name)
o = T._raw_raster(g)
sc = o["soft_colors"].cpu().numpy(); ref = g["soft_colors"]
err = np.abs(sc - ref)
for c in range(4):
    print("channel", c, "max err %.3e" % err[:, c].max(), "frac>1e-4 %.5f" % (err[:, c] > 1e-4).mean())
ag = o["aggrs_info"].cpu().numpy(); rag = g["aggrs_info"]
print("ssum rel err max %.3e" % (np.abs(ag[:, 0] - rag[:, 0]) / np.abs(rag[:, 0])).max(), "smax abs err max %.3e" % np.abs(ag[:, 1] - rag[:, 1]).max())
bad = np.argwhere(err > 1e-3)
print("n bad", len(bad))
for b in bad[:12]:
    n, c, y, x = b
    print(b, "got", sc[n, :, y, x], "ref", ref[n, :, y, x], "S", ag[n, 0, y, x], rag[n, 0, y, x], "m", ag[n, 1, y, x], rag[n, 1, y, x])
