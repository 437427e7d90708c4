import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from conftest import load_golden
import test_gpu_parity as T
for name in T.RASTER:
    g = load_golden(name)
    o = T._raw_raster(g)
    gf = o["grad_faces"].cpu().numpy(); gt = o["grad_textures"].cpu().numpy()
    bad = np.argwhere(~np.isfinite(gf))
    print(name, "nonfinite grad_faces", len(bad), bad[:6].tolist(), "grad_tex nonfinite", (~np.isfinite(gt)).sum(),
          "max|gf-ref|", np.nanmax(np.abs(gf - g["grad_faces"])), "ref max", np.abs(g["grad_faces"]).max())
