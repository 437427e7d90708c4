"""GPU box: the collapsed-edge faces of tests/test_gpu_zz_collapsed_edges.py through a given library variant (argv[1], ''
= the product build): prints how many output values are non-finite."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from umr_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", sys.argv[1], "libumr_hip.so")
from umr_amd import functional as UF
from test_gpu_zz_collapsed_edges import collapsed_edge_faces
fv_np, _ = collapsed_edge_faces(64, 64)
fv = torch.from_numpy(fv_np[None]).cuda()
tex = torch.rand(1, 64, 4, 3).cuda()
sc, p2f, aggr = UF.soft_rasterize(fv, tex, 64, [0.1, 0.2, 0.3], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, 'softmax')
al = UF.SilhouetteFunction.apply(fv, 64, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, False)
print("variant %r: non-finite soft_colors %d, alpha %d" % (sys.argv[1] if len(sys.argv) > 1 else "", int((~torch.isfinite(sc)).sum()), int((~torch.isfinite(al)).sum())))
