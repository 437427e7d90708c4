"""CPU count behind DESIGN 4.2: lane use of the face-major backward by sub-tile shape.  For every face of the bench scene the
contributing pixels (inside, or within the threshold distance) under the dilated bbox; a wave visit takes 64 / (w*h) sub-tiles
of w x h pixels of ONE face.  Prints visits per mesh and the share of lanes holding a contributing pixel (exact need, i.e. an
ideal sub-tile cull).  usage: sim_subtile.py [scale_lo scale_hi] [subdiv IS]"""
import sys
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_ref as TR
from tests.helpers import scene

lo, hi = (float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (0.6, 0.9)
sub, IS = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3, 512)
B = 2
verts, faces, cams, _ = scene(B, sub, seed=0, scale=(lo, hi))
proj = TR.orthographic_proj_withz(verts, cams, 5.) * torch.tensor([1., -1., 1.])
fv = TR.face_vertices(TR.look_at_ortho(proj), faces).numpy().astype(np.float64)
thr2 = np.log(1 / 1e-10 - 1) * 1e-5
thr = np.sqrt(thr2)
xs = (2 * np.arange(IS) + 1 - IS) / IS


def seg_d2(px, py, ax, ay, bx, by):
    ex, ey = bx - ax, by - ay
    l2 = ex * ex + ey * ey + 1e-30
    t = np.clip(((px - ax) * ex + (py - ay) * ey) / l2, 0, 1)
    dx = ax + t * ex - px; dy = ay + t * ey - py
    return dx * dx + dy * dy


shapes = [(8, 8), (4, 4), (4, 2), (2, 2), (8, 1), (4, 1)]
visits = {s: 0 for s in shapes}; blocks = {s: 0 for s in shapes}
pix = 0; inside_pix = 0; front = 0
for n in range(B):
    for f in range(fv.shape[1]):
        p = fv[n, f]; x = p[:, 0]; y = p[:, 1]
        xlo, xhi, ylo, yhi = x.min() - thr, x.max() + thr, y.min() - thr, y.max() + thr
        i0 = max(int(np.floor((xlo * IS + IS - 1) / 2)), 0); i1 = min(int(np.ceil((xhi * IS + IS - 1) / 2)), IS - 1)
        j0 = max(int(np.floor((ylo * IS + IS - 1) / 2)), 0); j1 = min(int(np.ceil((yhi * IS + IS - 1) / 2)), IS - 1)
        if i0 > i1 or j0 > j1:
            continue
        r0, r1 = IS - 1 - j1, IS - 1 - j0
        rows = np.arange(r0, r1 + 1); cols = np.arange(i0, i1 + 1)
        px, py = np.meshgrid(xs[cols], xs[IS - 1 - rows])
        d2 = np.minimum(np.minimum(seg_d2(px, py, x[0], y[0], x[1], y[1]), seg_d2(px, py, x[1], y[1], x[2], y[2])), seg_d2(px, py, x[2], y[2], x[0], y[0]))
        cr = lambda a, b: (x[b] - x[a]) * (py - y[a]) - (y[b] - y[a]) * (px - x[a])
        c0, c1, c2 = cr(0, 1), cr(1, 2), cr(2, 0)
        ins = ((c0 > 0) & (c1 > 0) & (c2 > 0)) | ((c0 < 0) & (c1 < 0) & (c2 < 0))
        need = ins | (d2 < thr2)
        pix += need.sum(); inside_pix += ins.sum()
        rr, cc = np.nonzero(need)
        rr = rr + r0; cc = cc + i0
        for (w, h) in shapes:
            nb = len(set(zip((rr // h).tolist(), (cc // w).tolist())))
            per = 64 // (w * h)
            blocks[(w, h)] += nb
            visits[(w, h)] += -(-nb // per)
print("scene scale %.2f-%.2f, subdiv %d, IS %d: contributing pairs per mesh %.0f (inside %.0f)" % (lo, hi, sub, IS, pix / B, inside_pix / B))
for s in shapes:
    print("  sub-tile %dx%d: %7.0f blocks, %6.0f visits per mesh, lane use %.3f (blocks alone %.3f)" % (
        s[0], s[1], blocks[s] / B, visits[s] / B, pix / (visits[s] * 64.0), pix / (blocks[s] * s[0] * s[1])))
