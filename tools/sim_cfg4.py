"""CPU count behind HISTORY.md section 4.1 (r3): raster work per mesh at BASELINE configs[1] (512^2 render, 1280 faces) and
configs[3] (1024^2 render, 5120 faces) -- (pixel, face) pairs under the sigma-dilated bounding boxes and 8x8 tiles that hold at
least one contributing pixel of a face (= the visits a perfect cull would make)."""
import sys
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_ref as TR
from umr_amd.synthetic import make_s1_inputs


def count(H, subdivide, B=2, T=8):
    IS = 2 * H
    tv, faces, out, batch = make_s1_inputs(B, H, subdivide, seed=100, device='cpu')
    verts, cams = out['pred_vs'].detach(), out['cam'].detach()
    proj = TR.orthographic_proj_withz(verts, cams, 5.) * torch.tensor([1., -1., 1.])
    fv = TR.face_vertices(TR.look_at_ortho(proj), faces[None].expand(B, -1, -1)).numpy().astype(np.float64)
    thr2 = np.log(1 / 1e-10 - 1) * 1e-5
    thr = np.sqrt(thr2)
    xs = (2 * np.arange(IS) + 1 - IS) / IS

    def seg_d2(px, py, ax, ay, bx, by):
        ex, ey = bx - ax, by - ay
        t = np.clip(((px - ax) * ex + (py - ay) * ey) / (ex * ex + ey * ey + 1e-30), 0, 1)
        return (ax + t * ex - px) ** 2 + (ay + t * ey - py) ** 2

    pairs = visits = contrib = 0
    for n in range(B):
        for f in range(fv.shape[1]):
            x, y = fv[n, f, :, 0], fv[n, f, :, 1]
            i0 = max(int(np.ceil(((x.min() - thr) * IS + IS - 1) / 2)), 0)
            i1 = min(int(np.floor(((x.max() + thr) * IS + IS - 1) / 2)), IS - 1)
            j0 = max(int(np.ceil(((y.min() - thr) * IS + IS - 1) / 2)), 0)
            j1 = min(int(np.floor(((y.max() + thr) * IS + IS - 1) / 2)), IS - 1)
            if i0 > i1 or j0 > j1:
                continue
            pairs += (i1 - i0 + 1) * (j1 - j0 + 1)
            px, py = np.meshgrid(xs[i0:i1 + 1], xs[j0:j1 + 1])
            d2 = np.minimum(np.minimum(seg_d2(px, py, x[0], y[0], x[1], y[1]), seg_d2(px, py, x[1], y[1], x[2], y[2])),
                            seg_d2(px, py, x[2], y[2], x[0], y[0]))
            cr = lambda a, b: (x[b] - x[a]) * (py - y[a]) - (y[b] - y[a]) * (px - x[a])
            c0, c1, c2 = cr(0, 1), cr(1, 2), cr(2, 0)
            live = ((c0 > 0) & (c1 > 0) & (c2 > 0)) | ((c0 < 0) & (c1 < 0) & (c2 < 0)) | (d2 < thr2)
            contrib += int(live.sum())
            jj, ii = np.nonzero(live)
            rows = (IS - 1 - (jj + j0)) // T
            cols = (ii + i0) // T
            visits += len(set((rows * 4096 + cols).tolist()))
    return pairs / B, visits / B, contrib / B


if __name__ == "__main__":
    a = count(256, 3)
    b = count(512, 4)
    for name, r in (("configs[1] 512^2 x 1280", a), ("configs[3] 1024^2 x 5120", b)):
        print("%s: pairs under dilated boxes %.0f, 8x8 visits with a contributing pixel %.0f, contributing pairs %.0f" % ((name,) + r))
    print("ratios: pairs %.2f  visits %.2f  contributing pairs %.2f" % tuple(b[i] / a[i] for i in range(3)))
