"""Seeded synthetic inputs of the CUB-shaped workload: the dataset, SCOPS maps and
pretrained weights of the reference are not distributed, so benchmarks and step-level tests use these."""
import numpy as np
import torch

from .mesh import create_sphere


def template(subdivide=3):
    v, f = create_sphere(subdivide)
    return torch.from_numpy(v).float(), torch.from_numpy(f).long()


def make_s1_inputs(B, image_size=256, subdivide=3, tex_size=6, seed=0, device="cpu"):
    """-> (template_verts, faces, outputs, batch) for RenderCompareS1 (leaves require grad)."""
    g = torch.Generator().manual_seed(seed)
    tv, faces = template(subdivide)
    V, F = tv.shape[0], faces.shape[0]
    delta_v = 0.05 * torch.randn(B, V, 3, generator=g)
    s = 0.6 + 0.3 * torch.rand(B, 1, generator=g)
    t = -0.1 + 0.2 * torch.rand(B, 2, generator=g)
    q = torch.randn(B, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    cam = torch.cat([s, t, q], 1)
    tex_flow = torch.rand(B, F, tex_size, tex_size, 2, generator=g) * 2 - 1
    imgs = torch.rand(B, 3, image_size, image_size, generator=g)
    # GT masks: discs of random radius (stand-in for hard renders of a second mesh)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, image_size), torch.linspace(-1, 1, image_size), indexing="ij")
    rad = 0.5 + 0.3 * torch.rand(B, 1, 1, generator=g)
    cen = 0.2 * torch.rand(B, 2, 1, 1, generator=g) - 0.1
    masks = (((xx[None] - cen[:, 0]) ** 2 + (yy[None] - cen[:, 1]) ** 2) < rad ** 2).float()
    dts = torch.rand(B, 1, image_size, image_size, generator=g)
    angles = torch.randint(0, 180, (B,), generator=g).float()
    dev = torch.device(device)
    delta_v = delta_v.to(dev).requires_grad_(True)
    outputs = dict(delta_v=delta_v, pred_vs=None, cam=cam.to(dev).requires_grad_(True),
                   tex_flow=tex_flow.to(dev).requires_grad_(True))
    outputs["mean_shape"] = tv.to(dev)
    outputs["pred_vs"] = outputs["mean_shape"][None] + delta_v
    batch = dict(imgs=imgs.to(dev), masks=masks.to(dev), dts_barrier=dts.to(dev), gan_angles=angles.to(dev))
    return tv, faces, outputs, batch
