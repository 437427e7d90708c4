"""Seeded synthetic inputs of the CUB-shaped workload: the dataset, SCOPS maps and
pretrained weights of the reference are not distributed, so benchmarks and step-level tests use these."""
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


def make_s2_inputs(B, K=8, image_size=256, subdivide=3, tex_size=6, seed=0, device="cpu"):
    """-> (template_verts, faces, outputs, batch, extras) for RenderCompareS2; extras holds the synthetic stand-ins
    for the SCOPS template: part vertex ids, the part-label UV image and the UV sampler."""
    from .model import compute_uvsampler
    tv, faces, outputs, batch = make_s1_inputs(B, image_size, subdivide, tex_size, seed, device)
    g = torch.Generator().manual_seed(seed + 7)
    dev = torch.device(device)
    s = 0.6 + 0.3 * torch.rand(B, K, 1, generator=g)
    t = -0.1 + 0.2 * torch.rand(B, K, 2, generator=g)
    q = torch.randn(B, K, 4, generator=g)
    q = q / q.norm(dim=2, keepdim=True)
    outputs["cam_hypotheses"] = torch.cat([s, t, q], 2).to(dev).requires_grad_(True)
    outputs["cam_probs"] = torch.softmax(torch.randn(B, K, generator=g), 1).to(dev).requires_grad_(True)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, image_size), torch.linspace(-1, 1, image_size), indexing="ij")
    parts = torch.stack([torch.sin(3 * xx + i) * torch.cos(2 * yy - i) for i in range(5)])[None].repeat(B, 1, 1, 1)
    batch["part_segs"] = (parts + 0.3 * torch.randn(parts.shape, generator=g)).to(dev)
    batch["random_imgs"] = (batch["imgs"] * batch["masks"].unsqueeze(1)).roll(1, 0)
    for name, n in (("head", 10), ("belly", 30), ("back", 10), ("neck", 30)):     # data/base.py:67-68
        batch[name + "_points"] = (torch.rand(B, n, 2, generator=g) * 2 - 1).to(dev)
    V = tv.shape[0]
    perm = torch.randperm(V, generator=g)
    c = [int(V * f) for f in (0.06, 0.22, 0.27, 0.41)]
    ids = dict(head=perm[:c[0]].numpy(), belly=perm[c[0]:c[1]].numpy(), neck=perm[c[1]:c[2]].numpy(),
               back=perm[c[2]:c[3]].numpy())
    uv_img = torch.randint(0, 5, (1, 1, 128, 256), generator=g).float()
    uv_sampler = torch.from_numpy(compute_uvsampler(tv.numpy().astype("float64"), faces.numpy(), tex_size)).float()
    uv_sampler = uv_sampler.view(1, faces.shape[0], tex_size * tex_size, 2).to(dev)
    return tv, faces, outputs, batch, dict(part_vertex_ids=ids, uv_img=uv_img, uv_sampler=uv_sampler)
