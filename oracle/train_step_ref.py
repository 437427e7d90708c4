"""CPU restatement of the train_s1 render-and-compare sequence (experiments/train_s1.py:177-265) on the
oracle (C raster + torch CPU losses).  TEST INFRASTRUCTURE: parity check of the whole step and the timed
`cpu_baseline` leg of bench.py.  Mirrors umr_amd/train_step.py term by term."""
import math

import torch
import torch.nn.functional as F

from . import torch_ref as TR


def rotate_cam_y(cam, angle_deg):
    """nnutils/geom_utils.py:167-193 with axis [0,1,0]: R_new = Rot_y(angle) R(q), returned as the w>=0
    unit quaternion (utils/transformations.py quaternion_from_matrix(isprecise=True) convention)."""
    half = angle_deg.to(cam.dtype) * (math.pi / 360.0)
    rw, ry = torch.cos(half), torch.sin(half)
    rot = torch.stack([rw, torch.zeros_like(rw), ry, torch.zeros_like(rw)], 1)
    q = TR.hamilton_product(rot, cam[:, 3:7])
    q = q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)
    q = torch.where(q[:, :1] < 0, -q, q)
    return torch.cat([cam[:, :3], q], 1)


class RenderCompareS1Ref:
    def __init__(self, template_verts, faces, image_size=256, weights=None, n_threads=1, backend="port"):
        from umr_amd.train_step import S1Weights  # plain constants (reference flag defaults)
        self.w = weights or S1Weights()
        self.faces = faces.long()
        mk = lambda kind: TR.SoftRenderer(image_size, kind, backend=backend, n_threads=n_threads)
        self.renderer, self.dis_renderer, self.hard_renderer, self.tex_renderer = mk("softmax"), mk("softmax"), mk("hard"), mk("softmax")
        self.tex_renderer.ambient_light_only()
        self.lap = TR.LaplacianLoss(template_verts, faces)
        self.flat = TR.FlattenLoss(faces)

    def __call__(self, outputs, batch):
        w = self.w
        pred_vs, delta_v, proj_cam, tex_flow = outputs["pred_vs"], outputs["delta_v"], outputs["cam"], outputs["tex_flow"]
        imgs, masks, dts = batch["imgs"], batch["masks"], batch["dts_barrier"]
        B = pred_vs.shape[0]
        faces = self.faces[None].expand(B, -1, -1)
        t = {}
        pred_seen, _, _ = self.renderer(pred_vs, faces, proj_cam)
        mask_pred_seen = pred_seen[:, 3]
        t["mask"] = TR.neg_iou_loss(mask_pred_seen, masks)
        t["triangle"] = self.lap(pred_vs).mean()
        t["flatten"] = self.flat(pred_vs).mean()
        t["deform"] = TR.deform_l2reg(delta_v)
        t["ori"] = TR.sym_reg(pred_vs)
        tex = TR.sample_textures(tex_flow, imgs).contiguous()
        bs, fs = tex.shape[:2]
        tex = tex.view(bs, fs, -1, 3)
        rgba, p2f, _ = self.tex_renderer(pred_vs.detach(), faces, proj_cam.detach(), tex)
        t["tex"] = TR.texture_loss_masks(rgba[:, :3], imgs, masks, mask_pred_seen)
        t["tex_dt"] = TR.texture_dt_loss(tex_flow, dts)
        _, _, aggr = self.hard_renderer(pred_vs.detach(), faces, proj_cam.detach())
        t["tex_cycle"], _ = TR.tex_cycle(tex_flow, p2f.detach(), aggr[:, 1].reshape(bs, -1).detach())
        pred_unseen, _, _ = self.dis_renderer(pred_vs, faces, rotate_cam_y(proj_cam.detach(), batch["gan_angles"]))
        t["gan"] = pred_unseen[:, 3].mean()
        total = t["mask"] * w.mask_loss_wt + t["triangle"] * w.triangle_reg_wt + t["flatten"] * w.flatten_reg_wt \
            + t["ori"] * w.ori_reg_wt + t["deform"] * w.deform_reg_wt + t["tex"] * w.tex_loss_wt \
            + t["tex_dt"] * w.tex_dt_loss_wt + t["tex_cycle"] * w.tex_cycle_loss_wt + t["gan"] * w.gan_loss_wt
        return total, t
