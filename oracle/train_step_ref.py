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
    def __init__(self, template_verts, faces, image_size=256, weights=None, n_threads=1, backend="port",
                 texture_loss=None, epoch=0, discriminator=None):
        """texture_loss: callable(img_pred, img_gt, mask_gt, mask_pred) (TR.PerceptualTextureLoss in the reference,
        train_s1.py:150); None = the masked L1 of loss_utils.py:103-116.  epoch: train_s1.py:250-255 gates the
        symmetry term (epoch < stop_ori_epoch) and the deformation term (epoch > update_template_freq)."""
        from umr_amd.train_step import S1Weights  # plain constants (reference flag defaults)
        self.w = weights or S1Weights()
        self.texture_loss = texture_loss or TR.texture_loss_masks
        self.epoch = epoch
        self.discriminator = discriminator           # train_s1.py:88-90, :243; None = mean of the unseen-view mask instead
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
        t["tex"] = self.texture_loss(rgba[:, :3], imgs, masks, mask_pred_seen)
        t["tex_dt"] = TR.texture_dt_loss(tex_flow, dts)
        _, _, aggr = self.hard_renderer(pred_vs.detach(), faces, proj_cam.detach())
        t["tex_cycle"], _ = TR.tex_cycle(tex_flow, p2f.detach(), aggr[:, 1].reshape(bs, -1).detach())
        pred_unseen, _, _ = self.dis_renderer(pred_vs, faces, rotate_cam_y(proj_cam.detach(), batch["gan_angles"]))
        if self.discriminator is not None:           # train_s1.py:238-244
            pred = torch.cat((pred_seen.detach(), pred_unseen))
            labels = torch.cat((torch.ones(pred_seen.shape[0]), torch.zeros(pred_unseen.shape[0])), dim=0)
            t["gan"] = torch.nn.functional.binary_cross_entropy_with_logits(self.discriminator(pred[:, 3].unsqueeze(1)).squeeze(), labels)
        else:
            t["gan"] = pred_unseen[:, 3].mean()
        total = t["mask"] * w.mask_loss_wt + t["triangle"] * w.triangle_reg_wt + t["flatten"] * w.flatten_reg_wt
        if self.epoch < w.stop_ori_epoch:            # train_s1.py:250-252
            total = total + t["ori"] * w.ori_reg_wt
        if self.epoch > w.update_template_freq:      # train_s1.py:253-255
            total = total + t["deform"] * w.deform_reg_wt
        total = total + t["tex"] * w.tex_loss_wt + t["tex_dt"] * w.tex_dt_loss_wt \
            + t["tex_cycle"] * w.tex_cycle_loss_wt + t["gan"] * w.gan_loss_wt
        return total, t


class RenderCompareS2Ref:
    """experiments/train_s2.py:201-316 on the oracle, term by term as umr_amd.train_step.RenderCompareS2, no
    discriminator network.  texture_loss: callable(img_pred, img_gt, mask_gt, mask_pred, avg=False) -- the reference's
    MultiTextureLoss uses PerceptualTextureLoss (loss_utils.py:291-292); None = masked L1 (loss_utils.py:103-116)."""

    def __init__(self, template_verts, faces, part_vertex_ids, uv_img, uv_sampler, image_size=256, num_hypo_cams=8,
                 weights=None, n_threads=1, backend="port", tex_size=6, texture_loss=None, discriminator=None, num_sym_faces=0):
        from umr_amd.train_step import S2Weights
        self.w = weights or S2Weights()
        self.texture_loss = texture_loss or TR.texture_loss_masks
        self.K, self.image_size = num_hypo_cams, image_size
        self.discriminator = discriminator            # train_s2.py:91-93, :262; None = mean of the unseen-view colours instead
        self.faces = faces.long()
        mk = lambda kind: TR.SoftRenderer(image_size, kind, backend=backend, n_threads=n_threads)
        self.mask_r, self.tex_r, self.hard_r, self.dis_r, self.part_r = mk("softmax"), mk("softmax"), mk("hard"), mk("softmax"), mk("softmax")
        for r in (self.tex_r, self.dis_r, self.part_r):
            r.ambient_light_only()
        self.lap, self.flat = TR.LaplacianLoss(template_verts, faces), TR.FlattenLoss(faces)
        names = ("head", "belly", "neck", "back")
        self.part_ids = [torch.as_tensor(part_vertex_ids[n]).long() for n in names]
        tex = TR.grid_sample(uv_img.float().view(1, 1, 128, 256), uv_sampler)
        tex = tex.view(1, -1, tex.size(2), tex_size, tex_size).permute(0, 2, 3, 4, 1)
        if num_sym_faces:                             # loss_utils.py:347-348: the sampler covers F - n faces, the last n are repeated
            tex = torch.cat([tex, tex[:, -num_sym_faces:]], 1)
        stex = torch.round(tex.reshape(tex.size(1), -1))
        nf, nt = stex.size()
        one_hot = torch.zeros(nf * nt, 5)
        one_hot.scatter_(1, stex.view(-1, 1).long().clamp(0, 4), 1)
        self.stex = one_hot.view(1, nf, nt, 5)

    def __call__(self, outputs, batch):
        w, K, H = self.w, self.K, self.image_size
        pred_vs, delta_v = outputs["pred_vs"], outputs["delta_v"]
        B = pred_vs.shape[0]
        faces = self.faces[None].expand(B, -1, -1)
        imgs, masks = batch["imgs"], batch["masks"]
        proj_cam = outputs["cam"].detach()
        cams_all, probs = outputs["cam_hypotheses"], outputs["cam_probs"]
        t = {}
        t["cam_div"] = -1 * (torch.log(probs + 1E-9) * probs).sum(1).mean()
        t["mask"], mask_all = TR.multi_mask_loss(self.mask_r, pred_vs, faces, cams_all, probs, masks, K, H)
        t["triangle"] = self.lap(pred_vs).mean()
        t["flatten"] = self.flat(pred_vs).mean()
        t["deform"] = TR.deform_l2reg(delta_v)
        tex_flow = outputs["tex_flow"]
        tex = TR.sample_textures(tex_flow, imgs).contiguous()
        bs, fs = tex.shape[:2]
        tex = tex.view(bs, fs, -1, 3)
        rep = lambda x: x.unsqueeze(1).repeat(1, K, *([1] * (x.dim() - 1))).view(-1, *x.shape[1:])
        rgba, _, _ = self.tex_r(rep(pred_vs.detach()), rep(faces), cams_all.detach().view(-1, 7), rep(tex))
        tl = self.texture_loss(rgba[:, :3], rep(imgs), rep(masks), mask_all, avg=False)
        t["tex"] = (tl.view(bs, -1) * probs.detach()).sum(dim=1).mean()
        t["tex_dt"] = TR.texture_dt_loss(tex_flow, batch["dts_barrier"])
        _, p2f, aggr = self.hard_r(pred_vs.detach(), faces, proj_cam)
        t["tex_cycle"], _ = TR.tex_cycle(tex_flow, p2f.detach(), aggr[:, 1].reshape(bs, -1).detach())
        pred_unseen, _, _ = self.dis_r(pred_vs, faces, rotate_cam_y(proj_cam, batch["gan_angles"]), tex.detach())
        if self.discriminator is not None:            # train_s2.py:256-263
            pred = torch.cat((batch["random_imgs"], pred_unseen[:, 0:3]))
            labels = torch.cat((torch.ones(batch["random_imgs"].shape[0]), torch.zeros(pred_vs.shape[0])), dim=0)
            t["gan"] = torch.nn.functional.binary_cross_entropy_with_logits(self.discriminator(pred).squeeze(), labels)
        else:
            t["gan"] = pred_unseen[:, 0:3].mean()
        projs = []
        for i in range(1, 5):
            stex = self.stex[:, :, :, i].unsqueeze(-1).repeat(B, 1, 1, 3)
            pr, _, _ = self.part_r(pred_vs, faces, proj_cam, stex)
            projs.append(torch.mean(pr[:, 0:3], dim=1).unsqueeze(1))
        t["part"] = TR.part_matching_core(projs, batch["part_segs"])
        mean_shape = outputs["mean_shape"][None].expand(B, -1, -1)
        # (head, belly, back, neck) passed into (head, belly, neck, back): train_s2.py:311 vs loss_utils.py:223
        pts = [rep(batch["head_points"]), rep(batch["belly_points"]), rep(batch["back_points"]), rep(batch["neck_points"])]
        ms, cf = rep(mean_shape), cams_all.reshape(-1, 7)
        coords = torch.cat([ms[:, ids, :] for ids in self.part_ids], dim=1)
        v2d = TR.orthographic_proj_withz(coords, cf)[:, :, :2]
        import numpy as np
        nums = np.cumsum([0] + [len(i) for i in self.part_ids])
        cds = []
        for i, wt in enumerate((1, 1, 0, 0)):
            d1, _, _, _ = TR.dist_chamfer(v2d[:, nums[i]:nums[i + 1], :], pts[i])
            cds.append(d1 * wt)
        corr = torch.mean(torch.cat(cds, dim=1), dim=1)
        t["corr"] = (corr.view(B, K) * probs.detach()).sum(dim=1).mean()
        total = t["mask"] * w.mask_loss_wt + t["triangle"] * w.triangle_reg_wt + t["flatten"] * w.flatten_reg_wt \
            + t["deform"] * w.deform_reg_wt + t["tex"] * w.tex_loss_wt + t["tex_dt"] * w.tex_dt_loss_wt \
            + t["tex_cycle"] * w.tex_cycle_loss_wt + t["gan"] * w.gan_loss_wt + t["cam_div"] * w.ent_loss_wt \
            + t["part"] * w.prob_loss_wt + t["corr"] * w.vertex_loss_wt
        return total, t
