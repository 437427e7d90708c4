"""The render-and-compare part of one UMR training step, as a module.

Mirrors `ShapenetTrainer.forward` of the reference from the point where the network outputs exist
(experiments/train_s1.py:177-265): the op sequence, detaches and loss weights are the reference's;
every render / sampling / reduction on the way is a HIP kernel (umr_amd/csrc).  Per image this is
4 raster forwards (mask :199, texture :217, hard :223, GAN view :235) and 3 raster backwards.
"""

import torch
import torch.nn as nn

from . import geom_utils, loss_utils
from .smr import SoftRenderer


class S1Weights:
    """experiments/train_s1.py:46-56 defaults."""
    mask_loss_wt = 3.0
    gan_loss_wt = 1.0
    triangle_reg_wt = 0.15
    flatten_reg_wt = 0.0004
    deform_reg_wt = 5.0
    ori_reg_wt = 0.4
    tex_loss_wt = 3.0
    tex_dt_loss_wt = 3.0
    tex_cycle_loss_wt = 0.5
    stop_ori_epoch = 3.0          # train_s1.py:53: the symmetry term is on while epoch < stop_ori_epoch
    update_template_freq = 5      # train_s1.py:66: the deformation term is on once epoch > update_template_freq


def rotate_cam_y(cam, angle_deg):
    """geom_utils.rotate_cam (nnutils/geom_utils.py:167-193) for axis=[0,1,0] without the per-sample
    numpy / cv2.Rodrigues / host round trip: new_q = q_y(angle) (x) q, renormalised, w >= 0 (one kernel,
    umr_rotate_cam_y).  cam [B,7] = [s,tx,ty,qw,qx,qy,qz]; angle_deg [B].  Forward only: the reference rotates a
    detached camera at both call sites (train_s1.py:233, train_s2.py:257)."""
    if cam.requires_grad or angle_deg.requires_grad:
        raise RuntimeError("rotate_cam_y is not differentiable; detach the camera (as the reference does)")
    from . import _lib
    c = cam.to(torch.float32).contiguous()
    a = angle_deg.to(torch.float32).contiguous()
    out = torch.empty_like(c)
    _lib.check(_lib.lib().umr_rotate_cam_y(_lib.ptr(c), _lib.ptr(a), _lib.ptr(out), c.shape[0], _lib.stream_ptr(c.device)),
               "umr_rotate_cam_y")
    return out


def gan_labels(module, B, device):
    """[1]*B + [0]*B real/fake targets of the mask discriminator (train_s1.py:240-241), built once per (B, device)."""
    cache = module.__dict__.setdefault("_label_cache", {})
    if (B, device) not in cache:
        cache[(B, device)] = torch.cat((torch.ones(B, device=device), torch.zeros(B, device=device)))
    return cache[(B, device)]


def weighted_total(module, terms, weights, skip=()):
    """sum_k weights[k] * terms[k] as ONE stack, ONE multiply and ONE sum (and their backward kernels) instead of a
    multiply and an add per term in each direction: the step's wall time equals its host enqueue time, so every
    elementwise launch on scalars costs ~10 us of it.  `weights`: list of (name, python float); the weight vector is
    uploaded once per device and cached on `module`.  `skip`: names LEFT OUT of the sum -- the terms the reference's own
    `if` statements leave out (train_s1.py:250-255: the epoch-gated regularisers); a term with a user-set weight of 0 stays in,
    as `0 * loss`, exactly like the reference's unconditional additions (its parameters then get zero gradients, not None)."""
    weights = [(k, v) for k, v in weights if k not in skip]
    if not weights:
        return next(iter(terms.values())).new_zeros(())
    vals = torch.stack([terms[k].reshape(()) for k, _ in weights])
    cache = module.__dict__.setdefault("_wvec_cache", {})
    key = (vals.device, tuple((k, float(v)) for k, v in weights))
    if key not in cache:
        cache[key] = torch.tensor([v for _, v in key[1]], dtype=vals.dtype, device=vals.device)
    return (vals * cache[key]).sum()


class RenderCompareS1(nn.Module):
    """train_s1 render-and-compare: forward(outputs, batch) -> (total_loss, dict of terms).

    outputs: pred_vs [B,V,3], delta_v [B,V',3], cam [B,7], tex_flow [B,F,T,T,2]
    batch:   imgs [B,3,H,H], masks [B,H,H], dts_barrier [B,1,H,H], gan_angles [B] (degrees)
    texture_loss: callable(img_pred, img_gt, mask_gt, mask_pred) -> scalar.  The reference uses
                  PerceptualTextureLoss (train_s1.py:150) -- pass umr_amd.perceptual.PerceptualTextureLoss(device), as
                  model.build_training_step does; None = loss_utils.texture_loss_masks (masked L1, :103-116).
    epoch: current epoch (see set_epoch).
    discriminator: optional nn.Module taking [2B,1,H,H] masks (train_s1.py:243).
    """

    def __init__(self, template_verts, faces, image_size=256, renderer_type="softmax", weights=None,
                 texture_loss=None, discriminator=None, epoch=0, share_mask_render=True):
        super().__init__()
        # True: the mask render (:199) is the alpha channel of the textured render (:217) of the same meshes and cameras
        # (SoftRenderer.forward(detach_rgb_geometry=True)) instead of a render of its own; False: two renders, as written there
        self.share_mask_render = share_mask_render
        self.w = weights or S1Weights()
        self.image_size = image_size
        self.register_buffer("faces", faces.long())
        self.renderer = SoftRenderer(image_size, renderer_type)            # train_s1.py:105
        self.dis_renderer = SoftRenderer(image_size, renderer_type)        # :106 (same settings: shares renderer's launch)
        self.hard_renderer = SoftRenderer(image_size, "hard")              # :107
        self.tex_renderer = SoftRenderer(image_size, renderer_type)        # :109-110
        self.tex_renderer.ambient_light_only()
        # outputs nobody reads are not computed: only tex_renderer's p2f is consumed (:226)
        self.renderer.need_p2f = False
        self.dis_renderer.need_p2f = False
        # ... and of the mask (:200) and unseen-view (:236) renders only the alpha channel is read
        self.renderer.alpha_only = True
        self.dis_renderer.alpha_only = True
        self.hard_renderer.ids_only = True      # only the face-id plane is read (:224)
        self.laplacian_loss_fn = loss_utils.LaplacianLoss(template_verts, faces)   # :141
        self.flatten_loss_fn = loss_utils.FlattenLoss(faces)                       # :142
        self.texture_cycle_fn = loss_utils.TexCycle()
        self.texture_loss = texture_loss or loss_utils.texture_loss_masks
        self.discriminator = discriminator
        self.epoch = epoch

    def set_epoch(self, epoch):
        """train_s1.py:250-255 reads `self.curr_epoch`: which regularisers enter the total depends on it."""
        self.epoch = epoch

    def forward(self, outputs, batch):
        w = self.w
        pred_vs, delta_v, proj_cam, tex_flow = outputs["pred_vs"], outputs["delta_v"], outputs["cam"], outputs["tex_flow"]
        imgs, masks, dts = batch["imgs"], batch["masks"], batch["dts_barrier"]
        B = pred_vs.shape[0]
        faces = self.faces[None].expand(B, -1, -1)
        terms = {}
        # shape losses (:199-206).  Of the mask render (:199) and the unseen-view render of the adversarial term (:235) only
        # alpha is read (:200, :236, :242).  share_mask_render = False runs them as ONE silhouette launch over 2B views (view
        # 2b = predicted camera, 2b+1 = rotated camera of image b), forward and backward
        random_cams = rotate_cam_y(proj_cam.detach(), batch["gan_angles"])                        # :232-233
        # texture losses (:209-230)
        tex = geom_utils.sample_textures(tex_flow, imgs)
        bs, fs = tex.shape[:2]
        tex = tex.reshape(bs, fs, -1, 3)
        # :217 textured soft-max render and :223-224 the hard render of the SAME mesh and camera of which only the face-id
        # plane is read: one launch (the z-buffer winner is tracked during the soft-max render's own visits)
        if self.share_mask_render:
            # ... and :199, the mask render of the same meshes and cameras again: its alpha channel, with the gradient to
            # vertices and camera the mask render has; the colour channels see the geometry detached as at :217
            # (lean_state: of this render the step reads the pooled image, p2f and the visible-face ids -- nothing else is written)
            texture_rgba, p2f_info, _, aggr_ids = self.tex_renderer(pred_vs, faces, proj_cam, tex, with_visibility=True,
                                                                    detach_rgb_geometry=True, lean_state=self.tex_renderer.anti_aliasing)
            if not self.tex_renderer.anti_aliasing:
                aggr_ids = aggr_ids[:, 1]
            mask_pred_seen = texture_rgba[:, 3]
            mask_pred_unseen = self.dis_renderer.silhouettes(pred_vs, faces, random_cams)
        else:
            both = self.renderer.silhouettes(pred_vs, faces, torch.stack((proj_cam, random_cams), dim=1).reshape(2 * B, 7))
            both = both.view(B, 2, both.shape[-2], both.shape[-1])
            mask_pred_seen, mask_pred_unseen = both[:, 0], both[:, 1]
            texture_rgba, p2f_info, _, aggr_info = self.tex_renderer(pred_vs.detach(), faces, proj_cam.detach(), tex,
                                                                     with_visibility=True)
            aggr_ids = aggr_info[:, 1]
        terms["mask"] = loss_utils.neg_iou_loss(mask_pred_seen, masks)
        terms["triangle"] = self.laplacian_loss_fn(pred_vs).mean()
        terms["flatten"] = self.flatten_loss_fn(pred_vs).mean()
        terms["deform"] = loss_utils.deform_l2reg(delta_v)
        terms["ori"] = loss_utils.sym_reg(pred_vs)
        texture_pred = texture_rgba[:, 0:3]
        terms["tex"] = self.texture_loss(texture_pred, imgs, masks, mask_pred_seen)
        terms["tex_dt"] = loss_utils.texture_dt_loss(tex_flow, dts)
        aggr_ids = aggr_ids.reshape(bs, -1)
        tex_cycle, _ = self.texture_cycle_fn(tex_flow, p2f_info.detach(), aggr_ids.detach())
        terms["tex_cycle"] = tex_cycle
        # adversarial term on the unseen view (:232-245)
        if self.discriminator is not None:
            pred = torch.cat((mask_pred_seen.detach(), mask_pred_unseen)).unsqueeze(1)             # :238, :243
            labels = gan_labels(self, B, pred.device)
            gan_preds = self.discriminator(pred)
            terms["gan"] = nn.functional.binary_cross_entropy_with_logits(gan_preds.view(-1), labels)
        else:  # keep the render + its backward in the step even without a discriminator network
            terms["gan"] = mask_pred_unseen.mean()
        # train_s1.py:247-265; the symmetry term only while epoch < stop_ori_epoch (:250), the deformation term only once
        # epoch > update_template_freq (:253) -- a gated-off term stays in `terms` for logging, as in the reference, and is
        # left out of the sum (weighted_total).  A HIP graph captured from this module bakes in the epoch's gating:
        # re-capture after set_epoch crosses stop_ori_epoch / update_template_freq (bench.py --graph runs one epoch).
        gated_off = [k for k, on in (("ori", self.epoch < w.stop_ori_epoch), ("deform", self.epoch > w.update_template_freq)) if not on]
        total = weighted_total(self, terms, [
            ("mask", w.mask_loss_wt), ("triangle", w.triangle_reg_wt), ("flatten", w.flatten_reg_wt),
            ("ori", w.ori_reg_wt), ("deform", w.deform_reg_wt), ("tex", w.tex_loss_wt), ("tex_dt", w.tex_dt_loss_wt),
            ("tex_cycle", w.tex_cycle_loss_wt), ("gan", w.gan_loss_wt)], skip=gated_off)
        return total, terms


class S2Weights:
    """experiments/train_s2.py:49-60 defaults."""
    mask_loss_wt = 2.5
    gan_loss_wt = 1.0
    triangle_reg_wt = 0.15
    flatten_reg_wt = 0.0005
    tex_loss_wt = 3.0
    tex_dt_loss_wt = 3.0
    tex_cycle_loss_wt = 1.0
    ent_loss_wt = 0.05
    prob_loss_wt = 5.0
    vertex_loss_wt = 10.0
    deform_reg_wt = 1.0


class RenderCompareS2(nn.Module):
    """train_s2 render-and-compare (experiments/train_s2.py:201-316): K=8 camera hypotheses, 22 raster forwards and
    21 backwards per image: MultiMaskLoss (8), MultiTextureLoss (8 + 1 hard), GAN view (1), part matching (4),
    chamfer vertex-part correspondence.

    outputs: pred_vs [B,V,3], delta_v, mean_shape [V,3], cam [B,7], cam_hypotheses [B,K,7], cam_probs [B,K], tex_flow
    batch:   imgs, masks, dts_barrier, gan_angles [B], part_segs [B,5,H,H], random_imgs [B,3,H,H] (previous step's
             masked images, :268), head/belly/back/neck_points [B,n,2]
    """

    def __init__(self, template_verts, faces, part_vertex_ids, uv_img, uv_sampler, image_size=256, num_hypo_cams=8,
                 weights=None, texture_loss_type="perceptual", discriminator=None, num_sym_faces=0, tex_size=6,
                 share_mask_render=True):
        super().__init__()
        # True: MultiMaskLoss's render of the B*K hypothesis views (loss_utils.py:265) is the alpha channel of
        # MultiTextureLoss's render of the same views (:313); False: two renders, as written there
        self.share_mask_render = share_mask_render
        self.w = weights or S2Weights()
        self.K = num_hypo_cams
        self.register_buffer("faces", faces.long())
        self.mask_loss_fn = loss_utils.MultiMaskLoss(image_size, "softmax", num_hypo_cams)          # :127-129
        self.laplacian_loss_fn = loss_utils.LaplacianLoss(template_verts, faces)                    # :135
        self.flatten_loss_fn = loss_utils.FlattenLoss(faces)                                        # :136
        self.texture_loss_fn = loss_utils.MultiTextureLoss(0, num_hypo_cams, image_size, "softmax", texture_loss_type)
        self.corr_loss_fn = loss_utils.CorrLossChamfer(part_vertex_ids, image_size)                 # :152
        self.part_loss_fn = loss_utils.part_matching_loss(uv_img, uv_sampler, num_sym_faces, im_size=image_size,
                                                          tex_size=tex_size)                        # :154-161
        self.dis_renderer = SoftRenderer(image_size, "softmax")                                     # :105-106
        self.dis_renderer.ambient_light_only()
        self.dis_renderer.need_p2f = False
        self.discriminator = discriminator

    def forward(self, outputs, batch):
        w, K = self.w, self.K
        pred_vs, delta_v = outputs["pred_vs"], outputs["delta_v"]
        B = pred_vs.shape[0]
        faces = self.faces[None].expand(B, -1, -1)
        imgs, masks = batch["imgs"], batch["masks"]
        proj_cam = outputs["cam"].detach()
        cams_all_hypo, cam_probs = outputs["cam_hypotheses"], outputs["cam_probs"]
        t = {}
        t["cam_div"] = -1 * (torch.log(cam_probs + 1E-9) * cam_probs).sum(1).mean()                 # :222
        tex_flow = outputs["tex_flow"]
        tex = geom_utils.sample_textures(tex_flow, imgs)
        bs, fs = tex.shape[:2]
        tex = tex.reshape(bs, fs, -1, 3)
        texture_rgba = self.texture_loss_fn.render_views(pred_vs, faces, cams_all_hypo, tex) if self.share_mask_render else None
        t["mask"], mask_all_hypo = self.mask_loss_fn(pred_vs, faces, cams_all_hypo, cam_probs, masks,
                                                     rendered_masks=None if texture_rgba is None else texture_rgba[:, 3])
        t["triangle"] = self.laplacian_loss_fn(pred_vs).mean()
        t["flatten"] = self.flatten_loss_fn(pred_vs).mean()
        t["deform"] = loss_utils.deform_l2reg(delta_v)
        t["tex"], t["tex_dt"], t["tex_cycle"], _ = self.texture_loss_fn(
            pred_vs.detach(), faces, cams_all_hypo.detach(), cam_probs.detach(), proj_cam, imgs, masks, mask_all_hypo,
            tex, tex_flow, batch["dts_barrier"], texture_rgba=texture_rgba)
        random_cams = rotate_cam_y(proj_cam, batch["gan_angles"])                                   # :257-258
        pred_unseen, _, _ = self.dis_renderer(pred_vs, faces, random_cams, tex.detach())            # :260
        if self.discriminator is not None:
            pred = torch.cat((batch["random_imgs"], pred_unseen[:, 0:3]))
            labels = gan_labels(self, B, pred.device)
            t["gan"] = nn.functional.binary_cross_entropy_with_logits(self.discriminator(pred).view(-1), labels)
        else:
            t["gan"] = pred_unseen[:, 0:3].mean()
        part_loss, _ = self.part_loss_fn(pred_vs, faces, proj_cam, batch["part_segs"])              # :294-296
        t["part"] = torch.mean(part_loss)
        mean_shape = outputs["mean_shape"][None].expand(B, -1, -1)
        rep = lambda x: x.unsqueeze(1).repeat(1, K, 1, 1).view(-1, x.size(1), x.size(2))
        # NOTE the reference passes (head, belly, back, neck) into parameters named (head, belly, neck, back)
        # (:311 vs loss_utils.py:223); the swapped pair carries weight 0, kept as is.
        corr = self.corr_loss_fn(rep(batch["head_points"]), rep(batch["belly_points"]), rep(batch["back_points"]),
                                 rep(batch["neck_points"]), rep(mean_shape), cams_all_hypo.reshape(-1, 7), avg=False)
        t["corr"] = (corr.view(B, K) * cam_probs.detach()).sum(dim=1).mean()                        # :313-314
        total = weighted_total(self, t, [
            ("mask", w.mask_loss_wt), ("triangle", w.triangle_reg_wt), ("flatten", w.flatten_reg_wt),
            ("deform", w.deform_reg_wt), ("tex", w.tex_loss_wt), ("tex_dt", w.tex_dt_loss_wt),
            ("tex_cycle", w.tex_cycle_loss_wt), ("gan", w.gan_loss_wt), ("cam_div", w.ent_loss_wt),
            ("part", w.prob_loss_wt), ("corr", w.vertex_loss_wt)])                                    # train_s2.py:300-316
        return total, t
