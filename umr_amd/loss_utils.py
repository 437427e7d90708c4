"""Losses of the render-and-compare loop with the reference's names and signatures
(nnutils/loss_utils.py, external/SoftRas/soft_renderer/losses.py), HIP-backed."""
import numpy as np
import torch
import torch.nn as nn

from . import functional as UF
from .chamfer_python import distChamfer
from .smr import SoftRenderer


def neg_iou_loss(predict, target, avg=True):
    """nnutils/loss_utils.py:41-48."""
    per = UF.neg_iou(predict, target)
    if avg:
        return per.sum() / per.nelement()   # == 1 - (I/U).sum()/n
    return per


def texture_dt_loss(texture_flow, dist_transf, vis_rend=None, cams=None, verts=None, tex_pred=None):
    """nnutils/loss_utils.py:50-90 (the interactive visualisation branch is not carried over)."""
    B, F, T = texture_flow.shape[0], texture_flow.shape[1], texture_flow.shape[-2]
    s = UF.grid_sample_cl(dist_transf, texture_flow.reshape(B, F * T * T, 2))
    return s.mean()


def texture_loss_masks(img_pred, img_gt, mask_gt, mask_pred, avg=True):
    """nnutils/loss_utils.py:103-116: L1(img_pred * mask_pred, img_gt * mask_gt); one kernel pair (umr_masked_l1_*)."""
    per = UF.masked_l1(img_pred, img_gt, mask_gt, mask_pred)     # [B]: mean over (C, H, W) per sample
    return per.mean() if avg else per


def deform_l2reg(V):
    """nnutils/loss_utils.py:118-123: mean row norm of V [B,N,3] (umr_row_norm_mean_*)."""
    return UF.row_norm_mean(V)


def sym_reg(verts):
    """nnutils/loss_utils.py:125-126: mean |y| of verts [B,V,3] (umr_abs_mean_*)."""
    return UF.abs_column_mean(verts, 1)


class TexCycle(nn.Module):
    """nnutils/loss_utils.py:152-182.  The per-sample torch.unique + host mask + H2D copy of the
    reference (:173-179) is one kernel (umr_visible_face_mask); id -1 still marks the last face."""

    def __init__(self, im_size=256, nf=1280, eps=1e-12):
        super(TexCycle, self).__init__()

    def forward(self, flow, prob, aggr_info):
        nb, nf = flow.shape[:2]
        avg_flow = torch.mean(flow.view(nb, nf, -1, 2), dim=2)
        mask = UF.visible_face_mask(aggr_info.reshape(nb, -1), nf).unsqueeze(-1)
        loss = torch.nn.MSELoss()(avg_flow * mask, prob * mask)
        return loss, avg_flow[0, 0:10, :]


class MultiMaskLoss(nn.Module):
    """nnutils/loss_utils.py:250-275."""

    def __init__(self, image_size=256, renderer_type="softmax", num_hypo_cams=8):
        super(MultiMaskLoss, self).__init__()
        self.renderer = SoftRenderer(image_size, renderer_type)
        self.renderer.need_p2f = False       # the reference discards p2f/aggr here (:265)
        self.renderer.alpha_only = True      # ... and reads the alpha channel only (:266)
        self.num_hypo_cams = num_hypo_cams
        self.image_size = image_size

    def forward(self, vs, fs, cams_all_hypo, cam_probs, masks_gt, rendered_masks=None):
        """rendered_masks [B*K,H,H]: the alpha channel of a render of the same views that keeps its gradient to vs and
        cams_all_hypo (MultiTextureLoss.render_views) -- the mask render (:265) is then not repeated."""
        bs = vs.size(0)
        K = self.num_hypo_cams
        if rendered_masks is not None:
            mask_all_hypo = rendered_masks
        else:
            # the reference materialises vs / fs K times (:260-262); here the K hypotheses of image b are views
            # b*K .. b*K+K-1 of mesh b (mesh_group indexing in the projection kernel), gradients summed over them
            cams_all_hypo_flat = cams_all_hypo.view(-1, 7)
            pred, _, _ = self.renderer.forward(vs, fs, cams_all_hypo_flat)
            mask_all_hypo = pred[:, 3, :, :]
        masks = masks_gt.unsqueeze(1).repeat(1, K, 1, 1).view(-1, self.image_size, self.image_size)
        loss = neg_iou_loss(mask_all_hypo, masks, avg=False)
        loss = loss.view(bs, -1) * cam_probs
        loss = loss.sum(dim=1)
        return loss.mean(), mask_all_hypo


class CorrLossChamfer(nn.Module):
    """nnutils/loss_utils.py:194-248.  `scops_path` may be a directory holding vertices_idx/*.npy as in
    the reference, or a dict {'head','belly','neck','back'} of index arrays (the SCOPS files are not
    distributed with the reference)."""

    def __init__(self, scops_path, image_size):
        super(CorrLossChamfer, self).__init__()
        import os.path as osp
        names = ("head", "belly", "neck", "back")
        if isinstance(scops_path, dict):
            ids = [np.asarray(scops_path[n]) for n in names]
        else:
            ids = [np.load(osp.join(scops_path, "vertices_idx/%s_vertices.npy" % n)) for n in names]
        for n, i in zip(names, ids):
            self.register_buffer(n + "_vertices", torch.from_numpy(i).long())
        self.renderer = SoftRenderer(image_size)
        self.weights = [1, 1, 0, 0]
        self.nums = list(np.cumsum([len(i) for i in ids]))

    def forward(self, head_points, belly_points, neck_points, back_points, verts, cams, avg=True):
        vert_coords = torch.cat((verts[:, self.head_vertices, :], verts[:, self.belly_vertices, :],
                                 verts[:, self.neck_vertices, :], verts[:, self.back_vertices, :]), dim=1)
        vert2d = self.renderer.project_points(vert_coords.contiguous(), cams)
        nums = [0] + self.nums
        pts = (head_points, belly_points, neck_points, back_points)
        cds = []
        for i in range(4):
            d1, _, _, _ = distChamfer(vert2d[:, nums[i]:nums[i + 1], :].contiguous(), pts[i])
            cds.append(d1 * self.weights[i])
        loss = torch.mean(torch.cat(cds, dim=1), dim=1)
        if avg:
            return torch.mean(loss), vert2d
        return loss


def _mesh_tables(faces):
    """CSR vertex adjacency + flatten edge quads from a [F,3] face array (host, init-time)."""
    faces = np.asarray(faces).astype(np.int64)
    nv = int(faces.max()) + 1
    nbrs = [set() for _ in range(nv)]
    for a, b, c in faces:
        nbrs[a].update((b, c)); nbrs[b].update((a, c)); nbrs[c].update((a, b))
    off = np.zeros(nv + 1, np.int32)
    off[1:] = np.cumsum([len(s) for s in nbrs])
    idx = np.concatenate([np.sort(list(s)) for s in nbrs]).astype(np.int32)
    return off, idx


class LaplacianLoss(nn.Module):
    """external/SoftRas/soft_renderer/losses.py:6-37, CSR neighbour walk instead of the dense [V,V] matmul."""

    def __init__(self, vertex, faces, average=False):
        super(LaplacianLoss, self).__init__()
        self.nv = vertex.size(0)
        self.nf = faces.size(0)
        self.average = average
        off, idx = _mesh_tables(faces.detach().cpu().numpy())
        assert len(off) == self.nv + 1
        self.register_buffer('nbr_off', torch.from_numpy(off))
        self.register_buffer('nbr_idx', torch.from_numpy(idx))

    def forward(self, x):
        loss = UF.laplacian(x, self.nbr_off, self.nbr_idx)
        if self.average:
            return loss.sum() / x.size(0)
        return loss


class FlattenLoss(nn.Module):
    """external/SoftRas/soft_renderer/losses.py:39-114."""

    def __init__(self, faces, average=False):
        super(FlattenLoss, self).__init__()
        self.nf = faces.size(0)
        self.average = average
        f = faces.detach().cpu().numpy().astype(np.int64)
        # unique undirected edges from (f0,f1),(f1,f2) as the reference collects them (:45), plus the two
        # opposite vertices in face order (:50-63), built with an edge->faces map instead of an O(E*F) scan
        edges = sorted(set(tuple(v) for v in np.sort(np.concatenate((f[:, 0:2], f[:, 1:3]), axis=0))))
        emap = {}
        for fi, face in enumerate(f):
            for a, b in ((0, 1), (1, 2), (2, 0)):
                emap.setdefault(tuple(sorted((face[a], face[b]))), []).append(fi)
        quads = []
        for v0, v1 in edges:
            opp = []
            for fi in emap[(v0, v1)]:
                face = f[fi]
                opp.append(int(face[(face != v0) & (face != v1)][0]))
            quads.append((v0, v1, opp[0], opp[1]))
        self.register_buffer('quads', torch.tensor(quads, dtype=torch.int32))

    def forward(self, vertices, eps=1e-6):
        loss = UF.flatten(vertices, self.quads)
        if self.average:
            return loss.sum() / vertices.size(0)
        return loss


class MultiTextureLoss(nn.Module):
    """nnutils/loss_utils.py:277-331 (renderer='smr').  texture_loss_type 'perceptual' uses the AlexNet
    cosine distance (umr_amd/perceptual.py, random weights offline); anything else the masked L1."""

    def __init__(self, samples_per_gpu=32, num_hypo_cams=8, image_size=256, renderer_type="softmax",
                 texture_loss_type="perceptual", renderer="smr"):
        super(MultiTextureLoss, self).__init__()
        self.renderer = SoftRenderer(image_size, renderer_type)
        self.renderer.ambient_light_only()
        self.renderer.need_p2f = False            # :313 discards p2f / aggr
        self.hard_renderer = SoftRenderer(image_size, "hard")
        self.hard_renderer.ids_only = True        # only aggr_info[:, 1] is read (:328)
        if texture_loss_type in "perceptual":
            from .perceptual import PerceptualTextureLoss
            self._ptl = PerceptualTextureLoss()
            self.pnet = self._ptl.perceptual_loss.model      # registered so .to(device) moves it
            self.texture_loss = self._ptl
        else:
            self.texture_loss = texture_loss_masks
        self.texture_cycle_fn = TexCycle(samples_per_gpu)
        self.num_hypo_cams = num_hypo_cams
        self.image_size = image_size

    def render_views(self, vs, fs, cams_all_hypo, tx):
        """The textured render of all B*K hypothesis views (:313) as the ONE render of those views in the step: colour
        channels with the geometry detached, as the reference passes it (vs.detach(), cams.detach()), alpha channel with
        its gradient to vs and cams_all_hypo -- which makes it MultiMaskLoss's render (:265) as well (same meshes, same
        cameras, same rasterizer settings; alpha depends on neither textures nor lighting).  -> [B*K,4,H,H]"""
        return self.renderer.forward(vs, fs, cams_all_hypo.view(-1, 7), tx, detach_rgb_geometry=True,
                                     lean_state=self.renderer.anti_aliasing)[0]

    def forward(self, vs, fs, cams_all_hypo, cam_probs, proj_cam, rgbs, masks_gt, masks_pred, tx, tex_flow,
                dts_barrier, texture_rgba=None):
        """texture_rgba: the result of render_views() when the caller shares that render with the mask term."""
        bs, K = vs.size(0), self.num_hypo_cams
        if texture_rgba is None:
            # :303-306 repeats vertices, faces and the [B,F,36,3] texels K times (70 MB at B*K = 128); folded into
            # mesh / texture group indexing of the kernels instead
            cams_all_hypo_flat = cams_all_hypo.view(-1, 7)
            texture_rgba, _, _ = self.renderer.forward(vs.detach(), fs, cams_all_hypo_flat, tx)
        texture_pred = texture_rgba[:, 0:3, :, :]
        imgs = rgbs.unsqueeze(1).repeat(1, K, 1, 1, 1).view(-1, 3, self.image_size, self.image_size)
        masks_gt = masks_gt.unsqueeze(1).repeat(1, K, 1, 1).view(-1, self.image_size, self.image_size)
        tex_loss = self.texture_loss(texture_pred, imgs, masks_gt, masks_pred, avg=False)
        tex_loss = (tex_loss.view(bs, -1) * cam_probs).sum(dim=1).mean()
        tex_dt_loss = texture_dt_loss(tex_flow, dts_barrier)
        # visibility map from the HARD renderer; its p2f is identically 0 (reference quirk)
        _, p2f_info, aggr_info = self.hard_renderer(vs.detach(), fs, proj_cam.detach())
        aggr_info = aggr_info[:, 1, :, :].reshape(bs, -1)
        tex_cycle_loss, _ = self.texture_cycle_fn(tex_flow, p2f_info.detach(), aggr_info.detach())
        return tex_loss, tex_dt_loss, tex_cycle_loss, texture_pred


class part_matching_loss(nn.Module):
    """nnutils/loss_utils.py:333-440 (loss_type 'mse').  `scops_path` may be the reference's directory holding
    semantic_seg.png, or a tensor `uv_img` [1,1,128,256] of part labels 0..4 (the SCOPS template is not
    distributed with the reference)."""

    def __init__(self, scops_path, uv_sampler, num_sym_faces, im_size=256, batch_size=32, loss_type='mse', tex_size=6,
                 num_cam=1):
        super(part_matching_loss, self).__init__()
        if torch.is_tensor(scops_path):
            uv_img = scops_path.float().view(1, 1, 128, 256)
        else:
            import os.path as osp
            import imageio
            uv_img = torch.from_numpy(imageio.imread(osp.join(scops_path, "semantic_seg.png"))).view(1, 1, 128, 256).float()
        uv_img = uv_img.to(uv_sampler.device)
        tex = torch.nn.functional.grid_sample(uv_img, uv_sampler, mode='bilinear', padding_mode='zeros',
                                              align_corners=True)
        tex = tex.view(tex.size(0), -1, tex.size(2), tex_size, tex_size).permute(0, 2, 3, 4, 1)
        if num_sym_faces:
            tex = torch.cat([tex, tex[:, -num_sym_faces:]], 1)
        stex = torch.round(tex.reshape(tex.size(1), -1))
        nf, nt = stex.size()
        one_hot = torch.zeros(nf * nt, 5, device=stex.device)
        one_hot.scatter_(1, stex.view(-1, 1).long().clamp(0, 4), 1)
        stex_one_hot = one_hot.view(1, nf, nt, 5)
        # one 3-channel one-hot texture per part; expanded per call instead of stored batch_size times (:360-363)
        for i in range(1, 5):
            self.register_buffer("stex%d" % i, stex_one_hot[:, :, :, i].unsqueeze(-1).repeat(1, 1, 1, 3))
        # parts 1-3 as the three colour channels of ONE texture set: the reference renders each part separately with its
        # one-hot replicated over r, g, b "because the renderer can only render 3-channel images" (:357-359) -- four
        # renders of identical geometry; channel k of one render of (part1, part2, part3) is the same image
        self.register_buffer("stex123", stex_one_hot[:, :, :, 1:4].contiguous(), persistent=False)
        self.renderer = SoftRenderer(im_size, "softmax")
        self.renderer.ambient_light_only()
        self.renderer.need_p2f = False
        self.im_size = im_size
        self.register_buffer("weights", torch.tensor([0, 5.0, 0.0, 0.0, 5.0]).view(1, 5, 1, 1))
        self._w5 = (0.0, 5.0, 0.0, 0.0, 5.0)            # the same weights by value for the fused reduction (:377-378)
        self.loss_type = loss_type

    def forward(self, verts, faces, cams, part_segs, cam_probs=None, avg=True):
        bs = verts.size(0)
        # two renders instead of four (:385-397), textures shared by the whole batch through group indexing instead of
        # being expanded to [bs,F,36,3]; mean over three identical channels (:386) == the channel
        proj_a, _, _ = self.renderer(verts, faces, cams, self.stex123)
        proj_b, _, _ = self.renderer(verts, faces, cams, self.stex4)
        projs = [proj_a[:, 0:1], proj_a[:, 1:2], proj_a[:, 2:3], proj_b[:, 0:1]]
        # everything after the renders (:399-440: background plane, soft-max over the 5 planes, SCOPS soft centroids of
        # both stacks, per-plane max normalisation, weighted MSE) is one fused op: 4 launches forward, 1 backward
        l_eqv, l_lm = UF.part_match(proj_a, proj_b, part_segs, self._w5, 0.1, 1e-3)
        H, W = proj_a.shape[2], proj_a.shape[3]
        if avg:
            loss_eqv = l_eqv.sum() / (bs * 5 * H * W)
            loss_lmeqv = l_lm.sum() / (bs * 8)
        else:
            loss_eqv = ((l_eqv / (5 * H * W)).view(cam_probs.size()) * cam_probs).sum(dim=1).mean()
            loss_lmeqv = ((l_lm / 8).view(cam_probs.size()) * cam_probs).sum(dim=1).mean()
        return (loss_eqv + loss_lmeqv) / 4.0, projs
