#!/usr/bin/env python3
"""Round-3 golden vectors, generated from the REFERENCE ITSELF in this container (python oracle/gen_golden_r3.py); same harness
and rules as oracle/gen_golden.py: the reference's modules are imported unmodified from /root/reference, nothing is written
there, only inputs + expected outputs are committed under tests/golden/.

  loss_masked_l1.npz   nnutils/loss_utils.py:103-116 texture_loss_masks, avg True / False, gradients wrt img_pred, mask_pred
  rotate_cam.npz       nnutils/geom_utils.py:167-193 rotate_cam for several axes and per-sample angles, through the imported
                       utils/transformations.py; cv2.Rodrigues (cv2 is absent) is restated below from its definition
  eval_kp.npz          the keypoint-transfer evaluation of experiments/test_kp.py:125-193: utils/kp_utils.py (create_grid,
                       draw_labelmap), nnutils/chamfer_python.py and nnutils/smr.py are imported; the two mapping methods are
                       trainer methods of a script that needs absl / the CUB loader, so their bodies (15 + 20 lines of torch
                       calls on those imported functions) are re-typed here with the reference line numbers
"""
import sys
sys.dont_write_bytecode = True
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as GG  # noqa: E402

OUT = GG.OUT
np_ = GG.np_


def rodrigues(rvec):
    """cv2.Rodrigues for a rotation vector: R = cos(t) I + (1 - cos(t)) r r^T + sin(t) [r]_x, t = |rvec|, r = rvec / t."""
    rvec = np.asarray(rvec, np.float64).reshape(3)
    t = np.linalg.norm(rvec)
    if t < 1e-12:
        return np.eye(3), None
    r = rvec / t
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return np.cos(t) * np.eye(3) + (1 - np.cos(t)) * np.outer(r, r) + np.sin(t) * K, None


def main():
    lib, sr, smr, loss_utils, geom_utils, chamfer_python, scops_utils, _ = GG.install_reference()
    g = torch.Generator().manual_seed(2026)

    # ---- masked L1 -------------------------------------------------------------------------------------------
    B, C, H = 3, 3, 20
    ip = torch.rand(B, C, H, H, generator=g).requires_grad_(True)
    ig = torch.rand(B, C, H, H, generator=g)
    mg = (torch.rand(B, H, H, generator=g) > 0.4).float()
    mp = torch.rand(B, H, H, generator=g).requires_grad_(True)
    la = loss_utils.texture_loss_masks(ip, ig, mg, mp, avg=True)
    la.backward()
    ga_ip, ga_mp = ip.grad.clone(), mp.grad.clone()
    ip.grad = None; mp.grad = None
    w = torch.rand(B, generator=g)
    lp = loss_utils.texture_loss_masks(ip, ig, mg, mp, avg=False)
    (lp * w).sum().backward()
    np.savez_compressed(os.path.join(OUT, "loss_masked_l1.npz"), img_pred=np_(ip), img_gt=np_(ig), mask_gt=np_(mg), mask_pred=np_(mp),
                        loss_avg=np_(la), grad_img_pred_avg=np_(ga_ip), grad_mask_pred_avg=np_(ga_mp), w=np_(w),
                        loss_per_sample=np_(lp), grad_img_pred_w=np_(ip.grad), grad_mask_pred_w=np_(mp.grad))
    print("loss_masked_l1: avg %.6f per-sample %s" % (la.item(), np_(lp)))

    # ---- rotate_cam ------------------------------------------------------------------------------------------
    sys.modules["cv2"].Rodrigues = rodrigues
    geom_utils.cv2 = sys.modules["cv2"]
    n = 12
    q = torch.randn(n, 4, generator=g); q = q / q.norm(dim=1, keepdim=True)
    q[0] = torch.tensor([1., 0, 0, 0]); q[1] = torch.tensor([0., 0, 1., 0]); q[2] = -q[2].abs()      # identity, pi about y, w < 0
    cam = torch.cat([0.5 + torch.rand(n, 1, generator=g), torch.rand(n, 2, generator=g) - 0.5, q], 1)
    angles = torch.cat([torch.tensor([0., 90., 180., 45.]), torch.randint(0, 360, (n - 4,), generator=g).float()])
    res = {}
    for name, axis in (("y", [0, 1, 0]), ("x", [1, 0, 0]), ("z", [0, 0, 1]), ("d", [0.6, 0.0, 0.8])):
        res["new_cam_" + name] = np_(geom_utils.rotate_cam(cam, angles.numpy(), axis=axis))
        res["axis_" + name] = np.asarray(axis, np.float32)
    np.savez_compressed(os.path.join(OUT, "rotate_cam.npz"), cam=np_(cam), angles=np_(angles), **res)
    print("rotate_cam: %d cameras x 4 axes" % n)

    # ---- keypoint transfer (test_kp.py) ----------------------------------------------------------------------
    from UMR.utils import kp_utils
    image_size, sigma, K, F_, T = 256, 3, 15, 320, 6     # 320 faces / 162 vertices keep the fixture small; the index logic is size-free
    P = 4                                           # pairs
    renderer = smr.SoftRenderer(image_size, 'softmax')
    verts, _, _, _ = GG.scene(1, 2, seed=3)
    mean_shape = (verts[0] * 0.9).contiguous()
    kps = torch.rand(P, 2, K, 3, generator=g) * 1.8 - 0.9
    kps[0, 0, 0, :2] = torch.tensor([-0.99, 0.98])          # Gaussian patch clipped by the image border
    kps[0, 0, 1, :2] = torch.tensor([1.2, 0.0])             # patch entirely outside: the heat map stays empty
    kps[:, :, :, 2] = (kps[:, :, :, 2] > -0.8).float()
    # smooth texture flows (a real network's flows vary slowly over neighbouring faces; random ones would make every face's
    # heat-map response a tie at 0)
    base = torch.rand(P, 2, F_, 1, 1, 2, generator=g) * 1.6 - 0.8
    flows = (base + 0.08 * (torch.rand(P, 2, F_, T, T, 2, generator=g) - 0.5)).clamp(-1, 1)
    flows = flows.half().float()                    # fp16-representable values: stored as float16, used as float32
    cams = torch.cat([0.6 + 0.3 * torch.rand(P, 2, 1, generator=g), 0.2 * torch.rand(P, 2, 2, generator=g) - 0.1,
                      torch.nn.functional.normalize(torch.randn(P, 2, 4, generator=g), dim=2)], 2)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, image_size), torch.linspace(-1, 1, image_size), indexing="ij")
    masks = torch.stack([torch.stack([(((xx - 0.1 * i) ** 2 + (yy + 0.05 * j) ** 2) < (0.3 + 0.05 * i)).float() for j in range(2)])
                         for i in range(P)])

    def map_flow(kp_src, flow_src, flow_tgt):           # experiments/test_kp.py:125-158
        grid_size = torch.Size((1, 2, image_size, image_size))
        sgrid = kp_utils.create_grid(grid_size)                                               # :131
        sgrid = sgrid.permute(0, 3, 1, 2)                                                     # :132
        nf = flow_tgt.size(0)
        p2face = torch.nn.functional.grid_sample(sgrid, flow_tgt.view(1, nf, -1, 2))          # :136
        p2face = torch.mean(p2face, dim=-1).permute(0, 2, 1)                                  # :138
        p2face = p2face.cpu().squeeze()
        kp_num = kp_src.size(0)
        hp = torch.zeros(1, kp_num, image_size, image_size)                                   # :144
        kp_src = (kp_src[:, 0:2] + 1) / 2.0 * 256                                             # :145
        for ccnt in range(kp_num):
            hp[0, ccnt] = kp_utils.draw_labelmap(hp[0, ccnt], (kp_src[ccnt][0], kp_src[ccnt][1]), sigma=sigma)   # :147
        k2face = torch.nn.functional.grid_sample(hp, flow_src.view(1, nf, -1, 2))             # :150
        k2face = torch.mean(k2face, dim=-1).cpu()                                             # :151
        _, k2face_idx = torch.max(k2face, dim=-1)                                             # :152
        return p2face[k2face_idx], k2face_idx, hp, k2face                                     # :155

    def map_cam(kp_src, cam_src, cam_tgt, mask_tgt):    # experiments/test_kp.py:160-193
        cam_src = cam_src.view(1, 7); cam_tgt = cam_tgt.view(1, 7)
        vert2ds_tgt = renderer.project_points(mean_shape[None], cam_tgt)                      # :168
        grid_size = torch.Size((1, 2, image_size, image_size))
        sgrid2D = kp_utils.create_grid(grid_size).squeeze()                                   # :173
        sgrid = sgrid2D.view(-1, 2)
        mask_tgt = mask_tgt.view(-1)
        fg_idx = torch.nonzero(mask_tgt).squeeze()                                            # :177
        fg_coords = sgrid[fg_idx, :]
        fg2proj, proj2fg, fg2proj_idx, proj2fg_idx = chamfer_python.distChamfer(fg_coords.unsqueeze(0), vert2ds_tgt)   # :180
        vert2ds_src = renderer.project_points(mean_shape[None], cam_src)                      # :183
        kp_src = kp_src[:, 0:2]
        kp2proj, _, kp2proj_idx, _ = chamfer_python.distChamfer(kp_src.unsqueeze(0), vert2ds_src)                      # :188
        kp2proj_idx = kp2proj_idx.squeeze().long(); proj2fg_idx = proj2fg_idx.squeeze().long()
        kp2fg = fg_coords[proj2fg_idx[kp2proj_idx], :]                                        # :192
        return kp2fg.view(1, kp_src.size(0), 2), kp2proj_idx, proj2fg_idx

    out = dict(kps=np_(kps), flows=np_(flows).astype(np.float16), cams=np_(cams), masks=masks.numpy().astype(np.uint8), mean_shape=np_(mean_shape),
               image_size=image_size, sigma=sigma)
    k12_f, k21_f, i12_f, i21_f, k12_c, k21_c, i12_c, i21_c = [], [], [], [], [], [], [], []
    for p in range(P):
        a, ia, hp, k2face = map_flow(kps[p, 0], flows[p, 0], flows[p, 1]); b, ib, _, _ = map_flow(kps[p, 1], flows[p, 1], flows[p, 0])
        k12_f.append(np_(a).reshape(K, 2)); k21_f.append(np_(b).reshape(K, 2)); i12_f.append(np_(ia.reshape(-1))); i21_f.append(np_(ib.reshape(-1)))
        if p == 0:
            out["k2face_pair0"] = np_(k2face[0])
        c, ic, jc = map_cam(kps[p, 0], cams[p, 0], cams[p, 1], masks[p, 1]); d, id_, jd = map_cam(kps[p, 1], cams[p, 1], cams[p, 0], masks[p, 0])
        k12_c.append(np_(c[0])); k21_c.append(np_(d[0])); i12_c.append(np_(ic)); i21_c.append(np_(id_))
    out.update(flow_k1_to_k2=np.stack(k12_f), flow_k2_to_k1=np.stack(k21_f), flow_face_12=np.stack(i12_f), flow_face_21=np.stack(i21_f),
               cam_k1_to_k2=np.stack(k12_c), cam_k2_to_k1=np.stack(k21_c), cam_vert_12=np.stack(i12_c), cam_vert_21=np.stack(i21_c))
    # PCK of the flow-mode transfers as test_kp.py:253-258, 317-323 computes it
    padding_frac = 0.05
    errs, vis = [], []
    for p in range(P):
        kps_gt = kps[p, :, :, 0:2].numpy()
        kps_vis = (kps[p, 0, :, 2] * kps[p, 1, :, 2]).view(1, K).repeat(2, 1)
        kps_pred = np.stack([out["flow_k2_to_k1"][p], out["flow_k1_to_k2"][p]])      # torch.cat((k2_to_k1, k1_to_k2), dim=0), :255
        e = kps_pred - kps_gt
        errs.append(np.sqrt(np.sum(e * e, axis=2)) * (1 + 2 * padding_frac) / 2.0); vis.append(kps_vis.numpy())
    errs, vis = np.concatenate(errs), np.concatenate(vis)
    n_vis = np.sum(vis, axis=0)
    out["pck1"] = (np.sum((errs < 0.1) * vis, axis=0) / n_vis).mean()
    out["pck15"] = (np.sum((errs < 0.15) * vis, axis=0) / n_vis).mean()
    np.savez_compressed(os.path.join(OUT, "eval_kp.npz"), **out)
    print("eval_kp: %d pairs, PCK.1 %.3f PCK.15 %.3f, distinct flow faces %d" % (P, out["pck1"], out["pck15"],
                                                                                 len(np.unique(out["flow_face_12"]))))


if __name__ == "__main__":
    main()
