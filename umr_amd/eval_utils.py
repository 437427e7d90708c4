"""Evaluation metrics of BASELINE config 5 on the GPU path: mask IoU (experiments/test_iou.py:101-110) and
keypoint transfer / PCK (experiments/test_kp.py:125-193, 253-324).  Rendering, texture-flow sampling, camera
projection and nearest-neighbour search are the HIP kernels of the training path; the per-keypoint python/numpy
loops of the reference (heat-map drawing, CPU round trips) are vectorised on the device."""
import torch

from . import functional as UF
from .chamfer_python import distChamfer


def mask_iou(mask_gt, mask_pred):
    """test_iou.py:104-109: soft IoU per sample, no threshold, no epsilon.  [B,H,W] x2 -> [B]."""
    g = mask_gt.reshape(mask_gt.shape[0], -1)
    p = mask_pred.reshape(mask_pred.shape[0], -1)
    inter = g * p
    return inter.sum(1) / (g + p - inter).sum(1)


def create_grid(image_size, device):
    """utils/kp_utils.py:9-21 (affine_grid of the identity, torch-1.1.0 = align_corners=True) -> [H,W,2]."""
    return UF.standard_grid(image_size, torch.device(device))


def draw_labelmaps(kp_xy_pix, image_size, sigma):
    """utils/kp_utils.py:42-69 (draw_labelmap) for K keypoints at once: an un-normalised (6 sigma + 1)^2 Gaussian
    patch whose upper-left corner is int(pt - 3 sigma) (truncation toward zero, as python's int()).
    kp_xy_pix [K,2] (x, y) in pixels -> [K,H,W]."""
    dev = kp_xy_pix.device
    size = 6 * sigma + 1
    c0 = size // 2
    ul = torch.trunc(kp_xy_pix - 3 * sigma)                     # [K,2]
    br = torch.trunc(kp_xy_pix + 3 * sigma + 1)
    ys, xs = torch.meshgrid(torch.arange(image_size, device=dev, dtype=torch.float32),
                            torch.arange(image_size, device=dev, dtype=torch.float32), indexing="ij")
    gx = xs[None] - ul[:, 0, None, None]
    gy = ys[None] - ul[:, 1, None, None]
    inside = (gx >= 0) & (gx < size) & (gy >= 0) & (gy < size) & (xs[None] < br[:, 0, None, None]) & (ys[None] < br[:, 1, None, None])
    g = torch.exp(-((gx - c0) ** 2 + (gy - c0) ** 2) / (2.0 * sigma ** 2))
    return torch.where(inside, g, torch.zeros_like(g))


def map_kp_flow(kp_src, flow_src, flow_tgt, image_size=256, sigma=3):
    """test_kp.py:125-158 (flow mode).  kp_src [K,>=2] in [-1,1]; flow_* [F,T,T,2] -> transferred keypoints [K,2]:
    keypoint -> face (arg-max of the keypoint heat map sampled at the source flow) -> target pixel (mean of the
    coordinate grid sampled at the target flow)."""
    nf = flow_tgt.size(0)
    dev = flow_tgt.device
    sgrid = create_grid(image_size, dev).permute(2, 0, 1).unsqueeze(0).contiguous()          # [1,2,H,W]
    p2face = UF.GridSampleCLFunction.apply(sgrid, flow_tgt.reshape(1, -1, 2))                   # [1, F*TT, 2]
    p2face = p2face.view(nf, -1, 2).mean(dim=1)                                                # [F,2]
    kp_pix = (kp_src[:, 0:2] + 1) / 2.0 * 256                                                  # :146 (hard-coded 256)
    hp = draw_labelmaps(kp_pix.to(dev), image_size, sigma).unsqueeze(0).contiguous()           # [1,K,H,W]
    k2face = UF.GridSampleCLFunction.apply(hp, flow_src.reshape(1, -1, 2))                      # [1, F*TT, K]
    k2face = k2face.view(nf, -1, hp.size(1)).mean(dim=1)                                       # [F,K]
    k2face_idx = torch.max(k2face, dim=0)[1]                                                   # [K]
    return p2face[k2face_idx]


def map_kp_cam(kp_src, cam_src, cam_tgt, mask_tgt, mean_shape, image_size=256):
    """test_kp.py:160-193 (cam mode): keypoint -> nearest projected template vertex under the source camera ->
    that vertex under the target camera -> nearest foreground pixel of the target mask."""
    dev = mean_shape.device
    ms = mean_shape.view(1, -1, 3).contiguous()
    v_tgt = UF.ProjectPointsFunction.apply(ms, cam_tgt.view(1, 7).contiguous(), 2, 0.0)
    sgrid = create_grid(image_size, dev).reshape(-1, 2)
    fg_coords = sgrid[torch.nonzero(mask_tgt.reshape(-1)).squeeze(1), :]
    _, _, _, proj2fg_idx = distChamfer(fg_coords.unsqueeze(0).contiguous(), v_tgt)
    v_src = UF.ProjectPointsFunction.apply(ms, cam_src.view(1, 7).contiguous(), 2, 0.0)
    _, _, kp2proj_idx, _ = distChamfer(kp_src[:, 0:2].to(dev).unsqueeze(0).contiguous(), v_src)
    return fg_coords[proj2fg_idx.squeeze(0).long()[kp2proj_idx.squeeze(0).long()], :]


def pck(kps_pred, kps_gt, kps_vis, padding_frac=0.05, thresholds=(0.1, 0.15)):
    """test_kp.py:253-258, 317-323.  kps_pred/kps_gt [P,K,2], kps_vis [P,K] -> tuple of PCK@t (mean over keypoints
    of correct/visible)."""
    err = torch.sqrt(((kps_pred - kps_gt) ** 2).sum(-1)) * ((1 + 2 * padding_frac) / 2.0)
    n_vis = kps_vis.sum(0)
    return tuple(float((((err < t).float() * kps_vis).sum(0) / n_vis).mean()) for t in thresholds)
