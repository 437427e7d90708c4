"""Evaluation metrics of BASELINE config 5 on the GPU path: mask IoU (experiments/test_iou.py:101-110) and
keypoint transfer / PCK (experiments/test_kp.py:125-193, 253-324).  Rendering, texture-flow sampling, camera
projection and nearest-neighbour search are the HIP kernels of the training path; the per-keypoint python/numpy loops of the
reference (heat-map drawing, host round trips, numpy PCK accumulation) are the kernels of csrc/eval.hip: a batch of pairs
is two (flow mode) or four (cam mode) launches, PCK counters stay on the device."""
import torch

from . import functional as UF


def mask_iou(mask_gt, mask_pred):
    """test_iou.py:104-109: soft IoU per sample, no threshold, no epsilon.  [B,H,W] x2 -> [B]."""
    g = mask_gt.reshape(mask_gt.shape[0], -1)
    p = mask_pred.reshape(mask_pred.shape[0], -1)
    inter = g * p
    return inter.sum(1) / (g + p - inter).sum(1)


def create_grid(image_size, device):
    """utils/kp_utils.py:9-21 (affine_grid of the identity, torch-1.1.0 = align_corners=True) -> [H,W,2]."""
    return UF.standard_grid(image_size, torch.device(device))


_PATCH_CACHE = {}


def gaussian_patch(sigma, device):
    """The (6 sigma + 1)^2 un-normalised Gaussian of utils/kp_utils.py:53-59, computed in float64 with numpy exactly as
    draw_labelmap does and rounded to float32 as its assignment into the float heat-map tensor does."""
    key = (int(sigma), str(device))
    if key not in _PATCH_CACHE:
        import numpy as np
        size = 6 * int(sigma) + 1
        x = np.arange(0, size, 1, float)
        y = x[:, np.newaxis]
        x0 = y0 = size // 2
        g = np.exp(- ((x - x0) ** 2 + (y - y0) ** 2) / (2 * int(sigma) ** 2))
        _PATCH_CACHE[key] = torch.from_numpy(g.astype(np.float32)).to(device).contiguous()
    return _PATCH_CACHE[key]


EVAL_MAX_K = 32      # keypoints per pair the device counters take (csrc/eval.hip)


def _check_pck_args(K, kp_gt, vis, counters):
    given = [a is not None for a in (kp_gt, vis, counters)]
    if any(given) and not all(given):
        raise ValueError("keypoint transfer: kp_gt, vis and counters go together (all three for PCK accumulation, or none)")
    if K > EVAL_MAX_K:
        raise ValueError("keypoint transfer: %d keypoints per pair; the device path takes at most %d" % (K, EVAL_MAX_K))


def pck(kps_pred, kps_gt, kps_vis, padding_frac=0.05, thresholds=(0.1, 0.15)):
    """test_kp.py:253-258 for one batch of transferred keypoints already on hand: kps_pred / kps_gt [P,K,>=2] in [-1,1],
    kps_vis [P,K] -> (PCK.1, PCK.15), per-keypoint hit rates averaged as the reference does.  (The batch transfer functions
    accumulate the same counts on the device; this is the stand-alone form of the reference's helper.)"""
    err = (kps_pred[..., :2] - kps_gt[..., :2]).norm(dim=-1) * ((1 + 2 * padding_frac) / 2)
    seen = kps_vis != 0
    n = seen.sum(0).double()
    return tuple(float((((err < t) & seen).sum(0).double() / n).mean()) for t in thresholds)


class PCKCounters:
    """Device-side accumulators of test_kp.py:253-258, 317-323: int32 [3,K] = (visible, err < 0.1, err < 0.15) per keypoint,
    added into by every transfer call that is given a ground truth; .pck() -> (PCK.1, PCK.15)."""

    def __init__(self, K, device, padding_frac=0.05, thresholds=(0.1, 0.15)):
        self.counts = torch.zeros(3, K, dtype=torch.int32, device=device)
        self.padding_frac, self.thresholds = float(padding_frac), thresholds

    def pck(self):
        c = self.counts.double()
        return float((c[1] / c[0]).mean()), float((c[2] / c[0]).mean())


def map_kp_flow_batch(kp_src, flow_src, flow_tgt, image_size=256, sigma=3, kp_gt=None, vis=None, counters=None):
    """test_kp.py:125-158 for `pairs` (source, target) entries at once: kp_src [P,K,>=2] in [-1,1], flow_* [P,F,T,T,2] ->
    (k2k [P,K,2], face index [P,K] int32).  Two launches (umr_kp_flow_transfer): per-face scores of every keypoint heat map
    and the face's image point, then arg-max + gather (+ the PCK counters when kp_gt [P,K,>=2], vis [P,K] and a PCKCounters
    are given)."""
    from . import _lib
    L = _lib.lib()
    dev = flow_src.device
    kp = kp_src.to(dev, torch.float32).contiguous()
    fs, ft = flow_src.to(torch.float32).contiguous(), flow_tgt.to(torch.float32).contiguous()
    P, K = kp.shape[0], kp.shape[1]
    _check_pck_args(K, kp_gt, vis, counters)
    F = fs.shape[1]
    TT = fs[0, 0].numel() // 2
    face_idx = torch.empty(P, K, dtype=torch.int32, device=dev)
    k2k = torch.empty(P, K, 2, dtype=torch.float32, device=dev)
    ws = torch.empty(L.umr_kp_flow_workspace_bytes(P, K, F), dtype=torch.uint8, device=dev)
    gt = kp_gt.to(dev, torch.float32).contiguous() if kp_gt is not None else None
    v = vis.to(dev, torch.float32).contiguous() if vis is not None else None
    pf, ta, tb = (counters.padding_frac, counters.thresholds[0], counters.thresholds[1]) if counters is not None else (0.05, 0.1, 0.15)
    _lib.check(L.umr_kp_flow_transfer(_lib.ptr(kp), kp.shape[2], _lib.ptr(fs), _lib.ptr(ft), _lib.ptr(gaussian_patch(sigma, dev)),
                                      _lib.ptr(face_idx), _lib.ptr(k2k), _lib.ptr(gt), gt.shape[2] if gt is not None else 0, _lib.ptr(v),
                                      _lib.ptr(counters.counts) if counters is not None else None, P, K, F, TT, int(image_size),
                                      int(sigma), pf, ta, tb, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "umr_kp_flow_transfer")
    return k2k, face_idx


def map_kp_flow(kp_src, flow_src, flow_tgt, image_size=256, sigma=3):
    """test_kp.py:125-158 (flow mode), one pair.  kp_src [K,>=2] in [-1,1]; flow_* [F,T,T,2] -> transferred keypoints [K,2]:
    keypoint -> face (arg-max of the keypoint heat map sampled at the source flow) -> target pixel (mean of the
    coordinate grid sampled at the target flow)."""
    return map_kp_flow_batch(kp_src[None], flow_src[None], flow_tgt[None], image_size, sigma)[0][0]


def map_kp_cam_batch(kp_src, cam_src, cam_tgt, mask_tgt, mean_shape, image_size=256, kp_gt=None, vis=None, counters=None):
    """test_kp.py:160-193 for `pairs` entries at once: kp_src [P,K,>=2], cam_* [P,7], mask_tgt [P,S,S], mean_shape [V,3] ->
    (k2k [P,K,2], nearest-vertex index [P,K] int32).  Four launches: two projections, nearest foreground pixel of every
    projected vertex, keypoint -> vertex -> pixel."""
    from . import _lib
    L = _lib.lib()
    dev = mean_shape.device
    P, K = kp_src.shape[0], kp_src.shape[1]
    _check_pck_args(K, kp_gt, vis, counters)
    ms = mean_shape.view(1, -1, 3).expand(P, -1, -1).contiguous()
    V = ms.shape[1]
    v_tgt = UF.project_points(ms, cam_tgt.view(P, 7).contiguous(), 2, 0.0).contiguous()
    v_src = UF.project_points(ms, cam_src.view(P, 7).contiguous(), 2, 0.0).contiguous()
    kp = kp_src.to(dev, torch.float32).contiguous()
    mask = mask_tgt.to(dev, torch.float32).contiguous()
    vert_idx = torch.empty(P, K, dtype=torch.int32, device=dev)
    pix = torch.empty(P, V, dtype=torch.int32, device=dev)
    k2k = torch.empty(P, K, 2, dtype=torch.float32, device=dev)
    gt = kp_gt.to(dev, torch.float32).contiguous() if kp_gt is not None else None
    v = vis.to(dev, torch.float32).contiguous() if vis is not None else None
    pf, ta, tb = (counters.padding_frac, counters.thresholds[0], counters.thresholds[1]) if counters is not None else (0.05, 0.1, 0.15)
    _lib.check(L.umr_kp_cam_transfer(_lib.ptr(kp), kp.shape[2], _lib.ptr(v_src), _lib.ptr(v_tgt), _lib.ptr(mask), _lib.ptr(vert_idx),
                                     _lib.ptr(pix), _lib.ptr(k2k), _lib.ptr(gt), gt.shape[2] if gt is not None else 0, _lib.ptr(v),
                                     _lib.ptr(counters.counts) if counters is not None else None, P, K, V, int(image_size), pf, ta, tb,
                                     _lib.stream_ptr(dev)), "umr_kp_cam_transfer")
    return k2k, vert_idx


def map_kp_cam(kp_src, cam_src, cam_tgt, mask_tgt, mean_shape, image_size=256):
    """test_kp.py:160-193 (cam mode), one pair: keypoint -> nearest projected template vertex under the source camera ->
    that vertex under the target camera -> nearest foreground pixel of the target mask."""
    return map_kp_cam_batch(kp_src[None], cam_src.view(1, 7), cam_tgt.view(1, 7), mask_tgt[None], mean_shape, image_size)[0][0]
