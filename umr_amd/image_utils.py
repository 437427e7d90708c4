"""utils/image.py pieces that sit on the training step's critical path, on the GPU."""
import torch

from . import _lib
from ._lib import ptr


def compute_dt_barrier(mask, k=50, return_squared=False):
    """utils/image.py:130-141 for a batch: mask [B,H,W] (or [H,W]) float, non-zero = foreground ->
    sigmoid(k * (EDT(1-mask) - EDT(mask)) / max(H,W)), same shape, float32.  The reference calls scipy on the host per
    image per step (experiments/train_s1.py:172, train_s2.py:196); this is two small kernels on the current stream."""
    L = _lib.lib()
    squeeze = mask.dim() == 2
    m = mask.detach().to(torch.float32).contiguous()
    if squeeze:
        m = m.unsqueeze(0)
    B, H, W = m.shape
    out = torch.empty_like(m)
    so = torch.empty(B, H, W, dtype=torch.int32, device=m.device) if return_squared else None
    si = torch.empty(B, H, W, dtype=torch.int32, device=m.device) if return_squared else None
    nb = L.umr_dt_barrier_workspace_bytes(B, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=m.device)
    _lib.check(L.umr_dt_barrier(ptr(m), ptr(out), ptr(so), ptr(si), B, H, W, float(k), ptr(ws), nb,
                                _lib.stream_ptr(m.device)), "umr_dt_barrier")
    if squeeze:
        out = out[0]
    return (out, so, si) if return_squared else out
