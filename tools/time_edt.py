import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umr_amd.image_utils import compute_dt_barrier
for B, H in ((16, 256), (128, 256), (16, 512)):
    m = (torch.rand(B, H, H, device="cuda") > 0.5).float()
    m[:, H // 4: 3 * H // 4, H // 4: 3 * H // 4] = 1; m[:, :H // 8] = 0
    for _ in range(3): compute_dt_barrier(m)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): compute_dt_barrier(m)
    e1.record(); torch.cuda.synchronize()
    print(B, H, "us per call", e0.elapsed_time(e1) * 1e3 / 20)
