"""Driver for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs):
raster forward+backward at N=128, F=1280, IS=512, TS=36 plus a calibration kernel with a KNOWN byte count and the
same per-lane access width (k_iou_partial: reads predict+target once with dword loads, writes nothing)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import bench
from umr_amd import functional as UF
print(bench(128, 3, 512, 36, iters=2))
p = torch.rand(128, 512 * 512, device="cuda")
t = torch.rand(128, 512 * 512, device="cuda")
for _ in range(3):
    UF.NegIoUFunction.apply(p, t)
x = torch.rand(64 * 1024 * 1024, device="cuda")   # 256 MB float4-vectorised copy: the guide's reference pattern
for _ in range(3):
    y = x.clone()
torch.cuda.synchronize()
