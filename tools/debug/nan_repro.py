"""GPU box: the bench's full train_s1 loop WITHOUT per-step host syncs, several trials; reports the step at which the loss
turns non-finite (checked after the fact from a per-step device-side log).  Switches: argv flags eager_cos / split_sil."""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from umr_amd.model import build_training_step
from umr_amd.synthetic import template
import umr_amd.perceptual as PC
from umr_amd.smr import SoftRenderer

flags = set(sys.argv[1:])
if "eager_cos" in flags:
    def eager(f0s, f1s, eps=1e-10):
        val = 0
        for in0, in1 in zip(f0s, f1s):
            n0 = in0 / (torch.sqrt(torch.sum(in0 ** 2, dim=1, keepdim=True)) + eps)
            n1 = in1 / (torch.sqrt(torch.sum(in1 ** 2, dim=1, keepdim=True)) + eps)
            val = val + (1. - torch.mean(torch.mean(torch.sum(n0 * n1, dim=1), dim=1), dim=1))
        return val
    PC.cos_sim_distance = eager
if "split_sil" in flags:
    orig = SoftRenderer.silhouettes
    def sil(self, v, f, cams):
        if cams.shape[0] == 2 * v.shape[0]:
            a, b = orig(self, v, f, cams[0::2].contiguous()), orig(self, v, f, cams[1::2].contiguous())
            return torch.stack((a, b), 1).reshape(cams.shape[0], a.shape[1], a.shape[2])
        return orig(self, v, f, cams)
    SoftRenderer.silhouettes = sil

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
args = argparse.Namespace(batch=16, image_size=256, subdivide=3, epoch=0)
tv, faces = template(3)
for trial in range(int(os.environ.get("TRIALS", "6"))):
    torch.manual_seed(1234)
    step = build_training_step(tv, faces, args, dev, 1)
    log = torch.zeros(64, device=dev)
    t0 = time.perf_counter()
    for i in range(50):
        loss = step()
        log[i] = loss            # device-side copy, no host sync
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    l = log[:50].cpu()
    bad = (~torch.isfinite(l)).nonzero().flatten()
    print("flags %s trial %d: first non-finite step %s, loss[0..3] %s last %.5f, %.1f ms/step" % (
        sorted(flags), trial, int(bad[0]) if len(bad) else None, [round(float(x), 4) for x in l[:4]], float(l[49]), 1e3 * dt / 50), flush=True)
