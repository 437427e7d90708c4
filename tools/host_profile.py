"""GPU box: cProfile of the hot-path-only step (host side), top functions by cumulative and own time."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from umr_amd.synthetic import make_s1_inputs
from umr_amd.train_step import RenderCompareS1
dev = torch.device("cuda:0")
tv, faces, outputs, batch = make_s1_inputs(16, 256, 3, seed=100, device=dev)
rc = RenderCompareS1(tv.to(dev), faces.to(dev), 256).to(dev)
leaves = [outputs["delta_v"], outputs["cam"], outputs["tex_flow"]]


def step():
    for l in leaves:
        l.grad = None
    outputs["pred_vs"] = outputs["mean_shape"][None] + outputs["delta_v"]
    total, _ = rc(outputs, batch)
    total.backward()


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:150] for l in s.getvalue().split("\n")[4:42]))
