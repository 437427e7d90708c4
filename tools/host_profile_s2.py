"""GPU box: cProfile of the train_s2 render-and-compare path (host side, network excluded)."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from umr_amd.synthetic import make_s2_inputs
from umr_amd.train_step import RenderCompareS2
dev = torch.device("cuda:0")
K = 8
tv, faces, out, batch, ex = make_s2_inputs(16, K, 256, 3, seed=100, device=dev)
rc = RenderCompareS2(tv.to(dev), faces.to(dev), ex["part_vertex_ids"], ex["uv_img"], ex["uv_sampler"], 256, K,
                     texture_loss_type="l1").to(dev)
leaves = [v for v in out.values() if torch.is_tensor(v) and v.requires_grad]


def step():
    for l in leaves:
        l.grad = None
    total, _ = rc(out, batch)
    total.backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10):
    step()
h = time.perf_counter() - t0
torch.cuda.synchronize()
print("host ms/step", 1e3 * h / 10, "wall", 1e3 * (time.perf_counter() - t0) / 10, flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    step()
pr.disable(); torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(24)
    print("\n".join(l[:140] for l in s.getvalue().split("\n")[4:36]))
