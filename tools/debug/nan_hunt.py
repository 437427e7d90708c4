"""GPU box: find where the full train_s1 step goes non-finite (many seeds / steps, per-step checks of every loss term
and of the gradients flowing back into the network outputs); dumps the step's inputs when it happens."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from umr_amd.model import build_training_step
from umr_amd.synthetic import template
import umr_amd.train_step as TS

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
orig_forward = TS.RenderCompareS1.forward
last = {}
def fwd(self, outputs, batch):
    hooks = {}
    for k in ("pred_vs", "cam", "tex_flow", "delta_v"):
        t = outputs[k]
        if t.requires_grad:
            t.register_hook(lambda g, k=k: hooks.__setitem__(k, (bool(torch.isfinite(g).all()), float(g.abs().max()))))
    total, terms = orig_forward(self, outputs, batch)
    last.update(terms={k: float(v) for k, v in terms.items()}, total=float(total), hooks=hooks,
                outs={k: outputs[k].detach().clone() for k in ("pred_vs", "cam", "tex_flow", "delta_v")},
                batch={k: v.detach().clone() for k, v in batch.items() if torch.is_tensor(v)})
    return total, terms
TS.RenderCompareS1.forward = fwd

def run(B, H, sub, steps, seed, sync):
    args = argparse.Namespace(batch=B, image_size=H, subdivide=sub, epoch=0)
    tv, faces = template(sub)
    torch.manual_seed(seed)
    step = build_training_step(tv, faces, args, dev, 1)
    for i in range(steps):
        loss = step()
        if sync or i == steps - 1 or i % 10 == 9:
            torch.cuda.synchronize()
            okh = all(v[0] for v in last["hooks"].values())
            okt = all(v == v and abs(v) < 1e30 for v in last["terms"].values())
            if not (okh and okt):
                print("seed %d step %d NON-FINITE total %s" % (seed, i, last["total"]))
                print("  terms", last["terms"]); print("  grad hooks", last["hooks"])
                print("  outs finite", {k: bool(torch.isfinite(v).all()) for k, v in last["outs"].items()},
                      {k: float(v.abs().max()) for k, v in last["outs"].items()})
                torch.save({"outs": {k: v.cpu() for k, v in last["outs"].items()}, "batch": {k: v.cpu() for k, v in last["batch"].items()},
                            "terms": last["terms"], "hooks": last["hooks"]}, "gpurun_out/r2c/nan_dump_seed%d.pt" % seed)
                return False
    print("seed %d: %d steps clean (sync=%s), final total %.5f" % (seed, steps, sync, last["total"]), flush=True)
    return True

for seed in (1234, 1, 2, 3):
    run(16, 256, 3, 60, seed, sync=True)
for seed in (1234, 11, 12):
    run(16, 256, 3, 60, seed, sync=False)
