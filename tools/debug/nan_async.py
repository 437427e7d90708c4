"""GPU box: the bench's full train_s1 loop, many fresh models, NO host synchronisation inside a run; per step a few tiny
device-side reductions record (asynchronously) whether the loss terms / network outputs / their gradients were finite, so
the first non-finite quantity of a diverging run can be named afterwards without disturbing the timing that provokes it."""
import argparse, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from umr_amd.model import build_training_step
from umr_amd.synthetic import template
import umr_amd.train_step as TS

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
STEPS = int(os.environ.get("STEPS", 45))
RUNS = int(os.environ.get("RUNS", 30))
NAMES = []
rec = {}
orig_forward = TS.RenderCompareS1.forward

def fwd(self, outputs, batch):
    i = rec["i"]
    total, terms = orig_forward(self, outputs, batch)
    keys = sorted(terms.keys())
    if not NAMES:
        NAMES.extend(["total"] + keys + ["out:" + k for k in ("pred_vs", "cam", "tex_flow", "delta_v")] +
                     ["grad:" + k for k in ("pred_vs", "cam", "tex_flow", "delta_v")])
        rec["buf"] = torch.zeros(RUNS, STEPS, len(NAMES), device=dev)
    row = rec["buf"][rec["run"], i]
    row[0] = total.detach()
    for j, k in enumerate(keys):
        row[1 + j] = terms[k].detach() if torch.is_tensor(terms[k]) else float(terms[k])
    base = 1 + len(keys)
    for j, k in enumerate(("pred_vs", "cam", "tex_flow", "delta_v")):
        row[base + j] = outputs[k].detach().abs().max()
        if outputs[k].requires_grad:
            outputs[k].register_hook(lambda g, r=row, c=base + 4 + j: r.__setitem__(c, g.abs().max()))
    return total, terms
TS.RenderCompareS1.forward = fwd

args = argparse.Namespace(batch=16, image_size=256, subdivide=3, epoch=0)
tv, faces = template(3)
bad = 0
for run in range(RUNS):
    torch.manual_seed(int(os.environ.get("SEED", 1234)))   # bench.py seeds 1234 + rank: only run-to-run noise differs
    step = build_training_step(tv, faces, args, dev, 1)
    rec["run"] = run
    for i in range(STEPS):
        rec["i"] = i
        loss = step()
    torch.cuda.synchronize()
    b = rec["buf"][run].cpu()
    fin = torch.isfinite(b)
    if not bool(fin.all()):
        bad += 1
        first = int((~fin).any(1).nonzero()[0])
        cols = [NAMES[c] for c in (~fin[first]).nonzero().flatten().tolist()]
        prev = {NAMES[c]: float(b[first - 1, c]) for c in range(len(NAMES))} if first > 0 else {}
        print(json.dumps({"run": run, "first_bad_step": first, "non_finite": cols, "previous_step": prev}), flush=True)
    else:
        print(json.dumps({"run": run, "ok": True, "final_total": float(b[-1, 0]), "max_pred_vs": float(b[:, NAMES.index("out:pred_vs")].max()),
                          "max_grad_pred_vs": float(b[:, NAMES.index("grad:pred_vs")].max())}), flush=True)
    del step
print("bad runs: %d of %d" % (bad, RUNS))
