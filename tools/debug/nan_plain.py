"""GPU box: bench.py's measurement loop (fresh model, 40 steps, no host sync, no extra device ops) repeated many times in
one process; prints the per-step loss history of any run that ends non-finite."""
import argparse, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from umr_amd.model import build_training_step
from umr_amd.synthetic import make_s1_inputs

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
RUNS = int(os.environ.get("RUNS", 60))
args = argparse.Namespace(batch=16, image_size=256, subdivide=3, epoch=0)
bad = 0
for run in range(RUNS):
    torch.manual_seed(1234)
    tv, faces, outputs, batch = make_s1_inputs(16, 256, 3, seed=100, device=dev)
    step = build_training_step(tv, faces, args, dev, 1)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    hist = [step() for _ in range(30)]
    torch.cuda.synchronize()
    h = torch.stack([x.reshape(()) for x in hist]).tolist()
    if not all(v == v and abs(v) < 1e30 for v in h):
        bad += 1
        bad_params = [n for n, p in step.model.named_parameters() if not bool(torch.isfinite(p).all())][:8]
        print(json.dumps({"run": run, "history": h, "nonfinite_params_sample": bad_params}), flush=True)
    del step, hist
print("bad runs: %d of %d" % (bad, RUNS), flush=True)
