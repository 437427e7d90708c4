"""GPU box: does any kernel of the full step read memory it did not write?  Fill most of the HBM with a poison pattern
(NaN, or argv[1] as a float), give it back to the driver, and run the bench's measurement in the same process: fresh
allocations then land on poisoned pages, and an uninitialised read that matters shows up at once instead of once in
twenty runs."""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from umr_amd.model import build_training_step
from umr_amd.synthetic import make_s1_inputs

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
poison = float(sys.argv[1]) if len(sys.argv) > 1 else float("nan")
gb = int(os.environ.get("POISON_GB", 64))
blocks = []
for _ in range(gb):
    blocks.append(torch.full((256 * 1024 * 1024,), poison, device=dev))      # 1 GiB of fp32 each
torch.cuda.synchronize()
del blocks
if os.environ.get("KEEP_CACHED", "0") != "1":
    torch.cuda.empty_cache()      # back to the driver (which may scrub pages); KEEP_CACHED=1: the caching allocator carves
                                  # every later torch.empty out of the poisoned blocks
args = argparse.Namespace(batch=16, image_size=256, subdivide=3, epoch=0)
torch.manual_seed(1234)
tv, faces, outputs, batch = make_s1_inputs(16, 256, 3, seed=100, device=dev)
step = build_training_step(tv, faces, args, dev, 1)
hist = [step() for _ in range(int(os.environ.get("STEPS", 40)))]
torch.cuda.synchronize()
h = torch.stack([x.reshape(()) for x in hist]).tolist()
bad = [i for i, v in enumerate(h) if not (v == v and abs(v) < 1e30)]
print(json.dumps({"poison": str(poison), "gb": gb, "first_bad_step": bad[0] if bad else None, "first": h[0], "last": h[-1]}))
