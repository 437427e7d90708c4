"""GPU box, ONE fresh process per call: MeshNet + discriminator forward / backward / Adam for 40 steps on the bench's inputs
with a surrogate loss (no render-and-compare kernels at all).  argv[1] = "hip" | "torch": which 2x up-sampling the decoder
uses.  Prints one line: the first non-finite step (or none) -- to tell whether the sporadic NaN of the full bench needs this
repo's kernels at all."""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as Fn
from umr_amd import model as M
from umr_amd.synthetic import make_s1_inputs

mode = sys.argv[1] if len(sys.argv) > 1 else "hip"
if mode == "torch":
    M.Upsample2x.forward = lambda self, x: Fn.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.manual_seed(1234)
tv, faces, outputs, batch = make_s1_inputs(16, 256, 3, seed=100, device=dev)
opts = M.default_opts(subdivide=3, batch_size=16)
net = M.MeshNet((256, 256), opts, nz_feat=opts.nz_feat).to(dev)
disc = M.Discriminator(opts.grl_wt, img_size=256).to(dev)
params = [p for p in list(net.parameters()) + list(disc.parameters()) if p.requires_grad]
opt = torch.optim.Adam(params, lr=opts.learning_rate, betas=(opts.beta1, 0.999), fused=True)
mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1)
std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)
x = (batch["imgs"] - mean) / std
hist = []
for it in range(int(os.environ.get("STEPS", 40))):
    opt.zero_grad(set_to_none=True)
    out = net(x)
    # surrogate: every head contributes, magnitudes comparable to the real step's gradients
    loss = (out["delta_v"] ** 2).mean() + (out["cam"] ** 2).mean() * 0.1 + (out["tex_flow"] ** 2).mean()
    fake = torch.sigmoid(out["tex_flow"].mean(dim=(1, 2, 3, 4), keepdim=False)).view(-1, 1, 1, 1).expand(-1, 1, 256, 256)
    loss = loss + disc(torch.cat([batch["masks"].unsqueeze(1), fake], 0)).mean() * 0.1
    loss.backward()
    opt.step()
    hist.append(loss.detach())
torch.cuda.synchronize()
h = torch.stack(hist).tolist()
bad = [i for i, v in enumerate(h) if not (v == v and abs(v) < 1e30)]
print(json.dumps({"mode": mode, "first_bad_step": bad[0] if bad else None, "last": h[-1]}))
