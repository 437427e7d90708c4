"""GPU box: isolate the part-render forward of the train_s2 step at bench size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from umr_amd.model import MeshNet, default_opts
from umr_amd.synthetic import make_s2_inputs
from umr_amd import loss_utils
from umr_amd.smr import SoftRenderer
dev = torch.device("cuda:0")
bs, IS = 16, 256
opts = default_opts(subdivide=3, batch_size=bs, multiple_cam_hypo=True)
net = MeshNet((IS, IS), opts, nz_feat=opts.nz_feat).to(dev)
_, _, _, batch, ex = make_s2_inputs(bs, opts.num_hypo_cams, IS, 3, seed=100, device=dev)
fn = loss_utils.part_matching_loss(ex["uv_img"], net.uv_sampler, net.texture_predictor.num_sym_faces, im_size=IS,
                                   batch_size=bs, tex_size=opts.tex_size).to(dev)
verts = net.get_mean_shape()[None].repeat(bs, 1, 1).detach()
cams = batch["cams"] if "cams" in batch else torch.cat([torch.full((bs, 1), 0.8), torch.zeros(bs, 2), torch.tensor([[1., 0, 0, 0]]).repeat(bs, 1)], 1).to(dev)
print("stex", fn.stex1.shape, "faces", net.faces.shape, "verts", verts.shape, flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "part"
r = SoftRenderer(IS, "softmax"); r.ambient_light_only(); r.need_p2f = False
F = net.faces.shape[-2]
if mode == "rand":
    tex = torch.rand(bs, F, 36, 3, device=dev)
else:
    tex = fn.stex1.expand(bs, -1, -1, -1).contiguous()
print("tex", tex.shape, tex.is_contiguous(), flush=True)
faces = net.faces if net.faces.dim() == 3 else net.faces[None].repeat(bs, 1, 1)
out = r(verts, faces, cams, tex)
torch.cuda.synchronize()
print("ok", out[0].shape, float(out[0].sum()), flush=True)
