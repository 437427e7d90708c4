"""MeshNet re-hosted for the MI355X training harness.

The network is OUT OF SCOPE as kernels (dense conv / GEMM -> PyTorch-ROCm, MIOpen
and hipBLASLt, i.e. MFMA); what matters is that its forward() keeps the reference's signature so the
render-and-compare step is a drop-in (nnutils/cub_mesh.py:450-485):

    MeshNet(input_shape, opts).forward(img [B,3,256,256]) -> dict(cam, cam_probs, cam_sample_inds, mean, logvar,
        noise, tex_flow [B,F,T,T,2], uvimage_pred [B,2,128,256], delta_v [B,V',3], ...)
    .symmetrize(V), .get_mean_shape(), .faces, .uv_sampler, .mean_v

torchvision is not available here, so the ResNet-18 trunk (cub_mesh.py:53-74) is written out in plain torch.nn
with random initialisation (the pretrained weights are not obtainable offline).  Parameter shapes follow the
reference, so the gradient all-reduce volume (~85 M fp32) is the reference's.
"""
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import mesh as mesh_utils


def default_opts(**kw):
    """The absl flags of nnutils/cub_mesh.py:29-48 + train_utils.py as plain attributes."""
    o = dict(symmetric=True, symmetric_texture=True, multiple_cam_hypo=False, nz_feat=200, z_dim=350, num_hypo_cams=8,
             use_texture=True,
             tex_size=6, subdivide=3, batch_size=16, gpu_num=1, scale_lr_decay=0.05, scale_bias=1.0, pred_cam=True,
             learning_rate=1e-4, beta1=0.9, grl_wt=0.2)
    o.update(kw)
    return SimpleNamespace(**o)


# ------------------------------------------------------------------ nnutils/net_blocks.py
def net_init(net):
    """net_blocks.py:225-252."""
    for m in net.modules():
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            m.weight.data.normal_(0, 0.02)
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


def fc(batch_norm, nc_inp, nc_out):
    if batch_norm:
        return nn.Sequential(nn.Linear(nc_inp, nc_out, bias=True), nn.BatchNorm1d(nc_out), nn.LeakyReLU(0.2, inplace=True))
    return nn.Sequential(nn.Linear(nc_inp, nc_out), nn.LeakyReLU(0.1, inplace=True))


def fc_stack(nc_inp, nc_out, nlayers, use_bn=True):
    mods = []
    for _ in range(nlayers):
        mods.append(fc(use_bn, nc_inp, nc_out))
        nc_inp = nc_out
    enc = nn.Sequential(*mods)
    net_init(enc)
    return enc


class _ConvStagedBiasGrad(torch.autograd.Function):
    """F.conv2d whose backward takes the input / weight gradients from the library's convolution backward and sums the BIAS
    gradient itself, in stages (over W, then H, then N)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, groups):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, padding, dilation, groups)
        return F.conv2d(x, weight, bias, stride, padding, dilation, groups)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.cfg
        g = g.contiguous()
        gi, gw, _ = torch.ops.aten.convolution_backward(g, x, weight, None, list(stride), list(padding), list(dilation), False, [0, 0], groups,
                                                        [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        gb = g.sum(3).sum(2).sum(0) if ctx.needs_input_grad[2] else None
        return gi, gw, gb, None, None, None, None


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters, same state_dict keys, same forward values) whose bias gradient over a LARGE output is not left
    to the convolution's backward.  There it is one reduction of N x H x W values per channel, which ATen splits over several
    workgroups that meet at a semaphore zeroed by a cudaMemsetAsync (ATen/native/cuda/Reduce.cuh: global_reduce).  On this
    ROCm stack a memset node of a captured HIP graph does not run again on replay (the same defect the library's own zero-fills
    met in round 2, umr_common.h: umr_k_zero), the semaphore keeps counting, no workgroup is ever 'last' and the bias gradient
    of every replay is whatever the buffer held: measured at bench size, the texture decoder's 64x128 and 128x256 layers --
    values up to 1e38 in a replayed step, eager steps correct (tests/test_gpu_round5.py::
    test_whole_training_step_replays_from_a_hip_graph compares every gradient moment of a replay with the eager step's).
    Summed in stages each reduction is short enough for one workgroup per output: no semaphore, no memset."""
    STAGED_FROM = 16384          # N x H x W per channel from which the bias gradient is summed here

    def forward(self, x):
        if self.bias is not None and x.dim() == 4 and isinstance(self.padding, tuple) and self.padding_mode == "zeros":
            k, st, pd, dl = self.kernel_size, self.stride, self.padding, self.dilation
            ho = (x.shape[2] + 2 * pd[0] - dl[0] * (k[0] - 1) - 1) // st[0] + 1      # output extent as F.conv2d computes it
            wo = (x.shape[3] + 2 * pd[1] - dl[1] * (k[1] - 1) - 1) // st[1] + 1
            if x.shape[0] * ho * wo >= self.STAGED_FROM:
                return _ConvStagedBiasGrad.apply(x, self.weight, self.bias, st, pd, dl, self.groups)
        return self._conv_forward(x, self.weight, self.bias)


def conv2d(batch_norm, cin, cout, kernel_size=3, stride=1):
    layers = [Conv2d(cin, cout, kernel_size=kernel_size, stride=stride, padding=(kernel_size - 1) // 2, bias=True)]
    if batch_norm:
        layers.append(nn.BatchNorm2d(cout))
    layers.append(nn.LeakyReLU(0.2, inplace=True))
    return nn.Sequential(*layers)


class Upsample2x(nn.Module):
    """nn.Upsample(scale_factor=2, mode='bilinear') (net_blocks.py upconv2d); on the GPU the fixed-2x HIP stencil
    replaces PyTorch-ROCm's generic kernel (same values), on CPU (tests of the DDP harness) plain torch runs."""

    def forward(self, x):
        if x.is_cuda:
            from .functional import upsample2x_bilinear
            return upsample2x_bilinear(x)
        return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)


def upconv2d(cin, cout, mode='bilinear'):
    return nn.Sequential(Upsample2x(), nn.ReflectionPad2d(1),
                         Conv2d(cin, cout, kernel_size=3, stride=1, padding=0), nn.LeakyReLU(0.2, inplace=True))


def decoder2d(nlayers, nc_input, nc_final, nc_min=8, use_bn=True):
    """net_blocks.py decoder2d with init_fc=False, use_deconv=False."""
    mods = []
    nc_output = nc_input
    for _ in range(nlayers):
        if nc_output // 2 >= nc_min:
            nc_output = nc_output // 2
        mods.append(upconv2d(nc_input, nc_output))
        nc_input = nc_output
        mods.append(conv2d(use_bn, nc_input, nc_output))
    mods.append(Conv2d(nc_output, nc_final, kernel_size=3, stride=1, padding=1, bias=True))
    dec = nn.Sequential(*mods)
    net_init(dec)
    return dec


# ------------------------------------------------------------------ ResNet-18 trunk (torchvision layout)
class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)), inplace=True)
        out = self.bn2(self.conv2(out))
        return F.relu(out + idt, inplace=True)


class ResNet18(nn.Module):
    """torchvision.models.resnet18 parameter layout (conv1, bn1, layer1..4, fc), so checkpoints interchange with the
    reference's `encoder.resnet_conv.resnet.*` keys.  `fc` is never evaluated (cub_mesh.py:62-73 stops after layer4) and
    never receives a gradient in the reference either; it is frozen here so DDP does not wait for it."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cfg = [(64, 64, 1), (64, 128, 2), (128, 256, 2), (256, 512, 2)]
        for i, (a, b, st) in enumerate(cfg):
            setattr(self, "layer%d" % (i + 1), nn.Sequential(BasicBlock(a, b, st), BasicBlock(b, b, 1)))
        self.fc = nn.Linear(512, 1000)
        for p in self.fc.parameters():
            p.requires_grad = False


class ResNetConv(nn.Module):
    """cub_mesh.py:53-74: the resnet18 trunk up to layer4 (no avgpool / fc)."""

    def __init__(self):
        super().__init__()
        self.resnet = ResNet18()

    def forward(self, x):
        r = self.resnet
        x = F.max_pool2d(F.relu(r.bn1(r.conv1(x)), inplace=True), 3, 2, 1)
        return r.layer4(r.layer3(r.layer2(r.layer1(x))))


class Encoder(nn.Module):
    """cub_mesh.py:77-118."""

    def __init__(self, input_shape, nz_feat=100, z_dim=200):
        super().__init__()
        self.resnet_conv = ResNetConv()
        self.enc_conv1 = conv2d(True, 512, 256, stride=2, kernel_size=4)
        nc_input = 256 * (input_shape[0] // 64) * (input_shape[1] // 64)
        self.enc_fc = fc_stack(nc_input, nz_feat, 2)
        self.mean_fc = nn.Sequential(nn.Linear(nz_feat, nz_feat), nn.LeakyReLU(), nn.Linear(nz_feat, z_dim))
        self.logvar_fc = nn.Sequential(nn.Linear(nz_feat, nz_feat), nn.LeakyReLU(), nn.Linear(nz_feat, z_dim))
        net_init(self.enc_conv1)

    def forward(self, img):
        f = self.enc_conv1(self.resnet_conv(img)).view(img.size(0), -1)
        feat = self.enc_fc(f)
        mean, logvar = self.mean_fc(feat), self.logvar_fc(feat)
        noise = torch.randn_like(mean) * logvar.mul(0.5).exp() + mean   # :103-107
        return feat, noise, mean, logvar


class TexturePredictorUV(nn.Module):
    """cub_mesh.py:120-165 (predict_flow=True)."""

    def __init__(self, nz_feat, num_faces, T, img_H=64, img_W=128, n_upconv=5, nc_init=256, num_sym_faces=0):
        super().__init__()
        self.feat_H, self.feat_W = img_H // (2 ** n_upconv), img_W // (2 ** n_upconv)
        self.nc_init, self.F, self.T, self.num_sym_faces = nc_init, num_faces, T, num_sym_faces
        self.enc = fc_stack(nz_feat, nc_init * self.feat_H * self.feat_W, 2)
        self.decoder = decoder2d(n_upconv, nc_init, nc_final=2)

    def forward(self, feat, uv_sampler):
        x = self.enc(feat).view(feat.size(0), self.nc_init, self.feat_H, self.feat_W)
        uvimage_pred = torch.tanh(self.decoder(x))
        tex = F.grid_sample(uvimage_pred, uv_sampler, mode='bilinear', padding_mode='zeros', align_corners=True)
        tex = tex.view(tex.size(0), -1, self.F, self.T, self.T).permute(0, 2, 3, 4, 1)
        if self.num_sym_faces:
            tex = torch.cat([tex, tex[:, -self.num_sym_faces:]], 1)   # :159-162
        return tex.contiguous(), uvimage_pred


class _Pred(nn.Module):
    """A Linear named `pred_layer`: ShapePredictor / QuatPredictor / ScalePredictor / TransPredictor of cub_mesh.py:169-233
    all keep their single layer under that name, which is what their state_dict keys look like."""

    def __init__(self, nz, nout):
        super().__init__()
        self.pred_layer = nn.Linear(nz, nout)

    def forward(self, feat):
        return self.pred_layer(feat)


class ShapePredictor(_Pred):
    """cub_mesh.py:169-184."""

    def __init__(self, nz_feat, num_verts):
        super().__init__(nz_feat, num_verts * 3)
        self.pred_layer.weight.data.normal_(0, 0.0001)           # :177


def _freeze(*mods):
    for m in mods:
        for p in m.parameters():
            p.requires_grad = False


class Camera(nn.Module):
    """cub_mesh.py:276-301 -> [quat(4), prob(1), scale(1), trans(2)]; sub-module names as the reference's."""

    def __init__(self, nz, scale_lr=1.0, scale_bias=1.0):
        super().__init__()
        self.fc_layer = fc_stack(nz, nz, 2)
        self.quat_predictor = _Pred(nz, 4)
        self.prob_predictor = nn.Linear(nz, 1)
        self.scale_predictor = _Pred(nz, 1)
        self.trans_predictor = _Pred(nz, 2)
        self.scale_lr, self.scale_bias = scale_lr, scale_bias    # ScalePredictor's (lr, bias), :208-212 defaults 1.0, 1.0
        net_init(self)
        self.init_quat_module()

    def init_quat_module(self):
        self.quat_predictor.pred_layer.bias.data = torch.tensor([1., 0., 0., 0.])   # initialize_to_zero_rotation (:200-203)

    def forward(self, feat):
        f = self.fc_layer(feat)
        quat = F.normalize(self.quat_predictor(f))
        scale = F.relu(self.scale_lr * self.scale_predictor(f) + self.scale_bias) + 1e-12     # ScalePredictor (:213-216)
        return torch.cat([quat, self.prob_predictor(f), scale, self.trans_predictor(f)], dim=1)


class MultiCamPredictor(nn.Module):
    """cub_mesh.py:303-362: K camera hypotheses + multinomial sampling."""

    def __init__(self, nz_feat, num_cams=8):
        super().__init__()
        self.fc = fc_stack(nz_feat, nz_feat, 2, use_bn=False)
        # shared scale / translation / probability / quaternion heads: the reference builds them, evaluates the first two and
        # throws the result away (:345-348), never touches the others -- none ever receives a gradient.  Kept (frozen)
        # so the state_dict has the reference's keys.
        self.scale_predictor = _Pred(nz_feat, 1)
        self.trans_predictor = _Pred(nz_feat, 2)
        self.prob_predictor = nn.Linear(nz_feat, num_cams)
        self.camera_predictor = nn.ModuleList([Camera(nz_feat) for _ in range(num_cams)])
        self.quat_predictor = _Pred(nz_feat, 4)
        net_init(self)
        for c in self.camera_predictor:
            c.init_quat_module()
        self.quat_predictor.pred_layer.bias.data = torch.tensor([1., 0., 0., 0.])
        _freeze(self.scale_predictor, self.trans_predictor, self.prob_predictor, self.quat_predictor)
        self.num_cams = num_cams
        # :326-332 (registered, never read by forward)
        base_rot, bias = (0.9239, 0., 0.3827, 0.), [(0.7071, 0.7071, 0., 0.)]
        for _ in range(1, num_cams):
            (a1, b1, c1, d1), (a2, b2, c2, d2) = base_rot, bias[-1]
            bias.append((a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2, a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
                         a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2, a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2))
        self.register_buffer("cam_biases", torch.tensor(bias, dtype=torch.float32))

    def forward(self, feat):
        f = self.fc(feat)
        cams = torch.stack([c(f) for c in self.camera_predictor], dim=1)      # [B,K,8]
        probs = F.softmax(cams[:, :, 4], dim=1)
        cam = torch.cat([cams[:, :, 5:6], cams[:, :, 6:8], cams[:, :, 0:4], probs.unsqueeze(-1)], dim=2)
        # per-rank RNG stream (:358-359).  torch.multinomial device-asserts on a non-finite probability and the assert takes the
        # whole process down (HSA hardware exception) -- a diverged run must reach the caller's own finite-loss check instead, so
        # non-finite rows are sampled uniformly (finite rows are untouched: nan_to_num is the identity on them)
        inds = torch.multinomial(torch.nan_to_num(probs.detach(), nan=1.0, posinf=1.0, neginf=0.0).clamp_min(1e-30), 1)
        sampled = torch.gather(cam, 1, inds.unsqueeze(-1).expand(-1, 1, 8)).squeeze(1)[:, 0:7]
        return sampled, inds, cam[:, :, 7], cam[:, :, 0:7], cams[:, :, 0:4]


class Discriminator(nn.Module):
    """nnutils/discriminators.py:60-86 with the gradient-reversal layer (:32-57)."""

    class _GRL(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, lam):
            ctx.lam = lam
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            return -ctx.lam * g, None

    def __init__(self, lambda_, in_dim=1, img_size=64):
        super().__init__()
        self.lambda_ = lambda_
        fc_size = int(img_size // 16)
        self.img_conv = Conv2d(in_dim, 32, 3, 2, 1)
        self.convs = nn.Sequential(Conv2d(32, 64, 3, 2, 1), nn.ReLU(True), Conv2d(64, 32, 3, 2, 1), nn.ReLU(True),
                                   Conv2d(32, 32, 3, 2, 1), nn.ReLU(True), Conv2d(32, 1, 1, 1, 0))
        self.fc = nn.Linear(fc_size * fc_size, 1)

    def forward(self, imgs):
        x = Discriminator._GRL.apply(imgs, self.lambda_)
        p = self.convs(F.relu(self.img_conv(x)))
        return self.fc(p.view(imgs.size(0), -1))


def spherical_uv(X):
    """utils/mesh.py get_spherical_coords: points on the unit sphere -> (u,v) in [-1,1]."""
    rad = np.linalg.norm(X, axis=1)
    theta = np.arccos(np.clip(X[:, 2] / rad, -1, 1))
    phi = np.arctan2(X[:, 1], X[:, 0])
    return np.stack([((phi + np.pi) / (2 * np.pi)) * 2 - 1, (theta / np.pi) * 2 - 1], 1)


def compute_uvsampler(verts, faces, tex_size):
    """utils/mesh.py:247-272 -> [F,T,T,2]."""
    a = np.arange(tex_size, dtype=np.float64) / (tex_size - 1)
    coords = np.stack(np.meshgrid(a, a, indexing="ij"), -1).reshape(-1, 2)      # product(alpha, beta)
    vs = verts[faces]
    v2, v0v2, v1v2 = vs[:, 2], vs[:, 0] - vs[:, 2], vs[:, 1] - vs[:, 2]
    samples = np.dstack([v0v2, v1v2]).dot(coords.T) + v2.reshape(-1, 3, 1)
    samples = np.transpose(samples, (0, 2, 1))
    return spherical_uv(samples.reshape(-1, 3)).reshape(-1, tex_size, tex_size, 2)


class MeshNet(nn.Module):
    """nnutils/cub_mesh.py:366-507.  Symmetric about `axis`: delta_v is predicted for the on-plane + right-half
    vertices and mirrored (:487-504); with `symmetric_texture` the texture-flow is predicted for the self-mirrored +
    right faces and copied to the left faces (:417-446, :159-162), faces being ordered by utils/mesh.py:102-195."""

    def __init__(self, input_shape, opts, nz_feat=100, axis=1, temp_path=None):
        super().__init__()
        self.opts = opts
        self.symmetric = opts.symmetric
        verts, faces = mesh_utils.create_sphere(opts.subdivide)
        self.symmetric_texture = bool(getattr(opts, "symmetric_texture", False)) and self.symmetric
        self.num_indept_faces, self.num_sym_faces = faces.shape[0], 0
        if self.symmetric:
            verts, faces, self.num_indept, self.num_sym = mesh_utils.make_symmetric(verts, faces, axis)
            if self.symmetric_texture:
                faces, self.num_indept_faces, self.num_sym_faces = mesh_utils.make_faces_symmetric(
                    verts, faces, self.num_indept, self.num_sym, axis)
            self.num_output = self.num_indept + self.num_sym
            flip = torch.ones(1, 3)
            flip[0, axis] = -1
            self.register_buffer("flip", flip, persistent=False)       # a plain attribute in the reference (:399)
        else:
            self.num_output = verts.shape[0]
        self.register_buffer("mean_v", torch.from_numpy(verts[:self.num_output]).float())
        self.register_buffer("faces", torch.from_numpy(faces).long(), persistent=False)   # attribute in the reference (:409)
        self.verts_np, self.faces_np = verts, faces
        self.encoder = Encoder(input_shape, nz_feat=nz_feat, z_dim=opts.z_dim)
        self.shape_predictor = ShapePredictor(opts.z_dim, self.num_output)
        self.cam_predictor = MultiCamPredictor(nz_feat, opts.num_hypo_cams) if opts.multiple_cam_hypo else Camera(nz_feat)
        T = opts.tex_size
        num_faces = self.num_indept_faces + self.num_sym_faces if self.symmetric_texture else faces.shape[0]   # :418-421
        uv = torch.from_numpy(compute_uvsampler(verts, faces[:num_faces], T)).float().view(1, num_faces, T * T, 2)
        self.register_buffer("uv_sampler", uv)
        img_H = int(2 ** np.floor(np.log2(np.sqrt(num_faces) * T)))                                            # :438
        self.texture_predictor = TexturePredictorUV(nz_feat, num_faces, T, img_H=img_H, img_W=2 * img_H,
                                                    num_sym_faces=self.num_sym_faces if self.symmetric_texture else 0)
        net_init(self.texture_predictor)

    def forward(self, img=None):
        out = {}
        img_feat, noise, mean, logvar = self.encoder(img)
        out['delta_v'] = self.shape_predictor(noise).view(img.size(0), -1, 3)
        if self.opts.multiple_cam_hypo:
            cam, inds, probs, all_cams, quats = self.cam_predictor(img_feat)
            out['cam_hypotheses'], out['base_quats'] = all_cams, quats[:, 0]
        else:
            c = self.cam_predictor(img_feat)
            cam = torch.cat([c[:, 5:6], c[:, 6:8], c[:, 0:4]], dim=1)
            inds = torch.zeros(cam.size(0), 1, dtype=torch.long, device=cam.device)
            probs = inds.float() + 1 + 0 * c[:, 4:5]   # keeps the (unused) prob head in the autograd graph for DDP
        out.update(mean=mean, logvar=logvar, cam_sample_inds=inds, cam_probs=probs, cam=cam, noise=noise, feat=noise)
        tex, uvimg = self.texture_predictor(img_feat, self.uv_sampler.expand(img.size(0), -1, -1, -1))
        out['tex_flow'], out['uvimage_pred'] = tex, uvimg
        return out

    def symmetrize(self, V):
        if not self.symmetric:
            return V
        if V.dim() == 2:
            return torch.cat([V, self.flip * V[-self.num_sym:]], 0)
        return torch.cat([V, self.flip * V[:, -self.num_sym:]], 1)

    def get_mean_shape(self):
        return self.symmetrize(self.mean_v)


# ------------------------------------------------------------------ training step (bench.py / examples)
def _data_parallel(model, net, disc, args, dev, world):
    """-> (net, disc, sync) of a data-parallel step over `world` ranks (nn.DataParallel's sites, train_s2.py:95-101).
    Default (`args.grad_sync` absent or "buckets"): the modules as they are + parallel.BucketedGradSync over all their parameters
    -- the bucketed, overlapped RCCL all-reduce written out, capturable into the step's HIP graph; "ddp": torch's
    DistributedDataParallel wrappers (eager steps only), sync None.  One rank: (net, disc, None)."""
    from .parallel import BucketedGradSync, wrap_ddp
    if world <= 1:
        return net, disc, None
    if getattr(args, "grad_sync", "buckets") == "ddp":
        return wrap_ddp(net, dev, world), wrap_ddp(disc, dev, world), None
    return net, disc, BucketedGradSync(model.parameters(), torch.distributed.get_world_size() if torch.distributed.is_initialized() else world)


def build_training_step(tv, faces, args, dev, world):
    """One full train_s1 iteration on resident synthetic data: MeshNet fwd -> render-and-compare (HIP) -> bwd with
    bucketed RCCL all-reduce overlapped (DDP) -> Adam with the reference's lr schedule (train_utils.py:186-194)."""
    from .image_utils import compute_dt_barrier
    from .synthetic import make_s1_inputs
    from .train_step import RenderCompareS1
    opts = default_opts(subdivide=args.subdivide, batch_size=args.batch)
    net = MeshNet((args.image_size, args.image_size), opts, nz_feat=opts.nz_feat).to(dev)
    disc = Discriminator(opts.grl_wt, img_size=args.image_size).to(dev)
    model = nn.ModuleDict(dict(net=net, disc=disc))
    ddp_net, ddp_disc, sync = _data_parallel(model, net, disc, args, dev, world)
    # train_s1.py:150: the texture term is the AlexNet perceptual distance (two feature passes over B images and one
    # backward per step); random weights offline, frozen (requires_grad False) and therefore not in the optimizer
    from .perceptual import PerceptualTextureLoss
    rc = RenderCompareS1(net.get_mean_shape().detach(), net.faces, args.image_size, discriminator=ddp_disc,
                         texture_loss=PerceptualTextureLoss(dev), epoch=getattr(args, "epoch", 0),
                         share_mask_render=bool(getattr(args, "share_mask_render", 1))).to(dev)
    # same update rule as train_utils.py:186-187; `fused` runs it as one multi-tensor kernel on the GPU
    # capturable: the step is going to be captured into ONE HIP graph (bench.py --graph 1) -- the learning rate and the step
    # counter then live on the device and the schedule below is device arithmetic inside the graph
    capt = bool(getattr(args, "graph", 0)) and torch.device(dev).type == "cuda"
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad],
                           lr=(torch.tensor(opts.learning_rate, device=dev) if capt else opts.learning_rate),
                           betas=(opts.beta1, 0.999), fused=(torch.device(dev).type == "cuda"), capturable=capt)
    it_dev = torch.zeros((), device=dev) if capt else None
    rank = torch.distributed.get_rank() if (world > 1 and torch.distributed.is_initialized()) else 0
    _, _, _, batch = make_s1_inputs(args.batch, args.image_size, args.subdivide, seed=100 + rank, device=dev)
    mean, std = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)
    input_imgs = (batch["imgs"] - mean) / std                         # train_s1.py:128-131,164
    state = dict(it=0)

    def step():
        for g in opt.param_groups:                                    # train_utils.py:194
            if capt:
                g['lr'].copy_(opts.learning_rate / (1 + it_dev * 5e-4))
            else:
                g['lr'] = opts.learning_rate / (1 + state["it"] * 5e-4)
        if sync is None:
            opt.zero_grad(set_to_none=True)
        else:
            sync.begin()
        # the reference computes the barrier distance transform on the host in set_input (train_s1.py:171-174)
        batch["dts_barrier"] = compute_dt_barrier(batch["masks"]).unsqueeze(1)
        out = ddp_net(input_imgs)
        out["pred_vs"] = net.get_mean_shape()[None] + net.symmetrize(out["delta_v"])   # train_s1.py:183-192
        total, _ = rc(out, batch)
        if sync is None:
            total.backward()
        else:       # bucket all-reduces start from the gradient hooks while backward runs; finish() waits for them
            (total * sync.loss_scale).backward()
            sync.finish()
        opt.step()
        state["it"] += 1
        if capt:
            it_dev.add_(1)
        return total.detach()

    step.model = model
    step.opt, step.it_dev = opt, it_dev          # (tests: snapshot / restore the optimizer state around a graph replay)
    step.sync = sync
    return step


def build_training_step_s2(args, dev, world):
    """One full train_s2 iteration (experiments/train_s2.py:201-316, 409-444): MeshNet with K=8 camera hypotheses ->
    22 raster forwards + 21 backwards per image + mask / texture (AlexNet perceptual) / part / chamfer losses ->
    backward with DDP all-reduce -> Adam.  SCOPS data being absent, part labels / points are synthetic."""
    from .image_utils import compute_dt_barrier
    from .synthetic import make_s2_inputs
    from .train_step import RenderCompareS2
    opts = default_opts(subdivide=args.subdivide, batch_size=args.batch, multiple_cam_hypo=True)
    net = MeshNet((args.image_size, args.image_size), opts, nz_feat=opts.nz_feat).to(dev)
    disc = Discriminator(opts.grl_wt, in_dim=3, img_size=args.image_size).to(dev)     # train_s2.py:91-93: rgb input
    model = nn.ModuleDict(dict(net=net, disc=disc))
    ddp_net, ddp_disc, sync = _data_parallel(model, net, disc, args, dev, world)
    rank = torch.distributed.get_rank() if (world > 1 and torch.distributed.is_initialized()) else 0
    _, _, _, batch, ex = make_s2_inputs(args.batch, opts.num_hypo_cams, args.image_size, args.subdivide,
                                        seed=getattr(args, "data_seed", 100) + rank, device=dev)
    rc = RenderCompareS2(net.get_mean_shape().detach(), net.faces, ex["part_vertex_ids"], ex["uv_img"],
                         net.uv_sampler, args.image_size, opts.num_hypo_cams, texture_loss_type="perceptual",
                         discriminator=ddp_disc, tex_size=opts.tex_size,
                         num_sym_faces=net.texture_predictor.num_sym_faces,
                         share_mask_render=bool(getattr(args, "share_mask_render", 1))).to(dev)      # train_s2.py:154-161
    # capturable (bench.py --graph 1): learning rate and step counter live on the device, as in build_training_step
    capt = bool(getattr(args, "graph", 0)) and torch.device(dev).type == "cuda"
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad],
                           lr=(torch.tensor(opts.learning_rate, device=dev) if capt else opts.learning_rate),
                           betas=(opts.beta1, 0.999), fused=(torch.device(dev).type == "cuda"), capturable=capt)
    it_dev = torch.zeros((), device=dev) if capt else None
    mean, std = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)
    input_imgs = (batch["imgs"] - mean) / std
    state = dict(it=0)
    import os
    watch = [] if os.environ.get("UMR_WATCH_TERMS") else None

    def step():
        for g in opt.param_groups:
            if capt:
                g['lr'].copy_(opts.learning_rate / (1 + it_dev * 5e-4))
            else:
                g['lr'] = opts.learning_rate / (1 + state["it"] * 5e-4)
        if sync is None:
            opt.zero_grad(set_to_none=True)
        else:
            sync.begin()
        batch["dts_barrier"] = compute_dt_barrier(batch["masks"]).unsqueeze(1)        # train_s2.py:196
        out = ddp_net(input_imgs)
        out["mean_shape"] = net.get_mean_shape()
        out["pred_vs"] = out["mean_shape"][None] + net.symmetrize(out["delta_v"])
        total, terms = rc(out, batch)
        if watch is not None:       # UMR_WATCH_TERMS=1: per-step loss terms + a few state norms, kept on the device (no sync)
            watch.append((sorted(terms), torch.stack([terms[k].detach().reshape(()) for k in sorted(terms)] +
                                                     [out["delta_v"].detach().abs().max(), out["cam_hypotheses"].detach().abs().max(),
                                                      out["tex_flow"].detach().abs().max()])))
        if sync is None:
            total.backward()
        else:
            (total * sync.loss_scale).backward()
            sync.finish()
        opt.step()
        # :268 -- written IN PLACE: the next step (and the next replay of a captured step) reads this very buffer
        torch.mul(batch["imgs"], batch["masks"].unsqueeze(1), out=batch["random_imgs"])
        state["it"] += 1
        if capt:
            it_dev.add_(1)
        return total.detach()

    step.model = model
    step.opt, step.it_dev = opt, it_dev          # (tests: snapshot / restore the optimizer state around a graph replay)
    step.watch = watch
    step.sync = sync
    return step
