"""Perceptual distance used by the texture loss (nnutils/perceptual_loss.py:38-57 ->
external/PerceptualSimilarity/models/{dist_model,networks_basic,pretrained_networks}.py, model='net', net='alex').

AlexNet's convolutions are dense contractions and stay on MIOpen (out of scope as kernels);
torchvision is not installed and the pretrained weights cannot be downloaded here, so the feature extractor is
written out (torchvision `alexnet().features` layout, taps after each of the 5 ReLUs,
pretrained_networks.py:59-95) with RANDOM weights unless a state_dict is supplied.  The distance head
(networks_basic.py:42-64 + util/util.py:71-83: sum over taps of 1 - mean_xy cos(f0, f1)) is one HIP launch for all
five taps in each direction (csrc/perceptual.hip) instead of ~8 elementwise / reduction launches per tap."""
import torch
import torch.nn as nn

from . import functional as UF


class AlexNetFeatures(nn.Module):
    def __init__(self):
        super().__init__()
        self.slices = nn.ModuleList([
            nn.Sequential(nn.Conv2d(3, 64, 11, 4, 2), nn.ReLU(inplace=False)),
            nn.Sequential(nn.MaxPool2d(3, 2), nn.Conv2d(64, 192, 5, 1, 2), nn.ReLU(inplace=False)),
            nn.Sequential(nn.MaxPool2d(3, 2), nn.Conv2d(192, 384, 3, 1, 1), nn.ReLU(inplace=False)),
            nn.Sequential(nn.Conv2d(384, 256, 3, 1, 1), nn.ReLU(inplace=False)),
            nn.Sequential(nn.Conv2d(256, 256, 3, 1, 1), nn.ReLU(inplace=False))])
        for p in self.parameters():
            p.requires_grad = False       # requires_grad=False in the reference (networks_basic.py:29)

    def forward(self, x):
        outs = []
        for s in self.slices:
            x = s(x)
            outs.append(x)
        return outs


def cos_sim_distance(feats0, feats1, eps=1e-10):
    """sum over the feature pairs of 1 - util.cos_sim(f0, f1) (networks_basic.py:50-58, util/util.py:71-83) -> [N].
    GPU tensors only (umr_cos_sim_forward / _backward); there is no eager fallback."""
    return UF.cos_sim_distance(eps, feats0, feats1)


class PNet(nn.Module):
    """networks_basic.py:13-64."""

    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1))
        self.register_buffer("scale", torch.tensor([.458, .448, .450]).view(1, 3, 1, 1))
        self.net = AlexNetFeatures()

    def forward(self, in0, in1):
        return self.forward_scaled((in0 - self.shift) / self.scale, (in1 - self.shift) / self.scale)

    def forward_scaled(self, x0, x1):
        """forward() behind the scaling layer (:45-46): x = (in - shift) / scale already applied by the caller."""
        # two passes as in the reference (:43-47); a side that carries no gradient (the ground-truth image) records
        # no autograd graph, so the convolutions' backward runs over the predicted half only
        def taps(x):
            if x.requires_grad:
                return self.net(x)
            with torch.no_grad():
                return self.net(x)
        return cos_sim_distance(taps(x0), taps(x1))


class PerceptualLoss(object):
    """nnutils/perceptual_loss.py:38-57."""

    def __init__(self, device=None, state_dict=None):
        self.model = PNet()
        if state_dict is not None:
            self.model.load_state_dict(state_dict)
        if device is not None:
            self.model.to(device)
        self.model.eval()

    def __call__(self, pred, target, normalize=True):
        if normalize:
            target = 2 * target - 1
            pred = 2 * pred - 1
        return self.model(target, pred)     # forward_pair(target, pred)


class PerceptualTextureLoss(object):
    """nnutils/loss_utils.py:128-150."""

    def __init__(self, device=None):
        self.perceptual_loss = PerceptualLoss(device)

    def __call__(self, img_pred, img_gt, mask_gt, mask_pred=None, avg=True):
        m_pred = mask_gt if mask_pred is None else mask_pred
        if img_pred.is_cuda and img_pred.dim() == 4 and img_pred.shape[1] == 3:
            # image * mask (:141-146), 2 x - 1 (perceptual_loss.py:52-54) and PNet's scaling layer
            # (networks_basic.py:45-46) in one launch per side (umr_perceptual_prologue_*), same rounding sequence
            net = self.perceptual_loss.model
            if getattr(net, "_host_consts", None) is None:     # read the two buffers back once, not once per step
                net._host_consts = (net.shift.flatten().tolist(), net.scale.flatten().tolist())
            sh, sc = net._host_consts
            x_pred = UF.perceptual_prologue(img_pred, m_pred, sh, sc)
            with torch.no_grad():
                x_gt = UF.perceptual_prologue(img_gt, mask_gt, sh, sc)
            dist = net.forward_scaled(x_gt, x_pred)          # forward_pair(target, pred)
        else:
            dist = self.perceptual_loss(img_pred * m_pred.unsqueeze(1), img_gt * mask_gt.unsqueeze(1))
        return dist.mean() if avg else dist
