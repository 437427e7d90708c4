"""Checkpoint and mesh IO in the reference's formats.

* checkpoints: `torch.save(state_dict)` to `<dir>/<network_label>_net_<epoch_label>.pth` with UN-PREFIXED keys --
  the reference strips DataParallel's `.module` before saving (nnutils/train_utils.py:106-115) and loads tolerantly,
  skipping buffers whose batch dimension differs (nnutils/test_utils.py:106-116, experiments/test_kp.py:101-113).
* meshes: Wavefront OBJ as `sr.functional.save_obj` writes it without textures
  (external/SoftRas/soft_renderer/functional/save_obj.py:30-46: 'v x y z' lines, 1-based 'f a b c').
"""
import os

import torch


def _unwrap(net):
    return net.module if hasattr(net, "module") else net        # DDP / DataParallel


def save_network(network, network_label, epoch_label, save_dir):
    """nnutils/train_utils.py:106-115."""
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, '{}_net_{}.pth'.format(network_label, epoch_label))
    torch.save({k: v.detach().cpu() for k, v in _unwrap(network).state_dict().items()}, path)
    return path


def load_network(network, network_label, epoch_label, save_dir, skip=("uv_sampler", "noise")):
    """nnutils/train_utils.py:117-125 with the tolerant filtering of test_utils.py:106-116: keys in `skip` (buffers
    that depend on the batch size) and keys whose shapes differ are left at their current values.
    Returns the list of keys that were loaded."""
    path = os.path.join(save_dir, '{}_net_{}.pth'.format(network_label, epoch_label))
    state = torch.load(path, map_location="cpu")
    net = _unwrap(network)
    own = net.state_dict()
    use = {k[len("module."):] if k.startswith("module.") else k: v for k, v in state.items()}
    use = {k: v for k, v in use.items() if k in own and not any(s in k for s in skip) and own[k].shape == v.shape}
    own.update(use)
    net.load_state_dict(own)
    return sorted(use)


def save_obj(filename, vertices, faces):
    """functional/save_obj.py without textures.  vertices [V,3], faces [F,3] (0-based)."""
    v = vertices.detach().cpu().numpy()
    f = faces.detach().cpu().numpy()
    with open(filename, 'w') as fh:
        fh.write('# %s\n\n' % os.path.basename(filename))
        for p in v:
            fh.write('v %.8f %.8f %.8f\n' % (p[0], p[1], p[2]))
        fh.write('\n')
        for t in f:
            fh.write('f %d %d %d\n' % (t[0] + 1, t[1] + 1, t[2] + 1))
