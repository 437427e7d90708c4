"""Checkpoint and mesh IO in the reference's formats.

* checkpoints: `torch.save(state_dict)` to `<dir>/<network_label>_net_<epoch_label>.pth` with UN-PREFIXED keys --
  the reference strips DataParallel's `.module` before saving (nnutils/train_utils.py:106-115) and loads tolerantly,
  skipping buffers whose batch dimension differs (nnutils/test_utils.py:106-116, experiments/test_kp.py:101-113).
* meshes: Wavefront OBJ + MTL + PNG texture atlas exactly as `sr.functional.save_obj` writes them
  (external/SoftRas/soft_renderer/functional/save_obj.py:38-93; `sr.Mesh.save_obj(path, save_texture=True)` at
  experiments/train_s1.py:370, train_s2.py:454, avg_uv.py:252,303).  The atlas is baked on the GPU
  (csrc/atlas.hip); the PNG is encoded here with zlib (the reference used skimage.io.imsave, absent in this image).
"""
import ctypes
import os
import struct
import zlib

import numpy as np
import torch

from . import _lib


def _unwrap(net):
    return net.module if hasattr(net, "module") else net        # DDP / DataParallel


def save_network(network, network_label, epoch_label, save_dir):
    """nnutils/train_utils.py:106-115."""
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, '{}_net_{}.pth'.format(network_label, epoch_label))
    torch.save({k: v.detach().cpu() for k, v in _unwrap(network).state_dict().items()}, path)
    return path


def load_network(network, network_label, epoch_label, save_dir, skip=("uv_sampler", "noise"), strict=True):
    """nnutils/train_utils.py:117-125 with the tolerant filtering of test_utils.py:106-116: keys in `skip` (buffers
    that depend on the batch size) are left at their current values.  umr_amd.model names its modules as the reference
    does (cub_mesh.py), so a reference checkpoint loads key for key.  Anything else that cannot be placed -- a key this
    model lacks, a key the file lacks, a shape mismatch -- raises when `strict` (the reference's filter would silently
    leave the model at its random initialisation); strict=False restores the caffe-like behaviour and returns what was
    loaded.  Returns the sorted list of loaded keys."""
    path = os.path.join(save_dir, '{}_net_{}.pth'.format(network_label, epoch_label))
    state = torch.load(path, map_location="cpu")
    net = _unwrap(network)
    own = net.state_dict()
    state = {k[len("module."):] if k.startswith("module.") else k: v for k, v in state.items()}
    skipped = lambda k: any(s in k for s in skip) or k.endswith("num_batches_tracked")
    use = {k: v for k, v in state.items() if k in own and not skipped(k) and own[k].shape == v.shape}
    bad = sorted(k for k in state if k not in use and not skipped(k)) + \
        sorted("(missing) " + k for k in own if k not in state and not skipped(k))
    if bad and strict:
        raise RuntimeError("load_network(%s): %d of %d entries not loaded (unknown key / shape mismatch / missing): %s%s"
                           % (path, len(bad), len(state), ", ".join(bad[:8]), " ..." if len(bad) > 8 else ""))
    own.update(use)
    net.load_state_dict(own)
    return sorted(use)


def texture_atlas(textures, texture_res=16, want=("image", "u8", "uv")):
    """GPU atlas bake.  textures [F, R*R, 3] float32 on the GPU -> dict of device tensors:
    image [H,W,3] f32 (un-flipped, as create_texture_image_cuda returns it), u8 [H,W,3] uint8 (clipped, x255,
    rows reversed: the array save_obj.py:50-53 hands to imsave), uv [F,3,2] (vt coordinates)."""
    if not (_lib.on_device(textures) and textures.dtype == torch.float32 and textures.dim() == 3 and textures.shape[2] == 3):
        raise RuntimeError("texture_atlas: textures must be a float32 [F, R*R, 3] GPU tensor")
    textures = textures.detach().contiguous()
    nf, r_in = textures.shape[0], int(round(textures.shape[1] ** 0.5))
    if r_in * r_in != textures.shape[1]:
        raise RuntimeError("texture_atlas: texel count %d is not a square" % textures.shape[1])
    h, w = ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.lib().umr_texture_atlas_shape(nf, int(texture_res), ctypes.byref(h), ctypes.byref(w)),
               "umr_texture_atlas_shape")
    dev = textures.device
    out = {}
    if "image" in want:
        out["image"] = torch.empty(h.value, w.value, 3, device=dev)
    if "u8" in want:
        out["u8"] = torch.empty(h.value, w.value, 3, dtype=torch.uint8, device=dev)
    if "uv" in want:
        out["uv"] = torch.empty(nf, 3, 2, device=dev)
    p = lambda k: _lib.ptr(out[k]) if k in out else None
    _lib.check(_lib.lib().umr_texture_atlas(_lib.ptr(textures), p("image"), p("u8"), p("uv"), nf, r_in,
                                            int(texture_res), 1e-5, _lib.stream_ptr(dev)), "umr_texture_atlas")
    return out


def create_texture_image(textures, texture_res=16):
    """functional/save_obj.py:9-35: returns (image [H,W,3] float32 numpy, rows reversed; vt [F,3,2] numpy)."""
    o = texture_atlas(textures, texture_res, want=("image", "uv"))
    return o["image"].cpu().numpy()[::-1, ::1], o["uv"].cpu().numpy()


def write_png(filename, rgb):
    """Minimal 8-bit RGB PNG encoder (IHDR / one zlib IDAT with filter 0 / IEND)."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, c = rgb.shape
    assert c == 3
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rgb.reshape(h, w * 3)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(filename, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def save_obj(filename, vertices, faces, textures=None, texture_res=16, texture_type='surface'):
    """functional/save_obj.py:38-93, byte-for-byte the same OBJ / MTL text.  vertices [V,3], faces [F,3] (0-based),
    textures [F,R*R,3] on the GPU for 'surface' (baked into <name>.png next to <name>.mtl) or [V,3] for 'vertex'."""
    assert vertices.dim() == 2 and faces.dim() == 2
    assert texture_type in ('surface', 'vertex') and texture_res >= 2
    base = os.path.basename(filename)
    filename_mtl, filename_tex = filename[:-4] + '.mtl', filename[:-4] + '.png'
    uv = None
    if textures is not None and texture_type == 'surface':
        o = texture_atlas(textures, texture_res, want=("u8", "uv"))
        write_png(filename_tex, o["u8"].cpu().numpy())
        uv = o["uv"].cpu().numpy().reshape(-1, 2)
    v = vertices.detach().cpu().numpy()
    f = faces.detach().cpu().numpy()
    with open(filename, 'w') as fh:
        fh.write('# %s\n#\n\n' % base)
        if textures is not None:
            fh.write('mtllib %s\n\n' % os.path.basename(filename_mtl))
        if textures is not None and texture_type == 'vertex':
            c = textures.detach().cpu().numpy()
            fh.write(''.join('v %.8f %.8f %.8f %.8f %.8f %.8f\n' % (p[0], p[1], p[2], q[0], q[1], q[2])
                             for p, q in zip(v, c)))
        else:
            fh.write(''.join('v %.8f %.8f %.8f\n' % (p[0], p[1], p[2]) for p in v))
        fh.write('\n')
        if uv is not None:
            fh.write(''.join('vt %.8f %.8f\n' % (t[0], t[1]) for t in uv))
            fh.write('\nusemtl material_1\n')
            fh.write(''.join('f %d/%d %d/%d %d/%d\n' % (t[0] + 1, 3 * i + 1, t[1] + 1, 3 * i + 2, t[2] + 1, 3 * i + 3)
                             for i, t in enumerate(f)))
            fh.write('\n')
        else:
            fh.write(''.join('f %d %d %d\n' % (t[0] + 1, t[1] + 1, t[2] + 1) for t in f))
    if uv is not None:
        with open(filename_mtl, 'w') as fh:
            fh.write('newmtl material_1\nmap_Kd %s\n' % os.path.basename(filename_tex))
