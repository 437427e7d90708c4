"""ctypes front-end for the CPU raster oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Nothing under umr_amd/ does (tests/test_cabi_and_layout.py::test_product_never_imports_the_oracle enforces it).

Two back-ends with one calling convention (that of the reference binding,
external/SoftRas/soft_renderer/cuda/soft_rasterize_cuda.cpp:62-138):
  * "port"  -> oracle/liboracle.so        (oracle/softras_oracle.c, our C restatement)
  * "ref"   -> oracle/_ref/libsoftras_ref.so  (the reference's own kernel bodies compiled
               for the host by oracle/ref_shim/build_ref.sh; exists only where it was built)

``soft_rasterize`` below restates the reference's autograd wrapper
(functional/soft_rasterize.py:9-125) on CPU tensors so that the whole render-and-compare
path can be differentiated on the host for parity tests.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = ctypes.POINTER(ctypes.c_float)
_libs = {}


def _ptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_F)


def have_backend(name):
    path = {"port": os.path.join(_HERE, "liboracle.so"),
            "ref": os.path.join(_HERE, "_ref", "libsoftras_ref.so")}[name]
    return os.path.exists(path)


def _lib(name):
    if name not in _libs:
        path = {"port": os.path.join(_HERE, "liboracle.so"),
                "ref": os.path.join(_HERE, "_ref", "libsoftras_ref.so")}[name]
        if not os.path.exists(path):
            raise RuntimeError("oracle backend %r not built (%s); run `make -C oracle`" % (name, path))
        _libs[name] = ctypes.CDLL(path)
    return _libs[name]


def max_threads():
    lib = _lib("port")
    lib.oracle_max_threads.restype = ctypes.c_int
    return int(lib.oracle_max_threads())


def raster_forward(faces, textures, image_size, background=(0, 0, 0), near=1.0, far=100.0, eps=1e-3,
                   sigma_val=1e-5, dist_eps_log=None, gamma_val=1e-4, func_id_rgb=1, double_side=True,
                   backend="port", n_threads=1, func_id_dist=2, func_id_alpha=2, texture_sample_type=0):
    """Innermost level.  faces [N,F,9|3x3] f32, textures [N,F,TS,3] f32 (numpy).
    ``dist_eps_log`` is the value the binding receives, log(1/dist_eps - 1).
    Returns dict(faces_info, aggrs_info, p2f_info, p2f_sum, soft_colors) (raw, un-normalised p2f)."""
    faces = np.ascontiguousarray(faces, np.float32).reshape(faces.shape[0], faces.shape[1], 9)
    textures = np.ascontiguousarray(textures, np.float32)
    n, f = faces.shape[:2]
    ts = textures.shape[2]
    IS = int(image_size)
    faces_info = np.zeros((n, f, 27), np.float32)
    aggrs_info = np.zeros((n, 2, IS, IS), np.float32)
    p2f_info = np.zeros((n, f, 2), np.float32)
    p2f_sum = np.zeros((n, f, 2), np.float32)
    soft_colors = np.ones((n, 4, IS, IS), np.float32)
    for k in range(3):
        soft_colors[:, k] *= np.float32(background[k])
    grid = standard_grid(IS)
    args = [_ptr(faces), _ptr(textures), _ptr(faces_info), _ptr(aggrs_info), _ptr(grid), _ptr(p2f_info),
            _ptr(p2f_sum), _ptr(soft_colors), ctypes.c_int(n), ctypes.c_int(f), ctypes.c_int(IS),
            ctypes.c_int(ts), ctypes.c_float(near), ctypes.c_float(far), ctypes.c_float(eps),
            ctypes.c_float(sigma_val), ctypes.c_int(func_id_dist), ctypes.c_float(dist_eps_log),
            ctypes.c_float(gamma_val), ctypes.c_int(func_id_rgb), ctypes.c_int(func_id_alpha),
            ctypes.c_int(texture_sample_type), ctypes.c_int(1 if double_side else 0)]
    lib = _lib(backend)
    if backend == "port":
        rc = lib.oracle_raster_forward(*args, ctypes.c_int(n_threads))
    else:
        rc = lib.ref_forward_soft_rasterize(*args)
    if rc != 0:
        raise RuntimeError("oracle forward rc=%d" % rc)
    return dict(faces=faces, textures=textures, faces_info=faces_info, aggrs_info=aggrs_info,
                p2f_info=p2f_info, p2f_sum=p2f_sum, soft_colors=soft_colors)


def raster_backward(faces, textures, soft_colors, faces_info, aggrs_info, grad_soft_colors, image_size,
                    near=1.0, far=100.0, eps=1e-3, sigma_val=1e-5, dist_eps_log=None, gamma_val=1e-4,
                    func_id_rgb=1, double_side=True, backend="port", n_threads=1, func_id_dist=2,
                    func_id_alpha=2, texture_sample_type=0):
    faces = np.ascontiguousarray(faces, np.float32).reshape(faces.shape[0], faces.shape[1], 9)
    textures = np.ascontiguousarray(textures, np.float32)
    n, f = faces.shape[:2]
    ts = textures.shape[2]
    IS = int(image_size)
    grad_faces = np.zeros((n, f, 9), np.float32)
    grad_textures = np.zeros((n, f, ts, 3), np.float32)
    g = np.ascontiguousarray(grad_soft_colors, np.float32)
    sc = np.ascontiguousarray(soft_colors, np.float32)
    fi = np.ascontiguousarray(faces_info, np.float32)
    ag = np.ascontiguousarray(aggrs_info, np.float32)
    args = [_ptr(faces), _ptr(textures), _ptr(sc), _ptr(fi), _ptr(ag), _ptr(grad_faces), _ptr(grad_textures),
            _ptr(g), ctypes.c_int(n), ctypes.c_int(f), ctypes.c_int(IS), ctypes.c_int(ts),
            ctypes.c_float(near), ctypes.c_float(far), ctypes.c_float(eps), ctypes.c_float(sigma_val),
            ctypes.c_int(func_id_dist), ctypes.c_float(dist_eps_log), ctypes.c_float(gamma_val),
            ctypes.c_int(func_id_rgb), ctypes.c_int(func_id_alpha), ctypes.c_int(texture_sample_type),
            ctypes.c_int(1 if double_side else 0)]
    lib = _lib(backend)
    if backend == "port":
        rc = lib.oracle_raster_backward(*args, ctypes.c_int(n_threads))
    else:
        rc = lib.ref_backward_soft_rasterize(*args)
    if rc != 0:
        raise RuntimeError("oracle backward rc=%d" % rc)
    return grad_faces, grad_textures


def standard_grid(image_size):
    """The `grid` argument the reference builds (functional/soft_rasterize.py:57-62):
    affine_grid(identity) under torch 1.1.0 semantics (requirements.txt:8), i.e. what
    torch>=1.3 calls align_corners=True: coordinates linspace(-1, 1, IS)."""
    theta = torch.tensor([[1, 0, 0], [0, 1, 0]], dtype=torch.float)
    g = torch.nn.functional.affine_grid(theta.unsqueeze(0), (1, 1, image_size, image_size), align_corners=True)
    return np.ascontiguousarray(g.view(image_size, image_size, 2).numpy(), np.float32)


class _SoftRasterizeCPU(torch.autograd.Function):
    """functional/soft_rasterize.py:9-108 on CPU tensors, kernels = the C oracle."""

    @staticmethod
    def forward(ctx, face_vertices, textures, image_size, background_color, near, far, fill_back, eps,
                sigma_val, dist_eps, gamma_val, aggr_func_rgb, backend, n_threads):
        ctx.cfg = dict(image_size=image_size, near=near, far=far, eps=eps, sigma_val=sigma_val,
                       dist_eps_log=float(np.log(1. / dist_eps - 1.)), gamma_val=gamma_val,
                       func_id_rgb={"hard": 0, "softmax": 1}[aggr_func_rgb], double_side=fill_back,
                       backend=backend, n_threads=n_threads)
        out = raster_forward(face_vertices.detach().numpy(), textures.detach().numpy(),
                             background=background_color, **ctx.cfg)
        ctx.saved = out
        p2f = torch.from_numpy(out["p2f_info"]) / torch.from_numpy(out["p2f_sum"]).clamp_min(1e-12)  # :73
        return torch.from_numpy(out["soft_colors"].copy()), p2f, torch.from_numpy(out["aggrs_info"].copy())

    @staticmethod
    def backward(ctx, grad_soft_colors, grad_p2f=None, grad_aggr=None):
        o = ctx.saved
        cfg = dict(ctx.cfg)
        gf, gt = raster_backward(o["faces"], o["textures"], o["soft_colors"], o["faces_info"], o["aggrs_info"],
                                 grad_soft_colors.contiguous().numpy(), **cfg)
        n, f = gf.shape[:2]
        return (torch.from_numpy(gf).view(n, f, 3, 3), torch.from_numpy(gt)) + (None,) * 12


def soft_rasterize(face_vertices, textures, image_size=256, background_color=(0, 0, 0), near=1, far=100,
                   fill_back=True, eps=1e-3, sigma_val=1e-5, dist_eps=1e-4, gamma_val=1e-4,
                   aggr_func_rgb="softmax", backend="port", n_threads=1):
    """Middle level: functional/soft_rasterize.py:111-125 (euclidean / prod / surface only)."""
    return _SoftRasterizeCPU.apply(face_vertices, textures, image_size, list(background_color), float(near),
                                   float(far), bool(fill_back), float(eps), float(sigma_val), float(dist_eps),
                                   float(gamma_val), aggr_func_rgb, backend, n_threads)


def texture_atlas(faces_uv, textures, image, res_out, tile_width, eps=1e-5, backend="port"):
    """cuda/create_texture_image_cuda_kernel.cu: bake per-face texels [F,R*R,3] into `image` [H,W,3] in place
    (numpy f32); faces_uv [F,3,2] in atlas pixel units."""
    faces_uv = np.ascontiguousarray(faces_uv, np.float32)
    textures = np.ascontiguousarray(textures, np.float32)
    assert image.dtype == np.float32 and image.flags["C_CONTIGUOUS"]
    nf, r_in = textures.shape[0], int(np.sqrt(textures.shape[1]))
    lib = _lib(backend)
    if backend == "ref":
        lib.ref_create_texture_image(_ptr(faces_uv), _ptr(textures), _ptr(image), ctypes.c_long(image.size), nf, r_in,
                                     int(res_out), int(tile_width), ctypes.c_float(eps))
    else:
        rc = lib.oracle_texture_atlas(_ptr(faces_uv), _ptr(textures), _ptr(image), image.shape[0], nf, r_in,
                                      int(res_out), int(tile_width), ctypes.c_float(eps))
        assert rc == 0
    return image


def atlas_layout(num_faces, texture_res):
    """functional/save_obj.py:10-22 with torch-1.1 integer division for `row` (LongTensor / int floors there)."""
    tile_width = int((num_faces - 1.) ** 0.5) + 1
    tile_height = int((num_faces - 1.) / tile_width) + 1
    fn = np.arange(num_faces)
    col, row = (fn % tile_width).astype(np.float32), (fn // tile_width).astype(np.float32)
    uv = np.zeros((num_faces, 3, 2), np.float32)
    uv[:, 0, 0] = col * texture_res + texture_res / 2
    uv[:, 0, 1] = row * texture_res + 1
    uv[:, 1, 0] = col * texture_res + 1
    uv[:, 1, 1] = (row + 1) * texture_res - 1 - 1
    uv[:, 2, 0] = (col + 1) * texture_res - 1 - 1
    uv[:, 2, 1] = (row + 1) * texture_res - 1 - 1
    return tile_width, tile_height, uv


def create_texture_image(textures, texture_res=16, backend="port"):
    """functional/save_obj.py:9-35: returns (image [H,W,3] flipped vertically, uv [F,3,2] normalised)."""
    textures = np.ascontiguousarray(textures, np.float32)
    tw, th, uv = atlas_layout(textures.shape[0], texture_res)
    image = np.ones((th * texture_res, tw * texture_res, 3), np.float32)
    texture_atlas(uv, textures, image, texture_res, tw, 1e-5, backend)
    uv = uv.copy()
    uv[:, :, 0] /= (image.shape[1] - 1)
    uv[:, :, 1] /= (image.shape[0] - 1)
    return image[::-1], uv


def obj_text(name, vertices, faces, uv=None):
    """The exact bytes functional/save_obj.py:55-89 writes (surface textures when uv is given)."""
    out = ['# %s\n' % name, '#\n', '\n']
    if uv is not None:
        out.append('mtllib %s\n\n' % (name[:-4] + '.mtl'))
    for v in vertices:
        out.append('v %.8f %.8f %.8f\n' % (v[0], v[1], v[2]))
    out.append('\n')
    if uv is not None:
        for t in uv.reshape(-1, 2):
            out.append('vt %.8f %.8f\n' % (t[0], t[1]))
        out.append('\n')
        out.append('usemtl material_1\n')
        for i, f in enumerate(faces):
            out.append('f %d/%d %d/%d %d/%d\n' % (f[0] + 1, 3 * i + 1, f[1] + 1, 3 * i + 2, f[2] + 1, 3 * i + 3))
        out.append('\n')
    else:
        for f in faces:
            out.append('f %d %d %d\n' % (f[0] + 1, f[1] + 1, f[2] + 1))
    return ''.join(out)
