"""CPU checks: the C-ABI library builds, loads and exports every symbol include/umr_hip.h declares; the
product package never touches the oracle; the product fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "umr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(umr_[a-z0-9_]+)\s*\(", hdr)))


def test_library_builds_and_exports_every_declared_symbol():
    from umr_amd import build, _lib
    path = build.build(verbose=False)
    h = ctypes.CDLL(path)
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(h, s), "missing export %s" % s
    # and the ctypes prototype table covers the same set
    assert set(_lib.SIGNATURES) | {"umr_version", "umr_build_id"} == set(syms)
    assert _lib.version().startswith("umr_hip")
    from umr_amd import build
    assert _lib.build_id() == build.source_hash()      # the library on disk was compiled from the sources on disk


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "umr_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle|liboracle|softras_ref|/root/reference", txt, flags=re.M):
                    bad.append(fn)
    assert not bad, bad


def test_no_cpu_fallback():
    from umr_amd.smr import SoftRenderer
    from helpers import scene
    verts, faces, cams, _ = scene(1, 1, seed=0)
    with pytest.raises(RuntimeError):
        SoftRenderer(32)(verts, faces, cams)      # CPU tensors: refused, never silently computed on the host


def test_size_queries_need_no_gpu():
    from umr_amd import _lib
    L = _lib.lib()
    assert L.umr_raster_workspace_bytes(2, 1280) >= 2 * 1280 * (16 + 160)
    assert L.umr_raster_workspace_bytes(0, 5) == 0
    assert L.umr_project_workspace_bytes(2, 642) == 2 * 642 * 12


def test_fused_loss_heads_have_no_eager_fallback():
    """The PNet distance head and the part-matching reductions are HIP kernels (csrc/perceptual.hip); handed CPU tensors
    they must refuse loudly instead of computing with eager torch (their CPU restatements live in oracle/ only)."""
    from umr_amd.perceptual import cos_sim_distance
    from umr_amd.functional import PartMatchFunction
    with pytest.raises(RuntimeError, match="GPU tensor"):
        cos_sim_distance([torch.zeros(1, 4, 3, 3)], [torch.zeros(1, 4, 3, 3)])
    with pytest.raises(RuntimeError, match="GPU tensor"):
        PartMatchFunction.apply(torch.zeros(1, 4, 8, 8), torch.zeros(1, 4, 8, 8), torch.zeros(1, 5, 8, 8),
                                (0., 5., 0., 0., 5.), 0.1, 1e-3)
    import umr_amd.loss_utils as LU
    assert not hasattr(LU, "batch_get_centers")       # the eager centroid helper is gone from the product


def test_symmetric_mesh_counts_match_reference_constants():
    """subdivide 3 -> 32 on-plane + 305 mirrored pairs (ShapePredictor emits 337 vertices)."""
    from umr_amd.mesh import create_sphere, make_symmetric
    import numpy as np
    v, f = create_sphere(3)
    v2, f2, n_ind, n_sym = make_symmetric(v, f, axis=1)
    assert (n_ind, n_sym) == (32, 305) and v2.shape == (642, 3) and f2.shape == (1280, 3)
    flip = np.array([1, -1, 1])
    assert np.array_equal(v2[n_ind + n_sym:], v2[n_ind:n_ind + n_sym] * flip)
    assert (v2[:n_ind, 1] == 0).all()


def test_argument_errors_are_reported_without_a_gpu():
    """Rejected calls return UMR_ERR_ARG before anything is enqueued (the reference only printf()s launch errors)."""
    import ctypes
    from umr_amd import _lib
    L = _lib.lib()
    one = ctypes.c_void_p(8)     # any non-NULL value: arguments are validated before being dereferenced on the device
    z = [one] * 9
    tail = [1.0, 100.0, 1e-3, 1e-5, 2, 23.0, 1e-4, 1, 2, 0, 1, 0, None, one, 1 << 30, None]
    assert L.umr_raster_forward(*z, 0, 5, 1, 64, *tail) == -1        # N = 0 (empty batch)
    assert L.umr_raster_forward(*z, 1, 0, 1, 64, *tail) == -1        # F = 0 (empty mesh)
    assert L.umr_raster_forward(*z, 1, 5, 1, 0, *tail) == -1         # empty image
    assert L.umr_raster_forward(*z, 1, 5, 1, 64, 1.0, 100.0, 1e-3, 1e-5, 2, 23.0, 1e-4, 1, 2, 0, 1, 0, None, one, 16, None) == -1  # workspace too small
    assert L.umr_raster_forward(None, *z[1:], 1, 5, 1, 64, *tail) == -1  # NULL faces
    assert L.umr_chamfer_forward(one, one, one, one, one, one, 1, 4, 4, 5, None) == -1   # D not in {2,3}
    assert L.umr_dt_barrier(one, one, None, None, 1, 16, 16, 50.0, one, 8, None) == -1   # workspace too small
    assert L.umr_project_faces_forward(one, one, one, None, one, 0, 4, 4, 5.0, -2.732, 1, None) == -1
    assert L.umr_debug_set(b"no_such_switch", 1) == -1
    # UMR_BWD_ALPHA_GEOMETRY (4) is routed by the face-major kernels only: where umr_raster_backward would take its pixel-major
    # pair -- more texels per face than the LDS accumulators hold, or the A/B switch -- the call is REJECTED (round 4 returned
    # UMR_OK with the rgb gradient in grad_faces).  soft_rasterize_cuda.cpp:122-129: the reference raises, never returns other data
    b = [one] * 8
    btail = [1.0, 100.0, 1e-3, 1e-5, 2, 23.0, 1e-4, 1, 2, 0, 1, one, 1 << 30, None]
    assert L.umr_raster_backward(*b, 1 | 4, 1, 1, 1, 8, 1024, 64, *btail) == -1      # TS = 1024 > 1023
    assert L.umr_raster_backward(*b, 4, 1, 1, 1, 8, 1024, 64, *btail) == -1          # ... with a full-resolution gradient as well
    assert L.umr_raster_backward(*b, 1 | 4, 1, 0, 1, 8, 36, 64, *btail) == -1        # the flag asks for both gradients
    assert L.umr_raster_backward(*b, 1 | 4 | 2, 1, 0, 1, 8, 36, 64, *btail) == -1    # ... and excludes ALPHA_ONLY
    assert L.umr_debug_set(b"bwd_pixel_major", 1) == 0
    try:
        assert L.umr_raster_backward(*b, 1 | 4, 1, 1, 1, 8, 36, 64, *btail) == -1    # pixel-major A/B route: rejected too
    finally:
        assert L.umr_debug_set(b"bwd_pixel_major", 0) == 0


def test_symmetric_face_ordering_matches_reference_constants():
    """utils/mesh.py:102-195 restated: 32 self-mirrored faces + 624 mirrored pairs for the 642-vertex sphere
    (the reference hard-codes num_sym_faces=624, nnutils/cub_mesh.py:125); pairs share vertex order."""
    import numpy as np
    from umr_amd.mesh import create_sphere, make_symmetric, make_faces_symmetric
    v, f = create_sphere(3)
    v2, f2, n_ind, n_sym = make_symmetric(v, f, axis=1)
    f3, nif, nsf = make_faces_symmetric(v2, f2, n_ind, n_sym, axis=1)
    assert (nif, nsf) == (32, 624) and f3.shape == (1280, 3)
    assert sorted(map(tuple, np.sort(f3, 1))) == sorted(map(tuple, np.sort(f2, 1)))       # same face set
    r, l = v2[f3[nif:nif + nsf]], v2[f3[nif + nsf:]]
    assert np.array_equal(r * np.array([1, -1, 1]), l)
    assert (r[..., 1].sum(1) >= l[..., 1].sum(1)).all()                                     # right = positive side
    v4, f4 = create_sphere(4)
    v4s, f4s, ni4, ns4 = make_symmetric(v4, f4, axis=1)
    _, nif4, nsf4 = make_faces_symmetric(v4s, f4s, ni4, ns4, axis=1)
    assert (ni4, ns4, nif4, nsf4) == (64, 1249, 64, 2528)


def _decode_png(data):
    """Test-side decoder for the 8-bit RGB, filter-0 PNGs io_utils.write_png emits."""
    import struct
    import zlib
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, shape = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xffffffff
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 2)
            shape = (h, w)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(shape[0], shape[1] * 3 + 1)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(shape[0], shape[1], 3)


def test_checkpoint_and_obj_formats(tmp_path):
    """Reference file formats: '<label>_net_<epoch>.pth' with un-prefixed keys; tolerant load; OBJ round trip."""
    from umr_amd import io_utils
    from umr_amd.model import MeshNet, default_opts
    opts = default_opts(subdivide=1, nz_feat=16, z_dim=8)
    a = MeshNet((64, 64), opts, nz_feat=16)
    path = io_utils.save_network(torch.nn.DataParallel(a), "pred", "latest", str(tmp_path))
    assert path.endswith("pred_net_latest.pth")
    sd = torch.load(path)
    assert not any(k.startswith("module.") for k in sd)
    # state_dict keys are the reference's (nnutils/cub_mesh.py: ResNetConv.resnet :56, Encoder :87-99, ShapePredictor.pred_layer
    # :176, Camera :281-290, TexturePredictorUV :136-141, buffers mean_v :395 / uv_sampler :431), so .pth files interchange
    ref_keys = ["encoder.resnet_conv.resnet.conv1.weight", "encoder.resnet_conv.resnet.bn1.running_mean",
                "encoder.resnet_conv.resnet.layer1.0.conv1.weight", "encoder.resnet_conv.resnet.layer2.0.downsample.0.weight",
                "encoder.resnet_conv.resnet.layer4.1.bn2.weight", "encoder.resnet_conv.resnet.fc.weight",
                "encoder.enc_conv1.0.weight", "encoder.enc_conv1.1.running_var", "encoder.enc_fc.0.0.weight",
                "encoder.enc_fc.1.1.weight", "encoder.mean_fc.0.weight", "encoder.mean_fc.2.bias", "encoder.logvar_fc.2.weight",
                "shape_predictor.pred_layer.weight", "shape_predictor.pred_layer.bias",
                "cam_predictor.fc_layer.0.0.weight", "cam_predictor.quat_predictor.pred_layer.bias",
                "cam_predictor.prob_predictor.weight", "cam_predictor.scale_predictor.pred_layer.weight",
                "cam_predictor.trans_predictor.pred_layer.bias", "texture_predictor.enc.0.0.weight",
                "texture_predictor.decoder.0.2.weight", "texture_predictor.decoder.1.0.weight", "mean_v", "uv_sampler"]
    assert [k for k in ref_keys if k not in sd] == []
    assert "faces" not in sd and "flip" not in sd                      # plain attributes in the reference (:399, :409)
    multi = MeshNet((64, 64), default_opts(subdivide=1, nz_feat=16, z_dim=8, multiple_cam_hypo=True), nz_feat=16).state_dict()
    for k in ("cam_predictor.fc.0.0.weight", "cam_predictor.camera_predictor.7.quat_predictor.pred_layer.weight",
              "cam_predictor.camera_predictor.0.fc_layer.1.1.running_mean", "cam_predictor.scale_predictor.pred_layer.weight",
              "cam_predictor.trans_predictor.pred_layer.weight", "cam_predictor.quat_predictor.pred_layer.bias",
              "cam_predictor.cam_biases"):                             # cub_mesh.py:309-332
        assert k in multi, k
    assert multi["cam_predictor.prob_predictor.weight"].shape == (8, 16) and multi["cam_predictor.cam_biases"].shape == (8, 4)
    b = MeshNet((64, 64), opts, nz_feat=16)
    loaded = io_utils.load_network(b, "pred", "latest", str(tmp_path))
    assert "uv_sampler" not in loaded and "shape_predictor.pred_layer.weight" in loaded
    assert torch.equal(a.shape_predictor.pred_layer.weight, b.shape_predictor.pred_layer.weight)
    # a checkpoint whose keys do not match is an error, not a silent random-init model
    torch.save({"encoder.resnet_conv.layers.0.weight": torch.zeros(1)}, str(tmp_path / "bad_net_latest.pth"))
    with pytest.raises(RuntimeError, match="not loaded"):
        io_utils.load_network(b, "bad", "latest", str(tmp_path))
    io_utils.save_obj(str(tmp_path / "m.obj"), a.get_mean_shape(), a.faces)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "save_obj.npz"))   # written by the reference
    io_utils.save_obj(str(tmp_path / "plain.obj"), torch.from_numpy(g["verts"]), torch.from_numpy(g["faces"]))
    assert open(tmp_path / "plain.obj", "rb").read() == g["obj_plain"].tobytes()
    with pytest.raises(RuntimeError):                                                # atlas bake is GPU-only
        io_utils.save_obj(str(tmp_path / "t.obj"), torch.from_numpy(g["verts"]), torch.from_numpy(g["faces"]),
                          textures=torch.from_numpy(g["textures"]))
    io_utils.write_png(str(tmp_path / "a.png"), g["png"])
    assert _decode_png(open(tmp_path / "a.png", "rb").read()).tobytes() == g["png"].tobytes()
    lines = open(tmp_path / "m.obj").read().split("\n")
    assert sum(l.startswith("v ") for l in lines) == 42 and sum(l.startswith("f ") for l in lines) == 80
    assert min(int(t) for l in lines if l.startswith("f ") for t in l.split()[1:]) == 1


def test_rasterize_rejects_mismatched_texture_batch():
    """The kernels index textures by (mesh, face): a texture tensor built for another face count (e.g. the 656-face
    symmetric texture space against the 1280-face mesh) must be refused on the host, not read out of bounds."""
    from umr_amd.functional import SoftRasterizeFunction
    fv = torch.zeros(2, 80, 3, 3)
    for tex in (torch.zeros(2, 42, 4, 3), torch.zeros(3, 80, 4, 3), torch.zeros(2, 80, 4)):   # [1,80,4,3] = a group of 2
        with pytest.raises(RuntimeError, match="face_vertices must be"):
            SoftRasterizeFunction.apply(fv, tex, 64)          # shape check precedes any device access


def test_mode_ids_pack_into_one_op_argument():
    """torch.ops.umr.soft_rasterize takes the reference binding's four mode ids (functional/soft_rasterize.py:21-24) as one
    integer; 0 / 1 keep meaning hard / soft-max colour with UMR's own modes."""
    from umr_amd import ops
    assert ops.pack_modes(0) == 0 and ops.pack_modes(1) == 1
    assert ops.unpack_modes(1) == (1, 2, 2, 0) and ops.unpack_modes(0) == (0, 2, 2, 0)
    for rgb in (0, 1):
        for dist in (0, 1, 2):
            for alpha in (0, 1, 2):
                for tex in (0, 1):
                    assert ops.unpack_modes(ops.pack_modes(rgb, dist, alpha, tex)) == (rgb, dist, alpha, tex)
    import math
    sc = ops._scalars(64, 1, 100, True, 1e-3, 1e-5, 1e-10, 1e-4, ops.pack_modes(1, 1, 0, 1))
    assert sc[0] == 64 and sc[5] == 1 and sc[8] == 1 and sc[9] == 0 and sc[10] == 1 and sc[11] == 1
    assert abs(sc[6] - math.log(1. / 1e-10 - 1.)) < 1e-9


def test_soft_rasterize_refuses_cpu_tensors_and_bad_vertex_textures():
    from umr_amd import functional as UF
    fv = torch.zeros(1, 4, 3, 3)
    with pytest.raises(TypeError):
        UF.soft_rasterize(fv, torch.zeros(1, 4, 1, 3), 8)
    with pytest.raises(RuntimeError):      # texture_type 'vertex' needs [N,F,3,3] (the reference reads w[j], j < texture_size)
        UF.SoftRasterizeFunction.apply(fv, torch.zeros(1, 4, 4, 3), 8, [0, 0, 0], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-4, 1e-4,
                                       'softmax', 'prod', 'vertex')


def test_ctypes_signatures_match_the_header_prototypes():
    """Every prototype of include/umr_hip.h has a ctypes signature in umr_amd/_lib.py with the same number of parameters and
    compatible kinds (pointer / integer / float): a drifted argtypes list is undefined behaviour at the first call."""
    import ctypes
    import re
    from umr_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "umr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    protos = re.findall(r"\b(?:int|size_t|long|const char \*)\s*(umr_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)
    assert len(protos) >= 40
    checked = 0
    for name, args in protos:
        params = [a.strip() for a in args.replace("\n", " ").split(",")] if args.strip() not in ("", "void") else []
        if name not in _lib.SIGNATURES:
            assert name in ("umr_version", "umr_build_id"), name + " has no ctypes signature"
            continue
        argtypes, _ = _lib.SIGNATURES[name]
        assert len(argtypes) == len(params), (name, len(argtypes), len(params))
        for a, p in zip(argtypes, params):
            is_ptr = "*" in p
            if is_ptr:
                assert a in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(a, "_type_") and not isinstance(a._type_, str), (name, p, a)
            elif re.search(r"\b(float|double)\b", p):
                assert a in (ctypes.c_float, ctypes.c_double), (name, p, a)
            else:
                assert a in (ctypes.c_int, ctypes.c_long, ctypes.c_size_t, ctypes.c_uint), (name, p, a)
        checked += 1
    assert checked >= 40
    # the flag values the Python layer passes are the header's
    raw = open(os.path.join(ROOT, "include", "umr_hip.h")).read()
    flag = lambda n: int(re.search(r"#define\s+%s\s+(\d+)" % n, raw).group(1))
    from umr_amd import ops
    assert (flag("UMR_BWD_GRAD_POOLED"), flag("UMR_BWD_ALPHA_ONLY"), flag("UMR_BWD_ALPHA_GEOMETRY")) == (1, 2, ops.BWD_ALPHA_GEOMETRY)


def test_every_kernel_of_the_path_is_a_registered_operator_with_a_fake_kernel():
    """north_star: "all exposed to PyTorch-ROCm as custom ops".  umr_amd/ops.py (rasterizer) and umr_amd/ops_losses.py (geometry
    and losses) register torch.ops.umr.*; here, without a GPU: every operator and its backward exist with a schema, and the
    fake (meta) kernels propagate shapes and dtypes under FakeTensorMode -- what torch.compile / export trace through."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from umr_amd import ops, ops_losses  # noqa: F401
    names = ["soft_rasterize", "soft_rasterize_backward", "silhouette", "silhouette_backward", "soft_rasterize_alpha_geometry",
             "soft_rasterize_alpha_geometry_backward"] + list(ops_losses.ALL_OPS) + \
        [n + "_backward" for n in ops_losses.ALL_OPS if n != "dt_barrier"]
    for n in names:
        assert hasattr(torch.ops.umr, n), n
        assert str(getattr(torch.ops.umr, n).default._schema).startswith("umr::" + n + "("), n
    with FakeTensorMode():
        f = lambda *s, dt=torch.float32: torch.empty(*s, device="cuda", dtype=dt)
        v, c, fi = f(4, 10, 3), f(4, 7), f(4, 6, 3, dt=torch.int32)
        assert torch.ops.umr.project_faces(v, c, fi, 5.0, -2.7).shape == (4, 6, 3, 3)
        assert torch.ops.umr.project_points(v, c, 3, 0.0).shape == (4, 10, 3)
        assert torch.ops.umr.grid_sample_cl(f(2, 3, 8, 8), f(2, 5, 2)).shape == (2, 5, 3)
        assert torch.ops.umr.laplacian(v, f(11, dt=torch.int32), f(40, dt=torch.int32))[0].shape == (4,)
        assert torch.ops.umr.flatten(v, f(9, 4, dt=torch.int32)).shape == (4,)
        assert torch.ops.umr.dt_barrier(f(2, 16, 16), 50.0).shape == (2, 16, 16)
        assert torch.ops.umr.row_norm_mean(v).shape == () and torch.ops.umr.abs_column_mean(v, 1).shape == ()
        d1, d2, i1, i2 = torch.ops.umr.chamfer(f(2, 5, 2), f(2, 7, 2))
        assert d1.shape == (2, 5) and i2.shape == (2, 7) and i1.dtype == torch.int32
        gv, gc = torch.ops.umr.project_points_backward(f(4, 10, 3), v, c, 3, True)
        assert gv.shape == v.shape and gc.shape == c.shape
