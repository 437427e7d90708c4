"""Round-6 GPU tests (MI355X, through the C ABI): heavy faces split into work items (k_face_order) -- against the unsplit kernel
and against the oracle on geometry a training step really rendered --, the binding INTEGRATION.md tells a maintainer to paste,
and the whole training step with its RCCL gradient exchange replayed from one HIP graph."""
import math
import os
import re
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import assert_close_frac, t2n  # noqa: E402
from tests.test_gpu_round5 import _forward_cabi, replay_equals_eager, UMR_BWD_GRAD_POOLED, UMR_BWD_ALPHA_GEOMETRY, UMR_BWD_PACKED_STATE  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
UMR_BWD_ALPHA_ONLY = 2
LIVE_SCENE = os.path.join(ROOT, "tests", "golden", "live_s1_scene_a.npz")   # = profiles/scenes/live_s1_a.npz (bench.py --capture-scene)


def _subtiles_under_bbox(fv, IS):
    """k_face_setup's work estimate (raster_core.h `cost`): 4x4 sub-tiles under the bbox dilated by sqrt(threshold)."""
    thr = np.float32(np.sqrt(np.float32(np.log(1. / 1e-10 - 1.)) * np.float32(1e-5)))
    f = fv.reshape(fv.shape[0], -1, 3, 3)
    x, y, h = f[..., 0], f[..., 1], 0.5 * IS
    px0 = np.maximum(np.floor((x.min(-1) - thr) * h + h - 0.5) - 1, 0); px1 = np.minimum(np.ceil((x.max(-1) + thr) * h + h - 0.5) + 1, IS - 1)
    py0 = np.maximum(np.floor((y.min(-1) - thr) * h + h - 0.5) - 1, 0); py1 = np.minimum(np.ceil((y.max(-1) + thr) * h + h - 0.5) + 1, IS - 1)
    return np.where((px0 <= px1) & (py0 <= py1), ((px1 // 4) - (px0 // 4) + 1) * ((py1 // 4) - (py0 // 4) + 1), 0)


def _backward(variant, fvd, texd, st, g, IS):
    from umr_amd import _lib
    L, p = _lib.lib(), _lib.ptr
    N, F, TS = fvd.shape[0], fvd.shape[1], texd.shape[2]
    gf = torch.zeros(N, F, 9, device=DEV)
    gt = torch.zeros(N, F, TS, 3, device=DEV)
    stream = _lib.stream_ptr(torch.device(DEV))
    if variant == "silhouette":
        alpha = st["sc"][:, 3].contiguous()
        rc = L.umr_raster_backward(p(fvd), None, p(alpha), None, None, p(gf), None, p(g[:, 3].contiguous()), UMR_BWD_GRAD_POOLED | UMR_BWD_ALPHA_ONLY,
                                   1, 0, N, F, 1, IS, *st["scal"], p(st["ws"]), st["wsb"], stream)
    elif variant == "texel_only":
        rc = L.umr_raster_backward(p(fvd), p(texd), p(st["sc"]), None, p(st["aggrs"]), None, p(gt), p(g), UMR_BWD_GRAD_POOLED,
                                   0, 1, N, F, TS, IS, *st["scal"], p(st["ws"]), st["wsb"], stream)
    elif variant == "vertex_only":       # (train_s2's unseen-view / part renders: textures detached)
        rc = L.umr_raster_backward(p(fvd), p(texd), p(st["sc"]), None, p(st["aggrs"]), p(gf), None, p(g), UMR_BWD_GRAD_POOLED,
                                   1, 0, N, F, TS, IS, *st["scal"], p(st["ws"]), st["wsb"], stream)
    else:
        packed = variant == "one_pass_packed"
        rc = L.umr_raster_backward(p(fvd), p(texd), None if packed else p(st["sc"]), None, p(st["aggrs"]), p(gf), p(gt), p(g),
                                   UMR_BWD_GRAD_POOLED | UMR_BWD_ALPHA_GEOMETRY | (UMR_BWD_PACKED_STATE if packed else 0),
                                   1, 1, N, F, TS, IS, *st["scal"], p(st["ws"]), st["wsb"], stream)
    assert rc == 0
    torch.cuda.synchronize()
    return gf, gt


@pytest.mark.parametrize("variant", ["one_pass_packed", "one_pass", "texel_only", "vertex_only", "silhouette"])
def test_split_faces_equal_the_unsplit_kernel_on_a_live_scene(variant):
    """The face-major backward on geometry a training step rendered (16 x 1280 faces at 512^2; faces of up to 1560 candidate
    sub-tiles where the median is 63): with k_face_order's default threshold the heavy faces are split into work items whose
    partial sums k_split_reduce adds in part order.  Against umr_debug_set("face_split", 0) -- one wave per face, round 5's
    kernel: faces of a single culling pass cannot split and come out bit-identical; split faces agree to summation order; two runs
    of the split kernel give the same bits (no float atomics, arrival order does not matter)."""
    from umr_amd import _lib
    z = np.load(LIVE_SCENE)
    fv = torch.from_numpy(z["fv_shared"]).reshape(16, -1, 9)
    N, F, IS, TS = fv.shape[0], fv.shape[1], 512, 36
    gen = torch.Generator().manual_seed(5)
    tex = torch.rand(N, F, TS, 3, generator=gen)
    g = torch.randn(N, 4, IS // 2, IS // 2, generator=gen).to(DEV)
    fvd, texd = fv.to(DEV).contiguous(), tex.to(DEV)
    packed = variant == "one_pass_packed"
    st = _forward_cabi(fvd, texd, IS, True, packed=packed)
    nt = _subtiles_under_bbox(fv.numpy(), IS)
    assert nt.max() > 1000 and np.median(nt) < 100
    try:
        _lib.debug_set("face_split", 0)
        ref = _backward(variant, fvd, texd, st, g, IS)
        _lib.debug_set("face_split", -1)         # the default threshold
        got = _backward(variant, fvd, texd, st, g, IS)
        again = _backward(variant, fvd, texd, st, g, IS)
    finally:
        _lib.debug_set("face_split", -1)
    single = torch.from_numpy(nt <= 64).to(DEV)
    changed = 0
    for a, b, c, name in zip(ref, got, again, ("grad_faces", "grad_textures")):
        if float(a.abs().max()) == 0:
            continue
        assert torch.equal(b, c), name + ": two runs of the split kernel differ"
        assert torch.equal(b[single], a[single]), name + ": an unsplittable face changed"
        s = float(a.abs().max())
        assert_close_frac(t2n(b), t2n(a), atol=2e-6 * s, rtol=1e-5, frac=1.0, name="split vs unsplit %s (%s)" % (name, variant))
        changed += int((a != b).flatten(2).any(2).sum())
    assert changed > 0, "no face was split"


def test_one_pass_backward_on_live_geometry_vs_oracle(oracle_built):
    """The kernel the timed step lives on, with its heavy faces split, against the ORACLE on the two heaviest meshes of a captured
    training step (faces far larger than the frame, vertices at +-3 screen half-widths): grad_faces vs the oracle's backward
    with upstream (0, 0, 0, g_alpha), grad_textures vs upstream (g_rgb, 0) -- the reference's backward is linear in the upstream
    gradient (soft_rasterize_cuda_kernel.cu:480-656) -- every element, at the bounds of test_alpha_geometry_backward_vs_oracle."""
    from oracle import softras
    z = np.load(LIVE_SCENE)
    nt = _subtiles_under_bbox(z["fv_shared"].reshape(16, -1, 9), 512)
    pick = np.argsort(-nt.max(1))[:2]
    fv = torch.from_numpy(np.ascontiguousarray(z["fv_shared"][pick])).reshape(2, -1, 9)
    N, F, IS, TS = 2, fv.shape[1], 512, 36
    gen = torch.Generator().manual_seed(9)
    tex = torch.rand(N, F, TS, 3, generator=gen)
    g = torch.randn(N, 4, IS // 2, IS // 2, generator=gen)
    g_full = (0.25 * g).repeat_interleave(2, 2).repeat_interleave(2, 3).contiguous()
    cfg = dict(near=1., far=100., eps=1e-3, sigma_val=1e-5, dist_eps_log=float(math.log(1e10 - 1.)), gamma_val=1e-4,
               func_id_rgb=1, double_side=True)
    nth = softras.max_threads()
    o = softras.raster_forward(fv.numpy().reshape(N, F, 3, 3), tex.numpy(), IS, backend="port", n_threads=nth, **cfg)
    g_a, g_rgb = g_full.clone(), g_full.clone()
    g_a[:, :3] = 0
    g_rgb[:, 3] = 0
    args = (o["faces"], o["textures"], o["soft_colors"], o["faces_info"], o["aggrs_info"])
    gf_ref, _ = softras.raster_backward(*args, g_a.numpy(), IS, backend="port", n_threads=nth, **cfg)
    _, gt_ref = softras.raster_backward(*args, g_rgb.numpy(), IS, backend="port", n_threads=nth, **cfg)
    fvd, texd = fv.to(DEV).contiguous(), tex.to(DEV)
    st = _forward_cabi(fvd, texd, IS, True, packed=True)
    gf, gt = _backward("one_pass_packed", fvd, texd, st, g.to(DEV), IS)
    ref_pool = torch.nn.functional.avg_pool2d(torch.from_numpy(o["soft_colors"]), 2, 2).numpy()
    assert_close_frac(t2n(st["pool"]), ref_pool, atol=1e-4, frac=1.0, max_outlier=1e-5, name="live scene pooled image")
    sf, stx = np.abs(gf_ref).max(), np.abs(gt_ref).max()
    assert sf > 0 and stx > 0
    assert_close_frac(t2n(gf).reshape(gf_ref.shape), gf_ref, atol=1e-5 * sf, rtol=1e-4, frac=1.0, name="live scene grad_faces (alpha term)")
    assert_close_frac(t2n(gt), gt_ref, atol=3e-6 * stx, rtol=1e-4, frac=1.0, name="live scene grad_textures (rgb term)")


def test_integration_md_ctypes_stub_is_the_tested_binding():
    """INTEGRATION.md section 1 shows the file a maintainer drops in as soft_renderer/cuda/soft_rasterize.py.  This test takes THAT
    code block out of the document, points it at the built library, executes it and drives its two functions exactly as
    functional/soft_rasterize.py:47-71, 95-106 does (buffers pre-allocated and pre-filled by the caller) on the reference's own
    golden (raster_softmax_ts36.npz): the binding that is documented is the binding that is tested."""
    from umr_amd import _lib
    from umr_amd.functional import standard_grid
    from conftest import load_golden
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## 1. Replace only the CUDA extension"):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert "def forward_soft_rasterize(" in code and "def backward_soft_rasterize(" in code
    assert code.count("/path/to/umr_amd/lib/libumr_hip.so") == 1
    mod = types.ModuleType("soft_rasterize_stub")
    exec(compile(code.replace("/path/to/umr_amd/lib/libumr_hip.so", _lib.LIB_PATH), "INTEGRATION.md#1", "exec"), mod.__dict__)
    g = load_golden("raster_softmax_ts36.npz")
    faces, tex = torch.from_numpy(g["faces"]).to(DEV), torch.from_numpy(g["textures"]).to(DEV)
    N, F, IS = faces.shape[0], faces.shape[1], int(g["image_size"])
    faces_info, aggrs = torch.zeros(N, F, 27, device=DEV), torch.zeros(N, 2, IS, IS, device=DEV)
    p2f_info, p2f_sum = torch.zeros(N, F, 2, device=DEV), torch.zeros(N, F, 2, device=DEV)
    sc = torch.ones(N, 4, IS, IS, device=DEV)
    for k in range(3):
        sc[:, k] *= float(g["background"][k])
    scal = (IS, float(g["near"]), float(g["far"]), float(g["eps"]), float(g["sigma_val"]), 2, float(g["dist_eps_log"]),
            float(g["gamma_val"]), int(g["func_id_rgb"]), 2, 0, bool(g["double_side"]))
    out = mod.forward_soft_rasterize(faces, tex, faces_info, aggrs, standard_grid(IS, torch.device(DEV)), p2f_info, p2f_sum, sc, *scal)
    assert out[0] is faces_info and out[4] is sc                      # in place, same handles back (soft_rasterize_cuda.cpp:96)
    gf, gt = torch.zeros(N, F, 9, device=DEV), torch.zeros_like(tex)
    gsc = torch.from_numpy(g["grad_soft_colors"]).to(DEV)
    out_b = mod.backward_soft_rasterize(faces, tex, sc, faces_info, aggrs, gf, gt, gsc, *scal)
    torch.cuda.synchronize()
    assert out_b[0] is gf and out_b[1] is gt
    np.testing.assert_array_equal(t2n(faces_info), g["faces_info"])
    assert_close_frac(t2n(sc), g["soft_colors"], atol=1e-4, frac=1.0, max_outlier=1e-5, name="stub soft_colors")
    assert_close_frac(t2n(aggrs), g["aggrs_info"], atol=0, rtol=3e-6, frac=1.0, name="stub aggrs")
    sf, stx = np.abs(g["grad_faces"]).max(), np.abs(g["grad_textures"]).max()
    assert_close_frac(t2n(gf), g["grad_faces"], atol=1.5e-5 * sf, rtol=1e-4, frac=1.0, name="stub grad_faces")
    assert_close_frac(t2n(gt), g["grad_textures"], atol=3e-6 * stx, rtol=1e-4, frac=1.0, name="stub grad_textures")
    with pytest.raises(RuntimeError):                                 # CHECK_INPUT (:57-59): a host tensor is refused
        mod.forward_soft_rasterize(faces.cpu(), tex, faces_info, aggrs, standard_grid(IS, torch.device(DEV)), p2f_info, p2f_sum, sc, *scal)


def test_whole_step_with_rccl_gradient_exchange_replays_from_a_hip_graph():
    """What `bench.py --gpus N` (N > 1) and `--force-ddp 1` time: the whole train_s1 step INCLUDING the gradient exchange --
    parallel.BucketedGradSync: gradients as views of one flat buffer, buckets all-reduced over RCCL from the gradient hooks
    while backward runs -- captured into one HIP graph.  One real RCCL rank here (the box has one GPU): the all-reduce kernels
    are in the graph, and a replay must equal the eager step from the same state (tests/test_gpu_round5.py's checker)."""
    import argparse
    import socket
    import torch.distributed as dist
    from umr_amd import model as M
    from umr_amd.synthetic import make_s1_inputs
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", TORCH_NCCL_ASYNC_ERROR_HANDLING="0")
    dev = torch.device(DEV)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(77)
        args = argparse.Namespace(graph=1, batch=4, image_size=64, subdivide=2, epoch=0, share_mask_render=1)
        tv, faces, _, _ = make_s1_inputs(4, 64, 2, seed=100, device=dev)
        step = M.build_training_step(tv, faces, args, dev, 2)          # world 2: the data-parallel form of the step
        assert step.sync is not None and len(step.sync.buckets) >= 1
        flat = step.sync.flat
        assert all(p.grad is not None and flat.data_ptr() <= p.grad.data_ptr() < flat.data_ptr() + 4 * flat.numel()
                   for p in step.model.parameters() if p.requires_grad)
        replay_equals_eager(step, dev, "s1 + RCCL bucket all-reduce (1 rank)", graph_kwargs={"capture_error_mode": "thread_local"}, bn_eval=True)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("workload", ["s1", "s2"])
def test_replayed_step_matches_the_eager_step_at_bench_size(workload):
    """The defect this guards against shows at BENCH size only: the texture decoder's 64x128 / 128x256 layers sum their bias
    gradient over enough values for ATen to split the reduction over several workgroups that meet at a semaphore zeroed by
    cudaMemsetAsync -- and a captured memset node is not executed again on replay on this ROCm stack, so every replay of the
    captured step held garbage (up to 1e38) in those gradients while the eager step was right.  umr_amd.model.Conv2d sums such
    bias gradients in stages; umr_amd.graph_check.replay_matches_eager (what bench.py runs before it times replays) compares
    EVERY gradient tensor of a replay with the eager step's from one saved state.  With the workaround switched off the check
    must fail -- it sees the defect -- unless a later stack has fixed the memset nodes (then it only says so)."""
    import argparse
    from umr_amd import model as M
    from umr_amd.graph_check import replay_matches_eager
    from umr_amd.synthetic import make_s1_inputs
    dev = torch.device(DEV)

    def captured(batch):
        torch.manual_seed(77)
        args = argparse.Namespace(graph=1, batch=batch, image_size=256, subdivide=3, epoch=0, share_mask_render=1, data_seed=100)
        if workload == "s2":
            step = M.build_training_step_s2(args, dev, 1)
        else:
            tv, faces, _, _ = make_s1_inputs(batch, 256, 3, seed=100, device=dev)
            step = M.build_training_step(tv, faces, args, dev, 1)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.autograd.set_multithreading_enabled(False), torch.cuda.graph(g, stream=side):
                static_loss = step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g.replay()                      # (a first replay is not the interesting one: the capture's own memsets ran once)
        torch.cuda.synchronize()
        return step, g, static_loss

    step, g, static_loss = captured(16 if workload == "s1" else 8)
    rep = replay_matches_eager(step, g, static_loss, dev)
    print("[graph-check] %s: %r" % (workload, rep))
    assert rep["ok"] and rep["compared"] >= rep["tensors"] // 3, rep
    del step, g, static_loss
    saved = M.Conv2d.STAGED_FROM
    M.Conv2d.STAGED_FROM = 1 << 62          # the library convolution's own bias gradient again
    try:
        step, g, static_loss = captured(16 if workload == "s1" else 8)
        rep = replay_matches_eager(step, g, static_loss, dev)
    finally:
        M.Conv2d.STAGED_FROM = saved
    print("[graph-check] %s, bias gradients left to the convolution's backward: ok=%s bad=%r" % (workload, rep["ok"], rep["bad"]))
    if rep["ok"]:
        print("[graph-check] this stack replays the reduction's memset node: the staged bias gradient is no longer needed")
    else:
        assert any("decoder" in n and n.endswith(".bias") for n, _ in rep["bad"]), rep


def test_backward_reuses_the_forward_workspace():
    """UMR_BWD_REUSE_WORKSPACE through the C ABI on the device: the one-pass backward handed the workspace its forward filled gives
    the bits of the stateless call that rebuilds the face records (one small launch less per backward -- what the training steps'
    shared render does: umr_amd/ops.py keeps the forward's workspace with the saved state)."""
    from umr_amd import _lib
    from tests.helpers import scene
    from oracle import torch_ref
    L, p = _lib.lib(), _lib.ptr
    verts, faces, cams, gen = scene(2, 3, seed=12)
    proj = torch_ref.orthographic_proj_withz(verts, cams, 5.) * torch.tensor([1., -1., 1.])
    fvd = torch_ref.face_vertices(torch_ref.look_at_ortho(proj), faces).reshape(2, -1, 9).contiguous().to(DEV)
    N, F, IS, TS = 2, fvd.shape[1], 256, 36
    texd = torch.rand(N, F, TS, 3, generator=gen).to(DEV)
    g = torch.randn(N, 4, IS // 2, IS // 2, generator=gen).to(DEV)
    st = _forward_cabi(fvd, texd, IS, True, packed=True)
    out = []
    for reuse in (False, True):
        gf, gt = torch.zeros(N, F, 9, device=DEV), torch.zeros(N, F, TS, 3, device=DEV)
        ws = st["ws"] if reuse else torch.empty_like(st["ws"])
        rc = L.umr_raster_backward(p(fvd), p(texd), None, None, p(st["aggrs"]), p(gf), p(gt), p(g),
                                   UMR_BWD_GRAD_POOLED | UMR_BWD_ALPHA_GEOMETRY | UMR_BWD_PACKED_STATE | (16 if reuse else 0),
                                   1, 1, N, F, TS, IS, *st["scal"], p(ws), st["wsb"], _lib.stream_ptr(torch.device(DEV)))
        assert rc == 0
        torch.cuda.synchronize()
        out.append((gf, gt))
    assert float(out[0][0].abs().max()) > 0 and float(out[0][1].abs().max()) > 0
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def _two_rank_worker(rank, world, port, q):
    import torch.distributed as dist
    from umr_amd.model import build_training_step
    from umr_amd.synthetic import template
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device(DEV)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)       # (one GPU on the box: RCCL refuses two ranks on a device)
    try:
        torch.manual_seed(0)                                           # identical initial replicas, as bench.py
        args = types.SimpleNamespace(batch=2, image_size=64, subdivide=2, epoch=0, graph=0, share_mask_render=1)
        tv, faces = template(2)
        step = build_training_step(tv, faces, args, dev, world)        # per-rank data: seed 100 + rank
        assert step.sync is not None
        losses, sums = [], []
        for _ in range(3):
            losses.append(float(step()))
            torch.cuda.synchronize()
            sums.append(float(sum(p.detach().double().sum() for p in step.model.parameters() if p.requires_grad)))
        g = step.sync.flat.double()
        q.put((rank, losses, sums, float(g.sum()), float(g.abs().sum())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_exchange_gradients():
    """The data-parallel step at WORLD SIZE 2 on the device: two processes share the box's one GPU, device tensors, the real kernels;
    the gradient exchange (parallel.BucketedGradSync: hooks fire on the autograd thread, buckets all-reduced asynchronously while
    backward runs, finish() before Adam) goes over gloo because RCCL does not take two ranks on one device -- the schedule, the
    hooks and the stream ordering are the N > 1 path's, only the transport differs.  Different data per rank, three steps: the
    ranks hold the same averaged gradient to the last bit and identical replicas after every step."""
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    (_, l0, s0, g0, a0), (_, l1, s1, g1, a1) = res
    assert all(math.isfinite(v) for v in l0 + l1) and l0 != l1          # finite, and the ranks really see different shards
    assert a0 > 0 and g0 == g1 and a0 == a1, (g0, g1, a0, a1)          # the same all-reduced gradient on both ranks
    assert s0 == s1 and s0[0] != s0[1], (s0, s1)                       # identical replicas after every step, and they move
