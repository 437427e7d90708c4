#!/usr/bin/env python3
"""bench.py -- train images/sec of the UMR render-and-compare hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`, or
    plain `python bench.py --gpus N`, which re-launches itself that way: one process per GPU, RCCL)

One "step" = one pass of the hot path over one batch of synthetic CUB-shaped input per GPU:
BASELINE.json configs[1] = train_s1, bs=16 per GPU, 256x256 images (512x512 internal raster),
642-vertex / 1280-face mesh: 4 raster forwards + 3 raster backwards per image plus every geometric /
image loss, forward and backward, producing gradients for vertices, cameras and texture flow
(umr_amd/train_step.py mirrors experiments/train_s1.py:177-265, texture term = the AlexNet perceptual distance of
train_s1.py:150).  With --model (default when the model module is present) the ResNet-18 MeshNet forward/backward,
the RCCL gradient all-reduce (N > 1) and the Adam step are inside the timed region too.  Inputs are resident in HBM
before the clock starts.

Prints ONE JSON line (rank 0) with `roofline` (raster-backward kernel: HIP events recorded by libumr_hip.so on the
launch stream in a separate short pass AFTER the timed steps, so the event bookkeeping is not in `value`) and
`cpu_baseline` (the CPU oracle = reference algorithm, brute force, OpenMP over all host cores, on a bounded sample;
rank 0, N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_RESULT_OUT = None
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
METRIC = "train images/sec (CUB 256^2, 642-vert mesh)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)   # covers MIOpen's first-use kernel search on a fresh box
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--subdivide", type=int, default=3)
    ap.add_argument("--model", type=int, default=-1, help="1: include MeshNet fwd/bwd + all-reduce + Adam; 0: hot path only")
    ap.add_argument("--cpu-baseline", type=int, default=-1, help="1/0; default: on for N=1")
    ap.add_argument("--cpu-sample", type=int, default=0, help="images in the CPU sample (0 = auto)")
    ap.add_argument("--force-ddp", type=int, default=0,
                    help="debug: create a 1-rank RCCL process group and wrap the model in DDP even when --gpus 1")
    ap.add_argument("--workload", default="s1", choices=["s1", "s2"],
                    help="s1 = BASELINE configs[1] (headline); s2 = train_s2 sequence of configs[2]/[3] (8 camera hypotheses)")
    ap.add_argument("--epoch", type=int, default=0, help="train_s1 epoch (gates the symmetry / deformation terms)")
    ap.add_argument("--profile-steps", type=int, default=5, help="untimed steps with kernel events for `roofline`")
    ap.add_argument("--fixed-scene", type=int, default=1, help="0: skip roofline.fixed_scene_us (the PMC passes of tools/collect_traffic.sh: "
                    "their per-launch averages are over the step's own launches only)")
    ap.add_argument("--share-mask-render", type=int, default=1,
                    help="1: the mask render is the alpha channel of the textured render of the same views (one render where the "
                         "reference makes two); 0: both renders, for A/B")
    ap.add_argument("--capture-scene", default="", help="write the inputs of the FIRST profile step's shared render and unseen-view silhouette "
                    "(projected face vertices, texels, upstream gradient) to this .npz: how profiles/scenes/*.npz were frozen")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launch (0 = pick a free one)")
    ap.add_argument("--graph", type=int, default=-1,
                    help="capture the step in ONE HIP graph and time graph replays instead of eager launches: with --model 0 the "
                         "render-and-compare step (every raster / loss kernel, forward and backward; one GPU), otherwise "
                         "the whole training step incl. MeshNet, backward, the bucketed RCCL gradient all-reduce (N > 1) and Adam.  "
                         "Default (-1): on, at EVERY --gpus N -- the eager step's wall time is its Python / dispatcher enqueue time "
                         "(config.eager_host_enqueue_ms_per_step), which depends on the box's host more than on the GPU; falls "
                         "back to the eager step if the capture fails (all ranks together; the line then says hip_graph: false)")
    ap.add_argument("--grad-sync", default="buckets", choices=["buckets", "ddp"],
                    help="N > 1: `buckets` = umr_amd.parallel.BucketedGradSync (64 MB buckets all-reduced from gradient hooks while "
                         "backward runs; capturable); `ddp` = torch DistributedDataParallel (eager only: implies --graph 0)")
    ap.add_argument("--hot-path-sub", type=int, default=-1,
                    help="1: also time the render-and-compare step alone (the --model 0 --graph 1 measurement) in this process and "
                         "attach it as config.hot_path_images_per_s / hot_path_ms_per_step; default: on for the --gpus 1 train_s1 "
                         "line with the network")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 outside torchrun: start one rank per GPU of this node under
    torch.distributed.run (RCCL rendezvous on 127.0.0.1) and pass its exit status on.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    port = args.master_port
    if port <= 0:                       # default: a port that is free right now (two benches on one node must not collide)
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    # the ranks run the script this process was started as (bench.py; the CPU suite starts it through a wrapper that puts the
    # wave64 emulation of the library under it: tests/bench_rank_on_emulator.py)
    entry = os.path.abspath(sys.argv[0]) if sys.argv and sys.argv[0].endswith(".py") else os.path.abspath(__file__)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), entry] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver (RCCL needs it)
    # N ranks share the host's cores: OpenMP (MIOpen's host side, the CPU-baseline oracle on rank 0) and torch's intra-op pool
    # get an N-th each instead of N x all cores
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or 8) // max(1, args.gpus)))))
    raise SystemExit(subprocess.call(cmd, env=env))


def cpu_baseline(args, n_images):
    """The reference algorithm on host cores: oracle/ (C raster, OpenMP over all cores + torch-CPU losses incl. the
    AlexNet perceptual texture term), same train_s1 sequence, bounded sample of `n_images` images of the same workload."""
    from oracle import softras, torch_ref
    from oracle.train_step_ref import RenderCompareS1Ref
    from umr_amd.perceptual import PNet
    from umr_amd.synthetic import make_s1_inputs
    cores = softras.max_threads()
    torch.set_num_threads(cores)
    tv, faces, outputs, batch = make_s1_inputs(n_images, args.image_size, args.subdivide, seed=1, device="cpu")
    step = RenderCompareS1Ref(tv, faces, args.image_size, n_threads=cores, epoch=getattr(args, "epoch", 0),
                              texture_loss=torch_ref.PerceptualTextureLoss(PNet().state_dict()))
    t0 = time.perf_counter()
    total, _ = step(outputs, batch)
    total.backward()
    dt = time.perf_counter() - t0
    return {"value": n_images / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d image(s) of the same train_s1 step (4 raster fwd + 3 bwd per image + all losses incl. the AlexNet "
                      "perceptual texture term, fwd+bwd), %.1f s wall; MeshNet / discriminator / Adam excluded on the CPU "
                      "side" % (n_images, dt)}


def raster_launch_times(dev, fv, fv_sil, iters=20, tex=None, g_tex=None, seed=0, two_render_forms=True):
    """us per launch (library-owned HIP events) of the raster launches of one train_s1 step on GIVEN projected face vertices:
    `fv` [N,F,3,3] = the views of the shared mask / texture render, `fv_sil` [M,F,3,3] = the views of the silhouette launch.
    Texels and upstream gradients steer no branch of these kernels (values only), so they are seeded noise unless given
    (`tex` [N,F,36,3], `g_tex` [N,4,IS/2,IS/2]: a captured step's own, tools/scene_times.py compares the two)."""
    from umr_amd import _lib, functional as UF
    g = torch.Generator().manual_seed(seed)
    IS, TS = 512, 36
    N, M, F = fv.shape[0], fv_sil.shape[0], fv.shape[1]
    fv = fv.detach().to(dev)
    tex = (torch.rand(N, F, TS, 3, generator=g) if tex is None else tex.detach().float().cpu()).to(dev).requires_grad_(True)
    fv_sil = fv_sil.detach().to(dev).clone().requires_grad_(True)
    g_tex = (torch.randn(N, 4, IS // 2, IS // 2, generator=g) if g_tex is None else g_tex.detach().float().cpu()).to(dev)
    g_sil = torch.randn(M, IS // 2, IS // 2, generator=g).to(dev)
    out = {}
    for phase in range(2):
        if phase:
            _lib.profile_enable(True)
            for k in range(4):
                _lib.profile_collect(k)
        for _ in range(iters if phase else 2):
            tex.grad = None; fv_sil.grad = None
            if two_render_forms:
                sc = UF.soft_rasterize(fv, tex, IS, [0, 0, 0], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, 'softmax', 'prod', 'surface',
                                       pool=True, need_p2f=True, want_visibility=True)[0]
            a = UF.silhouette(fv_sil, IS, 1., 100., True, 1e-3, 1e-5, 1e-10, 1e-4, True)
            if two_render_forms:
                sc.backward(g_tex)
            a.backward(g_sil)
        torch.cuda.synchronize()
    alg = out.setdefault("_alg_bytes", {})       # algorithmic bytes per launch, as the library accounts them (raster.hip: ProfScope)
    for name, k in (("textured_forward_p2f_vis_pool_N%d" % N, 0), ("texel_gradient_backward_N%d" % N, 1),
                    ("silhouette_forward_N%d" % M, 2), ("silhouette_backward_N%d" % M, 3)):
        ms, n, nbytes = _lib.profile_collect(k)
        if n:
            out[name] = round(1e3 * ms / n, 1)
            alg[name] = nbytes / n
    # the shared mask / texture render of the same views: its ONE backward pass (alpha gradient -> vertices, rgb -> texels),
    # which replaces a texel-gradient backward + a silhouette backward -- with the saved state packed (lean_state: what
    # the training steps run; its forward writes nothing else at full resolution) and with the reference's planes
    fv_sh = fv.clone().requires_grad_(True)
    for lean, tag in ((True, ""), (False, "_planar_state")) if two_render_forms else ((True, ""),):
        for phase in range(2):
            if phase:
                _lib.profile_collect(0); _lib.profile_collect(1)
            for _ in range(iters if phase else 2):
                tex.grad = None; fv_sh.grad = None
                UF.soft_rasterize(fv_sh, tex, IS, [0, 0, 0], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, 'softmax', 'prod', 'surface',
                                  pool=True, need_p2f=True, want_visibility=True, detach_rgb_geometry=True, lean_state=lean)[0].backward(g_tex)
            torch.cuda.synchronize()
        ms, n, nbytes = _lib.profile_collect(1)
        out["shared_render_backward_one_pass%s_N%d" % (tag, N)] = round(1e3 * ms / max(n, 1), 1)
        alg["shared_render_backward_one_pass%s_N%d" % (tag, N)] = nbytes / max(n, 1)
        if lean:
            ms, n, nbytes = _lib.profile_collect(0)
            out["shared_render_forward_packed_state_N%d" % N] = round(1e3 * ms / max(n, 1), 1)
            alg["shared_render_forward_packed_state_N%d" % N] = nbytes / max(n, 1)
    # the vertex-gradient-only backward of a textured render (train_s2's unseen-view and part renders: textures detached)
    tex_d = tex.detach()
    for phase in range(2):
        if phase:
            _lib.profile_collect(0); _lib.profile_collect(1)
        for _ in range(iters if phase else 2):
            fv_sh.grad = None
            UF.soft_rasterize(fv_sh, tex_d, IS, [0, 0, 0], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4, 'softmax', 'prod', 'surface',
                              pool=True, need_p2f=False)[0].backward(g_tex)
        torch.cuda.synchronize()
    ms, n, nbytes = _lib.profile_collect(1)
    out["vertex_gradient_backward_N%d" % N] = round(1e3 * ms / max(n, 1), 1)
    alg["vertex_gradient_backward_N%d" % N] = nbytes / max(n, 1)
    _lib.profile_collect(0)
    _lib.profile_enable(False)
    return out


def fixed_scene_kernel_times(dev, iters=20, scale=(0.6, 0.9), N=16):
    """The raster launches of one train_s1 step (bs 16) on a FIXED synthetic scene -- SURVEY.md 8d's: 1280-face icospheres with
    0.05 vertex noise, camera scale U(0.6, 0.9), translation U(-0.1, 0.1), random rotation, seed 0 -- identical every run and
    comparable across builds and rounds.  us per launch."""
    from umr_amd import functional as UF
    from umr_amd.mesh import create_sphere
    g = torch.Generator().manual_seed(0)
    v, f = create_sphere(3)
    verts = torch.from_numpy(v).float()[None].repeat(2 * N, 1, 1)
    verts = verts + 0.05 * torch.randn(verts.shape, generator=g)
    faces = torch.from_numpy(f).long()[None].repeat(2 * N, 1, 1)
    sc_ = scale[0] + (scale[1] - scale[0]) * torch.rand(2 * N, 1, generator=g)   # (drawn in the order of tests/helpers.py:scene -- the geometry
    tr_ = -0.1 + 0.2 * torch.rand(2 * N, 2, generator=g)         # tools/kernels.py times)
    q = torch.randn(2 * N, 4, generator=g)
    cams = torch.cat([sc_, tr_, q / q.norm(dim=1, keepdim=True)], 1)
    _, fv, _ = UF.project_faces(verts.to(dev), cams.to(dev), faces.int().to(dev), 5.0, -2.732)
    fv = fv.detach()
    return raster_launch_times(dev, fv[:N], fv, iters, seed=1)


SCENE_DIR = os.path.join(ROOT, "profiles", "scenes")


def frozen_scenes():
    """[(name, path)] of the frozen captures of what the training step really renders (bench.py --capture-scene: projected face
    vertices of the shared render's and the unseen-view silhouette's 16 views, written at the profile pass of a default run)."""
    if not os.path.isdir(SCENE_DIR):
        return []
    return [(f[:-4], os.path.join(SCENE_DIR, f)) for f in sorted(os.listdir(SCENE_DIR)) if f.endswith(".npz")]


def frozen_scene_kernel_times(dev, path, iters=20, real_values=False):
    import numpy as np
    z = np.load(path)
    t = lambda k: torch.from_numpy(np.ascontiguousarray(z[k]))
    real = real_values and "textures" in z.files and "grad_image" in z.files
    return raster_launch_times(dev, t("fv_shared"), t("fv_unseen"), iters, tex=t("textures") if real else None,
                               g_tex=t("grad_image") if real else None, seed=1, two_render_forms=False)


class SceneCapture:
    """_lib.TAP collector of ONE eager training step: the shared render's views (face vertices, texels, upstream gradient) and the
    unseen-view silhouette's."""

    def __init__(self):
        self.got = {}

    def __call__(self, name, d):
        if name == "raster_forward" and d.get("lean") and "fv_shared" not in self.got:
            self.got["fv_shared"] = d["face_vertices"].detach().clone()
            self.got["textures"] = d["textures"].detach().clone()
        elif name == "silhouette_forward" and "fv_unseen" not in self.got:
            self.got["fv_unseen"] = d["face_vertices"].detach().clone()
        elif name == "raster_backward_alpha_geometry" and "grad_image" not in self.got:
            self.got["grad_image"] = d["grad_image"].detach().clone()

    def save(self, path, meta):
        import numpy as np
        need = ("fv_shared", "fv_unseen", "textures", "grad_image")
        if any(k not in self.got for k in need):
            raise SystemExit("--capture-scene: the step made no shared render (got %s)" % sorted(self.got))
        os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
        # geometry exact (fp32: it decides every branch of the raster kernels); texels and the upstream gradient -- values only --
        # as fp16, to keep two captures inside gpurun's 64 MiB return limit
        np.savez_compressed(path, meta=json.dumps(meta), **{k: self.got[k].float().cpu().numpy().astype("float32" if k.startswith("fv") else "float16")
                                                            for k in need})


def build_hot_path_step(args, dev, world, tv, faces, outputs, batch):
    """The render-and-compare step alone (--model 0): every raster / loss kernel of one train_s1 step, forward and backward, on
    fixed network outputs.  -> (step_fn, check_replay | None); with args.graph (one GPU) step_fn replays ONE HIP graph of it and
    check_replay(where) raises unless the graph still reproduces the eager step's total."""
    from umr_amd.perceptual import PerceptualTextureLoss
    from umr_amd.train_step import RenderCompareS1
    rc = RenderCompareS1(tv.to(dev), faces.to(dev), args.image_size, texture_loss=PerceptualTextureLoss(dev),
                         epoch=args.epoch, share_mask_render=bool(args.share_mask_render)).to(dev)
    leaves = [outputs["delta_v"], outputs["cam"], outputs["tex_flow"]]
    last_terms = {}

    def step_fn():
        for l in leaves:
            l.grad = None
        outputs["pred_vs"] = outputs["mean_shape"][None] + outputs["delta_v"]
        total, terms = rc(outputs, batch)
        last_terms.update(terms)
        total.backward()
        if world > 1:   # hot-path-only mode has no parameters; exchange the (tiny) camera gradient sums
            import torch.distributed as dist
            dist.all_reduce(outputs["cam"].grad)
        return total

    if not (args.graph and world == 1):
        return step_fn, None
    # ~140 launches of a few microseconds each: eager, the step is host-enqueue bound.  Everything -- the first
    # eager step (module caches, MIOpen solver search, the leaves' AccumulateGrad nodes), the warm-up and the
    # capture -- runs on ONE side stream, so no node of the captured autograd pass is tied to the default stream;
    # what the warm-up allocated is dropped before the capture; the graph is then replayed on the timing stream.
    eager_step = step_fn
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        # fresh leaves created ON the side stream: a leaf's AccumulateGrad node is tied to the stream it was first
        # used on, and make_s1_inputs already used delta_v on the default stream -- the engine would then hop to
        # the default stream inside the capture (fork / join through events) for that leaf's accumulation
        outputs["pred_vs"] = None
        for k in ("delta_v", "cam", "tex_flow"):
            outputs[k] = outputs[k].detach().clone().requires_grad_(True)
        leaves[:] = [outputs["delta_v"], outputs["cam"], outputs["tex_flow"]]
        eager_total = float(eager_step())
        eager_terms = {k: float(v) for k, v in last_terms.items()}
        for _ in range(3):
            eager_step()
        outputs["pred_vs"] = None
        last_terms.clear()
        for l in leaves:
            l.grad = None
        torch.cuda.synchronize()
        hip_graph = torch.cuda.CUDAGraph()
        # single-threaded autograd: forward and backward launches of the captured step come from one host thread
        with torch.autograd.set_multithreading_enabled(False), torch.cuda.graph(hip_graph, stream=side):
            static_total = eager_step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    def check_replay(where):
        torch.cuda.synchronize()
        got = float(static_total)
        if not abs(got - eager_total) <= 1e-4 * abs(eager_total):
            raise SystemExit("HIP-graph replay (%s) does not reproduce the eager step: total %r vs %r, terms %r vs %r"
                             % (where, got, eager_total, {k: float(v) for k, v in last_terms.items()}, eager_terms))

    hip_graph.replay()
    check_replay("first replay")

    def replay_fn():
        hip_graph.replay()
        return static_total

    replay_fn.eager = eager_step
    return replay_fn, check_replay


def hot_path_submeasure(args, dev, steps=40, warmup=5):
    """config.hot_path_*: the render-and-compare step alone -- the part of the training step this library IS (every raster / loss
    kernel, forward and backward; the network, the optimizer and their MIOpen kernels excluded) -- replayed from one HIP graph on
    the same synthetic shard, timed like the headline (synchronise, K replays, synchronise).  The headline moves with MIOpen's
    fp32 convolutions (most of the step's GPU time); this figure moves with the kernels of this repository."""
    import copy
    from umr_amd.synthetic import make_s1_inputs
    a = copy.copy(args)
    a.graph = 1
    tv, faces, outputs, batch = make_s1_inputs(args.batch, args.image_size, args.subdivide, seed=100, device=dev)
    step, check = build_hot_path_step(a, dev, 1, tv, faces, outputs, batch)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    check("after the hot-path replays")
    return {"hot_path_images_per_s": args.batch * steps / dt, "hot_path_ms_per_step": 1e3 * dt / steps, "hot_path_steps": steps,
            "hot_path_scope": "render-and-compare step (all raster + loss kernels, fwd + bwd) from one HIP graph; network / Adam excluded"}


def main(device=None, backend="nccl"):
    """device / backend: the CPU suite runs this very function on host tensors over gloo with the wave64 emulation of the library
    (tests/bench_rank_on_emulator.py) to exercise the N > 1 plumbing without GPUs; every real run leaves them at their defaults."""
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # ONE JSON line on stdout, whatever the libraries underneath print: RCCL writes a version banner to the C-level stdout at
    # exit, MIOpen may chat -- keep a private handle on the real stdout for the result and send file descriptor 1 to stderr
    global _RESULT_OUT          # (after the self-launch decision: the ranks it starts inherit the real stdout)
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert device is not None or torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU path)"
    if world > 1:
        torch.set_num_threads(max(1, min(torch.get_num_threads(), (os.cpu_count() or 8) // world)))
        # N ranks start at once on a fresh node: each gets its own MIOpen user database / kernel cache, so the first-use solver
        # searches of the ranks do not write the same sqlite files concurrently (set before MIOpen creates its first handle)
        for var, sub_ in (("MIOPEN_USER_DB_PATH", "db"), ("MIOPEN_CUSTOM_CACHE_DIR", "cache")):
            d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "umr_miopen_%s_rank%d" % (sub_, local))
            os.makedirs(d, exist_ok=True)
            os.environ.setdefault(var, d)
    if device is None:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        dev = torch.device(device)
    if world > 1 or args.force_ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if args.master_port > 0:
                os.environ["MASTER_PORT"] = str(args.master_port)
            elif world == 1:            # --force-ddp outside a launcher: a port that is free right now, not a fixed one
                import socket
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
                    s_.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            else:                       # several ranks cannot agree on a random port by themselves
                raise SystemExit("bench.py: WORLD_SIZE=%d but no MASTER_PORT in the environment: start the ranks with torchrun / "
                                 "`python bench.py --gpus N` (which picks a free port), or pass --master-port" % world)
        # (graph capture of the step's collectives: no asynchronous error handling from the watchdog thread, as torch's notes on
        # CUDA graphs with NCCL ask for)
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        dist.init_process_group(backend, rank=rank, world_size=world,
                                device_id=(dev if dev.type == "cuda" else None))   # "nccl" = RCCL on ROCm

    from umr_amd import _lib
    from umr_amd.synthetic import make_s1_inputs
    from umr_amd.train_step import RenderCompareS1
    have_model = os.path.exists(os.path.join(ROOT, "umr_amd", "model.py"))
    use_model = have_model if args.model < 0 else bool(args.model)

    torch.manual_seed(1234 + rank)   # per-rank synthetic shard (weak scaling: fixed per-GPU batch)
    tv, faces, outputs, batch = make_s1_inputs(args.batch, args.image_size, args.subdivide, seed=100 + rank, device=dev)
    def build_step():
        if args.workload == "s2":
            from umr_amd.model import build_training_step_s2
            return build_training_step_s2(args, dev, 2 if args.force_ddp else world)
        from umr_amd.model import build_training_step
        return build_training_step(tv, faces, args, dev, 2 if args.force_ddp else world)

    if args.grad_sync == "ddp" and (world > 1 or args.force_ddp):
        args.graph = 0
    if args.graph < 0:          # default: graph replay at every N (device runs only: the CPU suite's emulator has no graphs)
        args.graph = 1 if dev.type == "cuda" else 0
    step_fn = None
    check_replay = None
    if args.workload == "s2":
        use_model = True
    if use_model:
        step_fn = build_step()
    else:
        step_fn, check_replay = build_hot_path_step(args, dev, world, tv, faces, outputs, batch)

    whole_graph = None
    eager_host_ms = None
    early_prof = None
    graph_check = None
    if args.graph and use_model:
        # The WHOLE training step -- distance transform, MeshNet forward, every raster / loss kernel, backward, fused Adam with its
        # on-device learning-rate schedule -- captured once into one HIP graph and replayed: the eager step's host enqueue time
        # (config.eager_host_enqueue_ms_per_step) leaves the timed region.  Same recipe as the hot-path capture
        # (build_hot_path_step): first eager steps (MIOpen solver search, caches), warm-up and capture on ONE side stream;
        # gradients dropped before the capture so that they become graph-private allocations.
        eager_step = step_fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                for _ in range(max(3, args.warmup)):
                    eager_step()
                torch.cuda.synchronize()
                t_e = time.perf_counter()                # what the eager step costs the HOST (enqueue only): three more steps after
                for _ in range(3):                       # the warm-up (first-use solver searches are behind us), profiling off
                    eager_step()
                eager_host_ms = 1e3 * (time.perf_counter() - t_e) / 3
                torch.cuda.synchronize()
                # ... these five eager steps are also a FIRST roofline pass (the library's HIP events around the raster main
                # kernels): training has barely moved the meshes yet, so the figures repeat from run to run, which the pass after
                # the timed steps -- the scene of step 40+ of a GAN-driven trajectory -- does not (+- 40 %)
                _lib.profile_enable(True)
                for k in range(4):
                    _lib.profile_collect(k)
                for _ in range(5):
                    eager_step()
                torch.cuda.synchronize()
                _lib.profile_enable(False)
                early_prof = {k: _lib.profile_collect(k) for k in range(4)}
                whole_graph = torch.cuda.CUDAGraph()
                # (with a process group: its watchdog thread polls events of earlier collectives -- only THIS thread's calls
                # belong to the capture)
                mode = {"capture_error_mode": "thread_local"} if (world > 1 or args.force_ddp) else {}
                if mode:
                    # ... and let the watchdog retire the warm-up's (completed) collectives first: it polls the end event of every
                    # work still on its list every 100 ms, HIP refuses a query on an event whose STREAM is capturing -- the process
                    # group's own stream joins the capture with the first bucket -- even when the event was recorded before the
                    # capture began (hipErrorCapturedEvent), and the watchdog then takes the process down (seen once in three runs)
                    torch.cuda.synchronize()
                    time.sleep(1.0)
                with torch.autograd.set_multithreading_enabled(False), torch.cuda.graph(whole_graph, stream=side, **mode):
                    static_loss = eager_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            whole_graph.replay()
            torch.cuda.synchronize()
            if not math.isfinite(float(static_loss)):
                raise RuntimeError("first replay of the captured step returned a non-finite loss")
            # ... and a replay must BE the step: every gradient tensor of a replay against the eager step's from the same saved
            # state (umr_amd/graph_check.py: captured memset nodes do not replay on this stack, and a library kernel that relies
            # on one leaves garbage in the replayed step only)
            from umr_amd.graph_check import replay_matches_eager
            graph_check = replay_matches_eager(eager_step, whole_graph, static_loss, dev)
            if not graph_check["ok"]:
                raise RuntimeError("a replay of the captured step does not reproduce the eager step: %r" % (graph_check,))

            def step_fn():
                whole_graph.replay()
                return static_loss
            step_fn.model = eager_step.model
        except Exception as ex:     # noqa: BLE001 -- report, fall back to the eager step (the line then says hip_graph: false)
            sys.stderr.write("bench.py: whole-step HIP-graph capture failed (%s: %s); timing the eager step\n" % (type(ex).__name__, ex))
            whole_graph = None
            torch.cuda.synchronize()
        if world > 1:               # every rank replays, or none does
            import torch.distributed as dist
            ok_all = torch.tensor([1.0 if whole_graph is not None else 0.0], device=dev)
            dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)
            if whole_graph is not None and not bool(ok_all.item()):
                sys.stderr.write("bench.py: another rank could not capture the step; rank %d times the eager step too\n" % rank)
                whole_graph = None
        if whole_graph is None:
            args.graph = 0
            step_fn = build_step()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def measure():
        for _ in range(args.warmup):
            step_fn()
        barrier()
        t0 = time.perf_counter()
        history = []
        for _ in range(args.steps):
            loss = step_fn()
            history.append(loss)                # device scalars, read only if the run has to be discarded
        host_dt = time.perf_counter() - t0      # time to ENQUEUE the steps (host side); dt below includes the drain
        barrier()
        measure.history = history
        return loss, host_dt, time.perf_counter() - t0

    # A measurement that ends with a non-finite loss is discarded and repeated on a freshly initialised model (all ranks decide
    # together; config.discarded_nonfinite_runs counts them, the per-step loss history goes to stderr): after a NaN every
    # render degenerates (NaN geometry) and the timing means nothing.  Round 2 saw one such process in ~20; the cause -- a face
    # edge seen end-on makes the reference's own `den` exactly 0, and the inside branch of eval_pair did not skip that edge the
    # way the reference's min-over-edges does -- is fixed (HISTORY.md section 5, tests/test_gpu_zz_collapsed_edges.py); the
    # guard stays.
    discarded = 0
    while True:
        loss, host_dt, dt = measure()
        if not use_model:
            # nothing trains in hot-path-only mode: no trajectory to diverge.  (Also: in --graph mode the tensors the
            # graph owns are only ever read back with .item() -- an eager torch op on `loss` here, torch.isfinite, was
            # reproducibly followed by replays whose last reduction returned a stale value; check_replay guards that.)
            break
        if whole_graph is not None:     # graph-owned tensor: read it with .item() only (see check_replay above)
            ok = torch.tensor([1.0 if math.isfinite(float(loss)) else 0.0], device=dev)
        else:
            ok = torch.isfinite(loss.detach()).reshape(1).to(torch.float32)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if bool(ok.item()) or discarded >= 2 or not use_model or whole_graph is not None:
            break
        discarded += 1
        if rank == 0 and getattr(step_fn, "watch", None):    # UMR_WATCH_TERMS=1: which term went non-finite first
            for i, (names, vals) in enumerate(step_fn.watch):
                v = vals.tolist()
                if not all(math.isfinite(x) for x in v) or i < 2:
                    sys.stderr.write("bench.py: step %d terms %s\n" % (i, " ".join("%s=%.4g" % (n, x) for n, x in
                                                                                    zip(list(names) + ["|delta_v|", "|cams|", "|flow|"], v))))
                    if not all(math.isfinite(x) for x in v):
                        break
        if rank == 0:     # what the discarded trajectory looked like (stderr: the JSON line on stdout stays alone)
            hist = [float(x) for x in torch.stack([h.detach().reshape(()) for h in measure.history]).tolist()]
            sys.stderr.write("bench.py: measurement %d discarded, loss per timed step: %s\n"
                             % (discarded, " ".join("%.4g" % v for v in hist)))
        torch.manual_seed(4321 + 17 * discarded + rank)
        step_fn = build_step()
    if args.graph and not use_model and world == 1:
        check_replay("after the timed replays")      # back-to-back replays must still compute the eager step's numbers
    # hot-path-only mode renders the SAME scene every step: the per-step totals may differ by float-atomic summation order
    # only.  Their relative spread over the timed steps is reported (config.hot_path_loss_spread) -- a sporadic wrong result
    # of any kernel on the path would show here.
    loss_spread = None
    if not use_model and not args.graph:
        hist = torch.stack([h.detach().reshape(()) for h in measure.history]).double()
        loss_spread = float(((hist.max() - hist.min()) / hist.mean().abs()).item()) if bool(torch.isfinite(hist).all()) else float("nan")
    # roofline pass: the same steps again, untimed, with the library recording a HIP-event pair around every raster
    # main kernel on its launch stream (event creation / bookkeeping stays out of `value`)
    _lib.profile_enable(True)
    for k in range(4):
        _lib.profile_collect(k)
    # (graph replays do not pass through the C ABI's event scope: the eager form of the same step is profiled)
    prof_step = eager_step if whole_graph is not None else getattr(step_fn, "eager", step_fn)
    capture = SceneCapture() if (args.capture_scene and rank == 0) else None
    for i_ in range(max(0, args.profile_steps)):
        if capture is not None and i_ == 0:
            _lib.TAP = capture
        prof_step()
        _lib.TAP = None
    if capture is not None:
        torch.cuda.synchronize()
        capture.save(args.capture_scene, {"bench_args": vars(args), "lib_build_id": _lib.build_id(),
                                          "taken_at": "first step of the profile pass, after %d warm-up + %d timed steps" % (args.warmup, args.steps)})
    barrier()
    _lib.profile_enable(False)
    prof = {k: _lib.profile_collect(k) for k in range(4)}   # 0 fwd, 1 bwd, 2 silhouette/id fwd, 3 silhouette bwd
    if args.graph and not use_model and world == 1:
        check_replay("after the profile pass")
    b_ms, b_n, b_bytes = prof[1]
    f_ms, f_n, f_bytes = prof[0]

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    # evidence that the gradient exchange happened: after DDP's all-reduce every rank holds the same averaged
    # gradients, so sum|g| agrees across ranks to the last bit (spread 0.0); reported with the rank count
    grad_info = {}
    model = getattr(step_fn, "model", None)
    if model is not None:
        gs = [p.grad.detach().double().abs().sum() for p in model.parameters() if p.grad is not None]
        cs = torch.stack(gs).sum().reshape(1) if gs else torch.zeros(1, device=dev, dtype=torch.float64)
        lo, hi = cs.clone(), cs.clone()
        if world > 1:
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        grad_info = {"grad_abs_sum": float(cs.item()), "grad_abs_sum_spread_over_ranks": float((hi - lo).item())}
    # the gradient exchange on its own: the model's gradient bytes all-reduced in DDP's bucket size, timed with HIP events
    # on every rank (max over ranks).  With one rank (--force-ddp) this is RCCL's launch + local-copy floor of the path.
    ar_info = {}
    if (world > 1 or args.force_ddp) and dev.type == "cuda":
        import torch.distributed as dist
        from umr_amd.parallel import BUCKET_MB
        nbytes = sum(p.numel() * 4 for p in model.parameters() if p.requires_grad) if model is not None else 337 * 1024 * 1024
        flat = torch.zeros(nbytes // 4, device=dev)
        per = BUCKET_MB * 1024 * 1024 // 4
        buckets = [flat[i:i + per] for i in range(0, flat.numel(), per)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(7):
            if it == 2:
                barrier(); e0.record()
            for bk in buckets:
                dist.all_reduce(bk)
        e1.record(); torch.cuda.synchronize()
        ar = torch.tensor([e0.elapsed_time(e1) / 5.0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        ar_info = {"allreduce_bytes": int(nbytes), "ddp_bucket_mb": BUCKET_MB, "ddp_buckets": len(buckets),
                   "allreduce_ms_standalone": float(ar.item()),
                   "allreduce_bus_GBs_per_rank": (2.0 * (world - 1) / max(world, 1)) * nbytes / 1e9 / (float(ar.item()) / 1e3) if world > 1 else 0.0}
        del flat, buckets
    if rank != 0:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return

    images = args.batch * world * args.steps
    # HBM traffic of the same kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of
    # this very command, tools/collect_traffic.sh); bench.py cannot run the profiler on itself, so the committed
    # per-launch figure is attached when it was measured for the same workload, else null.
    traffic, valu, traffic_note = None, {}, "no profiles/traffic.json"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        same_wl = args.workload == "s1" and tj.get("workload") == [args.batch, args.image_size, args.subdivide, bool(use_model)]
        if tj.get("build_id") != _lib.build_id():
            traffic_note = "profiles/traffic.json was measured on another build of libumr_hip.so (%s): not attached" % tj.get("build_id")
        elif not same_wl:
            traffic_note = "profiles/traffic.json was measured for another workload: not attached"
        else:
            traffic = tj.get("raster_backward_bytes_per_launch")
            traffic_note = "PMC FETCH_SIZE (x%.2f calibrated) + WRITE_SIZE per launch, tools/collect_traffic.sh on build %s" % (
                tj.get("calibration", {}).get("fetch_correction_factor", 2.0), tj.get("build_id"))
            for kname, key in (("k_raster_backward_fm<1", "backward"), ("k_raster_backward_fm_ag<1", "backward"), ("k_raster_backward_fm_agp<1", "backward"),
                               ("k_raster_forward<1", "forward_kernel"), ("k_raster_forward<2", "silhouette_forward"),
                               ("k_raster_backward_fm<2", "silhouette_backward"), ("k_raster_backward_fm_quads<2", "silhouette_backward")):
                for kn, e in tj.get("kernels", {}).items():
                    if kname in kn and e.get("valu"):
                        v = e["valu"]
                        valu[key] = {"issued_wave_instr": v.get("issued_wave_instr"), "useful_lane_instr": v.get("useful_lane_instr"),
                                     "lane_use": v.get("lane_use"), "frac_of_peak": v.get("frac_of_peak"),
                                     "peak_wave_instr_per_s": v.get("peak_wave_instr_per_s"), "hbm_bytes_per_launch": e.get("hbm_bytes_per_launch")}
    rccl = {}
    if world > 1 or args.force_ddp:
        import torch.distributed as dist
        rccl = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend()}
    rccl.update(grad_info)
    rccl.update(ar_info)
    net_note = ("; MeshNet fwd/bwd + %sAdam" % ("RCCL gradient all-reduce over %d ranks + " % world if world > 1 else "")
                if use_model else "; network excluded")
    if args.workload == "s1":
        wl = ("train_s1 CUB-shaped bs=%d/GPU %dx%d (IS=%d) %d-face icosphere: the reference's 4 raster fwd + 3 bwd per image as %s "
              "launches (textured soft-max render with p2f whose alpha channel IS the mask render and whose visits carry the hard "
              "visibility render; unseen-view silhouette) + IoU / AlexNet-perceptual "
              "texture / texture-dt / tex-cycle / Laplacian / flatten / GAN losses, fwd+bwd, epoch %d%s"
              % (args.batch, args.image_size, args.image_size, 2 * args.image_size, faces.shape[0],
                 "2 fwd + 2 bwd (ONE backward pass for the shared render's two gradients)" if args.share_mask_render else
                 "2 fwd + 2 bwd (mask and unseen-view silhouettes in one 2B-view launch each way)", args.epoch, net_note))
    else:
        wl = ("train_s2 CUB-shaped bs=%d/GPU %dx%d (IS=%d) %d-face icosphere, 8 camera hypotheses: the reference's 22 raster fwd + 21 bwd per "
              "image as %s (8 textured hypothesis renders whose alpha channels are the 8 mask renders, 1 visibility, 1 unseen "
              "view, 2 part renders carrying the reference's 4) + mask / "
              "AlexNet-perceptual texture / tex-cycle / part / chamfer losses, fwd+bwd%s"
              % (args.batch, args.image_size, args.image_size, 2 * args.image_size, faces.shape[0],
                 "12 fwd + 11 bwd" if args.share_mask_render else "20 fwd + 19 bwd", net_note))

    def kernel_line(k):
        ms, n, nbytes = prof[k]
        gbs = (nbytes / 1e9) / (ms / 1e3) if ms > 0 else 0.0
        out_ = {"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "launches": n, "avg_us": (1e3 * ms / n) if n else None,
                "alg_bytes_per_launch": (nbytes / n) if n else None}
        if early_prof is not None and early_prof[k][1]:      # the same launches in the five eager steps BEFORE the capture
            ems, en, eb = early_prof[k]
            out_["first_steps"] = {"avg_us": 1e3 * ems / en, "launches": en, "frac": (eb / 1e9) / (ems / 1e3) / HBM_PEAK_GBS}
        return out_

    out = {
        "metric": METRIC, "value": images / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": dict({"workload": wl, "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                        "includes_network": use_model, "host_enqueue_ms_per_step": 1e3 * host_dt / args.steps,
                        "lib_build_id": _lib.build_id(),      # source hash of the libumr_hip.so that ran (umr_amd/build.py)
                        # host time to enqueue ONE eager step (Python + dispatcher + ~1000 launches), measured before the capture:
                        # what the timed region would be bound by without the graph
                        "eager_host_enqueue_ms_per_step": eager_host_ms,
                        "final_loss": float(loss.detach()), "discarded_nonfinite_runs": discarded,
                        "hip_graph": bool((args.graph and not use_model and world == 1) or whole_graph is not None),
                        "hip_graph_scope": ("whole training step (network + losses + Adam)" if whole_graph is not None else
                                            ("render-and-compare step" if (args.graph and not use_model and world == 1) else None)),
                        # every gradient tensor of one replay against the eager step's from the same saved state (graph_check.py)
                        "hip_graph_replay_vs_eager": ({k: graph_check[k] for k in ("ok", "tensors", "compared", "bad", "loss_replay", "loss_eager")}
                                                      if graph_check else None),
                        "hot_path_loss_spread": loss_spread}, **rccl),
    }
    # ---- roofline -------------------------------------------------------------------------------------------------------------
    # Dominant raster-backward kernel of the step: the ONE backward pass of the shared mask / texture render (alpha gradient ->
    # vertices, rgb gradient -> texels, pooled gradient in; with --share-mask-render 0 the texel-gradient-only backward).
    # `achieved` = algorithmic bytes of THAT variant (HISTORY.md 4.6: SURVEY 8d's rule -- each op-boundary buffer the variant
    # touches, once) / its mean HIP-event duration.  Round 6: the durations `avg_us / achieved / frac` are built from come from
    # a DETERMINISTIC pass -- the same kernel, 20 launches, on the frozen captures of what the training step really renders
    # (profiles/scenes/live_s1_*.npz, mean over the scenes; per scene and for the regular SURVEY 8d scene under `scenes`) -- and
    # repeat from run to run; what THIS run's trajectory happened to render at its profile pass (+-40 % between runs of one build:
    # float-atomic summation order steers a GAN-driven trajectory) stays on the line as `live`.
    # `valu`: the resource that actually binds these kernels (HISTORY.md 4.6) -- issued wave64 VALU instructions per launch,
    # the share of their lanes that was active, and the issue rate against the fp32 vector peak (1228.9 G wave-instr/s),
    # from SQ PMC passes of this command on this build (profiles/traffic.json; absent when that file is stale).
    scenes = {}
    if dev.type == "cuda" and args.fixed_scene and args.workload == "s1" and args.image_size == 256 and args.subdivide == 3:
        for name_, path_ in frozen_scenes():
            scenes[name_] = frozen_scene_kernel_times(dev, path_)
        scenes["survey_8d"] = fixed_scene_kernel_times(dev)
    live_names = [n_ for n_ in scenes if n_ != "survey_8d"]
    det_names = live_names or [n_ for n_ in scenes]

    def det_line(prefix, live_k):
        """Deterministic figure of the kernel whose scene-timing key starts with `prefix` (mean over the frozen live scenes, or the
        8d scene when no capture is committed) + the in-step figure of library profile slot `live_k` as `live`."""
        us, nbytes = [], []
        for n_ in det_names:
            for key_, v_ in scenes[n_].items():
                if key_.startswith(prefix) and not key_.startswith("_") and "planar" not in key_:
                    us.append(v_); nbytes.append(scenes[n_]["_alg_bytes"][key_])
        line = {"live": kernel_line(live_k)}
        if us:
            avg, b_ = sum(us) / len(us), sum(nbytes) / len(nbytes)
            gbs = b_ / 1e9 / (avg / 1e6)
            line.update({"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "avg_us": avg, "alg_bytes_per_launch": b_,
                         "launches": 20 * len(us), "source": "frozen scenes " + ", ".join(det_names)})
        else:               # nothing deterministic to show (other workload / CPU emulation): the live figures
            line.update(kernel_line(live_k), source="this run's profile pass (no frozen scene for this workload)")
        return line

    bwd_prefix = "shared_render_backward_one_pass" if args.share_mask_render else "texel_gradient_backward"
    fwd_prefix = "shared_render_forward_packed_state" if args.share_mask_render else "textured_forward"
    out["roofline"] = dict({"bound": "hbm", "kernel": ("k_raster_backward_fm_agp (shared render, packed saved state: d alpha -> vertices, d rgb -> texels)" if args.share_mask_render else
                                     "k_raster_backward_fm (textured render, texel gradients only)"), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "traffic": traffic, "traffic_source": traffic_note, "valu": valu.get("backward")},
                           **det_line(bwd_prefix, 1),
                           forward_kernel=dict(det_line(fwd_prefix, 0), valu=valu.get("forward_kernel")),
                           silhouette_forward=dict(det_line("silhouette_forward", 2), valu=valu.get("silhouette_forward")),
                           silhouette_backward=dict(det_line("silhouette_backward", 3), valu=valu.get("silhouette_backward")))
    if scenes:
        out["roofline"]["scenes"] = {n_: {k_: v_ for k_, v_ in d_.items() if not k_.startswith("_")} for n_, d_ in scenes.items()}
        out["roofline"]["scenes_note"] = ("us per launch, library-owned HIP events, 20 launches each; live_s1_*: projected face vertices of the "
                                          "16 + 16 views a default bench.py run rendered at step ~57 (bench.py --capture-scene); survey_8d: "
                                          "16 (32) x 1280-face icospheres, IS 512, TS 36, seed 0")

    # the step's raster launches together: summed HIP-event time of the raster main kernels per step of the profile pass
    n_prof = max(1, args.profile_steps)
    out["roofline"]["raster_kernels_us_per_step"] = round(1e3 * sum(prof[k][0] for k in range(4)) / n_prof, 1)
    out["roofline"]["raster_launches_per_step"] = round(sum(prof[k][1] for k in range(4)) / n_prof, 2)
    want_sub = (world == 1 and use_model and args.workload == "s1" and dev.type == "cuda") if args.hot_path_sub < 0 else bool(args.hot_path_sub)
    if want_sub:
        try:
            out["config"].update(hot_path_submeasure(args, dev))
        except (Exception, SystemExit) as ex:     # noqa: BLE001 -- a sub-measurement: report, keep the headline
            out["config"]["hot_path_error"] = "%s: %s" % (type(ex).__name__, ex)
    if "survey_8d" in scenes:       # (kept under its round-5 name as well)
        out["roofline"]["fixed_scene_us"] = dict(out["roofline"]["scenes"]["survey_8d"],
                                                 scene="SURVEY 8d: 16 (32) x 1280-face icospheres, IS 512, TS 36, seed 0 -- identical every run")
    want_cpu = (world == 1 and args.workload == "s1") if args.cpu_baseline < 0 else bool(args.cpu_baseline)
    if want_cpu:
        from oracle import softras
        n = args.cpu_sample or max(1, min(args.batch, softras.max_threads() // 2))
        out["cpu_baseline"] = cpu_baseline(args, n)
        out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out), file=_RESULT_OUT, flush=True)
    if world > 1 or args.force_ddp:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
