#!/usr/bin/env python3
"""bench.py -- train images/sec of the UMR render-and-compare hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic CUB-shaped input per GPU:
BASELINE.json configs[1] = train_s1, bs=16 per GPU, 256x256 images (512x512 internal raster),
642-vertex / 1280-face mesh: 4 raster forwards + 3 raster backwards per image plus every geometric /
image loss, forward and backward, producing gradients for vertices, cameras and texture flow
(umr_amd/train_step.py mirrors experiments/train_s1.py:177-265).  With --model (default when the
model module is present) the ResNet-18 MeshNet forward/backward, the RCCL gradient all-reduce and the
Adam step are inside the timed region too.  Inputs are resident in HBM before the clock starts.

Prints ONE JSON line (rank 0) with `roofline` (raster-backward kernel, HIP events recorded by
libumr_hip.so on the launch stream during the timed steps) and `cpu_baseline` (the CPU oracle =
reference algorithm, brute force, OpenMP over all host cores, on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
    return ap.parse_args()


def cpu_baseline(args, n_images):
    """The reference algorithm on host cores: oracle/ (C raster, OpenMP over all cores + torch-CPU losses),
    same train_s1 sequence, bounded sample of `n_images` images of the same workload."""
    from oracle import softras
    from oracle.train_step_ref import RenderCompareS1Ref
    from umr_amd.synthetic import make_s1_inputs
    cores = softras.max_threads()
    torch.set_num_threads(cores)
    tv, faces, outputs, batch = make_s1_inputs(n_images, args.image_size, args.subdivide, seed=1, device="cpu")
    step = RenderCompareS1Ref(tv, faces, args.image_size, n_threads=cores)
    t0 = time.perf_counter()
    total, _ = step(outputs, batch)
    total.backward()
    dt = time.perf_counter() - t0
    return {"value": n_images / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d image(s) of the same train_s1 step (4 raster fwd + 3 bwd per image + losses, fwd+bwd), "
                      "%.1f s wall; network excluded on the CPU side" % (n_images, dt)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with torch.distributed.run (see module docstring)")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.force_ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" = RCCL on ROCm

    from umr_amd import _lib
    from umr_amd.synthetic import make_s1_inputs
    from umr_amd.train_step import RenderCompareS1
    have_model = os.path.exists(os.path.join(ROOT, "umr_amd", "model.py"))
    use_model = have_model if args.model < 0 else bool(args.model)

    torch.manual_seed(1234 + rank)   # per-rank synthetic shard (weak scaling: fixed per-GPU batch)
    tv, faces, outputs, batch = make_s1_inputs(args.batch, args.image_size, args.subdivide, seed=100 + rank, device=dev)
    step_fn = None
    if args.workload == "s2":
        from umr_amd.model import build_training_step_s2
        use_model = True
        step_fn = build_training_step_s2(args, dev, 2 if args.force_ddp else world)
    elif use_model:
        from umr_amd.model import build_training_step
        step_fn = build_training_step(tv, faces, args, dev, 2 if args.force_ddp else world)
    else:
        rc = RenderCompareS1(tv.to(dev), faces.to(dev), args.image_size).to(dev)
        leaves = [outputs["delta_v"], outputs["cam"], outputs["tex_flow"]]

        def step_fn():
            for l in leaves:
                l.grad = None
            outputs["pred_vs"] = outputs["mean_shape"][None] + outputs["delta_v"]
            total, _ = rc(outputs, batch)
            total.backward()
            if world > 1:   # hot-path-only mode has no parameters; exchange the (tiny) camera gradient sums
                import torch.distributed as dist
                dist.all_reduce(outputs["cam"].grad)
            return total

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step_fn()
    barrier()
    _lib.profile_enable(True)
    _lib.profile_collect(0); _lib.profile_collect(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step_fn()
    host_dt = time.perf_counter() - t0          # time to ENQUEUE the steps (host side); dt below includes the drain
    barrier()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    b_ms, b_n, b_bytes = _lib.profile_collect(1)
    f_ms, f_n, f_bytes = _lib.profile_collect(0)

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank != 0:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return

    images = args.batch * world * args.steps
    achieved = (b_bytes / 1e9) / (b_ms / 1e3) if b_ms > 0 else 0.0
    # HBM traffic of the same kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of
    # this very command, tools/collect_traffic.sh); bench.py cannot run the profiler on itself, so the committed
    # per-launch figure is attached when it was measured for the same workload, else null.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if args.workload == "s1" and tj.get("workload") == [args.batch, args.image_size, args.subdivide, bool(use_model)]:
            traffic = tj.get("raster_backward_bytes_per_launch")
    out = {
        "metric": METRIC, "value": images / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("train_s1 CUB-shaped bs=%d/GPU %dx%d (IS=%d) %d-face icosphere: 4 raster fwd + 3 bwd "
                                "per image + IoU/texture/tex-cycle/Laplacian/flatten losses, fwd+bwd%s"
                                if args.workload == "s1" else
                                "train_s2 CUB-shaped bs=%d/GPU %dx%d (IS=%d) %d-face icosphere, 8 camera hypotheses: 22 raster "
                                "fwd + 21 bwd per image + mask/perceptual-texture/tex-cycle/part/chamfer losses, fwd+bwd%s")
                               % (args.batch, args.image_size, args.image_size, 2 * args.image_size, faces.shape[0],
                                  "; MeshNet fwd/bwd + RCCL all-reduce + Adam" if use_model else "; network excluded"),
                   "global_batch": args.batch * world, "parallelism": "dp%d" % world, "includes_network": use_model,
                   "host_enqueue_ms_per_step": 1e3 * host_dt / args.steps,
                   "final_loss": float(loss.detach())},
        "roofline": {"bound": "hbm", "kernel": "k_raster_backward", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "launches": b_n, "avg_us": (1e3 * b_ms / b_n) if b_n else None,
                     "alg_bytes_per_launch": (b_bytes / b_n) if b_n else None,
                     "forward_kernel": {"achieved": (f_bytes / 1e9) / (f_ms / 1e3) if f_ms > 0 else 0.0,
                                        "launches": f_n, "avg_us": (1e3 * f_ms / f_n) if f_n else None}},
    }
    want_cpu = (world == 1 and args.workload == "s1") if args.cpu_baseline < 0 else bool(args.cpu_baseline)
    if want_cpu:
        from oracle import softras
        n = args.cpu_sample or max(1, min(args.batch, softras.max_threads() // 2))
        out["cpu_baseline"] = cpu_baseline(args, n)
        out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out), flush=True)
    if world > 1 or args.force_ddp:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
