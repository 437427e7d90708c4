"""CPU, world_size 2, gloo: bench.py's TIMED STEP itself -- umr_amd.model.build_training_step / build_training_step_s2: MeshNet
forward, the whole render-and-compare path through the product's kernels, backward with the bucketed gradient all-reduce, Adam -- on
two ranks, with the kernels running on the wave64 emulation of the library (tests/host_raster.py::emulated_product) instead of a
GPU.  tests/test_parallel_gloo.py checks the data-parallel harness around a surrogate loss; this runs the real thing (toy
size): different data per rank (seed 100 + rank, as in bench.py), identical replicas after every step, gradients that are the
mean over ranks, finite losses."""
import os
import socket
import types

import pytest
import torch
import torch.multiprocessing as mp

import host_raster as HR

pytestmark = pytest.mark.skipif(not HR.available(), reason="clang++ of the ROCm toolchain not present")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, stage, q, grad_sync="buckets"):
    import torch.distributed as dist
    from umr_amd import parallel
    from umr_amd.model import build_training_step, build_training_step_s2
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    parallel.init_distributed("gloo")
    torch.manual_seed(0)                                   # identical initial replicas, as bench.py
    args = types.SimpleNamespace(batch=2, image_size=64, subdivide=1, epoch=0, graph=0, grad_sync=grad_sync)
    with HR.emulated_product():
        step = (build_training_step(None, None, args, "cpu", world) if stage == 1 else build_training_step_s2(args, "cpu", world))
        losses, sums = [], []
        for _ in range(2):
            losses.append(float(step()))
            params = [p for p in step.model.parameters() if p.requires_grad]
            sums.append(float(sum(p.detach().double().sum() for p in params)))
        # the gradient every rank holds after backward is the all-reduced mean: identical on both ranks
        g = torch.cat([p.grad.flatten().double() for p in params if p.grad is not None])
        gsum, gabs = float(g.sum()), float(g.abs().sum())
    q.put((rank, losses, sums, gsum, gabs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("stage,grad_sync", [(1, "buckets"), (2, "buckets"), (1, "ddp")])
def test_bench_training_step_on_two_ranks(stage, grad_sync):
    """grad_sync: the step's gradient exchange -- parallel.BucketedGradSync (bench.py's default: flat gradient buffer, buckets
    all-reduced from the gradient hooks) or torch's DistributedDataParallel."""
    HR.lib(HR.build())                                     # built once, before the ranks start
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, stage, q, grad_sync)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    res, t0 = [], time.time()
    while len(res) < world:
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), "a rank died: exit codes %s" % [p.exitcode for p in procs]
            assert time.time() - t0 < 900, "timeout"
    res.sort()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, l0, s0, g0, a0), (_, l1, s1, g1, a1) = res
    assert all(map(lambda v: v == v and abs(v) < 1e30, l0 + l1)), (l0, l1)        # finite losses
    assert l0 != l1                                                                # the ranks really work on different shards
    assert a0 > 0 and abs(g0 - g1) <= 1e-9 * a0 and abs(a0 - a1) <= 1e-9 * a0, (g0, g1, a0, a1)   # all-reduced gradients
    for a, b in zip(s0, s1):
        assert abs(a - b) <= 1e-9 * max(1.0, abs(a)), (s0, s1)                     # identical replicas after every step
    assert s0[0] != s0[1]                                                          # and the optimizer did move them
