"""CPU, world_size 2, gloo: the data-parallel harness (umr_amd/parallel.py) around the re-hosted MeshNet.
The render-and-compare kernels have no CPU path by design, so the loss here is a plain torch surrogate over the
network outputs; what is checked is the distributed logic: sharding, bucketed gradient averaging, identical
replicas after an optimizer step, scalar reduction."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _surrogate_loss(out):
    return (out["delta_v"].pow(2).mean() + out["cam"].pow(2).mean() + out["tex_flow"].abs().mean()
            + 0.1 * out["mean"].pow(2).mean() + 0.0 * out["cam_probs"].sum())


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from umr_amd import parallel
    from umr_amd.model import MeshNet, default_opts
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)                      # identical initial replicas
    opts = default_opts(subdivide=1, nz_feat=32, z_dim=16)
    net = MeshNet((64, 64), opts, nz_feat=32)
    net.eval()                                # BatchNorm in eval: the single-process comparison needs batch-size independence
    full = torch.randn(4, 3, 64, 64, generator=torch.Generator().manual_seed(7))
    mine = parallel.shard(full, rank, world)
    assert mine.shape[0] == 2
    with pytest.raises(ValueError):           # a batch that does not split evenly would hang DDP on uneven inputs
        parallel.shard(torch.zeros(5, 1), rank, world)
    ddp = parallel.wrap_ddp(net, None, world)
    torch.manual_seed(100)                    # same VAE noise on both ranks -> comparable with the 1-process run below
    loss = _surrogate_loss(ddp(mine))
    loss.backward()
    trainable = lambda m: [p for p in m.parameters() if p.requires_grad]   # frozen: resnet fc (never evaluated)
    assert all(p.grad is not None for p in trainable(net))                 # DDP saw a gradient for every trainable parameter
    g = torch.cat([p.grad.flatten() for p in trainable(net)])
    # reference: average of the two shards' gradients computed without DDP
    ref_net = MeshNet((64, 64), opts, nz_feat=32)
    ref_net.load_state_dict(net.state_dict())
    ref_net.eval()
    acc = None
    for rr in range(world):
        ref_net.zero_grad()
        torch.manual_seed(100)
        _surrogate_loss(ref_net(parallel.shard(full, rr, world))).backward()
        gr = torch.cat([p.grad.flatten() for p in trainable(ref_net)])
        acc = gr if acc is None else acc + gr
    err = float((g - acc / world).abs().max())
    opt = torch.optim.Adam(trainable(net), lr=1e-3)
    opt.step()
    checksum = float(sum(p.double().sum() for p in net.parameters()))
    means = parallel.mean_scalars({"loss": float(loss), "rank": float(rank)}, world)
    # stage-1 template update: per-rank feature sums over different numbers of samples -> identical new template
    feats = torch.randn(5, 16, generator=torch.Generator().manual_seed(3))
    local = feats[:2] if rank == 0 else feats[2:]
    before = net.mean_v.clone()
    parallel.update_template(net, local.sum(0), local.shape[0], world)
    with torch.no_grad():
        expect = before + net.shape_predictor(feats.mean(0, keepdim=True)).view(-1, 3)
    terr = float((net.mean_v - expect).abs().max())
    # buffers are NOT re-broadcast every forward (replicas are kept identical by construction): a rank-local change to a
    # BatchNorm statistic survives the next forward, as under the reference's DataParallel-free per-rank statistics
    bn = net.encoder.resnet_conv.resnet.bn1
    bn.running_mean.fill_(float(rank + 1))
    with torch.no_grad():
        ddp(mine)
    assert float(bn.running_mean[0]) == float(rank + 1)
    q.put((rank, max(err, terr), checksum, means["rank"]))
    dist.destroy_process_group()


def test_ddp_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    assert all(e < 1e-5 for _, e, _, _ in res), res            # DDP grads == mean of per-shard grads
    assert abs(res[0][2] - res[1][2]) < 1e-9, res              # replicas identical after the step
    assert all(abs(m - 0.5) < 1e-12 for *_, m in res)          # scalar all-reduce mean


def _sync_worker(rank, world, port, q):
    """parallel.BucketedGradSync -- the gradient exchange bench.py's step runs by default (capturable into the step's HIP graph) --
    on two gloo ranks: tiny buckets (several per step, launched from the gradient hooks in bucket order), a parameter that gets no
    gradient, three steps with the flat buffer re-armed each time."""
    import torch.distributed as dist
    from umr_amd import parallel
    from umr_amd.model import MeshNet, default_opts
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_distributed("gloo")
    torch.manual_seed(0)
    opts = default_opts(subdivide=1, nz_feat=32, z_dim=16)
    net = MeshNet((64, 64), opts, nz_feat=32)
    net.eval()
    extra = torch.nn.Parameter(torch.ones(7))                  # never used: receives no gradient in any step
    params = list(net.parameters()) + [extra]
    full = torch.randn(4, 3, 64, 64, generator=torch.Generator().manual_seed(7))
    mine = parallel.shard(full, rank, world)
    sync = parallel.BucketedGradSync(params, world, bucket_mb=0.25)
    assert len(sync.buckets) > 3 and sum(b.numel() for b in sync.buckets) == sync.flat.numel()
    trainable = [p for p in params if p.requires_grad]
    assert all(p.grad is not None and p.grad.data_ptr() >= sync.flat.data_ptr() for p in trainable)   # views of the flat buffer
    opt = torch.optim.Adam(trainable, lr=1e-3)
    worst = 0.0
    for it in range(3):
        # reference: mean over the two shards' gradients, no communication, from the CURRENT parameters
        ref_net = MeshNet((64, 64), opts, nz_feat=32)
        ref_net.load_state_dict(net.state_dict())
        ref_net.eval()
        acc = None
        for rr in range(world):
            ref_net.zero_grad()
            torch.manual_seed(100 + it)
            _surrogate_loss(ref_net(parallel.shard(full, rr, world))).backward()
            gr = torch.cat([p.grad.flatten() for p in ref_net.parameters() if p.requires_grad])
            acc = gr if acc is None else acc + gr
        sync.begin()
        torch.manual_seed(100 + it)
        (_surrogate_loss(net(mine)) * sync.loss_scale).backward()
        sync.finish()
        g = torch.cat([p.grad.flatten() for p in net.parameters() if p.requires_grad])
        worst = max(worst, float((g - acc / world).abs().max()))
        assert float(extra.grad.abs().max()) == 0.0
        assert all(p.grad.data_ptr() >= sync.flat.data_ptr() for p in trainable)     # still views: nothing re-allocated a gradient
        opt.step()
    checksum = float(sum(p.double().sum() for p in net.parameters()))
    q.put((rank, worst, checksum))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_grad_sync_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    assert all(e < 1e-5 for _, e, _ in res), res               # gradients == mean of per-shard gradients, every step
    assert abs(res[0][2] - res[1][2]) < 1e-9, res              # replicas identical after three optimizer steps
