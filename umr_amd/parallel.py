"""Data parallelism: one process per GPU + RCCL, replacing the reference's single-process nn.DataParallel
(experiments/train_s2.py:95-164: per-step replicate/scatter/gather of ~85 M parameters, GPU-0 serial sections).

Every rank runs the model AND the whole render-and-compare path on its own shard of the batch; the only
collective is the bucketed gradient all-reduce (sum / world) that torch DDP overlaps with backward.  On ROCm the
"nccl" backend is RCCL; over xGMI each all-reduce bucket is per-link bound, so buckets are kept large.
BatchNorm statistics stay per rank (what DataParallel does as well).
"""
import os

import torch
import torch.distributed as dist

BUCKET_MB = 64   # ~340 MB of fp32 grads -> 6 buckets; the first fires while the trunk is still in backward


def init_distributed(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the torchrun environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # several ranks cannot agree on a free port by themselves, and a fixed default collides as soon as two jobs share
            # a node: the launcher has to name it (torchrun does; `python bench.py --gpus N` picks a free one: free_port())
            raise RuntimeError("init_distributed: WORLD_SIZE=%d but no MASTER_PORT in the environment; start the ranks with "
                               "torchrun / torch.distributed.run, or export MASTER_PORT=<umr_amd.parallel.free_port()> for all "
                               "of them" % world)
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def free_port(host="127.0.0.1"):
    """A TCP port that is free right now on `host` -- for the launcher of a single-node job (bench.py's self-launch, tests)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((host, 0))
        return s.getsockname()[1]


def wrap_ddp(module, device, world):
    """broadcast_buffers=False: replicas start identical (same seed / same checkpoint) and every update to a buffer is
    either rank-local by design (BatchNorm running statistics, as under DataParallel) or made identically on all ranks
    (update_template below), so rank 0's buffers are not re-broadcast before every forward."""
    if world <= 1:
        return module
    if device is not None and torch.device(device).type == "cuda":
        return torch.nn.parallel.DistributedDataParallel(module, device_ids=[torch.device(device).index],
                                                         bucket_cap_mb=BUCKET_MB, gradient_as_bucket_view=True,
                                                         broadcast_buffers=False)
    return torch.nn.parallel.DistributedDataParallel(module, bucket_cap_mb=BUCKET_MB, broadcast_buffers=False)


def shard(tensor, rank, world):
    """Contiguous batch shard of rank `rank` (images are independent in every kernel and loss).  The batch
    must split evenly: short or empty shards would give DDP uneven inputs (hang / crash in the bucketed all-reduce)."""
    n = tensor.shape[0]
    if n % world:
        raise ValueError("shard: batch of %d does not divide over %d ranks (drop the remainder first)" % (n, world))
    per = n // world
    return tensor[rank * per:(rank + 1) * per]


def mean_scalars(values, world):
    """All-reduce a dict of python/torch scalars to their mean over ranks (logging only)."""
    if world <= 1:
        return {k: float(v) for k, v in values.items()}
    keys = sorted(values)
    t = torch.tensor([float(values[k]) for k in keys], dtype=torch.float64,
                     device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t)
    return {k: float(x) / world for k, x in zip(keys, t)}


@torch.no_grad()
def update_template(net, feat_sum, n_samples, world):
    """Stage-1 template update (experiments/train_s1.py:386-411) under data parallelism: every rank accumulates the
    encoder feature of ITS shard over an epoch; sum and count are all-reduced so all ranks feed the same dataset-mean
    feature to the shape predictor and add the same delta to `mean_v` (replicas stay identical without a broadcast).
    net: the un-wrapped MeshNet; feat_sum [z_dim] local sum of outputs['feat'] rows; n_samples local row count."""
    t = torch.cat([feat_sum.reshape(-1).double(), torch.tensor([float(n_samples)], dtype=torch.float64,
                                                               device=feat_sum.device)])
    if world > 1:
        dist.all_reduce(t)
    mean_feat = (t[:-1] / t[-1]).to(feat_sum.dtype).unsqueeze(0)
    delta_v = net.shape_predictor(mean_feat).view(-1, 3)
    net.mean_v += delta_v
    return delta_v
