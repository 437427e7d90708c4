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


class BucketedGradSync:
    """The gradient exchange of one data-parallel training step, written out: one flat fp32 buffer laid out in reverse parameter
    order (the order backward produces gradients) holds every parameter's gradient and is cut into buckets of `bucket_mb`; a
    post-accumulate hook per parameter COPIES the gradient autograd just produced into the parameter's slice of the buffer,
    makes `.grad` that slice and counts its bucket down; a bucket whose gradients are all in is all-reduced at once --
    asynchronously (RCCL runs it on the process group's own stream), in bucket order on every rank, while backward goes on --
    and `finish()` waits for all of them before the optimizer reads the gradients.

    Why not torch's DistributedDataParallel here: this is the same schedule (64 MB buckets overlapped with backward, gradients
    averaged over ranks) with no host-side reducer state, so the WHOLE step -- forward, backward, the bucket all-reduces, Adam --
    captures into ONE HIP graph at every world size (bench.py: the N = 1 point of a scaling curve and the N > 1 points are then
    measured the same way; under DDP the N > 1 step was eager and host-enqueue-bound).  wrap_ddp stays for callers that want DDP.

    Use:   sync = BucketedGradSync(model.parameters(), world)
           sync.begin(); (loss * sync.loss_scale).backward(); sync.finish(); opt.step()       # ONE backward per begin() / finish()
    loss_scale = 1 / world: the all-reduce SUMS, so pre-scaled gradients come out averaged (what DataParallel's gather + mean
    and DDP both compute).  Parameters that received no gradient in a step contribute zeros.  (First form, round 6: `.grad` were
    permanent views, zeroed by one fill and accumulated into in place -- a fill and a read-modify-write of 337 MB per step where
    this form makes one copy: 0.2 ms of a 13 ms step.)"""

    def __init__(self, params, world, bucket_mb=BUCKET_MB, group=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("BucketedGradSync: no trainable parameters")
        self.world, self.group = max(1, int(world)), group
        self.loss_scale = 1.0 / self.world
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        cap = max(1, int(bucket_mb * 1024 * 1024) // 4)
        self.buckets, self._bucket_of, self._size, self._view = [], {}, [], {}
        off, start, count = 0, 0, 0
        for p in reversed(self.params):
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("BucketedGradSync: fp32 parameters on one device only")
            n = p.numel()
            self._view[id(p)] = self.flat[off:off + n].view_as(p)
            p.grad = self._view[id(p)]
            self._bucket_of[id(p)] = len(self.buckets)
            off += n
            count += 1
            if off - start >= cap:
                self.buckets.append(self.flat[start:off]); self._size.append(count)
                start, count = off, 0
        if off > start:
            self.buckets.append(self.flat[start:off]); self._size.append(count)
        self._pending, self._next, self._works, self._seen = list(self._size), 0, [], set()
        self._hooks = [p.register_post_accumulate_grad_hook(self._ready) for p in self.params]

    def begin(self):
        """Start of a step: drop every gradient (autograd then hands each hook a fresh tensor) and re-arm the buckets."""
        for p in self.params:
            p.grad = None
        self._pending, self._next, self._works, self._seen = list(self._size), 0, [], set()

    def _launch(self):
        while self._next < len(self.buckets) and self._pending[self._next] <= 0:
            if self.world > 1 or dist.is_initialized():
                self._works.append(dist.all_reduce(self.buckets[self._next], group=self.group, async_op=True))
            self._next += 1

    def _ready(self, p):
        b = self._bucket_of[id(p)]
        if id(p) in self._seen:         # a second backward since begin(): the bucket may already have been exchanged
            raise RuntimeError("BucketedGradSync: a parameter's gradient arrived twice in one step -- one backward() per "
                               "begin() / finish() (accumulate over micro-batches in the loss, or call begin() again)")
        self._seen.add(id(p))
        view = self._view[id(p)]
        if p.grad is not view:
            view.copy_(p.grad)
            p.grad = view
        self._pending[b] -= 1
        self._launch()

    def finish(self):
        """After backward: parameters without a gradient this step contribute zeros; launch the buckets that are still waiting,
        in order, and make the current stream wait for every all-reduce."""
        for p in self.params:
            if id(p) not in self._seen:
                view = self._view[id(p)]
                view.zero_()
                p.grad = view
                self._seen.add(id(p))
                self._pending[self._bucket_of[id(p)]] -= 1
        self._launch()
        for w in self._works:
            w.wait()
        self._works = []

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


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
