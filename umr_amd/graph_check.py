"""Does a captured HIP graph of a training step compute the training step?  A guard for callers that time or train from a
replayed graph (bench.py): on this ROCm stack a captured hipMemsetAsync node is not executed again on replay, and any library
kernel that counts on such a memset -- ATen's multi-workgroup reductions zero a semaphore that way (ATen/native/cuda/Reduce.cuh),
which is how a convolution's bias gradient over a large output is summed -- leaves garbage in every replay while the eager step
is correct (round 6: the texture decoder's 64x128 / 128x256 layers, values up to 1e38; umr_amd.model.Conv2d now sums those in
stages).  A replay and an eager step are run from ONE saved state (parameters, buffers, optimizer state, schedule counter, device
generator) and every gradient tensor is compared."""
import torch


def _state(step):
    model, opt = step.model, step.opt
    out = list(model.parameters()) + list(model.buffers())
    if getattr(step, "it_dev", None) is not None:
        out.append(step.it_dev)
    for st in opt.state.values():
        out += [v for _, v in sorted(st.items()) if torch.is_tensor(v)]
    return out + [g["lr"] for g in opt.param_groups if torch.is_tensor(g["lr"])]


def replay_matches_eager(step, graph, static_loss, device, rel_tol=0.3, significant=1e-3):
    """step: the eager closure the graph was captured from (attributes .model, .opt[, .it_dev]); graph: torch.cuda.CUDAGraph;
    static_loss: the graph's output tensor.  -> dict(ok, tensors, compared, bad [(name, rel)], loss_replay, loss_eager).
    A gradient tensor counts when its largest element is at least `significant` of the model's largest gradient element (the
    rest is rounding noise: biases in front of a BatchNorm); it is `bad` when replay and eager differ by more than `rel_tol` of
    its largest element and by more than 8x what two eager steps from that state differ by -- summation-order noise of the step's
    float atomics reaches tens of per cent on sums with heavy cancellation (camera heads), a stale or unwritten gradient is off by
    ~1 or by 1e20 in tensors whose eager noise is 1e-5.  Leaves the model one eager step past the saved state."""
    names = {id(p): n for n, p in step.model.named_parameters()}
    saved = [t.detach().clone() for t in _state(step)]
    rng = torch.cuda.get_rng_state(device)

    def restore():
        with torch.no_grad():
            for t, s in zip(_state(step), saved):
                t.copy_(s)
        torch.cuda.set_rng_state(rng, device)

    def grads():
        return {names[id(p)]: p.grad.detach().clone() for p in step.model.parameters() if p.grad is not None}

    graph.replay()
    torch.cuda.synchronize()
    loss_r, g_r = float(static_loss), grads()
    restore()
    float(step())
    torch.cuda.synchronize()
    g_e2 = grads()                      # a second eager step from the same state: what summation-order noise alone does to a tensor
    restore()
    loss_e = float(step())
    torch.cuda.synchronize()
    g_e = grads()
    gmax = max((float(t.abs().max()) for t in g_e.values()), default=0.0)
    bad, compared = [], 0
    for n, e in g_e.items():
        sc = float(e.abs().max())
        if n not in g_r or not sc >= significant * gmax:
            continue
        compared += 1
        rel = float((g_r[n] - e).abs().max()) / sc
        noise = float((g_e2[n] - e).abs().max()) / sc if n in g_e2 else 0.0
        # (camera-head gradients are sums over every vertex with heavy cancellation: two EAGER steps differ by tens of per cent
        # there; a tensor is bad when the replay is off by more than rel_tol AND by far more than the eager steps among themselves)
        if not rel <= max(rel_tol, 8.0 * noise):          # (NaN compares false: bad)
            bad.append((n, rel))
    ok = (not bad) and compared > 0 and abs(loss_r - loss_e) <= 1e-2 * max(1.0, abs(loss_e))
    return dict(ok=ok, tensors=len(g_e), compared=compared, bad=sorted(bad, key=lambda kv: -kv[1] if kv[1] == kv[1] else -1e300)[:8],
                loss_replay=loss_r, loss_eager=loss_e)
