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
    rest is rounding noise: biases in front of a BatchNorm); it is `bad` when two replays both lie farther than `rel_tol` of its
    largest element, and farther than 4x the eager steps' own scatter, from every one of three eager steps -- a stale or unwritten
    gradient is off by ~1 or by 1e20 in every replay, in tensors whose eager scatter is 1e-5.  Leaves the model one eager step past
    the saved state."""
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

    def run_replay():
        restore()
        graph.replay()
        torch.cuda.synchronize()
        return float(static_loss), grads()

    def run_eager():
        restore()
        loss = float(step())
        torch.cuda.synchronize()
        return loss, grads()

    # two replays and three eager steps, all from the same state.  Gradient tensors that are sums with heavy cancellation (the
    # camera heads of train_s2: four numbers summed over the batch's views) differ by tens of per cent between two EAGER steps, so
    # a fixed tolerance cannot tell them from a stale gradient; what can is where the replays lie relative to the eager steps' own
    # scatter: a tensor is bad when BOTH replays are farther than rel_tol AND farther than 4x that scatter from EVERY eager step.
    (loss_r, g_r1), (_, g_r2) = run_replay(), run_replay()
    eager = [run_eager() for _ in range(3)]
    loss_e, g_e = eager[-1]
    gmax = max((float(t.abs().max()) for t in g_e.values()), default=0.0)
    bad, compared = [], 0
    for n, e in g_e.items():
        sc = float(e.abs().max())
        if n not in g_r1 or n not in g_r2 or not sc >= significant * gmax or any(n not in g for _, g in eager):
            continue
        compared += 1
        es = [g[n] for _, g in eager]
        scatter = max(float((es[i] - es[j]).abs().max()) for i in range(3) for j in range(i)) / sc
        thr = max(rel_tol, 4.0 * scatter)
        dist = [min(float((r[n] - x).abs().max()) for x in es) / sc for r in (g_r1, g_r2)]
        far = max(0.9, 20.0 * scatter)                      # ... or when ONE replay is off by about the tensor's whole scale (the
        if not (dist[0] <= thr or dist[1] <= thr) or not (dist[0] <= far and dist[1] <= far):   # defect is not in every replay)
            bad.append((n, max(dist)))                      # (NaN compares false: bad)
    ok = (not bad) and compared > 0 and abs(loss_r - loss_e) <= 1e-2 * max(1.0, abs(loss_e))
    return dict(ok=ok, tensors=len(g_e), compared=compared, bad=sorted(bad, key=lambda kv: -kv[1] if kv[1] == kv[1] else -1e300)[:8],
                loss_replay=loss_r, loss_eager=loss_e)
