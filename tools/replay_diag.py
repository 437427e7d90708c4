"""GPU box: which gradient tensors of a REPLAYED training step differ from the eager step's?  Captures bench.py's train_s1 / train_s2
step at bench size into one HIP graph, then runs eager, replay x3, eager from ONE saved state (parameters, buffers, optimizer state,
device generator) and lists every tensor whose replays are off by more than 20x what the two eager steps differ by.  This is the
diagnosis that found round 6's stale conv-bias gradients (HISTORY 13; umr_amd/graph_check.py is the guard bench.py runs).
usage: replay_diag.py s1|s2 <bn_eval 0|1> <batch>"""
import argparse, sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umr_amd import model as M
from umr_amd.synthetic import make_s1_inputs
dev = torch.device("cuda:0")
wl, bn_eval, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(77)
args = argparse.Namespace(graph=1, batch=B, image_size=256, subdivide=3, epoch=0, share_mask_render=1, data_seed=100)
if wl == "s2":
    step = M.build_training_step_s2(args, dev, 1)
else:
    tv, faces, _, _ = make_s1_inputs(B, 256, 3, seed=100, device=dev)
    step = M.build_training_step(tv, faces, args, dev, 1)
model, opt = step.model, step.opt
if bn_eval: model.eval()
names = {id(p): n for n, p in model.named_parameters()}
def state():
    out = list(model.parameters()) + list(model.buffers()) + [step.it_dev]
    for st in opt.state.values():
        out += [v for _, v in sorted(st.items()) if torch.is_tensor(v)]
    return out + [g["lr"] for g in opt.param_groups if torch.is_tensor(g["lr"])]
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(4): step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.autograd.set_multithreading_enabled(False), torch.cuda.graph(g, stream=side):
        static_loss = step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
graph_grads = {names[id(p)]: p.grad for p in model.parameters() if p.grad is not None}
saved = [t.detach().clone() for t in state()]
rng = torch.cuda.get_rng_state(dev)
def restore():
    with torch.no_grad():
        for t, s in zip(state(), saved): t.copy_(s)
    torch.cuda.set_rng_state(rng, dev)
res = {}
for tag in ("eager1", "replay1", "replay2", "replay3", "eager2"):
    restore()
    if tag.startswith("replay"):
        g.replay(); torch.cuda.synchronize()
        res[tag] = ({n: t.detach().clone() for n, t in graph_grads.items()}, float(static_loss))
    else:
        l = float(step()); torch.cuda.synchronize()
        res[tag] = ({names[id(p)]: p.grad.detach().clone() for p in model.parameters() if p.grad is not None}, l)
print(wl, "bn_eval", bn_eval, "B", B, "losses", {t: round(v[1], 6) for t, v in res.items()})
gmax = max(float(a.abs().max()) for a in res["eager1"][0].values())
bad = 0
for n, a in res["eager1"][0].items():
    sc = float(a.abs().max())
    if sc < 1e-4 * gmax: continue
    e = {t: float((res[t][0][n] - a).abs().max()) / sc for t in ("eager2", "replay1", "replay2", "replay3")}
    if max(e["replay1"], e["replay2"], e["replay3"]) > 20 * max(e["eager2"], 1e-3):
        bad += 1
        print("  MISMATCH %-55s shape %-18s |g| %.2e  " % (n, tuple(a.shape), sc) + " ".join("%s %.1e" % kv for kv in e.items()))
print("tensors off in a replay:", bad, "of", len(res["eager1"][0]))
