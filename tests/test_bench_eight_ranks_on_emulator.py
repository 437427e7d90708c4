"""CPU, gloo: `bench.py --gpus 8` end to end the way the driver's scaling run starts it -- the self-launch under torch.distributed.run
with a rendezvous port that is free right now, eight ranks, per-rank MIOpen directories, bench's own training step (MeshNet, the
render-and-compare path, DDP's bucketed all-reduce, Adam; toy shapes, kernels on the wave64 emulation), barrier-bracketed timing
with the max over ranks, and ONE JSON line from rank 0.  No 8-GPU node has ever been available to the builder or the driver
(SCALE_r01..r03: skipped); this is what can be checked about that run without one."""
import json
import os
import subprocess
import sys

import pytest

import host_raster as HR

pytestmark = pytest.mark.skipif(not HR.available(), reason="clang++ of the ROCm toolchain not present")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENTRY = os.path.join(ROOT, "tests", "bench_rank_on_emulator.py")
TOY = ["--steps", "2", "--warmup", "1", "--batch", "2", "--image-size", "64", "--subdivide", "1", "--cpu-baseline", "0", "--profile-steps", "1"]


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env["TMPDIR"] = env.get("TMPDIR", "/tmp")
    return env


def test_bench_self_launch_with_eight_ranks():
    HR.lib(HR.build())                                     # the emulated library, built once before the ranks start
    r = subprocess.run([sys.executable, ENTRY, "--gpus", "8"] + TOY, capture_output=True, text=True, env=_env(), timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout, from rank 0: %r" % lines        # banners / other ranks stay off stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak" and j["value"] > 0
    c = j["config"]
    assert c["rccl_ranks"] == 8 and c["parallelism"] == "dp8" and c["global_batch"] == 16
    assert c["grad_abs_sum"] > 0 and c["grad_abs_sum_spread_over_ranks"] == 0.0          # every rank holds the all-reduced gradients
    assert c["discarded_nonfinite_runs"] == 0 and c["final_loss"] == c["final_loss"]
    assert abs(j["value"] - 16 * 2 / (j["ms_per_step"] * 2 / 1e3)) <= 1e-6 * j["value"]   # whole-job images / max-over-ranks time
    for rank in range(8):                                                                 # per-rank MIOpen database / cache directories
        for sub in ("db", "cache"):
            assert os.path.isdir(os.path.join(_env()["TMPDIR"], "umr_miopen_%s_rank%d" % (sub, rank)))


def test_two_single_rank_ddp_runs_at_once_do_not_collide():
    """--force-ddp outside a launcher used to fall back to the fixed port 29511: two such runs on one host at the same
    time must both come up."""
    HR.lib(HR.build())
    ps = [subprocess.Popen([sys.executable, ENTRY, "--gpus", "1", "--force-ddp", "1"] + TOY, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, env=_env(), cwd=ROOT) for _ in range(2)]
    outs = [p.communicate(timeout=900) for p in ps]
    for p, (o, e) in zip(ps, outs):
        assert p.returncode == 0, e[-2000:]
        j = json.loads([l for l in o.splitlines() if l.strip()][-1])
        assert j["config"]["rccl_ranks"] == 1 and j["n_gpus"] == 1
