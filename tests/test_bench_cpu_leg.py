"""CPU: bench.py's `cpu_baseline` leg (the oracle timed on host cores) runs and reports the contract fields;
the JSON-line schema helpers do not need a GPU."""
import argparse
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_cpu_baseline_leg_small(oracle_built):
    b = _load_bench()
    args = argparse.Namespace(image_size=32, subdivide=1, batch=2)
    r = b.cpu_baseline(args, 1)
    assert r["unit"] == "images/s" and r["kind"] == "port" and r["cores"] >= 1 and r["value"] > 0
    assert "sample" in r and "train_s1" in r["sample"]
    assert b.METRIC.startswith("train images/sec") and b.HBM_PEAK_GBS == 8000.0


def test_self_launch_builds_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` outside torchrun re-launches itself under torch.distributed.run: N ranks on this node,
    rendezvous on 127.0.0.1, the original arguments passed through, dmabuf IPC mode set for RCCL."""
    import subprocess
    import sys
    import pytest
    b = _load_bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    args = argparse.Namespace(gpus=4, master_port=29511)
    with pytest.raises(SystemExit) as e:
        b.self_launch(args)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
