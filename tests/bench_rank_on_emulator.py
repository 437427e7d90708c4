"""TEST INFRASTRUCTURE (not a test module): bench.py's main() as the CPU suite starts it -- the same function, argument parser,
self-launch, rendezvous, per-rank set-up, timed loop, reductions over ranks and result line, on host tensors over gloo with the
wave64 emulation of the library underneath (tests/host_raster.py::emulated_product).  Used by
tests/test_bench_eight_ranks_on_emulator.py; bench.py re-launches `sys.argv[0]`, i.e. this file, for its ranks."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

torch.set_num_threads(1)
import host_raster as HR  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
with HR.emulated_product():
    bench.main(device="cpu", backend="gloo")
