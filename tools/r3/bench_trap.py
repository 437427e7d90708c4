"""GPU box: bench.py on the tripwire build of the library (tools/build_variants.py trap="-DUMR_TRAP=1"): every kernel of
libumr_hip.so reports the first non-finite value it reads or writes (site ids: umr_amd/csrc/umr_common.h) without adding a
launch or a synchronisation.  After bench's own output, one line on stderr names the earliest report of the process.
CAPTURE=1 additionally keeps device copies of the render-and-compare inputs of the first 45 steps (the first measurement;
5 small copies per step) and, if a site fired, writes the last step with finite inputs and its predecessor to
gpurun_out/nan/repro_inputs_<pid>.pt for an offline replay of the failing step."""
import ctypes, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from umr_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", "trap", "libumr_hip.so")
ring = []
if os.environ.get("CAPTURE") == "1":
    import umr_amd.train_step as TS
    orig = TS.RenderCompareS1.forward
    def fwd(self, outputs, batch):
        if len(ring) < 45:
            ring.append({k: outputs[k].detach().clone() for k in ("pred_vs", "cam", "tex_flow", "delta_v")} |
                        {"gan_angles": batch["gan_angles"].detach().clone()})
        return orig(self, outputs, batch)
    TS.RenderCompareS1.forward = fwd
try:
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    h = _lib.lib()
    h.umr_debug_trap.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]
    h.umr_debug_trap.restype = ctypes.c_int
    when = ctypes.c_ulonglong(0)
    site = h.umr_debug_trap(0, ctypes.byref(when))
    sys.stderr.write("bench_trap: earliest non-finite report: site %d (0 = none) at device clock %d\n" % (site, when.value))
    if site and ring:
        finite = [all(bool(torch.isfinite(v).all()) for v in r.values()) for r in ring]
        last_ok = max([i for i, f in enumerate(finite) if f], default=None)
        sys.stderr.write("bench_trap: captured %d steps, inputs finite up to step %s\n" % (len(ring), last_ok))
        if last_ok is not None:
            keep = ring[max(0, last_ok - 1):last_ok + 1]
            os.makedirs(os.path.join(ROOT, "gpurun_out", "nan"), exist_ok=True)
            torch.save({"step": last_ok, "site": site, "inputs": [{k: v.cpu() for k, v in r.items()} for r in keep]},
                       os.path.join(ROOT, "gpurun_out", "nan", "repro_inputs_%d.pt" % os.getpid()))
