"""GPU box: bench.py on the tripwire build of the library (-DUMR_TRAP=1, built by tools/nan/nan_hunt.sh into umr_amd/lib/exp/):
every kernel of libumr_hip.so reports the first non-finite value it reads or writes (site ids: umr_amd/csrc/umr_common.h) without
adding a launch or a synchronisation.  After bench's own output one line on stderr names the earliest report of the process.
The geometry inputs of EVERY step (vertices, cameras: ~130 KB per step) and its loss terms are kept on the device; if a site fired or
a term went non-finite, the first step with a non-finite term and its predecessor go to gpurun_out/nan/repro_<pid>.pt for an offline
replay of the failing render on the CPU emulation (tools/nan/replay_nan.py)."""
import ctypes, math, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from umr_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "exp", "libumr_hip_%s.so" % os.environ.get("TRAP_LIB", "trap"))
for kv in os.environ.get("UMR_DEBUG_SET", "").split(","):      # e.g. UMR_DEBUG_SET=exact_edges=0
    if "=" in kv:
        _lib.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
import umr_amd.train_step as TS
ring, seen, extra = [], [0], {}
KEYS = ("pred_vs", "delta_v", "cam", "cam_hypotheses", "cam_probs")      # small tensors only: the ring has to fit gpurun_out (64 MiB)


def hook(cls):
    orig = cls.forward

    def fwd(self, outputs, batch):
        ring.append((seen[0], {k: outputs[k].detach().clone() for k in KEYS if k in outputs}))
        ring[-1][1]["gan_angles"] = batch["gan_angles"].detach().clone()
        if seen[0] == 0:
            ring[-1][1]["faces"] = self.faces.detach().clone()
            extra["faces"] = self.faces.detach().cpu()
        seen[0] += 1
        total, terms = orig(self, outputs, batch)
        ring[-1][1]["terms"] = {k: v.detach().clone() for k, v in terms.items()}
        return total, terms
    cls.forward = fwd


hook(TS.RenderCompareS1); hook(TS.RenderCompareS2)
try:
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    h = _lib.lib()
    h.umr_debug_trap.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]
    h.umr_debug_trap.restype = ctypes.c_int
    when = ctypes.c_ulonglong(0)
    site = h.umr_debug_trap(0, ctypes.byref(when))
    h.umr_debug_trap_where.restype = ctypes.c_ulonglong
    where = h.umr_debug_trap_where()
    info = int(where & 0xffffff) if where != 2 ** 64 - 1 else None
    if info is not None:
        sys.stderr.write("bench_trap: first raster-backward offender: launch with N %s 32, mesh-of-launch %d, face %d\n"
                         % (">" if info >> 23 else "<=", (info >> 16) & 127, info & 0xffff))
    bad_terms = [(s, k) for s, r in ring for k, v in r.get("terms", {}).items() if not math.isfinite(float(v))]
    first_bad = min([s for s, _ in bad_terms], default=None)
    sys.stderr.write("bench_trap: earliest non-finite report: site %d (0 = none) at device clock %d; steps seen %d; non-finite terms in the last steps: %s\n"
                     % (site, when.value, seen[0], bad_terms[:6]))
    if site or bad_terms:
        os.makedirs(os.path.join(ROOT, "gpurun_out", "nan"), exist_ok=True)
        keep = [(s, r) for s, r in ring if first_bad is None or first_bad - 1 <= s <= first_bad] if first_bad is not None else ring[-2:]
        torch.save({"site": site, "where": info, "first_bad_step": first_bad, "faces": extra.get("faces"),
                    "ring": [(s, {k: (v.cpu() if torch.is_tensor(v) else {a: b.cpu() for a, b in v.items()}) for k, v in r.items()}) for s, r in keep]},
                   os.path.join(ROOT, "gpurun_out", "nan", "repro_%d.pt" % os.getpid()))
