"""Shared test helpers: seeded synthetic scenes and tolerant comparisons."""
import numpy as np
import torch

from umr_amd.mesh import create_sphere


def scene(n_meshes, subdiv, seed, scale=(0.6, 0.9)):
    """Perturbed icosphere + random cameras.  Must stay identical to oracle/gen_golden.py:scene()."""
    g = torch.Generator().manual_seed(seed)
    v, f = create_sphere(subdiv)
    verts = torch.from_numpy(v).float()[None].repeat(n_meshes, 1, 1)
    verts = verts + 0.05 * torch.randn(verts.shape, generator=g)
    faces = torch.from_numpy(f).long()[None].repeat(n_meshes, 1, 1)
    s = scale[0] + (scale[1] - scale[0]) * torch.rand(n_meshes, 1, generator=g)
    t = -0.1 + 0.2 * torch.rand(n_meshes, 2, generator=g)
    q = torch.randn(n_meshes, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    return verts, faces, torch.cat([s, t, q], 1), g


def assert_close_frac(got, ref, atol, rtol=0.0, frac=0.999, max_outlier=None, name=""):
    """>= `frac` of the elements within atol + rtol*|ref|; the rest (isolated branch flips: texel index,
    closest-edge selection, threshold rejects flip on 1-ulp differences) bounded by `max_outlier`."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert np.isfinite(got).all(), name + ": non-finite values"
    err = np.abs(got - ref)
    ok = err <= atol + rtol * np.abs(ref)
    f = ok.mean() if ok.size else 1.0
    _record(name, got.size, f, float(err.max()) if err.size else 0.0, atol, rtol, frac, max_outlier)
    assert f >= frac, "%s: only %.5f within tol (max err %.3e, atol %.1e rtol %.1e)" % (name, f, err.max(), atol, rtol)
    if max_outlier is not None and err.size:
        assert err.max() <= max_outlier, "%s: outlier %.3e > %.3e" % (name, err.max(), max_outlier)


def _record(name, n, frac_ok, max_err, atol, rtol, need, max_outlier):
    """Measured parity figures, one JSON line per check: printed (pytest -s / failure reports) and appended to
    gpurun_out/parity_measured.jsonl so the numbers behind every tolerance are on file (profiles/archive_r01_r03/r02_parity_measured.jsonl)."""
    import json
    import os
    rec = dict(check=name, test=os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], elements=int(n),
               frac_within=float(frac_ok), max_abs_err=float(max_err), atol=float(np.max(atol)), rtol=float(rtol),
               frac_required=float(need), max_outlier_allowed=None if max_outlier is None else float(max_outlier))
    print("[parity] " + json.dumps(rec))
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        # runs of the same checks on the CPU emulation of the library (tests/test_gpu_suite_on_emulator.py,
        # tests/test_raster_library_on_host.py) are kept apart from the MI355X measurements
        emu = any(k in rec["test"] for k in ("on_emulator", "on_host"))
        with open(os.path.join(d, "parity_measured_emulator.jsonl" if emu else "parity_measured.jsonl"), "a") as fh:
            fh.write(json.dumps(rec) + "\n")
    except OSError:
        pass


def t2n(x):
    return x.detach().cpu().numpy()
