"""CPU checks: the C-ABI library builds, loads and exports every symbol include/umr_hip.h declares; the
product package never touches the oracle; the product fails loudly without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "umr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(umr_[a-z0-9_]+)\s*\(", hdr)))


def test_library_builds_and_exports_every_declared_symbol():
    from umr_amd import build, _lib
    path = build.build(verbose=False)
    h = ctypes.CDLL(path)
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(h, s), "missing export %s" % s
    # and the ctypes prototype table covers the same set
    assert set(_lib.SIGNATURES) | {"umr_version"} == set(syms)
    assert _lib.version().startswith("umr_hip")


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "umr_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle|liboracle|softras_ref|/root/reference", txt, flags=re.M):
                    bad.append(fn)
    assert not bad, bad


def test_no_cpu_fallback():
    from umr_amd.smr import SoftRenderer
    from helpers import scene
    verts, faces, cams, _ = scene(1, 1, seed=0)
    with pytest.raises(RuntimeError):
        SoftRenderer(32)(verts, faces, cams)      # CPU tensors: refused, never silently computed on the host


def test_size_queries_need_no_gpu():
    from umr_amd import _lib
    L = _lib.lib()
    assert L.umr_raster_workspace_bytes(2, 1280) >= 2 * 1280 * (16 + 160)
    assert L.umr_raster_workspace_bytes(0, 5) == 0
    assert L.umr_project_workspace_bytes(2, 642) == 2 * 642 * 12
