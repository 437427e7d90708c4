#!/usr/bin/env python3
"""Build compile-time variants of libumr_hip.so HERE (hipcc cross-compiles gfx950 without a GPU) so a GPU call only has to
time them: umr_amd/lib/variants/<tag>/libumr_hip.so (git-ignored, travels with the gpurun snapshot).

usage: tools/build_variants.py tag1="-DFM_SKIP_EMPTY=0" tag2="-DBWD_WPE=8 -DFOO=1" ...
Timing side: UMR_LIB_VARIANT=<tag> python tools/sweep_fm.py <tag>   (tools/sweep_fm.py points _lib.LIB_PATH at the variant)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from umr_amd import build as B  # noqa: E402


def one(spec):
    tag, flags = spec.split("=", 1)
    out = os.path.join(B.LIBDIR, "variants", tag)
    os.makedirs(out, exist_ok=True)
    cmd = [B.HIPCC] + B.FLAGS + ['-DUMR_SRC_HASH="variant-%s"' % tag] + flags.split() + \
        [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", os.path.join(out, "libumr_hip.so")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return tag, r.returncode, r.stderr[-2000:]


if __name__ == "__main__":
    with ThreadPoolExecutor(max_workers=8) as ex:
        for tag, rc, err in ex.map(one, sys.argv[1:]):
            print(tag, "ok" if rc == 0 else "FAILED\n" + err, flush=True)
