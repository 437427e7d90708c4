"""Register / spill / LDS figures of the raster kernels from the compiler's own metadata (no GPU): compiles raster.hip to gfx950
assembly with the product flags (+ extra -D flags given on the command line) and prints one line per kernel whose demangled name
contains the filter.
usage: regs.py [filter] [-DFLAG ...]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umr_amd import build as B  # noqa: E402


def kernel_table(extra=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "raster.s")
        flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")]
        subprocess.check_call([B.HIPCC] + flags + ['-DUMR_SRC_HASH="regs"', "-S", "--cuda-device-only", "-o", out] + list(extra) +
                              [os.path.join(B.CSRC, "raster.hip")], stderr=subprocess.DEVNULL)
        s = open(out).read()
    rows = []
    for m in re.finditer(r'- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)', s, re.S):
        blk = m.group(0)
        name = re.search(r'\.name:\s+(\S+)', blk).group(1)
        g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = dn.replace("(anonymous namespace)::", "").replace("(RasterArgs)", "").replace("void ", "")
        rows.append((dn, dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"),
                              scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"))))
    return rows, s


if __name__ == "__main__":
    filt = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else ""
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    for dn, r in kernel_table(extra)[0]:
        if filt in dn:
            print("%-62s %s" % (dn[:62], " ".join("%s=%d" % kv for kv in r.items())))
