"""Build an experimental copy of the library with extra compiler flags: umr_amd/lib/exp/libumr_hip_<tag>.so
usage: build_variant.py <tag> [flags...]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umr_amd import build as B
tag, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(B.LIBDIR, "exp", "libumr_hip_%s.so" % tag)
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call([B.HIPCC] + B.FLAGS + ['-DUMR_SRC_HASH="%s"' % tag] + flags + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", out])
print(out)
