"""Build libumr_hip.so (hipcc, gfx950 only) in-tree: umr_amd/lib/libumr_hip.so.

`python -m umr_amd.build` or umr_amd.build.build().  The .so is git-ignored but travels with the
gpurun snapshot, so the GPU box uses the prebuilt library (hipcc is present there too).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libumr_hip.so")
SOURCES = ["raster.hip", "geometry.hip", "losses.hip", "perceptual.hip", "edt.hip", "atlas.hip", "regs.hip", "eval.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
         "-munsafe-fp-atomics",   # native global_atomic_add_f32 instead of CAS loops
         "-ffp-contract=off",     # branch-deciding expressions round like the reference's float code
         "-fno-slp-vectorize",    # no COMPILER-made v_pk_{mul,add,fma}_f32 pairs: on VGPR operands a packed op issues in 4.3-4.5
                                  # cycles against 2 x 2.4 for the two scalar ops it replaces, and the pairing costs extra moves
                                  # (-2..11 %).  The hand-written packed ops of raster_core.h are the other case: their operand
                                  # is an SGPR PAIR of the face record, where the two scalar ops would cost 2 x 4.2
                                  # (profiles/r04_valu_ubench2.log)
         "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical"]


def _dep_files():
    return sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] +
                  [os.path.normpath(os.path.join(HERE, "..", "include", "umr_hip.h"))])


def source_hash():
    """sha256 over the kernel sources + the C-ABI header (file names and contents).  The build embeds it in the library
    (umr_build_id()); tests compare the two, so a stale .so is detected wherever the tests run -- file times do not
    survive the copy to the GPU box, contents do."""
    import hashlib
    h = hashlib.sha256()
    for f in _dep_files():
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:32]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [
        os.path.join(HERE, "..", "include", "umr_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, extra_flags=()):
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + ['-DUMR_SRC_HASH="%s"' % source_hash()] + list(extra_flags) + \
        [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print("[umr_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
