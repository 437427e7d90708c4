"""Build libumr_hip.so with a text pass over raster.hip's DEVICE assembly between hipcc's compile and assemble steps
(experiments on instruction forms the compiler cannot be told to choose).  The sub-commands are hipcc's own (-###), run one
by one; only the device compile is switched from -emit-obj to -S and the result assembled after the pass.

usage: build_asm.py <out.so> <pass> [extra -D flags...]      pass: none | cnd64  (every v_cndmask_b32_e32 .., vcc -> _e64)"""
import os
import re
import shlex
import subprocess
import sys
import tempfile

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from umr_amd import build as B  # noqa: E402


def passes(name, text):
    if name == "none":
        return text, 0
    if name == "cnd64":
        # VOP2 form reads VCC implicitly; the VOP3 form names it.  Same operands otherwise (src0 may be a constant / SGPR).
        pat = re.compile(r"^(\s*)v_cndmask_b32_e32 (v\d+), (.+), (v\d+), vcc\s*$", re.M)
        return pat.subn(r"\1v_cndmask_b32_e64 \2, \3, \4, vcc", text)
    raise SystemExit("unknown pass " + name)


def main():
    out, pname, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    tmp = tempfile.mkdtemp(prefix="umr_asm_")
    objs = []
    common = [B.HIPCC] + [f for f in B.FLAGS if f != "-shared"] + ['-DUMR_SRC_HASH="%s"' % B.source_hash()] + extra
    for s in B.SOURCES:
        src = os.path.join(B.CSRC, s)
        obj = os.path.join(tmp, s + ".o")
        if s != "raster.hip":
            subprocess.check_call(common + ["-c", src, "-o", obj])
            objs.append(obj)
            continue
        r = subprocess.run(common + ["-###", "-c", src, "-o", obj], capture_output=True, text=True)
        cmds = [shlex.split(l.strip()) for l in r.stderr.splitlines() if l.strip().startswith('"')]
        dev = next(c for c in cmds if "-fcuda-is-device" in c)
        dev_o = dev[dev.index("-o") + 1]
        asm = os.path.join(tmp, "raster.s")
        dev_s = [("-S" if a == "-emit-obj" else a) for a in dev]
        dev_s[dev_s.index("-o") + 1] = asm
        subprocess.check_call(dev_s)
        text, n = passes(pname, open(asm).read())
        open(asm, "w").write(text)
        print("[build_asm] pass %s: %d rewrites" % (pname, n), flush=True)
        subprocess.check_call([os.path.join(os.path.dirname(dev[0]), "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa",
                               "-mcpu=gfx950", "-c", asm, "-o", dev_o])
        for c in cmds:
            if c is dev:
                continue
            subprocess.check_call(c)
        objs.append(obj)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    subprocess.check_call([B.HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950"] + objs + ["-o", out])
    print(out)


if __name__ == "__main__":
    main()
