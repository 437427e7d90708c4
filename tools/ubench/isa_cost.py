#!/usr/bin/env python3
"""Estimate VALU issue cycles of a span of gfx950 ISA text with the per-class costs measured by valu_ubench.hip.

usage: isa_cost.py file.s first_line last_line      (1-based, inclusive; e.g. the body of a loop found with grep -n LBB)
Classes (cycles per wave64 instruction per SIMD, >= 2 resident waves): transcendental 8.3; half rate 4.2 (compares, selects,
min/max/med3, shifts, conversions, bit-field, 24-bit mad, DPP, lane reads/writes); full-rate opcode with an SGPR / literal
source 4.3; full-rate opcode on VGPRs and inline constants only 2.4; packed fp32 5.0.
"""
import re, sys, collections

TRANS = ("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")
HALF = ("v_cmp", "v_cndmask", "v_max", "v_min", "v_med3", "v_lshl", "v_lshr", "v_ashr", "v_cvt", "v_bfe", "v_bfi", "v_mad_u32",
        "v_mad_i32", "v_mul_u32", "v_mul_lo", "v_mul_hi", "v_readlane", "v_writelane", "v_readfirstlane", "v_mbcnt", "v_perm",
        "v_alignbit", "v_rndne", "v_floor", "v_ceil", "v_trunc", "v_fract", "v_ldexp", "v_frexp", "v_div", "v_lshlrev",
        "v_lshrrev", "v_ashrrev", "v_add3", "v_lshl_add", "v_add_lshl", "v_and_or", "v_or3", "v_xad", "v_sad", "v_mad_u64", "v_cmpx")
COST = {"trans": 8.3, "half": 4.2, "full_s": 4.3, "full_v": 2.4, "pk": 5.0}


def classify(line):
    t = line.split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith("."):
        return None
    op = t.split()[0]
    if op.startswith("s_"):
        return "salu" if not op.startswith(("s_waitcnt", "s_nop", "s_load", "s_buffer_load", "s_barrier", "s_cbranch", "s_branch")) else \
            ("smem" if "load" in op else "sctl")
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if not op.startswith("v_"):
        return "other"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith(TRANS):
        return "trans"
    if "dpp" in t or "row_" in t or "quad_perm" in t:
        return "half"
    if op.startswith(HALF):
        return "half"
    ops = t[len(op):]
    srcs = ops.split(",")[1:]
    def is_s(x):
        x = x.strip().strip("|-").replace("abs(", "").replace(")", "")
        if re.match(r"^(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|m0|ttmp)", x):
            return True
        if re.match(r"^0x[0-9a-f]+$", x):                   # literal (inline ints -16..64 print as decimals)
            return True
        return False
    return "full_s" if any(is_s(x) for x in srcs) else "full_v"


def main():
    fn, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    c = collections.Counter()
    ops = collections.Counter()
    for i, line in enumerate(open(fn), 1):
        if a <= i <= b:
            k = classify(line)
            if k:
                c[k] += 1
                if k in ("half", "full_s"):
                    ops[(k, line.split()[0])] += 1
    cyc = sum(COST[k] * n for k, n in c.items() if k in COST)
    print(dict(c))
    print("VALU instr %d, est. issue cycles %.0f" % (sum(n for k, n in c.items() if k in COST), cyc))
    if "-v" in sys.argv:
        for (k, o), n in sorted(ops.items(), key=lambda x: -x[1])[:25]:
            print("   %-7s %-24s %d" % (k, o, n))


if __name__ == "__main__":
    main()
