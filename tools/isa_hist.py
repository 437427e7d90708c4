"""ISA histogram of the raster kernels' visit loops by ISSUE CLASS (no GPU): compiles raster.hip to gfx950 assembly with the product
flags, finds in each shipped raster kernel the deepest loop that holds the sigmoid's v_exp_f32 (the per-(pixel block, face) visit),
and counts its instructions by the classes tools/ubench/valu_ubench2.hip measured on MI355X (profiles/r04_valu_ubench2.log; cycles
per wave64 instruction per SIMD at 8 waves): full-rate VALU on VGPRs 2.4, full-rate with an SGPR / literal source 4.3, half-rate
class (compare, select, min / max / med3, shift, cvt, bit-field, 24-bit mad, DPP, lane read / write) 4.2, packed fp32 4.4,
transcendental 8.3.  The loop body is the STATIC text: blocks of the rare paths (IEEE division of faces with degenerate depth, the
all-three-edges path of flagged faces) are listed separately where they can be told apart by their v_div_scale / trip marks.
usage: isa_hist.py [out.json] [-DFLAG ...]"""
import collections
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench"))
from tools.regs import kernel_table  # noqa: E402
import isa_cost  # noqa: E402

COST = {"trans": 8.3, "half": 4.2, "full_s": 4.3, "full_v": 2.4, "pk": 4.4}
SHIPPED = ["k_raster_forward<1, true, true, true>", "k_raster_forward<1, false, true, false>", "k_raster_forward<2, false, true, false>",
           "k_raster_backward_fm_agp<1, true, true, true>", "k_raster_backward_fm_quads<2, true, false, true>",
           "k_raster_backward_fm<1, false, true, true>", "k_raster_backward_fm<1, true, false, true>", "k_raster_backward_fm_w6<1, true, true, true>"]


def functions(asm):
    lines = asm.split("\n")
    out, i = {}, 0
    while i < len(lines):
        m = re.match(r"^(_ZN\S+):\s*; @", lines[i])
        if m:
            j = i
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            out[m.group(1)] = lines[i:j]
            i = j
        i += 1
    return out


def visit_loop(body):
    """Lines of the deepest loop that holds v_exp_f32: LLVM prints, on every basic block of a loop, `in Loop: Header=BBx_y Depth=d`
    (and `Loop Header: Depth=d` on the header itself), whatever the block layout (rotated loops have their latch in front)."""
    blocks, cur = [], None                       # (label, first line, last line, header it belongs to, depth)
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            if cur:
                cur[2] = i - 1
            cur = [m.group(1), i, len(body) - 1, None, 0]
            blocks.append(cur)
            text = " ".join(body[i:i + 4])
            h = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
            if h:
                cur[3], cur[4] = ".L" + h.group(1), int(h.group(2))
            else:
                for k in range(i, min(i + 4, len(body))):
                    if k > i and not body[k].strip().startswith(";"):
                        break
                    h2 = re.search(r"Loop Header: Depth=(\d+)", body[k])
                    if h2:
                        cur[3], cur[4] = cur[0], int(h2.group(1))
                        break
    loops = collections.defaultdict(list)
    for lab, a, b, head, depth in blocks:
        if head:
            loops[(head, depth)].append((a, b))
    best = None
    for (head, depth), segs in loops.items():
        lines = [l for a, b in segs for l in body[a:b + 1]]
        if any("v_exp_f32" in l for l in lines) and (best is None or depth > best[1]):
            best = (lines, depth)
    return best


def histogram(lines):
    c, ops = collections.Counter(), collections.Counter()
    for l in lines:
        k = isa_cost.classify(l)
        if k:
            c[k] += 1
            if k in COST:
                ops[(k, l.split()[0])] += 1
    return c, ops


if __name__ == "__main__":
    out_path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else None
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    rows, asm = kernel_table(extra)
    regs = dict(rows)
    import subprocess
    rep = {}
    for name, body in functions(asm).items():
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = dn.replace("(anonymous namespace)::", "").replace("(RasterArgs)", "").replace("void ", "")
        if dn not in SHIPPED:
            continue
        loop = visit_loop(body)
        if not loop:
            continue
        lines, depth = loop
        c, ops = histogram(lines)
        valu = sum(n for k, n in c.items() if k in COST)
        cyc = sum(COST[k] * n for k, n in c.items() if k in COST)
        slow = sum(1 for l in lines if "v_div_scale" in l)
        rep[dn] = {"registers": regs.get(dn), "visit_loop_lines": len(lines), "static_VALU": valu, "by_class": {k: c[k] for k in COST},
                   "other": {k: n for k, n in c.items() if k not in COST},
                   "weighted_cycles": round(cyc), "cycles_per_VALU": round(cyc / max(valu, 1), 2),
                   "share_of_cycles": {k: round(COST[k] * c[k] / cyc, 3) for k in COST},
                   "ieee_division_path_instr (rare: faces with degenerate depth)": slow * 79 // 14 if slow else 0,
                   "top_half_rate": [[o, n] for (k, o), n in sorted(ops.items(), key=lambda x: -x[1]) if k == "half"][:8],
                   "top_sgpr_sourced": [[o, n] for (k, o), n in sorted(ops.items(), key=lambda x: -x[1]) if k == "full_s"][:6]}
    if out_path:
        json.dump(rep, open(out_path, "w"), indent=1)
    print("%-52s %5s %5s %5s %5s %5s %5s | %6s %5s" % ("kernel (visit loop, static)", "VALU", "fullV", "fullS", "half", "pk", "trans", "cycles", "c/VALU"))
    for dn, r in rep.items():
        b = r["by_class"]
        print("%-52s %5d %5d %5d %5d %5d %5d | %6d %5.2f" % (dn[:52], r["static_VALU"], b["full_v"], b["full_s"], b["half"], b["pk"], b["trans"],
                                                              r["weighted_cycles"], r["cycles_per_VALU"]))
