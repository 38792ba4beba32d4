#!/usr/bin/env python3
"""Loops of one kernel in a gfx950 .s file: for every backward branch, the instruction mix of the block range it closes.

    python tools/isa_loops.py file.s KERNEL_SUBSTRING [--min N]

Counts per loop: vector ALU instructions (v_*, packed ones and DPP moves listed separately), plain v_mov, scalar ALU, s_nop
(with their wait states), s_waitcnt, vector memory, LDS.  Used by tests/test_build_hazards.py to pin the marching loop's size."""
import re
import sys
from collections import Counter

LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BR = re.compile(r"^\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)|^\s*s_branch\s+(\.LBB\d+_\d+)")


def kernel_lines(path, sub):
    lines = open(path).read().split("\n")
    out, name = [], None
    for i, l in enumerate(lines):
        if name is None:
            if l.endswith(":") and sub in l and not l.startswith((".", " ", "\t")) and "@" not in l.split(":")[0]:
                name = l.split(":")[0]
            elif sub in l and l.rstrip().endswith("; @" + l.split(":")[0]) :
                name = l.split(":")[0]
            if name is None:
                continue
            continue
        if l.startswith(".Lfunc_end") or l.strip().startswith("s_endpgm") and False:
            break
        out.append(l)
    return name, out


def classify(t):
    op = t.split()[0]
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith("v_mov") and "dpp" not in t and "row_" not in t and "wave_" not in t:
        return "v_mov"
    if "row_sh" in t or "wave_sh" in t or "dpp" in op or "row_bcast" in t or "quad_perm" in t:
        return "v_dpp"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith(("v_div_", "v_rcp", "v_rsq", "v_sqrt")):
        return "v_div/trans"
    if op.startswith("v_"):
        return "v_other"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_"):
        return "salu"
    return "other"


def loops(lines):
    pos = {}
    for i, l in enumerate(lines):
        m = LABEL.match(l)
        if m:
            pos[m.group(1)] = i
    res = []
    for i, l in enumerate(lines):
        m = BR.match(l)
        if not m:
            continue
        tgt = m.group(1) or m.group(2)
        if tgt in pos and pos[tgt] < i:
            c = Counter()
            nops = 0
            for t in lines[pos[tgt]:i + 1]:
                s = t.strip()
                if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
                    continue
                k = classify(s)
                c[k] += 1
                if k == "s_nop":
                    nops += int(s.split()[1]) + 1
            res.append((tgt, pos[tgt], i, c, nops))
    return res


def main():
    path, sub = sys.argv[1], sys.argv[2]
    mn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 100
    name, lines = kernel_lines(path, sub)
    print("kernel:", name, "lines", len(lines))
    for tgt, a, b, c, nops in loops(lines):
        tot = sum(c.values())
        if tot < mn:
            continue
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print(f"{tgt} [{a}:{b}] instr {tot}  VALU {valu}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + f"  nop_states={nops}")


if __name__ == "__main__":
    main()


def follow(lines, header, policy=None, limit=20000):
    """Executed instruction mix of ONE trip round the loop at label `header` on the path a plain-fluid, interior-window, no-clamp column takes:
    s_cbranch_execz taken (no lane needs the velocity clamp), s_cbranch_scc0 taken (no lane unsafe -> fast division), s_cbranch_vccnz
    taken (not a far-field window), s_cbranch_scc1 / execnz / vccz not taken; s_branch followed.  Stops when the header is reached again."""
    policy = policy or {"s_cbranch_execz": True, "s_cbranch_scc0": True, "s_cbranch_vccnz": True, "s_cbranch_scc1": False,
                        "s_cbranch_execnz": False, "s_cbranch_vccz": False}
    pos = {}
    for i, l in enumerate(lines):
        m = LABEL.match(l)
        if m:
            pos[m.group(1)] = i
    c, nops, i, n = Counter(), 0, pos[header] + 1, 0
    while n < limit:
        s = lines[i].strip()
        i += 1
        if LABEL.match(lines[i - 1]) and lines[i - 1].startswith(header + ":"):
            break
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        n += 1
        op = s.split()[0]
        k = classify(s)
        c[k] += 1
        if k == "s_nop":
            nops += int(s.split()[1]) + 1
        if op == "s_branch":
            i = pos[s.split()[1]]
        elif op.startswith("s_cbranch") and policy.get(op, False):
            i = pos[s.split()[1]]
    return c, nops


if __name__ == "__main__" and "--follow" in sys.argv:
    hdr = sys.argv[sys.argv.index("--follow") + 1]
    name, lines = kernel_lines(sys.argv[1], sys.argv[2])
    c, nops = follow(lines, hdr)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(f"follow {hdr}: instr {sum(c.values())} VALU {valu} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + f" nop_states={nops}")
