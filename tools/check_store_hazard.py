#!/usr/bin/env python3
"""Check generated gfx950 ISA for the buffer-store data hazard (see store_data_fence() in csrc/step_march.hpp).

A `buffer_store_dwordx3/x4 v[a:b], ..., sN offen` (register soffset) must not be followed, within two wait states, by
an instruction that writes one of v[a:b]: hipcc pads that hazard only for stores without a register soffset, and on
MI355X the unpadded form was seen to store the overwritten value.

    python tools/check_store_hazard.py file.s [file2.s ...]     # exit 1 and a listing if a violation is found
    python tools/check_store_hazard.py --build                  # compile csrc/windtunnel.hip with --save-temps and check it
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WAIT_STATES = 2
STORE = re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")
# instructions whose first vector operand is a SOURCE, not a destination
READS_FIRST = re.compile(r"^\s*(buffer_store|global_store|flat_store|scratch_store|ds_write|ds_store|v_cmp|v_cmpx|v_readlane|v_readfirstlane|s_|v_nop|exp)")
REG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")


def dest_regs(line):
    """VGPRs written by an instruction line (first operand; VOP3 with an SGPR-pair carry-out: still the first)."""
    if READS_FIRST.match(line):
        return set()
    parts = line.strip().split(None, 1)
    if len(parts) < 2:
        return set()
    first = parts[1].split(",")[0].strip()
    m = REG.match(first)
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def is_instr(line):
    t = line.strip()
    return bool(t) and not t.startswith((";", ".", "//")) and not t.endswith(":")


def check(path):
    lines = open(path).read().split("\n")
    bad = []
    func = "?"
    nstores = 0
    for i, l in enumerate(lines):
        if l.endswith(":") and l[:1].isalpha() or l.startswith("_") and l.endswith(":"):
            func = l[:-1]
        m = STORE.match(l)
        if not m or not m.group(4).startswith("s"):
            continue                                            # immediate / no soffset: hipcc pads it
        nstores += 1
        data = set(range(int(m.group(1)), int(m.group(2)) + 1))
        states, j = 0, i + 1
        while states < WAIT_STATES and j < len(lines):
            n = lines[j]
            j += 1
            if not is_instr(n):
                if n.strip().endswith(":") or n.strip().startswith(".LBB"):
                    continue                                    # a label: fall through into the next block (conservative)
                continue
            t = n.strip()
            if t.startswith("s_nop"):
                states += int(t.split()[1]) + 1
                continue
            if t.startswith(("s_endpgm", "s_branch", "s_cbranch", "s_setpc")):
                break                                           # control flow: a taken branch costs more than two states
            hit = dest_regs(n) & data
            if hit:
                bad.append((path, func, i + 1, l.strip(), j, t, sorted(hit)))
            states += 1
    return nstores, bad


RES = re.compile(r"^\s*\.set\s+(\S+)\.(num_vgpr|private_seg_size),\s*(\d+)")


def resources(path):
    """{kernel symbol: {"num_vgpr": n, "private_seg_size": bytes of scratch per lane}} from the .set lines of a gfx950 .s file."""
    out = {}
    for l in open(path):
        m = RES.match(l)
        if m:
            out.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
    return out


def build():
    tmp = tempfile.mkdtemp(prefix="wt_isa_")
    src = os.path.join(ROOT, "airfoil-cfd-tool_amd", "csrc", "windtunnel.hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--save-temps",
           "-c", src, "-o", os.path.join(tmp, "wt.o")]
    subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return [os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith("gfx950.s")]


def main():
    files = build() if sys.argv[1:] == ["--build"] else sys.argv[1:]
    total, bad = 0, []
    for f in files:
        n, b = check(f)
        total += n
        bad += b
    for path, func, ln, store, ln2, instr, regs in bad:
        print(f"{os.path.basename(path)}:{ln} [{func}] {store}\n    overwritten at line {ln2}: {instr}  (v{regs})")
    print(f"{total} buffer_store_dwordx3/x4 with register soffset checked, {len(bad)} hazard(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
