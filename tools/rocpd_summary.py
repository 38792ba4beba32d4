#!/usr/bin/env python3
"""Condense rocprofv3 (rocpd sqlite) outputs: per-kernel average duration and per-kernel average counter values.

    python tools/rocpd_summary.py gpurun_out/prof_x/*.db [--csv out.csv]
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("wt::", "").replace(", ", ",")


def main():
    args = sys.argv[1:]
    csv = None
    if "--csv" in args:
        i = args.index("--csv")
        csv = args[i + 1]
        del args[i:i + 2]
    paths = args
    rows = []
    for p in paths:
        c = sqlite3.connect(p)
        dur = defaultdict(list)
        for name, d in c.execute("select name, duration from kernels"):
            dur[short(name)].append(d)
        cnt = defaultdict(lambda: defaultdict(list))
        try:
            for kname, cname, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
                cnt[short(kname)][cname].append(val)
        except sqlite3.Error:
            pass
        for k in sorted(dur, key=lambda k: -sum(dur[k])):
            v = dur[k]
            row = {"file": p, "kernel": k, "calls": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3}
            for cname, vals in sorted(cnt.get(k, {}).items()):
                # one row per dispatch and (for per-XCD/SE counters) per instance: sum instances, average dispatches
                row[cname] = sum(vals) / len(v)
            rows.append(row)
    keys = []
    for r in rows:
        for k in r:
            if k not in keys:
                keys.append(k)
    import csv as csvmod
    out = open(csv, "w", newline="") if csv else sys.stdout
    wr = csvmod.writer(out)
    wr.writerow(keys)
    for r in rows:
        wr.writerow([f"{r.get(k, ''):.6g}" if isinstance(r.get(k), float) else str(r.get(k, "")) for k in keys])


if __name__ == "__main__":
    main()
