#!/usr/bin/env python3
"""Condense the rocprofv3 runs of bench.py (rocpd sqlite output: one --kernel-trace --stats run, one --pmc FETCH_SIZE
run, one --pmc WRITE_SIZE run) into the files kept under profiles/:

    <tag>_kernel_stats.csv     per-kernel calls / total / average / min / max duration of the trace run
    <tag>_pmc_traffic.csv      per-kernel average FETCH_SIZE and WRITE_SIZE (KB, as reported) and the HBM bytes derived
    pmc_traffic.json           the entry bench.py reports as roofline.traffic (+ where and when it was measured)

    python tools/summarize_profile.py <tag> <key> <trace.db> <fetch.db> <write.db>
        key, e.g. 4096x4096_float32_march3  (bench.py: "<nx>x<ny>_<dtype>" + "_march" / "_march3" for the two- / three-steps-per-pass kernel)

HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE (KB) x 1024 x 2 on gfx950 (the counter tallies 128-B requests
at 64 B for wide coalesced reads), WRITE_SIZE (KB) x 1024; the two counters do not fit one pass, hence two runs.
One "launch" of the marching path = k_march<false, FD> (non-emitting pass) + the k_halo_from_seams launch before it
(the first pass after an init / a single step builds its halo table with k_halo_rows instead: once per run, not counted).
"""
import csv
import datetime
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace(", ", ",")


def durations(db):
    c = sqlite3.connect(db)
    d = defaultdict(list)
    for name, dur in c.execute("select name, duration from kernels"):
        d[short(name)].append(dur)
    return d


def counters(db):
    c = sqlite3.connect(db)
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))     # kernel -> counter -> dispatch -> sum over instances
    for kname, cname, disp, val in c.execute("select kernel_name, counter_name, dispatch_id, value from counters_collection"):
        per[short(kname)][cname][disp] += val
    return {k: {cn: sum(v.values()) / len(v) for cn, v in cs.items()} for k, cs in per.items()}


def main():
    tag, key, trace_db, fetch_db, write_db = sys.argv[1:6]
    dst = os.path.join(ROOT, "profiles")
    dur = durations(trace_db)
    total = sum(sum(v) for v in dur.values())
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for k in sorted(dur, key=lambda k: -sum(dur[k])):
            v = dur[k]
            w.writerow([k, len(v), sum(v), f"{sum(v) / len(v):.1f}", f"{100.0 * sum(v) / total:.2f}", min(v), max(v)])
    fe, wr = counters(fetch_db), counters(write_db)
    march = "_march" in key
    march3 = key.endswith("_march3")
    march4 = key.endswith("_march4")
    rows = []
    fetch_kb = write_kb = 0.0
    for k in sorted(set(fe) | set(wr)):
        f = fe.get(k, {}).get("FETCH_SIZE")
        wv = wr.get(k, {}).get("WRITE_SIZE")
        rows.append([k, f, wv])
        if march4:
            use = ((k.startswith("wt::k_march3<") and ",4,false," in k) or k.startswith("wt::k_halo4<"))
        elif march3:
            use = ((k.startswith("wt::k_march3<") and ",3,false," in k) or k.startswith("wt::k_halo3<"))
        elif march:
            use = ((k.startswith("wt::k_march<") and ",false," in k) or k.startswith("wt::k_halo_from_seams"))
        else:
            use = k.startswith("wt::k_step<") and ",false," in k
        if use:
            fetch_kb += f or 0.0
            write_kb += wv or 0.0
    with open(os.path.join(dst, f"{tag}_pmc_traffic.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel", "FETCH_SIZE_KB_avg_per_dispatch", "WRITE_SIZE_KB_avg_per_dispatch", "HBM_bytes = FETCH x1024 x2 + WRITE x1024"])
        for k, f, wv in rows:
            w.writerow([k, "" if f is None else f"{f:.1f}", "" if wv is None else f"{wv:.1f}",
                        "" if f is None or wv is None else f"{f * 2048 + wv * 1024:.0f}"])
    fetch, write = fetch_kb * 1024 * 2, write_kb * 1024
    path = os.path.join(dst, "pmc_traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = {
        "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch, "write_bytes": write,
        "kernel": ("one pass = wt::k_halo4 + wt::k_march3<T,S,4,false,FD> (FOUR steps)" if march4 else
                   "one pass = wt::k_halo3 + wt::k_march3<T,S,3,false,FD> (THREE steps)" if march3 else
                   "one pass = wt::k_halo_from_seams + wt::k_march<T,S,false,FD> (TWO steps)" if march else "wt::k_step<float,false,...> (non-emitting step)"),
        "measured": datetime.date.today().isoformat() + ", rocprofv3 --pmc on one MI355X box of the gpurun pool, `python bench.py` default workload, "
                    "separate passes for FETCH_SIZE and WRITE_SIZE (not the run that prints the bench line)",
        "source": f"profiles/{tag}_pmc_traffic.csv: FETCH_SIZE KB x1024 x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KB x1024",
    }
    json.dump(data, open(path, "w"), indent=1)
    print(json.dumps(data[key], indent=1))


if __name__ == "__main__":
    main()
