#!/usr/bin/env python3
"""Condense a gpurun_out/prof_{trace,fetch,write} triple (rocprofv3 CSV output of bench.py) into the
files kept under profiles/:  <tag>_kernel_stats.csv, <tag>_pmc_{fetch,write}_k_step.csv, and the
pmc_traffic.json entry bench.py reports as roofline.traffic.
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE (KB) x 1024 x 2 on gfx950 (the counter tallies
128-B requests at 64 B for wide coalesced reads), WRITE_SIZE (KB) x 1024."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, key = sys.argv[1], sys.argv[2]            # e.g. r01_b 4096x4096_float32
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
shutil.copy(glob.glob(os.path.join(src, "prof_trace/*/*_kernel_stats.csv"))[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
vals = {}
fused = False
for name in ("fetch", "write"):
    f = glob.glob(os.path.join(src, f"prof_{name}/*/*_counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_step" in r["Kernel_Name"]]
    with open(os.path.join(dst, f"{tag}_pmc_{name}_k_step.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value_KB", "Grid_Size", "VGPR_Count", "SGPR_Count"])
        for r in rows:
            w.writerow([r["Kernel_Name"][:72], r["Counter_Name"], r["Counter_Value"], r["Grid_Size"], r["VGPR_Count"], r["SGPR_Count"]])
    groups = {}
    for r in rows:
        head = r["Kernel_Name"].split("(")[0]
        if "true" in head.replace("k_step_list<float, false", ""):      # skip the macro-emitting variants
            if "k_step2<true>" in head or "k_step<float, true" in head or "k_step_list<float, true" in head:
                continue
        groups.setdefault(head, []).append(float(r["Counter_Value"]))
    fused = any("k_step2" in k for k in groups)
    if fused:      # one pass over the lattice = k_step2 + the two zone passes (each launched once per pass)
        vals[name] = sum(sum(v) / len(v) for k, v in groups.items() if "k_step2" in k or "k_step_list" in k)
    else:
        v = [x for k, vv in groups.items() if "k_step<" in k for x in vv]
        vals[name] = sum(v) / len(v)
fetch, write = vals["fetch"] * 1024 * 2, vals["write"] * 1024
if fused:
    key += "_fused"
path = os.path.join(dst, "pmc_traffic.json")
data = json.load(open(path)) if os.path.exists(path) else {}
data[key] = {"hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch, "write_bytes": write,
             "kernel": ("one pass = wt::k_step2<false> + k_step_list<...,1> + k_step_list<...,2> (TWO steps)" if fused
                        else "wt::k_step<float,false,...> (non-emitting step)"),
             "source": f"profiles/{tag}_pmc_fetch_k_step.csv + {tag}_pmc_write_k_step.csv: separate --pmc passes; "
                       "FETCH_SIZE KB x1024 x2 (gfx950 correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE KB x1024"}
json.dump(data, open(path, "w"), indent=1)
print(json.dumps(data[key], indent=1))
