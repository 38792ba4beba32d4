#!/usr/bin/env python3
"""Per-kernel averages of the counters of one or more `rocprofv3 --pmc` runs (rocpd sqlite): python tools/sq_summary.py <kernel substring> a.db b.db ...
Values are summed over the chip per dispatch (one row per dispatch and counter instance), then averaged over the dispatches of the kernel."""
import sqlite3, sys
from collections import defaultdict
want = sys.argv[1]
tot, nd = defaultdict(float), defaultdict(set)
for db in sys.argv[2:]:
    c = sqlite3.connect(db)
    for kname, cname, disp, val in c.execute("select kernel_name, counter_name, dispatch_id, value from counters_collection"):
        if want in kname:
            tot[(kname, cname)] += val
            nd[(kname, cname)].add((db, disp))
for (k, cn) in sorted(tot):
    print(f"{k[:60]:60s} {cn:28s} {tot[(k, cn)] / len(nd[(k, cn)]):14.4g}   ({len(nd[(k, cn)])} dispatches)")
