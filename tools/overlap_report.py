#!/usr/bin/env python3
"""Is the ghost refresh of a column slab overlapped with its interior step?  Reads a rocprofv3 --kernel-trace run (rocpd sqlite) of
`bench.py --local-slabs P` and, for every refresh step of every slab, compares the interval of the ghost-copy kernel (wt k_ghost_copy, on
the slab's comm stream) with the interval of the interior-columns k_step launch that step_compute enqueued beside it (csrc/windtunnel.hip
halo_begin / step_compute): [refresh on s_comm] || [interior columns on s_compute] -> edge strips after the ev_halo event.

    python tools/overlap_report.py gpurun_out/local_slabs/trace8/t_results.db [out.txt]
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, stream_id, queue_id, start, end, grid_x, grid_y from kernels order by start"))
    t0 = rows[0][3]
    copies = [r for r in rows if "k_ghost_copy" in r[0]]
    steps = [r for r in rows if "k_step<" in r[0]]
    print(f"{db}: {len(rows)} kernel launches, {len(copies)} ghost-copy launches, {len(steps)} k_step launches", file=out)
    by_stream = defaultdict(list)
    for r in steps:
        by_stream[r[1]].append(r)
    own_inside = own_before_end = any_overlap = dep_ok = n = 0
    lines = []
    for cp in copies:
        # A handle creates its compute stream, then its comm stream (ids s, s + 1).  Of the three k_step launches of a refresh step on the
        # compute stream — interior columns, left strip, right strip — the interior one has by far the largest grid and comes first.
        mine = sorted([s for s in by_stream.get(cp[1] - 1, []) if abs(s[3] - cp[3]) < 3000000], key=lambda s: s[3])
        if not mine:
            continue
        big = max(s[5] for s in mine)
        interiors = [s for s in mine if s[5] == big]
        best = min(interiors, key=lambda s: abs(s[3] - cp[3]))
        strips = [s for s in mine if s[5] < big and s[3] >= best[3]][:2]
        n += 1
        ov_own = min(best[4], cp[4]) - max(best[3], cp[3])
        own_inside += best[3] <= cp[3] and cp[4] <= best[4]
        own_before_end += cp[3] < best[4]                         # the copy did not wait for its slab's interior kernel
        any_overlap += any(min(s[4], cp[4]) - max(s[3], cp[3]) > 0 for s in steps if s[5] >= big // 2)
        dep_ok += all(st[3] >= cp[4] for st in strips) if strips else 0
        if len(lines) < 16:
            lines.append(f"  slab stream {cp[1] - 1:3d}: copy [{(cp[3] - t0) / 1e3:10.2f}, {(cp[4] - t0) / 1e3:10.2f}] ({(cp[4] - cp[3]) / 1e3:6.2f} us)  "
                         f"interior k_step [{(best[3] - t0) / 1e3:10.2f}, {(best[4] - t0) / 1e3:10.2f}] ({(best[4] - best[3]) / 1e3:6.2f} us)  "
                         f"edge strips start {', '.join(f'{(st[3] - t0) / 1e3:.2f}' for st in strips) or '-'}  "
                         f"| copy inside own interior: {100 * max(ov_own, 0) / max(1, cp[4] - cp[3]):5.1f} %")
    print(f"refresh steps matched: {n}", file=out)
    print(f"  ghost copy starts before its own slab's interior kernel ends (never queued behind it): {own_before_end} of {n}", file=out)
    print(f"  ghost copy runs while SOME slab's interior k_step is executing on the device:          {any_overlap} of {n}", file=out)
    print(f"  ghost copy interval entirely inside its OWN slab's interior interval:                  {own_inside} of {n}", file=out)
    print(f"  both edge strips of the slab start after its ghost copy has ended (ev_halo):           {dep_ok} of {n}", file=out)
    print("first refresh steps (us relative to the first kernel of the run):", file=out)
    for l in lines:
        print(l, file=out)
    # per-kernel totals
    tot = defaultdict(lambda: [0, 0])
    for r in rows:
        k = r[0].split("(")[0].replace("void ", "")
        tot[k][0] += 1
        tot[k][1] += r[4] - r[3]
    print("kernel totals:", file=out)
    for k, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"  {k[:70]:70s} {n:6d} launches {d / 1e3:10.1f} us  avg {d / n / 1e3:8.2f} us", file=out)


if __name__ == "__main__":
    main()
