#!/usr/bin/env python3
"""per-item time stamps of lod_feed_kernel (GARLIC_TRACE=file): pace of the longest items, makespan, load per workgroup"""
import sys, collections
rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if l.strip()]
rows = [r for r in rows if r[2] > 0]
t0 = min(r[2] for r in rows)
end = max(r[4] for r in rows)
print("items", len(rows), "makespan ms", (end - t0) / 1e5)
byw = collections.defaultdict(list)
for r in rows:
    byw[r[1]].append(r)
print("workgroups", len(byw), "items per workgroup max", max(len(v) for v in byw.values()))
print(" item  wg  start_ms  setup_us  tiles  tile_ms  cyc/window  clk_GHz")
for r in rows[:12] + rows[len(rows) // 2: len(rows) // 2 + 4] + rows[-4:]:
    i, wg, b, tb, e, cb, ctb, ce, nt = r[:9]
    ticks = e - tb
    print("%5d %4d %8.3f %8.1f %6d %8.3f %8.1f %7.2f" % (i, wg, (b - t0) / 1e5, (tb - b) / 100.0, nt, ticks / 1e5,
          (ce - ctb) / max(1, nt) / 32.0, (ce - ctb) / max(1, ticks) / 10.0))
busy = sorted(((sum(r[4] - r[2] for r in v)) / 1e5, w) for w, v in byw.items())
print("busy ms per workgroup: min %.3f median %.3f max %.3f" % (busy[0][0], busy[len(busy) // 2][0], busy[-1][0]))
