#!/usr/bin/env python3
"""Per-kernel totals, GPU-busy time and span of a rocprofv3 --kernel-trace CSV directory.
usage: tools/trace_summary.py <dir> [first-kernel-substring]
With a substring the span starts at the first dispatch whose name contains it (to leave model init out)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
start_at = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
rows = []
for f in glob.glob(d + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
if start_at:
    k0 = next((i for i, r in enumerate(rows) if start_at in r[2]), 0)
    rows = rows[k0:]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    if "rocprim" in n:
        n = "rocprim::" + n.split("detail::")[-1][:40]
    agg[n][0] += 1
    agg[n][1] += e - s
busy = sum(v[1] for v in agg.values())
span = rows[-1][1] - rows[0][0] if rows else 0
print("%-56s %8s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-56s %8d %12.1f %12.1f %7.2f" % (n[:56], c, t / 1e3, t / 1e3 / c, 100.0 * t / max(busy, 1)))
print("span %.1f ms, busy %.1f ms (%.2f of the span), %d dispatches" % (span / 1e6, busy / 1e6, busy / max(span, 1), len(rows)))
if "--segments" in sys.argv:
    # cut the trace where no kernel starts for 0.1 s (e2e_demo sleeps between its repeated searches)
    seg, segs = [rows[0]], []
    for prev, cur in zip(rows, rows[1:]):
        if cur[0] - prev[1] > 100_000_000:
            segs.append(seg)
            seg = []
        seg.append(cur)
    segs.append(seg)
    for k, sg in enumerate(segs):
        b = sum(e - s for s, e, _ in sg)
        sp = sg[-1][1] - sg[0][0]
        # kernels of several streams overlap: the time at least one kernel runs (union of the intervals)
        u, cur_s, cur_e = 0, None, None
        for s_, e_, _ in sorted(sg):
            if cur_e is None or s_ > cur_e:
                if cur_e is not None:
                    u += cur_e - cur_s
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        if cur_e is not None:
            u += cur_e - cur_s
        print("segment %d: span %.1f ms, busy %.1f ms (%.2f of the span), at least one kernel running %.1f ms (%.2f), %d dispatches"
              % (k, sp / 1e6, b / 1e6, b / max(sp, 1), u / 1e6, u / max(sp, 1), len(sg)))
