#!/usr/bin/env python3
"""Prints the top kernels of a rocprofv3 rocpd database (names truncated): tools/rocpd_top.py <results.db> [n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cur = db.cursor()
print("%-60s %8s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit %d" % n):
    short = name.split("(")[0]
    if "rocprim" in short:
        short = "rocprim::" + short.split("detail::")[-1][:40]
    print("%-60s %8d %12.1f %12.1f %7.2f" % (short[:60], calls, total, avg, pct))
r = list(cur.execute("select min(start), max(end), sum(end-start), count(*) from kernels"))[0]
print("span %.1f ms, busy %.1f ms, %d dispatches" % ((r[1] - r[0]) / 1e6, r[2] / 1e6, r[3]))
