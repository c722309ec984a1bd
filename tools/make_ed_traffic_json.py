#!/usr/bin/env python3
"""L2->fabric bytes of ONE tm_score_hypotheses_early_drop_ball_dev call on the bench's batch, from a rocprofv3 --pmc
pass (TCC_EA0_RDREQ by size) and a --kernel-trace pass of `python3 bench.py --ed-profile` (three calls):
every dispatch of the early-drop path's kernels (k_ed_*, k_ball_*) is summed and divided by the three calls.
usage: tools/make_ed_traffic_json.py <pmc dir> <trace dir> <bench --ed-profile json> [out]"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pmc_dir, trace_dir, line = sys.argv[1], sys.argv[2], sys.argv[3]
dst = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "r03", "early_drop_traffic.json")
CALLS = 3.0
want = ("k_ed_", "k_ball_", "k_pending_sort", "k_segment_first")
n = {"TCC_EA0_RDREQ_128B_sum": 0.0, "TCC_EA0_RDREQ_64B_sum": 0.0, "TCC_EA0_RDREQ_32B_sum": 0.0, "TCC_EA0_RDREQ_sum": 0.0}
per_kernel = {}
for f in glob.glob(pmc_dir + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(w in k for w in want) or r["Counter_Name"] not in n:
            continue
        n[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_EA0_RDREQ_sum":
            per_kernel[k] = per_kernel.get(k, 0.0) + float(r["Counter_Value"])
other = n["TCC_EA0_RDREQ_sum"] - n["TCC_EA0_RDREQ_128B_sum"] - n["TCC_EA0_RDREQ_64B_sum"] - n["TCC_EA0_RDREQ_32B_sum"]
rd = 128.0 * n["TCC_EA0_RDREQ_128B_sum"] + 64.0 * n["TCC_EA0_RDREQ_64B_sum"] + 32.0 * n["TCC_EA0_RDREQ_32B_sum"] + 64.0 * max(other, 0.0)
busy = 0.0
disp = 0
for f in glob.glob(trace_dir + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if any(w in k for w in want):
            busy += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            disp += 1
fig = json.loads(open(line).read().strip().splitlines()[-1])["early_drop_ball_dev"]
res = {"what": "tm_score_hypotheses_early_drop_ball_dev on the bench's batch, one call",
       "source": "tools/profile_round5.sh: rocprofv3 --pmc TCC_EA0_RDREQ_* and --kernel-trace of `python3 bench.py --ed-profile`",
       "source_hash": bench.ed_source_hash(), "hypotheses": fig["hypotheses"],
       "fabric_read_requests_per_call": n["TCC_EA0_RDREQ_sum"] / CALLS, "fabric_bytes_per_call": rd / CALLS,
       "kernel_busy_ms_per_call": busy / CALLS, "dispatches_per_call": disp / CALLS,
       "requests_per_call_by_kernel": {k: v / CALLS for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1])},
       "ms_hip_events_unprofiled": fig["ms"]}
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1))
