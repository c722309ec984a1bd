#!/bin/bash
# kernel trace of steady-state searches of tools/e2e_demo: per-kernel counts and times of ONE search
# usage (GPU box): tools/search_trace.sh [outdir]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/search_trace}
mkdir -p $out
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_cpp_dropin import _fixture
_fixture("/tmp/c2.bin", model_n=100000, scene_n=1000000, copies=4)
PY
g++ -std=c++17 -O2 -Iinclude tools/e2e_demo.cpp -o /tmp/e2e_demo -Ltriplet_match_amd -ltm_hip -Wl,-rpath,$PWD/triplet_match_amd -pthread || exit 1
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- /tmp/e2e_demo /tmp/c2.bin 1 10 4 > $out/under_trace.txt 2>&1
python3 - $out <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
segs, seg = [], [rows[0]]
for prev, cur in zip(rows, rows[1:]):
    if cur[0] - prev[1] > 100_000_000:
        segs.append(seg)
        seg = []
    seg.append(cur)
segs.append(seg)
sg = segs[-1]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in sg:
    if "rocprim" in n:
        n = "rocprim::" + n.split("detail::")[-1][:48]
    agg[n][0] += 1
    agg[n][1] += e - s
print("last search: %d dispatches, %.2f ms of kernels, span %.2f ms" % (len(sg), sum(v[1] for v in agg.values()) / 1e6, (sg[-1][1] - sg[0][0]) / 1e6))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-60s %4d  %8.1f us" % (n[:60], c, t / 1e3))
PY
