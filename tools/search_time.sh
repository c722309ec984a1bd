#!/bin/bash
# steady-state searches of tools/e2e_demo at the configs[1] size, plain (no profiler): wall time and phases per call,
# then min / median over the repeats per phase
# usage (GPU box, repo root): tools/search_time.sh [repeats]     (environment switches such as TM_HASH_FIRST=1 pass through)
if [ ! -f /tmp/c2.bin ]; then
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_cpp_dropin import _fixture
_fixture("/tmp/c2.bin", model_n=100000, scene_n=1000000, copies=4)
PY
fi
g++ -std=c++17 -O2 -Iinclude tools/e2e_demo.cpp -o /tmp/e2e_demo -Ltriplet_match_amd -ltm_hip -Wl,-rpath,$PWD/triplet_match_amd -pthread || exit 1
/tmp/e2e_demo /tmp/c2.bin 1 10 ${1:-8} > /tmp/e2e_out.txt 2>&1 || { cat /tmp/e2e_out.txt; exit 1; }
grep -v "^   match\|further call\|phases \[ms\]\|sampling:" /tmp/e2e_out.txt
python3 - <<'PY'
import re, statistics
walls, phases = [], {}
for ln in open("/tmp/e2e_out.txt"):
    m = re.match(r"further call ([0-9.]+) s", ln)
    if m:
        walls.append(float(m.group(1)) * 1e3)
    if "phases [ms]" in ln and walls:
        for k, v in re.findall(r"([a-z+]+) ([0-9.]+)", ln.split("phases [ms]:")[1]):
            phases.setdefault(k, []).append(float(v))
if walls:
    print("steady-state calls: %d   wall ms: min %.2f  median %.2f  max %.2f" % (len(walls), min(walls), statistics.median(walls), max(walls)))
    print("   phase medians [ms]: " + "  ".join("%s %.2f" % (k, statistics.median(v[-len(walls):])) for k, v in phases.items()))
    print("   phase minima  [ms]: " + "  ".join("%s %.2f" % (k, min(v[-len(walls):])) for k, v in phases.items()))
PY
