#!/bin/bash
# A/B of the early-drop continuation: interval reductions (default) against bit rows (TM_ED_HIST=0) — the whole batch
# device-resident (bench.py --ed-profile) and the steady-state drop-in search (tools/e2e_demo) with a kernel trace
# usage (repo root, GPU box): tools/ab_hist.sh [outdir]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/ab_hist}
mkdir -p $out
for v in 1 0; do
  TM_ED_HIST=$v python3 bench.py --ed-profile > $out/ed_hist$v.json 2> $out/ed_hist$v.log
done
TM_ED_PREFIX_GROUP=0 python3 bench.py --ed-profile > $out/ed_group0.json 2> $out/ed_group0.log
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_cpp_dropin import _fixture
_fixture("/tmp/c2.bin", model_n=100000, scene_n=1000000, copies=4)
PY
g++ -std=c++17 -O2 -Iinclude tools/e2e_demo.cpp -o /tmp/e2e_demo -Ltriplet_match_amd -ltm_hip -Wl,-rpath,$PWD/triplet_match_amd -pthread
for v in 1 0; do
  TM_ED_HIST=$v /tmp/e2e_demo /tmp/c2.bin 1 10 3 3 > $out/e2e_hist$v.txt 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- /tmp/e2e_demo /tmp/c2.bin 1 10 4 > $out/e2e_under_trace.txt 2>&1
python3 tools/trace_summary.py $out/trace --segments > $out/e2e_kernels.txt
grep -h "ms\|early_drop" $out/ed_hist1.json $out/ed_hist0.json $out/ed_group0.json | cut -c1-200
tail -4 $out/e2e_hist1.txt; tail -4 $out/e2e_hist0.txt; tail -12 $out/e2e_kernels.txt
