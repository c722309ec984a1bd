#!/bin/bash
# kernel trace of the stream mode of tools/e2e_demo (new 1M-point clouds on T host threads with a context each):
# how much of the span has at least one kernel running.  usage (GPU box): tools/stream_trace.sh [outdir] [threads]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/stream_trace}; T=${2:-4}
mkdir -p $out
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_cpp_dropin import _fixture
_fixture("/tmp/c2.bin", model_n=100000, scene_n=1000000, copies=4)
PY
g++ -std=c++17 -O2 -Iinclude tools/e2e_demo.cpp -o /tmp/e2e_demo -Ltriplet_match_amd -ltm_hip -Wl,-rpath,$PWD/triplet_match_amd -pthread || exit 1
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- /tmp/e2e_demo /tmp/c2.bin 1 10 0 0 0 $T 1 > $out/under_trace.txt 2>&1
grep "stream" $out/under_trace.txt
python3 tools/trace_summary.py $out/trace --segments | grep "segment" > $out/segments.txt
cat $out/segments.txt
