#!/bin/bash
# rocprofv3 kernel trace of the end-to-end drop-in search at the C2 size
set -e
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_cpp_dropin import _fixture
_fixture("/tmp/c2.bin", model_n=100000, scene_n=1000000, copies=4)
PY
g++ -std=c++17 -O2 -Iinclude tools/e2e_demo.cpp -o /tmp/e2e_demo -Ltriplet_match_amd -ltm_hip -Wl,-rpath,$PWD/triplet_match_amd
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/e2e_prof -o e2e -- /tmp/e2e_demo /tmp/c2.bin ${1:-0} 10
