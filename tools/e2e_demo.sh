#!/bin/bash
# end-to-end drop-in demo at the C2 size (100k model / 1M scene, 4 planted instances) on the GPU box
set -e
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_cpp_dropin import _fixture
_fixture("/tmp/c2.bin", model_n=100000, scene_n=1000000, copies=4)
PY
g++ -std=c++17 -O2 -Iinclude tools/e2e_demo.cpp -o /tmp/e2e_demo -Ltriplet_match_amd -ltm_hip -Wl,-rpath,$PWD/triplet_match_amd -pthread
TM_SEARCH_VERBOSE=1 /tmp/e2e_demo /tmp/c2.bin 0 10
/tmp/e2e_demo /tmp/c2.bin 1 10
