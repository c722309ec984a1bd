#!/bin/bash
# steady-state searches of tools/e2e_demo at the configs[1] size, plain (no profiler): wall time and phases per call
# usage (GPU box, repo root): tools/search_time.sh [repeats]
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_cpp_dropin import _fixture
_fixture("/tmp/c2.bin", model_n=100000, scene_n=1000000, copies=4)
PY
g++ -std=c++17 -O2 -Iinclude tools/e2e_demo.cpp -o /tmp/e2e_demo -Ltriplet_match_amd -ltm_hip -Wl,-rpath,$PWD/triplet_match_amd -pthread || exit 1
/tmp/e2e_demo /tmp/c2.bin 1 10 ${1:-8}
