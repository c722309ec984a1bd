#!/bin/bash
# what one frame of a stream costs: the drop-in search on a kept scene, on a new scene object over the same cloud, on a
# new cloud (tools/e2e_demo at the C2 size); TM_FRAME_VERBOSE=1 prints where a new cloud's preparation goes
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_cpp_dropin import _fixture
_fixture("/tmp/c2.bin", model_n=100000, scene_n=1000000, copies=4)
PY
g++ -std=c++17 -O2 -Iinclude tools/e2e_demo.cpp -o /tmp/e2e_demo -Ltriplet_match_amd -ltm_hip -Wl,-rpath,$PWD/triplet_match_amd -pthread || exit 1
TM_FRAME_VERBOSE=1 /tmp/e2e_demo /tmp/c2.bin 1 10 3 4 4 ${STREAM_THREADS:-3} 2>&1 | grep -v "^\[search\]"
