#!/bin/bash
# the context's block cache (option pool_mib) off / on: the drop-in search on a prepared scene and on a NEW scene
# object per call (frames of a stream), tools/e2e_demo at the C2 size.  usage (GPU box): tools/ab_pool.sh [outdir]
cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/ab_pool}
mkdir -p $out
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_cpp_dropin import _fixture
_fixture("/tmp/c2.bin", model_n=100000, scene_n=1000000, copies=4)
PY
g++ -std=c++17 -O2 -Iinclude tools/e2e_demo.cpp -o /tmp/e2e_demo -Ltriplet_match_amd -ltm_hip -Wl,-rpath,$PWD/triplet_match_amd -pthread
for v in 0 1024 0 1024; do
  TM_POOL_MIB=$v /tmp/e2e_demo /tmp/c2.bin 1 10 3 5 > $out/e2e_pool$v.txt 2>&1
  echo "== pool_mib $v"; grep -h "model init\|second call\|further call\|new scene" $out/e2e_pool$v.txt
done
python3 tools/leak_check.py > $out/leak_check.txt 2>&1; tail -3 $out/leak_check.txt
