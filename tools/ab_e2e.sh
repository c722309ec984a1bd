#!/bin/bash
# same-run A/B of the end-to-end drop-in search (C2 size) under environment switches
# usage (GPU box): tools/ab_e2e.sh "name VAR=1 ..." "name2 ..."
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_cpp_dropin import _fixture
_fixture("/tmp/c2.bin", model_n=100000, scene_n=1000000, copies=4)
PY
g++ -std=c++17 -O2 -Iinclude tools/e2e_demo.cpp -o /tmp/e2e_demo -Ltriplet_match_amd -ltm_hip -Wl,-rpath,$PWD/triplet_match_amd
for round in 1 2; do
  for v in "$@"; do
    name=${v%% *}
    envs=${v#* }
    for ed in 1 0; do
      env $envs /tmp/e2e_demo /tmp/c2.bin $ed 10 4 > gpurun_out/e2e_${name}_ed$ed.txt 2>&1
      echo "$name round $round early_drop=$ed: $(grep -E 'find_all_parallel|second call|call [0-9]' gpurun_out/e2e_${name}_ed$ed.txt | tr '\n' ' ' | cut -c1-400)"
      grep "phases" gpurun_out/e2e_${name}_ed$ed.txt | tail -1
    done
  done
done
