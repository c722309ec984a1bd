#!/bin/bash
# rocprofv3 kernel trace (+ optional PMC pass) of the end-to-end drop-in search at the C2 size
# usage: tools/e2e_profile.sh <early_drop 0|1> <outdir>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ed=${1:-1}; out=${2:-gpurun_out/e2e_prof}
mkdir -p $out
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_cpp_dropin import _fixture
_fixture("/tmp/c2.bin", model_n=100000, scene_n=1000000, copies=4)
PY
g++ -std=c++17 -O2 -Iinclude tools/e2e_demo.cpp -o /tmp/e2e_demo -Ltriplet_match_amd -ltm_hip -Wl,-rpath,$PWD/triplet_match_amd -pthread
/tmp/e2e_demo /tmp/c2.bin $ed 10 > $out/e2e_plain.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- /tmp/e2e_demo /tmp/c2.bin $ed 10 4 > $out/e2e_under_trace.txt 2>&1
python3 tools/trace_summary.py $out/trace --segments > $out/e2e_kernels.txt
cat $out/e2e_plain.txt $out/e2e_kernels.txt
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- /tmp/e2e_demo /tmp/c2.bin $ed 10 > $out/pass$i.txt 2>&1 || echo "pass $i failed" >> $out/progress.log
done
python3 tools/pmc_summary.py $out | grep -A16 -E "k_ed_fused|k_ed_bits_wide|k_ed_score_rows" > $out/k_ed_fused_pmc.txt || true
cat $out/k_ed_fused_pmc.txt
