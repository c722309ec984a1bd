#!/bin/bash
# kernel durations and issue / wait counters of the early-drop batch (bench.py --ed-profile)
# usage (repo root, GPU box): tools/ed_pmc.sh [outdir]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/ed_pmc}
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --ed-profile > $out/trace.json 2> $out/trace.log
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- python3 bench.py --ed-profile > $out/pass$i.txt 2>&1 || echo "pass $i failed" >> $out/progress.log
done
python3 tools/pmc_summary.py $out > $out/summary.txt 2>&1 || true
python3 tools/trace_summary.py $out/trace | head -14
grep -A24 -E "k_ed_prefix_group|k_ed_hist_list" $out/summary.txt | head -80
