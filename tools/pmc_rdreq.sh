#!/bin/bash
# usage (GPU box, repo root): tools/pmc_rdreq.sh <outdir> [bench args...]
# Separate small rocprofv3 --pmc passes that settle how many bytes the scoring kernel really moves:
# L2->fabric read requests by size (32 / 64 / 128 B), the DRAM share, L2 requests, and L1 (TCP) accesses.
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum TCC_READ_sum TCC_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
           "TA_BUSY_avr TA_TA_BUSY_sum" "GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  # PMC_PASSES="9 10": only these passes (the others keep what an earlier call left in <outdir>)
  if [ -n "$PMC_PASSES" ] && ! echo " $PMC_PASSES " | grep -q " $i "; then continue; fi
  echo "pass $i: $set" >> $out/progress.log
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > $out/pass$i.json 2> $out/pass$i.log || echo "pass $i failed rc=$?" >> $out/progress.log
done
python3 tools/pmc_summary.py $out
# the raw counter tables are tens of MB per pass (gpurun copies back at most 64 MiB): the summary is what is kept
[ -n "$PMC_KEEP_RAW" ] || rm -rf $out/pass*/
