#!/bin/bash
# usage: tools/pmc_passes.sh <outdir> <bench args...>   — separate, small PMC passes over the same bench command
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
           "TA_BUSY_avr TA_TA_BUSY_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "pass $i: $set" >> $out/progress.log
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- python3 bench.py "$@" > $out/pass$i.json 2> $out/pass$i.log || echo "pass $i failed rc=$?" >> $out/progress.log
done
python3 - "$out" <<'PY'
import csv,glob,collections,json,sys
out=sys.argv[1]
agg=collections.defaultdict(list)
for f in glob.glob(out+"/pass*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'k_score' in k or 'k_gen' in k:
            agg[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
res={}
for (k,c),v in sorted(agg.items()):
    res.setdefault(k,{})[c]={"n":len(v),"mean":sum(v)/len(v)}
json.dump(res,open(out+"/summary.json","w"),indent=1)
for k in res:
    print(k)
    for c,v in res[k].items(): print("   %-40s n=%d mean=%.5g"%(c,v["n"],v["mean"]))
PY
