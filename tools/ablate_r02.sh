#!/bin/bash
# timing experiments of round 2 (results of some variants are invalid by construction)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # label, env...
  label=$1; shift
  env "$@" TM_DEBUG_STATS=1 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/abl_$label -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/abl_$label.json 2> gpurun_out/abl_$label.log || { echo "$label FAILED"; tail -3 gpurun_out/abl_$label.log; return 1; }
  echo "$label: $(grep '\[tm\] score' gpurun_out/abl_$label.log | tail -1)"
  python3 - gpurun_out/abl_$label <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "k_score" in k: agg[(k[-24:],r["Counter_Name"])].append(float(r["Counter_Value"]))
print("   "+"  ".join("%s=%.3g"%(c,sum(v)/len(v)) for (k,c),v in sorted(agg.items())))
PY
}
run base TM_X=1
run f64term TM_ABLATE=21
run base2 TM_X=1
run f64term2 TM_ABLATE=21
