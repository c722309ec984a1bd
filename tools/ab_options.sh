#!/bin/bash
# same-run A/B of context options on the headline bench command: tools/ab_options.sh <outdir> "NAME=ENV=VAL ..." ...
# e.g. tools/ab_options.sh gpurun_out/ab "base=" "novc=TM_VOXEL_CODES=0"
out=$1; shift
mkdir -p $out
for rep in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    env $envs timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $BENCH_ARGS > $out/$name.$rep.json 2> $out/$name.$rep.err || echo "$name rep $rep failed" >> $out/progress.log
    python3 - $out/$name.$rep.json $name $rep <<'PY' >> $out/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-12s rep %s  value %.4e  ms/step %.3f  kernel_ms %.3f" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"]))
except Exception as e:
    print("%-12s rep %s  FAILED %s" % (sys.argv[2], sys.argv[3], e))
PY
  done
done
cat $out/summary.txt
