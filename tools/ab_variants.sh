#!/bin/bash
# same-run A/B of kernel variants built by `python -m triplet_match_amd.build --variant NAME -D...` (libtm_hip_NAME.so):
# the headline bench command per library, three times round-robin so that clock / box drift hits all alike, then a short
# parity run (pytest -m gpu -k "score or parity") per variant.
# usage (GPU box, repo root): tools/ab_variants.sh <outdir> <variant names...>   ("base" = libtm_hip.so)
out=$1; shift
mkdir -p $out
for rep in 1 2 3; do
  for v in "$@"; do
    lib=$PWD/triplet_match_amd/libtm_hip_$v.so
    [ "$v" = base ] && lib=$PWD/triplet_match_amd/libtm_hip.so
    TM_LIB=$lib timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $BENCH_ARGS > $out/$v.$rep.json 2> $out/$v.$rep.err || echo "$v rep $rep failed" >> $out/progress.log
    python3 - $out/$v.$rep.json $v $rep <<'PY' >> $out/summary.txt
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
