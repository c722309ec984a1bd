#!/bin/bash
# same-run A/B of two builds of the library on the bench batch: tools/ab_lib.sh <libA.so> <libB.so> [reps] [bench args...]
# (alternating runs; prints value, ms_per_step and kernel_ms per run)
A=$1; B=$2; reps=${3:-3}; shift 3 2>/dev/null
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $reps); do
  for L in "$A" "$B"; do
    TM_LIB=$PWD/$L python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s value %.0f  ms/step %.3f  kernel_ms %.4f  best %s' % ('$L', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['best']['inlier_count']))
"
  done
done
