#!/bin/bash
# same-run comparison of environment settings on the bench batch: tools/ab_env.sh <reps> "<ENV=.. ENV=..>" "<...>" ...
# (an empty string = the defaults)
reps=$1; shift
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $reps); do
  for E in "$@"; do
    env $E python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s value %.0f  ms/step %.3f  kernel_ms %.4f' % ('[$E]', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))
"
  done
done
