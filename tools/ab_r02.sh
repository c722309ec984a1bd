#!/bin/bash
# A/B timing of scoring-kernel variants without a profiler attached: tools/ab_r02.sh "<label> <ENV=..>" ...
# every variant runs twice, interleaved, 20 timed steps each; prints the HIP-event kernel ms of bench.py
for round in 1 2; do
  for v in "$@"; do
    label=${v%% *}; envs=${v#* }
    env $envs python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/ab_$label.json 2> gpurun_out/ab_$label.log || { echo "$label FAILED"; tail -3 gpurun_out/ab_$label.log; continue; }
    python3 -c "import json;d=json.load(open('gpurun_out/ab_$label.json'));print('$label round $round: kernel_ms %.3f  ms_per_step %.3f' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
  done
done
