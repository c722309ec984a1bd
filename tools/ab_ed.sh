#!/bin/bash
# A/B of the early-drop path on bench.py's extras: prefix kernel + wide continuation (default), prefix kernel + one
# workgroup per pending hypothesis (TM_ED_NARROW=1), the full kernel alone in two steps (TM_ED_ONE_KERNEL=1),
# everything in the tile-culled pass (TM_ED_ONE_KERNEL=1 TM_ABLATE=30); the end-to-end search also without the
# segments of tm_search_batch (TM_SEARCH_SEGMENT=0) and without tm_search_batch (TM_SEARCH_UNFUSED=1)
for round in 1 2; do
  for v in "default X=1" "narrow TM_ED_NARROW=1" "no_segments TM_SEARCH_SEGMENT=0" "unfused_search TM_SEARCH_UNFUSED=1" "one_kernel TM_ED_ONE_KERNEL=1" "round1_shape TM_ED_ONE_KERNEL=1 TM_ABLATE=30"; do
    label=${v%% *}; envs=${v#* }
    env $envs python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/abed_$label.json 2> gpurun_out/abed_$label.log || { echo "$label FAILED"; tail -3 gpurun_out/abed_$label.log; continue; }
    python3 -c "import json;d=json.load(open('gpurun_out/abed_$label.json'));e=d['extras'];print('$label round $round: early_drop_ball %.0f hyp/s (%.1f ms), e2e first %.3f s, second %.3f s' % (e['early_drop_ball']['value'], e['early_drop_ball']['ms'], e['e2e_find_all_parallel']['find_all_parallel_s'], e['e2e_find_all_parallel']['second_call_s']))"
  done
done
