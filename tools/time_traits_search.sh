#!/bin/bash
# the two traits searches (configs 3 / 4 through scene<Point>::find_all_parallel<Traits>) with the driver's and the
# search's progress lines: where the seconds go
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 - <<'PY'
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_cpp_dropin import _compile, _traits_fixture
from triplet_match_amd import synth
_compile("test_traits_search")
model, _ = synth.synth_wall_panel(15_000, seed=5)
_traits_fixture("/tmp/indoor.bin", model, synth.synth_indoor(1_000_000, 15_000, seed=6))
model, _ = synth.synth_pipe_section(50_000, seed=3)
_traits_fixture("/tmp/pipes.bin", model, synth.synth_pipe_plant(500_000, 50_000, seed=4))
PY
for cfg in "indoor plane2" "pipes cylinder"; do
  set -- $cfg
  TM_SEARCH_VERBOSE=1 tests/cpp/_build/test_traits_search /tmp/$1.bin $2 0.05 > gpurun_out/traits_$1.txt 2>&1
  grep -E "find_all_parallel|PASSED|FAILED|model:" gpurun_out/traits_$1.txt
  grep -c "projector searched" gpurun_out/traits_$1.txt
done
