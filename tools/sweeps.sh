#!/bin/bash
# the randomised sweeps against the oracle on the GPU box: tools/sweeps.sh <tag> <parity cases> <traits cases> <searches> [seed0]
cd $GRAFT_REPO_ROOT
tag=$1; seed=${5:-61000}
python3 tools/parity_sweep.py $2 $seed > gpurun_out/${tag}_parity.txt 2>&1; tail -1 gpurun_out/${tag}_parity.txt
TM_SWEEP_TRAITS=1 python3 tools/parity_sweep.py $3 $((seed + 10000)) > gpurun_out/${tag}_traits.txt 2>&1; tail -1 gpurun_out/${tag}_traits.txt
python3 tools/search_sweep.py $4 > gpurun_out/${tag}_search.txt 2>&1; tail -1 gpurun_out/${tag}_search.txt
