#!/bin/bash
# the round's closing check on the GPU box: the whole -m gpu suite, then the default bench line with its extras
# usage (repo root): tools/check_round.sh [tag]
cd $GRAFT_REPO_ROOT
tag=${1:-check}
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gputests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/${tag}_gputests.log
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
python3 - "$tag" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s_bench.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.0f  ms/step %.3f  kernel_ms %.4f  frac %s  traffic %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["traffic"]))
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"])
x = d.get("extras") or {}
print("early drop: host buffers %.0f, device-resident %.0f hypotheses/s" % (x["early_drop_ball"]["value"], x["early_drop_ball_dev"]["value"]))
print("e2e:", json.dumps(x.get("e2e_find_all_parallel")))
PY
