#!/bin/bash
# usage (on the GPU box, repo root): tools/profile_round.sh <outdir>
# rocprofv3 --kernel-trace --stats of the default bench command + separate FETCH_SIZE / WRITE_SIZE passes
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/bench_stats.json 2> $out/bench_stats.log || echo "stats failed" >> $out/progress.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --no-cpu-baseline > $out/bench_fetch.json 2> $out/bench_fetch.log || echo "fetch failed" >> $out/progress.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --no-cpu-baseline > $out/bench_write.json 2> $out/bench_write.log || echo "write failed" >> $out/progress.log
python3 - "$out" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
res = {}
for name in ("fetch", "write"):
    for f in glob.glob(out + "/" + name + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            res.setdefault(k, {})[c] = {"dispatches": len(v), "mean_KB": sum(v) / len(v)}
json.dump(res, open(out + "/pmc_fetch_write.json", "w"), indent=1)
for f in glob.glob(out + "/stats/*/*_kernel_stats.csv"):
    print(open(f).read()[:3000])
PY
