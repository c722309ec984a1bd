#!/bin/bash
# L2->fabric read requests of the grouped scoring kernel on configs 3 / 4 / 5, one rocprofv3 --pmc pass per
# configuration (tools/report_configs.py --only), then the report itself with the measured bytes merged in.
# usage (GPU box, repo root): tools/pmc_configs.sh <outdir>
out=${1:-gpurun_out/cfgpmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
for cfg in 3 4 5.0 5.15 5.31; do
  echo "config $cfg" >> $out/progress.log
  timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/c$cfg/pass1 -- python3 tools/report_configs.py --only $cfg > $out/c$cfg.json 2> $out/c$cfg.log || echo "config $cfg failed rc=$?" >> $out/progress.log
done
python3 - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for cfg in ("3", "4", "5.0", "5.15", "5.31"):
    n128 = n64 = n32 = 0.0
    launches = 0
    for f in glob.glob("%s/c%s/pass1/*/*_counter_collection.csv" % (out, cfg)):
        for r in csv.DictReader(open(f)):
            if "k_score_items" not in r["Kernel_Name"]:
                continue
            v = float(r["Counter_Value"])
            c = r["Counter_Name"]
            if c == "TCC_EA0_RDREQ_128B_sum":
                n128 += v
            elif c == "TCC_EA0_RDREQ_64B_sum":
                n64 += v
            elif c == "TCC_EA0_RDREQ_32B_sum":
                n32 += v
            elif c == "TCC_EA0_RDREQ_sum":
                launches += 1
    if launches:
        res[cfg] = (128.0 * n128 + 64.0 * n64 + 32.0 * n32) / launches
json.dump(res, open(out + "/configs_traffic.json", "w"), indent=1)
print(res)
PY
python3 tools/report_configs.py --traffic $out/configs_traffic.json > $out/configs_3_4_5.jsonl 2> $out/report.log
cat $out/configs_3_4_5.jsonl | cut -c1-400
