#!/bin/bash
# round-2 evidence run on the GPU box: kernel stats + PMC passes of the default bench command, the same with the
# exhaustive voxel kernel, the kernel table, and profiles/r02/traffic.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_final
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-extras > $out/bench_under_stats.json 2> $out/bench_under_stats.log
tools/pmc_rdreq.sh $out/pmc > $out/pmc.txt 2>&1
TM_VOXEL_BRUTEFORCE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_brute -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_under_stats_brute.json 2> $out/bench_under_stats_brute.log
TM_VOXEL_BRUTEFORCE=1 tools/pmc_rdreq.sh $out/pmc_brute > $out/pmc_brute.txt 2>&1
python3 tools/make_traffic_json.py $out/pmc k_score_items $out/traffic.json > /dev/null
python3 tools/kernel_table.py $out/pmc $out/pmc_brute $out/stats/*/*kernel_stats.csv $out/stats_brute/*/*kernel_stats.csv > $out/kernel_table.md
cat $out/kernel_table.md
python3 bench.py --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.log
cut -c1-700 $out/bench_n1.json
