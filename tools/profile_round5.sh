#!/bin/bash
# round-5 evidence run on the GPU box: kernel stats of the default bench command, the PMC passes the roofline line
# quotes for every workload (profiles/r05/traffic.json for c2 with the voxel-index-folded pass of the timing-experiments
# build; traffic_c3.json / _c4 / _c5 from passes 1, 3, 5, 7), the early-drop batch, and the bench lines themselves.
# usage (repo root): tools/profile_round5.sh   (STAGES="stats pmc folded wl ed search bench" selects)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_final
mkdir -p $out profiles/r05
want() { [ -z "$STAGES" ] || echo " $STAGES " | grep -q " $1 "; }
echo "start" >> $out/progress.log
if want stats; then
echo "stats" >> $out/progress.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-extras > $out/bench_under_stats.json 2> $out/bench_under_stats.log
cp $out/stats/*/*_kernel_stats.csv $out/score_kernel_stats.csv; rm -rf $out/stats
fi
if want pmc; then
echo "pmc" >> $out/progress.log
tools/pmc_rdreq.sh $out/pmc > $out/pmc.txt 2>&1
fi
if want folded; then
echo "pmc folded" >> $out/progress.log
if [ -f triplet_match_amd/libtm_hip_experiments.so ]; then
  TM_LIB=$PWD/triplet_match_amd/libtm_hip_experiments.so TM_ABLATE=7 PMC_PASSES="1" tools/pmc_rdreq.sh $out/pmc_folded > $out/pmc_folded.txt 2>&1
  python3 tools/make_traffic_json.py $out/pmc k_score_items $out/traffic.json $out/pmc_folded > /dev/null
else
  python3 tools/make_traffic_json.py $out/pmc k_score_items $out/traffic.json > /dev/null
fi
fi
if want wl; then
for wl in c3 c4 c5; do
  echo "pmc $wl" >> $out/progress.log
  PMC_PASSES="1 3 5 7" tools/pmc_rdreq.sh $out/pmc_$wl --workload $wl > $out/pmc_$wl.txt 2>&1
  TM_WORKLOAD=$wl python3 tools/make_traffic_json.py $out/pmc_$wl k_score_items $out/traffic_$wl.json > /dev/null
done
fi
if want ed; then
echo "early drop" >> $out/progress.log
python3 bench.py --ed-profile > $out/ed_plain.json 2> $out/ed_plain.log
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/ed_pmc -- python3 bench.py --ed-profile > $out/ed_pmc.json 2> $out/ed_pmc.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ed_trace -- python3 bench.py --ed-profile > $out/ed_trace.json 2> $out/ed_trace.log
python3 tools/make_ed_traffic_json.py $out/ed_pmc $out/ed_trace $out/ed_plain.json $out/early_drop_traffic.json > /dev/null
cp $out/ed_trace/*/*_kernel_stats.csv $out/early_drop_kernel_stats.csv; rm -rf $out/ed_pmc $out/ed_trace
fi
if want search; then
echo "search" >> $out/progress.log
tools/search_time.sh 30 > $out/search_time.txt 2>&1
tools/search_trace.sh $out/search_trace > $out/search_timeline.txt 2>&1; rm -rf $out/search_trace
fi
if want bench; then
echo "bench" >> $out/progress.log
# the lines quote the passes made above (same sources, same box): put them where bench.py looks
cp $out/traffic*.json $out/early_drop_traffic.json profiles/r05/ 2>/dev/null
python3 bench.py --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.log
for wl in c3 c4 c5; do python3 bench.py --workload $wl --steps 10 --warmup 2 > $out/bench_$wl.json 2> $out/bench_$wl.log; done
cut -c1-600 $out/bench_n1.json
fi
echo "done" >> $out/progress.log
