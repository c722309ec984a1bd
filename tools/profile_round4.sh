#!/bin/bash
# round-4 evidence run on the GPU box (one gpurun call): kernel stats of the default bench command, the PMC passes
# the roofline line quotes (profiles/r04/traffic.json, with the voxel-index-folded pass of the timing-experiments
# build for voxel_line_utilisation), and the same two measurements on a 600k-point model whose near-surface brick
# set is larger than the 256 MB Infinity Cache (the DRAM share bracketed without MALL counters).
# usage (repo root): tools/profile_round4.sh   (STAGES="folded bracket" runs only those stages: stats pmc folded ed bracket bench;
# build the timing-experiments library first: python -m triplet_match_amd.build --timing-experiments)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_final
mkdir -p $out
want() { [ -z "$STAGES" ] || echo " $STAGES " | grep -q " $1 "; }
echo "start" >> $out/progress.log
if want stats; then
echo "stats" >> $out/progress.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-extras > $out/bench_under_stats.json 2> $out/bench_under_stats.log
fi
if want pmc; then
echo "pmc" >> $out/progress.log
tools/pmc_rdreq.sh $out/pmc > $out/pmc.txt 2>&1
fi
# voxel index folded into 1 MB (results invalid, timing-experiments build only): the lines that remain are not voxel lines
if want folded; then
echo "pmc folded" >> $out/progress.log
if [ -f triplet_match_amd/libtm_hip_experiments.so ]; then
  TM_LIB=$PWD/triplet_match_amd/libtm_hip_experiments.so TM_ABLATE=7 PMC_PASSES="1" tools/pmc_rdreq.sh $out/pmc_folded > $out/pmc_folded.txt 2>&1
  python3 tools/make_traffic_json.py $out/pmc k_score_items $out/traffic.json $out/pmc_folded > /dev/null
else
  python3 tools/make_traffic_json.py $out/pmc k_score_items $out/traffic.json > /dev/null
fi
cp $out/traffic.json profiles/r04/traffic.json 2>/dev/null || true
fi
if want ed; then
# the reference's default loop (neighbourhood + early-drop rule) on the same batch, device-resident: bytes and busy time
echo "early drop" >> $out/progress.log
python3 bench.py --ed-profile > $out/ed_plain.json 2> $out/ed_plain.log
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/ed_pmc -- python3 bench.py --ed-profile > $out/ed_pmc.json 2> $out/ed_pmc.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ed_trace -- python3 bench.py --ed-profile > $out/ed_trace.json 2> $out/ed_trace.log
python3 tools/make_ed_traffic_json.py $out/ed_pmc $out/ed_trace $out/ed_plain.json $out/early_drop_traffic.json > /dev/null
cp $out/early_drop_traffic.json profiles/r04/early_drop_traffic.json 2>/dev/null || true
fi
if want bracket; then
# DRAM-share bracket: a 600k-point model (brick set of the near-surface shell >> 256 MB)
echo "600k model" >> $out/progress.log
PMC_PASSES="1 2 3" tools/pmc_rdreq.sh $out/pmc_600k --model-points 600000 --copies 1 > $out/pmc_600k.txt 2>&1
python3 bench.py --steps 10 --warmup 3 --model-points 600000 --copies 1 --no-cpu-baseline --no-extras > $out/bench_600k.json 2> $out/bench_600k.log
fi
if want bench; then
echo "bench" >> $out/progress.log
python3 bench.py --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.log
cut -c1-900 $out/bench_n1.json
fi
