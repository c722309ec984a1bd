#!/bin/bash
# what bounds the kernels of the preprocessing rows (SURVEY 8f row 1 and model init): three rocprofv3 --pmc passes of
# tools/time_model_init.py (voxel packets, seeds, pair table, k-nearest walks of the model) and tools/time_curvature.py
# usage (GPU box, repo root): tools/pmc_preprocessing.sh <outdir>
out=${1:-gpurun_out/pmc_pre}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/model_pass$i -- python3 tools/time_model_init.py > $out/model_pass$i.log 2>&1 || echo "model pass $i failed"
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/curv_pass$i -- python3 tools/time_curvature.py > $out/curv_pass$i.log 2>&1 || echo "curvature pass $i failed"
  echo "pass $i done"
done
python3 - $out <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/*pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not any(k in name for k in ("k_voxel_nn_packet", "k_knn_groups", "k_curvature", "k_pairs", "k_lbvh_level", "k_knn2")):
            continue
        a = agg[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
res = {k: {c: v[0] / max(v[1], 1) for c, v in cs.items()} for k, cs in agg.items()}
for k, cs in sorted(res.items()):
    # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs; a wave64 VALU instruction holds a SIMD for 4 cycles
    valu, cyc = cs.get("SQ_INSTS_VALU", 0), cs.get("GRBM_GUI_ACTIVE", 0) / 8.0
    cs["valu_issue_share"] = (valu * 4.0 / (256 * 4 * cyc)) if cyc else 0
    cs["scalar_issue_share"] = ((cs.get("SQ_INSTS_SALU", 0) + cs.get("SQ_INSTS_SMEM", 0)) / (256 * cyc)) if cyc else 0
    print("%-70s VALU insts/dispatch %.3g  cycles %.3g  VALU issue share %.2f  scalar issue share %.2f  LDS insts %.3g  fabric rd %.3g wr %.3g  L2 hit %.2f" % (
        k[:70], valu, cyc, cs["valu_issue_share"], cs["scalar_issue_share"], cs.get("SQ_INSTS_LDS", 0),
        cs.get("TCC_EA0_RDREQ_sum", 0), cs.get("TCC_EA0_WRREQ_sum", 0),
        cs.get("TCC_HIT_sum", 0) / max(cs.get("TCC_HIT_sum", 0) + cs.get("TCC_MISS_sum", 0), 1)))
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
PY
