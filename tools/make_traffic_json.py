#!/usr/bin/env python3
"""Turns the PMC passes of tools/pmc_rdreq.sh into profiles/rNN/traffic.json (what bench.py quotes as
roofline.traffic).  usage: [TM_WORKLOAD=c3|c4|c5] tools/make_traffic_json.py <pmc outdir> [kernel substring] [out path] [folded outdir]
(the passes of `bench.py --workload cX` go to profiles/rNN/traffic_cX.json)
<folded outdir> = the pass-1 counters of the same command on the timing-experiments build with TM_ABLATE=7 (voxel
index folded into 1 MB): the difference of the line counts = the lines the voxel gathers fetch.

HBM-side bytes per launch = sum over the L2's fabric read requests by size (TCC_EA0_RDREQ_32B/64B/128B;
rocprofv3's FETCH_SIZE prices every non-32B request at 64 B, which is why it reads half of the real
bytes on gfx950 when the requests are 128 B) + the write requests. Infinity-Cache hits are included
(the counters sit on the L2's fabric side), so this is an upper bound of DRAM traffic.
The source hash ties the file to the kernel sources it was measured on."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_hash)

out = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_score"
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r03", "traffic.json")
folded_dir = sys.argv[4] if len(sys.argv) > 4 else None
summ = json.load(open(os.path.join(out, "summary.json")))
name = [k for k in summ if kern in k]
if not name:
    raise SystemExit("no kernel matching %r in %s" % (kern, list(summ)))
name = max(name, key=lambda k: summ[k].get("TCC_EA0_RDREQ_sum", {}).get("mean", 0))
c = {k: v["mean"] for k, v in summ[name].items()}
n128, n64, n32 = c["TCC_EA0_RDREQ_128B_sum"], c["TCC_EA0_RDREQ_64B_sum"], c["TCC_EA0_RDREQ_32B_sum"]
other = c["TCC_EA0_RDREQ_sum"] - n128 - n64 - n32
rd = 128.0 * n128 + 64.0 * n64 + 32.0 * n32 + 64.0 * max(other, 0.0)
wr64 = c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
wr = 64.0 * wr64 + 32.0 * max(c.get("TCC_EA0_WRREQ_sum", 0.0) - wr64, 0.0)
line = json.loads(open(os.path.join(out, "pass1.json")).read().strip().splitlines()[-1])
k_ms = line["roofline"]["kernel_ms"]
cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # summed over the 8 XCDs
res = {
    "kernel": name,
    "workload": ({"model_points": 100000, "scene_points": 1000000, "hyp_per_batch": 65536, "mode": "auto"}
                 if os.environ.get("TM_WORKLOAD", "c2") == "c2" else
                 {"workload": os.environ["TM_WORKLOAD"], "hyp_per_batch": 65536, "mode": "auto"}),
    "source_hash": bench.source_hash(),
    "source": "tools/pmc_rdreq.sh: separate rocprofv3 --pmc passes of `python3 bench.py --steps 2 --warmup 1 "
              "--no-cpu-baseline --no-extras`, mean per dispatch",
    "l2_fabric_read_requests": c["TCC_EA0_RDREQ_sum"], "l2_fabric_read_128B": n128, "l2_fabric_read_64B": n64,
    "l2_fabric_read_32B": n32, "l2_fabric_read_to_dram": c.get("TCC_EA0_RDREQ_DRAM_sum"),
    "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
    "kernel_ms_under_profiler": k_ms,
    "hbm_GBs_under_profiler": (rd + wr) / (k_ms * 1e-3) / 1e9,
    "l2_hit": c.get("TCC_HIT_sum"), "l2_miss": c.get("TCC_MISS_sum"), "l2_read_requests": c.get("TCC_READ_sum"),
    "tcp_total_accesses": c.get("TCP_TOTAL_ACCESSES_sum"), "tcp_cache_accesses": c.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
    "tcp_to_l2_read_requests": c.get("TCP_TCC_READ_REQ_sum"),
    "shader_cycles_per_launch": cycles,
    "sq_wave_cycles": c.get("SQ_WAVE_CYCLES"), "sq_wait_any": c.get("SQ_WAIT_ANY"),
    "sq_insts_vmem_rd": c.get("SQ_INSTS_VMEM_RD"), "sq_insts_valu": c.get("SQ_INSTS_VALU"),
    "ta_busy_sum": c.get("TA_TA_BUSY_sum"),
}
if folded_dir:
    fs = json.load(open(os.path.join(folded_dir, "summary.json")))
    fname = [k for k in fs if kern in k]
    if fname:
        fname = max(fname, key=lambda k: fs[k].get("TCC_EA0_RDREQ_sum", {}).get("mean", 0))
        folded = fs[fname]["TCC_EA0_RDREQ_sum"]["mean"]
        res["l2_fabric_read_requests_voxel_index_folded"] = folded
        res["voxel_lines_per_launch"] = max(c["TCC_EA0_RDREQ_sum"] - folded, 0.0)
if cycles and res["tcp_cache_accesses"]:
    # L1 (TCP): one 64-B tag lookup per cache access, 64 B/clk per CU, 256 CUs
    res["l1_frac_of_peak_at_profiled_ms"] = res["tcp_cache_accesses"] / (256.0 * cycles)
    res["ta_busy_frac"] = res["ta_busy_sum"] / (256.0 * cycles) if res["ta_busy_sum"] else None
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1))
