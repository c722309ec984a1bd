#!/usr/bin/env python3
"""profiles/r02 kernel table (the build kernels are compute / LDS bound and are reported as
achieved op rates + LDS traffic, not as an HBM fraction).
usage: tools/kernel_table.py <pmc dir default> <pmc dir TM_VOXEL_BRUTEFORCE=1> <kernel_stats.csv default> <kernel_stats.csv brute>
Per kernel: launches, mean duration, VALU lane-operations/s (SQ_INSTS_VALU x 64 lanes / time: an instruction
is counted once per wave; FMA contraction is off, so one lane-op = one flop or one integer op), the analytic
flop rate where the work is known in closed form, LDS instructions and the bytes they move at most
(x 64 lanes x 4 B), LDS bank-conflict cycles, VMEM read instructions, L2->fabric bytes."""
import csv
import json
import sys


def stats(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["Name"].split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
    return out


def main():
    pmc_d, pmc_b, st_d, st_b = sys.argv[1:5]
    rows = []
    for label, pmc_dir, st_path, want in (("default", pmc_d, st_d, ("k_voxel_nn_kd", "k_pairs", "k_gen_lookup", "k_gen_emit",
                                                                    "k_group_cull", "k_score_items")),
                                          ("TM_VOXEL_BRUTEFORCE=1", pmc_b, st_b, ("k_voxel_nn",))):
        pmc = json.load(open(pmc_dir + "/summary.json"))
        st = stats(st_path)
        for kname, c in pmc.items():
            short = kname.split("::")[-1].strip()
            base = short.split("<")[0]
            if base not in want:
                continue
            key = next((k for k in st if k.strip().endswith(short) or short in k), None)
            if key is None:
                continue
            calls, ms = st[key]
            g = lambda n: c.get(n, {}).get("mean", 0.0)
            valu = g("SQ_INSTS_VALU") * 64.0
            rd = 128.0 * g("TCC_EA0_RDREQ_128B_sum") + 64.0 * g("TCC_EA0_RDREQ_64B_sum") + 32.0 * g("TCC_EA0_RDREQ_32B_sum")
            rows.append(dict(kernel=short, mode=label, launches=calls, ms=ms, valu_glaneops=valu / (ms * 1e-3) / 1e9,
                             lds_insts=g("SQ_INSTS_LDS"), lds_GB=g("SQ_INSTS_LDS") * 256.0 / 1e9,
                             lds_GBs=g("SQ_INSTS_LDS") * 256.0 / (ms * 1e-3) / 1e9,
                             bank_conflict_cycles=g("SQ_LDS_BANK_CONFLICT"), vmem_rd=g("SQ_INSTS_VMEM_RD"),
                             fabric_GB=rd / 1e9, fabric_GBs=rd / (ms * 1e-3) / 1e9,
                             wait_frac=(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")) if g("SQ_WAVE_CYCLES") else 0.0))
    print("| kernel | mode | launches | ms | VALU G lane-op/s | LDS instr | LDS GB (<=) | LDS GB/s | bank-conflict cyc | VMEM rd instr | fabric GB | fabric GB/s | wait |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %s | %d | %.3f | %.0f | %.3g | %.2f | %.0f | %.3g | %.3g | %.3f | %.0f | %.2f |"
              % (r["kernel"], r["mode"], r["launches"], r["ms"], r["valu_glaneops"], r["lds_insts"], r["lds_GB"], r["lds_GBs"],
                 r["bank_conflict_cycles"], r["vmem_rd"], r["fabric_GB"], r["fabric_GBs"], r["wait_frac"]))
    json.dump(rows, open(pmc_d + "/kernel_table.json", "w"), indent=1)


if __name__ == "__main__":
    main()
