#!/usr/bin/env python3
"""Instruction mix of one kernel by STAGE, from the device assembly with line tables.

  hipcc <build flags> --cuda-device-only -S -gline-tables-only triplet_match_amd/csrc/tm_scene.hip -o /tmp/tm_scene.s
  python tools/isa_mix.py /tmp/tm_scene.s ILb1ELi4ELi16ELb0ELb1ELb1ELb0E > profiles/rNN/isa_mix_k_score_items.md

Every instruction of the kernel is attributed to the source line of its innermost `.loc` (inlined helpers count for
the stage that calls them: the stage table below maps file:line ranges to stages) and classified VALU / SALU / LDS /
VMEM / other. The counts are STATIC; the dynamic weight of a stage follows from how often its code runs: the stages of
the evaluation loop once per iteration of four chains (256 point evaluations), the others once per item."""
import collections
import re
import sys

path, want = sys.argv[1], sys.argv[2]
STAGES = [  # (file suffix, first line, last line, stage, per)
    ("tm_scene.hip", 626, 674, "hand-out / set-up", "item"),
    ("tm_scene.hip", 675, 713, "staging (chunk + poses -> LDS)", "item"),
    ("tm_scene.hip", 714, 757, "quarter-tile probes", "item"),
    ("tm_scene.hip", 758, 788, "entry list", "item"),
    ("tm_scene.hip", 789, 800, "entry decode", "iteration"),
    ("tm_scene.hip", 801, 813, "transform + in-grid test", "iteration"),
    ("tm_scene.hip", 814, 820, "voxel address + brick code", "iteration"),
    ("tm_scene.hip", 821, 832, "near test + voxel gather", "iteration"),
    ("tm_scene.hip", 833, 857, "model gather + distance", "iteration"),
    ("tm_scene.hip", 858, 904, "orientation + reduction", "iteration"),
    ("tm_scene.hip", 905, 919, "flush", "item"),
    ("tm_scene.hip", 920, 941, "stats epilogue", "kernel"),
]
HELPERS = {  # inlined helpers by function-ish line ranges
    ("tm_scene.hip", 575, 581): ("orientation + reduction", "iteration"),   # row16_sum
    ("tm_scene.hip", 589, 596): ("quarter-tile probes", "item"),            # lds_pose
    ("tm_scene.hip", 597, 601): ("transform + in-grid test", "iteration"),  # lds_xform_point
    ("tm_scene.hip", 602, 606): ("orientation + reduction", "iteration"),   # lds_xform_dir
    ("tm_scene.hip", 611, 615): ("gather addressing (ld_off32)", "iteration"),
    ("tm_scene.hip", 100, 160): ("quarter-tile probes", "item"),            # tile_probe / tile_decide
}


def classify(op):
    if op.startswith(("v_", "V_")):
        return "VALU"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_branch", "s_cbranch", "s_sleep", "s_setprio")):
            return "ctl"
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    return "other"


files = {}
inside = False
cur = ("?", 0)
mix = collections.defaultdict(collections.Counter)
per = {}
inherited = collections.Counter()
last_stage = None
total = collections.Counter()
for ln in open(path):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))
        continue
    if not inside:
        if s.startswith("_ZN3tmx13k_score_items" + want) and s.split(":")[0].endswith("group_argsE"):
            inside = True
        continue
    if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
        break
    if s.startswith(".loc"):
        p = s.split()
        cur = (files.get(int(p[1]), "?"), int(p[2]))
        continue
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
        continue
    op = s.split()[0]
    cls = classify(op)
    stage = None
    f, l = cur
    for (sf, a, b), (st, pr) in HELPERS.items():
        if f.endswith(sf) and a <= l <= b:
            stage, per[st] = st, pr
    if stage is None:
        for sf, a, b, st, pr in STAGES:
            if f.endswith(sf) and a <= l <= b:
                stage, per[st] = st, pr
    if stage is None:
        # a helper inlined from another file (tm_math.h: xform / dot / brick_of; warp shuffles; atomics): `.loc` names
        # the helper's own line, not the caller's, so the instruction goes to the stage of the last attributed
        # instruction before it in the stream (the scheduler keeps a chain's stages together: sched_barrier per chain)
        base = f.split("/")[-1]
        if base == "tm_score.h":
            stage = "quarter-tile probes"
            per.setdefault(stage, "item")
        elif last_stage is not None:
            stage = last_stage
            inherited[base] += 1
        else:
            stage = "other (%s)" % base
            per.setdefault(stage, "?")
    last_stage = stage
    mix[stage][cls] += 1
    total[cls] += 1
    if op.startswith("v_pk_"):
        mix[stage]["(v_pk)"] += 1
    if op.startswith(("v_cvt", "v_mul_f64", "v_add_f64", "v_fma_f64", "v_trunc_f64", "v_floor_f64", "v_ldexp_f64")):
        mix[stage]["(f64/cvt)"] += 1
    if "dpp" in s:
        mix[stage]["(dpp)"] += 1

cols = ["VALU", "SALU", "LDS", "VMEM", "ctl", "other", "(v_pk)", "(f64/cvt)", "(dpp)"]
print("| stage | runs per | " + " | ".join(cols) + " |")
print("|---|---|" + "---|" * len(cols))
order = [st for _, _, _, st, _ in STAGES] + sorted(set(mix) - {st for _, _, _, st, _ in STAGES})
seen = set()
for st in order:
    if st in seen or st not in mix:
        continue
    seen.add(st)
    print("| %s | %s | " % (st, per.get(st, "?")) + " | ".join(str(mix[st][c]) for c in cols) + " |")
print("| **total** | | " + " | ".join(str(total[c]) if c in total else "" for c in cols) + " |")
it = sum(mix[st]["VALU"] for st in mix if per.get(st) == "iteration")
print("\nVALU per iteration of four chains (256 evaluations = 4 wave-level chains): %d -> %.1f per chain" % (it, it / 4.0))
print("VALU per item outside the loop: %d" % sum(mix[st]["VALU"] for st in mix if per.get(st) == "item"))
print("instructions of inlined helpers attributed to the stage around them: %s" % dict(inherited))
