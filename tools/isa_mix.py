#!/usr/bin/env python3
"""Instruction mix of one kernel by STAGE, from the device assembly with line tables.

  hipcc <build flags> --cuda-device-only -S -gline-tables-only triplet_match_amd/csrc/tm_scene.hip -o /tmp/tm_scene.s
  python tools/isa_mix.py /tmp/tm_scene.s ILb1ELi4ELi16ELb0ELb1ELb1ELb0E > profiles/rNN/isa_mix_k_score_items.md
(run from the repo root: the stage boundaries are looked up by anchor text in the source the assembly was made from)

Every instruction of the kernel is attributed to the source line of its innermost `.loc` (inlined helpers count for
the stage that calls them: the stage table below maps file:line ranges to stages) and classified VALU / SALU / LDS /
VMEM / other. The counts are STATIC; the dynamic weight of a stage follows from how often its code runs: the stages of
the evaluation loop once per iteration of four chains (256 point evaluations), the others once per item."""
import collections
import re
import sys

path, want = sys.argv[1], sys.argv[2]  # [3] = the source file the anchors are looked up in
SRC = sys.argv[3] if len(sys.argv) > 3 else "triplet_match_amd/csrc/tm_scene.hip"
_src = open(SRC).read().split("\n")


def _line(text, after=0):
    """1-based number of the first source line containing `text` at or after line `after`"""
    for i in range(after, len(_src)):
        if text in _src[i]:
            return i + 1
    raise SystemExit("anchor not found in %s: %r" % (SRC, text))


# stage boundaries are found by anchor text in the source, so the table survives edits of the file
_k = _line("void k_score_items(score_args a, cull_args ca, group_args ga)")
_stage = _line("/* stage the chunk", _k)
_probe = _line("/* Which (hypothesis, quarter-tile) pairs are evaluated", _k)
_list = _line("if (n == 0u || TM_ABLATE_EQ(a, 10)) continue;", _k)
_turn = _line("auto turn = [&]", _k)
_xf = _line("const float4* pp = poses + 3u * gk[k];", _turn)
_addr = _line("uint32_t ix = in[k] ?", _turn)
_near = _line("near[k] = in[k] & (code[k] <= a.code_max);", _turn)
_model = _line("/* the model position and, for the near points", _turn)
_orient = _line("/* orientation term (:441-488)", _turn)
_flush = _line("/* flush the item's sums", _turn)
_stats = _line("if (!STATS) return;", _flush)
_end = _line("using items_kernel_t", _stats)
STAGES = [  # (file suffix, first line, last line, stage, per)
    ("tm_scene.hip", _k, _stage - 1, "hand-out / set-up", "item"),
    ("tm_scene.hip", _stage, _probe - 1, "staging (chunk + poses -> LDS)", "item"),
    ("tm_scene.hip", _probe, _list - 1, "two-level probe + entry list", "item"),
    ("tm_scene.hip", _list, _turn - 1, "loop set-up", "item"),
    ("tm_scene.hip", _turn, _xf - 3, "entry decode", "iteration"),
    ("tm_scene.hip", _xf - 2, _addr - 1, "transform + in-grid test", "iteration"),
    ("tm_scene.hip", _addr, _near - 3, "voxel address + brick code", "iteration"),
    ("tm_scene.hip", _near - 2, _model - 1, "near test + voxel gather", "iteration"),
    ("tm_scene.hip", _model, _orient - 1, "model + orientation records, distance", "iteration"),
    ("tm_scene.hip", _orient, _flush - 1, "orientation term + LDS atomic", "iteration"),
    ("tm_scene.hip", _flush, _stats - 1, "flush", "item"),
    ("tm_scene.hip", _stats, _end - 1, "stats epilogue", "kernel"),
]
_h = lambda t: _line(t)
HELPERS = {  # inlined helpers by their own line ranges
    ("tm_scene.hip", _h("pose12 lds_pose(const float4* p)"), _h("f3 lds_xform_point(") - 1): ("two-level probe + entry list", "item"),
    ("tm_scene.hip", _h("f3 lds_xform_point("), _h("f3 lds_xform_dir(") - 1): ("transform + in-grid test", "iteration"),
    ("tm_scene.hip", _h("f3 lds_xform_dir("), _h("T ld_off32(") - 8): ("orientation term + LDS atomic", "iteration"),
    ("tm_scene.hip", _h("T ld_off32(") - 1, _h("T ld_off32(") + 4): ("gather addressing (ld_off32)", "iteration"),
    ("tm_scene.hip", _h("tile_probe_t tile_probe("), _h("bool tile_decide(") + 4): ("two-level probe + entry list", "item"),
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
            stage = "two-level probe + entry list"
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
