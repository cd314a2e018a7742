#!/usr/bin/env python
"""
Per-phase instruction table of one kernel from a hipcc -S listing built with -DMPX_PHASE_MARKS (mpx_common.hpp: MPX_MARK):
the instructions between two markers are attributed to the first, by class.  Static counts of the listing -- a phase inside a
loop or a skipped branch executes another number of times (the window / gather rows: one block per 64 samples of the frame) --
next to SQ_INSTS_VALU per frame from the counters they say where the issue slots go.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-inline-asm -DMPX_PHASE_MARKS -S --cuda-device-only \\
          magphase_amd/csrc/magphase_comp.hip -o /tmp/comp_marks.s
    python tools/asm_phases.py /tmp/comp_marks.s k_roundtrip_pairILi32 [out.json]
"""
import collections
import json
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_permlane", "v_readlane", "v_writelane", "v_readfirstlane")) or "dpp" in op:
        return "valu_lane"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "valu_mov"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "valu_select"
    if op.startswith(("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshl", "v_lshr", "v_and", "v_or", "v_min_u32", "v_mad_u",
                      "v_mul_u", "v_mul_lo", "v_add_co", "v_addc", "v_bfe", "v_ashr", "v_mul_i", "v_add3", "v_xor", "v_sub_co",
                      "v_min_i", "v_max_i", "v_add_lshl", "v_xad", "v_bfi")):
        return "valu_int"
    if op.startswith(("v_rsq", "v_rcp", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "valu_trans"
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith("v_"):
        return "valu_float"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    txt = open(path).read().split("\n")
    start = next(i for i, l in enumerate(txt) if l.startswith("_ZN") and key in l.split(":")[0] and ":" in l)
    phases = collections.OrderedDict()
    cur = "prologue"
    for l in txt[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.search(r"; MPX_MARK (\S+)", l)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if not m:
            continue
        phases.setdefault(cur, collections.Counter())[classify(m.group(1))] += 1
    cols = ["valu_float", "valu_packed", "valu_trans", "valu_int", "valu_select", "valu_mov", "valu_lane", "salu", "lds", "vmem",
            "waitcnt"]
    print("%-24s %6s | %s" % ("phase", "VALU", " ".join("%11s" % c for c in cols)))
    tot = collections.Counter()
    out = {}
    for name, c in phases.items():
        valu = sum(v for k, v in c.items() if k.startswith("valu"))
        print("%-24s %6d | %s" % (name, valu, " ".join("%11d" % c.get(k, 0) for k in cols)))
        out[name] = dict(c, valu=valu)
        tot.update(c)
    valu = sum(v for k, v in tot.items() if k.startswith("valu"))
    print("%-24s %6d | %s" % ("TOTAL (static)", valu, " ".join("%11d" % tot.get(k, 0) for k in cols)))
    if len(sys.argv) > 3:
        json.dump({"kernel": key, "static_instructions_per_phase": out, "total": dict(tot, valu=valu)}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
