#!/usr/bin/env python
"""Instruction histogram of one kernel in a hipcc -S listing (whole function body, by class).

    python tools/asm_hist.py file.s k_synth_ola_pairILi32
"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    txt = open(path).read().split("\n")
    start = None
    for i, l in enumerate(txt):
        if l.startswith("_ZN") and key in l.split(":")[0] and ":" in l:
            start = i
            break
    assert start is not None, "kernel not found"
    hist = collections.Counter()
    for l in txt[start + 1:]:
        if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
            break
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if not m:
            continue
        hist[m.group(1)] += 1
    cls = collections.Counter()
    for k, v in hist.items():
        if k.startswith("v_mfma"):
            c = "mfma"
        elif k.startswith("v_"):
            c = "valu"
        elif k.startswith("ds_"):
            c = "lds"
        elif k.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c = "vmem"
        elif k.startswith("s_waitcnt"):
            c = "waitcnt"
        elif k.startswith("s_"):
            c = "salu"
        else:
            c = "other"
        cls[c] += v
    print(dict(cls), "total", sum(cls.values()))
    for k, v in hist.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
        print("%6d %s" % (v, k))


if __name__ == "__main__":
    main()
