#!/usr/bin/env python
"""
Where a frame's time goes in k_synth_ola_pair: s_memtime ticks per wave in 8 phases of the frame loop
(a -DMPX_PROBE_ENDTIME -DMPX_PROBE_PHASES build; the probe forces a full vmcnt(0) at the loop top, so "feature wait" is
the whole exposed memory latency).

    python tools/ab_bench.py --prepare php:-DMPX_PROBE_ENDTIME,-DMPX_PROBE_PHASES      # here
    python tools/phase_probe.py [variant] [--repeat]                                     # on the GPU box
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ab_bench  # noqa: E402
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "php"
em = ab_bench.load(name)
eng = em.Engine()
utts = bench.make_batch(0)
aplan = em.LosslessAnalysisPlan(eng, utts)
splan = em.LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
feats = aplan.run()
strips, pcm = eng.empty((splan.strip_floats,)), eng.empty((splan.total_out,))
repeat = "--repeat" in sys.argv
for rep in range(4):
    aplan.run(out=feats)
    splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm)
    if repeat:
        splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm)
torch.cuda.synchronize()
wpb = int(os.environ.get("WPB", 12))
n = 256 * wpb
b1 = (ctypes.c_ulonglong * (4 * n))()
b2 = (ctypes.c_ulonglong * (8 * n))()
assert eng.lib.mpx_probe_endtimes(b1, 4 * n) == 0
assert eng.lib.mpx_probe_phases(b2, 8 * n) == 0
a = np.frombuffer(b1, dtype=np.uint64).reshape(n, 4).astype(np.float64)
ph = np.frombuffer(b2, dtype=np.uint64).reshape(n, 8).astype(np.float64)
fr = a[:, 2]
ok = fr > 0
wall_us = (a[:, 1] - a[:, 0]) / 100.0
cyc = a[:, 3]
print("launch: max end %.1f us; shader clock median %.0f MHz; ticks per frame and wave median %.0f" % (
    ((a[ok, 1] - a[ok, 0].min()) / 100.0).max(), np.median(cyc[ok] / wall_us[ok]), np.median(cyc[ok] / fr[ok])))
names = ["feature wait", "merge", "transform", "prefetch+scalars", "ticket wait", "flush", "overlap-add", "tail"]
widx = np.arange(n) % wpb
print("%-10s %8s " % ("waves", "us/frame") + " ".join("%16s" % s for s in names) + "   (ticks per frame; share of the wave's loop)")
for cls, sel in (("0-3", widx < 4), ("4-7", (widx >= 4) & (widx < 8)), ("8-11", widx >= 8), ("all", widx >= 0)):
    m = sel & ok
    per = ph[m] / fr[m][:, None]
    tot = per.sum(axis=1)
    print("%-10s %8.2f " % (cls, np.median(wall_us[m] / fr[m])) + " ".join("%8.0f (%4.1f%%)" % (np.median(per[:, i]), 100 * np.median(per[:, i] / tot)) for i in range(8)))
