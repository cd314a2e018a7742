#!/usr/bin/env python
"""
Where a frame's time goes in k_roundtrip_pair (the one-launch copy synthesis): s_memtime ticks per wave in 7 phases of
the frame loop (a -DMPX_PROBE_RT build).

    python tools/ab_bench.py --prepare rtp:-DMPX_PROBE_RT      # here
    python tools/roundtrip_phase_probe.py [variant]            # on the GPU box
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

name = sys.argv[1] if len(sys.argv) > 1 else "rtp"
em = ab_bench.load(name)
eng = em.Engine()
utts = bench.make_batch(0)
rt = em.LosslessRoundTripPlan(eng, utts)
feats, out = rt.run()
for _ in range(4):
    rt.run(feats=feats, out=out)
torch.cuda.synchronize()
wpb = 12
n = 256 * wpb
b1 = (ctypes.c_ulonglong * (4 * n))()
b2 = (ctypes.c_ulonglong * (8 * n))()
eng.lib.mpx_probe_roundtrip.restype = ctypes.c_int
eng.lib.mpx_probe_roundtrip.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert eng.lib.mpx_probe_roundtrip(b1, b2, n) == 0
a = np.frombuffer(b1, dtype=np.uint64).reshape(n, 4).astype(np.float64)
ph = np.frombuffer(b2, dtype=np.uint64).reshape(n, 8).astype(np.float64)[:, :7]
fr = a[:, 2]
ok = fr > 0
wall_us = (a[:, 1] - a[:, 0]) / 100.0
cyc = a[:, 3]
print("launch: max end %.1f us; shader clock median %.0f MHz; ticks per frame and wave median %.0f; frames per wave %.1f" % (
    ((a[ok, 1] - a[ok, 0].min()) / 100.0).max(), np.median(cyc[ok] / wall_us[ok]), np.median(cyc[ok] / fr[ok]), np.median(fr[ok])))
names = ["staging wait", "window+FFT+split", "feats+stores+merge", "inverse FFT", "scalars+ticket", "flush", "overlap-add"]
widx = np.arange(n) % wpb
print("%-8s %8s " % ("waves", "us/frame") + " ".join("%20s" % s for s in names) + "   (ticks per frame; share of the wave's loop)")
for cls, sel in (("0-3", widx < 4), ("4-7", (widx >= 4) & (widx < 8)), ("8-11", widx >= 8), ("all", widx >= 0)):
    m = sel & ok
    per = ph[m] / fr[m][:, None]
    tot = per.sum(axis=1)
    print("%-8s %8.2f " % (cls, np.median(wall_us[m] / fr[m])) + " ".join("%12.0f (%4.1f%%)" % (np.median(per[:, i]), 100 * np.median(per[:, i] / tot)) for i in range(7)))
end_us = (a[ok, 1] - a[ok, 0].min()) / 100.0
print("wave end times (us): min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (
    end_us.min(), np.percentile(end_us, 10), np.median(end_us), np.percentile(end_us, 90), end_us.max()))
