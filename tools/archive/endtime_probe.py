#!/usr/bin/env python
"""
Launch tail of k_synth_ola_pair: start / end clock of every wave of one launch (a -DMPX_PROBE_ENDTIME build).

    python tools/ab_bench.py --prepare endp:-DMPX_PROBE_ENDTIME        # here
    python tools/endtime_probe.py [variant]                            # on the GPU box

Prints the distribution of the waves' end times relative to the first start, per XCD (block b runs on XCD b % 8), and
what the launch would take if every wave ended at the mean.
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

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "endp"
em = ab_bench.load(name)
eng = em.Engine()
utts = bench.make_batch(0)
aplan = em.LosslessAnalysisPlan(eng, utts)
splan = em.LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
feats = aplan.run()
strips, pcm = eng.empty((splan.strip_floats,)), eng.empty((splan.total_out,))
repeat = "--repeat" in sys.argv     # probe a synthesis launch that follows another synthesis (no freshly written features)
for rep in range(4):
    aplan.run(out=feats)
    splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm)
    if repeat:
        splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm)
torch.cuda.synchronize()
waves_per_block = int(os.environ.get("WPB", 12))
n = 256 * waves_per_block
buf = (ctypes.c_ulonglong * (4 * n))()
assert eng.lib.mpx_probe_endtimes(buf, 4 * n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.float64)
t0, t1, fr = a[:, 0], a[:, 1], a[:, 2]
ok = fr > 0
base = t0[ok].min()
s, e = (t0[ok] - base) / 100.0, (t1[ok] - base) / 100.0     # microseconds (100 MHz clock)
print("waves %d (with frames: %d), frames per wave %.1f .. %.1f" % (n, ok.sum(), fr[ok].min(), fr[ok].max()))
print("start  us: min %.1f  median %.1f  max %.1f" % (s.min(), np.median(s), s.max()))
print("end    us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  p99 %.1f  max %.1f" % (
    e.min(), np.percentile(e, 10), np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max()))
cyc = a[:, 3][ok]
print("shader clock during the launch: median %.0f MHz (s_memtime ticks / wall time per wave); cycles per frame and wave: median %.0f" % (np.median(cyc / (e - s)), np.median(cyc / fr[ok])))
print("busy   us: mean %.1f  (launch = max end %.1f: %.0f %% of the waves' mean)" % ((e - s).mean(), e.max(), 100 * e.max() / (e - s).mean()))
blk = np.arange(n)[ok] // waves_per_block
for x in range(8):
    m = (blk % 8) == x
    print("  XCD %d: end median %.1f  max %.1f   us per frame (median) %.2f" % (x, np.median(e[m]), e[m].max(), np.median((e[m] - s[m]) / fr[ok][m])))

# by position in the batch: block b works on the frames [b, b + 1) * F / 256 (slots are dealt in frame order)
nb = 16
print("end time (median us) and us per frame by sixteenth of the batch (first frames ... last frames):")
print("  " + " ".join("%6.1f" % np.median(e[(blk * nb) // 256 == i]) for i in range(nb)))
print("  " + " ".join("%6.2f" % np.median(((e - s) / fr[ok])[(blk * nb) // 256 == i]) for i in range(nb)))
widx = np.arange(n)[ok] % waves_per_block
print("end time (median / max us) and frames by wave index within the workgroup:")
for w in range(waves_per_block):
    m = widx == w
    print("  wave %2d: end %6.1f / %6.1f   us per frame %.2f   frames %.1f" % (w, np.median(e[m]), e[m].max(), np.median(((e - s) / fr[ok])[m]), fr[ok][m].mean()))
