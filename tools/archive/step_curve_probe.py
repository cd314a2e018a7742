#!/usr/bin/env python
"""Per-step kernel times of the lossless step from an idle GPU (HIP events, no host sync inside the loop), after different
preconditioning: nothing / 0.5 s idle / 0.3 s of back-to-back streaming-copy kernels / 40 steps synchronised one by one.
What it showed: after any idle period the analysis launch runs 0.31 -> 0.41 (steps 5-8) -> 0.30 ms
(from step ~35 on): a power-management transient of about 30 ms of busy time.
    python tools/step_curve_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from magphase_amd import _lib  # noqa: E402
from magphase_amd.engine import LosslessAnalysisPlan, LosslessSynthesisPlan, get_engine  # noqa: E402

eng = get_engine()
utts = bench.make_batch(0)
aplan = LosslessAnalysisPlan(eng, utts)
splan = LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
H, F = aplan.fft_len // 2 + 1, aplan.total_frames
f_, s_, p_ = (tuple(eng.empty_feats(F, H) for _ in range(3)), eng.empty((max(splan.strip_floats, 1),)), eng.empty((splan.total_out,)))
n = 60
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n + 1)]
npr = 1 << 28
pa, pb = eng.empty((npr,)), eng.empty((npr,))


def curve(tag):
    ev[0].record()
    for i in range(n):
        aplan.run(out=f_)
        ev[2 * i + 1].record()
        splan.run(f_[0], f_[1], f_[2], strips=s_, out=p_)
        ev[2 * i + 2].record()
    torch.cuda.synchronize()
    a = [ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(n)]
    s = [ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(n)]
    print("%-28s ana %s" % (tag, " ".join("%.3f" % x for x in a[::3])))
    print("%-28s syn %s" % ("", " ".join("%.3f" % x for x in s[::3])), flush=True)


torch.cuda.synchronize()
curve("cold")
time.sleep(0.5)
curve("after 0.5 s idle")
time.sleep(0.5)
for _ in range(700):
    _lib.check(eng.lib.mpx_bw_probe(eng.stream_ptr(), 2, pa.data_ptr(), pb.data_ptr(), npr), "probe")
curve("after 0.3 s of copy kernels")
time.sleep(0.5)
for _ in range(40):
    aplan.run(out=f_)
    splan.run(f_[0], f_[1], f_[2], strips=s_, out=p_)
    torch.cuda.synchronize()
curve("after 40 synchronised steps")
time.sleep(0.5)
for _ in range(150):
    aplan.run(out=f_)
curve("after 150 analysis launches")
time.sleep(0.5)
for _ in range(150):
    splan.run(f_[0], f_[1], f_[2], strips=s_, out=p_)
curve("after 150 synthesis launches")
