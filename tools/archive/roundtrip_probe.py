#!/usr/bin/env python
"""configs[1] (64 x 5 s @48 kHz, lossless analysis -> synthesis): the two-launch step against the one-launch round trip
(mpx_roundtrip_lossless_ola), interleaved, with board power.   python tools/roundtrip_probe.py [n_utts]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from magphase_amd import synthetic as syn  # noqa: E402
from magphase_amd.engine import (LosslessAnalysisPlan, LosslessRoundTripPlan, LosslessSynthesisPlan,  # noqa: E402
                                 get_engine)

n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = get_engine()
utts = [(lambda r: (r[0], 48000, r[1], r[2]))(syn.make_utterance(u, dur_s=5.0, fs=48000)) for u in range(n_utts)]
pa = LosslessAnalysisPlan(eng, utts)
ps = LosslessSynthesisPlan(eng, pa.v_f0, pa.fs, pa.fft_len)
rt = LosslessRoundTripPlan(eng, utts)
H = pa.fft_len // 2 + 1
feats = tuple(eng.empty_feats(pa.total_frames, H) for _ in range(3))
strips = eng.empty((max(ps.strip_floats, rt.synthesis.strip_floats, 1),))
out = eng.empty((ps.total_out,))


def two():
    pa.run(out=feats)
    ps.run(*feats, strips=strips, out=out)


def one():
    rt.run(feats=feats, strips=strips, out=out)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {"frames": pa.total_frames, "two_launch_ms": [], "one_launch_ms": []}
for _ in range(4):
    res["two_launch_ms"].append(round(timeit(two), 4))
    res["one_launch_ms"].append(round(timeit(one), 4))
try:
    hw = B._hwmon_dir(torch, 0)
    for name, fn in (("two_launch", two), ("one_launch", one)):
        smp = B.PowerSampler(hw)
        t_end = time.time() + 2.5
        while time.time() < t_end:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        smp.stop()
        res[name + "_watts"] = round(smp.mean_w(1.0), 1)
except Exception as ex:  # noqa: BLE001
    res["power_error"] = repr(ex)
a = feats[0].clone()
one()
torch.cuda.synchronize()
y1 = out.clone()
two()
torch.cuda.synchronize()
res["pcm_max_abs_diff"] = float((y1 - out).abs().max())
res["pcm_peak"] = float(out.abs().max())
res["mag_max_rel_diff"] = float(((a - feats[0]).abs().max(dim=1).values / feats[0].abs().max(dim=1).values.clamp_min(1e-30)).max())
print(json.dumps(res))
