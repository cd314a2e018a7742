#!/usr/bin/env python
"""Which pairs of HIP streams really run side by side?  Eight streams from torch's pool; the lossless step alternating
between stream 0 and stream j (and between a few other pairs), 60 steps each, against one stream alone.
    python tools/stream_pair_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from magphase_amd.engine import LosslessAnalysisPlan, LosslessSynthesisPlan, get_engine  # noqa: E402

eng = get_engine()
utts = bench.make_batch(0)
aplan = LosslessAnalysisPlan(eng, utts)
splan = LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
H, F = aplan.fft_len // 2 + 1, aplan.total_frames
bufs = [(tuple(eng.empty_feats(F, H) for _ in range(3)), eng.empty((max(splan.strip_floats, 1),)), eng.empty((splan.total_out,)))
        for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(8)]


def block(pair, steps=60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % len(pair)
        f_, s_, p_ = bufs[k]
        with torch.cuda.stream(pair[k]):
            aplan.run(out=f_)
            splan.run(f_[0], f_[1], f_[2], strips=s_, out=p_)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


block([streams[0]], 100)
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
print("one stream      ", " ".join("%d:%.3f" % (j, block([streams[j]])) for j in range(8)))
print("pairs (0, j)    ", " ".join("%d:%.3f" % (j, block([streams[0], streams[j]])) for j in range(1, 8)))
print("pairs (j, j + 1)", " ".join("%d:%.3f" % (j, block([streams[j], streams[j + 1]])) for j in range(1, 7)))
cur = torch.cuda.current_stream()
print("pairs (null, j) ", " ".join("%d:%.3f" % (j, block([cur, streams[j]])) for j in range(0, 8)))
