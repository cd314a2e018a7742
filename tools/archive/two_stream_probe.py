#!/usr/bin/env python
"""Lossless step (analysis -> synthesis -> fix-up) with consecutive batches alternating between TWO HIP streams and two
sets of feature / output buffers: does the next batch's analysis fill the synthesis launch's tail?
    python tools/two_stream_probe.py"""
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


def buffers():
    return (tuple(eng.empty_feats(F, H) for _ in range(3)), eng.empty((max(splan.strip_floats, 1),)), eng.empty((splan.total_out,)))


def run(n_streams, steps=200, warm=20):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    bufs = [buffers() for _ in range(n_streams)]

    def step(i):
        k = i % n_streams
        feats, strips, pcm = bufs[k]
        with torch.cuda.stream(streams[k]):
            aplan.run(out=feats)
            splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm)

    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(3):
    print("1 stream %.4f ms   2 streams %.4f ms   3 streams %.4f ms" % (run(1), run(2), run(3)), flush=True)
