#!/usr/bin/env python
"""How the step time depends on how long the GPU has been busy (clock / power ramp) and on the number of streams, at the
driver's bench size (20 timed steps after 5): consecutive 20-step blocks from a cold start, alternating 2 and 1 streams,
then the same after 0.3 s of streaming probe kernels.
    python tools/ramp_probe.py"""
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
streams = [torch.cuda.Stream() for _ in range(2)]
bufs = [(tuple(eng.empty_feats(F, H) for _ in range(3)), eng.empty((max(splan.strip_floats, 1),)), eng.empty((splan.total_out,)))
        for _ in range(2)]


def block(n_streams, steps=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % n_streams
        f_, s_, p_ = bufs[k]
        with torch.cuda.stream(streams[k]):
            aplan.run(out=f_)
            splan.run(f_[0], f_[1], f_[2], strips=s_, out=p_)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if mode == 1:      # pre-heat with streaming kernels
    n = 1 << 28
    a, b = eng.empty((n,)), eng.empty((n,))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        _lib.check(eng.lib.mpx_bw_probe(eng.stream_ptr(), 2, a.data_ptr(), b.data_ptr(), n), "probe")
        torch.cuda.synchronize()
block(2, 5)
print("mode", mode, " ".join("%d:%.4f" % (s, block(s)) for s in (2, 1, 2, 1, 2, 1, 2, 1, 2, 1)), flush=True)
