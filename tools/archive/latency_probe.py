#!/usr/bin/env python
"""Single-utterance latency of the array API (serving use case): where does the time of ONE call go?"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from magphase_amd import libutils as lu  # noqa: E402
from magphase_amd import magphase as mp  # noqa: E402

d = os.path.join(ROOT, "demos", "data_48k", "params_predicted")
m_mag = lu.read_binfile(os.path.join(d, "hvd_704.mag"), dim=60)
m_real = lu.read_binfile(os.path.join(d, "hvd_704.real"), dim=45)
m_imag = lu.read_binfile(os.path.join(d, "hvd_704.imag"), dim=45)
v_lf0 = lu.read_binfile(os.path.join(d, "hvd_704.lf0"), dim=1)


def once():
    y = mp.synthesis_from_compressed(mp.post_filter(m_mag, 48000), m_real, m_imag, v_lf0, 48000)
    return y


for _ in range(3):
    y = once()
torch.cuda.synchronize()
t = time.time()
n = 20
for _ in range(n):
    y = once()
torch.cuda.synchronize()
dt = (time.time() - t) / n
print("synthesis_from_compressed(hvd_704: %d frames, %.2f s of audio): %.2f ms per call = %.0f x real time" %
      (m_mag.shape[0], y.size / 48000.0, dt * 1e3, y.size / 48000.0 / dt))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    once()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

# ---- lossless analysis + synthesis and low-dimensional analysis of one 5 s utterance (array API)
from magphase_amd import synthetic as syn  # noqa: E402

pcm, pm, voi = syn.make_utterance(3, dur_s=5.0)
x = syn.pcm_to_float(pcm)


def lossless():
    a = mp.analysis_lossless_from_epochs(x, 48000, pm, voi)
    return mp.synthesis_from_lossless(a[0], a[1], a[2], a[3], 48000)


def lowdim():
    return mp.analysis_compressed_batch([(x, 48000, pm, voi)], mag_dim=60, phase_dim=45)[0]


for name, fn in (("lossless analysis + synthesis (numpy in, numpy out)", lossless), ("analysis_compressed", lowdim)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print("%s, 5 s utterance: %.2f ms per call" % (name, (time.time() - t) / 10 * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        fn()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(8)
