#!/usr/bin/env python
"""Device time of one draw of numpy's MT19937 stream (Engine.numpy_global_uniform, 30 M samples = the noise of 128
utterances of 5 s at 48 kHz) on an otherwise idle GPU.

    python tools/mt_ladder_probe.py [n_samples]

Round 6 used it with a temporary switch for the number of workgroups that share one jump's mask (kMtJumpSplit in
magphase_noise.hip): 16 / 8 / 4 / 2 / 1 = 0.95 / 1.06 / 1.41 / 2.36 / 4.22 ms per draw, and for the ladder's radix
(kMtRadix: 16 instead of 2 = slower inside a generation job); the constants stayed at 16 and 2 and the switch was removed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from magphase_amd.engine import get_engine

e = get_engine()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
np.random.seed(1)
for _ in range(3):
    e.numpy_global_uniform(n, defer=True)
torch.cuda.synchronize()
rng = e.copy_stream("rng")
ts = []
for _ in range(15):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(rng)
    e.numpy_global_uniform(n, defer=True)
    b.record(rng)
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
e.mt_sync()
ts.sort()
print("%d samples per draw: median %.3f ms, min %.3f ms" % (n, ts[len(ts) // 2], ts[0]))
