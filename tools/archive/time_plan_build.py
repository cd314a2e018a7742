#!/usr/bin/env python
"""Host time of building the analysis + synthesis plans of the bench batch (64 x 5 s), first and steady state."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from magphase_amd.engine import LosslessAnalysisPlan, LosslessSynthesisPlan, get_engine  # noqa: E402

eng = get_engine()
utts = bench.make_batch(0)
for i in range(4):
    t = time.time()
    ap = LosslessAnalysisPlan(eng, utts)
    torch.cuda.synchronize()
    t1 = time.time() - t
    t = time.time()
    sp = LosslessSynthesisPlan(eng, ap.v_f0, ap.fs, ap.fft_len)
    torch.cuda.synchronize()
    print("plan build %d: analysis %.1f ms, synthesis %.1f ms" % (i, t1 * 1e3, (time.time() - t) * 1e3))
