#!/usr/bin/env python
"""Kernel times of the variable-frame-rate generation path (batch_waveform_generation's default) on the bench batch."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from magphase_amd import magphase as mp  # noqa: E402
from magphase_amd.engine import CompressedSynthesisPlan, get_engine  # noqa: E402

eng = get_engine()
utts = bench.make_batch(0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    res = mp.analysis_compressed_batch(utts, mag_dim=60, phase_dim=45, b_const_rate=False)
np.random.seed(0)
ppt = os.environ.get("PER_PHASE", "magphase")   # magphase | min_phase | linear
plan = CompressedSynthesisPlan(eng, [(r[0], r[1], r[2], r[3]) for r in res], 48000, b_const_rate=False, post_filter=True,
                               per_phase_type=ppt)
for _ in range(3):
    plan.run()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
    plan.run()
ev[1].record()
torch.cuda.synchronize()
print("variable-rate synthesis_from_compressed (+ post-filter, per_phase_type=%s), %d frames: %.3f ms per batch" %
      (ppt, plan.total_frames, ev[0].elapsed_time(ev[1]) / 20))
