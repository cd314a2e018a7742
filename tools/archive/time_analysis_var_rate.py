#!/usr/bin/env python
"""Kernel times of the variable-frame-rate low-dimensional analysis (the extraction script's default) on the bench batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from magphase_amd.engine import CompressedAnalysisPlan, get_engine  # noqa: E402

eng = get_engine()
utts = bench.make_batch(0)
plan = CompressedAnalysisPlan(eng, utts, mag_dim=60, phase_dim=45, b_const_rate=False)
feats = tuple(eng.empty_feats(plan.lossless.total_frames, plan.fft_len // 2 + 1) for _ in range(3))
for _ in range(3):
    out = plan.run(feats=feats)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
    out = plan.run(feats=feats, out=out)
ev[1].record()
torch.cuda.synchronize()
print("variable-rate analysis_compressed, %d frames: %.3f ms per batch" % (plan.total_out_frames, ev[0].elapsed_time(ev[1]) / 20))
