#!/usr/bin/env python
"""Where the numpy array API's time goes: cProfile of mp.analysis_lossless_batch on 16 utterances (float64 out)."""
import cProfile
import os
import pstats
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from magphase_amd import magphase as mp  # noqa: E402

utts = bench.make_batch(0)[:16]
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(3):
        f = mp.analysis_lossless_batch(utts, copy=False)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        f = mp.analysis_lossless_batch(utts, copy=False)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(16)
