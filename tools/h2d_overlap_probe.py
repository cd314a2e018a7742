#!/usr/bin/env python
"""Does a host -> device copy on its own stream run beside the compute stream's kernels?  A page-locked buffer of 70 MB (the
coefficient rows of a 128-utterance generation launch) is copied on a second stream (a) alone, (b) while a configs[2] synthesis
step runs on the compute stream, (c) while a large matrix product runs; the copy's own duration (event pair on its stream) and
the time from its enqueue to its completion are printed.

    python tools/h2d_overlap_probe.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from magphase_amd import engine as em

eng = em.get_engine()
utts = bench.make_batch(0)
sa, ss = bench.lowdim_plans(em, eng, utts)
sa(), ss()
torch.cuda.synchronize()
nbytes = 70 << 20
pinned = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
pinned.fill_(1)
up = torch.cuda.Stream()
a = torch.randn(8192, 8192, device="cuda")


def work_synth():
    for _ in range(4):
        ss()


def work_mm():
    for _ in range(3):
        torch.mm(a, a)


for name, work in (("alone", None), ("beside 4 synthesis steps", work_synth), ("beside 3 matrix products", work_mm),
                   ("alone", None)):
    for rep in range(3):
        torch.cuda.synchronize()
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record()
        if work:
            work()
        k1.record()
        with torch.cuda.stream(up):
            c0.record(up)
            d = pinned.to("cuda", non_blocking=True)
            c1.record(up)
        torch.cuda.synchronize()
    print("%-28s compute %.2f ms | copy itself %.2f ms (%.1f GB/s) | compute start -> copy done %.2f ms"
          % (name, k0.elapsed_time(k1), c0.elapsed_time(c1), nbytes / c0.elapsed_time(c1) / 1e6, k0.elapsed_time(c1)), flush=True)
