#!/usr/bin/env python
"""
Per-phase time of k_synth_ola_pair from a -DMPX_PROBE_TIMING build (s_memtime stamps summed over all waves).

    python tools/ab_bench.py --prepare probe:-DMPX_PROBE_TIMING        # here
    python tools/probe_timing.py                                       # on the GPU box
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ab_bench  # noqa: E402
import bench  # noqa: E402

PHASES = ["cursor", "feature loads (issue + land)", "convert + merge", "inverse FFT", "ticket wait", "ring flush",
          "ring add", "run end + ticket pass"]


def main():
    torch.cuda.set_device(0)
    em = ab_bench.load("probe")
    eng = em.Engine()
    utts = bench.make_batch(0)
    aplan = em.LosslessAnalysisPlan(eng, utts)
    splan = em.LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
    feats = aplan.run()
    buf = (ctypes.c_ulonglong * 16)()
    eng.lib.mpx_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for _ in range(3):
        splan.run(*feats)
    eng.lib.mpx_probe_read(buf, 1)
    reps = 10
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        splan.run(*feats)
    ev[1].record()
    torch.cuda.synchronize()
    eng.lib.mpx_probe_read(buf, 1)
    acc = np.array(list(buf)[:8], dtype=np.float64) / reps
    F = aplan.total_frames
    print("synthesis (probe build): %.4f ms per launch pair, %d frames, %d runs" % (ev[0].elapsed_time(ev[1]) / reps, F, splan.n_runs))
    print("%-34s %12s %8s" % ("phase", "cyc / frame", "share"))
    for name, a in zip(PHASES, acc):
        print("%-34s %12.0f %7.1f%%" % (name, a / F, 100 * a / acc.sum()))
    print("%-34s %12.0f" % ("total (s_memtime ticks per frame)", acc.sum() / F))


if __name__ == "__main__":
    main()
