#!/usr/bin/env python
"""How long the host waits for the device inside the corpus workload's timed loops (HostTicket.wait + the final
mt_sync): wall ~ wait + host work.  A loop whose wait is small is host-bound."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

import corpus_workload as cw
from magphase_amd import engine as em

acc = {"wait": 0.0}
_w, _s = em.HostTicket.wait, em.Engine.mt_sync


def _timed(fn):
    def f(self, *a, **k):
        t = time.perf_counter()
        r = fn(self, *a, **k)
        acc["wait"] += time.perf_counter() - t
        return r
    return f


em.HostTicket.wait, em.Engine.mt_sync = _timed(_w), _timed(_s)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 640
for name, mixed, fn in (("extraction", False, cw.run_extraction), ("generation", True, cw.run_generation)):
    dur, fs = cw.corpus_spec(n, mixed)
    mine = np.arange(n)
    fn(0, mine, dur, fs)
    acc["wait"] = 0.0
    r = fn(0, mine, dur, fs)
    print("%s: %.4f s (%.0f x real time), of which the host waited for the device %.4f s"
          % (name, r["seconds"], r["audio_s"] / r["seconds"], acc["wait"]))
