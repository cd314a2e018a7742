#!/usr/bin/env python
"""How busy the device is inside the corpus workload's timed loops (tools/corpus_workload.py, one rank).

Every launch of the batch API (mp.analysis_compressed_batch / mp.synthesis_from_compressed_batch) is bracketed by two HIP
events on the compute stream: the first completes when everything enqueued before the launch has finished, the second when
the launch's own work (uploads waited for, kernels, the result's hand-over to the download stream) has.  Launches are
serialised on that stream, so the sum of the brackets' durations is the time the stream was busy or had work pending from an
already-built plan; wall minus that is the time the device waited for the HOST (plan build, enqueue).  The bracket of a launch
the host enqueued late starts late, so host stalls show up as a low busy fraction.

    python tools/corpus_wait_probe.py [n_utts] [out.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import corpus_workload as cw
from magphase_amd import magphase as mp

brackets = []


def _bracket(fn):
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        brackets.append((e0, e1))
        return r
    return f


mp.analysis_compressed_batch = _bracket(mp.analysis_compressed_batch)
mp.synthesis_from_compressed_batch = _bracket(mp.synthesis_from_compressed_batch)
_clock = cw.time.perf_counter


n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
out = {}
for name, mixed, fn in (("extraction", False, cw.run_extraction), ("generation", True, cw.run_generation)):
    dur, fs = cw.corpus_spec(n, mixed)
    mine = np.arange(n)
    fn(0, mine, dur, fs)
    rows = []
    for rep in range(3):
        del brackets[:]
        r = fn(0, mine, dur, fs)
        torch.cuda.synchronize()
        # the timed loop's launches are the LAST ones of the call (warm-up launches come first): count them from the end
        n_launch = len(cw._batches(list(mine), cw.BATCH_GEN if mixed else cw.BATCH)) * (2 if mixed else 1)
        busy = sum(a.elapsed_time(b) for a, b in brackets[-n_launch:]) * 1e-3
        rows.append({"seconds": round(r["seconds"], 5), "x_realtime": round(r["audio_s"] / r["seconds"], 1),
                     "launches": n_launch, "stream_busy_s": round(busy, 5),
                     "device_busy_fraction": round(min(1.0, busy / r["seconds"]), 3)})
        print("%s pass %d: %.4f s (%.0f x real time), %d launches, compute stream busy %.4f s = %.0f %% of the pass"
              % (name, rep, r["seconds"], r["audio_s"] / r["seconds"], n_launch, busy, 100.0 * min(1.0, busy / r["seconds"])),
              flush=True)
    out[name] = rows
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fh:
        json.dump({"utterances": n, "what": __doc__.strip().split("\n\n")[1].replace("\n", " "), "passes": out}, fh, indent=1)
