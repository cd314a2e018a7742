#!/usr/bin/env python
"""Per-kernel device time of the generation loop of tools/corpus_workload.py WITHOUT a profiler: HIP events recorded at the
marks of CompressedSynthesisPlan.run (the same ones bench.py's configs2 block uses), summed over one timed pass, and event pairs
on the compute stream around the rest of a launch (plan construction with its upload and noise draw, output high-pass, 16-bit
conversion), with the host's wall time beside them and the planner thread's time per launch.  Round 6 found the generation
loop waiting 13-19 ms of a 45 ms pass for the coefficient upload this way: three staging slots for four launches in flight.

    python tools/corpus_marks_probe.py [n_utts]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import corpus_workload as cw
from magphase_amd import engine as em

marks = []
_run = em.CompressedSynthesisPlan.run


def run(self, out=None, keep=False, mark=None):
    ev = []

    def m(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.append((name, e))
    r = _run(self, out=out, keep=keep, mark=m)
    m("tail")       # (high-pass, peak, 16-bit conversion follow outside run(): see the "rest" line)
    marks.append(ev)
    return r


em.CompressedSynthesisPlan.run = run

# the rest of a launch: plan construction (uploads, the noise draw), output high-pass, 16-bit conversion + hand-over -- device
# time by event pairs on the compute stream, host time by the wall clock
sections = {}
import time


def bracket(obj, name, label):
    fn = getattr(obj, name)

    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        dt = time.perf_counter() - t0
        e1.record()
        sections.setdefault(label, []).append((e0, e1, dt))
        return r
    setattr(obj, name, f)


_init = em.CompressedSynthesisPlan.__init__


def init(self, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    _init(self, *a, **k)
    dt = time.perf_counter() - t0
    e1.record()
    sections.setdefault("plan_init", []).append((e0, e1, dt))


em.CompressedSynthesisPlan.__init__ = init
bracket(em.Engine, "output_hpf", "output_hpf")
bracket(em.Engine, "output_pcm16", "output_pcm16")
bracket(em.HostTicket, "wait", "ticket_wait")
_prep = em.Engine.prepare_synthesis
prep_times = []


def prep(self, *a, **k):
    t0 = time.perf_counter()
    r = _prep(self, *a, **k)
    prep_times.append((t0, time.perf_counter()))
    return r


em.Engine.prepare_synthesis = prep
_acq = em.Engine._slot_acquire
acq_times = []


def acq(self, *a, **k):
    t0 = time.perf_counter()
    r = _acq(self, *a, **k)
    acq_times.append(time.perf_counter() - t0)
    return r


em.Engine._slot_acquire = acq
bracket(em.Engine, "_slot_upload", "  slot_upload")
bracket(em.Engine, "numpy_global_uniform", "  noise_draw")
bracket(em.Engine, "to_device_packed", "  to_device_packed")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
dur, fs = cw.corpus_spec(n, True)
mine = np.arange(n)
cw.run_generation(0, mine, dur, fs)
for rep in range(3):
    del marks[:]
    sections.clear()
    del prep_times[:]
    del acq_times[:]
    r = cw.run_generation(0, mine, dur, fs)
    torch.cuda.synchronize()
    n_launch = len(cw._batches(list(mine), cw.BATCH_GEN)) * 2
    acc = {}
    for ev in marks[-n_launch:]:
        for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
            acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
    first = marks[-n_launch][0][1]
    last = marks[-1][-1][1]
    print("pass %d: %.1f ms wall (%.0f x real time); first mark -> last mark %.1f ms; inside run(): %s = %.1f ms"
          % (rep, r["seconds"] * 1e3, r["audio_s"] / r["seconds"], first.elapsed_time(last),
             "  ".join("%s %.2f" % (k, v) for k, v in acc.items()), sum(acc.values())), flush=True)
    pt = prep_times[-n_launch:]
    print("        prepare_synthesis on the planner thread: %.2f ms over %d calls (%.2f ms each), of which waiting for a staging "
          "slot %.2f ms" % (1e3 * sum(b - a for a, b in pt), len(pt), 1e3 * sum(b - a for a, b in pt) / max(1, len(pt)),
                            1e3 * sum(acq_times[-n_launch:])), flush=True)
    for k, v in sections.items():
        v = v[-n_launch:] if k != "ticket_wait" else v
        print("        %-14s device (event pair) %.2f ms, host %.2f ms over %d calls"
              % (k, sum(a.elapsed_time(b) for a, b, _ in v), 1e3 * sum(d for _a, _b, d in v), len(v)), flush=True)
