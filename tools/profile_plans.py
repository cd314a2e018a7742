#!/usr/bin/env python
"""Per-line wall time of the plan constructors inside the corpus workload (tools/lineprof.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

import corpus_workload as cw
import lineprof
from magphase_amd import engine as em
from magphase_amd import magphase as mp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 640
for name, mixed, fn, watch in (("extraction", False, cw.run_extraction,
                                (em.CompressedAnalysisPlan.__init__, em.LosslessAnalysisPlan.__init__, mp.analysis_compressed_batch)),
                               ("generation", True, cw.run_generation,
                                (em.CompressedSynthesisPlan.__init__, mp.synthesis_from_compressed_batch))):
    dur, fs = cw.corpus_spec(n, mixed)
    mine = np.arange(n)
    fn(0, mine, dur, fs)
    lineprof.watch(*watch)
    r = fn(0, mine, dur, fs)
    lineprof.report()
    lineprof._codes.clear(); lineprof._acc.clear(); lineprof._hits.clear()
    print("%s: %.4f s traced" % (name, r["seconds"]))
