#!/usr/bin/env python
"""Where the corpus workload's wall time goes (tools/corpus_workload.py, one rank): cProfile of the timed loops of
configs[3] extraction and configs[4] generation.    python tools/profile_corpus.py [n_utts]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

import corpus_workload as cw

n = int(sys.argv[1]) if len(sys.argv) > 1 else 640
for name, mixed, fn in (("extraction", False, cw.run_extraction), ("generation", True, cw.run_generation)):
    dur, fs = cw.corpus_spec(n, mixed)
    mine = np.arange(n)
    r = fn(0, mine, dur, fs)          # warm (tables, pools, plans' constants)
    r = fn(0, mine, dur, fs)
    print("%s: %.4f s for %d utterances (%.0f x real time), %.2f ms per 64-utterance launch"
          % (name, r["seconds"], n, r["audio_s"] / r["seconds"], 1e3 * r["seconds"] / max(1, (n + 63) // 64)), flush=True)
    pr = cProfile.Profile()
    pr.enable()
    fn(0, mine, dur, fs)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.print_callees("engine.py.*__init__")
