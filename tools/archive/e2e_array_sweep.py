#!/usr/bin/env python
"""Array API (numpy in -> float64 numpy out) lossless analysis + synthesis of 16 utterances for a few native thread counts."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import json, os, sys, time, warnings
sys.path.insert(0, %r)
import bench
from magphase_amd import magphase as mp
utts = bench.make_batch(0)[:16]
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    f = mp.analysis_lossless_batch(utts, copy=False); mp.synthesis_from_lossless_batch([x[:5] for x in f][:2])
    runs = []
    for _ in range(5):
        t0 = time.perf_counter(); f = mp.analysis_lossless_batch(utts, copy=False); ta = time.perf_counter() - t0
        fin = [x[:5] for x in f]
        t0 = time.perf_counter(); mp.synthesis_from_lossless_batch(fin); ts = time.perf_counter() - t0
        runs.append((ta + ts, ta, ts))
runs.sort()
print(json.dumps({"threads": os.environ.get("MAGPHASE_IO_NATIVE_THREADS"), "median_s": runs[2], "x_realtime": 80.0 / runs[2][0]}))
''' % ROOT
for thr in ("8", "16", "32", "64"):
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, MAGPHASE_IO_NATIVE_THREADS=thr), capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
