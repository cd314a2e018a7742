#!/usr/bin/env python
"""Array API (numpy float64 in / out) of the lossless pair on 16 utterances, by the host threads and the D2H chunk size
of the pinned pipeline (MAGPHASE_IO_NATIVE_THREADS / MAGPHASE_D2H_CHUNK_MB): analysis_s, synthesis_s, frames/s."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import warnings

    import bench
    from magphase_amd import magphase as mp
    sub = bench.make_batch(0)[:16]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        feats = mp.analysis_lossless_batch(sub, copy=False)
        fin = [(f[0], f[1], f[2], f[3], f[4]) for f in feats]
        mp.synthesis_from_lossless_batch(fin[:2])
        runs = []
        for _ in range(5):
            t0 = time.perf_counter()
            feats = mp.analysis_lossless_batch(sub, copy=False)
            t_a = time.perf_counter() - t0
            fin = [(f[0], f[1], f[2], f[3], f[4]) for f in feats]
            t0 = time.perf_counter()
            mp.synthesis_from_lossless_batch(fin)
            runs.append((t_a + time.perf_counter() - t0, t_a))
    runs.sort()
    tot, t_a = runs[len(runs) // 2]
    nfr = int(sum(f[0].shape[0] for f in feats))
    print("threads %s chunk %s MB: analysis %.4f s, synthesis %.4f s, %.0f frames/s" % (
        os.environ.get("MAGPHASE_IO_NATIVE_THREADS", "auto"), os.environ.get("MAGPHASE_D2H_CHUNK_MB", "32"), t_a, tot - t_a,
        nfr / tot), flush=True)
else:
    for thr, chunk in (("", "32"), ("16", "32"), ("32", "16"), ("32", "8"), ("64", "16"), ("64", "32"), ("32", "64")):
        env = dict(os.environ, MAGPHASE_D2H_CHUNK_MB=chunk)
        env.pop("MAGPHASE_IO_NATIVE_THREADS", None)
        if thr:
            env["MAGPHASE_IO_NATIVE_THREADS"] = thr
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
