#!/usr/bin/env python
"""Device time of one draw of numpy's MT19937 stream (Engine.numpy_global_uniform, 30 M samples = the noise of 128
utterances of 5 s at 48 kHz) on an otherwise idle GPU, by the jump ladder's mask split (MAGPHASE_MT_SPLIT: workgroups that
share one jump; unset = adaptive).  One process per setting (the split is read once)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    from magphase_amd.engine import get_engine
    e = get_engine()
    n = 30_000_000
    np.random.seed(1)
    for _ in range(3):
        e.numpy_global_uniform(n, defer=True)
    torch.cuda.synchronize()
    rng = e.copy_stream("rng")
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(rng)
        e.numpy_global_uniform(n, defer=True)
        b.record(rng)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    e.mt_sync()
    ts.sort()
    print("split %s: median %.3f ms, min %.3f ms" % (os.environ.get("MAGPHASE_MT_SPLIT", "adaptive"), ts[len(ts) // 2], ts[0]))
else:
    for sp in ("", "16", "8", "4", "2", "1"):
        env = dict(os.environ)
        env.pop("MAGPHASE_MT_SPLIT", None)
        if sp:
            env["MAGPHASE_MT_SPLIT"] = sp
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
