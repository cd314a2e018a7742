#!/usr/bin/env python
"""numpy's MT19937 stream continued on the device (Engine.numpy_global_uniform): ms per draw of 7.7 M / 30 M samples (the
noise of 32 / 128 utterances of 5 s @ 48 kHz), against np.random.uniform on the host."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magphase_amd.engine import get_engine
e = get_engine()
for n in (7_700_000, 30_000_000):
    for rep in range(4):
        np.random.seed(1)
        torch.cuda.synchronize(); t = time.perf_counter()
        x = e.numpy_global_uniform(n)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(n, "samples: %.2f ms" % (dt * 1e3))
    t = time.perf_counter(); np.random.uniform(-1, 1, n); print("  numpy host: %.1f ms" % ((time.perf_counter() - t) * 1e3))
