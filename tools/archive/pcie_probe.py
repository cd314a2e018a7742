#!/usr/bin/env python
"""Pinned / pageable host <-> device copy rates of this box (what bounds the numpy array API).
    python tools/pcie_probe.py"""
import time

import numpy as np
import torch

n = 1 << 28                      # 1 GiB of float32
dev = torch.empty(n, dtype=torch.float32, device="cuda")
pin = torch.empty(n, dtype=torch.float32).pin_memory()
page = torch.empty(n, dtype=torch.float32)


def rate(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return 4.0 * n / sorted(ts)[len(ts) // 2] / 1e9


print("D2H pinned   %.1f GB/s" % rate(lambda: pin.copy_(dev, non_blocking=True)))
print("H2D pinned   %.1f GB/s" % rate(lambda: dev.copy_(pin, non_blocking=True)))
print("D2H pageable %.1f GB/s" % rate(lambda: page.copy_(dev)))
print("H2D pageable %.1f GB/s" % rate(lambda: dev.copy_(page)))
a = np.empty(n // 4, dtype=np.float32)
b = np.empty(n // 4, dtype=np.float64)
t = time.perf_counter(); b[:] = a; dt = time.perf_counter() - t
print("numpy float32 -> float64, one thread: %.1f GB/s of float32 read" % (a.nbytes / dt / 1e9))
