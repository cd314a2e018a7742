#!/usr/bin/env python
"""Where should the lossless feature matrices live?  Times the bench step (analysis -> fused synthesis) with the three
[F x H] matrices in ordinary device memory (torch's allocator) and in memory from hipExtMallocWithFlags (fine-grained /
uncached): the synthesis kernel reads what the analysis just wrote, and 256 MB of it sit dirty in the Infinity Cache
(DESIGN.md 3.5)."""
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from magphase_amd import engine as em  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipExtMallocWithFlags.restype = ctypes.c_int


class _Raw:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2}


def alloc(rows, cols, flags):
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), rows * cols * 4, flags)
    if rc != 0:
        raise RuntimeError("hipExtMallocWithFlags(%d) -> %d" % (flags, rc))
    return torch.as_tensor(_Raw(p.value, (rows, cols)), device="cuda")


torch.cuda.set_device(0)
utts = bench.make_batch(0)
eng = em.Engine()
aplan = em.LosslessAnalysisPlan(eng, utts)
splan = em.LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
H, F = aplan.fft_len // 2 + 1, aplan.total_frames
strips = eng.empty((max(splan.strip_floats, 1),))
pcm = eng.empty((splan.total_out,))
variants = {"default": tuple(eng.empty_feats(F, H) for _ in range(3))}
for name, flags in (("finegrained", 1), ("uncached", 3)):
    try:
        variants[name] = tuple(alloc(F, H, flags) for _ in range(3))
    except Exception as e:  # noqa: BLE001
        print(name, "not available:", e)
ref = None
times = {n: ([], []) for n in variants}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for r in range(20):
    for name, feats in variants.items():
        ev[0].record()
        aplan.run(out=feats)
        ev[1].record()
        splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm)
        ev[2].record()
        torch.cuda.synchronize()
        if r == 0:
            if ref is None:
                ref = pcm.clone()
            assert torch.equal(pcm, ref), name
        if r >= 2:
            times[name][0].append(ev[0].elapsed_time(ev[1]))
            times[name][1].append(ev[1].elapsed_time(ev[2]))
for name, (a, s) in times.items():
    print("%-12s analysis %.4f  synthesis+fixup %.4f  step %.4f ms" % (name, statistics.median(a), statistics.median(s),
                                                                      statistics.median(a) + statistics.median(s)))
