#!/usr/bin/env python
"""The same kernels on the same data differ by +-5 % between instances in one process (tools/ab_bench.py with identical
variants).  Which allocation carries it?  One engine; alternatives of one buffer class at a time (feature matrices, PCM
input, PCM output + strips, plan descriptors), everything else shared; analysis -> fused synthesis timed interleaved."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from magphase_amd import engine as em  # noqa: E402

torch.cuda.set_device(0)
utts = bench.make_batch(0)
eng = em.Engine()
K = 4
aplans = [em.LosslessAnalysisPlan(eng, utts) for _ in range(K)]
splans = [em.LosslessSynthesisPlan(eng, aplans[0].v_f0, aplans[0].fs, aplans[0].fft_len) for _ in range(K)]
H, F = aplans[0].fft_len // 2 + 1, aplans[0].total_frames
feats = [tuple(eng.empty_feats(F, H) for _ in range(3)) for _ in range(K)]
outs = [(eng.empty((max(splans[0].strip_floats, 1),)), eng.empty((splans[0].total_out,))) for _ in range(K)]
junk = [torch.empty(int(37e6) + 1000 * i, dtype=torch.float32, device="cuda") for i in range(3)]   # shifts later allocations
sigs = [aplans[0].sig.clone() for _ in range(K)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]


def run(ap, sp, ft, out, sig):
    keep = ap.sig
    ap.sig = sig
    ev[0].record()
    ap.run(out=ft)
    ev[1].record()
    sp.run(ft[0], ft[1], ft[2], strips=out[0], out=out[1])
    ev[2].record()
    torch.cuda.synchronize()
    ap.sig = keep
    return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])


def sweep(label, configs):
    t = [([], []) for _ in configs]
    for r in range(14):
        for i, c in enumerate(configs):
            a, s = run(*c)
            if r >= 2:
                t[i][0].append(a)
                t[i][1].append(s)
    print(label)
    for i, (a, s) in enumerate(t):
        print("   alternative %d: analysis %.4f  synthesis+fixup %.4f" % (i, statistics.median(a), statistics.median(s)))


sweep("feature matrices:", [(aplans[0], splans[0], feats[i], outs[0], sigs[0]) for i in range(K)])
sweep("PCM input:", [(aplans[0], splans[0], feats[0], outs[0], sigs[i]) for i in range(K)])
sweep("PCM output + strips:", [(aplans[0], splans[0], feats[0], outs[i], sigs[0]) for i in range(K)])
sweep("analysis plan descriptors:", [(aplans[i], splans[0], feats[0], outs[0], sigs[0]) for i in range(K)])
sweep("synthesis plan descriptors:", [(aplans[0], splans[i], feats[0], outs[0], sigs[0]) for i in range(K)])
sweep("same everything, four times:", [(aplans[0], splans[0], feats[0], outs[0], sigs[0]) for i in range(K)])
