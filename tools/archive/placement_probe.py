#!/usr/bin/env python
"""Does the placement of the three lossless feature matrices (mag, real, imag: written / read row by row at the same
offsets by every wave) in device memory matter?  One slab, the matrices carved at base, base + S + skew, base + 2 (S +
skew) for a range of skews (S = the matrix size rounded up to 2 MB), analysis -> fused synthesis timed interleaved."""
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
aplan = em.LosslessAnalysisPlan(eng, utts)
splan = em.LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
H, F = aplan.fft_len // 2 + 1, aplan.total_frames
strips = eng.empty((max(splan.strip_floats, 1),))
pcm = eng.empty((splan.total_out,))
n = F * H
S = ((4 * n + (2 << 20) - 1) // (2 << 20)) * (2 << 20) // 4          # floats, 2 MB multiple
skews = [int(x) for x in (sys.argv[1:] or [0, 64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 1 << 20])]
slab = torch.empty(3 * S + 3 * max(skews) // 4 + 1024, dtype=torch.float32, device="cuda")
base_off = (-slab.data_ptr() % (2 << 20)) // 4                       # 2 MB aligned start
variants = {"separate tensors": tuple(eng.empty_feats(F, H) for _ in range(3))}
for sk in skews:
    o = [base_off + i * (S + sk // 4) for i in range(3)]
    variants["skew %7d B" % sk] = tuple(slab[a:a + n].view(F, H) for a in o)
times = {k: ([], []) for k in variants}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for r in range(14):
    for name, feats in variants.items():
        ev[0].record()
        aplan.run(out=feats)
        ev[1].record()
        splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm)
        ev[2].record()
        torch.cuda.synchronize()
        if r >= 2:
            times[name][0].append(ev[0].elapsed_time(ev[1]))
            times[name][1].append(ev[1].elapsed_time(ev[2]))
for name, (a, s) in times.items():
    print("%-18s analysis %.4f  synthesis+fixup %.4f  step %.4f ms   (ptr %% 2MB: %s)" % (
        name, statistics.median(a), statistics.median(s), statistics.median(a) + statistics.median(s),
        [hex(t.data_ptr() % (2 << 20)) for t in variants[name]]))
