#!/usr/bin/env python
"""Does it matter WHEN the feature matrices are allocated?  mode 'first': three matrices (or one slab) right after the
context exists, before the engine / plans; mode 'last': after everything else, as bench.py does."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "last"
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
F_GUESS, H = 57100, 2049
pre = None
if mode == "first":
    pre = tuple(torch.empty((F_GUESS, H), dtype=torch.float32, device="cuda") for _ in range(3))
elif mode == "slab-first":
    slab = torch.empty(3 * F_GUESS * H, dtype=torch.float32, device="cuda")
    pre = tuple(slab[i * F_GUESS * H:(i + 1) * F_GUESS * H].view(F_GUESS, H) for i in range(3))
import bench  # noqa: E402
from magphase_amd import engine as em  # noqa: E402

utts = bench.make_batch(0)
eng = em.Engine()
aplan = em.LosslessAnalysisPlan(eng, utts)
splan = em.LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
F = aplan.total_frames
assert F <= F_GUESS
strips = eng.empty((max(splan.strip_floats, 1),))
pcm = eng.empty((splan.total_out,))
if mode == "slab-last":
    slab = torch.empty(3 * F * H + (48 << 20), dtype=torch.float32, device="cuda")
    st = F * H + (16 << 20)                      # 64 MB between the matrices (tools/placement_probe.py)
    feats = tuple(slab[i * st:i * st + F * H].view(F, H) for i in range(3))
elif pre is None:
    feats = tuple(eng.empty_feats(F, H) for _ in range(3))
else:
    feats = tuple(p[:F] for p in pre)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
ta, ts = [], []
for r in range(40):
    ev[0].record()
    aplan.run(out=feats)
    ev[1].record()
    splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm)
    ev[2].record()
    torch.cuda.synchronize()
    if r >= 5:
        ta.append(ev[0].elapsed_time(ev[1]))
        ts.append(ev[1].elapsed_time(ev[2]))
print("%-10s analysis %.4f  synthesis+fixup %.4f  step %.4f ms" % (mode, statistics.median(ta), statistics.median(ts),
                                                                    statistics.median(ta) + statistics.median(ts)))
