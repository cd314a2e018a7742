#!/usr/bin/env python
"""Large-batch robustness run: 2400 utterances (2 h of audio) through ONE launch of each kernel -- index ranges, grid
limits and position independence (the same utterance gives bit-identical PCM wherever it sits in the batch)."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magphase_amd import synthetic as syn, magphase as mp
from magphase_amd.engine import get_engine, LosslessAnalysisPlan, LosslessSynthesisPlan, CompressedAnalysisPlan
eng = get_engine()
base = []
for u in range(40):
    pcm, pm, voi = syn.make_utterance(u, dur_s=3.0)
    base.append((syn.pcm_to_float(pcm), 48000, pm, voi))
utts = [base[i % 40] for i in range(2400)]          # 2400 utterances x 3 s = 2 h of audio in ONE launch
t = time.time(); ap = LosslessAnalysisPlan(eng, utts); print("analysis plan %.2fs, frames %d" % (time.time()-t, ap.total_frames))
t = time.time(); mag, real, imag = ap.run(); torch.cuda.synchronize(); print("analysis %.1f ms, feats %.1f GB" % ((time.time()-t)*1e3, 3*mag.numel()*4/1e9))
t = time.time(); sp = LosslessSynthesisPlan(eng, ap.v_f0, ap.fs, ap.fft_len); print("synthesis plan %.2fs, runs %d" % (time.time()-t, sp.n_runs))
t = time.time(); pcm = sp.run(mag, real, imag); torch.cuda.synchronize(); print("synthesis %.1f ms" % ((time.time()-t)*1e3))
# round trip on first / last utterance
for u in (0, 2399):
    a, b = int(sp.out_off_host[u]), int(sp.out_off_host[u+1])
    y = pcm[a:b].cpu().numpy(); x = utts[u][0][:y.size]
    n0, n1 = 2400, min(x.size, y.size) - 2400
    err = np.max(np.abs(y[n0:n1]-x[n0:n1])); print("utt", u, "roundtrip max err %.2e" % err); assert err < 2e-5
# identical utterances give the same output wherever they sit in the batch -- up to the fp32 re-association at the run
# boundaries (the batch's frames are cut into equal shares, so the cuts fall differently in utterance 0 and 2360)
a0, b0 = int(sp.out_off_host[0]), int(sp.out_off_host[1]); a1, b1 = int(sp.out_off_host[2360]), int(sp.out_off_host[2361])
d = (pcm[a0:b0] - pcm[a1:b1]).abs().max().item(); print("position dependence max |d| %.2e" % d)
assert d <= 2e-6 * pcm[a0:b0].abs().max().item()
del mag, real, imag, pcm
cp = CompressedAnalysisPlan(eng, utts, mag_dim=60, phase_dim=45, b_const_rate=True)
t = time.time(); out = cp.run(); torch.cuda.synchronize(); print("compressed analysis %.1f ms, const-rate frames %d" % ((time.time()-t)*1e3, cp.total_out_frames))
assert all(torch.isfinite(o).all() for o in out)
print("OK")
