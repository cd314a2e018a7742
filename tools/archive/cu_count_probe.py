"""k_synth_ola_pair on fewer CUs (MAGPHASE_SYN_CUS), interleaved A/B: at the power limit the CU count hardly matters (GPU box)."""
import os, sys, numpy as np
sys.path.insert(0,'.')
import torch, bench
from magphase_amd.engine import LosslessAnalysisPlan, LosslessSynthesisPlan, get_engine
eng = get_engine(); utts = bench.make_batch(0)
aplan = LosslessAnalysisPlan(eng, utts)
H, F = aplan.fft_len//2+1, aplan.total_frames
feats = tuple(eng.empty_feats(F, H) for _ in range(3)); aplan.run(out=feats)
cfgs = ("256", "248", "240", "232", "224", "216", "208", "200")
plans = {}
for cus in cfgs:
    os.environ["MAGPHASE_SYN_CUS"] = cus
    sp = LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
    plans[cus] = (sp, eng.empty((max(sp.strip_floats,1)+65536,)))
pcm = eng.empty((plans["256"][0].total_out,))
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
rep = {c: [] for c in cfgs}; ins = {c: [] for c in cfgs}
for r in range(24):
    for c in cfgs:
        sp, st = plans[c]
        aplan.run(out=feats)
        e[0].record(); sp.run(feats[0], feats[1], feats[2], strips=st, out=pcm); e[1].record(); sp.run(feats[0], feats[1], feats[2], strips=st, out=pcm); e[2].record(); torch.cuda.synchronize()
        if r >= 4: ins[c].append(e[0].elapsed_time(e[1])); rep[c].append(e[1].elapsed_time(e[2]))
for c in cfgs:
    print("CUs %s: after analysis %.4f (min %.4f)   repeated %.4f (min %.4f)" % (c, np.median(ins[c]), min(ins[c]), np.median(rep[c]), min(rep[c])), flush=True)
