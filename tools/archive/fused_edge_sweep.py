"""Edge-case sweep of mpx_analysis_compressed_fused against the oracle: rates, coefficient counts 1..64 / 1..48, one to three\nshort utterances (partial rounds), very long and truncated frames (GPU box)."""
import os, sys, warnings, numpy as np
sys.path.insert(0,'.')
from oracle import magphase_oracle as orc
from magphase_amd import synthetic as syn, magphase as mp
from magphase_amd.engine import CompressedAnalysisPlan, get_engine
eng = get_engine()
def check(utts, fs, md, pd, ap=None, fb=False):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p = CompressedAnalysisPlan(eng, utts, mag_dim=md, phase_dim=pd, alpha_phase=ap, b_mag_fbank_mel=fb)
        assert p.fused
        a = [t.cpu().numpy().astype(np.float64) for t in p.run()]
        e = [0,0,0]
        for u,(x,_f,pm,voi) in enumerate(utts):
            ol = orc.analysis_lossless_from_epochs(x, fs, pm, voi)
            o = orc.format_for_modelling(ol[0], ol[1], ol[2], ol[3], fs, mag_dim=md, phase_dim=pd, alpha_phase=ap, b_mag_fbank_mel=fb)
            s0,s1 = int(p.out_off[u]), int(p.out_off[u+1])
            fl = o[0] == -1e10
            e[0]=max(e[0], np.max(np.abs(a[0][s0:s1]-o[0])[~fl], initial=0)); e[1]=max(e[1], np.max(np.abs(a[1][s0:s1]-o[1]))); e[2]=max(e[2], np.max(np.abs(a[2][s0:s1]-o[2])))
    return p.lossless.total_frames, e
for fs in (48000, 16000, 44100, 22050):
    for (md, pd) in ((64, 48), (3, 1), (16, 16), (17, 17), (60, 32)):
        for nu, dur in ((1, 0.05), (1, 0.3), (3, 0.21)):
            utts=[]
            for u in range(nu):
                pcm, pm, voi = syn.make_utterance(900+u+nu, dur_s=dur, fs=fs); utts.append((syn.pcm_to_float(pcm), fs, pm, voi))
            try:
                F, e = check(utts, fs, md, pd)
                flag = "" if (e[0] < 2e-5 and e[1] < 3e-6 and e[2] < 3e-6) else "   <<<<<< OVER"
                print("fs %5d dims %2d/%2d utts %d dur %.2f frames %4d: %.2e %.2e %.2e%s" % (fs, md, pd, nu, dur, F, *e, flag), flush=True)
            except Exception as ex:
                print("fs %5d dims %2d/%2d utts %d dur %.2f: %s: %s" % (fs, md, pd, nu, dur, type(ex).__name__, str(ex)[:150]), flush=True)
# long frames (very low pitch: window longer than the LDS region) and frames longer than N
fs=48000
x = np.random.RandomState(1).uniform(-0.3,0.3,60000)
pm = np.array([0.02, 0.05, 0.09, 0.20, 0.2002, 0.26, 0.40, 0.47, 0.48, 0.60, 0.75, 0.95, 1.0, 1.1, 1.2]); voi = np.ones_like(pm)
F, e = check([(x, fs, pm, voi)], fs, 60, 45)
print("long / truncated frames:", F, e)
