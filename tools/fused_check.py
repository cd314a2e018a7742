"""Fused vs staged compressed analysis: agreement with each other and with the oracle, and kernel times (GPU box)."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, '.')
import torch
from oracle import magphase_oracle as orc
from magphase_amd import synthetic as syn
from magphase_amd.engine import CompressedAnalysisPlan, get_engine
eng = get_engine()
for fs, md, pd, ap in ((48000, 60, 10, False), (48000, 60, 45, None), (16000, 60, 45, None), (16000, 24, 16, None)):
    utts = []
    for u in range(6):
        pcm, pm, voi = syn.make_utterance(40 + u, dur_s=1.5, fs=fs); utts.append((pcm, fs, pm, voi))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        os.environ["MAGPHASE_COMP_FUSED"] = "1"
        pf = CompressedAnalysisPlan(eng, utts, mag_dim=md, phase_dim=pd, alpha_phase=ap)
        assert pf.fused
        a = [t.cpu().numpy().astype(np.float64) for t in pf.run()]
        os.environ["MAGPHASE_COMP_FUSED"] = "0"
        ps = CompressedAnalysisPlan(eng, utts, mag_dim=md, phase_dim=pd, alpha_phase=ap)
        assert not ps.fused
        b = [t.cpu().numpy().astype(np.float64) for t in ps.run()]
        print("fs %d dims %d/%d: fused vs staged: mag %.2e  real %.2e  imag %.2e" % (fs, md, pd, *(np.max(np.abs(x - y)) for x, y in zip(a, b))), flush=True)
        e = [0, 0, 0]; es = [0, 0, 0]
        for u in range(len(utts)):
            pcm, _f, pm, voi = utts[u]
            o = orc.analysis_compressed_from_epochs(pcm.astype(np.float64) / 32768.0, fs, pm, voi, mag_dim=md, phase_dim=pd, alpha_phase=ap)
            s0, s1 = int(pf.out_off[u]), int(pf.out_off[u + 1])
            for k in range(3):
                e[k] = max(e[k], np.max(np.abs(a[k][s0:s1] - o[k]))); es[k] = max(es[k], np.max(np.abs(b[k][s0:s1] - o[k])))
        print("   vs oracle: fused %.2e %.2e %.2e | staged %.2e %.2e %.2e" % (*e, *es), flush=True)
# timing at configs[3] size
utts = []
for u in range(64):
    pcm, pm, voi = syn.make_utterance(u, dur_s=5.0, fs=48000); utts.append((pcm, 48000, pm, voi))
for md, pd, ap in ((60, 10, False), (60, 45, None)):
    for fused in ("1", "0"):
        os.environ["MAGPHASE_COMP_FUSED"] = fused
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p = CompressedAnalysisPlan(eng, utts, mag_dim=md, phase_dim=pd, alpha_phase=ap)
        out = p.run(); feats = None
        if fused == "0":
            feats = tuple(eng.empty_feats(p.lossless.total_frames, 2049) for _ in range(3))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for r in range(12):
            e0.record(); p.run(out=out) if fused == "1" else p.run(feats=feats, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("dims %d/%d fused=%s: %.4f ms (min %.4f) for %d frames" % (md, pd, fused, np.median(ts[2:]), min(ts), p.lossless.total_frames), flush=True)
