"""Constant-rate compressed analysis, one kernel (mpx_analysis_compressed_fused_cr) vs the staged pair: agreement with each
other and with the oracle, and times (GPU box).   python tools/fused_cr_check.py [quick]"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, '.')
import torch
from oracle import magphase_oracle as orc
from magphase_amd import synthetic as syn
from magphase_amd.engine import CompressedAnalysisPlan, get_engine
eng = get_engine()
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
for fs, md, pd, nu, dur in ((48000, 60, 45, 6, 1.5), (16000, 60, 45, 6, 1.5), (48000, 60, 10, 3, 0.4), (16000, 24, 16, 70, 1.0)):
    utts = []
    for u in range(nu):
        pcm, pm, voi = syn.make_utterance(40 + u, dur_s=dur, fs=fs); utts.append((pcm, fs, pm, voi))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        os.environ["MAGPHASE_COMP_FUSED_CR"] = "1"
        pf = CompressedAnalysisPlan(eng, utts, mag_dim=md, phase_dim=pd, b_const_rate=True)
        assert pf.fused_cr
        a = [t.cpu().numpy().astype(np.float64) for t in pf.run()]
        os.environ["MAGPHASE_COMP_FUSED_CR"] = "0"
        ps = CompressedAnalysisPlan(eng, utts, mag_dim=md, phase_dim=pd, b_const_rate=True)
        assert not ps.fused_cr
        b = [t.cpu().numpy().astype(np.float64) for t in ps.run()]
        print("fs %d dims %d/%d, %d utts, %d -> %d frames: one kernel vs staged: mag %.2e  real %.2e  imag %.2e"
              % (fs, md, pd, nu, pf.lossless.total_frames, pf.total_out_frames, *(np.max(np.abs(x - y)) for x, y in zip(a, b))), flush=True)
        if nu <= 6:
            e = [0, 0, 0]; es = [0, 0, 0]
            for u in range(len(utts)):
                pcm, _f, pm, voi = utts[u]
                o = orc.analysis_compressed_from_epochs(pcm.astype(np.float64) / 32768.0, fs, pm, voi, mag_dim=md, phase_dim=pd, b_const_rate=True)
                s0, s1 = int(pf.out_off[u]), int(pf.out_off[u + 1])
                for k in range(3):
                    e[k] = max(e[k], np.max(np.abs(a[k][s0:s1] - o[k]))); es[k] = max(es[k], np.max(np.abs(b[k][s0:s1] - o[k])))
            print("   vs oracle: one kernel %.2e %.2e %.2e | staged %.2e %.2e %.2e" % (*e, *es), flush=True)
if not quick:   # timing at configs[2] size
    utts = []
    for u in range(64):
        pcm, pm, voi = syn.make_utterance(u, dur_s=5.0, fs=48000); utts.append((pcm, 48000, pm, voi))
    for flag in ("1", "0"):
        os.environ["MAGPHASE_COMP_FUSED_CR"] = flag
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p = CompressedAnalysisPlan(eng, utts, mag_dim=60, phase_dim=45, b_const_rate=True)
        out = p.run(); feats = None
        if flag == "0":
            feats = tuple(eng.empty_feats(p.lossless.total_frames, 2049) for _ in range(3))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for r in range(14):
            e0.record(); p.run(feats=feats, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("configs[2] analysis, one kernel=%s: %.4f ms (min %.4f) for %d -> %d frames" % (flag, np.median(ts[2:]), min(ts), p.lossless.total_frames, p.total_out_frames), flush=True)
