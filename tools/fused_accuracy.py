#!/usr/bin/env python
"""
Accuracy of build variants' fused compressed analysis (mpx_analysis_compressed_fused) against the oracle, on the utterances of
tests/test_gpu_compressed.py::test_fused_compressed_analysis_matches_oracle_and_staged_path: worst case and sum of squared
errors of the magnitude and the phase features.  Variants are tools/ab_bench.py builds:

    python tools/ab_bench.py --prepare a b:-DFOO=1 && python tools/fused_accuracy.py a b        (on an MI355X)
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import ab_bench
    from magphase_amd import synthetic as syn
    from oracle import magphase_oracle as orc   # the checker

    warnings.simplefilter("ignore")
    for fs, md, pd, ap in [(48000, 60, 10, False), (48000, 60, 45, None), (16000, 60, 45, None)]:
        utts = []
        for u in range(4):
            pcm, pm, voi = syn.make_utterance(70 + u, dur_s=0.9 + 0.13 * u, fs=fs)
            utts.append((syn.pcm_to_float(pcm), fs, pm, voi))
        refs = []
        for (x, _fs, pm, voi) in utts:
            ol = orc.analysis_lossless_from_epochs(x, fs, pm, voi)
            refs.append(orc.format_for_modelling(ol[0], ol[1], ol[2], ol[3], fs, mag_dim=md, phase_dim=pd, alpha_phase=ap))
        for name in sys.argv[1:]:
            em = ab_bench.load(name)
            pf = em.CompressedAnalysisPlan(em.Engine(), utts, mag_dim=md, phase_dim=pd, alpha_phase=ap)
            assert pf.fused
            a = [t.cpu().numpy().astype(np.float64) for t in pf.run()]
            wm, wp, sm, sp = 0.0, 0.0, 0.0, 0.0
            for u, o in enumerate(refs):
                s0, s1 = int(pf.out_off[u]), int(pf.out_off[u + 1])
                d = np.abs(a[0][s0:s1] - o[0])[o[0] != -1.0e10]
                dp = np.concatenate((np.abs(a[1][s0:s1] - o[1]).ravel(), np.abs(a[2][s0:s1] - o[2]).ravel()))
                wm, wp, sm, sp = max(wm, d.max()), max(wp, dp.max()), sm + float(np.sum(d ** 2)), sp + float(np.sum(dp ** 2))
            print("%d Hz %d/%d %-8s mag max %.3g (sum sq %.3g)  phase max %.3g (sum sq %.3g)" % (fs, md, pd, name, wm, sm, wp, sp),
                  flush=True)


if __name__ == "__main__":
    main()
