#!/usr/bin/env python
"""Board power of the compressed-path kernels (are they at the cap like the lossless ones?):
    python tools/power_lowdim.py        (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from magphase_amd import engine as em  # noqa: E402

torch.cuda.init()
eng = em.get_engine()
utts = bench.make_batch(0)
st = bench._lowdim_state(em, eng, utts)
os.environ["MAGPHASE_COMP_FUSED"] = "1"
xf = em.CompressedAnalysisPlan(eng, utts, mag_dim=60, phase_dim=10, alpha_phase=False)
xo = xf.run()
sp = st["splan"]
phases = [("configs2 analysis (k_analysis_f64 + k_mel_warp_mfma)", lambda: st["aplan"].run(feats=st["feats"], out=st["out"])),
          ("configs2 synthesis (unwarp, noise, k_synth_comp_pair)", lambda: sp.run(out=st["pcm"])),
          ("configs3 fused analysis (k_analysis_warp_fused)", lambda: xf.run(out=xo))]
pw = bench.measure_power(torch, 0, phases, seconds=2.0)
print("cap %.0f W idle %.0f W" % (pw["cap_W"], pw["idle_W"]))
for k, v in pw["phases"].items():
    print("%-62s %8.4f ms  %6.0f W  (%.3f of cap)  %.4f J above idle" % (k, v["ms"], v["board_W"], v["frac_of_cap"], v["energy_above_idle_J"]))
# per kernel of the synthesis side through the plan's marks
marks = bench._Marks(torch)
for _ in range(5):
    marks.ev = []
    sp.run(out=st["pcm"], mark=marks)
    torch.cuda.synchronize()
print(marks.durations())
