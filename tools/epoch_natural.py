#!/usr/bin/env python
"""
The built-in epoch / voicing front end (reference: REAPER, libaudio.py:421-455, magphase.py:2875-2879)
on the reference's ten natural recordings (demos/data_48k/wavs_nat) against the voicing their phone labels imply
(demos/data_48k/labs: magphase_amd.epochs.score_against_labels) -- at 48 kHz and, decimated by 3, at 16 kHz.

    python tools/epoch_natural.py [out.json]          (needs an MI355X: the tracker's kernels have no CPU path)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scipy import signal

from magphase_amd import epochs, libaudio as la


def main():
    d = os.path.join(ROOT, "demos", "data_48k")
    toks = sorted(f[:-4] for f in os.listdir(os.path.join(d, "wavs_nat")) if f.endswith(".wav"))
    out = {"what": __doc__.strip().split("\n\n")[0].replace("\n", " "), "files": {}}
    for rate in (48000, 16000):
        sigs = []
        for t in toks:
            x, fs = la.read_audio_file(os.path.join(d, "wavs_nat", t + ".wav"))
            if rate != fs:
                x = signal.resample_poly(x, 1, fs // rate)
            sigs.append(x)
        res = epochs.track_epochs_batch(sigs, rate)
        rows = {}
        for t, (pm, voi) in zip(toks, res):
            rows[t] = epochs.score_against_labels(pm, voi, os.path.join(d, "labs", t + ".lab"))
        w = np.array([r["voiced_points"] + r["unvoiced_points"] for r in rows.values()], dtype=np.float64)
        pooled = {k: float(np.sum(w * np.array([r[k] for r in rows.values()])) / w.sum())
                  for k in ("voiced_recall", "unvoiced_recall", "agreement", "f0_jump_rate")}
        pooled["worst_file_agreement"] = float(min(r["agreement"] for r in rows.values()))
        out["files"][str(rate)] = rows
        out.setdefault("pooled", {})[str(rate)] = pooled
        print("%d Hz: %s" % (rate, ", ".join("%s %.4f" % kv for kv in pooled.items())), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
