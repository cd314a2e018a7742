#!/usr/bin/env python
"""Accuracy figures of the built-in epoch / voicing tracker on synthetic truth, on the GPU box:
    python tools/epoch_accuracy.py > gpurun_out/epoch_accuracy.json      (copied to profiles/r03_epoch_accuracy.json)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magphase_amd import epochs, synthetic as syn  # noqa: E402

out = {"what": "magphase_amd.epochs.track_epochs_batch against the generator's own epochs (synthetic.make_utterance, 5 s "
               "utterances), scores of epochs.accuracy_against_truth pooled over the utterances by their number of true "
               "voiced epochs"}
keys = ("identification_rate", "miss_rate", "false_alarm_rate", "jitter_us", "bias_us", "gross_f0_error_rate",
        "f0_fine_error_percent", "voicing_error_rate")
for fs, us in ((48000, range(100, 164)), (16000, range(200, 264))):
    data = [syn.make_utterance(u, dur_s=5.0, fs=fs) for u in us]
    res = epochs.track_epochs_batch([d[0] for d in data], fs)
    rows = [epochs.accuracy_against_truth(pm, voi, e_pm, e_voi) for (_p, pm, voi), (e_pm, e_voi) in zip(data, res)]
    w = np.array([r["true_voiced_epochs"] for r in rows], dtype=np.float64)
    out["fs_%d" % fs] = dict(utterances=len(rows), true_voiced_epochs=int(w.sum()),
                             **{k: round(float(np.sum(w * np.array([r.get(k, np.nan) for r in rows])) / w.sum()), 5) for k in keys},
                             worst_utterance_identification_rate=round(min(r["identification_rate"] for r in rows), 4))
print(json.dumps(out, indent=1))
