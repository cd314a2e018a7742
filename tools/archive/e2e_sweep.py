#!/usr/bin/env python
"""File-interface throughput for a few corpus / launch sizes (tools/corpus_throughput.run): python tools/e2e_sweep.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import corpus_throughput  # noqa: E402

for n_utt, batch in ((128, 32), (256, 32), (256, 64), (512, 64)):
    r = corpus_throughput.run(n_utt=n_utt, batch_utts=batch, noise_mode="reference")
    print(json.dumps({"n_utt": n_utt, "batch": batch, "extraction_x": r["extraction_x_realtime"], "extraction_s": r["extraction_samples_s"],
                      "generation_x": r["generation_x_realtime"], "gen_s": r["generation"]["reference"]["samples_s"],
                      "ext_busy": r["stage_busy_s"]["extraction"], "gen_busy": r["generation"]["reference"]["stage_busy_s"]}), flush=True)
