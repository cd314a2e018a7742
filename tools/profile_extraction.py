#!/usr/bin/env python
"""Where extraction's wall time goes: iobatch.extract_features_corpus on 128 synthetic utterances, stage timings of three
runs, then a cProfile of the compute stage (mp.analysis_for_acoustic_modelling's batch form) inside the pipeline."""
import cProfile
import os
import pstats
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "demos")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_demo_data  # noqa: E402

from magphase_amd import iobatch, libaudio as la, synthetic as syn  # noqa: E402

tmp = tempfile.mkdtemp(prefix="mpx_corpus_")
wav_dir = os.path.join(tmp, "wavs")
os.makedirs(wav_dir)
wavs = []
for u in range(128):
    pcm, pm, voi = syn.make_utterance(3000 + u, dur_s=5.0)
    tok = "u%04d" % u
    la.write_audio_file(os.path.join(wav_dir, tok + ".wav"), pcm / 32768.0, 48000, norm=None)
    make_demo_data.write_est(os.path.join(wav_dir, tok + ".est"), pm, voi)
    wavs.append(os.path.join(wav_dir, tok + ".wav"))
iobatch.extract_features_corpus(wavs[:32], os.path.join(tmp, "warm"), batch_utts=32, phase_dim=45, verbose=False)
for rep in range(3):
    r = iobatch.CorpusReport()
    t = time.time()
    iobatch.extract_features_corpus(wavs, os.path.join(tmp, "f%d" % rep), batch_utts=32, phase_dim=45, verbose=False, report=r)
    print("extraction %.1f ms" % ((time.time() - t) * 1e3), {k: round(v, 3) for k, v in r.items() if k.endswith("_s")})
pr = cProfile.Profile()
orig = iobatch.pipeline


def prof_pipeline(items, load, compute, store, **kw):
    def c2(x):
        pr.enable()
        try:
            return compute(x)
        finally:
            pr.disable()
    return orig(items, load, c2, store, **kw)


iobatch.pipeline = prof_pipeline
iobatch.extract_features_corpus(wavs, os.path.join(tmp, "fp"), batch_utts=32, phase_dim=45, verbose=False)
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
shutil.rmtree(tmp)
