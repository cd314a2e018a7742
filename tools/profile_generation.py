#!/usr/bin/env python
"""Where generation's wall time goes: iobatch.generate_waveforms_corpus on 128 synthetic utterances, both noise modes,
per-launch duration of mp.synthesis_from_compressed_batch inside the pipeline, then a cProfile of that call."""
import os, sys, time, tempfile, shutil, threading, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'demos')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import make_demo_data
from magphase_amd import iobatch, libaudio as la, synthetic as syn, magphase as mp
tmp = tempfile.mkdtemp(prefix="mpx_corpus_")
wav_dir = os.path.join(tmp, "wavs"); os.makedirs(wav_dir)
toks = []
for u in range(128):
    pcm, pm, voi = syn.make_utterance(3000 + u, dur_s=5.0)
    tok = "u%04d" % u
    la.write_audio_file(os.path.join(wav_dir, tok + ".wav"), pcm / 32768.0, 48000, norm=None)
    make_demo_data.write_est(os.path.join(wav_dir, tok + ".est"), pm, voi)
    toks.append(tok)
wavs = [os.path.join(wav_dir, t + ".wav") for t in toks]
feats = os.path.join(tmp, "feats")
iobatch.extract_features_corpus(wavs, feats, batch_utts=32, phase_dim=45, verbose=False)
orig = mp.synthesis_from_compressed_batch
calls = []
def timed(*a, **k):
    t = time.perf_counter(); r = orig(*a, **k); calls.append(time.perf_counter() - t); return r
mp.synthesis_from_compressed_batch = timed
for mode in ("reference", "device", "reference"):
    for rep in range(3):
        calls.clear()
        np.random.seed(1)
        rep_g = iobatch.CorpusReport()
        t = time.time()
        iobatch.generate_waveforms_corpus(feats, toks, os.path.join(tmp, "syn_" + mode), 60, 45, 48000, pf_type="magphase",
                                          batch_utts=32, verbose=False, report=rep_g, noise_mode=mode)
        print(mode, "total %.1f ms" % ((time.time() - t) * 1e3), "calls ms:", ["%.1f" % (c * 1e3) for c in calls],
              {k: round(v, 3) for k, v in rep_g.items() if k.endswith("_s")})
# profile the compute call alone in reference mode, inside the pipeline
pr = cProfile.Profile()
def prof(*a, **k):
    pr.enable(); r = orig(*a, **k); pr.disable(); return r
mp.synthesis_from_compressed_batch = prof
np.random.seed(1)
iobatch.generate_waveforms_corpus(feats, toks, os.path.join(tmp, "syn_p"), 60, 45, 48000, pf_type="magphase", batch_utts=32, verbose=False, noise_mode="reference")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
shutil.rmtree(tmp)
