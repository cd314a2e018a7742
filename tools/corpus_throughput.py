#!/usr/bin/env python
"""
End-to-end corpus throughput through the file interface: wav + .est files on disk ->
iobatch.extract_features_corpus (reader thread, batched kernels, writer thread) -> feature files ->
iobatch.generate_waveforms_corpus -> wav files.  Everything the reference's two batch scripts do, timed wall-clock.
"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "demos"))
import numpy as np  # noqa: E402

import make_demo_data  # noqa: E402
from magphase_amd import iobatch, libaudio as la, synthetic as syn  # noqa: E402

N_UTT, DUR = int(os.environ.get("N_UTT", 128)), 5.0
tmp = tempfile.mkdtemp(prefix="mpx_corpus_")
try:
    wav_dir = os.path.join(tmp, "wavs")
    os.makedirs(wav_dir)
    toks = []
    for u in range(N_UTT):
        pcm, pm, voi = syn.make_utterance(3000 + u, dur_s=DUR)
        tok = "u%04d" % u
        la.write_audio_file(os.path.join(wav_dir, tok + ".wav"), pcm / 32768.0, 48000, norm=None)
        make_demo_data.write_est(os.path.join(wav_dir, tok + ".est"), pm, voi)
        toks.append(tok)
    wavs = [os.path.join(wav_dir, t + ".wav") for t in toks]
    feats = os.path.join(tmp, "feats")
    iobatch.extract_features_corpus(wavs[:8], os.path.join(tmp, "warm"), batch_utts=8, phase_dim=45, verbose=False)
    t = time.time()
    iobatch.extract_features_corpus(wavs, feats, batch_utts=32, phase_dim=45, verbose=False)
    t_ext = time.time() - t
    np.random.seed(1)
    iobatch.generate_waveforms_corpus(feats, toks[:8], os.path.join(tmp, "warm_syn"), 60, 45, 48000, pf_type="magphase",
                                      batch_utts=8, verbose=False)
    t = time.time()
    iobatch.generate_waveforms_corpus(feats, toks, os.path.join(tmp, "syn"), 60, 45, 48000, pf_type="magphase",
                                      batch_utts=32, verbose=False)
    t_gen = time.time() - t
    audio = N_UTT * DUR
    print("feature extraction (wav+est -> .mag/.real/.imag/.lf0/.shift): %d utterances, %.0f s of audio in %.2f s = %.0f x real time"
          % (N_UTT, audio, t_ext, audio / t_ext))
    print("waveform generation (features -> post-filter -> wav):          %d utterances, %.0f s of audio in %.2f s = %.0f x real time"
          % (N_UTT, audio, t_gen, audio / t_gen))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
