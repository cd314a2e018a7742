"""
Corpus-level I/O batching around the hot path.

The reference's batch scripts push ONE utterance at a time through analysis / synthesis (one Pool worker each,
libutils.py:32-63).  On a GPU a single utterance is ~900 frames -- a few microseconds of kernel time behind
milliseconds of file reading, epoch parsing, host index math and feature-file writing.  Here a corpus is processed in
batches of `batch_utts` utterances per kernel launch, and the host side is a three-stage pipeline:

    reader thread  : wav + epochs (or feature files) of batch i+1          (disk, numpy)
    main thread    : plan build + kernels of batch i, D2H through pinned staging (engine.to_host_f64)
    writer thread  : raw float32 feature files / 16-bit wavs of batch i-1  (disk)

so disk, host math and the device overlap.  File formats and names are the reference's (raw float32 `.mag .real .imag
.lf0 .shift`, magphase.py:3014-3020; 16-bit wav at 0.98 peak, libaudio.py:352-365).  Multi-GPU: call with the shard of
tokens of this rank (magphase_amd.sharding) -- no data is exchanged between ranks.
"""
import os
import queue
import threading

import numpy as np

from . import libaudio as la
from . import libutils as lu


class _Stage(threading.Thread):
    """Runs fn(item) for every item of an input queue in order, pushes results to an output queue; None ends it."""

    def __init__(self, fn, q_in, q_out=None):
        super().__init__(daemon=True)
        self.fn, self.q_in, self.q_out, self.error = fn, q_in, q_out, None

    def run(self):
        while True:
            item = self.q_in.get()
            if item is None:
                break
            if self.error is None:
                try:
                    res = self.fn(item)
                    if self.q_out is not None:
                        self.q_out.put(res)
                except BaseException as e:   # re-raised by pipeline() in the caller's thread
                    self.error = e
                    if self.q_out is not None:
                        self.q_out.put(_Failed(e))
        if self.q_out is not None:
            self.q_out.put(None)


class _Failed:
    def __init__(self, err):
        self.err = err


def batches(items, n):
    items = list(items)
    return [items[i:i + n] for i in range(0, len(items), n)]


def pipeline(work, load, compute, store, depth=2):
    """
    Three-stage pipeline over the list `work`: load(w) in a reader thread (at most `depth` results ahead),
    compute(loaded) in the calling thread, store(result) in a writer thread.  Order is preserved; an exception in
    any stage is re-raised here after the threads have been shut down.  Returns the number of items completed.
    """
    q_work, q_loaded, q_store = queue.Queue(), queue.Queue(maxsize=depth), queue.Queue(maxsize=depth)
    reader = _Stage(load, q_work, q_loaded)
    writer = _Stage(store, q_store)
    reader.start()
    writer.start()
    for w in work:
        q_work.put(w)
    q_work.put(None)
    done, err = 0, None
    try:
        while True:
            item = q_loaded.get()
            if item is None:
                break
            if isinstance(item, _Failed):
                err = item.err
                continue
            if err is None and writer.error is None:
                try:
                    q_store.put(compute(item))
                    done += 1
                except BaseException as e:
                    err = e
    finally:
        q_store.put(None)
        writer.join()
        reader.join()
    err = err or reader.error or writer.error
    if err is not None:
        raise err
    return done


# ----------------------------------------------------------------------------------------------------
# feature extraction (scripts/batch_feature_extraction_for_tts.py)
# ----------------------------------------------------------------------------------------------------
def extract_features_corpus(wav_files, out_dir, batch_utts=16, fft_len=None, mag_dim=60, phase_dim=10,
                            b_const_rate=False, engine=None, verbose=True):
    """
    mp.analysis_for_acoustic_modelling (magphase.py:2992-3022) for a list of wav files, `batch_utts` per launch.
    All files of one call must share the sample rate (as every corpus the reference's script handles does); a batch
    with mixed rates is split.  Writes <token>.mag/.real/.imag/.lf0 (+ .shift for variable rate) into out_dir.
    """
    from . import magphase as mp

    lu.mkdir(out_dir)

    def load(files):
        utts = []
        for f in files:
            v_sig, fs = la.read_audio_file(f)
            v_pm_sec, v_voi = mp._epochs_for(f)
            utts.append((v_sig, fs, v_pm_sec, v_voi))
        return files, utts

    def compute(loaded):
        files, utts = loaded
        out = []
        for fs in sorted(set(u[1] for u in utts)):
            idx = [i for i, u in enumerate(utts) if u[1] == fs]
            # Q7: the reference forwards alpha_phase=b_mag_fbank_mel (False) -- see mp.analysis_for_acoustic_modelling
            res = mp.analysis_compressed_batch([utts[i] for i in idx], fft_len=fft_len, mag_dim=mag_dim,
                                               phase_dim=phase_dim, b_const_rate=b_const_rate, alpha_phase=False,
                                               engine=engine)
            out.extend((files[i], r) for i, r in zip(idx, res))
        return out

    def store(results):
        for f, (m_mag, m_real, m_imag, v_lf0, v_shift, _fs, _n) in results:
            tok = os.path.basename(f).split(".")[0]
            mp.write_featfile(m_mag, out_dir, tok + ".mag")
            mp.write_featfile(m_real, out_dir, tok + ".real")
            mp.write_featfile(m_imag, out_dir, tok + ".imag")
            mp.write_featfile(v_lf0, out_dir, tok + ".lf0")
            if not b_const_rate:
                mp.write_featfile(v_shift, out_dir, tok + ".shift")
            if verbose:
                print("extracted " + tok)

    return pipeline(batches(wav_files, batch_utts), load, compute, store)


# ----------------------------------------------------------------------------------------------------
# waveform generation (scripts/batch_waveform_generation.py)
# ----------------------------------------------------------------------------------------------------
def generate_waveforms_corpus(in_feats_dir, tokens, out_syn_dir, mag_dim, phase_dim, fs, fft_len=None, pf_type="no",
                              b_const_rate=False, batch_utts=16, engine=None, verbose=True):
    """
    mp.synthesis_from_acoustic_modelling (magphase.py:3229-3275) for a list of tokens, `batch_utts` per launch:
    reads <token>.mag/.real/.imag/.lf0, post-filters (pf_type 'magphase' on the device, 'merlin' on the host, 'no'),
    synthesises and writes <token>.wav.
    """
    from . import magphase as mp

    lu.mkdir(out_syn_dir)
    if pf_type not in ("no", "magphase", "merlin"):
        raise ValueError("pf_type must be 'no', 'magphase' or 'merlin'")

    def load(toks):
        utts = []
        for t in toks:
            base = os.path.join(in_feats_dir, t)
            m_mag = lu.read_binfile(base + ".mag", dim=mag_dim)
            if pf_type == "merlin":
                m_mag = mp.post_filter_merlin(m_mag, fs)       # host arithmetic: done in the reader thread
            utts.append((m_mag, lu.read_binfile(base + ".real", dim=phase_dim),
                         lu.read_binfile(base + ".imag", dim=phase_dim), lu.read_binfile(base + ".lf0", dim=1)))
        return toks, utts

    def compute(loaded):
        toks, utts = loaded
        sigs = mp.synthesis_from_compressed_batch(utts, fs, fft_len=fft_len, b_const_rate=b_const_rate,
                                                  b_post_filter=(pf_type == "magphase"), engine=engine)
        return list(zip(toks, sigs))

    def store(results):
        for t, v_sig in results:
            la.write_audio_file(os.path.join(out_syn_dir, t + ".wav"), v_sig, fs)
            if verbose:
                print("synthesised " + t)

    return pipeline(batches(tokens, batch_utts), load, compute, store)
