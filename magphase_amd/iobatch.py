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
    """Runs fn(item) for every item of an input queue in order, pushes results to an output queue; None ends it.
    stop: threading.Event -- once set, remaining items are dropped and blocking puts give up (abnormal shutdown)."""

    def __init__(self, fn, q_in, q_out=None, stop=None):
        super().__init__(daemon=True)
        self.fn, self.q_in, self.q_out, self.error = fn, q_in, q_out, None
        self.stop = stop or threading.Event()

    def _put(self, item):
        while not self.stop.is_set():
            try:
                self.q_out.put(item, timeout=0.2)
                return
            except queue.Full:
                continue

    def run(self):
        while True:
            item = self.q_in.get()
            if item is None or self.stop.is_set():
                break
            if self.error is None:
                try:
                    res = self.fn(item)
                    if self.q_out is not None:
                        self._put(res)
                except BaseException as e:   # re-raised by pipeline() in the caller's thread
                    self.error = e
                    if self.q_out is not None:
                        self._put(_Failed(e))
        if self.q_out is not None:
            self._put(None)


class _Failed:
    def __init__(self, err):
        self.err = err


def batches(items, n):
    items = list(items)
    return [items[i:i + n] for i in range(0, len(items), n)]


def pipeline(work, load, compute, store, depth=2, timings=None):
    """
    Three-stage pipeline over the list `work`: load(w) in a reader thread (at most `depth` results ahead),
    compute(loaded) in the calling thread, store(result) in a writer thread.  Order is preserved; an exception in
    any stage is re-raised here after the threads have been shut down (a KeyboardInterrupt in the calling thread too:
    the stages are told to stop and never block on a full queue, so Ctrl-C does not hang).
    Returns the number of items completed.
    """
    q_work, q_loaded, q_store = queue.Queue(), queue.Queue(maxsize=depth), queue.Queue(maxsize=depth)
    stop = threading.Event()
    if timings is not None:      # busy seconds of the three stages (they overlap: the slowest one sets the wall time)
        import time

        def timed(fn, key):
            def g(x):
                t0 = time.perf_counter()
                try:
                    return fn(x)
                finally:
                    timings[key] = timings.get(key, 0.0) + time.perf_counter() - t0
            return g
        load, compute, store = timed(load, "load_s"), timed(compute, "compute_s"), timed(store, "store_s")
    if os.environ.get("MAGPHASE_IO_PIPELINE", "1") == "0":   # no threads: load, compute, store one after the other
        done = 0
        for w in work:
            store(compute(load(w)))
            done += 1
        return done
    reader = _Stage(load, q_work, q_loaded, stop)
    writer = _Stage(store, q_store, None, stop)
    # The compute stage is hundreds of short ctypes / torch calls, each of which hands the GIL over; with CPython's
    # default 5 ms switch interval it then waits up to 5 ms to get it back from the reader or writer thread (measured:
    # 36 ms per batch instead of 11).  A 0.1 ms interval for the duration of the pipeline removes that.
    import gc
    import sys
    old_switch = sys.getswitchinterval()
    sys.setswitchinterval(float(os.environ.get("MAGPHASE_SWITCH_INTERVAL", "1e-4")))
    # The cyclic collector runs full collections over the interpreter's whole heap (torch's modules included) every few
    # hundred container allocations of the planners: 5-40 ms pauses, measured 0.037-0.082 s per 128-utterance
    # generation run with it against a steady 0.035 s without.  The stages allocate arrays and tuples, no cycles; one
    # collection runs at the end.
    gc_was_on = gc.isenabled() and os.environ.get("MAGPHASE_GC_PAUSE", "1") != "0"
    if gc_was_on:
        gc.disable()
    reader.start()
    writer.start()
    for w in work:
        q_work.put(w)
    q_work.put(None)
    done, err, clean = 0, None, False
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
                except Exception as e:
                    err = e
        clean = True
    finally:
        if not clean:                 # interrupted: nobody will drain the queues any more
            stop.set()
            q_work.put(None)
        try:
            q_store.put(None, timeout=1.0 if not clean else None)
        except queue.Full:
            pass
        writer.join(timeout=None if clean else 2.0)
        reader.join(timeout=None if clean else 2.0)
        sys.setswitchinterval(old_switch)
        if gc_was_on:
            gc.enable()
    err = err or reader.error or writer.error
    if err is not None:
        raise err
    return done


class CorpusReport(dict):
    """Filled by the two corpus functions: done (utterances written), failed [(token, 'ExcType: message')], crash_list
    (path of the crash_file_list_<host>_<pid>.scp the failed tokens were appended to, or None)."""


def _record_failures(report, out_dir, failed):
    """The reference's crash-list convention (scripts/batch_convert_label_state_aligned_to_variable_frame_rate.py:48,
    59-70): tokens that raised are appended to crash_file_list_<host>_<pid>.scp and the run goes on."""
    if not failed:
        return
    path = lu.ins_pid(os.path.join(out_dir, "crash_file_list.scp"))
    with open(path, "a") as fh:
        for tok, _msg in failed:
            fh.write(tok + "\n")
    if report is not None:
        report.setdefault("failed", []).extend(failed)
        report["crash_list"] = path


def _isolate(items, fn_batch, keep_numpy_rng=False, rng_engine=None):
    """fn_batch(list) -> list of results.  If the whole batch raises, every item is retried on its own so that one bad
    utterance costs one utterance: returns (results, failed) with failed = [(index, exception)].
    keep_numpy_rng: the batch draws from numpy's GLOBAL generator (reference noise, magphase.py:883); the failed attempt
    may already have advanced it, so its state is put back before the retries -- the good utterances then get the noise
    they would have got without the bad one in the batch (a bad utterance that draws before it fails still shifts the
    stream for the ones after it, as it would in the reference's own loop)."""
    # (rng_engine: the generator's state may be a deferred one on the device, Engine.mt_snapshot / mt_restore)
    state = None
    if keep_numpy_rng:
        state = rng_engine.mt_snapshot() if rng_engine is not None else ("host", np.random.get_state())
    try:
        return list(zip(range(len(items)), fn_batch(items))), []
    except (KeyboardInterrupt, SystemExit):
        raise
    except Exception:
        if state is not None:
            if rng_engine is not None:
                rng_engine.mt_restore(state)
            else:
                np.random.set_state(state[1])
        ok, failed = [], []
        for i, it in enumerate(items):
            try:
                ok.append((i, fn_batch([it])[0]))
            except (KeyboardInterrupt, SystemExit):
                raise
            except Exception as e:
                failed.append((i, e))
        return ok, failed


def _tok(path):
    return os.path.basename(path).split(".")[0]


_IO_POOL = None


def _io_map(fn, items):
    """fn over items, results in order; an item's exception is returned in place of its result.  Sequential by default:
    the per-file work is short numpy / parsing calls that hold the GIL, and on the MI355X host a 4-thread pool measured
    7x SLOWER per file (1.96 ms vs 0.27 ms) than the plain loop.  MAGPHASE_IO_THREADS > 1 enables a pool for slow
    (network) file systems, where the reads themselves dominate."""
    global _IO_POOL
    items = list(items)
    n = int(os.environ.get("MAGPHASE_IO_THREADS", "1"))
    if n <= 1 or len(items) <= 1:
        pool_map = map
    else:
        if _IO_POOL is None:
            from concurrent.futures import ThreadPoolExecutor

            _IO_POOL = ThreadPoolExecutor(max_workers=n, thread_name_prefix="mpx_io")
        pool_map = _IO_POOL.map

    def safe(it):
        try:
            return fn(it)
        except (KeyboardInterrupt, SystemExit):
            raise
        except Exception as e:
            return e

    return list(pool_map(safe, items))


def token_seed(token):
    """64-bit noise seed of an utterance for noise_mode='device': FNV-1a of the token's UTF-8 bytes."""
    h = 0xCBF29CE484222325
    for b in str(token).encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


# ----------------------------------------------------------------------------------------------------
# feature extraction (scripts/batch_feature_extraction_for_tts.py)
# ----------------------------------------------------------------------------------------------------
def extract_features_corpus(wav_files, out_dir, batch_utts=16, fft_len=None, mag_dim=60, phase_dim=10,
                            b_const_rate=False, engine=None, verbose=True, report=None):
    """
    mp.analysis_for_acoustic_modelling (magphase.py:2992-3022) for a list of wav files, `batch_utts` per launch.
    Sample rates may be mixed: every batch is split by rate (one launch per rate).  Writes <token>.mag/.real/.imag/.lf0
    (+ .shift for variable rate) into out_dir.  A file that cannot be read or analysed does not stop the corpus: its
    token goes to crash_file_list_<host>_<pid>.scp in out_dir (report, a CorpusReport / dict, gets the details).
    Returns the number of batches completed.
    """
    from . import magphase as mp

    lu.mkdir(out_dir)
    # the engine's device, resolved in THIS thread (torch's current device is thread-local: in the reader thread it would
    # be device 0 on every rank) -- only needed when a wav has no .est and the built-in tracker is the opt-in route
    tracker_device = None
    if engine is not None:
        tracker_device = engine.device
    elif os.environ.get("MAGPHASE_EPOCHS", "") == "builtin":
        from .engine import get_engine
        tracker_device = get_engine().device
    # the engine whose planner prepares a batch's launches in the READER thread, one batch ahead of the compute stage
    # (Engine.prepare_analysis: native planners, samples into page-locked memory, no stream touched) -- resolved here for the
    # same reason
    from .engine import get_engine as _ge
    prep_engine = engine or _ge()

    def load(files):
        # the bytes of the wavs and the parsed epoch tracks (text) each come from one native call (a few threads, no GIL)
        wavs = la.read_audio_files_pcm_batch(files)      # 16-bit PCM stays int16: the plan converts it in one pass
        eps = mp._epochs_for_batch(files, device=tracker_device)
        utts, failed = [], []
        for f, w, ep in zip(files, wavs, eps):
            bad = w if isinstance(w, Exception) else (ep if isinstance(ep, Exception) else None)
            if bad is not None:
                failed.append((_tok(f), "%s: %s" % (type(bad).__name__, bad)))
            else:
                utts.append((f, (w[0], w[1], ep[0], ep[1])))
        prepared = {}
        for fs in sorted(set(u[1][1] for u in utts)):   # one launch per sample rate: its host side, here in the reader thread
            try:
                prepared[fs] = prep_engine.prepare_analysis([u[1] for u in utts if u[1][1] == fs], fft_len)
            except (KeyboardInterrupt, SystemExit):
                raise
            except Exception:
                prepared[fs] = None      # the compute stage's own attempt raises (and isolates) what there is to raise
        return utts, failed, prepared

    def compute(loaded):
        # The device results are NOT waited for here: they land in a page-locked ring slot (engine.HostTicket) while this
        # thread plans the next batch; the writer thread waits for the copy and hands the slot back.
        utts, failed, prepared = loaded
        out, tickets = [], []
        for fs in sorted(set(u[1][1] for u in utts)):
            group = [u for u in utts if u[1][1] == fs]

            def analyse(g, whole=len(group), fs=fs):
                # Q7: the reference forwards alpha_phase=b_mag_fbank_mel (False) -- see mp.analysis_for_acoustic_modelling
                kw = dict(fft_len=fft_len, mag_dim=mag_dim, phase_dim=phase_dim, b_const_rate=b_const_rate,
                          alpha_phase=False, engine=engine, as_float32=True)
                if len(g) != whole:   # _isolate's one-by-one retries after a failed batch: plain synchronous results (a
                    return mp.analysis_compressed_batch([u[1] for u in g], **kw)   # ring slot each would exhaust the ring)
                res, ticket = mp.analysis_compressed_batch([u[1] for u in g], async_out=True,
                                                           prepared=prepared.pop(fs, None), **kw)
                tickets.append(ticket)
                return res

            ok, bad = _isolate(group, analyse)
            out.extend((group[i][0], r) for i, r in ok)
            failed = failed + [(_tok(group[i][0]), "%s: %s" % (type(e).__name__, e)) for i, e in bad]
        return out, failed, tickets

    def store(res):
        import time

        results, failed, tickets = res
        t0 = time.perf_counter()
        for t in tickets:
            t.wait()
        t1 = time.perf_counter()
        try:
            _store(results, failed)
        finally:
            for t in tickets:
                t.release()
            if report is not None:   # the writer stage's two parts: waiting for the device's results, writing the files
                report["store_wait_device_s"] = report.get("store_wait_device_s", 0.0) + t1 - t0
                report["store_write_files_s"] = report.get("store_write_files_s", 0.0) + time.perf_counter() - t1

    def _store(results, failed):
        # float32 from the device as it is (what write_featfile stores); all files of the batch in one native call
        paths, bodies, owner = [], [], []
        for f, (m_mag, m_real, m_imag, v_lf0, v_shift, _fs, _n) in results:
            base = os.path.join(out_dir, _tok(f))
            items = [(".mag", m_mag), (".real", m_real), (".imag", m_imag), (".lf0", np.array(v_lf0, "float32"))]
            if not b_const_rate:
                items.append((".shift", np.array(v_shift, "float32")))
            for ext, a in items:
                paths.append(base + ext), bodies.append(a), owner.append(_tok(f))
        bad = {}
        for tok, st in zip(owner, la.write_files_batch(paths, bodies)):
            if st is not None and tok not in bad:
                bad[tok] = "%s: %s" % (type(st).__name__, st)
        failed = failed + list(bad.items())
        if verbose:
            for f, _r in results:
                if _tok(f) not in bad:
                    print("extracted " + _tok(f))
        if report is not None:
            report["done"] = report.get("done", 0) + len(results) - sum(1 for t, _m in failed if t in set(_tok(f) for f, _r in results))
        _record_failures(report, out_dir, failed)
        for tok, msg in failed:
            print("FAILED " + tok + " (" + msg + ")")

    return pipeline(batches(wav_files, batch_utts), load, compute, store, timings=report)


# ----------------------------------------------------------------------------------------------------
# waveform generation (scripts/batch_waveform_generation.py)
# ----------------------------------------------------------------------------------------------------
def generate_waveforms_corpus(in_feats_dir, tokens, out_syn_dir, mag_dim, phase_dim, fs, fft_len=None, pf_type="no",
                              b_const_rate=False, batch_utts=16, engine=None, verbose=True, report=None,
                              noise_mode="reference"):
    """
    mp.synthesis_from_acoustic_modelling (magphase.py:3229-3275) for a list of tokens, `batch_utts` per launch:
    reads <token>.mag/.real/.imag/.lf0, post-filters on the device (pf_type 'magphase' / 'merlin', or 'no'),
    synthesises and writes <token>.wav.
    fs: one sample rate for all tokens (the reference's script), or a dict / callable token -> fs for corpora that mix
    rates (feature files carry no rate): every batch is split by rate, one launch per rate.
    noise_mode: 'reference' draws the aperiodic source from numpy's global RNG exactly like magphase.py:883 (seed it
    and the output is the reference's); 'device' lets the GPU generate it (counter-based, per-utterance seed: not the
    reference's sample values, same statistics; removes the largest host cost of generation).
    A token whose files are missing or malformed does not stop the corpus: crash_file_list_<host>_<pid>.scp.
    Returns the number of batches completed.
    """
    from . import magphase as mp

    lu.mkdir(out_syn_dir)
    if pf_type not in ("no", "magphase", "merlin"):
        raise ValueError("pf_type must be 'no', 'magphase' or 'merlin'")
    from .engine import get_engine
    rng_engine = (engine or get_engine()) if noise_mode == "reference" else None   # resolved in THIS thread (device)
    prep_engine = engine or get_engine()   # its planner prepares a batch's launches in the reader thread (see extraction)
    fs_of = (lambda t: int(fs[t])) if isinstance(fs, dict) else ((lambda t: int(fs(t))) if callable(fs) else (lambda t: int(fs)))

    def load(toks):
        # lu.read_binfile without the float64 copy (the plan uploads float32 anyway); every file of the batch in one call
        exts = ((".mag", mag_dim), (".real", phase_dim), (".imag", phase_dim), (".lf0", 1))
        raw = la.read_files_batch([os.path.join(in_feats_dir, t) + e for t in toks for e, _d in exts])
        utts, failed = [], []
        for k, t in enumerate(toks):
            try:
                rate = fs_of(t)
                mats = []
                for (ext, dim), v in zip(exts, raw[4 * k:4 * k + 4]):
                    if isinstance(v, Exception):
                        raise v
                    if v.size % dim != 0:
                        raise ValueError("Dimension provided not compatible with file size.")
                    mats.append(v.reshape(-1, dim) if dim > 1 else v)
                utts.append((t, rate, tuple(mats)))
            except (KeyboardInterrupt, SystemExit):
                raise
            except Exception as e:
                failed.append((t, "%s: %s" % (type(e).__name__, e)))
        prepared = {}
        for rate in sorted(set(u[1] for u in utts)):   # the launches' host side, here in the reader thread (see extraction)
            try:
                prepared[rate] = prep_engine.prepare_synthesis([u[2] for u in utts if u[1] == rate], rate, fft_len=fft_len,
                                                               b_const_rate=b_const_rate)
            except (KeyboardInterrupt, SystemExit):
                raise
            except Exception:
                prepared[rate] = None
        return utts, failed, prepared

    def compute(loaded):
        # (the device results are not waited for here: see extract_features_corpus.compute)
        utts, failed, prepared = loaded
        out, tickets = [], []
        for rate in sorted(set(u[1] for u in utts)):
            group = [u for u in utts if u[1] == rate]
            def synth(g, whole=len(group), rate=rate):
                kw = {}
                if noise_mode != "reference":   # seed = a hash of the token: the same wav whatever the batching / sharding
                    kw = {"noise_mode": noise_mode, "noise_seeds": [token_seed(u[0]) for u in g]}
                # pcm16_norm: la.write_audio_file's peak normalisation and 16-bit conversion done on the device
                kw.update(fft_len=fft_len, b_const_rate=b_const_rate, b_post_filter=(pf_type if pf_type != "no" else False),
                          engine=engine, pcm16_norm=0.98, defer_rng=(rng_engine is not None))
                if len(g) != whole:   # _isolate's one-by-one retries: synchronous results (see extract_features_corpus)
                    return mp.synthesis_from_compressed_batch([u[2] for u in g], rate, **kw)
                sigs, ticket = mp.synthesis_from_compressed_batch([u[2] for u in g], rate, async_out=True,
                                                                  prepared=prepared.pop(rate, None), **kw)
                tickets.append(ticket)
                return sigs

            ok, bad = _isolate(group, synth, keep_numpy_rng=(noise_mode == "reference"), rng_engine=rng_engine)
            out.extend((group[i][0], rate, sig) for i, sig in ok)
            failed = failed + [(group[i][0], "%s: %s" % (type(e).__name__, e)) for i, e in bad]
        order = {u[0]: k for k, u in enumerate(utts)}
        out.sort(key=lambda r: order[r[0]])
        return out, failed, tickets

    def store(res):
        results, failed, tickets = res
        for t in tickets:
            t.wait()
        try:
            _store(results, failed)
        finally:
            for t in tickets:
                t.release()

    def _store(results, failed):
        paths = [os.path.join(out_syn_dir, t + ".wav") for t, _r, _p in results]
        pcms = [np.ascontiguousarray(p, dtype="<i2") for _t, _r, p in results]
        heads = [la.wav_header_pcm16(p.size, rate) for (_t, rate, _p), p in zip(results, pcms)]
        bad = {}
        for (t, _r, _p), st in zip(results, la.write_files_batch(paths, pcms, heads)):
            if st is not None:
                bad[t] = "%s: %s" % (type(st).__name__, st)
        failed = failed + list(bad.items())
        if verbose:
            for t, _r, _p in results:
                if t not in bad:
                    print("synthesised " + t)
        if report is not None:
            report["done"] = report.get("done", 0) + len(results) - sum(1 for t, _m in failed if t in set(x[0] for x in results))
        _record_failures(report, out_syn_dir, failed)
        for tok, msg in failed:
            print("FAILED " + tok + " (" + msg + ")")

    try:
        return pipeline(batches(tokens, batch_utts), load, compute, store, timings=report)
    finally:
        if rng_engine is not None:   # the noise stream's state was kept on the device between the batches: back to numpy
            rng_engine.mt_sync()
