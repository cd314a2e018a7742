"""-m gpu: the python-3 counterparts of the reference's demos / batch scripts run end to end on the device path."""
import os
import subprocess
import sys
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, cwd):
    out = subprocess.run([sys.executable] + args, cwd=cwd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    return out.stdout


def _wav_len(path):
    with wave.open(path, "rb") as w:
        return w.getnframes(), w.getframerate()


def test_demos_and_batch_scripts(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "demos"))
    import make_demo_data
    wav_dir = tmp_path / "wavs_nat"
    toks = make_demo_data.main(n=2, out_dir=str(wav_dir))
    scp = tmp_path / "file_id.scp"
    assert scp.read_text().split() == toks
    syn_dir = tmp_path / "syn"
    # lossless copy synthesis
    _run([os.path.join(ROOT, "demos", "demo_copy_synthesis_lossless.py"), "--wav", str(wav_dir / "syn_000.wav"),
          "--out-dir", str(syn_dir)], ROOT)
    n_in, fs = _wav_len(str(wav_dir / "syn_000.wav"))
    n_out, fs2 = _wav_len(str(syn_dir / "syn_000_copy_syn_lossless.wav"))
    assert fs == fs2 == 48000 and abs(n_out - n_in) < 2000
    # ... and as one device launch: the same wav up to one 16-bit step (float32 last bits before the rounding)
    one_dir = tmp_path / "syn_one"
    _run([os.path.join(ROOT, "demos", "demo_copy_synthesis_lossless.py"), "--wav", str(wav_dir / "syn_000.wav"),
          "--out-dir", str(one_dir), "--one-launch"], ROOT)
    import wave
    pcm = []
    for d in (syn_dir, one_dir):
        with wave.open(str(d / "syn_000_copy_syn_lossless.wav")) as w:
            pcm.append(np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.int32))
    assert pcm[0].size == pcm[1].size and np.max(np.abs(pcm[0] - pcm[1])) <= 1
    # low-dim copy synthesis
    _run([os.path.join(ROOT, "demos", "demo_copy_synthesis_low_dim.py"), "--wav", str(wav_dir / "syn_000.wav"),
          "--out-dir", str(syn_dir)], ROOT)
    assert os.path.isfile(str(syn_dir / "syn_000_copy_syn_low_dim_mag_dim_60_ph_dim_45_const_rate_0.wav"))
    # batch feature extraction (Q7: phase_dim 10, alpha_phase False -> 44 linear-cepstral bins cut to 10)
    feats = tmp_path / "params_nat"
    _run([os.path.join(ROOT, "scripts", "batch_feature_extraction_for_tts.py"), "--scp", str(scp), "--wav-dir",
          str(wav_dir), "--out-dir", str(feats)], ROOT)
    for tok in toks:
        mag = np.fromfile(str(feats / (tok + ".mag")), dtype=np.float32)
        real = np.fromfile(str(feats / (tok + ".real")), dtype=np.float32)
        lf0 = np.fromfile(str(feats / (tok + ".lf0")), dtype=np.float32)
        shift = np.fromfile(str(feats / (tok + ".shift")), dtype=np.float32)
        assert mag.size == 60 * lf0.size and real.size == 10 * lf0.size and shift.size == lf0.size
    # batch generation from the bundled predicted features
    gen = tmp_path / "gen"
    _run([os.path.join(ROOT, "scripts", "batch_waveform_generation.py"), "--out-dir", str(gen)], ROOT)
    for tok in ("hvd_704", "hvd_705", "hvd_706", "hvd_708"):
        n, fsw = _wav_len(str(gen / (tok + ".wav")))
        assert fsw == 48000 and n > 40000


def test_corpus_batching_equals_the_per_file_calls(tmp_path):
    """iobatch (reader thread / batched kernels / writer thread) writes the same files as one call per utterance."""
    sys.path.insert(0, os.path.join(ROOT, "demos"))
    import make_demo_data
    from magphase_amd import iobatch, magphase as mp
    wav_dir = tmp_path / "wavs"
    toks = make_demo_data.main(n=3, out_dir=str(wav_dir))
    wavs = [str(wav_dir / (t + ".wav")) for t in toks]
    one, many = tmp_path / "one", tmp_path / "many"
    os.makedirs(str(one))
    for w in wavs:
        mp.analysis_for_acoustic_modelling(w, str(one))
    assert iobatch.extract_features_corpus(wavs, str(many), batch_utts=2, verbose=False) == 2
    for t in toks:
        for ext in (".mag", ".real", ".imag", ".lf0", ".shift"):
            a = np.fromfile(str(one / (t + ext)), dtype=np.float32)
            b = np.fromfile(str(many / (t + ext)), dtype=np.float32)
            assert a.size == b.size and np.array_equal(a, b), (t, ext)
    # generation: per-file (seeded) vs batched (same seed: the noise of the utterances is drawn in the same order)
    gen1, gen2 = tmp_path / "gen1", tmp_path / "gen2"
    os.makedirs(str(gen1))
    feats = os.path.join(ROOT, "demos", "data_48k", "params_predicted")
    gtoks = ["hvd_704", "hvd_705", "hvd_706"]
    np.random.seed(11)
    for t in gtoks:
        mp.synthesis_from_acoustic_modelling(feats, t, str(gen1), 60, 45, 48000, pf_type="magphase")
    np.random.seed(11)
    iobatch.generate_waveforms_corpus(feats, gtoks, str(gen2), 60, 45, 48000, pf_type="magphase", batch_utts=3,
                                      verbose=False)
    for t in gtoks:
        with wave.open(str(gen1 / (t + ".wav")), "rb") as w1, wave.open(str(gen2 / (t + ".wav")), "rb") as w2:
            a = np.frombuffer(w1.readframes(w1.getnframes()), dtype=np.int16).astype(np.int32)
            b = np.frombuffer(w2.readframes(w2.getnframes()), dtype=np.int16).astype(np.int32)
        assert a.size == b.size and np.max(np.abs(a - b)) <= 2, t   # 16-bit PCM at 0.98 peak: +-1 LSB of fp32 re-association
    # Merlin post-filter branch runs end to end
    iobatch.generate_waveforms_corpus(feats, gtoks[:1], str(tmp_path / "gen3"), 60, 45, 48000, pf_type="merlin",
                                      verbose=False)
    assert _wav_len(str(tmp_path / "gen3" / "hvd_704.wav"))[0] > 40000


def test_output_ring_wraps_and_small_uploads_wrap_their_arena(tmp_path):
    """The compute stage of iobatch does not wait for the device: results land in a 4-slot page-locked ring that the writer
    thread hands back (engine.HostTicket), small tables go up through an 8 MB page-locked arena that wraps around.  Nine
    batches of one utterance (the ring wraps twice) with a 64 KB arena (it wraps every batch) must write the files of one
    batch of nine; the failed-batch path (one bad utterance -> one-by-one retries, synchronous) still isolates the failure."""
    sys.path.insert(0, os.path.join(ROOT, "demos"))
    import make_demo_data
    from magphase_amd import iobatch
    from magphase_amd.engine import Engine
    wav_dir = tmp_path / "wavs"
    toks = make_demo_data.main(n=9, out_dir=str(wav_dir), dur_s=0.6)
    wavs = [str(wav_dir / (t + ".wav")) for t in toks]
    eng_small = Engine()
    eng_small._ARENA_BYTES = 1 << 16
    eng_small._ARENA_MAX_ITEM = 1 << 13
    a_dir, b_dir = tmp_path / "a", tmp_path / "b"
    assert iobatch.extract_features_corpus(wavs, str(a_dir), batch_utts=9, phase_dim=45, verbose=False) == 1
    assert iobatch.extract_features_corpus(wavs, str(b_dir), batch_utts=1, phase_dim=45, verbose=False,
                                           engine=eng_small) == 9
    for t in toks:
        for ext in (".mag", ".real", ".imag", ".lf0", ".shift"):
            a = np.fromfile(str(a_dir / (t + ext)), dtype=np.float32)
            b = np.fromfile(str(b_dir / (t + ext)), dtype=np.float32)
            assert a.size == b.size and np.array_equal(a, b), (t, ext)
    g1, g2 = tmp_path / "g1", tmp_path / "g2"
    kw = dict(pf_type="magphase", verbose=False, noise_mode="device")   # per-token seeds: independent of the batching
    iobatch.generate_waveforms_corpus(str(a_dir), toks, str(g1), 60, 45, 48000, batch_utts=9, **kw)
    iobatch.generate_waveforms_corpus(str(b_dir), toks, str(g2), 60, 45, 48000, batch_utts=1, engine=eng_small, **kw)
    for t in toks:
        assert (g1 / (t + ".wav")).read_bytes() == (g2 / (t + ".wav")).read_bytes(), t
    # a truncated feature file in the middle of a batch: the other utterances of the batch are written, the token is listed
    bad = toks[4]
    with open(str(a_dir / (bad + ".lf0")), "r+b") as fh:
        fh.truncate(40)
    rep = iobatch.CorpusReport()
    iobatch.generate_waveforms_corpus(str(a_dir), toks, str(tmp_path / "g3"), 60, 45, 48000, batch_utts=3, report=rep, **kw)
    assert [t for t, _m in rep["failed"]] == [bad] and rep["done"] == 8
    for t in toks:
        if t != bad:
            assert (tmp_path / "g3" / (t + ".wav")).read_bytes() == (g1 / (t + ".wav")).read_bytes(), t


def test_pinned_d2h_matches_plain_copy():
    import torch
    from magphase_amd.engine import get_engine
    eng = get_engine()
    t = torch.randn(1000, 2049, device=eng.device)
    got = eng.to_host_f64(t, chunk_bytes=1 << 20)       # 8 chunks
    assert got.dtype == np.float64 and np.array_equal(got, t.cpu().numpy().astype(np.float64))
    v = eng.empty_feats(37, 2049, ld=2112)
    v.copy_(torch.randn(37, 2049, device=eng.device))
    assert np.array_equal(eng.to_host_f64(v), v.cpu().numpy().astype(np.float64))
    assert np.array_equal(eng.to_host_f64(t[:, 0].contiguous()), t[:, 0].cpu().numpy().astype(np.float64))


def test_builtin_epoch_tracker_end_to_end(tmp_path):
    """wav without epochs -> built-in ZFF tracker on the device -> lossless analysis -> synthesis reproduces the signal
    (the half-window pairs of consecutive frames sum to one whatever the epochs are)."""
    from magphase_amd import epochs, libaudio as la, magphase as mp, synthetic as syn
    pcm, pm, voi = syn.make_utterance(7, dur_s=1.5)
    e_dev = epochs.track_epochs(pcm, 48000)                          # batched HIP kernels (csrc/magphase_epochs.hip)
    assert abs(int(e_dev[1].sum()) - int(voi.sum())) <= max(5, int(0.1 * voi.sum()))
    wav = str(tmp_path / "x.wav")
    la.write_audio_file(wav, pcm / 32768.0, 48000, norm=None)
    if la.find_reaper() is None and os.environ.get("MAGPHASE_EPOCHS", "") != "builtin":
        with pytest.raises(RuntimeError):        # no .est, no REAPER, no opt-in: an error, never a silent substitution
            mp.analysis_lossless(wav)
    la.write_audio_file(wav, pcm / 32768.0, 48000, norm=None)
    mp.use_builtin_epoch_tracker()
    try:
        m_mag, m_real, m_imag, v_f0, fs, v_shift = mp.analysis_lossless(wav)
        v_syn = mp.synthesis_from_lossless(m_mag, m_real, m_imag, v_f0, fs)
    finally:
        mp.set_epoch_provider(None)
    x = pcm[:v_syn.size] / 32768.0
    n0, n1 = int(0.05 * fs), min(x.size, v_syn.size) - int(0.05 * fs)
    err = v_syn[n0:n1] - x[n0:n1]
    snr = 10 * np.log10(np.sum(x[n0:n1] ** 2) / np.sum(err ** 2))
    assert snr > 80.0, snr


def test_library_helpers_on_the_device():
    """la.sp_mel_unwarp / la.build_min_phase_from_mag_spec / mp.format_for_modelling (names the reference exposes) run the
    same kernels as the main path; checked against the oracle / against analysis_compressed."""
    from magphase_amd import libaudio as la, magphase as mp, synthetic as syn
    from oracle import magphase_oracle as orc
    rng = np.random.RandomState(4)
    x = rng.randn(37, 60) * 0.7 - 2.0
    for in_type in ("log", "abs"):
        got = la.sp_mel_unwarp(np.exp(x) if in_type == "abs" else x, 2049, alpha=0.77, in_type=in_type)
        ref = orc.sp_mel_unwarp(np.exp(x) if in_type == "abs" else x, 2049, alpha=0.77, in_type=in_type)
        assert got.shape == ref.shape == (37, 2049)
        assert np.max(np.abs(got - ref)) <= 5e-6 * max(1.0, np.max(np.abs(ref)))
    pcm, pm, voi = syn.make_utterance(31, dur_s=0.6)
    sig = syn.pcm_to_float(pcm)
    a = mp.analysis_lossless_from_epochs(sig, 48000, pm, voi)
    mph = la.build_min_phase_from_mag_spec(a[0][:20] + 1e-4)
    ref = orc.build_min_phase_from_mag_spec(a[0][:20] + 1e-4)
    assert mph.shape == ref.shape and np.max(np.abs(np.abs(mph) - np.abs(ref))) <= 1e-5 * np.max(np.abs(ref))
    big = np.abs(ref) > 1e-2 * np.max(np.abs(ref), axis=1, keepdims=True)
    assert np.max(np.abs(mph - ref)[big] / np.abs(ref)[big]) < 2e-3
    f = mp.format_for_modelling(a[0], a[1], a[2], a[3], 48000, mag_dim=60, phase_dim=45)
    c = mp.analysis_compressed_batch([(sig, 48000, pm, voi)], mag_dim=60, phase_dim=45)[0]
    for k in range(3):
        assert f[k].shape == c[k].shape and np.max(np.abs(f[k] - c[k])) < 2e-3   # float64 -> float32 re-quantised input
    assert np.array_equal(f[3], c[3])


def test_device_pcm16_equals_the_host_wav_conversion(tmp_path):
    """mpx_pcm16 (peak normalisation + libsndfile-style rounding on the device) gives exactly the samples
    la.write_audio_file stores -- with and without the output high-pass (float64 / float32 device input)."""
    import wave
    from magphase_amd import libaudio as la, libutils as lu, magphase as mp
    d = os.path.join(ROOT, "demos", "data_48k", "params_predicted")
    utts = [tuple(lu.read_binfile(os.path.join(d, t + e), dim=k) for e, k in ((".mag", 60), (".real", 45), (".imag", 45), (".lf0", 1)))
            for t in ("hvd_704", "hvd_706")]
    for hpf in (True, False):
        np.random.seed(5)
        sigs = mp.synthesis_from_compressed_batch(utts, 48000, b_out_hpf=hpf)
        np.random.seed(5)
        pcm = mp.synthesis_from_compressed_batch(utts, 48000, b_out_hpf=hpf, pcm16_norm=0.98)
        for u in range(2):
            path = str(tmp_path / ("h%d_%d.wav" % (hpf, u)))
            la.write_audio_file(path, sigs[u], 48000)
            with wave.open(path, "rb") as w:
                ref = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
            assert pcm[u].dtype == np.int16 and pcm[u].size == ref.size
            if hpf:      # float64 on both sides: bit-identical
                assert np.array_equal(pcm[u], ref)
            else:        # the host widens the device's float32 PCM first: same values, same result
                assert np.array_equal(pcm[u], ref)
            assert 32000 < int(np.max(np.abs(ref.astype(np.int32)))) <= 32112      # 0.98 * 32767


def test_device_pcm16_on_ragged_lengths_around_the_peak_kernels_blocks():
    """mpx_pcm16's peak is an atomic maximum over blocks of 4 096 samples (round 5): utterances of 1 .. 100 001 samples, a
    silent one among them, float64 and float32 input, against the host's operations (libaudio.py:352-365) sample for sample."""
    import torch
    from magphase_amd.engine import get_engine
    eng = get_engine()
    rng = np.random.RandomState(9)
    lens = [1, 7, 4095, 4096, 4097, 8193, 100001, 300]
    for dt in (np.float64, np.float32):
        sigs = [(rng.uniform(-1, 1, n) * rng.uniform(0.01, 3.0)).astype(dt) for n in lens]
        sigs[5][:] = 0.0                                     # a silent utterance: 0 / 0 -> the bound, as on the host
        sigs[6][77777] = -3.5                                # the peak deep inside a long utterance, negative
        off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        y = torch.from_numpy(np.concatenate(sigs)).to(eng.device)
        got = eng.output_pcm16(y, off, norm=0.98)
        for u, x in enumerate(sigs):
            x64 = x.astype(np.float64)
            with np.errstate(invalid="ignore", divide="ignore"):
                v = 0.98 * x64 / np.max(np.abs(x64))
            ref = np.clip(np.rint(v * 32767.0), -32768, 32767)
            g = got[off[u]:off[u + 1]].astype(np.float64)
            ok = ~np.isnan(ref)
            assert np.array_equal(g[ok], ref[ok]), (dt, u)


# ----------------------------------------------------------------------------------------------------------------------
# prepared launches (Engine.prepare_analysis / prepare_synthesis: native whole-launch planners, planner thread) give the
# generic path's results bit for bit
# ----------------------------------------------------------------------------------------------------------------------
def _prep_utts(n=5, fs=48000):
    from magphase_amd import synthetic as syn
    return [(lambda r: (r[0], fs, r[1], r[2]))(syn.make_utterance(300 + u, dur_s=0.5 + 0.2 * u, fs=fs)) for u in range(n)]


def test_prepared_analysis_equals_generic_path(monkeypatch):
    from magphase_amd import hostplan, magphase as mp
    from magphase_amd.engine import get_engine
    if hostplan.pyhost() is None:
        pytest.skip("_mpx_pyhost not built")
    eng = get_engine()
    utts = _prep_utts()
    kw = dict(mag_dim=60, phase_dim=10, alpha_phase=False, as_float32=True)
    monkeypatch.setenv("MAGPHASE_NATIVE_PREPARE", "0")
    ref = mp.analysis_compressed_batch(utts, **kw)
    ref_ll = mp.analysis_lossless_batch(utts, return_device=True)
    monkeypatch.delenv("MAGPHASE_NATIVE_PREPARE")
    prep = eng.prepare_async("analysis", utts).result()
    assert prep is not None
    for got in (mp.analysis_compressed_batch(utts, prepared=prep, **kw), mp.analysis_compressed_batch(utts, **kw)):
        for a, b in zip(ref, got):
            for x, y in zip(a[:5], b[:5]):
                assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True)
            assert a[5:] == b[5:]
    got_ll = mp.analysis_lossless_batch(utts, return_device=True)
    for a, b in zip(ref_ll, got_ll):
        for x, y in zip(a[:3], b[:3]):
            assert bool((x == y).all())
        assert np.array_equal(a[3], b[3], equal_nan=True) and np.array_equal(a[5], b[5])
    # float64 / float32 samples take the float32 staging of the native path
    utts64 = [(u[0].astype(np.float64) / 32768.0, u[1], u[2], u[3]) for u in utts]
    monkeypatch.setenv("MAGPHASE_NATIVE_PREPARE", "0")
    ref64 = mp.analysis_compressed_batch(utts64, **kw)
    monkeypatch.delenv("MAGPHASE_NATIVE_PREPARE")
    for a, b in zip(ref64, mp.analysis_compressed_batch(utts64, **kw)):
        for x, y in zip(a[:5], b[:5]):
            assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True)
    # a prepared object of another batch is refused
    with pytest.raises(ValueError):
        mp.analysis_compressed_batch(utts[:3], prepared=eng.prepare_analysis(utts), **kw)


@pytest.mark.parametrize("b_const_rate", [False, True])
def test_prepared_synthesis_equals_generic_path(monkeypatch, b_const_rate):
    from magphase_amd import hostplan, magphase as mp
    from magphase_amd.engine import get_engine
    if hostplan.pyhost() is None:
        pytest.skip("_mpx_pyhost not built")
    eng = get_engine()
    feats = [r[:4] for r in mp.analysis_compressed_batch(_prep_utts(4), mag_dim=60, phase_dim=45, b_const_rate=b_const_rate,
                                                         as_float32=True)]
    feats[1] = tuple(np.asarray(x, dtype=np.float64) for x in feats[1])   # one utterance in float64
    kw = dict(b_const_rate=b_const_rate, b_out_hpf=True, b_post_filter=True, pcm16_norm=0.98)
    monkeypatch.setenv("MAGPHASE_NATIVE_PREPARE", "0")
    np.random.seed(5)
    ref = mp.synthesis_from_compressed_batch(feats, 48000, **kw)
    np.random.seed(5)
    ref_f = mp.synthesis_from_compressed_batch(feats, 48000, b_const_rate=b_const_rate)
    monkeypatch.delenv("MAGPHASE_NATIVE_PREPARE")
    prep = eng.prepare_async("synthesis", feats, 48000, b_const_rate=b_const_rate).result()
    assert prep is not None
    np.random.seed(5)
    got = mp.synthesis_from_compressed_batch(feats, 48000, prepared=prep, **kw)
    np.random.seed(5)
    got_f = mp.synthesis_from_compressed_batch(feats, 48000, b_const_rate=b_const_rate)
    for a, b in zip(ref, got):
        assert a.dtype == np.int16 and np.array_equal(a, b)
    for a, b in zip(ref_f, got_f):
        assert np.array_equal(a, b)
    # a prepared object built for other flags is dropped (the constructor prepares again), not misused
    prep2 = eng.prepare_synthesis(feats, 48000, b_const_rate=not b_const_rate)
    if prep2 is not None:
        np.random.seed(5)
        again = mp.synthesis_from_compressed_batch(feats, 48000, prepared=prep2, **kw)
        for a, b in zip(ref, again):
            assert np.array_equal(a, b)


def test_staging_slot_pool_hands_out_ready_slots_first_and_never_blocks_a_pipeline_of_two_batches():
    """Engine._slot_acquire / _slot_release (round 6): six page-locked staging slots -- a generation batch is two launches and
    the planner works one batch ahead, so four are held at once and a fifth / sixth must come without waiting (with three
    the planner thread, the enqueuing thread and the device waited for each other); a slot whose upload event has not completed
    is passed over while another one is free; an exhausted pool returns None to a non-blocking caller."""
    import torch
    from magphase_amd.engine import Engine
    e = Engine()
    n = e._N_SLOTS
    assert n >= 6
    held = [e._slot_acquire(1 << 20, 1 << 16, wait=False) for _ in range(n)]
    assert all(s is not None for s in held) and len({id(s) for s in held}) == n
    assert e._slot_acquire(1 << 20, 1 << 16, wait=False) is None          # exhausted: no blocking, no slot
    assert all(s["stage"].is_pinned() and s["stage"].numel() >= (1 << 20) for s in held)
    # one slot goes back behind an event that cannot complete yet (a long kernel chain is queued in front of it), another one
    # with a completed upload: the next acquire must take the READY one although the pending one was returned last
    a = torch.randn(4096, 4096, device="cuda")
    ready_ev = torch.cuda.Event()
    ready_ev.record()
    torch.cuda.synchronize()
    e._slot_release(held[0], ready_ev)
    for _ in range(40):
        a = a @ a * 1e-4
    pending_ev = torch.cuda.Event()
    pending_ev.record()
    e._slot_release(held[1], pending_ev)
    assert not pending_ev.query()
    got = e._slot_acquire(1 << 20, 1 << 16, wait=False)
    assert got is held[0]
    got2 = e._slot_acquire(1 << 20, 1 << 16, wait=False)                  # only the pending one is left: waited for, not refused
    assert got2 is held[1] and pending_ev.query()
    for s in [got, got2] + held[2:]:
        e._slot_release(s)
    torch.cuda.synchronize()
