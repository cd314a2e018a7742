"""
-m gpu parity tests of the compressed-feature synthesis path (mpx_mel_unwarp, mpx_noise_stats,
mpx_synthesis_compressed_ola through magphase_amd.magphase) against the committed outputs of the real reference
(tests/golden G5: bundled predicted features hvd_704; G8: constant-rate case) and against the CPU oracle.

Tolerances (fp32 device vs fp64 reference): unwarped spectra rel 5e-6 (mag) / abs 2e-6 (real, imag);
noise gains rel 1e-6; resynthesised PCM <= COMP_PCM_TOL of the signal peak.  Noise is the reference's own draw:
np.random.seed(k) then one np.random.uniform(-1, 1, ns_len) (magphase.py:883).
"""
import os
import warnings

import numpy as np
import pytest

from _tol import note, within

pytestmark = pytest.mark.gpu

COMP_PCM_TOL = 2e-6   # measured worst case on MI355X 6.0e-7 of the signal peak (profiles/r03_tolerance_report.json)


@pytest.fixture(scope="module")
def mp():
    from magphase_amd import magphase as m
    return m


@pytest.fixture(scope="module")
def orc():
    from oracle import magphase_oracle as o
    return o


def _hvd704(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_generation_hvd704.npz"))
    mm = g["in_mag"].reshape(-1, 60).astype(np.float64)
    rr = g["in_real"].reshape(-1, 45).astype(np.float64)
    ii = g["in_imag"].reshape(-1, 45).astype(np.float64)
    return g, mm, rr, ii, g["in_lf0"].astype(np.float64)


def test_unwarp_and_noise_gains_match_oracle(orc, golden_dir):
    from magphase_amd.engine import CompressedSynthesisPlan, get_engine
    g, mm, rr, ii, lf = _hvd704(golden_dir)
    np.random.seed(11)
    plan = CompressedSynthesisPlan(get_engine(), [(mm, rr, ii, lf)], 48000)
    plan.run(keep=True)
    np.random.seed(11)
    _, dbg = orc.synthesis_from_compressed(mm, rr, ii, lf, 48000, return_debug=True)
    mag = plan.debug["mag"].cpu().numpy().astype(np.float64)
    real = plan.debug["real"].cpu().numpy().astype(np.float64)
    imag = plan.debug["imag"].cpu().numpy().astype(np.float64)
    assert np.max(np.abs(mag - dbg["m_mag"]) / dbg["m_mag"]) < 5e-6   # fp32 sum of 60 terms in the exponent
    # the phase rows exist where the synthesis reads them (magphase.py:925-931: voiced frames, bins below the end of the
    # periodic / aperiodic crossfade); since round 5 that also holds for variable-rate input (mpx_mel_unwarp_rows with
    # identity row tables).  Voiced frames are checked; the plain form, all frames and bins, below.
    v, n_per = plan.voiced_host, int(plan.n_per)
    assert np.any(v) and 0 < n_per <= 512
    assert np.max(np.abs(real[v, :n_per] - dbg["m_real"][v, :n_per])) < 2e-6
    assert np.max(np.abs(imag[v, :n_per] - dbg["m_imag"][v, :n_per])) < 2e-6
    os.environ["MAGPHASE_UNWARP_ROWS_VAR"] = "0"
    try:
        np.random.seed(11)
        plain = CompressedSynthesisPlan(get_engine(), [(mm, rr, ii, lf)], 48000)
        plain.run(keep=True)
    finally:
        os.environ.pop("MAGPHASE_UNWARP_ROWS_VAR")
    assert not plain.unwarp_rows and plan.unwarp_rows
    assert np.array_equal(plain.debug["mag"].cpu().numpy(), plan.debug["mag"].cpu().numpy())   # the same values bit for bit
    assert np.max(np.abs(plain.debug["real"].cpu().numpy().astype(np.float64) - dbg["m_real"])) < 2e-6
    assert np.max(np.abs(plain.debug["imag"].cpu().numpy().astype(np.float64) - dbg["m_imag"])) < 2e-6
    assert np.array_equal(plain.debug["real"].cpu().numpy()[v, :n_per], plan.debug["real"].cpu().numpy()[v, :n_per])
    g_voi, g_unv = plan.gains[0]
    assert abs(g_voi - dbg["g_voi"]) < 1e-6 * dbg["g_voi"]
    assert abs(g_unv - dbg["g_unv"]) < 1e-6 * dbg["g_unv"]


def test_unwarp_rows_equals_unwarp_then_interpolation(golden_dir):
    """mpx_mel_unwarp_rows (constant -> variable rate folded into the GEMM: two magnitude products interpolated after
    the exp, coefficient rows of the linear phase unwarp interpolated before it) against the plain unwarp of the
    constant-rate rows followed by interp_from_const_to_variable_rate's lerp (magphase.py:2242-2252) in float64."""
    import torch
    from magphase_amd import _lib
    from magphase_amd.engine import CompressedSynthesisPlan, get_engine
    g = np.load(os.path.join(golden_dir, "g8_compressed_analysis.npz"))
    e = get_engine()
    np.random.seed(3)
    plan = CompressedSynthesisPlan(e, [(g["cr45_mag"], g["cr45_real"], g["cr45_imag"], g["cr45_lf0"])], int(g["fs"]),
                                   b_const_rate=True)
    plan.run(keep=True)
    H = plan.fft_len // 2 + 1
    ld = int(e.lib.mpx_spec_ld(H))
    full = [e.empty((plan.n_rows, ld))[:, :H] for _ in range(3)]
    _lib.check(e.lib.mpx_mel_unwarp(e.stream_ptr(), plan.n_rows, H, plan.a_mag.data_ptr(), plan.mag_dim,
                                    plan.u_mag.data_ptr(), full[0].data_ptr(), plan.a_real.data_ptr(),
                                    plan.a_imag.data_ptr(), plan.phase_dim, plan.u_phase.data_ptr(), full[1].data_ptr(),
                                    full[2].data_ptr(), ld), "mpx_mel_unwarp")
    torch.cuda.synchronize()
    # the same through the two-products-per-frame form (tile_first = NULL): bit-identical magnitudes
    alt = [e.empty((plan.total_frames, ld))[:, :H] for _ in range(3)]
    _lib.check(e.lib.mpx_mel_unwarp_rows(e.stream_ptr(), plan.total_frames, H, plan.a_mag.data_ptr(), plan.mag_dim,
                                         plan.u_mag.data_ptr(), alt[0].data_ptr(), plan.a_real.data_ptr(),
                                         plan.a_imag.data_ptr(), plan.phase_dim, plan.u_phase.data_ptr(), alt[1].data_ptr(),
                                         alt[2].data_ptr(), ld, plan.row0.data_ptr(), plan.row1.data_ptr(),
                                         plan.rowt.data_ptr(), 0, None, None, 0), "mpx_mel_unwarp_rows")
    torch.cuda.synchronize()
    voi = plan.voiced.cpu().numpy().astype(bool)          # the plan's unwarp skips phase tiles without a voiced frame
    assert np.any(voi) and np.any(~voi)
    vt = torch.from_numpy(voi).to(alt[0].device)
    assert torch.equal(alt[0], plan.debug["mag"])
    nc = (plan.n_per + 63) // 64 * 64                    # ... and produces the phase rows below the crossfade only
    assert 0 < plan.n_per < H // 2
    for k, name in ((1, "real"), (2, "imag")):
        assert torch.equal(alt[k][vt][:, :nc], plan.debug[name][vt][:, :nc]), name
    r0, r1 = plan.row0.cpu().numpy(), plan.row1.cpu().numpy()
    t = plan.rowt.cpu().numpy().astype(np.float64)[:, None]
    assert plan.total_frames != plan.n_rows and np.any(t > 0)
    for k, name in enumerate(("mag", "real", "imag")):
        x = full[k].cpu().numpy().astype(np.float64)
        want = x[r0] + (x[r1] - x[r0]) * t
        got = plan.debug[name].cpu().numpy().astype(np.float64)
        assert got.shape == want.shape
        if name != "mag":                      # phase rows: voiced frames, bins below the crossfade (the rest is never read)
            got, want = got[voi][:, :nc], want[voi][:, :nc]
        if name == "mag":
            assert np.max(np.abs(got - want) / want) < 3e-7          # one fp32 rounding of the lerp
        else:
            assert np.max(np.abs(got - want)) < 2e-6                 # 45-term fp32 sums in a different association


def test_row_table_form_of_the_synthesis_kernel(golden_dir):
    """mpx_synthesis_compressed_ola with row0 / row1 / row_t (the kernel interpolates between constant-rate spectrum rows
    itself: the C ABI's other form -- the Python side interpolates in the unwarp instead) against the plan's own run."""
    import torch
    from magphase_amd import _lib
    from magphase_amd.engine import CompressedSynthesisPlan, get_engine
    g = np.load(os.path.join(golden_dir, "g8_compressed_analysis.npz"))
    e = get_engine()
    np.random.seed(3)
    plan = CompressedSynthesisPlan(e, [(g["cr45_mag"], g["cr45_real"], g["cr45_imag"], g["cr45_lf0"])], int(g["fs"]),
                                   b_const_rate=True)
    want = plan.run(keep=True).clone()
    N = plan.fft_len
    H = N // 2 + 1
    ld = int(e.lib.mpx_spec_ld(H))
    full = [e.empty((plan.n_rows, ld))[:, :H] for _ in range(3)]
    _lib.check(e.lib.mpx_mel_unwarp(e.stream_ptr(), plan.n_rows, H, plan.a_mag.data_ptr(), plan.mag_dim,
                                    plan.u_mag.data_ptr(), full[0].data_ptr(), plan.a_real.data_ptr(),
                                    plan.a_imag.data_ptr(), plan.phase_dim, plan.u_phase.data_ptr(), full[1].data_ptr(),
                                    full[2].data_ptr(), ld), "mpx_mel_unwarp")
    buf = plan._buffers()
    strips = torch.zeros_like(buf["strips"])
    pcm = torch.zeros_like(want)
    _lib.check(e.lib.mpx_synthesis_compressed_ola(
        e.stream_ptr(), N, e.tables(N).data_ptr(), full[0].data_ptr(), full[1].data_ptr(), full[2].data_ptr(),
        plan.noise.data_ptr(), plan.npos.data_ptr(), plan.nleft.data_ptr(), plan.nright.data_ptr(), plan.wtype.data_ptr(),
        plan.voiced.data_ptr(), buf["inv_gain"].data_ptr(), plan.row0.data_ptr(), plan.row1.data_ptr(), plan.rowt.data_ptr(),
        plan.win_l.data_ptr(), plan.win_r.data_ptr(), plan.pm_rel.data_ptr(), plan.per_v.data_ptr(), plan.ap_v.data_ptr(),
        plan.ap_u.data_ptr(), plan.runs.data_ptr(), plan.n_runs, plan.slot_off.data_ptr(), plan.slot_runs.data_ptr(),
        plan.n_slots, strips.data_ptr(), pcm.data_ptr(), ld, 0), "mpx_synthesis_compressed_ola")
    e.ola_fixup(N, plan, strips, pcm)
    torch.cuda.synchronize()
    a, b = pcm.cpu().numpy().astype(np.float64), want.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(b)) > 1e-3
    # (same kernel, same arithmetic, only the anti-ringing window's source differs: measured 8.9e-8)
    within(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))), 2.5e-7, "COMP_PCM_TOL:row_tables")


def test_generation_from_predicted_features_matches_reference_golden(mp, golden_dir):
    g, mm, rr, ii, lf = _hvd704(golden_dir)
    seed = int(g["seed"])
    pf = mp.post_filter(mm, 48000)
    assert np.max(np.abs(pf - g["pf48"])) < 1e-12
    for hpf in (True, False):
        np.random.seed(seed)
        v = mp.synthesis_from_compressed(pf, rr, ii, lf, 48000, b_out_hpf=hpf)
        ref = g["syn_pf_hpf%d" % int(hpf)]
        assert len(v) == len(ref)
        within((np.max(np.abs(v - ref))) / (np.max(np.abs(ref))), COMP_PCM_TOL, "COMP_PCM_TOL:123")
    np.random.seed(seed)
    v = mp.synthesis_from_compressed(mm, rr, ii, lf, 48000, b_voi_ap_win=False)
    ref = g["syn_nopf_novoiwin"]
    assert len(v) == len(ref)
    within((np.max(np.abs(v - ref))) / (np.max(np.abs(ref))), COMP_PCM_TOL, "COMP_PCM_TOL:128")


@pytest.mark.parametrize("f0", [55.0, 93.0, 94.5])
def test_low_pitch_noise_frames_longer_than_one_staging_tile(mp, orc, golden_dir, f0):
    """k_synth_comp_pair at 12 waves per CU stages a noise frame in tiles of 1024 samples: below 94 Hz at 48 kHz a
    voiced frame (two pitch periods + 1 samples) is longer than that and takes a second tile.  The bundled predicted
    features with their pitch replaced, against the oracle."""
    g, mm, rr, ii, lf = _hvd704(golden_dir)
    lf = np.where(lf > 0.0, np.log(f0), lf)[:60]
    mm, rr, ii = mm[:60], rr[:60], ii[:60]
    assert 2 * int(round(48000 / f0)) + 1 > 1024 or f0 > 94.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.random.seed(3)
        v = mp.synthesis_from_compressed(mm, rr, ii, lf, 48000)
        np.random.seed(3)
        ref = orc.synthesis_from_compressed(mm, rr, ii, lf, 48000)
    assert v.shape == ref.shape
    within(np.max(np.abs(v - ref)) / max(1.0, np.max(np.abs(ref))), COMP_PCM_TOL, "COMP_PCM_TOL:low_pitch")


def test_constant_rate_input_matches_reference_golden(mp, golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_compressed_analysis.npz"))
    np.random.seed(int(g["cr45_seed"]))
    v = mp.synthesis_from_compressed(g["cr45_mag"], g["cr45_real"], g["cr45_imag"], g["cr45_lf0"], int(g["fs"]),
                                     b_const_rate=True, b_out_hpf=False)
    ref = g["cr45_syn"]
    assert len(v) == len(ref)
    within((np.max(np.abs(v - ref))) / (np.max(np.abs(ref))), COMP_PCM_TOL, "COMP_PCM_TOL:138")


def test_16k_and_batch_match_oracle(mp, orc):
    from magphase_amd import synthetic as syn
    feats = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for u in range(3):
            pcm, pm, voi = syn.make_utterance(90 + u, dur_s=0.5 + 0.2 * u, fs=16000)
            r = orc.analysis_compressed_from_epochs(syn.pcm_to_float(pcm), 16000, pm, voi, mag_dim=60, phase_dim=45)
            feats.append(r[:4])
        refs = []
        np.random.seed(21)
        for f in feats:
            refs.append(orc.synthesis_from_compressed(f[0], f[1], f[2], f[3], 16000))
        np.random.seed(21)
        got = mp.synthesis_from_compressed_batch(feats, 16000)
    for v, ref in zip(got, refs):
        assert len(v) == len(ref)
        within((np.max(np.abs(v - ref))) / (np.max(np.abs(ref))), COMP_PCM_TOL, "COMP_PCM_TOL:158")


def test_min_phase_and_linear_branches(mp, orc, golden_dir):
    """per_phase_type='min_phase' (complex-cepstrum minimum phase, libaudio.py:920-934) vs the reference's golden;
    'linear' vs the oracle (the reference's own 'linear' branch raises under numpy 2: complex into a float array)."""
    g, mm, rr, ii, lf = _hvd704(golden_dir)
    np.random.seed(int(g["seed"]))
    v = mp.synthesis_from_compressed(mm, rr, ii, lf, 48000, per_phase_type="min_phase")
    ref = g["syn_nopf_minphase"]
    assert len(v) == len(ref)
    within((np.max(np.abs(v - ref))) / (np.max(np.abs(ref))), COMP_PCM_TOL, "COMP_PCM_TOL:169")
    np.random.seed(4)
    v = mp.synthesis_from_compressed(mm, rr, ii, lf, 48000, per_phase_type="linear", b_out_hpf=False)
    np.random.seed(4)
    ref = orc.synthesis_from_compressed(mm, rr, ii, lf, 48000, per_phase_type="linear", b_out_hpf=False)
    assert len(v) == len(ref)
    within((np.max(np.abs(v - ref))) / (np.max(np.abs(ref))), COMP_PCM_TOL, "COMP_PCM_TOL:175")
    # constant-rate input: the minimum phase is taken of the row-interpolated magnitude (magphase.py:861-865, 937-938)
    g8 = np.load(os.path.join(golden_dir, "g8_compressed_analysis.npz"))
    np.random.seed(8)
    v = mp.synthesis_from_compressed(g8["cr45_mag"], g8["cr45_real"], g8["cr45_imag"], g8["cr45_lf0"], 48000,
                                     b_const_rate=True, per_phase_type="min_phase", b_out_hpf=False)
    np.random.seed(8)
    ref = orc.synthesis_from_compressed(g8["cr45_mag"], g8["cr45_real"], g8["cr45_imag"], g8["cr45_lf0"], 48000,
                                        b_const_rate=True, per_phase_type="min_phase", b_out_hpf=False)
    assert len(v) == len(ref)
    within((np.max(np.abs(v - ref))) / (np.max(np.abs(ref))), COMP_PCM_TOL, "COMP_PCM_TOL:185")


def test_unsupported_branches_and_errors(mp, golden_dir):
    g, mm, rr, ii, lf = _hvd704(golden_dir)
    with pytest.raises(ValueError):
        mp.synthesis_from_compressed(mm, rr, ii, lf, 44100 + 1)   # define_alpha: unsupported rate


# ---------------------------------------------------------------------------------------------------------------
# compressed analysis (mel warp).  The SPTK mcep arithmetic is a restatement -> golden G8 is "oracle-with-our-mcep".
# Tolerances: log-mag mel abs 2e-5 nepers (0.0002 dB), real/imag abs 3e-6, lf0 and shifts exact (host fp64).
# The lossless features feeding the warp come from the float64-transform analysis kernel (magphase_f64.hip): they are
# the correctly rounded float32 values of the reference's float64 features (test_f64_analysis_features_are_correctly_
# rounded below), so what is left is the fp32 log / GEMM of the warp itself: observed 3.3e-5 (mag), 3.7e-6 (real/imag).
# (With the fp32 transform the same outputs were off by 6.6e-4 / 5.6e-4: ~1e-6 of the frame peak of FFT noise on every
# bin is a 1e-3..1e-2 relative error on bins 60-80 dB down, which ln() and the division by |X| pass on.)
# ---------------------------------------------------------------------------------------------------------------
WARP_TOL = 2e-5         # log-mel magnitudes: measured 8.5e-6 on the golden / configs[2] cases (the random sweep of
                        # tests/test_gpu_fuzz.py keeps 1e-4: worst seen in 340 batches 4.1e-5)
WARP_PHASE_TOL = 3e-6   # phase coefficients: measured 8.9e-7


def test_f64_analysis_features_are_correctly_rounded(orc):
    """mpx_analysis_frames_f64: every feature is the float32 nearest to the reference's float64 value (half an ulp),
    at the three transform sizes -- including bins 100 dB below the frame peak."""
    from magphase_amd import synthetic as syn
    from magphase_amd.engine import LosslessAnalysisPlan, get_engine
    eng = get_engine()
    for fs, n_fft in ((48000, None), (16000, None), (8000, 1024)):
        pcm, pm, voi = syn.make_utterance(5, dur_s=1.0, fs=fs)
        x = syn.pcm_to_float(pcm)
        o = orc.analysis_lossless_from_epochs(x, fs, pm, voi, fft_len=n_fft)
        plan = LosslessAnalysisPlan(eng, [(x, fs, pm, voi)], fft_len=n_fft)
        m, r, i = (t.cpu().numpy().astype(np.float64) for t in plan.run(precise=True))
        peak = o[0].max(axis=1, keepdims=True)
        ok = o[0] > 1e-9 * peak
        assert np.max((np.abs(m - o[0]) / np.maximum(o[0], 1e-300))[ok]) <= 6.5e-8      # 2^-24 = 5.96e-8, + the fp64 chain
        assert np.max(np.abs(r - o[1])[ok]) <= 3.5e-8 and np.max(np.abs(i - o[2])[ok]) <= 3.5e-8
        assert np.array_equal(plan.v_f0[0], o[3])


def test_exactly_cancelling_bins_carry_the_references_residue(mp, orc):
    """Found by tools/fuzz_vs_oracle.py: over a stretch of exactly periodic pitch periods (synthetic utterance 447741 at
    44.1 kHz, frames 132-138) the Nyquist bin cancels exactly; numpy's FFT returns 0.0 -- the reference stores (0, 0, 0)
    (magphase.py:466-472) -- or a residue of a few 2^-53, which it normalises to (+-1, 0).  With numpy's own window weights
    (mpx_analysis_frames_f64w) the device's products are the reference's and so is the residue: zero exactly where the
    reference is zero, the same phasor elsewhere.  (Round 3: everything below 2^-45 of the frame was flushed to zero.)"""
    from magphase_amd import synthetic as syn
    from magphase_amd.engine import LosslessAnalysisPlan, get_engine
    fs = 44100
    pcm, pm, voi = syn.make_utterance(447741, dur_s=0.813, fs=fs)
    x = syn.pcm_to_float(pcm)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        o = orc.analysis_lossless_from_epochs(x, fs, pm, voi)
        zero = o[0] == 0.0
        tiny = o[0] <= 1e-13 * o[0].max(axis=1, keepdims=True)
        assert zero.sum() >= 5 and zero[132, 2048]
        plan = LosslessAnalysisPlan(get_engine(), [(x, fs, pm, voi)])
        m, r, i = (t.cpu().numpy() for t in plan.run(precise=True))
        assert np.array_equal(m == 0.0, zero)                        # zero exactly where the reference is, nowhere else
        assert np.all(r[zero] == 0.0) and np.all(i[zero] == 0.0)
        within(max(np.max(np.abs(r[tiny] - o[1][tiny])), np.max(np.abs(i[tiny] - o[2][tiny]))), 1e-6, "F64_RESIDUE_PHASOR:447741")
        oc = orc.analysis_compressed_from_epochs(x, fs, pm, voi, mag_dim=60, phase_dim=45)
        g = mp.analysis_compressed_batch([(x, fs, pm, voi)], mag_dim=60, phase_dim=45)[0]
    within(np.max(np.abs(g[1] - oc[1])), WARP_PHASE_TOL, "WARP_PHASE_TOL:247r")
    within(np.max(np.abs(g[2] - oc[2])), WARP_PHASE_TOL, "WARP_PHASE_TOL:247i")
    within(np.max(np.abs(g[0] - oc[0])), WARP_TOL, "WARP_TOL:248")


def test_f64_analysis_rows_in_use_skips_only_the_phase_rows():
    """mpx_analysis_frames_f64(rows_in_use): frames flagged 0 get their magnitude row only -- the real / imag rows keep
    what the buffers held; everything else is bit-identical to the unflagged launch."""
    import torch
    from magphase_amd import synthetic as syn
    from magphase_amd.engine import LosslessAnalysisPlan, get_engine
    eng = get_engine()
    pcm, pm, voi = syn.make_utterance(9, dur_s=1.0, fs=48000)
    plan = LosslessAnalysisPlan(eng, [(syn.pcm_to_float(pcm), 48000, pm, voi)])
    full = [t.clone() for t in plan.run(precise=True)]
    F = plan.total_frames
    use = (torch.arange(F, device=eng.device) % 3 != 1).float()
    out = tuple(torch.full_like(t, -7.0) for t in full)
    got = plan.run(out=out, precise=True, rows_in_use=use)
    keep = use.bool()
    assert torch.equal(got[0], full[0])
    for k in (1, 2):
        assert torch.equal(got[k][keep], full[k][keep])
        assert bool((got[k][~keep] == -7.0).all()) and int((~keep).sum()) > 50


@pytest.mark.parametrize("tag,kw", [("vr45", dict(phase_dim=45)), ("cr45", dict(phase_dim=45, b_const_rate=True)),
                                     ("q7", dict(phase_dim=10, alpha_phase=False))])
def test_compressed_analysis_matches_golden(mp, golden_dir, tag, kw):
    from magphase_amd import synthetic as syn
    g = np.load(os.path.join(golden_dir, "g8_compressed_analysis.npz"))
    fs = int(g["fs"])
    x = syn.pcm_to_float(g["pcm"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = mp.analysis_compressed_batch([(x, fs, g["pm_sec"], g["voi"])], mag_dim=60, **kw)[0]
    assert r[0].shape == g[tag + "_mag"].shape and r[1].shape == g[tag + "_real"].shape
    within(np.max(np.abs(r[0] - g[tag + "_mag"])), WARP_TOL, "WARP_TOL:283")
    within(np.max(np.abs(r[1] - g[tag + "_real"])), WARP_PHASE_TOL, "WARP_PHASE_TOL:284")
    within(np.max(np.abs(r[2] - g[tag + "_imag"])), WARP_PHASE_TOL, "WARP_PHASE_TOL:285")
    assert np.array_equal(r[3], g[tag + "_lf0"])
    assert np.array_equal(r[4], g[tag + "_shift"])
    assert r[5] == fs and r[6] == 4096


def test_phase_warp_on_variable_rate_rows_equals_warp_of_interpolated_rows(monkeypatch):
    """mpx_mel_warp_rows: the phase streams warped once per variable-rate row and interpolated afterwards against the
    reference's order (rows interpolated, then warped: mpx_mel_warp with row tables).  The warp is linear up to its
    1e-8 e^{-2x} floor term: the two orders agree to fp32 rounding of sums of 2049 terms (WARP_PHASE_TOL / 7); unvoiced constant-rate frames are
    +0 in both, the magnitudes are the same launch."""
    from magphase_amd import synthetic as syn
    from magphase_amd.engine import CompressedAnalysisPlan, get_engine
    eng = get_engine()
    utts = []
    for u, fs in enumerate((48000, 48000, 48000)):
        pcm, pm, voi = syn.make_utterance(40 + u, dur_s=1.5 + 0.5 * u, fs=fs)
        utts.append((pcm, fs, pm, voi))
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MAGPHASE_WARP_PHASE_ROWS", flag)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            plan = CompressedAnalysisPlan(eng, utts, mag_dim=60, phase_dim=45, b_const_rate=True)
        assert plan.phase_on_rows == (flag == "1")
        outs[flag] = [t.cpu().numpy() for t in plan.run()]
        voi = plan.voi.cpu().numpy()
    assert np.array_equal(outs["1"][0], outs["0"][0])
    assert 0.2 < voi.mean() < 0.9
    for k in (1, 2):
        a, b = outs["1"][k], outs["0"][k]
        assert np.all(a[voi == 0] == 0.0) and np.all(b[voi == 0] == 0.0)
        assert np.max(np.abs(a)) <= 1.0
        # both are fp32 sums of 2049 terms of size ~1 in different association: each is ~1e-6 from the float64 value
        assert np.max(np.abs(a - b)) < 3e-6, np.max(np.abs(a - b))
        assert np.sqrt(np.mean((a - b) ** 2)) < 2e-7, np.sqrt(np.mean((a - b) ** 2))


def test_low_dim_copy_synthesis_roundtrip(mp, orc):
    """demo_copy_synthesis_low_dim call sequence on the device path vs the same sequence in the oracle."""
    from magphase_amd import synthetic as syn
    pcm, pm, voi = syn.make_utterance(77, dur_s=0.8, fs=48000)
    x = syn.pcm_to_float(pcm)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = mp.analysis_compressed_batch([(x, 48000, pm, voi)], mag_dim=60, phase_dim=45, b_const_rate=True)[0]
        o = orc.analysis_compressed_from_epochs(x, 48000, pm, voi, mag_dim=60, phase_dim=45, b_const_rate=True)
        np.random.seed(3)
        v = mp.synthesis_from_compressed(a[0], a[1], a[2], a[3], 48000, b_const_rate=True, b_out_hpf=False)
        np.random.seed(3)
        ref = orc.synthesis_from_compressed(o[0], o[1], o[2], o[3], 48000, b_const_rate=True, b_out_hpf=False)
    assert len(v) == len(ref)
    # features differ by ~3e-5 (fp32 log / GEMM of the warp) -> the waveform by ~2e-6 of peak (observed)
    within(np.max(np.abs(a[0] - o[0])), WARP_TOL, "WARP_TOL:338")
    within(np.max(np.abs(a[1] - o[1])), WARP_PHASE_TOL, "WARP_PHASE_TOL:338")
    within((np.max(np.abs(v - ref))) / (np.max(np.abs(ref))), COMP_PCM_TOL, "COMP_PCM_TOL:339")


def test_device_post_filter_and_gains(mp, orc, golden_dir):
    """mpx_post_filter vs the reference's post_filter golden (G6); batched generation with the device post-filter."""
    from magphase_amd.engine import get_engine
    g, mm, rr, ii, lf = _hvd704(golden_dir)
    eng = get_engine()
    for fs, key in ((48000, "pf48"), (16000, "pf16")):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = eng.post_filter(eng.to_device(mm, np.float32), fs).cpu().numpy().astype(np.float64)
        assert np.max(np.abs(out - g[key])) < 5e-6
    np.random.seed(int(g["seed"]))
    v = mp.synthesis_from_compressed_batch([(mm, rr, ii, lf)], 48000, b_post_filter=True)[0]
    ref = g["syn_pf_hpf1"]
    assert len(v) == len(ref)
    within(np.max(np.abs(v - ref)) / np.max(np.abs(ref)), COMP_PCM_TOL, "COMP_PCM_TOL:355")


def test_device_output_hpf_matches_lfilter(mp):
    """mpx_output_hpf (cascade of biquads, blocked float64 scan) vs scipy.signal.lfilter on ragged utterances."""
    from scipy import signal
    from magphase_amd.engine import get_engine
    eng = get_engine()
    rng = np.random.RandomState(2)
    for fs in (48000, 16000):
        # around the scan's units: blocks of 256 samples, 64 x 64 tiles, 64 blocks (16 384 samples) per wave and per carry
        # step, an utterance of several carry steps; (1 023-1 025: the block size of rounds 2-4)
        lens = [1, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 16383, 16384, 16385, 50000, 3000, 240001]
        sigs = [rng.uniform(-1, 1, n).astype(np.float32) for n in lens]
        off = np.concatenate(([0], np.cumsum(lens)))
        y = eng.output_hpf(eng.to_device(np.concatenate(sigs), np.float32), off, fs).cpu().numpy()
        b_, a_ = signal.butter(4, 40 / (fs / 2.0), btype="highpass")
        for u, x in enumerate(sigs):
            ref = signal.lfilter(b_, a_, x.astype(np.float64))
            assert np.max(np.abs(y[off[u]:off[u + 1]] - ref)) <= 1e-6 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("fs", [44100, 22050])
def test_other_sample_rates_match_oracle(mp, orc, fs):
    """fs = 44.1 / 22.05 kHz (alpha 0.76 / 0.65, FFT 4096 / 2048, the reference's 'untuned crossfade' warning): lossless
    and low-dimensional paths against the oracle."""
    from magphase_amd import synthetic as syn
    pcm, pm, voi = syn.make_utterance(90 + fs % 7, dur_s=0.8, fs=fs)
    x = syn.pcm_to_float(pcm)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        al = mp.analysis_lossless_from_epochs(x, fs, pm, voi)
        ol = orc.analysis_lossless_from_epochs(x, fs, pm, voi)
        assert np.array_equal(al[5], ol[5]) and np.array_equal(al[3], ol[3], equal_nan=True)
        peak = np.max(ol[0], axis=1, keepdims=True)
        assert np.max(np.abs(al[0] - ol[0]) / peak) <= 4e-6
        a = mp.analysis_compressed_batch([(x, fs, pm, voi)], mag_dim=60, phase_dim=45)[0]
        o = orc.analysis_compressed_from_epochs(x, fs, pm, voi, mag_dim=60, phase_dim=45)
        for k in range(3):
            assert a[k].shape == o[k].shape
            within(np.max(np.abs(a[k] - o[k])), WARP_TOL if k == 0 else WARP_PHASE_TOL, "WARP_TOL:392/%d" % min(k, 1))
        assert a[6] == o[6] == (4096 if fs == 44100 else 2048)
        np.random.seed(5)
        v = mp.synthesis_from_compressed(o[0], o[1], o[2], o[3], fs)
        np.random.seed(5)
        ref = orc.synthesis_from_compressed(o[0], o[1], o[2], o[3], fs)
    assert v.shape == ref.shape
    within(np.max(np.abs(v - ref)) / max(1.0, np.max(np.abs(ref))), COMP_PCM_TOL, "COMP_PCM_TOL:398")


def test_fbank_unwarp_generation_matches_reference(mp, golden_dir):
    """synthesis_from_compressed(b_fbank_mel=True) on the bundled predicted features vs the reference's own output (G10)."""
    g10 = np.load(os.path.join(golden_dir, "g10_fbank.npz"))
    g, mm, rr, ii, lf = _hvd704(golden_dir)
    np.random.seed(int(g10["seed"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        v = mp.synthesis_from_compressed(mm, rr, ii, lf, 48000, b_fbank_mel=True)
    ref = g10["syn_fbank"]
    assert v.shape == ref.shape
    within(np.max(np.abs(v - ref)) / max(1.0, np.max(np.abs(ref))), COMP_PCM_TOL, "COMP_PCM_TOL:398")


def test_fbank_warp_analysis_matches_reference(mp, golden_dir):
    """format_for_modelling(b_mag_fbank_mel=True): the mel filter bank on the device (mpx_mel_warp_fbank) against the
    reference's own output (G11, pinned); la.log's -1e10 floor for zero magnitudes comes back exactly."""
    from magphase_amd.engine import get_engine
    g = np.load(os.path.join(golden_dir, "g11_fbank_warp.npz"))
    g2 = np.load(os.path.join(golden_dir, "g2_lossless_48k.npz"))
    r = mp.format_for_modelling(g2["mag32"], g2["real32"], g2["imag32"], g2["v_f0"], 48000, mag_dim=60, phase_dim=45,
                                b_mag_fbank_mel=True)
    assert r[0].shape == g["ffm_mag_mel_log"].shape
    within(np.max(np.abs(r[0] - g["ffm_mag_mel_log"])), WARP_TOL, "WARP_TOL:422")
    assert np.array_equal(r[3], g["ffm_lf0"])
    plain = mp.format_for_modelling(g2["mag32"], g2["real32"], g2["imag32"], g2["v_f0"], 48000, mag_dim=60, phase_dim=45)
    assert np.array_equal(r[1], plain[1]) and np.array_equal(r[2], plain[2])      # the phase streams do not change
    e = get_engine()
    for nb, nbins, fs in ((60, 2049, 48000), (60, 1025, 16000)):
        x, y = g["x_%d_%d" % (nb, nbins)], g["y_%d_%d" % (nb, nbins)]
        dev = [e.feats_to_device(a) for a in (x, np.zeros_like(x), np.zeros_like(x))]
        out = e.mel_warp_feats(dev[0], dev[1], dev[2], np.ones(x.shape[0]), fs, nb, 10, b_mag_fbank_mel=True)
        got = e.to_host_f64(out[0])
        floor = y == -1.0e10
        assert np.array_equal(got == -1.0e10, floor)
        within(np.max(np.abs(got[~floor] - y[~floor])), 2e-6, "WARP_TOL:434")   # filter-bank magnitudes: measured 6.7e-7


def test_full_size_config3_constant_rate_post_filter(mp, orc):
    """
    BASELINE configs[2] at full size on the device-resident batch path: 64 x 5 s @ 48 kHz, analysis_compressed (mag 60,
    phase 45, constant 5 ms rate) -> post-filter -> synthesis_from_compressed(b_const_rate=True).
      * all 64 utterances: frame counts of the constant-rate grid, finite features, phase features in [-1, 1] and zero
        in unvoiced frames, output lengths from the reference's slicing rules, finite PCM at a sane level, bit-identical
        PCM on a second run;
      * utterances 0, 31, 63: features, lf0, and the waveform sample by sample against the oracle fed the same noise
        draw (np.random state captured where the batch drew that utterance's noise).
    """
    import torch
    from scipy import signal
    from magphase_amd import synthetic as syn
    from magphase_amd import hostmath as hm
    from magphase_amd.engine import CompressedAnalysisPlan, CompressedSynthesisPlan, get_engine
    eng = get_engine()
    fs = 48000
    utts = []
    for u in range(64):
        pcm, pm, voi = syn.make_utterance(u, dur_s=5.0, fs=fs)
        utts.append((pcm, fs, pm, voi))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        aplan = CompressedAnalysisPlan(eng, utts, mag_dim=60, phase_dim=45, b_const_rate=True)
        feats = [t.cpu().numpy().astype(np.float64) for t in aplan.run()]
        lossless_dev = aplan.lossless.run(precise=True)      # every row (the plan's own run skips unread phase rows)
    assert aplan.total_out_frames > 60000 and all(np.all(np.isfinite(f)) for f in feats)
    assert np.max(np.abs(feats[1])) <= 1.0 and np.max(np.abs(feats[2])) <= 1.0
    sutts, lf0s = [], []
    for u in range(64):
        a, b = int(aplan.out_off[u]), int(aplan.out_off[u + 1])
        n_expected = int(np.ceil(np.cumsum(aplan.lossless.v_shift[u])[-1] / 240.0)) - 1       # Q15: centres at 5 ms steps
        assert b - a == n_expected
        v_f0 = aplan.f0_out[u]
        with np.errstate(divide="ignore"):
            v_lf0 = orc.f0_to_lf0((v_f0 > 0).astype(float) * signal.medfilt(v_f0))
        unv = v_f0 == 0
        assert np.all(feats[1][a:b][unv] == 0.0) and np.all(feats[2][a:b][unv] == 0.0)        # magphase.py:2527-2529
        sutts.append((feats[0][a:b], feats[1][a:b], feats[2][a:b], v_lf0))
        lf0s.append(v_lf0)
    rs = np.random.RandomState(2024)
    np.random.seed(2024)
    splan = CompressedSynthesisPlan(eng, sutts, fs, b_const_rate=True, post_filter=True)
    states = []
    for u in range(64):                       # the batch drew utterance u's noise right here in numpy's global stream
        states.append(rs.get_state())
        rs.uniform(-1, 1, splan.ns_len[u])
    pcm1 = splan.run().cpu().numpy().astype(np.float64)
    pcm2 = splan.run().cpu().numpy().astype(np.float64)
    assert np.array_equal(pcm1, pcm2) and np.all(np.isfinite(pcm1))
    for u in range(64):
        y = pcm1[splan.out_off_host[u]:splan.out_off_host[u + 1]]
        x = utts[u][0].astype(np.float64) / 32768.0
        assert abs(len(y) - len(x)) < 0.02 * len(x)
        r = np.sqrt(np.mean(y ** 2)) / np.sqrt(np.mean(x ** 2))
        assert 0.3 < r < 3.0, (u, r)
    for u in (0, 31, 63):
        pcm, _fs, pm, voi = utts[u]
        x = pcm.astype(np.float64) / 32768.0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = orc.analysis_compressed_from_epochs(x, fs, pm, voi, mag_dim=60, phase_dim=45, b_const_rate=True)
            a, b = int(aplan.out_off[u]), int(aplan.out_off[u + 1])
            assert o[0].shape == (b - a, 60)
            within(np.max(np.abs(feats[0][a:b] - o[0])), WARP_TOL, "WARP_TOL:499")
            # Phase streams, stage by stage AND end to end on every frame -- no allowance, no exclusion (round 4).  The
            # synthetic generator has stretches of exactly constant pitch; over them the Nyquist / DC bin of a frame
            # cancels EXACTLY and numpy's FFT returns 0.0 or a residue of a few 2^-53, which the reference normalises to a
            # unit phasor.  k_analysis_f64 now multiplies by numpy's own np.hanning weights (mpx_analysis_frames_f64w,
            # hostmath.hann_half_table), so its products -- and with them the residue and its sign -- are the reference's;
            # round 3 flushed such bins to zero and this test skipped the 0.1 % of constant-rate frames interpolating from
            # them (errors up to 3.2e-5 there).  (1) the device's LOSSLESS features are the correctly rounded reference
            # values on every bin above the reference FFT's rounding floor (|X| > 1e-9 of the frame peak); (2) the
            # compression of those features equals the oracle's compression of the SAME features on every value; (3) end to
            # end against the oracle, EVERY constant-rate frame agrees to the phase tolerance.
            a0, b0 = int(aplan.lossless.frame_off[u]), int(aplan.lossless.frame_off[u + 1])
            ol = orc.analysis_lossless_from_epochs(x, fs, pm, voi)
            dl = [t[a0:b0].cpu().numpy().astype(np.float64) for t in lossless_dev]
            peak = ol[0].max(axis=1, keepdims=True)
            ok = ol[0] > 1e-9 * peak
            within(np.max((np.abs(dl[0] - ol[0]) / np.maximum(ol[0], 1e-300))[ok]), 6.5e-8, "F64_MAG_REL:cfg2")
            within(np.max(np.abs(dl[1] - ol[1])[ok]), 3.5e-8, "F64_PHASE_ABS:cfg2r")
            within(np.max(np.abs(dl[2] - ol[2])[ok]), 3.5e-8, "F64_PHASE_ABS:cfg2i")
            oc = orc.to_const_rate(dl[0], dl[1], dl[2], ol[3], ol[5], fs)
            of = orc.format_for_modelling(oc[0], oc[1], oc[2], oc[3], fs, mag_dim=60, phase_dim=45)
            within(np.max(np.abs(feats[0][a:b] - of[0])), WARP_TOL, "WARP_TOL:cfg2_same_inputs")
            for k in (1, 2):
                within(np.max(np.abs(feats[k][a:b] - of[k])), WARP_PHASE_TOL, "WARP_PHASE_TOL:cfg2_same_inputs")
            note("configs2_end_to_end_phase_check:excluded_fraction_of_frames", 0.0)
            for k in (1, 2):   # every frame: excluded fraction 0
                within(np.max(np.abs(feats[k][a:b] - o[k])), WARP_PHASE_TOL, "WARP_PHASE_TOL:cfg2_end_to_end_all_frames")
            # the exactly cancelling bins themselves: zero where the reference's are zero, the same unit phasor elsewhere
            tiny = ~ok
            if tiny.any():
                assert np.array_equal(dl[0][tiny] == 0.0, ol[0][tiny] == 0.0)
                within(max(np.max(np.abs(dl[1][tiny] - ol[1][tiny])), np.max(np.abs(dl[2][tiny] - ol[2][tiny]))), 1e-6,
                       "F64_RESIDUE_PHASOR:cfg2")
            assert np.array_equal(lf0s[u], o[3])
            # the oracle on OUR features (so that the waveform check isolates the synthesis side), same noise draw
            np.random.set_state(states[u])
            ref = orc.synthesis_from_compressed(orc.post_filter(sutts[u][0], fs), sutts[u][1], sutts[u][2], sutts[u][3], fs,
                                                b_const_rate=True, b_out_hpf=False)
        y = pcm1[splan.out_off_host[u]:splan.out_off_host[u + 1]]
        assert len(y) == len(ref)
        within((np.max(np.abs(y - ref))) / (np.max(np.abs(ref))), COMP_PCM_TOL, "COMP_PCM_TOL:514")


@pytest.mark.parametrize("fs,mag_dim,phase_dim,alpha_phase,fbank", [(48000, 60, 10, False, False), (48000, 60, 45, None, False),
                                                                   (16000, 60, 45, None, False), (16000, 24, 16, None, False),
                                                                   (48000, 40, 33, None, True), (48000, 64, 48, None, False),
                                                                   (16000, 3, 1, None, False)])
def test_fused_compressed_analysis_matches_oracle_and_staged_path(orc, fs, mag_dim, phase_dim, alpha_phase, fbank):
    """mpx_analysis_compressed_fused (variable frame rate: transform + both warps in one kernel, no lossless features in
    HBM) against the oracle at the staged path's tolerances, against the staged pair k_analysis_f64 -> k_mel_warp_mfma
    (same arithmetic up to the float32 summation order of the GEMM), bit-reproducible, with a frame count that is not a
    multiple of the eight frames of a round, an all-unvoiced utterance and the filter-bank magnitudes."""
    from magphase_amd import synthetic as syn
    from magphase_amd.engine import CompressedAnalysisPlan, get_engine
    eng = get_engine()
    utts = []
    for u in range(4):
        pcm, pm, voi = syn.make_utterance(70 + u, dur_s=0.9 + 0.13 * u, fs=fs)
        if u == 2:
            voi = np.zeros_like(voi)
        utts.append((syn.pcm_to_float(pcm), fs, pm, voi))
    kw = dict(mag_dim=mag_dim, phase_dim=phase_dim, alpha_phase=alpha_phase, b_mag_fbank_mel=fbank)
    old = os.environ.get("MAGPHASE_COMP_FUSED")
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            os.environ["MAGPHASE_COMP_FUSED"] = "1"
            pf = CompressedAnalysisPlan(eng, utts, **kw)
            if pf.lossless.total_frames % 8 == 0:   # the last round must be a partial one: drop one epoch
                x0, f0_, pm0, voi0 = utts[0]
                utts[0] = (x0, f0_, pm0[:-1], voi0[:-1])
                pf = CompressedAnalysisPlan(eng, utts, **kw)
            assert pf.fused and pf.lossless.total_frames % 8 != 0
            a = [t.cpu().numpy() for t in pf.run()]
            a2 = [t.cpu().numpy() for t in pf.run()]
            os.environ["MAGPHASE_COMP_FUSED"] = "0"
            ps = CompressedAnalysisPlan(eng, utts, **kw)
            assert not ps.fused
            b = [t.cpu().numpy() for t in ps.run()]
    finally:
        if old is None:
            os.environ.pop("MAGPHASE_COMP_FUSED", None)
        else:
            os.environ["MAGPHASE_COMP_FUSED"] = old
    for x, y in zip(a, a2):
        assert np.array_equal(x, y)
    floor = b[0] == -1.0e10
    assert np.array_equal(a[0] == -1.0e10, floor)
    within(np.max(np.abs(a[0].astype(np.float64) - b[0])[~floor]), 1.5e-5, "FUSED_VS_STAGED:mag")
    within(max(np.max(np.abs(a[1].astype(np.float64) - b[1])), np.max(np.abs(a[2].astype(np.float64) - b[2]))), 2e-6,
           "FUSED_VS_STAGED:phase")
    for u, (x, _fs, pm, voi) in enumerate(utts):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ol = orc.analysis_lossless_from_epochs(x, fs, pm, voi)
            o = orc.format_for_modelling(ol[0], ol[1], ol[2], ol[3], fs, mag_dim=mag_dim, phase_dim=phase_dim,
                                         alpha_phase=alpha_phase, b_mag_fbank_mel=fbank)
        s0, s1 = int(pf.out_off[u]), int(pf.out_off[u + 1])
        assert o[0].shape == (s1 - s0, mag_dim)
        fl = o[0] == -1.0e10
        assert np.array_equal(a[0][s0:s1] == -1.0e10, fl)
        within(np.max(np.abs(a[0][s0:s1] - o[0])[~fl], initial=0.0), 1e-5, "WARP_TOL:fused")          # measured 3.4e-6 (staged: 8e-6)
        within(max(np.max(np.abs(a[1][s0:s1] - o[1])), np.max(np.abs(a[2][s0:s1] - o[2]))), 2e-6, "WARP_PHASE_TOL:fused")   # measured 6.5e-7
        if u == 2:
            assert np.all(a[1][s0:s1] == 0.0) and np.all(a[2][s0:s1] == 0.0)


def test_float64_front_end_with_windows_longer_than_its_lds_region(orc):
    """Very low pitch / sparse epochs: half windows of 1 400 - 9 600 samples do not fit a wave's window region (928 doubles) and
    take several passes; frames longer than fft_len are truncated like the reference's (magphase.py:311-315).  Fused kernel and
    staged float64 analysis against the oracle."""
    from magphase_amd.engine import CompressedAnalysisPlan, LosslessAnalysisPlan, get_engine
    fs = 48000
    x = np.round(np.random.RandomState(1).uniform(-0.3, 0.3, 60000) * 32768.0) / 32768.0   # 16-bit PCM: exact in float32
    pm = np.array([0.02, 0.05, 0.09, 0.20, 0.2002, 0.26, 0.40, 0.47, 0.48, 0.60, 0.75, 0.95, 1.0, 1.1, 1.2])
    voi = np.ones_like(pm)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ol = orc.analysis_lossless_from_epochs(x, fs, pm, voi)
        o = orc.format_for_modelling(ol[0], ol[1], ol[2], ol[3], fs, mag_dim=60, phase_dim=45)
        p = CompressedAnalysisPlan(get_engine(), [(x, fs, pm, voi)], mag_dim=60, phase_dim=45)
        assert p.fused
        a = [t.cpu().numpy() for t in p.run()]
        lp = LosslessAnalysisPlan(get_engine(), [(x, fs, pm, voi)])
        m, r, i = (t.cpu().numpy() for t in lp.run(precise=True))
    within(np.max(np.abs(a[0] - o[0])), 2.5e-6, "WARP_TOL:long_frames")            # white noise: no weak bins; measured 8.2e-7
    within(max(np.max(np.abs(a[1] - o[1])), np.max(np.abs(a[2] - o[2]))), 8e-7, "WARP_PHASE_TOL:long_frames")   # measured 2.7e-7
    peak = ol[0].max(axis=1, keepdims=True)
    ok = ol[0] > 1e-9 * peak
    within(np.max((np.abs(m - ol[0]) / np.maximum(ol[0], 1e-300))[ok]), 1.2e-7, "F64_MAG_REL:long_frames")
    within(max(np.max(np.abs(r - ol[1])[ok]), np.max(np.abs(i - ol[2])[ok])), 6e-8, "F64_PHASE_ABS:long_frames")


def test_stored_noise_spectra_matches_reference_golden_and_recomputed(mp, orc, golden_dir, monkeypatch):
    """mpx_noise_stats_spectra + mpx_synthesis_compressed_ola_spectra (opt-in, MAGPHASE_NOISE_SPECTRA=store): every noise
    frame is transformed once, its spectrum stored by the statistics launch and loaded by the synthesis launch
    (magphase.py:886-903, :908-976).  Against the reference's golden outputs (variable rate G5, constant rate G8), the
    oracle at low pitch (noise frames longer than one staging tile) and the default (recomputing) pair on a batch."""
    from magphase_amd import engine as em

    g, mm, rr, ii, lf = _hvd704(golden_dir)
    eng = em.get_engine()
    rng = np.random.RandomState(5)
    utts = []
    for u in range(6):
        n = min(mm.shape[0] - 1, 150 + 37 * u)
        a = rng.randint(0, mm.shape[0] - n)
        utts.append((mm[a:a + n], rr[a:a + n], ii[a:a + n], lf[a:a + n]))
    np.random.seed(11)
    plan = em.CompressedSynthesisPlan(eng, utts, 48000, b_const_rate=True, frames_per_run=23)
    base = plan.run().cpu().numpy().astype(np.float64)
    assert plan._buf.get("nspec") is None

    monkeypatch.setenv("MAGPHASE_NOISE_SPECTRA", "store")
    np.random.seed(11)
    plan = em.CompressedSynthesisPlan(eng, utts, 48000, b_const_rate=True, frames_per_run=23)
    got = plan.run().cpu().numpy().astype(np.float64)
    assert plan._buf.get("nspec") is not None     # the stored form really ran
    within(np.max(np.abs(got - base)) / np.max(np.abs(base)), COMP_PCM_TOL, "COMP_PCM_TOL:nspec_vs_recomputed")

    seed = int(g["seed"])
    pf = mp.post_filter(mm, 48000)
    np.random.seed(seed)
    v = mp.synthesis_from_compressed(pf, rr, ii, lf, 48000, b_out_hpf=False)
    ref = g["syn_pf_hpf0"]
    assert len(v) == len(ref)
    within(np.max(np.abs(v - ref)) / np.max(np.abs(ref)), COMP_PCM_TOL, "COMP_PCM_TOL:nspec_var")
    g8 = np.load(os.path.join(golden_dir, "g8_compressed_analysis.npz"))
    np.random.seed(int(g8["cr45_seed"]))
    v = mp.synthesis_from_compressed(g8["cr45_mag"], g8["cr45_real"], g8["cr45_imag"], g8["cr45_lf0"], int(g8["fs"]),
                                     b_const_rate=True, b_out_hpf=False)
    ref = g8["cr45_syn"]
    assert len(v) == len(ref)
    within(np.max(np.abs(v - ref)) / np.max(np.abs(ref)), COMP_PCM_TOL, "COMP_PCM_TOL:nspec_const")
    for f0 in (55.0, 400.0):
        lf2 = np.where(lf > 0.0, np.log(f0), lf)[:80]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            np.random.seed(3)
            v = mp.synthesis_from_compressed(mm[:80], rr[:80], ii[:80], lf2, 48000, b_const_rate=True)
            np.random.seed(3)
            ref = orc.synthesis_from_compressed(mm[:80], rr[:80], ii[:80], lf2, 48000, b_const_rate=True)
        assert v.shape == ref.shape
        within(np.max(np.abs(v - ref)) / max(1.0, np.max(np.abs(ref))), COMP_PCM_TOL, "COMP_PCM_TOL:nspec_f0")
    # the other periodic-phase branches run the kernel's any-crossfade instantiation (all bins may carry a periodic part)
    np.random.seed(seed)
    v = mp.synthesis_from_compressed(mm, rr, ii, lf, 48000, per_phase_type="min_phase")
    ref = g["syn_nopf_minphase"]
    assert len(v) == len(ref)
    within(np.max(np.abs(v - ref)) / np.max(np.abs(ref)), COMP_PCM_TOL, "COMP_PCM_TOL:nspec_minphase")
    np.random.seed(4)
    v = mp.synthesis_from_compressed(mm, rr, ii, lf, 48000, per_phase_type="linear", b_out_hpf=False)
    np.random.seed(4)
    ref = orc.synthesis_from_compressed(mm, rr, ii, lf, 48000, per_phase_type="linear", b_out_hpf=False)
    assert len(v) == len(ref)
    within(np.max(np.abs(v - ref)) / np.max(np.abs(ref)), COMP_PCM_TOL, "COMP_PCM_TOL:nspec_linear")
    # FFT lengths other than 4096 keep the recomputing pair (no stored form there): same call, same result as the oracle
    from magphase_amd import synthetic as syn
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pcm, pm, voi = syn.make_utterance(90, dur_s=0.6, fs=16000)
        f = orc.analysis_compressed_from_epochs(syn.pcm_to_float(pcm), 16000, pm, voi, mag_dim=60, phase_dim=45)[:4]
        np.random.seed(21)
        ref = orc.synthesis_from_compressed(f[0], f[1], f[2], f[3], 16000)
        np.random.seed(21)
        v = mp.synthesis_from_compressed(f[0], f[1], f[2], f[3], 16000)
    assert len(v) == len(ref)
    within(np.max(np.abs(v - ref)) / np.max(np.abs(ref)), COMP_PCM_TOL, "COMP_PCM_TOL:nspec_16k")


@pytest.mark.parametrize("fs,mag_dim,phase_dim,n_utts", [(48000, 60, 45, 12), (16000, 60, 45, 70), (16000, 24, 16, 5),
                                                         (48000, 64, 48, 4), (16000, 3, 1, 4)])
def test_one_kernel_constant_rate_analysis_matches_oracle_and_staged_path(orc, monkeypatch, fs, mag_dim, phase_dim, n_utts):
    """mpx_analysis_compressed_fused_cr + mpx_warp_phase_rows (MAGPHASE_COMP_FUSED_CR=1: transform, interpolation to the 5 ms
    grid and both warps without the lossless rows in HBM) against the oracle at the constant-rate path's tolerances and
    against the staged pair.  The batch holds what the kernel's bookkeeping has to get right: more frames than one round per
    workgroup (the halo row handed from round to round), an utterance with a single voiced frame (rounds without phase
    tiles), a 60 Hz voice (more than
    16 constant-rate frames per window of eight: a second sweep), a 400 Hz voice (frames without a constant-rate frame of
    their own), a one-frame tail (frame counts that are no multiple of eight) and bit-reproducibility."""
    from magphase_amd import synthetic as syn
    from magphase_amd.engine import CompressedAnalysisPlan, get_engine
    eng = get_engine()
    utts = []
    for u in range(n_utts):
        pcm, pm, voi = syn.make_utterance(90 + u, dur_s=0.8 + 0.11 * (u % 7), fs=fs)
        x = syn.pcm_to_float(pcm)
        dur = len(x) / float(fs)
        if u == 1:       # one voiced frame only (none at all: the reference's constant-rate f0 interpolation has nothing to
            voi = np.zeros_like(voi)   # interpolate from and raises, magphase.py:2975-2980)
            voi[len(voi) // 2] = 1.0
        elif u == 2:     # 60 Hz: 3.3 constant-rate frames per pitch period
            pm = np.round(np.arange(0.02, dur - 0.03, 1.0 / 60.0), 6)
            voi = np.ones_like(pm)
        elif u == 3:     # 400 Hz: two frames per 5 ms
            pm = np.round(np.arange(0.01, dur - 0.02, 1.0 / 400.0), 6)
            voi = np.ones_like(pm)
        utts.append((x, fs, pm, voi))
    kw = dict(mag_dim=mag_dim, phase_dim=phase_dim, b_const_rate=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkeypatch.setenv("MAGPHASE_COMP_FUSED_CR", "1")
        pf = CompressedAnalysisPlan(eng, utts, **kw)
        if pf.lossless.total_frames % 8 == 0:
            x0, f0_, pm0, voi0 = utts[0]
            utts[0] = (x0, f0_, pm0[:-1], voi0[:-1])
            pf = CompressedAnalysisPlan(eng, utts, **kw)
        assert pf.fused_cr and not pf.fused and pf.lossless.total_frames % 8 != 0
        a = [t.cpu().numpy() for t in pf.run()]
        a2 = [t.cpu().numpy() for t in pf.run()]
        monkeypatch.setenv("MAGPHASE_COMP_FUSED_CR", "0")
        ps = CompressedAnalysisPlan(eng, utts, **kw)
        assert not ps.fused_cr
        b = [t.cpu().numpy() for t in ps.run()]
    if n_utts >= 12:   # several rounds per workgroup: the halo path is in use
        assert pf.lossless.total_frames > 8 * 256
    for x, y in zip(a, a2):
        assert np.array_equal(x, y)
    assert all(np.all(np.isfinite(x)) for x in a)
    within(np.max(np.abs(a[0].astype(np.float64) - b[0])), 1.5e-5, "ONE_KERNEL_CR_VS_STAGED:mag")
    within(max(np.max(np.abs(a[1].astype(np.float64) - b[1])), np.max(np.abs(a[2].astype(np.float64) - b[2]))), 2e-6,
           "ONE_KERNEL_CR_VS_STAGED:phase")
    voi_c = pf.voi.cpu().numpy()
    assert np.all(a[1][voi_c == 0] == 0.0) and np.all(a[2][voi_c == 0] == 0.0)
    for u in ([0, 1, 2, 3] + ([n_utts - 1] if n_utts > 4 else [])):
        x, _fs, pm, voi = utts[u]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = orc.analysis_compressed_from_epochs(x, fs, pm, voi, mag_dim=mag_dim, phase_dim=phase_dim, b_const_rate=True)
        s0, s1 = int(pf.out_off[u]), int(pf.out_off[u + 1])
        assert o[0].shape == (s1 - s0, mag_dim)
        if u == 3:   # frames of 240 samples with a 4096-point transform: both forms are 3.6e-5 from the oracle on the first
            # coefficient of one frame (measured, round 6; the typical utterance: 2.8e-6 here, 5.5e-6 staged)
            within(np.max(np.abs(a[0][s0:s1] - o[0])), 6e-5, "WARP_TOL:one_kernel_cr_400Hz")
            within(np.max(np.abs(b[0][s0:s1] - o[0])), 6e-5, "WARP_TOL:staged_cr_400Hz")
        else:
            within(np.max(np.abs(a[0][s0:s1] - o[0])), WARP_TOL, "WARP_TOL:one_kernel_cr")
        within(max(np.max(np.abs(a[1][s0:s1] - o[1])), np.max(np.abs(a[2][s0:s1] - o[2]))), WARP_PHASE_TOL,
               "WARP_PHASE_TOL:one_kernel_cr")


def test_one_kernel_constant_rate_analysis_through_the_batch_api(mp, orc, monkeypatch):
    """analysis_compressed_batch(..., b_const_rate=True) with MAGPHASE_COMP_FUSED_CR=1: the caller-level results (features,
    lf0, shifts) of a single short utterance -- fewer frames than one round -- equal the default form's to the kernels'
    tolerances, lf0 / shifts exactly."""
    from magphase_amd import synthetic as syn
    pcm, pm, voi = syn.make_utterance(5, dur_s=0.06, fs=48000)
    x = syn.pcm_to_float(pcm)
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MAGPHASE_COMP_FUSED_CR", flag)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            outs[flag] = mp.analysis_compressed_batch([(x, 48000, pm, voi)], mag_dim=60, phase_dim=45, b_const_rate=True)[0]
    r1, r0 = outs["1"], outs["0"]
    assert r1[0].shape == r0[0].shape and r1[0].shape[0] >= 1
    within(np.max(np.abs(r1[0] - r0[0])), 1.5e-5, "ONE_KERNEL_CR_VS_STAGED:mag")
    within(max(np.max(np.abs(r1[1] - r0[1])), np.max(np.abs(r1[2] - r0[2]))), 2e-6, "ONE_KERNEL_CR_VS_STAGED:phase")
    assert np.array_equal(r1[3], r0[3]) and np.array_equal(r1[4], r0[4])
