"""
Drop-in counterpart of the reference's vocoder API (src/magphase.py) for the analysis/synthesis hot path.

Same function names, positional order, defaults and return arity as the reference
(SURVEY.md section 8b); numpy float64 arrays in and out, float32 feature files on disk.  The per-frame
arithmetic runs in hand-written gfx950 kernels (libmagphase_hip.so via ctypes); the float64
epoch/index arithmetic stays on the host (hostmath.py).  There is no CPU fallback.

Epochs: the reference shells out to the REAPER binary (magphase.py:2875-2876).  REAPER is outside the hot path
(SURVEY.md section 8f #1); epochs come, in this order, from a provider set with ``set_epoch_provider``,
from ``<wav stem>.est`` next to the wav (REAPER text format), from the built-in tracker if MAGPHASE_EPOCHS=builtin
is set (explicit opt-in: it is not REAPER), or from a REAPER binary if one is installed; otherwise RuntimeError.
"""
import os
import warnings

import numpy as np

from . import hostmath as hm
from . import libaudio as la
from . import libutils as lu
from .engine import (CompressedAnalysisPlan, CompressedSynthesisPlan, LosslessAnalysisPlan, LosslessRoundTripPlan,
                     LosslessSynthesisPlan,
                     get_engine)

_epoch_provider = None

_WARN_LONG = ("fft_len (%d) is shorter than the current detected frame length (%d). "
              "This issue is not very critical, but if it occurs often "
              "(e.g., more than 3 times per utterance), please increase de FFT length.")


def set_epoch_provider(fn):
    """fn(wav_file) -> (v_pm_sec, v_voi) or None.  Replaces the REAPER call of magphase.py:2875-2876."""
    global _epoch_provider
    _epoch_provider = fn


def use_builtin_epoch_tracker(device=None):
    """Installs magphase_amd.epochs.track_epochs as the epoch provider (instead of <wav>.est files / REAPER).
    The device is resolved HERE, in the caller's thread: torch's "current device" is thread-local and iobatch calls the
    provider from its reader thread, where it is device 0 on every rank."""
    from . import epochs

    dev = device if device is not None else get_engine().device

    def provider(wav_file):
        v_sig, fs = la.read_audio_file(wav_file)
        return epochs.track_epochs(v_sig, fs, device=dev)

    set_epoch_provider(provider)


def _epochs_for(wav_file, device=None):
    """device: the engine's device for the built-in tracker (MAGPHASE_EPOCHS=builtin); callers that run this in a worker
    thread pass it (see use_builtin_epoch_tracker), None = this thread's current device."""
    if _epoch_provider is not None:
        r = _epoch_provider(wav_file)
        if r is not None:
            return np.asarray(r[0], dtype=np.float64), np.asarray(r[1], dtype=np.float64)
    est = os.path.splitext(wav_file)[0] + ".est"
    if os.path.isfile(est):
        return la.read_est_fast(est)      # == np.loadtxt(est, skiprows=7, usecols=[0, 1]) columns, 5x faster
    if os.environ.get("MAGPHASE_EPOCHS", "") == "builtin":
        # explicit opt-in (or use_builtin_epoch_tracker()): the built-in zero-frequency-filtering tracker
        # (magphase_amd/epochs.py).  Not REAPER: the epochs differ, hence so do the pitch-synchronous features --
        # parity unpinned for this front end, so it is never substituted silently.
        from . import epochs
        v_sig, fs = la.read_audio_file(wav_file)
        return epochs.track_epochs(v_sig, fs, device=device if device is not None else get_engine().device)
    if la.find_reaper() is None:
        raise RuntimeError(
            "no epochs for %s: neither an epoch provider (set_epoch_provider), nor %s, nor a REAPER binary.  Features "
            "are pitch-synchronous, so the epoch source is part of the result: opt into the built-in tracker "
            "explicitly with MAGPHASE_EPOCHS=builtin or magphase.use_builtin_epoch_tracker() (not REAPER: parity "
            "unpinned)." % (wav_file, est))
    est_tmp = lu.ins_pid("temp.est")
    la.reaper(wav_file, est_tmp)
    try:
        m = np.atleast_2d(np.loadtxt(est_tmp, skiprows=7, usecols=[0, 1]))
    finally:
        if os.path.exists(est_tmp):
            os.remove(est_tmp)
    return m[:, 0], m[:, 1]


def _epochs_for_batch(wav_files, device=None):
    """_epochs_for over a list: [(v_pm_sec, v_voi) | Exception].  Without an epoch provider the .est files next to the
    wavs are parsed in one native call (la.read_est_batch); a wav without one takes _epochs_for's remaining routes
    (device: see _epochs_for -- iobatch's reader thread passes its engine's device)."""
    if _epoch_provider is None:
        res = la.read_est_batch([os.path.splitext(f)[0] + ".est" for f in wav_files])
    else:
        res = [FileNotFoundError()] * len(wav_files)
    out = []
    for f, r in zip(wav_files, res):
        if isinstance(r, FileNotFoundError):
            try:
                r = _epochs_for(f, device=device)
            except (KeyboardInterrupt, SystemExit):
                raise
            except Exception as e:
                r = e
        out.append(r)
    return out


# ======================================================================================================
# helper names of the reference's module that Merlin-side callers import (array in, float64 array out)
# ======================================================================================================
def windowing(v_sig, v_pm, win_func=np.hanning):
    """
    magphase.py:74-119: pitch-synchronous frames sig[pm_{f-1} .. pm_{f+1}] times the non-symmetric window of their two
    half lengths.  win_func: a window function, a list of them (one per frame, Q11) or None (no window).
    Returns (l_frames, v_lens, v_pm_plus, v_shift, v_rights) -- float64 frames on the host, like the reference (the
    batched device form of the same step is the front end of mpx_analysis_frames).
    """
    v_sig = np.asarray(v_sig)
    pm, left, right = hm.frame_bounds(v_pm, np.size(v_sig))
    v_pm_plus = np.hstack((0, pm, np.size(v_sig) - 1))
    l_frames = []
    for f in range(pm.size):
        v_frm = v_sig[v_pm_plus[f]:v_pm_plus[f + 2] + 1]
        fn = win_func[f] if isinstance(win_func, list) else win_func
        if fn is not None:
            v_frm = v_frm * la.gen_non_symmetric_win(left[f], right[f], fn)
        l_frames.append(v_frm)
    v_lens = np.array([len(x) for x in l_frames], dtype=int)
    return l_frames, v_lens, v_pm_plus, left.astype(int), right.astype(int)


def ola(m_frm, v_pm, win_func=None, device=None):
    """
    magphase.py:34-62 (PSOLA): frame i added at pm[i] - pm[0], head and tail trimmed so that frame centres land on the
    epochs.  Frames of 1024 / 2048 / 4096 samples without a window go through the device gather (mpx_ola_gather: the
    reference's ascending summation order, float32); anything else is summed on the host in float64.  With win_func the
    frames are multiplied by the centred anti-ringing window first -- IN PLACE, like the reference (magphase.py:48).
    device=False (or MAGPHASE_OLA_HOST=1) keeps every call on the float64 host sum, the reference's own arithmetic.
    """
    if device is None:
        device = os.environ.get("MAGPHASE_OLA_HOST", "0") != "1"
    v_pm = np.asarray(v_pm).astype(int)
    nfrms, frmlen = np.shape(m_frm)
    rel, start, out_len = hm.ola_plan(v_pm, frmlen)
    if win_func is not None:
        v_shift = np.append(la.pm_to_shift(v_pm), v_pm[-1] - v_pm[-2] if nfrms > 1 else v_pm[-1])
        for i in range(nfrms):
            m_frm[i, :] *= la.gen_centr_win(v_shift[i], v_shift[i + 1], frmlen, win_func=win_func)
    if device and win_func is None and frmlen in (1024, 2048, 4096) and nfrms > 0 and out_len > 0:
        e = get_engine()
        frames = e.to_device(np.ascontiguousarray(m_frm, dtype=np.float32), np.float32)
        t = e.to_device_packed([("fo", np.array([0, nfrms]), np.int32), ("rel", rel, np.int32),
                                ("st", np.array([start]), np.int32), ("oo", np.array([0, out_len]), np.int64)])
        out = e.ola_gather(frmlen, frames, t["fo"], t["rel"], t["st"], t["oo"], out_len, out_len)
        return e.to_host_f64(out)
    v_sig = np.zeros(int(v_pm[-1]) + frmlen)
    for i in range(nfrms):
        v_sig[rel[i]:rel[i] + frmlen] += m_frm[i, :]
    return v_sig[start:start + out_len]


def get_shifts_and_frm_locs_from_const_shifts(v_shift_c_rate, frm_rate_ms, fs, interp_type='linear'):
    """magphase.py:1426-1449 (Q16): the serial backward scan from the last constant-rate centre, in the library's host
    function (scipy interp1d's float64 operation sequence, bit-identical: golden G7)."""
    from .engine import _const_to_variable_scan, _const_to_variable_scan_scipy

    if interp_type != 'linear':
        return _const_to_variable_scan_scipy(v_shift_c_rate, frm_rate_ms, fs)
    return _const_to_variable_scan(v_shift_c_rate, frm_rate_ms, fs)


def interp_from_variable_to_const_frm_rate(m_data, v_pm_smpls, const_rate_ms, fs, interp_type='linear'):
    """magphase.py:2219-2239 (Q15): rows at the epochs -> rows on the constant-rate grid, float64 on the host (scipy's
    interp1d like the reference; the batched device form is the operand load of mpx_mel_warp)."""
    from scipy import interpolate

    m_data = np.asarray(m_data, dtype=np.float64)
    squeeze = m_data.ndim == 1
    m2 = m_data[:, None] if squeeze else m_data
    v_pm_smpls = np.asarray(v_pm_smpls)
    grid = np.arange(fs * const_rate_ms / 1000, v_pm_smpls[-1], fs * const_rate_ms / 1000)
    if v_pm_smpls[0] > 0:   # the first row is held back to sample 0
        f = interpolate.interp1d(np.r_[0, v_pm_smpls], np.vstack((m2[0, :], m2)), axis=0, kind=interp_type)
    else:
        f = interpolate.interp1d(v_pm_smpls, m2, axis=0, kind=interp_type)
    out = f(grid)
    return out[:, 0] if squeeze else out


def interp_from_const_to_variable_rate(m_data, v_frm_locs_smpls, frm_rate_ms, fs, interp_type='linear'):
    """magphase.py:2242-2252: rows on the constant-rate grid -> rows at the frame locations (host float64)."""
    from scipy import interpolate

    m_data = np.asarray(m_data, dtype=np.float64)
    centres = (fs * frm_rate_ms / 1000) * np.arange(1, np.size(m_data, 0) + 1)
    return interpolate.interp1d(centres, m_data, axis=0, kind=interp_type)(v_frm_locs_smpls)


# constants (magphase.py:3279-3317)
define_alpha = hm.define_alpha
define_fft_len = hm.define_fft_len
define_crossfade_params = hm.define_crossfade_params
shift_to_f0_raw = hm.shift_to_f0
f0_to_shift = hm.f0_to_shift


def write_featfile(m_data, out_dir, filename):
    """magphase.py:2787-2791."""
    lu.write_binfile(m_data, os.path.join(out_dir, filename))


# ======================================================================================================
# lossless analysis
# ======================================================================================================
def analysis_lossless_batch(utts, fft_len=None, engine=None, return_device=False, copy=True):
    """
    Batched magphase.py:2869-2906 for utterances that already have epochs.
    utts: list of (v_sig, fs, v_pm_sec, v_voi).  Returns a list of
    (m_mag, m_real, m_imag, v_f0, fs, v_shift) in float64 numpy (or device tensors if return_device).
    copy=True (default): every utterance owns its arrays, like the reference's (independent, freeing one frees its
    memory).  copy=False: the utterances' matrices are ROW VIEWS of three arrays that hold the whole batch -- no second
    pass over the data (0.7 GB for 16 utterances), but keeping one utterance alive keeps the batch alive and in-place
    edits are made in the shared arrays; bench.py and iobatch use it.
    """
    engine = engine or get_engine()
    plan = LosslessAnalysisPlan(engine, utts, fft_len=fft_len)
    for lens in plan.long_frame_lens:
        for n in lens:  # Q19: truncation warns, it does not raise (magphase.py:311-315)
            warnings.warn(_WARN_LONG % (plan.fft_len, n))
    mag, real, imag = plan.run()
    if not return_device:   # one pinned, chunked D2H per stream for the whole batch (engine.to_host_f64)
        h_feats = tuple(engine.to_host_f64_many([mag, real, imag]))
    out = []
    for u in range(len(utts)):
        a, b = int(plan.frame_off[u]), int(plan.frame_off[u + 1])
        if return_device:
            feats = (mag[a:b], real[a:b], imag[a:b])
        else:
            # the utterance's rows of the batch's arrays (disjoint views: no second pass over the 0.7 GB a 16-utterance
            # batch returns; a one-utterance batch is its own array)
            feats = h_feats if len(utts) == 1 else tuple((h[a:b].copy() if copy else h[a:b]) for h in h_feats)
        out.append(feats + (plan.v_f0[u], plan.fs[u], plan.v_shift[u].astype(int)))
    return out


def analysis_lossless_from_epochs(v_sig, fs, v_pm_sec, v_voi, fft_len=None):
    """magphase.py:2869-2906 from the point where the epochs have been read (array interface)."""
    return analysis_lossless_batch([(v_sig, fs, v_pm_sec, v_voi)], fft_len=fft_len)[0]


def analysis_lossless(wav_file, fft_len=None, out_dir=None):
    """magphase.py:2869-2906."""
    v_sig, fs = la.read_audio_file(wav_file)
    v_pm_sec, v_voi = _epochs_for(wav_file)
    m_mag, m_real, m_imag, v_f0, fs, v_shift = analysis_lossless_from_epochs(v_sig, fs, v_pm_sec, v_voi, fft_len)
    if type(out_dir) is str:
        file_id = os.path.basename(wav_file).split(".")[0]
        write_featfile(m_mag, out_dir, file_id + ".mag")
        write_featfile(m_real, out_dir, file_id + ".real")
        write_featfile(m_imag, out_dir, file_id + ".imag")
        write_featfile(v_f0, out_dir, file_id + ".f0")
        write_featfile(v_shift, out_dir, file_id + ".shift")
        return
    return m_mag, m_real, m_imag, v_f0, fs, v_shift


# ======================================================================================================
# lossless synthesis
# ======================================================================================================
def synthesis_from_lossless_batch(feats, engine=None):
    """
    Batched magphase.py:1759-1776.  feats: list of (m_mag, m_real, m_imag, v_f0, fs), all with the same
    number of bins.  Returns a list of float64 numpy signals.
    """
    engine = engine or get_engine()
    torch = __import__("torch")
    H = int(np.shape(feats[0][0])[1])
    fft_len = 2 * (H - 1)
    plan = LosslessSynthesisPlan(engine, [f[3] for f in feats], [f[4] for f in feats], fft_len)
    rows = np.concatenate(([0], np.cumsum([int(np.shape(f[0])[0]) for f in feats])))
    cat = []
    for k in range(3):   # one device matrix per stream (engine.empty_feats)
        if len(feats) == 1 and torch.is_tensor(feats[0][k]) and feats[0][k].dtype == torch.float32:
            cat.append(feats[0][k])
            continue
        if not any(torch.is_tensor(f[k]) for f in feats):   # host arrays: narrowed into pinned staging, one DMA
            buf = engine.feats_cat_to_device([f[k] for f in feats], H)
            if buf is not None:
                cat.append(buf)
                continue
        buf = engine.empty_feats(int(rows[-1]), H)
        for u, f in enumerate(feats):
            part = f[k] if torch.is_tensor(f[k]) else torch.from_numpy(np.ascontiguousarray(f[k], dtype=np.float32))
            buf[int(rows[u]):int(rows[u + 1])].copy_(part)
        cat.append(buf)
    pcm = engine.to_host_f64(plan.run(cat[0], cat[1], cat[2]))
    return [pcm[plan.out_off_host[u]:plan.out_off_host[u + 1]] for u in range(len(feats))]


def synthesis_from_lossless(m_mag, m_real, m_imag, v_f0, fs):
    """magphase.py:1759-1776."""
    return synthesis_from_lossless_batch([(m_mag, m_real, m_imag, v_f0, fs)])[0]


def copy_synthesis_lossless_batch(utts, fft_len=None, engine=None, return_device=False, with_feats=True):
    """
    analysis_lossless followed by synthesis_from_lossless on the same frames (magphase.py:2869-2906, :1759-1776: what
    demos/demo_copy_synthesis_lossless.py:44-50 does per file) for a batch, as ONE device launch
    (mpx_roundtrip_lossless_ola): the feature rows are written and the waveform is built from them without reading them
    back.  utts: list of (v_sig, fs, v_pm_sec, v_voi), one fft_len per batch.  Returns a list of
    ((m_mag, m_real, m_imag, v_f0, fs, v_shift), v_syn_sig): analysis_lossless' tuple and synthesis_from_lossless' signal,
    float64 numpy (device tensors with return_device); with_feats=False skips the download of the three matrices
    (None in their place) when only the waveform is wanted.
    """
    engine = engine or get_engine()
    if not utts:
        return []
    plan = LosslessRoundTripPlan(engine, utts, fft_len=fft_len)
    a = plan.analysis
    for lens in a.long_frame_lens:
        for n in lens:  # Q19: truncation warns, it does not raise (magphase.py:311-315)
            warnings.warn(_WARN_LONG % (plan.fft_len, n))
    (mag, real, imag), pcm = plan.run()
    if not return_device:
        h_feats = tuple(engine.to_host_f64_many([mag, real, imag])) if with_feats else None
        h_pcm = engine.to_host_f64(pcm)
    out = []
    for u in range(len(utts)):
        fa, fb = int(a.frame_off[u]), int(a.frame_off[u + 1])
        oa, ob = int(plan.out_off_host[u]), int(plan.out_off_host[u + 1])
        if return_device:
            feats, sig = (mag[fa:fb], real[fa:fb], imag[fa:fb]), pcm[oa:ob]
        else:
            feats = tuple(h[fa:fb].copy() for h in h_feats) if with_feats else (None, None, None)
            sig = h_pcm[oa:ob]
        out.append((feats + (a.v_f0[u], a.fs[u], a.v_shift[u].astype(int)), sig))
    return out


def copy_synthesis_lossless(wav_file, fft_len=None):
    """analysis_lossless(wav_file) + synthesis_from_lossless of its result in one launch: returns
    ((m_mag, m_real, m_imag, v_f0, fs, v_shift), v_syn_sig)."""
    v_sig, fs = la.read_audio_file(wav_file)
    v_pm_sec, v_voi = _epochs_for(wav_file)
    return copy_synthesis_lossless_batch([(v_sig, fs, v_pm_sec, v_voi)], fft_len=fft_len)[0]


# ======================================================================================================
# compressed-feature synthesis
# ======================================================================================================
def post_filter(m_mag_mel_log, fs, av_len_at_zero=None, av_len_at_nyq=None, boost_at_zero=None, boost_at_nyq=None):
    """
    magphase.py:2300-2378 (Q20).  [F x 60] float64 on the host, vectorised over frames: 0.5 KB per frame, not worth
    a kernel launch.  Same defaults, warnings and ValueError as the reference.
    """
    m_mag_mel_log = np.asarray(m_mag_mel_log, dtype=np.float64)
    nfrms, mag_dim = m_mag_mel_log.shape
    if mag_dim != 60:
        warnings.warn('Post-filter: It has been only tested with 60 dimensional mag data. '
                      'If you use another dimension, the result may be suboptimal.')
    opts = [av_len_at_zero, av_len_at_nyq, boost_at_zero, boost_at_nyq]
    if fs == 48000:
        defaults = [hm.round_to_int(11.0 * (mag_dim / 60.0)), hm.round_to_int(3.0 * (mag_dim / 60.0)), 1.8, 2.0]
    elif fs == 16000:
        if any(o is None for o in opts):
            warnings.warn('Post-filter: The default parameters for 16kHz sample rate have not being tunned.')
        defaults = [hm.round_to_int(9.0 * (mag_dim / 60.0)), hm.round_to_int(12.0 * (mag_dim / 60.0)), 2.0, 1.6]
    else:
        if any(o is None for o in opts):
            raise ValueError('Post-filter: It has only been tested with 16kHz and 48kHz sample rates.'
                             '\nProvide your own values for the options: av_len_at_zero, av_len_at_nyq, '
                             'boost_at_zero,\nboost_at_nyq if you use another sample rate')
        defaults = opts
    av0, avn, b0, bn = [d if o is None else o for o, d in zip(opts, defaults)]
    v_nx = np.arange(np.floor(av0 / 2), mag_dim - np.floor(avn / 2)).astype(int)
    v_lens = (2 * np.ceil(np.linspace(av0, avn, v_nx.size) / 2) - 1).astype(int)
    half = v_lens // 2
    m_ave = np.zeros((nfrms, mag_dim))
    for j, nxb in enumerate(v_nx):
        m_ave[:, nxb] = np.mean(m_mag_mel_log[:, nxb - half[j]:nxb + half[j] + 1], axis=1)
    m_ave[:, :v_nx[0]] = m_ave[:, [v_nx[0]]]
    m_ave[:, v_nx[-1]:] = m_ave[:, [v_nx[-1]]]
    m_enh = (m_mag_mel_log - m_ave) * np.linspace(b0, bn, mag_dim)[None, :] + m_ave
    m_enh[:, 0] = m_mag_mel_log[:, 0]
    m_enh[:, -1] = m_mag_mel_log[:, -1]
    return m_enh


def post_filter_merlin(m_mag_mel_log, fs, pf_coef=1.4):
    """
    magphase.py:3375-3465: Merlin / HTS style formant enhancement of the log mel magnitudes.  The reference pipes the
    frames through nine SPTK-3.9 binaries; the same arithmetic is restated here on the host in float64 with a float32
    rounding at every pipe boundary (SPTK's stream format) -- **parity unpinned**: neither SPTK nor any reference
    output of this branch exists in the build container, the checks in tests/test_post_filter_merlin.py are
    known-answer properties (pf_coef = 1 is the identity of the cepstral round trip; the frame energy r0 is kept).
      mcep   = rceps(mag, 'log', 'compact')                           -> temp.mcep (float32)
      r0     = c2acr_0(freqt_{alpha->0, 2047}(mcep)),  p_r0 = the same of mcep * lifter, lifter = (1, 1, pf, pf, ...)
      b      = mc2b(mcep * lifter, alpha);  b0 += ln(r0 / p_r0) / 2;  mcep_pf = b2mc(b, alpha)
      out    = cosine-matrix log spectrum of mcep_pf on mag_dim points (alpha = 0), NaN -> la.MAGIC
    Host-side on purpose: [F x 60] per utterance, a few matrix products.
    """
    fft_len = 4096
    minph_ord = fft_len // 2 - 1
    alpha = define_alpha(fs)
    ncoeffs = np.shape(m_mag_mel_log)[1]
    m_mcep = hm._f32(hm.rceps_compact(m_mag_mel_log))
    v_lifter = hm._f32(np.concatenate(([1.0, 1.0], np.full(ncoeffs - 2, float("%1.2f" % pf_coef)))))
    m_mcep_w = hm._f32(m_mcep * v_lifter)                                    # vopr -m
    v_r0 = hm._f32(hm.sptk_c2acr_r0(hm._f32(hm.sptk_freqt(m_mcep, minph_ord, alpha)), fft_len))
    v_p_r0 = hm._f32(hm.sptk_c2acr_r0(hm._f32(hm.sptk_freqt(m_mcep_w, minph_ord, alpha)), fft_len))
    m_b = hm._f32(hm.sptk_mc2b(m_mcep_w, alpha))
    with np.errstate(divide="ignore", invalid="ignore"):
        v_p_b0 = hm._f32(hm._f32(np.log(hm._f32(v_r0 / v_p_r0)) / 2.0) + m_b[:, 0])   # vopr -d | sopr -LN -d 2 | vopr -a
    m_b[:, 0] = v_p_b0                                                       # bcp 1..order | merge
    m_mcep_pf = hm._f32(hm.sptk_b2mc(m_b, alpha))
    m_out = hm.cos_matrix_log_spectrum(m_mcep_pf, ncoeffs)
    m_out[np.isnan(m_out)] = la.MAGIC
    return m_out


def post_filter_merlin_device(m_mag_mel_log, fs, pf_coef=1.4, engine=None):
    """post_filter_merlin on the device (mpx_post_filter_merlin): the same chain in float32 for all frames at once --
    what iobatch.generate_waveforms_corpus(pf_type='merlin') and synthesis_from_compressed_batch(b_post_filter='merlin')
    run.  Host array in, float64 array out; differs from the host form by float32 rounding (tests: -m gpu)."""
    engine = engine or get_engine()
    x = engine.to_device(np.atleast_2d(np.asarray(m_mag_mel_log)), np.float32)
    return engine.to_host_f64(engine.post_filter_merlin(x, fs, pf_coef=pf_coef))


def _output_hpf(v_syn_sig, fs):
    """magphase.py:981-995: 4th-order Butterworth high-pass at 40 Hz, float64 on the host (poles at |z|~0.997)."""
    from scipy import signal

    v_b, v_a = signal.butter(4, 40 / (fs / 2.0), btype='highpass')
    return signal.lfilter(v_b, v_a, v_syn_sig)


def synthesis_from_compressed_batch(utts, fs, fft_len=None, b_voi_ap_win=True, b_const_rate=False, alpha_phase=None,
                                    b_out_hpf=True, noise=None, engine=None, per_phase_type='magphase',
                                    b_post_filter=False, b_fbank_mel=False, noise_mode='reference', noise_seeds=None,
                                    pcm16_norm=False, async_out=False, defer_rng=False, prepared=None):
    """Batched synthesis_from_compressed; utts: list of (m_mag_mel_log, m_real_mel, m_imag_mel, v_lf0).
    defer_rng: reference noise only -- numpy's advanced generator state stays on the device between calls; the caller owes
    engine.mt_sync() before numpy's global generator is used again (iobatch does this for a corpus run).
    async_out (with pcm16_norm): returns (signals, ticket) -- the int16 signals are views of a page-locked buffer the
    device is still copying into; ticket.wait() before reading them, ticket.release() when done (engine.HostTicket).
    b_post_filter: apply a post-filter to the log-mel magnitudes on the device first: True / 'magphase' = the MagPhase
    post-filter (pf_type='magphase'), 'merlin' = the Merlin-style one (pf_type='merlin', mpx_post_filter_merlin).
    noise_mode: 'reference' (default) draws the aperiodic source from numpy's global RNG like magphase.py:883;
    'device' generates it on the GPU (Philox, one uint64 seed per utterance in noise_seeds, default 0, 1, ...): same
    distribution, not the reference's sample values, independent of batching and sharding.
    pcm16_norm: False (default) returns float64 signals; a number (la.write_audio_file's norm, 0.98) or None returns
    the int16 samples la.write_audio_file(..., norm=pcm16_norm) would store, converted on the device (mpx_pcm16).
    prepared: the host side of THIS batch built ahead of time (engine.prepare_async("synthesis", utts, fs, fft_len=...,
    b_voi_ap_win=..., b_const_rate=...).result(): the planner thread prepares launch i + 1 while this thread enqueues
    launch i; None: prepared here)."""
    engine = engine or get_engine()
    plan = CompressedSynthesisPlan(engine, utts, fs, fft_len=fft_len, b_voi_ap_win=b_voi_ap_win,
                                   b_const_rate=b_const_rate, alpha_phase=alpha_phase, noise=noise,
                                   per_phase_type=per_phase_type, post_filter=b_post_filter, b_fbank_mel=b_fbank_mel,
                                   noise_mode=noise_mode, noise_seeds=noise_seeds, defer_rng=defer_rng, prepared=prepared)
    pcm_dev = plan.run()
    if b_out_hpf:   # magphase.py:981-995, float64 on the device (engine.output_hpf); _output_hpf is the host form
        pcm_dev = engine.output_hpf(pcm_dev, plan.out_off_host, fs)
    if pcm16_norm is not False and async_out:
        pcm, ticket = engine.output_pcm16(pcm_dev, plan.out_off_host, norm=pcm16_norm, async_out=True)
        return [pcm[plan.out_off_host[u]:plan.out_off_host[u + 1]] for u in range(len(utts))], ticket
    if async_out:
        raise ValueError("async_out needs pcm16_norm")
    if pcm16_norm is not False:
        pcm = engine.output_pcm16(pcm_dev, plan.out_off_host, norm=pcm16_norm)
    elif b_out_hpf:
        pcm = pcm_dev.cpu().numpy()
    else:
        pcm = engine.to_host_f64(pcm_dev)
    return [pcm[plan.out_off_host[u]:plan.out_off_host[u + 1]] for u in range(len(utts))]


def synthesis_from_compressed(m_mag_mel_log, m_real_mel, m_imag_mel, v_lf0, fs, fft_len=None, b_voi_ap_win=True,
                              b_fbank_mel=False, b_const_rate=False, per_phase_type='magphase', alpha_phase=None,
                              b_out_hpf=True):
    """magphase.py:825-997, per_phase_type in {'magphase', 'min_phase', 'linear'}; b_fbank_mel: magnitudes unwarped by
    the filter-bank interpolation (la.sp_mel_unwarp_fbank) instead of the cepstral cosine matrix."""
    return synthesis_from_compressed_batch([(m_mag_mel_log, m_real_mel, m_imag_mel, v_lf0)], fs, fft_len=fft_len,
                                           b_voi_ap_win=b_voi_ap_win, b_const_rate=b_const_rate,
                                           alpha_phase=alpha_phase, b_out_hpf=b_out_hpf,
                                           per_phase_type=per_phase_type, b_fbank_mel=b_fbank_mel)[0]


def synthesis_from_acoustic_modelling(in_feats_dir, filename_token, out_syn_dir, mag_dim, phase_dim, fs,
                                      fft_len=None, pf_type='no', b_const_rate=False):
    """magphase.py:3229-3275."""
    print("\nSynthesising file: " + filename_token + '.wav............................')
    m_mag_mel_log = lu.read_binfile(in_feats_dir + '/' + filename_token + '.mag', dim=mag_dim)
    m_real_mel = lu.read_binfile(in_feats_dir + '/' + filename_token + '.real', dim=phase_dim)
    m_imag_mel = lu.read_binfile(in_feats_dir + '/' + filename_token + '.imag', dim=phase_dim)
    v_lf0 = lu.read_binfile(in_feats_dir + '/' + filename_token + '.lf0', dim=1)
    if pf_type == 'magphase':
        print('Using MagPhase postfilter...')
        m_mag_mel_log = post_filter(m_mag_mel_log, fs)
    elif pf_type == 'merlin':
        # the device form, as iobatch.generate_waveforms_corpus(pf_type='merlin') runs it: one pf_type, one result whichever
        # entry point writes the wav (post_filter_merlin, the float64 host chain, stays the array API)
        print('Using Merlin postfilter...')
        m_mag_mel_log = post_filter_merlin_device(m_mag_mel_log, fs)
    elif pf_type == 'no':
        print('No postfilter...')
    v_syn_sig = synthesis_from_compressed(m_mag_mel_log, m_real_mel, m_imag_mel, v_lf0, fs, fft_len=fft_len,
                                          b_const_rate=b_const_rate)
    la.write_audio_file(out_syn_dir + '/' + filename_token + '.wav', v_syn_sig, fs)
    return


# ======================================================================================================
# compressed-feature analysis
# ======================================================================================================
def analysis_compressed_batch(utts, fft_len=None, mag_dim=60, phase_dim=10, b_const_rate=False, alpha_phase=None,
                              engine=None, as_float32=False, async_out=False, prepared=None):
    """
    Batched magphase.py:2947-2988 for utterances with epochs: utts = list of (v_sig, fs, v_pm_sec, v_voi), one
    sample rate per call.  Lossless analysis (k_analysis) stays on the device; the mel warp runs on it directly.
    Returns a list of (m_mag_mel_log, m_real_mel, m_imag_mel, v_lf0_smth, v_shift, fs, fft_len).
    async_out (with as_float32): returns (list, ticket) -- the three matrices are views of a page-locked buffer the device
    is still copying into; ticket.wait() before reading them, ticket.release() when done (engine.HostTicket).
    prepared: the host side of THIS batch built ahead of time (engine.prepare_async("analysis", utts, fft_len).result():
    the planner thread prepares launch i + 1 while this thread enqueues launch i; None: prepared here).
    """
    engine = engine or get_engine()
    if len(utts) == 0:   # nothing to do (the per-utterance loop of the reference would run zero times)
        from .engine import HostTicket

        return ([], HostTicket(None, None, None, None)) if async_out else []
    plan = CompressedAnalysisPlan(engine, utts, fft_len=fft_len, mag_dim=mag_dim, phase_dim=phase_dim,
                                  b_const_rate=b_const_rate, alpha_phase=alpha_phase, prepared=prepared)
    for lens in plan.lossless.long_frame_lens:
        for n in lens:
            warnings.warn(_WARN_LONG % (plan.fft_len, n))
    # as_float32: the device's float32 values as they are (what the feature files store), no widening to float64
    ticket = None
    if async_out:
        if not as_float32:
            raise ValueError("async_out needs as_float32")
        (h_mag, h_real, h_imag), ticket = engine.to_host_f32_async(list(plan.run()))
    else:
        h_mag, h_real, h_imag = ((engine.to_host_f32 if as_float32 else engine.to_host_f64)(t_) for t_ in plan.run())
    res = []
    try:
        # signal.medfilt of every utterance's f0 in one pass (hostmath.medfilt3_batch: bit-identical, also for 0 or 1
        # vectors; 30 us per scipy call)
        med_flat = None if b_const_rate else getattr(plan.lossless, "f0_med_flat", None)
        if med_flat is not None:   # the native planner already holds f0 and its median-3 for the whole batch
            f0_flat = plan.f0_out.flat
            lf0_cat = la.f0_to_lf0((f0_flat > 0).astype('float') * med_flat)   # magphase.py:2499-2501
            off = np.asarray(plan.out_off).tolist()
        else:
            f0_med = hm.medfilt3_batch(plan.f0_out)
            # lf0 of the whole batch in one pass (la.f0_to_lf0 per utterance: the same element-wise operations)
            sizes = [int(np.size(f)) for f in plan.f0_out]
            lf0_cat = la.f0_to_lf0((np.concatenate(plan.f0_out) > 0).astype('float') * np.concatenate(f0_med)) if sizes else np.zeros(0)  # magphase.py:2499-2501
            off = np.concatenate(([0], np.cumsum(sizes))).tolist()
        o_off = np.asarray(plan.out_off).tolist()
        shifts = plan.lossless.v_shift
        for u in range(len(utts)):
            a, b = o_off[u], o_off[u + 1]
            v_lf0 = lf0_cat[off[u]:off[u + 1]].copy()
            res.append((h_mag[a:b], h_real[a:b], h_imag[a:b], v_lf0, shifts[u].astype(int), plan.fs, plan.fft_len))
    except BaseException:
        if ticket is not None:   # the page-locked slot goes back to the ring when the host part fails
            ticket.release()
        raise
    return (res, ticket) if async_out else res


def format_for_modelling(m_mag, m_real, m_imag, v_f0, fs, mag_dim=60, phase_dim=45, b_mag_fbank_mel=False,
                         alpha_phase=None):
    """
    magphase.py:2490-2544 for lossless features that are already in host arrays: f0 smoothing / lf0 on the host (fp64),
    the two mel warps on the device (mpx_mel_warp).  Returns (m_mag_mel_log, m_real_mel, m_imag_mel, v_lf0_smth).
    b_mag_fbank_mel=True: the magnitudes go through the mel filter bank (la.sp_mel_warp_fbank, magphase.py:2504-2505;
    mpx_mel_warp_fbank) instead of the cepstral warp.  In the reference this branch is reachable only through this
    function (analysis_compressed never forwards the flag, Q7).
    """
    from scipy import signal
    engine = get_engine()
    v_f0 = np.asarray(v_f0, dtype=np.float64)
    v_voi = (v_f0 > 0).astype('float')                                       # magphase.py:2497
    v_lf0_smth = la.f0_to_lf0(v_voi * signal.medfilt(v_f0))                  # :2499-2501
    mag, real, imag = (engine.feats_to_device(x) for x in (m_mag, m_real, m_imag))
    out = engine.mel_warp_feats(mag, real, imag, v_voi, fs, mag_dim, phase_dim, alpha_phase=alpha_phase,
                                b_mag_fbank_mel=bool(b_mag_fbank_mel))
    return tuple(engine.to_host_f64(t) for t in out) + (v_lf0_smth,)


shift_to_f0 = hm.shift_to_f0
get_num_full_mel_coeffs_from_num_phase_coeffs = hm.get_num_full_mel_coeffs_from_num_phase_coeffs


def analysis_compressed(wav_file, fft_len=None, mag_dim=60, phase_dim=10, b_const_rate=False, b_mag_fbank_mel=False,
                        alpha_phase=None):
    """magphase.py:2947-2988 (b_mag_fbank_mel is accepted and, as in the reference, never forwarded)."""
    v_sig, fs = la.read_audio_file(wav_file)
    v_pm_sec, v_voi = _epochs_for(wav_file)
    return analysis_compressed_batch([(v_sig, fs, v_pm_sec, v_voi)], fft_len=fft_len, mag_dim=mag_dim,
                                     phase_dim=phase_dim, b_const_rate=b_const_rate, alpha_phase=alpha_phase)[0]


def analysis_for_acoustic_modelling(wav_file, out_dir, fft_len=None, mag_dim=60, phase_dim=10, b_const_rate=False,
                                    b_mag_fbank_mel=False, alpha_phase=None):
    """magphase.py:2992-3022, including Q7: alpha_phase=b_mag_fbank_mel is what the reference forwards (:3010)."""
    m_mag_mel_log, m_real_mel, m_imag_mel, v_lf0_smth, v_shift, fs, fft_len = analysis_compressed(
        wav_file, fft_len=fft_len, mag_dim=mag_dim, phase_dim=phase_dim, b_const_rate=b_const_rate,
        b_mag_fbank_mel=b_mag_fbank_mel, alpha_phase=b_mag_fbank_mel)
    file_id = os.path.basename(wav_file).split(".")[0]
    write_featfile(m_mag_mel_log, out_dir, file_id + '.mag')
    write_featfile(m_real_mel, out_dir, file_id + '.real')
    write_featfile(m_imag_mel, out_dir, file_id + '.imag')
    write_featfile(v_lf0_smth, out_dir, file_id + '.lf0')
    if not b_const_rate:
        write_featfile(v_shift, out_dir, file_id + '.shift')
    return


# ======================================================================================================
# label re-timing for constant-frame-rate trainers
# ======================================================================================================
def get_num_of_frms_per_state(v_shift, lab_state_align_file, fs, b_prevent_zeros=False, n_states_x_phone=5,
                              nfrms_tolerance=6):
    """
    magphase.py:2111-2150: number of pitch-synchronous frames (epochs) that fall inside every line of an HTS
    state-aligned label file ([start, end) in units of 100 ns).  Frames left over after the last line (at most
    `nfrms_tolerance`: label files often end early) go to the last state; a total mismatch or a phone without any
    frame raises ValueError like the reference.  Returns float64[n_states].
    """
    m_lab = np.loadtxt(lab_state_align_file, usecols=(0, 1), ndmin=2)
    m_lab_ms = m_lab / 10000.0
    v_ep_ms = np.cumsum(v_shift) * 1000.0 / fs
    # epochs e with start <= e < end == (#epochs < end) - (#epochs < start); the epoch times are non-decreasing
    v_ep_sorted = np.sort(v_ep_ms)
    v_n = (np.searchsorted(v_ep_sorted, m_lab_ms[:, 1], side="left")
           - np.searchsorted(v_ep_sorted, m_lab_ms[:, 0], side="left")).astype(np.float64)
    v_n = np.maximum(v_n, 0.0)
    n_left = np.size(v_shift) - np.sum(v_n)
    if 0 < n_left <= nfrms_tolerance:
        v_n[-1] += n_left
    if np.sum(v_n) != np.size(v_shift):
        raise ValueError("Total number of frames is different to the number of frames of the shifts.")
    v_n_ph = v_n.reshape((v_n.size // n_states_x_phone, n_states_x_phone)).sum(axis=1)
    if np.any(v_n_ph == 0.0):
        raise ValueError("There is some phoneme(s) that do(es) not contain any frame.")
    if b_prevent_zeros:
        v_n[v_n == 0] = 1
    return v_n
