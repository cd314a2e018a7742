"""
TEST INFRASTRUCTURE ONLY -- CPU oracle for the MagPhase analysis/synthesis hot path.

A numpy/scipy float64 restatement of the live functions of the reference
(CSTR-Edinburgh/magphase, /root/reference/src/{magphase,libaudio,libutils}.py), written from
the reference's behaviour, one function per reference function, each citing the file:line it
follows.  Who may import this module: tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg -- as the checker / the timed CPU baseline, never as a product path.
The product (magphase_amd/) does not import it and fails loudly without its HIP library.

Pinning status
  * Everything except the two external binaries is PINNED: oracle/gen_golden.py imports the
    real reference (oracle/ref_shim.py) in the build container and tests/test_oracle_vs_golden.py
    compares this restatement with the committed outputs of the reference itself
    (tests/golden/*.npz; re-running oracle/gen_golden.py where /root/reference exists
    regenerates them bit for bit).
  * ``sptk_mcep`` / ``freqt`` restate SPTK-3.9 ``mcep -j 0`` (an external C binary fetched by
    the reference's tools/download_and_compile_tools.sh:5,36; source absent from
    /root/reference): PARITY UNPINNED.  Pinned only by self-consistency KATs
    (tests/test_oracle_mcep_kats.py).  Golden vectors that pass through it are labelled
    "oracle-with-our-mcep".
  * Epoch detection (REAPER, external binary) is out of scope: epochs are an input.

All arithmetic is float64; index arithmetic reproduces the reference's IEEE-754 op sequence
(np.round half-to-even, truncating astype(int), sequential cumsum) -- SURVEY.md F5, Q1-Q3.
"""
import warnings

import numpy as np
from scipy import interpolate, signal

MAGIC = -1.0e10  # libaudio.py:17 -- logarithm floor (same constant SPTK uses)


# =============================================================================================
# libutils.py
# =============================================================================================
def round_to_int(x):
    """libutils.py:131-133 -- np.round (half-to-even) then truncating int cast."""
    return np.round(x).astype(int)


def read_binfile(filename, dim=60):
    """libutils.py:112-120 -- raw little-endian float32, row-major, no header -> float64, squeezed."""
    v = np.fromfile(filename, dtype=np.float32)
    if v.size % dim != 0:
        raise ValueError("Dimension provided not compatible with file size.")
    return np.squeeze(v.reshape((-1, dim)).astype("float64"))


def write_binfile(m_data, filename):
    """libutils.py:122-127."""
    np.array(m_data, "float32").tofile(filename)


# =============================================================================================
# constants  (magphase.py:3279-3317)
# =============================================================================================
def define_alpha(fs):
    """magphase.py:3279-3290."""
    table = {16000: 0.58, 22050: 0.65, 44100: 0.76, 48000: 0.77}
    if fs not in table:
        raise ValueError("Sample rate %d not supported yet." % fs)
    return table[fs]


def define_fft_len(fs):
    """magphase.py:3292-3299."""
    if fs in (22050, 16000):
        return 2048
    if fs == 8000:
        return 1024
    return 4096


def define_crossfade_params(fs):
    """magphase.py:3301-3317 (same warning condition for untuned rates)."""
    crsf_bw = 2000
    if fs == 48000:
        return 5000, crsf_bw
    if fs == 16000:
        return 2500, crsf_bw
    warnings.warn("Constant crsf_cf not tested nor tunned to synthesise at fs=%d Hz." % fs)
    if fs == 44100:
        return 4500, crsf_bw
    return 3500, crsf_bw


# =============================================================================================
# small helpers of libaudio.py
# =============================================================================================
def log_protected(m_x):
    """libaudio.py:241-248 -- log with inf/nan replaced by MAGIC."""
    with np.errstate(divide="ignore", invalid="ignore"):
        m_y = np.log(m_x)
    m_y = np.array(m_y, dtype=np.float64)
    m_y[~np.isfinite(m_y)] = MAGIC
    return m_y


def f0_to_lf0(v_f0):
    """libaudio.py:458-465 -- only infinities are floored (nan stays nan, as in the reference)."""
    with np.errstate(divide="ignore"):
        v_lf0 = np.log(v_f0)
    v_lf0[np.isinf(v_lf0)] = MAGIC
    return v_lf0


def db(m_data, b_inv=False):
    """libaudio.py:635-639."""
    if not b_inv:
        return 20 * np.log10(m_data)
    return 10 ** (m_data / 20)


def shift_to_pm(v_shift):
    """libaudio.py:60-62."""
    return np.cumsum(v_shift)


def pm_to_shift(v_pm):
    """libaudio.py:65-67."""
    return np.diff(np.hstack((0, v_pm)))


def warp_axis(alpha, nbins):
    """Frequency-warped axis of the first-order all-pass: libaudio.py:612-614 and :711-715."""
    w = np.linspace(0, np.pi, num=nbins)
    ww = np.arctan((1 - alpha ** 2) * np.sin(w) / ((1 + alpha ** 2) * np.cos(w) - 2 * alpha))
    ww[ww < 0] += np.pi
    return ww


def build_mel_curve(alpha, nbins, amp=np.pi):
    """libaudio.py:711-718."""
    return warp_axis(alpha, nbins) * (amp / np.pi)


def half_windows(left_len, right_len, win_func=np.hanning):
    """
    libaudio.py:70-84 (gen_non_symmetric_win): rising half of win(1+2*left) followed by the
    falling half of win(1+2*right) without its first sample; total length left+right+1.
    """
    wl = win_func(1 + 2 * left_len)[: left_len + 1]
    wr = win_func(1 + 2 * right_len)[: right_len + 1][::-1]
    return np.concatenate((wl, wr[1:]))


def centred_window(winlen_l, winlen_r, totlen, win_func, fill_with_bound=False):
    """libaudio.py:90-103 (gen_centr_win): asymmetric window whose peak sits at floor(totlen/2)."""
    w_short = half_windows(winlen_l, winlen_r, win_func)
    centre = int(np.floor(totlen / 2.0))
    start = centre - winlen_l
    v_win = np.zeros(totlen)
    if fill_with_bound:
        v_win += w_short[0]
    if start < 0 or start + len(w_short) > totlen:
        # numpy slice-assignment semantics of the reference: a shape mismatch raises
        raise ValueError("could not broadcast window of len %d into frame of len %d" % (len(w_short), totlen))
    v_win[start : start + len(w_short)] = w_short
    return v_win


def raised_hanning(length, att=1.0):
    """magphase.py:25-31."""
    return (1 - att) + att * np.hanning(length)


def voi_noise_window(length):
    """magphase.py:67-68 (Q11)."""
    return np.bartlett(length) ** 2.5


def add_hermitian_half_real(m):
    """libaudio.py:371-372 -- even extension [x0..x_{H-1}, x_{H-2}..x_1]."""
    return np.hstack((m, m[:, -2:0:-1]))


def hermitian_full_spectrum(m_cplx):
    """
    libaudio.py:369-388 with data_type='complex' (Q5): real part even-extended; imaginary part
    has DC and Nyquist forced to 0 and is odd-extended.  (The reference zeroes the caller's
    imag view in place; the returned spectrum is what matters here.)
    """
    re = add_hermitian_half_real(m_cplx.real)
    im = np.array(m_cplx.imag, dtype=np.float64)
    im[:, 0] = 0.0
    im[:, -1] = 0.0
    im = np.hstack((im, -im[:, -2:0:-1]))
    return re + 1j * im


def spectral_crossfade_lowpass_curve(nbins_half, cut_off, bw, fs):
    """
    libaudio.py:160-186 evaluated for m_sp_l = ones, m_sp_r = zeros (the only use on the path,
    magphase.py:873-875): 1 below bin_l, falling Hann half on [bin_l, bin_r], 0 above.
    """
    nfft = (nbins_half - 1) * 2
    bin_l = round_to_int((cut_off - bw / 2.0) * nfft / float(fs))
    bin_r = round_to_int((cut_off + bw / 2.0) * nfft / float(fs))
    bw_bin = bin_r - bin_l
    w = np.hanning(2 * bw_bin + 1)[bw_bin:]
    return np.hstack((np.ones(bin_l), w, np.zeros(nbins_half - bin_r - 1)))


# =============================================================================================
# epochs (libaudio.py:421-447) -- starts AFTER the REAPER text parse
# =============================================================================================
def clean_epochs(v_pm_sec, v_voi, check_len_smpls=-1, fs=-1):
    """libaudio.py:435-447: drop non-increasing epoch times; drop epochs landing at/after n-1."""
    v_pm_sec = np.asarray(v_pm_sec, dtype=np.float64)
    v_voi = np.asarray(v_voi, dtype=np.float64)
    ok = np.hstack((True, np.diff(v_pm_sec) > 0))
    v_pm_sec, v_voi = v_pm_sec[ok], v_voi[ok]
    if check_len_smpls > 0:
        pm = round_to_int(v_pm_sec * fs)
        if pm[-1] >= (check_len_smpls - 1):
            ok2 = pm < (check_len_smpls - 1)
            v_pm_sec, v_voi = v_pm_sec[ok2], v_voi[ok2]
    return v_pm_sec, v_voi


def read_est_file(est_file, check_len_smpls=-1, fs=-1):
    """libaudio.py:421-447 -- REAPER .est: 7 header lines, columns 0 (seconds) and 1 (voiced)."""
    m = np.atleast_2d(np.loadtxt(est_file, skiprows=7, usecols=[0, 1]))
    return clean_epochs(m[:, 0], m[:, 1], check_len_smpls, fs)


# =============================================================================================
# analysis  (magphase.py:74-119, 266-334, 457-476, 2198-2207, 2869-2906)
# =============================================================================================
def frame_bounds(v_pm_smpls, n_smpls):
    """
    magphase.py:77-83,90-98,112-117 -- epochs rounded (Q1) and extended with 0 and n-1 (Q4).
    Returns (v_pm_plus int[F+2], v_left int[F], v_right int[F], v_len int[F]).
    """
    pm = round_to_int(np.asarray(v_pm_smpls))
    pm_plus = np.hstack((0, pm, n_smpls - 1))
    left = pm_plus[1:-1] - pm_plus[:-2]
    right = pm_plus[2:] - pm_plus[1:-1]
    return pm_plus, left, right, left + right + 1


def windowing(v_sig, v_pm_smpls, win_func=np.hanning):
    """magphase.py:74-119.  win_func: a window function or a per-frame list of them."""
    pm_plus, left, right, lens = frame_bounds(v_pm_smpls, np.size(v_sig))
    frames = []
    for f in range(left.size):
        wf = win_func[f] if isinstance(win_func, list) else win_func
        seg = v_sig[pm_plus[f] : pm_plus[f + 2] + 1]
        frames.append(seg * half_windows(left[f], right[f], wf))
    lens = np.array([len(x) for x in frames], dtype=int)
    return frames, lens, pm_plus, left.astype(int), right.astype(int)


def analysis_frames(v_sig, v_pm_smpls, fft_len):
    """
    magphase.py:266-334 (analysis_with_del_comp_from_pm, nwin_per_pitch_period=0.5):
    zero-pad (or truncate + warn, Q19) each windowed frame to fft_len, rotate left by the
    frame's left length so the epoch sits at index 0, FFT, keep bins 0..fft_len/2.
    Returns (m_fft complex128 [F x H], v_shift int[F]).
    """
    frames, lens, _, v_shift, _ = windowing(v_sig, v_pm_smpls)
    nfrms = len(frames)
    m_frms = np.zeros((nfrms, fft_len))
    for f in range(nfrms):
        if lens[f] <= fft_len:
            row = np.zeros(fft_len)
            row[: lens[f]] = frames[f]
        else:
            row = frames[f][:fft_len].copy()
            warnings.warn(
                "fft_len (%d) is shorter than the current detected frame length (%d)." % (fft_len, lens[f])
            )
        s = v_shift[f]
        m_frms[f, :] = np.concatenate((row[s:], row[:s]))  # python slicing: s >= fft_len -> no rotation
    m_fft = np.fft.fft(m_frms)[:, : fft_len // 2 + 1]
    return m_fft, v_shift


def shift_to_f0(v_shift, v_voi, fs):
    """magphase.py:2198-2207 with b_smooth=False: f0 = voi * fs / shift (left-to-right, Q2)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return v_voi * fs / v_shift.astype("float64")


def compute_lossless_feats(m_fft, v_shift, v_voi, fs):
    """magphase.py:457-476."""
    m_mag = np.absolute(m_fft)
    zero = m_mag == 0.0
    div = m_mag.copy()
    div[zero] = 1.0
    m_real = m_fft.real / div
    m_imag = m_fft.imag / div
    m_real[zero] = 0.0
    m_imag[zero] = 0.0
    return m_mag, m_real, m_imag, shift_to_f0(v_shift, v_voi, fs)


def analysis_lossless_from_epochs(v_sig, fs, v_pm_sec, v_voi, fft_len=None):
    """
    magphase.py:2869-2906 from the point where REAPER's epochs have been read
    (:2877-2879, :2891, :2894).  v_pm_sec / v_voi are the two .est columns.
    Returns (m_mag, m_real, m_imag, v_f0, fs, v_shift).
    """
    v_pm_sec, v_voi = clean_epochs(v_pm_sec, v_voi, check_len_smpls=len(v_sig), fs=fs)
    if fft_len is None:
        fft_len = define_fft_len(fs)
    m_fft, v_shift = analysis_frames(v_sig, v_pm_sec * fs, fft_len)
    m_mag, m_real, m_imag, v_f0 = compute_lossless_feats(m_fft, v_shift, v_voi, fs)
    return m_mag, m_real, m_imag, v_f0, fs, v_shift


# =============================================================================================
# lossless synthesis  (magphase.py:34-62, 1759-1776, 2210-2215)
# =============================================================================================
def f0_to_shift(v_f0_in, fs, unv_frm_rate_ms=5):
    """magphase.py:2210-2215 (Q3)."""
    v_f0 = v_f0_in.copy()
    v_f0[v_f0 == 0] = 1000.0 / unv_frm_rate_ms
    return fs / v_f0


def ola(m_frm, v_pm):
    """
    magphase.py:34-62 (PSOLA, win_func=None): frame i is added at offset pm_i - pm_0 of a
    buffer of pm_{F-1}+frmlen samples; then ``[frmlen/2 - pm_0:]`` and
    ``[:pm_{F-1} + shift_{F-1} + 1]`` python slices (negative start keeps its python meaning).
    """
    v_pm = v_pm.astype(int)
    nfrms, frmlen = m_frm.shape
    v_sig = np.zeros(v_pm[-1] + frmlen)
    starts = v_pm - v_pm[0]
    for i in range(nfrms):
        v_sig[starts[i] : starts[i] + frmlen] += m_frm[i, :]
    last_shift = v_pm[-1] - v_pm[-2] if nfrms > 1 else v_pm[-1]
    v_sig = v_sig[(frmlen // 2 - v_pm[0]) :]
    return v_sig[: (v_pm[-1] + last_shift + 1)]


def synthesis_from_lossless(m_mag, m_real, m_imag, v_f0, fs):
    """magphase.py:1759-1776."""
    ph = m_real + 1j * m_imag
    ph_mag = np.absolute(ph)
    ph_mag[ph_mag == 0.0] = 1.0
    m_fft = hermitian_full_spectrum(m_mag * ph / ph_mag)
    m_frm = np.fft.fftshift(np.fft.ifft(m_fft).real, axes=1)
    v_pm = shift_to_pm(f0_to_shift(v_f0, fs))
    return ola(m_frm, v_pm)


# =============================================================================================
# SPTK-3.9 ``mcep`` restatement  -- PARITY UNPINNED (external binary, source not in /root/reference)
# =============================================================================================
def freqt(c1, order_out, alpha):
    """
    Frequency transformation of cepstra (SPTK-3.9 ``freqt``; Tokuda et al., "Recursive calculation
    of mel-cepstrum from LPC coefficients", 1994): input cepstrum c1[0..m1] (rows of a 2-D array),
    output order_out+1 coefficients for all-pass constant alpha.  Same recursion order as SPTK:
    the input is consumed from c1[m1] down to c1[0].
    """
    c1 = np.atleast_2d(np.asarray(c1, dtype=np.float64))
    nfr, n1 = c1.shape
    m2 = order_out
    b = 1.0 - alpha * alpha
    g = np.zeros((nfr, m2 + 1))
    for i in range(n1 - 1, -1, -1):
        d = g.copy()
        g[:, 0] = c1[:, i] + alpha * d[:, 0]
        if m2 >= 1:
            g[:, 1] = b * d[:, 0] + alpha * d[:, 1]
        for j in range(2, m2 + 1):
            g[:, j] = d[:, j - 1] + alpha * (d[:, j] - g[:, j - 1])
    return g


def freqt_matrix(n_in, order_out, alpha):
    """The linear map of ``freqt`` as a matrix A[(order_out+1) x n_in] (mc = A @ c)."""
    return freqt(np.eye(n_in), order_out, alpha).T.copy()


def sptk_mcep(m_sp, n_coeffs=60, alpha=0.77, in_type=3, fft_len=0):
    """
    What the reference obtains at libaudio.py:575-601 from
    ``mcep -a <alpha %1.2f> -m n_coeffs-1 -l fft_len -e 1.0E-8 -j 0 -f 0.0 -q in_type``:
      float32 input file -> periodogram (q=3: x^2, q=2: exp(x)^2, q=1: dB) + 1e-8 -> log ->
      real IFFT -> c[0]/2, c[N/2]/2 -> freqt(c[0..N/2], n_coeffs-1, alpha) -> float32 output file.
    ``-j 0`` = zero Newton iterations, i.e. SPTK's initial estimate is returned.
    """
    m_sp = np.atleast_2d(np.asarray(m_sp))
    x = m_sp.astype(np.float32).astype(np.float64)  # lu.write_binfile -> float32 file (libaudio.py:582)
    if fft_len == 0:
        fft_len = 2 * (x.shape[1] - 1)
    a = float("%1.2f" % alpha)  # the command line prints alpha with two decimals (libaudio.py:589)
    if in_type == 3:
        p = x * x
    elif in_type == 2:
        e = np.exp(x)
        p = e * e
    elif in_type == 1:
        e = np.exp((x / 20.0) * np.log(10.0))
        p = e * e
    else:
        raise ValueError("in_type must be 1, 2 or 3")
    logp = np.log(p + 1.0e-8)
    c = np.fft.ifft(add_hermitian_half_real(logp)).real  # symmetric input -> real cepstrum
    half = fft_len // 2
    c = c[:, : half + 1].copy()
    c[:, 0] /= 2.0
    c[:, half] /= 2.0
    mc = freqt(c, n_coeffs - 1, a)
    return mc.astype(np.float32).astype(np.float64)  # float32 .mgc file read back (libutils.py:112-120)


# =============================================================================================
# mel warp / unwarp  (libaudio.py:605-684)
# =============================================================================================
def cos_matrix(n_cep, n_spbins, alpha):
    """libaudio.py:611-619 -- trans[i, k] = cos(i * warp_alpha(pi k/(n_spbins-1)))."""
    return np.cos(np.arange(n_cep)[:, None] * warp_axis(alpha, n_spbins)[None, :])


def mcep_to_sp_cosmat(m_mcep, n_spbins, alpha=0.77, out_type="abs"):
    """libaudio.py:605-631."""
    m_sp = np.dot(m_mcep, cos_matrix(m_mcep.shape[1], n_spbins, alpha))
    if out_type == "abs":
        return np.exp(m_sp)
    if out_type == "db":
        return m_sp * (20 / np.log(10))
    return m_sp


def sp_mel_warp(m_sp, nbins_out, alpha=0.77, in_type=3):
    """libaudio.py:643-661: SPTK mcep, then cosine matrix with alpha=0 on nbins_out points."""
    m_mcep = sptk_mcep(m_sp, n_coeffs=nbins_out, alpha=alpha, in_type=in_type)
    out_type = {3: "abs", 1: "db", 2: "log"}[in_type]
    return mcep_to_sp_cosmat(m_mcep, nbins_out, alpha=0.0, out_type=out_type)


def sp_mel_unwarp(m_sp_mel, nbins_out, alpha=0.77, in_type="log"):
    """libaudio.py:667-684 (Q6: coefficients 1..ncoeffs-3 doubled, ncoeffs-2 left as is)."""
    ncoeffs = m_sp_mel.shape[1]
    if in_type == "abs":
        m_sp_mel = np.log(m_sp_mel)
    m_mcep = np.fft.ifft(add_hermitian_half_real(m_sp_mel)).real
    m_mcep[:, 1 : (ncoeffs - 2)] *= 2
    return mcep_to_sp_cosmat(m_mcep[:, :ncoeffs], nbins_out, alpha=alpha, out_type=in_type)


def unwarp_matrix(ncoeffs, nbins_out, alpha):
    """sp_mel_unwarp(in_type='log') is linear: returns U[ncoeffs x nbins_out], out = x @ U."""
    return sp_mel_unwarp(np.eye(ncoeffs), nbins_out, alpha=alpha, in_type="log")


def get_num_full_mel_coeffs_from_num_phase_coeffs(freq_hz, phase_dim, alpha, fs):
    """magphase.py:2479-2487."""
    cw = 2 * np.pi * freq_hz / float(fs)
    cf_mel = np.arctan((1 - alpha ** 2) * np.sin(cw) / ((1 + alpha ** 2) * np.cos(cw) - 2 * alpha))
    if cf_mel < 0:
        cf_mel += np.pi
    return round_to_int(1 + (np.pi * (phase_dim - 1) / float(cf_mel)))


def phase_uncompress_type1_mcep(m_real_mel, m_imag_mel, alpha, fft_len, fs):
    """magphase.py:1219-1235: nearest-neighbour extension pd -> K (last bin repeated), then unwarp."""
    pd = m_real_mel.shape[1]
    cf = define_crossfade_params(fs)[0]
    k_full = int(get_num_full_mel_coeffs_from_num_phase_coeffs(cf, pd, alpha, fs))
    idx = np.minimum(np.arange(k_full), pd - 1)
    half = 1 + fft_len // 2
    m_real = sp_mel_unwarp(m_real_mel[:, idx], half, alpha=alpha, in_type="log")
    m_imag = sp_mel_unwarp(m_imag_mel[:, idx], half, alpha=alpha, in_type="log")
    return m_real, m_imag


# =============================================================================================
# constant <-> variable frame rate  (magphase.py:1426-1449, 2219-2252)
# =============================================================================================
def interp_from_variable_to_const_frm_rate(m_data, v_pm_smpls, const_rate_ms, fs):
    """magphase.py:2219-2239 (Q15): grid arange(step, pm[-1], step); first row duplicated at t=0."""
    data = np.asarray(m_data)
    was_1d = data.ndim == 1
    if was_1d:
        data = data.reshape((-1, 1))
    step = fs * const_rate_ms / 1000
    centres = np.arange(step, v_pm_smpls[-1], step)
    if v_pm_smpls[0] > 0:
        f = interpolate.interp1d(np.r_[0, v_pm_smpls], np.vstack((data[0, :], data)), axis=0, kind="linear")
    else:
        f = interpolate.interp1d(v_pm_smpls, data, axis=0, kind="linear")
    out = f(centres)
    return out[:, 0] if was_1d else out


def get_shifts_and_frm_locs_from_const_shifts(v_shift_c_rate, frm_rate_ms, fs):
    """
    magphase.py:1426-1449 (Q16): backward serial scan from the last constant-rate centre:
    pos_k, shift_k = lerp(shift)(pos_k), pos_{k-1} = pos_k - shift_k, until pos leaves the grid.
    At most 2n-1 steps are taken (the reference allocates 2n slots and never fills slot 0).
    """
    n = np.size(v_shift_c_rate, 0)
    step = fs * frm_rate_ms / 1000
    centres = step * np.arange(1, n + 1)
    f = interpolate.interp1d(centres, v_shift_c_rate, axis=0, kind="linear")
    shifts = np.zeros(n * 2)
    locs = np.zeros(n * 2)
    pos = centres[-1]
    for i in range(2 * n - 1, 0, -1):
        locs[i] = pos
        try:
            shifts[i] = f(pos)
        except ValueError:
            locs = locs[i + 1 :]
            shifts = shifts[i + 1 :]
            break
        pos = pos - shifts[i]
    return shifts, locs


def interp_from_const_to_variable_rate(m_data, v_frm_locs_smpls, frm_rate_ms, fs):
    """magphase.py:2242-2252."""
    n = np.size(m_data, 0)
    step = fs * frm_rate_ms / 1000
    centres = step * np.arange(1, n + 1)
    return interpolate.interp1d(centres, m_data, axis=0, kind="linear")(v_frm_locs_smpls)


def fbank_matrix(v_bins_warp, nbands, win_func=np.hanning):
    """
    libaudio.py:721-749 (apply_fbank, the filter-bank construction): band centres equally spaced on the warped axis,
    mapped back to bins through a quadratic interp1d and rounded; band b is the normalised asymmetric window
    (half_windows) from centre b-1 to centre b+1.  Returns m_fbank [nbins x nbands].
    """
    nbins = v_bins_warp.size
    v_cntrs_mel = np.linspace(0, v_bins_warp[-1], nbands)
    v_cntrs = round_to_int(interpolate.interp1d(v_bins_warp, np.arange(nbins), kind="quadratic")(v_cntrs_mel))
    m_fbank = np.zeros((nbins, nbands))
    ext = np.r_[v_cntrs[0], v_cntrs, v_cntrs[-1]]
    for b in range(1, nbands + 1):
        v_win = half_windows(ext[b] - ext[b - 1], ext[b + 1] - ext[b])
        v_win = v_win / np.sum(v_win)
        m_fbank[ext[b - 1]:ext[b - 1] + v_win.size, b - 1] = v_win
    return m_fbank


def sp_mel_warp_fbank(m_mag, n_melbands, alpha=0.77):
    """libaudio.py:763-769: exp(log_protected(m_mag) . fbank) -- 'average' mode of apply_fbank.  PINNED (golden G11)."""
    m_fbank = fbank_matrix(build_mel_curve(alpha, m_mag.shape[1]), n_melbands)
    with np.errstate(under="ignore"):
        return np.exp(np.dot(log_protected(m_mag), m_fbank))


# =============================================================================================
# compressed analysis  (magphase.py:2490-2544, 2947-2988)
# =============================================================================================
def format_for_modelling(m_mag, m_real, m_imag, v_f0, fs, mag_dim=60, phase_dim=45, alpha_phase=None,
                         b_mag_fbank_mel=False):
    """magphase.py:2490-2544.  The phase streams (and the magnitudes unless b_mag_fbank_mel) pass through the UNPINNED
    mcep; the filter-bank magnitudes (b_mag_fbank_mel=True, :2504-2505) are pure numpy in the reference: pinned, G11."""
    alpha = define_alpha(fs)
    v_voi = (v_f0 > 0).astype("float")
    v_lf0 = f0_to_lf0(v_voi * signal.medfilt(v_f0))
    if b_mag_fbank_mel:
        m_mag_mel_log = log_protected(sp_mel_warp_fbank(m_mag, mag_dim, alpha=alpha))
    else:
        m_mag_mel_log = log_protected(sp_mel_warp(m_mag, mag_dim, alpha=alpha, in_type=3))
    cf, _ = define_crossfade_params(fs)
    if alpha_phase is None:
        alpha_phase = alpha
    k_full = int(get_num_full_mel_coeffs_from_num_phase_coeffs(cf, phase_dim, alpha_phase, fs))
    m_real_mel = sp_mel_warp(m_real, k_full, alpha=alpha_phase, in_type=2)[:, :phase_dim]
    m_imag_mel = sp_mel_warp(m_imag, k_full, alpha=alpha_phase, in_type=2)[:, :phase_dim]
    # masking + clipping (executed twice in the reference, Q9: idempotent)
    m_real_mel = np.clip(m_real_mel * v_voi[:, None], -1, 1)
    m_imag_mel = np.clip(m_imag_mel * v_voi[:, None], -1, 1)
    return m_mag_mel_log, m_real_mel, m_imag_mel, v_lf0


def to_const_rate(m_mag, m_real, m_imag, v_f0, v_shift, fs, const_rate_ms=5.0):
    """magphase.py:2967-2980 (Q15)."""
    v_pm = shift_to_pm(v_shift)
    m_mag_c = interp_from_variable_to_const_frm_rate(m_mag, v_pm, const_rate_ms, fs)
    m_real_c = interp_from_variable_to_const_frm_rate(m_real, v_pm, const_rate_ms, fs)
    m_imag_c = interp_from_variable_to_const_frm_rate(m_imag, v_pm, const_rate_ms, fs)
    v_voi = v_f0 > 1.0
    v_f0_c = interp_from_variable_to_const_frm_rate(
        np.r_[v_f0[v_voi][0], v_f0[v_voi], v_f0[v_voi][-1]], np.r_[0, v_pm[v_voi], v_pm[-1]], const_rate_ms, fs
    )
    v_voi_c = interp_from_variable_to_const_frm_rate(v_voi.astype(np.float64), v_pm, const_rate_ms, fs) > 0.5
    return m_mag_c, m_real_c, m_imag_c, v_f0_c * v_voi_c


def analysis_compressed_from_epochs(v_sig, fs, v_pm_sec, v_voi, fft_len=None, mag_dim=60, phase_dim=10,
                                    b_const_rate=False, alpha_phase=None):
    """magphase.py:2947-2988 after the epoch read."""
    m_mag, m_real, m_imag, v_f0, fs, v_shift = analysis_lossless_from_epochs(v_sig, fs, v_pm_sec, v_voi, fft_len)
    if b_const_rate:
        m_mag, m_real, m_imag, v_f0 = to_const_rate(m_mag, m_real, m_imag, v_f0, v_shift, fs)
    feats = format_for_modelling(m_mag, m_real, m_imag, v_f0, fs, mag_dim=mag_dim, phase_dim=phase_dim,
                                 alpha_phase=alpha_phase)
    return feats + (v_shift, fs, 2 * (np.size(m_mag, 1) - 1))


# =============================================================================================
# post filter  (magphase.py:2300-2378)
# =============================================================================================
def post_filter(m_mag_mel_log, fs, av_len_at_zero=None, av_len_at_nyq=None, boost_at_zero=None, boost_at_nyq=None):
    """magphase.py:2300-2378 (Q20), vectorised over frames (moving averages via prefix sums)."""
    nfrms, mag_dim = m_mag_mel_log.shape
    if mag_dim != 60:
        warnings.warn("Post-filter: It has been only tested with 60 dimensional mag data.")
    opts = [av_len_at_zero, av_len_at_nyq, boost_at_zero, boost_at_nyq]
    if fs == 48000:
        defaults = [round_to_int(11.0 * (mag_dim / 60.0)), round_to_int(3.0 * (mag_dim / 60.0)), 1.8, 2.0]
    elif fs == 16000:
        if any(o is None for o in opts):
            warnings.warn("Post-filter: The default parameters for 16kHz sample rate have not being tunned.")
        defaults = [round_to_int(9.0 * (mag_dim / 60.0)), round_to_int(12.0 * (mag_dim / 60.0)), 2.0, 1.6]
    else:
        if any(o is None for o in opts):
            raise ValueError("Post-filter: It has only been tested with 16kHz and 48kHz sample rates.")
        defaults = opts
    av0, avn, b0, bn = [d if o is None else o for o, d in zip(opts, defaults)]
    v_nx = np.arange(np.floor(av0 / 2), mag_dim - np.floor(avn / 2)).astype(int)
    v_lens = (2 * np.ceil(np.linspace(av0, avn, v_nx.size) / 2) - 1).astype(int)
    half = v_lens // 2
    m_ave = np.zeros((nfrms, mag_dim))
    for j, nxb in enumerate(v_nx):
        m_ave[:, nxb] = np.mean(m_mag_mel_log[:, nxb - half[j] : nxb + half[j] + 1], axis=1)
    m_ave[:, : v_nx[0]] = m_ave[:, [v_nx[0]]]
    m_ave[:, v_nx[-1] :] = m_ave[:, [v_nx[-1]]]
    tilt = np.linspace(b0, bn, mag_dim)
    m_enh = (m_mag_mel_log - m_ave) * tilt[None, :] + m_ave
    m_enh[:, 0] = m_mag_mel_log[:, 0]
    m_enh[:, -1] = m_mag_mel_log[:, -1]
    return m_enh


# =============================================================================================
# compressed synthesis  (magphase.py:825-997)
# =============================================================================================
# =============================================================================================
# Merlin-style post-filter (magphase.py:3375-3465) -- PARITY UNPINNED for the SPTK legs
# =============================================================================================
def rceps_compact(m_log):
    """la.rceps(m, in_type='log', out_type='compact') (libaudio.py:252-269): even extension, real IFFT, coefficients
    1 .. n-3 doubled, first n kept.  Pinned: golden g12 (the reference's own function)."""
    m_log = np.asarray(m_log, dtype=np.float64)
    n = m_log.shape[1]
    m_c = np.fft.ifft(add_hermitian_half_real(m_log)).real
    m_c[:, 1:(n - 2)] *= 2
    return m_c[:, :n]


def _pipe(x):
    """An SPTK pipe / temp file: float32 on the wire (x2x +af, lu.write_binfile), double inside every tool."""
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def _sptk_freqt_aA(c, order_out, a_in, a_out):
    """``freqt -m <len-1> -a a_in -M order_out -A a_out``: SPTK derives ONE all-pass constant
    a = (a_out - a_in) / (1 - a_in a_out) and runs the freqt recursion with it (frame by frame)."""
    a = (a_out - a_in) / (1.0 - a_in * a_out)
    return freqt(c, order_out, a)


def _sptk_c2acr_r0(c, fft_len):
    """``c2acr -m <len-1> -M 0 -l fft_len``, frame by frame as the tool does: zero-pad the cepstrum to fft_len, real FFT,
    x = exp(2 Re C) on every bin, inverse FFT, r[0] = x-sum / fft_len."""
    c = np.atleast_2d(c)
    out = np.empty(c.shape[0])
    for f in range(c.shape[0]):
        x = np.zeros(fft_len)
        x[:c.shape[1]] = c[f]
        re = np.fft.rfft(x).real                       # fftr: bins 0 .. l/2; the upper half mirrors them
        p = np.exp(2.0 * re)
        out[f] = (p[0] + p[-1] + 2.0 * np.sum(p[1:-1])) / fft_len
    return out


def _sptk_mc2b(mc, a):
    """``mc2b -m order -a a``: b[order] = mc[order]; b[m] = mc[m] - a b[m+1] downwards (per frame)."""
    b = np.array(mc, dtype=np.float64)
    for f in range(b.shape[0]):
        for m in range(b.shape[1] - 2, -1, -1):
            b[f, m] = mc[f, m] - a * b[f, m + 1]
    return b


def _sptk_b2mc(b, a):
    """``b2mc -m order -a a``: d = b[order]; for m = order-1 .. 0: o = b[m] + a d; d = b[m]; mc[m] = o."""
    b = np.asarray(b, dtype=np.float64)
    mc = np.array(b)
    for f in range(b.shape[0]):
        d = b[f, -1]
        for m in range(b.shape[1] - 2, -1, -1):
            o = b[f, m] + a * d
            d = b[f, m]
            mc[f, m] = o
    return mc


def post_filter_merlin(m_mag_mel_log, fs, pf_coef=1.4):
    """
    magphase.py:3375-3465, command by command (each ``|`` and each temp file is a float32 stream):
      temp.mcep   = la.rceps(mag, 'log', 'compact')                                               (:3398-3399)
      temp.lift   = echo 1 1 pf pf ... | x2x +af          (pf printed with %1.2f)                   (:3404, :3418-3419)
      temp.r0     = freqt -m n-1 -a alpha -M 2047 -A 0 < mcep | c2acr -m 2047 -M 0 -l 4096          (:3422-3424)
      temp.p_r0   = vopr -m mcep lift | freqt ... | c2acr ...                                       (:3426-3429)
      temp.b0     = vopr -m mcep lift | mc2b -a alpha | bcp -s 0 -e 0                               (:3432-3434)
      temp.p_b0   = vopr -d r0 p_r0 | sopr -LN -d 2 | vopr -a b0                                    (:3437-3439)
      temp.mcep_pf= vopr -m mcep lift | mc2b | bcp -s 1 -e order | merge -s 0 -N 0 p_b0 | b2mc      (:3442-3445)
      out         = la.mcep_to_sp_cosmat(mcep_pf, n, alpha=0.0, out_type='log'); NaN -> la.MAGIC    (:3451-3455)
    PARITY UNPINNED: SPTK-3.9 (x2x, freqt, c2acr, vopr, sopr, mc2b, bcp, merge, b2mc) is an external binary package absent
    from /root/reference and from this image; its tools are restated from their published algorithms (Tokuda et al.
    1994 for freqt; the MLSA coefficient recursions for mc2b / b2mc).  The two legs that ARE the reference's own Python
    (la.rceps, la.mcep_to_sp_cosmat) are pinned by golden g12.  Written independently of magphase_amd.hostmath's
    table form (matrices) -- frame-by-frame recursions here.
    """
    m = np.asarray(m_mag_mel_log, dtype=np.float64)
    fft_len = 4096
    minph_ord = fft_len // 2 - 1
    alpha = define_alpha(fs)
    n = m.shape[1]
    mcep = _pipe(rceps_compact(m))
    lift = _pipe(np.array([1.0, 1.0] + [float("%1.2f" % pf_coef)] * (n - 2)))
    r0 = _pipe(_sptk_c2acr_r0(_pipe(_sptk_freqt_aA(mcep, minph_ord, alpha, 0.0)), fft_len))
    mcep_w = _pipe(mcep * lift[None, :])                                        # vopr -m
    p_r0 = _pipe(_sptk_c2acr_r0(_pipe(_sptk_freqt_aA(mcep_w, minph_ord, alpha, 0.0)), fft_len))
    b = _pipe(_sptk_mc2b(mcep_w, alpha))
    b0 = b[:, 0]                                                                # bcp -s 0 -e 0
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = _pipe(r0 / p_r0)                                                # vopr -d
        half_ln = _pipe(np.log(ratio) / 2.0)                                    # sopr -LN -d 2
    p_b0 = _pipe(half_ln + b0)                                                  # vopr -a
    b_pf = np.hstack((p_b0[:, None], b[:, 1:]))                                 # bcp -s 1 -e order | merge -s 0 -N 0
    mcep_pf = _pipe(_sptk_b2mc(b_pf, alpha))
    out = mcep_to_sp_cosmat(mcep_pf, n, alpha=0.0, out_type="log")
    out[np.isnan(out)] = MAGIC
    return out


def frm_list_to_matrix(frames, v_shift, nfft):
    """libaudio.py:122-140: each ragged frame placed so that its epoch lands on index nfft/2."""
    m = np.zeros((len(v_shift), nfft))
    for i, fr in enumerate(frames):
        start = nfft // 2 - v_shift[i]
        if start < 0 or start + len(fr) > nfft:
            raise ValueError("negative dimensions are not allowed")  # np.zeros(<0) in the reference
        m[i, start : start + len(fr)] = fr
    return m


def noise_gain(m_ns_mag_rows):
    """magphase.py:902-903 (Q10): sqrt(exp(mean(log(|N|)**2))) over bins 1..H-2 -- square OF THE LOG."""
    with np.errstate(invalid="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        return np.sqrt(np.exp(np.mean(log_protected(m_ns_mag_rows[:, 1:-1]) ** 2)))


def build_min_phase_from_mag_spec(m_mag):
    """libaudio.py:920-934 -- complex-cepstrum minimum-phase spectrum of a magnitude spectrum."""
    half = m_mag.shape[1]
    ceps = np.fft.ifft(add_hermitian_half_real(log_protected(m_mag))).real
    ceps[:, half:] = 0.0
    ceps[:, 1 : (half - 1)] *= 2.0
    return np.exp(np.fft.fft(ceps)[:, :half])


def sp_mel_unwarp_fbank(m_mag_mel, nbins, alpha=0.77, interp_kind="quadratic"):
    """libaudio.py:815-864 (sp_mel_unwarp_fbank -> unwarp_from_fbank): per-frame interp1d through the band centres."""
    from scipy import interpolate

    m_mag_mel = np.asarray(m_mag_mel, dtype=np.float64)
    nfrms, n_melbands = m_mag_mel.shape
    v_bins_warp = build_mel_curve(alpha, nbins, amp=np.pi)
    v_cntrs_mel = np.linspace(0, v_bins_warp[-1], n_melbands)
    f_interp = interpolate.interp1d(v_bins_warp, np.arange(nbins), kind=interp_kind)
    v_cntrs = round_to_int(f_interp(v_cntrs_mel))
    v_bins = np.arange(nbins)
    m_mag = np.zeros((nfrms, nbins))
    for nxf in range(nfrms):
        m_mag[nxf, :] = interpolate.interp1d(v_cntrs, m_mag_mel[nxf, :], kind=interp_kind)(v_bins)
    return m_mag


def synthesis_from_compressed(m_mag_mel_log, m_real_mel, m_imag_mel, v_lf0, fs, fft_len=None, b_voi_ap_win=True,
                              b_const_rate=False, per_phase_type="magphase", alpha_phase=None, b_out_hpf=True,
                              v_noise=None, return_debug=False, b_fbank_mel=False):
    """
    magphase.py:825-997.  b_fbank_mel: magnitudes unwarped by la.sp_mel_unwarp_fbank (magphase.py:851-852).
    ``v_noise``: if given, used instead of the
    ``np.random.uniform(-1, 1, ns_len)`` draw of magphase.py:883 (must have length ns_len).
    """
    cf, bw = define_crossfade_params(fs)
    alpha = define_alpha(fs)
    if fft_len is None:
        fft_len = define_fft_len(fs)
    half = fft_len // 2 + 1
    nfrms = m_mag_mel_log.shape[0]

    v_f0 = np.exp(v_lf0)
    v_voi = v_f0 > 1.0
    v_shift = f0_to_shift(v_f0, fs)
    if b_fbank_mel:
        m_mag = np.exp(sp_mel_unwarp_fbank(m_mag_mel_log, half, alpha=alpha))
    else:
        m_mag = np.exp(sp_mel_unwarp(m_mag_mel_log, half, alpha=alpha, in_type="log"))
    if alpha_phase is None:
        alpha_phase = alpha
    m_real, m_imag = phase_uncompress_type1_mcep(m_real_mel, m_imag_mel, alpha_phase, fft_len, fs)

    if b_const_rate:
        v_shift, v_locs = get_shifts_and_frm_locs_from_const_shifts(v_shift, 5.0, fs)
        m_mag = interp_from_const_to_variable_rate(m_mag, v_locs, 5.0, fs)
        m_real = interp_from_const_to_variable_rate(m_real, v_locs, 5.0, fs)
        m_imag = interp_from_const_to_variable_rate(m_imag, v_locs, 5.0, fs)
        v_voi = interp_from_const_to_variable_rate(v_voi, v_locs, 5.0, fs) > 0.5
        v_f0 = shift_to_f0(v_shift, v_voi, fs)
        nfrms = v_shift.size

    # periodic/aperiodic crossfade mask (voiced rows only)
    v_curve = spectral_crossfade_lowpass_curve(half, cf, bw, fs)
    m_mask = np.zeros(m_mag.shape)
    m_mask[v_voi, :] = v_curve[None, :]

    # noise
    v_shift = v_shift.astype(int)
    v_pm = shift_to_pm(v_shift)
    ns_len = v_pm[-1] + (v_pm[-1] - v_pm[-2])
    if v_noise is None:
        v_noise = np.random.uniform(-1, 1, ns_len)
    assert len(v_noise) == ns_len
    wins = [np.hanning] * nfrms
    if b_voi_ap_win:
        wins = [voi_noise_window if v_voi[i] else np.hanning for i in range(nfrms)]
    frames, _, _, _, _ = windowing(v_noise, v_pm, win_func=wins)
    m_ns = frm_list_to_matrix(frames, v_shift, fft_len)
    m_ns_spec = np.fft.fft(np.fft.fftshift(m_ns, axes=1))[:, :half]
    m_ns_mag = np.absolute(m_ns_spec)
    g_voi = noise_gain(m_ns_mag[v_voi, :])
    g_unv = noise_gain(m_ns_mag[~v_voi, :])
    m_ns_spec[v_voi, :] = m_ns_spec[v_voi, :] / g_voi
    m_ns_spec[~v_voi, :] = m_ns_spec[~v_voi, :] / g_unv

    # aperiodic component (+ unvoiced tilt, Q13)
    m_ap = m_ns_spec * m_mag
    m_ap[~v_voi, :] *= db(build_mel_curve(alpha, half, amp=3.5) - 3.5, b_inv=True)

    # periodic component (+ voiced tilt, Q13)
    if per_phase_type == "magphase":
        ph = m_real + 1j * m_imag
        ph_mag = np.absolute(ph)
        ph_mag[ph_mag == 0.0] = 1.0
        m_per = m_mag * (ph / ph_mag)
    elif per_phase_type == "linear":
        m_per = m_mag.astype(complex)
    elif per_phase_type == "min_phase":
        m_per = build_min_phase_from_mag_spec(m_mag)
    else:
        raise ValueError("per_phase_type")
    m_per[v_voi, :] *= db(build_mel_curve(0.6, half, amp=2.0), b_inv=True)

    # masks (Q12)
    m_per = m_per * (m_mask ** 0.5)
    m_ap = m_ap * ((1 - m_mask) ** 0.5)
    m_per[m_mask == 0.0] = 0
    m_ap[m_mask == 1.0] = 0
    m_syn = m_per + m_ap
    m_syn[:, 0] = np.absolute(m_syn[:, 0])
    m_syn[:, -1] = np.absolute(m_syn[:, -1])

    m_frms = np.fft.fftshift(np.fft.ifft(hermitian_full_spectrum(m_syn)).real, axes=1)

    # anti-ringing window (Q14)
    se = np.r_[v_shift[0], v_shift, v_shift[-1], v_shift[-1]]
    for n in range(nfrms):
        m_frms[n, :] *= centred_window(se[n] + se[n + 1], se[n + 2] + se[n + 3], fft_len, raised_hanning, True)

    v_syn = ola(m_frms, v_pm)
    v_pre_hpf = v_syn
    if b_out_hpf:
        b_, a_ = signal.butter(4, 40 / (fs / 2.0), btype="highpass")
        v_syn = signal.lfilter(b_, a_, v_syn)
    if return_debug:
        return v_syn, dict(v_shift=v_shift, v_pm=v_pm, g_voi=g_voi, g_unv=g_unv, v_voi=v_voi, ns_len=ns_len,
                           v_pre_hpf=v_pre_hpf, m_mag=m_mag, m_real=m_real, m_imag=m_imag, m_syn=m_syn)
    return v_syn


def normalise_for_wav(v_signal, norm=0.98):
    """libaudio.py:352-363 (Q17) -- the scaling applied before the wav write."""
    return norm * v_signal / np.max(np.abs(v_signal))
