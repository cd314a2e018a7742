"""
Python-3 counterpart of the src/libaudio.py names used by the reference's scripts and by magphase.py
(SURVEY.md section 8b): wav / .est I/O, dB, protected logs.  Signal arithmetic of the hot path is in the HIP
library; the small float64 vectors here are constants and file glue.
"""
import os
import shutil
import subprocess
import wave

import numpy as np

from . import hostmath as hm

MAGIC = hm.MAGIC


def read_audio_file(filepath):
    """What ``soundfile.read`` gives the reference (magphase.py:2872): float64 in [-1,1), fs.  Mono only."""
    try:
        import soundfile as sf  # the reference's own reader, when installed

        return sf.read(filepath)
    except ImportError:
        pass
    with wave.open(filepath, "rb") as w:
        if w.getnchannels() != 1:
            raise ValueError("mono wav expected")
        fs, width, raw = w.getframerate(), w.getsampwidth(), w.readframes(w.getnframes())
    if width == 2:
        return np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0, fs
    if width == 4:
        return np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0, fs
    raise ValueError("unsupported wav sample width %d" % width)


def write_audio_file(filepath, v_signal, fs, norm=0.98):
    """libaudio.py:352-365 (Q17): peak-normalise to ``norm`` then write 16-bit PCM."""
    v_signal = np.asarray(v_signal, dtype=np.float64)
    if norm is not None:
        v_signal = norm * v_signal / np.max(np.abs(v_signal))
    try:
        import soundfile as sf

        sf.write(filepath, v_signal, fs)
        return
    except ImportError:
        pass
    pcm = np.clip(np.round(v_signal * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(filepath, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(fs))
        w.writeframes(pcm.tobytes())


def read_reaper_est_file(est_file, check_len_smpls=-1, fs=-1, skiprows=7, usecols=[0, 1]):
    """libaudio.py:421-447."""
    if (check_len_smpls > 0) and (fs == -1):
        raise ValueError("If check_len_smpls given, fs must be provided as well.")
    m_data = np.atleast_2d(np.loadtxt(est_file, skiprows=skiprows, usecols=usecols))
    return hm.clean_epochs(m_data[:, 0], m_data[:, 1], check_len_smpls, fs)


def find_reaper():
    """libaudio.py:20-34: tools/bin/reaper next to the package, or [TOOLS] bin_dir, or PATH."""
    cand = [os.environ.get("MAGPHASE_REAPER_BIN", ""),
            os.path.realpath(os.path.join(os.path.dirname(__file__), "..", "tools", "bin", "reaper")),
            shutil.which("reaper") or ""]
    for c in cand:
        if c and os.path.isfile(c) and os.access(c, os.X_OK):
            return c
    return None


def reaper(in_wav_file, out_est_file):
    """libaudio.py:450-455 -- same command line; REAPER is an external binary (out of scope)."""
    binary = find_reaper()
    if binary is None:
        raise RuntimeError("REAPER binary not found (set MAGPHASE_REAPER_BIN, or provide epochs: a <wav>.est "
                           "file next to the wav, or magphase.set_epoch_provider)")
    subprocess.call("%s -s -x 400 -m 50 -a -u 0.005 -i %s -p %s" % (binary, in_wav_file, out_est_file), shell=True)


def db(m_data, b_inv=False):
    """libaudio.py:635-639."""
    if not b_inv:
        return 20 * np.log10(m_data)
    return 10 ** (m_data / 20)


def log(m_x):
    """libaudio.py:241-248."""
    with np.errstate(divide="ignore", invalid="ignore"):
        m_y = np.array(np.log(m_x), dtype=np.float64)
    m_y[~np.isfinite(m_y)] = MAGIC
    return m_y


def f0_to_lf0(v_f0):
    """libaudio.py:458-465."""
    with np.errstate(divide="ignore"):
        v_lf0 = np.log(v_f0)
    v_lf0[np.isinf(v_lf0)] = MAGIC
    return v_lf0


def shift_to_pm(v_shift):
    """libaudio.py:60-62."""
    return np.cumsum(v_shift)


def pm_to_shift(v_pm):
    """libaudio.py:65-67."""
    return np.diff(np.hstack((0, v_pm)))


def convert_label_state_align_to_var_frame_rate(in_lab_st_file, v_dur_state, out_lab_st_file):
    """
    libaudio.py:687-708: rewrites the times of an HTS state-aligned label file so that state i lasts
    v_dur_state[i] frames of 5 ms -- the "variable frame rate" labels a constant-rate trainer is given.
    Times are written in units of 100 ns, the label strings (third column) are kept.
    """
    with open(in_lab_st_file, "r") as f:
        l_names = [ln.split(" ")[2].rstrip("\n") for ln in f if ln.strip()]
    v_edges = np.concatenate(([0.0], np.cumsum(np.asarray(v_dur_state, dtype=np.float64) * 5.0 * 10000.0)))
    v_edges = v_edges.astype(int)
    with open(out_lab_st_file, "w") as f:
        for i, name in enumerate(l_names):
            f.write("%d %d %s\n" % (v_edges[i], v_edges[i + 1], name))


# ---- helpers of the reference's libaudio that the live path is built from, for callers that use them directly
read_est_file = read_reaper_est_file      # libaudio.py:421 under its older name
build_mel_curve = hm.build_mel_curve      # libaudio.py:711-718


def hz_to_bin(v_hz, nFFT, fs):
    """libaudio.py:151-152."""
    return v_hz * nFFT / float(fs)


def bin_to_hz(v_bin, nFFT, fs):
    """libaudio.py:154-155."""
    return v_bin * fs / float(nFFT)


def add_hermitian_half(m_data, data_type="mag"):
    """libaudio.py:369-388: [F x H] half spectra -> [F x 2(H-1)] ('phase' zeroes DC/Nyquist IN PLACE like the reference)."""
    if data_type in ("mag", "magnitude"):
        return np.hstack((m_data, np.fliplr(m_data[:, 1:-1])))
    if data_type == "phase":
        m_data[:, 0] = 0
        m_data[:, -1] = 0
        return np.hstack((m_data, -np.fliplr(m_data[:, 1:-1])))
    if data_type == "zeros":
        return np.hstack((m_data, np.zeros((m_data.shape[0], m_data.shape[1] - 2))))
    if data_type == "complex":
        return add_hermitian_half(m_data.real) + 1j * add_hermitian_half(m_data.imag, data_type="phase")
    return m_data


def remove_hermitian_half(m_data):
    """libaudio.py:392-400."""
    m_data = np.asarray(m_data)
    if m_data.ndim == 1:
        return m_data[:m_data.size // 2 + 1].copy()
    return m_data[:, :m_data.shape[1] // 2 + 1].copy()


def sp_mel_unwarp(m_sp_mel, nbins_out, alpha=0.77, in_type="log"):
    """
    libaudio.py:667-684 on the device (mpx_mel_unwarp, the GEMM against hostmath.unwarp_matrix): [F x n] mel-warped
    log (or 'abs') spectra -> [F x nbins_out], float64 like the reference's.
    """
    from .engine import get_engine
    e = get_engine()
    x = np.log(m_sp_mel) if in_type == "abs" else np.asarray(m_sp_mel, dtype=np.float64)
    out = e.mel_unwarp_single(x, int(nbins_out), float(alpha), exp_out=(in_type == "abs"))
    return out


def build_min_phase_from_mag_spec(m_mag):
    """
    libaudio.py:920-934 on the device (mpx_min_phase: complex cepstrum, causal fold): [F x H] magnitudes -> complex
    [F x H] minimum-phase spectra |X| e^{j phi}.  H - 1 must be 512, 1024 or 2048.
    """
    from .engine import get_engine
    return get_engine().min_phase_single(np.asarray(m_mag, dtype=np.float64))
