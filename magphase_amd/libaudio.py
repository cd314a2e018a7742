"""
Python-3 counterpart of the src/libaudio.py names used by the reference's scripts and by magphase.py
(SURVEY.md section 8b): wav / .est I/O, dB, protected logs.  Signal arithmetic of the hot path is in the HIP
library; the small float64 vectors here are constants and file glue.
"""
import os
import shutil
import subprocess

import numpy as np

from . import hostmath as hm

MAGIC = hm.MAGIC


def _read_wav_raw(filepath):
    """(samples as stored, fs, scale): mono PCM 16 / 24 / 32-bit or IEEE float32 RIFF wav, parsed directly (the wave
    module copies the frames twice and knows no float wavs)."""
    with open(filepath, "rb") as fh:
        buf = fh.read()
    return _parse_wav(buf, filepath)


def _parse_wav(buf, filepath="<buffer>"):
    """_read_wav_raw on the file's bytes (bytes, or a uint8 numpy array as read_files_batch returns them)."""
    import struct

    if isinstance(buf, np.ndarray):
        buf = memoryview(np.ascontiguousarray(buf, dtype=np.uint8)).cast("B")
    if len(buf) < 12 or bytes(buf[:4]) != b"RIFF" or bytes(buf[8:12]) != b"WAVE":
        raise ValueError("%s: not a RIFF/WAVE file" % filepath)
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(buf):
        cid, size = bytes(buf[pos:pos + 4]), struct.unpack_from("<I", buf, pos + 4)[0]
        if cid == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", buf, pos + 8)
            if fmt[0] == 0xFFFE and size >= 26:      # WAVE_FORMAT_EXTENSIBLE: the real tag is the sub-format's first word
                fmt = (struct.unpack_from("<H", buf, pos + 8 + 24)[0],) + fmt[1:]
        elif cid == b"data":
            data = (pos + 8, min(size, len(buf) - pos - 8))
            break
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError("%s: wav without fmt / data chunk" % filepath)
    tag, nch, fs, _rate, _align, bits = fmt
    if nch != 1:
        raise ValueError("mono wav expected")
    off, n = data
    if tag == 1 and bits == 16:
        return np.frombuffer(buf, dtype="<i2", count=n // 2, offset=off), fs, 1.0 / 32768.0
    if tag == 1 and bits == 32:
        return np.frombuffer(buf, dtype="<i4", count=n // 4, offset=off), fs, 1.0 / 2147483648.0
    if tag == 1 and bits == 24:
        b = np.frombuffer(buf, dtype=np.uint8, count=(n // 3) * 3, offset=off).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        return v - ((v & 0x800000) << 1), fs, 1.0 / 8388608.0
    if tag == 3 and bits == 32:
        return np.frombuffer(buf, dtype="<f4", count=n // 4, offset=off), fs, 1.0
    raise ValueError("unsupported wav format (tag %d, %d bits)" % (tag, bits))


def read_audio_file(filepath):
    """What ``soundfile.read`` gives the reference (magphase.py:2872): float64 in [-1,1), fs.  Mono only."""
    try:
        import soundfile as sf  # the reference's own reader, when installed

        return sf.read(filepath)
    except ImportError:
        pass
    v, fs, scale = _read_wav_raw(filepath)
    return v.astype(np.float64) * scale, fs


def read_audio_file_pcm(filepath):
    """(int16 PCM or float64 samples, fs): 16-bit wavs stay int16 -- the analysis plan converts them to float32 in one
    pass (exactly int16 / 32768), so the corpus reader skips a float64 round trip.  Same values as read_audio_file."""
    v, fs, scale = _read_wav_raw(filepath)
    if v.dtype == np.dtype("<i2"):
        return v, fs
    return v.astype(np.float64) * scale, fs


def read_audio_files_pcm_batch(paths):
    """read_audio_file_pcm for a list of files: the bytes of all files in one native call (read_files_batch), the RIFF
    headers parsed here.  [(samples, fs) | Exception] in input order."""
    out = []
    for p, raw in zip(paths, read_files_batch(paths, dtype=np.uint8)):
        if isinstance(raw, Exception):
            out.append(raw)
            continue
        try:
            v, fs, scale = _parse_wav(raw, p)
            out.append((v, fs) if v.dtype == np.dtype("<i2") else (v.astype(np.float64) * scale, fs))
        except (KeyboardInterrupt, SystemExit):
            raise
        except Exception as e:
            out.append(e)
    return out


def write_audio_file(filepath, v_signal, fs, norm=0.98):
    """libaudio.py:352-365 (Q17): peak-normalise to ``norm`` then write 16-bit PCM through soundfile.  Without the
    soundfile package the samples are converted like libsndfile converts floats to PCM_16 (lrint(x * 0x7FFF), no
    clipping of in-range input) and written behind a 44-byte RIFF header."""
    v_signal = np.asarray(v_signal, dtype=np.float64)
    if norm is not None:
        v_signal = norm * v_signal / np.max(np.abs(v_signal))       # the reference's expression, operation for operation
    try:
        import soundfile as sf

        sf.write(filepath, v_signal, fs)
        return
    except ImportError:
        pass
    pcm = np.rint(v_signal * 32767.0)
    np.clip(pcm, -32768.0, 32767.0, out=pcm)
    write_pcm16_file(filepath, pcm.astype("<i2"), fs)


def wav_header_pcm16(n_samples, fs):
    """The 44-byte RIFF header of a mono 16-bit wav with n_samples samples."""
    import struct

    n = int(n_samples) * 2
    return struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + n, b"WAVE", b"fmt ", 16, 1, 1, int(fs), int(fs) * 2, 2, 16,
                       b"data", n)


def write_pcm16_file(filepath, pcm, fs):
    """int16 samples -> mono 16-bit RIFF wav (44-byte header + the samples)."""
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    with open(filepath, "wb") as fh:
        fh.write(wav_header_pcm16(pcm.size, fs))
        fh.write(memoryview(pcm))


def read_est_fast(est_file, skiprows=7):
    """Columns 0 and 1 of a REAPER .est file (what np.loadtxt(est, skiprows=7, usecols=[0, 1]) returns, ~10x faster):
    the body is `time voicing f0` per line."""
    with open(est_file, "r") as fh:
        txt = fh.read()
    pos = 0
    for _ in range(skiprows):
        pos = txt.index("\n", pos) + 1
    body = txt[pos:]
    first = body.split("\n", 1)[0].split()
    vals = np.fromstring(body, dtype=np.float64, sep=" ")   # C strtod: correctly rounded, the values np.loadtxt gives
    m = vals.reshape(-1, max(len(first), 1))
    return m[:, 0], m[:, 1]


# ----------------------------------------------------------------------------------------------------
# many files at once through the library's host helpers (csrc/magphase_host.cpp): a few native threads, no GIL held
# ----------------------------------------------------------------------------------------------------
def _io_threads():
    return max(1, int(os.environ.get("MAGPHASE_IO_NATIVE_THREADS", "8")))


def _c_paths(paths):
    import ctypes

    return (ctypes.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])


def read_est_batch(est_files, skiprows=7):
    """read_est_fast for a list of files in one native call: [(v_time, v_voi) | Exception] in input order."""
    import ctypes

    from . import _lib

    lib, n = _lib.load(), len(est_files)
    if n == 0:
        return []
    cp = _c_paths(est_files)
    sizes = np.empty(n, dtype=np.int64)
    lib.mpx_host_file_sizes(n, cp, sizes.ctypes.data)
    rows = np.where(sizes > 0, sizes // 4 + 1, 1)          # a row is at least "0 0\n"
    off = np.concatenate(([0], np.cumsum(rows))).astype(np.int64)
    c0, c1 = np.empty(int(off[-1])), np.empty(int(off[-1]))
    counts = np.empty(n, dtype=np.int64)
    rc = lib.mpx_host_read_est_batch(n, cp, int(skiprows), off.ctypes.data, c0.ctypes.data, c1.ctypes.data,
                                     counts.ctypes.data, _io_threads())
    if rc != 0:
        raise RuntimeError("mpx_host_read_est_batch: error %d" % rc)
    out = []
    for i, f in enumerate(est_files):
        k = int(counts[i])
        if k < 0:
            out.append(OSError(-k, os.strerror(-k), f) if -k != 22 else ValueError("%s: malformed epoch file" % f))
        else:
            a = int(off[i])
            out.append((c0[a:a + k].copy(), c1[a:a + k].copy()))
    return out


def write_files_batch(paths, bodies, headers=None):
    """bodies[i] (C-contiguous numpy array or bytes) -> paths[i], optionally preceded by headers[i] (bytes), all in one
    native call.  Returns [None | OSError] in input order."""
    import ctypes

    from . import _lib

    lib, n = _lib.load(), len(paths)
    if n == 0:
        return []
    keep = [np.ascontiguousarray(b) if isinstance(b, np.ndarray) else np.frombuffer(b, dtype=np.uint8) for b in bodies]
    bp = (ctypes.c_void_p * n)(*[k.ctypes.data if k.size else None for k in keep])
    bn = np.array([k.nbytes for k in keep], dtype=np.int64)
    if headers is not None:
        hk = [np.frombuffer(h, dtype=np.uint8) for h in headers]
        hp = (ctypes.c_void_p * n)(*[k.ctypes.data for k in hk])
        hn = np.array([k.nbytes for k in hk], dtype=np.int64)
        hp_arg, hn_arg = hp, hn.ctypes.data
    else:
        hp_arg, hn_arg = None, None
    status = np.zeros(n, dtype=np.int32)
    rc = lib.mpx_host_write_files(n, _c_paths(paths), hp_arg, hn_arg, bp, bn.ctypes.data, status.ctypes.data, _io_threads())
    if rc != 0:
        raise RuntimeError("mpx_host_write_files: error %d" % rc)
    return [None if s == 0 else OSError(int(s), os.strerror(int(s)), p) for s, p in zip(status, paths)]


def read_files_batch(paths, dtype=np.float32):
    """np.fromfile(path, dtype) for a list of files in one native call: [array | OSError] in input order."""
    import ctypes

    from . import _lib

    lib, n = _lib.load(), len(paths)
    if n == 0:
        return []
    cp = _c_paths(paths)
    sizes = np.empty(n, dtype=np.int64)
    lib.mpx_host_file_sizes(n, cp, sizes.ctypes.data)
    item = np.dtype(dtype).itemsize
    bufs = [np.empty(max(int(s), 0) // item, dtype=dtype) for s in sizes]
    bp = (ctypes.c_void_p * n)(*[b.ctypes.data if b.size else None for b in bufs])
    cap = np.array([b.nbytes for b in bufs], dtype=np.int64)
    got = np.empty(n, dtype=np.int64)
    rc = lib.mpx_host_read_files(n, cp, bp, cap.ctypes.data, got.ctypes.data, _io_threads())
    if rc != 0:
        raise RuntimeError("mpx_host_read_files: error %d" % rc)
    out = []
    for i, p in enumerate(paths):
        if sizes[i] < 0 or got[i] < 0:
            e = int(-sizes[i]) if sizes[i] < 0 else int(-got[i])
            out.append(OSError(e, os.strerror(e), p))
        else:
            out.append(bufs[i][:int(got[i]) // item])
    return out


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


def gen_non_symmetric_win(left_len, right_len, win_func, b_norm=False):
    """libaudio.py:70-84: rising half of win_func(1 + 2 left) joined to the falling half of win_func(1 + 2 right)."""
    left_len, right_len = int(left_len), int(right_len)
    v_rise = win_func(1 + 2 * left_len)[:left_len + 1]
    v_fall = win_func(1 + 2 * right_len)[:right_len + 1][::-1]
    v_win = np.concatenate((v_rise, v_fall[1:]))
    return v_win / np.sum(v_win) if b_norm else v_win


def gen_centr_win(winlen_l, winlen_r, totlen, win_func=None, b_fill_w_bound_val=False):
    """libaudio.py:90-103: the non-symmetric window placed so that its centre sits at index totlen // 2."""
    v_short = gen_non_symmetric_win(winlen_l, winlen_r, win_func)
    first = int(np.floor(totlen / 2.0)) - int(winlen_l)
    v_win = np.zeros(totlen)
    if b_fill_w_bound_val:
        v_win += v_short[0]
    v_win[first:first + len(v_short)] = v_short
    return v_win


def mcep_to_sp_cosmat(m_mcep, n_spbins, alpha=0.77, out_type="abs"):
    """
    libaudio.py:605-631: mel-cepstra -> spectra by the cosine matrix on the all-pass warped axis (the reference fills the
    [n_cep x n_spbins] matrix in a Python double loop on every call; here it is one outer product, cached per
    configuration by hostmath).  Host float64 like the reference: out_type 'abs' (exp), 'db', 'log'.
    """
    from . import hostmath as hm

    m_mcep = np.asarray(m_mcep, dtype=np.float64)
    m_sp = np.dot(m_mcep, hm.cos_matrix(m_mcep.shape[1], int(n_spbins), float(alpha)))
    if out_type == "abs":
        return np.exp(m_sp)
    if out_type == "db":
        return m_sp * (20 / np.log(10))
    return m_sp


def sp_mel_warp(m_sp, nbins_out, alpha=0.77, in_type=3):
    """
    libaudio.py:643-661 on the device: SPTK ``mcep -j 0`` followed by the cosine matrix with alpha = 0 is ONE linear
    map of the log-periodogram (hostmath.warp_matrix; SPTK restated, parity unpinned) -- the GEMM of mpx_mel_warp.
    m_sp [F x H] with H - 1 in {512, 1024, 2048}; in_type 3: |f(w)|, 2: ln|f(w)|, 1: 20 log10|f(w)|.  Returns float64
    [F x nbins_out] of the same kind as the input ('abs' / 'log' / 'db').
    """
    from .engine import get_engine

    if in_type not in (1, 2, 3):
        raise ValueError("in_type must be 1, 2 or 3")
    m_sp = np.atleast_2d(np.asarray(m_sp, dtype=np.float64))
    x = m_sp if in_type == 3 else np.exp(m_sp if in_type == 2 else m_sp * (np.log(10.0) / 20.0))
    out = get_engine().mel_warp_single(x, int(nbins_out), float(alpha))      # ln-domain result
    if in_type == 3:
        return np.exp(out)
    return out if in_type == 2 else out * (20 / np.log(10))
