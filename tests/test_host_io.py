"""
Host-side file helpers of the library (csrc/magphase_host.cpp through la.read_est_batch / write_files_batch /
read_files_batch): same values as the numpy calls the reference makes (np.loadtxt of the REAPER track, libaudio.py:421-447;
ndarray.tofile / np.fromfile, libutils.py:193-211), per-file errors instead of a failed batch.  No GPU needed.
"""
import os

import numpy as np
import pytest

from magphase_amd import libaudio as la


def _write_est(path, rows, header=7):
    with open(path, "w") as fh:
        for i in range(header - 1):
            fh.write("header line %d\n" % i)
        fh.write("EST_Header_End\n")
        for r in rows:
            fh.write(r + "\n")


def test_est_batch_equals_loadtxt(tmp_path):
    rng = np.random.default_rng(0)
    files = []
    for k in range(5):
        t = np.cumsum(rng.uniform(0.002, 0.012, 400 + 50 * k))
        rows = ["%.6f %d %.3f" % (x, rng.integers(0, 2), rng.uniform(-1, 300)) for x in t]
        _write_est(tmp_path / ("a%d.est" % k), rows)
        files.append(str(tmp_path / ("a%d.est" % k)))
    got = la.read_est_batch(files)
    for f, (c0, c1) in zip(files, got):
        ref = np.loadtxt(f, skiprows=7, usecols=[0, 1])
        assert np.array_equal(c0, ref[:, 0]) and np.array_equal(c1, ref[:, 1])      # bit-equal float64
        f0, f1 = la.read_est_fast(f)
        assert np.array_equal(c0, f0) and np.array_equal(c1, f1)


def test_est_number_forms_and_blank_lines(tmp_path):
    rows = ["0.1 1 0", "", "  2.5e-3\t0  7", "-0.000001 1 0", "123456789.123456789012345 0 0", "1e3 1 0", "   ",
            "0.30000000000000004 1 0", "+4.25 0 1", "17 1 0"]
    _write_est(tmp_path / "n.est", rows)
    (c0, c1), = la.read_est_batch([str(tmp_path / "n.est")])
    want = [float(r.split()[0]) for r in rows if r.strip()]
    assert c0.tolist() == want                       # Python's float() is correctly rounded: so must these be
    assert c1.tolist() == [float(r.split()[1]) for r in rows if r.strip()]


def test_est_errors_are_per_file(tmp_path):
    _write_est(tmp_path / "ok.est", ["0.1 1 0", "0.2 0 0"])
    _write_est(tmp_path / "bad.est", ["0.1 1 0", "0.2"])
    _write_est(tmp_path / "empty.est", [])
    r = la.read_est_batch([str(tmp_path / "ok.est"), str(tmp_path / "missing.est"), str(tmp_path / "bad.est"),
                           str(tmp_path / "empty.est")])
    assert r[0][0].tolist() == [0.1, 0.2]
    assert isinstance(r[1], FileNotFoundError)
    assert isinstance(r[2], ValueError)
    assert r[3][0].size == 0 and r[3][1].size == 0
    assert la.read_est_batch([]) == []


@pytest.mark.parametrize("threads", ["1", "4"])
def test_write_and_read_files_round_trip(tmp_path, monkeypatch, threads):
    monkeypatch.setenv("MAGPHASE_IO_NATIVE_THREADS", threads)
    rng = np.random.default_rng(1)
    arrs = [rng.standard_normal((50 + k, 7)).astype(np.float32) for k in range(9)] + [np.zeros((0, 7), np.float32)]
    paths = [str(tmp_path / ("f%d.mag" % k)) for k in range(len(arrs))]
    st = la.write_files_batch(paths, arrs)
    assert st == [None] * len(arrs)
    for p, a in zip(paths, arrs):
        assert np.array_equal(np.fromfile(p, dtype=np.float32), a.reshape(-1))        # what tofile would have written
    back = la.read_files_batch(paths + [str(tmp_path / "nope.mag")])
    for a, b in zip(arrs, back[:-1]):
        assert b.dtype == np.float32 and np.array_equal(a.reshape(-1), b)
    assert isinstance(back[-1], FileNotFoundError)
    # a failing path does not stop the others
    st = la.write_files_batch([str(tmp_path / "no_dir" / "x.mag"), paths[0]], [arrs[0], arrs[1]])
    assert isinstance(st[0], OSError) and st[1] is None
    assert np.array_equal(np.fromfile(paths[0], dtype=np.float32), arrs[1].reshape(-1))


def test_wav_written_with_header(tmp_path):
    pcm = (np.arange(-500, 500) * 13).astype(np.int16)
    p = str(tmp_path / "x.wav")
    assert la.write_files_batch([p], [pcm], [la.wav_header_pcm16(pcm.size, 16000)]) == [None]
    q = str(tmp_path / "y.wav")
    la.write_pcm16_file(q, pcm, 16000)
    assert open(p, "rb").read() == open(q, "rb").read()
    sig, fs = la.read_audio_file_pcm(p)
    assert fs == 16000 and np.array_equal(sig, pcm)


def test_wav_batch_reader_formats_and_errors(tmp_path):
    """la.read_audio_files_pcm_batch: the files' bytes in one native call, RIFF parsing as read_audio_file_pcm (16-bit PCM
    stays int16; 24 / 32-bit PCM and float32 become float64 in [-1, 1)); a bad file costs one entry."""
    import struct
    rng = np.random.default_rng(2)
    pcm = rng.integers(-32768, 32767, 3000).astype(np.int16)
    la.write_pcm16_file(str(tmp_path / "a.wav"), pcm, 16000)
    f32 = rng.uniform(-1, 1, 777).astype("<f4")
    with open(tmp_path / "f.wav", "wb") as fh:
        fh.write(struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + f32.nbytes, b"WAVE", b"fmt ", 16, 3, 1, 48000, 48000 * 4, 4, 32,
                             b"data", f32.nbytes))
        fh.write(f32.tobytes())
    i24 = rng.integers(-(1 << 23), (1 << 23) - 1, 500)
    b24 = np.stack([(i24 >> s) & 0xFF for s in (0, 8, 16)], axis=1).astype(np.uint8)
    with open(tmp_path / "t.wav", "wb") as fh:
        fh.write(struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + b24.size, b"WAVE", b"fmt ", 16, 1, 1, 22050, 22050 * 3, 3, 24,
                             b"data", b24.size))
        fh.write(b24.tobytes())
    (tmp_path / "junk.wav").write_bytes(b"not a wav at all")
    paths = [str(tmp_path / n) for n in ("a.wav", "f.wav", "missing.wav", "t.wav", "junk.wav")]
    r = la.read_audio_files_pcm_batch(paths)
    assert r[0][1] == 16000 and r[0][0].dtype == np.int16 and np.array_equal(r[0][0], pcm)
    assert r[1][1] == 48000 and np.array_equal(r[1][0], f32.astype(np.float64))
    assert isinstance(r[2], FileNotFoundError)
    assert r[3][1] == 22050 and np.array_equal(r[3][0], i24 / 8388608.0)
    assert isinstance(r[4], ValueError)
    for p, got in zip(paths, r):
        if not isinstance(got, Exception):
            one = la.read_audio_file_pcm(p)
            assert one[1] == got[1] and np.array_equal(one[0], got[0])


def test_mkdir_is_race_free_between_ranks(tmp_path):
    """Two ranks of a batch job create the same output directory at the same moment (libutils.py:146-156 checks and
    then creates: the loser of that race raised FileExistsError)."""
    import threading
    from magphase_amd import libutils as lu
    errs = []
    for rep in range(50):
        d = str(tmp_path / ("out%d" % rep))
        gate = threading.Barrier(8)

        def work():
            gate.wait()
            try:
                lu.mkdir(d)
            except Exception as e:  # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=work) for _ in range(8)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert os.path.isdir(d)
    assert not errs
    f = tmp_path / "afile"
    f.write_text("x")
    with pytest.raises(FileExistsError):
        lu.mkdir(str(f))


def test_medfilt3_batch_equals_scipy_bit_for_bit():
    import warnings
    from scipy import signal
    from magphase_amd import hostmath as hm
    rng = np.random.RandomState(0)
    vs = [rng.uniform(0, 300, n) * (rng.rand(n) > 0.3) for n in (1, 2, 3, 7, 500, 5)]
    vs += [np.array([np.inf, 0.0, 100.0]), np.array([5.0, np.inf, np.inf, 3.0])]      # f0 = inf where the shift is 0 (Q2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for v, o in zip(vs, hm.medfilt3_batch(vs)):
            assert np.array_equal(signal.medfilt(v), o)
    assert hm.medfilt3_batch([]) == []


def test_native_copy_many():
    import ctypes
    from magphase_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(1)
    arrs = [rng.randint(-30000, 30000, n).astype(np.int16) for n in (1, 0, 100000, 480001, 7)]
    nb = np.asarray([a.nbytes for a in arrs], dtype=np.int64)
    off = np.concatenate(([0], np.cumsum(nb)[:-1])).astype(np.int64)
    dst = np.zeros(int(nb.sum()) // 2, dtype=np.int16)
    src = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    assert lib.mpx_host_copy_many(len(arrs), src, nb.ctypes.data, off.ctypes.data, dst.ctypes.data, 8) == 0
    assert np.array_equal(dst, np.concatenate(arrs))
    assert lib.mpx_host_copy_many(-1, src, nb.ctypes.data, off.ctypes.data, dst.ctypes.data, 8) != 0
