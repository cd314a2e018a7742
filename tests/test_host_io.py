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
