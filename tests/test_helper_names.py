"""
The helper names of the reference's modules that Merlin-side callers import besides the main entry points
(mp.windowing, mp.ola, mp.interp_from_variable_to_const_frm_rate, mp.get_shifts_and_frm_locs_from_const_shifts,
la.sp_mel_warp, la.mcep_to_sp_cosmat; /root/reference/src/magphase.py:74,34,2219,1426, libaudio.py:643,605) against the
oracle.  Host forms: exact on CPU; device-backed forms (ola, sp_mel_warp): -m gpu.
"""
import numpy as np
import pytest

from magphase_amd import synthetic as syn
from oracle import magphase_oracle as orc


def _import_api():
    # the drop-in modules import without a GPU; only calls into the engine need one
    from magphase_amd import libaudio as la, magphase as mp
    return mp, la


def test_names_exist_in_the_dropin_and_src_shims():
    mp, la = _import_api()
    for n in ("windowing", "ola", "interp_from_variable_to_const_frm_rate", "interp_from_const_to_variable_rate",
              "get_shifts_and_frm_locs_from_const_shifts"):
        assert callable(getattr(mp, n))
    for n in ("sp_mel_warp", "mcep_to_sp_cosmat", "sp_mel_unwarp", "gen_non_symmetric_win", "gen_centr_win"):
        assert callable(getattr(la, n))


def test_windowing_matches_oracle_exactly():
    mp, _la = _import_api()
    pcm, pm, _voi = syn.make_utterance(11, dur_s=0.4, fs=16000)
    x = syn.pcm_to_float(pcm)
    for wf in (np.hanning, None):
        got = mp.windowing(x, pm * 16000, win_func=wf) if wf else mp.windowing(x, pm * 16000, win_func=None)
        if wf is None:
            ref = orc.windowing(x, pm * 16000, win_func=lambda n: np.ones(n))
        else:
            ref = orc.windowing(x, pm * 16000, win_func=wf)
        assert len(got[0]) == len(ref[0])
        for a, b in zip(got[0], ref[0]):
            assert np.array_equal(a, b)
        for k in (1, 2, 3, 4):
            assert np.array_equal(got[k], ref[k])
    lst = [np.hanning if f % 2 else orc.voi_noise_window for f in range(len(ref[0]))]
    got, ref = mp.windowing(x, pm * 16000, win_func=lst), orc.windowing(x, pm * 16000, win_func=lst)
    assert all(np.array_equal(a, b) for a, b in zip(got[0], ref[0]))


def test_rate_interpolations_match_oracle_exactly():
    mp, _la = _import_api()
    rng = np.random.RandomState(3)
    pm = np.cumsum(rng.randint(80, 400, 60))
    m = rng.randn(60, 7)
    a = mp.interp_from_variable_to_const_frm_rate(m, pm, 5.0, 48000)
    b = orc.interp_from_variable_to_const_frm_rate(m, pm, 5.0, 48000)
    assert a.shape == b.shape and np.array_equal(a, b)
    v = mp.interp_from_variable_to_const_frm_rate(m[:, 0], pm, 5.0, 48000)
    assert v.ndim == 1 and np.array_equal(v, b[:, 0])
    locs = np.sort(rng.uniform(240, 240 * 59, 40))
    c = mp.interp_from_const_to_variable_rate(m, locs, 5.0, 48000)
    assert np.array_equal(c, orc.interp_from_const_to_variable_rate(m, locs, 5.0, 48000))


def test_const_shift_scan_matches_oracle_exactly():
    mp, _la = _import_api()
    rng = np.random.RandomState(5)
    shift_c = rng.uniform(150, 500, 200)
    s1, l1 = mp.get_shifts_and_frm_locs_from_const_shifts(shift_c, 5.0, 48000)
    s2, l2 = orc.get_shifts_and_frm_locs_from_const_shifts(shift_c, 5.0, 48000)
    assert np.array_equal(s1, s2) and np.array_equal(l1, l2)


def test_mcep_to_sp_cosmat_and_windows_match_oracle():
    _mp, la = _import_api()
    rng = np.random.RandomState(7)
    mc = rng.randn(5, 60) * 0.1
    for out_type in ("abs", "log", "db"):
        a = la.mcep_to_sp_cosmat(mc, 2049, alpha=0.77, out_type=out_type)
        b = orc.mcep_to_sp_cosmat(mc, 2049, alpha=0.77, out_type=out_type)
        assert np.max(np.abs(a - b)) <= 1e-12 * max(1.0, np.max(np.abs(b)))
    for L, R in ((0, 5), (7, 0), (12, 31)):
        assert np.array_equal(la.gen_non_symmetric_win(L, R, np.hanning), orc.half_windows(L, R, np.hanning))
    w = la.gen_centr_win(10, 20, 256, win_func=np.hanning)
    assert w.shape == (256,) and w[128] == 1.0 and w[128 - 10] == 0.0 and np.count_nonzero(w) == 29


@pytest.mark.gpu
def test_ola_device_and_host_forms_match_oracle():
    mp, _la = _import_api()
    rng = np.random.RandomState(9)
    for frmlen, first in ((4096, 300), (2048, 1500), (512, 100)):     # 2048 with pm[0] > frmlen/2: python negative slice
        pm = first + np.cumsum(np.r_[0, rng.randint(100, 600, 30)])
        m = rng.randn(pm.size, frmlen)
        ref = orc.ola(m.copy(), pm)
        got = mp.ola(m.copy(), pm)
        assert got.shape == ref.shape
        tol = 1e-12 if frmlen == 512 else 2e-6 * np.max(np.abs(ref))   # host float64 / device float32 gather
        assert np.max(np.abs(got - ref)) <= tol
    # with the anti-ringing window: frames modified in place like the reference (magphase.py:48)
    pm = 200 + np.cumsum(np.r_[0, rng.randint(100, 300, 10)])
    m = rng.randn(pm.size, 1024)
    m2 = m.copy()
    out = mp.ola(m2, pm, win_func=np.hanning)
    assert not np.array_equal(m, m2) and np.all(np.isfinite(out))


@pytest.mark.gpu
def test_sp_mel_warp_matches_oracle():
    _mp, la = _import_api()
    rng = np.random.RandomState(13)
    sp = np.abs(rng.randn(40, 2049)) * 0.1 + 0.01
    for in_type, x in ((3, sp), (2, np.log(sp))):
        got = la.sp_mel_warp(x, 60, alpha=0.77, in_type=in_type)
        ref = orc.sp_mel_warp(x, 60, alpha=0.77, in_type=in_type)
        lg, lr = (np.log(got), np.log(ref)) if in_type == 3 else (got, ref)
        # the oracle quantises SPTK's files to float32; the device GEMM accumulates in float32: a few 1e-5 in the log domain
        assert got.shape == ref.shape and np.max(np.abs(lg - lr)) < 1e-4
