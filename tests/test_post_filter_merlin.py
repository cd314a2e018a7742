"""
Merlin-style post-filter (magphase.py:3375-3465).  PARITY UNPINNED: the reference runs nine SPTK
binaries that exist neither in the build container nor on the GPU box; these are known-answer properties of the
restated arithmetic (magphase_amd.magphase.post_filter_merlin).
"""
import numpy as np

from magphase_amd import hostmath as hm
from magphase_amd import magphase as mp


def _log_mel_mags(n_frames=40, dim=60, seed=5):
    rng = np.random.RandomState(seed)
    k = np.arange(dim)
    out = []
    for f in range(n_frames):
        f1, f2, f3 = rng.uniform(4, 12), rng.uniform(15, 28), rng.uniform(32, 50)
        env = -0.04 * k + 2.2 * np.exp(-0.5 * ((k - f1) / 2.0) ** 2) + 1.6 * np.exp(-0.5 * ((k - f2) / 2.5) ** 2) \
            + 1.0 * np.exp(-0.5 * ((k - f3) / 3.0) ** 2)
        out.append(env + rng.uniform(-6, -2))
    return np.array(out)


def test_mc2b_b2mc_are_inverse():
    rng = np.random.RandomState(0)
    m = rng.randn(7, 60)
    assert np.allclose(hm.sptk_b2mc(hm.sptk_mc2b(m, 0.77), 0.77), m, atol=1e-12)


def test_freqt_round_trip_keeps_the_spectrum():
    """freqt(alpha -> 0) of a mel cepstrum describes the same log spectrum on the unwarped axis."""
    rng = np.random.RandomState(1)
    mc = rng.randn(3, 20) * 0.3
    c = hm.sptk_freqt(mc, 1023, 0.58)
    w = np.linspace(0, np.pi, 257)
    ww = hm.warp_axis(0.58, 257)                       # warped frequency of each linear-frequency point
    s_lin = c @ np.cos(np.outer(np.arange(c.shape[1]), w))
    s_mel = mc @ np.cos(np.outer(np.arange(20), ww))
    assert np.max(np.abs(s_lin - s_mel)) < 1e-6


def test_unit_lifter_is_the_cepstral_round_trip():
    x = _log_mel_mags()
    y = mp.post_filter_merlin(x, 48000, pf_coef=1.0)
    ref = hm.cos_matrix_log_spectrum(hm._f32(hm.rceps_compact(x)), x.shape[1])
    assert np.max(np.abs(y - ref)) < 2e-5


def test_energy_is_kept_and_formants_are_sharpened():
    x = _log_mel_mags()
    y = mp.post_filter_merlin(x, 48000)                # pf_coef 1.4
    assert y.shape == x.shape and np.all(np.isfinite(y))
    alpha = mp.define_alpha(48000)
    r0_in = hm.sptk_c2acr_r0(hm.sptk_freqt(hm.rceps_compact(x), 2047, alpha), 4096)
    r0_out = hm.sptk_c2acr_r0(hm.sptk_freqt(hm.rceps_compact(y), 2047, alpha), 4096)
    assert np.max(np.abs(r0_out / r0_in - 1.0)) < 2e-2        # the b0 correction restores the frame energy
    y1 = mp.post_filter_merlin(x, 48000, pf_coef=1.0)
    assert np.mean(y.std(axis=1)) > 1.08 * np.mean(y1.std(axis=1))   # deeper valleys / higher peaks (c[2:] x 1.4)


def test_silent_frames_do_not_produce_nans():
    x = np.full((3, 60), -23.0)
    y = mp.post_filter_merlin(x, 16000)
    assert np.all(np.isfinite(y))
