"""
Built-in epoch / voicing front end (magphase_amd/epochs.py).  PARITY UNPINNED (REAPER is an external
binary that is not available): quality is measured on synthetic utterances whose epochs are known exactly.
"""
import numpy as np
import pytest
import torch

from magphase_amd import epochs
from magphase_amd import synthetic as syn

CPU = torch.device("cpu")


def _score(pm, voi, e_pm, e_voi):
    tv, ev = pm[voi > 0], e_pm[e_voi > 0]
    err = ev[np.abs(ev[None, :] - tv[:, None]).argmin(1)] - tv
    hit = float((np.abs(err) < 0.0005).mean())                                   # within 0.5 ms
    fa = float((np.abs(tv[None, :] - ev[:, None]).min(1) > 0.001).mean())        # estimated epoch with no true one within 1 ms
    return hit, fa, float(np.median(err))


@pytest.mark.parametrize("u,fs", [(0, 48000), (2, 48000), (3, 48000), (5, 16000), (6, 16000)])
def test_epochs_of_synthetic_utterances(u, fs):
    pcm, pm, voi = syn.make_utterance(u, dur_s=2.5, fs=fs)
    e_pm, e_voi = epochs.track_epochs(pcm, fs, device=CPU)
    assert np.all(np.diff(e_pm) > 0) and set(np.unique(e_voi)) <= {0.0, 1.0}
    assert e_pm[0] > 0 and e_pm[-1] * fs < pcm.size - 1
    hit, fa, med = _score(pm, voi, e_pm, e_voi)
    assert hit > 0.85 and fa < 0.08 and abs(med) < 0.0002, (hit, fa, med)
    # unvoiced marks every 5 ms, about as many as the generator placed
    n_unv, n_unv_true = int((e_voi == 0).sum()), int((voi == 0).sum())
    assert abs(n_unv - n_unv_true) < 0.25 * n_unv_true + 10


def test_polarity_does_not_matter():
    pcm, _pm, _voi = syn.make_utterance(1, dur_s=2.0)
    a = epochs.track_epochs(pcm.astype(np.float64) / 32768.0, 48000, device=CPU)
    b = epochs.track_epochs(-pcm.astype(np.float64) / 32768.0, 48000, device=CPU)
    assert a[0].size == b[0].size and np.allclose(a[0], b[0], atol=1.0 / 48000) and np.array_equal(a[1], b[1])


def test_noise_and_silence_are_unvoiced():
    rng = np.random.RandomState(0)
    for sig in (0.1 * rng.randn(32000), np.zeros(32000)):
        pm, voi = epochs.track_epochs(sig, 16000, device=CPU)
        assert voi.sum() <= 0.02 * voi.size
        assert np.allclose(np.diff(pm)[voi[1:] + voi[:-1] == 0], 0.005, atol=1e-6)
