"""CPU tests of the host-side fp64/int plan builders (magphase_amd/hostmath.py) against the oracle."""
import numpy as np
import pytest

from magphase_amd import hostmath as hm
from oracle import magphase_oracle as orc


def _emulate_ola(frames, pm_rel, start, out_len, pm0=0):
    N = frames.shape[1]
    buf = np.zeros(int(pm_rel[-1]) + N + int(pm0))  # the reference's buffer is pm[-1] + N long (absolute pm)
    for i, r in enumerate(pm_rel):
        buf[r:r + N] += frames[i]
    return buf[start:start + out_len]


@pytest.mark.parametrize("first_shift", [120.0, 2047.0, 2048.0, 2049.0, 3300.0, 9000.0])
def test_ola_plan_matches_reference_slicing(first_shift):
    rng = np.random.RandomState(int(first_shift))
    N = 4096
    nfr = 9
    shifts = np.r_[first_shift, rng.uniform(100, 900, nfr - 1)]
    v_pm = np.cumsum(shifts).astype(int)
    frames = rng.randn(nfr, N)
    ref = orc.ola(frames, v_pm)
    rel, start, out_len = hm.ola_plan(v_pm, N)
    got = _emulate_ola(frames, rel, start, out_len, v_pm[0])
    assert len(got) == len(ref)
    assert np.array_equal(got, ref)


def test_ola_plan_single_frame():
    N = 2048
    frames = np.random.RandomState(0).randn(1, N)
    v_pm = np.array([300])
    ref = orc.ola(frames, v_pm)
    rel, start, out_len = hm.ola_plan(v_pm, N)
    assert np.array_equal(_emulate_ola(frames, rel, start, out_len, v_pm[0]), ref)


@pytest.mark.parametrize("N,T", [(4096, 4096), (4096, 2048), (4096, 8192), (2048, 1024), (2048, 2048 + 64)])
def test_ola_chunks_cover_every_frame_once_and_strips_cover_the_buffer(N, T):
    rng = np.random.RandomState(N + T)
    rels = []
    for u in range(4):
        sh = rng.randint(60, 1000, size=rng.randint(3, 200))
        sh[rng.randint(0, len(sh))] = 7000  # a gap: empty territories
        rel = np.cumsum(sh) - sh[0]
        rels.append(rel.astype(np.int64))
    rows, terr_off, owner = hm.ola_chunks(rels, N, T)
    nfr_total = sum(len(r) for r in rels)
    seen = np.zeros(nfr_total, dtype=int)
    for fb, fe, x0, _ in rows:
        seen[fb:fe] += 1
        assert (x0 + N // 2) % T == 0
    assert np.all(seen == 1)
    assert np.all(np.diff(rows[:, 1] - rows[:, 0]) <= 0)  # longest first
    # emulate: strips + 3-neighbour fixup == plain OLA
    frames = rng.randn(nfr_total, N)
    strips = np.zeros((len(rows), T + N))
    allrel = np.concatenate(rels)
    for ci, (fb, fe, x0, _) in enumerate(rows):
        for f in range(fb, fe):
            x = allrel[f] - x0
            assert 0 <= x < T + 0 * N and x + N <= T + N
            strips[ci, x:x + N] += frames[f]
    f0 = 0
    for u, rel in enumerate(rels):
        n = len(rel)
        ref = np.zeros(rel[-1] + N)
        for i in range(n):
            ref[rel[i]:rel[i] + N] += frames[f0 + i]
        got = np.zeros_like(ref)
        nc = terr_off[u + 1] - terr_off[u]
        for b in range(len(ref)):
            c = b // T
            for cc in (c - 1, c, c + 1):
                if 0 <= cc < nc and owner[terr_off[u] + cc] >= 0:
                    idx = b - (cc * T - N // 2)
                    if 0 <= idx < T + N:
                        got[b] += strips[owner[terr_off[u] + cc], idx]
        assert np.max(np.abs(got - ref)) < 1e-12
        f0 += n


class _FakeEngine:
    """Stands in for engine.Engine on a box without a GPU: descriptor tensors stay numpy arrays."""

    def to_device(self, arr, dtype):
        return np.ascontiguousarray(arr, dtype=dtype)

    def to_device_packed(self, items):
        return {name: np.ascontiguousarray(arr, dtype=dt) for name, arr, dt in items}


def test_plans_build_without_gpu_and_match_oracle_indices():
    from magphase_amd import synthetic as syn
    from magphase_amd.engine import LosslessAnalysisPlan, LosslessSynthesisPlan
    utts = []
    for u in range(3):
        pcm, pm, voi = syn.make_utterance(20 + u, dur_s=0.6, fs=48000)
        utts.append((pcm, 48000, pm, voi))
    ap = LosslessAnalysisPlan(_FakeEngine(), utts)
    sp = LosslessSynthesisPlan(_FakeEngine(), ap.v_f0, ap.fs, ap.fft_len)
    off = 0
    for u, (pcm, fs, pm, voi) in enumerate(utts):
        x = syn.pcm_to_float(pcm)
        o = orc.analysis_lossless_from_epochs(x, fs, pm, voi)
        assert np.array_equal(ap.v_shift[u], o[5])
        assert np.array_equal(ap.v_f0[u], o[3])
        a, b = ap.frame_off[u], ap.frame_off[u + 1]
        assert np.array_equal(ap.left[a:b], o[5])
        assert np.array_equal(ap.pos[a:b] - off, orc.round_to_int(orc.clean_epochs(pm, voi, len(x), fs)[0] * fs))
        ref = orc.synthesis_from_lossless(o[0], o[1], o[2], o[3], fs)
        assert sp.out_len[u] == len(ref)
        off += len(x)
    assert sp.total_frames == ap.total_frames
    assert sp.strip_floats == sp.n_chunks * (sp.territory + sp.fft_len)
    assert sp.chunks.shape == (sp.n_chunks, 4)
