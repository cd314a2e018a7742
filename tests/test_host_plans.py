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


def _emulate_runs(runs, rels, frames, N, total_out):
    """numpy model of k_synth_ola_pair's flush rules + k_ola_fixup on the planner's runs (include/magphase_hip.h)."""
    allrel = np.concatenate(rels)
    pcm = np.full(total_out, np.nan)
    writes = np.zeros(total_out, dtype=int)
    strips = np.zeros((len(runs), N + 64))
    for ri, r in enumerate(runs):
        fb, fe, x0 = int(r["frame_begin"]), int(r["frame_end"]), int(r["x0"])
        span = max(int(r["flush_end"]), int(allrel[fe - 1]) - x0 + N)
        acc = np.zeros(span + 64)
        for f in range(fb, fe):
            x = int(allrel[f]) - x0
            assert x >= 0
            acc[x:x + N] += frames[f]
        assert int(r["flush_end"]) >= int(allrel[fe - 1]) - x0 + N      # the ring ends up cleared
        he, lo, hi = int(r["head_end"]), int(r["out_lo"]), int(r["out_hi"])
        assert 0 <= he <= N + 64 and int(r["out_base"]) % 64 == 0
        strips[ri, :he] = acc[:he]
        if hi > lo:
            assert lo >= he or he == 0 or lo >= 0
            idx = int(r["out_base"]) + np.arange(lo, hi)
            pcm[idx] = acc[lo:hi]
            writes[idx] += 1
    assert np.all(writes == 1)          # every kept sample written exactly once by the main kernel
    for ri, r in enumerate(runs):       # fix-up: predecessor's sum (in pcm) + head strip
        lo, hi = int(r["fix_lo"]), int(r["fix_hi"])
        if hi > lo:
            assert hi <= int(r["head_end"])
            pcm[int(r["out_base"]) + np.arange(lo, hi)] += strips[ri, lo:hi]
    return pcm


@pytest.mark.parametrize("N,fpr,n_slots", [(4096, None, 1024), (4096, 7, 8), (4096, 1, 4), (2048, None, 16),
                                           (2048, 3, 2), (1024, None, 4096), (4096, 1000, 1)])
def test_ola_runs_write_every_sample_once_and_sum_to_plain_ola(N, fpr, n_slots):
    rng = np.random.RandomState(N + (fpr or 0) + n_slots)
    rels, starts, lens, v_pms = [], [], [], []
    for u in range(5):
        sh = rng.randint(60, 1000, size=rng.randint(2, 200))
        if u != 1:
            sh[rng.randint(0, len(sh))] = 7000                   # consecutive frames further apart than N: a gap of zeros
        sh[0] = [150, N // 2 + 700, 90, 9000, N // 2][u]           # incl. first epoch beyond N/2 (negative slice start)
        v_pm = np.cumsum(sh)
        rel, start, out_len = hm.ola_plan(v_pm, N)
        rels.append(rel), starts.append(start), lens.append(out_len), v_pms.append(v_pm)
    out_off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    runs, slot_off, slot_runs = hm.ola_runs(rels, starts, lens, out_off, N, n_slots, frames_per_run=fpr)
    assert sorted(slot_runs.tolist()) == list(range(runs.size)) and slot_off[0] == 0 and slot_off[-1] == runs.size
    nfr_total = sum(len(r) for r in rels)
    seen = np.zeros(nfr_total, dtype=int)
    for r in runs:
        seen[int(r["frame_begin"]):int(r["frame_end"])] += 1
    assert np.all(seen == 1) and np.all(np.diff(runs["frame_begin"]) > 0)
    frames = rng.randn(nfr_total, N)
    got = _emulate_runs(runs, rels, frames, N, int(out_off[-1]))
    f0 = 0
    for u, rel in enumerate(rels):
        ref = _emulate_ola(frames[f0:f0 + len(rel)], rel, starts[u], lens[u], v_pms[u][0])
        assert len(ref) == lens[u]
        assert np.max(np.abs(got[out_off[u]:out_off[u + 1]] - ref)) < 1e-12 if lens[u] else True
        f0 += len(rel)
    # only adjacent runs of an utterance overlap
    f0 = 0
    allrel = np.concatenate(rels)
    for u, rel in enumerate(rels):
        mine = [r for r in runs if f0 <= int(r["frame_begin"]) < f0 + len(rel)]
        for a, c in zip(mine[:-2], mine[2:]):
            assert allrel[int(a["frame_end"]) - 1] + N <= allrel[int(c["frame_begin"])]
        f0 += len(rel)


def test_ola_runs_balance_for_the_bench_shape():
    """64 utterances x ~890 frames over 1024 slots: every slot gets the same number of frames +- 1."""
    rng = np.random.RandomState(5)
    rels, starts, lens = [], [], []
    for u in range(64):
        v_pm = np.cumsum(rng.randint(200, 350, size=rng.randint(860, 920)))
        rel, start, out_len = hm.ola_plan(v_pm, 4096)
        rels.append(rel), starts.append(start), lens.append(out_len)
    out_off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    runs, slot_off, slot_runs = hm.ola_runs(rels, starts, lens, out_off, 4096, 1024)
    n = (runs["frame_end"] - runs["frame_begin"]).astype(np.int64)
    loads = np.add.reduceat(n[slot_runs], slot_off[:-1])
    assert slot_off.size == 1025 and loads.max() - loads.min() <= 1
    assert runs.size <= 1024 + 63       # at most one extra run per utterance boundary


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
    assert sp.strip_floats == sp.n_runs * (sp.fft_len + 64)
    assert sp.runs.dtype == np.uint8 and sp.runs.size == sp.n_runs * 56


def test_native_const_to_variable_scan_is_bit_identical_to_the_scipy_form(golden_dir):
    """mpx_host_const_to_var_scan (host C++ in the library) vs one scipy interp1d call per step, and vs the reference's
    own output (golden G7)."""
    import os
    import time
    from magphase_amd import engine
    rng = np.random.RandomState(3)
    for n, fs in ((997, 48000), (400, 16000), (2, 48000), (57, 48000)):
        f0 = np.where(rng.rand(n) < 0.3, 0.0, rng.uniform(60, 400, n))
        sh = hm.f0_to_shift(f0, fs)
        a = engine._const_to_variable_scan(sh, 5.0, fs)
        b = engine._const_to_variable_scan_scipy(sh, 5.0, fs)
        assert a[0].shape == b[0].shape and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    g = np.load(os.path.join(golden_dir, "g7_const_rate.npz"))
    shifts, locs = engine._const_to_variable_scan(hm.f0_to_shift(g["v_f0_c"], int(g["fs"])), 5.0, int(g["fs"]))
    assert np.array_equal(shifts, g["v_shift_vr"]) and np.array_equal(locs, g["v_locs"])
    sh = hm.f0_to_shift(rng.uniform(80, 300, 1000), 48000)
    t0 = time.perf_counter()
    engine._const_to_variable_scan(sh, 5.0, 48000)
    t1 = time.perf_counter()
    engine._const_to_variable_scan_scipy(sh, 5.0, 48000)
    t2 = time.perf_counter()
    assert (t1 - t0) * 20 < (t2 - t1)


# ----------------------------------------------------------------------------------------------------------------------
# native batch planners (csrc/magphase_plan.cpp through magphase_amd/hostplan.py) == the numpy forms, bit for bit
# ----------------------------------------------------------------------------------------------------------------------
def test_native_analysis_planner_equals_numpy_form():
    from magphase_amd import hostplan as hp, synthetic as syn
    rng = np.random.default_rng(0)
    utts = []
    for u in range(8):
        fs = (48000, 16000, 8000)[u % 3]
        pcm, pm, voi = syn.make_utterance(10 + u, dur_s=0.4 + 0.15 * u, fs=fs)
        utts.append((len(pcm), fs, pm, voi))
    n0, fs0, pm0, voi0 = utts[0]
    utts.append((n0, fs0, np.r_[pm0, pm0[-1], pm0[-1] - 1e-4, n0 / fs0 + 0.01], np.r_[voi0, 1, 0, 1]))   # repeats, past the end
    utts.append((n0, fs0, np.r_[0.0, pm0], np.r_[1.0, voi0]))                                            # first epoch at sample 0
    utts.append((n0, fs0, (np.arange(1, 200) + 0.5) / fs0 * 37, rng.integers(0, 2, 199).astype(float)))   # half-even ties
    sig_off = np.concatenate(([0], np.cumsum([u[0] for u in utts])))[:-1]
    r = hp.plan_analysis([u[2] for u in utts], [u[3] for u in utts], [u[0] for u in utts], [u[1] for u in utts], sig_off)
    for i, (n, fs, pm_sec, voi) in enumerate(utts):
        ps, vv = hm.clean_epochs(pm_sec, voi, check_len_smpls=n, fs=fs)
        pm, lft, rgt = hm.frame_bounds(ps * fs, n)
        f0 = hm.shift_to_f0(lft, vv, fs)
        a, b = int(r["frame_off"][i]), int(r["frame_off"][i + 1])
        assert np.array_equal(r["pm"][a:b], pm) and np.array_equal(r["pos"][a:b], pm + sig_off[i])
        assert np.array_equal(r["left"][a:b], lft) and np.array_equal(r["right"][a:b], rgt)
        assert np.array_equal(r["f0"][a:b], f0, equal_nan=True)
    with pytest.raises(hp.PlanFallback):       # an utterance without epochs: left to the numpy form (which raises)
        hp.plan_analysis([np.zeros(0)], [np.zeros(0)], [100], [48000], [0])


@pytest.mark.parametrize("b_const_rate", [False, True])
@pytest.mark.parametrize("fs,N", [(48000, 4096), (16000, 2048)])
def test_native_synthesis_planner_equals_numpy_form(b_const_rate, fs, N):
    from magphase_amd import engine, hostplan as hp
    rng = np.random.RandomState(7 + int(b_const_rate))
    lf0s = []
    for u in range(7):
        n = int(rng.randint(3, 600))
        f0 = rng.uniform(70, 420, n)
        seg = rng.rand(n) < 0.35
        lf0 = np.where(seg, -1.0e10, np.log(f0))                  # la.f0_to_lf0's floor for unvoiced frames
        lf0s.append(lf0)
    for b_win in (True, False):
        a = hp.plan_synthesis([np.exp(l) for l in lf0s], fs, N, b_const_rate, b_win)
        b = engine.plan_synthesis_numpy(lf0s, fs, N, b_const_rate, b_win)
        for k in b:
            assert a[k].dtype == b[k].dtype or k == "rowt", k
            assert np.array_equal(a[k], b[k]), k
    # what the reference's arithmetic raises on, the native planner hands back to the numpy form
    bad = [np.log(np.full(10, 5.0))]                              # f0 = 5 Hz: shifts of fs / 5 samples > N / 2
    with pytest.raises(hp.PlanFallback):
        hp.plan_synthesis([np.exp(l) for l in bad], fs, N, b_const_rate, True)
    with pytest.raises(ValueError):
        engine.plan_synthesis_numpy(bad, fs, N, False, True)


def test_native_ola_runs_equal_numpy_form():
    from magphase_amd import hostplan as hp
    rng = np.random.default_rng(1)
    for trial in range(120):
        U, N = int(rng.integers(1, 9)), int(rng.choice([1024, 2048, 4096]))
        n_slots = int(rng.choice([1, 3, 16, 64, 1024]))
        rels, starts, lens = [], [], []
        for u in range(U):
            n = int(rng.integers(2, 400))
            sh = rng.integers(20, N // 2 - 1, size=n)
            if rng.random() < 0.15:
                sh[rng.integers(0, n)] = N + rng.integers(1, 500)   # frames further apart than N: a gap of zeros
            rel, st, ol = hm.ola_plan(np.cumsum(sh), N)
            rels.append(rel), starts.append(st), lens.append(ol)
        offs = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        a = hm.ola_runs(rels, starts, lens, offs, N, n_slots)
        sizes = [r.size for r in rels]
        b = hp.ola_runs(np.concatenate(rels), np.concatenate(([0], np.cumsum(sizes))), starts, lens, offs[:U], N, n_slots)
        assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_native_lossless_synthesis_planner_equals_numpy_form():
    from magphase_amd import hostplan as hp
    rng = np.random.RandomState(11)
    f0s, fss = [], []
    for u in range(9):
        n = int(rng.randint(1, 500))
        f0 = np.where(rng.rand(n) < 0.3, 0.0, rng.uniform(55, 450, n))
        if u == 3:
            f0[0] = np.inf                                  # first epoch at sample 0: L = 0 -> f0 = inf -> shift 0
        if u == 4:
            f0[:3] = 20.0                                   # first epoch beyond N / 2: ola's negative slice start
        f0s.append(f0), fss.append((48000, 16000)[u % 2])
    for N in (4096, 2048):
        r = hp.plan_lossless_synthesis(f0s, fss, N)
        for u, (f0, fs) in enumerate(zip(f0s, fss)):
            v_pm = np.cumsum(hm.f0_to_shift(f0, fs)).astype(int)
            rel, start, out_len = hm.ola_plan(v_pm, N)
            a, b = int(r["frame_off"][u]), int(r["frame_off"][u + 1])
            assert np.array_equal(r["v_pm"][a:b], v_pm) and np.array_equal(r["pm_rel"][a:b], rel)
            assert (int(r["out_start"][u]), int(r["out_len"][u])) == (start, out_len)
    with pytest.raises(hp.PlanFallback):
        hp.plan_lossless_synthesis([np.array([np.nan, 100.0])], [48000], 4096)


def test_weighted_slot_shares_native_equals_numpy():
    """hostmath.slot_cuts / ola_runs(weights=...): shares in proportion to the slots' relative speeds
    (mpx_synth_ola_slot_weights: a SIMD serves its waves by age), native planner == numpy planner, every frame once."""
    from magphase_amd import _lib, hostmath as hm, hostplan
    lib = _lib.load()
    n_slots = 24
    w = np.zeros(n_slots, dtype=np.float32)
    assert lib.mpx_synth_ola_slot_weights(w.ctypes.data, n_slots) == 0
    assert np.all(w > 0) and w[0] >= w[-1] and len(set(w.tolist())) >= 2      # older wave groups are faster
    cuts = hm.slot_cuts(10000, n_slots, w)
    assert cuts[0] == 0 and cuts[-1] == 10000 and np.all(np.diff(cuts) > 0)
    shares = np.diff(cuts) / 10000.0
    assert np.max(np.abs(shares - w / w.sum())) < 1e-3
    assert np.array_equal(hm.slot_cuts(7, 24, w), np.round(7 * np.concatenate(([0], np.cumsum(w[:7]))) / w[:7].sum()))
    rng = np.random.RandomState(4)
    rels, starts, lens = [], [], []
    for u in range(5):
        pm = 300 + np.cumsum(rng.randint(150, 500, 400 + 37 * u))
        rel, start, out_len = hm.ola_plan(pm, 4096)
        rels.append(rel), starts.append(start), lens.append(out_len)
    out_off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    r1, so1, sr1 = hm.ola_runs(rels, starts, lens, out_off, 4096, n_slots, weights=w)
    sizes = [r.size for r in rels]
    r2, so2, sr2 = hostplan.ola_runs(np.concatenate(rels), np.concatenate(([0], np.cumsum(sizes))), starts, lens, out_off[:5],
                                     4096, n_slots, weights=w)
    assert r1.tobytes() == r2.tobytes() and np.array_equal(so1, so2) and np.array_equal(sr1, sr2)
    assert r1["frame_begin"][0] == 0 and r1["frame_end"][-1] == sum(sizes)
    assert np.array_equal(r1["frame_begin"][1:], r1["frame_end"][:-1])
    per_slot = np.array([sum(int(r1["frame_end"][k] - r1["frame_begin"][k]) for k in sr1[so1[s]:so1[s + 1]]) for s in range(n_slots)])
    assert per_slot[0] > per_slot[-1] and per_slot.sum() == sum(sizes)


def test_hann_half_table_layout_matches_the_references_window():
    """hostmath.hann_half_table (what mpx_analysis_frames_f64w / mpx_analysis_compressed_fused read): half length h at offset
    h (h + 1) / 2, entries np.hanning(2 h + 1)[0 .. h] -- from which a frame's window is assembled exactly as the reference
    does (libaudio.py:70-84: rising half of the left window, flipped rising half of the right one)."""
    from oracle import magphase_oracle as orc
    cap = 300
    tab = hm.hann_half_table(cap)
    assert tab.size == (cap + 1) * (cap + 2) // 2 and tab[0] == 1.0          # np.hanning(1) == [1.0]
    for left, right in ((0, 5), (7, 0), (1, 1), (123, 300), (300, 299), (17, 64)):
        tl = tab[left * (left + 1) // 2:][:left + 1]
        tr = tab[right * (right + 1) // 2:][:right + 1]
        k = np.arange(left + right + 1)
        w = np.where(k <= left, tl[np.minimum(k, left)], tr[np.clip(left + right - k, 0, right)])   # the kernels' indexing
        assert np.array_equal(w, orc.half_windows(left, right)), (left, right)


def test_too_early_cuts_move_forward_instead_of_merging_runs():
    """hostmath._enforce_span (and its native twin): a run with both neighbours must span >= N output samples between
    its predecessor's last frame and its successor's first.  A cut that violates it is moved to the first frame that
    satisfies it -- the slot keeps a run -- and dropped only when its successor has no such frame."""
    rng = np.random.RandomState(8)
    N = 4096
    rel = np.cumsum(rng.randint(150, 260, 3000))                 # high-pitched: ~20 frames per N samples
    cuts = np.arange(0, 3001, 12)                                 # 12-frame shares: every cut comes too early
    out = hm._enforce_span(rel, N, cuts)
    assert out[0] == 0 and out[-1] == 3000 and np.all(np.diff(out) > 0)
    for k in range(1, len(out) - 2):
        assert rel[out[k + 1]] - rel[out[k] - 1] >= N
    dropped_all = []                                              # the round-3 rule, for comparison
    c = [int(x) for x in cuts]
    k = 1
    while k < len(c) - 2:
        if rel[c[k + 1]] - rel[c[k] - 1] < N:
            del c[k + 1]
        else:
            k += 1
    dropped_all = c
    assert len(out) >= len(dropped_all)                           # never fewer runs than dropping gives
    lens_new, lens_old = np.diff(out), np.diff(dropped_all)
    assert lens_new[1:-1].max() <= lens_old[1:-1].max()           # and no longer ones
    # cuts that already satisfy the condition are left alone
    wide = np.arange(0, 3001, 100)
    assert np.array_equal(hm._enforce_span(rel, N, wide), wide)
    # no frame of the successor far enough: the cut is dropped, as before
    rel2 = np.array([0, 10, 20, 30, 40, 50, 5000, 5010])
    assert np.array_equal(hm._enforce_span(rel2, N, np.array([0, 2, 4, 6, 8])), np.array([0, 2, 6, 8]))


# ----------------------------------------------------------------------------------------------------------------------
# whole-launch planners (mpx_host_plan_analysis_batch / _synthesis_batch through _mpx_pyhost) == the list-based forms
# ----------------------------------------------------------------------------------------------------------------------
def _pyhost():
    from magphase_amd import hostplan as hp
    ph = hp.pyhost()
    if ph is None:
        pytest.skip("_mpx_pyhost not built (no Python.h)")
    return ph


def _analysis_batch_utts():
    from magphase_amd import synthetic as syn
    rng = np.random.default_rng(1)
    utts = []
    for u in range(9):
        fs = (48000, 16000)[u % 2] if u < 6 else 48000
        pcm, pm, voi = syn.make_utterance(30 + u, dur_s=0.3 + 0.1 * u, fs=fs)
        utts.append([pcm, fs, pm, voi])
    pcm0, fs0, pm0, voi0 = utts[0]
    n0 = len(pcm0)
    utts.append([pcm0, fs0, np.r_[pm0, pm0[-1], pm0[-1] - 1e-4, n0 / fs0 + 0.01], np.r_[voi0, 1, 0, 1]])   # repeats, past the end
    utts.append([pcm0, fs0, np.r_[0.0, pm0], np.r_[1.0, voi0]])                                            # first epoch at sample 0
    utts.append([pcm0, fs0, (np.arange(1, 200) + 0.5) / fs0 * 37, rng.integers(0, 2, 199).astype(float)])   # half-even ties
    utts.append([pcm0, fs0, np.r_[pm0[:5], pm0[5] + 0.2, pm0[6:] + 0.2], voi0])                              # a frame longer than fft_len
    return utts


@pytest.mark.parametrize("sig_kind", ["i16", "mixed"])
def test_native_analysis_batch_equals_list_planner(sig_kind):
    from magphase_amd import hostplan as hp, synthetic as syn
    ph = _pyhost()
    utts = _analysis_batch_utts()
    if sig_kind == "mixed":
        for k, u in enumerate(utts):
            if k % 3 == 1:
                u[0] = syn.pcm_to_float(u[0]).astype(np.float32)
            elif k % 3 == 2:
                u[0] = syn.pcm_to_float(u[0]).astype(np.float64) * 0.999
    utts = [tuple(u) for u in utts]
    m = ph.analysis_marshal(utts)
    assert m is not None
    U, total, E, all_i16 = ph.analysis_info(m)
    assert U == len(utts) and total == sum(len(u[0]) for u in utts) and E == sum(len(u[2]) for u in utts)
    assert all_i16 == (sig_kind == "i16")
    stage = np.zeros(total + 8, dtype=np.int16 if all_i16 else np.float32)
    pos, pm, left64 = (np.empty(E, dtype=np.int64) for _ in range(3))
    left32, right32 = np.empty(E, dtype=np.int32), np.empty(E, dtype=np.int32)
    voi32, f0, f0_med = np.empty(E, dtype=np.float32), np.empty(E), np.empty(E)
    frame_off = np.empty(U + 1, dtype=np.int64)
    lf, ll = np.empty(64, dtype=np.int64), np.empty(64, dtype=np.int64)
    for n_thr in (1, 5):
        F, n_long = ph.analysis_run(m, stage.ctypes.data, 0 if all_i16 else 1, pos, left32, right32, voi32, pm, left64, f0,
                                    f0_med, frame_off, 4096, lf, ll, n_thr)
        sig_off = np.concatenate(([0], np.cumsum([len(u[0]) for u in utts])))[:-1]
        r = hp.plan_analysis([u[2] for u in utts], [u[3] for u in utts], [len(u[0]) for u in utts], [u[1] for u in utts], sig_off)
        assert F == r["pos"].size and np.array_equal(frame_off, r["frame_off"])
        assert np.array_equal(pos[:F], r["pos"]) and np.array_equal(pm[:F], r["pm"])
        assert np.array_equal(left64[:F], r["left"]) and np.array_equal(left32[:F], r["left"])
        assert np.array_equal(right32[:F], r["right"]) and np.array_equal(f0[:F], r["f0"], equal_nan=True)
        with np.errstate(invalid="ignore"):
            assert np.array_equal(voi32[:F], (r["f0"] > 0).astype(np.float32))
        med = hm.medfilt3_batch([r["f0"][int(a):int(b)] for a, b in zip(frame_off[:-1], frame_off[1:])])
        assert np.array_equal(f0_med[:F], np.concatenate(med), equal_nan=True)
        tot = r["left"] + r["right"] + 1
        hit = np.flatnonzero(tot > 4096)
        assert n_long == hit.size and n_long >= 1
        assert np.array_equal(lf[:n_long], hit) and np.array_equal(ll[:n_long], tot[hit])
        o = 0
        for u in utts:   # the staged samples: int16 as they are / float32 as the generic path converts them
            x = np.asarray(u[0])
            want = x if all_i16 else (x.astype(np.float32) * np.float32(1.0 / 32768.0) if x.dtype == np.int16 else x.astype(np.float32))
            assert np.array_equal(stage[o:o + len(x)], want)
            o += len(x)
    # not the plain shape: a Python list of samples, float32 epochs, unequal lengths -> None (the generic path takes over)
    assert ph.analysis_marshal([(list(range(10)), 48000, utts[0][2], utts[0][3])]) is None
    assert ph.analysis_marshal([(utts[0][0], 48000, utts[0][2].astype(np.float32), utts[0][3])]) is None
    assert ph.analysis_marshal([(utts[0][0], 48000, utts[0][2], utts[0][3][:-1])]) is None
    # an utterance without epochs: negative code naming it
    m2 = ph.analysis_marshal([utts[0], (utts[0][0], 48000, np.zeros(0), np.zeros(0))])
    F, _ = ph.analysis_run(m2, 0, 1, pos, left32, right32, voi32, pm, left64, f0, f0_med, frame_off, 4096, lf, ll, 2)
    assert F == -3


@pytest.mark.parametrize("b_const_rate", [False, True])
@pytest.mark.parametrize("fs,N,weighted", [(48000, 4096, True), (16000, 2048, False)])
def test_native_synthesis_batch_equals_list_planner(b_const_rate, fs, N, weighted):
    from magphase_amd import hostplan as hp
    ph = _pyhost()
    rng = np.random.RandomState(11 + int(b_const_rate))
    utts, mag_dim, phase_dim = [], 60, 45
    for u in range(7):
        rows = int(rng.randint(40, 260))
        f0 = np.where(rng.rand(rows) < 0.3, 0.0, rng.uniform(70, 300, rows))
        f0[:3] = 0.0 if u % 2 else 150.0
        with np.errstate(divide="ignore"):
            lf0 = np.where(f0 > 0, np.log(np.maximum(f0, 1e-300)), -1e10)
        dt = np.float32 if u % 2 == 0 else np.float64
        utts.append((rng.randn(rows, mag_dim).astype(dt), rng.randn(rows, phase_dim).astype(dt),
                     rng.randn(rows, phase_dim).astype(dt), lf0.astype(np.float32 if u % 3 == 0 else np.float64)))
    m = ph.synthesis_marshal(utts)
    assert m is not None
    U, R, md, pd = ph.synthesis_info(m)
    assert (U, md, pd) == (len(utts), mag_dim, phase_dim) and R == sum(u[0].shape[0] for u in utts)
    lf0_cat = np.empty(R)
    ph.synthesis_lf0(m, lf0_cat)
    assert np.array_equal(lf0_cat, np.concatenate([u[3].astype(np.float64) for u in utts]))
    f0 = np.exp(lf0_cat)
    n_slots = 37
    w = (1.0 + 0.5 * rng.rand(n_slots)).astype(np.float32) if weighted else None
    wcum = np.concatenate(([0.0], np.cumsum(np.asarray(w, dtype=np.float64)))) if weighted else None
    wsum = float(np.asarray(w, dtype=np.float64).sum()) if weighted else 0.0
    stage = np.zeros(R * (mag_dim + 2 * phase_dim), dtype=np.float32)
    cap = 2 * R + 2 * U
    desc = np.zeros(hp.synth_desc_bytes(R, U, n_slots, True), dtype=np.uint8)
    desc_off, counts = np.zeros(18, dtype=np.int64), np.zeros(8, dtype=np.int64)
    v_shift, v_pm, voiced = np.empty(cap, dtype=np.int64), np.empty(cap, dtype=np.int64), np.empty(cap, dtype=np.int32)
    frame_off = np.empty(U + 1, dtype=np.int64)
    ns_len, out_start, out_len = (np.empty(U, dtype=np.int64) for _ in range(3))
    runs_host = np.zeros(U + n_slots + 1, dtype=hm.OLA_RUN_DTYPE)
    for n_thr in (1, 4):
        F = ph.synthesis_run(m, stage.ctypes.data, f0, float(fs), N, int(b_const_rate), 1, n_slots, wcum, wsum, 1, desc, desc_off,
                             v_shift, v_pm, voiced, frame_off, ns_len, out_start, out_len, runs_host, counts, n_thr)
        r = hp.plan_synthesis([np.exp(u[3].astype(np.float64)) for u in utts], fs, N, b_const_rate, True)
        assert F == r["v_pm"].size == counts[0] and np.array_equal(frame_off, r["frame_off"])
        assert np.array_equal(v_shift[:F], r["v_shift"]) and np.array_equal(v_pm[:F], r["v_pm"])
        assert np.array_equal(voiced[:F], r["voiced"]) and np.array_equal(ns_len, r["ns_len"])
        assert np.array_equal(out_start, r["out_start"]) and np.array_equal(out_len, r["out_len"])
        out_off = np.concatenate(([0], np.cumsum(r["out_len"]))).astype(np.int64)
        runs, slot_off, slot_runs = hp.ola_runs(r["pm_rel"], r["frame_off"], r["out_start"], r["out_len"], out_off[:U], N,
                                                n_slots, weights=w)
        nr = int(counts[1])
        assert nr == runs.size and counts[2] == slot_off.size - 1
        assert np.array_equal(runs_host[:nr], runs)
        tab = {}
        sizes = {"utt_frame_off": U + 1, "tile_first": int(counts[6]), "out_start": U, "out_off": U + 1, "runs": 56 * nr,
                 "slot_off": int(counts[2]) + 1, "slot_runs": nr}
        for (name, dt), off in zip(hp.SYNTH_TABLES, desc_off.tolist()):
            assert off % 256 == 0
            n = sizes.get(name, F)
            tab[name] = desc[off:off + n * np.dtype(dt).itemsize].view(dt)
        assert np.array_equal(tab["utt_frame_off"], r["frame_off"]) and np.array_equal(tab["npos"], r["npos"])
        for k in ("nleft", "nright", "wtype", "voiced", "row0", "row1", "win_l", "win_r", "pm_rel"):
            assert np.array_equal(tab[k], r[k]), k
        assert np.array_equal(tab["rowt"], r["rowt"].astype(np.float32))
        assert np.array_equal(tab["out_start"], r["out_start"]) and np.array_equal(tab["out_off"], out_off)
        assert np.array_equal(tab["runs"].view(hm.OLA_RUN_DTYPE), runs)
        assert np.array_equal(tab["slot_off"], slot_off) and np.array_equal(tab["slot_runs"], slot_runs)
        assert np.array_equal(tab["tile_first"], np.searchsorted(r["row0"], 31 * np.arange((R + 30) // 31 + 1), side="left"))
        assert counts[4] == int(r["ns_len"].sum()) and counts[5] == int(out_off[-1]) and counts[7] == R
        n_m, n_p = R * mag_dim, R * phase_dim
        assert np.array_equal(stage[:n_m].reshape(R, mag_dim), np.concatenate([u[0] for u in utts]).astype(np.float32))
        assert np.array_equal(stage[n_m:n_m + n_p].reshape(R, phase_dim), np.concatenate([u[1] for u in utts]).astype(np.float32))
        assert np.array_equal(stage[n_m + n_p:].reshape(R, phase_dim), np.concatenate([u[2] for u in utts]).astype(np.float32))
    # shapes the native path leaves to the generic one
    bad = list(utts)
    bad[2] = (bad[2][0], bad[2][1][:, :44], bad[2][2][:, :44], bad[2][3])            # another phase dimension
    assert ph.synthesis_marshal(bad) is None
    bad[2] = (utts[2][0][:-1], utts[2][1], utts[2][2], utts[2][3])                   # frame counts disagree
    assert ph.synthesis_marshal(bad) is None
    bad[2] = (utts[2][0].T.copy().T, utts[2][1], utts[2][2], utts[2][3])             # not C-contiguous
    assert ph.synthesis_marshal(bad) is None
    # an utterance the numpy form raises on (one row): negative code naming it
    one = list(utts) + [(utts[0][0][:1], utts[0][1][:1], utts[0][2][:1], utts[0][3][:1])]
    m1 = ph.synthesis_marshal(one)
    _, R1, _, _ = ph.synthesis_info(m1)
    f1 = np.empty(R1)
    ph.synthesis_lf0(m1, f1)
    np.exp(f1, out=f1)
    st1 = np.zeros(R1 * (mag_dim + 2 * phase_dim), dtype=np.float32)
    big = lambda n, dt=np.int64: np.empty(n, dtype=dt)   # noqa: E731
    rc = ph.synthesis_run(m1, st1.ctypes.data, f1, float(fs), N, int(b_const_rate), 1, n_slots, None, 0.0, 1,
                          np.zeros(hp.synth_desc_bytes(R1, U + 1, n_slots, True), dtype=np.uint8), desc_off, big(2 * R1 + 2 * U + 2),
                          big(2 * R1 + 2 * U + 2), big(2 * R1 + 2 * U + 2, np.int32), big(U + 2), big(U + 1), big(U + 1), big(U + 1),
                          np.zeros(U + n_slots + 2, dtype=hm.OLA_RUN_DTYPE), counts, 2)
    assert rc == -(U + 2)


def test_native_synthesis_batch_with_fewer_frames_than_slots():
    """Weighted shares and fewer frames than slots: the native planner asks for the first F weights' cumsum / sum (their sum
    is numpy's pairwise one) and then equals hostmath.slot_cuts' cut-down shares."""
    from magphase_amd import hostplan as hp
    ph = _pyhost()
    rng = np.random.RandomState(3)
    utts = []
    for u in range(3):
        rows = 30 + 5 * u
        lf0 = np.log(rng.uniform(80, 250, rows))
        utts.append((rng.randn(rows, 60).astype(np.float32), rng.randn(rows, 45).astype(np.float32),
                     rng.randn(rows, 45).astype(np.float32), lf0))
    m = ph.synthesis_marshal(utts)
    U, R, _, _ = ph.synthesis_info(m)
    f0 = np.empty(R)
    ph.synthesis_lf0(m, f0)
    np.exp(f0, out=f0)
    n_slots = 1536
    w = (1.0 + 0.5 * rng.rand(n_slots)).astype(np.float32)
    w64 = np.asarray(w, dtype=np.float64)
    cap = 2 * R + 2 * U
    desc = np.zeros(hp.synth_desc_bytes(R, U, n_slots, True), dtype=np.uint8)
    desc_off, counts = np.zeros(18, dtype=np.int64), np.zeros(8, dtype=np.int64)
    outs = (np.empty(cap, dtype=np.int64), np.empty(cap, dtype=np.int64), np.empty(cap, dtype=np.int32),
            np.empty(U + 1, dtype=np.int64), np.empty(U, dtype=np.int64), np.empty(U, dtype=np.int64), np.empty(U, dtype=np.int64))
    runs_host = np.zeros(U + n_slots + 1, dtype=hm.OLA_RUN_DTYPE)

    def run(ns, wc, ws):
        return ph.synthesis_run(m, 0, f0, 48000.0, 4096, 0, 1, ns, wc, ws, 1, desc, desc_off, *outs, runs_host, counts, 2)

    assert run(n_slots, np.concatenate(([0.0], np.cumsum(w64))), float(w64.sum())) == -4000000
    F = int(counts[0])
    assert F == R < n_slots
    assert run(F, np.concatenate(([0.0], np.cumsum(w64[:F]))), float(w64[:F].sum())) == F
    r = hp.plan_synthesis([np.exp(u[3]) for u in utts], 48000, 4096, False, True)
    out_off = np.concatenate(([0], np.cumsum(r["out_len"]))).astype(np.int64)
    runs, slot_off, _ = hp.ola_runs(r["pm_rel"], r["frame_off"], r["out_start"], r["out_len"], out_off[:U], 4096, n_slots, weights=w)
    nr = int(counts[1])
    assert nr == runs.size and np.array_equal(runs_host[:nr], runs) and counts[2] == slot_off.size - 1 == F
