"""
CPU model of the bookkeeping of k_analysis_warp_fused_cr (magphase_f64.hip; mpx_analysis_compressed_fused_cr): which workgroup
and round owns which constant-rate frame, which of the round's nine rows (eight window frames + the halo) its two source rows
are, and how many sweeps a round takes -- the same integer arithmetic as the launcher and the kernel, on row tables from the
real constant-rate planner (hostmath.var_to_const_rate_table, magphase.py:2219-2239).  Every constant-rate frame must be
produced exactly once, from rows the round actually has.
"""
import numpy as np
import pytest

from magphase_amd import hostmath as hm

K_ROWS = 16      # kCrRows: constant-rate frames per sweep


def _tables(rng, n_utts, fs, f0_lo, f0_hi):
    row0, row1, base = [], [], 0
    for _ in range(n_utts):
        n = int(rng.randint(3, 400))
        shift = np.round(fs / rng.uniform(f0_lo, f0_hi, n)).astype(np.int64)
        pm = np.cumsum(shift)
        if rng.rand() < 0.3:
            pm = pm - pm[0]                      # first epoch at 0: no duplicated first row
        lo, hi, _t = hm.var_to_const_rate_table(pm, 5.0, fs)
        row0.append(lo + base), row1.append(hi + base)
        base += n
    return np.concatenate(row0).astype(np.int64), np.concatenate(row1).astype(np.int64), base


def _launch(n_frames, slots):
    per = (n_frames + slots - 1) // slots
    fw = 8 * ((per + 1 + 7) // 8) - 1                       # frames a workgroup owns: 8 R - 1
    return fw, (n_frames + fw - 1) // fw


@pytest.mark.parametrize("seed,slots,f0", [(0, 256, (90, 260)), (1, 256, (50, 70)), (2, 7, (300, 500)), (3, 1, (90, 260)),
                                           (4, 256, (60, 400)), (5, 3, (48, 55))])
def test_every_constant_rate_frame_is_owned_once_and_finds_its_rows(seed, slots, f0):
    rng = np.random.RandomState(seed)
    row0, row1, n_frames = _tables(rng, int(rng.randint(1, 40)), 48000, *f0)
    n_const = row1.size
    assert np.all(np.diff(row1) >= 0) and np.all((row1 - row0 == 0) | (row1 - row0 == 1))
    cstart = np.searchsorted(row1, np.arange(n_frames + 1), side="left")        # k_cr_index
    fw, grid = _launch(n_frames, slots)
    assert grid <= slots and grid * fw >= n_frames
    owner = np.full(n_const, -1)
    max_sweeps, transformed = 0, 0
    for wg in range(grid):
        own_lo, own_end = wg * fw, min((wg + 1) * fw, n_frames)
        prev_c1 = None
        r = 0
        while True:
            a = own_lo - 1 + 8 * r
            hi_lo = a + 1 if r == 0 else a
            if hi_lo >= own_end:
                break
            hi_hi = min(a + 8, own_end)
            c0, c1 = int(cstart[hi_lo]), int(cstart[hi_hi])
            if prev_c1 is not None:
                assert c0 == prev_c1                 # the kernel takes the next round's c0 from this round's c1
            prev_c1 = c1
            have = [f for f in range(a, a + 8) if 0 <= f < own_end]          # the window's frames this round transforms
            transformed += len(have)
            halo = (a - 1) if r >= 1 else None                               # left by the previous round's wave 7
            if r >= 1:
                assert 0 <= halo < own_end and halo == (own_lo - 1 + 8 * (r - 1)) + 7
            for c in range(c0, c1):
                assert owner[c] == -1
                owner[c] = wg
                lo_rel, hi_rel = int(row0[c] - a), int(row1[c] - a)
                assert 0 <= hi_rel <= 7 and a + hi_rel in have
                assert -1 <= lo_rel <= 7
                if lo_rel == -1:
                    assert halo is not None and row0[c] == halo              # tile row 8
                else:
                    assert a + lo_rel in have
            max_sweeps = max(max_sweeps, -(-(c1 - c0) // K_ROWS))
            r += 1
    assert np.all(owner >= 0)                                                # every constant-rate frame produced, once
    assert np.all(np.diff(owner) >= 0)                                       # ... in workgroup order (contiguous output ranges)
    assert n_frames <= transformed <= n_frames + grid                        # one frame per workgroup boundary is transformed twice
    if f0[1] <= 70:
        assert max_sweeps >= 2                                               # 50-70 Hz: more than 16 frames per window of eight


def test_launch_geometry_small_and_large():
    for n_frames in (1, 2, 7, 8, 9, 255, 256, 257, 2047, 56985, 10 ** 6):
        fw, grid = _launch(n_frames, 256)
        assert fw % 8 == 7 and 1 <= grid <= 256 and grid * fw >= n_frames and (grid - 1) * fw < n_frames
