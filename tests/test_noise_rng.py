"""
Device noise source (mpx_noise_uniform, noise_mode='device'): Philox4x32-10 restated in numpy -- integer arithmetic,
so the kernel must agree BIT FOR BIT -- plus the known-answer vectors of the Random123 distribution (kat_vectors).
"""
import numpy as np
import pytest

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr, key):
    """ctr: uint32 [n x 4], key: (k0, k1) -> uint32 [n x 4]."""
    c = [ctr[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & mask, p1 & mask, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & mask, p0 & mask]
        k0 = (k0 + np.uint64(W0)) & mask
        k1 = (k1 + np.uint64(W1)) & mask
    return np.stack(c, axis=1).astype(np.uint32)


def uniform_noise(seed, n):
    """What mpx_noise_uniform writes for one utterance: float32 [n] in [-1, 1)."""
    q = np.arange((n + 3) // 4, dtype=np.uint64)
    ctr = np.zeros((q.size, 4), dtype=np.uint32)
    ctr[:, 0] = (q & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (q >> np.uint64(32)).astype(np.uint32)
    w = philox4x32_10(ctr, (int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)).reshape(-1)[:n]
    return ((w >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0)).astype(np.float32)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kat:
        got = philox4x32_10(np.asarray([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(x) for x in got) == exp


def test_uniform_noise_statistics():
    x = uniform_noise(12345, 400001).astype(np.float64)
    assert x.min() >= -1.0 and x.max() < 1.0
    assert abs(x.mean()) < 5e-3 and abs(x.var() - 1.0 / 3.0) < 5e-3
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 5e-3
    assert not np.array_equal(uniform_noise(1, 64), uniform_noise(2, 64))


@pytest.mark.gpu
def test_device_noise_matches_numpy_bit_for_bit():
    import torch
    from magphase_amd import _lib
    from magphase_amd.engine import get_engine
    eng = get_engine()
    lens = [1, 4, 7, 1023, 1024, 1025, 50001]
    seeds = np.asarray([0, 1, 2 ** 63 + 5, 0xFFFFFFFFFFFFFFFF, 42, 7, 0x123456789ABCDEF], dtype=np.uint64)
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    d_seeds = torch.from_numpy(seeds.view(np.int64)).to(eng.device)
    d_off = torch.from_numpy(off).to(eng.device)
    out = torch.full((int(off[-1]) + 8,), 7.0, dtype=torch.float32, device=eng.device)
    with torch.cuda.device(eng.device):
        _lib.check(eng.lib.mpx_noise_uniform(eng.stream_ptr(), len(lens), d_seeds.data_ptr(), d_off.data_ptr(), max(lens),
                                             out.data_ptr()), "mpx_noise_uniform")
    h = out.cpu().numpy()
    assert np.all(h[int(off[-1]):] == 7.0)                       # nothing written past the end
    for u, n in enumerate(lens):
        assert np.array_equal(h[off[u]:off[u + 1]], uniform_noise(int(seeds[u]), n)), u


@pytest.mark.gpu
def test_device_noise_generation_is_independent_of_batching():
    """noise_mode='device': the same utterance gives the same PCM alone, in a batch, and at another batch position."""
    import os
    from magphase_amd import libutils as lu, magphase as mp
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "demos", "data_48k", "params_predicted")
    utts = [tuple(lu.read_binfile(os.path.join(d, t + e), dim=k) for e, k in ((".mag", 60), (".real", 45), (".imag", 45), (".lf0", 1)))
            for t in ("hvd_704", "hvd_705", "hvd_706")]
    seeds = [11, 22, 33]
    a = mp.synthesis_from_compressed_batch(utts, 48000, noise_mode="device", noise_seeds=seeds)
    b = mp.synthesis_from_compressed_batch(utts[::-1], 48000, noise_mode="device", noise_seeds=seeds[::-1])[::-1]
    for u in range(3):
        one = mp.synthesis_from_compressed_batch([utts[u]], 48000, noise_mode="device", noise_seeds=[seeds[u]])[0]
        assert np.all(np.isfinite(one)) and np.max(np.abs(one)) > 1e-3
        # run partition differs with the batch: fp32 re-association at run boundaries only
        assert np.max(np.abs(a[u] - one)) <= 2e-6 * np.max(np.abs(one))
        assert np.max(np.abs(b[u] - one)) <= 2e-6 * np.max(np.abs(one))
    # the statistics the reference normalises by (Q10) behave: same gains to ~1 % as with numpy's generator
    np.random.seed(3)
    ref = mp.synthesis_from_compressed_batch([utts[0]], 48000)[0]
    assert abs(np.std(ref) / np.std(a[0]) - 1.0) < 0.05


@pytest.mark.gpu
def test_numpy_global_generator_continued_on_the_device():
    """mpx_noise_numpy_mt19937: np.random.uniform(-1, 1, n) from numpy's GLOBAL MT19937 state, produced on the device --
    the same float32 samples as the host draw and the same generator state afterwards (any position in the 624-word
    block, lengths around the block and the 454-word step boundaries)."""
    from magphase_amd.engine import get_engine
    e = get_engine()
    for seed, warm, n in ((1, 0, 5), (2, 311, 700), (3, 1, 1000), (5, 623, 312), (7, 0, 312), (9, 100, 311), (11, 17, 227),
                          (13, 0, 1), (4, 5, 300001),
                          # many-workgroup form (segments of 159 744 samples, jump-ahead windows): 2, 3, 20 segments, a
                          # draw ending exactly on a segment / block boundary, one ending a word after it
                          (21, 0, 159744 + 312), (22, 77, 400000), (23, 623, 3111111), (24, 0, 2 * 159744 + 312),
                          (25, 0, 2 * 159744 + 313), (26, 1, 159744 * 9),
                          # 188 segments (the noise of 128 utterances of 5 s), and more than 256: segments of twice the length
                          (27, 5, 30_000_000), (28, 3, 45_000_000)):
        np.random.seed(seed)
        np.random.uniform(size=warm)
        st = np.random.get_state()
        want = np.random.uniform(-1, 1, n).astype(np.float32)
        after = np.random.get_state()
        nxt = np.random.uniform(-1, 1, 7)
        np.random.set_state(st)
        got = e.numpy_global_uniform(n).cpu().numpy()
        now = np.random.get_state()
        assert np.array_equal(got, want), (seed, warm, n)
        assert np.array_equal(now[1], after[1]) and now[2] == after[2] and now[3:] == after[3:]
        assert np.array_equal(np.random.uniform(-1, 1, 7), nxt)          # and the stream goes on identically


@pytest.mark.gpu
def test_deferred_generator_state_chains_on_the_device():
    """numpy_global_uniform(defer=True): consecutive draws continue from the state the previous one left ON THE DEVICE (no
    download per call); mt_sync() puts numpy's global generator where the host draws would have left it; mt_snapshot /
    mt_restore rewind a deferred state; the host-tracked word cursor equals the device's."""
    from magphase_amd.engine import get_engine
    e = get_engine()
    sizes = (300001, 5, 159744 + 312, 700, 1, 2 * 159744 + 313, 312)
    np.random.seed(77)
    np.random.uniform(size=311)
    st = np.random.get_state()
    want = [np.random.uniform(-1, 1, n).astype(np.float32) for n in sizes]
    after = np.random.get_state()
    np.random.set_state(st)
    got = []
    for k, n in enumerate(sizes):
        if k == 3:                                   # a failed batch: rewind and draw again
            snap = e.mt_snapshot()
            e.numpy_global_uniform(12345, defer=True)
            e.mt_restore(snap)
        got.append(e.numpy_global_uniform(n, defer=True).cpu().numpy())
    assert np.array_equal(np.random.get_state()[1], st[1])          # numpy's own state is stale until the sync
    e.mt_sync()
    now = np.random.get_state()
    for a, b, n in zip(got, want, sizes):
        assert np.array_equal(a, b), n
    assert np.array_equal(now[1], after[1]) and now[2] == after[2] and now[3:] == after[3:]
    e.mt_sync()                                      # idempotent
    assert np.random.get_state()[2] == after[2]
    for p0, w in ((0, 0), (624, 1), (620, 4), (620, 5), (1, 1247), (1, 1248), (300, 624 * 7 + 324), (300, 624 * 7 + 325)):
        pos = p0                                     # randomkit's cursor, word by word
        for _ in range(w):
            if pos == 624:
                pos = 0
            pos += 1
        assert e._mt_next_pos(p0, w) == pos, (p0, w)


def _mt_raw_stream(key, n_words):
    """Raw (untempered) words X[0 .. n_words) of MT19937 continued from a 624-word key (X[0..623] = key)."""
    x = np.zeros(n_words + 624, dtype=np.uint32)
    x[:624] = key
    up, lo, mag = np.uint32(0x80000000), np.uint32(0x7FFFFFFF), np.uint32(0x9908B0DF)
    n = 0
    while n + 624 < x.size:       # 227 new words per step depend on old ones only
        m = min(227, x.size - 624 - n)
        y = (x[n:n + m] & up) | (x[n + 1:n + m + 1] & lo)
        x[n + 624:n + 624 + m] = x[n + 397:n + 397 + m] ^ (y >> np.uint32(1)) ^ np.where(y & np.uint32(1), mag, np.uint32(0))
        n += m
    return x[:n_words]


def test_mt19937_jump_polynomials_against_the_recurrence():
    """mpx_host_mt19937_jump_poly (host, no device): X[n + J] == xor of X[n + i] over the set bits i of x^J mod phi, for
    the words the recurrence produces from one of numpy's own states; the ladder's level l is the jump J 2^l."""
    import ctypes
    from magphase_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(12345)
    key = rs.get_state()[1].astype(np.uint32)
    # the key really is numpy's stream: tempering X[0..] gives random_raw's words (pos = 624 after seeding)
    J, levels = 624 * 40, 3
    polys = np.zeros((levels, 624), dtype=np.uint32)
    assert lib.mpx_host_mt19937_jump_poly(J, levels, polys.ctypes.data_as(ctypes.c_void_p)) == 0
    x = _mt_raw_stream(key, 624 + 19937 + 624 + J * 4 + 700)
    for l in range(levels):
        bits = np.unpackbits(polys[l].view(np.uint8), bitorder="little")
        assert bits.size == 19968 and not bits[19937:].any()
        idx = np.flatnonzero(bits)
        assert idx.size > 100
        for n in (624, 625, 624 + 623, 2000):
            acc = np.bitwise_xor.reduce(x[n + idx])
            assert acc == x[n + J * (1 << l)], (l, n)
    # the same call again (cached ladder) and a longer ladder agree
    more = np.zeros((levels + 2, 624), dtype=np.uint32)
    assert lib.mpx_host_mt19937_jump_poly(J, levels + 2, more.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(more[:levels], polys)
    assert lib.mpx_host_mt19937_jump_poly(0, 1, more.ctypes.data_as(ctypes.c_void_p)) != 0
