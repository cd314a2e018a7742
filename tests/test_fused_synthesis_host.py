"""Host side of the fused unwarp -> synthesis launch (mpx_synthesis_compressed_fused): segment planner and the MFMA
fragment packing of the unwarp matrices (no GPU)."""
import numpy as np


def test_segment_planner_properties():
    """hostmath.plan_segments: every frame in exactly one segment, <= 16 frames, rows within 16 of the segment's first."""
    from magphase_amd import hostmath as hm

    rng = np.random.RandomState(0)
    row0 = np.cumsum(rng.randint(0, 4, 500))
    row1 = row0 + rng.randint(0, 2, 500)
    fb = np.array([0, 37, 38, 120, 400])
    fe = np.array([37, 38, 120, 400, 500])
    seg_fb, seg_rb, off = hm.plan_segments(fb, fe, row0, row1)
    assert off[0] == 0 and off[-1] == seg_fb.size and np.all(np.diff(off) >= 1)
    for r in range(fb.size):
        b = list(seg_fb[off[r]:off[r + 1]]) + [fe[r]]
        assert b[0] == fb[r] and np.all(np.diff(b) >= 1) and np.all(np.diff(b) <= 16)
        for k in range(len(b) - 1):
            rb = seg_rb[off[r] + k]
            assert rb == row0[b[k]] and row1[b[k + 1] - 1] - rb <= 15


def test_pack_unwarp_frag_layout():
    """out[ct][q][lane][e] = U[4 (4 q + e) + (lane >> 4)][16 ct + (lane & 15)], zero outside the matrix."""
    from magphase_amd import hostmath as hm

    rng = np.random.RandomState(1)
    u = rng.randn(45, 700)
    pk = hm.pack_unwarp_frag(u, 12, 32)
    assert pk.shape == (32, 3, 64, 4) and pk.dtype == np.float32
    for ct, q, lane, e in ((0, 0, 0, 0), (31, 2, 63, 3), (17, 1, 40, 2), (5, 2, 20, 3)):
        k, c = 4 * (4 * q + e) + (lane >> 4), 16 * ct + (lane & 15)
        want = u[k, c] if (k < 45 and c < 700) else 0.0
        assert pk[ct, q, lane, e] == np.float32(want)
    # a product formed from the packed fragments equals a @ U
    a = rng.randn(16, 45)
    acc = np.zeros((16, 16 * 32))
    for ct in range(32):
        for q in range(3):
            for e in range(4):
                for g in range(4):
                    k = 4 * (4 * q + e) + g
                    if k < 45:
                        acc[:, 16 * ct:16 * ct + 16] += a[:, k][:, None] * pk[ct, q, 16 * g:16 * g + 16, e][None, :]
    assert np.allclose(acc[:, :512], (a @ u)[:, :512], atol=1e-5)
