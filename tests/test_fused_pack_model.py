"""
CPU model of mpx_analysis_compressed_fused's GEMM data flow (csrc/magphase_f64.hip: k_analysis_warp_fused): the published
tile's column -> bin map (hostmath.fused_chunk_bins), the MFMA fragment order of the packed warp matrices
(hostmath.pack_warp_fused) and the K-split over the eight waves reproduce out = X . W^T, bin M/2 through `whalf`.
v_mfma_f32_16x16x4_f32: lane l supplies A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16]; D[i][j] += sum_k A B.
"""
import numpy as np
import pytest

from magphase_amd import hostmath as hm


@pytest.mark.parametrize("n_waves", [4, 8])
@pytest.mark.parametrize("fft_len,mag_dim,phase_dim", [(4096, 60, 10), (4096, 60, 45), (2048, 24, 16), (2048, 64, 33)])
def test_packed_fragments_reproduce_the_matrix_product(fft_len, mag_dim, phase_dim, n_waves):
    rng = np.random.RandomState(fft_len + mag_dim + phase_dim)
    H, M, P = fft_len // 2 + 1, fft_len // 2, fft_len // 128
    w_mag, w_ph = rng.standard_normal((mag_dim, H)), rng.standard_normal((phase_dim, H))
    wpack, whalf = hm.pack_warp_fused(w_mag, w_ph, fft_len, n_waves=n_waves)
    ntm, ntp = 4, (phase_dim + 15) // 16
    T = ntm + ntp                                                              # MFMA tiles: real and imaginary share the phase tiles
    cols, kh = 128 // n_waves, 128 // n_waves // 16
    wpack = wpack.reshape(P // 2, n_waves, kh, T, 64, 4).astype(np.float64)
    whalf = whalf.reshape(T, 16).astype(np.float64)
    bins = hm.fused_chunk_bins(fft_len)
    assert sorted(bins.reshape(-1).tolist() + [M // 2]) == list(range(H))      # every bin once, M/2 left out
    nf = n_waves                                                               # frames per round
    x = rng.standard_normal((3, nf, H))                                        # [stream][frame row][bin] operands
    lane = np.arange(64)
    li, g = lane & 15, lane >> 4
    acc = np.zeros((n_waves, T, 16, 16))                                       # per wave: D[tile][i][j]
    for q in range(P // 2):
        tile = x[:, :, bins[q]]                                                # the published tile [stream][row][128 columns]
        for w in range(n_waves):
            for h in range(kh):
                for t in range(T):
                    for e in range(4):
                        col = cols * w + 16 * h + 4 * g + e
                        if t < ntm:
                            a = tile[0, li & (nf - 1), col]                    # magnitudes: row li mod frames
                        else:                                                  # rows 0..7 real operands, 8..15 imaginary
                            a = np.where(li < 8, tile[1, li & (nf - 1), col], tile[2, li & (nf - 1), col])
                        b = wpack[q, w, h, t, lane, e]
                        for gg in range(4):                                    # D[i][j] += A[i][k = gg] B[k = gg][j]
                            acc[w, t] += np.outer(a[16 * gg:16 * gg + 16], b[16 * gg:16 * gg + 16])
    d = acc.sum(axis=0)                                                        # the round-end reduction over the waves
    for t in range(T):
        wsrc, jt = (w_mag, t) if t < ntm else (w_ph, t - ntm)
        rows = wsrc[16 * jt:16 * jt + 16]
        if rows.shape[0] == 0:                                                 # a tile past the matrix: all padding
            assert np.all(d[t] == 0.0)
            continue
        for sa, r0 in ((0, 0),) if t < ntm else ((1, 0), (2, 8)):              # output rows: frames (real), 8 + frames (imaginary)
            got = d[t][r0:r0 + nf, :rows.shape[0]] + np.outer(x[sa][:, M // 2], whalf[t][:rows.shape[0]])
            want = x[sa] @ rows.T
            assert np.max(np.abs(got - want)) < 2e-5 * max(1.0, np.max(np.abs(want)))   # float32 rounding of the packed weights
        assert np.all(d[t][:, rows.shape[0]:] == 0.0)                          # padding columns: zero weights


@pytest.mark.parametrize("fft_len,mag_dim,phase_dim", [(4096, 60, 10), (2048, 24, 33), (2048, 64, 48)])
def test_layout_1_magnitudes_on_4x4_blocks(fft_len, mag_dim, phase_dim):
    """pack_warp_fused(layout=1) with v_mfma_f32_4x4x1_16b_f32's register layout (measured: tools/archive/mfma4x4_layout_probe.hip
    -- block b = lane >> 2; D[lane 4 b + j][register r] += A(lane 4 b + r) * B(lane 4 b + j)): lane 4 b + x supplies frame
    4 (lane >> 5) + x and coefficient lane & 31 (+ 32 for the second half); the phase fragments are layout 0's."""
    rng = np.random.RandomState(7 + fft_len + mag_dim)
    H, M, P = fft_len // 2 + 1, fft_len // 2, fft_len // 128
    w_mag, w_ph = rng.standard_normal((mag_dim, H)), rng.standard_normal((phase_dim, H))
    wp1, wh1 = hm.pack_warp_fused(w_mag, w_ph, fft_len, n_waves=8, layout=1)
    wp0, wh0 = hm.pack_warp_fused(w_mag, w_ph, fft_len, n_waves=8, layout=0)
    ntp = (phase_dim + 15) // 16
    wp1 = wp1.reshape(P // 2, 8, 8 + ntp, 64, 4)
    assert np.array_equal(wh0, wh1)
    assert np.array_equal(wp1[:, :, 8:], wp0.reshape(P // 2, 8, 1, 4 + ntp, 64, 4)[:, :, 0, 4:])      # phase fragments unchanged
    wp1 = wp1.astype(np.float64)
    bins = hm.fused_chunk_bins(fft_len)
    x = rng.standard_normal((8, H))
    lane = np.arange(64)
    blk = lane >> 2
    acc = np.zeros((8, 2, 64, 4))                                              # [wave][half][lane][register]
    for q in range(P // 2):
        tile = x[:, bins[q]]                                                   # published magnitudes [frame row][128 columns]
        for w in range(8):
            for kg in range(4):
                for e in range(4):
                    a = tile[4 * (lane >> 5) + (lane & 3), 16 * w + 4 * kg + e]
                    for hf in range(2):
                        b = wp1[q, w, 2 * kg + hf, :, e]
                        for r in range(4):
                            acc[w, hf, :, r] += a[4 * blk + r] * b
    d = acc.sum(axis=0)                                                        # the round-end reduction over the waves
    got = np.empty((8, 64))
    for fr in range(8):
        for n in range(64):                                                    # the output loop's addressing
            got[fr, n] = d[n >> 5, (n & 31) + 32 * (fr >> 2), fr & 3]
    got += np.outer(x[:, M // 2], wh1.reshape(-1, 16)[:4].reshape(-1).astype(np.float64))
    want = x @ w_mag.T
    assert np.max(np.abs(got[:, :mag_dim] - want)) < 2e-5 * max(1.0, np.max(np.abs(want)))
    assert np.all(got[:, mag_dim:] == 0.0)
