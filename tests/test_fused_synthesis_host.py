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


def test_bf16_split_is_exact_to_2_pow_minus_24_and_packs_in_fragment_order():
    """hostmath.bf16_split3 / pack_unwarp_frag_bf16: the three bfloat16 parts sum back to the float32 value to 2^-24 of it,
    and out[ct][kb][s][lane][j] = split_s(U)[32 kb + 8 (lane >> 4) + j][16 ct + (lane & 15)]."""
    from magphase_amd import hostmath as hm

    rng = np.random.RandomState(2)
    v = (rng.randn(4000) * np.exp(rng.uniform(-20, 5, 4000))).astype(np.float32)
    a, b, c = hm.bf16_split3(v)
    for part in (a, b, c):   # each part IS a bfloat16 value: its low 16 bits are zero
        assert np.all((part.view(np.uint32) & 0xFFFF) == 0)
    rec = a.astype(np.float64) + b.astype(np.float64) + c.astype(np.float64)
    assert np.max(np.abs(rec - v.astype(np.float64)) / np.abs(v.astype(np.float64))) <= 2.0 ** -23
    u = rng.randn(45, 700) * 0.4
    pk = hm.pack_unwarp_frag_bf16(u, 32)
    assert pk.shape == (32, 2, 3, 64, 8) and pk.dtype == np.uint16
    parts = hm.bf16_split3(u.astype(np.float32))
    for ct, kb, s_, lane, j in ((0, 0, 0, 0, 0), (31, 1, 2, 63, 7), (9, 1, 1, 21, 3), (17, 0, 2, 40, 6)):
        k, col = 32 * kb + 8 * (lane >> 4) + j, 16 * ct + (lane & 15)
        want = parts[s_][k, col] if (k < 45 and col < 700) else np.float32(0)
        got = (np.uint32(pk[ct, kb, s_, lane, j]) << 16).view(np.float32) if hasattr(np.uint32(0), "view") else None
        assert np.uint16(np.float32(want).view(np.uint32) >> 16) == pk[ct, kb, s_, lane, j]


def test_six_term_bf16_product_sum_matches_the_float32_chain():
    """sum_k a_k u_k from the six partial products a_i u_j (i + j <= 2) of the three-way splits, float32 accumulation:
    as accurate as a float32 fmaf chain on operands of the unwarp's size (what v_mfma_f32_16x16x32_bf16 computes)."""
    from magphase_amd import hostmath as hm

    rng = np.random.RandomState(3)
    a = (rng.randn(16, 60) * 3.0).astype(np.float32)
    u = (rng.randn(60, 256) * 0.3).astype(np.float32)
    ref = a.astype(np.float64) @ u.astype(np.float64)
    chain = np.zeros((16, 256), dtype=np.float32)
    for k in range(60):
        chain = (chain + a[:, k:k + 1] * u[k][None, :]).astype(np.float32)
    sa, su = hm.bf16_split3(a), hm.bf16_split3(u)
    acc = np.zeros((16, 256), dtype=np.float32)
    for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):
        for k in range(60):
            acc = (acc + sa[i][:, k:k + 1] * su[j][k][None, :]).astype(np.float32)
    e_chain, e_split = np.max(np.abs(chain - ref)), np.max(np.abs(acc - ref))
    assert e_split <= 2.0 * e_chain + 1e-7, (e_split, e_chain)


def test_data_flow_model_of_the_fused_product():
    """CPU model of csrc/magphase_comp.hip: fuse_unwarp_steps_bf16 + fuse_interp_store for one segment -- the lane <-> element
    maps of v_mfma_f32_16x16x32_bf16 as the kernel uses them (A: row lane & 15, k-slots 32 kb + 8 (lane >> 4) + j; B: the
    packed fragments; C: column lane & 15, rows 4 (lane >> 4) + r), the 16 x 64 tile, the frame table and the four-frames-
    per-instruction interpolation -- against exp(A U) interpolated directly."""
    from magphase_amd import hostmath as hm

    rng = np.random.RandomState(7)
    K, H, n_rows = 60, 2049, 40
    u = rng.randn(K, H) * 0.2
    a_rows = (rng.randn(n_rows, K) * 1.5).astype(np.float32)
    pk = hm.pack_unwarp_frag_bf16(u, 132)                       # [ct][kb][split][lane][8] bf16 bits
    pkf = (pk.astype(np.uint32) << 16).view(np.float32)
    rb, nf = 17, 11                                             # segment: rows 17 .. 32, 11 frames
    row0 = rb + np.sort(rng.randint(0, 15, nf))
    row1 = np.minimum(row0 + rng.randint(0, 2, nf), rb + 15)
    wt = rng.rand(nf).astype(np.float32)
    lane = np.arange(64)
    li, gq = lane & 15, lane >> 4
    # A fragments: three-way split of the 16 rows' coefficients, k-slots of every lane
    asp = hm.bf16_split3(np.pad(a_rows[rb:rb + 16], ((0, 0), (0, 64 - K))))
    afrag = np.stack([[sp[li][:, 32 * kb + 8 * gq[:, None] + np.arange(8)[None, :]][np.arange(64), np.arange(64)]
                       for sp in asp] for kb in range(2)])     # [kb][split][lane][8]
    out = np.zeros((nf, 64 * 33), dtype=np.float32)
    for s in (0, 5, 32):                                        # three of the 33 column steps (the last: bin 2048 alone)
        tile = np.zeros((16, 64), dtype=np.float32)
        for c in range(4):                                      # four 16-bin column tiles of the step
            ct = 4 * s + c
            acc = np.zeros((16, 16), dtype=np.float64)          # C[row][col]
            for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):
                for kb in range(2):
                    # D[m][n] += sum over lanes with (lane & 15 == m resp. n) and the same k-slot of A[m][slot] B[slot][n]
                    A = np.zeros((16, 32))
                    B = np.zeros((32, 16))
                    for l in range(64):
                        A[li[l], 8 * gq[l]:8 * gq[l] + 8] = afrag[kb][i][l]
                        B[8 * gq[l]:8 * gq[l] + 8, li[l]] = pkf[ct, kb, j, l]
                    acc += A @ B
            tile[:, 16 * c:16 * c + 16] = np.exp(acc.astype(np.float32))
        for i0 in range(0, nf, 8):                              # fuse_interp_store: lane = (frame 4 u + (lane >> 4), bin quad lane & 15)
            for uu in range(2):
                if i0 + 4 * uu >= nf:
                    continue
                for l in range(64):
                    f = min(i0 + 4 * uu + (l >> 4), nf - 1)     # table entries past the last frame repeat it
                    q4 = 4 * (l & 15)
                    m0, m1 = tile[row0[f] - rb, q4:q4 + 4], tile[row1[f] - rb, q4:q4 + 4]
                    out[f, 64 * s + q4:64 * s + q4 + 4] = (m1 - m0) * wt[f] + m0
    ref = np.exp(a_rows.astype(np.float64) @ u)
    for s in (0, 5, 32):
        cols = np.arange(64 * s, min(64 * s + 64, H))
        want = ref[row0][:, cols] + (ref[row1][:, cols] - ref[row0][:, cols]) * wt[:, None]
        got = out[:, cols]
        assert np.max(np.abs(got - want) / want) < 5e-6, (s, np.max(np.abs(got - want) / want))
