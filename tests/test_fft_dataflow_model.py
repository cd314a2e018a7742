"""
CPU model of the one-wavefront FFT data flow used by magphase_amd/csrc/wave_fft.hpp (lanes x registers,
bit-reversed register order, LDS transpose, cross-lane radix-2 stages, kappa lane permutation, the real-FFT
split/merge with the (64-kappa) partner lane).  It pins the *layout contract* the HIP code relies on, on the
CPU; the HIP code itself is checked against the oracle in the -m gpu tests.
"""
import numpy as np
import pytest


def brev(i, bits):
    r = 0
    for b in range(bits):
        r |= ((i >> b) & 1) << (bits - 1 - b)
    return r


def fft_inreg(x, sign):
    """x: [64, P] complex (lane, reg). radix-2 DIF, output reg i holds index brev(i)."""
    P = x.shape[1]
    x = x.copy()
    s = P // 2
    while s >= 1:
        for g in range(0, P, 2 * s):
            for k in range(s):
                i0, i1 = g + k, g + k + s
                a, b = x[:, i0].copy(), x[:, i1].copy()
                x[:, i0] = a + b
                x[:, i1] = (a - b) * np.exp(sign * 2j * np.pi * k / (2 * s))
        s //= 2
    return x


def kappa(lane, P):
    """lane bits above log2(P) reversed: identity (P=32), bits 4<->5 (P=16), bits 3..5 reversed (P=8)."""
    lp = int(np.log2(P))
    return (lane & (P - 1)) | (brev(lane >> lp, 6 - lp) << lp)


def wave_fft(x, sign):
    """x[lane, j] = z[lane + 64 j]  ->  out[lane, i] = Z[kappa(lane) + 64*brev(i)]"""
    P = x.shape[1]
    M = 64 * P
    LB = int(np.log2(P))
    lanes = np.arange(64)
    y = fft_inreg(x, sign)
    for i in range(P):
        y[:, i] *= np.exp(sign * 2j * np.pi * lanes * brev(i, LB) / M)
    lds = np.zeros((P, 65), dtype=complex)
    for i in range(P):
        lds[brev(i, LB), :64] = y[:, i]
    v = np.zeros((64, P), dtype=complex)
    for lam in range(64):
        v[lam] = lds[lam % P, (lam // P) * P:(lam // P) * P + P]

    # 64-point DIF over l = (lane // P) * P + l': strides S >= P across lanes.  Upper lane of a stride-S butterfly:
    # (oth - own) * W_{2S}^{l'} (by register) * W_{2S/P}^{e}, e = (lane // P) mod (S // P):
    #   S == P: nothing, S == 2P: (sign i)^e (the "rot" of the HIP code), S == 4P (P = 8, S = 32): an eighth root.
    S = 32
    while S >= P:
        out = np.zeros_like(v)
        for lam in range(64):
            own, oth = v[lam], v[lam ^ S]
            if lam & S:
                e = (lam // P) % (S // P)
                out[lam] = (oth - own) * np.exp(sign * 2j * np.pi * np.arange(P) / (2 * S)) \
                    * np.exp(sign * 2j * np.pi * e * P / (2 * S))
            else:
                out[lam] = own + oth
        v = out
        S //= 2
    return fft_inreg(v, sign)


@pytest.mark.parametrize("P", [32, 16, 8])
@pytest.mark.parametrize("sign", [-1, 1])
def test_wave_fft_layout(P, sign):
    rng = np.random.RandomState(P + sign)
    M = 64 * P
    LB = int(np.log2(P))
    z = rng.randn(M) + 1j * rng.randn(M)
    out = wave_fft(z.reshape(P, 64).T.copy(), sign)
    Z = np.fft.fft(z) if sign < 0 else np.fft.ifft(z) * M
    for lam in range(64):
        for i in range(P):
            assert abs(out[lam, i] - Z[kappa(lam, P) + 64 * brev(i, LB)]) < 1e-9


@pytest.mark.parametrize("P", [32, 16, 8])
def test_real_fft_split_with_partner_lane(P):
    """analysis epilogue: X[k] = E + W_N^k O from Z[k] and Z[M-k] fetched from lane kappa^-1((64-kappa)&63)."""
    rng = np.random.RandomState(7)
    M, N, LB = 64 * P, 128 * P, int(np.log2(P))
    y = rng.randn(N)
    z = y[0::2] + 1j * y[1::2]
    Zl = wave_fft(z.reshape(P, 64).T.copy(), -1)
    R = np.fft.rfft(y)
    for lam in range(64):
        kap = kappa(lam, P)
        src = kappa((64 - kap) & 63, P)  # kappa is an involution
        for i in range(P):
            q = brev(i, LB)
            k = kap + 64 * q
            if kap != 0:
                zp = Zl[src, P - 1 - i]
            else:
                zp = Zl[lam, brev((P - q) % P, LB)]
            E = 0.5 * (Zl[lam, i] + np.conj(zp))
            O = -0.5j * (Zl[lam, i] - np.conj(zp))
            X = E + np.exp(-2j * np.pi * kap / N) * np.exp(-2j * np.pi * q / (2 * P)) * O
            assert abs(X - R[k]) < 1e-9
    assert abs((Zl[0, 0].real - Zl[0, 0].imag) - R[M].real) < 1e-9


@pytest.mark.parametrize("P", [32, 16, 8])
def test_real_ifft_merge_with_partner_lane(P):
    """synthesis prologue: Z[k] = E + iO from X[k], X[M-k]; fftshift folded in as (-1)^k; DC/Nyquist imag dropped."""
    rng = np.random.RandomState(8)
    M, N, LB = 64 * P, 128 * P, int(np.log2(P))
    X = np.fft.rfft(rng.randn(N))
    X[0] += 0.3j
    X[M] -= 0.2j
    Xh = X.copy()
    Xh[0], Xh[M] = X[0].real, X[M].real
    ref = np.fft.fftshift(np.fft.irfft(Xh, N))
    Xc = np.zeros((64, P), dtype=complex)
    for l in range(64):
        for j in range(P):
            k = l + 64 * j
            v = X[k] * (-1.0 if (l & 1) else 1.0)
            if k == 0:
                v = v.real
            Xc[l, j] = v
    XM = X[M].real
    Zin = np.zeros((64, P), dtype=complex)
    for l in range(64):
        src = (64 - l) & 63
        for j in range(P):
            k = l + 64 * j
            if l != 0:
                xp = Xc[src, P - 1 - j]
            else:
                xp = Xc[0, (P - j) % P] if j > 0 else XM
            E = Xc[l, j] + np.conj(xp)
            T = Xc[l, j] - np.conj(xp)
            O = np.conj(np.exp(-2j * np.pi * l / N) * np.exp(-2j * np.pi * j / (2 * P))) * T
            Zin[l, j] = (E + 1j * O) * (0.5 / M)
    zl = wave_fft(Zin, +1)
    y = np.zeros(N)
    for lam in range(64):
        for i in range(P):
            m = kappa(lam, P) + 64 * brev(i, LB)
            y[2 * m], y[2 * m + 1] = zl[lam, i].real, zl[lam, i].imag
    assert np.max(np.abs(y - ref)) < 1e-9


def _permlane16_swap(a, b):
    """v_permlane16_swap: the odd 16-lane rows of `a` are exchanged with the even rows of `b` (per wave of 64 lanes)."""
    a, b = a.copy(), b.copy()
    for row in (1, 3):
        lo, hi = 16 * row, 16 * row + 16
        ea, eb = slice(lo, hi), slice(lo - 16, hi - 16)
        a[ea], b[eb] = b[eb].copy(), a[ea].copy()
    return a, b


def _half_swz(row):
    """lds_transpose_half (round 5): the rows 4-11 store their two 16-column halves swapped."""
    return ((row + 4) >> 3) & 1


def test_half_height_transpose_equals_full_transpose():
    """wave_fft.hpp lds_transpose_half (P = 32): two phases of 16 rows, every lane reads 16 columns of row lane % 16, one
    permlane16 swap per register pair -> the full transpose's layout: register l' holds (k1 = lane % 32, l = 32 (lane / 32) + l').
    Rows 4-11 are stored with column c at c ^ 16 (bank layout, see the next test): writes and reads use the same map."""
    P, LB, HP = 32, 5, 16
    rng = np.random.RandomState(3)
    y = rng.randn(64, P)                       # y[lane, i]: register i holds k1 = brev(i), column l = lane
    full = np.zeros((64, P))
    lds = np.zeros((P, 64))
    for i in range(P):
        lds[brev(i, LB)] = y[:, i]
    for lam in range(64):
        full[lam] = lds[lam % P, (lam // P) * P:(lam // P) * P + P]
    t = np.zeros((64, P))
    for h in range(2):
        buf = np.zeros((HP, 64))
        for i in range(P):
            k1 = brev(i, LB)
            if (k1 >> (LB - 1)) == h:
                r = k1 & (HP - 1)
                for lane in range(64):
                    buf[r, lane ^ (HP if _half_swz(r) else 0)] = y[lane, i]
        for lam in range(64):
            r = lam & (HP - 1)
            col = (lam // P) * P + (((lam >> (LB - 1)) & 1) ^ _half_swz(r)) * HP
            t[lam, h * HP:(h + 1) * HP] = buf[r, col:col + HP]
    for j in range(HP):
        t[:, j], t[:, HP + j] = _permlane16_swap(t[:, j], t[:, HP + j])
    assert np.array_equal(t, full)


def test_half_height_transpose_bank_layout():
    """The half-height transpose's ds_read_b128 on MI355X's lane groups: conflict-free with the rows 4-11 swapped; the
    round-4 layout (every row in column order) cost 4 extra LDS cycles per read = 64 per transform -- exactly the
    SQ_LDS_BANK_CONFLICT / frame measured on k_synth_ola_pair (profiles/r05_conflict_ablation.txt).  The dword writes
    (two 32-lane groups, 32 banks) stay conflict-free: the swap permutes lanes inside a group."""
    S = 68
    for q in range(4):
        new = [(l & 15) * S + (l // 32) * 32 + ((((l >> 4) & 1) ^ _half_swz(l & 15)) * 16) + 4 * q for l in range(64)]
        old = [(l & 15) * S + (l // 32) * 32 + ((l >> 4) & 1) * 16 + 4 * q for l in range(64)]
        assert _conflicts(new, B128_READ_GROUPS) == 0
        assert _conflicts(old, B128_READ_GROUPS) == 4
    for r in range(16):
        for half in (range(0, 32), range(32, 64)):
            banks = [(r * S + (l ^ (16 if _half_swz(r) else 0))) % 32 for l in half]
            assert len(set(banks)) == 32


def test_half_twiddle_table_identity():
    """The odd registers' first-pass twiddles are the even ones times W_128^lane (wave_fft_front_compact)."""
    P, LB, M = 32, 5, 2048
    for lane in (0, 1, 17, 63):
        for i in range(0, P, 2):
            w_even = np.exp(2j * np.pi * lane * brev(i, LB) / M)
            w_odd = np.exp(2j * np.pi * lane * brev(i + 1, LB) / M)
            assert brev(i + 1, LB) == brev(i, LB) + 16
            assert abs(w_even * np.exp(2j * np.pi * lane / 128) - w_odd) < 1e-14


# ds_read_b128 / ds_write_b128 are served in fixed lane groups; lanes of a group that touch the same bank (of 64, one per
# 4 bytes) at different addresses cost extra LDS cycles (MI355X_MICROARCH.md, LDS table)
B128_READ_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
                    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
B128_READ_GROUPS += [[l + 32 for l in g] for g in B128_READ_GROUPS]
B128_WRITE_GROUPS = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


def _conflicts(addr_floats, groups):
    """Extra LDS cycles of one 16-byte-per-lane instruction: per group, (max lanes on one bank) - 1, summed."""
    extra = 0
    for g in groups:
        banks = {}
        for lane in g:
            for b in range(4):
                banks.setdefault((addr_floats[lane] + b) % 64, set()).add(addr_floats[lane])
        extra += max(len(v) for v in banks.values()) - 1
    return extra


def _warp_swz(r15):
    return 2 * (r15 - 4) if 4 <= r15 < 12 else 2 * (r15 if r15 < 4 else r15 - 8) + 1


def test_mel_warp_lds_swizzle_is_conflict_free_and_round2_layout_was_not():
    """k_mel_warp_mfma (magphase_comp.hip): fragment reads (lane -> row li = lane & 15, k group g = lane >> 4, step q) and
    staging writes (thread -> chunk c4 = t & 15 of row t >> 4) on the swizzled dense layout; the padded layout of round 2
    (row stride 68) conflicts in every read -- the 31 % SQ_LDS_BANK_CONFLICT of profiles/r02_v12."""
    for wave in range(4):
        for q in range(4):
            new = [64 * (16 * wave + (l & 15)) + 4 * (((4 * (l >> 4)) ^ _warp_swz(l & 15)) ^ q) for l in range(64)]
            old = [68 * (16 * wave + (l & 15)) + 16 * (l >> 4) + 4 * q for l in range(64)]
            assert _conflicts(new, B128_READ_GROUPS) == 0
            assert _conflicts(old, B128_READ_GROUPS) >= 4
    for w in range(4):        # writes: wave w of the 256 staging threads, pass p
        for p in range(4):
            t = [64 * w + l for l in range(64)]
            new = [64 * ((x >> 4) + 16 * p) + 4 * ((x & 15) ^ _warp_swz((x >> 4) & 15)) for x in t]
            assert _conflicts(new, B128_WRITE_GROUPS) == 0
    # the swizzle is a bijection on a row's chunks, and what is written is what is read
    for r in range(16):
        assert sorted((c ^ _warp_swz(r)) for c in range(16)) == list(range(16))


def fft_inreg_dit(x, sign):
    """wave_fft.hpp fft_inreg_dit: x[lane, r] holds element brev(r); radix-2 DIT, out0 = a + w b, out1 = 2 a - out0;
    output register i holds index i."""
    P = x.shape[1]
    x = x.copy()
    s = 1
    while s < P:
        for g in range(0, P, 2 * s):
            for k in range(s):
                i0, i1 = g + k, g + k + s
                w = np.exp(sign * 2j * np.pi * k / (2 * s))
                o0 = x[:, i0] + w * x[:, i1]
                x[:, i1] = 2 * x[:, i0] - o0
                x[:, i0] = o0
        s *= 2
    return x


@pytest.mark.parametrize("sign", [-1, 1])
def test_dit_compact_wave_fft_layout(sign):
    """wave_fft_dit_compact (P = 32): input register brev(j) <- z[l + 64 j], natural-order first-pass twiddles, natural-
    order transposes, cross-lane stage, DIT second pass on bit-reversed registers -> register i holds Z[lane + 64 i]."""
    P, LB, M = 32, 5, 2048
    rng = np.random.RandomState(11 + sign)
    z = rng.randn(M) + 1j * rng.randn(M)
    x = z.reshape(P, 64).T.copy()                          # x[l, j] = z[l + 64 j]
    xin = np.zeros_like(x)
    for j in range(P):
        xin[:, brev(j, LB)] = x[:, j]
    y = fft_inreg_dit(xin, sign)                           # register i: k1 = i
    lanes = np.arange(64)
    for i in range(P):
        y[:, i] *= np.exp(sign * 2j * np.pi * lanes * i / M)
    v = np.zeros((64, P), dtype=complex)
    for lam in range(64):                                  # transposes: register l' <- (row k1 = lam % 32, column 32 (lam // 32) + l')
        for lp in range(P):
            v[lam, lp] = y[(lam // P) * P + lp, lam % P]
    out = np.zeros_like(v)
    for lam in range(64):                                  # cross-lane stage, stride 32
        own, oth = v[lam], v[lam ^ 32]
        out[lam] = (oth - own) * np.exp(sign * 2j * np.pi * np.arange(P) / 64) if lam & 32 else own + oth
    t = np.zeros_like(out)
    for r in range(P):
        t[:, brev(r, LB)] = out[:, r]
    res = fft_inreg_dit(t, sign)
    Z = np.fft.fft(z) if sign < 0 else np.fft.ifft(z) * M
    for lam in range(64):
        for i in range(P):
            assert abs(res[lam, i] - Z[lam + 64 * i]) < 1e-9


def wave_fft_dit(x_brev, sign):
    """wave_fft.hpp wave_fft_dit (any P): x_brev[lane, brev(j)] = z[lane + 64 j] -> out[lane, i] = Z[kappa(lane) + 64 i]:
    DIT first pass (natural k1), natural-order twiddles and transposes, the cross-lane stages of wave_fft, DIT second pass on
    the bit-reversed renaming of the registers."""
    P = x_brev.shape[1]
    M, LB = 64 * P, int(np.log2(P))
    lanes = np.arange(64)
    y = fft_inreg_dit(x_brev, sign)
    for i in range(P):
        y[:, i] *= np.exp(sign * 2j * np.pi * lanes * i / M)
    v = np.zeros((64, P), dtype=complex)
    for lam in range(64):
        for lp in range(P):
            v[lam, lp] = y[(lam // P) * P + lp, lam % P]
    S = 32
    while S >= P:
        out = np.zeros_like(v)
        for lam in range(64):
            own, oth = v[lam], v[lam ^ S]
            if lam & S:
                e = (lam // P) % (S // P)
                out[lam] = (oth - own) * np.exp(sign * 2j * np.pi * np.arange(P) / (2 * S)) \
                    * np.exp(sign * 2j * np.pi * e * P / (2 * S))
            else:
                out[lam] = own + oth
        v = out
        S //= 2
    t = np.zeros_like(v)
    for r in range(P):
        t[:, brev(r, LB)] = v[:, r]
    return fft_inreg_dit(t, sign)


@pytest.mark.parametrize("P", [32, 16, 8])
@pytest.mark.parametrize("sign", [-1, 1])
def test_dit_wave_fft_layout(P, sign):
    rng = np.random.RandomState(3 * P + sign)
    M, LB = 64 * P, int(np.log2(P))
    z = rng.randn(M) + 1j * rng.randn(M)
    x = z.reshape(P, 64).T.copy()
    xin = np.zeros_like(x)
    for j in range(P):
        xin[:, brev(j, LB)] = x[:, j]
    res = wave_fft_dit(xin, sign)
    Z = np.fft.fft(z) if sign < 0 else np.fft.ifft(z) * M
    for lam in range(64):
        for i in range(P):
            assert abs(res[lam, i] - Z[kappa(lam, P) + 64 * i]) < 1e-9


@pytest.mark.parametrize("P", [32, 16, 8])
def test_real_fft_split_in_natural_register_order(P):
    """The split of the DIT form (k_analysis_f64, noise_spectrum_paired): lane kappa owns k = kappa + 64 q in register q;
    Z[M - k] lives in lane (64 - kappa) & 63, register P - 1 - q (kappa == 0: own register (P - q) % P); bin M / 2 is
    register P / 2 of the kappa == 0 lane."""
    M, N, LB = 64 * P, 128 * P, int(np.log2(P))
    rng = np.random.RandomState(P)
    xr = rng.randn(N)
    z = xr[0::2] + 1j * xr[1::2]
    x = z.reshape(P, 64).T.copy()
    xin = np.zeros_like(x)
    for j in range(P):
        xin[:, brev(j, LB)] = x[:, j]
    Zl = wave_fft_dit(xin, -1)
    X = np.fft.fft(xr)
    inv = {kappa(l, P): l for l in range(64)}
    for lam in range(64):
        kap = kappa(lam, P)
        src = inv[(64 - kap) & 63]
        for q in range(P // 2):
            own = Zl[lam, q]
            par = Zl[lam, (P - q) % P] if kap == 0 else Zl[src, P - 1 - q]
            k = kap + 64 * q
            E = 0.5 * (own + np.conj(par))
            O = -0.5j * (own - np.conj(par))
            T = np.exp(-2j * np.pi * k / N) * O
            assert abs(E + T - X[k]) < 1e-9
            if not (kap == 0 and q == 0):
                assert abs(np.conj(E - T) - X[M - k]) < 1e-9
    l0 = inv[0]
    assert abs(np.conj(Zl[l0, P // 2]) - X[M // 2]) < 1e-9
