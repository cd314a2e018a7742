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
