// wave_fft_f64.hpp -- double-precision instantiation of the one-wavefront FFT (forward transform only), used by the
// compressed-analysis path: its log / division epilogues amplify the transform's round-off on bins 60-80 dB below the
// frame peak, where an fp32 FFT (noise floor ~1e-6 of the peak) leaves 1e-3 relative errors.  Same data flow and layout
// contract as wave_fft.hpp (lane l, register j <-> z[l + 64 j] in; lane l, register i <-> Z[kappa(l) + 64 brev(i)] out;
// tests/test_fft_dataflow_model.py), with
//   * float64 registers and butterflies (literal twiddles in double),
//   * a float64 first-pass twiddle table in LDS: one row per lane, entry i = (cos, sin)(2 pi lane brev(i) / M), 16 bytes
//     per entry = one ds_read_b128, rows padded by 16 bytes (conflict-free across lanes),
//   * the per-wave LDS transpose run on the low and the high 32-bit words of a plane separately (the fp32 buffer and
//     code: 4 plane transposes per transform instead of 2),
//   * lane exchanges on both words (v_permlane32_swap / v_permlane16_swap / DPP).
// fp64 vector rate on gfx950 is half the fp32 rate; the path this serves is not bandwidth-bound.
#pragma once
#include "wave_fft.hpp"

namespace mpx {

// cos / sin (2 pi k / 64), k < 32, in double
__device__ __forceinline__ constexpr double dc64(int k) {
    constexpr double t[32] = {1.00000000000000000e+00, 9.95184726672196929e-01, 9.80785280403230431e-01, 9.56940335732208824e-01, 9.23879532511286738e-01, 8.81921264348355050e-01, 8.31469612302545236e-01, 7.73010453362736993e-01, 7.07106781186547573e-01, 6.34393284163645488e-01, 5.55570233019602289e-01, 4.71396736825997809e-01, 3.82683432365089837e-01, 2.90284677254462331e-01, 1.95090322016128331e-01, 9.80171403295607702e-02, 6.12323399573676604e-17, -9.80171403295606453e-02, -1.95090322016128193e-01, -2.90284677254462165e-01, -3.82683432365089726e-01, -4.71396736825997698e-01, -5.55570233019601956e-01, -6.34393284163645377e-01, -7.07106781186547462e-01, -7.73010453362736993e-01, -8.31469612302545347e-01, -8.81921264348354939e-01, -9.23879532511286738e-01, -9.56940335732208824e-01, -9.80785280403230431e-01, -9.95184726672196818e-01};
    return t[k];
}
__device__ __forceinline__ constexpr double ds64(int k) {
    constexpr double t[32] = {0.00000000000000000e+00, 9.80171403295606036e-02, 1.95090322016128248e-01, 2.90284677254462331e-01, 3.82683432365089782e-01, 4.71396736825997642e-01, 5.55570233019602178e-01, 6.34393284163645488e-01, 7.07106781186547462e-01, 7.73010453362736993e-01, 8.31469612302545236e-01, 8.81921264348354939e-01, 9.23879532511286738e-01, 9.56940335732208935e-01, 9.80785280403230431e-01, 9.95184726672196818e-01, 1.00000000000000000e+00, 9.95184726672196929e-01, 9.80785280403230431e-01, 9.56940335732208935e-01, 9.23879532511286738e-01, 8.81921264348355050e-01, 8.31469612302545458e-01, 7.73010453362737104e-01, 7.07106781186547573e-01, 6.34393284163645488e-01, 5.55570233019602178e-01, 4.71396736825997864e-01, 3.82683432365089893e-01, 2.90284677254462387e-01, 1.95090322016128609e-01, 9.80171403295608257e-02};
    return t[k];
}

template <int P>
__host__ __device__ constexpr int tw64_stride() { return 2 * P + 2; }     // doubles per lane row (2 of padding)
template <int P>
__host__ __device__ constexpr int tw64_doubles() { return 64 * tw64_stride<P>(); }

__device__ __forceinline__ void split64(double v, unsigned& lo, unsigned& hi) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    lo = (unsigned)b;
    hi = (unsigned)(b >> 32);
}
__device__ __forceinline__ double join64(unsigned lo, unsigned hi) {
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <int P, int SIGN>
__device__ __forceinline__ void fft_inreg_f64(double (&re)[P], double (&im)[P]) {
#pragma unroll
    for (int s = P / 2; s >= 1; s >>= 1) {
#pragma unroll
        for (int g = 0; g < P; g += 2 * s) {
#pragma unroll
            for (int k = 0; k < s; ++k) {
                const int i0 = g + k, i1 = g + k + s;
                const int t = k * (16 / s);  // twiddle W_{2s}^k = W_32^t = W_64^{2t}
                const double ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
                re[i0] = ar + br;
                im[i0] = ai + bi;
                const double tr = ar - br, ti = ai - bi;
                if (t == 0) {
                    re[i1] = tr;
                    im[i1] = ti;
                } else if (t == 8) {
                    re[i1] = (SIGN < 0) ? ti : -ti;
                    im[i1] = (SIGN < 0) ? -tr : tr;
                } else {
                    const double c = dc64(2 * t), sn = (SIGN < 0) ? -ds64(2 * t) : ds64(2 * t);
                    re[i1] = tr * c - ti * sn;
                    im[i1] = tr * sn + ti * c;
                }
            }
        }
    }
}

// The DIT form of the in-register transform (wave_fft.hpp fft_inreg_dit): input register r holds element brev(r), output
// register i holds index i; out0 = a + w b, out1 = 2 a - out0 in fused multiply-adds -- 6 instead of 8 float64
// instructions per general butterfly (68 fewer per 32-point transform), and float64 instructions are what k_analysis_f64
// is made of.
template <int P, int SIGN>
__device__ __forceinline__ void fft_inreg_dit_f64(double (&re)[P], double (&im)[P]) {
#pragma unroll
    for (int s = 1; s < P; s <<= 1) {
#pragma unroll
        for (int g = 0; g < P; g += 2 * s) {
#pragma unroll
            for (int k = 0; k < s; ++k) {
                const int i0 = g + k, i1 = g + k + s;
                const int t = k * (16 / s);  // twiddle W_{2s}^k = W_32^t = W_64^{2t}
                const double ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
                if (t == 0) {
                    re[i0] = ar + br;
                    im[i0] = ai + bi;
                    re[i1] = ar - br;
                    im[i1] = ai - bi;
                } else if (t == 8) {  // w = SIGN * i: w b = SIGN (-bi, br)
                    re[i0] = (SIGN < 0) ? ar + bi : ar - bi;
                    im[i0] = (SIGN < 0) ? ai - br : ai + br;
                    re[i1] = (SIGN < 0) ? ar - bi : ar + bi;
                    im[i1] = (SIGN < 0) ? ai + br : ai - br;
                } else {
                    const double c = dc64(2 * t), sn = (SIGN < 0) ? -ds64(2 * t) : ds64(2 * t);
                    const double o0r = fma(br, c, fma(-bi, sn, ar));
                    const double o0i = fma(br, sn, fma(bi, c, ai));
                    re[i0] = o0r;
                    im[i0] = o0i;
                    re[i1] = fma(2.0, ar, -o0r);
                    im[i1] = fma(2.0, ai, -o0i);
                }
            }
        }
    }
}

// plane transpose through the fp32 buffer: low words, then high words (NAT: natural input register order)
template <int P, bool NAT = false>
__device__ __forceinline__ void lds_transpose_f64(double (&x)[P], float* xbuf, int lane) {
    float lo[P], hi[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        unsigned a, b;
        split64(x[i], a, b);
        lo[i] = __builtin_bit_cast(float, a);
        hi[i] = __builtin_bit_cast(float, b);
    }
    lds_transpose<P, NAT>(lo, xbuf, lane);
    lds_transpose<P, NAT>(hi, xbuf, lane);
#pragma unroll
    for (int i = 0; i < P; ++i) x[i] = join64(__builtin_bit_cast(unsigned, lo[i]), __builtin_bit_cast(unsigned, hi[i]));
}

// value of lane ^ PARTNER (one double)
template <int PARTNER>
__device__ __forceinline__ double lane_xor_f64(double v) {
    unsigned lo, hi;
    split64(v, lo, hi);
    if (PARTNER == 8) {
        lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x128, 0xf, 0xf, false);
        hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x128, 0xf, 0xf, false);
    } else {
        lo = (unsigned)__shfl_xor((int)lo, PARTNER);
        hi = (unsigned)__shfl_xor((int)hi, PARTNER);
    }
    return join64(lo, hi);
}

template <int P, int SIGN, int PARTNER, int TWN>
__device__ __forceinline__ void cross_lane_stage_f64(double (&re)[P], double (&im)[P], bool upper, bool rot, int lane) {
    if (PARTNER >= 16) {
        // in place, registers in pairs (lp, lp + 1), on the low and the high words: swap / add-subtract / swap (see
        // cross_lane_stage in wave_fft.hpp) -- no second register set (2P lane exchanges in flight at once would need
        // 4P more registers than the 4P of the data)
        auto swp = [](double& a, double& b) {
            unsigned al, ah, bl, bh;
            split64(a, al, ah);
            split64(b, bl, bh);
            float fal = __builtin_bit_cast(float, al), fah = __builtin_bit_cast(float, ah);
            float fbl = __builtin_bit_cast(float, bl), fbh = __builtin_bit_cast(float, bh);
            if (PARTNER == 32) { swap32(fal, fbl); swap32(fah, fbh); } else { swap16(fal, fbl); swap16(fah, fbh); }
            a = join64(__builtin_bit_cast(unsigned, fal), __builtin_bit_cast(unsigned, fah));
            b = join64(__builtin_bit_cast(unsigned, fbl), __builtin_bit_cast(unsigned, fbh));
        };
#pragma unroll
        for (int lp = 0; lp < P; lp += 2) {
            swp(re[lp], re[lp + 1]);
            swp(im[lp], im[lp + 1]);
        }
#pragma unroll
        for (int lp = 0; lp < P; lp += 2) {
            const double a = re[lp], b = re[lp + 1], c = im[lp], d = im[lp + 1];
            re[lp] = a + b;
            re[lp + 1] = a - b;
            im[lp] = c + d;
            im[lp + 1] = c - d;
        }
#pragma unroll
        for (int lp = 0; lp < P; lp += 2) {
            swp(re[lp], re[lp + 1]);
            swp(im[lp], im[lp + 1]);
        }
    } else {
        const double sg = upper ? -1.0 : 1.0;
#pragma unroll
        for (int lp = 0; lp < P; ++lp) {   // own * (+-1) + partner: lower lane own + oth, upper lane oth - own
            const double orr = lane_xor_f64<PARTNER>(re[lp]);
            const double oii = lane_xor_f64<PARTNER>(im[lp]);
            re[lp] = fma(re[lp], sg, orr);
            im[lp] = fma(im[lp], sg, oii);
        }
    }
    if (upper) {
#pragma unroll
        for (int lp = 1; lp < P; ++lp) {
            const int k = (TWN == 64) ? lp : ((TWN == 32) ? 2 * lp : 4 * lp);   // W_TWN^lp = W_64^k
            const double cw = dc64(k), sw = (SIGN < 0) ? -ds64(k) : ds64(k);
            const double xr = re[lp] * cw - im[lp] * sw;
            const double xi = re[lp] * sw + im[lp] * cw;
            re[lp] = xr;
            im[lp] = xi;
        }
    }
    if (PARTNER == 4 * P) {   // P = 8, stride 32: W_8^e, e = (lane / P) & 3, on the upper lanes
        const int e = (lane >> 3) & 3;
        constexpr double kR2 = 7.07106781186547524e-01;
        const double fc = (e == 0) ? 1.0 : ((e == 1) ? kR2 : ((e == 2) ? 0.0 : -kR2));
        const double fs0 = (e == 0) ? 0.0 : ((e == 1) ? kR2 : ((e == 2) ? 1.0 : kR2));
        const double fs = (SIGN < 0) ? -fs0 : fs0;
        if (upper) {
#pragma unroll
            for (int lp = 0; lp < P; ++lp) {
                const double xr = re[lp] * fc - im[lp] * fs;
                const double xi = re[lp] * fs + im[lp] * fc;
                re[lp] = xr;
                im[lp] = xi;
            }
        }
    }
    if (PARTNER == 2 * P) {
        if (rot) {
#pragma unroll
            for (int lp = 0; lp < P; ++lp) {
                const double xr = re[lp], xi = im[lp];
                re[lp] = (SIGN < 0) ? xi : -xi;
                im[lp] = (SIGN < 0) ? -xr : xr;
            }
        }
    }
}

// tw: this workgroup's float64 twiddle table in LDS (tw64_doubles<P>() doubles); xbuf: the wave's fp32 transpose buffer
template <int P, int SIGN>
__device__ __forceinline__ void wave_fft_f64(double (&re)[P], double (&im)[P], const double* tw, float* xbuf, int lane) {
    fft_inreg_f64<P, SIGN>(re, im);
    const double2* trow = reinterpret_cast<const double2*>(tw + lane * tw64_stride<P>());
#pragma unroll
    for (int i = 0; i < P; ++i) {
        // eight table reads in flight at a time: left alone, the scheduler issues all P of them up front (4 registers
        // each on top of the 4P of the data: the P = 32 kernel spills)
        if ((i & 7) == 0) asm volatile("" ::: "memory");
        const double2 w = trow[i];
        const double ws = (SIGN < 0) ? -w.y : w.y;
        const double xr = re[i] * w.x - im[i] * ws;
        const double xi = re[i] * ws + im[i] * w.x;
        re[i] = xr;
        im[i] = xi;
    }
    lds_transpose_f64<P>(re, xbuf, lane);
    lds_transpose_f64<P>(im, xbuf, lane);
    if (P == 32) {
        cross_lane_stage_f64<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, false, lane);
    } else if (P == 16) {
        cross_lane_stage_f64<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, (lane & 48) == 48, lane);
        cross_lane_stage_f64<P, SIGN, 16, 32>(re, im, (lane & 16) != 0, false, lane);
    } else {
        cross_lane_stage_f64<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, false, lane);
        cross_lane_stage_f64<P, SIGN, 16, 32>(re, im, (lane & 16) != 0, (lane & 24) == 24, lane);
        cross_lane_stage_f64<P, SIGN, 8, 16>(re, im, (lane & 8) != 0, false, lane);
    }
    fft_inreg_f64<P, SIGN>(re, im);
}

// DIT form (layout of wave_fft_dit in wave_fft.hpp): input register brev(j) <- z[l + 64 j], output register i holds
// Z[kappa(l) + 64 i]; twn: the float64 table with its rows in NATURAL order (entry i of lane l = W_M^{l i}).
template <int P, int SIGN>
__device__ __forceinline__ void wave_fft_dit_f64(double (&re)[P], double (&im)[P], const double* twn, float* xbuf, int lane) {
    constexpr int LB = ilog2(P);
    fft_inreg_dit_f64<P, SIGN>(re, im);
    const double2* trow = reinterpret_cast<const double2*>(twn + lane * tw64_stride<P>());
#pragma unroll
    for (int i = 0; i < P; ++i) {
        if ((i & 7) == 0) asm volatile("" ::: "memory");   // eight table reads in flight at a time (see wave_fft_f64)
        const double2 w = trow[i];
        const double ws = (SIGN < 0) ? -w.y : w.y;
        const double xr = re[i] * w.x - im[i] * ws;
        const double xi = re[i] * ws + im[i] * w.x;
        re[i] = xr;
        im[i] = xi;
    }
    lds_transpose_f64<P, true>(re, xbuf, lane);
    lds_transpose_f64<P, true>(im, xbuf, lane);
    if (P == 32) {
        cross_lane_stage_f64<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, false, lane);
    } else if (P == 16) {
        cross_lane_stage_f64<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, (lane & 48) == 48, lane);
        cross_lane_stage_f64<P, SIGN, 16, 32>(re, im, (lane & 16) != 0, false, lane);
    } else {
        cross_lane_stage_f64<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, false, lane);
        cross_lane_stage_f64<P, SIGN, 16, 32>(re, im, (lane & 16) != 0, (lane & 24) == 24, lane);
        cross_lane_stage_f64<P, SIGN, 8, 16>(re, im, (lane & 8) != 0, false, lane);
    }
    double tr[P], ti[P];
#pragma unroll
    for (int r = 0; r < P; ++r) {
        tr[brev(r, LB)] = re[r];
        ti[brev(r, LB)] = im[r];
    }
    fft_inreg_dit_f64<P, SIGN>(tr, ti);
#pragma unroll
    for (int r = 0; r < P; ++r) {
        re[r] = tr[r];
        im[r] = ti[r];
    }
}

}  // namespace mpx
