// wave_fft.hpp -- one-wavefront complex FFT for gfx950 (CDNA4), M = 64*P points, P in {8, 16, 32}.
//
// Layout contract (validated against numpy in tests/test_fft_dataflow_model.py):
//   input : lane l, register j      holds z[l + 64*j]                       (natural order)
//   pass 1: P-point DIF FFT in registers over j            -> register i holds k1 = brev(i)
//           multiply by W_M^(l*k1)  (table in LDS, shared by the workgroup, one padded row per lane in register
//           order: two twiddles per ds_read_b128)
//           transpose through a per-wave LDS buffer (row stride 68 floats: conflict-free dword writes with the lane
//           along a row, 16-byte aligned rows so that a lane reads its P consecutive floats as P/4 ds_read_b128 --
//           a quarter of the LDS instructions of the dword form; bank model in tests/test_fft_dataflow_model.py)
//   pass 2: 64-point FFT over l = radix-(64/P) butterflies ACROSS lanes + P-point DIF in registers
//   output: lane l, register i      holds Z[kappa(l) + 64*brev(i)]
//           kappa(l) = l with its bits above log2(P) reversed: identity for P=32, bits 4 and 5 swapped for P=16,
//           bits 3..5 reversed for P=8 (an involution in every case).
//   pass 2 in full: the 64-point DIF FFT over l = (lane / P) * P + l' has its strides S >= P across lanes (partner =
//   lane ^ S, S = 32 .. P; exchanged with v_permlane32_swap / v_permlane16_swap / DPP row_ror:8 -- VALU lane
//   crossbars, no LDS round trip) and the strides < P in registers.  The upper lane of a stride-S butterfly multiplies by
//   W_{2S}^(l mod S) = W_{2S}^{l'} (literal, by register) x W_{2S/P}^{e}, e = (lane / P) mod (S / P): nothing for
//   S == P, (SIGN i)^e for S == 2P ("rot"), a general eighth root for S == 4P (only P = 8, S = 32).
//
// No __syncthreads(): every wave owns its LDS buffer; ordering is wave-local (LDS ops of one wave
// execute in issue order), the fences below only stop the compiler from reordering.
#pragma once
#include <hip/hip_runtime.h>

namespace mpx {

constexpr int kXStride = 68;  // floats per k1 row of the per-wave transpose buffer (16-byte aligned rows)

// Twiddle table of the first pass: one row per lane, entry i = (cos, sin)(2 pi lane brev(i) / M) in REGISTER order,
// rows padded to 2P + 4 floats (16-byte aligned, conflict-free ds_read_b128 across lanes).
template <int P>
__host__ __device__ constexpr int tw_stride() { return 2 * P + 4; }
template <int P>
__host__ __device__ constexpr int tw_floats() { return 64 * tw_stride<P>(); }

__host__ __device__ constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

__host__ __device__ constexpr int brev(int i, int bits) {
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((i >> b) & 1) << (bits - 1 - b);
    return r;
}

// cos/sin(2*pi*k/32), k < 16
__device__ __forceinline__ constexpr float c32(int k) {
    constexpr float t[16] = {1.000000000e+00f, 9.807852507e-01f, 9.238795042e-01f, 8.314695954e-01f,
                             7.071067691e-01f, 5.555702448e-01f, 3.826834261e-01f, 1.950903237e-01f,
                             0.0f,             -1.950903237e-01f, -3.826834261e-01f, -5.555702448e-01f,
                             -7.071067691e-01f, -8.314695954e-01f, -9.238795042e-01f, -9.807852507e-01f};
    return t[k];
}
__device__ __forceinline__ constexpr float s32(int k) {
    constexpr float t[16] = {0.000000000e+00f, 1.950903237e-01f, 3.826834261e-01f, 5.555702448e-01f,
                             7.071067691e-01f, 8.314695954e-01f, 9.238795042e-01f, 9.807852507e-01f,
                             1.000000000e+00f, 9.807852507e-01f, 9.238795042e-01f, 8.314695954e-01f,
                             7.071067691e-01f, 5.555702448e-01f, 3.826834261e-01f, 1.950903237e-01f};
    return t[k];
}
// cos/sin(2*pi*k/64), k < 32
__device__ __forceinline__ constexpr float c64(int k) {
    constexpr float t[32] = {1.000000000e+00f, 9.951847196e-01f, 9.807852507e-01f, 9.569403529e-01f,
                             9.238795042e-01f, 8.819212914e-01f, 8.314695954e-01f, 7.730104327e-01f,
                             7.071067691e-01f, 6.343932748e-01f, 5.555702448e-01f, 4.713967443e-01f,
                             3.826834261e-01f, 2.902846634e-01f, 1.950903237e-01f, 9.801714122e-02f,
                             0.0f,             -9.801714122e-02f, -1.950903237e-01f, -2.902846634e-01f,
                             -3.826834261e-01f, -4.713967443e-01f, -5.555702448e-01f, -6.343932748e-01f,
                             -7.071067691e-01f, -7.730104327e-01f, -8.314695954e-01f, -8.819212914e-01f,
                             -9.238795042e-01f, -9.569403529e-01f, -9.807852507e-01f, -9.951847196e-01f};
    return t[k];
}
__device__ __forceinline__ constexpr float s64(int k) {
    constexpr float t[32] = {0.000000000e+00f, 9.801714122e-02f, 1.950903237e-01f, 2.902846634e-01f,
                             3.826834261e-01f, 4.713967443e-01f, 5.555702448e-01f, 6.343932748e-01f,
                             7.071067691e-01f, 7.730104327e-01f, 8.314695954e-01f, 8.819212914e-01f,
                             9.238795042e-01f, 9.569403529e-01f, 9.807852507e-01f, 9.951847196e-01f,
                             1.000000000e+00f, 9.951847196e-01f, 9.807852507e-01f, 9.569403529e-01f,
                             9.238795042e-01f, 8.819212914e-01f, 8.314695954e-01f, 7.730104327e-01f,
                             7.071067691e-01f, 6.343932748e-01f, 5.555702448e-01f, 4.713967443e-01f,
                             3.826834261e-01f, 2.902846634e-01f, 1.950903237e-01f, 9.801714122e-02f};
    return t[k];
}

// e^{SIGN * 2*pi*i * q / (2P)}: register part of the real-FFT split twiddle W_N^{64 q}, N = 128 P.
template <int P>
__device__ __forceinline__ constexpr float cos2p(int q) { return P == 32 ? c64(q) : (P == 16 ? c32(q) : c32(2 * q)); }
template <int P>
__device__ __forceinline__ constexpr float sin2p(int q) { return P == 32 ? s64(q) : (P == 16 ? s32(q) : s32(2 * q)); }

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------
// P-point radix-2 DIF FFT on statically indexed registers; output register i holds index brev(i).
// SIGN = -1: forward (e^{-i...}), +1: inverse (unnormalised).
// ---------------------------------------------------------------------------------------------
template <int P, int SIGN>
__device__ __forceinline__ void fft_inreg(float (&re)[P], float (&im)[P]) {
#pragma unroll
    for (int s = P / 2; s >= 1; s >>= 1) {
#pragma unroll
        for (int g = 0; g < P; g += 2 * s) {
#pragma unroll
            for (int k = 0; k < s; ++k) {
                const int i0 = g + k, i1 = g + k + s;
                const int t = k * (16 / s);  // twiddle W_{2s}^k = W_32^t, t in [0,16)
                const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
                re[i0] = ar + br;
                im[i0] = ai + bi;
                const float tr = ar - br, ti = ai - bi;
                if (t == 0) {
                    re[i1] = tr;
                    im[i1] = ti;
                } else if (t == 8) {  // W = SIGN * i
                    re[i1] = (SIGN < 0) ? ti : -ti;
                    im[i1] = (SIGN < 0) ? -tr : tr;
                } else {
                    const float c = c32(t), sn = (SIGN < 0) ? -s32(t) : s32(t);
                    re[i1] = tr * c - ti * sn;
                    im[i1] = tr * sn + ti * c;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// P-point radix-2 DIT FFT on statically indexed registers: input register r holds element brev(r), output register i
// holds index i (natural).  Same transform as fft_inreg; the butterflies are out0 = a + w b, out1 = 2 a - out0 in fused
// multiply-adds: 6 instructions for a general twiddle where the DIF form (sum, difference, complex multiply) takes 8 --
// 68 fewer per 32-point transform.  Used where VALU issue is the bound (k_synth_ola_pair).
// ---------------------------------------------------------------------------------------------
template <int P, int SIGN>
__device__ __forceinline__ void fft_inreg_dit(float (&re)[P], float (&im)[P]) {
#pragma unroll
    for (int s = 1; s < P; s <<= 1) {
#pragma unroll
        for (int g = 0; g < P; g += 2 * s) {
#pragma unroll
            for (int k = 0; k < s; ++k) {
                const int i0 = g + k, i1 = g + k + s;
                const int t = k * (16 / s);  // twiddle W_{2s}^k = W_32^t, t in [0,16)
                const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
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
                    const float c = c32(t), sn = (SIGN < 0) ? -s32(t) : s32(t);
                    const float o0r = fmaf(br, c, fmaf(-bi, sn, ar));
                    const float o0i = fmaf(br, sn, fmaf(bi, c, ai));
                    re[i0] = o0r;
                    im[i0] = o0i;
                    re[i1] = fmaf(2.0f, ar, -o0r);
                    im[i1] = fmaf(2.0f, ai, -o0i);
                }
            }
        }
    }
}

// Per-wave LDS transpose of one plane: in: register i holds (k1 = brev(i), l = lane);
// out: register l' holds (k1 = lane % P, l = (lane / P) * P + l').
// NAT: the input registers are in natural order (register i holds k1 = i: the DIT first pass) instead of bit-reversed.
template <int P, bool NAT = false>
__device__ __forceinline__ void lds_transpose(float (&x)[P], float* xbuf, int lane) {
    constexpr int LB = ilog2(P);
#pragma unroll
    for (int i = 0; i < P; ++i) xbuf[(NAT ? i : brev(i, LB)) * kXStride + lane] = x[i];
    wave_sync();
    const float4* src = reinterpret_cast<const float4*>(xbuf + (lane % P) * kXStride + (lane / P) * P);
#pragma unroll
    for (int q = 0; q < P / 4; ++q) {
        const float4 v = src[q];
        x[4 * q + 0] = v.x;
        x[4 * q + 1] = v.y;
        x[4 * q + 2] = v.z;
        x[4 * q + 3] = v.w;
    }
    wave_sync();
}

// value of lane ^ PARTNER for the two registers (a, b) at once.  PARTNER 32 / 16: a swap instruction exchanges the
// upper half (odd 16-lane rows) of one register with the lower half (even rows) of the other: two swaps around the
// add / subtract give both butterflies with no LDS traffic.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// (the elements are copied to scalars before the bit cast: __builtin_bit_cast straight from a vector element reads
// element 0 for both under hipcc / ROCm 7.2)
__device__ __forceinline__ void swap32(float& a, float& b) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b),
                                                     false, false);
    const unsigned r0 = r.x, r1 = r.y;
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void swap16(float& a, float& b) {
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b),
                                                     false, false);
    const unsigned r0 = r.x, r1 = r.y;
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}

// Radix-2 DIF butterflies across lanes (partner = lane ^ PARTNER).  Lower lane: own + oth; upper lane:
// (oth - own) * e^{SIGN 2 pi i lp / TWN}, times (SIGN*i) where `rot` (only the P=16 first stage).
// The twiddle multiply sits in ONE exec-masked region with literal constants: selecting the constants per
// lane instead makes them loop-invariant VGPRs that LICM hoists out of the frame loop (64 registers).
template <int P, int SIGN, int PARTNER, int TWN>
__device__ __forceinline__ void cross_lane_stage(float (&re)[P], float (&im)[P], bool upper, bool rot, int lane) {
    if (PARTNER >= 16) {
        // registers in pairs (lp, lp + 1): after the first swap a = [lower halves of both], b = [upper halves of both];
        // a + b / a - b are the lower / upper outputs of both registers; the second swap sorts them back.
        // Three batches (swap / add-subtract / swap): a swap must not read a register written by one of the two VALU
        // instructions before it (the compiler pads with s_nop otherwise); batched, every operand is older than that.
#pragma unroll
        for (int lp = 0; lp < P; lp += 2) {
            if (PARTNER == 32) { swap32(re[lp], re[lp + 1]); swap32(im[lp], im[lp + 1]); }
            else { swap16(re[lp], re[lp + 1]); swap16(im[lp], im[lp + 1]); }
        }
#pragma unroll
        for (int lp = 0; lp < P; lp += 2) {
            const float a = re[lp], b = re[lp + 1], c = im[lp], d = im[lp + 1];
            re[lp] = a + b;
            re[lp + 1] = a - b;
            im[lp] = c + d;
            im[lp + 1] = c - d;
        }
#pragma unroll
        for (int lp = 0; lp < P; lp += 2) {
            if (PARTNER == 32) { swap32(re[lp], re[lp + 1]); swap32(im[lp], im[lp + 1]); }
            else { swap16(re[lp], re[lp + 1]); swap16(im[lp], im[lp + 1]); }
        }
    } else {   // PARTNER == 8 (only P = 8): rotation by 8 inside each 16-lane row (DPP row_ror:8)
        const float sg = upper ? -1.0f : 1.0f;
#pragma unroll
        for (int lp = 0; lp < P; ++lp) {
            const float orr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, re[lp]), 0x128, 0xf, 0xf, false));
            const float oii = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, im[lp]), 0x128, 0xf, 0xf, false));
            re[lp] = fmaf(re[lp], sg, orr);
            im[lp] = fmaf(im[lp], sg, oii);
        }
    }
    if (upper) {
#pragma unroll
        for (int lp = 1; lp < P; ++lp) {
            const float cw = (TWN == 64) ? c64(lp) : ((TWN == 32) ? c32(lp) : c32(2 * lp));
            const float sw0 = (TWN == 64) ? s64(lp) : ((TWN == 32) ? s32(lp) : s32(2 * lp));
            const float sw = (SIGN < 0) ? -sw0 : sw0;
            const float xr = re[lp] * cw - im[lp] * sw;
            const float xi = re[lp] * sw + im[lp] * cw;
            re[lp] = xr;
            im[lp] = xi;
        }
    }
    if (PARTNER == 4 * P) {   // P = 8, stride 32: W_8^e, e = (lane / P) & 3, on the upper lanes
        const int e = (lane >> 3) & 3;
        constexpr float kR2 = 7.071067691e-01f;
        const float fc = (e == 0) ? 1.0f : ((e == 1) ? kR2 : ((e == 2) ? 0.0f : -kR2));
        const float fs0 = (e == 0) ? 0.0f : ((e == 1) ? kR2 : ((e == 2) ? 1.0f : kR2));
        const float fs = (SIGN < 0) ? -fs0 : fs0;
        if (upper) {
#pragma unroll
            for (int lp = 0; lp < P; ++lp) {
                const float xr = re[lp] * fc - im[lp] * fs;
                const float xi = re[lp] * fs + im[lp] * fc;
                re[lp] = xr;
                im[lp] = xi;
            }
        }
    }
    if (PARTNER == 2 * P) {
        if (rot) {  // W_4^{e}: multiply by SIGN*i
#pragma unroll
            for (int lp = 0; lp < P; ++lp) {
                const float xr = re[lp], xi = im[lp];
                re[lp] = (SIGN < 0) ? xi : -xi;
                im[lp] = (SIGN < 0) ? -xr : xr;
            }
        }
    }
}

// kappa(lane): which residue (mod 64) of the output index this lane holds after wave_fft.
template <int P>
__device__ __forceinline__ int kappa(int lane) {
    if (P == 32) return lane;
    if (P == 16) return (lane & 15) | (((lane >> 5) & 1) << 4) | (((lane >> 4) & 1) << 5);
    return (lane & 7) | (((lane >> 5) & 1) << 3) | (lane & 16) | (((lane >> 3) & 1) << 5);   // P == 8: bits 3..5 reversed
}

// Full M = 64*P point FFT of one wave, in two halves so that a caller can place independent work
// (e.g. the next frame's global loads) between them.  tw: LDS twiddle table (tw_floats<P>() floats, layout above; sign
// applied here).  xbuf: this wave's P * kXStride float LDS buffer.
template <int P, int SIGN>
__device__ __forceinline__ void wave_fft_front(float (&re)[P], float (&im)[P], const float* tw, float* xbuf, int lane) {
    fft_inreg<P, SIGN>(re, im);
    const float4* trow = reinterpret_cast<const float4*>(tw + lane * tw_stride<P>());
#pragma unroll
    for (int q = 0; q < P / 2; ++q) {
        const float4 w = trow[q];   // twiddles of registers 2q, 2q + 1
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * q + e;
            const float wc = e ? w.z : w.x;
            const float ws = (SIGN < 0) ? -(e ? w.w : w.y) : (e ? w.w : w.y);
            const float xr = re[i] * wc - im[i] * ws;
            const float xi = re[i] * ws + im[i] * wc;
            re[i] = xr;
            im[i] = xi;
        }
    }
    lds_transpose<P>(re, xbuf, lane);
    lds_transpose<P>(im, xbuf, lane);
    if (P == 32) {
        cross_lane_stage<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, false, lane);
    } else if (P == 16) {
        cross_lane_stage<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, (lane & 48) == 48, lane);
        cross_lane_stage<P, SIGN, 16, 32>(re, im, (lane & 16) != 0, false, lane);
    } else {
        cross_lane_stage<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, false, lane);
        cross_lane_stage<P, SIGN, 16, 32>(re, im, (lane & 16) != 0, (lane & 24) == 24, lane);
        cross_lane_stage<P, SIGN, 8, 16>(re, im, (lane & 8) != 0, false, lane);
    }
}

template <int P, int SIGN>
__device__ __forceinline__ void wave_fft(float (&re)[P], float (&im)[P], const float* tw, float* xbuf, int lane) {
    wave_fft_front<P, SIGN>(re, im, tw, xbuf, lane);
    fft_inreg<P, SIGN>(re, im);
}

// ---------------------------------------------------------------------------------------------
// The same transform in the DIT form (fft_inreg_dit: 6 instead of 8 instructions per general butterfly), any P:
//   input : lane l, register brev(j)  holds z[l + 64 j]            (a static renaming for the caller)
//   output: lane l, register i        holds Z[kappa(l) + 64 i]     (natural register order)
// twn: the first-pass twiddle table with its rows in NATURAL register order (entry i of lane l = W_M^{l i}; the kernels
// that use this form permute the global table while copying it to LDS: tw_nat_index).  Two halves as wave_fft_front /
// fft_inreg, so that a caller can start a copy into xbuf between them.
// ---------------------------------------------------------------------------------------------
// index into the global (register-order) table of element i of the natural-order table
template <int P>
__device__ __forceinline__ int tw_nat_index(int i) {
    constexpr int LB = ilog2(P);
    const int l = i / tw_stride<P>(), c = i - l * tw_stride<P>();
    return (c < 2 * P) ? l * tw_stride<P>() + 2 * brev(c >> 1, LB) + (c & 1) : i;
}

template <int P, int SIGN>
__device__ __forceinline__ void wave_fft_dit_front(float (&re)[P], float (&im)[P], const float* twn, float* xbuf, int lane) {
    fft_inreg_dit<P, SIGN>(re, im);
    const float4* trow = reinterpret_cast<const float4*>(twn + lane * tw_stride<P>());
#pragma unroll
    for (int q = 0; q < P / 2; ++q) {
        const float4 w = trow[q];   // twiddles of registers 2q, 2q + 1 (k1 = 2q, 2q + 1)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * q + e;
            const float wc = e ? w.z : w.x;
            const float ws = (SIGN < 0) ? -(e ? w.w : w.y) : (e ? w.w : w.y);
            const float xr = re[i] * wc - im[i] * ws;
            const float xi = re[i] * ws + im[i] * wc;
            re[i] = xr;
            im[i] = xi;
        }
    }
    lds_transpose<P, true>(re, xbuf, lane);
    lds_transpose<P, true>(im, xbuf, lane);
    if (P == 32) {
        cross_lane_stage<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, false, lane);
    } else if (P == 16) {
        cross_lane_stage<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, (lane & 48) == 48, lane);
        cross_lane_stage<P, SIGN, 16, 32>(re, im, (lane & 16) != 0, false, lane);
    } else {
        cross_lane_stage<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, false, lane);
        cross_lane_stage<P, SIGN, 16, 32>(re, im, (lane & 16) != 0, (lane & 24) == 24, lane);
        cross_lane_stage<P, SIGN, 8, 16>(re, im, (lane & 8) != 0, false, lane);
    }
}

// second pass: the P-point transforms over l' in DIT form want register brev(l') <- element l'
template <int P, int SIGN>
__device__ __forceinline__ void wave_fft_dit_back(float (&re)[P], float (&im)[P]) {
    constexpr int LB = ilog2(P);
    float tr[P], ti[P];
#pragma unroll
    for (int r = 0; r < P; ++r) {
        tr[brev(r, LB)] = re[r];
        ti[brev(r, LB)] = im[r];
    }
    fft_inreg_dit<P, SIGN>(tr, ti);
#pragma unroll
    for (int r = 0; r < P; ++r) {
        re[r] = tr[r];
        im[r] = ti[r];
    }
}

template <int P, int SIGN>
__device__ __forceinline__ void wave_fft_dit(float (&re)[P], float (&im)[P], const float* twn, float* xbuf, int lane) {
    wave_fft_dit_front<P, SIGN>(re, im, twn, xbuf, lane);
    wave_fft_dit_back<P, SIGN>(re, im);
}

// ---------------------------------------------------------------------------------------------
// Compact form (P == 32 only) for kernels that need their LDS for something else (k_synth_ola_pair at 12 waves per CU:
// six overlap-add rings): HALF-height transpose buffer (P/2 rows) and HALF twiddle table, the same data flow.
//   * twiddles: the table keeps the even registers only (k1 = brev(i) < 16; rows of P floats + 4 pad); register i + 1
//     holds k1 + 16, its twiddle is the even one times W_M^{16 l} = W_128^l, a per-lane constant (lc, ls): 4 VALU per
//     odd register instead of 8.7 KB of LDS.
//   * transpose in two phases, rows k1 < 16 first: all 64 lanes write their 16 registers of the phase; lane lam reads
//     16 consecutive columns of row lam % 16 -- the 16 x 64 block is read by all 64 lanes, so a lane whose own row
//     (k1 = lam % 32) belongs to the OTHER phase reads the half of its neighbour lam ^ 16 that the neighbour does not
//     read itself (columns 16 b .. 16 b + 15, b = bit 4 of the lane).  After both phases every lane holds its own row's
//     columns [16 b, 16 b + 16) and the other 16 columns of its neighbour's row: one v_permlane16_swap per register
//     pair (odd 16-lane rows of A <-> even rows of B) sorts them out -- 16 VALU per plane.
//   Same output as lds_transpose: register l' holds (k1 = lane % P, l = (lane / P) * P + l').
// ---------------------------------------------------------------------------------------------
template <int P>
__host__ __device__ constexpr int tw_half_stride() { return P + 4; }
template <int P>
__host__ __device__ constexpr int tw_half_floats() { return 64 * tw_half_stride<P>(); }

// The four padding floats of a lane's half-table row: a kernel may keep per-lane constants there and read them where they
// are used (one ds_read_b128) instead of holding them in registers across its frame loop (k_synth_comp_pair at <= 168
// VGPRs: (cos, sin) of the split twiddle and W_128^lane).
template <int P>
__device__ __forceinline__ float4 tw_half_pad(const float* twh, int lane) {
    return *reinterpret_cast<const float4*>(twh + lane * tw_half_stride<P>() + P);
}

// NAT: the input registers are in natural order (register i holds k1 = i: the DIT first pass) instead of bit-reversed.
// Bank layout (round 5).  ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32): with
// every row stored in column order, the rows 8-11 read at column 16 by the lanes 24-27 share their banks with the rows
// 12-15 read at column 0 by the lanes 12-15 of the same group -- 4 extra LDS cycles per read, 64 per transform
// (SQ_LDS_BANK_CONFLICT 3.65 M per launch of k_synth_ola_pair = 64.0 per frame, none with the transform ablated; 7.3 of
// the 8.4 M of the two-transform kernels).  The rows 4-11 therefore store their two 16-column halves SWAPPED (column
// c at c ^ 16): a compile-time choice between two lane offsets on the write side (register i <-> row is static), a
// per-lane constant on the read side; conflict-free by MI355X_MICROARCH.md's group table, model in
// tests/test_fft_dataflow_model.py.
template <int P, bool NAT = false>
__device__ __forceinline__ void lds_transpose_half(float (&x)[P], float* xbuf, int lane) {
    static_assert(P == 32, "the half-height transpose pairs lanes lam, lam ^ 16: P == 32 only");
    constexpr int LB = ilog2(P), HP = P / 2;
    float t[P];
    const int row = lane & (HP - 1);
    const int swz = ((row + 4) >> 3) & 1;   // rows 4-11
    const float4* src = reinterpret_cast<const float4*>(xbuf + row * kXStride + (lane / P) * P +
                                                        (((lane >> (LB - 1)) & 1) ^ swz) * HP);
    const int lane_s = lane ^ HP;           // the swapped rows' column of this lane
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int k1 = NAT ? i : brev(i, LB);
            const int r = k1 & (HP - 1);
            if ((k1 >> (LB - 1)) == h) xbuf[r * kXStride + ((((r + 4) >> 3) & 1) ? lane_s : lane)] = x[i];
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < HP / 4; ++q) {
            const float4 v = src[q];
            t[h * HP + 4 * q + 0] = v.x;
            t[h * HP + 4 * q + 1] = v.y;
            t[h * HP + 4 * q + 2] = v.z;
            t[h * HP + 4 * q + 3] = v.w;
        }
        wave_sync();
    }
#pragma unroll
    for (int j = 0; j < HP; ++j) swap16(t[j], t[HP + j]);
#pragma unroll
    for (int j = 0; j < P; ++j) x[j] = t[j];
}

// wave_fft_front with the half table (twh: tw_half_floats<P>() floats) and the half-height buffer (P/2 * kXStride floats).
// (lc, ls) = (cos, sin)(2 pi lane / 128), the caller's per-lane constant.
template <int P, int SIGN>
__device__ __forceinline__ void wave_fft_front_compact(float (&re)[P], float (&im)[P], const float* twh, float* xbuf,
                                                       int lane, float lc, float ls) {
    static_assert(P == 32, "compact form: P == 32 only");
    fft_inreg<P, SIGN>(re, im);
    const float4* trow = reinterpret_cast<const float4*>(twh + lane * tw_half_stride<P>());
#pragma unroll
    for (int q = 0; q < P / 4; ++q) {
        const float4 w = trow[q];   // twiddles of the even registers 4q, 4q + 2
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 4 * q + 2 * e;
            const float wc = e ? w.z : w.x;
            const float ws0 = e ? w.w : w.y;
            const float wc1 = wc * lc - ws0 * ls, ws1 = wc * ls + ws0 * lc;   // register i + 1: times W_128^lane
            const float ws = (SIGN < 0) ? -ws0 : ws0, wsb = (SIGN < 0) ? -ws1 : ws1;
            const float xr = re[i] * wc - im[i] * ws, xi = re[i] * ws + im[i] * wc;
            const float yr = re[i + 1] * wc1 - im[i + 1] * wsb, yi = re[i + 1] * wsb + im[i + 1] * wc1;
            re[i] = xr;
            im[i] = xi;
            re[i + 1] = yr;
            im[i + 1] = yi;
        }
    }
    lds_transpose_half<P>(re, xbuf, lane);
    lds_transpose_half<P>(im, xbuf, lane);
    cross_lane_stage<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, false, lane);
}

// The whole inverse / forward transform in the DIT form with the compact front (half-height buffer, half twiddle table
// in NATURAL register order: entry e of a lane's row = W_M^{lane e}, e < P/2; register e + P/2 = that times W_128^lane):
//   input : lane l, register brev(j)  holds z[l + 64 j]          (bit-reversed register order; a static renaming for the caller)
//   output: lane l, register i        holds Z[kappa(l) + 64 i]   (natural register order)
// (two halves, so that a caller can start a copy into xbuf between them: the front leaves the exchange buffer idle)
template <int P, int SIGN>
__device__ __forceinline__ void wave_fft_dit_compact_front(float (&re)[P], float (&im)[P], const float* twh, float* xbuf,
                                                           int lane, float lc, float ls) {
    static_assert(P == 32, "compact form: P == 32 only");
    constexpr int HP = P / 2;
    fft_inreg_dit<P, SIGN>(re, im);
    const float4* trow = reinterpret_cast<const float4*>(twh + lane * tw_half_stride<P>());
#pragma unroll
    for (int q = 0; q < HP / 2; ++q) {
        const float4 w = trow[q];   // twiddles of the registers 2q, 2q + 1 (k1 = 2q, 2q + 1 < 16)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * q + e;
            const float wc = e ? w.z : w.x;
            const float ws0 = e ? w.w : w.y;
            const float wc1 = wc * lc - ws0 * ls, ws1 = wc * ls + ws0 * lc;   // register i + 16: times W_128^lane
            const float ws = (SIGN < 0) ? -ws0 : ws0, wsb = (SIGN < 0) ? -ws1 : ws1;
            const float xr = re[i] * wc - im[i] * ws, xi = re[i] * ws + im[i] * wc;
            const float yr = re[i + HP] * wc1 - im[i + HP] * wsb, yi = re[i + HP] * wsb + im[i + HP] * wc1;
            re[i] = xr;
            im[i] = xi;
            re[i + HP] = yr;
            im[i + HP] = yi;
        }
    }
    lds_transpose_half<P, true>(re, xbuf, lane);
    lds_transpose_half<P, true>(im, xbuf, lane);
    cross_lane_stage<P, SIGN, 32, 64>(re, im, (lane & 32) != 0, false, lane);
}

template <int P, int SIGN>
__device__ __forceinline__ void wave_fft_dit_compact(float (&re)[P], float (&im)[P], const float* twh, float* xbuf, int lane,
                                                     float lc, float ls) {
    constexpr int LB = ilog2(P);
    wave_fft_dit_compact_front<P, SIGN>(re, im, twh, xbuf, lane, lc, ls);
    // second pass: the 32-point transforms over l' in DIT form want register brev(l') <- element l'
    float tr[P], ti[P];
#pragma unroll
    for (int r = 0; r < P; ++r) {
        tr[brev(r, LB)] = re[r];
        ti[brev(r, LB)] = im[r];
    }
    fft_inreg_dit<P, SIGN>(tr, ti);
#pragma unroll
    for (int r = 0; r < P; ++r) {
        re[r] = tr[r];
        im[r] = ti[r];
    }
}

}  // namespace mpx
