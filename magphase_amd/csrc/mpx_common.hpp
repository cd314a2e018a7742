// mpx_common.hpp -- shared host/device helpers of libmagphase_hip.so (lossless + compressed translation units).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/magphase_hip.h"
#include "wave_fft.hpp"

namespace mpx {

inline thread_local char g_err[512] = "";

inline int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

#define MPX_HIP_CHECK(expr)                                                       \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) return fail(MPX_ERR_HIP, #expr ": %s", hipGetErrorString(e__)); \
    } while (0)

// Build-time knobs (tools/ab_bench.py builds variants with -D... and times them interleaved in one process).
#ifndef MPX_WAVES_PER_BLOCK
#define MPX_WAVES_PER_BLOCK 8
#endif
constexpr int kWavesPerBlock = MPX_WAVES_PER_BLOCK;  // 8: 512 threads, one block per CU, 2 waves per SIMD; measured best of 8/10/12/16 (tools/ab_bench.py); must not spill: scratch traffic counts in vmcnt
constexpr int kThreads = kWavesPerBlock * 64;

template <int P>
constexpr size_t lds_bytes() {
    return sizeof(float) * (size_t)(tw_floats<P>() + kWavesPerBlock * P * kXStride);
}

#ifndef MPX_ANA_WAVES
#define MPX_ANA_WAVES 12
#endif
// k_analysis: 12 waves = 3 per SIMD (<= 168 VGPRs).  A wave issues at most one instruction per ~5 cycles on this
// chip (tools/archive/clock_probe.hip), so 2 waves per SIMD leave the VALU idle half the time; the third wave only pays once
// every store is a full aligned 256-byte block (tools/ab_bench.py: 8 -> 12 waves = +10 % time with the old store
// shape, -20 % with the aligned one).
constexpr int kAnaWaves = MPX_ANA_WAVES;
constexpr int kAnaThreads = kAnaWaves * 64;
template <int P>
constexpr size_t lds_bytes_ana() {   // twiddles + one transpose buffer per wave + the workgroup's frame queue (4 words)
    return sizeof(float) * (size_t)(tw_floats<P>() + kAnaWaves * P * kXStride + 4);
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Phase markers for the per-phase instruction table of a kernel's ISA (tools/asm_phases.py; builds with -DMPX_PHASE_MARKS
// only: a comment line in the listing, and a compiler barrier for memory operations so that a phase's loads and stores stay
// inside it -- plain arithmetic may still drift across by a few instructions).
#ifdef MPX_PHASE_MARKS
#define MPX_MARK(name) asm volatile("; MPX_MARK " name ::: "memory")
#else
#define MPX_MARK(name) do { } while (0)
#endif
// ... and, in those builds, the values that cross a phase boundary pinned at it (an empty asm that "uses and defines" them):
// the arithmetic of the phase before cannot sink below the marker, that of the phase after cannot rise above it.
template <int K>
__device__ __forceinline__ void mpx_pin(float (&a)[K]) {
#ifdef MPX_PHASE_MARKS
    static_assert(K % 4 == 0, "pinned in groups of four");
#pragma unroll
    for (int i = 0; i < K; i += 4) asm volatile("" : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]));
#else
    (void)a;
#endif
}

// ---------------------------------------------------------------------------------------------
// Frame queue of a workgroup.  A SIMD serves its resident waves by AGE (MI355X_MICROARCH.md, "Two waves per SIMD"): with
// a static frame list per wave (grid-stride) the first-dispatched wave of every SIMD ran 14 us per frame, the last one 21
// (tools/archive/endtime_probe.py), the old waves finished a third earlier and the SIMDs idled through the tail.  The frame-per-
// wave kernels therefore give every workgroup a contiguous range of the batch's frames and let its waves PULL the next
// frame from a counter in LDS (one ds_add_rtn_u32 by lane 0 per frame): fast waves simply take more frames, all waves
// of a workgroup finish together.  Frames are independent, so the order changes nothing in the output.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ long long queue_pull(unsigned* ctr, long long begin) {
    unsigned v = 0u;
    if ((threadIdx.x & 63) == 0) v = atomicAdd(ctr, 1u);
    return begin + (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void block_frame_range(long long nframes, long long& fb, long long& fe) {
    fb = (long long)blockIdx.x * nframes / (long long)gridDim.x;
    fe = (long long)(blockIdx.x + 1) * nframes / (long long)gridDim.x;
}

// ---------------------------------------------------------------------------------------------
// analysis
// ---------------------------------------------------------------------------------------------
// sin(pi t / 2)^2 = 0.5 + 0.5 sin(pi (t - 0.5)) for t in [0,1]: one odd degree-9 polynomial for sin(pi u) on
// u in [-0.5, 0.5] (least squares on Chebyshev nodes, |err| < 4e-9; 1.2e-7 after fp32 evaluation), no range
// reduction, no branches: 8 VALU per window sample.
__device__ __forceinline__ float sin2_halfpi(float t) {
    const float u = t - 0.5f;
    const float u2 = u * u;
    float p = fmaf(u2, 7.721838616e-02f, -5.980441939e-01f);
    p = fmaf(u2, p, 2.550031194e+00f);
    p = fmaf(u2, p, -5.167706866e+00f);
    p = fmaf(u2, p, 3.141592580e+00f);
    return fmaf(0.5f * u, p, 0.5f);
}

// Half windows of libaudio.py:70-84 evaluated analytically.  t = k/L on the rising half (k <= L; t = 1 when L == 0:
// win(1) == [1.]) and t = (L+R-k)/R on the falling half:
//   wtype 0: np.hanning(1+2L)[k]      = sin^2(pi t / 2)
//   wtype 1: np.bartlett(1+2L)[k]**2.5 = t^2.5      (voiced noise window, magphase.py:67-68, Q11)
__device__ __forceinline__ float half_window(int k, int L, int LR, int kadd, float invL, float invR, int wtype) {
    const bool rising = k <= L;
    const int num = rising ? k + kadd : LR - k;
    const float inv = rising ? invL : invR;
    const float t = (float)num * inv;
    // t = k / L in [0, 1], never a denormal: the hardware square root as it is (sqrtf adds a rescaling sequence)
    return (wtype == 0) ? sin2_halfpi(t) : t * t * __builtin_amdgcn_sqrtf(t);
}

__device__ __forceinline__ float hann_half(int k, int L, int LR, int kadd, float invL, float invR) {
    return half_window(k, L, LR, kadd, invL, invR, 0);
}

struct FrameGeom {
    const float* base;  // &sig[pos - L]: sample k of the windowed frame is base[k]
    int L, LR, len, rot, kadd;
    float invL, invR;
};

__device__ __forceinline__ FrameGeom frame_geom(const float* __restrict__ sig, long long pos, int L, int R, int N) {
    FrameGeom g;
    g.L = L;
    g.LR = L + R;
    g.len = min(g.LR + 1, N);            // Q19: frames longer than N are truncated
    g.rot = (L < N) ? L : 0;             // python slicing: rotation by >= N is the identity
    g.kadd = (L == 0) ? 1 : 0;           // L == 0: the single rising sample has weight np.hanning(1) == 1
    g.invL = (L > 0) ? 1.0f / (float)L : 1.0f;
    g.invR = (R > 0) ? 1.0f / (float)R : 0.0f;
    g.base = sig + (pos - L);
    return g;
}

// Asynchronous HBM -> LDS copy of the frame's samples [tile0, tile0 + tile_len) in sample order (clamped reads;
// samples >= len are masked by the window later): one global_load_lds_dword per 64 samples, no VGPR destination.
// Issued through inline asm on purpose: the compiler does not track these copies, so it cannot pessimise them into
// vmcnt(0) waits (which on gfx9 -- one in-order counter for loads AND stores -- would drain the 99 stores of the
// previous frame); the caller waits with staged_wait<N>() instead.  LDS address = M0 + 4*lane (wave-uniform base).
__device__ __forceinline__ void stage_samples_async(const FrameGeom& g, int tile0, int tile_len, unsigned lds_byte,
                                                    int lane) {
    const int nrow = (min(g.len - tile0, tile_len) + 63) >> 6;
    for (int c = 0; c < nrow; ++c) {
        const float* src = g.base + min(tile0 + 64 * c + lane, g.len - 1);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_byte + 256u * (unsigned)c);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(m0v) : "m0", "memory");
    }
}

// Waits until at most N vector-memory operations of this wave are outstanding (gfx9: ONE in-order counter for loads and
// stores, 6 bits).  After a staged copy, N = the number of VMEM operations issued SINCE the copy's last load: the copy
// has then landed.  N >= 63 saturates the counter field (63 is the most that can be asked for).
template <int N>
__device__ __forceinline__ void staged_wait() {
    static_assert(N >= 0, "staged_wait: negative count");
    constexpr int kCnt = N > 63 ? 63 : N;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kCnt) : "memory");
}

// Hermitian merge Z[k] = E[k] + i O[k] of the half spectrum held as lane l, register j <-> bin l + 64 j
// (values pre-scaled by 0.5/M, fftshift sign folded in), in place, two bins (j, P-1-j) per step so that no
// second register array is live.  Partner bin M-k: lane (64-lane)&63, register P-1-j; lane 0 pairs (j, P-j)
// instead, served from a one-bin stash of the original register P-j (overwritten one step earlier).
// xm = Nyquist bin (real), only meaningful on lane 0.  (wl_c, wl_s) = e^{+2 pi i lane/N}.
template <int P>
__device__ __forceinline__ void hermitian_merge(float (&xr)[P], float (&xi)[P], float xm, int lane, float wl_c,
                                                float wl_s) {
    const int src_lane = (64 - lane) & 63;
    const bool lane0 = (lane == 0);
    // 1) fetch every partner bin first (64 ds_bpermute in flight together: one LDS latency instead of 16);
    //    partner of bin (lane, j) = bin M-k: lane (64-lane)&63, register P-1-j; on lane 0: own register (P-j)%P, and
    //    the Nyquist bin for j == 0.
    float pr[P], pi[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        pr[j] = __shfl(xr[P - 1 - j], src_lane);
        pi[j] = __shfl(xi[P - 1 - j], src_lane);
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const float l0r = (j == 0) ? xm : xr[(P - j) % P];
        const float l0i = (j == 0) ? 0.0f : xi[(P - j) % P];
        pr[j] = lane0 ? l0r : pr[j];
        pi[j] = lane0 ? l0i : pi[j];
    }
    // 2) E = X + conj(Xp), T = X - conj(Xp), O = conj(W_N^k) T, Z = E + i O   (conj(W_N^k) = e^{+2 pi i (lane/N + j/2P)})
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const float er = xr[j] + pr[j], ei = xi[j] - pi[j], tr = xr[j] - pr[j], ti = xi[j] + pi[j];
        const float cq = cos2p<P>(j), sq = sin2p<P>(j);
        const float wr = wl_c * cq - wl_s * sq, wi = wl_c * sq + wl_s * cq;
        xr[j] = er - (wr * ti + wi * tr);
        xi[j] = ei + (wr * tr - wi * ti);
    }
}

// Loads one frame's lossless features (bins lane + 64 j, plus the Nyquist bin on lane 0) and turns them into the
// scaled unit-phase spectrum X = mag (R + jI)/|R + jI| (0 where |R + jI| == 0), magphase.py:1761-1766, with
// DC/Nyquist imaginary parts dropped (Q5) and the (-1)^k fftshift sign and the 0.5/M scale folded in.
template <int P>
struct FrameFeat {
    float m[P], a[P], b[P];
    float mM, aM, bM;
};

template <int P>
__device__ __forceinline__ void feat_load(FrameFeat<P>& ff, const float* __restrict__ mrow,
                                          const float* __restrict__ rrow, const float* __restrict__ irow, int lane) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int k = lane + 64 * j;
        ff.m[j] = mrow[k];
        ff.a[j] = rrow[k];
        ff.b[j] = irow[k];
    }
    ff.mM = ff.aM = ff.bM = 0.0f;
    if (lane == 0) {
        ff.mM = mrow[64 * P];
        ff.aM = rrow[64 * P];
        ff.bM = irow[64 * P];
    }
}

template <int P>
__device__ __forceinline__ void feat_convert(const FrameFeat<P>& ff, float (&xr)[P], float (&xi)[P], float& xm,
                                             int lane) {
    constexpr int M = 64 * P;
    const float sgn_scale = ((lane & 1) ? -1.0f : 1.0f) * (0.5f / (float)M);
#pragma unroll
    for (int j = 0; j < P; ++j) {
        // |R + jI| == 0 -> 0 (magphase.py:1764-1766): rsq(max(s, tiny)) is finite and multiplies a == b == 0
        const float s = ff.a[j] * ff.a[j] + ff.b[j] * ff.b[j];
        const float g = ff.m[j] * sgn_scale * __builtin_amdgcn_rsqf(fmaxf(s, 1.0e-37f));
        xr[j] = ff.a[j] * g;
        xi[j] = ff.b[j] * g;
    }
    xm = 0.0f;
    if (lane == 0) {
        xi[0] = 0.0f;
        const float s = ff.aM * ff.aM + ff.bM * ff.bM;
        xm = (s > 0.0f) ? ff.mM * (0.5f / (float)M) * ff.aM * __builtin_amdgcn_rsqf(s) : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// Paired form of feat_load / feat_convert / hermitian_merge (k_synth_ola_pair).  The Hermitian merge works on bin
// pairs (k, M - k): Z[k] = E + iO and Z[M - k] = conj(E) + i conj(O) come from ONE E / O evaluation.  So lane l loads
// its bins k = l + 64 j for j < P/2 only, plus their mirrors M - k straight from memory (descending addresses: still
// one contiguous 256-byte block per instruction), computes both outputs, keeps Z[k] in register j and hands Z[M - k]
// to the lane that owns it: bin M - k = (64 - l) + 64 (P-1-j) is register P-1-j of lane 64 - l (lane 0: its own
// register P - j; bin M, the Nyquist bin, is the mirror of bin 0 and has no register; bin M/2 = register P/2 of lane 0
// is its own mirror: Z = 2 conj(X), one extra single-lane load per stream).
// Against the per-bin form: half the merge arithmetic (16 instead of 32 E/O evaluations per lane), P instead of 2P
// lane exchanges, the same 3P + 3 loads.
// ---------------------------------------------------------------------------------------------
template <int P>
struct PairFeat {
    float m[P / 2], a[P / 2], b[P / 2];      // own bins k = lane + 64 j
    float mq[P / 2], aq[P / 2], bq[P / 2];   // mirrors M - k
    float mH, aH, bH;                        // bin M/2 + lane (used on lane 0: bin M/2)
};

template <int P>
__device__ __forceinline__ void feat_load_paired(PairFeat<P>& ff, const float* __restrict__ mrow,
                                                 const float* __restrict__ rrow, const float* __restrict__ irow,
                                                 int lane) {
    constexpr int M = 64 * P;
    const float* mlo = mrow + lane;
    const float* rlo = rrow + lane;
    const float* ilo = irow + lane;
    const float* mhi = mrow + (M - lane);
    const float* rhi = rrow + (M - lane);
    const float* ihi = irow + (M - lane);
#ifdef MPX_FEAT_NT
#define MPX_LD(p) __builtin_nontemporal_load(p)
#else
#define MPX_LD(p) (*(p))
#endif
#pragma unroll
    for (int j = 0; j < P / 2; ++j) {
        ff.m[j] = MPX_LD(mlo + 64 * j);
        ff.a[j] = MPX_LD(rlo + 64 * j);
        ff.b[j] = MPX_LD(ilo + 64 * j);
        ff.mq[j] = MPX_LD(mhi - 64 * j);
        ff.aq[j] = MPX_LD(rhi - 64 * j);
        ff.bq[j] = MPX_LD(ihi - 64 * j);
    }
    // bin M/2 is needed on lane 0 only; every lane loads "its" bin M/2 + lane instead (in range, one more coalesced
    // vector load per stream): a lane-0-only load has a wave-uniform address, which the compiler turns into a SCALAR
    // load followed by an immediate lgkmcnt(0) wait -- a full memory round trip in the middle of the prefetch.
    ff.mH = mlo[M / 2];
    ff.aH = rlo[M / 2];
    ff.bH = ilo[M / 2];
}

// The same loads for the bin pairs J0 <= j < J1 only (+ bin M/2 + lane if WITH_H): k_synth_ola_pair issues a frame's
// prefetch in parts, as registers come free.
template <int P, int J0, int J1, bool WITH_H>
__device__ __forceinline__ void feat_load_paired_part(PairFeat<P>& ff, const float* __restrict__ mrow,
                                                      const float* __restrict__ rrow, const float* __restrict__ irow,
                                                      int lane) {
    constexpr int M = 64 * P;
    const float* mlo = mrow + lane;
    const float* rlo = rrow + lane;
    const float* ilo = irow + lane;
    const float* mhi = mrow + (M - lane);
    const float* rhi = rrow + (M - lane);
    const float* ihi = irow + (M - lane);
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        ff.m[j] = mlo[64 * j];
        ff.a[j] = rlo[64 * j];
        ff.b[j] = ilo[64 * j];
        ff.mq[j] = mhi[-64 * j];
        ff.aq[j] = rhi[-64 * j];
        ff.bq[j] = ihi[-64 * j];
    }
    if (WITH_H) {   // every lane loads "its" bin M/2 + lane (see feat_load_paired)
        ff.mH = mlo[M / 2];
        ff.aH = rlo[M / 2];
        ff.bH = ilo[M / 2];
    }
}

template <int P>
__device__ __forceinline__ void feat_merge_paired(const PairFeat<P>& ff, float (&xr)[P], float (&xi)[P], int lane,
                                                  float wl_c, float wl_s) {
    constexpr int M = 64 * P, HP = P / 2;
    const float sgn_scale = ((lane & 1) ? -1.0f : 1.0f) * (0.5f / (float)M);   // (-1)^k fftshift sign, IFFT scale
    const bool lane0 = (lane == 0);
    float zr[HP], zi[HP];   // Z[M - k]
#pragma unroll
    for (int j = 0; j < HP; ++j) {
        // X = mag (R + jI) / |R + jI|, 0 where |R + jI| == 0 (magphase.py:1761-1766)
        const float s = ff.a[j] * ff.a[j] + ff.b[j] * ff.b[j];
        const float g = ff.m[j] * sgn_scale * __builtin_amdgcn_rsqf(fmaxf(s, 1.0e-37f));
        const float sq_ = ff.aq[j] * ff.aq[j] + ff.bq[j] * ff.bq[j];
        const float gq = ff.mq[j] * sgn_scale * __builtin_amdgcn_rsqf(fmaxf(sq_, 1.0e-37f));
        const float x_r = ff.a[j] * g, p_r = ff.aq[j] * gq;
        float x_i = ff.b[j] * g, p_i = ff.bq[j] * gq;
        if (j == 0) {   // DC and Nyquist: imaginary parts dropped (Q5)
            x_i = lane0 ? 0.0f : x_i;
            p_i = lane0 ? 0.0f : p_i;
        }
        // E = X + conj(Xp), T = X - conj(Xp), O = conj(W_N^k) T   (conj(W_N^k) = e^{+2 pi i (lane/N + j/2P)})
        const float er = x_r + p_r, ei = x_i - p_i, tr = x_r - p_r, ti = x_i + p_i;
        const float cq = cos2p<P>(j), sq = sin2p<P>(j);
        const float wr = wl_c * cq - wl_s * sq, wi = wl_c * sq + wl_s * cq;
        const float orr = wr * tr - wi * ti, oi = wr * ti + wi * tr;
        xr[j] = er - oi;
        xi[j] = ei + orr;
        zr[j] = er + oi;
        zi[j] = orr - ei;
    }
    // bin M/2 (lane 0): Z = 2 conj(X); M/2 is even and so is lane 0: sgn_scale is +0.5/M there
    const float sH = ff.aH * ff.aH + ff.bH * ff.bH;
    const float gH = 2.0f * ff.mH * sgn_scale * __builtin_amdgcn_rsqf(fmaxf(sH, 1.0e-37f));
    const float hr = ff.aH * gH, hi = -ff.bH * gH;
    // hand-over: register r >= P/2 of lane l' comes from lane (64 - l') & 63, which publishes Z[M - k] of its
    // j = P-1-r; lane 0 serves itself: register P/2 = bin M/2, register r > P/2 = mirror of its j = P - r.
    const int src_lane = (64 - lane) & 63;
#pragma unroll
    for (int r = HP; r < P; ++r) {
        const float pr = lane0 ? ((r == HP) ? hr : zr[P - r]) : zr[P - 1 - r];
        const float pi = lane0 ? ((r == HP) ? hi : zi[P - r]) : zi[P - 1 - r];
        xr[r] = __shfl(pr, src_lane);
        xi[r] = __shfl(pi, src_lane);
    }
}

// feat_merge_paired's second half for spectra that are already complex (k_synth_comp_pair): xo = X[k] of the own bins
// k = lane + 64 j, xm = X[M - k] of their mirrors (j < P/2), xh = X[M/2] (lane 0), all including the (-1)^k / 2M factor.
template <int P>
__device__ __forceinline__ void merge_paired_complex(const float (&xo_r)[P / 2], const float (&xo_i)[P / 2],
                                                     const float (&xm_r)[P / 2], const float (&xm_i)[P / 2], float xh_r,
                                                     float xh_i, float (&xr)[P], float (&xi)[P], int lane, float wl_c,
                                                     float wl_s) {
    constexpr int HP = P / 2;
    const bool lane0 = (lane == 0);
    float zr[HP], zi[HP];   // Z[M - k]
#pragma unroll
    for (int j = 0; j < HP; ++j) {
        // E = X + conj(Xp), T = X - conj(Xp), O = conj(W_N^k) T   (conj(W_N^k) = e^{+2 pi i (lane/N + j/2P)})
        const float er = xo_r[j] + xm_r[j], ei = xo_i[j] - xm_i[j], tr = xo_r[j] - xm_r[j], ti = xo_i[j] + xm_i[j];
        const float cq = cos2p<P>(j), sq = sin2p<P>(j);
        const float wr = wl_c * cq - wl_s * sq, wi = wl_c * sq + wl_s * cq;
        const float orr = wr * tr - wi * ti, oi = wr * ti + wi * tr;
        xr[j] = er - oi;
        xi[j] = ei + orr;
        zr[j] = er + oi;
        zi[j] = orr - ei;
    }
    const float hr = 2.0f * xh_r, hi = -2.0f * xh_i;   // bin M/2 (lane 0): Z = 2 conj(X)
    const int src_lane = (64 - lane) & 63;
#pragma unroll
    for (int r = HP; r < P; ++r) {
        const float pr = lane0 ? ((r == HP) ? hr : zr[P - r]) : zr[P - 1 - r];
        const float pi = lane0 ? ((r == HP) ? hi : zi[P - r]) : zi[P - 1 - r];
        xr[r] = __shfl(pr, src_lane);
        xi[r] = __shfl(pi, src_lane);
    }
}

// Ring of R strip elements, stored as two halves: even strip positions b in ringE[b/2 mod R/2], odd ones in
// ringO.  A lane's two samples (2m, 2m+1) of a frame then hit ringE/ringO[c + m] with m consecutive across
// lanes: conflict-free 4-byte accesses whatever the parity of the frame position (ring_add below).
template <int P>
constexpr int ring_len() { return 128 * P + 128; }

// One RUN of consecutive frames of one utterance, overlap-added by one wave pair in an LDS ring (host planner:
// hostmath.ola_runs; C ABI: mpx_ola_run).  Coordinates e are "strip elements": e = OLA-buffer position - x0.
//   e <  head_end           : positions the previous run's last frames reach too -> stored in this run's head strip,
//                             added to the output by k_ola_fixup (previous run's sum first: fixed order)
//   out_lo <= e < out_hi    : final for this run (its frames are the only or the first contributors) -> written
//                             straight to pcm_out[out_base + e]   (out_base is a multiple of 64: aligned 256-byte blocks)
//   everything else         : outside the kept part of the reference's OLA buffer (magphase.py:59-61), or owned by the
//                             next run -> dropped
struct RunDesc {
    int frame_begin, frame_end;   // global frame indices (rows of mag/real/imag, entries of pm_rel)
    int x0;                       // OLA-buffer position of element 0
    int head_end;
    int out_lo, out_hi;
    int flush_end;                // the ring is streamed out and cleared up to here when the run ends
    int fix_lo, fix_hi;           // k_ola_fixup: pcm_out[out_base + e] += head strip[e] for fix_lo <= e < fix_hi
    int pad;
    long long out_base;           // pcm_out index of element 0
    long long strip_off;          // float offset of the head strip in `strips`
};
static_assert(sizeof(RunDesc) == 56, "RunDesc must match mpx_ola_run");

// ---------------------------------------------------------------------------------------------
// Shared front of the pair kernels (k_synth_comp_pair, k_roundtrip_pair): LDS tables and the cursor over a wave's frames.
// ---------------------------------------------------------------------------------------------
// First-pass twiddle table into LDS -- COMPACT: the half table (wave_fft.hpp: the even registers' twiddles, or with DIT the
// first P/2 in natural register order) with this lane's constants in the pad of its row (tw_half_pad: W_N^{-lane} and
// W_128^lane as (cos, sin)) -- then the pairs' rings and tickets cleared, one __syncthreads().
template <int P, bool COMPACT, bool DIT>
__device__ __forceinline__ void pair_kernel_prologue(float* tw, const float* __restrict__ tw_g, float* rings, int ring_floats,
                                                     int* tickets, int n_tickets, int n_threads) {
    constexpr int N = 128 * P;
    if constexpr (COMPACT) {
        for (int i = threadIdx.x; i < tw_half_floats<P>(); i += n_threads) {
            const int l = i / tw_half_stride<P>(), c = i - l * tw_half_stride<P>();
            float v = 0.0f;
            if (c < P) {
                v = DIT ? tw_g[l * tw_stride<P>() + 2 * brev(c >> 1, ilog2(P)) + (c & 1)]
                        : tw_g[l * tw_stride<P>() + 4 * (c >> 1) + (c & 1)];
            } else {
                float sn, cs;
                if (c < P + 2) sincospif(-2.0f * (float)l / (float)N, &sn, &cs);
                else sincospif((float)l / 64.0f, &sn, &cs);
                v = ((c - P) & 1) ? sn : cs;
            }
            tw[i] = v;
        }
    } else {
        for (int i = threadIdx.x; i < tw_floats<P>(); i += n_threads) tw[i] = tw_g[i];
    }
    for (int i = threadIdx.x; i < ring_floats; i += n_threads) rings[i] = 0.0f;
    if ((int)threadIdx.x < n_tickets) tickets[threadIdx.x] = 0;
    __syncthreads();
}

// Cursor over this wave's frames: every second frame (the wave's half) of every run of the pair's work list; ticket_base
// counts the frames of the runs already passed (the order of the overlap-adds within the pair).
struct PairCursor {
    int wi, fi, ci, ticket_base, fb, fe, x0, valid;
};
__device__ __forceinline__ void pair_cursor_settle(PairCursor& c, int wi_end, int half, const RunDesc* __restrict__ runs,
                                                   const int* __restrict__ slot_runs) {
    while (c.wi < wi_end) {
        c.ci = slot_runs[c.wi];
        c.fb = runs[c.ci].frame_begin;
        c.fe = runs[c.ci].frame_end;
        c.x0 = runs[c.ci].x0;
        c.fi = c.fb + half;
        if (c.fi < c.fe) {
            c.valid = 1;
            return;
        }
        c.ticket_base += c.fe - c.fb;
        ++c.wi;
    }
    c.valid = 0;
}
__device__ __forceinline__ void pair_cursor_advance(PairCursor& c, int wi_end, int half, const RunDesc* __restrict__ runs,
                                                    const int* __restrict__ slot_runs) {
    c.fi += 2;
    if (c.fi >= c.fe) {
        c.ticket_base += c.fe - c.fb;
        ++c.wi;
        pair_cursor_settle(c, wi_end, half, runs, slot_runs);
    }
}

// Streams elements [from, to) out of the ring (head strip / pcm_out / nowhere, see RunDesc) and clears their slots.
// from is a multiple of 64.  One step = 256 elements, FOUR consecutive elements b .. b+3 per lane (b = b0 + 4 lane):
// elements 2j / 2j+1 live in the even / odd half of the ring at index j, so a lane's quad is one 8-byte read from each
// half (index b/2 is even and RH is even: aligned, and a pair never straddles the wrap), interleaved into ONE 16-byte
// global store (1 KB per wave instruction; out_base is a multiple of 64 elements, so the stores are aligned) and cleared
// by two 8-byte LDS writes -- 5 memory instructions per 256 elements instead of 12 with one element per lane.  Blocks
// that touch a region boundary (head_end / out_lo / out_hi: at most a handful per run) take the per-element path.
// This runs inside the ordered section of the ring, where every cycle is serial.
template <int R>
__device__ __forceinline__ void flush_ring(float* ring, float* __restrict__ strip, float* __restrict__ pcm0,
                                           int head_end, int out_lo, int out_hi, int from, int to, int lane) {
    constexpr int RH = R / 2;
    int j = ((from >> 1) % RH) + 2 * lane;      // index of the lane's first pair in each half
    j = (j >= RH) ? j - RH : j;
    for (int b0 = from; b0 < to; b0 += 256) {
        const int b = b0 + 4 * lane;
        const bool act = b < to;
        float2 ev = make_float2(0.0f, 0.0f), od = ev;
        if (act) {
            ev = *reinterpret_cast<const float2*>(ring + j);
            od = *reinterpret_cast<const float2*>(ring + RH + j);
        }
        const int bend = (min(b0 + 256, to) + 3) & ~3;   // end of the last active quad (to need not be a multiple of 4)
        const bool whole_out = (b0 >= out_lo) && (bend <= out_hi) && (b0 >= head_end);   // wave-uniform
        if (whole_out) {
            if (act) *reinterpret_cast<float4*>(pcm0 + b) = make_float4(ev.x, od.x, ev.y, od.y);
        } else if (act) {
            const float v[4] = {ev.x, od.x, ev.y, od.y};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (b + e < head_end) strip[b + e] = v[e];
                if (b + e >= out_lo && b + e < out_hi) pcm0[b + e] = v[e];
            }
        }
        if (act) {
            *reinterpret_cast<float2*>(ring + j) = make_float2(0.0f, 0.0f);
            *reinterpret_cast<float2*>(ring + RH + j) = make_float2(0.0f, 0.0f);
        }
        j += 128;
        j = (j >= RH) ? j - RH : j;
    }
}

// Overlap-add of one frame held as lane l, register i <-> samples 2n, 2n + 1 with n = kappa(l) + 64 brev(i) (the
// inverse wave FFT's output) at strip position x into the ring.  Even strip positions live in ring[0, RH), odd ones in
// ring[RH, 2RH) (index b/2 mod RH): a lane's two samples hit the two halves at consecutive indices across lanes --
// conflict-free 4-byte accesses whatever the parity of x.  Element i of a plane sits at index (c + 64 q) mod RH,
// q = brev(i), c = c_base + kappa: 64 q goes into the instruction's immediate offset, the wrap (-RH once the index
// passes the end; the wrap point differs by at most one q between lanes) is a per-lane bit mask: bit q set <=>
// wrapped, one v_bfe_i32 + v_bfi_b32 per access instead of compare / select pairs (and their hazard nops).
// All reads of a plane first, then the adds, then the writes: one LDS latency per plane.  combine(old, value, n): new ring value
// for sample n of the frame (plain sum, or windowed sum); live(q): false for register rows that add nothing.
// (LDS float atomics -- ds_add_f32 -- measured ~190 LDS cycles per wave instruction on gfx950: plain read/add/write.)
template <int P, typename CFn, typename LFn>
__device__ __forceinline__ void ring_add(float* smem_base, unsigned ring_byte, int x, const float (&xr)[P],
                                         const float (&xi)[P], int lane, CFn combine, LFn live) {
    constexpr int LB = ilog2(P), R = ring_len<P>(), RH = R / 2;
    const int kap = kappa<P>(lane);
    const int odd = x & 1;
    const int c0 = ((x >> 1) % RH) + kap;          // plane 0 (samples 2n): half `odd`
    const int c1 = (((x + 1) >> 1) % RH) + kap;    // plane 1 (samples 2n + 1): the other half
    const unsigned a0 = ring_byte + 4u * (unsigned)((odd ? RH : 0) + c0);
    const unsigned a1 = ring_byte + 4u * (unsigned)((odd ? 0 : RH) + c1);
    const unsigned b0 = a0 - 4u * RH, b1 = a1 - 4u * RH;
    const int w0 = (RH - c0 + 63) >> 6;            // first q with c0 + 64 q >= RH (may be > P - 1: never wraps)
    const int w1 = (RH - c1 + 63) >> 6;
    const unsigned m0 = (w0 >= 32) ? 0u : (~0u << w0);
    const unsigned m1 = (w1 >= 32) ? 0u : (~0u << w1);
    char* base = reinterpret_cast<char*>(smem_base);
    auto at = [&](unsigned a, unsigned b, unsigned m, int q) -> float* {
        const unsigned sel = (unsigned)__builtin_amdgcn_sbfe((int)m, q, 1);   // 0 or ~0
        return reinterpret_cast<float*>(base + ((sel & b) | (~sel & a)) + 256 * q);
    };
    // plane by plane: P ring values in registers at a time (two LDS round trips per frame; all 2P at once cost P more
    // registers, which the feature prefetch of the next frame needs)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const unsigned a = pl ? a1 : a0, b = pl ? b1 : b0, m = pl ? m1 : m0;
        float o[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int q = brev(i, LB);
            o[i] = 0.0f;
            if (live(q)) o[i] = *at(a, b, m, q);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int q = brev(i, LB);
            if (live(q)) *at(a, b, m, q) = combine(o[i], pl ? xi[i] : xr[i], 2 * (kap + 64 * q) + pl);
        }
    }
}

// ring_add one plane at a time, CH ring values in registers at a time (k_synth_ola_pair at 3 waves per SIMD: the caller
// places the next part of its feature prefetch between the planes).  PL 0: samples 2n (from xv = the real parts),
// PL 1: samples 2n + 1 (the imaginary parts).
struct RingAddr {
    unsigned a0, a1, b0, b1, m0, m1;
    int cu0, cu1;   // the wave-uniform parts of the two planes' start indices (without the lane's kappa)
};

template <int P>
__device__ __forceinline__ RingAddr ring_addr(unsigned ring_byte, int x, int lane) {
    constexpr int R = ring_len<P>(), RH = R / 2;
    const int kap = kappa<P>(lane);
    const int odd = x & 1;
    const int c0 = ((x >> 1) % RH) + kap;          // plane 0 (samples 2n): half `odd`
    const int c1 = (((x + 1) >> 1) % RH) + kap;    // plane 1 (samples 2n + 1): the other half
    RingAddr r;
    r.a0 = ring_byte + 4u * (unsigned)((odd ? RH : 0) + c0);
    r.a1 = ring_byte + 4u * (unsigned)((odd ? 0 : RH) + c1);
    r.b0 = r.a0 - 4u * RH;
    r.b1 = r.a1 - 4u * RH;
    const int w0 = (RH - c0 + 63) >> 6;            // first q with c0 + 64 q >= RH (may be > P - 1: never wraps)
    const int w1 = (RH - c1 + 63) >> 6;
    r.m0 = (w0 >= 32) ? 0u : (~0u << w0);
    r.m1 = (w1 >= 32) ? 0u : (~0u << w1);
    r.cu0 = (x >> 1) % RH;
    r.cu1 = ((x + 1) >> 1) % RH;
    return r;
}

// LDS pointer from a byte address (address space 3: no generic-pointer arithmetic, the immediate offset folds)
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ lds_float* lds_ptr(unsigned byte_addr) {
    return reinterpret_cast<lds_float*>(static_cast<uintptr_t>(byte_addr));
}

// NAT: register i holds row q = i (the DIT transform's natural output order) instead of q = brev(i).
// ROT: the rows are taken half a transform apart -- row q of the frame = row (q + P/2) mod P of xv: the fftshift of the
// inverse transform's output (a spectrum multiplied by (-1)^k) as a renaming of registers instead of a multiplication.
template <int P, int PL, int CH, bool NAT = false, bool ROT = false, typename CFn, typename LFn>
__device__ __forceinline__ void ring_add_plane(float* smem_base, const RingAddr& ra, const float (&xv)[P], int lane,
                                               CFn combine, LFn live) {
    constexpr int LB = ilog2(P);
    constexpr int XR = ROT ? (NAT ? P / 2 : 1) : 0;   // register of row q + P/2: i ^ (P/2) in natural order, i ^ 1 bit-reversed
    static_assert(P % CH == 0, "chunk size must divide P");
    const int kap = kappa<P>(lane);
    const unsigned a = PL ? ra.a1 : ra.a0, b = PL ? ra.b1 : ra.b0, m = PL ? ra.m1 : ra.m0;
#ifndef MPX_RING_ADDR_MODE
#define MPX_RING_ADDR_MODE 2
#endif
    if constexpr (P == 32 && MPX_RING_ADDR_MODE != 0) {
        // kappa(lane) == lane: row q wraps for the lanes >= thr_q = RH - c - 64 q, a WAVE-UNIFORM threshold.  The lane
        // mask of every row is therefore formed on the scalar unit and the address is ONE v_cndmask per element (unwrapped
        // / wrapped base + the row's immediate offset).  The per-lane bit-mask form below compiled to v_and + v_cmp +
        // v_cndmask + v_add per element: 256 of the kernel's 2 350 VALU instructions per frame were ring addresses, and
        // the kernel is bound by VALU issue (tools/archive/endtime_probe.py: 2.8 cycles per instruction and SIMD at 3 waves).
        constexpr int RH = ring_len<P>() / 2;
        const int cu = PL ? ra.cu1 : ra.cu0;
#ifndef MPX_RING_ADDR_MODE
#define MPX_RING_ADDR_MODE 2
#endif
#if MPX_RING_ADDR_MODE == 1   // lane mask per row on the scalar unit (8 SALU) + one v_cndmask
        auto addr = [&](int q) -> unsigned {
            const int thr = RH - cu - 64 * q;
            const unsigned long long mask = (thr <= 0) ? ~0ull : ((thr >= 64) ? 0ull : (~0ull << thr));
            unsigned r;
            asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
            return r + 256u * (unsigned)q;
        };
#else                         // one v_cmp against a literal + one v_cndmask: lane + 64 q >= RH - c  <=>  d >= -64 q
        const int d = kap - (RH - cu);
        auto addr = [&](int q) -> unsigned { return ((d >= -64 * q) ? b : a) + 256u * (unsigned)q; };
#endif
#pragma unroll
        for (int c = 0; c < P; c += CH) {
            float o[CH];
            unsigned ad[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int q = NAT ? (c + i) : brev(c + i, LB);
                o[i] = 0.0f;
                ad[i] = addr(q);
                if (live(q)) o[i] = *lds_ptr(ad[i]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int q = NAT ? (c + i) : brev(c + i, LB);
                if (live(q)) *lds_ptr(ad[i]) = combine(o[i], xv[(c + i) ^ XR], 2 * (kap + 64 * q) + PL);
            }
        }
        (void)smem_base;
        (void)m;
        return;
    }
    char* base = reinterpret_cast<char*>(smem_base);
    auto at = [&](int q) -> float* {
        const unsigned sel = (unsigned)__builtin_amdgcn_sbfe((int)m, q, 1);   // 0 or ~0
        return reinterpret_cast<float*>(base + ((sel & b) | (~sel & a)) + 256 * q);
    };
#pragma unroll
    for (int c = 0; c < P; c += CH) {
        float o[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int q = NAT ? (c + i) : brev(c + i, LB);
            o[i] = 0.0f;
            if (live(q)) o[i] = *at(q);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int q = NAT ? (c + i) : brev(c + i, LB);
            if (live(q)) *at(q) = combine(o[i], xv[(c + i) ^ XR], 2 * (kap + 64 * q) + PL);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
inline int p_of(int fft_len) { return fft_len == 4096 ? 32 : (fft_len == 2048 ? 16 : (fft_len == 1024 ? 8 : 0)); }

// Compute units of the current device (cached per device: hipGetDeviceProperties costs tens of microseconds, and this
// is asked on every launch).
inline int device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        hipDeviceProp_t prop;
        cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return cached[dev];
}

inline int grid_for(long long nframes, int waves_per_block = kWavesPerBlock) {
    long long need = (nframes + waves_per_block - 1) / waves_per_block;
    return (int)std::max<long long>(1, std::min<long long>(need, device_cus()));
}

// Raises a kernel's dynamic LDS limit; remembered per (kernel, device, size) so that the runtime call is made once,
// not on every launch.
template <typename K>
inline int set_lds(K kernel, size_t bytes) {
    struct Entry {
        const void* fn;
        int dev;
        size_t bytes;
    };
    static std::mutex mu;
    static std::vector<Entry> done;
    const void* fn = reinterpret_cast<const void*>(kernel);
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const Entry& e : done)
            if (e.fn == fn && e.dev == dev && e.bytes == bytes) return MPX_OK;
    }
    MPX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    std::lock_guard<std::mutex> lock(mu);
    done.push_back({fn, dev, bytes});
    return MPX_OK;
}


}  // namespace mpx
