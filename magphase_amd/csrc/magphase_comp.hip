// magphase_comp.hip -- compressed-feature synthesis kernels (synthesis_from_compressed, magphase.py:825-997).
//
//   k_mel_unwarp_mfma / _tiled   [F x K] x [K x H] -> exp / identity: la.sp_mel_unwarp and phase_uncompress_type1_mcep as the
//                         linear maps they are, K <= 64, on the f32 matrix instructions
//   k_noise_stats<P>      per frame: windowed noise frame -> FFT -> sum_k (ln|Ns[k]|)^2, k = 1..N/2-1 (Q10 gain statistics)
//   k_synth_comp_pair<P>  per chunk of frames: noise FFT (recomputed) + periodic/aperiodic spectrum assembly
//                         (Appendix A2 steps 9-12) + inverse FFT + anti-ringing window + LDS overlap-add (as k_synth_ola_pair)
#include <type_traits>

#include "mpx_common.hpp"

// waves per workgroup of k_synth_comp_pair (8 = two per SIMD, 12 = three: compact transform front, <= 168 VGPRs)
#ifndef MPX_COMP_PAIR_WAVES
#define MPX_COMP_PAIR_WAVES 12
#endif
#ifndef MPX_COMP_DIT
#define MPX_COMP_DIT 0   // 1: both transforms of the compact form in the DIT form (wave_fft.hpp; parity-green): -1 % on the
                         // synthesis side of configs[2], but 10 registers spill again at 12 waves per CU: off
#endif

namespace mpx {

// ---------------------------------------------------------------------------------------------
// mel unwarp GEMM
// ---------------------------------------------------------------------------------------------
struct UnwarpJob {
    const float* A;  // [F x K] mel-domain features
    const float* U;  // [K x H] unwarp matrix
    float* out;      // [F x H]
    int K;
    int op;          // 0: identity, 1: exp
};
struct UnwarpJobs {
    UnwarpJob j[3];
};

constexpr int kGemmFrames = 64;   // frames per block
constexpr int kGemmKMax = 64;

// ---------------------------------------------------------------------------------------------
// mel warp GEMM (compressed analysis): out[f][i] = post( sum_k W[i][k] * pre(x[f][k]) ), i < nout <= 64, k < H
// ---------------------------------------------------------------------------------------------
// la.sp_mel_warp (libaudio.py:643-661) = SPTK ``mcep -j 0`` (log-periodogram -> real IFFT -> halve c0, c_{N/2} ->
// freqt) followed by the alpha = 0 cosine matrix: a LINEAR map of the log-periodogram, precomputed on the host as W
// (hostmath.warp_matrix).  pre: mode 0 (|f(w)|, mcep -q 3): ln(x^2 + 1e-8); mode 1 (ln|f(w)|, -q 2): ln(exp(x)^2 + 1e-8).
// Rows may be interpolated on the fly (variable -> constant frame rate, magphase.py:2219-2239) BEFORE pre().
// post: mode 0: none (the reference's exp followed by la.log); mode 1: * voiced, clip to [-1, 1] (magphase.py:2527-2532).
struct WarpJob {
    const float* x;      // [rows x H]
    const float* W;      // [nout x H]
    float* out;          // [F x nout]
    const float* voi;    // [F] or null
    int nout;
    int mode;
    long long F;         // frames (rows of out) of THIS job: the phase jobs may run on other rows than the magnitudes
};
struct WarpJobs {
    WarpJob j[3];
};

// Prologue of the mel warp's operand (WarpJob::mode) and the matching epilogue:
//   0  magnitudes, cepstral warp: ln(x^2 + 1e-8)                                   (mcep -q 3 -e 1e-8 on x, libaudio.py:575-661)
//   1  phase streams (mcep -q 2 on exp(x), |x| <= 1): ln(e^{2x} + 1e-8) = 2x + 1e-8 e^{-2x} (the next term is 5e-17);
//      epilogue voicing mask + clip
//   3  phase streams on the VARIABLE-rate rows (mpx_mel_warp_rows): prologue of mode 1, no epilogue -- the 45 outputs
//      are interpolated to the constant rate, masked and clipped by k_warp_phase_rows; ``voi`` = the rows in use
//   2  magnitudes, filter bank: la.log(x) = ln x with -1e10 for x == 0 (libaudio.py:241-248, :763-769); the reference then
//      takes exp and la.log again (magphase.py:2505-2510): the identity unless the exp underflows to 0 (sum < ln of the
//      smallest float64, -745.13), which comes back as -1e10
__device__ __forceinline__ float warp_prologue(int mode, float x) {
    // the argument is >= 1e-8 (never a denormal): the hardware log2 as it is, without __logf's denormal rescaling
    if (mode == 0) return __builtin_amdgcn_logf(fmaf(x, x, 1.0e-8f)) * 0.69314718055994531f;
    if (mode == 1 || mode == 3) return fmaf(1.0e-8f, __expf(-2.0f * x), 2.0f * x);
    return (x > 0.0f) ? __logf(x) : -1.0e10f;
}
__device__ __forceinline__ float warp_epilogue(int mode, float y, float vo) {
    if (mode == 1) return (vo == 0.0f) ? 0.0f : fminf(fmaxf(y * vo, -1.0f), 1.0f);   // +0 for unvoiced frames, always
    if (mode == 2) return (y < -745.13321f) ? -1.0e10f : y;
    return y;
}

constexpr int kWarpTile = 64;
#ifndef MPX_WARP_STRIDE
#define MPX_WARP_STRIDE 68
#endif
constexpr int kWarpStride = MPX_WARP_STRIDE;   // floats per LDS row (multiple of 4 for float4 reads)
#ifndef MPX_WARP_KC
#define MPX_WARP_KC 64
#endif
constexpr int kWarpKC = MPX_WARP_KC;                 // bins per staged chunk of the MFMA warp (64 or 128)
// LDS rows of the MFMA warp.  Round 2 padded them to 68 floats and measured SQ_LDS_BANK_CONFLICT = 31 % of the kernel's
// LDS cycles: a ds_read_b128 is served in groups of 16 lanes ({0-3, 12-15, 20-27}, ...; MI355X_MICROARCH.md, LDS), a
// group holds every fragment row li = lane & 15 once but from TWO k groups g = lane >> 4, and with the k offset 16 g
// added to the row's padding offset 4 li two rows of a group always met in a bank.  Now: dense rows (64 floats: every
// row starts in bank 0) and the 16-byte chunk c of row r stored at chunk c ^ swz(r & 15), swz even for r in 4..11 and
// odd otherwise -- within a lane group the g = 0 lanes then read the odd (even) chunks ^ q and the g = 1 lanes the even
// (odd) ones: 16 distinct chunks, conflict-free; the staging writes (8 lanes = 8 consecutive chunks of one row) stay so.
#ifndef MPX_WARP_SWIZZLE
#define MPX_WARP_SWIZZLE (MPX_WARP_KC == 64)
#endif
constexpr bool kWarpSwizzle = MPX_WARP_SWIZZLE;
constexpr int kWarpKStride = kWarpSwizzle ? kWarpKC : kWarpKC + (kWarpStride - 64);
static_assert(!kWarpSwizzle || kWarpKC == 64, "the chunk swizzle is defined for 64-bin chunks (16 chunks of 16 bytes per row)");
__device__ __forceinline__ int warp_swz(int r15) {
    if (!kWarpSwizzle) return 0;
    const bool mid = (r15 >= 4) && (r15 < 12);
    return mid ? 2 * (r15 - 4) : 2 * ((r15 < 4) ? r15 : r15 - 8) + 1;
}

// ---------------------------------------------------------------------------------------------
// noise frame -> half spectrum in registers
// ---------------------------------------------------------------------------------------------
// Windowed noise frame (magphase.py:886-897: windowing() with per-frame window list, epoch moved to index 0 by
// frm_list_to_matrix + fftshift) -> N-point real FFT -> Ns[k] for the bins kappa(lane) + 64 j, j = 0..P-1
// (natural j), plus the Nyquist bin (real) on the lane with kappa == 0.  Synchronous staging (no prefetch).
// COMPACT (P == 32): half-height exchange buffer (tiles of 32 P samples) and half twiddle table (wave_fft_front_compact;
// (lc, ls) = W_128^lane) -- k_synth_comp_pair at 12 waves per CU.
template <int P, bool PRESTAGED = false, bool COMPACT = false>   // PRESTAGED: the caller already copied tile 0 into xbuf and waited for it
__device__ __forceinline__ void noise_fft(const FrameGeom& g, int wtype, const float* tw, float* xbuf,
                                          unsigned xbuf_byte, int lane, float (&re)[P], float (&im)[P], float lc = 1.0f,
                                          float ls = 0.0f) {
    constexpr int M = 64 * P, N = 2 * M, kTile = COMPACT ? 32 * P : 64 * P;
    MPX_MARK("zero_init");
#pragma unroll
    for (int j = 0; j < P; ++j) re[j] = im[j] = 0.0f;
    const int ntiles = (g.len + kTile - 1) / kTile;
    for (int t = 0; t < ntiles; ++t) {
        MPX_MARK("window");
        const int tile0 = t * kTile;
        if (!PRESTAGED || t > 0) {
            stage_samples_async(g, tile0, kTile, xbuf_byte, lane);
            staged_wait<0>();
        }
        const int hi = min(g.len, tile0 + kTile);
        // the window type is per frame (wave-uniform): one loop per type, not a select per sample
        if (wtype == 0) {
            for (int k = tile0 + lane; k < hi; k += 64)
                xbuf[k - tile0] *= half_window(k, g.L, g.LR, g.kadd, g.invL, g.invR, 0);
        } else {
            for (int k = tile0 + lane; k < hi; k += 64)
                xbuf[k - tile0] *= half_window(k, g.L, g.LR, g.kadd, g.invL, g.invR, 1);
        }
        wave_sync();
        MPX_MARK("gather");
        // Gather y[n] = x[(n + rot) mod N]: lane l, row j <-> n = 128 j + 2 l (real part), + 1 (imaginary part).  Rows that
        // hold no sample of the frame are skipped (wave-uniform: n < len - rot is the frame's right half, n >= N - rot its
        // left half); inside an active row every lane reads -- clamped address, no exec-masked branch, so all reads of the
        // row batch are in flight together instead of one LDS round trip per element -- and keeps the value only where its
        // sample index falls into the tile.  (Round 6: the branchy form was ~5 SALU + a serialised s_waitcnt per element, and
        // its 32 hoisted row masks were 64 SGPRs of the pair kernels' spills.)
        {
            int n_lo = g.len - g.rot, n_hi = N - g.rot;
            asm volatile("" : "+s"(n_lo), "+s"(n_hi));   // evaluated here, per tile: not hoisted as 32 lane masks
            const unsigned span = (unsigned)(hi - tile0);
            const unsigned b0 = (unsigned)(2 * lane + g.rot) - (unsigned)tile0;
#pragma unroll
            for (int j = 0; j < P; ++j) {
                constexpr int LBJ = ilog2(P);
                const int m0 = 128 * j;
                if ((m0 < n_lo) || (m0 + 127 >= n_hi)) {
                    const int rj = (COMPACT && MPX_COMP_DIT) ? brev(j, LBJ) : j;   // the DIT form wants register brev(j) <- z[l + 64 j]
                    // sample index relative to the tile, modulo N (tile0 is a multiple of the tile length, which divides N)
                    const unsigned d0 = (b0 + (unsigned)m0) & (unsigned)(N - 1), d1 = (d0 + 1u) & (unsigned)(N - 1);
                    const float v0 = xbuf[min(d0, (unsigned)(kTile - 1))], v1 = xbuf[min(d1, (unsigned)(kTile - 1))];
                    re[rj] = (d0 < span) ? v0 : re[rj];
                    im[rj] = (d1 < span) ? v1 : im[rj];
                }
            }
        }
        wave_sync();
    }
    mpx_pin(re), mpx_pin(im);
    MPX_MARK("fft_forward");
    if constexpr (COMPACT) {   // (lc, ls) = W_128^lane from the pad of the lane's table row (k_synth_comp_pair keeps them there)
        const float4 pk = tw_half_pad<P>(tw, lane);
#if MPX_COMP_DIT
        wave_fft_dit_compact<P, -1>(re, im, tw, xbuf, lane, pk.z, pk.w);   // output: register i <-> Z[lane + 64 i]
#else
        wave_fft_front_compact<P, -1>(re, im, tw, xbuf, lane, pk.z, pk.w);
        fft_inreg<P, -1>(re, im);
#endif
        (void)lc;
        (void)ls;
    } else {
        wave_fft<P, -1>(re, im, tw, xbuf, lane);
    }
}

// Spectrum of the windowed noise frame in PAIRED layout: lane l (kappa = kappa(l)) gets, for q < P/2, its own bin
// k = kappa + 64 q (no_*) and the mirrored bin M - k (nm_*) from ONE evaluation of the real-FFT split's E / T terms
// (X[k] = E + T, X[M-k] = conj(E - T)): P lane exchanges and P/2 split evaluations per lane instead of 2P and P of the
// per-bin form.  The kappa == 0 lane's q == 0 pair is (DC, Nyquist); bin M/2 is its own mirror and comes out separately
// (nh_*, meaningful on the kappa == 0 lane).
template <int P, bool PRESTAGED = false, bool COMPACT = false>
__device__ __forceinline__ void noise_spectrum_paired(const FrameGeom& g, int wtype, const float* tw, float* xbuf,
                                                      unsigned xbuf_byte, int lane, float wl_c, float wl_s,
                                                      float (&no_r)[P / 2], float (&no_i)[P / 2], float (&nm_r)[P / 2],
                                                      float (&nm_i)[P / 2], float& nh_r, float& nh_i, float lc = 1.0f,
                                                      float ls = 0.0f) {
    constexpr int LB = ilog2(P);
    float re[P], im[P];
    noise_fft<P, PRESTAGED, COMPACT>(g, wtype, tw, xbuf, xbuf_byte, lane, re, im, lc, ls);
    mpx_pin(re), mpx_pin(im);
    MPX_MARK("split");
    if constexpr (COMPACT) {   // the split twiddle W_N^kappa from the table row's pad: not live across the transform
        const float4 pk = tw_half_pad<P>(tw, lane);
        wl_c = pk.x;
        wl_s = pk.y;
    }
    const int kap = kappa<P>(lane);
    const int src_lane = kappa<P>((64 - kap) & 63);
    const bool lane0 = (kap == 0);
    // partner bins SB at a time: 2 SB lane exchanges in flight together.  All P/2 at once (8 waves per CU) keeps 64 inputs +
    // 32 partners + the growing outputs live; in batches the own (even) and the source (odd) registers of a batch die as
    // its four outputs per bin pair appear -- what the 12-wave form (<= 168 VGPRs) needs.
#ifndef MPX_NOISE_SPLIT_BATCH
#define MPX_NOISE_SPLIT_BATCH (MPX_COMP_PAIR_WAVES > 8 ? 8 : 16)
#endif
    constexpr int SB = (P / 2 < MPX_NOISE_SPLIT_BATCH) ? P / 2 : MPX_NOISE_SPLIT_BATCH;
#pragma unroll
    for (int qb = 0; qb < P / 2; qb += SB) {
        float zpr[SB], zpi[SB];
        // register of row q: brev(q) after the DIF transform, q itself after the DIT one (compact form with MPX_COMP_DIT)
        constexpr bool NATR = COMPACT && MPX_COMP_DIT;
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int i = NATR ? qb + u : brev(qb + u, LB);
            zpr[u] = __shfl(re[P - 1 - i], src_lane);
            zpi[u] = __shfl(im[P - 1 - i], src_lane);
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int q = qb + u;
            const int i = NATR ? q : brev(q, LB);
            const int i0 = NATR ? (P - q) % P : brev((P - q) % P, LB);
            const float pr = lane0 ? re[i0] : zpr[u];
            const float pi = lane0 ? im[i0] : zpi[u];
            const float er = 0.5f * (re[i] + pr), ei = 0.5f * (im[i] - pi);
            const float orr = 0.5f * (im[i] + pi), oi = -0.5f * (re[i] - pr);
            const float cq = cos2p<P>(q), sq = -sin2p<P>(q);   // W_N^k = W_N^kappa * e^{-2 pi i q/(2P)}
            const float wr = wl_c * cq - wl_s * sq, wi = wl_c * sq + wl_s * cq;
            const float tr = wr * orr - wi * oi, ti = wr * oi + wi * orr;
            no_r[q] = er + tr;
            no_i[q] = ei + ti;
            nm_r[q] = er - tr;
            nm_i[q] = ti - ei;
        }
    }
    constexpr int ih = (COMPACT && MPX_COMP_DIT) ? P / 2 : 1;   // bin M/2 = register brev(P/2) = 1 (DIF) / P/2 (DIT) of the kappa == 0 lane
    nh_r = re[ih];   // X = conj Z
    nh_i = -im[ih];
}

template <int P, bool PRESTAGED = false, bool COMPACT = false>   // PRESTAGED: the caller already copied tile 0 into xbuf and waited for it
__device__ __forceinline__ void noise_spectrum(const FrameGeom& g, int wtype, const float* tw, float* xbuf,
                                               unsigned xbuf_byte, int lane, float wl_c, float wl_s,
                                               float (&nr)[P], float (&ni)[P], float& nM, float lc = 1.0f, float ls = 0.0f) {
    constexpr int LB = ilog2(P);
    float re[P], im[P];
    noise_fft<P, PRESTAGED, COMPACT>(g, wtype, tw, xbuf, xbuf_byte, lane, re, im, lc, ls);
    if constexpr (COMPACT) {
        const float4 pk = tw_half_pad<P>(tw, lane);
        wl_c = pk.x;
        wl_s = pk.y;
    }
    // real-FFT split for every own bin (redundant form: each lane evaluates X[k] for all its bins)
    const int kap = kappa<P>(lane);
    const int src_lane = kappa<P>((64 - kap) & 63);
    const bool lane0 = (kap == 0);
    // partner fetches in batches of 8 bins: 16 lane exchanges in flight per LDS latency instead of one
#pragma unroll
    for (int ib = 0; ib < P; ib += 8) {
        float prb[8], pib[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            prb[u] = __shfl(re[P - 1 - (ib + u)], src_lane);
            pib[u] = __shfl(im[P - 1 - (ib + u)], src_lane);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = ib + u;
            constexpr bool NATR = COMPACT && MPX_COMP_DIT;   // register i holds row q = i (DIT) / brev(i) (DIF)
            const int q = NATR ? i : brev(i, LB);
            const int i0 = NATR ? (P - q) % P : brev((P - q) % P, LB);
            const float pr = lane0 ? re[i0] : prb[u];
            const float pi = lane0 ? im[i0] : pib[u];
            const float er = 0.5f * (re[i] + pr), ei = 0.5f * (im[i] - pi);
            const float orr = 0.5f * (im[i] + pi), oi = -0.5f * (re[i] - pr);
            const float cq = cos2p<P>(q), sq = -sin2p<P>(q);
            const float wr = wl_c * cq - wl_s * sq, wi = wl_c * sq + wl_s * cq;
            nr[q] = er + (wr * orr - wi * oi);
            ni[q] = ei + (wr * oi + wi * orr);
        }
    }
    nM = re[0] - im[0];   // Nyquist bin X[M] = Re Z[0] - Im Z[0] (meaningful on the kappa == 0 lane)
}

// floats per frame of the stored noise spectra (N = 4096): 17 slots x 64 lanes x float4
constexpr int kSpecFrameFloats = 17 * 64 * 4;
typedef float spec_f32x4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(kAnaThreads) void k_noise_stats(const float* __restrict__ noise,
                                                          const long long* __restrict__ npos,
                                                          const int* __restrict__ nleft,
                                                          const int* __restrict__ nright,
                                                          const int* __restrict__ wtype, long long nframes,
                                                          const float* __restrict__ tw_g,
                                                          float* __restrict__ out_sum,
                                                          float* __restrict__ spec_out) {
    constexpr int M = 64 * P, N = 2 * M;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tw = smem;
    const int lane_id = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* xbuf = smem + tw_floats<P>() + wave * (P * kXStride);
    const unsigned xbuf_byte = 4u * (unsigned)(tw_floats<P>() + rfl(wave) * (P * kXStride));
    for (int i = threadIdx.x; i < tw_floats<P>(); i += kAnaThreads) tw[i] = tw_g[i];
    unsigned* queue = reinterpret_cast<unsigned*>(smem + tw_floats<P>() + kAnaWaves * (P * kXStride));
    if (threadIdx.x == 0) *queue = 0u;
    __syncthreads();
    float wl_s0, wl_c0;
    sincospif(-2.0f * (float)kappa<P>(lane_id) / (float)N, &wl_s0, &wl_c0);
    // compute-only kernel (one float out per frame): 12 waves per CU like k_analysis (3 per SIMD); the workgroup's frames
    // are pulled from its LDS queue (queue_pull, mpx_common.hpp: a SIMD serves its waves by age)
    long long fb, fe;
    block_frame_range(nframes, fb, fe);
    for (long long f = queue_pull(queue, fb); f < fe; f = queue_pull(queue, fb)) {
        int lane = lane_id;
        float wl_s = wl_s0, wl_c = wl_c0;
        asm volatile("" : "+v"(lane), "+v"(wl_s), "+v"(wl_c));
        const FrameGeom g = frame_geom(noise, npos[f], nleft[f], nright[f], N);
        float no_r[P / 2], no_i[P / 2], nm_r[P / 2], nm_i[P / 2], nh_r, nh_i;
        noise_spectrum_paired<P>(g, wtype[f], tw, xbuf, xbuf_byte, lane, wl_c, wl_s, no_r, no_i, nm_r, nm_i, nh_r, nh_i);
        if constexpr (P == 32) {
            // "noise spectra once" form (mpx_noise_stats_spectra): the frame's paired spectrum goes to HBM as the registers
            // hold it -- 17 slots of 64 lanes x float4 (no_r, no_i, nm_r, nm_i four rows at a time, then the bin-M/2 pair),
            // 1 KB per wave store -- and k_synth_comp_pair<.., SPEC> loads it back instead of transforming the frame again
            if (spec_out) {
                spec_f32x4* d = reinterpret_cast<spec_f32x4*>(spec_out + f * (long long)kSpecFrameFloats) + lane_id;
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    __builtin_nontemporal_store(spec_f32x4{no_r[4 * s_], no_r[4 * s_ + 1], no_r[4 * s_ + 2], no_r[4 * s_ + 3]}, d + 64 * s_);
                    __builtin_nontemporal_store(spec_f32x4{no_i[4 * s_], no_i[4 * s_ + 1], no_i[4 * s_ + 2], no_i[4 * s_ + 3]}, d + 64 * (4 + s_));
                    __builtin_nontemporal_store(spec_f32x4{nm_r[4 * s_], nm_r[4 * s_ + 1], nm_r[4 * s_ + 2], nm_r[4 * s_ + 3]}, d + 64 * (8 + s_));
                    __builtin_nontemporal_store(spec_f32x4{nm_i[4 * s_], nm_i[4 * s_ + 1], nm_i[4 * s_ + 2], nm_i[4 * s_ + 3]}, d + 64 * (12 + s_));
                }
                __builtin_nontemporal_store(spec_f32x4{nh_r, nh_i, 0.0f, 0.0f}, d + 64 * 16);
            }
        }
        // sum over bins 1..M-1 of (ln|Ns|)^2 = (0.5 ln |Ns|^2)^2 ; |Ns| == 0 -> protected log MAGIC = -1e10 (libaudio.py:241-248)
        // paired layout: own bin + mirror per step; the kappa == 0 lane's first pair is (DC, Nyquist), both excluded, and
        // that lane adds bin M/2
        const bool lane0 = (kappa<P>(lane) == 0);
        // |Ns|^2 of every bin first; the logarithm is the hardware log2 as it is (3 instructions per bin) when no bin of the
        // frame is zero or a denormal -- always, in practice -- and __logf with its rescaling sequence otherwise (one
        // wave-uniform branch per frame instead of a select per bin)
        float so[P / 2], sm[P / 2];
        const float sh = nh_r * nh_r + nh_i * nh_i;
        float smin = lane0 ? sh : 1.0f;
#pragma unroll
        for (int q = 0; q < P / 2; ++q) {
            so[q] = no_r[q] * no_r[q] + no_i[q] * no_i[q];
            sm[q] = nm_r[q] * nm_r[q] + nm_i[q] * nm_i[q];
            smin = fminf(smin, fminf(so[q], sm[q]));
        }
        float acc;
        if (!__any(smin < 1.1754944e-38f)) {
            auto term = [](float s_) {
                const float lg = 0.34657359027997264f * __builtin_amdgcn_logf(s_);   // 0.5 ln 2 log2(s)
                return lg * lg;
            };
            acc = lane0 ? term(sh) : 0.0f;
#pragma unroll
            for (int q = 0; q < P / 2; ++q) {
                const float tq = term(so[q]) + term(sm[q]);
                acc += (q == 0 && lane0) ? 0.0f : tq;
            }
        } else {
            auto term = [](float s_) {
                const float lg = (s_ > 0.0f) ? 0.5f * __logf(s_) : -1.0e10f;
                return lg * lg;
            };
            acc = lane0 ? term(sh) : 0.0f;
#pragma unroll
            for (int q = 0; q < P / 2; ++q) {
                const float tq = term(so[q]) + term(sm[q]);
                acc += (q == 0 && lane0) ? 0.0f : tq;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane_id == 0) out_sum[f] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// noise gains (magphase.py:902-906, Q10): per utterance and class (voiced / unvoiced)
//   g = sqrt(exp(mean over the class's frames and bins 1..N/2-1 of (ln|Ns|)^2)),  inv_gain[f] = 1 / g(class of f)
// from the per-frame sums of k_noise_stats.  One block per utterance, float64 accumulation; an empty class gives
// NaN exactly like np.mean of an empty selection (it is never applied to a frame).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_noise_gains(const float* __restrict__ sums, const int* __restrict__ voiced,
                                                     const int* __restrict__ utt_frame_off, int bins_per_frame,
                                                     float* __restrict__ inv_gain, double* __restrict__ gains) {
    __shared__ double s_sum[2][256];
    __shared__ int s_cnt[2][256];
    const int u = blockIdx.x;
    const int f0 = utt_frame_off[u], f1 = utt_frame_off[u + 1];
    double acc[2] = {0.0, 0.0};
    int cnt[2] = {0, 0};
    for (int f = f0 + threadIdx.x; f < f1; f += 256) {
        const int c = voiced[f] ? 0 : 1;
        acc[c] += (double)sums[f];
        cnt[c] += 1;
    }
    for (int c = 0; c < 2; ++c) {
        s_sum[c][threadIdx.x] = acc[c];
        s_cnt[c][threadIdx.x] = cnt[c];
    }
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) {
            for (int c = 0; c < 2; ++c) {
                s_sum[c][threadIdx.x] += s_sum[c][threadIdx.x + off];
                s_cnt[c][threadIdx.x] += s_cnt[c][threadIdx.x + off];
            }
        }
        __syncthreads();
    }
    double g[2];
    for (int c = 0; c < 2; ++c)
        g[c] = (s_cnt[c][0] > 0) ? sqrt(exp(s_sum[c][0] / ((double)s_cnt[c][0] * (double)bins_per_frame))) : nan("");
    if (threadIdx.x == 0 && gains) {
        gains[2 * u + 0] = g[0];
        gains[2 * u + 1] = g[1];
    }
    for (int f = f0 + threadIdx.x; f < f1; f += 256) inv_gain[f] = (float)(1.0 / g[voiced[f] ? 0 : 1]);
}

// ---------------------------------------------------------------------------------------------
// post-filter (magphase.py:2300-2378, Q20) on the log-mel magnitude [F x D]: per bin a centred moving average of odd
// length lens[b] (host table, linearly shrinking/growing with frequency), enhancement
// y = (x - ave) * tilt[b] + ave, the two end bins copied.  One thread per (frame, bin); D <= 256.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_post_filter(const float* __restrict__ x, long long F, int D,
                                                     const int* __restrict__ half_len, int nx0, int nx1,
                                                     const float* __restrict__ tilt, float* __restrict__ y) {
    extern __shared__ float row[];   // rows_per_block x D
    const int rows_per_block = blockDim.x / D;
    const int rl = threadIdx.x / D, b = threadIdx.x - rl * D;
    const long long f = (long long)blockIdx.x * rows_per_block + rl;
    const bool live = (rl < rows_per_block) && (f < F);
    if (live) row[rl * D + b] = x[f * D + b];
    __syncthreads();
    if (!live) return;
    const float* r = row + rl * D;
    // averages exist for bins nx0..nx1 (inclusive); outside they repeat the boundary value (magphase.py:2357-2358:
    // v_ave[:v_nx[0]] = v_ave[v_nx[0]] ; v_ave[v_nx[-1]:] = v_ave[v_nx[-1]])
    const int bc = min(max(b, nx0), nx1);
    const int h = half_len[bc - nx0];
    float acc = 0.0f;
    for (int k = bc - h; k <= bc + h; ++k) acc += r[k];
    const float ave = acc / (float)(2 * h + 1);
    float out = (r[b] - ave) * tilt[b] + ave;
    if (b == 0 || b == D - 1) out = r[b];
    y[f * D + b] = out;
}

// ---------------------------------------------------------------------------------------------
// output high-pass (magphase.py:981-995: butter(4, 40 Hz) + lfilter), float64, blocked scan over a CASCADE of two
// second-order sections.  Each section's direct-form-II-transposed recurrence
//   y = b0 x + z0 ; z0 = b1 x + z1 - a1 y ; z1 = b2 x - a2 y
// is linear in (z, x): k_hpf_zero_state runs it per block of kHpfBlock samples from z = 0 (parallel over blocks),
// k_hpf_carry chains the block end states z_{j+1} = A^B z_j + zs_j per utterance (2x2, serial, tiny), k_hpf_apply adds
// each block's free response G[n] . z_start (G[n] = C A^n, host table).  Why a cascade: chaining the states of the
// 4th-order direct form is hopeless in float64 (four poles at |z| ~ 0.997 within 0.005 of each other: A^1024 has
// entries of 3e8 and the block hand-over loses everything -- measured), while the biquads' tables stay below 120.
// The cascade differs from scipy's direct-form lfilter by ~1e-7 of peak, which is lfilter's own round-off noise.
// ---------------------------------------------------------------------------------------------
// Round 5: on the corpus generation path (32 utterances per launch) the two section passes took 0.57 ms of the launch's
// 2.8 ms of device time -- one THREAD per 1024-sample block, every lane of a wave reading its own cache line.  Now a
// block is 256 samples and a WAVE takes 64 consecutive blocks of an utterance: 64 x 64 tiles go through LDS (row stride
// 65 doubles: the wave-wide loads / stores are 64 consecutive samples, a lane's serial pass reads its own row without bank
// conflicts), and the carry kernel is one wave per utterance with the block states staged through LDS the same way.
constexpr int kHpfBlock = 256;
constexpr int kHpfTile = 64;                    // samples of a block per LDS pass (and blocks per wave)
constexpr int kHpfTileStride = kHpfTile + 1;    // doubles per LDS row

struct BiquadCoef {
    double b0, b1, b2, a1, a2;
};

// cg / cz (second section only): the PREVIOUS section's free response is added while loading -- x[n] + G[n mod 256] . z_start
// of the block -- instead of by a k_hpf_apply pass over the whole signal in between (0.49 GB of traffic per 128 utterances).
template <typename TIn>
__global__ __launch_bounds__(64) void k_hpf_zero_state(const TIn* __restrict__ x, const long long* __restrict__ off,
                                                       const int* __restrict__ blk_off, BiquadCoef c,
                                                       double* __restrict__ y, double* __restrict__ zend,
                                                       const double* __restrict__ cg, const double* __restrict__ cz) {
    // one wave per 64 consecutive blocks of an utterance, lane t = block 64 blockIdx.x + t; blk_off[u] = first global
    // block index of utterance u
    __shared__ double tile[kHpfTile * kHpfTileStride];
    const int u = blockIdx.y;
    const int nb = blk_off[u + 1] - blk_off[u];
    const int j0 = blockIdx.x * 64;
    if (j0 >= nb) return;
    const int t = threadIdx.x;
    const long long base = off[u] + (long long)j0 * kHpfBlock;   // first sample of the wave's blocks
    const long long end = off[u + 1];
    __shared__ double czs[2 * kHpfTile];
    if (cg) {   // start states of the wave's 64 blocks in the previous section
        const int jb = min(j0 + t, nb - 1);
        czs[2 * t] = cz[2 * (long long)(blk_off[u] + jb)];
        czs[2 * t + 1] = cz[2 * (long long)(blk_off[u] + jb) + 1];
        __syncthreads();
    }
    double z0 = 0, z1 = 0;
    for (int ch = 0; ch < kHpfBlock / kHpfTile; ++ch) {
        // row r = block j0 + r, its samples [ch * 64, ch * 64 + 64): lane t loads column t of every row (coalesced)
        {   // all 64 row loads in flight before the first LDS write (eight at a time left the pass waiting on memory
            // latency: 266 us per 128 utterances; see the round-5 notes)
            TIn xv[kHpfTile];
#pragma unroll
            for (int r = 0; r < kHpfTile; ++r) {
                const long long n = base + (long long)r * kHpfBlock + ch * kHpfTile + t;
                xv[r] = x[min(n, end - 1)];
            }
            double g0 = 0.0, g1 = 0.0;
            if (cg) {
                g0 = cg[2 * (ch * kHpfTile + t)];
                g1 = cg[2 * (ch * kHpfTile + t) + 1];
            }
#pragma unroll
            for (int r = 0; r < kHpfTile; ++r) {
                const long long n = base + (long long)r * kHpfBlock + ch * kHpfTile + t;
                double xd = (double)xv[r];
                if (cg) xd += g0 * czs[2 * r] + g1 * czs[2 * r + 1];   // the same operations as k_hpf_apply's
                tile[r * kHpfTileStride + t] = (n < end) ? xd : 0.0;
            }
        }
        __syncthreads();
        {   // the lane's row into registers first: 64 independent LDS reads in flight, then the recurrence alone is the chain
            double v[kHpfTile];
#pragma unroll
            for (int i = 0; i < kHpfTile; ++i) v[i] = tile[t * kHpfTileStride + i];
#pragma unroll
            for (int i = 0; i < kHpfTile; ++i) {
                const double xv = v[i];
                const double yv = c.b0 * xv + z0;
                z0 = c.b1 * xv + z1 - c.a1 * yv;
                z1 = c.b2 * xv - c.a2 * yv;
                v[i] = yv;
            }
#pragma unroll
            for (int i = 0; i < kHpfTile; ++i) tile[t * kHpfTileStride + i] = v[i];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kHpfTile; ++r) {
            const long long n = base + (long long)r * kHpfBlock + ch * kHpfTile + t;
            if (n < end) y[n] = tile[r * kHpfTileStride + t];
        }
        __syncthreads();
    }
    // (a block that ends before its 256th sample ran on zero padding: its end state is never used -- it is the
    // utterance's last block)
    if (j0 + t < nb) {
        double* ze = zend + 2 * (long long)(blk_off[u] + j0 + t);
        ze[0] = z0;
        ze[1] = z1;
    }
}

__global__ __launch_bounds__(64) void k_hpf_carry(const int* __restrict__ blk_off, int n_utts,
                                                  const double* __restrict__ pmat /* A^B, row-major 2x2 */,
                                                  const double* __restrict__ zend, double* __restrict__ zstart) {
    // z_{j+1} = P z_j + e_j over an utterance's blocks (e_j = the zero-state end state of block j), one wave per utterance,
    // 64 blocks per step as a SCAN across the lanes: after log-step k lane t holds sum_{t - 2^{k+1} < i <= t} P^{t-i} e_i
    // (c_t += P^{2^k} c_{t - 2^k}), so block t starts from P^t z_tile + c_{t-1} -- 6 exchange steps per 64 blocks instead
    // of 64 dependent LDS round trips (the serial form: 59 us per 32 utterances of 938 blocks).
    const int u = blockIdx.x;
    if (u >= n_utts) return;
    const int g0 = blk_off[u], g1 = blk_off[u + 1], t = threadIdx.x;
    double pw[7][4];   // P^(2^k)
    pw[0][0] = pmat[0], pw[0][1] = pmat[1], pw[0][2] = pmat[2], pw[0][3] = pmat[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double a = pw[k][0], b = pw[k][1], c = pw[k][2], d = pw[k][3];
        pw[k + 1][0] = a * a + b * c;
        pw[k + 1][1] = a * b + b * d;
        pw[k + 1][2] = c * a + d * c;
        pw[k + 1][3] = c * b + d * d;
    }
    double q0 = 1.0, q1 = 0.0, q2 = 0.0, q3 = 1.0;   // P^t of this lane (binary expansion of t)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if ((t >> k) & 1) {
            const double a = q0 * pw[k][0] + q1 * pw[k][2], b = q0 * pw[k][1] + q1 * pw[k][3];
            const double c = q2 * pw[k][0] + q3 * pw[k][2], d = q2 * pw[k][1] + q3 * pw[k][3];
            q0 = a, q1 = b, q2 = c, q3 = d;
        }
    }
    double zt0 = 0.0, zt1 = 0.0;   // state at the start of the tile
    for (int g = g0; g < g1; g += 64) {
        const int cnt = min(64, g1 - g);
        const double e0 = (t < cnt) ? zend[2 * (long long)(g + t)] : 0.0;
        const double e1 = (t < cnt) ? zend[2 * (long long)(g + t) + 1] : 0.0;
        double c0 = e0, c1 = e1;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double o0 = __shfl_up(c0, 1 << k), o1 = __shfl_up(c1, 1 << k);
            if (t >= (1 << k)) {
                c0 += pw[k][0] * o0 + pw[k][1] * o1;
                c1 += pw[k][2] * o0 + pw[k][3] * o1;
            }
        }
        double m0 = __shfl_up(c0, 1), m1 = __shfl_up(c1, 1);   // c_{t-1}
        if (t == 0) m0 = m1 = 0.0;
        const double zs0 = q0 * zt0 + q1 * zt1 + m0, zs1 = q2 * zt0 + q3 * zt1 + m1;
        if (t < cnt) {
            zstart[2 * (long long)(g + t)] = zs0;
            zstart[2 * (long long)(g + t) + 1] = zs1;
        }
        // the state after block t = one more step of the recurrence; the next tile starts from lane cnt - 1's
        const double n0 = pw[0][0] * zs0 + pw[0][1] * zs1 + e0, n1 = pw[0][2] * zs0 + pw[0][3] * zs1 + e1;
        zt0 = __shfl(n0, cnt - 1);
        zt1 = __shfl(n1, cnt - 1);
    }
}

__global__ __launch_bounds__(256) void k_hpf_apply(const long long* __restrict__ off, const int* __restrict__ blk_off,
                                                   const double* __restrict__ gtab /* [kHpfBlock x 2] */,
                                                   const double* __restrict__ zstart, double* __restrict__ y) {
    const int u = blockIdx.y;
    const long long len = off[u + 1] - off[u];
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= len) return;
    const int j = (int)(t / kHpfBlock), r = (int)(t - (long long)j * kHpfBlock);
    const double* z = zstart + 2 * (long long)(blk_off[u] + j);
    y[off[u] + t] += gtab[2 * r] * z[0] + gtab[2 * r + 1] * z[1];
}

// ---------------------------------------------------------------------------------------------
// 16-bit PCM for the wav writer (libaudio.py:352-365, Q17, as soundfile / libsndfile writes it): per utterance
// v = norm * y / max|y| in float64, then lrint(v * 0x7FFF) (round half to even; no clipping of in-range input).  Same
// IEEE operations in the same order as the host form (la.write_audio_file) -- __dmul_rn / __ddiv_rn keep the compiler
// from fusing them -- so the samples are bit-identical; the device hands the writer thread ready int16 samples and the
// D2H copy is a quarter of the float64 one.  k_peak_abs: one block per utterance; k_pcm16: one thread per sample.
// ---------------------------------------------------------------------------------------------
constexpr int kPeakPerThread = 16;   // elements per thread of k_peak_abs
template <typename T>
__global__ __launch_bounds__(256) void k_peak_abs(const T* __restrict__ y, const long long* __restrict__ off,
                                                  double* __restrict__ peak) {
    // peak[u] = max |y| over utterance u; peak[] zeroed by the caller.  blockIdx.y = utterance, 4096 elements per block
    // (one block per utterance ran 240 us per 32 x 5 s: a single wave front of loads in flight per CU).  The maximum is
    // order-independent, so the result is the serial one bit for bit; non-negative doubles order like their bit patterns.
    __shared__ double s_max[256];
    const int u = blockIdx.y;
    const long long b0 = off[u] + (long long)blockIdx.x * (256 * kPeakPerThread), b1 = off[u + 1];
    if (b0 >= b1) return;
    double m = 0.0;
#pragma unroll
    for (int k = 0; k < kPeakPerThread; ++k) {
        const long long i = b0 + k * 256 + threadIdx.x;
        if (i < b1) m = fmax(m, fabs((double)y[i]));
    }
    s_max[threadIdx.x] = m;
    __syncthreads();
    for (int k = 128; k >= 1; k >>= 1) {
        if (threadIdx.x < k) s_max[threadIdx.x] = fmax(s_max[threadIdx.x], s_max[threadIdx.x + k]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // (fmax drops NaNs, as the serial chain did: s_max[0] is a non-negative number)
        atomicMax(reinterpret_cast<unsigned long long*>(peak + u), (unsigned long long)__double_as_longlong(s_max[0]));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_pcm16(const T* __restrict__ y, const long long* __restrict__ off,
                                               const double* __restrict__ peak, double norm, short* __restrict__ out) {
    const int u = blockIdx.y;
    const long long i = off[u] + (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= off[u + 1]) return;
    double v = (double)y[i];
    if (norm > 0.0) v = __ddiv_rn(__dmul_rn(norm, v), peak[u]);
    const double r = rint(__dmul_rn(v, 32767.0));
    out[i] = (short)fmin(fmax(r, -32768.0), 32767.0);   // NaN (silent utterance: 0 / 0) -> fmax/fmin pick the bound
}

// int16 PCM -> float32 in [-1, 1): x * 2^-15, exact (what the host did with np.multiply before uploading float32 --
// the int16 samples cross PCIe at half the bytes and the host pass is gone).  4 samples per thread.
__global__ __launch_bounds__(256) void k_pcm16_to_f32(const short* __restrict__ in, long long n, float* __restrict__ out) {
    const long long i = 4 * ((long long)blockIdx.x * 256 + threadIdx.x);
    if (i + 3 < n) {
        const short4 v = *reinterpret_cast<const short4*>(in + i);
        *reinterpret_cast<float4*>(out + i) = make_float4((float)v.x * (1.0f / 32768.0f), (float)v.y * (1.0f / 32768.0f),
                                                          (float)v.z * (1.0f / 32768.0f), (float)v.w * (1.0f / 32768.0f));
    } else {
        for (long long k = i; k < n; ++k) out[k] = (float)in[k] * (1.0f / 32768.0f);
    }
}

// ---------------------------------------------------------------------------------------------
// minimum-phase spectrum from a magnitude spectrum (complex cepstrum), la.build_min_phase_from_mag_spec
// (libaudio.py:920-934): ln|X| -> even extension -> real IFFT (cepstrum c) -> causal fold (c[1..N/2-1] *= 2,
// c[N/2+1..] = 0) -> FFT -> exp.  Since Re FFT(fold c) == ln|X|, only the phase phi = Im FFT(fold c) is new: the
// kernel writes the unit phasor (cos phi, sin phi) where the synthesis kernel expects the (real, imag) phase
// features, and the (row-interpolated) magnitude it was computed from.  One wavefront per frame, two FFTs.
// ---------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(kThreads) void k_min_phase(const float* __restrict__ mag, const int* __restrict__ row0,
                                                        const int* __restrict__ row1,
                                                        const float* __restrict__ rowt, long long nframes,
                                                        const float* __restrict__ tw_g, float* __restrict__ omag,
                                                        float* __restrict__ oreal, float* __restrict__ oimag,
                                                        long long ld) {
    constexpr int M = 64 * P, N = 2 * M, H = M + 1, LB = ilog2(P);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tw = smem;
    const int lane_id = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* xbuf = smem + tw_floats<P>() + wave * (P * kXStride);
    for (int i = threadIdx.x; i < tw_floats<P>(); i += kThreads) tw[i] = tw_g[i];
    __syncthreads();
    float wa_s0, wa_c0, ws_s0, ws_c0;
    sincospif(-2.0f * (float)kappa<P>(lane_id) / (float)N, &wa_s0, &wa_c0);
    sincospif(2.0f * (float)lane_id / (float)N, &ws_s0, &ws_c0);
    const int wave_u = rfl(wave);
    for (long long f = (long long)blockIdx.x * kWavesPerBlock + wave_u; f < nframes;
         f += (long long)gridDim.x * kWavesPerBlock) {
        int lane = lane_id;
        float wa_s = wa_s0, wa_c = wa_c0, ws_s = ws_s0, ws_c = ws_c0;
        asm volatile("" : "+v"(lane), "+v"(wa_s), "+v"(wa_c), "+v"(ws_s), "+v"(ws_c));
        const int r0 = row0[f], r1 = row1[f];
        const float rt = rowt[f];
        const float* m0p = mag + (long long)r0 * ld;
        const float* m1p = mag + (long long)r1 * ld;
        // ---- ln|X| (protected log, libaudio.py:241-248) on bins lane + 64 j, scaled for the inverse transform
        float xr[P], xi[P], mv[P];
        const float scale = 0.5f / (float)M;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const float a = m0p[lane + 64 * j], b = m1p[lane + 64 * j];
            mv[j] = fmaf(b - a, rt, a);
        }
        float mM = 0.0f;
        if (lane == 0) {
            const float a = m0p[M], b = m1p[M];
            mM = fmaf(b - a, rt, a);
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            xr[j] = ((mv[j] > 0.0f) ? logf(mv[j]) : -1.0e10f) * scale;
            xi[j] = 0.0f;
        }
        const float xm = ((mM > 0.0f) ? logf(mM) : -1.0e10f) * scale;
        hermitian_merge<P>(xr, xi, xm, lane, ws_c, ws_s);
        wave_fft<P, +1>(xr, xi, tw, xbuf, lane);
        // ---- cepstrum samples n = 2m, 2m+1 with m = kappa + 64 brev(i): causal fold
        const int kap = kappa<P>(lane);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int n0 = 2 * (kap + 64 * brev(i, LB));
            const float w0 = (n0 == 0) ? 1.0f : ((n0 < M) ? 2.0f : ((n0 == M) ? 1.0f : 0.0f));
            const float w1 = (n0 + 1 < M) ? 2.0f : ((n0 + 1 == M) ? 1.0f : 0.0f);
            xr[i] *= w0;
            xi[i] *= w1;
        }
        // ---- forward real FFT of the folded cepstrum: input register j must hold z[lane + 64 j]
        float re[P], im[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            re[brev(i, LB)] = xr[i];
            im[brev(i, LB)] = xi[i];
        }
        if (P != 32) {
            const int src = kappa<P>(lane);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                re[j] = __shfl(re[j], src);
                im[j] = __shfl(im[j], src);
            }
        }
        wave_fft<P, -1>(re, im, tw, xbuf, lane);
        const int src_lane = kappa<P>((64 - kap) & 63);
        const bool lane0 = (kap == 0);
        float* mo = omag + f * ld;
        float* ro = oreal + f * ld;
        float* io = oimag + f * ld;
        // the magnitude row is stored from the lanes that loaded it (bins lane + 64 j)
#pragma unroll
        for (int j = 0; j < P; ++j) mo[lane + 64 * j] = mv[j];
        if (lane == 0) mo[M] = mM;
        // partner fetches in batches of 8 bins (16 lane exchanges in flight per LDS latency instead of one); phase ->
        // unit phasor with the hardware sin / cos (|error| ~ 5e-7, the phase itself carries ~1e-6 of fp32 FFT noise)
#pragma unroll
        for (int ib = 0; ib < P; ib += 8) {
            float prb[8], pib[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                prb[u] = __shfl(re[P - 1 - (ib + u)], src_lane);
                pib[u] = __shfl(im[P - 1 - (ib + u)], src_lane);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = ib + u;
                const int q = brev(i, LB);
                const int i0 = brev((P - q) % P, LB);
                const float pr = lane0 ? re[i0] : prb[u];
                const float pi = lane0 ? im[i0] : pib[u];
                const float ei = 0.5f * (im[i] - pi);
                const float orr = 0.5f * (im[i] + pi), oi = -0.5f * (re[i] - pr);
                const float cq = cos2p<P>(q), sq = -sin2p<P>(q);
                const float wr = wa_c * cq - wa_s * sq, wi = wa_c * sq + wa_s * cq;
                const float phi = ei + (wr * oi + wi * orr);   // Im S[k]
                float sn, cs;
                __sincosf(phi, &sn, &cs);
                const int k = kap + 64 * q;
                ro[k] = cs;
                io[k] = sn;
            }
        }
        if (lane0) {   // Nyquist bin of a real sequence: phase 0
            ro[M] = 1.0f;
            io[M] = 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// compressed synthesis + PSOLA
// ---------------------------------------------------------------------------------------------
struct CompFrameTabs {
    const long long* npos;   // noise frame epoch (absolute index into the noise buffer)
    const int* nleft;
    const int* nright;
    const int* wtype;        // 0 hann / 1 bartlett^2.5
    const int* voiced;       // 0 / 1
    const float* inv_gain;   // 1 / noise gain of the frame's class (Q10)
    const int* row0;         // feature rows to interpolate between (constant -> variable rate), row0 == row1 if none
    const int* row1;
    const float* rowt;       // interpolation weight of row1
    const int* win_l;        // anti-ringing window half lengths (Q14)
    const int* win_r;
    const int* pm_rel;
    const float* nspec;      // SPEC form: the frames' noise spectra as k_noise_stats stored them (else unused)
};

// ---------------------------------------------------------------------------------------------
// Compressed-feature synthesis + PSOLA, pair form: two waves share one LDS ring and alternate over the frames of the
// pair's runs (tickets in LDS, exactly as k_synth_ola_pair).  Round 3: 12 waves per CU (three per SIMD, <= 168 VGPRs)
// instead of 8 -- the kernel is parked in s_waitcnt a third of its wave cycles and moves 1 TB/s, a third wave per SIMD
// fills those gaps: synthesis side of configs[2] 1.350 -> 1.25 ms (interleaved A/B), 0.727 -> 0.632 ms for the launch.
// What made it fit: the compact transform front (half-height exchange buffer, noise staged in tiles of 1024 samples:
// frames longer than that -- f0 below 94 Hz at 48 kHz -- take a second, synchronous tile; half twiddle table), the lane
// constants of the split / merge / compact twiddles kept in the PAD of the lane's table row and read where they are used
// (tw_half_pad: six registers that are not live across the frame loop), the feature loads in the form SGPR row pointer +
// one zero-extended 32-bit lane offset (the int-indexed form made a 64-bit address per load: 208 v_lshl_add_u64 and as
// many register pairs in the ISA), the overlap-add 16 ring values at a time (ring_add_plane), and -- what finally removed
// the spills -- the assembly's load batches sized by what a row can need (template parameter NPQ below): with room for 10
// values per bin pair on every row, batches of 1 / 2 / 4 pairs spilled 0 / 6-18 / 17 registers at 1.448 / 1.273 / 1.253
// ms (the loads in flight are what the kernel lives on), and the spilled registers' scratch lines, evicted from L2 by the
// feature stream, cost 176-340 MB of HBM traffic per launch.  The feature rows -- read once, by one wave -- are loaded
// non-temporally.  The single-wave form it replaced (git
// history) held both feature rows, the per-bin curves and the noise FFT at once (466 VGPRs, ONE wave per SIMD: at one
// instruction per ~5.4 cycles and wave its ~7.5 k instructions per frame were the whole 1.35 ms).  Here the noise spectrum is
// computed first and the features are folded into it in place, half a spectrum (16 register rows) at a time:
// 64 + 128 live registers instead of 64 + 262, so two waves fit a SIMD.
// ---------------------------------------------------------------------------------------------
constexpr int kCompPairWaves = MPX_COMP_PAIR_WAVES;
constexpr int kCompPairs = kCompPairWaves / 2;
// More than 8 waves per CU: P == 32 does not fit the LDS with full-height exchange buffers (6 rings + 12 buffers + the
// table = 223 KB) -- the compact transform front of wave_fft.hpp (half-height buffers, noise staged in tiles of 1024
// samples, half twiddle table): 162.9 KB, as k_synth_ola_pair.
template <int P>
constexpr bool comp_compact() { return P == 32 && kCompPairWaves > 8; }
template <int P>
constexpr int comp_tw_floats() { return comp_compact<P>() ? tw_half_floats<P>() : tw_floats<P>(); }
template <int P>
constexpr int comp_xbuf_floats() { return (comp_compact<P>() ? P / 2 : P) * kXStride; }
template <int P>
constexpr size_t lds_bytes_comp_pair() {
    return sizeof(float) * (size_t)(comp_tw_floats<P>() + kCompPairWaves * comp_xbuf_floats<P>() + kCompPairs * ring_len<P>() + 16);
}

// LERP: every frame interpolates between two spectrum rows (row0 / row1 / rowt tables); false: one row per frame, row
// index = frame index (variable-rate input, or rows already interpolated by mpx_mel_unwarp_rows) -- half the feature loads.
// NPQ >= 0 (one-row-per-frame form): the caller's promise n_per <= 64 NPQ at compile time -- only the own bins of the
// register rows q < NPQ can have a periodic component, no mirror bin has (M - k > 64 NPQ).  The assembly then loads 7
// values per bin pair on those rows and 4 on the others instead of keeping room for 10 everywhere: batches of (4, 4, 8)
// pairs = 28 / 28 / 32 loads in flight for P == 32, NPQ == 8 (48 / 44.1 kHz: the crossfade ends at bin 512), against
// four batches of 40 -- the 17 registers the 40-wide batches spilled at 12 waves per CU are gone (166 VGPRs, no scratch),
// and the freed registers allow batches of (8, 8) = 56 / 32 loads: two exposed load latencies per frame instead of four.
// NPQ < 0: anything goes (run-time tests only).
// SPEC: the noise spectra come from HBM (tb.nspec, stored by k_noise_stats) instead of a second transform of the frame.
template <int P, bool LERP, int NPQ = -1, bool SPEC = false>
__global__ __launch_bounds__(kCompPairWaves * 64) void k_synth_comp_pair(const float* __restrict__ mag,
                                                                        const float* __restrict__ real,
                                                                        const float* __restrict__ imag,
                                                                        const float* __restrict__ noise,
                                                                        CompFrameTabs tb,
                                                                        const float* __restrict__ per_v,
                                                                        const float* __restrict__ ap_v,
                                                                        const float* __restrict__ ap_u,
                                                                        const RunDesc* __restrict__ runs,
                                                                        const int* __restrict__ slot_off,
                                                                        const int* __restrict__ slot_runs,
                                                                        int nslots,
                                                                        const float* __restrict__ tw_g,
                                                                        float* __restrict__ strips,
                                                                        float* __restrict__ pcm, long long ld, int n_per) {
    static_assert(!SPEC || (P == 32 && !LERP), "stored noise spectra: N = 4096, one row per frame");
    constexpr int M = 64 * P, N = 2 * M, R = ring_len<P>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tw = smem;
    const int lane_id = threadIdx.x & 63;
    const int wave = rfl((int)(threadIdx.x >> 6));
    const int pair = wave >> 1, half = wave & 1;
    constexpr bool kCompact = comp_compact<P>();
    float* xbuf = smem + comp_tw_floats<P>() + wave * comp_xbuf_floats<P>();
    const unsigned xbuf_byte = 4u * (unsigned)(comp_tw_floats<P>() + wave * comp_xbuf_floats<P>());
    constexpr int kRing0 = comp_tw_floats<P>() + kCompPairWaves * comp_xbuf_floats<P>();
    float* ring = smem + kRing0 + pair * R;
    const unsigned ring_byte = 4u * (unsigned)(kRing0 + pair * R);
    int* turn = reinterpret_cast<int*>(smem + kRing0 + kCompPairs * R) + pair;
    static_assert(kCompPairs <= 16, "the tickets fit the 16 floats behind the rings");
    pair_kernel_prologue<P, kCompact, MPX_COMP_DIT != 0>(tw, tw_g, smem + kRing0, kCompPairs * R, turn - pair, kCompPairs,
                                                         kCompPairWaves * 64);

    float wa_s0, wa_c0, ws_s0, ws_c0;   // analysis-side lane twiddle W_N^kappa and synthesis-side conj(W_N^lane)
    sincospif(-2.0f * (float)kappa<P>(lane_id) / (float)N, &wa_s0, &wa_c0);
    sincospif(2.0f * (float)lane_id / (float)N, &ws_s0, &ws_c0);
    // (compact form: these constants live in the pad of the lane's table row and are read where they are used -- six
    // registers that are not live across the frame loop)
    const int slot = blockIdx.x * kCompPairs + pair;
    if (slot >= nslots) return;

    // cursor over this wave's frames: every second frame of every run of the pair's work list (see k_synth_ola_pair)
    typedef PairCursor Cursor;
    const int wi_end = slot_off[slot + 1];
    auto advance = [&](Cursor& c) { pair_cursor_advance(c, wi_end, half, runs, slot_runs); };
    Cursor cur;
    cur.wi = slot_off[slot];
    cur.ticket_base = 0;
    pair_cursor_settle(cur, wi_end, half, runs, slot_runs);

    if (!cur.valid) return;

    // the noise samples of a frame are copied HBM -> LDS (into the transpose buffer) while the previous frame's
    // inverse FFT finishes and its overlap-add runs (same scheme as k_analysis)
    constexpr int kTile = kCompact ? 32 * P : 64 * P;
    FrameGeom g = frame_geom(noise, tb.npos[cur.fi], tb.nleft[cur.fi], tb.nright[cur.fi], N);
    if constexpr (!SPEC) stage_samples_async(g, 0, kTile, xbuf_byte, lane_id);

    while (cur.valid) {
        int lane = lane_id;
        float wa_s = 0.0f, wa_c = 1.0f, ws_s = 0.0f, ws_c = 1.0f;
        constexpr float lc = 1.0f, ls = 0.0f;
        if constexpr (kCompact) {
            asm volatile("" : "+v"(lane));
        } else {
            wa_s = wa_s0, wa_c = wa_c0, ws_s = ws_s0, ws_c = ws_c0;
            asm volatile("" : "+v"(lane), "+v"(wa_s), "+v"(wa_c), "+v"(ws_s), "+v"(ws_c));
        }
        Cursor nxt = cur;
        advance(nxt);
        const int fi = cur.fi;

        float xr[P], xi[P];
        const int voiced = tb.voiced[fi];
        const float ig = tb.inv_gain[fi];
        const float* apc = voiced ? ap_v : ap_u;    // aperiodic curve of the frame's class (uniform select)
        const float pvs = voiced ? 1.0f : 0.0f;     // periodic component only in voiced frames
        const float sgn_scale = ((lane & 1) ? -1.0f : 1.0f) * (0.5f / (float)M);
        if constexpr (!LERP) {
            // ---- aperiodic source in PAIRED layout: own bins k = lane + 64 q and their mirrors M - k, q < P/2 (one
            // evaluation of the split per pair), then the spectrum assembly (Appendix A2 steps 9-12) on the same pairs
            // with the features loaded ascending / descending, and the Hermitian merge without a second round of lane
            // exchanges (merge_paired_complex): against the per-bin form 2P fewer lane exchanges and half the split /
            // merge arithmetic per frame.
            constexpr int HP = P / 2;
            float no_r[HP], no_i[HP], nm_r[HP], nm_i[HP], nh_r, nh_i;
            if constexpr (SPEC) {
                const spec_f32x4* sp = reinterpret_cast<const spec_f32x4*>(tb.nspec + (long long)fi * kSpecFrameFloats) + lane;
                spec_f32x4 v[16];
#pragma unroll
                for (int s_ = 0; s_ < 16; ++s_) v[s_] = __builtin_nontemporal_load(sp + 64 * s_);
                const spec_f32x4 vh = __builtin_nontemporal_load(sp + 64 * 16);
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    no_r[4 * s_] = v[s_].x, no_r[4 * s_ + 1] = v[s_].y, no_r[4 * s_ + 2] = v[s_].z, no_r[4 * s_ + 3] = v[s_].w;
                    no_i[4 * s_] = v[4 + s_].x, no_i[4 * s_ + 1] = v[4 + s_].y, no_i[4 * s_ + 2] = v[4 + s_].z, no_i[4 * s_ + 3] = v[4 + s_].w;
                    nm_r[4 * s_] = v[8 + s_].x, nm_r[4 * s_ + 1] = v[8 + s_].y, nm_r[4 * s_ + 2] = v[8 + s_].z, nm_r[4 * s_ + 3] = v[8 + s_].w;
                    nm_i[4 * s_] = v[12 + s_].x, nm_i[4 * s_ + 1] = v[12 + s_].y, nm_i[4 * s_ + 2] = v[12 + s_].z, nm_i[4 * s_ + 3] = v[12 + s_].w;
                }
                nh_r = vh.x, nh_i = vh.y;
            } else {
                staged_wait<0>();
                noise_spectrum_paired<P, true, kCompact>(g, tb.wtype[fi], tw, xbuf, xbuf_byte, lane, wa_c, wa_s, no_r, no_i,
                                                         nm_r, nm_i, nh_r, nh_i, lc, ls);
            }
            if (P != 32) {   // FFT output lanes hold bins kappa(lane) + 64 q; everything below wants bins lane + 64 q
                const int src = kappa<P>(lane);
#pragma unroll
                for (int q = 0; q < HP; ++q) {
                    no_r[q] = __shfl(no_r[q], src);
                    no_i[q] = __shfl(no_i[q], src);
                    nm_r[q] = __shfl(nm_r[q], src);
                    nm_i[q] = __shfl(nm_i[q], src);
                }
                nh_r = __shfl(nh_r, src);
                nh_i = __shfl(nh_i, src);
            }
            const float* mrow = mag + (long long)fi * ld;
            const float* arow = real + (long long)fi * ld;
            const float* brow = imag + (long long)fi * ld;
            // Unvoiced frames have no periodic component (its mask is zero, magphase.py:873-876): the phase features and
            // the periodic curve are neither loaded nor used -- one wave-uniform branch per frame, 4 instead of 10 loads and
            // a third of the arithmetic per bin pair for about a third of the frames.
            float xh_r, xh_i;
            auto assemble_all = [&](auto voiced_tag) {
                constexpr bool V = decltype(voiced_tag)::value;
                auto assemble = [&](float m, float a, float b, float cpv, float cap, float n_r, float n_i, bool real_only,
                                    float& o_r, float& o_i) {
                    const float apf = m * cap * ig;
                    float vr, vi;
                    if (V) {
                        const float s = a * a + b * b;
                        const float u = (s > 0.0f) ? m * cpv * __builtin_amdgcn_rsqf(s) : 0.0f;
                        vr = fmaf(n_r, apf, a * u);
                        vi = fmaf(n_i, apf, b * u);
                    } else {
                        vr = n_r * apf;
                        vi = n_i * apf;
                    }
                    if (real_only) {   // DC and Nyquist: X = |X| (magphase.py:958-961)
                        vr = __builtin_sqrtf(vr * vr + vi * vi);
                        vi = 0.0f;
                    }
                    o_r = vr * sgn_scale;
                    o_i = vi * sgn_scale;
                };
#ifndef MPX_COMP_QB
#define MPX_COMP_QB (MPX_COMP_PAIR_WAVES > 8 ? 4 : 8)
#endif
#ifndef MPX_COMP_SADDR
#define MPX_COMP_SADDR 1
#endif
#ifndef MPX_COMP_NT
#define MPX_COMP_NT 1   // feature rows are read once, by one wave: non-temporal loads keep them from evicting the other
#endif                  // waves' lines (and any scratch lines) out of L2
                constexpr int QB = (HP < MPX_COMP_QB) ? HP : MPX_COMP_QB;   // pairs per batch: 10 QB loads in flight
                // bin M/2 (lane 0) first: every lane loads "its" bin M/2 + lane (a lane-0-only load becomes a scalar load with
                // an immediate wait, see feat_load_paired); only lane 0's value is used by the merge.  First, because the
                // split's nh_* then die here instead of living through the whole assembly; at 12 waves per CU the result
                // waits in the (idle) exchange buffer, not in two registers.
                {
                    const bool ph = V && M / 2 < n_per;
                    const unsigned bl = 4u * (unsigned)lane;
                    auto gh = [&](const float* row) {
                        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(row + M / 2) + (size_t)bl);
                    };
                    assemble(gh(mrow), ph ? gh(arow) : 0.0f, ph ? gh(brow) : 0.0f, ph ? gh(per_v) : 0.0f, gh(apc), nh_r, nh_i,
                             false, xh_r, xh_i);
                    if constexpr (kCompact) {
                        xbuf[lane] = xh_r;
                        xbuf[64 + lane] = xh_i;
                    }
                }
                // one batch = the bin pairs q0 .. q0 + QN - 1; OWN / MIR: their own / mirror bins may carry a periodic component
                auto batch = [&](auto q0_, auto qn_, auto own_, auto mir_) {
                    constexpr int Q0 = decltype(q0_)::value, QN = decltype(qn_)::value;
                    constexpr bool OWN = V && decltype(own_)::value, MIR = V && decltype(mir_)::value;
                    float m0[QN], d0[QN], m1[QN], d1[QN], a0[OWN ? QN : 1], b0[OWN ? QN : 1], c0[OWN ? QN : 1];
                    float a1[MIR ? QN : 1], b1[MIR ? QN : 1], c1[MIR ? QN : 1];
                    // keep each batch's loads where they are written (the compiler moves loads of read-only memory across
                    // plain barriers; hoisted above the noise spectrum they spill): the lane offset is laundered with a fake
                    // dependency on what has been produced so far
                    int lo = lane;
                    {
                        constexpr int c0_ = (Q0 == 0) ? 0 : ((Q0 >= 4) ? Q0 - 4 : 0), c1_ = (Q0 == 0) ? HP : Q0;
#pragma unroll
                        for (int c = c0_; c < c1_; c += 4)
                            asm volatile("" : "+v"(lo)
                                         : "v"(no_r[c]), "v"(no_r[c + 1]), "v"(no_r[c + 2]), "v"(no_r[c + 3]), "v"(no_i[c]),
                                           "v"(no_i[c + 1]), "v"(no_i[c + 2]), "v"(no_i[c + 3]), "v"(nm_r[c]), "v"(nm_r[c + 1]),
                                           "v"(nm_r[c + 2]), "v"(nm_r[c + 3]), "v"(nm_i[c]), "v"(nm_i[c + 1]), "v"(nm_i[c + 2]),
                                           "v"(nm_i[c + 3]));
                    }
                    // the row pointer (+ the constant part of the index) stays in SGPRs, the lane part is ONE zero-extended
                    // 32-bit byte offset (global_load_dword v, v, s[..] offset:imm); with int element indices every load got
                    // its own 64-bit address (v_lshl_add_u64 + a register pair: 208 of them in the kernel's ISA)
                    const unsigned blo = 4u * (unsigned)lo, bhi = 4u * (unsigned)(M - lo);
                    auto gl = [](const float* row, int kel, unsigned boff) {
#if MPX_COMP_SADDR && MPX_COMP_NT
                        return __builtin_nontemporal_load(
                            reinterpret_cast<const float*>(reinterpret_cast<const char*>(row + kel) + (size_t)boff));
#elif MPX_COMP_SADDR
                        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(row + kel) + (size_t)boff);
#else
                        return row[kel + (int)(boff >> 2)];
#endif
                    };
#pragma unroll
                    for (int jj = 0; jj < QN; ++jj) {
                        const int k = 64 * (Q0 + jj);
                        m0[jj] = gl(mrow, k, blo);
                        d0[jj] = gl(apc, k, blo);
                        m1[jj] = gl(mrow, -k, bhi);
                        d1[jj] = gl(apc, -k, bhi);
                        // the periodic curve is zero from bin n_per on (above the crossfade): the phase rows and the curve
                        // are read only for the register rows that reach below it (wave-uniform conditions)
                        // (the offset is laundered inside the conditional block: instruction selection works block by block
                        // and only recognises base + zext(offset) when the zero-extension sits in the same block)
                        if constexpr (OWN) {
                            a0[jj] = b0[jj] = c0[jj] = 0.0f;
                            if (k < n_per) {               // own bins lane + k
                                unsigned bq = blo;
                                asm volatile("" : "+v"(bq));
                                a0[jj] = gl(arow, k, bq);
                                b0[jj] = gl(brow, k, bq);
                                c0[jj] = gl(per_v, k, bq);
                            }
                        }
                        if constexpr (MIR) {
                            a1[jj] = b1[jj] = c1[jj] = 0.0f;
                            if (M - k - 63 < n_per) {      // mirrors M - lane - k
                                unsigned bq = bhi;
                                asm volatile("" : "+v"(bq));
                                a1[jj] = gl(arow, -k, bq);
                                b1[jj] = gl(brow, -k, bq);
                                c1[jj] = gl(per_v, -k, bq);
                            }
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < QN; ++jj) {
                        const int q = Q0 + jj;
                        const bool ends = (q == 0) && (lane == 0);   // the (DC, Nyquist) pair
                        assemble(m0[jj], OWN ? a0[OWN ? jj : 0] : 0.0f, OWN ? b0[OWN ? jj : 0] : 0.0f,
                                 OWN ? c0[OWN ? jj : 0] : 0.0f, d0[jj], no_r[q], no_i[q], ends, no_r[q], no_i[q]);
                        assemble(m1[jj], MIR ? a1[MIR ? jj : 0] : 0.0f, MIR ? b1[MIR ? jj : 0] : 0.0f,
                                 MIR ? c1[MIR ? jj : 0] : 0.0f, d1[jj], nm_r[q], nm_i[q], ends, nm_r[q], nm_i[q]);
                    }
                };
                using std::integral_constant;
                constexpr auto yes = std::true_type{};
                constexpr auto no = std::false_type{};
#ifndef MPX_COMP_QA
#define MPX_COMP_QA 8   // pairs per batch on the rows with phase (7 loads per pair); (QA, QR) = (8, 8) / (4, 8) / (4, 4) /
                        // (2, 8): 1.182 / 1.191 / 1.193 / 1.201 ms for the synthesis side of configs[2], none spills
#endif
#ifndef MPX_COMP_QR
#define MPX_COMP_QR 8   // pairs per batch on the rows without (4 loads per pair)
#endif
                if constexpr (NPQ >= 0 && P == 32 && NPQ == 8) {   // (4, 4, 8): rows 0-7 with the own bins' phase, rows 8-15 without
                    constexpr int QA = MPX_COMP_QA, QR = MPX_COMP_QR;
                    static_assert(8 % QA == 0 && 8 % QR == 0, "MPX_COMP_QA / MPX_COMP_QR divide 8");
                    batch(integral_constant<int, 0>{}, integral_constant<int, QA>{}, yes, no);
                    if constexpr (QA < 8) batch(integral_constant<int, QA>{}, integral_constant<int, QA>{}, yes, no);
                    if constexpr (QA < 4) batch(integral_constant<int, 2 * QA>{}, integral_constant<int, QA>{}, yes, no);
                    if constexpr (QA < 4) batch(integral_constant<int, 3 * QA>{}, integral_constant<int, QA>{}, yes, no);
                    batch(integral_constant<int, 8>{}, integral_constant<int, QR>{}, no, no);
                    if constexpr (QR < 8) batch(integral_constant<int, 8 + QR>{}, integral_constant<int, QR>{}, no, no);
                    if constexpr (QR < 4) batch(integral_constant<int, 8 + 2 * QR>{}, integral_constant<int, QR>{}, no, no);
                    if constexpr (QR < 4) batch(integral_constant<int, 8 + 3 * QR>{}, integral_constant<int, QR>{}, no, no);
                } else {
                    static_assert(NPQ < 0 || (P == 32 && NPQ == 8), "k_synth_comp_pair: NPQ schedules exist for P == 32, NPQ == 8");
                    static_assert(QB == 1 || QB == 2 || QB == 4 || QB == 8 || QB == 16, "MPX_COMP_QB");
                    // uniform batches of QB pairs (unrolled by hand: the batch sizes are template arguments)
                    if constexpr (HP / QB >= 1) batch(integral_constant<int, 0 * QB>{}, integral_constant<int, QB>{}, yes, yes);
                    if constexpr (HP / QB >= 2) batch(integral_constant<int, 1 * QB>{}, integral_constant<int, QB>{}, yes, yes);
                    if constexpr (HP / QB >= 3) batch(integral_constant<int, 2 * QB>{}, integral_constant<int, QB>{}, yes, yes);
                    if constexpr (HP / QB >= 4) batch(integral_constant<int, 3 * QB>{}, integral_constant<int, QB>{}, yes, yes);
                    if constexpr (HP / QB >= 5) batch(integral_constant<int, 4 * QB>{}, integral_constant<int, QB>{}, yes, yes);
                    if constexpr (HP / QB >= 6) batch(integral_constant<int, 5 * QB>{}, integral_constant<int, QB>{}, yes, yes);
                    if constexpr (HP / QB >= 7) batch(integral_constant<int, 6 * QB>{}, integral_constant<int, QB>{}, yes, yes);
                    if constexpr (HP / QB >= 8) batch(integral_constant<int, 7 * QB>{}, integral_constant<int, QB>{}, yes, yes);
                    static_assert(HP / QB <= 8, "more batches than the hand-unrolled schedule covers");
                }
            };
            if (voiced) assemble_all(std::true_type{});
            else assemble_all(std::false_type{});
            if constexpr (kCompact) {   // kappa(lane) == lane: the synthesis-side twiddle is the conjugate of the split's
                const float4 pk = tw_half_pad<P>(tw, lane);
                ws_c = pk.x;
                ws_s = -pk.y;
                xh_r = xbuf[lane];
                xh_i = xbuf[64 + lane];
            }
            merge_paired_complex<P>(no_r, no_i, nm_r, nm_i, xh_r, xh_i, xr, xi, lane, ws_c, ws_s);
        } else {
            // ---- aperiodic source: spectrum of this frame's windowed noise, bins k = lane + 64 j
            float nM;
            {
                staged_wait<0>();
                noise_spectrum<P, true, kCompact>(g, tb.wtype[fi], tw, xbuf, xbuf_byte, lane, wa_c, wa_s, xr, xi, nM, lc, ls);
                if (P != 32) {   // FFT output lanes hold bins kappa(lane)+64j; everything below wants bins lane+64j
                    const int src = kappa<P>(lane);
    #pragma unroll
                    for (int j = 0; j < P; ++j) {
                        xr[j] = __shfl(xr[j], src);
                        xi[j] = __shfl(xi[j], src);
                    }
                    nM = __shfl(nM, src);
                }
            }

            // ---- spectrum assembly (Appendix A2 steps 9-12) in place, HB register rows at a time: all loads of a batch are
            // issued together and branch-free (one memory latency per batch)
            const int r0 = LERP ? tb.row0[fi] : fi, r1 = LERP ? tb.row1[fi] : fi;
            const float rt = LERP ? tb.rowt[fi] : 0.0f;   // 0 when r0 == r1: the lerp below is then exact
            const float* m0p = mag + (long long)r0 * ld;
            const float* a0p = real + (long long)r0 * ld;
            const float* b0p = imag + (long long)r0 * ld;
            const float* m1p = mag + (long long)r1 * ld;
            const float* a1p = real + (long long)r1 * ld;
            const float* b1p = imag + (long long)r1 * ld;
            float xm = 0.0f;
    #ifndef MPX_COMP_FEAT_ROWS
    #define MPX_COMP_FEAT_ROWS 16
    #endif
            constexpr int HB = MPX_COMP_FEAT_ROWS;   // register rows per batch: 8 x HB loads in flight
    #pragma unroll
            for (int h = 0; h < P / HB; ++h) {
                float m0[HB], a0[HB], b0[HB], m1[HB], a1[HB], b1[HB], cpv[HB], cap[HB];
                // keep each half's loads where they are written: hoisted above the noise spectrum (or into the other
                // half) they cost 190 spilled registers
                // (the compiler moves loads of read-only memory across plain barriers: the lane offset is laundered with a
                // fake dependency on the last value produced before this half)
                int lo = lane;
                {
                    const int c0 = (h == 0) ? 0 : (h - 1) * HB, c1 = (h == 0) ? P : h * HB;   // everything produced so far
    #pragma unroll
                    for (int c = c0; c < c1; c += 8)
                        asm volatile("" : "+v"(lo)
                                     : "v"(xr[c]), "v"(xr[c + 1]), "v"(xr[c + 2]), "v"(xr[c + 3]), "v"(xr[c + 4]), "v"(xr[c + 5]),
                                       "v"(xr[c + 6]), "v"(xr[c + 7]), "v"(xi[c]), "v"(xi[c + 1]), "v"(xi[c + 2]), "v"(xi[c + 3]),
                                       "v"(xi[c + 4]), "v"(xi[c + 5]), "v"(xi[c + 6]), "v"(xi[c + 7]));
                }
    #pragma unroll
                for (int jj = 0; jj < HB; ++jj) {
                    const int k = 64 * (h * HB + jj);
                    m0[jj] = m0p[lo + k];
                    a0[jj] = a0p[lo + k];
                    b0[jj] = b0p[lo + k];
                    if (LERP) {
                        m1[jj] = m1p[lo + k];
                        a1[jj] = a1p[lo + k];
                        b1[jj] = b1p[lo + k];
                    }
                    cpv[jj] = per_v[lo + k];
                    cap[jj] = apc[lo + k];
                }
    #pragma unroll
                for (int jj = 0; jj < HB; ++jj) {
                    const int j = h * HB + jj;
                    // linear interpolation between constant-rate rows (magphase.py:2242-2252); rt == 0 for variable-rate input
                    const float m = LERP ? fmaf(m1[jj] - m0[jj], rt, m0[jj]) : m0[jj];
                    const float a = LERP ? fmaf(a1[jj] - a0[jj], rt, a0[jj]) : a0[jj];
                    const float b = LERP ? fmaf(b1[jj] - b0[jj], rt, b0[jj]) : b0[jj];
                    const float s = a * a + b * b;
                    const float u = (s > 0.0f) ? m * cpv[jj] * pvs * __builtin_amdgcn_rsqf(s) : 0.0f;
                    const float apf = m * cap[jj] * ig;
                    float vr = fmaf(xr[j], apf, a * u), vi = fmaf(xi[j], apf, b * u);
                    if (j == 0 && lane == 0) {   // DC: X = |X| (magphase.py:958-961)
                        vr = __builtin_sqrtf(vr * vr + vi * vi);
                        vi = 0.0f;
                    }
                    xr[j] = vr * sgn_scale;
                    xi[j] = vi * sgn_scale;
                }
            }
            if (lane == 0) {   // Nyquist bin: the noise spectrum is real there; X = |X|
                const float m = LERP ? fmaf(m1p[M] - m0p[M], rt, m0p[M]) : m0p[M];
                const float a = LERP ? fmaf(a1p[M] - a0p[M], rt, a0p[M]) : a0p[M];
                const float b = LERP ? fmaf(b1p[M] - b0p[M], rt, b0p[M]) : b0p[M];
                const float s = a * a + b * b;
                const float u = (s > 0.0f) ? m * per_v[M] * pvs * __builtin_amdgcn_rsqf(s) : 0.0f;
                const float apf = m * apc[M] * ig;
                const float vr = fmaf(nM, apf, a * u), vi = b * u;
                xm = __builtin_sqrtf(vr * vr + vi * vi) * (0.5f / (float)M);   // (-1)^M = +1
            }

            if constexpr (kCompact) {
                const float4 pk = tw_half_pad<P>(tw, lane);
                ws_c = pk.x;
                ws_s = -pk.y;
            }
            hermitian_merge<P>(xr, xi, xm, lane, ws_c, ws_s);
        }
        constexpr bool kDit = kCompact && MPX_COMP_DIT;
        if constexpr (kDit) {
            // DIT form: its input wants register brev(j) <- bin lane + 64 j, its output is register i <-> samples 2 n, 2 n + 1
            // with n = lane + 64 i: static renamings on both sides
            constexpr int LBJ = ilog2(P);
            float yr[P], yi[P];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                yr[brev(j, LBJ)] = xr[j];
                yi[brev(j, LBJ)] = xi[j];
            }
            const float4 pk = tw_half_pad<P>(tw, lane);
            wave_fft_dit_compact_front<P, +1>(yr, yi, tw, xbuf, lane, pk.z, pk.w);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                xr[j] = yr[j];
                xi[j] = yi[j];
            }
        } else if constexpr (kCompact) {
            const float4 pk = tw_half_pad<P>(tw, lane);
            wave_fft_front_compact<P, +1>(xr, xi, tw, xbuf, lane, pk.z, pk.w);
        } else {
            wave_fft_front<P, +1>(xr, xi, tw, xbuf, lane);
        }
        if (!SPEC && nxt.valid) {   // the exchange buffer is idle from here on: start the copy of the next frame's noise
            g = frame_geom(noise, tb.npos[nxt.fi], tb.nleft[nxt.fi], tb.nright[nxt.fi], N);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stage_samples_async(g, 0, kTile, xbuf_byte, lane);
        }
        if constexpr (kDit) wave_fft_dit_back<P, +1>(xr, xi);
        else fft_inreg<P, +1>(xr, xi);

        // ---- anti-ringing window (magphase.py:969-973, Q14): centred asymmetric Hann, zero outside
        const int wl = tb.win_l[fi], wr = tb.win_r[fi];
        const float inv_wl = (wl > 0) ? 1.0f / (float)wl : 1.0f;
        const float inv_wr = (wr > 0) ? 1.0f / (float)wr : 0.0f;
        const int kadd = (wl == 0) ? 1 : 0;
        const int n_lo = N / 2 - wl, n_hi = N / 2 + wr;   // support [n_lo, n_hi]

        // ---- ordered section: wait for this frame's ticket
        const RunDesc rd = runs[cur.ci];
        float* strip = strips + rd.strip_off;
        float* pcm0 = pcm + rd.out_base;
        const int ticket = cur.ticket_base + (fi - cur.fb);
        const int x = tb.pm_rel[fi] - cur.x0;   // strip position of the frame's first sample
        const int target = x & ~63;
        const int flushed = (fi == cur.fb) ? 0 : ((tb.pm_rel[fi - 1] - cur.x0) & ~63);
        asm volatile("" ::"s"(x), "s"(flushed), "s"(rd.head_end), "s"(rd.out_lo), "s"(rd.out_hi), "s"(rd.flush_end));
        while (__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ticket)
            __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        if (flushed < target) flush_ring<R>(ring, strip, pcm0, rd.head_end, rd.out_lo, rd.out_hi, flushed, target, lane);
        wave_sync();
        // register rows whose samples all lie outside the window support add nothing: skipped (wave-uniform)
        auto win_add = [&](float o, float v, int n) {
            const int ks = n - n_lo;
            const float w = (ks >= 0 && n <= n_hi) ? half_window(ks, wl, wl + wr, kadd, inv_wl, inv_wr, 0) : 0.0f;
            return fmaf(v, w, o);
        };
        auto row_live = [&](int q) { return !(128 * q + 127 < n_lo || 128 * q > n_hi); };
        if constexpr (kCompPairWaves > 8) {   // 16 ring values in registers at a time (<= 168 VGPRs)
#ifndef MPX_COMP_CH
#define MPX_COMP_CH 16
#endif
            constexpr int CH = (P < MPX_COMP_CH) ? P : MPX_COMP_CH;
            const RingAddr ra = ring_addr<P>(ring_byte, x, lane);
            ring_add_plane<P, 0, CH, kDit>(smem, ra, xr, lane, win_add, row_live);
            ring_add_plane<P, 1, CH, kDit>(smem, ra, xi, lane, win_add, row_live);
        } else {
            ring_add<P>(smem, ring_byte, x, xr, xi, lane, win_add, row_live);
        }
        wave_sync();
        if (fi == cur.fe - 1) {   // last frame of the run: stream out the rest, leave the ring cleared
            flush_ring<R>(ring, strip, pcm0, rd.head_end, rd.out_lo, rd.out_hi, target, rd.flush_end, lane);
            wave_sync();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __hip_atomic_store(turn, ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------
// Lossless analysis -> synthesis of the same frames in ONE launch (copy synthesis: analysis_lossless followed by
// synthesis_from_lossless, demos/demo_copy_synthesis_lossless.py:44-50; magphase.py:2869-2906, 1759-1776).  The wave that
// analysed frame f still holds X[k] for the bin pairs (k, M - k) the synthesis wants from it (PairFeat: lane l <-> bins
// l + 64 q and their mirrors): it writes the three feature rows -- the API's output -- and rebuilds the frame from the
// very float32 values it stored, so the 3 x 4 H bytes per frame that k_synth_ola_pair reads back from HBM (1.4 GB per
// 57 k frames: 0.14 J of the step's 0.7 J on a board that runs both kernels at its power limit) are not
// read at all.  Structure = k_synth_comp_pair without the spectrum assembly: frame geometry from the ANALYSIS tables
// (position / left / right of the epoch in the recording), samples staged by LDS-DMA into the exchange buffer while the
// previous frame overlap-adds, Hann halves, real FFT + split in the paired layout (noise_spectrum_paired), features
// (magphase.py:466-474) stored in k_analysis' line-friendly shape, unit-phase spectrum + Hermitian merge
// (feat_merge_paired: exactly what k_synth_ola_pair computes from the loaded values), inverse transform, overlap-add
// into the pair's LDS ring in frame order (runs / slots / tickets / head strips as k_synth_ola_pair; k_ola_fixup
// afterwards).  The output samples are what k_synth_ola_pair gives on the features this kernel wrote up to float32
// rounding (the same expressions; the compiler contracts their multiply-adds differently in the two kernels: measured
// 3.6e-7 of the signal peak); the features differ from k_analysis' in the last bits (DIT instead of DIF forward
// transform).  Both are held to the two-launch path's tolerances against the oracle.
// Every frame index of [0, n_frames) must belong to exactly one run: a frame outside the runs is not analysed.
// ---------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(kCompPairWaves * 64) void k_roundtrip_pair(const float* __restrict__ sig,
                                                                       const long long* __restrict__ fpos,
                                                                       const int* __restrict__ fleft,
                                                                       const int* __restrict__ fright,
                                                                       const RunDesc* __restrict__ runs,
                                                                       const int* __restrict__ slot_off,
                                                                       const int* __restrict__ slot_runs, int nslots,
                                                                       const int* __restrict__ pm_rel,
                                                                       const float* __restrict__ tw_g,
                                                                       float* __restrict__ omag, float* __restrict__ oreal,
                                                                       float* __restrict__ oimag, float* __restrict__ strips,
                                                                       float* __restrict__ pcm, long long ld) {
    constexpr int M = 64 * P, N = 2 * M, R = ring_len<P>(), HP = P / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tw = smem;
    const int lane_id = threadIdx.x & 63;
    const int wave = rfl((int)(threadIdx.x >> 6));
    const int pair = wave >> 1, half = wave & 1;
    constexpr bool kCompact = comp_compact<P>();
    float* xbuf = smem + comp_tw_floats<P>() + wave * comp_xbuf_floats<P>();
    const unsigned xbuf_byte = 4u * (unsigned)(comp_tw_floats<P>() + wave * comp_xbuf_floats<P>());
    constexpr int kRing0 = comp_tw_floats<P>() + kCompPairWaves * comp_xbuf_floats<P>();
    float* ring = smem + kRing0 + pair * R;
    const unsigned ring_byte = 4u * (unsigned)(kRing0 + pair * R);
    int* turn = reinterpret_cast<int*>(smem + kRing0 + kCompPairs * R) + pair;
    pair_kernel_prologue<P, kCompact, MPX_COMP_DIT != 0>(tw, tw_g, smem + kRing0, kCompPairs * R, turn - pair, kCompPairs,
                                                         kCompPairWaves * 64);

    float wa_s0, wa_c0, ws_s0, ws_c0;   // analysis-side lane twiddle W_N^kappa and synthesis-side conj(W_N^lane)
    sincospif(-2.0f * (float)kappa<P>(lane_id) / (float)N, &wa_s0, &wa_c0);
    sincospif(2.0f * (float)lane_id / (float)N, &ws_s0, &ws_c0);
    const int slot = blockIdx.x * kCompPairs + pair;
    if (slot >= nslots) return;

    typedef PairCursor Cursor;
    const int wi_end = slot_off[slot + 1];
    auto advance = [&](Cursor& c) { pair_cursor_advance(c, wi_end, half, runs, slot_runs); };
    Cursor cur;
    cur.wi = slot_off[slot];
    cur.ticket_base = 0;
    pair_cursor_settle(cur, wi_end, half, runs, slot_runs);
    if (!cur.valid) return;

    constexpr int kTile = kCompact ? 32 * P : 64 * P;
    FrameGeom g = frame_geom(sig, fpos[cur.fi], fleft[cur.fi], fright[cur.fi], N);
    stage_samples_async(g, 0, kTile, xbuf_byte, lane_id);

    while (cur.valid) {
        int lane = lane_id;
        float wa_s = 0.0f, wa_c = 1.0f, ws_s = 0.0f, ws_c = 1.0f;
        constexpr float lc = 1.0f, ls = 0.0f;
        if constexpr (kCompact) {
            asm volatile("" : "+v"(lane));
        } else {
            wa_s = wa_s0, wa_c = wa_c0, ws_s = ws_s0, ws_c = ws_c0;
            asm volatile("" : "+v"(lane), "+v"(wa_s), "+v"(wa_c), "+v"(ws_s), "+v"(ws_c));
        }
        Cursor nxt = cur;
        advance(nxt);
        const int fi = cur.fi;

        float xr[P], xi[P];
        {
            // ---- analysis: X[k] of the own bins k = lane + 64 q and of their mirrors M - k, bin M/2 on lane 0
            float no_r[HP], no_i[HP], nm_r[HP], nm_i[HP], nh_r, nh_i;
            MPX_MARK("frame_setup");
            staged_wait<0>();
            noise_spectrum_paired<P, true, kCompact>(g, 0, tw, xbuf, xbuf_byte, lane, wa_c, wa_s, no_r, no_i, nm_r, nm_i, nh_r,
                                                     nh_i, lc, ls);
            if (P != 32) {   // FFT output lanes hold bins kappa(lane) + 64 q; the rows and the merge want bins lane + 64 q
                const int src = kappa<P>(lane);
#pragma unroll
                for (int q = 0; q < HP; ++q) {
                    no_r[q] = __shfl(no_r[q], src);
                    no_i[q] = __shfl(no_i[q], src);
                    nm_r[q] = __shfl(nm_r[q], src);
                    nm_i[q] = __shfl(nm_i[q], src);
                }
                nh_r = __shfl(nh_r, src);
                nh_i = __shfl(nh_i, src);
            }
            // ---- per bin pair q: lossless features (magphase.py:466-474; as k_analysis: X == 0 -> all three 0), their
            // stores, and the pair's step of the Hermitian merge -- feat_merge_paired's arithmetic on the values just
            // stored (X = mag (R + jI) / |R + jI|, magphase.py:1761-1766), pair by pair so that a pair's four inputs die
            // as its four outputs appear (all 99 features at once, as k_synth_ola_pair holds them: 49 spilled registers)
            mpx_pin(no_r), mpx_pin(no_i), mpx_pin(nm_r), mpx_pin(nm_i);
            MPX_MARK("features_stores_merge");
            auto feat = [](float x_r, float x_i, float& m, float& a, float& b) {
                const float s2 = x_r * x_r + x_i * x_i;
                const float r = __builtin_amdgcn_rsqf(fmaxf(s2, 1.0e-37f));
                m = s2 * r;
                a = x_r * r;
                b = x_i * r;
            };
            if constexpr (kCompact) {   // kappa(lane) == lane: the synthesis-side twiddle is the conjugate of the split's
                const float4 pk = tw_half_pad<P>(tw, lane);
                ws_c = pk.x;
                ws_s = -pk.y;
            }
            const bool lane0 = (lane == 0);
            // the three rows in k_analysis' store shape: ascending 256-byte blocks of the own bins in natural q order; the
            // mirrors of step q regrouped into aligned blocks [M - 64 q - 64, M - 64 q - 1] whose lowest float is lane 0's
            // mirror of step q + 1 (bin M/2 for the last block); bin M alone from lane 0
            float* row_m = omag + (long long)fi * ld;
            float* row_r = oreal + (long long)fi * ld;
            float* row_i = oimag + (long long)fi * ld;
            float* mlo = row_m + lane;
            float* rlo = row_r + lane;
            float* ilo = row_i + lane;
            const int hoff = lane0 ? M - 64 : M - lane;
            float* mhi = row_m + hoff;
            float* rhi = row_r + hoff;
            float* ihi = row_i + hoff;
            float zr[HP], zi[HP];   // Z[M - k]
            float hm = 0.0f, ha = 0.0f, hb = 0.0f;   // mirror features of the previous step
#pragma unroll
            for (int q = 0; q < HP; ++q) {
                float m, a, b, mq, aq, bq;
                feat(no_r[q], no_i[q], m, a, b);
                feat(nm_r[q], nm_i[q], mq, aq, bq);
                mlo[64 * q] = m;
                rlo[64 * q] = a;
                ilo[64 * q] = b;
                if (q == 0) {
                    if (lane0) {
                        row_m[M] = mq;
                        row_r[M] = aq;
                        row_i[M] = bq;
                    }
                } else {
                    mhi[-64 * (q - 1)] = lane0 ? mq : hm;
                    rhi[-64 * (q - 1)] = lane0 ? aq : ha;
                    ihi[-64 * (q - 1)] = lane0 ? bq : hb;
                }
                hm = mq;
                ha = aq;
                hb = bq;
                // feat_merge_paired, step j = q, on X itself: mag (R + jI) / |R + jI| of the three values just formed IS X up
                // to their float32 rounding (m = |X|^2 r, (a, b) = X r with r = rsq(|X|^2): the product differs from X by the
                // relative error of r, < 2.5e-7), so the magnitude, the second inverse square root and four products per bin
                // that feat_merge_paired spends on rebuilding it are not spent here.  The (-1)^k of the fftshift is a rotation
                // of the inverse transform's output by half its length -- a renaming of registers in the overlap-add below --
                // and the 0.5 / M of the inverse transform rides on the overlap-add's multiply-add: both exact.
                const float x_r = no_r[q], p_r = nm_r[q];
                float x_i = no_i[q], p_i = nm_i[q];
                if (q == 0) {   // DC and Nyquist: imaginary parts dropped (Q5)
                    x_i = lane0 ? 0.0f : x_i;
                    p_i = lane0 ? 0.0f : p_i;
                }
                const float er = x_r + p_r, ei = x_i - p_i, tr = x_r - p_r, ti = x_i + p_i;
                const float cq = cos2p<P>(q), sq = sin2p<P>(q);
                const float wr = ws_c * cq - ws_s * sq, wi = ws_c * sq + ws_s * cq;
                const float orr = wr * tr - wi * ti, oi = wr * ti + wi * tr;
                xr[q] = er - oi;
                xi[q] = ei + orr;
                zr[q] = er + oi;
                zi[q] = orr - ei;
            }
            float mH, aH, bH;
            feat(nh_r, nh_i, mH, aH, bH);
            mhi[-64 * (HP - 1)] = lane0 ? mH : hm;
            rhi[-64 * (HP - 1)] = lane0 ? aH : ha;
            ihi[-64 * (HP - 1)] = lane0 ? bH : hb;
            // bin M/2 (lane 0): Z = 2 conj(X); then the hand-over of Z[M - k] to the lanes that own those registers
            const float hr = 2.0f * nh_r, hi = -2.0f * nh_i;
            const int src_lane = (64 - lane) & 63;
#pragma unroll
            for (int r = HP; r < P; ++r) {
                const float pr = lane0 ? ((r == HP) ? hr : zr[P - r]) : zr[P - 1 - r];
                const float pi = lane0 ? ((r == HP) ? hi : zi[P - r]) : zi[P - 1 - r];
                xr[r] = __shfl(pr, src_lane);
                xi[r] = __shfl(pi, src_lane);
            }
        }
        mpx_pin(xr), mpx_pin(xi);
        MPX_MARK("fft_inverse");
        constexpr bool kDit = kCompact && MPX_COMP_DIT;
        if constexpr (kDit) {
            constexpr int LBJ = ilog2(P);
            float yr[P], yi[P];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                yr[brev(j, LBJ)] = xr[j];
                yi[brev(j, LBJ)] = xi[j];
            }
            const float4 pk = tw_half_pad<P>(tw, lane);
            wave_fft_dit_compact_front<P, +1>(yr, yi, tw, xbuf, lane, pk.z, pk.w);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                xr[j] = yr[j];
                xi[j] = yi[j];
            }
        } else if constexpr (kCompact) {
            const float4 pk = tw_half_pad<P>(tw, lane);
            wave_fft_front_compact<P, +1>(xr, xi, tw, xbuf, lane, pk.z, pk.w);
        } else {
            wave_fft_front<P, +1>(xr, xi, tw, xbuf, lane);
        }
        if (nxt.valid) {   // the exchange buffer is idle from here on: start the copy of the next frame's samples
            g = frame_geom(sig, fpos[nxt.fi], fleft[nxt.fi], fright[nxt.fi], N);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stage_samples_async(g, 0, kTile, xbuf_byte, lane);
        }
        if constexpr (kDit) wave_fft_dit_back<P, +1>(xr, xi);
        else fft_inreg<P, +1>(xr, xi);

        // ---- ordered section: wait for this frame's ticket
        mpx_pin(xr), mpx_pin(xi);
        MPX_MARK("ticket");
        const RunDesc rd = runs[cur.ci];
        float* strip = strips + rd.strip_off;
        float* pcm0 = pcm + rd.out_base;
        const int ticket = cur.ticket_base + (fi - cur.fb);
        const int x = pm_rel[fi] - cur.x0;   // strip position of the frame's first sample
        const int target = x & ~63;
        const int flushed = (fi == cur.fb) ? 0 : ((pm_rel[fi - 1] - cur.x0) & ~63);
        asm volatile("" ::"s"(x), "s"(flushed), "s"(rd.head_end), "s"(rd.out_lo), "s"(rd.out_hi), "s"(rd.flush_end));
        while (__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ticket)
            __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        MPX_MARK("flush");
        if (flushed < target) flush_ring<R>(ring, strip, pcm0, rd.head_end, rd.out_lo, rd.out_hi, flushed, target, lane);
        wave_sync();
        MPX_MARK("overlap_add");
        constexpr float kScale = 0.5f / (float)M;   // the inverse transform's scale, on the overlap-add's multiply-add
        auto plain_add = [](float o, float v, int) { return fmaf(v, kScale, o); };
        auto all_rows = [](int) { return true; };
        {   // 16 ring values in registers at a time (<= 168 VGPRs); rows taken half a transform apart: the fftshift
            constexpr int CH = (P < MPX_COMP_CH) ? P : MPX_COMP_CH;
            const RingAddr ra = ring_addr<P>(ring_byte, x, lane);
            ring_add_plane<P, 0, CH, kDit, true>(smem, ra, xr, lane, plain_add, all_rows);
            ring_add_plane<P, 1, CH, kDit, true>(smem, ra, xi, lane, plain_add, all_rows);
        }
        wave_sync();
        if (fi == cur.fe - 1) {   // last frame of the run: stream out the rest, leave the ring cleared
            flush_ring<R>(ring, strip, pcm0, rd.head_end, rd.out_lo, rd.out_hi, target, rd.flush_end, lane);
            wave_sync();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __hip_atomic_store(turn, ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        MPX_MARK("loop_end");
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------
// Mel warp on the matrix cores: out[F x nout] = ln-prologue(x)[F x H] . W^T[H x nout], v_mfma_f32_16x16x4_f32.
// One workgroup = 64 output frames x up to 64 outputs; the reduction runs over the H bins in chunks of 64.  Per chunk
// the 256 threads stage the prologue values and the W slab into LDS as [row][k] (k contiguous, row stride 68 floats:
// conflict-free dword writes with k across lanes, 16-byte fragment reads); wave w then owns the frames
// 16 w .. 16 w + 15 and every 16-wide column tile.  The MFMA sums over k in any order, so lane group g = lane >> 4
// takes k = 16 g + 4 q + {0..3} for the four instructions of step q: a lane's fragment is one ds_read_b128 per step.
// Instructions of the column tiles are interleaved (dependent-accumulator latency 40 cycles > issue 32).
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NT: 16-wide column tiles in use, ceil(nout / 16); INTERP: rows interpolated (row0 given); MODE: the job's prologue /
// epilogue at compile time (as a run-time value it was a scalar branch per staged element: a third of the kernel's
// instructions were SALU, and the kernel is bound by instruction issue)
template <int NT, bool INTERP, int MODE>
__device__ __forceinline__ void mel_warp_block(const WarpJob& job, float (*As)[kWarpKStride], float (*Ws)[kWarpKStride],
                                               long long* s_o0, long long* s_o1, float* s_rt, long long F, int H,
                                               const int* __restrict__ row0, const int* __restrict__ row1,
                                               const float* __restrict__ rowt, long long ld) {
    const long long f0 = (long long)blockIdx.x * kWarpTile;   // row tiles on x: no 65535 limit on the frame count
    if (f0 >= F) return;                                      // the jobs of a launch need not have the same frame count
    const int kk = threadIdx.x & 63, fq = threadIdx.x >> 6;   // staging roles: bin within the chunk, frame quarter
    const int wave = rfl((int)(threadIdx.x >> 6));
    const int li = kk & 15, g = kk >> 4;                      // fragment roles
    if ((MODE == 1 || MODE == 3) && job.voi) {   // phase streams are masked by the voicing (magphase.py:2527-2529): a tile without a
        // voiced frame is all zeros -- written as such, nothing read
        const int pred = (threadIdx.x < kWarpTile) && (job.voi[min(f0 + (long long)threadIdx.x, F - 1)] != 0.0f);
        if (!__syncthreads_or(pred)) {
            const long long n_el = min((long long)kWarpTile, F - f0) * job.nout;
            for (long long i = threadIdx.x; i < n_el; i += 256) job.out[f0 * job.nout + i] = 0.0f;
            return;
        }
    }
    if (threadIdx.x < kWarpTile) {
        const long long f = min(f0 + (long long)threadIdx.x, F - 1);
        s_o0[threadIdx.x] = (long long)(row0 ? row0[f] : (int)f) * ld;
        s_o1[threadIdx.x] = (long long)(row0 ? row1[f] : (int)f) * ld;
        s_rt[threadIdx.x] = row0 ? rowt[f] : 0.0f;
    }
    __syncthreads();
    // Two-level accumulation over the H bins.  The operands are log spectra (|v| ~ 10): one fp32 chain over 2049 terms
    // carries partial sums of that size and loses ~sqrt(2049) * 6e-7 = 3e-5 -- measured 4e-5 against a float64 product of
    // the same device features, the whole residual error of the compressed analysis once its FFT is float64.  So every
    // 64-bin chunk is summed in a fresh MFMA accumulator (small partial sums) and the 33 chunk sums are added up
    // separately: ~sqrt(33) * 6e-7.  (An error-free TwoSum of the totals costs 16 more registers: the kernel, capped at
    // 128 VGPRs for four workgroups per CU, spills.)
    f32x4 tot[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) tot[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // Staging roles: thread (c4, rr) handles the 4 consecutive bins k0 + 4 c4 .. + 3 of the rows rr + 16 p, p < 4: one
    // 16-byte load per row and operand (12 per thread and chunk instead of 48 dword loads: the kernel issued 19 M VMEM
    // instructions per launch) and one ds_write_b128 per row (16 lanes cover a row's 256 bytes: conflict-free).  Rows
    // are dense (pitch H floats): the loads are 4-byte aligned only, which global_load_dwordx4 allows.  Every load is
    // unconditional; the chunk that crosses the end of the row (H = 64 q + 1: the last bin alone) clamps per element.
    // The NEXT chunk's loads are issued right after this chunk's values are in LDS, so they fly behind the fragment
    // reads and the MFMAs.
    constexpr int LPR = kWarpKC / 4, RPT = 256 / LPR, NP = kWarpTile / RPT;   // lanes per row, rows per pass, passes
    constexpr int NPW = (16 * NT + RPT - 1) / RPT;                            // passes that touch a W row in use
    const int c4 = threadIdx.x % LPR, rr = threadIdx.x / LPR;
    static_assert(!kWarpSwizzle || RPT == 16, "swizzle: a staging thread's rows rr + 16 p share rr & 15");
    const int swz_w = warp_swz(rr & 15), swz_r = warp_swz(li);
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load from a 4-byte aligned address
    auto ld4 = [](const float* q) {
        const f32x4u v = *reinterpret_cast<const f32x4u*>(q);
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    float4 xv[NP], xw[NP], wv4[NP];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int fl = rr + RPT * p;
            const int k = k0 + 4 * c4;
#ifdef MPX_PROBE_WARP_NOLOAD   // ablation (tools/ab_bench.py): operands faked, no global loads in the chunk loop
            xv[p] = xw[p] = make_float4(1.0f + 0.001f * (float)(k + fl), 1.1f, 1.2f, 1.3f);
            wv4[p] = make_float4(0.001f * (float)fl, 0.002f, 0.003f, 0.004f);
            asm volatile("" : "+v"(xv[p].x), "+v"(xw[p].y), "+v"(wv4[p].z));
#else
            xv[p] = ld4(job.x + s_o0[fl] + k);
            if (INTERP) xw[p] = ld4(job.x + s_o1[fl] + k);
            if (p < NPW) wv4[p] = ld4(job.W + (long long)min(fl, job.nout - 1) * H + k);
#endif
        }
    };
    const int Hfull = H & ~(kWarpKC - 1);   // bins covered by whole chunks; the rest (one bin for H = 64 q + 1) below
    if (Hfull > 0) fetch(0);
    for (int k0 = 0; k0 < Hfull; k0 += kWarpKC) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int fl = rr + RPT * p;
            // No masking of the padding: frames past F repeat the last frame's rows (s_o0 / s_o1 are clamped) and W rows
            // past nout repeat the last row -- rows and columns of the product are independent, and the epilogue stores
            // neither (a select per staged element was 32 of the 180 instructions of this block).
            const float rt = INTERP ? s_rt[fl] : 0.0f;
            const float xin[4] = {xv[p].x, xv[p].y, xv[p].z, xv[p].w};
            const float xin1[4] = {xw[p].x, xw[p].y, xw[p].z, xw[p].w};
            float av[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = INTERP ? fmaf(xin1[e] - xin[e], rt, xin[e]) : xin[e];
                av[e] = warp_prologue(MODE, x);
            }
            const int cw = kWarpSwizzle ? 4 * (c4 ^ swz_w) : 4 * c4;   // (fl & 15 == rr & 15: RPT is 16 for 64-bin chunks)
            *reinterpret_cast<float4*>(&As[fl][cw]) = make_float4(av[0], av[1], av[2], av[3]);
            if (p < NPW) *reinterpret_cast<float4*>(&Ws[fl][cw]) = wv4[p];
        }
        __syncthreads();
        if (k0 + kWarpKC < Hfull) fetch(k0 + kWarpKC);
#ifdef MPX_PROBE_WARP_NOMFMA   // ablation: staging, barriers and fragment reads only
#define MPX_WARP_MFMA(a_, b_, c_) ((c_) + f32x4{(a_) * (b_), 0.0f, 0.0f, 0.0f})
#else
#define MPX_WARP_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32((a_), (b_), (c_), 0, 0, 0)
#endif
#pragma unroll
        for (int h = 0; h < kWarpKC / 64; ++h) {   // a fresh accumulator per 64 bins (two-level accumulation, above)
            const float* arow = &As[16 * wave + li][kWarpSwizzle ? 0 : 64 * h + 16 * g];
            f32x4 acc[NT];
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) acc[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // 4 k per lane group and step: one 16-byte read per fragment
                // column of the fragment: k = 16 g + 4 q .. + 3, i.e. chunk 4 g + q (swizzled: ^ swz(li), one v_xor per step)
                const int col = kWarpSwizzle ? 4 * (((4 * g) ^ swz_r) ^ q) : 64 * h + 16 * g + 4 * q;
                const float4 aq = *reinterpret_cast<const float4*>(arow + (kWarpSwizzle ? col : 4 * q));
                float4 bq[NT];
#pragma unroll
                for (int jt = 0; jt < NT; ++jt)
                    bq[jt] = *reinterpret_cast<const float4*>(&Ws[16 * jt + li][col]);
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) acc[jt] = MPX_WARP_MFMA(aq.x, bq[jt].x, acc[jt]);
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) acc[jt] = MPX_WARP_MFMA(aq.y, bq[jt].y, acc[jt]);
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) acc[jt] = MPX_WARP_MFMA(aq.z, bq[jt].z, acc[jt]);
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) acc[jt] = MPX_WARP_MFMA(aq.w, bq[jt].w, acc[jt]);
            }
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) tot[jt] += acc[jt];
        }
        __syncthreads();
    }
    // C: column li of tile jt, row 4 g + r of this wave's 16 frames.  The bins past the last whole chunk are added here,
    // one fmaf per bin and output (H = 64 q + 1 for every transform size: the Nyquist bin).
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long f = f0 + 16 * wave + 4 * g + r;
        if (f >= F) continue;
        const int fl = 16 * wave + 4 * g + r;
        for (int k = Hfull; k < H; ++k) {
            const float x0 = job.x[s_o0[fl] + k];
            const float x = INTERP ? fmaf(job.x[s_o1[fl] + k] - x0, s_rt[fl], x0) : x0;
            const float v = warp_prologue(MODE, x);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const int i = 16 * jt + li;
                tot[jt][r] = fmaf(v, job.W[(long long)min(i, job.nout - 1) * H + k], tot[jt][r]);
            }
        }
        const float vo = (MODE == 1 && job.voi) ? job.voi[f] : 1.0f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const int i = 16 * jt + li;
            if (i >= job.nout) continue;
            float y = tot[jt][r];
            y = warp_epilogue(MODE, y, vo);
            job.out[f * job.nout + i] = y;
        }
    }
}

// One launch for the three jobs (separate launches end in a half-empty last round of workgroups); the magnitude job
// (blockIdx.y == 0) and the two phase jobs get their own column-tile count (60 outputs -> 4 tiles, 45 -> 3: a quarter
// fewer MFMAs on two thirds of the workgroups).
// PHV: the phase jobs run on the variable-rate rows themselves (mode 3, no row interpolation, their own frame count);
// k_warp_phase_rows then interpolates their outputs to the constant rate.
template <int NTM, int NTP, bool INTERP, int MAGMODE, bool PHV>
#if MPX_WARP_KC == 64
__attribute__((amdgpu_waves_per_eu(4, 4)))   // <= 128 VGPRs: four workgroups (35 KB of LDS each) per CU, measured -5 %
#else
__attribute__((amdgpu_waves_per_eu(2, 2)))   // 128-bin chunks: 68 KB of LDS, two workgroups per CU
#endif
__global__ __launch_bounds__(256) void k_mel_warp_mfma(WarpJobs jobs, int H, const int* __restrict__ row0,
                                                       const int* __restrict__ row1, const float* __restrict__ rowt,
                                                       long long ld) {
    __shared__ __attribute__((aligned(16))) float As[kWarpTile][kWarpKStride];   // As[f][k]
    __shared__ __attribute__((aligned(16))) float Ws[kWarpTile][kWarpKStride];   // Ws[i][k]
    __shared__ long long s_o0[kWarpTile], s_o1[kWarpTile];   // element offsets of the two input rows of a frame
    __shared__ float s_rt[kWarpTile];
    if (blockIdx.y == 0)
        mel_warp_block<NTM, INTERP, MAGMODE>(jobs.j[0], As, Ws, s_o0, s_o1, s_rt, jobs.j[0].F, H, row0, row1, rowt, ld);
    else if (PHV)
        mel_warp_block<NTP, false, 3>(jobs.j[blockIdx.y], As, Ws, s_o0, s_o1, s_rt, jobs.j[blockIdx.y].F, H, nullptr, nullptr,
                                      nullptr, ld);
    else
        mel_warp_block<NTP, INTERP, 1>(jobs.j[blockIdx.y], As, Ws, s_o0, s_o1, s_rt, jobs.j[blockIdx.y].F, H, row0, row1, rowt,
                                       ld);
}

// Phase streams of the compressed analysis at the constant rate from their variable-rate warp (mode 3): row
// interpolation of the phase_dim outputs, then the epilogue of mode 1 (voicing mask, clip; magphase.py:2527-2532).
// The warp is linear up to its 1e-8 e^{-2x} floor term, so interpolating after it instead of before differs by < 1e-8 per
// bin; rows no voiced frame uses were not computed and are not read (the select discards them).
__global__ __launch_bounds__(256) void k_warp_phase_rows(const float* __restrict__ tr, const float* __restrict__ ti,
                                                         const int* __restrict__ row0, const int* __restrict__ row1,
                                                         const float* __restrict__ rowt, const float* __restrict__ voi,
                                                         long long F, int nout, float* __restrict__ out_r,
                                                         float* __restrict__ out_i) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= F * nout) return;
    const long long f = idx / nout;
    const int i = (int)(idx - f * nout);
    const float vo = voi[f];
    float yr = 0.0f, yi = 0.0f;
    if (vo != 0.0f) {
        const long long a = (long long)row0[f] * nout + i, b = (long long)row1[f] * nout + i;
        const float t = rowt[f];
        yr = warp_epilogue(1, fmaf(tr[b] - tr[a], t, tr[a]), vo);
        yi = warp_epilogue(1, fmaf(ti[b] - ti[a], t, ti[a]), vo);
    }
    out_r[idx] = yr;
    out_i[idx] = yi;
}

// ---------------------------------------------------------------------------------------------
// Mel unwarp on the matrix cores: out[F x H] = op(A[F x K] . U[K x H]) with v_mfma_f32_32x32x2_f32 (f32 in, f32
// accumulate: bit-for-bit an fmaf chain in k order, so the result equals the VALU form's).  One wave = one task =
// 32 frames x kUnwarpColTiles column tiles of 32 bins.  Nothing is staged in LDS: A (F x K, a few MB) and U (K x H,
// < 0.5 MB) are L2 resident and a lane's fragment element is one dword either way; the kernel is bound by the
// 12 H bytes per frame it writes.  A fragment: lane l holds A[f0 + (l & 31)][2t + (l >> 5)] for t < KH (kept for
// the whole task); B fragment: U[2t + (l >> 5)][j0 + (l & 31)], the next tile's loads in flight behind this
// tile's KH MFMAs; C: column j0 + (l & 31), row (r & 3) + 8 (r >> 2) + 4 (l >> 5) of register r.
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef MPX_UNWARP_COL_TILES
#define MPX_UNWARP_COL_TILES 16
#endif
constexpr int kUnwarpColTiles = MPX_UNWARP_COL_TILES;   // 512 bins per task

// MODE 0: one output row per row of A.  MODE 1 / 2 (constant -> variable frame rate, magphase.py:2242-2252 folded in):
// output row f is the interpolation between rows row0[f] and row1[f] of the unwarped A with weight rowt[f] --
//   MODE 1 (linear maps, op 0): the coefficient rows are interpolated, out = U . lerp(A[r0], A[r1]) (the unwarp is
//          linear: equal to lerp(U A[r0], U A[r1]) up to rounding);
//   MODE 2 (op 1, exp): both products are formed (the A fragments of r0 and r1 share every B fragment) and the
//          interpolation runs on the exponentials, lerp(exp(U A[r0]), exp(U A[r1])), as the reference does on m_mag.
// The synthesis kernel then reads ONE row per frame instead of two, and the spectra are written at the variable rate only.
template <int KH, int MODE>
__global__ __launch_bounds__(256) void k_mel_unwarp_mfma(UnwarpJobs jobs, int job0, long long F, int H,
                                                         int col_parts, long long n_tasks, int ld,
                                                         const int* __restrict__ row0, const int* __restrict__ row1,
                                                         const float* __restrict__ rowt, const int* __restrict__ voiced,
                                                         int Hc) {   // Hc <= H: output columns produced
    const UnwarpJob job = jobs.j[job0 + blockIdx.y];
    const int lane = threadIdx.x & 63;
    const long long task = (long long)blockIdx.x * 4 + rfl((int)(threadIdx.x >> 6));   // wave-uniform, and known to be
    if (task >= n_tasks) return;
    // row tile fastest: the 4 waves of a workgroup work on the same columns at the same time and share their B
    // fragments (the U slab) through the vector L1 -- with the column part fastest the kernel is bound by L2 -> L1 traffic
    const long long row_tiles = n_tasks / col_parts;
    const int cp = (int)(task / row_tiles);
    const long long rt = task - cp * row_tiles;
    const long long f0 = rt * 32;
    const int K = job.K;
    const int kk = lane >> 5, li = lane & 31;
    if (MODE == 1 && voiced) {   // phase rows of unvoiced frames are never read: skip tiles without a voiced frame
        if (!__any(voiced[min(f0 + li, F - 1)] != 0)) return;
    }

    float a[KH];
    float a1[MODE == 2 ? KH : 1];   // MODE 2: fragment of the second row
    float tr[MODE == 2 ? 16 : 1];   // MODE 2: interpolation weight of the output row of accumulator register r
    {
        const long long f = min(f0 + li, F - 1);   // rows past F are computed on a copy of the last row, never stored
        const float* arow = job.A + (MODE == 0 ? f : (long long)row0[f]) * K;
        const float* arow1 = (MODE == 0) ? arow : job.A + (long long)row1[f] * K;
        const float wt = (MODE == 0) ? 0.0f : rowt[f];
#pragma unroll
        for (int t = 0; t < KH; ++t) {
            const int k = 2 * t + kk;
            a[t] = arow[min(k, K - 1)];
            a[t] = (k < K) ? a[t] : 0.0f;
            if (MODE != 0) {
                float x1 = arow1[min(k, K - 1)];
                x1 = (k < K) ? x1 : 0.0f;
                if (MODE == 1) a[t] = fmaf(x1 - a[t], wt, a[t]);
                else a1[t] = x1;
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[r] = rowt[min(f0 + (r & 3) + 8 * (r >> 2) + 4 * kk, F - 1)];
        }
    }
    const int jbeg = cp * (kUnwarpColTiles * 32);
    const int jend = min(Hc, jbeg + kUnwarpColTiles * 32);
    if (jbeg >= jend) return;
    // Two column tiles (64 bins) per step: their stores go out back to back, so every row gets 256 contiguous bytes at
    // once.  The B fragments are refilled for the NEXT step right behind the MFMA that consumed them (a whole step of
    // MFMA time for the L2 round trip, no second register set).
    float b0[KH], b1[KH];
    // U[2t + kk][col] through a buffer descriptor: wave-uniform scalar offset 2t H (bytes) + one per-lane offset
    // (kk H + col) -- no per-t address registers; rows >= K fall outside the descriptor and read as 0 (their A element
    // is 0 anyway), columns >= H read the next row's start and are never stored.
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(job.U), 0, K * H * 4, 0x00020000);
    const int row2 = 8 * H;   // bytes between rows 2t and 2t + 2
    // MODE 1 is the linear (phase) jobs, MODE 2 the exp (magnitude) job: known at compile time (as a run-time flag both
    // values are computed and selected for every output element)
    const bool op_exp = (MODE == 1) ? false : ((MODE == 2) ? true : (rfl(job.op) != 0));
    const int nrows = (int)min((long long)32, F - f0);
    const int HO = ld;   // output row pitch
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(job.out + f0 * HO, 0, nrows * HO * 4, 0x00020000);
    // rows k >= K (K odd, or K/2 rounded up to the even KH) multiply a zero of A but must still read finite memory
    // inside U: 2t >= K reads rows 0/1 (scalar select), 2t + 1 == K reads row 2t in both half-waves.
    auto soff = [&](int t) { return (2 * t < K) ? t * row2 : 0; };
    {
        const int v0 = 4 * (kk * H + jbeg + li), v0e = 4 * (jbeg + li);
#pragma unroll
        for (int t = 0; t < KH; ++t) {
            const int v = (2 * t + 1 == K) ? v0e : v0;
            b0[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(urs, v, soff(t), 0));
            b1[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(urs, v + 128, soff(t), 0));
        }
    }
    const bool full_rows = nrows == 32;
    for (int j0 = jbeg; j0 < jend; j0 += 64) {
        const int vn0 = 4 * (kk * H + j0 + 64 + li), vn0e = 4 * (j0 + 64 + li);
        f32x16 acc0, acc1, acc2, acc3;   // acc2 / acc3: the second row's products (MODE 2)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = acc3[r] = 0.0f;
#pragma unroll
        for (int t = 0; t < KH; ++t) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b0[t], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b1[t], acc1, 0, 0, 0);
            if (MODE == 2) {
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc3, 0, 0, 0);
            }
            const int v = (2 * t + 1 == K) ? vn0e : vn0;
            b0[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(urs, v, soff(t), 0));
            b1[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(urs, v + 128, soff(t), 0));
        }
        // (Forming the product transposed -- lane = frame, four consecutive bins in four registers, one 16-byte store per
        // lane and 8 instead of 32 store instructions per tile -- was measured: 0.61 -> 0.74 ms.  A row then gets 32 bytes
        // per instruction; what this memory system rewards is whole 128-byte lines per instruction, cf. k_analysis.)
        // Stores through a descriptor of this task's output rows: one wave-uniform scalar offset per accumulator
        // register, no address arithmetic.  The bounds check of a raw buffer does not see the scalar offset, so the
        // rows >= F of the last row tile are masked through the per-lane offset (out of range = dropped), like the
        // columns >= H of the last column part.
        const int c0 = j0 + li, c1 = c0 + 32;
        const int vo0 = (c0 < jend) ? 4 * (4 * kk * HO + c0) : 0x7ffffff0;
        const int vo1 = (c1 < jend) ? 4 * (4 * kk * HO + c1) : 0x7ffffff0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2);
            const int so = row * (4 * HO);
            const bool ok = full_rows || (row + 4 * kk < nrows);
            float v0 = op_exp ? __expf(acc0[r]) : acc0[r];
            float v1 = op_exp ? __expf(acc1[r]) : acc1[r];
            if (MODE == 2) {   // fmaf(m1 - m0, t, m0): the form the synthesis kernel used on the two rows
                const float w0 = op_exp ? __expf(acc2[r]) : acc2[r];
                const float w1 = op_exp ? __expf(acc3[r]) : acc3[r];
                v0 = fmaf(w0 - v0, tr[r], v0);
                v1 = fmaf(w1 - v1, tr[r], v1);
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), ors, ok ? vo0 : 0x7ffffff0, so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), ors, ok ? vo1 : 0x7ffffff0, so, 0);
        }
    }
}

struct UnwarpRows {   // constant -> variable rate interpolation tables (all null: one output row per input row)
    const int* row0;
    const int* row1;
    const float* rowt;
    const int* voiced;   // optional (MODE 1): a tile of 32 frames none of which is voiced is skipped -- the synthesis does
                         // not read the phase rows of unvoiced frames
    int phase_cols;      // MODE 1: only the first phase_cols bins of the phase rows are produced (0: all)
};

template <int KH>
static int launch_unwarp_mfma(hipStream_t s, const UnwarpJobs& jobs, int job0, int njobs, long long F, int H, int ld,
                              const UnwarpRows& rw) {
    // the phase rows are only needed below the periodic / aperiodic crossfade (rw.phase_cols, rounded up to a 64-bin step)
    const bool phase = rw.row0 && jobs.j[job0].op == 0;
    const int Hc = (phase && rw.phase_cols > 0) ? min(H, (rw.phase_cols + 63) / 64 * 64) : H;
    const int col_parts = (Hc + kUnwarpColTiles * 32 - 1) / (kUnwarpColTiles * 32);
    const long long n_tasks = ((F + 31) / 32) * col_parts;
    const dim3 grid((unsigned)((n_tasks + 3) / 4), (unsigned)njobs);
    if (!rw.row0)
        hipLaunchKernelGGL((k_mel_unwarp_mfma<KH, 0>), grid, dim3(256), 0, s, jobs, job0, F, H, col_parts, n_tasks, ld,
                           rw.row0, rw.row1, rw.rowt, rw.voiced, Hc);
    else if (phase)
        hipLaunchKernelGGL((k_mel_unwarp_mfma<KH, 1>), grid, dim3(256), 0, s, jobs, job0, F, H, col_parts, n_tasks, ld,
                           rw.row0, rw.row1, rw.rowt, rw.voiced, Hc);
    else
        hipLaunchKernelGGL((k_mel_unwarp_mfma<KH, 2>), grid, dim3(256), 0, s, jobs, job0, F, H, col_parts, n_tasks, ld,
                           rw.row0, rw.row1, rw.rowt, rw.voiced, Hc);
    return MPX_OK;
}

static int dispatch_unwarp_mfma(hipStream_t s, const UnwarpJobs& jobs, int job0, int njobs, int K, long long F, int H,
                                int ld, const UnwarpRows& rw) {
    switch ((K + 3) / 4) {   // KH = K/2 rounded up to even: 16 instantiations cover K <= 64
#define MPX_UNWARP_CASE(q) case q: return launch_unwarp_mfma<2 * q>(s, jobs, job0, njobs, F, H, ld, rw);
        MPX_UNWARP_CASE(1) MPX_UNWARP_CASE(2) MPX_UNWARP_CASE(3) MPX_UNWARP_CASE(4) MPX_UNWARP_CASE(5) MPX_UNWARP_CASE(6)
        MPX_UNWARP_CASE(7) MPX_UNWARP_CASE(8) MPX_UNWARP_CASE(9) MPX_UNWARP_CASE(10) MPX_UNWARP_CASE(11) MPX_UNWARP_CASE(12)
        MPX_UNWARP_CASE(13) MPX_UNWARP_CASE(14) MPX_UNWARP_CASE(15) MPX_UNWARP_CASE(16)
#undef MPX_UNWARP_CASE
    }
    return fail(MPX_ERR_ARG, "mpx_mel_unwarp: coefficient count must be in 1..64%s");
}

// Magnitude unwarp with the constant -> variable rate interpolation, ONE product per constant-rate row (MODE 2 above
// forms two per variable-rate frame: adjacent frames share their rows, 44 % of its MFMAs are repeats).  A task = 32
// constant-rate rows (tiles advance by 31 rows: a frame's two rows are adjacent, so both lie in the tile its row0 falls
// into) x kUnwarpColTiles column tiles.  Per 64-bin step the exp'd 32 x 64 tile goes through a per-wave LDS buffer
// (row stride 72 floats: the accumulator's two half-waves write rows a and a + 4, 32 banks apart) and every
// variable-rate frame of the tile -- [tile_first[T], tile_first[T + 1]), planned on the host -- reads its two rows
// back (lane = bin: 256 contiguous bytes), interpolates with fmaf(m1 - m0, t, m0) and stores its 256-byte row segment.
// Same values as MODE 2 (each row's product is the same fmaf chain).
#ifndef MPX_UNWARP_QUADS
#define MPX_UNWARP_QUADS 1
#endif
constexpr int kTileRows = 31;          // new constant-rate rows per tile
constexpr int kTileStride = 72;        // floats per LDS row

// (Dropping the two special cases for U rows past K when K == 2 KH -- fewer instructions -- made the kernel SLOWER,
// 271 -> 350 us: the selects space the B-fragment loads out between the MFMAs; without them the compiler batches them.)
template <int KH>
__global__ __launch_bounds__(256) void k_mel_unwarp_tiled(UnwarpJob job, long long F, long long n_rows, int H,
                                                          int col_parts, long long n_tasks, int ld,
                                                          const int* __restrict__ row0, const int* __restrict__ row1,
                                                          const float* __restrict__ rowt,
                                                          const int* __restrict__ tile_first) {
    __shared__ __attribute__((aligned(16))) float tiles[4][32 * kTileStride];
    __shared__ __attribute__((aligned(16))) float4 tabs[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = rfl((int)(threadIdx.x >> 6));
    const long long task = (long long)blockIdx.x * 4 + wave;
    if (task >= n_tasks) return;
    const long long row_tiles = n_tasks / col_parts;
    const int cp = (int)(task / row_tiles);
    const long long T = task - cp * row_tiles;
    const long long rb = T * kTileRows;
    const int K = job.K;
    const int kk = lane >> 5, li = lane & 31;
    float* tile = tiles[wave];
    float4* tab = tabs[wave];

    float a[KH];
    {
        const float* arow = job.A + min(rb + li, n_rows - 1) * K;
#pragma unroll
        for (int t = 0; t < KH; ++t) {
            const int k = 2 * t + kk;
            a[t] = arow[min(k, K - 1)];
            a[t] = (k < K) ? a[t] : 0.0f;
        }
    }
    const int fa = rfl(tile_first[T]), fb = rfl(tile_first[T + 1]);
    const int jbeg = cp * (kUnwarpColTiles * 32);
    const int jend = min(H, jbeg + kUnwarpColTiles * 32);
    if (jbeg >= jend || fa >= fb) return;
    float b0[KH], b1[KH];
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(job.U), 0, K * H * 4, 0x00020000);
    const int row2 = 8 * H;
    constexpr bool op_exp = true;   // the tiled form is the magnitudes' (exp epilogue)
    auto soff = [&](int t) { return (2 * t < K) ? t * row2 : 0; };
    {
        const int v0 = 4 * (kk * H + jbeg + li), v0e = 4 * (jbeg + li);
#pragma unroll
        for (int t = 0; t < KH; ++t) {
            const int v = (2 * t + 1 == K) ? v0e : v0;
            b0[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(urs, v, soff(t), 0));
            b1[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(urs, v + 128, soff(t), 0));
        }
    }
    for (int j0 = jbeg; j0 < jend; j0 += 64) {
        const int vn0 = 4 * (kk * H + j0 + 64 + li), vn0e = 4 * (j0 + 64 + li);
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
#pragma unroll
        for (int t = 0; t < KH; ++t) {
#ifdef MPX_PROBE_UNWARP_NOMFMA   // ablation (timing only): one VALU op per step instead of the two matrix instructions
            acc0[t & 15] = fmaf(a[t], b0[t], acc0[t & 15]);
            acc1[t & 15] = fmaf(a[t], b1[t], acc1[t & 15]);
#else
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b0[t], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b1[t], acc1, 0, 0, 0);
#endif
#ifndef MPX_PROBE_UNWARP_NOBLOAD   // ablation: the first step's U fragments for every step
            const int v = (2 * t + 1 == K) ? vn0e : vn0;
            b0[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(urs, v, soff(t), 0));
            b1[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(urs, v + 128, soff(t), 0));
#else
            asm volatile("" : "+v"(b0[t]), "+v"(b1[t]) : "s"(vn0 + vn0e));
#endif
        }
        wave_sync();   // the previous step's readers are done with the tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
            tile[row * kTileStride + li] = op_exp ? __expf(acc0[r]) : acc0[r];
            tile[row * kTileStride + 32 + li] = op_exp ? __expf(acc1[r]) : acc1[r];
        }
        wave_sync();
        const bool col_ok = j0 + lane < jend;
#ifdef MPX_PROBE_UNWARP_NOINTERP   // ablation (timing only): no interpolation / store phase
        if (tile[lane] == 123.456f) job.out[lane] = 1.0f;
        const int fb_ = fa;
#else
        const int fb_ = fb;
#endif
        for (int fc = fa; fc < fb_; fc += 64) {
            // lane i writes the tables of frame fc + i (tile offsets of its two rows, weight) into a small per-wave LDS
            // table; the loop below reads one entry per frame as a broadcast -- no v_readlane / SGPR round trips (that
            // form spent its time in scalar hazards: 26 M SALU instructions per launch)
            const int fl = min(fc + lane, fb - 1);
            const int r_ = row0[fl] - (int)rb;
            const int d_ = row1[fl] - row0[fl];
            wave_sync();
            tab[lane] = make_float4(__builtin_bit_cast(float, r_ * kTileStride), __builtin_bit_cast(float, (r_ + d_) * kTileStride),
                                    rowt[fl], 0.0f);
            wave_sync();
            const int cnt = min(64, fb - fc);
#if MPX_UNWARP_QUADS
            // FOUR frames per wave instruction (round 5): lane = (frame i + (lane >> 4), bin quad lane & 15) reads 16 bytes of
            // each of its two rows (a 16-lane group reads one row's 256 contiguous bytes: conflict-free), interpolates four
            // bins and stores them as one 16-byte store -- a quarter of the LDS reads and store instructions of the
            // lane-per-bin form, the same values.  Rows are ld floats apart (ld a multiple of 32: 16-byte aligned); the quad
            // that holds bin H - 1 writes up to three floats of the row's padding.
            const int fq = lane >> 4, q4 = 4 * (lane & 15);
            const bool q_ok = j0 + q4 < jend;
            (void)col_ok;
            float* obase = job.out + (long long)fc * ld + j0 + q4;
            for (int i = 0; i < cnt; i += 8) {   // two batches of four frames in flight
                float4 e[2], m0[2], m1[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    e[u] = tab[min(i + 4 * u + fq, cnt - 1)];
                    m0[u] = *reinterpret_cast<const float4*>(tile + __builtin_bit_cast(int, e[u].x) + q4);
                    m1[u] = *reinterpret_cast<const float4*>(tile + __builtin_bit_cast(int, e[u].y) + q4);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int f = i + 4 * u + fq;
                    const float w = e[u].z;
                    float4 v;
                    v.x = fmaf(m1[u].x - m0[u].x, w, m0[u].x);
                    v.y = fmaf(m1[u].y - m0[u].y, w, m0[u].y);
                    v.z = fmaf(m1[u].z - m0[u].z, w, m0[u].z);
                    v.w = fmaf(m1[u].w - m0[u].w, w, m0[u].w);
                    if (q_ok && f < cnt) *reinterpret_cast<float4*>(obase + (long long)f * ld) = v;
                }
            }
#else
            float* orow = job.out + (long long)fc * ld + j0 + lane;
            for (int i = 0; i < cnt; i += 4) {   // four frames per iteration: their LDS reads are in flight together
                float m0[4], m1[4], w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 e = tab[min(i + u, cnt - 1)];
                    m0[u] = tile[__builtin_bit_cast(int, e.x) + lane];
                    m1[u] = tile[__builtin_bit_cast(int, e.y) + lane];
                    w[u] = e.z;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#ifdef MPX_PROBE_UNWARP_NOSTORE    // ablation: everything but the global stores
                    asm volatile("" ::"v"(fmaf(m1[u] - m0[u], w[u], m0[u])), "v"(orow));
#else
                    if (col_ok && i + u < cnt) orow[(long long)(i + u) * ld] = fmaf(m1[u] - m0[u], w[u], m0[u]);
#endif
                }
            }
#endif
        }
    }
}

template <int KH>
static int launch_unwarp_tiled(hipStream_t s, const UnwarpJob& job, long long F, long long n_rows, int H, int ld,
                               const UnwarpRows& rw, const int* tile_first) {
    const int col_parts = (H + kUnwarpColTiles * 32 - 1) / (kUnwarpColTiles * 32);
    const long long row_tiles = (n_rows + kTileRows - 1) / kTileRows;
    const long long n_tasks = row_tiles * col_parts;
    hipLaunchKernelGGL((k_mel_unwarp_tiled<KH>), dim3((unsigned)((n_tasks + 3) / 4)), dim3(256), 0, s, job, F, n_rows, H,
                       col_parts, n_tasks, ld, rw.row0, rw.row1, rw.rowt, tile_first);
    return MPX_OK;
}

static int dispatch_unwarp_tiled(hipStream_t s, const UnwarpJob& job, int K, long long F, long long n_rows, int H, int ld,
                                 const UnwarpRows& rw, const int* tile_first) {
    if (job.op != 1) return fail(MPX_ERR_ARG, "mpx_mel_unwarp_rows: the tiled form is the exp job's%s");
    if (MPX_UNWARP_QUADS && ((ld & 3) || (reinterpret_cast<uintptr_t>(job.out) & 15)))
        return fail(MPX_ERR_ARG, "mpx_mel_unwarp_rows: with tile_first the output rows must be 16-byte aligned (ld a multiple of 4: mpx_spec_ld)%s");
    switch ((K + 3) / 4) {
#define MPX_UNWARP_CASE(q) case q: return launch_unwarp_tiled<2 * q>(s, job, F, n_rows, H, ld, rw, tile_first);
        MPX_UNWARP_CASE(1) MPX_UNWARP_CASE(2) MPX_UNWARP_CASE(3) MPX_UNWARP_CASE(4) MPX_UNWARP_CASE(5) MPX_UNWARP_CASE(6)
        MPX_UNWARP_CASE(7) MPX_UNWARP_CASE(8) MPX_UNWARP_CASE(9) MPX_UNWARP_CASE(10) MPX_UNWARP_CASE(11) MPX_UNWARP_CASE(12)
        MPX_UNWARP_CASE(13) MPX_UNWARP_CASE(14) MPX_UNWARP_CASE(15) MPX_UNWARP_CASE(16)
#undef MPX_UNWARP_CASE
    }
    return fail(MPX_ERR_ARG, "mpx_mel_unwarp: coefficient count must be in 1..64%s");
}

}  // namespace mpx

using namespace mpx;

extern "C" {

int64_t mpx_spec_ld(int32_t n_bins) { return n_bins <= 0 ? 0 : ((int64_t)n_bins + 31) / 32 * 32; }

static int mel_unwarp_impl(void* stream, int64_t n_frames, int32_t n_bins, const float* a_mag, int32_t k_mag,
                           const float* u_mag, float* out_mag, const float* a_real, const float* a_imag, int32_t k_phase,
                           const float* u_phase, float* out_real, float* out_imag, int64_t ld, const UnwarpRows& rw,
                           int64_t n_rows = 0, const int32_t* tile_first = nullptr) {
    if (n_frames < 0 || n_bins <= 0 || ld < n_bins || ld > (1 << 20)) return fail(MPX_ERR_ARG, "mpx_mel_unwarp: bad size%s");
    if (k_mag <= 0 || k_mag > kGemmKMax || k_phase <= 0 || k_phase > kGemmKMax)
        return fail(MPX_ERR_ARG, "mpx_mel_unwarp: coefficient count must be in 1..64%s");
    if (n_frames == 0) return MPX_OK;
    if (!a_mag || !u_mag || !out_mag || !a_real || !a_imag || !u_phase || !out_real || !out_imag)
        return fail(MPX_ERR_ARG, "mpx_mel_unwarp: null pointer%s");
    UnwarpJobs jobs;
    jobs.j[0] = {a_mag, u_mag, out_mag, (int)k_mag, 1};
    jobs.j[1] = {a_real, u_phase, out_real, (int)k_phase, 0};
    jobs.j[2] = {a_imag, u_phase, out_imag, (int)k_phase, 0};
    if (tile_first) {   // magnitudes: one product per constant-rate row, interpolated out of an LDS tile
        if (int rc = dispatch_unwarp_tiled((hipStream_t)stream, jobs.j[0], (int)k_mag, (long long)n_frames, (long long)n_rows,
                                           (int)n_bins, (int)ld, rw, tile_first)) return rc;
    } else {
        if (int rc = dispatch_unwarp_mfma((hipStream_t)stream, jobs, 0, 1, (int)k_mag, (long long)n_frames, (int)n_bins, (int)ld, rw)) return rc;
    }
    if (int rc = dispatch_unwarp_mfma((hipStream_t)stream, jobs, 1, 2, (int)k_phase, (long long)n_frames, (int)n_bins, (int)ld, rw)) return rc;
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_mel_unwarp(void* stream, int64_t n_frames, int32_t n_bins, const float* a_mag, int32_t k_mag,
                   const float* u_mag, float* out_mag, const float* a_real, const float* a_imag, int32_t k_phase,
                   const float* u_phase, float* out_real, float* out_imag, int64_t ld) {
    return mel_unwarp_impl(stream, n_frames, n_bins, a_mag, k_mag, u_mag, out_mag, a_real, a_imag, k_phase, u_phase,
                           out_real, out_imag, ld, UnwarpRows{nullptr, nullptr, nullptr, nullptr, 0});
}

int mpx_mel_unwarp_rows(void* stream, int64_t n_frames, int32_t n_bins, const float* a_mag, int32_t k_mag,
                        const float* u_mag, float* out_mag, const float* a_real, const float* a_imag, int32_t k_phase,
                        const float* u_phase, float* out_real, float* out_imag, int64_t ld, const int32_t* row0,
                        const int32_t* row1, const float* row_t, int64_t n_rows, const int32_t* tile_first,
                        const int32_t* voiced, int32_t n_phase_bins) {
    if (!row0 || !row1 || !row_t) return fail(MPX_ERR_ARG, "mpx_mel_unwarp_rows: null row table%s");
    if (n_phase_bins < 0) return fail(MPX_ERR_ARG, "mpx_mel_unwarp_rows: negative n_phase_bins%s");
    if (tile_first && n_rows <= 0) return fail(MPX_ERR_ARG, "mpx_mel_unwarp_rows: tile_first needs n_rows%s");
    return mel_unwarp_impl(stream, n_frames, n_bins, a_mag, k_mag, u_mag, out_mag, a_real, a_imag, k_phase, u_phase,
                           out_real, out_imag, ld, UnwarpRows{row0, row1, row_t, voiced, (int)n_phase_bins}, n_rows, tile_first);
}

static int noise_stats_impl(void* stream, int fft_len, const void* tables, const float* noise, const int64_t* frame_pos,
                            const int32_t* frame_left, const int32_t* frame_right, const int32_t* frame_wtype,
                            int64_t n_frames, float* out_sum, float* spectra) {
    const int P = p_of(fft_len);
    if (!P) return fail(MPX_ERR_ARG, "mpx_noise_stats: fft_len must be 1024, 2048 or 4096%s");
    if (n_frames < 0) return fail(MPX_ERR_ARG, "mpx_noise_stats: negative n_frames%s");
    if (n_frames == 0) return MPX_OK;
    if (!tables || !noise || !frame_pos || !frame_left || !frame_right || !frame_wtype || !out_sum)
        return fail(MPX_ERR_ARG, "mpx_noise_stats: null pointer%s");
    const dim3 grid(grid_for(n_frames, kAnaWaves)), block(kAnaThreads);
    hipStream_t s = (hipStream_t)stream;
    if (P == 32) {
        if (int rc = set_lds(k_noise_stats<32>, lds_bytes_ana<32>())) return rc;
        hipLaunchKernelGGL(k_noise_stats<32>, grid, block, lds_bytes_ana<32>(), s, noise, (const long long*)frame_pos,
                           frame_left, frame_right, frame_wtype, (long long)n_frames, (const float*)tables, out_sum, spectra);
    } else if (P == 16) {
        if (int rc = set_lds(k_noise_stats<16>, lds_bytes_ana<16>())) return rc;
        hipLaunchKernelGGL(k_noise_stats<16>, grid, block, lds_bytes_ana<16>(), s, noise, (const long long*)frame_pos,
                           frame_left, frame_right, frame_wtype, (long long)n_frames, (const float*)tables, out_sum, spectra);
    } else {
        if (int rc = set_lds(k_noise_stats<8>, lds_bytes_ana<8>())) return rc;
        hipLaunchKernelGGL(k_noise_stats<8>, grid, block, lds_bytes_ana<8>(), s, noise, (const long long*)frame_pos,
                           frame_left, frame_right, frame_wtype, (long long)n_frames, (const float*)tables, out_sum, spectra);
    }
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_noise_stats(void* stream, int fft_len, const void* tables, const float* noise, const int64_t* frame_pos,
                    const int32_t* frame_left, const int32_t* frame_right, const int32_t* frame_wtype,
                    int64_t n_frames, float* out_sum) {
    return noise_stats_impl(stream, fft_len, tables, noise, frame_pos, frame_left, frame_right, frame_wtype, n_frames,
                            out_sum, nullptr);
}

int64_t mpx_noise_spectra_floats(int fft_len, int64_t n_frames) {
    return (fft_len == 4096 && n_frames > 0) ? n_frames * (int64_t)kSpecFrameFloats : 0;
}

int mpx_noise_stats_spectra(void* stream, int fft_len, const void* tables, const float* noise, const int64_t* frame_pos,
                            const int32_t* frame_left, const int32_t* frame_right, const int32_t* frame_wtype,
                            int64_t n_frames, float* out_sum, float* spectra) {
    if (fft_len != 4096) return fail(MPX_ERR_ARG, "mpx_noise_stats_spectra: fft_len must be 4096%s");
    if (!spectra && n_frames > 0) return fail(MPX_ERR_ARG, "mpx_noise_stats_spectra: null pointer%s");
    return noise_stats_impl(stream, fft_len, tables, noise, frame_pos, frame_left, frame_right, frame_wtype, n_frames,
                            out_sum, spectra);
}

int mpx_synth_comp_slots(void) { return device_cus() * kCompPairs; }

// Relative speed of the slots' wave pairs (see mpx_synth_ola_slot_weights): the pairs (0, 1), (2, 3), (4, 5) of a workgroup
// hold the oldest / middle / youngest wave of every SIMD.  12 waves, interleaved A/B of the configs[2] synthesis side:
// equal shares 1.332 ms, 100:80:60 1.301, 100:75:55 1.314, 100:90:80 1.287, 100:85:70 .. 100:88:76 1.270-1.277 (flat).
// Re-tuned on the final kernel (no scratch, more of its time in the VALU: the age effect is stronger): 100:86:73 1.190,
// 100:90:80 1.194, 100:92:86 1.215, 100:82:66 1.167, 100:84:62 1.168, 100:80:62 1.178, 100:78:58 1.177, 100:76:62 1.179.
#ifndef MPX_COMP_W0
#define MPX_COMP_W0 100
#endif
#ifndef MPX_COMP_W1
#define MPX_COMP_W1 (MPX_COMP_PAIR_WAVES > 8 ? 82 : 80)
#endif
#ifndef MPX_COMP_W2
#define MPX_COMP_W2 66
#endif
int mpx_synth_comp_slot_weights(float* weights_host, int32_t n_slots) {
    if (!weights_host || n_slots < 0) return fail(MPX_ERR_ARG, "mpx_synth_comp_slot_weights: bad arguments%s");
    for (int s = 0; s < n_slots; ++s) {
        const int age = ((s % kCompPairs) * 2) / 4;   // age rank of the pair's waves on their SIMDs
        weights_host[s] = (age == 0) ? (float)MPX_COMP_W0 : ((age == 1) ? (float)MPX_COMP_W1 : (float)MPX_COMP_W2);
    }
    return MPX_OK;
}

static int synthesis_compressed_ola_impl(void* stream, int fft_len, const void* tables, const float* mag, const float* real,
                                 const float* imag, const float* noise, const int64_t* noise_pos,
                                 const int32_t* noise_left, const int32_t* noise_right, const int32_t* noise_wtype,
                                 const int32_t* voiced, const float* inv_gain, const int32_t* row0,
                                 const int32_t* row1, const float* row_t, const int32_t* win_left,
                                 const int32_t* win_right, const int32_t* pm_rel, const float* per_v,
                                 const float* ap_v, const float* ap_u, const mpx_ola_run* runs, int32_t n_runs,
                                 const int32_t* slot_off, const int32_t* slot_runs, int32_t n_slots,
                                 float* strips, float* pcm_out, int64_t ld, int32_t n_per_bins,
                                 const float* spectra) {
    const int P = p_of(fft_len);
    if (!P) return fail(MPX_ERR_ARG, "mpx_synthesis_compressed_ola: fft_len must be 1024, 2048 or 4096%s");
    const int n_per = (n_per_bins <= 0 || n_per_bins > fft_len / 2 + 1) ? fft_len / 2 + 1 : (int)n_per_bins;
    if (n_runs < 0 || n_slots < 0) return fail(MPX_ERR_ARG, "mpx_synthesis_compressed_ola: negative count%s");
    if (n_runs == 0 || n_slots == 0) return MPX_OK;
    if (!tables || !mag || !real || !imag || !noise || !noise_pos || !noise_left || !noise_right || !noise_wtype ||
        !voiced || !inv_gain || !win_left || !win_right || !pm_rel || !per_v || !ap_v ||
        !ap_u || !runs || !slot_off || !slot_runs || !strips || !pcm_out)
        return fail(MPX_ERR_ARG, "mpx_synthesis_compressed_ola: null pointer%s");
    const bool lerp = row0 || row1 || row_t;   // all three null: one spectrum row per frame, row index = frame index
    if (lerp && (!row0 || !row1 || !row_t))
        return fail(MPX_ERR_ARG, "mpx_synthesis_compressed_ola: row0 / row1 / row_t must be given together%s");
    CompFrameTabs tb{(const long long*)noise_pos, noise_left, noise_right, noise_wtype, voiced, inv_gain,
                     row0, row1, row_t, win_left, win_right, pm_rel, spectra};
    hipStream_t s = (hipStream_t)stream;
    const dim3 pgrid((n_slots + kCompPairs - 1) / kCompPairs), pblock(kCompPairWaves * 64);
    if (spectra) {   // stored noise spectra (mpx_noise_stats_spectra): N = 4096, one row per frame
        if (P != 32 || lerp)
            return fail(MPX_ERR_ARG, "mpx_synthesis_compressed_ola_spectra: fft_len 4096 and one row per frame only%s");
        if (n_per <= 512) {
            if (int rc = set_lds(k_synth_comp_pair<32, false, 8, true>, lds_bytes_comp_pair<32>())) return rc;
            hipLaunchKernelGGL((k_synth_comp_pair<32, false, 8, true>), pgrid, pblock, lds_bytes_comp_pair<32>(),
                               s, mag, real, imag, noise, tb, per_v, ap_v, ap_u, (const RunDesc*)runs, slot_off, slot_runs,
                               (int)n_slots, (const float*)tables, strips, pcm_out, (long long)ld, n_per);
        } else {
            if (int rc = set_lds(k_synth_comp_pair<32, false, -1, true>, lds_bytes_comp_pair<32>())) return rc;
            hipLaunchKernelGGL((k_synth_comp_pair<32, false, -1, true>), pgrid, pblock, lds_bytes_comp_pair<32>(),
                               s, mag, real, imag, noise, tb, per_v, ap_v, ap_u, (const RunDesc*)runs, slot_off, slot_runs,
                               (int)n_slots, (const float*)tables, strips, pcm_out, (long long)ld, n_per);
        }
        MPX_HIP_CHECK(hipGetLastError());
        return MPX_OK;
    }
#define MPX_LAUNCH_COMP(PP, LL)                                                                                        \
    do {                                                                                                             \
        if (int rc = set_lds(k_synth_comp_pair<PP, LL>, lds_bytes_comp_pair<PP>())) return rc;                       \
        hipLaunchKernelGGL((k_synth_comp_pair<PP, LL>), pgrid, pblock, lds_bytes_comp_pair<PP>(), s, mag, real, imag, \
                           noise, tb, per_v, ap_v, ap_u, (const RunDesc*)runs, slot_off, slot_runs, (int)n_slots,    \
                           (const float*)tables, strips, pcm_out, (long long)ld, n_per);                 \
    } while (0)
    if (lerp) {
        if (P == 32) MPX_LAUNCH_COMP(32, true);
        else if (P == 16) MPX_LAUNCH_COMP(16, true);
        else MPX_LAUNCH_COMP(8, true);
    } else {
        if (P == 32 && n_per <= 512) {   // the crossfade ends at or below bin 512 (48 / 44.1 kHz): the NPQ == 8 schedule
            if (int rc = set_lds(k_synth_comp_pair<32, false, 8>, lds_bytes_comp_pair<32>())) return rc;
            hipLaunchKernelGGL((k_synth_comp_pair<32, false, 8>), pgrid, pblock, lds_bytes_comp_pair<32>(), s, mag, real, imag,
                               noise, tb, per_v, ap_v, ap_u, (const RunDesc*)runs, slot_off, slot_runs, (int)n_slots,
                               (const float*)tables, strips, pcm_out, (long long)ld, n_per);
        } else if (P == 32) MPX_LAUNCH_COMP(32, false);
        else if (P == 16) MPX_LAUNCH_COMP(16, false);
        else MPX_LAUNCH_COMP(8, false);
    }
#undef MPX_LAUNCH_COMP
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_synthesis_compressed_ola(void* stream, int fft_len, const void* tables, const float* mag, const float* real,
                                 const float* imag, const float* noise, const int64_t* noise_pos,
                                 const int32_t* noise_left, const int32_t* noise_right, const int32_t* noise_wtype,
                                 const int32_t* voiced, const float* inv_gain, const int32_t* row0,
                                 const int32_t* row1, const float* row_t, const int32_t* win_left,
                                 const int32_t* win_right, const int32_t* pm_rel, const float* per_v,
                                 const float* ap_v, const float* ap_u, const mpx_ola_run* runs, int32_t n_runs,
                                 const int32_t* slot_off, const int32_t* slot_runs, int32_t n_slots,
                                 float* strips, float* pcm_out, int64_t ld, int32_t n_per_bins) {
    return synthesis_compressed_ola_impl(stream, fft_len, tables, mag, real, imag, noise, noise_pos, noise_left, noise_right,
                                         noise_wtype, voiced, inv_gain, row0, row1, row_t, win_left, win_right, pm_rel, per_v,
                                         ap_v, ap_u, runs, n_runs, slot_off, slot_runs, n_slots, strips, pcm_out, ld, n_per_bins, nullptr);
}

int mpx_synthesis_compressed_ola_spectra(void* stream, int fft_len, const void* tables, const float* mag, const float* real,
                                 const float* imag, const float* noise, const int64_t* noise_pos,
                                 const int32_t* noise_left, const int32_t* noise_right, const int32_t* noise_wtype,
                                 const int32_t* voiced, const float* inv_gain, const int32_t* row0,
                                 const int32_t* row1, const float* row_t, const int32_t* win_left,
                                 const int32_t* win_right, const int32_t* pm_rel, const float* per_v,
                                 const float* ap_v, const float* ap_u, const mpx_ola_run* runs, int32_t n_runs,
                                 const int32_t* slot_off, const int32_t* slot_runs, int32_t n_slots,
                                 float* strips, float* pcm_out, int64_t ld, int32_t n_per_bins, const float* spectra) {
    if (!spectra && n_runs > 0 && n_slots > 0)
        return fail(MPX_ERR_ARG, "mpx_synthesis_compressed_ola_spectra: null pointer%s");
    return synthesis_compressed_ola_impl(stream, fft_len, tables, mag, real, imag, noise, noise_pos, noise_left, noise_right,
                                         noise_wtype, voiced, inv_gain, row0, row1, row_t, win_left, win_right, pm_rel, per_v,
                                         ap_v, ap_u, runs, n_runs, slot_off, slot_runs, n_slots, strips, pcm_out, ld, n_per_bins, spectra);
}

// Slot weights of k_roundtrip_pair (see mpx_synth_comp_slot_weights: pairs of the oldest / middle / youngest waves of the
// SIMDs).  Interleaved sweeps of the configs[1] step on two boxes: equal shares 0.592 ms, 100:90:80 0.574, 100:86:73 0.567,
// 100:82:66 0.555, 100:72:60 0.547, 100:75:55 0.539-0.545; a smallest share below ~25 frames per run (100:70:48, 100:78:50)
// fell off a cliff (0.80 ms) while the planner DROPPED cuts that would let non-adjacent runs overlap (runs shorter than
// fft_len output samples: one slot idle, its neighbour with twice the frames).  It now moves such a cut forward
// (hostmath._enforce_span); on that planner, another box: 100:77:58 0.522, 100:75:55 0.523, 100:72:52 0.512, 100:71:49 0.511,
// 100:68:46 0.516, 100:65:42 0.527, 100:60:38 0.548 -- the per-frame times by age (tools/roundtrip_phase_probe.py: 22.5 /
// 31.6 / 46.3 us) say 100:71:49.
#ifndef MPX_RT_W1
#define MPX_RT_W1 71
#endif
#ifndef MPX_RT_W2
#define MPX_RT_W2 50
#endif
int mpx_roundtrip_slot_weights(float* weights_host, int32_t n_slots) {
    if (!weights_host || n_slots < 0) return fail(MPX_ERR_ARG, "mpx_roundtrip_slot_weights: bad arguments%s");
    for (int s = 0; s < n_slots; ++s) {
        const int age = ((s % kCompPairs) * 2) / 4;
        weights_host[s] = (age == 0) ? 100.0f : ((age == 1) ? (float)MPX_RT_W1 : (float)MPX_RT_W2);
    }
    return MPX_OK;
}

int mpx_roundtrip_lossless_ola(void* stream, int fft_len, const void* tables, const float* sig, const int64_t* frame_pos,
                               const int32_t* frame_left, const int32_t* frame_right, int64_t n_frames,
                               const mpx_ola_run* runs, int32_t n_runs, const int32_t* slot_off, const int32_t* slot_runs,
                               int32_t n_slots, const int32_t* pm_rel, float* out_mag, float* out_real, float* out_imag,
                               float* strips, float* pcm_out, int64_t ld) {
    const int P = p_of(fft_len);
    if (!P) return fail(MPX_ERR_ARG, "mpx_roundtrip_lossless_ola: fft_len must be 1024, 2048 or 4096%s");
    if (n_frames < 0 || n_runs < 0 || n_slots < 0) return fail(MPX_ERR_ARG, "mpx_roundtrip_lossless_ola: negative count%s");
    if (ld < fft_len / 2 + 1) return fail(MPX_ERR_ARG, "mpx_roundtrip_lossless_ola: ld < fft_len/2 + 1%s");
    if (n_frames == 0 || n_runs == 0 || n_slots == 0) return MPX_OK;
    if (!tables || !sig || !frame_pos || !frame_left || !frame_right || !runs || !slot_off || !slot_runs || !pm_rel ||
        !out_mag || !out_real || !out_imag || !strips || !pcm_out)
        return fail(MPX_ERR_ARG, "mpx_roundtrip_lossless_ola: null pointer%s");
    hipStream_t s = (hipStream_t)stream;
    const dim3 pgrid((n_slots + kCompPairs - 1) / kCompPairs), pblock(kCompPairWaves * 64);
#define MPX_LAUNCH_RT(PP)                                                                                              \
    do {                                                                                                             \
        if (int rc = set_lds(k_roundtrip_pair<PP>, lds_bytes_comp_pair<PP>())) return rc;            \
        hipLaunchKernelGGL(k_roundtrip_pair<PP>, pgrid, pblock, lds_bytes_comp_pair<PP>(), s, sig,   \
                           (const long long*)frame_pos, frame_left, frame_right, (const RunDesc*)runs, slot_off,     \
                           slot_runs, (int)n_slots, pm_rel, (const float*)tables, out_mag, out_real, out_imag, strips, \
                           pcm_out, (long long)ld);                                                                  \
    } while (0)
    if (P == 32) MPX_LAUNCH_RT(32);
    else if (P == 16) MPX_LAUNCH_RT(16);
    else MPX_LAUNCH_RT(8);
#undef MPX_LAUNCH_RT
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

// n_var_rows / need / tmp_real / tmp_imag: the phase jobs on the variable-rate rows (mpx_mel_warp_rows), or 0 / null
static int mel_warp_impl(void* stream, int64_t n_frames, int32_t n_bins, const float* mag, const float* real,
                         const float* imag, const int32_t* row0, const int32_t* row1, const float* row_t,
                         const float* w_mag, int32_t mag_dim, const float* w_phase, int32_t phase_dim, const float* voiced,
                         float* out_mag, float* out_real, float* out_imag, int64_t ld, int mag_mode,
                         int64_t n_var_rows = 0, const float* need = nullptr, float* tmp_real = nullptr,
                         float* tmp_imag = nullptr) {
    if (n_frames < 0 || n_bins <= 0 || ld < n_bins) return fail(MPX_ERR_ARG, "mpx_mel_warp: bad size%s");
    if (mag_dim <= 0 || mag_dim > kWarpTile || phase_dim <= 0 || phase_dim > kWarpTile)
        return fail(MPX_ERR_ARG, "mpx_mel_warp: output dimension must be in 1..64%s");
    if (n_frames == 0) return MPX_OK;
    if (!mag || !real || !imag || !w_mag || !w_phase || !voiced || !out_mag || !out_real || !out_imag)
        return fail(MPX_ERR_ARG, "mpx_mel_warp: null pointer%s");
    if ((row0 == nullptr) != (row1 == nullptr) || (row0 == nullptr) != (row_t == nullptr))
        return fail(MPX_ERR_ARG, "mpx_mel_warp: row0/row1/row_t must be all null or all given%s");
    const bool phv = tmp_real != nullptr;
    if (phv && (!row0 || !tmp_imag || !need || n_var_rows <= 0))
        return fail(MPX_ERR_ARG, "mpx_mel_warp_rows: the variable-rate phase warp needs row tables, both scratch matrices and the row flags%s");
    WarpJobs jobs;
    jobs.j[0] = {mag, w_mag, out_mag, nullptr, (int)mag_dim, mag_mode, (long long)n_frames};
    if (phv) {
        jobs.j[1] = {real, w_phase, tmp_real, need, (int)phase_dim, 3, (long long)n_var_rows};
        jobs.j[2] = {imag, w_phase, tmp_imag, need, (int)phase_dim, 3, (long long)n_var_rows};
    } else {
        jobs.j[1] = {real, w_phase, out_real, voiced, (int)phase_dim, 1, (long long)n_frames};
        jobs.j[2] = {imag, w_phase, out_imag, voiced, (int)phase_dim, 1, (long long)n_frames};
    }
    const long long max_f = phv ? ((long long)n_frames > (long long)n_var_rows ? (long long)n_frames : (long long)n_var_rows) : (long long)n_frames;
    const dim3 grid((unsigned)((max_f + kWarpTile - 1) / kWarpTile), 3);
    {
        const dim3 g2(grid.x, 3);
        const int ntm = ((int)mag_dim + 15) / 16, ntp = ((int)phase_dim + 15) / 16;
#define MPX_WARP_GO(NTM, NTP, IN, MM, PV)                                                                              \
    hipLaunchKernelGGL((k_mel_warp_mfma<NTM, NTP, IN, MM, PV>), g2, dim3(256), 0, (hipStream_t)stream, jobs, (int)n_bins, \
                       row0, row1, row_t, (long long)ld)
#define MPX_WARP_LAUNCH(NTM, NTP)                                   \
    do {                                                            \
        if (phv && mag_mode == 0) MPX_WARP_GO(NTM, NTP, true, 0, true);      \
        else if (phv) MPX_WARP_GO(NTM, NTP, true, 2, true);         \
        else if (row0 && mag_mode == 0) MPX_WARP_GO(NTM, NTP, true, 0, false);  \
        else if (row0) MPX_WARP_GO(NTM, NTP, true, 2, false);       \
        else if (mag_mode == 0) MPX_WARP_GO(NTM, NTP, false, 0, false);      \
        else MPX_WARP_GO(NTM, NTP, false, 2, false);                \
    } while (0)
#define MPX_WARP_ROW(NTM)                       \
    switch (ntp) {                              \
        case 1: MPX_WARP_LAUNCH(NTM, 1); break; \
        case 2: MPX_WARP_LAUNCH(NTM, 2); break; \
        case 3: MPX_WARP_LAUNCH(NTM, 3); break; \
        default: MPX_WARP_LAUNCH(NTM, 4); break; \
    }
        switch (ntm) {
            case 1: MPX_WARP_ROW(1) break;
            case 2: MPX_WARP_ROW(2) break;
            case 3: MPX_WARP_ROW(3) break;
            default: MPX_WARP_ROW(4) break;
        }
#undef MPX_WARP_ROW
#undef MPX_WARP_LAUNCH
#undef MPX_WARP_GO
    }
    if (phv) {
        const long long n_el = (long long)n_frames * phase_dim;
        hipLaunchKernelGGL(k_warp_phase_rows, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)tmp_real, (const float*)tmp_imag, row0, row1, row_t, voiced, (long long)n_frames,
                           (int)phase_dim, out_real, out_imag);
    }
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

// The second half of the constant-rate phase streams on its own (mpx_analysis_compressed_fused_cr leaves their
// variable-rate warp in tmp_real / tmp_imag): row interpolation, voicing mask, clip.
int mpx_warp_phase_rows(void* stream, int64_t n_frames, int32_t phase_dim, const float* tmp_real, const float* tmp_imag,
                        const int32_t* row0, const int32_t* row1, const float* row_t, const float* voiced, float* out_real,
                        float* out_imag) {
    if (n_frames < 0 || phase_dim <= 0) return fail(MPX_ERR_ARG, "mpx_warp_phase_rows: bad size%s");
    if (n_frames == 0) return MPX_OK;
    if (!tmp_real || !tmp_imag || !row0 || !row1 || !row_t || !voiced || !out_real || !out_imag)
        return fail(MPX_ERR_ARG, "mpx_warp_phase_rows: null pointer%s");
    const long long n_el = (long long)n_frames * phase_dim;
    hipLaunchKernelGGL(k_warp_phase_rows, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tmp_real,
                       tmp_imag, row0, row1, row_t, voiced, (long long)n_frames, (int)phase_dim, out_real, out_imag);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_mel_warp_rows(void* stream, int64_t n_frames, int32_t n_bins, const float* mag, const float* real,
                      const float* imag, const int32_t* row0, const int32_t* row1, const float* row_t, const float* w_mag,
                      int32_t mag_dim, const float* w_phase, int32_t phase_dim, const float* voiced, float* out_mag,
                      float* out_real, float* out_imag, int64_t ld, int32_t mag_fbank, int64_t n_var_rows,
                      const float* rows_in_use, float* tmp_real, float* tmp_imag) {
    return mel_warp_impl(stream, n_frames, n_bins, mag, real, imag, row0, row1, row_t, w_mag, mag_dim, w_phase, phase_dim,
                         voiced, out_mag, out_real, out_imag, ld, mag_fbank ? 2 : 0, n_var_rows, rows_in_use, tmp_real,
                         tmp_imag);
}

int mpx_mel_warp(void* stream, int64_t n_frames, int32_t n_bins, const float* mag, const float* real, const float* imag,
                 const int32_t* row0, const int32_t* row1, const float* row_t, const float* w_mag, int32_t mag_dim,
                 const float* w_phase, int32_t phase_dim, const float* voiced, float* out_mag, float* out_real,
                 float* out_imag, int64_t ld) {
    return mel_warp_impl(stream, n_frames, n_bins, mag, real, imag, row0, row1, row_t, w_mag, mag_dim, w_phase, phase_dim,
                         voiced, out_mag, out_real, out_imag, ld, 0);
}

int mpx_mel_warp_fbank(void* stream, int64_t n_frames, int32_t n_bins, const float* mag, const float* real,
                       const float* imag, const int32_t* row0, const int32_t* row1, const float* row_t,
                       const float* w_fbank, int32_t mag_dim, const float* w_phase, int32_t phase_dim, const float* voiced,
                       float* out_mag, float* out_real, float* out_imag, int64_t ld) {
    return mel_warp_impl(stream, n_frames, n_bins, mag, real, imag, row0, row1, row_t, w_fbank, mag_dim, w_phase,
                         phase_dim, voiced, out_mag, out_real, out_imag, ld, 2);
}

int mpx_min_phase(void* stream, int fft_len, const void* tables, const float* mag, const int32_t* row0,
                  const int32_t* row1, const float* row_t, int64_t n_frames, float* out_mag, float* out_real,
                  float* out_imag, int64_t ld) {
    const int P = p_of(fft_len);
    if (!P) return fail(MPX_ERR_ARG, "mpx_min_phase: fft_len must be 1024, 2048 or 4096%s");
    if (n_frames < 0) return fail(MPX_ERR_ARG, "mpx_min_phase: negative n_frames%s");
    if (n_frames == 0) return MPX_OK;
    if (!tables || !mag || !row0 || !row1 || !row_t || !out_mag || !out_real || !out_imag)
        return fail(MPX_ERR_ARG, "mpx_min_phase: null pointer%s");
    const dim3 grid(grid_for(n_frames)), block(kThreads);
    hipStream_t s = (hipStream_t)stream;
    if (P == 32) {
        if (int rc = set_lds(k_min_phase<32>, lds_bytes<32>())) return rc;
        hipLaunchKernelGGL(k_min_phase<32>, grid, block, lds_bytes<32>(), s, mag, row0, row1, row_t,
                           (long long)n_frames, (const float*)tables, out_mag, out_real, out_imag, (long long)ld);
    } else if (P == 16) {
        if (int rc = set_lds(k_min_phase<16>, lds_bytes<16>())) return rc;
        hipLaunchKernelGGL(k_min_phase<16>, grid, block, lds_bytes<16>(), s, mag, row0, row1, row_t,
                           (long long)n_frames, (const float*)tables, out_mag, out_real, out_imag, (long long)ld);
    } else {
        if (int rc = set_lds(k_min_phase<8>, lds_bytes<8>())) return rc;
        hipLaunchKernelGGL(k_min_phase<8>, grid, block, lds_bytes<8>(), s, mag, row0, row1, row_t,
                           (long long)n_frames, (const float*)tables, out_mag, out_real, out_imag, (long long)ld);
    }
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_noise_gains(void* stream, const float* sums, const int32_t* voiced, const int32_t* utt_frame_off,
                    int32_t n_utts, int32_t bins_per_frame, float* inv_gain, double* gains) {
    if (n_utts < 0 || bins_per_frame <= 0) return fail(MPX_ERR_ARG, "mpx_noise_gains: bad size%s");
    if (n_utts == 0) return MPX_OK;
    if (!sums || !voiced || !utt_frame_off || !inv_gain) return fail(MPX_ERR_ARG, "mpx_noise_gains: null pointer%s");
    hipLaunchKernelGGL(k_noise_gains, dim3((unsigned)n_utts), dim3(256), 0, (hipStream_t)stream, sums, voiced,
                       utt_frame_off, (int)bins_per_frame, inv_gain, gains);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_post_filter(void* stream, const float* mag_mel_log, int64_t n_frames, int32_t dim, const int32_t* half_len,
                    int32_t nx_first, int32_t nx_last, const float* tilt, float* out) {
    if (n_frames < 0 || dim < 3 || dim > 256) return fail(MPX_ERR_ARG, "mpx_post_filter: dim must be in 3..256%s");
    if (nx_first < 0 || nx_last >= dim || nx_first > nx_last) return fail(MPX_ERR_ARG, "mpx_post_filter: bad bin range%s");
    if (n_frames == 0) return MPX_OK;
    if (!mag_mel_log || !half_len || !tilt || !out) return fail(MPX_ERR_ARG, "mpx_post_filter: null pointer%s");
    const int rows = 256 / dim;
    const dim3 grid((unsigned)((n_frames + rows - 1) / rows)), block(256);
    hipLaunchKernelGGL(k_post_filter, grid, block, sizeof(float) * (size_t)rows * dim, (hipStream_t)stream,
                       mag_mel_log, (long long)n_frames, (int)dim, half_len, (int)nx_first, (int)nx_last, tilt, out);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_pcm16(void* stream, const void* y, int32_t y_is_f64, const int64_t* out_off, int32_t n_utts, int64_t max_len,
              double norm, double* peaks, int16_t* out) {
    if (n_utts < 0 || max_len < 0) return fail(MPX_ERR_ARG, "mpx_pcm16: negative size%s");
    if (n_utts == 0 || max_len == 0) return MPX_OK;
    if (!y || !out_off || !peaks || !out) return fail(MPX_ERR_ARG, "mpx_pcm16: null pointer%s");
    if (n_utts > 65535) return fail(MPX_ERR_ARG, "mpx_pcm16: at most 65535 utterances per call%s");
    hipStream_t s = (hipStream_t)stream;
    const dim3 g2((unsigned)((max_len + 255) / 256), (unsigned)n_utts);
    const dim3 gp((unsigned)((max_len + 256 * kPeakPerThread - 1) / (256 * kPeakPerThread)), (unsigned)n_utts);
    MPX_HIP_CHECK(hipMemsetAsync(peaks, 0, sizeof(double) * (size_t)n_utts, s));
    if (y_is_f64) {
        hipLaunchKernelGGL(k_peak_abs<double>, gp, dim3(256), 0, s, (const double*)y, (const long long*)out_off, peaks);
        hipLaunchKernelGGL(k_pcm16<double>, g2, dim3(256), 0, s, (const double*)y, (const long long*)out_off, peaks, norm, (short*)out);
    } else {
        hipLaunchKernelGGL(k_peak_abs<float>, gp, dim3(256), 0, s, (const float*)y, (const long long*)out_off, peaks);
        hipLaunchKernelGGL(k_pcm16<float>, g2, dim3(256), 0, s, (const float*)y, (const long long*)out_off, peaks, norm, (short*)out);
    }
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_pcm16_to_f32(void* stream, const int16_t* pcm, int64_t n, float* out) {
    if (n < 0) return fail(MPX_ERR_ARG, "mpx_pcm16_to_f32: negative size%s");
    if (n == 0) return MPX_OK;
    if (!pcm || !out) return fail(MPX_ERR_ARG, "mpx_pcm16_to_f32: null pointer%s");
    if (((uintptr_t)pcm & 7) || ((uintptr_t)out & 15)) return fail(MPX_ERR_ARG, "mpx_pcm16_to_f32: pcm must be 8-byte, out 16-byte aligned%s");
    const long long blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(k_pcm16_to_f32, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const short*)pcm,
                       (long long)n, out);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_hpf_block(void) { return kHpfBlock; }

int mpx_output_hpf(void* stream, const float* pcm, const int64_t* out_off, const int32_t* blk_off, int32_t n_utts,
                   int64_t max_len, const double* sos_host, const double* pmat, const double* gtab, double* zend,
                   double* zstart, double* y_tmp, double* y) {
    if (n_utts < 0 || max_len < 0) return fail(MPX_ERR_ARG, "mpx_output_hpf: negative size%s");
    if (n_utts == 0 || max_len == 0) return MPX_OK;
    if (!pcm || !out_off || !blk_off || !sos_host || !pmat || !gtab || !zend || !zstart || !y_tmp || !y)
        return fail(MPX_ERR_ARG, "mpx_output_hpf: null pointer%s");
    if (n_utts > 65535) return fail(MPX_ERR_ARG, "mpx_output_hpf: at most 65535 utterances per call%s");
    hipStream_t s = (hipStream_t)stream;
    const unsigned max_blocks = (unsigned)((max_len + kHpfBlock - 1) / kHpfBlock);
    const dim3 gz((max_blocks + 63) / 64, (unsigned)n_utts), gc((unsigned)n_utts),
        ga((unsigned)((max_len + 255) / 256), (unsigned)n_utts);
    for (int sec = 0; sec < 2; ++sec) {
        const double* q = sos_host + 6 * sec;   // scipy sos row: b0 b1 b2 a0 a1 a2
        const BiquadCoef c{q[0] / q[3], q[1] / q[3], q[2] / q[3], q[4] / q[3], q[5] / q[3]};
        double* out = (sec == 0) ? y_tmp : y;
        // section 0's free response is folded into section 1's loads: no k_hpf_apply pass over y_tmp in between
        if (sec == 0)
            hipLaunchKernelGGL(k_hpf_zero_state<float>, gz, dim3(64), 0, s, pcm, (const long long*)out_off, blk_off, c,
                               out, zend, (const double*)nullptr, (const double*)nullptr);
        else
            hipLaunchKernelGGL(k_hpf_zero_state<double>, gz, dim3(64), 0, s, (const double*)y_tmp,
                               (const long long*)out_off, blk_off, c, out, zend, gtab, (const double*)zstart);
        // (section 1's zero-state pass has consumed zstart: the carry may overwrite it -- same stream, in order)
        hipLaunchKernelGGL(k_hpf_carry, gc, dim3(64), 0, s, blk_off, (int)n_utts, pmat + 4 * sec, zend, zstart);
        if (sec == 1)
            hipLaunchKernelGGL(k_hpf_apply, ga, dim3(256), 0, s, (const long long*)out_off, blk_off,
                               gtab + 2 * (size_t)kHpfBlock * sec, zstart, out);
    }
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

}  // extern "C"
