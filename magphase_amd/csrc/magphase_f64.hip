// magphase_f64.hip -- analysis of pitch-synchronous frames with a float64 transform (features still float32).
//
// Same operation as k_analysis (magphase_hip.hip; magphase.py:74-119, 309-325, 457-476), for the callers whose next step
// amplifies the transform's round-off: the compressed analysis takes ln(mag^2 + 1e-8) and divides by |X|
// (magphase.py:2508-2521, libaudio.py:575-601), and speech frames have bins 60-80 dB below the frame's peak.  An fp32
// FFT leaves an absolute error of ~1e-6 of the peak on every bin -- 1e-3 .. 1e-2 relative on those bins, 7e-4 on the
// mel-warped outputs; with the window, the transform and the epilogue in float64 the features are correctly rounded
// float32 values of the reference's float64 ones (a few 1e-8 relative), and the warp's error falls to the fp32 GEMM's.
// Cost: float64 vector rate is half the fp32 rate and the 64 complex registers of a frame take 128 VGPRs, so 8 waves
// per CU; the kernel is compute-bound at ~2x the fp32 kernel's time (profiles/), which the compressed path can afford:
// it is not bandwidth-bound.
#include <type_traits>

#include "mpx_common.hpp"
#include "wave_fft_f64.hpp"

namespace mpx {

constexpr int kAna64Waves = 8;
// floats of LDS per wave for the frame's two half windows (f64_frame_transform): 928 doubles for N = 4096, in proportion for
// the shorter transforms (their sample rates have proportionally shorter pitch periods)
template <int P>
constexpr int f64_win_floats() { return 58 * P; }
#ifndef MPX_F64_DIT
#define MPX_F64_DIT 1   // the transform in the DIT form (fft_inreg_dit_f64: 136 fewer float64 instructions per frame); 0: DIF
#endif
constexpr bool kF64Dit = MPX_F64_DIT != 0;
template <int P>
constexpr size_t lds_bytes_ana64() {
    return sizeof(double) * (size_t)tw64_doubles<P>() +
           sizeof(float) * (size_t)(kAna64Waves * P * kXStride + 4 + kAna64Waves * f64_win_floats<P>());   // + frame queue + window regions
}

// sin(x), |x| <= pi/2: Taylor polynomial to x^17 (remainder (pi/2)^19 / 19! = 4e-14)
__device__ __forceinline__ double sin_halfpi_range(double x) {
    const double x2 = x * x;
    double p = 1.0 / 355687428096000.0;
    p = fma(x2, p, -1.0 / 1307674368000.0);
    p = fma(x2, p, 1.0 / 6227020800.0);
    p = fma(x2, p, -1.0 / 39916800.0);
    p = fma(x2, p, 1.0 / 362880.0);
    p = fma(x2, p, -1.0 / 5040.0);
    p = fma(x2, p, 1.0 / 120.0);
    p = fma(x2, p, -1.0 / 6.0);
    return fma(x * x2, p, x);
}

// np.hanning(1 + 2L)[k] on the rising half (t = k/L; L == 0: weight 1) and np.hanning(1 + 2R) on the falling half
// (t = (L + R - k)/R): 0.5 - 0.5 cos(pi t) = 0.5 + 0.5 sin(pi (t - 0.5))      (libaudio.py:70-84)
__device__ __forceinline__ double hann_half_f64(int k, int L, int LR, int kadd, double invL, double invR) {
    const bool rising = k <= L;
    const int num = rising ? k + kadd : LR - k;
    const double t = (double)num * (rising ? invL : invR);
    return fma(0.5, sin_halfpi_range(3.14159265358979323846 * (t - 0.5)), 0.5);
}

// 1/sqrt(s) from the fp32 hardware estimate (relative error e0 <= 1.2e-7) and ONE Newton step in double: the error
// becomes 1.5 e0^2 < 3e-14, six orders below the float32 rounding the features get next (a second step bought nothing
// and cost 4 of the ~20 fp64 instructions per bin)
__device__ __forceinline__ double rsqrt_f64(double s) {
    const double r = (double)__builtin_amdgcn_rsqf((float)s);
    return r * fma(-0.5 * s, r * r, 1.5);
}

// |X|, Re X / |X|, Im X / |X| as float32 (0, 0, 0 where X == 0: magphase.py:466-472).  PH == false: the magnitude only
// (rows whose phase features nobody reads, mpx_analysis_frames_f64's rows_in_use)
// zero2: |X|^2 at or below it is the transform's own rounding noise (k_analysis_f64: (2^-45 sum |windowed samples|)^2,
// never less than 1e-36) -- an exactly cancelling bin (the Nyquist bin of a frame of two exactly periodic pitch periods,
// say) comes out of numpy's FFT as 0.0 and the reference stores (0, 0, 0) for it; a residue of 2^-50 normalised to a
// "phase" of (1, 0) moved the compressed phase features by 6e-5.
template <bool PH>
__device__ __forceinline__ void feat_store(double xr, double xi, double zero2, float mag_scale, float* pm, float* pr, float* pi) {
    const double s = xr * xr + xi * xi;
    // (the fp32 estimate needs a normal float: |X|^2 below 1e-38 is zero for the float32 features anyway)
    const bool nz = s > zero2;
    const double r = nz ? rsqrt_f64(s) : 0.0;
    *pm = (float)(s * r) * mag_scale;
    if (PH) {
        *pr = (float)(xr * r);
        *pi = (float)(xi * r);
    }
}

// register that holds z[l + 64 j] on entry / bin kappa + 64 q on exit of the transform
template <int P>
__device__ __forceinline__ constexpr int f64_in_reg(int j) { return kF64Dit ? brev(j, ilog2(P)) : j; }
template <int P>
__device__ __forceinline__ constexpr int f64_out_reg(int q) { return kF64Dit ? q : brev(q, ilog2(P)); }

// The front of the float64 analysis of ONE frame by one wave: samples HBM -> LDS, window (numpy's own weights from win_tab,
// or the analytic half Hann), gather in FFT order, transform.  On return lane kappa, register f64_out_reg(q) holds bin row q
// of the half-size complex transform (see the split in the callers); zero2 / mag_scale as explained inline.
template <int P>
__device__ __forceinline__ void f64_frame_transform(const float* __restrict__ sig, long long pos, int Lf, int Rf,
                                                    const double* __restrict__ win_tab, int win_cap, const double* tw,
                                                    float* xbuf, unsigned xbuf_byte, float* wbuf, unsigned wbuf_byte,
                                                    int lane, double (&re)[P], double (&im)[P], double& zero2_out,
                                                    float& mag_scale_out) {
    constexpr int N = 128 * P, kTile = 64 * P;
    const FrameGeom g = frame_geom(sig, pos, Lf, Rf, N);
    const double invL = (g.L > 0) ? 1.0 / (double)g.L : 1.0;
    const double invR = (g.LR > g.L) ? 1.0 / (double)(g.LR - g.L) : 0.0;
    // Window weights from the HOST's np.hanning (win_tab: half window of half length h at h (h + 1) / 2, h <= win_cap;
    // hostmath.hann_half_table): the products x[n] w[n] are then the reference's own doubles, and a bin that cancels
    // exactly (DC / Nyquist over exactly periodic pitch periods) leaves the reference's own residue -- 0.0 or a few
    // 2^-53 with the reference's sign -- instead of this kernel's polynomial's (round 3 flushed such bins to zero and the
    // configs[2] test had to skip the frames that interpolate from them).  Frames with a half longer than win_cap (F0
    // below 23 Hz at 48 kHz) keep the analytic window.  Wave-uniform.
    const int gR = g.LR - g.L;
    const bool tabw = rfl((int)(win_tab != nullptr && g.L <= win_cap && gR <= win_cap)) != 0;
    const double* tl = win_tab + ((long long)g.L * (g.L + 1) >> 1);
    const double* tr = win_tab + ((long long)gR * (gR + 1) >> 1);
    // The table path scales the frame by 2^30 (exact: a power of two changes no mantissa anywhere in the transform), so
    // that a residue of 2^-53 x amplitude squares to a normal float for the reciprocal square root's fp32 seed; the
    // magnitude is scaled back after its conversion to float32 (exact as well).
    const float in_scale = tabw ? 1073741824.0f : 1.0f;
    const float mag_scale = tabw ? 9.313225746154785e-10f : 1.0f;

    // ---- samples HBM -> LDS (fp32: the PCM is exact in float32), window in float64 while gathering in FFT order.
    // The window goes through LDS as well: the frame's weights, index i = k on the rising half (k <= L) and
    // i = L + 1 + (LR - k) on the falling one, are laid out in this wave's window region `wbuf` -- copied from the table
    // by global_load_lds_dword (no registers; lanes past a half's end are switched off, the copy honours EXEC) or
    // evaluated by ONE compact loop -- and the unrolled gather below reads them with ds_read_b64.  With the weights
    // evaluated / loaded from global memory inside the unrolled gather (64 copies of the polynomial or of the table
    // branches) the kernel had grown from 8 300 to 12 300 instructions, past the instruction cache, and from 0.45 to
    // 0.61 ms.  Windows longer than the region (L + R > 926 samples at N = 4096: F0 below 52 Hz) take several passes.
    double s_abs = 0.0;   // this lane's share of sum |windowed sample|: the scale of the transform's rounding noise
#pragma unroll
    for (int j = 0; j < P; ++j) re[j] = im[j] = 0.0;
    constexpr int capD = f64_win_floats<P>() / 2;                    // doubles per window region
    double* wl = reinterpret_cast<double*>(wbuf);
    const int nidx = g.LR + 2;                                       // weights of the frame
    const int ntiles = (g.len + kTile - 1) / kTile;   // 1 except for frames longer than 64 P samples (Q19)
    for (int t = 0; t < ntiles; ++t) {
        const int tile0 = t * kTile;
        const int hi = min(g.len, tile0 + kTile);
        stage_samples_async(g, tile0, kTile, xbuf_byte, lane);
        for (int seg0 = 0; seg0 < nidx; seg0 += capD) {
            const int seg1 = min(seg0 + capD, nidx);
            if (tabw) {
                // rising half: indices [seg0, min(L + 1, seg1)) from tl; falling half: [max(L + 1, seg0), seg1) from tr
                const int a0 = seg0, a1 = min(g.L + 1, seg1), b0 = max(g.L + 1, seg0), b1 = seg1;
                const float* srcl = reinterpret_cast<const float*>(tl + a0);
                const float* srcr = reinterpret_cast<const float*>(tr + (b0 - (g.L + 1)));
                const int nl = 2 * max(a1 - a0, 0), nr = 2 * max(b1 - b0, 0);   // dwords
                for (int c = 0; c < (nl + 63) >> 6; ++c) {
                    const unsigned m0v = __builtin_amdgcn_readfirstlane(wbuf_byte + 256u * (unsigned)c);
                    if (64 * c + lane < nl) {
                        const float* src = srcl + 64 * c + lane;
                        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(m0v) : "m0", "memory");
                    }
                }
                for (int c = 0; c < (nr + 63) >> 6; ++c) {
                    const unsigned m0v = __builtin_amdgcn_readfirstlane(wbuf_byte + 8u * (unsigned)(b0 - seg0) + 256u * (unsigned)c);
                    if (64 * c + lane < nr) {
                        const float* src = srcr + 64 * c + lane;
                        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(m0v) : "m0", "memory");
                    }
                }
            } else {
                for (int i = seg0 + lane; i < seg1; i += 64) {
                    const int k = (i <= g.L) ? i : g.LR - (i - (g.L + 1));
                    wl[i - seg0] = hann_half_f64(k, g.L, g.LR, g.kadd, invL, invR);
                }
            }
            staged_wait<0>();
            wave_sync();
            // the gather depends on this copy of the lane id: the compiler cannot start it ahead of the copies
            int lane_g = lane;
            asm volatile("" : "+v"(lane_g));
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const int m0 = 128 * j;
                // buffer index m holds windowed sample k = (m + rot) mod N, valid iff k < len
                const bool any = (m0 < g.len - g.rot) || (m0 + 127 >= N - g.rot);
                if (any) {
                    const int m = m0 + 2 * lane_g;
                    int k0 = m + g.rot;
                    k0 = (k0 >= N) ? k0 - N : k0;
                    int k1 = m + 1 + g.rot;
                    k1 = (k1 >= N) ? k1 - N : k1;
                    const int i0 = ((k0 <= g.L) ? k0 : g.L + 1 + (g.LR - k0)) - seg0;
                    const int i1 = ((k1 <= g.L) ? k1 : g.L + 1 + (g.LR - k1)) - seg0;
                    // (s_abs counts a register when it is WRITTEN: a window that takes several passes / tiles must not
                    // count it once per pass -- the flush threshold of the analytic path would grow with the pass count)
                    if (k0 >= tile0 && k0 < hi && i0 >= 0 && i0 < capD) {
                        const double v = (double)(xbuf[k0 - tile0] * in_scale) * wl[i0];
                        re[f64_in_reg<P>(j)] = v;
                        s_abs += fabs(v);
                    }
                    if (k1 >= tile0 && k1 < hi && i1 >= 0 && i1 < capD) {
                        const double v = (double)(xbuf[k1 - tile0] * in_scale) * wl[i1];
                        im[f64_in_reg<P>(j)] = v;
                        s_abs += fabs(v);
                    }
                }
            }
            wave_sync();
        }
    }
    float s_all = (float)s_abs;   // a scale: float is plenty
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s_all += __shfl_xor(s_all, d);
    const double zt = 2.842170943040401e-14 * (double)s_all;   // 2^-45 * sum: ~20 x the transform's error bound
#ifdef MPX_F64_NOFLUSH   // A/B of the policy (tools/fuzz_vs_oracle.py): the residue normalised like any other value
    const double zero2 = 1.0e-36 + 0.0 * zt;
#else
    // table path: the residue IS the reference's -- nothing is flushed (only |X|^2 too small for the fp32 seed: 1e-27 of
    // a sample after the 2^30 scaling); analytic path: the round-3 policy
    const double zero2 = tabw ? 1.0e-36 : fmax(zt * zt, 1.0e-36);
#endif

    if constexpr (kF64Dit) wave_fft_dit_f64<P, -1>(re, im, tw, xbuf, lane);
    else wave_fft_f64<P, -1>(re, im, tw, xbuf, lane);
    // scheduling fence: left alone, the last butterfly stage is interleaved with the split below and its inputs AND
    // outputs are live together (16 doubles spilled: 1 GB of scratch traffic per launch)
#pragma unroll
    for (int j = 0; j < P; j += 4)
        asm volatile("" : "+v"(re[j]), "+v"(re[j + 1]), "+v"(re[j + 2]), "+v"(re[j + 3]), "+v"(im[j]), "+v"(im[j + 1]),
                          "+v"(im[j + 2]), "+v"(im[j + 3]));

    zero2_out = zero2;
    mag_scale_out = mag_scale;
}

template <int P>
__global__ __launch_bounds__(kAna64Waves * 64) void k_analysis_f64(const float* __restrict__ sig,
                                                                  const long long* __restrict__ fpos,
                                                                  const int* __restrict__ fleft,
                                                                  const int* __restrict__ fright, long long nframes,
                                                                  const double* __restrict__ tw_g,
                                                                  float* __restrict__ omag, float* __restrict__ oreal,
                                                                  float* __restrict__ oimag, long long ld,
                                                                  const float* __restrict__ rows_in_use,
                                                                  const double* __restrict__ win_tab, int win_cap) {
    constexpr int M = 64 * P, N = 2 * M, LB = ilog2(P), kTile = 64 * P;
    extern __shared__ __attribute__((aligned(16))) double smem64[];
    double* tw = smem64;
    float* xbase = reinterpret_cast<float*>(smem64 + tw64_doubles<P>());
    const int lane_id = threadIdx.x & 63;
    const int wave = rfl((int)(threadIdx.x >> 6));
    float* xbuf = xbase + wave * (P * kXStride);
    const unsigned xbuf_byte = 8u * (unsigned)tw64_doubles<P>() + 4u * (unsigned)(wave * (P * kXStride));
    for (int i = threadIdx.x; i < tw64_doubles<P>(); i += kAna64Waves * 64) {
        // the global table is in the DIF form's register order (entry i = W_M^{l brev(i)}); the DIT form reads natural rows
        const int l = i / tw64_stride<P>(), c = i - l * tw64_stride<P>();
        const int src = (kF64Dit && c < 2 * P) ? l * tw64_stride<P>() + 2 * brev(c >> 1, LB) + (c & 1) : i;
        tw[i] = tw_g[src];
    }
    unsigned* queue = reinterpret_cast<unsigned*>(xbase + kAna64Waves * (P * kXStride));
    float* wbuf = xbase + kAna64Waves * (P * kXStride) + 4 + wave * f64_win_floats<P>();   // this wave's window region
    const unsigned wbuf_byte = 8u * (unsigned)tw64_doubles<P>() + 4u * (unsigned)(kAna64Waves * (P * kXStride) + 4 + wave * f64_win_floats<P>());
    if (threadIdx.x == 0) *queue = 0u;
    __syncthreads();

    // lane part of the split twiddle W_N^kappa = e^{-2 pi i kappa / N}
    double wl_s0, wl_c0;
    sincospi(-2.0 * (double)kappa<P>(lane_id) / (double)N, &wl_s0, &wl_c0);

    // the workgroup's frames are pulled from its LDS queue (queue_pull, mpx_common.hpp: a SIMD serves its waves by age)
    long long fb, fe;
    block_frame_range(nframes, fb, fe);
    for (long long f = queue_pull(queue, fb); f < fe; f = queue_pull(queue, fb)) {
        int lane = lane_id;   // laundered per frame: keeps per-lane products out of loop-invariant registers
        double wl_s = wl_s0, wl_c = wl_c0;
        asm volatile("" : "+v"(lane), "+v"(wl_s), "+v"(wl_c));
        const int kap = kappa<P>(lane);
        const int src_lane = kappa<P>((64 - kap) & 63);
        const bool lane0 = (kap == 0);
        double re[P], im[P];
        double zero2;
        float mag_scale;
        f64_frame_transform<P>(sig, fpos[f], fleft[f], fright[f], win_tab, win_cap, tw, xbuf, xbuf_byte, wbuf, wbuf_byte, lane, re,
                               im, zero2, mag_scale);

        // ---- real-FFT split, one (k, M-k) bin pair per step q (see k_analysis): lane kappa owns k = kappa + 64 q
        // (register brev(q)); Z[M-k] lives in lane (64-kappa)&63, register P-1-i (kappa == 0: own register of bin
        // (P-q)%P).  X[k] = E + T, X[M-k] = conj(E - T).  Bin M/2 is its own mirror (kappa == 0, register 1).
        float* row_m = omag + f * ld;
        float* row_r = oreal + f * ld;
        float* row_i = oimag + f * ld;
#ifndef MPX_F64_EPI_BATCH
#define MPX_F64_EPI_BATCH 4
#endif
        constexpr int EB = MPX_F64_EPI_BATCH;
        auto epilogue = [&](auto ph_tag) {
            constexpr bool PH = decltype(ph_tag)::value;
#pragma unroll
            for (int qb = 0; qb < P / 2; qb += EB) {
                double zpr[EB], zpi[EB];   // EB partner bins per batch: 4 EB lane exchanges in flight
#pragma unroll
                for (int u = 0; u < EB; ++u) {
                    const int i = f64_out_reg<P>(qb + u);   // Z[M - k]: register of row P - 1 - q (P - 1 - brev(q) == brev(P - 1 - q))
                    unsigned a, b, c, d;
                    split64(re[P - 1 - i], a, b);
                    split64(im[P - 1 - i], c, d);
                    zpr[u] = join64((unsigned)__shfl((int)a, src_lane), (unsigned)__shfl((int)b, src_lane));
                    zpi[u] = join64((unsigned)__shfl((int)c, src_lane), (unsigned)__shfl((int)d, src_lane));
                }
#pragma unroll
                for (int u = 0; u < EB; ++u) {
                    const int q = qb + u;
                    const int i = f64_out_reg<P>(q);
                    const int i0 = f64_out_reg<P>((P - q) % P);
                    const double pr = lane0 ? re[i0] : zpr[u];
                    const double pi = lane0 ? im[i0] : zpi[u];
                    const double er = 0.5 * (re[i] + pr), ei = 0.5 * (im[i] - pi);
                    const double orr = 0.5 * (im[i] + pi), oi = -0.5 * (re[i] - pr);
                    constexpr int kq = 64 / (2 * P);   // e^{-2 pi i q / (2P)} = W_64^{q kq}
                    const double cq = dc64(q * kq), sq = -ds64(q * kq);
                    const double wr = wl_c * cq - wl_s * sq, wi = wl_c * sq + wl_s * cq;
                    const double tr = wr * orr - wi * oi, ti = wr * oi + wi * orr;
                    const int k = kap + 64 * q;
                    feat_store<PH>(er + tr, ei + ti, zero2, mag_scale, row_m + k, row_r + k, row_i + k);
                    const int km = M - k;              // kappa == 0, q == 0: bin M
                    feat_store<PH>(er - tr, ti - ei, zero2, mag_scale, row_m + km, row_r + km, row_i + km);
                }
            }
            constexpr int ih = f64_out_reg<P>(P / 2);   // bin M/2
            if (lane0) feat_store<PH>(re[ih], -im[ih], zero2, mag_scale, row_m + M / 2, row_r + M / 2, row_i + M / 2);
        };
        // rows whose phase features no consumer reads (the compressed analysis: unvoiced stretches) get the magnitude only:
        // a third of the stores and two conversions per bin less, one wave-uniform branch per frame
        if (rows_in_use == nullptr || rfl((int)(rows_in_use[f] != 0.0f))) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    }
}

// ---------------------------------------------------------------------------------------------
// FUSED compressed analysis at the variable frame rate ("lossless features never hit HBM"):
// float64 analysis of 8 frames per workgroup round (one per wave) -> log-power / unit phasor of every bin straight from
// the registers into an LDS tile -> mel warp [8 frames x H] . W^T[H x (mag_dim + 2 phase_dim)] on the matrix cores
// (v_mfma_f32_16x16x4_f32, f32 in / f32 accumulate) -> mask / clip -> the compressed features.  What reaches HBM is the
// samples in and the 60 + 2 x 10 (or 45) coefficients per frame out; the staged form wrote 12 H bytes per frame and read
// them back (1.4 GB + 1.6 GB per 64 utterances).
//
// Round structure.  Phase A: every wave transforms its frame (f64_frame_transform: same code as k_analysis_f64).
// Phase B: P/2 chunk steps; step q splits the bin pair rows (kappa + 64 q, M - kappa - 64 q) exactly like k_analysis_f64's
// epilogue, applies the warp's prologue (the staged path's float32 values and formulas, so both paths agree to the
// summation order) and PUBLISHES them to As[buf][stream][wave][column]: column kappa = the low bin, 64 + kappa = its
// mirror.  After one barrier the eight waves split the chunk's 128 columns (16 each), read their A fragments and the
// matching W fragments from `wpack` (host-packed in fragment order: 16-byte loads, L2-resident,
// hostmath.pack_warp_fused), and accumulate: the magnitudes as 4 x 4 blocks on v_mfma_f32_4x4x1_16b_f32 (8 frames x 32
// coefficients per instruction, every product used), the two phase streams as rows 0..7 / 8..15 of 16 x 16 x 4 tiles
// (round 5; before, the magnitudes ran on 16 x 16 x 4 tiles whose rows 8..15 repeated rows 0..7).  The tile is double
// buffered: one barrier per chunk.  A fragment of the NEXT chunk is requested right behind the last use of this chunk's,
// and the compiler sinks the next chunk's operand arithmetic (float64 -> log / phase operands) into the loop: L2 latency
// and matrix instructions run under VALU work.
// Round end: the eight K-slices are added through LDS (deterministic order), bin M/2 (its own mirror, held by lane 0) is
// added as one product per output, epilogue as k_mel_warp_mfma's, rows stored.
// Accumulation: 256 terms per wave and accumulator + 8 partial sums -- the same error level as the staged GEMM's fresh
// accumulator per 64 bins + 33 chunk sums.
// ---------------------------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// Waves (= frames per round) of a workgroup.  8: one workgroup per CU; its transform (float64 VALU) and its GEMM (MFMA)
// alternate -- 0.96 ms per 57 k frames at 60 / 10 coefficients.  4 (-DMPX_FUSED_WAVES=4): TWO independent workgroups per
// CU, so that one's MFMA phase could run under the other's float64 VALU phase; the tiles are then a quarter / half full
// (twice the MFMA instructions per frame) and the kernel got SLOWER, 1.20-1.26 ms, with or without a raised priority of the
// GEMM phase (MPX_FUSED_PRIO): the two workgroups' phases do not interleave as hoped.  (Not a power effect: this kernel draws
// 899 W of the 1400 W cap, tools/power_lowdim.py.)
#ifndef MPX_FUSED_WAVES
#define MPX_FUSED_WAVES 8
#endif
constexpr int kFusedWaves = MPX_FUSED_WAVES;
static_assert(kFusedWaves == 4 || kFusedWaves == 8, "4 or 8 waves per fused workgroup");
constexpr int kFusedCols = 128 / kFusedWaves;      // tile columns per wave and chunk (its K slice): 16 or 32
constexpr int kFusedKH = kFusedCols / 16;          // 16-column groups per wave: 4 MFMA k-steps each
#ifndef MPX_FUSED_ASTRIDE
#define MPX_FUSED_ASTRIDE 136
#endif
// floats per published row: 128 columns + 8.  A fragment read is one ds_read_b128 at row (li & 7), column 16 h + 4 g: in the
// 16-lane groups a b128 read is served in ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: MI355X_MICROARCH.md) the lanes
// hold all eight rows at two adjacent g, so the 16-byte bank quads (row x stride / 4 + g) mod 16 must be distinct for row <
// 8, g in {0, 1}: stride / 4 = 2 mod 16.  (132: stride / 4 = 1 mod 16 -- row r at g + 1 met row r + 1 at g: 8.7 M conflict
// cycles per launch, round 5.)
constexpr int kFusedAStride = MPX_FUSED_ASTRIDE;
static_assert(kFusedAStride % 4 == 0 && kFusedAStride >= 128, "published rows: 128 columns, 16-byte aligned");

// floats per (wave, tile, register) row of the round's reduction buffer: 64 lanes + 4.  The output loop's 64 consecutive
// threads read column c of four consecutive tiles: 4 rows = 4 x 68 floats apart = 16 banks apart (64: the same bank, 4-way).
constexpr int kFusedRedStride = 68;
#ifndef MPX_FUSED_PAIR    // two chunks per barrier where four tile buffers fit the window regions (N = 4096) and it pays (fused_pair)
#define MPX_FUSED_PAIR 1
#endif
#ifndef MPX_FUSED_VOIBRANCH
#define MPX_FUSED_VOIBRANCH 1
#endif
#ifndef MPX_FUSED_F32NORM   // |X| and X / |X| of the float64 spectrum formed in float32 (see `operand`): measured error-neutral against
#define MPX_FUSED_F32NORM 1 // the oracle (worst case and sum of squares within 3 %), -1 % (60 / 10) / -3.4 % (60 / 45)
#endif
#ifndef MPX_FUSED_M4      // magnitude product on v_mfma_f32_4x4x1_16b_f32 (full blocks) -- the default; 0: 16 x 16 x 4 tiles
#define MPX_FUSED_M4 1
#endif

// the warp's operand prologue / epilogue (same formulas as magphase_comp.hip: warp_prologue / warp_epilogue)
__device__ __forceinline__ float fused_prologue_mag(int mode, float x) {
    if (mode == 0) return __builtin_amdgcn_logf(fmaf(x, x, 1.0e-8f)) * 0.69314718055994531f;
    return (x > 0.0f) ? __logf(x) : -1.0e10f;
}
// 2 x + 1e-8 e^{-2 x} (the phase streams' operand, warp_prologue's mode 1).  |x| <= 1 here (a component of X / |X|), and the
// exponential only scales a 1e-8 floor term: a quadratic fit of e^{-2x} on [-1, 1] (Chebyshev nodes, max error 0.55) moves
// the operand by < 6e-9 -- three fmas instead of a v_exp_f32 (quarter rate) and three more operations.
#ifndef MPX_FUSED_EXPFIT
#define MPX_FUSED_EXPFIT 1
#endif
__device__ __forceinline__ float fused_prologue_phase(float x) {
#if MPX_FUSED_EXPFIT
    const float t = fmaf(x, fmaf(x, 2.75579379e-8f, -3.18127371e-8f), 0.90168841e-8f);
    return fmaf(2.0f, x, t);
#else
    return fmaf(1.0e-8f, __expf(-2.0f * x), 2.0f * x);
#endif
}

// Operand values of one bin: (ln-power, phase operands of Re X/|X| and Im X/|X|) as the staged path forms them.  RAW: the
// magnitude operand is |X| itself -- the constant-rate form interpolates it between two frames BEFORE the logarithm
// (k_analysis_warp_fused_cr).
template <int MAGMODE, bool RAW>
__device__ __forceinline__ void fused_operand(double xr, double xi, double zero2, float mag_scale, bool voi, float& a_m, float& a_r,
                                              float& a_i) {
    const double s2 = xr * xr + xi * xi;
    const bool nz = s2 > zero2;
#if MPX_FUSED_F32NORM   // |X| and the unit phasor in float32 from the float64 spectrum: v_rsq_f32 (1 ulp) and float32 products, no
    // Newton step -- <= 2 ulp on operands that the float32 matrix product then sums over 2 049 bins (the staged path
    // rounds the float64 quotient once instead: k_analysis_f64's rows are API outputs, these operands are not)
    const float s2f = (float)s2;
    const float rf = nz ? __builtin_amdgcn_rsqf(s2f) : 0.0f;
    a_m = RAW ? (s2f * rf) * mag_scale : fused_prologue_mag(MAGMODE, (s2f * rf) * mag_scale);
    // (pinned: the two products were sunk to their first use after the phase loop, s2f AND rf of every bin live until then --
    // 64 registers instead of 32 and 200 spills)
    if (RAW) asm volatile("" : "+v"(a_m));
    if (voi) {
        a_r = fused_prologue_phase((float)xr * rf);
        a_i = fused_prologue_phase((float)xi * rf);
    } else {
        a_r = a_i = 0.0f;
    }
    return;
#endif
    const double rr = nz ? rsqrt_f64(s2) : 0.0;
    a_m = RAW ? (float)(s2 * rr) * mag_scale : fused_prologue_mag(MAGMODE, (float)(s2 * rr) * mag_scale);
    if (RAW) asm volatile("" : "+v"(a_m));
#if MPX_FUSED_VOIBRANCH   // a real (wave-uniform) branch: an unvoiced frame skips the phase operands' arithmetic instead of discarding it
    if (voi) {
        a_r = fused_prologue_phase((float)(xr * rr));
        a_i = fused_prologue_phase((float)(xi * rr));
    } else {
        a_r = a_i = 0.0f;
    }
#else
    a_r = voi ? fused_prologue_phase((float)(xr * rr)) : 0.0f;
    a_i = voi ? fused_prologue_phase((float)(xi * rr)) : 0.0f;
#endif
}

#ifndef MPX_FUSED_EB
#define MPX_FUSED_EB 2
#endif
// The split of every bin pair row of a frame's half-size transform (k_analysis_f64's epilogue arithmetic) into float32 operand
// registers: entry 2 q = the low bin kappa + 64 q of step q, 2 q + 1 = its mirror M - kappa - 64 q; (m0, m1, m2) = bin M/2
// (its own mirror; lane kappa == 0 holds it).
template <int P, int MAGMODE, bool RAW>
__device__ __forceinline__ void fused_split_operands(double (&re)[P], double (&im)[P], double zero2, float mag_scale, bool voi,
                                                     bool lane0, int src_lane, double wl_c, double wl_s, float (&vm)[P],
                                                     float (&vr)[P], float (&vi)[P], float& m0, float& m1, float& m2) {
    constexpr int EB = MPX_FUSED_EB;
#pragma unroll
    for (int qb = 0; qb < P / 2; qb += EB) {
        // (fence: the partner exchanges of LATER batches must not be scheduled ahead -- every batch in flight
        // holds 4 EB more registers while the float64 spectrum is still live)
        asm volatile("" : "+v"(re[f64_out_reg<P>(qb)]), "+v"(im[f64_out_reg<P>(qb)]));
        double zpr[EB], zpi[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) {   // Z[M - k]: lane (64 - kappa) & 63, register of row P - 1 - q
            const int i = f64_out_reg<P>(qb + u);
            unsigned a, b, c, d;
            split64(re[P - 1 - i], a, b);
            split64(im[P - 1 - i], c, d);
            zpr[u] = join64((unsigned)__shfl((int)a, src_lane), (unsigned)__shfl((int)b, src_lane));
            zpi[u] = join64((unsigned)__shfl((int)c, src_lane), (unsigned)__shfl((int)d, src_lane));
        }
#pragma unroll
        for (int u = 0; u < EB; ++u) {
            const int q = qb + u;
            const int i = f64_out_reg<P>(q);
            const int i0 = f64_out_reg<P>((P - q) % P);
            const double pr_ = lane0 ? re[i0] : zpr[u];
            const double pi_ = lane0 ? im[i0] : zpi[u];
            const double er = 0.5 * (re[i] + pr_), ei = 0.5 * (im[i] - pi_);
            const double orr = 0.5 * (im[i] + pi_), oi = -0.5 * (re[i] - pr_);
            constexpr int kq = 64 / (2 * P);
            const double cq = dc64(q * kq), sq = -ds64(q * kq);
            const double wr = wl_c * cq - wl_s * sq, wi = wl_c * sq + wl_s * cq;
            const double tr = wr * orr - wi * oi, ti = wr * oi + wi * orr;
            fused_operand<MAGMODE, RAW>(er + tr, ei + ti, zero2, mag_scale, voi, vm[2 * q], vr[2 * q], vi[2 * q]);
            fused_operand<MAGMODE, RAW>(er - tr, ti - ei, zero2, mag_scale, voi, vm[2 * q + 1], vr[2 * q + 1], vi[2 * q + 1]);
        }
    }
    constexpr int ih = f64_out_reg<P>(P / 2);
    fused_operand<MAGMODE, RAW>(re[ih], -im[ih], zero2, mag_scale, voi, m0, m1, m2);
}

// (measured, 57 k frames: 60 / 45 coefficients 1.043 -> 1.018 ms; 60 / 10: 0.893 -> 0.909 ms -- so only with three phase tiles)
template <int P, int NTP>
constexpr bool fused_pair() {
    return MPX_FUSED_PAIR != 0 && NTP >= 3 && kFusedWaves == 8 &&
           kFusedWaves * f64_win_floats<P>() >= 4 * 3 * kFusedWaves * kFusedAStride;
}

template <int P, int NTM, int NTP>
constexpr size_t lds_bytes_fused() {
    // [twiddles][transpose buffers][window regions, whose head doubles as the published tiles][bin M/2 values]; the round's
    // reduction buffer aliases the transpose buffers and the window regions
    static_assert(kFusedWaves * f64_win_floats<P>() >= 2 * 3 * kFusedWaves * kFusedAStride, "the tiles live inside the window regions");
    constexpr size_t tiles = NTM + NTP;
    constexpr size_t work = sizeof(float) * (size_t)(kFusedWaves * P * kXStride + kFusedWaves * f64_win_floats<P>());
    constexpr size_t red = sizeof(float) * (size_t)(kFusedWaves * tiles * 4 * kFusedRedStride);
    static_assert(work >= red, "the reduction buffer must fit the regions it aliases");
    return sizeof(double) * (size_t)tw64_doubles<P>() + work + sizeof(float) * 32;
}

template <int P, int NTM, int NTP, int MAGMODE>
__attribute__((amdgpu_waves_per_eu(2, 2)))   // <= 256 registers (VGPR + AGPR): two workgroups of four waves per CU
__global__ __launch_bounds__(kFusedWaves * 64) void k_analysis_warp_fused(
    const float* __restrict__ sig, const long long* __restrict__ fpos, const int* __restrict__ fleft,
    const int* __restrict__ fright, long long nframes, const double* __restrict__ tw_g,
    const double* __restrict__ win_tab, int win_cap, const float* __restrict__ wpack, const float* __restrict__ whalf,
    const float* __restrict__ voiced, int mag_dim, int phase_dim, float* __restrict__ omag, float* __restrict__ oreal,
    float* __restrict__ oimag) {
    // T column tiles per chunk: NTM for the magnitudes (rows 0..7 of the 16-row MFMA tile = the round's frames, rows 8..15
    // repeat them and are dropped) and NTP for BOTH phase streams at once -- they share W_phase, so rows 0..7 take the real
    // operands of the eight frames and rows 8..15 the imaginary ones: full tiles, half the phase MFMAs.
    constexpr int M = 64 * P, N = 2 * M, LB = ilog2(P), T = NTM + NTP;
    extern __shared__ __attribute__((aligned(16))) double smem64[];
    double* tw = smem64;
    float* xbase = reinterpret_cast<float*>(smem64 + tw64_doubles<P>());
    const int lane_id = threadIdx.x & 63;
    const int wave = rfl((int)(threadIdx.x >> 6));
    float* xbuf = xbase + wave * (P * kXStride);
    const unsigned xbuf_byte = 8u * (unsigned)tw64_doubles<P>() + 4u * (unsigned)(wave * (P * kXStride));
    float* As = xbase + kFusedWaves * (P * kXStride);          // [2][3][8][kFusedAStride]: the head of the window regions
    float* wbuf = As + wave * f64_win_floats<P>();                    // phase A: this wave's two half windows (f64_frame_transform)
    const unsigned wbuf_byte = 8u * (unsigned)tw64_doubles<P>() + 4u * (unsigned)(kFusedWaves * (P * kXStride) + wave * f64_win_floats<P>());
    float* red = xbase;                                         // round end: aliases the transpose buffers (and the windows)
    float* mid = xbase + kFusedWaves * (P * kXStride) + kFusedWaves * f64_win_floats<P>();   // [3][8]: bin M/2 of the round's frames
    for (int i = threadIdx.x; i < tw64_doubles<P>(); i += kFusedWaves * 64) {
        const int l = i / tw64_stride<P>(), c = i - l * tw64_stride<P>();
        const int src = (kF64Dit && c < 2 * P) ? l * tw64_stride<P>() + 2 * brev(c >> 1, LB) + (c & 1) : i;
        tw[i] = tw_g[src];
    }
    __syncthreads();

    double wl_s0, wl_c0;
    sincospi(-2.0 * (double)kappa<P>(lane_id) / (double)N, &wl_s0, &wl_c0);
    const long long nrounds = (nframes + kFusedWaves - 1) / kFusedWaves;
    const int li = lane_id & 15, g = lane_id >> 4;
    const f32x4_t* wp_wave = reinterpret_cast<const f32x4_t*>(wpack) + ((long long)wave * kFusedKH * T) * 64;   // wave-uniform

    for (long long rnd = blockIdx.x; rnd < nrounds; rnd += gridDim.x) {
        int lane = lane_id;   // laundered per round (see k_analysis_f64)
        double wl_s = wl_s0, wl_c = wl_c0;
        asm volatile("" : "+v"(lane), "+v"(wl_s), "+v"(wl_c));
        const int kap = kappa<P>(lane);
        const int src_lane = kappa<P>((64 - kap) & 63);
        const bool lane0 = (kap == 0);
        const long long f = rnd * kFusedWaves + wave;
        const bool has = f < nframes;                    // wave-uniform
        double re[P], im[P];
        double zero2 = 1.0e-36;
        float mag_scale = 1.0f;
        bool voi = false;
        if (has) {
            f64_frame_transform<P>(sig, fpos[f], fleft[f], fright[f], win_tab, win_cap, tw, xbuf, xbuf_byte, wbuf, wbuf_byte, lane,
                                   re, im, zero2, mag_scale);
            voi = rfl((int)(voiced[f] != 0.0f)) != 0;
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) re[j] = im[j] = 0.0;
        }

        // the split of every bin pair row into float32 operand registers (fused_split_operands): the float64 spectrum (128
        // registers) is dead after it; what stays live through the chunk loop is 3 x P floats
        float vm[P], vr[P], vi[P];
        {
            float m0, m1, m2;
            fused_split_operands<P, MAGMODE, false>(re, im, zero2, mag_scale, voi, lane0, src_lane, wl_c, wl_s, vm, vr, vi, m0, m1, m2);
            if (lane0) {
                mid[0 * kFusedWaves + wave] = m0;
                mid[1 * kFusedWaves + wave] = m1;
                mid[2 * kFusedWaves + wave] = m2;
            }
        }
        // a round without a voiced frame has nothing for the phase tiles to do (their outputs are masked to +0)
        // the float32 form's magnitude product on v_mfma_f32_4x4x1_16b_f32 (M4, see the chunk loop): two accumulators of
        // [8 frames x 32 coefficients] instead of four half-empty 16 x 16 tiles
        constexpr bool M4 = kFusedKH == 1 && MPX_FUSED_M4 != 0;
        constexpr int TR = M4 ? 2 + NTP : T;              // accumulator tiles per wave (= per K slice in `red`)
        f32x4_t acc[TR];
#pragma unroll
        for (int t = 0; t < TR; ++t) acc[t] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();   // the tiles overlay the other waves' window regions: every wave must be through its gather
#ifdef MPX_FUSED_PRIO   // the GEMM phase above the other workgroup's transform phase on the same SIMD (an MFMA needs one issue
        __builtin_amdgcn_s_setprio(MPX_FUSED_PRIO);   // slot per 32 cycles; losing it to the elder wave's VALU stream stalls the round)
#endif
        if constexpr (M4) {
            // MAGNITUDES on v_mfma_f32_4x4x1_16b_f32: 16 independent [4 x 1] . [1 x 4] blocks per instruction.  Block
            // b = lane >> 2 = (frame group fg = lane >> 5, column group cg = (lane >> 2) & 7): lane 4 b + x supplies the
            // operand of frame 4 fg + x and the weight of coefficient 4 cg + x = lane & 31 (+ 32 for the second half);
            // D: lane 4 b + j, register r = frame 4 fg + r, coefficient 4 cg + j (layout: tools/archive/mfma4x4_layout_probe.hip).
            // One instruction = 8 frames x 32 coefficients x 1 bin in 8 cycles, every product used; the 16 x 16 x 4 tiles
            // of the form below ran half empty (rows 8..15 repeat the frames): 256 matrix-pipe cycles per wave and chunk
            // for the 64 magnitude columns instead of 512.  The phase tiles (real rows 0..7, imaginary rows 8..15: full)
            // stay on v_mfma_f32_16x16x4_f32.  Weights: hostmath.pack_warp_fused(..., layout=1): per (chunk, wave) eight
            // magnitude fragments [bin group kg][half] then the phase fragments; a fragment is reloaded for the NEXT chunk
            // right behind its last use, so the L2 round trip runs under the other fragments' matrix instructions.
            constexpr int FR = 8 + NTP;
            const f32x4_t* wp4 = reinterpret_cast<const f32x4_t*>(wpack) + ((long long)wave * FR) * 64;   // wave-uniform
            f32x4_t bm[4][2], bp[NTP];
            {
                int wo = 0;
                asm volatile("" : "+s"(wo));   // (the offset, not the pointer: see the form below)
#pragma unroll
                for (int f = 0; f < 8; ++f) bm[f >> 1][f & 1] = wp4[wo + f * 64 + lane_id];
#pragma unroll
                for (int t = 0; t < NTP; ++t) bp[t] = wp4[wo + (8 + t) * 64 + lane_id];
            }
            const int arow4 = (4 * (lane_id >> 5) + (lane_id & 3)) * kFusedAStride + kFusedCols * wave;
            // N = 4096: FOUR tile buffers, two chunks published per barrier (half the barriers of a round: MPX_FUSED_PAIR)
            constexpr bool PAIR = fused_pair<P, NTP>();
            constexpr int NB = PAIR ? 4 : 2;
            auto publish = [&](int q) {
                float* row = As + (q & (NB - 1)) * (3 * kFusedWaves * kFusedAStride) + wave * kFusedAStride;
                row[kap] = vm[2 * q];
                row[64 + kap] = vm[2 * q + 1];
                row += kFusedWaves * kFusedAStride;
                row[kap] = vr[2 * q];
                row[64 + kap] = vr[2 * q + 1];
                row += kFusedWaves * kFusedAStride;
                row[kap] = vi[2 * q];
                row[64 + kap] = vi[2 * q + 1];
            };
#pragma unroll
            for (int q = 0; q < P / 2; ++q) {
                const float* tile = As + (q & (NB - 1)) * (3 * kFusedWaves * kFusedAStride);
                if (!PAIR) {
                    publish(q);
                    __syncthreads();
                } else if ((q & 1) == 0) {
                    publish(q);
                    publish(q + 1);
                    __syncthreads();
                }
                f32x4_t a4[4];
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) a4[kg] = *reinterpret_cast<const f32x4_t*>(tile + arow4 + 4 * kg);
                const f32x4_t ap = *reinterpret_cast<const f32x4_t*>(
                    tile + ((li < 8 ? 1 : 2) * kFusedWaves + (li & (kFusedWaves - 1))) * kFusedAStride + kFusedCols * wave + 4 * g);
                int wo = (q + 1 < P / 2 ? q + 1 : q) * (kFusedWaves * FR * 64);
                asm volatile("" : "+s"(wo));
                // FRESH accumulators per chunk (see the form below); two per half so that consecutive instructions are independent
                f32x4_t cm[2][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) cm[i >> 1][i & 1] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
#ifdef MPX_PROBE_FUSEDA_NOMFMA
                            cm[hf][e & 1][e] = fmaf(a4[kg][e], bm[kg][hf][e], cm[hf][e & 1][e]);
#else
                            cm[hf][e & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[kg][e], bm[kg][hf][e], cm[hf][e & 1], 0, 0, 0);
#endif
                        }
                    }
                    if (q + 1 < P / 2) {
                        bm[kg][0] = wp4[wo + (2 * kg) * 64 + lane_id];
                        bm[kg][1] = wp4[wo + (2 * kg + 1) * 64 + lane_id];
                    }
                }
#pragma unroll
                for (int t = 0; t < NTP; ++t) {
                    f32x4_t ca = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#ifdef MPX_PROBE_FUSEDA_NOMFMA
                        ca[e] = fmaf(ap[e], bp[t][e], ca[e]);
#else
                        ca = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[e], bp[t][e], ca, 0, 0, 0);
#endif
                    }
                    if (q + 1 < P / 2) bp[t] = wp4[wo + (8 + t) * 64 + lane_id];
                    acc[2 + t] += ca;
                }
                acc[0] += cm[0][0] + cm[0][1];
                acc[1] += cm[1][0] + cm[1][1];
            }
        } else
#pragma unroll
        for (int q = 0; q < P / 2; ++q) {
            float* tile = As + (q & 1) * (3 * kFusedWaves * kFusedAStride);
            {
                float* row = tile + wave * kFusedAStride;
                row[kap] = vm[2 * q];
                row[64 + kap] = vm[2 * q + 1];
                row += kFusedWaves * kFusedAStride;
                row[kap] = vr[2 * q];
                row[64 + kap] = vr[2 * q + 1];
                row += kFusedWaves * kFusedAStride;
                row[kap] = vi[2 * q];
                row[64 + kap] = vi[2 * q + 1];
            }
            // W fragments of this wave's 16 columns (one 16-byte load per tile, from a scalar base + the lane offset; the
            // base is laundered so that the compiler cannot hoist every chunk's loads to the top of the unrolled loop)
            // (the OFFSET is laundered, not the pointer: a pointer out of an asm statement is a generic one, its loads become
            // flat_load, those count on lgkmcnt too, and the barrier's LDS wait then waits for the weights as well -- round 5)
            int wo = q * (kFusedWaves * kFusedKH * T * 64);
            asm volatile("" : "+s"(wo));
            const f32x4_t* wq = wp_wave + wo;
            f32x4_t bw[T];   // the first 16-column group's fragments fly across the barrier; later groups reuse the registers
#pragma unroll
            for (int t = 0; t < T; ++t) bw[t] = wq[t * 64 + lane_id];
            __syncthreads();
            // a FRESH accumulator per chunk (16 or 32 terms), added to the round's totals afterwards: the partial sums of
            // one long float32 chain over operands of size ~10 cost 2e-6 on the phase features (the staged GEMM's two-level
            // accumulation, magphase_comp.hip)
            f32x4_t ca[T];
#pragma unroll
            for (int t = 0; t < T; ++t) ca[t] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int h = 0; h < kFusedKH; ++h) {
                // A fragments: magnitudes row li mod frames; phases: rows 0..7 the real operands, rows 8..15 the imaginary ones
                const int col = kFusedCols * wave + 16 * h + 4 * g;
                const f32x4_t am = *reinterpret_cast<const f32x4_t*>(tile + (li & (kFusedWaves - 1)) * kFusedAStride + col);
                const f32x4_t ap = *reinterpret_cast<const f32x4_t*>(
                    tile + ((li < 8 ? 1 : 2) * kFusedWaves + (li & (kFusedWaves - 1))) * kFusedAStride + col);
                if (h > 0) {
                    int who = h * (T * 64);
                    asm volatile("" : "+s"(who));
                    const f32x4_t* wh = wq + who;
#pragma unroll
                    for (int t = 0; t < T; ++t) bw[t] = wh[t * 64 + lane_id];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#ifdef MPX_PROBE_FUSEDA_NOMFMA   // ablation (timing only): one VALU operation per matrix instruction
#pragma unroll
                    for (int t = 0; t < NTM; ++t) ca[t][e] = fmaf(am[e], bw[t][e], ca[t][e]);
#pragma unroll
                    for (int t = NTM; t < T; ++t) ca[t][e] = fmaf(ap[e], bw[t][e], ca[t][e]);
#else
#pragma unroll
                    for (int t = 0; t < NTM; ++t) ca[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[e], bw[t][e], ca[t], 0, 0, 0);
#pragma unroll
                    for (int t = NTM; t < T; ++t) ca[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[e], bw[t][e], ca[t], 0, 0, 0);
#endif
                }
            }
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] += ca[t];
        }
#ifdef MPX_FUSED_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __syncthreads();   // every wave is done with the tiles and its transpose buffer: `red` may overwrite them
#pragma unroll
        for (int t = 0; t < TR; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave * TR + t) * 4 + r) * kFusedRedStride + lane_id] = acc[t][r];
        __syncthreads();
        // outputs of the round.  C of a tile: column c, row 4 g' + r.  Magnitude tiles: row = frame fr.  Phase tiles: row fr =
        // the real stream, row 8 + fr = the imaginary one.
        constexpr int kOuts = NTM + 2 * NTP;   // output column tiles per frame: magnitudes, real, imaginary
        for (int idx = threadIdx.x; idx < kFusedWaves * kOuts * 16; idx += kFusedWaves * 64) {
            const int c = idx & 15, to = (idx >> 4) % kOuts, fr = idx / (16 * kOuts);
            const long long fo = rnd * kFusedWaves + fr;
            if (fo >= nframes) continue;
            const int sa = (to < NTM) ? 0 : ((to < NTM + NTP) ? 1 : 2);
            const int t = (sa == 2) ? to - NTP : to;               // the MFMA tile that holds it
            const int row = (sa == 2) ? 8 + fr : fr;
            const int gg = row >> 2, r = row & 3;
            float y = 0.0f;
            if (M4 && sa == 0) {   // 4 x 4 blocks: coefficient n of frame fr is register fr & 3 of lane (n & 31) + 32 (fr >> 2), half n >> 5
                const int n = 16 * to + c;
#pragma unroll
                for (int w = 0; w < kFusedWaves; ++w)
                    y += red[((w * TR + (n >> 5)) * 4 + (fr & 3)) * kFusedRedStride + (n & 31) + 32 * (fr >> 2)];
            } else {
                const int tr = M4 ? t - NTM + 2 : t;
#pragma unroll
                for (int w = 0; w < kFusedWaves; ++w) y += red[((w * TR + tr) * 4 + r) * kFusedRedStride + c + 16 * gg];
            }
            y = fmaf(mid[sa * kFusedWaves + fr], whalf[t * 16 + c], y);
            if (sa == 0) {
                const int n = 16 * t + c;
                if (n >= mag_dim) continue;
                if (MAGMODE == 2) y = (y < -745.13321f) ? -1.0e10f : y;
                omag[fo * mag_dim + n] = y;
            } else {
                const int n = 16 * (t - NTM) + c;
                if (n >= phase_dim) continue;
                const float vo = voiced[fo];
                y = (vo == 0.0f) ? 0.0f : fminf(fmaxf(y * vo, -1.0f), 1.0f);
                (sa == 1 ? oreal : oimag)[fo * phase_dim + n] = y;
            }
        }
        __syncthreads();   // `red` / `mid` are reused by the next round
    }
}

// ---------------------------------------------------------------------------------------------
// The same launch at the CONSTANT frame rate (magphase.py:2967-2983, 2219-2239: the features of the pitch-synchronous frames
// are interpolated to a 5 ms grid BEFORE the warp).  Constant-rate frame c lies between the variable-rate frames row0[c] and
// row1[c] = row0[c] + 1 (or == row0[c]: the duplicated first row) with weight rowt[c]:
//   magnitudes  ln((|X_lo| + t (|X_hi| - |X_lo|))^2 + 1e-8) . W_mag -- the logarithm follows the interpolation, so the matrix
//               product's operand rows are the CONSTANT-rate frames: built in LDS from the published |X| rows of the round's
//               eight frames and of the frame before them (the halo), two passes of eight rows per sweep of the chunk loop
//               (a round with more than 16 constant-rate frames -- F0 below ~100 Hz -- sweeps again);
//   phases      linear up to the 1e-8 floor term of their operand: the product runs on the variable-rate frames as in
//               k_analysis_warp_fused and its phase_dim outputs are interpolated afterwards (k_warp_phase_rows,
//               magphase_comp.hip), exactly as the staged path does since round 3 (mpx_mel_warp_rows).
// A workgroup takes a CONTIGUOUS range of frames (frames_per_wg = 8 R - 1): its first round's window starts one frame early
// (that frame is transformed twice per launch boundary: once per workgroup), every later round finds the halo -- |X| of the
// previous round's last frame, 8 KB -- where wave 7 left it in global memory (`halo`, two round parities per workgroup; the
// LDS is full), fetched four chunks ahead of its use.  A constant-rate frame belongs to the round whose window holds
// row1[c]; cstart[f] = the first c with row1[c] >= f (k_cr_index).  Nothing of the lossless features reaches HBM: the staged
// pair k_analysis_f64 -> k_mel_warp_mfma writes and re-reads 24.6 KB per frame.
//
// Two chunk loops per round.  Phases: two chunks per barrier.  Magnitudes, one barrier per chunk: before barrier q every wave
// publishes its raw row of chunk q + 2 (three raw buffers); after it the matrix instructions read the operand rows of chunk q
// while the workgroup builds those of chunk q + 1 (two operand buffers).  Weight fragments are fetched two chunks ahead.
//
// Measured (round 6, configs[2]: 56 985 -> 63 926 frames): 1.36 ms against 0.98 ms of the staged pair, HBM traffic of the
// analysis side 0.37 GB against 2.87 GB.  Where the time goes (ablations, tools/ab_bench.py): transform + split 0.57 ms (as
// in k_analysis_warp_fused), phase loop 0.25 ms, magnitude sweeps 0.55 ms -- of which the weights' reload 0.11, the matrix
// instructions 0.10, the operand build 0.09, the raw publish 0.05, the halo 0.05: a chain of short steps between barriers
// in which the two waves of a SIMD do the same thing at the same time.  (One loop for both products, as in the variable-rate
// kernel: 60 registers spilled at N = 4096, 1.56 ms.)  The staged pair stays the default (engine.py: MAGPHASE_COMP_FUSED_CR).
// ---------------------------------------------------------------------------------------------
constexpr int kCrRows = 16;             // constant-rate frames per sweep: two passes of eight operand rows
constexpr int kCrRawRows = 9;          // published magnitude rows per chunk: |X| of the window's eight frames + the halo (row 8)
constexpr int kCrPhRows = 2 * 2 * 2 * kFusedWaves;   // the phase loop's tiles: two buffers of two chunks of [real 8][imaginary 8]
template <int P>
constexpr int cr_tile_floats() {
    // (the phase loop's two tiles of 16 rows lie over the same space)
    constexpr int mag_rows = 3 * kCrRawRows + 2 * kCrRows;   // three raw buffers, two operand buffers
    constexpr int win = kFusedWaves * f64_win_floats<P>(), tiles = (mag_rows > kCrPhRows ? mag_rows : kCrPhRows) * kFusedAStride;
    return win > tiles ? win : tiles;
}
template <int P>
constexpr int cr_halo_floats() { return 64 * P + 64; }   // per round parity: |X| registers of a frame [P][64] + bin M/2
template <int P, int NTP>
constexpr size_t lds_bytes_fused_cr() {
    constexpr size_t work = sizeof(float) * (size_t)(kFusedWaves * P * kXStride + cr_tile_floats<P>());
    constexpr size_t red = sizeof(float) * (size_t)(kFusedWaves * (4 + NTP) * 4 * kFusedRedStride);
    static_assert(work >= red, "the reduction buffer must fit the regions it aliases");
    return sizeof(double) * (size_t)tw64_doubles<P>() + work + sizeof(float) * (48 + 3 * kCrRows);   // + bin M/2 values [25], frame flags [8], the sweep's row table
}

__global__ __launch_bounds__(256) void k_cr_index(const int* __restrict__ row1, int n_const, long long nframes,
                                                  int* __restrict__ cstart) {
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    if (f > nframes) return;
    int lo = 0, hi = n_const;   // first c with row1[c] >= f (row1 ascends over the batch)
    while (lo < hi) {
        const int m = (lo + hi) >> 1;
        if ((long long)row1[m] < f) lo = m + 1; else hi = m;
    }
    cstart[f] = lo;
}

template <int P, int NTP>
__attribute__((amdgpu_waves_per_eu(2, 2)))
__global__ __launch_bounds__(kFusedWaves * 64) void k_analysis_warp_fused_cr(
    const float* __restrict__ sig, const long long* __restrict__ fpos, const int* __restrict__ fleft,
    const int* __restrict__ fright, long long nframes, const double* __restrict__ tw_g,
    const double* __restrict__ win_tab, int win_cap, const float* __restrict__ wpack, const float* __restrict__ whalf,
    const float* __restrict__ rows_in_use, const int* __restrict__ row0, const int* __restrict__ row1,
    const float* __restrict__ rowt, const int* __restrict__ cstart, int frames_per_wg, int mag_dim, int phase_dim,
    float* __restrict__ omag, float* __restrict__ tr_out, float* __restrict__ ti_out, float* halo) {
    static_assert(kFusedWaves == 8 && kFusedKH == 1 && MPX_FUSED_M4 != 0, "the constant-rate form is written for 8 waves and the 4 x 4 magnitude product");
    constexpr int M = 64 * P, N = 2 * M, LB = ilog2(P), NC = P / 2, AS = kFusedAStride;
    constexpr int TR = 4 + NTP;   // accumulators per wave: [pass][half] of the magnitudes, then the phase tiles
    constexpr int FR = 8 + NTP;   // weight fragments per (chunk, wave): hostmath.pack_warp_fused(..., layout=1)
    extern __shared__ __attribute__((aligned(16))) double smem64[];
    double* tw = smem64;
    float* xbase = reinterpret_cast<float*>(smem64 + tw64_doubles<P>());
    const int lane_id = threadIdx.x & 63;
    const int wave = rfl((int)(threadIdx.x >> 6));
    float* xbuf = xbase + wave * (P * kXStride);
    const unsigned xbuf_byte = 8u * (unsigned)tw64_doubles<P>() + 4u * (unsigned)(wave * (P * kXStride));
    float* rawb = xbase + kFusedWaves * (P * kXStride);   // [3][kCrRawRows][AS]: the head of the window regions
    float* abuf = rawb + 3 * kCrRawRows * AS;               // [2][kCrRows][AS]: the magnitudes' operand rows
    float* wbuf = rawb + wave * f64_win_floats<P>();
    const unsigned wbuf_byte = 8u * (unsigned)tw64_doubles<P>() + 4u * (unsigned)(kFusedWaves * (P * kXStride) + wave * f64_win_floats<P>());
    float* red = xbase;
    float* mid = rawb + cr_tile_floats<P>();   // bin M/2: [0..7] |X| of the window's frames, [8] the halo's, [9..16] real, [17..24] imaginary; [32..39] frame flags
    float* prm = mid + 48;                     // [kCrRows][3]: (row0, row1, weight) of the sweep's constant-rate frames
    for (int i = threadIdx.x; i < tw64_doubles<P>(); i += kFusedWaves * 64) {
        const int l = i / tw64_stride<P>(), c = i - l * tw64_stride<P>();
        const int src = (kF64Dit && c < 2 * P) ? l * tw64_stride<P>() + 2 * brev(c >> 1, LB) + (c & 1) : i;
        tw[i] = tw_g[src];
    }
    __syncthreads();

    double wl_s0, wl_c0;
    sincospi(-2.0 * (double)kappa<P>(lane_id) / (double)N, &wl_s0, &wl_c0);
    const int li = lane_id & 15, g = lane_id >> 4;
    const f32x4_t* wp4 = reinterpret_cast<const f32x4_t*>(wpack) + ((long long)wave * FR) * 64;   // wave-uniform
    float* hs = halo + (long long)blockIdx.x * (2 * cr_halo_floats<P>());
    const long long own_lo = (long long)blockIdx.x * frames_per_wg;
    const long long own_end = (own_lo + frames_per_wg < nframes) ? own_lo + frames_per_wg : nframes;
    const int arow4 = (4 * (lane_id >> 5) + (lane_id & 3)) * AS + kFusedCols * wave;
    const int jr = (int)(threadIdx.x >> 5), col4 = 4 * (int)(threadIdx.x & 31);   // operand-row build: row of the sweep, four columns

    int c_next0 = 0, c_next1 = 0;
    bool prm_ready = false;   // the first sweep's row table is already in `prm` (left there by the previous round)
    for (int r = 0;; ++r) {
        // window [a, a + 8); owned: the constant-rate frames with row1 in [a + (r == 0), min(a + 8, own_end))
        const long long a = own_lo - 1 + 8ll * r;
        const long long hi_lo = (r == 0) ? a + 1 : a;
        if (hi_lo >= own_end) break;
        const long long hi_hi = (a + 8 < own_end) ? a + 8 : own_end;
        // constant-rate frames of this round [c0, c1) and the end of the NEXT round's (its begin is this round's c1): the
        // scalar loads fly under the transform
        const int c0 = (r == 0) ? cstart[hi_lo] : c_next0;
        const int c1 = (r == 0) ? cstart[hi_hi] : c_next1;
        const bool more = a + 8 < own_end;   // another round follows
        c_next0 = c1;
        c_next1 = more ? cstart[(a + 16 < own_end) ? a + 16 : own_end] : c1;
        int lane = lane_id;   // laundered per round (see k_analysis_f64)
        double wl_s = wl_s0, wl_c = wl_c0;
        asm volatile("" : "+v"(lane), "+v"(wl_s), "+v"(wl_c));
        const int kap = kappa<P>(lane);
        const int src_lane = kappa<P>((64 - kap) & 63);
        const bool lane0 = (kap == 0);
        const long long f = a + wave;
        const bool has = f >= 0 && f < own_end;   // wave-uniform
        double re[P], im[P];
        double zero2 = 1.0e-36;
        float mag_scale = 1.0f;
        bool voi = false;
        if (has) {
            f64_frame_transform<P>(sig, fpos[f], fleft[f], fright[f], win_tab, win_cap, tw, xbuf, xbuf_byte, wbuf, wbuf_byte, lane,
                                   re, im, zero2, mag_scale);
            voi = rfl((int)(rows_in_use[f] != 0.0f)) != 0;
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) re[j] = im[j] = 0.0;
        }
        float vm[P], vr[P], vi[P];
        const bool halo_on = r >= 1;
        const float* hp = hs + ((r + 1) & 1) * cr_halo_floats<P>();   // where the previous round left its last frame
        {
            float m0, m1, m2;
            fused_split_operands<P, 0, true>(re, im, zero2, mag_scale, voi, lane0, src_lane, wl_c, wl_s, vm, vr, vi, m0, m1, m2);
            if (lane0) {
                mid[wave] = m0;
                mid[9 + wave] = m1;
                mid[17 + wave] = m2;
            }
            if (wave == kFusedWaves - 1) {   // the next round's halo
                float* hw = hs + (r & 1) * cr_halo_floats<P>();
#pragma unroll
                for (int j = 0; j < P; ++j) hw[j * 64 + lane_id] = vm[j];
                if (lane0) {
                    hw[64 * P] = m0;
                    if (halo_on) mid[8] = hp[64 * P];
                }
            }
        }
        // A round without a frame in use has nothing for the phase tiles to do.  (Not __syncthreads_or: its workgroup reduction
        // brings a static LDS variable with it, the dynamic array then no longer starts at LDS address 0 and the byte
        // offsets the transform's direct-to-LDS loads are given -- xbuf_byte, wbuf_byte -- point elsewhere.)
        if (lane_id == 0) mid[32 + wave] = voi ? 1.0f : 0.0f;
        __syncthreads();   // also the barrier the tiles need: they overlay the other waves' window regions
        bool any_ph = false;
#pragma unroll
        for (int w = 0; w < kFusedWaves; ++w) any_ph |= mid[32 + w] != 0.0f;
        any_ph = rfl((int)any_ph) != 0;

        // The halo row comes back from global memory (L2) kHaloAhead chunks ahead of its use: with one chunk of lead the whole
        // workgroup waited at every barrier for wave 7's load (1.2 us per chunk instead of 0.4).
        constexpr int kHaloAhead = 4;
        const bool halo_w = halo_on && wave == kFusedWaves - 1;   // wave-uniform: this wave also publishes the halo row
        float hq[kHaloAhead][2];
#pragma unroll
        for (int u = 0; u < kHaloAhead; ++u) hq[u][0] = hq[u][1] = 0.0f;
        if (halo_w && c1 > c0) {
#pragma unroll
            for (int u = 0; u < kHaloAhead; ++u) {
                hq[u][0] = hp[(2 * u) * 64 + lane_id];
                hq[u][1] = hp[(2 * u + 1) * 64 + lane_id];
            }
        }

        // ---- phase streams of the window's frames (variable rate): their own chunk loop, two chunks per barrier, so that the
        // 64 operand registers (vr, vi) are dead before the magnitude sweeps need theirs (one loop for both: 60 registers
        // spilled at N = 4096 and 1.56 ms instead of 1.38)
        f32x4_t accp[NTP];
#pragma unroll
        for (int t = 0; t < NTP; ++t) accp[t] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
        if (any_ph) {
            // weight fragments TWO chunks ahead (an L2 round trip is longer than one chunk of this loop: with the reload right
            // behind the use, as in k_analysis_warp_fused where the magnitude product covers it, every chunk waited for it)
            f32x4_t bp[2][NTP];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int wo = u * (kFusedWaves * FR * 64);
                asm volatile("" : "+s"(wo));   // (the offset, not the pointer: see k_analysis_warp_fused)
#pragma unroll
                for (int t = 0; t < NTP; ++t) bp[u][t] = wp4[wo + (8 + t) * 64 + lane_id];
            }
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                // buffer (q >> 1) & 1, half q & 1: [real rows 0..7][imaginary rows 8..15]
                float* tile = rawb + (((q >> 1) & 1) * 2 + (q & 1)) * (2 * kFusedWaves * AS);
                if ((q & 1) == 0) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        float* row = tile + u * (2 * kFusedWaves * AS) + wave * AS;
                        row[kap] = vr[2 * (q + u)];
                        row[64 + kap] = vr[2 * (q + u) + 1];
                        row += kFusedWaves * AS;
                        row[kap] = vi[2 * (q + u)];
                        row[64 + kap] = vi[2 * (q + u) + 1];
                    }
                    __syncthreads();
                }
                const f32x4_t ap = *reinterpret_cast<const f32x4_t*>(tile + li * AS + kFusedCols * wave + 4 * g);
                int wo = (q + 2 < NC ? q + 2 : q) * (kFusedWaves * FR * 64);
                asm volatile("" : "+s"(wo));
#pragma unroll
                for (int t = 0; t < NTP; ++t) {
                    f32x4_t ca = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) ca = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[e], bp[q & 1][t][e], ca, 0, 0, 0);
                    if (q + 2 < NC) bp[q & 1][t] = wp4[wo + (8 + t) * 64 + lane_id];
                    accp[t] += ca;
                }
            }
            __syncthreads();   // the magnitude sweep's first rows go where the last phase tiles were read
        }

        // ---- magnitudes: sweeps of up to kCrRows constant-rate frames
        for (int cb = c0, sweep = 0; sweep == 0 || cb < c1; cb += kCrRows, ++sweep) {
            const bool ph = sweep == 0 && any_ph;               // the phase accumulators go out with the first sweep
            const int nrows = (c1 - cb < kCrRows) ? c1 - cb : kCrRows;
            const int npass = nrows > 8 ? 2 : (nrows > 0 ? 1 : 0);
            // The sweep's row table in LDS: prm[j] = (row0, row1, weight) of constant-rate frame cb + j.  The first sweep of a
            // round finds it there (the previous round fetched it under its output stage); otherwise sixteen threads fetch it
            // now -- three dependent-free loads whose latency every thread of the workgroup would otherwise wait for twice.
            if (!(sweep == 0 && prm_ready)) {
                if (threadIdx.x < kCrRows && (int)threadIdx.x < nrows) {
                    prm[3 * threadIdx.x] = __int_as_float(row0[cb + threadIdx.x]);
                    prm[3 * threadIdx.x + 1] = __int_as_float(row1[cb + threadIdx.x]);
                    prm[3 * threadIdx.x + 2] = rowt[cb + threadIdx.x];
                }
                __syncthreads();
            }
            // this thread's operand row of the sweep: its two source rows (8 = the halo) and the weight; rows past the sweep's
            // end are built from row 0 and never read
            int lo_rel = 0, hi_rel = 0;
            float tt = 0.0f;
            if (jr < nrows) {
                lo_rel = __float_as_int(prm[3 * jr]) - (int)a;
                hi_rel = __float_as_int(prm[3 * jr + 1]) - (int)a;
                tt = prm[3 * jr + 2];
                if (lo_rel < 0) lo_rel = 8;
            }
            f32x4_t acc[4];   // [pass][half]
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            // The chunk loop for NP passes: straight-line code per pass count.  (A run-time `p < npass` inside the unrolled loop
            // put every chunk's accumulator update into a block of its own; the compiler sank all of them behind the loop and
            // kept 16 chunks' matrix results alive until then -- 200 registers spilled, 2.4 ms.  The accumulators are pinned
            // where they are formed for the same reason.)
            // Pipeline, one barrier per chunk: before barrier q every wave publishes its raw row of chunk q + 2 (three raw
            // buffers); after it the matrix instructions read the operand rows of chunk q while the workgroup builds those of
            // chunk q + 1 (two operand buffers) from the raw rows of chunk q + 1, published before barrier q - 1.
            auto chunks = [&](auto np_tag) {
                constexpr int NP = decltype(np_tag)::value;
                f32x4_t bm[2][4][2];   // weight fragments of two chunks: chunk q + 2's are fetched behind chunk q's last use
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    int wo = u * (kFusedWaves * FR * 64);
                    asm volatile("" : "+s"(wo));
#pragma unroll
                    for (int fi = 0; fi < 8; ++fi) bm[u][fi >> 1][fi & 1] = wp4[wo + fi * 64 + lane_id];
                }
                const bool builder = NP == 2 || wave < kFusedWaves / 2;   // wave-uniform: operand rows 8..15 belong to waves 4..7
                if (sweep > 0 && halo_w) {   // (the first sweep's ring was filled before the phase loop)
#pragma unroll
                    for (int u = 0; u < kHaloAhead; ++u) {
                        hq[u][0] = hp[(2 * u) * 64 + lane_id];
                        hq[u][1] = hp[(2 * u + 1) * 64 + lane_id];
                    }
                }
                auto publish = [&](int q) {
                    float* row = rawb + (q % 3) * (kCrRawRows * AS) + wave * AS;
                    row[kap] = vm[2 * q];
                    row[64 + kap] = vm[2 * q + 1];
                    if (halo_w) {
                        float* hr = rawb + (q % 3) * (kCrRawRows * AS) + 8 * AS;
                        hr[kap] = hq[q % kHaloAhead][0];
                        hr[64 + kap] = hq[q % kHaloAhead][1];
                        if (q + kHaloAhead < NC) {
                            hq[q % kHaloAhead][0] = hp[(2 * (q + kHaloAhead)) * 64 + lane_id];
                            hq[q % kHaloAhead][1] = hp[(2 * (q + kHaloAhead) + 1) * 64 + lane_id];
                        }
                    }
                };
                auto build = [&](int q) {   // operand row jr of chunk q: ln((lo + t (hi - lo))^2 + 1e-8), four columns per thread
                    const float* rb = rawb + (q % 3) * (kCrRawRows * AS);
                    const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(rb + lo_rel * AS + col4);
                    const f32x4_t x1 = *reinterpret_cast<const f32x4_t*>(rb + hi_rel * AS + col4);
                    f32x4_t av;
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[e] = fused_prologue_mag(0, fmaf(x1[e] - x0[e], tt, x0[e]));
                    *reinterpret_cast<f32x4_t*>(abuf + (q & 1) * (kCrRows * AS) + jr * AS + col4) = av;
                };
                publish(0);
                publish(1);
                __syncthreads();
                if (builder) build(0);
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const float* ab = abuf + (q & 1) * (kCrRows * AS);
                    if (q + 2 < NC) publish(q + 2);
                    __syncthreads();
                    int wo = (q + 2 < NC ? q + 2 : q) * (kFusedWaves * FR * 64);
                    asm volatile("" : "+s"(wo));
                    if (q + 1 < NC && builder) build(q + 1);
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        f32x4_t a4[4];
#pragma unroll
                        for (int kg = 0; kg < 4; ++kg) a4[kg] = *reinterpret_cast<const f32x4_t*>(ab + 8 * p * AS + arow4 + 4 * kg);
                        // FRESH accumulators per chunk, two per half (see k_analysis_warp_fused)
                        f32x4_t cm[2][2];
#pragma unroll
                        for (int i = 0; i < 4; ++i) cm[i >> 1][i & 1] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
#pragma unroll
                                for (int hf = 0; hf < 2; ++hf)
                                    cm[hf][e & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[kg][e], bm[q & 1][kg][hf][e], cm[hf][e & 1], 0, 0, 0);
                        acc[2 * p] += cm[0][0] + cm[0][1];
                        acc[2 * p + 1] += cm[1][0] + cm[1][1];
                        asm volatile("" : "+v"(acc[2 * p]), "+v"(acc[2 * p + 1]));   // (formed HERE, see above)
                    }
                    if (q + 2 < NC) {
#pragma unroll
                        for (int fi = 0; fi < 8; ++fi) bm[q & 1][fi >> 1][fi & 1] = wp4[wo + fi * 64 + lane_id];
                    }
                }
            };
            if (npass == 2) chunks(std::integral_constant<int, 2>{});
            else if (npass == 1) chunks(std::integral_constant<int, 1>{});
            // the NEXT round's first sweep: its row table is fetched now and goes to LDS behind this sweep's output stage
            const bool last_sweep = cb + kCrRows >= c1;
            const bool fetch = last_sweep && more && threadIdx.x < kCrRows && c_next0 + (int)threadIdx.x < c_next1;
            int n_lo = 0, n_hi = 0;
            float n_t = 0.0f;
            if (fetch) {
                n_lo = row0[c_next0 + threadIdx.x];
                n_hi = row1[c_next0 + threadIdx.x];
                n_t = rowt[c_next0 + threadIdx.x];
            }
            __syncthreads();   // every wave is done with the tiles and its transpose buffer: `red` may overwrite them
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) red[((wave * TR + t) * 4 + rr) * kFusedRedStride + lane_id] = acc[t][rr];
            if (ph) {
#pragma unroll
                for (int t = 0; t < NTP; ++t)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) red[((wave * TR + 4 + t) * 4 + rr) * kFusedRedStride + lane_id] = accp[t][rr];
            }
            __syncthreads();
            // magnitudes of the sweep's constant-rate frames: row jo = 8 p + fr of pass p is register fr & 3 of lane
            // (n & 31) + 32 (fr >> 2) of accumulator 2 p + (n >> 5)
            for (int idx = threadIdx.x; idx < kCrRows * 64; idx += kFusedWaves * 64) {
                const int jo = idx >> 6, n = idx & 63;
                if (jo >= nrows || n >= mag_dim) continue;
                const int p = jo >> 3, fr = jo & 7;
                float y = 0.0f;
#pragma unroll
                for (int w = 0; w < kFusedWaves; ++w)
                    y += red[((w * TR + 2 * p + (n >> 5)) * 4 + (fr & 3)) * kFusedRedStride + (n & 31) + 32 * (fr >> 2)];
                int lo = __float_as_int(prm[3 * jo]) - (int)a;
                const int hi = __float_as_int(prm[3 * jo + 1]) - (int)a;
                if (lo < 0) lo = 8;
                const float xm = fmaf(mid[hi] - mid[lo], prm[3 * jo + 2], mid[lo]);
                omag[(long long)(cb + jo) * mag_dim + n] = fmaf(fused_prologue_mag(0, xm), whalf[n], y);
            }
            if (ph) {   // phase coefficients of the window's frames, at the variable rate (no mask, no clip: k_warp_phase_rows)
                for (int idx = threadIdx.x; idx < kFusedWaves * 2 * NTP * 16; idx += kFusedWaves * 64) {
                    const int c = idx & 15, to = (idx >> 4) % (2 * NTP), fr = idx / (16 * 2 * NTP);
                    const long long fo = a + fr;
                    if (fo < 0 || fo >= own_end) continue;
                    const int sa = (to < NTP) ? 1 : 2;
                    const int tp = (sa == 2) ? to - NTP : to;
                    const int row = (sa == 2) ? 8 + fr : fr;
                    const int n = 16 * tp + c;
                    if (n >= phase_dim || mid[32 + fr] == 0.0f) continue;   // (the frame's flag: rows_in_use[fo])
                    float y = 0.0f;
#pragma unroll
                    for (int w = 0; w < kFusedWaves; ++w)
                        y += red[((w * TR + 4 + tp) * 4 + (row & 3)) * kFusedRedStride + c + 16 * (row >> 2)];
                    y = fmaf(mid[(sa == 1 ? 9 : 17) + fr], whalf[(4 + tp) * 16 + c], y);
                    (sa == 1 ? tr_out : ti_out)[fo * phase_dim + n] = y;
                }
            }
            __syncthreads();   // `red` and `prm` are reused by the next sweep / round
            if (fetch) {
                prm[3 * threadIdx.x] = __int_as_float(n_lo);
                prm[3 * threadIdx.x + 1] = __int_as_float(n_hi);
                prm[3 * threadIdx.x + 2] = n_t;
            }
            prm_ready = last_sweep && more;
        }
    }
}

// First-pass twiddle table in double (layout: wave_fft_f64.hpp).  One block per lane row, one thread per entry.
__global__ void k_tables_init_f64(int P, double* __restrict__ tab) {
    const int l = blockIdx.x, i = threadIdx.x;
    const int M = 64 * P, stride = 2 * P + 2;
    int lb = 0;
    while ((1 << lb) < P) ++lb;
    if (i < P) {
        int k1 = 0;
        for (int b = 0; b < lb; ++b) k1 |= ((i >> b) & 1) << (lb - 1 - b);
        const int r = (int)(((long long)l * k1) % M);
        double sn, cs;
        sincospi(2.0 * (double)r / (double)M, &sn, &cs);
        tab[l * stride + 2 * i + 0] = cs;
        tab[l * stride + 2 * i + 1] = sn;
    } else if (i == P) {
        tab[l * stride + 2 * i + 0] = 0.0;
        tab[l * stride + 2 * i + 1] = 0.0;
    }
}

}  // namespace mpx

using namespace mpx;

extern "C" {

size_t mpx_tables_f64_bytes(int fft_len) {
    const int P = p_of(fft_len);
    return P ? sizeof(double) * 64 * (size_t)(2 * P + 2) : 0;
}

int mpx_tables_f64_init(void* stream, int fft_len, void* tables) {
    const int P = p_of(fft_len);
    if (!P) return fail(MPX_ERR_ARG, "mpx_tables_f64_init: fft_len must be 1024, 2048 or 4096%s");
    if (!tables) return fail(MPX_ERR_ARG, "mpx_tables_f64_init: null tables%s");
    hipLaunchKernelGGL(k_tables_init_f64, dim3(64), dim3(64), 0, (hipStream_t)stream, P, (double*)tables);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_analysis_frames_f64(void* stream, int fft_len, const void* tables_f64, const float* sig, const int64_t* frame_pos,
                            const int32_t* frame_left, const int32_t* frame_right, int64_t n_frames, float* out_mag,
                            float* out_real, float* out_imag, int64_t ld, const float* rows_in_use) {
    return mpx_analysis_frames_f64w(stream, fft_len, tables_f64, sig, frame_pos, frame_left, frame_right, n_frames, out_mag,
                                    out_real, out_imag, ld, rows_in_use, nullptr, 0);
}

int mpx_analysis_frames_f64w(void* stream, int fft_len, const void* tables_f64, const float* sig, const int64_t* frame_pos,
                             const int32_t* frame_left, const int32_t* frame_right, int64_t n_frames, float* out_mag,
                             float* out_real, float* out_imag, int64_t ld, const float* rows_in_use, const double* win_tab,
                             int32_t win_cap) {
    const int P = p_of(fft_len);
    if (!P) return fail(MPX_ERR_ARG, "mpx_analysis_frames_f64: fft_len must be 1024, 2048 or 4096%s");
    if (n_frames < 0) return fail(MPX_ERR_ARG, "mpx_analysis_frames_f64: negative n_frames%s");
    if (ld < fft_len / 2 + 1) return fail(MPX_ERR_ARG, "mpx_analysis_frames_f64: ld < fft_len/2 + 1%s");
    if (win_tab && win_cap < 0) return fail(MPX_ERR_ARG, "mpx_analysis_frames_f64w: negative win_cap%s");
    if (n_frames == 0) return MPX_OK;
    if (!tables_f64 || !sig || !frame_pos || !frame_left || !frame_right || !out_mag || !out_real || !out_imag)
        return fail(MPX_ERR_ARG, "mpx_analysis_frames_f64: null pointer%s");
    const dim3 grid(grid_for(n_frames, kAna64Waves)), block(kAna64Waves * 64);
    hipStream_t s = (hipStream_t)stream;
#define MPX_LAUNCH_A64(PP)                                                                                        \
    do {                                                                                                          \
        if (int rc = set_lds(k_analysis_f64<PP>, lds_bytes_ana64<PP>())) return rc;                               \
        hipLaunchKernelGGL(k_analysis_f64<PP>, grid, block, lds_bytes_ana64<PP>(), s, sig,                        \
                           (const long long*)frame_pos, frame_left, frame_right, (long long)n_frames,             \
                           (const double*)tables_f64, out_mag, out_real, out_imag, (long long)ld, rows_in_use,    \
                           win_tab, (int)win_cap);                                                                \
    } while (0)
    if (P == 32) MPX_LAUNCH_A64(32);
    else if (P == 16) MPX_LAUNCH_A64(16);
    else MPX_LAUNCH_A64(8);
#undef MPX_LAUNCH_A64
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}


/* (declared in include/magphase_hip.h) */
int mpx_analysis_compressed_fused(void* stream, int fft_len, const void* tables_f64, const float* sig,
                                  const int64_t* frame_pos, const int32_t* frame_left, const int32_t* frame_right,
                                  int64_t n_frames, const double* win_tab, int32_t win_cap, const float* wpack,
                                  const float* whalf, int32_t mag_dim, int32_t phase_dim, const float* voiced,
                                  int32_t mag_fbank, float* out_mag, float* out_real, float* out_imag) {
    const int P = p_of(fft_len);
    if (P != 32 && P != 16) return fail(MPX_ERR_ARG, "mpx_analysis_compressed_fused: fft_len must be 2048 or 4096%s");
    if (n_frames < 0) return fail(MPX_ERR_ARG, "mpx_analysis_compressed_fused: negative n_frames%s");
    if (mag_dim <= 0 || mag_dim > 64 || phase_dim <= 0 || phase_dim > 48)
        return fail(MPX_ERR_ARG, "mpx_analysis_compressed_fused: mag_dim must be in 1..64 and phase_dim in 1..48%s");
    if (win_tab && win_cap < 0) return fail(MPX_ERR_ARG, "mpx_analysis_compressed_fused: negative win_cap%s");
    if (n_frames == 0) return MPX_OK;
    if (!tables_f64 || !sig || !frame_pos || !frame_left || !frame_right || !wpack || !whalf || !voiced || !out_mag ||
        !out_real || !out_imag)
        return fail(MPX_ERR_ARG, "mpx_analysis_compressed_fused: null pointer%s");
    const long long nrounds = (n_frames + kFusedWaves - 1) / kFusedWaves;
    const long long slots = (long long)device_cus() * (8 / kFusedWaves);   // resident workgroups: two per CU with four waves each
    const dim3 grid((unsigned)(nrounds < slots ? nrounds : slots)), block(kFusedWaves * 64);
    hipStream_t s = (hipStream_t)stream;
    const int ntp = (phase_dim + 15) / 16;
#define MPX_FUSED_GO(PP, NTP_, MM)                                                                                        \
    do {                                                                                                                  \
        if (int rc = set_lds(k_analysis_warp_fused<PP, 4, NTP_, MM>, (lds_bytes_fused<PP, 4, NTP_>()))) return rc;        \
        hipLaunchKernelGGL((k_analysis_warp_fused<PP, 4, NTP_, MM>), grid, block, (lds_bytes_fused<PP, 4, NTP_>()), s, sig, \
                           (const long long*)frame_pos, frame_left, frame_right, (long long)n_frames,                     \
                           (const double*)tables_f64, win_tab, (int)win_cap, wpack, whalf, voiced, (int)mag_dim,          \
                           (int)phase_dim, out_mag, out_real, out_imag);                                                  \
    } while (0)
#define MPX_FUSED_P(PP)                                                     \
    do {                                                                    \
        if (ntp == 1 && !mag_fbank) MPX_FUSED_GO(PP, 1, 0);                 \
        else if (ntp == 1) MPX_FUSED_GO(PP, 1, 2);                          \
        else if (ntp == 2 && !mag_fbank) MPX_FUSED_GO(PP, 2, 0);            \
        else if (ntp == 2) MPX_FUSED_GO(PP, 2, 2);                          \
        else if (!mag_fbank) MPX_FUSED_GO(PP, 3, 0);                        \
        else MPX_FUSED_GO(PP, 3, 2);                                        \
    } while (0)
    if (P == 32) MPX_FUSED_P(32);
    else MPX_FUSED_P(16);
#undef MPX_FUSED_P
#undef MPX_FUSED_GO
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

// Scratch of mpx_analysis_compressed_fused_cr: cstart (int32 x (n_frames + 1), padded to 256 bytes), then the workgroups' halos
int64_t mpx_analysis_compressed_fused_cr_work_bytes(int fft_len, int64_t n_frames) {
    const int P = p_of(fft_len);
    if ((P != 32 && P != 16) || n_frames < 0) return MPX_ERR_ARG;
    const int64_t idx = ((int64_t)sizeof(int32_t) * (n_frames + 1) + 255) / 256 * 256;
    return idx + (int64_t)sizeof(float) * device_cus() * 2 * (64 * P + 64);
}

/* (declared in include/magphase_hip.h) */
int mpx_analysis_compressed_fused_cr(void* stream, int fft_len, const void* tables_f64, const float* sig,
                                     const int64_t* frame_pos, const int32_t* frame_left, const int32_t* frame_right,
                                     int64_t n_frames, const double* win_tab, int32_t win_cap, const float* wpack,
                                     const float* whalf, int32_t mag_dim, int32_t phase_dim, const float* rows_in_use,
                                     const int32_t* row0, const int32_t* row1, const float* row_t, int64_t n_const,
                                     float* out_mag, float* tmp_real, float* tmp_imag, void* work) {
    const int P = p_of(fft_len);
    if (P != 32 && P != 16) return fail(MPX_ERR_ARG, "mpx_analysis_compressed_fused_cr: fft_len must be 2048 or 4096%s");
    if (n_frames < 0 || n_const < 0 || n_const > 2147483647ll || n_frames > 2147483647ll)
        return fail(MPX_ERR_ARG, "mpx_analysis_compressed_fused_cr: bad frame count%s");
    if (mag_dim <= 0 || mag_dim > 64 || phase_dim <= 0 || phase_dim > 48)
        return fail(MPX_ERR_ARG, "mpx_analysis_compressed_fused_cr: mag_dim must be in 1..64 and phase_dim in 1..48%s");
    if (win_tab && win_cap < 0) return fail(MPX_ERR_ARG, "mpx_analysis_compressed_fused_cr: negative win_cap%s");
    if (kFusedWaves != 8 || kFusedKH != 1 || MPX_FUSED_M4 == 0)
        return fail(MPX_ERR_ARG, "mpx_analysis_compressed_fused_cr: this build's fused kernel has another tile shape%s");
    if (n_frames == 0) return MPX_OK;
    if (!tables_f64 || !sig || !frame_pos || !frame_left || !frame_right || !wpack || !whalf || !rows_in_use || !work ||
        !tmp_real || !tmp_imag || (n_const > 0 && (!row0 || !row1 || !row_t || !out_mag)))
        return fail(MPX_ERR_ARG, "mpx_analysis_compressed_fused_cr: null pointer%s");
    hipStream_t s = (hipStream_t)stream;
    int* cstart = (int*)work;
    float* halo = (float*)((char*)work + ((int64_t)sizeof(int32_t) * (n_frames + 1) + 255) / 256 * 256);
    hipLaunchKernelGGL(k_cr_index, dim3((unsigned)((n_frames + 1 + 255) / 256)), dim3(256), 0, s, row1, (int)n_const,
                       (long long)n_frames, cstart);
    // contiguous frame ranges of 8 R - 1 frames (the first round's window starts one frame early), one workgroup per CU
    const long long slots = device_cus();
    const long long per = (n_frames + slots - 1) / slots;
    const long long fw = 8 * ((per + 1 + 7) / 8) - 1;
    const dim3 grid((unsigned)((n_frames + fw - 1) / fw)), block(kFusedWaves * 64);
    const int ntp = (phase_dim + 15) / 16;
#define MPX_FUSED_CR_GO(PP, NTP_)                                                                                       \
    do {                                                                                                                \
        if (int rc = set_lds(k_analysis_warp_fused_cr<PP, NTP_>, (lds_bytes_fused_cr<PP, NTP_>()))) return rc;          \
        hipLaunchKernelGGL((k_analysis_warp_fused_cr<PP, NTP_>), grid, block, (lds_bytes_fused_cr<PP, NTP_>()), s, sig, \
                           (const long long*)frame_pos, frame_left, frame_right, (long long)n_frames,                   \
                           (const double*)tables_f64, win_tab, (int)win_cap, wpack, whalf, rows_in_use, row0, row1,     \
                           row_t, (const int*)cstart, (int)fw, (int)mag_dim, (int)phase_dim, out_mag, tmp_real,         \
                           tmp_imag, halo);                                                                             \
    } while (0)
#define MPX_FUSED_CR_P(PP)                       \
    do {                                         \
        if (ntp == 1) MPX_FUSED_CR_GO(PP, 1);    \
        else if (ntp == 2) MPX_FUSED_CR_GO(PP, 2); \
        else MPX_FUSED_CR_GO(PP, 3);             \
    } while (0)
    if (P == 32) MPX_FUSED_CR_P(32);
    else MPX_FUSED_CR_P(16);
#undef MPX_FUSED_CR_P
#undef MPX_FUSED_CR_GO
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_analysis_compressed_fused_tiles(int32_t mag_dim, int32_t phase_dim, int32_t* ntm, int32_t* ntp) {
    if (mag_dim <= 0 || mag_dim > 64 || phase_dim <= 0 || phase_dim > 48 || !ntm || !ntp)
        return fail(MPX_ERR_ARG, "mpx_analysis_compressed_fused_tiles: bad arguments%s");
    *ntm = 4;                        // the magnitude job always runs four 16-wide column tiles (mag_dim <= 64)
    *ntp = (phase_dim + 15) / 16;
    return MPX_OK;
}

int mpx_analysis_compressed_fused_waves(void) { return kFusedWaves; }

// fragment layout mpx_analysis_compressed_fused expects in `wpack` (hostmath.pack_warp_fused's `layout`): 0 = one 16 x 16 x 4
// fragment per tile, 1 = eight 4 x 4 x 1 magnitude fragments [bin group][half], then the phase tiles' 16 x 16 x 4 fragments
int mpx_analysis_compressed_fused_layout(void) { return (kFusedKH == 1 && MPX_FUSED_M4 != 0) ? 1 : 0; }

// resident workgroups per CU of the fused kernel for (fft_len, phase_dim) as the runtime's occupancy query sees them, or < 0
int mpx_analysis_compressed_fused_blocks_per_cu(int fft_len, int32_t phase_dim) {
    const int P = p_of(fft_len);
    if ((P != 32 && P != 16) || phase_dim <= 0 || phase_dim > 48) return MPX_ERR_ARG;
    const int ntp = (phase_dim + 15) / 16;
    int nb = -1;
#define MPX_FUSED_OCC(PP, NTP_)                                                                                       \
    do {                                                                                                              \
        if (set_lds(k_analysis_warp_fused<PP, 4, NTP_, 0>, (lds_bytes_fused<PP, 4, NTP_>()))) return -1;              \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_analysis_warp_fused<PP, 4, NTP_, 0>, kFusedWaves * 64, \
                                                         (lds_bytes_fused<PP, 4, NTP_>())) != hipSuccess)             \
            return -1;                                                                                                \
    } while (0)
    if (P == 32) {
        if (ntp == 1) MPX_FUSED_OCC(32, 1); else if (ntp == 2) MPX_FUSED_OCC(32, 2); else MPX_FUSED_OCC(32, 3);
    } else {
        if (ntp == 1) MPX_FUSED_OCC(16, 1); else if (ntp == 2) MPX_FUSED_OCC(16, 2); else MPX_FUSED_OCC(16, 3);
    }
#undef MPX_FUSED_OCC
    return nb;
}   // frames per round = K slices of the packed weights

}  // extern "C"
