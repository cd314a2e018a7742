// magphase_hip.hip -- gfx950 kernels + C ABI (include/magphase_hip.h) for the MagPhase hot path.
//
// Kernel plan:
//   k_analysis<P>           one wavefront per frame, persistent waves (grid-stride over frames):
//                           gather+window+rotate -> 64*P-point complex FFT in registers/LDS -> real-FFT split
//                           -> mag/real/imag epilogue, 3 x H coalesced float stores.   HBM-write bound.
//   k_synth_lossless<P>     one wavefront per frame: 3 x H coalesced loads -> unit-phase spectrum ->
//                           Hermitian merge -> inverse FFT -> epoch-centred frame.      HBM-read bound.
//   k_ola_gather            one thread per output sample, ascending-frame gather (deterministic PSOLA).
//   k_synth_ola_pair<P>     the production form: synthesis + PSOLA fused, one LDS ring per wave pair (below).
// No MFMA in this file: nothing on the lossless path is a dense contraction; the mel
// warp / unwarp GEMMs of the compressed path (magphase_comp.hip) run on the fp32 MFMA.
#include "mpx_common.hpp"

namespace mpx {

// Feature stores of k_analysis: plain by default; MPX_ANA_NT makes them non-temporal (streaming: the rows are not
// read again by this kernel) -- an A/B knob, see DESIGN.md for what was measured.
#ifdef MPX_ANA_NT
#define MPX_ST(dst, val) __builtin_nontemporal_store((val), &(dst))
#else
#define MPX_ST(dst, val) ((dst) = (val))
#endif

#ifndef MPX_ANA_QUEUE
#define MPX_ANA_QUEUE 0
#endif
template <int P>
__global__ __launch_bounds__(kAnaThreads) void k_analysis(const float* __restrict__ sig,
                                                          const long long* __restrict__ fpos,
                                                          const int* __restrict__ fleft,
                                                          const int* __restrict__ fright, long long nframes,
                                                          const float* __restrict__ tw_g, float* __restrict__ omag,
                                                          float* __restrict__ oreal, float* __restrict__ oimag,
                                                          long long ld) {
    constexpr int M = 64 * P, N = 2 * M, LB = ilog2(P), kTile = 64 * P;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tw = smem;
    const int lane_id = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* xbuf = smem + tw_floats<P>() + wave * (P * kXStride);
    // byte address of xbuf in LDS (the dynamic segment starts at 0: the kernel has no static __shared__)
    const unsigned xbuf_byte = 4u * (unsigned)(tw_floats<P>() + rfl(wave) * (P * kXStride));
    for (int i = threadIdx.x; i < tw_floats<P>(); i += kAnaThreads) tw[i] = tw_g[i];
    unsigned* queue = reinterpret_cast<unsigned*>(smem + tw_floats<P>() + kAnaWaves * (P * kXStride));
    if (threadIdx.x == 0) *queue = 0u;
    __syncthreads();

    // lane part of the split twiddle W_N^kappa = e^{-2 pi i kappa / N}
    float wl_s0, wl_c0;
    sincospif(-2.0f * (float)kappa<P>(lane_id) / (float)N, &wl_s0, &wl_c0);

    // Frames: grid-stride by default (wave w takes frames w, w + W, ...: neighbouring rows of the three feature matrices
    // are written at the same time); MPX_ANA_QUEUE = 1 gives every workgroup a contiguous range pulled frame by frame
    // from an LDS counter (queue_pull, mpx_common.hpp) -- what balances the compute-bound frame kernels (k_analysis_f64,
    // k_noise_stats) measured 3 % SLOWER here (0.3285 vs 0.3183 ms): this kernel runs at the device's store ceiling, and
    // the store pattern, not the waves' age, is what it is sensitive to.
#if MPX_ANA_QUEUE
    long long fb, fe;
    block_frame_range(nframes, fb, fe);
    long long f = queue_pull(queue, fb);
#else
    const long long fb = 0, fe = nframes, fstep = (long long)gridDim.x * kAnaWaves;
    long long f = (long long)blockIdx.x * kAnaWaves + rfl(wave);
#endif
    if (f >= fe) return;

    // Software pipeline: the samples of the wave's next frame are copied HBM -> LDS (into the transpose buffer, idle
    // after the FFT's exchange) while this frame's second FFT pass and epilogue run.  All 99 stores of the epilogue
    // are issued after that copy, so "copy landed" == vmcnt <= 63: no wait on the store drain.
    FrameGeom g = frame_geom(sig, fpos[f], fleft[f], fright[f], N);
    stage_samples_async(g, 0, kTile, xbuf_byte, lane_id);
    staged_wait<0>();

    while (true) {
        // Launder the per-lane invariants once per frame: otherwise LICM hoists every (lane x register)
        // twiddle product out of this loop and the kernel spills.
        int lane = lane_id;
        float wl_s = wl_s0, wl_c = wl_c0;
        asm volatile("" : "+v"(lane), "+v"(wl_s), "+v"(wl_c));
        const int kap = kappa<P>(lane);
        const int src_lane = kappa<P>((64 - kap) & 63);
        const bool lane0 = (kap == 0);

        // ---- window in sample order (in place in LDS), then gather in FFT order
        float re[P], im[P];
#pragma unroll
        for (int j = 0; j < P; ++j) re[j] = im[j] = 0.0f;
        const int ntiles = (g.len + kTile - 1) / kTile;   // 1 except for frames longer than 64*P samples
        for (int t = 0; t < ntiles; ++t) {
            const int tile0 = t * kTile;
            if (t > 0) {                                   // rare slow path: not prefetched
                stage_samples_async(g, tile0, kTile, xbuf_byte, lane);
                staged_wait<0>();
            }
            const int hi = min(g.len, tile0 + kTile);
            for (int kb = tile0 + lane; kb < hi; kb += 256) {   // 4 rows per step: 4 LDS reads in flight, not 1
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (kb + 64 * r < hi) ? xbuf[kb + 64 * r - tile0] : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = kb + 64 * r;
                    if (k < hi) xbuf[k - tile0] = v[r] * hann_half(k, g.L, g.LR, g.kadd, g.invL, g.invR);
                }
            }
            wave_sync();
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const int m0 = 128 * j;
                // buffer index m holds windowed sample k = (m + rot) mod N, valid iff k < len
                const bool any = (m0 < g.len - g.rot) || (m0 + 127 >= N - g.rot);
                if (any) {
                    const int m = m0 + 2 * lane;
                    int k0 = m + g.rot;
                    k0 = (k0 >= N) ? k0 - N : k0;
                    int k1 = m + 1 + g.rot;
                    k1 = (k1 >= N) ? k1 - N : k1;
                    if (k0 >= tile0 && k0 < hi) re[j] = xbuf[k0 - tile0];
                    if (k1 >= tile0 && k1 < hi) im[j] = xbuf[k1 - tile0];
                }
            }
            wave_sync();
        }

        wave_fft_front<P, -1>(re, im, tw, xbuf, lane);

        // ---- the exchange buffer is idle from here on: start the copy of the next frame's samples into it
#if MPX_ANA_QUEUE
        const long long fn = queue_pull(queue, fb);
#else
        const long long fn = f + fstep;
#endif
        FrameGeom gn = g;
        if (fn < fe) {
            gn = frame_geom(sig, fpos[fn], fleft[fn], fright[fn], N);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the exchange's own LDS reads have returned
            stage_samples_async(gn, 0, kTile, xbuf_byte, lane);
        }

        fft_inreg<P, -1>(re, im);

        // Real-FFT split, one (k, M-k) bin pair per step q = 0 .. P/2-1: lane kappa owns k = kappa + 64 q (register
        // brev(q)) and also produces the mirrored bin M-k from the same E/T terms:
        //   E = (Z[k] + conj Z[M-k])/2, T = W_N^k (Z[k] - conj Z[M-k])/(2i), X[k] = E + T, X[M-k] = conj(E - T).
        // Z[M-k] lives in lane (64-kappa)&63, register P-1-i (lane 0: own register holding bin (P-q)%P).
        // Bin M/2 is its own mirror (lane 0, register 1: X = conj Z); bin M comes out of the k = 0 pair.
        //
        // Store shape.  Measured (tools/ab_bench.py, 12 waves per CU): what costs is a 128-byte line that leaves L2
        // partially written.  So (1) the steps run in natural q order -- each stream writes consecutive 256-byte
        // blocks back to back and the shared lines between blocks complete at once (bit-reversed order: +35 %);
        // (2) the mirrored bins of step q, M-64q-63 .. M-64q, are regrouped into blocks that start at a multiple of
        // 64 floats from the row start, like the ascending stream's: lanes 1..63 hold the upper 63 floats of block
        // S_q = [M-64q-64, M-64q-1], its lowest float, bin M-64(q+1), is lane 0's output of step q+1, so S_q is
        // stored during step q+1 with data (lane 0 ? step q+1 : step q); lane 0's bin M (step 0) is a single-lane
        // store and the lowest float of the last block is bin M/2.  With 128-byte aligned rows every block is then
        // two full lines; with the dense pitch ld = H it is still 3 % faster than the unshifted form.
        // (3) Dense rows (ld = H) beat padded ones (H -> 2080 / 2112: +7 %): the lone bin M shares its line with the
        // next row, written by the neighbouring wave, instead of leaving a partial line per row.
        float* row_m = omag + f * ld;
        float* row_r = oreal + f * ld;
        float* row_i = oimag + f * ld;
        float* mlo = row_m + kap;                     // X[k]   : ascending lanes, +64 q
        float* rlo = row_r + kap;
        float* ilo = row_i + kap;
        const int hoff = lane0 ? M - 64 : M - kap;    // X[M-k] : descending lanes, -64 q (lane 0: one block lower)
        float* mhi = row_m + hoff;
        float* rhi = row_r + hoff;
        float* ihi = row_i + hoff;
        float zpr[P / 2], zpi[P / 2];   // all partner bins first: P lane exchanges in flight together
#pragma unroll
        for (int q = 0; q < P / 2; ++q) {
            const int i = brev(q, LB);
            zpr[q] = __shfl(re[P - 1 - i], src_lane);
            zpi[q] = __shfl(im[P - 1 - i], src_lane);
        }
        float hm = 0.0f, hr = 0.0f, hi_ = 0.0f;       // mirror outputs of the previous step
#pragma unroll
        for (int q = 0; q < P / 2; ++q) {
            const int i = brev(q, LB);                // even register
            const int i0 = brev((P - q) % P, LB);
            const float pr = lane0 ? re[i0] : zpr[q];
            const float pi = lane0 ? im[i0] : zpi[q];
            const float er = 0.5f * (re[i] + pr), ei = 0.5f * (im[i] - pi);
            const float orr = 0.5f * (im[i] + pi), oi = -0.5f * (re[i] - pr);
            const float cq = cos2p<P>(q), sq = -sin2p<P>(q);   // W_N^k = W_N^kappa * e^{-2 pi i q/(2P)}
            const float wr = wl_c * cq - wl_s * sq, wi = wl_c * sq + wl_s * cq;
            const float tr = wr * orr - wi * oi, ti = wr * oi + wi * orr;
            {
                const float xr = er + tr, xi = ei + ti;
                const float s2 = xr * xr + xi * xi;
                const float r = __builtin_amdgcn_rsqf(fmaxf(s2, 1.0e-37f));   // X == 0 -> xr = xi = s2 = 0 -> all three outputs 0
#ifdef MPX_PROBE_NOSTORE
                asm volatile("" ::"v"(s2 * r), "v"(xr * r), "v"(xi * r));
#else
                MPX_ST(mlo[64 * q], s2 * r);
                MPX_ST(rlo[64 * q], xr * r);
                MPX_ST(ilo[64 * q], xi * r);
#endif
            }
            {
                const float xr = er - tr, xi = ti - ei;
                const float s2 = xr * xr + xi * xi;
                const float r = __builtin_amdgcn_rsqf(fmaxf(s2, 1.0e-37f));
                const float cm = s2 * r, cr = xr * r, ci = xi * r;
#ifdef MPX_PROBE_NOSTORE
                asm volatile("" ::"v"(cm), "v"(cr), "v"(ci));
#else
                if (q == 0) {
                    if (lane0) {                      // bin M
                        MPX_ST(row_m[M], cm);
                        MPX_ST(row_r[M], cr);
                        MPX_ST(row_i[M], ci);
                    }
                } else {                              // block S_{q-1}
                    MPX_ST(mhi[-64 * (q - 1)], lane0 ? cm : hm);
                    MPX_ST(rhi[-64 * (q - 1)], lane0 ? cr : hr);
                    MPX_ST(ihi[-64 * (q - 1)], lane0 ? ci : hi_);
                }
#endif
                hm = cm;
                hr = cr;
                hi_ = ci;
            }
        }
        {   // block S_{P/2-1} = [M/2, M/2+63]: lane 0 supplies bin M/2 (register 1 holds q = P/2): X = conj Z
            const float xr = re[1], xi = -im[1];
            const float s2 = xr * xr + xi * xi;
            const float r = __builtin_amdgcn_rsqf(fmaxf(s2, 1.0e-37f));
#ifdef MPX_PROBE_NOSTORE
            asm volatile("" ::"v"(s2 * r), "v"(xr * r), "v"(xi * r), "v"(hm), "v"(hr), "v"(hi_));
#else
            MPX_ST(mhi[-64 * (P / 2 - 1)], lane0 ? s2 * r : hm);
            MPX_ST(rhi[-64 * (P / 2 - 1)], lane0 ? xr * r : hr);
            MPX_ST(ihi[-64 * (P / 2 - 1)], lane0 ? xi * r : hi_);
#endif
        }
        g = gn;
        // The epilogue issued 3P + 3 stores after the copy's loads (3 per step q for X[k], 3 for bin M or a mirror block,
        // 3 for the last block): at most that many operations outstanding <=> the copy has landed.  (P = 32: 99 > 63,
        // the counter's range; P = 16 / 8: 51 / 27 -- a fixed 63 would not wait at all there.)
        if (fn >= fe) break;
#ifdef MPX_PROBE_NOSTORE
        staged_wait<0>();
#else
        staged_wait<3 * P + 3>();
#endif
        f = fn;
    }
}

// ---------------------------------------------------------------------------------------------
// lossless synthesis: per-frame spectrum rebuild + inverse real FFT (epoch at N/2)
// ---------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(kThreads) void k_synth_lossless(const float* __restrict__ mag,
                                                             const float* __restrict__ real,
                                                             const float* __restrict__ imag, long long nframes,
                                                             const float* __restrict__ tw_g,
                                                             float* __restrict__ frames, long long ld) {
    constexpr int M = 64 * P, N = 2 * M, H = M + 1, LB = ilog2(P);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tw = smem;
    const int lane_id = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* xbuf = smem + tw_floats<P>() + wave * (P * kXStride);
    for (int i = threadIdx.x; i < tw_floats<P>(); i += kThreads) tw[i] = tw_g[i];
    __syncthreads();

    // input layout is natural (k = lane + 64 j): lane twiddle conj(W_N^lane) = e^{+2 pi i lane/N}
    float wl_s0, wl_c0;
    sincospif(2.0f * (float)lane_id / (float)N, &wl_s0, &wl_c0);

    const int wave_u = rfl(wave);
    for (long long f = (long long)blockIdx.x * kWavesPerBlock + wave_u; f < nframes;
         f += (long long)gridDim.x * kWavesPerBlock) {
        int lane = lane_id;  // laundered per frame (see k_analysis)
        float wl_s = wl_s0, wl_c = wl_c0;
        asm volatile("" : "+v"(lane), "+v"(wl_s), "+v"(wl_c));
        const int kap = kappa<P>(lane);
        FrameFeat<P> ff;
        feat_load<P>(ff, mag + f * ld, real + f * ld, imag + f * ld, lane);
        float xr[P], xi[P], xm;
        feat_convert<P>(ff, xr, xi, xm, lane);
        hermitian_merge<P>(xr, xi, xm, lane, wl_c, wl_s);

        wave_fft<P, +1>(xr, xi, tw, xbuf, lane);

        float2* out = reinterpret_cast<float2*>(frames + f * N);
#pragma unroll
        for (int i = 0; i < P; ++i) out[kap + 64 * brev(i, LB)] = make_float2(xr[i], xi[i]);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused lossless synthesis + PSOLA.  The frames of every utterance are cut into RUNS of consecutive frames (host:
// hostmath.ola_runs -- about total_frames / wave-pair-slots frames each, so every slot gets one run of equal length);
// a PAIR of wavefronts owns a run: the two waves rebuild alternate frames concurrently (as k_synth_lossless) and enter
// the overlap-add strictly in frame order (a ticket word in LDS; LDS serves one CU's waves in order, so "ticket seen"
// implies the partner's ring writes are done).  The sum lives in a ring buffer in LDS; what no later frame of the run
// can reach any more streams out -- straight to pcm_out for the positions only this run (or this run first) contributes
// to, to a small head strip for the first <= N positions, which the previous run's last frames also reach;
// k_ola_fixup adds those strips afterwards (previous run's sum + head strip: fixed order, deterministic).
// Why pairs: one ring per wave (16.5 KB) would cap the kernel at 5 waves per CU, and a lone wave cannot hide its own LDS /
// memory latencies.  HBM traffic: features read once, every output sample written once, + (N + 64) floats written and
// read once more per run boundary (the [F x N] frame scratch of the two-kernel form, 16 KB per frame each way, and the
// territory strips of the first fused form, T + N floats per T output samples each way, are gone).
// ---------------------------------------------------------------------------------------------
#ifndef MPX_SYN_PAIR_WAVES
#define MPX_SYN_PAIR_WAVES 12
#endif
#ifndef MPX_SYN_GROUP
#define MPX_SYN_GROUP 2
#endif
// Waves per workgroup: 12 = 3 per SIMD (<= 168 VGPRs).  A wave issues at most one VALU instruction per ~4-5 cycles while a
// SIMD can start one every 2: with 2 waves per SIMD (round 2: 243 VGPRs, 8 waves) the kernel ran its VALU 33 % and its
// memory pipe 47 % busy -- bound by neither, by latency.  The third wave needs (a) the LDS: six rings + twelve transpose
// buffers + the twiddle table are 223 KB in the round-2 form; P == 32 therefore uses the COMPACT transform front of
// wave_fft.hpp (half-height transpose buffer, half twiddle table: 162.9 KB), and (b) <= 168 VGPRs: the prefetch of the
// next frame's 99 feature values is issued in THREE parts as registers come free (after the transform; between the two
// planes of the overlap-add; after the ordered section) and the overlap-add reads the ring 16 values at a time.
constexpr int kPairWaves = MPX_SYN_PAIR_WAVES;   // waves per workgroup
constexpr int kGroup = MPX_SYN_GROUP;            // waves sharing one ring (2: the "pair"; 4 at the same occupancy measured
                                                 // +11 %, 6 with 12 waves per CU +15 %: the in-order ring hand-over couples the
                                                 // waves of a group, so more waves only pay with their own rings)
constexpr int kPairs = kPairWaves / kGroup;      // rings (= work-list slots) per workgroup
static_assert(kPairWaves % kGroup == 0, "waves per workgroup must be a multiple of the group size");

#ifndef MPX_PRIO_ROTATE
#define MPX_PRIO_ROTATE 0
#endif
#ifndef MPX_PRIO_SHIFT
#define MPX_PRIO_SHIFT 10
#endif
#ifdef MPX_PROBE_ENDTIME
// Probe build (tools/archive/endtime_probe.py): every wave of k_synth_ola_pair stores the constant-rate clock (100 MHz) when it
// enters and when it leaves its frame loop, and its frame count -- the spread of the end times is the launch tail.
__device__ unsigned long long g_endprobe[4 * 8192];   // per wave: start, end (100 MHz clock), frames, shader cycles
#ifdef MPX_PROBE_PHASES
// per wave: s_memtime ticks spent in 8 phases of the frame loop (feature wait, merge, transform, prefetch issue + scalars,
// ticket wait, flush, overlap-add, tail) -- tools/archive/phase_probe.py
__device__ unsigned long long g_phaseprobe[8 * 8192];
#define MPX_PHASE(i)                                                                              \
    do {                                                                                          \
        const unsigned t_ = (unsigned)clock64();                                                  \
        if (lane_id == 0) atomicAdd(probe_ph + (i), t_ - probe_last);   /* LDS: ds_add_u32 */     \
        probe_last = t_;                                                                          \
    } while (0)
#else
#define MPX_PHASE(i) do { } while (0)
#endif
#else
#define MPX_PHASE(i) do { } while (0)
#endif

template <int P>
constexpr bool pair_compact() { return P == 32 && kPairWaves > 8; }
template <int P>
constexpr int pair_tw_floats() { return pair_compact<P>() ? tw_half_floats<P>() : tw_floats<P>(); }
template <int P>
constexpr int pair_xbuf_floats() { return (pair_compact<P>() ? P / 2 : P) * kXStride; }
template <int P>
constexpr size_t lds_bytes_pair() {
#ifdef MPX_PROBE_PHASES
    return sizeof(float) * (size_t)(pair_tw_floats<P>() + kPairWaves * pair_xbuf_floats<P>() + kPairs * ring_len<P>() + 16 + 8 * kPairWaves);
#else
    return sizeof(float) * (size_t)(pair_tw_floats<P>() + kPairWaves * pair_xbuf_floats<P>() + kPairs * ring_len<P>() + 16);
#endif
}

template <int P>
__global__ __launch_bounds__(kPairWaves * 64) void k_synth_ola_pair(const float* __restrict__ mag,
                                                                    const float* __restrict__ real,
                                                                    const float* __restrict__ imag,
                                                                    const RunDesc* __restrict__ runs,
                                                                    const int* __restrict__ slot_off,
                                                                    const int* __restrict__ slot_runs, int nslots,
                                                                    const int* __restrict__ pm_rel,
                                                                    const float* __restrict__ tw_g,
                                                                    float* __restrict__ strips,
                                                                    float* __restrict__ pcm, long long ld) {
    constexpr int M = 64 * P, N = 2 * M, R = ring_len<P>();
    constexpr bool kCompact = pair_compact<P>();
    // feature prefetch in parts (bin pairs j: six values each): j < JA and bin M/2 right after the transform, JA <= j < JB
    // between the two planes of the overlap-add (the first plane's 32 registers are free then), the rest after the
    // ordered section.  The short transforms have the registers for one part.  (JA, JB, CH) = (6, 11, 8) is the largest
    // first part that compiles without a spill at 168 VGPRs (8 / 13 / 16: 20 spilled registers; tools/kres.py).
#ifndef MPX_JA
#define MPX_JA 6
#endif
#ifndef MPX_JB
#define MPX_JB 11
#endif
#ifndef MPX_CH
#define MPX_CH 8
#endif
    constexpr int JA = kCompact ? MPX_JA : P / 2, JB = kCompact ? MPX_JB : P / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tw = smem;
    const int lane_id = threadIdx.x & 63;
    const int wave = rfl(threadIdx.x >> 6);
    const int pair = wave / kGroup, half = wave % kGroup;   // ring, member index
    float* xbuf = smem + pair_tw_floats<P>() + wave * pair_xbuf_floats<P>();
    constexpr int kRing0 = pair_tw_floats<P>() + kPairWaves * pair_xbuf_floats<P>();   // first ring (floats from the LDS base)
    float* ring = smem + kRing0 + pair * R;
    const unsigned ring_byte = 4u * (unsigned)(kRing0 + pair * R);
    int* turn = reinterpret_cast<int*>(smem + kRing0 + kPairs * R) + pair;
    if constexpr (kCompact) {   // even registers' twiddles only: row l = (P/2 complex + 4 pad floats) out of the full table's row
        for (int i = threadIdx.x; i < tw_half_floats<P>(); i += kPairWaves * 64) {
            const int l = i / tw_half_stride<P>(), c = i % tw_half_stride<P>();
            // natural register order for the DIT first pass: entry e = W_M^{l e}, which the full table keeps at register brev(e)
            tw[i] = (c < P) ? tw_g[l * tw_stride<P>() + 2 * brev(c >> 1, ilog2(P)) + (c & 1)] : 0.0f;
        }
    } else {
        for (int i = threadIdx.x; i < tw_floats<P>(); i += kPairWaves * 64) tw[i] = tw_g[i];
    }
    for (int i = threadIdx.x; i < kPairs * R; i += kPairWaves * 64) smem[kRing0 + i] = 0.0f;
    if (threadIdx.x < kPairs) turn[threadIdx.x - pair] = 0;   // thread t < kPairs has pair == 0
    __syncthreads();

    float wl_s0, wl_c0, lc0 = 1.0f, ls0 = 0.0f;
    sincospif(2.0f * (float)lane_id / (float)N, &wl_s0, &wl_c0);
    if constexpr (kCompact) sincospif((float)lane_id / 64.0f, &ls0, &lc0);   // W_128^lane: the odd registers' twiddle factor
    const int slot = blockIdx.x * kPairs + pair;
    if (slot >= nslots) return;

    // Cursor over this wave's frames: every kGroup-th frame of every run of the ring's work list, as ONE stream.
    struct Cursor {   // plain ints only: a bool member made the struct copies go through scratch (VMEM -> vmcnt waits)
        int wi, fi, ci, ticket_base, fb, fe, x0, valid;
    };
    const int wi_end = slot_off[slot + 1];
    auto settle = [&](Cursor& c) {   // move to the first run (from c.wi on) that has a frame for this wave
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
    };
    auto advance = [&](Cursor& c) {
        c.fi += kGroup;
        if (c.fi >= c.fe) {
            c.ticket_base += c.fe - c.fb;
            ++c.wi;
            settle(c);
        }
    };
    Cursor cur;
    cur.wi = slot_off[slot];
    cur.ticket_base = 0;
    settle(cur);
    if (!cur.valid) return;

#ifdef MPX_PROBE_ENDTIME
    const unsigned long long probe_t0 = wall_clock64();
    const unsigned long long probe_c0 = clock64();
    int probe_frames = 0;
#ifdef MPX_PROBE_PHASES
    unsigned* probe_ph = reinterpret_cast<unsigned*>(smem + kRing0 + kPairs * R + 16) + 8 * wave;   // LDS accumulators
    if (lane_id < 8) probe_ph[lane_id] = 0;
    unsigned probe_last = (unsigned)probe_c0;
#endif
#endif
    // Software pipeline over the wave's frames: the features of the wave's NEXT frame are loaded while this frame waits
    // for its ticket and overlap-adds; convert + merge of the next iteration then starts on data that is (mostly) there.
    PairFeat<P> ff;
    {
        const long long f = cur.fi;
        feat_load_paired_part<P, 0, P / 2, true>(ff, mag + f * ld, real + f * ld, imag + f * ld, lane_id);
    }
    while (cur.valid) {
        int lane = lane_id;  // laundered per frame (see k_analysis)
        float wl_s = wl_s0, wl_c = wl_c0, lc = lc0, ls = ls0;
        asm volatile("" : "+v"(lane), "+v"(wl_s), "+v"(wl_c), "+v"(lc), "+v"(ls));
#if MPX_PRIO_ROTATE
        // Fair shares of a SIMD.  The waves of a SIMD are arbitrated by priority, then AGE: at equal priority the oldest
        // wave issues whenever it can and the youngest gets what is left (MI355X_MICROARCH.md, "Two waves per SIMD").
        // Every wave here has the same static amount of work, so the old waves finished at 220 us, the young ones at
        // 370-400 us (tools/archive/endtime_probe.py: launch = 126-134 % of the waves' mean busy time) and the SIMDs idled through
        // the tail.  The priority therefore rotates with the constant-rate clock: the waves w, w + 4, w + 8 of a workgroup
        // share a SIMD (dispatch order 0 -> 2 -> 1 -> 3) and take the priorities (t + w / 4) mod kPerSimd in turn, t
        // advancing every 2^MPX_PRIO_SHIFT ticks of 10 ns.
        {
            constexpr int kPerSimd = (kPairWaves + 3) / 4;
            const unsigned t = (unsigned)(wall_clock64() >> MPX_PRIO_SHIFT);
            const int pr = (int)((t + (unsigned)(wave >> 2)) % (unsigned)kPerSimd);
            if (pr == 0) __builtin_amdgcn_s_setprio(0);
            else if (pr == 1) __builtin_amdgcn_s_setprio(1);
            else if (pr == 2) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(3);
        }
#endif
        Cursor nxt = cur;
        advance(nxt);
#ifdef MPX_PROBE_PHASES
        probe_last = (unsigned)clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MPX_PHASE(0);
#endif
        float xr[P], xi[P];
#ifdef MPX_ABL_NOMERGE   // energy ablation (tools/energy_probe.py): the loaded values go straight into the transform
#pragma unroll
        for (int j = 0; j < P / 2; ++j) {
            xr[j] = ff.m[j] + ff.a[j];
            xi[j] = ff.b[j];
            xr[j + P / 2] = ff.mq[j] + ff.aq[j];
            xi[j + P / 2] = ff.bq[j] + ff.mH;
        }
#else
        feat_merge_paired<P>(ff, xr, xi, lane, wl_c, wl_s);
#endif
#ifdef MPX_PROBE_PHASES
        MPX_PHASE(1);
#endif
#ifdef MPX_ABL_NOFFT
        if constexpr (false) {
#else
        if constexpr (kCompact) {
#endif
            // DIT form (fused multiply-add butterflies): its input wants register brev(j) <- bin lane + 64 j, its output is
            // register i <-> samples 2 n, 2 n + 1 with n = lane + 64 i: static renamings on both sides
            constexpr int LB = ilog2(P);
            float yr[P], yi[P];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                yr[brev(j, LB)] = xr[j];
                yi[brev(j, LB)] = xi[j];
            }
            wave_fft_dit_compact<P, +1>(yr, yi, tw, xbuf, lane, lc, ls);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                xr[j] = yr[j];
                xi[j] = yi[j];
            }
        } else {
#ifndef MPX_ABL_NOFFT
            wave_fft<P, +1>(xr, xi, tw, xbuf, lane);
#endif
        }
#ifdef MPX_FFT_FENCE
        // scheduling fence: without it the compiler sinks the transform's second pass (~500 VALU) below the ticket wait,
        // into the ordered section of the ring, where the pair's other wave waits for it
        {
#define MPX_F8(a, o) "+v"(a[o]), "+v"(a[o + 1]), "+v"(a[o + 2]), "+v"(a[o + 3]), "+v"(a[o + 4]), "+v"(a[o + 5]), "+v"(a[o + 6]), "+v"(a[o + 7])
            if constexpr (P == 32) {
                asm volatile("" : MPX_F8(xr, 0), MPX_F8(xr, 8), MPX_F8(xr, 16));
                asm volatile("" : MPX_F8(xr, 24), MPX_F8(xi, 0), MPX_F8(xi, 8));
                asm volatile("" : MPX_F8(xi, 16), MPX_F8(xi, 24));
            }
#undef MPX_F8
        }
#endif
#ifdef MPX_PROBE_PHASES
        MPX_PHASE(2);
#endif
        // (an exhausted cursor loads row 0 -- no branch around the loads; every finishing wave reads the same 24 KB, which
        // stay in L2: re-reading its own last frame cost 75 MB of HBM fetches per launch, 5 % of the kernel's traffic)
        const long long fnx = nxt.valid ? nxt.fi : 0;
        const float* nm = mag + fnx * ld;
        const float* nr = real + fnx * ld;
        const float* ni = imag + fnx * ld;
        asm volatile("" ::: "memory");
#ifdef MPX_ABL_NOLOAD   // energy ablation: no feature loads in the loop; the registers are (re)defined by an empty asm, as a load would
#define MPX_FEAT_LOAD(J0, J1, WH)                                                                           \
    do {                                                                                                    \
        _Pragma("unroll") for (int j_ = J0; j_ < J1; ++j_)                                                  \
            asm volatile("" : "=v"(ff.m[j_]), "=v"(ff.a[j_]), "=v"(ff.b[j_]), "=v"(ff.mq[j_]), "=v"(ff.aq[j_]), "=v"(ff.bq[j_])); \
    } while (0)
#else
#define MPX_FEAT_LOAD(J0, J1, WH) feat_load_paired_part<P, J0, J1, WH>(ff, nm, nr, ni, lane)
#endif
        MPX_FEAT_LOAD(0, JA, true);

        // ---- ordered section: wait for this frame's ticket
        const int fi = cur.fi;
        const int ticket = cur.ticket_base + (fi - cur.fb);
        // run geometry and frame positions first: their (scalar) loads must not sit behind the ticket
        const RunDesc rd = runs[cur.ci];
        float* strip = strips + rd.strip_off;
        float* pcm0 = pcm + rd.out_base;
        const int x = pm_rel[fi] - cur.x0;   // strip position of the frame's first sample
        const int target = x & ~63;
        const int flushed = (fi == cur.fb) ? 0 : ((pm_rel[fi - 1] - cur.x0) & ~63);
        asm volatile("" ::"s"(x), "s"(flushed), "s"(rd.head_end), "s"(rd.out_lo), "s"(rd.out_hi), "s"(rd.flush_end));
        MPX_PHASE(3);
        while (__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ticket)
            __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        MPX_PHASE(4);
#ifdef MPX_ABL_NOOLA   // energy ablation: no flush, no ring adds (the transform's outputs are kept alive)
#define MPX_OLA(stmt) do { } while (0)
#pragma unroll
        for (int i = 0; i < P; ++i) asm volatile("" ::"v"(xr[i]), "v"(xi[i]));
#else
#define MPX_OLA(stmt) stmt
#endif
        MPX_OLA(if (flushed < target) flush_ring<R>(ring, strip, pcm0, rd.head_end, rd.out_lo, rd.out_hi, flushed, target, lane));
        wave_sync();
        MPX_PHASE(5);
        const RingAddr ra = ring_addr<P>(ring_byte, x, lane);
        MPX_OLA((ring_add_plane<P, 0, kCompact ? MPX_CH : P, kCompact>(smem, ra, xr, lane, [](float o, float v, int) { return o + v; },
                                                [](int) { return true; })));
        if constexpr (JB > JA) {   // the first plane's registers are free: the second part of the prefetch
            asm volatile("" ::: "memory");
            MPX_FEAT_LOAD(JA, JB, false);
        }
        MPX_OLA((ring_add_plane<P, 1, kCompact ? MPX_CH : P, kCompact>(smem, ra, xi, lane, [](float o, float v, int) { return o + v; },
                                                [](int) { return true; })));
        wave_sync();
        if (fi == cur.fe - 1) {   // last frame of the run: stream out the rest, leave the ring cleared
            MPX_OLA(flush_ring<R>(ring, strip, pcm0, rd.head_end, rd.out_lo, rd.out_hi, target, rd.flush_end, lane));
            wave_sync();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        MPX_PHASE(6);
        __hip_atomic_store(turn, ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (JB < P / 2) {
            asm volatile("" ::: "memory");
            MPX_FEAT_LOAD(JB, P / 2, false);
        }
        cur = nxt;
        MPX_PHASE(7);
#ifdef MPX_PROBE_ENDTIME
        ++probe_frames;
#endif
    }
#ifdef MPX_PROBE_ENDTIME
    if (lane_id == 0) {
        const int w = (blockIdx.x * kPairWaves + wave) % 8192;
        g_endprobe[4 * w + 0] = probe_t0;
        g_endprobe[4 * w + 1] = wall_clock64();
        g_endprobe[4 * w + 2] = (unsigned long long)probe_frames;
        g_endprobe[4 * w + 3] = clock64() - probe_c0;
#ifdef MPX_PROBE_PHASES
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = 0; i < 8; ++i) g_phaseprobe[8 * w + i] = probe_ph[i];
#endif
    }
#endif
}

// pcm_out[out_base + e] += head strip[e], fix_lo <= e < fix_hi, for every run that has a predecessor in its utterance:
// the predecessor's tail sum (already in pcm_out) + this run's head sum, in that order.  One block per (run, 1024 elements).
__global__ __launch_bounds__(256) void k_ola_fixup(const RunDesc* __restrict__ runs, const float* __restrict__ strips,
                                                   float* __restrict__ pcm) {
    const RunDesc rd = runs[blockIdx.x];
    const int e0 = (rd.fix_lo & ~63) + (int)blockIdx.y * 1024;
    if (e0 >= rd.fix_hi) return;
    const float* strip = strips + rd.strip_off;
    float* out = pcm + rd.out_base;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = e0 + 256 * r + (int)threadIdx.x;
        if (e >= rd.fix_lo && e < rd.fix_hi) out[e] += strip[e];
    }
}

// ---------------------------------------------------------------------------------------------
// PSOLA gather (deterministic): magphase.py:34-62
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ola_gather(const float* __restrict__ frames, int N,
                                                    const int* __restrict__ utt_frame_off,
                                                    const int* __restrict__ pm_rel,
                                                    const int* __restrict__ out_start,
                                                    const long long* __restrict__ out_off,
                                                    float* __restrict__ pcm) {
    const int u = blockIdx.y;
    const long long o0 = out_off[u];
    const long long len = out_off[u + 1] - o0;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= len) return;
    const int f0 = utt_frame_off[u], f1 = utt_frame_off[u + 1];
    const long long b = t + out_start[u];  // index in the reference's un-trimmed OLA buffer
    // first frame i in [f0,f1) with pm_rel[i] > b - N   (pm_rel non-decreasing)
    int lo = f0, hi = f1;
    const long long thr = b - N;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)pm_rel[mid] > thr) hi = mid; else lo = mid + 1;
    }
    float acc = 0.0f;
    for (int i = lo; i < f1; ++i) {
        const long long off = b - pm_rel[i];
        if (off < 0) break;
        acc += frames[(long long)i * N + off];
    }
    pcm[o0 + t] = acc;
}

// First-pass twiddle table (layout: wave_fft.hpp): row = lane l, entry i = e^{2 pi i l brev(i) / M} as (cos, sin), rows
// padded to 2P + 4 floats (pad zeroed).  One block per lane, one thread per entry.
__global__ void k_tables_init(int P, float* __restrict__ tab) {
    const int l = blockIdx.x, i = threadIdx.x;
    const int M = 64 * P, stride = 2 * P + 4;
    int lb = 0;
    while ((1 << lb) < P) ++lb;
    if (i < P) {
        int k1 = 0;
        for (int b = 0; b < lb; ++b) k1 |= ((i >> b) & 1) << (lb - 1 - b);
        const int r = (int)(((long long)l * k1) % M);
        double sn, cs;
        sincospi(2.0 * (double)r / (double)M, &sn, &cs);
        tab[l * stride + 2 * i + 0] = (float)cs;
        tab[l * stride + 2 * i + 1] = (float)sn;
    } else if (i < P + 2) {
        tab[l * stride + 2 * i + 0] = 0.0f;
        tab[l * stride + 2 * i + 1] = 0.0f;
    }
}

}  // namespace mpx

using namespace mpx;

extern "C" {

int mpx_version(void) { return MPX_ABI_VERSION; }

const char* mpx_last_error(void) { return g_err; }

size_t mpx_tables_bytes(int fft_len) {
    const int P = p_of(fft_len);
    return P ? sizeof(float) * 64 * (size_t)(2 * P + 4) : 0;
}

int mpx_tables_init(void* stream, int fft_len, void* tables) {
    const int P = p_of(fft_len);
    if (!P) return fail(MPX_ERR_ARG, "mpx_tables_init: fft_len must be 1024, 2048 or 4096%s");
    if (!tables) return fail(MPX_ERR_ARG, "mpx_tables_init: null tables%s");
    // built on the device (float64 sincospi, rounded once): nothing is copied from the host, nothing synchronises
    hipLaunchKernelGGL(k_tables_init, dim3(64), dim3(64), 0, (hipStream_t)stream, P, (float*)tables);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int64_t mpx_feat_ld(int fft_len) {
    if (!p_of(fft_len)) return 0;
    return (int64_t)(fft_len / 2 + 1);   // dense: measured best (see the header)
}

int mpx_analysis_frames(void* stream, int fft_len, const void* tables, const float* sig, const int64_t* frame_pos,
                        const int32_t* frame_left, const int32_t* frame_right, int64_t n_frames, float* out_mag,
                        float* out_real, float* out_imag, int64_t ld) {
    const int P = p_of(fft_len);
    if (!P) return fail(MPX_ERR_ARG, "mpx_analysis_frames: fft_len must be 1024, 2048 or 4096%s");
    if (n_frames < 0) return fail(MPX_ERR_ARG, "mpx_analysis_frames: negative n_frames%s");
    if (ld < fft_len / 2 + 1) return fail(MPX_ERR_ARG, "mpx_analysis_frames: ld < fft_len/2 + 1%s");
    if (n_frames == 0) return MPX_OK;
    if (!tables || !sig || !frame_pos || !frame_left || !frame_right || !out_mag || !out_real || !out_imag)
        return fail(MPX_ERR_ARG, "mpx_analysis_frames: null pointer%s");
    const dim3 grid(grid_for(n_frames, kAnaWaves)), block(kAnaThreads);
    hipStream_t s = (hipStream_t)stream;
    if (P == 32) {
        if (int rc = set_lds(k_analysis<32>, lds_bytes_ana<32>())) return rc;
        hipLaunchKernelGGL(k_analysis<32>, grid, block, lds_bytes_ana<32>(), s, sig, (const long long*)frame_pos,
                           frame_left, frame_right, (long long)n_frames, (const float*)tables, out_mag, out_real,
                           out_imag, (long long)ld);
    } else if (P == 16) {
        if (int rc = set_lds(k_analysis<16>, lds_bytes_ana<16>())) return rc;
        hipLaunchKernelGGL(k_analysis<16>, grid, block, lds_bytes_ana<16>(), s, sig, (const long long*)frame_pos,
                           frame_left, frame_right, (long long)n_frames, (const float*)tables, out_mag, out_real,
                           out_imag, (long long)ld);
    } else {
        if (int rc = set_lds(k_analysis<8>, lds_bytes_ana<8>())) return rc;
        hipLaunchKernelGGL(k_analysis<8>, grid, block, lds_bytes_ana<8>(), s, sig, (const long long*)frame_pos,
                           frame_left, frame_right, (long long)n_frames, (const float*)tables, out_mag, out_real,
                           out_imag, (long long)ld);
    }
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_synthesis_lossless_frames(void* stream, int fft_len, const void* tables, const float* mag, const float* real,
                                  const float* imag, int64_t n_frames, float* frames_out, int64_t ld) {
    const int P = p_of(fft_len);
    if (!P) return fail(MPX_ERR_ARG, "mpx_synthesis_lossless_frames: fft_len must be 1024, 2048 or 4096%s");
    if (n_frames < 0) return fail(MPX_ERR_ARG, "mpx_synthesis_lossless_frames: negative n_frames%s");
    if (ld < fft_len / 2 + 1) return fail(MPX_ERR_ARG, "mpx_synthesis_lossless_frames: ld < fft_len/2 + 1%s");
    if (n_frames == 0) return MPX_OK;
    if (!tables || !mag || !real || !imag || !frames_out)
        return fail(MPX_ERR_ARG, "mpx_synthesis_lossless_frames: null pointer%s");
    const dim3 grid(grid_for(n_frames)), block(kThreads);
    hipStream_t s = (hipStream_t)stream;
    if (P == 32) {
        if (int rc = set_lds(k_synth_lossless<32>, lds_bytes<32>())) return rc;
        hipLaunchKernelGGL(k_synth_lossless<32>, grid, block, lds_bytes<32>(), s, mag, real, imag,
                           (long long)n_frames, (const float*)tables, frames_out, (long long)ld);
    } else if (P == 16) {
        if (int rc = set_lds(k_synth_lossless<16>, lds_bytes<16>())) return rc;
        hipLaunchKernelGGL(k_synth_lossless<16>, grid, block, lds_bytes<16>(), s, mag, real, imag,
                           (long long)n_frames, (const float*)tables, frames_out, (long long)ld);
    } else {
        if (int rc = set_lds(k_synth_lossless<8>, lds_bytes<8>())) return rc;
        hipLaunchKernelGGL(k_synth_lossless<8>, grid, block, lds_bytes<8>(), s, mag, real, imag,
                           (long long)n_frames, (const float*)tables, frames_out, (long long)ld);
    }
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_ola_gather(void* stream, int fft_len, const float* frames, int32_t n_utts, const int32_t* utt_frame_off,
                   const int32_t* pm_rel, const int32_t* out_start, const int64_t* out_off, int64_t max_out_len,
                   float* pcm_out) {
    if (!p_of(fft_len)) return fail(MPX_ERR_ARG, "mpx_ola_gather: fft_len must be 1024, 2048 or 4096%s");
    if (n_utts < 0 || max_out_len < 0) return fail(MPX_ERR_ARG, "mpx_ola_gather: negative size%s");
    if (n_utts == 0 || max_out_len == 0) return MPX_OK;
    if (!frames || !utt_frame_off || !pm_rel || !out_start || !out_off || !pcm_out)
        return fail(MPX_ERR_ARG, "mpx_ola_gather: null pointer%s");
    if (n_utts > 65535) return fail(MPX_ERR_ARG, "mpx_ola_gather: at most 65535 utterances per call%s");
    const dim3 block(256), grid((unsigned)((max_out_len + 255) / 256), (unsigned)n_utts);
    hipLaunchKernelGGL(k_ola_gather, grid, block, 0, (hipStream_t)stream, frames, fft_len, utt_frame_off, pm_rel,
                       out_start, (const long long*)out_off, pcm_out);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

/*
 * Host-side planner (no device work): the reference's serial constant -> variable frame-rate scan
 * (magphase.py:1426-1449, Q16), pos_{k-1} = pos_k - lerp(shift)(pos_k) from the last constant-rate centre backwards
 * until the position leaves the grid.  The reference evaluates scipy.interpolate.interp1d once per step from a Python
 * loop (8 us per step, ~1000 steps per utterance); this is the same float64 operation sequence -- searchsorted (left),
 * indices clipped to [1, n-1], slope = (y_hi - y_lo) / (x_hi - x_lo), y = slope * (x_new - x_lo) + y_lo, no fused
 * multiply-add -- so the shifts and locations are bit-identical (golden G7, tests/test_host_plans.py).
 * centres: float64[n] ascending (step * (1 .. n)); shift_c: float64[n]; shifts_out / locs_out: float64[2n].
 * Returns the index of the first valid element of the two outputs (the result is out[start : 2n]).
 */
int64_t mpx_host_const_to_var_scan(const double* centres, const double* shift_c, int64_t n, double* shifts_out,
                                   double* locs_out) {
#pragma clang fp contract(off)
    if (!centres || !shift_c || !shifts_out || !locs_out || n < 2) return -1;
    for (int64_t i = 0; i < 2 * n; ++i) shifts_out[i] = locs_out[i] = 0.0;
    double pos = centres[n - 1];
    for (int64_t i = 2 * n - 1; i >= 1; --i) {
        locs_out[i] = pos;
        if (!(pos >= centres[0] && pos <= centres[n - 1])) return i + 1;   // interp1d's bounds_error (NaN included)
        int64_t lo = 0, hi = n;                                            // np.searchsorted(centres, pos, 'left')
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (centres[mid] < pos) lo = mid + 1; else hi = mid;
        }
        int64_t idx = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
        const double x_lo = centres[idx - 1], x_hi = centres[idx], y_lo = shift_c[idx - 1], y_hi = shift_c[idx];
        const double slope = (y_hi - y_lo) / (x_hi - x_lo);
        const double prod = slope * (pos - x_lo);
        const double y = prod + y_lo;
        shifts_out[i] = y;
        pos = pos - y;
    }
    return 0;
}


#ifdef MPX_PROBE_ENDTIME
int mpx_probe_endtimes(unsigned long long* host, int n_words) {   // probe builds only (not part of the ABI)
    MPX_HIP_CHECK(hipDeviceSynchronize());
    MPX_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_endprobe), sizeof(unsigned long long) * (size_t)n_words));
    return MPX_OK;
}
#ifdef MPX_PROBE_PHASES
int mpx_probe_phases(unsigned long long* host, int n_words) {
    MPX_HIP_CHECK(hipDeviceSynchronize());
    MPX_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phaseprobe), sizeof(unsigned long long) * (size_t)n_words));
    return MPX_OK;
}
#endif
#endif

int mpx_synth_ola_slots(void) { return device_cus() * kPairs; }

// Relative speed of the slots' wave groups.  The waves i, i + 4, i + 8 of a workgroup share a SIMD, and a SIMD serves its
// waves by age: measured on MI355X (tools/archive/endtime_probe.py) the first four waves of a 12-wave workgroup take 14.4 us per
// frame, the next four 17.2, the last four 20.5 while all are resident (18.0 / 23.9 corrected for the time they run
// without their elders) -- with equal shares the launch lasted 126-134 % of the waves' mean busy time.  The planner
// therefore deals the frames in proportion to these weights; a rotating s_setprio (MPX_PRIO_ROTATE) narrows the spread
// only by a third.
#ifndef MPX_SYN_W0
#define MPX_SYN_W0 100
#endif
#ifndef MPX_SYN_W1
#define MPX_SYN_W1 80
#endif
#ifndef MPX_SYN_W2
#define MPX_SYN_W2 60
#endif
int mpx_synth_ola_slot_weights(float* weights_host, int32_t n_slots) {
    if (!weights_host || n_slots < 0) return fail(MPX_ERR_ARG, "mpx_synth_ola_slot_weights: bad arguments%s");
    const float w[4] = {(float)MPX_SYN_W0, (float)MPX_SYN_W1, (float)MPX_SYN_W2, (float)MPX_SYN_W2};
    for (int s = 0; s < n_slots; ++s) {
        const int age = ((s % kPairs) * kGroup) / 4;   // first wave of the group: its age rank on its SIMD
        weights_host[s] = w[age > 3 ? 3 : age];
    }
    return MPX_OK;
}

int64_t mpx_ola_strip_floats(int fft_len) { return p_of(fft_len) ? (int64_t)fft_len + 64 : 0; }

int mpx_synthesis_lossless_ola(void* stream, int fft_len, const void* tables, const float* mag, const float* real,
                               const float* imag, const mpx_ola_run* runs, int32_t n_runs, const int32_t* slot_off,
                               const int32_t* slot_runs, int32_t n_slots, const int32_t* pm_rel, float* strips,
                               float* pcm_out, int64_t ld) {
    const int P = p_of(fft_len);
    if (!P) return fail(MPX_ERR_ARG, "mpx_synthesis_lossless_ola: fft_len must be 1024, 2048 or 4096%s");
    if (n_runs < 0 || n_slots < 0) return fail(MPX_ERR_ARG, "mpx_synthesis_lossless_ola: negative count%s");
    if (ld < fft_len / 2 + 1) return fail(MPX_ERR_ARG, "mpx_synthesis_lossless_ola: ld < fft_len/2 + 1%s");
    if (n_runs == 0 || n_slots == 0) return MPX_OK;
    if (!tables || !mag || !real || !imag || !runs || !slot_off || !slot_runs || !pm_rel || !strips || !pcm_out)
        return fail(MPX_ERR_ARG, "mpx_synthesis_lossless_ola: null pointer%s");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((n_slots + kPairs - 1) / kPairs), block(kPairWaves * 64);
#define MPX_LAUNCH_PAIR(PP)                                                                                          \
    do {                                                                                                             \
        if (int rc = set_lds(k_synth_ola_pair<PP>, lds_bytes_pair<PP>())) return rc;                                 \
        hipLaunchKernelGGL(k_synth_ola_pair<PP>, grid, block, lds_bytes_pair<PP>(), s, mag, real, imag,              \
                           (const RunDesc*)runs, slot_off, slot_runs, (int)n_slots, pm_rel, (const float*)tables,    \
                           strips, pcm_out, (long long)ld);                                                          \
    } while (0)
    if (P == 32) MPX_LAUNCH_PAIR(32);
    else if (P == 16) MPX_LAUNCH_PAIR(16);
    else MPX_LAUNCH_PAIR(8);
#undef MPX_LAUNCH_PAIR
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_ola_fixup(void* stream, int fft_len, const mpx_ola_run* runs, int32_t n_runs, const float* strips,
                  float* pcm_out) {
    if (!p_of(fft_len)) return fail(MPX_ERR_ARG, "mpx_ola_fixup: fft_len must be 1024, 2048 or 4096%s");
    if (n_runs < 0) return fail(MPX_ERR_ARG, "mpx_ola_fixup: negative count%s");
    if (n_runs == 0) return MPX_OK;
    if (!runs || !strips || !pcm_out) return fail(MPX_ERR_ARG, "mpx_ola_fixup: null pointer%s");
    if (n_runs > 2147483647 / 2) return fail(MPX_ERR_ARG, "mpx_ola_fixup: too many runs%s");
    // a head strip holds at most fft_len + 64 elements: ceil((N + 64 + 63) / 1024) column blocks cover any fix range
    const dim3 block(256), grid((unsigned)n_runs, (unsigned)((fft_len + 127 + 1023) / 1024));
    hipLaunchKernelGGL(k_ola_fixup, grid, block, 0, (hipStream_t)stream, (const RunDesc*)runs, strips, pcm_out);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

}  // extern "C"
