// magphase_hip.hip -- gfx950 kernels + C ABI (include/magphase_hip.h) for the MagPhase hot path.
//
// Kernel plan:
//   k_analysis<P>           one wavefront per frame, persistent waves (grid-stride over frames):
//                           gather+window+rotate -> 64*P-point complex FFT in registers/LDS -> real-FFT split
//                           -> mag/real/imag epilogue, 3 x H coalesced float stores.   HBM-write bound.
//   k_synth_lossless<P>     one wavefront per frame: 3 x H coalesced loads -> unit-phase spectrum ->
//                           Hermitian merge -> inverse FFT -> epoch-centred frame.      HBM-read bound.
//   k_ola_gather            one thread per output sample, ascending-frame gather (deterministic PSOLA).
// No MFMA in this file: nothing on the lossless path is a dense contraction; the mel
// warp / unwarp GEMMs of the compressed path (magphase_comp.hip) run on the fp32 MFMA.
#include "mpx_common.hpp"

namespace mpx {

template <int P>
__global__ __launch_bounds__(kAnaThreads) void k_analysis(const float* __restrict__ sig,
                                                          const long long* __restrict__ fpos,
                                                          const int* __restrict__ fleft,
                                                          const int* __restrict__ fright, long long nframes,
                                                          const float2* __restrict__ tw_g, float* __restrict__ omag,
                                                          float* __restrict__ oreal, float* __restrict__ oimag,
                                                          long long ld) {
    constexpr int M = 64 * P, N = 2 * M, LB = ilog2(P), kTile = 64 * P;
    extern __shared__ float smem[];
    float2* tw = reinterpret_cast<float2*>(smem);
    const int lane_id = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* xbuf = smem + P * 64 * 2 + wave * (P * kXStride);
    // byte address of xbuf in LDS (the dynamic segment starts at 0: the kernel has no static __shared__)
    const unsigned xbuf_byte = 4u * (unsigned)(P * 64 * 2 + rfl(wave) * (P * kXStride));
    for (int i = threadIdx.x; i < P * 64; i += kAnaThreads) tw[i] = tw_g[i];
    __syncthreads();

    // lane part of the split twiddle W_N^kappa = e^{-2 pi i kappa / N}
    float wl_s0, wl_c0;
    sincospif(-2.0f * (float)kappa<P>(lane_id) / (float)N, &wl_s0, &wl_c0);

    const int wave_u = rfl(wave);
    const long long fstep = (long long)gridDim.x * kAnaWaves;
    long long f = (long long)blockIdx.x * kAnaWaves + wave_u;
    if (f >= nframes) return;

    // Software pipeline: the samples of the wave's next frame are copied HBM -> LDS (into the transpose buffer, idle
    // after the FFT's exchange) while this frame's second FFT pass and epilogue run.  All 99 stores of the epilogue
    // are issued after that copy, so "copy landed" == vmcnt <= 63: no wait on the store drain.
    FrameGeom g = frame_geom(sig, fpos[f], fleft[f], fright[f], N);
    stage_samples_async(g, 0, kTile, xbuf_byte, lane_id);
    staged_wait<0>();

    for (; f < nframes; f += fstep) {
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
        const long long fn = f + fstep;
        FrameGeom gn = g;
        if (fn < nframes) {
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
                mlo[64 * q] = s2 * r;
                rlo[64 * q] = xr * r;
                ilo[64 * q] = xi * r;
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
                        row_m[M] = cm;
                        row_r[M] = cr;
                        row_i[M] = ci;
                    }
                } else {                              // block S_{q-1}
                    mhi[-64 * (q - 1)] = lane0 ? cm : hm;
                    rhi[-64 * (q - 1)] = lane0 ? cr : hr;
                    ihi[-64 * (q - 1)] = lane0 ? ci : hi_;
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
            mhi[-64 * (P / 2 - 1)] = lane0 ? s2 * r : hm;
            rhi[-64 * (P / 2 - 1)] = lane0 ? xr * r : hr;
            ihi[-64 * (P / 2 - 1)] = lane0 ? xi * r : hi_;
#endif
        }
        g = gn;
        if (fn < nframes) staged_wait<63>();   // >= 63 stores were issued after the copy: it has landed
    }
}

// ---------------------------------------------------------------------------------------------
// lossless synthesis: per-frame spectrum rebuild + inverse real FFT (epoch at N/2)
// ---------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(kThreads) void k_synth_lossless(const float* __restrict__ mag,
                                                             const float* __restrict__ real,
                                                             const float* __restrict__ imag, long long nframes,
                                                             const float2* __restrict__ tw_g,
                                                             float* __restrict__ frames, long long ld) {
    constexpr int M = 64 * P, N = 2 * M, H = M + 1, LB = ilog2(P);
    extern __shared__ float smem[];
    float2* tw = reinterpret_cast<float2*>(smem);
    const int lane_id = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* xbuf = smem + P * 64 * 2 + wave * (P * kXStride);
    for (int i = threadIdx.x; i < P * 64; i += kThreads) tw[i] = tw_g[i];
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
// fused lossless synthesis + PSOLA.  One wavefront owns one CHUNK = the consecutive frames of one
// utterance whose centres fall in a territory of T samples of the reference's OLA buffer; a PAIR of wavefronts owns a
// work list of chunks and alternates over their frames (k_synth_ola_pair below; the single-wave form it replaced is
// in the history: 5 waves per CU, 0.53 ms); each frame is rebuilt
// each frame (as k_synth_lossless) and overlap-adds it, in ascending frame order, into a ring buffer in
// LDS, streaming the finished part out to the chunk's private strip (T + N floats: the territory plus
// N/2 of halo on each side).  k_ola_fixup then sums the <= 3 strips covering each output sample.
// HBM traffic: features read once, (T+N)/T * 4 B per output sample written -- the [F x N] frame
// scratch of the two-kernel path (16 KB per frame written + read) is gone.
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// k_synth_ola_pair: same work as k_synth_ola, but TWO wavefronts share one chunk and one LDS ring: they take
// alternate frames, rebuild them concurrently, and enter the overlap-add strictly in frame order (a ticket word
// in LDS; LDS serves one CU's waves in order, so "ticket seen" implies the partner's ring writes are done).
// Why: the ring (16.5 KB) limits k_synth_ola to 5 waves per CU = 1.25 per SIMD, and a lone wave cannot hide its own
// LDS / memory latencies (measured 53 % issue-active).  Sharing the ring gives 8 waves per CU in the same LDS.
// Summation order is unchanged (ascending frames), so results are bit-identical to k_synth_ola.
// ---------------------------------------------------------------------------------------------
#ifndef MPX_SYN_PAIR_WAVES
#define MPX_SYN_PAIR_WAVES 8
#endif
#ifndef MPX_SYN_GROUP
#define MPX_SYN_GROUP 2
#endif
constexpr int kPairWaves = MPX_SYN_PAIR_WAVES;   // waves per workgroup
constexpr int kGroup = MPX_SYN_GROUP;            // waves sharing one ring (2: the "pair"; 4 at the same occupancy measured
                                                 // +11 %, 6 with 12 waves per CU +15 %: the in-order ring hand-over couples the
                                                 // waves of a group, so more waves only pay with their own rings -- no LDS left)
constexpr int kPairs = kPairWaves / kGroup;      // rings (= work-list slots) per workgroup
static_assert(kPairWaves % kGroup == 0, "waves per workgroup must be a multiple of the group size");
template <int P>
constexpr size_t lds_bytes_pair() {
    return sizeof(float) * (size_t)(P * 64 * 2 + kPairWaves * (P * kXStride) + kPairs * ring_len<P>() + 16);
}

template <int P>
__global__ __launch_bounds__(kPairWaves * 64) void k_synth_ola_pair(const float* __restrict__ mag,
                                                                    const float* __restrict__ real,
                                                                    const float* __restrict__ imag,
                                                                    const ChunkDesc* __restrict__ chunks,
                                                                    const int* __restrict__ slot_off,
                                                                    const int* __restrict__ slot_chunks, int nslots,
                                                                    const int* __restrict__ pm_rel, int T,
                                                                    const float2* __restrict__ tw_g,
                                                                    float* __restrict__ strips, long long ld) {
    constexpr int M = 64 * P, N = 2 * M, H = M + 1, LB = ilog2(P), R = ring_len<P>();
    extern __shared__ float smem[];
    float2* tw = reinterpret_cast<float2*>(smem);
    const int lane_id = threadIdx.x & 63;
    const int wave = rfl(threadIdx.x >> 6);
    const int pair = wave / kGroup, half = wave % kGroup;   // ring, member index
    float* xbuf = smem + P * 64 * 2 + wave * (P * kXStride);
    float* ring = smem + P * 64 * 2 + kPairWaves * (P * kXStride) + pair * R;
    int* turn = reinterpret_cast<int*>(smem + P * 64 * 2 + kPairWaves * (P * kXStride) + kPairs * R) + pair;
    for (int i = threadIdx.x; i < P * 64; i += kPairWaves * 64) tw[i] = tw_g[i];
    for (int i = threadIdx.x; i < kPairs * R; i += kPairWaves * 64)
        smem[P * 64 * 2 + kPairWaves * (P * kXStride) + i] = 0.0f;
    if (threadIdx.x < kPairs) turn[threadIdx.x - pair] = 0;   // thread t < kPairs has pair == 0
    __syncthreads();

    float wl_s0, wl_c0;
    sincospif(2.0f * (float)lane_id / (float)N, &wl_s0, &wl_c0);
    const int strip_len = T + N;
    const int slot = blockIdx.x * kPairs + pair;
    if (slot >= nslots) return;

    // Cursor over this wave's frames: every second frame of every chunk of the pair's work list, as ONE stream, so
    // that the feature prefetch runs across chunk boundaries (no per-chunk start-up bubble).
    struct Cursor {   // plain ints only: a bool member made the struct copies go through scratch (VMEM -> vmcnt waits)
        int wi, fi, ci, ticket_base, fb, fe, x0, valid;
    };
    const int wi_end = slot_off[slot + 1];
    auto settle = [&](Cursor& c) {   // move to the first chunk (from c.wi on) that has a frame for this wave
        while (c.wi < wi_end) {
            c.ci = slot_chunks[c.wi];
            const ChunkDesc cd = chunks[c.ci];
            c.fb = cd.frame_begin;
            c.fe = cd.frame_end;
            c.x0 = cd.x0;
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

    while (cur.valid) {
        int lane = lane_id;  // laundered per frame (see k_analysis)
        float wl_s = wl_s0, wl_c = wl_c0;
        asm volatile("" : "+v"(lane), "+v"(wl_s), "+v"(wl_c));
        Cursor nxt = cur;
        advance(nxt);

        // Features are loaded right where they are used: no register prefetch.  With two waves per SIMD the partner
        // wave covers the memory latency (a prefetch behind the FFT measured 4 % SLOWER), and the 99 registers are
        // worth more as room to keep many LDS operations in flight -- this kernel's stalls are LDS latency
        // (bpermutes of the merge, the exchange, the ring's read-add-write), not HBM.
        FrameFeat<P> ff;
        {
            const long long f = cur.fi;
            feat_load<P>(ff, mag + f * ld, real + f * ld, imag + f * ld, lane);
        }
        float xr[P], xi[P], xm;
        feat_convert<P>(ff, xr, xi, xm, lane);
        hermitian_merge<P>(xr, xi, xm, lane, wl_c, wl_s);
        wave_fft<P, +1>(xr, xi, tw, xbuf, lane);

        // ---- ordered section: wait for this frame's ticket
        const int fi = cur.fi;
        float* strip = strips + (long long)cur.ci * strip_len;
        const int ticket = cur.ticket_base + (fi - cur.fb);
        // frame positions first: their (scalar) loads must not sit behind the ticket inside the ordered section
        const int x = pm_rel[fi] - cur.x0;   // in [0, T)
        const int target = x & ~63;
        const int flushed = (fi == cur.fb) ? 0 : ((pm_rel[fi - 1] - cur.x0) & ~63);
        asm volatile("" ::"s"(x), "s"(flushed));
#ifndef MPX_PROBE_NOTICKET
        while (__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ticket)
            __builtin_amdgcn_s_sleep(1);
#endif
        asm volatile("" ::: "memory");
        if (flushed < target) flush_ring<R>(ring, strip, flushed, target, strip_len, lane);
        wave_sync();
        {
            constexpr int RH = R / 2;
            const int kap = kappa<P>(lane);
            const int odd = x & 1;
            float* r0 = ring + (odd ? RH : 0);
            float* r1 = ring + (odd ? 0 : RH);
            // element i of a plane sits at ring index (c + 64 b) mod RH, b = brev(i), c = cb + kappa: no index arrays --
            // the un-wrapped / wrapped base pointers pA / pB = pA - RH are selected per element (the wrap point differs
            // by at most one b between lanes) and 64 b goes into the instruction's immediate offset.
            const int c0 = ((x >> 1) % RH) + kap;
            const int c1 = (((x + 1) >> 1) % RH) + kap;
            float* pA0 = r0 + c0;
            float* pB0 = pA0 - RH;
            float* pA1 = r1 + c1;
            float* pB1 = pA1 - RH;
            const int w0 = (RH - c0 + 63) >> 6;   // first b with c0 + 64 b >= RH
            const int w1 = (RH - c1 + 63) >> 6;
            // read all ring values first, then add, then write: one LDS latency instead of 2P dependent chains
            constexpr int kRingBatch = P;
#pragma unroll
            for (int i0 = 0; i0 < P; i0 += kRingBatch) {
                float o0[kRingBatch], o1[kRingBatch];
#pragma unroll
                for (int i = 0; i < kRingBatch; ++i) {
                    const int bq = brev(i0 + i, LB);
                    o0[i] = ((bq >= w0) ? pB0 : pA0)[64 * bq];
                    o1[i] = ((bq >= w1) ? pB1 : pA1)[64 * bq];
                }
#pragma unroll
                for (int i = 0; i < kRingBatch; ++i) {
                    const int bq = brev(i0 + i, LB);
                    ((bq >= w0) ? pB0 : pA0)[64 * bq] = o0[i] + xr[i0 + i];
                    ((bq >= w1) ? pB1 : pA1)[64 * bq] = o1[i] + xi[i0 + i];
                }
            }
        }
        wave_sync();
        if (fi == cur.fe - 1) {   // last frame of the chunk: stream out the rest, leave the ring cleared
            flush_ring<R>(ring, strip, target, strip_len, strip_len, lane);
            wave_sync();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __hip_atomic_store(turn, ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        cur = nxt;
    }
}

// out[t] = sum over the strips of chunks c-1, c, c+1 (c = territory of b = t + out_start) -- fixed order.
__global__ __launch_bounds__(256) void k_ola_fixup(const float* __restrict__ strips, int N, int T,
                                                   const int* __restrict__ utt_chunk_off,
                                                   const int* __restrict__ strip_id,
                                                   const int* __restrict__ out_start,
                                                   const long long* __restrict__ out_off,
                                                   float* __restrict__ pcm) {
    // One block per (utterance, territory): the three strip ids are block-uniform, every load is independent.
    const int u = blockIdx.y;
    const int c = blockIdx.x;
    const int c0 = utt_chunk_off[u], nc = utt_chunk_off[u + 1] - c0;
    const long long o0 = out_off[u];
    const long long len = out_off[u + 1] - o0;
    const long long start = out_start[u];
    if ((long long)c * T >= start + len) return;   // no output sample in this territory
    const int strip_len = T + N;
    // territories at or beyond nc hold no frame: their output samples (the reference's zero tail) are written as 0
    const int sid_p = (c > 0 && c - 1 < nc) ? strip_id[c0 + c - 1] : -1;
    const int sid_o = (c < nc) ? strip_id[c0 + c] : -1;
    const int sid_n = (c + 1 < nc) ? strip_id[c0 + c + 1] : -1;
    const float* sp = strips + (long long)max(sid_p, 0) * strip_len;
    const float* so = strips + (long long)max(sid_o, 0) * strip_len;
    const float* sn = strips + (long long)max(sid_n, 0) * strip_len;
    // territory sample r = b - c*T in [0, T): strip indices  prev: r + T + N/2 (valid r < N/2),  own: r + N/2,
    // next: r - T + N/2 (valid r >= T - N/2); summed in the order prev, own, next (fixed -> deterministic)
    for (int r = threadIdx.x; r < T; r += 256) {
        const long long t = (long long)c * T + r - start;   // output index of buffer position b = c*T + r
        if (t < 0 || t >= len) continue;
        float acc = 0.0f;
        if (sid_p >= 0 && r < N / 2) acc += sp[r + T + N / 2];
        if (sid_o >= 0) acc += so[r + N / 2];
        if (sid_n >= 0 && r >= T - N / 2) acc += sn[r - T + N / 2];
        pcm[o0 + t] = acc;
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

}  // namespace mpx

using namespace mpx;

extern "C" {

int mpx_version(void) { return MPX_ABI_VERSION; }

const char* mpx_last_error(void) { return g_err; }

size_t mpx_tables_bytes(int fft_len) {
    const int P = p_of(fft_len);
    return P ? sizeof(float) * 2 * 64 * (size_t)P : 0;
}

int mpx_tables_init(void* stream, int fft_len, void* tables) {
    const int P = p_of(fft_len);
    if (!P) return fail(MPX_ERR_ARG, "mpx_tables_init: fft_len must be 1024, 2048 or 4096%s");
    if (!tables) return fail(MPX_ERR_ARG, "mpx_tables_init: null tables%s");
    const int M = 64 * P;
    std::vector<float> h(2 * 64 * (size_t)P);
    for (int k1 = 0; k1 < P; ++k1)
        for (int l = 0; l < 64; ++l) {
            const double a = 2.0 * M_PI * (double)((long long)l * k1 % M) / (double)M;
            h[2 * (k1 * 64 + l) + 0] = (float)std::cos(a);
            h[2 * (k1 * 64 + l) + 1] = (float)std::sin(a);
        }
    // pageable source: hipMemcpyAsync stages it before returning, so the local vector may die
    MPX_HIP_CHECK(hipMemcpyAsync(tables, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice,
                                 (hipStream_t)stream));
    MPX_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
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
                           frame_left, frame_right, (long long)n_frames, (const float2*)tables, out_mag, out_real,
                           out_imag, (long long)ld);
    } else if (P == 16) {
        if (int rc = set_lds(k_analysis<16>, lds_bytes_ana<16>())) return rc;
        hipLaunchKernelGGL(k_analysis<16>, grid, block, lds_bytes_ana<16>(), s, sig, (const long long*)frame_pos,
                           frame_left, frame_right, (long long)n_frames, (const float2*)tables, out_mag, out_real,
                           out_imag, (long long)ld);
    } else {
        if (int rc = set_lds(k_analysis<8>, lds_bytes_ana<8>())) return rc;
        hipLaunchKernelGGL(k_analysis<8>, grid, block, lds_bytes_ana<8>(), s, sig, (const long long*)frame_pos,
                           frame_left, frame_right, (long long)n_frames, (const float2*)tables, out_mag, out_real,
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
                           (long long)n_frames, (const float2*)tables, frames_out, (long long)ld);
    } else if (P == 16) {
        if (int rc = set_lds(k_synth_lossless<16>, lds_bytes<16>())) return rc;
        hipLaunchKernelGGL(k_synth_lossless<16>, grid, block, lds_bytes<16>(), s, mag, real, imag,
                           (long long)n_frames, (const float2*)tables, frames_out, (long long)ld);
    } else {
        if (int rc = set_lds(k_synth_lossless<8>, lds_bytes<8>())) return rc;
        hipLaunchKernelGGL(k_synth_lossless<8>, grid, block, lds_bytes<8>(), s, mag, real, imag,
                           (long long)n_frames, (const float2*)tables, frames_out, (long long)ld);
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

int mpx_synth_ola_slots(void) { return device_cus() * kPairs; }

int mpx_synthesis_lossless_ola(void* stream, int fft_len, const void* tables, const float* mag, const float* real,
                               const float* imag, const void* chunks, int32_t n_chunks, const int32_t* slot_off,
                               const int32_t* slot_chunks, int32_t n_slots, const int32_t* pm_rel,
                               int32_t territory, float* strips, int64_t ld) {
    const int P = p_of(fft_len);
    if (!P) return fail(MPX_ERR_ARG, "mpx_synthesis_lossless_ola: fft_len must be 1024, 2048 or 4096%s");
    if (n_chunks < 0 || n_slots < 0) return fail(MPX_ERR_ARG, "mpx_synthesis_lossless_ola: negative count%s");
    if (ld < fft_len / 2 + 1) return fail(MPX_ERR_ARG, "mpx_synthesis_lossless_ola: ld < fft_len/2 + 1%s");
    if (territory < fft_len / 2 || (territory % 64) != 0)
        return fail(MPX_ERR_ARG, "mpx_synthesis_lossless_ola: territory must be a multiple of 64 and >= fft_len/2%s");
    if (n_chunks == 0 || n_slots == 0) return MPX_OK;
    if (!tables || !mag || !real || !imag || !chunks || !slot_off || !slot_chunks || !pm_rel || !strips)
        return fail(MPX_ERR_ARG, "mpx_synthesis_lossless_ola: null pointer%s");
    hipStream_t s = (hipStream_t)stream;
    {
        const dim3 grid((n_slots + kPairs - 1) / kPairs), block(kPairWaves * 64);
        if (P == 32) {
            if (int rc = set_lds(k_synth_ola_pair<32>, lds_bytes_pair<32>())) return rc;
            hipLaunchKernelGGL(k_synth_ola_pair<32>, grid, block, lds_bytes_pair<32>(), s, mag, real, imag,
                               (const ChunkDesc*)chunks, slot_off, slot_chunks, (int)n_slots, pm_rel, (int)territory,
                               (const float2*)tables, strips, (long long)ld);
        } else if (P == 16) {
            if (int rc = set_lds(k_synth_ola_pair<16>, lds_bytes_pair<16>())) return rc;
            hipLaunchKernelGGL(k_synth_ola_pair<16>, grid, block, lds_bytes_pair<16>(), s, mag, real, imag,
                               (const ChunkDesc*)chunks, slot_off, slot_chunks, (int)n_slots, pm_rel, (int)territory,
                               (const float2*)tables, strips, (long long)ld);
        } else {
            if (int rc = set_lds(k_synth_ola_pair<8>, lds_bytes_pair<8>())) return rc;
            hipLaunchKernelGGL(k_synth_ola_pair<8>, grid, block, lds_bytes_pair<8>(), s, mag, real, imag,
                               (const ChunkDesc*)chunks, slot_off, slot_chunks, (int)n_slots, pm_rel, (int)territory,
                               (const float2*)tables, strips, (long long)ld);
        }
        MPX_HIP_CHECK(hipGetLastError());
        return MPX_OK;
    }
}

int mpx_ola_fixup(void* stream, int fft_len, int32_t territory, const float* strips, int32_t n_utts,
                  const int32_t* utt_chunk_off, const int32_t* strip_id, const int32_t* out_start,
                  const int64_t* out_off, int32_t max_territories, float* pcm_out) {
    if (!p_of(fft_len)) return fail(MPX_ERR_ARG, "mpx_ola_fixup: fft_len must be 1024, 2048 or 4096%s");
    if (n_utts < 0 || max_territories < 0) return fail(MPX_ERR_ARG, "mpx_ola_fixup: negative size%s");
    if (territory < fft_len / 2 || (territory % 64) != 0)
        return fail(MPX_ERR_ARG, "mpx_ola_fixup: territory must be a multiple of 64 and >= fft_len/2%s");
    if (n_utts == 0 || max_territories == 0) return MPX_OK;
    if (!strips || !utt_chunk_off || !strip_id || !out_start || !out_off || !pcm_out)
        return fail(MPX_ERR_ARG, "mpx_ola_fixup: null pointer%s");
    if (n_utts > 65535) return fail(MPX_ERR_ARG, "mpx_ola_fixup: at most 65535 utterances per call%s");
    const dim3 block(256), grid((unsigned)max_territories, (unsigned)n_utts);
    hipLaunchKernelGGL(k_ola_fixup, grid, block, 0, (hipStream_t)stream, strips, fft_len, (int)territory,
                       utt_chunk_off, strip_id, out_start, (const long long*)out_off, pcm_out);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

}  // extern "C"
