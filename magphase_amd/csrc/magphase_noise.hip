// magphase_noise.hip -- the aperiodic source of compressed-feature synthesis (magphase.py:883, np.random.uniform(-1, 1, ns_len)):
//   mpx_noise_numpy_mt19937   numpy's GLOBAL MT19937 stream continued on the device, bit for bit (jump-ahead ladder + one
//                             workgroup per segment), the reference's own sample values
//   mpx_noise_uniform         counter-based Philox source (opt-in: another stream, the same distribution)
#include "mpx_common.hpp"

namespace mpx {

// ---------------------------------------------------------------------------------------------
// Device noise source (opt-in replacement of the reference's np.random.uniform(-1, 1, n), magphase.py:883, which draws
// from numpy's global Mersenne twister on the host: 31 M draws per 128 utterances, the largest host cost of waveform
// generation).  Counter-based Philox4x32-10 (Salmon et al., SC'11): sample i of utterance u is word i & 3 of
// philox(counter = (i >> 2, 0, 0, 0) as 64 + 64 bits, key = seed_u), mapped to (u32 >> 8) * 2^-23 - 1 in [-1, 1).
// The value of a sample depends on (seed, i) only -- not on batching, sharding or launch geometry -- so a corpus
// generated on 1 or 8 GPUs, 1 or 64 utterances per launch, is bit-identical.  Integer arithmetic: the numpy restatement
// in tests/test_noise_rng.py must agree bit for bit.  NOT the reference's sample values (same distribution).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

__global__ __launch_bounds__(256) void k_noise_uniform(const unsigned long long* __restrict__ seeds,
                                                       const long long* __restrict__ off, float* __restrict__ out) {
    const int u = blockIdx.y;
    const long long n = off[u + 1] - off[u];
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;   // group of 4 samples
    if (4 * q >= n) return;
    const unsigned long long seed = seeds[u];
    unsigned c[4] = {(unsigned)q, (unsigned)((unsigned long long)q >> 32), 0u, 0u};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    float* o = out + off[u] + 4 * q;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (4 * q + e < n) o[e] = (float)(c[e] >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

// ---------------------------------------------------------------------------------------------
// numpy's global generator (MT19937) continued on the device: the reference draws the aperiodic source with
// np.random.uniform(-1, 1, ns_len) per utterance (magphase.py:883) -- 31 M doubles per 128 utterances, ~0.13 s of
// host time, four times everything else generation does.  The recurrence X[n+624] = X[n+397] ^ f(X[n], X[n+1]) yields
// 227 new words from the previous 624 in parallel, and a thread's word of the next 227 needs only its own new word
// plus old ones: ONE workgroup produces 454 words per barrier out of a 2048-word ring in LDS (measured ~12 ms per
// 7.7 M samples = 15 M words: 0.65 G samples/s against numpy's 0.24 G/s on the host, which no longer waits for it).
// k_mt19937_stream: key_in[624], pos = numpy's state; emits the tempered words pos .. pos + n_words - 1 of the stream
// and the state numpy would be left in.  k_mt_uniform: pairs of words -> random_sample's 53-bit double -> -1 + 2 d
// (numpy's legacy uniform: loc + scale * d) -> float32, the value the host path uploads.  Bit-identical by construction;
// tests/test_noise_rng.py compares with numpy on the GPU box.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned mt_twist(unsigned x, unsigned y) {
    const unsigned m = (x & 0x80000000u) | (y & 0x7fffffffu);
    return (m >> 1) ^ ((m & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// With ``windows`` the stream is cut into segments of J words (J a multiple of 624), one workgroup each: segment k
// starts from the 624-word window X[624 + k J ..] (k_mt_seq / k_mt_xor below), emits its own words and stops at the next
// segment's start; workgroup 0 also emits what is left of numpy's current block (X[pos .. 623], straight from the key)
// and the last one leaves the state.  Without windows: one segment, from the key.
__global__ __launch_bounds__(256) void k_mt19937_stream(const unsigned* __restrict__ key_in,
                                                        const unsigned* __restrict__ windows, long long J, int pos,
                                                        long long n_words, unsigned* __restrict__ raw,
                                                        unsigned* __restrict__ key_out, int* __restrict__ pos_out) {
    constexpr int R = 2048;   // ring: X[c + i] lives in ring[i & (R - 1)]
    __shared__ unsigned ring[R];
    const int t = threadIdx.x;
    const long long end = (long long)pos + n_words;                 // stream indices [pos, end) are emitted
    const long long B = (end > 0) ? (end - 1) / 624 : 0;            // block numpy's state ends in
    const long long gen_end = 624 * (B + 1);                        // that block is produced completely
    const long long c = windows ? 624 + (long long)blockIdx.x * J : 0;          // stream index of the segment's window
    const long long seg_end = windows ? min(c + J, gen_end) : gen_end;
    const unsigned* src = windows ? windows + 624ll * blockIdx.x : key_in;
    for (int i = t; i < 624; i += 256) ring[i] = src[i];
    __syncthreads();
    auto emit = [&](long long i, unsigned v) {
        if (i >= pos && i < end && i < seg_end) raw[i - pos] = mt_temper(v);
    };
    for (int i = t; i < 624; i += 256) emit(c + i, ring[i]);
    if (windows && blockIdx.x == 0)
        for (int i = t; i < 624; i += 256)
            if (i >= pos && i < end) raw[i - pos] = mt_temper(key_in[i]);
    // Thread t < 227 owns the chain X[c + 624 + 227 j + t], j = 0, 1, ...: each link is the previous one (a register) xor
    // the twist of two words 624 / 623 places back, which were written at least one barrier ago as long as only TWO links
    // are made per barrier (the third would read words of this very interval).  Per interval: 4 LDS reads, 2 writes, 2
    // stores.  (One wavefront running all 227 chains without barriers -- LDS serves a wave in order -- was measured
    // slower: 120 vs 71 ms per 128 utterances; the four waves overlap their LDS round trips.)
    unsigned v = (t < 227) ? ring[t + 397] : 0u;                     // X[c + 397 + t]: the "previous link" of the first step
    const long long n_int = (seg_end > c + 624) ? (seg_end - c - 624 + 453) / 454 : 0;
    unsigned k = 0;                                                   // ring offset of the interval's first input word
    long long o = c + 624 + t;                                        // stream index of this thread's next output
    for (long long it = 0; it < n_int; ++it) {
        if (t < 227) {
            const unsigned x = ring[(k + t) & (R - 1)], y = ring[(k + t + 1) & (R - 1)];
            const unsigned x2 = ring[(k + 227 + t) & (R - 1)], y2 = ring[(k + 228 + t) & (R - 1)];
            const unsigned v1 = v ^ mt_twist(x, y);                  // X[c + k + 624 + t]
            v = v1 ^ mt_twist(x2, y2);                               // X[c + k + 851 + t]
            ring[(k + 624 + t) & (R - 1)] = v1;
            ring[(k + 851 + t) & (R - 1)] = v;
            emit(o, v1);
            emit(o + 227, v);
        }
        k = (k + 454) & (R - 1);
        o += 454;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1) {
        for (int i = t; i < 624; i += 256) key_out[i] = ring[(624 * B - c + i) & (R - 1)];
        if (t == 0) pos_out[0] = (int)(end - 624 * B);
    }
}

// windows[0] = X[624 .. 1247], the first block the recurrence produces from numpy's key: every word of it (and after
// it) is a linear function of the generator's 19937 state bits, which the low 31 bits of X[0] are not.
__global__ __launch_bounds__(256) void k_mt_first(const unsigned* __restrict__ key_in, unsigned* __restrict__ windows) {
    __shared__ unsigned xs[624 + 908];
    const int t = threadIdx.x;
    for (int i = t; i < 624; i += 256) xs[i] = key_in[i];
    __syncthreads();
    for (int base = 0; base < 624; base += 454) {
        if (t < 227) {
            const unsigned v1 = xs[base + 397 + t] ^ mt_twist(xs[base + t], xs[base + t + 1]);
            xs[base + 624 + t] = v1;
            xs[base + 851 + t] = v1 ^ mt_twist(xs[base + 227 + t], xs[base + 228 + t]);
        }
        __syncthreads();
    }
    for (int i = t; i < 624; i += 256) windows[i] = xs[624 + i];
}

// Jump-ahead: dst[w] = the window J' words after windows[w], J' the jump whose polynomial g = x^J' mod phi is given as a
// 19968-bit mask (magphase_mtjump.cpp):  X[n + J'] = xor_{i : g_i} X[n + i].  Two launches per round of the doubling ladder
// (round 6; before: ONE kernel in which each of the 16 workgroups that share a jump's mask regenerated the 20 561 words behind
// the source window itself, 85 KB of LDS and 13 of its 20 us -- 4 080 such workgroups per draw, which took the CUs away from
// the synthesis kernels running beside the generator's stream):
//   k_mt_seq  one workgroup per SOURCE window: the recurrence for the 19937 + 623 words after it, through an 8 KB ring, written
//             to global memory once (seq, 84 KB per window);
//   k_mt_xor  kMtJumpSplit workgroups per jump: each stages the 1 248 + 768 words its share of the mask (39 mask words = bits
//             1248 p .. 1248 p + 1247) can reach, xors the words i + j of the set bits i for three j per thread and xors the
//             partial window into the (zeroed) result.
constexpr int kMtJumpSplit = 16;                  // 624 mask words = 16 x 39
constexpr int kMtSeqWords = 19937 + 624;          // words of a source window's sequence the mask can reach
constexpr int kMtSeqStride = 21376;               // + the read-ahead slack of the threads without a third j (zeros), 256-byte rows
constexpr int kMtXorSpan = 1248 + 768;            // words a share's workgroup stages
__global__ __launch_bounds__(256) void k_mt_seq(const unsigned* __restrict__ windows, unsigned* __restrict__ seq) {
    constexpr int R = 2048;   // ring: X[i] lives in ring[i & (R - 1)] (k_mt19937_stream's scheme)
    __shared__ unsigned ring[R];
    const int t = threadIdx.x;
    const unsigned* src = windows + 624ll * blockIdx.x;
    unsigned* out = seq + (long long)blockIdx.x * kMtSeqStride;
    for (int i = t; i < 624; i += 256) {
        const unsigned v = src[i];
        ring[i] = v;
        out[i] = v;
    }
    for (int i = kMtSeqWords + t; i < kMtSeqStride; i += 256) out[i] = 0u;
    __syncthreads();
    unsigned v = (t < 227) ? ring[t + 397] : 0u;
    unsigned k = 0;
    int o = 624 + t;
    for (int it = 0; it < (kMtSeqWords - 624 + 453) / 454; ++it) {
        if (t < 227) {
            const unsigned x = ring[(k + t) & (R - 1)], y = ring[(k + t + 1) & (R - 1)];
            const unsigned x2 = ring[(k + 227 + t) & (R - 1)], y2 = ring[(k + 228 + t) & (R - 1)];
            const unsigned v1 = v ^ mt_twist(x, y);
            v = v1 ^ mt_twist(x2, y2);
            ring[(k + 624 + t) & (R - 1)] = v1;
            ring[(k + 851 + t) & (R - 1)] = v;
            if (o < kMtSeqWords) out[o] = v1;
            if (o + 227 < kMtSeqWords) out[o + 227] = v;
        }
        k = (k + 454) & (R - 1);
        o += 454;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_mt_xor(const unsigned* __restrict__ seq, const unsigned* __restrict__ poly,
                                                unsigned* __restrict__ dst_windows) {
    __shared__ unsigned xs[kMtXorSpan];
    const int t = threadIdx.x, part = blockIdx.y;
    const unsigned* src = seq + (long long)blockIdx.x * kMtSeqStride + 1248 * part;
    for (int i = t; i < kMtXorSpan; i += 256) xs[i] = src[i];
    __syncthreads();
    unsigned a0 = 0u, a1 = 0u, a2 = 0u;
    constexpr int QW = 624 / kMtJumpSplit;
    for (int q = 0; q < QW; ++q) {
        unsigned bits = __builtin_amdgcn_readfirstlane(poly[part * QW + q]);
        while (bits) {   // wave-uniform: the mask is the same for every lane
            const int i = 32 * q + __builtin_ctz(bits);
            bits &= bits - 1u;
            a0 ^= xs[i + t];
            a1 ^= xs[i + t + 256];
            a2 ^= xs[i + t + 512];
        }
    }
    unsigned* dst = dst_windows + 624ll * blockIdx.x;
    atomicXor(&dst[t], a0);
    atomicXor(&dst[t + 256], a1);
    if (t + 512 < 624) atomicXor(&dst[t + 512], a2);
}

__global__ __launch_bounds__(256) void k_mt_uniform(const unsigned* __restrict__ raw, long long n, float* __restrict__ out) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const unsigned a = raw[2 * j] >> 5, b = raw[2 * j + 1] >> 6;
    const double d = __ddiv_rn(__dadd_rn(__dmul_rn((double)a, 67108864.0), (double)b), 9007199254740992.0);
    out[j] = (float)__dadd_rn(-1.0, __dmul_rn(2.0, d));
}

}  // namespace mpx

using namespace mpx;

extern "C" {

int mpx_noise_uniform(void* stream, int32_t n_utts, const uint64_t* seeds, const int64_t* offsets, int64_t max_len,
                      float* out) {
    if (n_utts < 0 || max_len < 0) return fail(MPX_ERR_ARG, "mpx_noise_uniform: negative size%s");
    if (n_utts == 0 || max_len == 0) return MPX_OK;
    if (!seeds || !offsets || !out) return fail(MPX_ERR_ARG, "mpx_noise_uniform: null pointer%s");
    if (n_utts > 65535) return fail(MPX_ERR_ARG, "mpx_noise_uniform: at most 65535 utterances per call%s");
    const dim3 grid((unsigned)((max_len + 1023) / 1024), (unsigned)n_utts);
    hipLaunchKernelGGL(k_noise_uniform, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)seeds,
                       (const long long*)offsets, out);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

// Segments of the parallel form: kMtSegWords words each (doubled until at most kMtMaxSegs are needed)
constexpr long long kMtSegWords = 624ll * 512;
constexpr int kMtMaxSegs = 256, kMtMaxLevels = 8;
// (A radix-16 ladder -- two rounds of fifteen jumps per source window instead of eight doubling rounds -- and narrower mask
// splits were measured in round 6 and are slower inside a generation job: docs/LAB_NOTES.md, round 6.)

// windows [kMtMaxSegs + kMtMaxLevels][624], then the sequences of a round's source windows [kMtMaxSegs / 2][kMtSeqStride]
int64_t mpx_noise_numpy_mt19937_work_words(void) {
    return (int64_t)(kMtMaxSegs + kMtMaxLevels) * 624 + (int64_t)(kMtMaxSegs / 2) * kMtSeqStride;
}

int mpx_noise_numpy_mt19937(void* stream, const uint32_t* key, int32_t pos, int64_t n_samples, uint32_t* raw,
                            float* out, uint32_t* key_out, int32_t* pos_out, uint32_t* work) {
    if (n_samples < 0 || pos < 0 || pos > 624) return fail(MPX_ERR_ARG, "mpx_noise_numpy_mt19937: bad size / position%s");
    if (!key || !key_out || !pos_out || (n_samples > 0 && (!raw || !out)))
        return fail(MPX_ERR_ARG, "mpx_noise_numpy_mt19937: null pointer%s");
    hipStream_t s = (hipStream_t)stream;
    const long long n_words = 2 * (long long)n_samples, end = pos + n_words;
    const long long gen_end = 624 * (((end > 0) ? (end - 1) / 624 : 0) + 1);
    long long J = kMtSegWords;
    int shift = 0;
    while ((gen_end - 624 + J - 1) / J > kMtMaxSegs) {
        J *= 2;
        ++shift;
    }
    const int K = (int)((gen_end - 624 + J - 1) / J);
    if (!work || K < 2) {   // short draws: one workgroup, from the key
        hipLaunchKernelGGL(k_mt19937_stream, dim3(1), dim3(256), 0, s, (const unsigned*)key, (const unsigned*)nullptr, 0ll,
                           (int)pos, n_words, (unsigned*)raw, (unsigned*)key_out, (int*)pos_out);
    } else {
        int levels = 0;   // rounds of the doubling ladder
        while ((1 << levels) < K) ++levels;
        // The ladder's jump polynomials (x^(J 2^l) mod the characteristic polynomial, l < levels) live on the device, one copy
        // per (device, shift, levels), uploaded once: round 5 found this call copying them from pageable memory and then
        // SYNCHRONISING the stream on every launch of a generation job -- the host could never run ahead of the device.
        unsigned* windows = (unsigned*)work;
        unsigned* seq = windows + (size_t)(kMtMaxSegs + kMtMaxLevels) * 624;
        const unsigned* dpoly = nullptr;
        {
            struct Entry { int dev, shift, levels; unsigned* ptr; };
            static std::mutex mu;
            static std::vector<Entry> cache;
            int dev = 0;
            (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> lock(mu);
            for (const Entry& e : cache)
                if (e.dev == dev && e.shift == shift && e.levels >= levels) dpoly = e.ptr;
            if (!dpoly) {
                std::vector<int64_t> jumps((size_t)levels);
                for (int l = 0; l < levels; ++l) jumps[(size_t)l] = J << l;
                std::vector<uint32_t> polys((size_t)levels * 624);
                if (mpx_host_mt19937_jump_polys(jumps.data(), levels, polys.data(), 16) != MPX_OK)
                    return fail(MPX_ERR_ARG, "mpx_noise_numpy_mt19937: jump polynomials unavailable%s");
                unsigned* p = nullptr;
                MPX_HIP_CHECK(hipMalloc((void**)&p, polys.size() * sizeof(uint32_t)));
                MPX_HIP_CHECK(hipMemcpy(p, polys.data(), polys.size() * sizeof(uint32_t), hipMemcpyHostToDevice));   // synchronous, once
                cache.push_back(Entry{dev, shift, levels, p});
                dpoly = p;
            }
        }
        MPX_HIP_CHECK(hipMemsetAsync(windows, 0, (size_t)K * 624 * sizeof(unsigned), s));   // jump results are xor-ed in
        hipLaunchKernelGGL(k_mt_first, dim3(1), dim3(256), 0, s, (const unsigned*)key, windows);
        for (int l = 0; l < levels; ++l) {
            const int n_src = 1 << l;
            const int n_jump = min(n_src, K - n_src);   // window w -> window n_src + w
            hipLaunchKernelGGL(k_mt_seq, dim3((unsigned)n_jump), dim3(256), 0, s, (const unsigned*)windows, seq);
            hipLaunchKernelGGL(k_mt_xor, dim3((unsigned)n_jump, kMtJumpSplit), dim3(256), 0, s, (const unsigned*)seq,
                               (const unsigned*)(dpoly + 624ll * l), windows + 624ll * n_src);
        }
        hipLaunchKernelGGL(k_mt19937_stream, dim3((unsigned)K), dim3(256), 0, s, (const unsigned*)key,
                           (const unsigned*)windows, J, (int)pos, n_words, (unsigned*)raw, (unsigned*)key_out,
                           (int*)pos_out);
    }
    if (n_samples > 0)
        hipLaunchKernelGGL(k_mt_uniform, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0, s, (const unsigned*)raw,
                           (long long)n_samples, out);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

}  // extern "C"
