// magphase_merlin.hip -- Merlin / HTS style post-filter of the log mel magnitudes (magphase.py:3375-3465, SURVEY.md 8f
// rank 3) as batched device work.  The reference pipes every utterance through nine SPTK-3.9 binaries (x2x, freqt, c2acr,
// vopr, mc2b, bcp, sopr, merge, b2mc); SPTK is not available here, the arithmetic is restated from its published
// algorithms -- PARITY UNPINNED, held by the known-answer tests of tests/test_post_filter_merlin.py and, for this device
// form, by a -m gpu comparison with the host form (magphase_amd.magphase.post_filter_merlin, float64 with a float32
// rounding at every pipe boundary).  All frames of a batch at once:
//   k_rows_gemm     mcep   = x . C1            la.rceps('log', 'compact') as the [D x D] matrix it is
//                   mcep_w = mcep * lifter     (1, 1, pf, pf, ...)                                   (vopr -m)
//   k_merlin_r0     r0, p_r0 = mean over the 4096 bins of exp(2 Re FFT(freqt(mcep | mcep_w, 2047, alpha -> 0)))   (freqt | c2acr)
//                   = sum_k w_k exp(2 (c . G)[k]) with G [D x 2049] = freqt matrix x cosine matrix, built on the host:
//                   a [64 frames x D] . [D x 2049] product whose exp'd outputs are reduced on the fly -- nothing of the
//                   [F x 2049] spectra is written
//   k_merlin_b      b = mc2b(mcep_w); b[0] += ln(r0 / p_r0) / 2; mcep_pf = b2mc(b)                    (mc2b | bcp | merge | b2mc)
//   k_rows_gemm     out = mcep_pf . Cf         cosine matrix, alpha = 0 (la.mcep_to_sp_cosmat 'log'); NaN -> la.MAGIC
// float32 throughout: the pipe boundaries of the SPTK chain are float32 as well.
#include "mpx_common.hpp"

namespace mpx {

// out[f][i] = (sum_k a[f][k] * mat[k][i]) * (scale ? scale[i] : 1), K, n <= 64.  Optionally out2 = the unscaled product.
// 256 threads = 4 frames x 64 outputs; the matrix lives in LDS.
__global__ __launch_bounds__(256) void k_rows_gemm(const float* __restrict__ a, long long F, int K, int n,
                                                   const float* __restrict__ mat, const float* __restrict__ scale,
                                                   float* __restrict__ out_plain, float* __restrict__ out_scaled,
                                                   float nan_value, int replace_nan) {
    __shared__ float ms[64 * 65];
    __shared__ float as[4][64];
    for (int i = threadIdx.x; i < K * n; i += 256) ms[(i / n) * 65 + (i % n)] = mat[i];
    const int i = threadIdx.x & 63, r = threadIdx.x >> 6;
    for (long long f0 = (long long)blockIdx.x * 4; f0 < F; f0 += (long long)gridDim.x * 4) {
        __syncthreads();
        const long long f = f0 + r;
        if (f < F && i < K) as[r][i] = a[f * K + i];
        __syncthreads();
        if (f < F && i < n) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc = fmaf(as[r][k], ms[k * 65 + i], acc);
            if (replace_nan && acc != acc) acc = nan_value;
            if (out_plain) out_plain[f * n + i] = acc;
            if (out_scaled) out_scaled[f * n + i] = acc * (scale ? scale[i] : 1.0f);
        }
    }
}

// r0[f] = sum_k w[k] exp(2 sum_n c[f][n] G[n][k]) for TWO coefficient sets (c1 -> r1, c2 -> r2) against the same G.
// One workgroup = 64 frames; thread t: frame t & 63, bins 16 (t >> 6) .. + 15 of every 64-bin chunk.  G chunk and both
// coefficient tiles in LDS; a wave's 64 lanes share their G values (broadcast reads), every lane its own frame's
// coefficients (row stride D + 1: conflict-free).  Deterministic: fixed summation order, no atomics.
__global__ __launch_bounds__(256) void k_merlin_r0(const float* __restrict__ c1, const float* __restrict__ c2,
                                                   long long F, int D, const float* __restrict__ G,
                                                   const float* __restrict__ wk, int nb, float* __restrict__ r1,
                                                   float* __restrict__ r2) {
    __shared__ float gs[64][64 + 4];    // [n][bin of the chunk]
    __shared__ float a1[64][65], a2[64][65];
    __shared__ float part[2][4][64];
    const int fl = threadIdx.x & 63, q = rfl((int)(threadIdx.x >> 6));
    const long long f0 = (long long)blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * D; i += 256) {
        const int r = i / D, n = i % D;
        const long long f = min(f0 + r, F - 1);
        a1[r][n] = c1[f * D + n];
        a2[r][n] = c2[f * D + n];
    }
    float s1 = 0.0f, s2 = 0.0f;
    for (int k0 = 0; k0 < nb; k0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < D * 64; i += 256) {
            const int n = i >> 6, kb = i & 63;
            gs[n][kb] = (k0 + kb < nb) ? G[(long long)n * nb + k0 + kb] : 0.0f;
        }
        __syncthreads();
        float acc1[16], acc2[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[e] = acc2[e] = 0.0f;
        for (int n = 0; n < D; ++n) {
            const float x1 = a1[fl][n], x2 = a2[fl][n];
            const float4* g4 = reinterpret_cast<const float4*>(&gs[n][16 * q]);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 g = g4[v];
                acc1[4 * v + 0] = fmaf(x1, g.x, acc1[4 * v + 0]);
                acc1[4 * v + 1] = fmaf(x1, g.y, acc1[4 * v + 1]);
                acc1[4 * v + 2] = fmaf(x1, g.z, acc1[4 * v + 2]);
                acc1[4 * v + 3] = fmaf(x1, g.w, acc1[4 * v + 3]);
                acc2[4 * v + 0] = fmaf(x2, g.x, acc2[4 * v + 0]);
                acc2[4 * v + 1] = fmaf(x2, g.y, acc2[4 * v + 1]);
                acc2[4 * v + 2] = fmaf(x2, g.z, acc2[4 * v + 2]);
                acc2[4 * v + 3] = fmaf(x2, g.w, acc2[4 * v + 3]);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = k0 + 16 * q + e;
            const float w = (k < nb) ? wk[k] : 0.0f;
            s1 = fmaf(w, expf(2.0f * acc1[e]), s1);
            s2 = fmaf(w, expf(2.0f * acc2[e]), s2);
        }
    }
    part[0][q][fl] = s1;
    part[1][q][fl] = s2;
    __syncthreads();
    if (threadIdx.x < 64 && f0 + threadIdx.x < F) {
        const int t = threadIdx.x;
        r1[f0 + t] = (part[0][0][t] + part[0][1][t]) + (part[0][2][t] + part[0][3][t]);
        r2[f0 + t] = (part[1][0][t] + part[1][1][t]) + (part[1][2][t] + part[1][3][t]);
    }
}

// Per frame: b = mc2b(mcep_w) (b[m] = mc[m] - alpha b[m + 1], from the top down); b[0] += ln(r0 / p_r0) / 2;
// mcep_pf = b2mc(b) (mc[m] = b[m] + alpha b[m + 1]).  One thread per frame, the D coefficients in registers' stead in a
// small per-thread loop over global memory (D floats per frame: 13 MB for 57 k frames).
__global__ __launch_bounds__(256) void k_merlin_b(const float* __restrict__ mcw, long long F, int D, float alpha,
                                                  const float* __restrict__ r0, const float* __restrict__ p_r0,
                                                  float* __restrict__ out) {
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const float* x = mcw + f * D;
    float* y = out + f * D;
    float b_next = x[D - 1];            // b[D-1] = mc[D-1]
    y[D - 1] = b_next;                  // mcep_pf[D-1] = b[D-1]
    for (int m = D - 2; m >= 0; --m) {
        float b = x[m] - alpha * b_next;                        // mc2b
        if (m == 0) b += 0.5f * logf(r0[f] / p_r0[f]);        // vopr -d | sopr -LN -d 2 | vopr -a, merged into b[0]
        y[m] = b + alpha * b_next;                              // b2mc (b[m+1] of the unmodified tail: only b[0] changes)
        b_next = b;
    }
}

}  // namespace mpx

using namespace mpx;

extern "C" int mpx_post_filter_merlin(void* stream, const float* mag_mel_log, int64_t n_frames, int32_t dim,
                                      const float* c1, const float* lifter, const float* g, const float* wk,
                                      int32_t n_bins, double alpha, const float* cf, double magic, float* mcep,
                                      float* mcep_w, float* r0, float* p_r0, float* out) {
    if (n_frames < 0 || dim < 3 || dim > 64 || n_bins < 1)
        return fail(MPX_ERR_ARG, "mpx_post_filter_merlin: need 3 <= dim <= 64, n_frames >= 0, n_bins >= 1%s");
    if (n_frames == 0) return MPX_OK;
    if (!mag_mel_log || !c1 || !lifter || !g || !wk || !cf || !mcep || !mcep_w || !r0 || !p_r0 || !out)
        return fail(MPX_ERR_ARG, "mpx_post_filter_merlin: null pointer%s");
    hipStream_t s = (hipStream_t)stream;
    const long long F = n_frames;
    const unsigned gb = (unsigned)std::min<long long>((F + 3) / 4, 4096);
    hipLaunchKernelGGL(k_rows_gemm, dim3(gb), dim3(256), 0, s, mag_mel_log, F, (int)dim, (int)dim, c1, lifter, mcep, mcep_w,
                       0.0f, 0);
    hipLaunchKernelGGL(k_merlin_r0, dim3((unsigned)((F + 63) / 64)), dim3(256), 0, s, (const float*)mcep,
                       (const float*)mcep_w, F, (int)dim, g, wk, (int)n_bins, r0, p_r0);
    // mcep is free now: it receives the post-filtered mel cepstrum
    hipLaunchKernelGGL(k_merlin_b, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, s, (const float*)mcep_w, F, (int)dim,
                       (float)alpha, (const float*)r0, (const float*)p_r0, mcep);
    hipLaunchKernelGGL(k_rows_gemm, dim3(gb), dim3(256), 0, s, (const float*)mcep, F, (int)dim, (int)dim, cf,
                       (const float*)nullptr, out, (float*)nullptr, (float)magic, 1);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}
