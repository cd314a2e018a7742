// magphase_epochs.hip -- batched kernels of the built-in epoch / voicing front end.
//
// The reference shells out to REAPER (libaudio.py:450-455), an external binary; this is NOT REAPER and makes no parity
// claim (magphase_amd/epochs.py: parity unpinned, quality-checked on signals with known epochs).  Two stages, both
// batched over the utterances of a call:
//   1. F0 / voicing candidates: normalised cross-correlation (NCCF) of the signal decimated to ~4 kHz, 40 ms frames every
//      5 ms, lags for 60 .. 400 Hz.  k_epoch_decimate (box average, stride dec) + k_epoch_nccf (one wavefront per frame,
//      one lane per lag, float64 sums; the wave picks the shortest lag within 0.06 of the best and refines it by a
//      parabola through its neighbours).
//   2. Epochs by zero-frequency filtering (Murty & Yegnanarayana 2008): the differenced signal through two
//      zero-frequency resonators (four cumulative sums) with the local mean over ~1.5 pitch periods removed after each
//      resonator and twice more at the end.  Every cumulative sum and every moving mean is a float64 prefix sum over
//      the utterance: k_epoch_scan (one workgroup per utterance, tiles of 4096 elements, carry in a register) and
//      k_epoch_movmean (windowed sum from the prefix sums, replicate padding).  k_epoch_crossings lists the zero
//      crossings of both directions with their slope and an excitation-energy score (the host keeps the direction the
//      energy sits on = the recording's polarity).
// Nothing here is on the analysis / synthesis hot path: it runs once per utterance, ahead of mpx_analysis_frames.
#include "mpx_common.hpp"

namespace mpx {

// ---------------------------------------------------------------------------------------------
// stage 1
// ---------------------------------------------------------------------------------------------
// xd[j] = mean of x[j*dec - dec/2 .. j*dec - dec/2 + 2 dec) with zeros outside the utterance (the divisor stays 2 dec),
// minus nothing (the utterance mean is removed by the caller's choice of `mean`: x - mean inside the utterance).
__global__ __launch_bounds__(256) void k_epoch_decimate(const float* __restrict__ sig, const long long* __restrict__ off,
                                                        const double* __restrict__ mean, int dec,
                                                        const long long* __restrict__ doff, double* __restrict__ xd) {
    const int u = blockIdx.y;
    const long long n = off[u + 1] - off[u], nd = doff[u + 1] - doff[u];
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= nd) return;
    const float* x = sig + off[u];
    const double m = mean[u];
    double acc = 0.0;
    const long long s0 = j * dec - dec / 2;
    for (int k = 0; k < 2 * dec; ++k) {
        const long long i = s0 + k;
        if (i >= 0 && i < n) acc += (double)x[i] - m;
    }
    xd[doff[u] + j] = acc / (double)(2 * dec);
}

// One wavefront per (utterance, frame): lane l handles lag l_min + l.  Outputs per frame: f0 candidate (fs_d / refined lag),
// the NCCF value at the chosen lag, and the frame energy (the host turns them into the voicing decision).
__global__ __launch_bounds__(256) void k_epoch_nccf(const double* __restrict__ xd, const long long* __restrict__ doff,
                                                    const double* __restrict__ dmean, const long long* __restrict__ foff,
                                                    int hop, int win, int l_min, int n_lags, double fs_d,
                                                    float* __restrict__ f0, float* __restrict__ peak,
                                                    float* __restrict__ energy) {
    const int u = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long T = foff[u + 1] - foff[u];
    if (t >= T) return;
    const long long nd = doff[u + 1] - doff[u];
    const double* x = xd + doff[u];
    const double m = dmean[u];
    const long long s0 = t * hop;
    const int lag = l_min + lane;
    double num = 0.0, e_sh = 0.0, e_ref = 0.0;
    for (int k = 0; k < win; ++k) {
        const long long i0 = s0 + k, i1 = s0 + k + lag;
        const double a = (i0 < nd) ? x[i0] - m : 0.0;          // frames past the end see zeros (torch pads)
        const double b = (lane < n_lags && i1 < nd) ? x[i1] - m : 0.0;
        e_ref = fma(a, a, e_ref);
        num = fma(a, b, num);
        e_sh = fma(b, b, e_sh);
    }
    double r = (lane < n_lags) ? num / (sqrt(e_ref * e_sh) + 1.0e-20) : -2.0;
    // best over lanes
    double best = r;
    for (int o = 32; o >= 1; o >>= 1) {
        const double v = __shfl_xor(best, o);
        best = fmax(best, v);
    }
    const unsigned long long okm = __ballot(lane < n_lags && r >= best - 0.06);
    const int first = okm ? __builtin_ctzll(okm) : 0;
    const int li = min(max(first, 1), n_lags - 2);
    const double y0 = __shfl(r, li - 1), y1 = __shfl(r, li), y2 = __shfl(r, li + 1);
    double delta = 0.5 * (y0 - y2) / (y0 - 2.0 * y1 + y2 - 1.0e-20);
    delta = fmin(fmax(delta, -1.0), 1.0);
    const double lag_r = (double)(li + l_min) + delta;
    const double pk = __shfl(r, first);
    if (lane == 0) {
        f0[foff[u] + t] = (float)(fs_d / lag_r);
        peak[foff[u] + t] = (float)pk;
        energy[foff[u] + t] = (float)e_ref;
    }
}

// ---------------------------------------------------------------------------------------------
// stage 2: float64 prefix sums per utterance
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    unsigned lo = (unsigned)__builtin_bit_cast(unsigned long long, v), hi = (unsigned)(__builtin_bit_cast(unsigned long long, v) >> 32);
    lo = (unsigned)__shfl_up((int)lo, d);
    hi = (unsigned)__shfl_up((int)hi, d);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// mode 0: out = inclusive cumsum(in).  mode 1: in is float32 PCM x, out = cumsum(dx), dx[0] = 0, dx[i] = x[i] - x[i-1]
// (i.e. x[i] - x[0]: written as a scan so that it shares the code path and the rounding of the torch form).
// mode 2: in is float32 PCM, out = cumsum(dx^2).  One workgroup of 1024 threads per utterance.
template <int MODE>
__global__ __launch_bounds__(1024) void k_epoch_scan(const void* in_, const long long* __restrict__ off, double* out) {   // in_ may alias out
    __shared__ double s_wave[16];
    __shared__ double s_carry;
    const int u = blockIdx.x;
    const long long n = off[u + 1] - off[u];
    const double* ind = (const double*)in_ + off[u];
    const float* inf = (const float*)in_ + off[u];
    double* o = out + off[u];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0.0;
    __syncthreads();
    for (long long base = 0; base < n; base += 4096) {
        const long long i0 = base + 4 * (long long)threadIdx.x;
        double v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long i = i0 + e;
            double x = 0.0;
            if (i < n) {
                if (MODE == 0) x = ind[i];
                else {
                    const double d = (i > 0) ? (double)inf[i] - (double)inf[i - 1] : 0.0;
                    x = (MODE == 1) ? d : d * d;
                }
            }
            v[e] = x;
        }
        v[1] += v[0];
        v[2] += v[1];
        v[3] += v[2];
        double incl = v[3];                                    // inclusive scan of the threads' totals within the wave
        for (int d = 1; d < 64; d <<= 1) {
            const double t = shfl_up_f64(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        double wave_off = 0.0;
        for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
        const double carry = s_carry;
        const double pre = carry + wave_off + (incl - v[3]);   // sum of everything before this thread's 4 elements
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (i0 + e < n) o[i0 + e] = pre + v[e];
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = pre + v[3];
        __syncthreads();
    }
}

// out[i] = y[i] - (sum of y over [i - h, i + h], indices clamped to the utterance: replicate padding) / (2h + 1), from
// the inclusive prefix sums S of y.  Out of place (the clamped terms read y[0] and y[n-1] from every thread).
__global__ __launch_bounds__(256) void k_epoch_movmean(const double* __restrict__ y, const double* __restrict__ S,
                                                       const long long* __restrict__ off, const int* __restrict__ half,
                                                       double* __restrict__ out) {
    const int u = blockIdx.y;
    const long long n = off[u + 1] - off[u];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long h = half[u];
    const double* yu = y + off[u];
    const double* Su = S + off[u];
    const long long lo = i - h, hi = i + h;
    const long long lo_c = lo < 0 ? 0 : lo, hi_c = hi > n - 1 ? n - 1 : hi;
    double w = Su[hi_c] - (lo_c > 0 ? Su[lo_c - 1] : 0.0);
    if (lo < 0) w += (double)(-lo) * yu[0];
    if (hi > n - 1) w += (double)(hi - (n - 1)) * yu[n - 1];
    out[off[u] + i] = yu[i] - w / (double)(2 * h + 1);
}

// mean of one utterance's samples (float32 PCM or float64), one workgroup per utterance
template <typename T>
__global__ __launch_bounds__(256) void k_epoch_mean(const T* __restrict__ x, const long long* __restrict__ off,
                                                    double* __restrict__ mean) {
    __shared__ double s_acc[256];
    const int u = blockIdx.x;
    double a = 0.0;
    for (long long i = off[u] + threadIdx.x; i < off[u + 1]; i += 256) a += (double)x[i];
    s_acc[threadIdx.x] = a;
    __syncthreads();
    for (int k = 128; k >= 1; k >>= 1) {
        if (threadIdx.x < k) s_acc[threadIdx.x] += s_acc[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean[u] = (off[u + 1] > off[u]) ? s_acc[0] / (double)(off[u + 1] - off[u]) : 0.0;
}

// Zero crossings of y in both directions.  For direction p (0: negative-going = closure of a positive-polarity
// recording, 1: positive-going): index i (1 <= i < n) with sign change between i-1 and i; slope |y[i] - y[i-1]|; score =
// excitation energy in the millisecond after minus the millisecond before (c2 = inclusive cumsum of dx^2).
// Appended (unordered) to the utterance's list of that direction; cnt[2u + p] counts them, cap entries per list.
__global__ __launch_bounds__(256) void k_epoch_crossings(const double* __restrict__ y, const double* __restrict__ c2,
                                                         const long long* __restrict__ off, int w_score, int cap,
                                                         int* __restrict__ cnt, int* __restrict__ idx,
                                                         float* __restrict__ slope, float* __restrict__ score,
                                                         float* __restrict__ frac) {
    const int u = blockIdx.y;
    const long long n = off[u + 1] - off[u];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x + 1;
    if (i >= n) return;
    const double* yu = y + off[u];
    const double a = yu[i - 1], b = yu[i];
    int p = -1;
    if (a > 0.0 && b <= 0.0) p = 0;          // -y: up-crossing (z[i-1] < 0, z[i] >= 0 with z = -y)
    else if (a < 0.0 && b >= 0.0) p = 1;
    if (p < 0) return;
    const int k = atomicAdd(&cnt[2 * u + p], 1);
    if (k >= cap) return;
    const double* c = c2 + off[u];
    long long q = i;
    q = q < w_score ? w_score : (q > n - w_score - 1 ? n - w_score - 1 : q);
    // torch form: c = [0, cumsum(dx^2)]; (c[q + w] - c[q]) - (c[q] - c[q - w])  ->  inclusive sums shifted by one
    const double cq = c[q - 1], cp = c[q + w_score - 1], cm = (q - w_score - 1 >= 0) ? c[q - w_score - 1] : 0.0;
    const long long slot = (long long)(2 * u + p) * cap + k;
    idx[slot] = (int)i;
    // where between samples i - 1 and i the line through the two values crosses zero, as a fraction of the step back from i
    // (0: at sample i, towards 1: at sample i - 1): sub-sample epochs -- at 16 kHz a sample is 62 us, the tracker's jitter 40
    if (frac) frac[slot] = (float)(b / (b - a));
    slope[slot] = (float)fabs(b - a);
    score[slot] = (float)((cp - cq) - (cq - cm));
}

}  // namespace mpx

using namespace mpx;

extern "C" {

int mpx_epoch_f0_track(void* stream, const float* sig, const int64_t* off, int32_t n_utts, int32_t dec,
                       const int64_t* dec_off, int64_t max_dec_len, double* xd, double* means, const int64_t* frame_off,
                       int64_t max_frames, int32_t hop, int32_t win, int32_t l_min, int32_t n_lags, double fs_d,
                       float* f0, float* peak, float* energy) {
    if (n_utts < 0) return fail(MPX_ERR_ARG, "mpx_epoch_f0_track: negative count%s");
    if (n_utts == 0 || max_dec_len <= 0 || max_frames <= 0) return MPX_OK;
    if (n_utts > 65535) return fail(MPX_ERR_ARG, "mpx_epoch_f0_track: at most 65535 utterances per call%s");
    if (dec < 1 || n_lags < 3 || n_lags > 64 || win < 1 || hop < 1)
        return fail(MPX_ERR_ARG, "mpx_epoch_f0_track: bad geometry (1 <= dec, 3 <= n_lags <= 64)%s");
    if (!sig || !off || !dec_off || !xd || !means || !frame_off || !f0 || !peak || !energy)
        return fail(MPX_ERR_ARG, "mpx_epoch_f0_track: null pointer%s");
    hipStream_t s = (hipStream_t)stream;
    double* mean_x = means;            // [n_utts]
    double* mean_d = means + n_utts;   // [n_utts]
    hipLaunchKernelGGL(k_epoch_mean<float>, dim3((unsigned)n_utts), dim3(256), 0, s, sig, (const long long*)off, mean_x);
    hipLaunchKernelGGL(k_epoch_decimate, dim3((unsigned)((max_dec_len + 255) / 256), (unsigned)n_utts), dim3(256), 0, s, sig,
                       (const long long*)off, mean_x, (int)dec, (const long long*)dec_off, xd);
    hipLaunchKernelGGL(k_epoch_mean<double>, dim3((unsigned)n_utts), dim3(256), 0, s, xd, (const long long*)dec_off, mean_d);
    hipLaunchKernelGGL(k_epoch_nccf, dim3((unsigned)((max_frames + 3) / 4), (unsigned)n_utts), dim3(256), 0, s, xd,
                       (const long long*)dec_off, mean_d, (const long long*)frame_off, (int)hop, (int)win, (int)l_min,
                       (int)n_lags, fs_d, f0, peak, energy);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

int mpx_epoch_zff(void* stream, const float* sig, const int64_t* off, int32_t n_utts, int64_t max_len,
                  const int32_t* half_win, int32_t w_score, double* buf_a, double* buf_b, double* buf_c, int32_t cap,
                  int32_t* counts, int32_t* cross_idx, float* cross_slope, float* cross_score, float* cross_frac) {
    if (n_utts < 0 || max_len < 0 || cap < 1) return fail(MPX_ERR_ARG, "mpx_epoch_zff: bad size%s");
    if (n_utts == 0 || max_len == 0) return MPX_OK;
    if (n_utts > 65535) return fail(MPX_ERR_ARG, "mpx_epoch_zff: at most 65535 utterances per call%s");
    if (!sig || !off || !half_win || !buf_a || !buf_b || !buf_c || !counts || !cross_idx || !cross_slope || !cross_score)
        return fail(MPX_ERR_ARG, "mpx_epoch_zff: null pointer%s");
    hipStream_t s = (hipStream_t)stream;
    const dim3 gs((unsigned)n_utts), bs(1024);
    const dim3 ge((unsigned)((max_len + 255) / 256), (unsigned)n_utts), be(256);
    const long long* o = (const long long*)off;
    double *A = buf_a, *B = buf_b, *C = buf_c;
    // first resonator: cumsum(cumsum(dx)), local mean removed
    hipLaunchKernelGGL(k_epoch_scan<1>, gs, bs, 0, s, (const void*)sig, o, A);
    hipLaunchKernelGGL(k_epoch_scan<0>, gs, bs, 0, s, (const void*)A, o, A);
    hipLaunchKernelGGL(k_epoch_scan<0>, gs, bs, 0, s, (const void*)A, o, B);
    hipLaunchKernelGGL(k_epoch_movmean, ge, be, 0, s, A, B, o, half_win, C);
    // second resonator, local mean removed three times (C -> A -> C -> A)
    hipLaunchKernelGGL(k_epoch_scan<0>, gs, bs, 0, s, (const void*)C, o, C);
    hipLaunchKernelGGL(k_epoch_scan<0>, gs, bs, 0, s, (const void*)C, o, C);
    hipLaunchKernelGGL(k_epoch_scan<0>, gs, bs, 0, s, (const void*)C, o, B);
    hipLaunchKernelGGL(k_epoch_movmean, ge, be, 0, s, C, B, o, half_win, A);
    hipLaunchKernelGGL(k_epoch_scan<0>, gs, bs, 0, s, (const void*)A, o, B);
    hipLaunchKernelGGL(k_epoch_movmean, ge, be, 0, s, A, B, o, half_win, C);
    hipLaunchKernelGGL(k_epoch_scan<0>, gs, bs, 0, s, (const void*)C, o, B);
    hipLaunchKernelGGL(k_epoch_movmean, ge, be, 0, s, C, B, o, half_win, A);
    // excitation energy prefix sums (B), crossings of the filtered signal (A)
    hipLaunchKernelGGL(k_epoch_scan<2>, gs, bs, 0, s, (const void*)sig, o, B);
    MPX_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int32_t) * 2 * (size_t)n_utts, s));
    hipLaunchKernelGGL(k_epoch_crossings, ge, be, 0, s, A, B, o, (int)w_score, (int)cap, counts, cross_idx, cross_slope,
                       cross_score, cross_frac);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}

}  // extern "C"
