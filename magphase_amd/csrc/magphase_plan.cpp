// Host-side batch planners (plain C++, no device code): the reference's float64 / integer index arithmetic for a whole
// batch of utterances in one call, operation for operation what magphase_amd/hostmath.py and engine.py do in numpy (those
// stay as the readable form and as the reference the tests compare these against: tests/test_host_plans.py).  "Bit-exact
// indices"  means the same IEEE-754 sequence: products and quotients in double, np.round = round
// half to even (nearbyint in the default rounding mode), astype(int) = truncation, cumsum sequential, comparisons as
// written.  No fused multiply-add: the file is compiled with contraction off.
//
//   mpx_host_plan_analysis    libaudio.py:435-447 (epoch clean-up), magphase.py:77-98 (frame bounds), :2198-2207 (f0)
//   mpx_host_plan_synthesis   magphase.py:846-848, 861-868 (const -> variable rate), 879-882, 77-98, 969-973, 34-62
//   mpx_host_ola_runs         the run planner of the fused overlap-add (hostmath.ola_runs, default equal-share mode)
#pragma clang fp contract(off)
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../../include/magphase_hip.h"
#include "host_pool.hpp"

#include <atomic>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

extern "C" int64_t mpx_host_const_to_var_scan(const double* centres, const double* shift_c, int64_t n,
                                              double* shifts_out, double* locs_out);

namespace {

inline int64_t round_to_int(double x) { return (int64_t)std::nearbyint(x); }   // np.round(x).astype(int)

// memcpy into a page-locked staging buffer with streaming (non-temporal) stores: the destination is read next by the DMA
// engine, not by a core -- written through the cache every line is first READ for ownership, a third of the copy's memory
// traffic, and eight ranks staging 30 MB per launch each share one memory system (bench.py host_contention)
inline void stream_copy(void* dst, const void* src, size_t n) {
#if defined(__SSE2__) && !defined(MPX_HOST_NO_STREAM)
    char* d = (char*)dst;
    const char* s = (const char*)src;
    size_t head = (16 - ((uintptr_t)d & 15u)) & 15u;
    if (head > n) head = n;
    if (head) memcpy(d, s, head);
    d += head, s += head, n -= head;
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i*)(s + i)), b = _mm_loadu_si128((const __m128i*)(s + i + 16));
        const __m128i c = _mm_loadu_si128((const __m128i*)(s + i + 32)), e = _mm_loadu_si128((const __m128i*)(s + i + 48));
        _mm_stream_si128((__m128i*)(d + i), a);
        _mm_stream_si128((__m128i*)(d + i + 16), b);
        _mm_stream_si128((__m128i*)(d + i + 32), c);
        _mm_stream_si128((__m128i*)(d + i + 48), e);
    }
    _mm_sfence();
    if (i < n) memcpy(d + i, s + i, n - i);
#else
    memcpy(dst, src, n);
#endif
}

// np.minimum / np.maximum (a NaN operand propagates)
inline double np_min(double a, double b) { return std::isnan(a) ? a : (std::isnan(b) ? b : (a < b ? a : b)); }
inline double np_max(double a, double b) { return std::isnan(a) ? a : (std::isnan(b) ? b : (a > b ? a : b)); }
// scipy.signal.medfilt(v, 3) at one position (hostmath.medfilt3_batch: the median of three SELECTED, zero-padded ends)
inline double med3(double lo, double mid, double hi) { return np_max(np_min(lo, mid), np_min(np_max(lo, mid), hi)); }

// the epoch clean-up of libaudio.py:435-447 for one utterance (hostmath.clean_epochs): kept times / voicing flags
inline void clean_epochs(const double* pm_sec, const double* voi, int64_t n_ep, int64_t n, double rate,
                         std::vector<double>& t, std::vector<double>& v) {
    t.clear();
    v.clear();
    for (int64_t i = 0; i < n_ep; ++i) {   // keep = [True] + (diff(pm_sec) > 0)
        if (i == 0 || (pm_sec[i] - pm_sec[i - 1]) > 0) {
            t.push_back(pm_sec[i]);
            v.push_back(voi[i]);
        }
    }
    // epochs at or beyond the last sample are dropped (only if the LAST one is, as the reference tests it)
    if (n > 0 && !t.empty() && round_to_int(t.back() * rate) >= n - 1) {
        size_t k = 0;
        for (size_t i = 0; i < t.size(); ++i)
            if (round_to_int(t[i] * rate) < n - 1) {
                t[k] = t[i];
                v[k] = v[i];
                ++k;
            }
        t.resize(k);
        v.resize(k);
    }
}

}  // namespace

extern "C" {

int64_t mpx_host_plan_analysis(int32_t n_utts, const double* pm_sec, const double* voi, const int64_t* ep_off,
                               const int64_t* n_smpls, const double* fs, const int64_t* sig_off, int64_t* pos,
                               int64_t* pm_out, int64_t* left, int64_t* right, double* f0, int64_t* frame_off) {
    if (n_utts < 0 || (n_utts > 0 && (!pm_sec || !voi || !ep_off || !n_smpls || !fs || !sig_off || !pos || !pm_out ||
                                      !left || !right || !f0 || !frame_off)))
        return -1;
    int64_t w = 0;
    frame_off[0] = 0;
    std::vector<double> t, v;
    for (int32_t u = 0; u < n_utts; ++u) {
        const int64_t a = ep_off[u], b = ep_off[u + 1], n = n_smpls[u];
        const double rate = fs[u];
        if (b <= a) return -(int64_t)(u + 2);   // no epochs: the numpy form raises; so does the caller
        clean_epochs(pm_sec + a, voi + a, b - a, n, rate, t, v);
        const int64_t F = (int64_t)t.size();
        if (F == 0) return -(int64_t)(u + 2);
        // ext = [0, pm..., n - 1]; left = ext[1:-1] - ext[:-2]; right = ext[2:] - ext[1:-1]
        int64_t prev = 0;
        for (int64_t i = 0; i < F; ++i) {
            const int64_t p = round_to_int(t[i] * rate);
            const int64_t nxt = (i + 1 < F) ? round_to_int(t[i + 1] * rate) : n - 1;
            pm_out[w + i] = p;
            pos[w + i] = p + sig_off[u];
            left[w + i] = p - prev;
            right[w + i] = nxt - p;
            f0[w + i] = (v[i] * rate) / (double)(p - prev);   // v_voi * fs / v_shift (0 / 0 = nan, x / 0 = inf: as numpy)
            prev = p;
        }
        w += F;
        frame_off[u + 1] = w;
    }
    return w;
}

// LosslessAnalysisPlan's whole host side for a batch, utterances by pointer (see include/magphase_hip.h).
int64_t mpx_host_plan_analysis_batch(int32_t n_utts, const void* const* pcm, const int32_t* pcm_kind,
                                     const int64_t* n_smpls, const double* fs, const double* const* pm_sec,
                                     const double* const* voi, const int64_t* n_epochs, void* stage, int32_t stage_kind,
                                     int64_t* pos, int32_t* left32, int32_t* right32, float* voi32, int64_t* pm_out,
                                     int64_t* left64, double* f0, double* f0_med, int64_t* frame_off,
                                     int32_t fft_len, int64_t* long_frame, int64_t* long_len, int64_t long_cap,
                                     int64_t* n_long_out, int32_t n_threads) {
    if (n_utts < 0) return -1;
    if (n_utts > 0 && (!n_smpls || !fs || !pm_sec || !voi || !n_epochs || !pos || !left32 || !right32 || !pm_out ||
                       !left64 || !f0 || !frame_off))
        return -1;
    if (stage && (!pcm || !pcm_kind)) return -1;
    if (n_long_out) *n_long_out = 0;
    frame_off[0] = 0;
    if (n_utts == 0) return 0;
    const int U = n_utts;
    // sample offsets of the utterances in the staged buffer, copy tasks of at most 256 KB of input
    std::vector<int64_t> sig_off((size_t)U + 1, 0), first((size_t)U + 1, 0);
    const int64_t kBlock = 128 << 10;   // samples per copy task
    for (int u = 0; u < U; ++u) {
        if (n_smpls[u] < 0 || n_epochs[u] < 0) return -1;
        if (stage && n_smpls[u] > 0 && (!pcm[u] || pcm_kind[u] < 0 || pcm_kind[u] > 2)) return -1;
        if (stage && stage_kind == 0 && pcm_kind[u] != 0) return -1;   // int16 staging: every utterance must be int16
        sig_off[(size_t)u + 1] = sig_off[(size_t)u] + n_smpls[u];
        first[(size_t)u + 1] = first[(size_t)u] + (stage ? (n_smpls[u] + kBlock - 1) / kBlock : 0);
    }
    const int n_copy = (int)first[(size_t)U];
    std::vector<std::vector<double>> tt((size_t)U), vv((size_t)U);
    std::atomic<int> bad(1 << 30);
    auto fail_at = [&](int u) {
        int cur = bad.load();
        while (u < cur && !bad.compare_exchange_weak(cur, u)) {
        }
    };
    auto copy_task = [&](int t) {
        const int u = (int)(std::upper_bound(first.begin(), first.end(), (int64_t)t) - first.begin()) - 1;
        const int64_t a = ((int64_t)t - first[(size_t)u]) * kBlock;
        const int64_t e = (a + kBlock < n_smpls[u]) ? a + kBlock : n_smpls[u];
        const int64_t o = sig_off[(size_t)u];
        if (stage_kind == 0) {   // int16 in, int16 staged (widened on the device: mpx_pcm16_to_f32)
            stream_copy((int16_t*)stage + o + a, (const int16_t*)pcm[u] + a, (size_t)(e - a) * 2);
            return;
        }
        float* d = (float*)stage + o;
        if (pcm_kind[u] == 0) {          // int16 * 2^-15: exact (== astype(float32) / 32768)
            const int16_t* s = (const int16_t*)pcm[u];
            for (int64_t i = a; i < e; ++i) d[i] = (float)s[i] * (1.0f / 32768.0f);
        } else if (pcm_kind[u] == 1) {
            stream_copy(d + a, (const float*)pcm[u] + a, (size_t)(e - a) * 4);
        } else {                         // float64 -> float32, round to nearest even (numpy's astype)
            const double* s = (const double*)pcm[u];
            for (int64_t i = a; i < e; ++i) d[i] = (float)s[i];
        }
    };
    try {
        // pass 1: the staged copy's tasks and, per utterance, the cleaned epoch list (its length = the frame count)
        mpx_host::parallel_for(n_copy + U, n_threads, [&](int t) {
            if (t < n_copy) {
                copy_task(t);
                return;
            }
            const int u = t - n_copy;
            if (n_epochs[u] <= 0 || !pm_sec[u] || !voi[u]) {
                fail_at(u);
                return;
            }
            clean_epochs(pm_sec[u], voi[u], n_epochs[u], n_smpls[u], fs[u], tt[(size_t)u], vv[(size_t)u]);
            if (tt[(size_t)u].empty()) fail_at(u);
        });
        if (bad.load() != (1 << 30)) return -(int64_t)(bad.load() + 2);
        for (int u = 0; u < U; ++u) frame_off[u + 1] = frame_off[u] + (int64_t)tt[(size_t)u].size();
        // pass 2: frame bounds (magphase.py:77-98), f0 (:2198-2207), its median-3 (signal.medfilt), the device tables
        mpx_host::parallel_for(U, n_threads, [&](int u) {
            const std::vector<double>&t = tt[(size_t)u], &v = vv[(size_t)u];
            const int64_t F = (int64_t)t.size(), w = frame_off[u], n = n_smpls[u];
            const double rate = fs[u];
            int64_t prev = 0;
            for (int64_t i = 0; i < F; ++i) {
                const int64_t p = round_to_int(t[(size_t)i] * rate);
                const int64_t nxt = (i + 1 < F) ? round_to_int(t[(size_t)i + 1] * rate) : n - 1;
                pm_out[w + i] = p;
                pos[w + i] = p + sig_off[(size_t)u];
                left64[w + i] = p - prev;
                left32[w + i] = (int32_t)(p - prev);
                right32[w + i] = (int32_t)(nxt - p);
                const double f = (v[(size_t)i] * rate) / (double)(p - prev);   // v_voi * fs / v_shift (as numpy: nan / inf kept)
                f0[w + i] = f;
                if (voi32) voi32[w + i] = (f > 0) ? 1.0f : 0.0f;
                prev = p;
            }
            if (f0_med)
                for (int64_t i = 0; i < F; ++i)
                    f0_med[w + i] = med3(i > 0 ? f0[w + i - 1] : 0.0, f0[w + i], i + 1 < F ? f0[w + i + 1] : 0.0);
        });
    } catch (...) {
        return -1;
    }
    const int64_t F_tot = frame_off[U];
    if (fft_len > 0 && n_long_out) {   // frames longer than fft_len (the reference warns once per such frame; rare)
        int64_t k = 0;
        for (int64_t i = 0; i < F_tot; ++i) {
            const int64_t tot = (int64_t)left32[i] + (int64_t)right32[i] + 1;
            if (tot > fft_len) {
                if (k < long_cap && long_frame && long_len) {
                    long_frame[k] = i;
                    long_len[k] = tot;
                }
                ++k;
            }
        }
        *n_long_out = k;
    }
    return F_tot;
}

}  // extern "C"

namespace {

// One utterance of CompressedSynthesisPlan's loop (magphase.py:846-848, 861-868, 879-882, 77-98, 969-973, 34-62).
// f0 = exp(lf0) of the utterance's n_rows rows (numpy's exp, evaluated by the caller for the whole batch: the one
// transcendental of the planner stays bit-identical to the array API's).  Row indices are relative to the utterance,
// noise positions to the utterance's own noise.  Returns false where the numpy form raises.
struct SynthUtt {
    std::vector<int64_t> sft, pm, rel;
    std::vector<int32_t> nleft, nright, wtype, voiced, row0, row1, win_l, win_r;
    std::vector<double> rowt;
    int64_t n = 0, ns_len = 0, out_start = 0, out_len = 0;
};
struct SynthScratch {
    std::vector<double> shift_c, centres, sh, loc;
    std::vector<char> voi_c;
};

bool plan_synth_utt(const double* f0, int64_t n_rows, double fs, int32_t fft_len, int32_t b_const_rate,
                    int32_t b_voi_ap_win, SynthUtt& o, SynthScratch& w) {
    const int64_t N = fft_len, half = N / 2;
    if (n_rows < 2) return false;   // v_pm[-2] below: the numpy form raises IndexError
    // v_voi = f0 > 1; v_shift = fs / where(f0 == 0, 200, f0)
    w.shift_c.resize((size_t)n_rows);
    w.voi_c.resize((size_t)n_rows);
    for (int64_t i = 0; i < n_rows; ++i) {
        const double f = f0[i];
        w.voi_c[(size_t)i] = f > 1.0;
        w.shift_c[(size_t)i] = fs / ((f == 0) ? 200.0 : f);
    }
    int64_t n = n_rows;
    const double* shp = w.shift_c.data();
    int64_t s0 = 0;
    if (b_const_rate) {
        const double step = fs * 5.0 / 1000;
        w.centres.resize((size_t)n_rows);
        for (int64_t i = 0; i < n_rows; ++i) w.centres[(size_t)i] = step * (double)(i + 1);
        w.sh.assign((size_t)(2 * n_rows), 0.0);
        w.loc.assign((size_t)(2 * n_rows), 0.0);
        s0 = mpx_host_const_to_var_scan(w.centres.data(), w.shift_c.data(), n_rows, w.sh.data(), w.loc.data());
        if (s0 < 0) return false;
        n = 2 * n_rows - s0;
        shp = w.sh.data() + s0;
    }
    if (n < 2) return false;
    o.n = n;
    o.sft.resize((size_t)n), o.pm.resize((size_t)n), o.rel.resize((size_t)n);
    o.nleft.resize((size_t)n), o.nright.resize((size_t)n), o.wtype.resize((size_t)n), o.voiced.resize((size_t)n);
    o.row0.resize((size_t)n), o.row1.resize((size_t)n), o.win_l.resize((size_t)n), o.win_r.resize((size_t)n);
    o.rowt.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const double s = shp[i];
        if (!std::isfinite(s) || std::fabs(s) > 1.0e15) return false;
        o.sft[(size_t)i] = (int64_t)s;   // astype(int): truncation
    }
    // rows / weights / voicing of the variable-rate frames
    const std::vector<double>& centres = w.centres;
    for (int64_t i = 0; i < n; ++i) {
        if (b_const_rate) {
            const double x = w.loc[(size_t)(s0 + i)];
            // outside the constant-rate grid scipy's interp1d raises (bounds_error): hand the utterance to the numpy
            // form, which raises the reference's ValueError, instead of extrapolating silently
            if (!(x >= centres[0] && x <= centres[(size_t)(n_rows - 1)])) return false;
            int64_t lo = 0, hi = n_rows;   // np.searchsorted(centres, x, 'left')
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (centres[(size_t)mid] < x) lo = mid + 1; else hi = mid;
            }
            const int64_t idx = lo < 1 ? 1 : (lo > n_rows - 1 ? n_rows - 1 : lo);
            const double x_lo = centres[(size_t)(idx - 1)], x_hi = centres[(size_t)idx];
            const double y_lo = w.voi_c[(size_t)(idx - 1)] ? 1.0 : 0.0, y_hi = w.voi_c[(size_t)idx] ? 1.0 : 0.0;
            const double slope = (y_hi - y_lo) / (x_hi - x_lo);   // scipy interp1d._call_linear
            const double y = slope * (x - x_lo) + y_lo;
            o.voiced[(size_t)i] = y > 0.5;
            o.row0[(size_t)i] = (int32_t)(idx - 1);
            o.row1[(size_t)i] = (int32_t)idx;
            o.rowt[(size_t)i] = (x - x_lo) / (x_hi - x_lo);
        } else {
            o.voiced[(size_t)i] = w.voi_c[(size_t)i];
            o.row0[(size_t)i] = o.row1[(size_t)i] = (int32_t)i;
            o.rowt[(size_t)i] = 0.0;
        }
        o.wtype[(size_t)i] = (o.voiced[(size_t)i] && b_voi_ap_win) ? 1 : 0;
    }
    // v_pm = cumsum(v_shift); ns_len = v_pm[-1] + (v_pm[-1] - v_pm[-2])
    int64_t acc = 0;
    for (int64_t i = 0; i < n; ++i) {
        acc += o.sft[(size_t)i];
        o.pm[(size_t)i] = acc;
    }
    const int64_t last = o.pm[(size_t)(n - 1)], ns_len = last + (last - o.pm[(size_t)(n - 2)]);
    // frame_bounds(v_pm, ns_len): ext = [0, pm..., ns_len - 1]
    int64_t prev = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t p = o.pm[(size_t)i], nxt = (i + 1 < n) ? o.pm[(size_t)(i + 1)] : ns_len - 1;
        const int64_t l = p - prev, r = nxt - p;
        if (l > half || r + 1 > half) return false;   // "negative dimensions are not allowed"
        o.nleft[(size_t)i] = (int32_t)l;
        o.nright[(size_t)i] = (int32_t)r;
        prev = p;
    }
    // anti-ringing window half lengths: se = [s0, s..., s_last, s_last]; wl = se[i] + se[i+1]; wr = se[i+2] + se[i+3]
    auto se = [&](int64_t k) { return o.sft[(size_t)(k <= 0 ? 0 : (k > n ? n - 1 : k - 1))]; };
    for (int64_t i = 0; i < n; ++i) {
        const int64_t wl = se(i) + se(i + 1), wr = se(i + 2) + se(i + 3);
        if (wl > half || wr + 1 > half) return false;
        o.win_l[(size_t)i] = (int32_t)wl;
        o.win_r[(size_t)i] = (int32_t)wr;
    }
    // ola_plan(v_pm, N)
    const int64_t first = o.pm[0], buf_len = last + N;
    int64_t start = half - first;
    if (start < 0) start = (buf_len + start > 0) ? buf_len + start : 0;
    if (start > buf_len) start = buf_len;
    const int64_t len1 = buf_len - start;
    int64_t stop = last + (last - o.pm[(size_t)(n - 2)]) + 1;
    if (stop < 0) stop = (len1 + stop > 0) ? len1 + stop : 0;
    for (int64_t i = 0; i < n; ++i) o.rel[(size_t)i] = o.pm[(size_t)i] - first;
    o.out_start = start;
    o.out_len = len1 < stop ? len1 : stop;
    o.ns_len = ns_len;
    return true;
}

}  // namespace

extern "C" {

int64_t mpx_host_plan_synthesis(int32_t n_utts, const double* f0, const int64_t* row_off, double fs, int32_t fft_len,
                                int32_t b_const_rate, int32_t b_voi_ap_win, int64_t cap, int64_t* v_shift, int64_t* v_pm,
                                int64_t* npos, int32_t* nleft, int32_t* nright, int32_t* wtype, int32_t* voiced,
                                int32_t* row0, int32_t* row1, double* rowt, int32_t* win_l, int32_t* win_r,
                                int64_t* pm_rel, int64_t* frame_off, int64_t* ns_len_out, int64_t* out_start,
                                int64_t* out_len) {
    if (n_utts < 0 || fft_len <= 0 || !(fs > 0)) return -1;
    if (n_utts > 0 && (!f0 || !row_off || !v_shift || !v_pm || !npos || !nleft || !nright || !wtype || !voiced || !row0 ||
                       !row1 || !rowt || !win_l || !win_r || !pm_rel || !frame_off || !ns_len_out || !out_start || !out_len))
        return -1;
    int64_t w = 0, noise_base = 0;
    frame_off[0] = 0;
    SynthUtt o;
    SynthScratch scr;
    for (int32_t u = 0; u < n_utts; ++u) {
        const int64_t a = row_off[u], n_rows = row_off[u + 1] - row_off[u];
        if (!plan_synth_utt(f0 + a, n_rows, fs, fft_len, b_const_rate, b_voi_ap_win, o, scr)) return -(int64_t)(u + 2);
        const int64_t n = o.n;
        if (w + n > cap) return -1000000;
        for (int64_t i = 0; i < n; ++i) {
            v_shift[w + i] = o.sft[(size_t)i];
            v_pm[w + i] = o.pm[(size_t)i];
            npos[w + i] = o.pm[(size_t)i] + noise_base;
            nleft[w + i] = o.nleft[(size_t)i];
            nright[w + i] = o.nright[(size_t)i];
            wtype[w + i] = o.wtype[(size_t)i];
            voiced[w + i] = o.voiced[(size_t)i];
            row0[w + i] = o.row0[(size_t)i] + (int32_t)a;
            row1[w + i] = o.row1[(size_t)i] + (int32_t)a;
            rowt[w + i] = o.rowt[(size_t)i];
            win_l[w + i] = o.win_l[(size_t)i];
            win_r[w + i] = o.win_r[(size_t)i];
            pm_rel[w + i] = o.rel[(size_t)i];
        }
        out_start[u] = o.out_start;
        out_len[u] = o.out_len;
        ns_len_out[u] = o.ns_len;
        noise_base += o.ns_len;
        w += n;
        frame_off[u + 1] = w;
    }
    return w;
}

// LosslessSynthesisPlan's loop: v_pm = cumsum(f0_to_shift(f0, fs)).astype(int) (float cumsum, then truncation: Q3,
// magphase.py:1771-1772) and ola's offsets (magphase.py:34-62) per utterance.
int64_t mpx_host_plan_lossless_synthesis(int32_t n_utts, const double* f0, const int64_t* frame_off, const double* fs,
                                         int32_t fft_len, int64_t* v_pm, int64_t* pm_rel, int64_t* out_start,
                                         int64_t* out_len) {
    if (n_utts < 0 || fft_len <= 0) return -1;
    if (n_utts > 0 && (!f0 || !frame_off || !fs || !v_pm || !pm_rel || !out_start || !out_len)) return -1;
    const int64_t N = fft_len, half = N / 2;
    for (int32_t u = 0; u < n_utts; ++u) {
        const int64_t a = frame_off[u], n = frame_off[u + 1] - frame_off[u];
        if (n < 1) return -(int64_t)(u + 2);
        double acc = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            const double f = f0[a + i];
            acc += fs[u] / ((f == 0) ? 200.0 : f);
            if (!std::isfinite(acc) || std::fabs(acc) > 1.0e15) return -(int64_t)(u + 2);
            v_pm[a + i] = (int64_t)acc;
        }
        const int64_t first = v_pm[a], last = v_pm[a + n - 1], buf_len = last + N;
        int64_t start = half - first;
        if (start < 0) start = (buf_len + start > 0) ? buf_len + start : 0;
        if (start > buf_len) start = buf_len;
        const int64_t len1 = buf_len - start;
        const int64_t last_shift = (n > 1) ? last - v_pm[a + n - 2] : last;
        int64_t stop = last + last_shift + 1;
        if (stop < 0) stop = (len1 + stop > 0) ? len1 + stop : 0;
        for (int64_t i = 0; i < n; ++i) pm_rel[a + i] = v_pm[a + i] - first;
        out_start[u] = start;
        out_len[u] = len1 < stop ? len1 : stop;
    }
    return n_utts > 0 ? frame_off[n_utts] : 0;
}

// hostmath.ola_runs in its default mode (global equal shares `gcuts`, computed by the caller).  runs: capacity
// n_utts + n_gcuts records; returns the number of runs or a negative error.
int64_t mpx_host_ola_runs(int32_t n_utts, const int64_t* pm_rel, const int64_t* frame_off, const int64_t* starts,
                          const int64_t* out_lens, const int64_t* out_offs, int32_t fft_len, const int64_t* gcuts,
                          int64_t n_gcuts, mpx_ola_run* runs, int64_t cap_runs) {
    if (n_utts < 0 || fft_len <= 0 || n_gcuts < 1) return -1;
    if (n_utts > 0 && (!pm_rel || !frame_off || !starts || !out_lens || !out_offs || !gcuts || !runs)) return -1;
    const int64_t N = fft_len, strip_floats = N + 64;
    int64_t nr = 0;
    std::vector<int64_t> cuts;
    for (int32_t u = 0; u < n_utts; ++u) {
        const int64_t f_base = frame_off[u], n = frame_off[u + 1] - frame_off[u];
        if (n == 0) continue;
        const int64_t* rel = pm_rel + f_base;
        const int64_t start = starts[u], out_len = out_lens[u], o0 = out_offs[u];
        // cuts = [0] + (global cuts strictly inside this utterance) + [n]
        cuts.clear();
        cuts.push_back(0);
        for (int64_t g = 0; g < n_gcuts; ++g)
            if (gcuts[g] > f_base && gcuts[g] < f_base + n) cuts.push_back(gcuts[g] - f_base);
        cuts.push_back(n);
        // _enforce_span (hostmath.py): a run with both neighbours must satisfy rel[next run's first] - rel[own first - 1]
        // >= N; a cut that comes too early moves forward to the first frame that is far enough if that leaves its
        // successor a frame, otherwise it is dropped (the run grows into its successor)
        {
            size_t k = 1;
            while (k + 2 < cuts.size()) {
                const int64_t need = rel[cuts[k] - 1] + N;
                if (rel[cuts[k + 1]] < need) {
                    const int64_t c2 = std::lower_bound(rel, rel + n, need) - rel;   // rel ascends within an utterance
                    if (c2 < cuts[k + 2]) {
                        cuts[k + 1] = c2;
                        ++k;
                    } else {
                        cuts.erase(cuts.begin() + (long)(k + 1));
                    }
                } else {
                    ++k;
                }
            }
        }
        const int64_t k = (int64_t)cuts.size() - 1;
        if (nr + k > cap_runs) return -1000000;
        int64_t prev_hi = 0;
        for (int64_t i = 0; i < k; ++i) {
            const int64_t fb = cuts[(size_t)i], fe = cuts[(size_t)i + 1];
            const int64_t hi = rel[fe - 1] + N;
            const int64_t lo = (rel[fb] < prev_hi) ? rel[fb] : prev_hi;
            int64_t m = (lo - start + o0) % 64;   // numpy's % : non-negative for a positive divisor
            if (m < 0) m += 64;
            const int64_t x0 = lo - m;
            const int64_t head_end = (i > 0) ? prev_hi - x0 : 0;
            if (head_end > strip_floats) return -2000000;
            const int64_t own_lo = (i > 0) ? prev_hi : 0;
            int64_t own_hi = hi;
            if (i == k - 1 && start + out_len > own_hi) own_hi = start + out_len;
            const int64_t out_lo = (own_lo > start ? own_lo : start) - x0;
            int64_t out_hi = (own_hi < start + out_len ? own_hi : start + out_len) - x0;
            if (out_hi < out_lo) out_hi = out_lo;
            const int64_t flush_end = (hi > own_hi ? hi : own_hi) - x0;
            const int64_t fix_lo = (lo > start ? lo : start) - x0;
            int64_t fix_hi = (prev_hi < start + out_len ? prev_hi : start + out_len) - x0;
            fix_hi = (i > 0) ? (fix_hi > fix_lo ? fix_hi : fix_lo) : fix_lo;
            mpx_ola_run& r = runs[nr];
            std::memset(&r, 0, sizeof r);
            r.frame_begin = (int32_t)(fb + f_base);
            r.frame_end = (int32_t)(fe + f_base);
            r.x0 = (int32_t)x0;
            r.head_end = (int32_t)head_end;
            r.out_lo = (int32_t)out_lo;
            r.out_hi = (int32_t)out_hi;
            r.flush_end = (int32_t)flush_end;
            r.fix_lo = (int32_t)fix_lo;
            r.fix_hi = (int32_t)fix_hi;
            r.out_base = o0 + x0 - start;
            r.strip_off = nr * strip_floats;
            ++nr;
            prev_hi = hi;
        }
    }
    return nr;
}

// hostmath.slot_cuts: frame indices that deal `total` frames to the slots (cuts[0] == 0, n_cuts = ns + 1).  wcum / wsum:
// np.concatenate(([0.], np.cumsum(w))) and w.sum() of the slots' float64 weights, evaluated by numpy ONCE per engine (the
// pairwise sum is numpy's); null = equal shares, np.round(np.linspace(0, total, ns + 1)).
static void slot_cuts(int64_t total, int32_t ns, const double* wcum, double wsum, std::vector<int64_t>& cuts) {
    cuts.resize((size_t)ns + 1);
    if (wcum) {
        for (int32_t i = 0; i <= ns; ++i) cuts[(size_t)i] = round_to_int((double)total * wcum[i] / wsum);
    } else {
        const double step = (double)total / (double)ns;   // np.linspace: arange(num) * step + start, last = stop
        for (int32_t i = 0; i <= ns; ++i) cuts[(size_t)i] = round_to_int((double)i * step + 0.0);
        cuts[(size_t)ns] = total;
    }
}

// CompressedSynthesisPlan's whole host side for a batch (see include/magphase_hip.h).
int64_t mpx_host_plan_synthesis_batch(int32_t n_utts, const void* const* mag, const void* const* real,
                                      const void* const* imag, const int32_t* kind, const int64_t* n_rows,
                                      int32_t mag_dim, int32_t phase_dim, float* stage, const double* f0, double fs,
                                      int32_t fft_len, int32_t b_const_rate, int32_t b_voi_ap_win, int32_t n_slots,
                                      const double* wcum, double wsum, int32_t want_tiles, uint8_t* desc,
                                      int64_t desc_cap, int64_t* desc_off, int64_t* v_shift, int64_t* v_pm,
                                      int32_t* voiced_host, int64_t* frame_off, int64_t* ns_len, int64_t* out_start,
                                      int64_t* out_len, mpx_ola_run* runs_host, int64_t runs_cap, int64_t* counts,
                                      int32_t n_threads) {
    if (n_utts <= 0 || fft_len <= 0 || !(fs > 0) || n_slots < 1 || mag_dim < 1 || phase_dim < 1) return -1;
    if (!n_rows || !f0 || !desc || !desc_off || !v_shift || !v_pm || !voiced_host || !frame_off || !ns_len ||
        !out_start || !out_len || !runs_host || !counts)
        return -1;
    if (stage && (!mag || !real || !imag || !kind)) return -1;
    const int U = n_utts;
    std::vector<int64_t> row_off((size_t)U + 1, 0);
    for (int u = 0; u < U; ++u) {
        if (n_rows[u] < 0) return -1;
        if (stage && n_rows[u] > 0 && (!mag[u] || !real[u] || !imag[u] || kind[u] < 1 || kind[u] > 2)) return -1;
        row_off[(size_t)u + 1] = row_off[(size_t)u] + n_rows[u];
    }
    const int64_t R = row_off[(size_t)U];
    // ---- staged coefficient rows [R x mag_dim | R x phase_dim | R x phase_dim] (float32) + the per-utterance plans
    const int64_t kBlock = 64 << 10;   // elements per copy task
    struct CopyJob {
        const void* src;
        int32_t kind;
        int64_t n, dst;
    };
    std::vector<CopyJob> jobs;
    std::vector<int64_t> first(1, 0);
    if (stage) {
        const int64_t n_m = R * mag_dim, n_p = R * phase_dim;
        for (int u = 0; u < U; ++u) {
            const int64_t r0 = row_off[(size_t)u], nr = n_rows[u];
            if (nr == 0) continue;
            jobs.push_back({mag[u], kind[u], nr * mag_dim, r0 * mag_dim});
            jobs.push_back({real[u], kind[u], nr * phase_dim, n_m + r0 * phase_dim});
            jobs.push_back({imag[u], kind[u], nr * phase_dim, n_m + n_p + r0 * phase_dim});
        }
        for (const CopyJob& j : jobs) first.push_back(first.back() + (j.n + kBlock - 1) / kBlock);
    }
    const int n_copy = (int)first.back();
    std::vector<SynthUtt> plans((size_t)U);
    std::atomic<int> bad(1 << 30);
    try {
        mpx_host::parallel_for(n_copy + U, n_threads, [&](int t) {
            if (t < n_copy) {
                const int j = (int)(std::upper_bound(first.begin(), first.end(), (int64_t)t) - first.begin()) - 1;
                const CopyJob& c = jobs[(size_t)j];
                const int64_t a = ((int64_t)t - first[(size_t)j]) * kBlock, e = (a + kBlock < c.n) ? a + kBlock : c.n;
                float* d = stage + c.dst;
                if (c.kind == 1) {
                    stream_copy(d + a, (const float*)c.src + a, (size_t)(e - a) * 4);
                } else {   // float64 -> float32, round to nearest even (numpy's astype)
                    const double* sp = (const double*)c.src;
                    for (int64_t i = a; i < e; ++i) d[i] = (float)sp[i];
                }
                return;
            }
            const int u = t - n_copy;
            static thread_local SynthScratch scr;
            if (!plan_synth_utt(f0 + row_off[(size_t)u], n_rows[u], fs, fft_len, b_const_rate, b_voi_ap_win,
                                plans[(size_t)u], scr)) {
                int cur = bad.load();
                while (u < cur && !bad.compare_exchange_weak(cur, u)) {
                }
            }
        });
    } catch (...) {
        return -1;
    }
    if (bad.load() != (1 << 30)) return -(int64_t)(bad.load() + 2);
    // ---- offsets
    frame_off[0] = 0;
    std::vector<int64_t> noise_base((size_t)U + 1, 0), out_off((size_t)U + 1, 0);
    for (int u = 0; u < U; ++u) {
        const SynthUtt& o = plans[(size_t)u];
        frame_off[u + 1] = frame_off[u] + o.n;
        noise_base[(size_t)u + 1] = noise_base[(size_t)u] + o.ns_len;
        out_off[(size_t)u + 1] = out_off[(size_t)u] + o.out_len;
        ns_len[u] = o.ns_len;
        out_start[u] = o.out_start;
        out_len[u] = o.out_len;
    }
    const int64_t F = frame_off[U];
    // ---- device tables in their final types, 256-byte aligned, in desc
    const int64_t n_tiles1 = want_tiles ? (R + 30) / 31 + 1 : 0;
    const int64_t runs_max = (int64_t)U + n_slots + 1;
    enum { T_UFO, T_NPOS, T_NLEFT, T_NRIGHT, T_WTYPE, T_VOICED, T_TILE, T_ROW0, T_ROW1, T_ROWT, T_WINL, T_WINR, T_PMREL,
           T_OSTART, T_OOFF, T_RUNS, T_SLOTOFF, T_SLOTRUNS, T_COUNT };
    const int64_t bytes[T_COUNT] = {4 * ((int64_t)U + 1), 8 * F, 4 * F, 4 * F, 4 * F, 4 * F, 4 * n_tiles1, 4 * F, 4 * F, 4 * F,
                                    4 * F, 4 * F, 4 * F, 4 * (int64_t)U, 8 * ((int64_t)U + 1),
                                    (int64_t)sizeof(mpx_ola_run) * runs_max, 4 * ((int64_t)n_slots + 1), 4 * runs_max};
    int64_t off = 0;
    for (int k = 0; k < T_COUNT; ++k) {
        off = (off + 255) / 256 * 256;
        desc_off[k] = off;
        off += bytes[k];
    }
    if (off > desc_cap) return -1000000;
    int32_t* d_ufo = (int32_t*)(desc + desc_off[T_UFO]);
    int64_t* d_npos = (int64_t*)(desc + desc_off[T_NPOS]);
    int32_t* d_nleft = (int32_t*)(desc + desc_off[T_NLEFT]);
    int32_t* d_nright = (int32_t*)(desc + desc_off[T_NRIGHT]);
    int32_t* d_wtype = (int32_t*)(desc + desc_off[T_WTYPE]);
    int32_t* d_voiced = (int32_t*)(desc + desc_off[T_VOICED]);
    int32_t* d_tile = (int32_t*)(desc + desc_off[T_TILE]);
    int32_t* d_row0 = (int32_t*)(desc + desc_off[T_ROW0]);
    int32_t* d_row1 = (int32_t*)(desc + desc_off[T_ROW1]);
    float* d_rowt = (float*)(desc + desc_off[T_ROWT]);
    int32_t* d_winl = (int32_t*)(desc + desc_off[T_WINL]);
    int32_t* d_winr = (int32_t*)(desc + desc_off[T_WINR]);
    int32_t* d_pmrel = (int32_t*)(desc + desc_off[T_PMREL]);
    int32_t* d_ostart = (int32_t*)(desc + desc_off[T_OSTART]);
    int64_t* d_ooff = (int64_t*)(desc + desc_off[T_OOFF]);
    mpx_ola_run* d_runs = (mpx_ola_run*)(desc + desc_off[T_RUNS]);
    int32_t* d_slotoff = (int32_t*)(desc + desc_off[T_SLOTOFF]);
    int32_t* d_slotruns = (int32_t*)(desc + desc_off[T_SLOTRUNS]);
    std::vector<int64_t> rel64((size_t)(F > 0 ? F : 1));
    try {
        mpx_host::parallel_for(U, n_threads, [&](int u) {
            const SynthUtt& o = plans[(size_t)u];
            const int64_t w = frame_off[u], a = row_off[(size_t)u], nb = noise_base[(size_t)u];
            for (int64_t i = 0; i < o.n; ++i) {
                v_shift[w + i] = o.sft[(size_t)i];
                v_pm[w + i] = o.pm[(size_t)i];
                voiced_host[w + i] = o.voiced[(size_t)i];
                d_npos[w + i] = o.pm[(size_t)i] + nb;
                d_nleft[w + i] = o.nleft[(size_t)i];
                d_nright[w + i] = o.nright[(size_t)i];
                d_wtype[w + i] = o.wtype[(size_t)i];
                d_voiced[w + i] = o.voiced[(size_t)i];
                d_row0[w + i] = o.row0[(size_t)i] + (int32_t)a;
                d_row1[w + i] = o.row1[(size_t)i] + (int32_t)a;
                d_rowt[w + i] = (float)o.rowt[(size_t)i];
                d_winl[w + i] = o.win_l[(size_t)i];
                d_winr[w + i] = o.win_r[(size_t)i];
                d_pmrel[w + i] = (int32_t)o.rel[(size_t)i];
                rel64[(size_t)(w + i)] = o.rel[(size_t)i];
            }
            d_ufo[u] = (int32_t)w;
            d_ostart[u] = (int32_t)o.out_start;
            d_ooff[u] = out_off[(size_t)u];
        });
    } catch (...) {
        return -1;
    }
    d_ufo[U] = (int32_t)F;
    d_ooff[U] = out_off[(size_t)U];
    if (want_tiles) {   // frames of every 31-row tile of the coefficient matrix (mpx_mel_unwarp_rows): row0 must ascend over
        // the batch with row1 - row0 in {0, 1} (CompressedSynthesisPlan._check_rows_for_tiles)
        for (int64_t i = 0; i < F; ++i) {
            const int32_t dr = d_row1[i] - d_row0[i];
            if (dr < 0 || dr > 1 || (i > 0 && d_row0[i] < d_row0[i - 1])) return -3000000;
        }
        int64_t f = 0;
        for (int64_t t = 0; t < n_tiles1; ++t) {   // np.searchsorted(row0, 31 t, 'left')
            while (f < F && d_row0[f] < 31 * t) ++f;
            d_tile[t] = (int32_t)f;
        }
    }
    // ---- OLA runs and the slots' work lists (hostmath.ola_runs / hostplan.ola_runs)
    const int32_t ns = (int32_t)std::min<int64_t>(n_slots, std::max<int64_t>(F, 1));
    if (wcum && ns != n_slots) {   // fewer frames than slots: the weights are cut to the first F (their sum is numpy's):
        counts[0] = F;             // the caller calls again with n_slots = F and the cut weights' cumsum / sum
        return -4000000;
    }
    std::vector<int64_t> gcuts;
    slot_cuts(F, ns, wcum, wsum, gcuts);
    const int64_t nr = mpx_host_ola_runs(U, rel64.data(), frame_off, out_start, out_len, out_off.data(), fft_len,
                                         gcuts.data(), (int64_t)gcuts.size(), d_runs, runs_max);
    if (nr < 0) return nr <= -1000000 ? nr : -5000000;
    if (nr > runs_cap) return -1000000;
    memcpy(runs_host, d_runs, (size_t)nr * sizeof(mpx_ola_run));
    {   // slot_of = clip(searchsorted(gcuts, frame_begin, 'right') - 1, 0, ns - 1); slot_off = searchsorted(slot_of, arange(ns + 1))
        int64_t r = 0;
        for (int32_t sl = 0; sl <= ns; ++sl) {
            while (r < nr) {
                const int64_t fb = d_runs[r].frame_begin;
                int64_t so = (std::upper_bound(gcuts.begin(), gcuts.end(), fb) - gcuts.begin()) - 1;
                so = so < 0 ? 0 : (so > ns - 1 ? ns - 1 : so);
                if (so >= sl) break;
                ++r;
            }
            d_slotoff[sl] = (int32_t)r;
        }
        for (int64_t i = 0; i < nr; ++i) d_slotruns[i] = (int32_t)i;
    }
    counts[0] = F;
    counts[1] = nr;
    counts[2] = ns;
    counts[3] = off;              // bytes of desc in use
    counts[4] = noise_base[(size_t)U];
    counts[5] = out_off[(size_t)U];
    counts[6] = n_tiles1;
    counts[7] = R;
    return F;
}

}  // extern "C"
