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

extern "C" int64_t mpx_host_const_to_var_scan(const double* centres, const double* shift_c, int64_t n,
                                              double* shifts_out, double* locs_out);

namespace {

inline int64_t round_to_int(double x) { return (int64_t)std::nearbyint(x); }   // np.round(x).astype(int)

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
        // keep = [True] + (diff(pm_sec) > 0)
        t.clear();
        v.clear();
        for (int64_t i = a; i < b; ++i) {
            if (i == a || (pm_sec[i] - pm_sec[i - 1]) > 0) {
                t.push_back(pm_sec[i]);
                v.push_back(voi[i]);
            }
        }
        // epochs at or beyond the last sample are dropped (only if the LAST one is, as the reference tests it)
        if (n > 0) {
            if (round_to_int(t.back() * rate) >= n - 1) {
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

// One utterance of CompressedSynthesisPlan's loop.  Inputs: f0 = exp(lf0) (numpy's exp, evaluated by the caller for the
// whole batch: the one transcendental of the planner stays bit-identical to the array API's).
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
    const int64_t N = fft_len, half = N / 2;
    int64_t w = 0, noise_base = 0;
    frame_off[0] = 0;
    std::vector<double> shift_c, centres, sh, loc;
    std::vector<int64_t> sft;
    std::vector<char> voi_c;
    for (int32_t u = 0; u < n_utts; ++u) {
        const int64_t a = row_off[u], n_rows = row_off[u + 1] - row_off[u];
        if (n_rows < 2) return -(int64_t)(u + 2);   // v_pm[-2] below: the numpy form raises IndexError
        // v_voi = f0 > 1; v_shift = fs / where(f0 == 0, 200, f0)
        shift_c.resize((size_t)n_rows);
        voi_c.resize((size_t)n_rows);
        for (int64_t i = 0; i < n_rows; ++i) {
            const double f = f0[a + i];
            voi_c[(size_t)i] = f > 1.0;
            shift_c[(size_t)i] = fs / ((f == 0) ? 200.0 : f);
        }
        int64_t n = n_rows;
        const double* shp = shift_c.data();
        int64_t s0 = 0;
        if (b_const_rate) {
            const double step = fs * 5.0 / 1000;
            centres.resize((size_t)n_rows);
            for (int64_t i = 0; i < n_rows; ++i) centres[(size_t)i] = step * (double)(i + 1);
            sh.assign((size_t)(2 * n_rows), 0.0);
            loc.assign((size_t)(2 * n_rows), 0.0);
            s0 = mpx_host_const_to_var_scan(centres.data(), shift_c.data(), n_rows, sh.data(), loc.data());
            if (s0 < 0) return -(int64_t)(u + 2);
            n = 2 * n_rows - s0;
            shp = sh.data() + s0;
        }
        if (n < 2) return -(int64_t)(u + 2);
        if (w + n > cap) return -1000000;
        sft.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            const double s = shp[i];
            if (!std::isfinite(s) || std::fabs(s) > 1.0e15) return -(int64_t)(u + 2);
            sft[(size_t)i] = (int64_t)s;   // astype(int): truncation
        }
        // rows / weights / voicing of the variable-rate frames
        for (int64_t i = 0; i < n; ++i) {
            if (b_const_rate) {
                const double x = loc[(size_t)(s0 + i)];
                // outside the constant-rate grid scipy's interp1d raises (bounds_error): hand the utterance to the numpy
                // form, which raises the reference's ValueError, instead of extrapolating silently
                if (!(x >= centres[0] && x <= centres[(size_t)(n_rows - 1)])) return -(int64_t)(u + 2);
                int64_t lo = 0, hi = n_rows;   // np.searchsorted(centres, x, 'left')
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (centres[(size_t)mid] < x) lo = mid + 1; else hi = mid;
                }
                const int64_t idx = lo < 1 ? 1 : (lo > n_rows - 1 ? n_rows - 1 : lo);
                const double x_lo = centres[(size_t)(idx - 1)], x_hi = centres[(size_t)idx];
                const double y_lo = voi_c[(size_t)(idx - 1)] ? 1.0 : 0.0, y_hi = voi_c[(size_t)idx] ? 1.0 : 0.0;
                const double slope = (y_hi - y_lo) / (x_hi - x_lo);   // scipy interp1d._call_linear
                const double y = slope * (x - x_lo) + y_lo;
                voiced[w + i] = y > 0.5;
                row0[w + i] = (int32_t)(idx - 1 + a);
                row1[w + i] = (int32_t)(idx + a);
                rowt[w + i] = (x - x_lo) / (x_hi - x_lo);
            } else {
                voiced[w + i] = voi_c[(size_t)i];
                row0[w + i] = row1[w + i] = (int32_t)(i + a);
                rowt[w + i] = 0.0;
            }
            wtype[w + i] = (voiced[w + i] && b_voi_ap_win) ? 1 : 0;
        }
        // v_pm = cumsum(v_shift); ns_len = v_pm[-1] + (v_pm[-1] - v_pm[-2])
        int64_t acc = 0;
        for (int64_t i = 0; i < n; ++i) {
            acc += sft[(size_t)i];
            v_shift[w + i] = sft[(size_t)i];
            v_pm[w + i] = acc;
        }
        const int64_t last = v_pm[w + n - 1], ns_len = last + (last - v_pm[w + n - 2]);
        // frame_bounds(v_pm, ns_len): ext = [0, pm..., ns_len - 1]
        int64_t prev = 0;
        for (int64_t i = 0; i < n; ++i) {
            const int64_t p = v_pm[w + i], nxt = (i + 1 < n) ? v_pm[w + i + 1] : ns_len - 1;
            const int64_t l = p - prev, r = nxt - p;
            if (l > half || r + 1 > half) return -(int64_t)(u + 2);   // "negative dimensions are not allowed"
            nleft[w + i] = (int32_t)l;
            nright[w + i] = (int32_t)r;
            npos[w + i] = p + noise_base;
            prev = p;
        }
        // anti-ringing window half lengths: se = [s0, s..., s_last, s_last]; wl = se[i] + se[i+1]; wr = se[i+2] + se[i+3]
        auto se = [&](int64_t k) { return sft[(size_t)(k <= 0 ? 0 : (k > n ? n - 1 : k - 1))]; };
        for (int64_t i = 0; i < n; ++i) {
            const int64_t wl = se(i) + se(i + 1), wr = se(i + 2) + se(i + 3);
            if (wl > half || wr + 1 > half) return -(int64_t)(u + 2);
            win_l[w + i] = (int32_t)wl;
            win_r[w + i] = (int32_t)wr;
        }
        // ola_plan(v_pm, N)
        const int64_t first = v_pm[w], buf_len = last + N;
        int64_t start = half - first;
        if (start < 0) start = (buf_len + start > 0) ? buf_len + start : 0;
        if (start > buf_len) start = buf_len;
        const int64_t len1 = buf_len - start;
        int64_t stop = last + (last - v_pm[w + n - 2]) + 1;
        if (stop < 0) stop = (len1 + stop > 0) ? len1 + stop : 0;
        for (int64_t i = 0; i < n; ++i) pm_rel[w + i] = v_pm[w + i] - first;
        out_start[u] = start;
        out_len[u] = len1 < stop ? len1 : stop;
        ns_len_out[u] = ns_len;
        noise_base += ns_len;
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

}  // extern "C"
