// Host-side file helpers of the batch scripts (the file interface either side of the hot path).
// Plain C++ (no device code): epoch-track parsing and many-file reads / writes with a few threads, called through the C
// ABI from the reader / writer threads of iobatch.py -- ctypes drops the GIL for the duration of the call, so the
// stages of the corpus pipeline really overlap (numpy.fromstring / ndarray.tofile hold it).
//
// Reference behaviour restated here:
//   mpx_host_read_est_batch   np.loadtxt(est_file, skiprows=7, usecols=[0, 1])        (libaudio.py:421-447)
//   mpx_host_write_files      lu.write_binfile / ndarray.tofile, la.write_audio_file's file write
//                                                                                      (libutils.py:193-199, libaudio.py:352-365)
//   mpx_host_read_files       lu.read_binfile's np.fromfile                            (libutils.py:201-211)
#include <algorithm>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <exception>
#include <vector>

#include "../../include/magphase_hip.h"
#include "host_pool.hpp"

namespace {

using mpx_host::parallel_for;

// No exception may leave an extern "C" entry point (ctypes would terminate the process): MPX_ERR_HOST instead.
template <typename F>
int32_t guarded(F body) {
    try {
        body();
    } catch (...) {
        return MPX_ERR_HOST;
    }
    return MPX_OK;
}

bool read_whole(const char* path, std::string& out, int* err) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        *err = errno;
        return false;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        *err = errno;
        close(fd);
        return false;
    }
    out.resize((size_t)st.st_size);
    size_t got = 0;
    while (got < out.size()) {
        const ssize_t r = read(fd, &out[got], out.size() - got);
        if (r < 0) {
            if (errno == EINTR) continue;
            *err = errno;
            close(fd);
            return false;
        }
        if (r == 0) break;
        got += (size_t)r;
    }
    out.resize(got);
    close(fd);
    return true;
}

// One decimal number at p (leading blanks skipped) -> value, end pointer; nullptr when there is no number.
// Plain decimals with <= 15 significant digits and <= 22 fractional digits are mantissa / 10^k with both operands exact
// in float64, and the IEEE division rounds correctly: the value strtod returns (Clinger's fast path).  Everything else
// (exponents, long mantissas, inf / nan) goes to strtod itself.
const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

const char* parse_double(const char* p, const char* end, double* v) {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
    if (p >= end || *p == '\n') return nullptr;
    const char* s = p;
    bool neg = false;
    if (*p == '-' || *p == '+') {
        neg = (*p == '-');
        ++p;
    }
    uint64_t mant = 0;
    int digits = 0, frac = 0;
    bool any = false;
    while (p < end && *p >= '0' && *p <= '9') {
        if (digits < 19) {
            mant = mant * 10 + (uint64_t)(*p - '0');
            if (mant) ++digits;
        } else {
            digits = 99;
        }
        any = true;
        ++p;
    }
    if (p < end && *p == '.') {
        ++p;
        while (p < end && *p >= '0' && *p <= '9') {
            if (digits < 19) {
                mant = mant * 10 + (uint64_t)(*p - '0');
                if (mant) ++digits;
                ++frac;
            } else {
                digits = 99;
            }
            any = true;
            ++p;
        }
    }
    const bool plain = any && (p >= end || *p == ' ' || *p == '\t' || *p == '\n' || *p == '\r');
    if (plain && digits <= 15 && frac <= 22) {
        const double x = (double)mant / kPow10[frac];
        *v = neg ? -x : x;
        return p;
    }
    char* e = nullptr;
    const double x = strtod(s, &e);   // the buffer is NUL-terminated by the caller
    if (e == s) return nullptr;
    *v = x;
    return e;
}

}  // namespace

extern "C" {

int32_t mpx_host_widen_f32(const float* src, double* dst, int64_t n, int32_t n_threads) {
    if (n < 0 || (n > 0 && (!src || !dst))) return MPX_ERR_ARG;
    const int64_t kBlock = 1 << 18;   // 1 MB of input per task
    const int nb = (int)((n + kBlock - 1) / kBlock);
    return guarded([&] { parallel_for(nb, n_threads, [&](int b) {
        const int64_t a = (int64_t)b * kBlock, e = (a + kBlock < n) ? a + kBlock : n;
        int64_t i = a;
#if defined(__SSE2__) && !defined(MPX_HOST_NO_STREAM)
        // streaming (non-temporal) stores: the destination is hundreds of megabytes the caller reads later, if at all --
        // written through the cache every line is first READ for ownership (0.7 GB of extra traffic per 0.7 GB written)
        for (; i < e && (reinterpret_cast<uintptr_t>(dst + i) & 15u); ++i) dst[i] = (double)src[i];
        for (; i + 4 <= e; i += 4) {
            const __m128 v = _mm_loadu_ps(src + i);
            _mm_stream_pd(dst + i, _mm_cvtps_pd(v));
            _mm_stream_pd(dst + i + 2, _mm_cvtps_pd(_mm_movehl_ps(v, v)));
        }
        _mm_sfence();
#endif
        for (; i < e; ++i) dst[i] = (double)src[i];
    }); });
}

int32_t mpx_host_narrow_f64(const double* src, float* dst, int64_t n, int32_t n_threads) {
    if (n < 0 || (n > 0 && (!src || !dst))) return MPX_ERR_ARG;
    const int64_t kBlock = 1 << 17;   // 1 MB of input per task
    const int nb = (int)((n + kBlock - 1) / kBlock);
    return guarded([&] { parallel_for(nb, n_threads, [&](int b) {
        const int64_t a = (int64_t)b * kBlock, e = (a + kBlock < n) ? a + kBlock : n;
        int64_t i = a;
#if defined(__SSE2__) && !defined(MPX_HOST_NO_STREAM)
        // (streaming stores as in mpx_host_widen_f32: the destination is the page-locked staging buffer the DMA engine reads)
        for (; i < e && (reinterpret_cast<uintptr_t>(dst + i) & 15u); ++i) dst[i] = (float)src[i];
        for (; i + 4 <= e; i += 4) {
            const __m128 lo = _mm_cvtpd_ps(_mm_loadu_pd(src + i)), hi = _mm_cvtpd_ps(_mm_loadu_pd(src + i + 2));
            _mm_stream_ps(dst + i, _mm_movelh_ps(lo, hi));   // cvtpd2ps rounds to nearest even (MXCSR default), as numpy's astype
        }
        _mm_sfence();
#endif
        for (; i < e; ++i) dst[i] = (float)src[i];   // round to nearest even, as numpy's astype
    }); });
}

// n byte ranges copied into one destination buffer at the given offsets, on a few threads: the PCM of a batch's utterances
// into the page-locked staging buffer (one numpy slice assignment per utterance is one thread at ~12 GB/s: 1.2 ms of the 5 ms
// a 32-utterance extraction batch spends on the host).
int32_t mpx_host_copy_many(int32_t n, const void* const* src, const int64_t* nbytes, const int64_t* dst_off, void* dst,
                           int32_t n_threads) {
    if (n < 0 || (n > 0 && (!src || !nbytes || !dst_off || !dst))) return MPX_ERR_ARG;
    // tasks of at most 256 KB so that a few long utterances still spread over the threads
    const int64_t kBlock = 256 << 10;
    std::vector<int64_t> first(n + 1, 0);
    for (int i = 0; i < n; ++i) {
        if (nbytes[i] < 0 || (nbytes[i] > 0 && !src[i])) return MPX_ERR_ARG;
        first[i + 1] = first[i] + (nbytes[i] + kBlock - 1) / kBlock;
    }
    return guarded([&] { parallel_for((int)first[n], n_threads, [&](int t) {
        int i = (int)(std::upper_bound(first.begin(), first.end(), (int64_t)t) - first.begin()) - 1;
        const int64_t a = ((int64_t)t - first[i]) * kBlock, e = (a + kBlock < nbytes[i]) ? a + kBlock : nbytes[i];
        memcpy((char*)dst + dst_off[i] + a, (const char*)src[i] + a, (size_t)(e - a));
    }); });
}

int32_t mpx_host_file_sizes(int32_t n, const char* const* paths, int64_t* sizes) {
    if (n < 0 || (n > 0 && (!paths || !sizes))) return MPX_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        struct stat st;
        sizes[i] = (stat(paths[i], &st) == 0) ? (int64_t)st.st_size : -(int64_t)errno;
    }
    return MPX_OK;
}

int32_t mpx_host_read_est_batch(int32_t n, const char* const* paths, int32_t skiprows, const int64_t* row_off,
                                double* col0, double* col1, int64_t* counts, int32_t n_threads) {
    if (n < 0 || skiprows < 0 || (n > 0 && (!paths || !row_off || !col0 || !col1 || !counts))) return MPX_ERR_ARG;
    return guarded([&] { parallel_for(n, n_threads, [&](int i) {
        std::string txt;
        int err = 0;
        if (!read_whole(paths[i], txt, &err)) {
            counts[i] = -(int64_t)(err ? err : EIO);
            return;
        }
        const char* p = txt.c_str();          // NUL-terminated: strtod cannot run past the end
        const char* end = p + txt.size();
        for (int r = 0; r < skiprows && p < end; ++r) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
            p = nl ? nl + 1 : end;
        }
        const int64_t cap = row_off[i + 1] - row_off[i];
        double* o0 = col0 + row_off[i];
        double* o1 = col1 + row_off[i];
        int64_t rows = 0;
        while (p < end) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
            const char* le = nl ? nl : end;
            double a, b;
            const char* q = parse_double(p, le, &a);
            if (q) {                          // blank lines are skipped, as np.loadtxt does
                q = parse_double(q, le, &b);
                if (!q) {
                    counts[i] = -(int64_t)EINVAL;   // a row with fewer than two columns: np.loadtxt raises
                    return;
                }
                if (rows >= cap) {
                    counts[i] = -(int64_t)ENOSPC;
                    return;
                }
                o0[rows] = a;
                o1[rows] = b;
                ++rows;
            }
            p = nl ? nl + 1 : end;
        }
        counts[i] = rows;
    }); });
}

int32_t mpx_host_write_files(int32_t n, const char* const* paths, const void* const* headers, const int64_t* header_bytes,
                             const void* const* bodies, const int64_t* body_bytes, int32_t* status, int32_t n_threads) {
    if (n < 0 || (n > 0 && (!paths || !bodies || !body_bytes || !status))) return MPX_ERR_ARG;
    return guarded([&] { parallel_for(n, n_threads, [&](int i) {
        status[i] = 0;
        // (round 6 tried temporary names + rename for atomic completion: the 640 renames of a 128-utterance batch doubled
        // the file stage, 0.018 -> 0.036-0.047 s; an interrupted rank's directory is left where it is instead: scripts/batch_*.py)
        const int fd = open(paths[i], O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd < 0) {
            status[i] = errno;
            return;
        }
        auto put = [&](const void* buf, int64_t len) {
            const char* p = (const char*)buf;
            while (len > 0) {
                const ssize_t w = write(fd, p, (size_t)len);
                if (w < 0) {
                    if (errno == EINTR) continue;
                    status[i] = errno;
                    return false;
                }
                p += w;
                len -= w;
            }
            return true;
        };
        bool ok = true;
        if (headers && header_bytes && headers[i] && header_bytes[i] > 0) ok = put(headers[i], header_bytes[i]);
        if (ok && body_bytes[i] > 0) put(bodies[i], body_bytes[i]);
        if (close(fd) != 0 && status[i] == 0) status[i] = errno;
    }); });
}

int32_t mpx_host_read_files(int32_t n, const char* const* paths, void* const* bufs, const int64_t* cap, int64_t* got,
                            int32_t n_threads) {
    if (n < 0 || (n > 0 && (!paths || !bufs || !cap || !got))) return MPX_ERR_ARG;
    return guarded([&] { parallel_for(n, n_threads, [&](int i) {
        const int fd = open(paths[i], O_RDONLY);
        if (fd < 0) {
            got[i] = -(int64_t)errno;
            return;
        }
        int64_t total = 0;
        char* p = (char*)bufs[i];
        while (total < cap[i]) {
            const ssize_t r = read(fd, p + total, (size_t)(cap[i] - total));
            if (r < 0) {
                if (errno == EINTR) continue;
                total = -(int64_t)errno;
                break;
            }
            if (r == 0) break;
            total += r;
        }
        close(fd);
        got[i] = total;
    }); });
}

}  // extern "C"
