// Jump-ahead polynomials of MT19937 (host side, plain C++): what lets the device continue numpy's global generator
// (np.random.uniform(-1, 1, ns_len), magphase.py:883) in MANY workgroups instead of one.
//
// The raw word sequence X[n] of the generator obeys X[n+624] = X[n+397] ^ f(X[n], X[n+1]); every bit of it is a linear
// recurring sequence over GF(2) whose characteristic polynomial phi has degree 19937.  With g(x) = x^J mod phi,
//     X[n + J] = xor over the set bits i of g of X[n + i]            (n >= 624: words the recurrence produced)
// so the 624-word window J words further on is a xor of windows of the next 19937 + 623 words -- a parallel reduction
// (k_mt_jump in magphase_comp.hip) instead of J sequential steps.  This file computes phi once (Berlekamp-Massey on
// 2 x 19937 output bits) and the ladder x^(J 2^l) mod phi by square-and-multiply, bit-packed in 64-bit words.
// Nothing here is taken from an implementation: the recurrence constants are the published MT19937 parameters
// (Matsumoto & Nishimura 1998), the jump identity is Haramoto et al. 2008.
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/magphase_hip.h"

namespace {

constexpr int kDeg = 19937;
constexpr int kW = 312;              // 64-bit words of a residue (19968 bits)
typedef std::vector<uint64_t> Bits;

inline bool get_bit(const Bits& v, int i) { return (v[(size_t)i >> 6] >> (i & 63)) & 1u; }
inline void flip_bit(Bits& v, int i) { v[(size_t)i >> 6] ^= (uint64_t)1 << (i & 63); }

// dst ^= src << sh (src: n_src words; dst large enough)
inline void xor_shifted(Bits& dst, const uint64_t* src, int n_src, int sh) {
    const int ws = sh >> 6, bs = sh & 63;
    if (bs == 0) {
        for (int w = 0; w < n_src; ++w) dst[(size_t)w + ws] ^= src[w];
    } else {
        for (int w = 0; w < n_src; ++w) {
            dst[(size_t)w + ws] ^= src[w] << bs;
            dst[(size_t)w + ws + 1] ^= src[w] >> (64 - bs);
        }
    }
}

// Raw (untempered) words of the recurrence, block by block
struct RawStream {
    uint32_t s[624];
    int idx = 624;
    explicit RawStream(uint32_t seed) {
        s[0] = seed;   // Knuth's initialisation of the published generator: any state with a non-zero reduced part does
        for (int i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
    }
    void regen() {
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
            s[i] = s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        idx = 0;
    }
    uint32_t next() {
        if (idx >= 624) regen();
        return s[idx++];
    }
};

// phi, bit j = coefficient of x^j (bit 19937 set), from the connection polynomial Berlekamp-Massey finds for one
// output bit of the stream: sum_{i=0..L} c_i s[n-i] = 0  <=>  phi(E) s = 0 with phi_j = c_{L-j}, E the shift.
Bits characteristic_polynomial() {
    const int T = 2 * kDeg + 64;
    RawStream g(5489u);
    g.regen();                                   // start at X[624]
    Bits rev((size_t)(T + 127) / 64 + 2, 0);     // rev bit t = s[T-1-t]
    for (int n = 0; n < T; ++n)
        if (g.next() & 1u) flip_bit(rev, T - 1 - n);
    const size_t nw = (size_t)(kDeg + 64) / 64 + 2;
    Bits C(nw, 0), B(nw, 0), Tm(nw, 0);
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    for (int n = 0; n < T; ++n) {
        // d = parity of sum_{i=0..L} c_i s[n-i];  s[n-i] = rev bit (T-1-n+i)
        const int off = T - 1 - n, w0 = off >> 6, sh = off & 63;
        uint64_t acc = 0;
        const int lw = L / 64 + 1;
        for (int w = 0; w < lw; ++w) {
            uint64_t r = rev[(size_t)w0 + w] >> sh;
            if (sh) r |= rev[(size_t)w0 + w + 1] << (64 - sh);
            acc ^= C[w] & r;
        }
        if (!__builtin_parityll(acc)) {
            ++m;
        } else if (2 * L <= n) {
            Tm = C;
            xor_shifted(C, B.data(), (int)nw - m / 64 - 2, m);
            L = n + 1 - L;
            B = Tm;
            m = 1;
        } else {
            xor_shifted(C, B.data(), (int)nw - m / 64 - 2, m);
            ++m;
        }
    }
    Bits phi;
    if (L != kDeg) return phi;                   // cannot happen for this generator; the caller reports it
    phi.assign(kW + 1, 0);
    for (int j = 0; j <= L; ++j)
        if (get_bit(C, L - j)) flip_bit(phi, j);
    return phi;
}

const Bits& phi_poly() {
    static std::once_flag once;
    static Bits phi;
    std::call_once(once, [] { phi = characteristic_polynomial(); });
    return phi;
}

inline uint64_t spread32(uint32_t x) {          // bit i -> bit 2 i
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}

// a (degree < 19937, kW words) -> a^2 mod phi
void square_mod(Bits& a, const Bits& phi) {
    Bits p(2 * kW + 2, 0);
    for (int w = 0; w < kW; ++w) {
        p[2 * (size_t)w] = spread32((uint32_t)a[w]);
        p[2 * (size_t)w + 1] = spread32((uint32_t)(a[w] >> 32));
    }
    for (int d = 2 * kDeg - 2; d >= kDeg; --d)
        if (get_bit(p, d)) xor_shifted(p, phi.data(), kW, d - kDeg);
    for (int w = 0; w < kW; ++w) a[w] = p[w];
}

// a -> a x mod phi
void times_x_mod(Bits& a, const Bits& phi) {
    uint64_t carry = 0;
    for (int w = 0; w < kW; ++w) {
        const uint64_t nc = a[w] >> 63;
        a[w] = (a[w] << 1) | carry;
        carry = nc;
    }
    if (get_bit(a, kDeg))
        for (int w = 0; w < kW; ++w) a[w] ^= phi[w];
}

struct Ladder {
    std::vector<Bits> g;   // g[l] = x^(J 2^l) mod phi
};
std::mutex g_mu;
std::map<int64_t, Ladder> g_ladders;

}  // namespace

extern "C" int32_t mpx_host_mt19937_jump_poly(int64_t jump_words, int32_t n_levels, uint32_t* out) {
    if (jump_words <= 0 || n_levels <= 0 || n_levels > 40 || !out) return MPX_ERR_ARG;
    const Bits& phi = phi_poly();
    if (phi.empty()) return MPX_ERR_ARG;
    std::lock_guard<std::mutex> lock(g_mu);
    Ladder& lad = g_ladders[jump_words];
    if (lad.g.empty()) {
        Bits a(kW + 1, 0);
        a[0] = 1;                                 // x^0; square-and-multiply over the bits of jump_words
        for (int b = 62; b >= 0; --b) {
            square_mod(a, phi);
            if ((jump_words >> b) & 1) times_x_mod(a, phi);
        }
        lad.g.push_back(a);
    }
    while ((int)lad.g.size() < n_levels) {
        Bits a = lad.g.back();
        square_mod(a, phi);
        lad.g.push_back(a);
    }
    for (int l = 0; l < n_levels; ++l) std::memcpy(out + (size_t)l * 624, lad.g[l].data(), 624 * sizeof(uint32_t));
    return MPX_OK;
}

// x^(jumps[i]) mod phi for n arbitrary jumps, independent of each other, on up to n_threads threads (no cache): the
// polynomials of a radix-R jump ladder -- multiples p * J * R^l, p < R, of the segment length -- which the doubling ladder
// above does not contain.  out [n x 624] uint32 (HOST), layout as mpx_host_mt19937_jump_poly.
extern "C" int32_t mpx_host_mt19937_jump_polys(const int64_t* jumps, int32_t n, uint32_t* out, int32_t n_threads) {
    if (n < 0 || (n > 0 && (!jumps || !out))) return MPX_ERR_ARG;
    for (int i = 0; i < n; ++i)
        if (jumps[i] <= 0) return MPX_ERR_ARG;
    const Bits& phi = phi_poly();
    if (phi.empty()) return MPX_ERR_ARG;
    auto one = [&](int i) {
        Bits a(kW + 1, 0);
        a[0] = 1;
        bool started = false;
        for (int b = 62; b >= 0; --b) {
            if (started) square_mod(a, phi);
            if ((jumps[i] >> b) & 1) {
                times_x_mod(a, phi);
                started = true;
            }
        }
        std::memcpy(out + (size_t)i * 624, a.data(), 624 * sizeof(uint32_t));
    };
    const int nt = n_threads < 1 ? 1 : (n_threads > n ? n : n_threads);
    try {
        if (nt <= 1) {
            for (int i = 0; i < n; ++i) one(i);
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&, t] {
                    for (int i = t; i < n; i += nt) one(i);
                });
            for (auto& x : th) x.join();
        }
    } catch (...) {
        return MPX_ERR_HOST;
    }
    return MPX_OK;
}
