// glibc_jump.hpp -- host-side arithmetic for jumping around glibc's rand() stream.
//
// The reference's CPU walker (csrc/cpu/rw_cpu.cpp:20,60,118,169) draws from
// glibc's TYPE_3 generator: r[k] = r[k-3] + r[k-31] (mod 2^32), output r[k]>>1,
// seeded by srand() through a 16807 Lehmer fill and 310 discarded outputs.  The
// recurrence is linear over Z/2^32 with characteristic polynomial
// x^31 = x^28 + 1, so "advance by J draws" is multiplication by x^J modulo that
// polynomial: 31 coefficients per jump.  The device holds x^(d*256^k) for
// k < 6, d < 256 (6 radix-256 digits = 2^48 draws) and composes a jump to any
// stream position from at most six 31x31 multiply-accumulate passes.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

namespace trw {

constexpr int kGlDeg = 31;      // generator order
constexpr int kGlSep = 3;       // short lag
constexpr int kJumpDigits = 6;  // radix-256 digits of a stream position
constexpr int kJumpRadix = 256;
constexpr int kJumpRow = 32;    // coefficients per table row (31 + 1 pad, 128 B)

struct Poly {
    uint32_t c[kGlDeg];
};

inline Poly poly_mulmod(const Poly &a, const Poly &b)
{
    uint32_t t[2 * kGlDeg - 1] = {0};
    for (int i = 0; i < kGlDeg; ++i)
        for (int j = 0; j < kGlDeg; ++j) t[i + j] += a.c[i] * b.c[j];
    // x^d = x^(d-3) + x^(d-31) for d >= 31
    for (int d = 2 * kGlDeg - 2; d >= kGlDeg; --d) {
        t[d - kGlSep] += t[d];
        t[d - kGlDeg] += t[d];
    }
    Poly r;
    std::memcpy(r.c, t, sizeof r.c);
    return r;
}

// tab[(k*256 + d)*32 + i] = coefficient i of x^(d * 256^k)
inline std::vector<uint32_t> build_jump_table()
{
    std::vector<uint32_t> tab((size_t)kJumpDigits * kJumpRadix * kJumpRow, 0u);
    Poly base{};  // x^(256^k)
    base.c[1] = 1;
    for (int k = 0; k < kJumpDigits; ++k) {
        Poly cur{};
        cur.c[0] = 1;
        for (int d = 0; d < kJumpRadix; ++d) {
            std::memcpy(&tab[((size_t)k * kJumpRadix + d) * kJumpRow], cur.c, sizeof cur.c);
            cur = poly_mulmod(cur, base);
        }
        base = cur;  // x^(256^(k+1))
    }
    return tab;
}

// The 31 generator words in front of draw 0 after srand(seed): r[313..343].
inline void seed_window(int32_t seed, uint32_t win[kGlDeg])
{
    uint32_t r[344];
    uint32_t s = (uint32_t)seed;
    if (s == 0) s = 1;
    r[0] = s;
    int32_t word = (int32_t)s;  // signed: the Lehmer step uses C's truncating division
    for (int i = 1; i < kGlDeg; ++i) {
        const int64_t hi = word / 127773;
        const int64_t lo = word % 127773;
        int64_t nw = 16807 * lo - 2836 * hi;
        if (nw < 0) nw += 2147483647;
        word = (int32_t)nw;
        r[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; ++i) r[i] = r[i - 31];
    for (int i = 34; i < 344; ++i) r[i] = r[i - 31] + r[i - 3];
    for (int j = 0; j < kGlDeg; ++j) win[j] = r[313 + j];
}

// win <- (x^J) applied to win, g = coefficients of x^J.
inline void window_apply(const uint32_t *g, uint32_t win[kGlDeg])
{
    uint32_t ext[2 * kGlDeg - 1];
    std::memcpy(ext, win, kGlDeg * sizeof(uint32_t));
    for (int t = 0; t < kGlDeg - 1; ++t) ext[kGlDeg + t] = ext[t] + ext[t + kGlDeg - kGlSep];
    for (int m = 0; m < kGlDeg; ++m) {
        uint32_t acc = 0;
        for (int i = 0; i < kGlDeg; ++i) acc += g[i] * ext[i + m];
        win[m] = acc;
    }
}

}  // namespace trw
