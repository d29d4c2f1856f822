// Jump-ahead for the MT19937 generator (host side): the polynomials that let the device start many
// independent pieces of ONE NumPy RandomState stream at once.
//
// MT19937's state transition is linear over GF(2) with a primitive characteristic polynomial phi of
// degree 19937, so the state J words ahead is  sum_i g_i * (state i words ahead)  with
// g(x) = x^J mod phi(x)  (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer: "Efficient jump ahead
// for F2-linear random number generators", 2008).  In terms of the raw word sequence x[t] of the
// generator, word j of the jumped state is  XOR_{i : g_i = 1} x[i + j].
//
// phi is recovered once per process with Berlekamp-Massey from 2*19937 bits of the generator's own
// output (no table of constants to get wrong), then g_b = x^(624 * 2^b) mod phi is built by repeated
// squaring and cached.  The only strides the device path uses are whole 624-word blocks times a power
// of two, so one small table serves every problem size.
#pragma once
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

namespace mtjump {

constexpr int DEG = 19937;
constexpr int NW = (DEG + 63) / 64 + 1;   // 64-bit words for coefficients 0 .. DEG (and a little slack)

// the raw (untempered) MT19937 word recurrence on a circular buffer
struct RawMt {
    uint32_t s[624];
    int i = 0;
    void seed(uint32_t v) {
        s[0] = v;
        for (int j = 1; j < 624; ++j) s[j] = 1812433253u * (s[j - 1] ^ (s[j - 1] >> 30)) + (uint32_t)j;
        i = 0;
    }
    uint32_t next() {       // x[t + 624] from x[t], x[t + 1], x[t + 397]
        const uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
        const uint32_t v = s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        s[i] = v;
        i = (i + 1) % 624;
        return v;
    }
};

using Poly = std::vector<uint64_t>;   // bit i = coefficient of x^i

inline void xor_shifted(Poly &dst, const Poly &src, int shift) {     // dst ^= src << shift
    const int ws = shift >> 6, bs = shift & 63;
    const int n = (int)dst.size();
    for (int w = n - 1 - ws; w >= 0; --w) {
        uint64_t v = src[w] << bs;
        if (bs && w > 0) v |= src[w - 1] >> (64 - bs);
        dst[w + ws] ^= v;
    }
}

// Berlekamp-Massey over GF(2) on the low bit of the raw word stream; returns phi (degree DEG) or an
// empty polynomial if the linear complexity found is not 19937 (cannot happen for MT19937)
inline Poly characteristic_polynomial() {
    const int N = 2 * DEG + 64;
    RawMt g;
    g.seed(5489u);
    // the sequence is stored reversed (r_j = s_{N-1-j}) so that s_n, s_{n-1}, ... is a contiguous,
    // ascending bit window and the discrepancy is an AND + parity over words
    std::vector<uint64_t> r((N + 63) / 64 + 8, 0);
    for (int n = 0; n < N; ++n) {
        const int j = N - 1 - n;
        if (g.next() & 1u) r[j >> 6] |= 1ull << (j & 63);
    }
    const int W = NW + 8;
    Poly C(W, 0), B(W, 0), T;
    C[0] = 1; B[0] = 1;
    int L = 0, m = 1;
    for (int n = 0; n < N; ++n) {
        const int off = N - 1 - n, wo = off >> 6, sh = off & 63;
        uint64_t acc = 0;
        for (int w = 0; w <= L / 64; ++w) {
            uint64_t rv = r[wo + w] >> sh;
            if (sh) rv |= r[wo + w + 1] << (64 - sh);
            acc ^= C[w] & rv;
        }
        if (!__builtin_parityll(acc)) { ++m; continue; }
        if (2 * L <= n) {
            T = C;
            xor_shifted(C, B, m);
            L = n + 1 - L;
            B.swap(T);
            m = 1;
        } else {
            xor_shifted(C, B, m);
            ++m;
        }
    }
    if (L != DEG) return Poly();
    Poly phi(NW, 0);                     // phi_j = C_{L-j}: the reciprocal of the connection polynomial
    for (int j = 0; j <= L; ++j) {
        const int i = L - j;
        if ((C[i >> 6] >> (i & 63)) & 1ull) phi[j >> 6] |= 1ull << (j & 63);
    }
    return phi;
}

inline uint64_t spread_bits(uint32_t v) {      // bit i -> bit 2i
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

inline Poly square_mod(const Poly &a, const Poly &phi) {
    Poly sq(2 * NW + 2, 0);
    for (int w = 0; w < NW; ++w) {
        sq[2 * w] = spread_bits((uint32_t)a[w]);
        sq[2 * w + 1] = spread_bits((uint32_t)(a[w] >> 32));
    }
    Poly ph(sq.size(), 0);
    std::memcpy(ph.data(), phi.data(), sizeof(uint64_t) * NW);
    for (int i = 2 * (DEG - 1); i >= DEG; --i)
        if ((sq[i >> 6] >> (i & 63)) & 1ull) {
            // sq ^= phi << (i - DEG), touching only the words phi reaches
            const int shift = i - DEG, ws = shift >> 6, bs = shift & 63;
            for (int w = NW - 1; w >= 0; --w) {
                uint64_t v = ph[w] << bs;
                if (bs && w > 0) v |= ph[w - 1] >> (64 - bs);
                sq[w + ws] ^= v;
            }
            if (bs) sq[NW + ws] ^= ph[NW - 1] >> (64 - bs);
        }
    sq.resize(NW);
    return sq;
}

// g_b = x^(624 * 2^b) mod phi as 624 32-bit words (bit i of the array = coefficient of x^i);
// returns false if the characteristic polynomial could not be established
inline bool block_jump_polynomial(int b, uint32_t out[624]) {
    static std::mutex mu;
    static Poly phi;
    static std::vector<Poly> table;
    static bool failed = false;
    std::lock_guard<std::mutex> lock(mu);
    if (failed) return false;
    if (phi.empty()) {
        phi = characteristic_polynomial();
        if (phi.empty()) { failed = true; return false; }
        Poly g0(NW, 0);
        g0[624 >> 6] |= 1ull << (624 & 63);
        table.push_back(g0);
    }
    while ((int)table.size() <= b) table.push_back(square_mod(table.back(), phi));
    const Poly &g = table[b];
    for (int w = 0; w < 624; ++w) out[w] = (uint32_t)(g[w >> 1] >> (32 * (w & 1)));
    return true;
}

// host application of a jump polynomial to a 624-word state block (tests and tiny cases)
inline void apply_jump(uint32_t key[624], const uint32_t g[624]) {
    std::vector<uint32_t> x(DEG + 624 + 624);
    RawMt mt;
    std::memcpy(mt.s, key, sizeof(mt.s));
    mt.i = 0;
    std::memcpy(x.data(), key, sizeof(mt.s));
    for (size_t t = 624; t < x.size(); ++t) x[t] = mt.next();
    uint32_t acc[624] = {0};
    for (int i = 0; i < DEG; ++i)
        if ((g[i >> 5] >> (i & 31)) & 1u)
            for (int j = 0; j < 624; ++j) acc[j] ^= x[i + j];
    std::memcpy(key, acc, sizeof(acc));
}

}  // namespace mtjump
