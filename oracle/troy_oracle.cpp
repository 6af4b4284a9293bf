// troy_oracle.cpp -- CPU parity oracle.  TEST INFRASTRUCTURE ONLY (see troy_oracle.h).
//
// A scalar restatement, written from the algorithm description in SURVEY.md Appendix A, of the
// reference's CPU path for the hot path (NTT/INTT, dyadic ops, BEHZ base conversion, key switching,
// multiply / relinearize / rotate / rescale).  Every function cites the reference file:line whose
// results it must reproduce bit-for-bit.  Parity: PINNED (tests/test_oracle_*.py check it against
// the reference's own unit-test vectors and against outputs of the reference compiled in
// oracle/_ref).  Nothing here is used by the product path.

#include "troy_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

typedef uint64_t u64;
typedef unsigned __int128 u128;

namespace {

// ------------------------------------------------------------------------------------------------
// Scalar modular arithmetic (reference: src/modulus.cpp:27-37, src/utils/uintarithsmallmod.h)
// ------------------------------------------------------------------------------------------------
struct Mod {
    u64 p = 0, cr0 = 0, cr1 = 0; // const_ratio = floor(2^128 / p) as (lo, hi)  (modulus.cpp:27-37)
    int bits = 0;
    Mod() {}
    explicit Mod(u64 v) : p(v) {
        if (v == 0) return;
        // floor(2^128 / p): long division of 2^128 by p
        u128 hi_part = ((u128)1 << 64) / p;             // floor(2^64 / p) (fits 64 bits since p >= 2)
        u128 rem = ((u128)1 << 64) - hi_part * p;       // 2^64 mod p
        u128 lo_part = (rem << 64) / p;                 // floor(rem * 2^64 / p)
        cr1 = (u64)hi_part;
        cr0 = (u64)lo_part;
        bits = 64 - __builtin_clzll(v);
    }
};

inline u64 mulhi(u64 a, u64 b) { return (u64)(((u128)a * b) >> 64); }

// uintarithsmallmod.h:128-141
inline u64 barrett64(u64 x, const Mod &m) {
    u64 q = mulhi(x, m.cr1);
    u64 r = x - q * m.p;
    return r >= m.p ? r - m.p : r;
}
// uintarithsmallmod.h:95-121
inline u64 barrett128(u128 z, const Mod &m) {
    u64 lo = (u64)z, hi = (u64)(z >> 64);
    // floor(z * const_ratio / 2^128), dropping the lowest partial product's low half
    u64 carry = mulhi(lo, m.cr0);
    u128 t2 = (u128)lo * m.cr1;
    u64 tmp1 = (u64)t2 + carry;
    u64 tmp3 = (u64)(t2 >> 64) + (tmp1 < carry);
    u128 t4 = (u128)hi * m.cr0;
    u64 tmp1b = tmp1 + (u64)t4;
    u64 c2 = (u64)(t4 >> 64) + (tmp1b < tmp1);
    u64 q = hi * m.cr1 + tmp3 + c2;
    u64 r = lo - q * m.p;
    return r >= m.p ? r - m.p : r;
}
inline u64 mulmod(u64 a, u64 b, const Mod &m) { return barrett128((u128)a * b, m); } // :349-355
inline u64 addmod(u64 a, u64 b, const Mod &m) { u64 s = a + b; return s >= m.p ? s - m.p : s; }
inline u64 submod(u64 a, u64 b, const Mod &m) { return a >= b ? a - b : a + m.p - b; }
inline u64 negmod(u64 a, const Mod &m) { return a ? m.p - a : 0; }

// MultiplyUIntModOperand (uintarithsmallmod.h:166-186): quotient = floor(operand * 2^64 / p)
struct Shoup {
    u64 op = 0, quo = 0;
    Shoup() {}
    Shoup(u64 w, const Mod &m) : op(w), quo((u64)((((u128)w) << 64) / m.p)) {}
};
// :217-230  result in [0, 2p)
inline u64 mul_lazy(u64 x, const Shoup &w, u64 p) { return w.op * x - mulhi(x, w.quo) * p; }
// :196-210
inline u64 mul_shoup(u64 x, const Shoup &w, u64 p) { u64 r = mul_lazy(x, w, p); return r >= p ? r - p : r; }

u64 powmod(u64 a, u64 e, const Mod &m) { // uintarithsmallmod.cpp exponentiateUintMod
    u64 r = 1 % m.p;
    a = barrett64(a, m);
    while (e) {
        if (e & 1) r = mulmod(r, a, m);
        a = mulmod(a, a, m);
        e >>= 1;
    }
    return r;
}
// tryInvertUintMod (uintarithsmallmod.h / numth xgcd): inverse exists iff gcd == 1
bool invmod(u64 a, u64 p, u64 &out) {
    if (a == 0) return false;
    __int128 r0 = p, r1 = a % p, s0 = 0, s1 = 1;
    if (r1 == 0) return false;
    while (r1) {
        __int128 q = r0 / r1, t = r0 - q * r1;
        r0 = r1; r1 = t;
        t = s0 - q * s1; s0 = s1; s1 = t;
    }
    if (r0 != 1) return false;
    if (s0 < 0) s0 += p;
    out = (u64)s0;
    return true;
}
u64 invmod_or_throw(u64 a, const Mod &m) {
    u64 r;
    if (!invmod(barrett64(a, m), m.p, r)) throw std::logic_error("invalid rns bases");
    return r;
}
// dotProductMod (uintarithsmallmod.cpp): canonical sum of products
u64 dotmod(const u64 *a, const u64 *b, size_t n, const Mod &m) {
    u128 acc = 0;
    u64 r = 0;
    for (size_t i = 0; i < n; i++) {
        acc += (u128)a[i] * b[i];
        if ((i & 7) == 7) { r = addmod(r, barrett128(acc, m), m); acc = 0; } // keep acc < 2^128 for 61-bit terms
    }
    return addmod(r, barrett128(acc, m), m);
}

// ------------------------------------------------------------------------------------------------
// Number theory (reference: src/utils/numth.cpp:162-363, src/modulus.cpp:80-121)
// ------------------------------------------------------------------------------------------------
bool is_prime(u64 v) { // numth.cpp:162-247 -- any correct primality test gives the same answers
    if (v < 2) return false;
    for (u64 s : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        if (v == s) return true;
        if (v % s == 0) return false;
    }
    Mod m(v);
    u64 d = v - 1;
    int r = 0;
    while (!(d & 1)) { d >>= 1; r++; }
    // deterministic Miller-Rabin for 64-bit integers
    for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        u64 x = powmod(a, d, m);
        if (x == 1 || x == v - 1) continue;
        bool comp = true;
        for (int i = 1; i < r; i++) {
            x = mulmod(x, x, m);
            if (x == v - 1) { comp = false; break; }
        }
        if (comp) return false;
    }
    return true;
}
// numth.cpp:261-285
std::vector<u64> get_primes(u64 factor, int bits, size_t count) {
    std::vector<u64> out;
    u64 value = ((u64(1) << bits) - 1) / factor * factor + 1;
    u64 lower = u64(1) << (bits - 1);
    while (count > 0 && value > lower) {
        if (is_prime(value)) { out.push_back(value); count--; }
        value -= factor;
    }
    if (count > 0) throw std::logic_error("failed to find enough qualifying primes");
    return out;
}
// modulus.cpp:80-121: per distinct size the `count` largest primes, handed out from the back
std::vector<u64> coeff_modulus_create(u64 N, const std::vector<int> &bits) {
    std::map<int, size_t> cnt;
    for (int b : bits) cnt[b]++;
    std::map<int, std::vector<u64>> table;
    for (auto &kv : cnt) table[kv.first] = get_primes(2 * N, kv.first, kv.second);
    std::vector<u64> out;
    for (int b : bits) { out.push_back(table[b].back()); table[b].pop_back(); }
    return out;
}
// numth.cpp:335-363: minimal primitive `degree`-th root (degree a power of two).  The reference
// finds a random primitive root then scans all odd powers for the minimum, so the result is the
// minimum over ALL primitive degree-th roots, independent of the starting root.
bool min_primitive_root(u64 degree, const Mod &m, u64 &out) {
    u64 group = m.p - 1;
    if (group % degree) return false;
    u64 quot = group / degree, root = 0;
    for (u64 g = 2; g < 1000; g++) { // deterministic search for any primitive degree-th root
        u64 c = powmod(g, quot, m);
        if (powmod(c, degree >> 1, m) == m.p - 1) { root = c; break; }
    }
    if (!root) return false;
    u64 sq = mulmod(root, root, m), cur = root;
    for (u64 i = 0; i < degree; i += 2) {
        if (cur < root) root = cur;
        cur = mulmod(cur, sq, m);
    }
    out = root;
    return true;
}
inline uint32_t bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}
// numth.h:16-36
std::vector<int> naf(int value) {
    std::vector<int> res;
    bool sign = value < 0;
    value = std::abs(value);
    for (int i = 0; value; i++) {
        int zi = (value & 1) ? 2 - (value & 3) : 0;
        value = (value - zi) >> 1;
        if (zi) res.push_back((sign ? -zi : zi) * (1 << i));
    }
    return res;
}

// ------------------------------------------------------------------------------------------------
// NTT tables and transforms (reference: src/utils/ntt.cpp:17-66,158-213; dwthandler.h:88-372; ntt.h:17-64)
// ------------------------------------------------------------------------------------------------
struct NttTable {
    Mod m;
    int logn = 0;
    size_t n = 0;
    u64 psi = 0;
    std::vector<Shoup> root, iroot;
    Shoup inv_n;
    NttTable() {}
    NttTable(int logn_, u64 p) : m(p), logn(logn_), n(size_t(1) << logn_) {
        if (!min_primitive_root(2 * n, m, psi)) throw std::invalid_argument("invalid modulus");
        u64 ipsi = invmod_or_throw(psi, m);
        root.resize(n);
        iroot.resize(n);
        u64 pw = psi;
        for (size_t i = 1; i < n; i++) { // ntt.cpp:38-43
            root[bitrev((uint32_t)i, logn)] = Shoup(pw, m);
            pw = mulmod(pw, psi, m);
        }
        root[0] = Shoup(1, m);
        pw = ipsi;
        for (size_t i = 1; i < n; i++) { // ntt.cpp:49-54
            iroot[bitrev((uint32_t)(i - 1), logn) + 1] = Shoup(pw, m);
            pw = mulmod(pw, ipsi, m);
        }
        iroot[0] = Shoup(1, m);
        inv_n = Shoup(invmod_or_throw(n, m), m);
    }
};

// dwthandler.h:88-204 (transformToRev): natural -> bit-reversed, inputs < 4p, outputs < 4p
void ntt_fwd_lazy(u64 *a, const NttTable &t) {
    const u64 p = t.m.p, two_p = 2 * p;
    size_t n = t.n, gap = n >> 1, m = 1, ridx = 0;
    for (; m < n; m <<= 1, gap >>= 1) {
        size_t off = 0;
        for (size_t i = 0; i < m; i++) {
            const Shoup &w = t.root[++ridx];
            u64 *x = a + off, *y = x + gap;
            for (size_t j = 0; j < gap; j++) {
                u64 u = x[j] >= two_p ? x[j] - two_p : x[j]; // ntt.h:55-58 guard (>=)
                u64 v = mul_lazy(y[j], w, p);
                x[j] = u + v;
                y[j] = u + two_p - v;
            }
            off += gap << 1;
        }
    }
}
void ntt_fwd(u64 *a, const NttTable &t) { // ntt.cpp:164-187
    ntt_fwd_lazy(a, t);
    const u64 p = t.m.p, two_p = 2 * p;
    for (size_t i = 0; i < t.n; i++) {
        u64 v = a[i];
        if (v >= two_p) v -= two_p;
        if (v >= p) v -= p;
        a[i] = v;
    }
}
// dwthandler.h:215-372 (transformFromRev) with n^{-1} folded into the last stage: bit-reversed ->
// natural, inputs < 2p, outputs < 2p
void ntt_inv_lazy(u64 *a, const NttTable &t) {
    const u64 p = t.m.p, two_p = 2 * p;
    size_t n = t.n, gap = 1, m = n >> 1, ridx = 0;
    auto guard = [&](u64 v) { return v >= two_p ? v - two_p : v; };
    for (; m > 1; m >>= 1, gap <<= 1) {
        size_t off = 0;
        for (size_t i = 0; i < m; i++) {
            const Shoup &w = t.iroot[++ridx];
            u64 *x = a + off, *y = x + gap;
            for (size_t j = 0; j < gap; j++) {
                u64 u = x[j], v = y[j];
                x[j] = guard(u + v);
                y[j] = mul_lazy(u + two_p - v, w, p);
            }
            off += gap << 1;
        }
    }
    // last stage, scalar n^{-1} merged (dwthandler.h:289-330): scaled_r = r * n^{-1}
    const Shoup &w = t.iroot[++ridx];
    Shoup scaled_w(mul_shoup(w.op, t.inv_n, p), t.m);
    u64 *x = a, *y = a + gap;
    for (size_t j = 0; j < gap; j++) {
        u64 u = guard(x[j]), v = y[j];
        x[j] = mul_lazy(guard(u + v), t.inv_n, p);
        y[j] = mul_lazy(u + two_p - v, scaled_w, p);
    }
}
void ntt_inv(u64 *a, const NttTable &t) { // ntt.cpp:196-213
    ntt_inv_lazy(a, t);
    const u64 p = t.m.p;
    for (size_t i = 0; i < t.n; i++) if (a[i] >= p) a[i] -= p;
}

// ------------------------------------------------------------------------------------------------
// Galois (reference: src/utils/galois.cpp:18-35, 44-86, 143-177)
// ------------------------------------------------------------------------------------------------
void apply_galois(size_t N, int logn, uint32_t elt, const Mod &m, const u64 *in, u64 *out) {
    u64 idx_raw = 0;
    for (size_t i = 0; i < N; i++, idx_raw += elt) {
        u64 idx = idx_raw & (N - 1);
        u64 v = in[i];
        if ((idx_raw >> logn) & 1) v = v ? m.p - v : 0;
        out[idx] = v;
    }
}
void apply_galois_ntt(size_t N, int logn, uint32_t elt, const u64 *in, u64 *out) {
    for (size_t i = 0; i < N; i++) {
        uint32_t rev = bitrev((uint32_t)(i + N), logn + 1);
        u64 raw = ((u64)elt * rev) >> 1;
        raw &= (N - 1);
        out[i] = in[bitrev((uint32_t)raw, logn)];
    }
}
uint32_t elt_from_step(size_t N, int step) { // galois.cpp:44-86
    uint32_t n = (uint32_t)N, m32 = 2 * n;
    if (step == 0) return m32 - 1;
    bool sign = step < 0;
    uint32_t pos = (uint32_t)std::abs(step);
    if (pos >= (n >> 1)) throw std::invalid_argument("step count too large");
    pos &= m32 - 1;
    int s = sign ? (int)(n >> 1) - (int)pos : (int)pos;
    u64 g = 1;
    while (s--) { g *= 3; g &= (u64)m32 - 1; }
    return (uint32_t)g;
}

// ------------------------------------------------------------------------------------------------
// RNS bases and the BEHZ tool (reference: src/utils/rns.cpp:415-459, 556-1146)
// ------------------------------------------------------------------------------------------------
struct BaseConv { // rns.cpp:556-572: matrix[o][i] = (prod_{k != i} ibase_k) mod obase_o
    std::vector<Mod> ib, ob;
    std::vector<Shoup> inv_punct; // (prod_{k != i} ibase_k)^{-1} mod ibase_i
    std::vector<std::vector<u64>> mat;
    BaseConv() {}
    BaseConv(const std::vector<Mod> &i_, const std::vector<Mod> &o_) : ib(i_), ob(o_) {
        size_t ni = ib.size(), no = ob.size();
        inv_punct.resize(ni);
        for (size_t i = 0; i < ni; i++) {
            u64 pr = 1 % ib[i].p;
            for (size_t k = 0; k < ni; k++) if (k != i) pr = mulmod(pr, barrett64(ib[k].p, ib[i]), ib[i]);
            inv_punct[i] = Shoup(invmod_or_throw(pr, ib[i]), ib[i]);
        }
        mat.assign(no, std::vector<u64>(ni));
        for (size_t o = 0; o < no; o++)
            for (size_t i = 0; i < ni; i++) {
                u64 pr = 1 % ob[o].p;
                for (size_t k = 0; k < ni; k++) if (k != i) pr = mulmod(pr, barrett64(ib[k].p, ob[o]), ob[o]);
                mat[o][i] = pr;
            }
    }
    // rns.cpp:415-459 fastConvertArray: in [ni][count] -> out [no][count]
    void fast_convert(const u64 *in, u64 *out, size_t count) const {
        size_t ni = ib.size(), no = ob.size();
        std::vector<u64> temp(count * ni);
        for (size_t i = 0; i < ni; i++) {
            if (inv_punct[i].op == 1)
                for (size_t j = 0; j < count; j++) temp[j * ni + i] = barrett64(in[i * count + j], ib[i]);
            else
                for (size_t j = 0; j < count; j++) temp[j * ni + i] = mul_shoup(in[i * count + j], inv_punct[i], ib[i].p);
        }
        for (size_t o = 0; o < no; o++)
            for (size_t j = 0; j < count; j++) out[o * count + j] = dotmod(&temp[j * ni], mat[o].data(), ni, ob[o]);
    }
    // rns.cpp:462-548 exactConvertArray (one output modulus; double-precision rounding term v)
    void exact_convert(const u64 *in, u64 *out, size_t count) const {
        size_t ni = ib.size();
        const Mod &p = ob[0];
        std::vector<u64> temp(count * ni);
        std::vector<double> v(count * ni);
        for (size_t i = 0; i < ni; i++) {
            double divisor = (double)ib[i].p;
            for (size_t j = 0; j < count; j++) {
                u64 x = inv_punct[i].op == 1 ? barrett64(in[i * count + j], ib[i]) : mul_shoup(in[i * count + j], inv_punct[i], ib[i].p);
                temp[j * ni + i] = x;
                v[j * ni + i] = (double)x / divisor;
            }
        }
        u64 q_mod_p = 1 % p.p;
        for (size_t k = 0; k < ni; k++) q_mod_p = mulmod(q_mod_p, barrett64(ib[k].p, p), p);
        for (size_t j = 0; j < count; j++) {
            double agg = 0.0;
            for (size_t i = 0; i < ni; i++) agg += v[j * ni + i];
            agg += 0.5;
            u64 rounded = (u64)agg;
            u64 sum = dotmod(&temp[j * ni], mat[0].data(), ni, p);
            out[j] = submod(sum, mulmod(rounded, q_mod_p, p), p);
        }
    }
};

// significant bit count of prod(values) via schoolbook multi-word multiply (uintarith multiplyManyUint64)
int product_bit_count(const std::vector<Mod> &v) {
    std::vector<u64> acc{1};
    for (auto &m : v) {
        u64 carry = 0;
        for (auto &w : acc) {
            u128 t = (u128)w * m.p + carry;
            w = (u64)t;
            carry = (u64)(t >> 64);
        }
        if (carry) acc.push_back(carry);
    }
    while (acc.size() > 1 && acc.back() == 0) acc.pop_back();
    return (int)(64 * (acc.size() - 1) + (64 - __builtin_clzll(acc.back())));
}
u64 prod_mod(const std::vector<Mod> &v, const Mod &m) {
    u64 r = 1 % m.p;
    for (auto &x : v) r = mulmod(r, barrett64(x.p, m), m);
    return r;
}

struct Orc;

struct RnsTool { // rns.cpp:581-803
    size_t N = 0;
    Mod t, m_tilde, m_sk, gamma;
    std::vector<Mod> q, B, Bsk, Bsk_mt;
    std::vector<const NttTable *> bsk_tables;
    BaseConv q_to_Bsk, q_to_mt, B_to_q, B_to_msk, q_to_tg, q_to_t;
    std::vector<u64> prod_B_mod_q, prod_q_mod_Bsk;
    std::vector<Shoup> inv_prod_q_mod_Bsk, inv_mt_mod_Bsk, prod_tg_mod_q, neg_inv_q_mod_tg, inv_q_last_mod_q;
    Shoup inv_prod_B_mod_msk, neg_inv_prod_q_mod_mt, inv_gamma_mod_t;
    u64 inv_q_last_mod_t = 1, q_last_mod_t = 0;

    void init(Orc *o, const std::vector<Mod> &q_, const Mod &t_);

    // rns.cpp:1012-1037
    void fastbconv_mtilde(const u64 *in, u64 *out) const {
        size_t nq = q.size(), nb = Bsk.size();
        std::vector<u64> temp(nq * N);
        for (size_t i = 0; i < nq; i++) {
            Shoup s(barrett64(m_tilde.p, q[i]), q[i]);
            for (size_t j = 0; j < N; j++) temp[i * N + j] = mul_shoup(in[i * N + j], s, q[i].p);
        }
        q_to_Bsk.fast_convert(temp.data(), out, N);
        q_to_mt.fast_convert(temp.data(), out + nb * N, N);
    }
    // rns.cpp:943-983
    void sm_mrq(const u64 *in, u64 *out) const {
        size_t nb = Bsk.size();
        const u64 *in_mt = in + nb * N;
        const u64 mt_half = m_tilde.p >> 1;
        std::vector<u64> r(N);
        for (size_t j = 0; j < N; j++) r[j] = mul_shoup(in_mt[j], neg_inv_prod_q_mod_mt, m_tilde.p);
        for (size_t i = 0; i < nb; i++) {
            const Mod &b = Bsk[i];
            for (size_t j = 0; j < N; j++) {
                u64 temp = r[j];
                if (temp >= mt_half) temp += b.p - m_tilde.p;
                u64 s = barrett128((u128)temp * prod_q_mod_Bsk[i] + in[i * N + j], b);
                out[i * N + j] = mul_shoup(s, inv_mt_mod_Bsk[i], b.p);
            }
        }
    }
    // rns.cpp:985-1010
    void fast_floor(const u64 *in, u64 *out) const {
        size_t nq = q.size(), nb = Bsk.size();
        q_to_Bsk.fast_convert(in, out, N);
        const u64 *in_b = in + nq * N;
        for (size_t i = 0; i < nb; i++)
            for (size_t j = 0; j < N; j++)
                out[i * N + j] = mul_shoup(in_b[i * N + j] + (Bsk[i].p - out[i * N + j]), inv_prod_q_mod_Bsk[i], Bsk[i].p);
    }
    // rns.cpp:879-941
    void fastbconv_sk(const u64 *in, u64 *out) const {
        size_t nq = q.size(), nB = B.size();
        B_to_q.fast_convert(in, out, N);
        std::vector<u64> temp(N), alpha(N);
        B_to_msk.fast_convert(in, temp.data(), N);
        for (size_t j = 0; j < N; j++)
            alpha[j] = mul_shoup(temp[j] + (m_sk.p - in[nB * N + j]), inv_prod_B_mod_msk, m_sk.p);
        const u64 msk_half = m_sk.p >> 1;
        for (size_t i = 0; i < nq; i++) {
            const Mod &b = q[i];
            for (size_t j = 0; j < N; j++) {
                u64 &d = out[i * N + j];
                if (alpha[j] > msk_half)
                    d = barrett128((u128)negmod(alpha[j], m_sk) * prod_B_mod_q[i] + d, b);
                else
                    d = barrett128((u128)alpha[j] * (b.p - prod_B_mod_q[i]) + d, b);
            }
        }
    }
    // rns.cpp:805-830
    void divide_round_qlast(u64 *x) const {
        size_t nq = q.size();
        const Mod &last = q[nq - 1];
        u64 *xl = x + (nq - 1) * N;
        u64 half = last.p >> 1;
        for (size_t j = 0; j < N; j++) xl[j] = addmod(xl[j], barrett64(half, last), last);
        for (size_t i = 0; i + 1 < nq; i++) {
            const Mod &b = q[i];
            u64 half_mod = barrett64(half, b);
            for (size_t j = 0; j < N; j++) {
                u64 tmp = submod(barrett64(xl[j], b), half_mod, b);
                x[i * N + j] = mul_shoup(submod(x[i * N + j], tmp, b), inv_q_last_mod_q[i], b.p);
            }
        }
    }
    // rns.cpp:832-877
    void divide_round_qlast_ntt(u64 *x, const std::vector<const NttTable *> &tables) const {
        size_t nq = q.size();
        const Mod &last = q[nq - 1];
        u64 *xl = x + (nq - 1) * N;
        ntt_inv(xl, *tables[nq - 1]);
        u64 half = last.p >> 1;
        for (size_t j = 0; j < N; j++) xl[j] = addmod(xl[j], barrett64(half, last), last);
        std::vector<u64> temp(N);
        for (size_t i = 0; i + 1 < nq; i++) {
            const Mod &b = q[i];
            if (b.p < last.p) for (size_t j = 0; j < N; j++) temp[j] = barrett64(xl[j], b);
            else std::copy(xl, xl + N, temp.begin());
            u64 neg_half_mod = b.p - barrett64(half, b);
            for (size_t j = 0; j < N; j++) temp[j] += neg_half_mod;
            u64 qi_lazy = b.p << 2;
            ntt_fwd_lazy(temp.data(), *tables[i]);
            for (size_t j = 0; j < N; j++)
                x[i * N + j] = mul_shoup(x[i * N + j] + qi_lazy - temp[j], inv_q_last_mod_q[i], b.p);
        }
    }
    // rns.cpp:1097-1140
    void modt_divide_qlast(u64 *x) const {
        size_t nq = q.size();
        u64 *xl = x + (nq - 1) * N;
        u64 last_value = q[nq - 1].p;
        std::vector<u64> neg_c(N);
        for (size_t j = 0; j < N; j++) {
            u64 v = negmod(barrett64(xl[j], t), t);
            if (inv_q_last_mod_t != 1) v = mulmod(v, inv_q_last_mod_t, t);
            neg_c[j] = v;
        }
        for (size_t i = 0; i + 1 < nq; i++) {
            const Mod &b = q[i];
            u64 lv = barrett64(last_value, b);
            for (size_t j = 0; j < N; j++) {
                u64 delta = mulmod(barrett64(neg_c[j], b), lv, b);
                u64 v = x[i * N + j] + 2 * b.p - barrett64(xl[j], b) - delta;
                x[i * N + j] = mul_shoup(v, inv_q_last_mod_q[i], b.p);
            }
        }
    }
    // rns.cpp:1039-1095
    void decrypt_scale_and_round(const u64 *in, u64 *out) const {
        size_t nq = q.size();
        std::vector<u64> temp(nq * N), tg(2 * N);
        for (size_t i = 0; i < nq; i++)
            for (size_t j = 0; j < N; j++) temp[i * N + j] = mul_shoup(in[i * N + j], prod_tg_mod_q[i], q[i].p);
        q_to_tg.fast_convert(temp.data(), tg.data(), N);
        const Mod tgm[2] = {t, gamma};
        for (size_t i = 0; i < 2; i++)
            for (size_t j = 0; j < N; j++) tg[i * N + j] = mul_shoup(tg[i * N + j], neg_inv_q_mod_tg[i], tgm[i].p);
        u64 gamma_half = gamma.p >> 1;
        for (size_t j = 0; j < N; j++) {
            u64 d;
            if (tg[N + j] > gamma_half) d = addmod(tg[j], barrett64(gamma.p - tg[N + j], t), t);
            else d = submod(tg[j], barrett64(tg[N + j], t), t);
            if (d != 0) d = mul_shoup(d, inv_gamma_mod_t, t.p);
            out[j] = d;
        }
    }
};

struct Ct {
    std::vector<u64> d;
    int size = 0, limbs = 0;
    bool ntt = false;
    double scale = 1.0;
    u64 cf = 1;
    u64 *poly(size_t i, size_t N) { return d.data() + i * limbs * N; }
    const u64 *poly(size_t i, size_t N) const { return d.data() + i * limbs * N; }
};

struct Orc {
    int scheme = 0, logn = 0;
    size_t N = 0, K = 0;
    Mod t;
    std::vector<Mod> key_q;
    std::map<u64, std::unique_ptr<NttTable>> cache;
    std::vector<const NttTable *> key_tables;
    std::map<int, std::unique_ptr<RnsTool>> tools; // by limb count
    int first_limbs = 0, last_limbs = 0, n_levels = 0;
    std::vector<u64> relin_key;                  // index 0 (power 2)
    std::map<int, std::vector<u64>> relin_keys_hi; // index >= 1 (power index + 2)
    std::map<uint32_t, std::vector<u64>> galois_keys;
    std::string err;

    const NttTable *table(u64 p) {
        auto it = cache.find(p);
        if (it == cache.end()) it = cache.emplace(p, std::make_unique<NttTable>(logn, p)).first;
        return it->second.get();
    }
    const RnsTool &tool(int limbs) {
        auto it = tools.find(limbs);
        if (it == tools.end()) throw std::invalid_argument("no such level");
        return *it->second;
    }
    std::vector<Mod> level_q(int limbs) const { return std::vector<Mod>(key_q.begin(), key_q.begin() + limbs); }
    bool level_exists(int limbs) const { return tools.count(limbs) != 0; }
};

void RnsTool::init(Orc *o, const std::vector<Mod> &q_, const Mod &t_) {
    N = o->N;
    q = q_;
    t = t_;
    size_t nq = q.size();
    int total_bits = product_bit_count(q);
    size_t nB = nq;
    if (32 + t.bits + total_bits >= 61 * (int)nq + 61) nB++; // rns.cpp:610-615
    size_t nBsk = nB + 1;
    auto aux = get_primes(2 * N, 61, nBsk + 1);             // rns.cpp:629-635
    m_sk = Mod(aux[0]);
    gamma = Mod(aux[1]);
    for (size_t i = 0; i < nB; i++) B.emplace_back(aux[2 + i]);
    m_tilde = Mod(u64(1) << 32);
    Bsk = B; Bsk.push_back(m_sk);
    Bsk_mt = Bsk; Bsk_mt.push_back(m_tilde);
    for (auto &m : Bsk) bsk_tables.push_back(o->table(m.p));
    q_to_Bsk = BaseConv(q, Bsk);
    q_to_mt = BaseConv(q, {m_tilde});
    B_to_q = BaseConv(B, q);
    B_to_msk = BaseConv(B, {m_sk});
    if (t.p) { q_to_tg = BaseConv(q, {t, gamma}); q_to_t = BaseConv(q, {t}); }
    prod_B_mod_q.resize(nq);
    for (size_t i = 0; i < nq; i++) prod_B_mod_q[i] = prod_mod(B, q[i]);
    inv_prod_q_mod_Bsk.resize(nBsk); inv_mt_mod_Bsk.resize(nBsk); prod_q_mod_Bsk.resize(nBsk);
    for (size_t i = 0; i < nBsk; i++) {
        u64 pq = prod_mod(q, Bsk[i]);
        prod_q_mod_Bsk[i] = pq;
        inv_prod_q_mod_Bsk[i] = Shoup(invmod_or_throw(pq, Bsk[i]), Bsk[i]);
        inv_mt_mod_Bsk[i] = Shoup(invmod_or_throw(m_tilde.p, Bsk[i]), Bsk[i]);
    }
    inv_prod_B_mod_msk = Shoup(invmod_or_throw(prod_mod(B, m_sk), m_sk), m_sk);
    {
        u64 inv = invmod_or_throw(prod_mod(q, m_tilde), m_tilde);
        neg_inv_prod_q_mod_mt = Shoup(negmod(inv, m_tilde), m_tilde);
    }
    if (t.p) {
        inv_gamma_mod_t = Shoup(invmod_or_throw(gamma.p, t), t);
        prod_tg_mod_q.resize(nq);
        for (size_t i = 0; i < nq; i++) prod_tg_mod_q[i] = Shoup(mulmod(barrett64(t.p, q[i]), barrett64(gamma.p, q[i]), q[i]), q[i]);
        const Mod tgm[2] = {t, gamma};
        neg_inv_q_mod_tg.resize(2);
        for (int i = 0; i < 2; i++) neg_inv_q_mod_tg[i] = Shoup(negmod(invmod_or_throw(prod_mod(q, tgm[i]), tgm[i]), tgm[i]), tgm[i]);
    }
    inv_q_last_mod_q.resize(nq - 1);
    for (size_t i = 0; i + 1 < nq; i++) inv_q_last_mod_q[i] = Shoup(invmod_or_throw(q[nq - 1].p, q[i]), q[i]);
    if (t.p) {
        inv_q_last_mod_t = invmod_or_throw(q[nq - 1].p, t);
        q_last_mod_t = barrett64(q[nq - 1].p, t);
    }
}

// ------------------------------------------------------------------------------------------------
// Evaluator (reference: src/evaluator.cpp)
// ------------------------------------------------------------------------------------------------
struct Eval {
    Orc *o;
    size_t N;
    explicit Eval(Orc *o_) : o(o_), N(o_->N) {}

    void check_level(const Ct &c) const {
        if (!o->level_exists(c.limbs) || (o->K > 1 && (size_t)c.limbs == o->K)) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    }

    // evaluator.cpp:69-128 balanceCorrectionFactors
    std::tuple<u64, u64, u64> balance(u64 f1, u64 f2) const {
        const Mod &tm = o->t;
        u64 t = tm.p, half_t = t / 2;
        auto sum_abs = [&](u64 x, u64 y) {
            int64_t xb = (int64_t)(x > half_t ? x - t : x), yb = (int64_t)(y > half_t ? y - t : y);
            return std::llabs(xb) + std::llabs(yb);
        };
        u64 ratio;
        if (!invmod(f1, t, ratio)) throw std::logic_error("invalid correction factor1");
        ratio = mulmod(ratio, f2, tm);
        u64 e1 = ratio, e2 = 1;
        int64_t sum = sum_abs(e1, e2);
        int64_t prev_a = (int64_t)t, prev_b = 0, a = (int64_t)ratio, b = 1;
        auto gcd = [](u64 x, u64 y) { while (y) { u64 r = x % y; x = y; y = r; } return x; };
        while (a != 0) {
            int64_t qq = prev_a / a, temp = prev_a % a;
            prev_a = a; a = temp;
            temp = prev_b - b * qq; prev_b = b; b = temp;
            u64 a_mod = barrett64((u64)std::llabs(a), tm);
            if (a < 0) a_mod = negmod(a_mod, tm);
            u64 b_mod = barrett64((u64)std::llabs(b), tm);
            if (b < 0) b_mod = negmod(b_mod, tm);
            if (a_mod != 0 && gcd(a_mod, t) == 1) {
                int64_t ns = sum_abs(a_mod, b_mod);
                if (ns < sum) { sum = ns; e1 = a_mod; e2 = b_mod; }
            }
        }
        return std::make_tuple(mulmod(e1, f1, tm), e1, e2);
    }
    void scalar_mul(Ct &c, u64 s) const { // polyarithsmallmod multiplyPolyScalarCoeffmod
        for (int i = 0; i < c.size; i++)
            for (int l = 0; l < c.limbs; l++) {
                const Mod &m = o->key_q[l];
                Shoup w(barrett64(s, m), m);
                u64 *p = c.poly(i, N) + l * N;
                for (size_t j = 0; j < N; j++) p[j] = mul_shoup(p[j], w, m.p);
            }
    }
    static bool same_scale(const Ct &a, const Ct &b) { return std::fabs(a.scale - b.scale) < std::ldexp(1.0, -40) * std::max(std::fabs(a.scale), 1.0) || a.scale == b.scale; }

    // evaluator.cpp:167-235 addInplace / :262-347 subInplace
    void add_sub(Ct &a, const Ct &b_in, bool sub) const {
        check_level(a);
        if (a.limbs != b_in.limbs) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        if (a.ntt != b_in.ntt) throw std::invalid_argument("NTT form mismatch");
        if (!same_scale(a, b_in)) throw std::invalid_argument("scale mismatch");
        const Ct *bp = &b_in;
        Ct b_copy;
        if (a.cf != b_in.cf) {
            auto f = balance(a.cf, b_in.cf);
            scalar_mul(a, std::get<1>(f));
            b_copy = b_in;
            scalar_mul(b_copy, std::get<2>(f));
            a.cf = std::get<0>(f);
            b_copy.cf = std::get<0>(f);
            bp = &b_copy;
        }
        const Ct &b = *bp;
        int mx = std::max(a.size, b.size), mn = std::min(a.size, b.size), a_size = a.size;
        a.d.resize((size_t)mx * a.limbs * N, 0);
        a.size = mx;
        for (int i = 0; i < mn; i++)
            for (int l = 0; l < a.limbs; l++) {
                const Mod &m = o->key_q[l];
                u64 *x = a.poly(i, N) + l * N;
                const u64 *y = b.poly(i, N) + l * N;
                for (size_t j = 0; j < N; j++) x[j] = sub ? submod(x[j], y[j], m) : addmod(x[j], y[j], m);
            }
        if (a_size < b.size)
            for (int i = mn; i < b.size; i++)
                for (int l = 0; l < a.limbs; l++) {
                    const Mod &m = o->key_q[l];
                    u64 *x = a.poly(i, N) + l * N;
                    const u64 *y = b.poly(i, N) + l * N;
                    for (size_t j = 0; j < N; j++) x[j] = sub ? negmod(y[j], m) : y[j];
                }
    }
    void negate(Ct &a) const { // evaluator.cpp:147-165
        check_level(a);
        for (int i = 0; i < a.size; i++)
            for (int l = 0; l < a.limbs; l++) {
                const Mod &m = o->key_q[l];
                u64 *x = a.poly(i, N) + l * N;
                for (size_t j = 0; j < N; j++) x[j] = negmod(x[j], m);
            }
    }

    // dest[i] = sum_{j+k=i} a[j] (.) b[k]  (evaluator.cpp:507-570 BEHZ step 4, :626-702 ckks, :747-780 bgv)
    static void tensor(const u64 *a, int asz, const u64 *b, int bsz, u64 *dest, const std::vector<Mod> &mods, size_t N) {
        size_t nl = mods.size(), d = nl * N;
        int dsz = asz + bsz - 1;
        std::fill(dest, dest + (size_t)dsz * d, 0);
        for (int i = 0; i < dsz; i++) {
            int a_last = std::min(i, asz - 1), b_first = std::min(i, bsz - 1), a_first = i - b_first;
            int steps = a_last - a_first + 1;
            for (int s = 0; s < steps; s++)
                for (size_t l = 0; l < nl; l++) {
                    const u64 *x = a + (size_t)(a_first + s) * d + l * N, *y = b + (size_t)(b_first - s) * d + l * N;
                    u64 *z = dest + (size_t)i * d + l * N;
                    for (size_t j = 0; j < N; j++) z[j] = addmod(z[j], mulmod(x[j], y[j], mods[l]), mods[l]);
                }
        }
    }

    // evaluator.cpp:385-624 bfvMultiply
    void bfv_multiply(Ct &a, const Ct &b) const {
        if (a.ntt || b.ntt) throw std::invalid_argument("encrypted1 or encrypted2 cannot be in NTT form");
        const RnsTool &rt = o->tool(a.limbs);
        size_t nq = a.limbs, nb = rt.Bsk.size();
        int dsz = a.size + b.size - 1;
        auto extend = [&](const Ct &c, std::vector<u64> &cq, std::vector<u64> &cb) {
            cq.assign(c.d.begin(), c.d.end());
            cb.resize((size_t)c.size * nb * N);
            std::vector<u64> temp((nb + 1) * N);
            for (int i = 0; i < c.size; i++) {
                for (size_t l = 0; l < nq; l++) ntt_fwd_lazy(&cq[(i * nq + l) * N], *o->key_tables[l]);
                rt.fastbconv_mtilde(c.poly(i, N), temp.data());
                rt.sm_mrq(temp.data(), &cb[(size_t)i * nb * N]);
                for (size_t l = 0; l < nb; l++) ntt_fwd_lazy(&cb[(i * nb + l) * N], *rt.bsk_tables[l]);
            }
        };
        std::vector<u64> aq, ab, bq, bb;
        extend(a, aq, ab);
        extend(b, bq, bb);
        // the tensor's mulmod (Barrett-128) accepts the lazy [0,4p) NTT outputs and returns canonical values
        std::vector<u64> dq((size_t)dsz * nq * N), db((size_t)dsz * nb * N);
        tensor(aq.data(), a.size, bq.data(), b.size, dq.data(), rt.q, N);
        tensor(ab.data(), a.size, bb.data(), b.size, db.data(), rt.Bsk, N);
        for (int i = 0; i < dsz; i++) {
            for (size_t l = 0; l < nq; l++) ntt_inv_lazy(&dq[(i * nq + l) * N], *o->key_tables[l]);
            for (size_t l = 0; l < nb; l++) ntt_inv_lazy(&db[(i * nb + l) * N], *rt.bsk_tables[l]);
        }
        a.d.assign((size_t)dsz * nq * N, 0);
        a.size = dsz;
        std::vector<u64> tqb((nq + nb) * N), tb(nb * N);
        for (int i = 0; i < dsz; i++) {
            for (size_t l = 0; l < nq; l++) {
                Shoup w(barrett64(o->t.p, rt.q[l]), rt.q[l]);
                for (size_t j = 0; j < N; j++) tqb[l * N + j] = mul_shoup(dq[(i * nq + l) * N + j], w, rt.q[l].p);
            }
            for (size_t l = 0; l < nb; l++) {
                Shoup w(barrett64(o->t.p, rt.Bsk[l]), rt.Bsk[l]);
                for (size_t j = 0; j < N; j++) tqb[(nq + l) * N + j] = mul_shoup(db[(i * nb + l) * N + j], w, rt.Bsk[l].p);
            }
            rt.fast_floor(tqb.data(), tb.data());
            rt.fastbconv_sk(tb.data(), a.poly(i, N));
        }
    }
    // evaluator.cpp:626-702 ckksMultiply
    void ckks_multiply(Ct &a, const Ct &b) const {
        if (!(a.ntt && b.ntt)) throw std::invalid_argument("encrypted1 or encrypted2 must be in NTT form");
        int dsz = a.size + b.size - 1;
        std::vector<u64> dest((size_t)dsz * a.limbs * N);
        tensor(a.d.data(), a.size, b.d.data(), b.size, dest.data(), o->level_q(a.limbs), N);
        a.d.swap(dest);
        a.size = dsz;
        a.scale *= b.scale;
        if (!scale_ok(a.scale, a.limbs)) throw std::invalid_argument("scale out of bounds");
    }
    bool scale_ok(double scale, int limbs) const { // evaluator.cpp:40-66 isScaleWithinBounds
        int bound = product_bit_count(o->level_q(limbs));
        return !(scale <= 0 || ((int)std::log2(scale) >= bound));
    }
    // evaluator.cpp:704-794 bgvMultiply
    void bgv_multiply(Ct &a, const Ct &b) const {
        if (a.ntt || b.ntt) throw std::invalid_argument("encryped1 or encrypted2 must be not in NTT form");
        int dsz = a.size + b.size - 1;
        size_t nq = a.limbs;
        std::vector<u64> x(a.d), y(b.d), dest((size_t)dsz * nq * N);
        for (int i = 0; i < a.size; i++) for (size_t l = 0; l < nq; l++) ntt_fwd(&x[(i * nq + l) * N], *o->key_tables[l]);
        for (int i = 0; i < b.size; i++) for (size_t l = 0; l < nq; l++) ntt_fwd(&y[(i * nq + l) * N], *o->key_tables[l]);
        tensor(x.data(), a.size, y.data(), b.size, dest.data(), o->level_q(a.limbs), N);
        for (int i = 0; i < dsz; i++) for (size_t l = 0; l < nq; l++) ntt_inv(&dest[(i * nq + l) * N], *o->key_tables[l]);
        a.d.swap(dest);
        a.size = dsz;
        a.cf = mulmod(a.cf, b.cf, o->t);
    }
    void multiply(Ct &a, const Ct &b) const { // evaluator.cpp:349-383
        check_level(a);
        if (a.limbs != b.limbs) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        if (o->scheme == ORC_BFV) bfv_multiply(a, b);
        else if (o->scheme == ORC_CKKS) ckks_multiply(a, b);
        else bgv_multiply(a, b);
    }

    // evaluator.cpp:2310-2653 switchKeyInplace.  key layout [decomp j][2][K][N]
    void switch_key(Ct &ct, const u64 *target, const std::vector<u64> &key) const {
        if (o->K < 2) throw std::logic_error("keyswitching is not supported by the context");
        if (o->scheme == ORC_BFV && ct.ntt) throw std::invalid_argument("BFV encrypted cannot be in NTT form");
        if (o->scheme == ORC_CKKS && !ct.ntt) throw std::invalid_argument("CKKS encrypted must be in NTT form");
        if (o->scheme == ORC_BGV && ct.ntt) throw std::invalid_argument("BGV encrypted cannot be in NTT form");
        size_t K = o->K, dl = ct.limbs, rl = dl + 1;
        const RnsTool &key_rt = o->tool((int)K);
        std::vector<u64> t_target(target, target + dl * N);
        if (o->scheme == ORC_CKKS) for (size_t l = 0; l < dl; l++) ntt_inv(&t_target[l * N], *o->key_tables[l]);
        std::vector<u64> prod(2 * rl * N, 0), t_ntt(N);
        std::vector<u128> acc(2 * N);
        for (size_t i = 0; i < rl; i++) {
            size_t ki = (i == dl ? K - 1 : i);
            const Mod &km = o->key_q[ki];
            std::fill(acc.begin(), acc.end(), (u128)0);
            for (size_t j = 0; j < dl; j++) {
                const u64 *operand;
                if (o->scheme == ORC_CKKS && i == j) operand = target + j * N;
                else {
                    if (o->key_q[j].p <= km.p) std::copy(&t_target[j * N], &t_target[(j + 1) * N], t_ntt.begin());
                    else for (size_t l = 0; l < N; l++) t_ntt[l] = barrett64(t_target[j * N + l], km);
                    ntt_fwd_lazy(t_ntt.data(), *o->key_tables[ki]);
                    operand = t_ntt.data();
                }
                for (size_t k = 0; k < 2; k++) {
                    const u64 *kp = &key[((j * 2 + k) * K + ki) * N];
                    u128 *ac = &acc[k * N];
                    // 128-bit lazy accumulation; reduce before it can overflow (the reference reduces
                    // every 256 summands, evaluator.cpp:2401-2460 -- same value mod p either way)
                    for (size_t l = 0; l < N; l++) {
                        u128 pr = (u128)operand[l] * kp[l];
                        u128 s = ac[l] + pr;
                        if (s < pr) s = (u128)barrett128(ac[l], km) + pr; // overflow guard (never hit for j < 256)
                        ac[l] = s;
                    }
                }
            }
            for (size_t k = 0; k < 2; k++)
                for (size_t l = 0; l < N; l++) prod[(k * rl + i) * N + l] = barrett128(acc[k * N + l], km);
        }
        const Mod &qk_mod = o->key_q[K - 1];
        u64 qk = qk_mod.p;
        for (size_t k = 0; k < 2; k++) {
            u64 *t_last = &prod[(k * rl + dl) * N];
            if (o->scheme == ORC_BGV) { // evaluator.cpp:2528-2577
                u64 qk_inv_t = key_rt.inv_q_last_mod_t;
                ntt_inv(t_last, *o->key_tables[K - 1]);
                std::vector<u64> kk(N);
                for (size_t l = 0; l < N; l++) {
                    u64 v = negmod(barrett64(t_last[l], o->t), o->t);
                    if (qk_inv_t != 1) v = mulmod(v, qk_inv_t, o->t);
                    kk[l] = v;
                }
                for (size_t j = 0; j < dl; j++) {
                    const Mod &qj = o->key_q[j];
                    u64 *pp = &prod[(k * rl + j) * N];
                    ntt_inv(pp, *o->key_tables[j]);
                    u64 qk_mod_qj = barrett64(qk, qj);
                    u64 *cp = ct.poly(k, N) + j * N;
                    for (size_t l = 0; l < N; l++) {
                        u64 delta = mulmod(barrett64(kk[l], qj), qk_mod_qj, qj);
                        u64 c_mod = barrett64(t_last[l], qj);
                        u64 v = pp[l] + 2 * qj.p - (delta + c_mod);
                        v = mul_shoup(v, key_rt.inv_q_last_mod_q[j], qj.p);
                        cp[l] = addmod(v, cp[l], qj);
                    }
                }
            } else { // evaluator.cpp:2579-2648
                ntt_inv_lazy(t_last, *o->key_tables[K - 1]);
                u64 half = qk >> 1;
                for (size_t l = 0; l < N; l++) t_last[l] = barrett64(t_last[l] + half, qk_mod);
                for (size_t j = 0; j < dl; j++) {
                    const Mod &qj = o->key_q[j];
                    u64 *pp = &prod[(k * rl + j) * N];
                    if (qk > qj.p) for (size_t l = 0; l < N; l++) t_ntt[l] = barrett64(t_last[l], qj);
                    else std::copy(t_last, t_last + N, t_ntt.begin());
                    u64 fix = qj.p - barrett64(half, qj);
                    for (size_t l = 0; l < N; l++) t_ntt[l] += fix;
                    u64 qi_lazy = qj.p << 1;
                    if (o->scheme == ORC_CKKS) { ntt_fwd_lazy(t_ntt.data(), *o->key_tables[j]); qi_lazy = qj.p << 2; }
                    else ntt_inv_lazy(pp, *o->key_tables[j]);
                    u64 *cp = ct.poly(k, N) + j * N;
                    for (size_t l = 0; l < N; l++) {
                        u64 v = mul_shoup(pp[l] + qi_lazy - t_ntt[l], key_rt.inv_q_last_mod_q[j], qj.p);
                        cp[l] = addmod(v, cp[l], qj);
                    }
                }
            }
        }
    }
    // evaluator.cpp:1113-1163 relinearizeInternal, destination size 2.  As in the reference, `encrypted_iter` keeps pointing at the LAST
    // polynomial through all relins_needed steps (evaluator.cpp:1149-1156): step i switches that same polynomial with the key of
    // index getIndex(size - 1 - i) = size - 3 - i, and the polynomials 2 .. size-2 are dropped by the final resize.
    void relinearize(Ct &a) const {
        check_level(a);
        if (a.size == 2) return;
        const int size = a.size;
        auto key_of = [&](int index) -> const std::vector<u64> & {
            if (index == 0) { if (o->relin_key.empty()) throw std::invalid_argument("not enough relinearization keys"); return o->relin_key; }
            auto it = o->relin_keys_hi.find(index);
            if (it == o->relin_keys_hi.end()) throw std::invalid_argument("not enough relinearization keys");
            return it->second;
        };
        for (int i = 0; i < size - 2; i++) key_of(i); // relin_keys.size() >= size - 2 (evaluator.cpp:1134)
        std::vector<u64> target(a.poly(size - 1, N), a.poly(size - 1, N) + (size_t)a.limbs * N);
        for (int i = 0; i < size - 2; i++) switch_key(a, target.data(), key_of(size - 3 - i));
        a.size = 2;
        a.d.resize((size_t)2 * a.limbs * N);
    }
    // evaluator.cpp:1165-1225 modSwitchScaleToNext
    void mod_switch_scale(Ct &a) const {
        check_level(a);
        if (!o->level_exists(a.limbs - 1) || a.limbs < 2) throw std::invalid_argument("end of modulus switching chain reached");
        if (o->scheme == ORC_BFV && a.ntt) throw std::invalid_argument("BFV encrypted cannot be in NTT form");
        if (o->scheme == ORC_CKKS && !a.ntt) throw std::invalid_argument("CKKS encrypted must be in NTT form");
        if (o->scheme == ORC_BGV && a.ntt) throw std::invalid_argument("BGV encrypted cannot be in NTT form");
        const RnsTool &rt = o->tool(a.limbs);
        int nl = a.limbs - 1;
        std::vector<u64> out((size_t)a.size * nl * N);
        for (int i = 0; i < a.size; i++) {
            u64 *x = a.poly(i, N);
            if (o->scheme == ORC_BFV) rt.divide_round_qlast(x);
            else if (o->scheme == ORC_CKKS) rt.divide_round_qlast_ntt(x, o->key_tables);
            else rt.modt_divide_qlast(x);
            std::copy(x, x + (size_t)nl * N, &out[(size_t)i * nl * N]);
        }
        if (o->scheme == ORC_CKKS) a.scale = a.scale / (double)o->key_q[a.limbs - 1].p;
        else if (o->scheme == ORC_BGV) a.cf = mulmod(a.cf, rt.inv_q_last_mod_t, o->t);
        a.d.swap(out);
        a.limbs = nl;
    }
    void mod_switch_drop(Ct &a) const { // evaluator.cpp:1227-1290
        check_level(a);
        if (!o->level_exists(a.limbs - 1) || a.limbs < 2) throw std::invalid_argument("end of modulus switching chain reached");
        if (o->scheme == ORC_CKKS && !a.ntt) throw std::invalid_argument("CKKS encrypted must be in NTT form");
        if (!scale_ok(a.scale, a.limbs - 1)) throw std::invalid_argument("scale out of bounds");
        int nl = a.limbs - 1;
        std::vector<u64> out((size_t)a.size * nl * N);
        for (int i = 0; i < a.size; i++) std::copy(a.poly(i, N), a.poly(i, N) + (size_t)nl * N, &out[(size_t)i * nl * N]);
        a.d.swap(out);
        a.limbs = nl;
    }
    void mod_switch_to_next(Ct &a) const { // evaluator.cpp:1330-1370
        if (o->scheme == ORC_CKKS) mod_switch_drop(a);
        else mod_switch_scale(a);
    }
    void rescale_to_next(Ct &a) const { // evaluator.cpp:1440-1470
        if (o->scheme != ORC_CKKS) throw std::invalid_argument("unsupported operation for scheme type");
        mod_switch_scale(a);
    }
    // evaluator.cpp:2153-2249 applyGaloisInplace
    void apply_galois_ct(Ct &a, uint32_t elt) const {
        check_level(a);
        auto it = o->galois_keys.find(elt);
        if (it == o->galois_keys.end()) throw std::invalid_argument("Galois key not present");
        if (!(elt & 1) || elt >= 2 * N) throw std::invalid_argument("Galois element is not valid");
        if (a.size > 2) throw std::invalid_argument("encrypted size must be 2");
        size_t nl = a.limbs;
        std::vector<u64> temp(nl * N);
        for (int pidx = 0; pidx < 2; pidx++) {
            for (size_t l = 0; l < nl; l++) {
                if (o->scheme == ORC_CKKS) apply_galois_ntt(N, o->logn, elt, a.poly(pidx, N) + l * N, &temp[l * N]);
                else apply_galois(N, o->logn, elt, o->key_q[l], a.poly(pidx, N) + l * N, &temp[l * N]);
            }
            if (pidx == 0) std::copy(temp.begin(), temp.end(), a.poly(0, N));
        }
        std::fill(a.poly(1, N), a.poly(1, N) + nl * N, 0);
        switch_key(a, temp.data(), it->second);
    }
    // evaluator.cpp:2251-2307 rotateInternal
    void rotate(Ct &a, int steps) const {
        if (steps == 0) return;
        uint32_t elt = elt_from_step(N, steps);
        if (o->galois_keys.count(elt)) { apply_galois_ct(a, elt); return; }
        auto ns = naf(steps);
        if (ns.size() == 1) throw std::invalid_argument("Galois key not present");
        for (int s : ns) if ((size_t)std::abs(s) != (N >> 1)) rotate(a, s);
    }
    void to_ntt(Ct &a) const { // evaluator.cpp transformToNttInplace(Ciphertext)
        check_level(a);
        if (a.ntt) throw std::invalid_argument("encrypted is already in NTT form");
        for (int i = 0; i < a.size; i++) for (int l = 0; l < a.limbs; l++) ntt_fwd(a.poly(i, N) + l * N, *o->key_tables[l]);
        a.ntt = true;
    }
    void from_ntt(Ct &a) const {
        check_level(a);
        if (!a.ntt) throw std::invalid_argument("encrypted_ntt is not in NTT form");
        for (int i = 0; i < a.size; i++) for (int l = 0; l < a.limbs; l++) ntt_inv(a.poly(i, N) + l * N, *o->key_tables[l]);
        a.ntt = false;
    }
    void multiply_plain_ntt(Ct &a, const u64 *plain, double pscale) const { // evaluator.cpp multiplyPlainNtt
        check_level(a);
        if (!a.ntt) throw std::invalid_argument("encrypted_ntt is not in NTT form");
        for (int i = 0; i < a.size; i++)
            for (int l = 0; l < a.limbs; l++) {
                const Mod &m = o->key_q[l];
                u64 *x = a.poly(i, N) + l * N;
                for (size_t j = 0; j < N; j++) x[j] = mulmod(x[j], plain[l * N + j], m);
            }
        a.scale *= pscale;
        if (o->scheme == ORC_CKKS && !scale_ok(a.scale, a.limbs)) throw std::invalid_argument("scale out of bounds");
    }

    // EvaluatorCuda::applyKeySwitchingInplace (evaluator_cuda.cu:1365-1378; CUDA-only API): c1 becomes the key-switch target,
    // c1 := 0, then switchKeyInplace with the single key of `kswitch_keys` (held in the relin-key slot here)
    void apply_key_switching(Ct &a) const {
        check_level(a);
        if (a.size != 2) throw std::invalid_argument("encrypted.size() != 2");
        std::vector<u64> target(a.poly(1, N), a.poly(1, N) + (size_t)a.limbs * N);
        std::fill(a.poly(1, N), a.poly(1, N) + (size_t)a.limbs * N, 0);
        switch_key(a, target.data(), o->relin_key);
    }
    // EvaluatorCuda::negacyclicShift (evaluator_cuda.cu:2342-2351) = util::negacyclicShiftPolyCoeffmod on every limb of every
    // polynomial (polyarithsmallmod.cpp:128-152): result[(i + shift) mod N] = +-poly[i], negated on wrap-around, 0 stays 0
    void negacyclic_shift(Ct &a, size_t shift) const {
        check_level(a);
        std::vector<u64> tmp(N);
        for (int i = 0; i < a.size; i++)
            for (int l = 0; l < a.limbs; l++) {
                const Mod &m = o->key_q[l];
                u64 *x = a.poly(i, N) + (size_t)l * N;
                std::copy(x, x + N, tmp.begin());
                if (shift == 0) continue;
                for (size_t j = 0; j < N; j++) {
                    const size_t raw = j + shift, idx = raw & (N - 1);
                    x[idx] = (!(raw & N) || !tmp[j]) ? tmp[j] : m.p - tmp[j];
                }
            }
    }

    // ---- plaintext operands ----
    // context.cpp:307-351: Delta_l = floor(q/t) mod q_l, q mod t, plain_upper_half_threshold = (t+1)/2.  floor(q/t) =
    // (q - q mod t)/t, so modulo q_l (q = 0 there) Delta_l = -(q mod t) * t^-1; the reference divides the multi-word q.
    struct PlainConsts {
        std::vector<u64> delta;
        u64 q_mod_t, thr;
    };
    PlainConsts plain_consts(int limbs) const {
        PlainConsts c;
        auto q = o->level_q(limbs);
        c.q_mod_t = prod_mod(q, o->t);
        c.thr = (o->t.p + 1) >> 1;
        for (auto &m : q) c.delta.push_back(mulmod(negmod(barrett64(c.q_mod_t, m), m), invmod_or_throw(barrett64(o->t.p, m), m), m));
        return c;
    }
    // evaluator.cpp:1791-1930 / 1972-2070 lifting of a plaintext coefficient m (mod t) to q_l: m if m < (t+1)/2, else
    // m + (q - t).  fast lift (every q_l > t, context.cpp:299-305): m + (q_l - t), left unreduced for the lazy NTT;
    // otherwise the multi-word sum is decomposed (rns.cpp decomposeArray).  Both are the residue returned here (q = 0 mod q_l),
    // and everything downstream canonicalises (nttNegacyclicHarvey), so the stored limbs are identical.
    void lift_plain(const u64 *plain, size_t n, int limbs, u64 *out) const {
        const u64 thr = (o->t.p + 1) >> 1;
        for (int l = 0; l < limbs; l++) {
            const Mod &m = o->key_q[l];
            const u64 t_mod = barrett64(o->t.p, m);
            for (size_t j = 0; j < N; j++) {
                u64 v = 0;
                if (j < n) {
                    v = barrett64(plain[j], m);
                    if (plain[j] >= thr) v = submod(v, t_mod, m);
                }
                out[(size_t)l * N + j] = v;
            }
        }
    }
    void check_plain(const u64 *plain, size_t n) const {
        if (n > N) throw std::invalid_argument("plain is not valid for encryption parameters");
        for (size_t j = 0; j < n; j++)
            if (plain[j] >= o->t.p) throw std::invalid_argument("plain is not valid for encryption parameters");
    }
    // evaluator.cpp:1603-1763 addPlainInplace / subPlainInplace; scalingvariant.cpp:17-134
    void add_plain(Ct &a, const u64 *plain, size_t n, double pscale, bool sub) const {
        check_level(a);
        if (o->scheme == ORC_BFV && a.ntt) throw std::invalid_argument("BFV encrypted cannot be in NTT form");
        if (o->scheme == ORC_CKKS && !a.ntt) throw std::invalid_argument("CKKS encrypted must be in NTT form");
        if (o->scheme == ORC_BGV && a.ntt) throw std::invalid_argument("BGV encrypted cannot be in NTT form");
        if (o->scheme == ORC_CKKS) { // plain: [limbs][N] NTT form at the same level
            if (std::fabs(a.scale - pscale) >= std::ldexp(1.0, -23)) throw std::invalid_argument("scale mismatch");
            for (int l = 0; l < a.limbs; l++) {
                const Mod &m = o->key_q[l];
                u64 *x = a.poly(0, N) + (size_t)l * N;
                for (size_t j = 0; j < N; j++) x[j] = sub ? submod(x[j], plain[l * N + j], m) : addmod(x[j], plain[l * N + j], m);
            }
            return;
        }
        check_plain(plain, n);
        if (o->scheme == ORC_BFV) { // multiplyAdd/SubPlainWithScalingVariant: round(q m / t) = Delta m + floor((q mod t) m + (t+1)/2) / t)
            const PlainConsts c = plain_consts(a.limbs);
            for (size_t j = 0; j < n; j++) {
                const u128 num = (u128)plain[j] * c.q_mod_t + c.thr;
                const u64 fix = (u64)(num / o->t.p);
                for (int l = 0; l < a.limbs; l++) {
                    const Mod &m = o->key_q[l];
                    const u64 v = barrett128((u128)plain[j] * c.delta[l] + fix, m); // multiplyAddUintMod
                    u64 &x = a.poly(0, N)[(size_t)l * N + j];
                    x = sub ? submod(x, v, m) : addmod(x, v, m);
                }
            }
        } else { // BGV: plain * correction_factor mod t, then add/subPlainWithoutScalingVariant
            for (size_t j = 0; j < n; j++) {
                const u64 pc = mulmod(plain[j], a.cf, o->t);
                for (int l = 0; l < a.limbs; l++) {
                    const Mod &m = o->key_q[l];
                    const u64 v = barrett64(pc, m);
                    u64 &x = a.poly(0, N)[(size_t)l * N + j];
                    x = sub ? submod(x, v, m) : addmod(x, v, m);
                }
            }
        }
    }
    // evaluator.cpp:1791-1930 multiplyPlainNormal (generic branch; the monomial branch gives the same canonical limbs)
    // The CPU reference short-cuts a plaintext with ONE nonzero coefficient (evaluator.cpp:1816-1867): scalar multiply +
    // negacyclic shift (polyarithsmallmod.h:271-279, .cpp:128-152), and in fast-lift mode it multiplies by the coefficient
    // ITSELF even when it is in the upper half (no q - t adjustment), so those limbs differ from the generic branch by a
    // multiple of t.  The reference's CUDA evaluator has no such branch (evaluator_cuda.cu:1757-1815): `generic_only`
    // selects that behaviour (the drop-in target of the GPU product).
    void multiply_plain_normal(Ct &a, const u64 *plain, size_t n, bool generic_only) const {
        check_level(a);
        if (a.ntt) throw std::invalid_argument("NTT form mismatch");
        check_plain(plain, n);
        size_t nonzero = 0, mono = 0;
        for (size_t j = 0; j < n; j++)
            if (plain[j]) { nonzero++; mono = j; }
        if (nonzero == 1 && !generic_only) {
            const u64 c = plain[mono], thr = (o->t.p + 1) >> 1;
            bool fast = true;
            for (int l = 0; l < a.limbs; l++) fast = fast && o->key_q[l].p > o->t.p;
            std::vector<u64> tmp(N);
            for (int i = 0; i < a.size; i++)
                for (int l = 0; l < a.limbs; l++) {
                    const Mod &m = o->key_q[l];
                    u64 cl = barrett64(c, m);
                    if (c >= thr && !fast) cl = submod(cl, barrett64(o->t.p, m), m); // (q - t + c) mod q_l
                    u64 *x = a.poly(i, N) + (size_t)l * N;
                    for (size_t j = 0; j < N; j++) tmp[j] = mulmod(x[j], cl, m);
                    for (size_t j = 0; j < N; j++) {
                        const size_t raw = j + mono, idx = raw & (N - 1);
                        x[idx] = (!(raw & N) || !tmp[j]) ? tmp[j] : m.p - tmp[j];
                    }
                }
            return;
        }
        std::vector<u64> temp((size_t)a.limbs * N);
        lift_plain(plain, n, a.limbs, temp.data());
        for (int l = 0; l < a.limbs; l++) ntt_fwd(temp.data() + (size_t)l * N, *o->key_tables[l]);
        for (int i = 0; i < a.size; i++)
            for (int l = 0; l < a.limbs; l++) {
                const Mod &m = o->key_q[l];
                u64 *x = a.poly(i, N) + (size_t)l * N;
                ntt_fwd_lazy(x, *o->key_tables[l]);
                for (size_t j = 0; j < N; j++) x[j] = mulmod(x[j], temp[(size_t)l * N + j], m);
                ntt_inv(x, *o->key_tables[l]);
            }
    }
    // evaluator.cpp:1972-2070 transformToNttInplace(Plaintext, parms_id): out [limbs][N]
    void plain_to_ntt(const u64 *plain, size_t n, int limbs, u64 *out) const {
        if (!o->level_exists(limbs) && limbs != (int)o->K) throw std::invalid_argument("parms_id is not valid for the current context");
        check_plain(plain, n);
        lift_plain(plain, n, limbs, out);
        for (int l = 0; l < limbs; l++) ntt_fwd(out + (size_t)l * N, *o->key_tables[l]);
    }

    // decryptor.cpp:284-371 dotProductCtSkArray + :115-153 bfvDecrypt / :185-222 bgvDecrypt / :155-183 ckks
    void decrypt(const Ct &c, const u64 *sk, u64 *out) const {
        size_t nl = c.limbs, K = o->K;
        std::vector<u64> acc(nl * N, 0), tmp(N), skp(nl * N);
        for (size_t l = 0; l < nl; l++) std::copy(sk + l * N, sk + (l + 1) * N, &skp[l * N]);
        std::vector<u64> skpow(skp);
        for (int i = 1; i < c.size; i++) {
            for (size_t l = 0; l < nl; l++) {
                const Mod &m = o->key_q[l];
                std::copy(c.poly(i, N) + l * N, c.poly(i, N) + (l + 1) * N, tmp.begin());
                if (!c.ntt) ntt_fwd_lazy(tmp.data(), *o->key_tables[l]);
                for (size_t j = 0; j < N; j++) acc[l * N + j] = addmod(acc[l * N + j], mulmod(tmp[j], skpow[l * N + j], m), m);
            }
            if (i + 1 < c.size)
                for (size_t l = 0; l < nl; l++)
                    for (size_t j = 0; j < N; j++) skpow[l * N + j] = mulmod(skpow[l * N + j], skp[l * N + j], o->key_q[l]);
        }
        for (size_t l = 0; l < nl; l++) {
            const Mod &m = o->key_q[l];
            if (!c.ntt) ntt_inv(&acc[l * N], *o->key_tables[l]);
            for (size_t j = 0; j < N; j++) acc[l * N + j] = addmod(acc[l * N + j], c.poly(0, N)[l * N + j], m);
        }
        (void)K;
        const RnsTool &rt = o->tool(c.limbs);
        if (o->scheme == ORC_BFV) rt.decrypt_scale_and_round(acc.data(), out);
        else if (o->scheme == ORC_BGV) {
            rt.q_to_t.exact_convert(acc.data(), out, N);
            if (c.cf != 1) {
                u64 fix;
                if (!invmod(c.cf, o->t.p, fix)) throw std::logic_error("invalid correction factor");
                for (size_t j = 0; j < N; j++) out[j] = mulmod(out[j], fix, o->t);
            }
        } else std::copy(acc.begin(), acc.end(), out); // CKKS: raw RNS plaintext [limbs][N] in NTT form
    }
};

Ct make_ct(Orc *o, const orc_ct_desc *d, const u64 *data) {
    Ct c;
    c.size = d->size; c.limbs = d->limbs; c.ntt = d->is_ntt != 0; c.scale = d->scale; c.cf = d->correction_factor;
    c.d.assign(data, data + (size_t)c.size * c.limbs * o->N);
    return c;
}

template <class F> int guarded(Orc *o, F f) {
    try { f(); return 0; } catch (const std::exception &e) { o->err = e.what(); return -1; }
}

} // namespace

// ================================================================================================
extern "C" {

uint64_t orc_barrett_reduce_64(uint64_t x, uint64_t p) { return barrett64(x, Mod(p)); }
uint64_t orc_barrett_reduce_128(uint64_t lo, uint64_t hi, uint64_t p) { return barrett128(((u128)hi << 64) | lo, Mod(p)); }
uint64_t orc_multiply_uint_mod(uint64_t a, uint64_t b, uint64_t p) { return mulmod(a, b, Mod(p)); }
uint64_t orc_shoup_quotient(uint64_t w, uint64_t p) { return Shoup(w, Mod(p)).quo; }
uint64_t orc_multiply_uint_mod_lazy(uint64_t x, uint64_t w, uint64_t p) { Mod m(p); return mul_lazy(x, Shoup(w, m), p); }
uint64_t orc_exponentiate_uint_mod(uint64_t a, uint64_t e, uint64_t p) { return powmod(a, e, Mod(p)); }
int orc_try_invert_uint_mod(uint64_t a, uint64_t p, uint64_t *out) { return invmod(a, p, *out) ? 1 : 0; }
uint64_t orc_dot_product_mod(const uint64_t *a, const uint64_t *b, int n, uint64_t p) { return dotmod(a, b, n, Mod(p)); }
int orc_is_prime(uint64_t p) { return is_prime(p) ? 1 : 0; }
int orc_try_minimal_primitive_root(uint64_t degree, uint64_t p, uint64_t *out) { return min_primitive_root(degree, Mod(p), *out) ? 1 : 0; }
void orc_modulus_const_ratio(uint64_t p, uint64_t *out3) {
    Mod m(p);
    out3[0] = m.cr0; out3[1] = m.cr1;
    // remainder 2^128 - floor(2^128/p)*p (modulus.cpp:27-37)
    u128 q = ((u128)m.cr1 << 64) | m.cr0;
    out3[2] = (u64)((u128)0 - q * p);
}
void orc_naf(int value, int *out, int *n_out) {
    auto v = naf(value);
    for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
    *n_out = (int)v.size();
}
int orc_get_primes(uint64_t factor, int bits, int count, uint64_t *out) {
    try { auto v = get_primes(factor, bits, count); std::copy(v.begin(), v.end(), out); return 0; } catch (...) { return -1; }
}
int orc_coeff_modulus_create(uint64_t N, const int *bits, int n, uint64_t *out) {
    try { auto v = coeff_modulus_create(N, std::vector<int>(bits, bits + n)); std::copy(v.begin(), v.end(), out); return 0; } catch (...) { return -1; }
}
uint64_t orc_plain_batching(uint64_t N, int bits) { // modulus.h:528-531; 0 = no such prime (the reference throws logic_error)
    try { return coeff_modulus_create(N, {bits})[0]; } catch (...) { return 0; }
}

void *orc_create(int scheme, uint64_t N, const uint64_t *primes, int K, uint64_t t) {
    auto *o = new Orc();
    try {
        o->scheme = scheme; o->N = N; o->K = K;
        o->logn = 63 - __builtin_clzll(N);
        if ((u64(1) << o->logn) != N) throw std::invalid_argument("poly_modulus_degree");
        o->t = (scheme == ORC_CKKS) ? Mod() : Mod(t);
        for (int i = 0; i < K; i++) { o->key_q.emplace_back(primes[i]); o->key_tables.push_back(o->table(primes[i])); }
        // level chain (context.cpp:426-531): key level, then drop the last prime repeatedly while valid
        int first = K > 1 ? K - 1 : K;
        o->first_limbs = first;
        o->n_levels = 0;
        for (int limbs = K; limbs >= 1; limbs--) {
            auto q = o->level_q(limbs);
            if (scheme != ORC_CKKS) { // context.cpp:253-262: plain modulus must be smaller than the coeff modulus
                int bits = product_bit_count(q);
                if (bits <= 64) {
                    u128 prod = 1;
                    for (auto &m : q) prod *= m.p;
                    if (prod <= (u128)o->t.p) break;
                }
            }
            auto rt = std::make_unique<RnsTool>();
            rt->init(o, q, o->t);
            o->tools[limbs] = std::move(rt);
            o->last_limbs = limbs;
            o->n_levels++;
        }
        if (!o->level_exists(first)) throw std::invalid_argument("invalid parameters");
    } catch (const std::exception &e) {
        fprintf(stderr, "orc_create: %s\n", e.what());
        delete o;
        return nullptr;
    }
    return o;
}
void orc_destroy(void *h) { delete (Orc *)h; }
const char *orc_last_error(void *h) { return ((Orc *)h)->err.c_str(); }
int orc_chain(void *h, int *first_limbs, int *last_limbs) {
    Orc *o = (Orc *)h;
    *first_limbs = o->first_limbs; *last_limbs = o->last_limbs;
    return o->n_levels;
}
int orc_ntt_tables(void *h, int prime_idx, uint64_t *rop, uint64_t *rquo, uint64_t *iop, uint64_t *iquo,
                   uint64_t *inv_degree2, uint64_t *root) {
    Orc *o = (Orc *)h;
    const NttTable &t = *o->key_tables[prime_idx];
    for (size_t i = 0; i < o->N; i++) { rop[i] = t.root[i].op; rquo[i] = t.root[i].quo; iop[i] = t.iroot[i].op; iquo[i] = t.iroot[i].quo; }
    inv_degree2[0] = t.inv_n.op; inv_degree2[1] = t.inv_n.quo;
    *root = t.psi;
    return 0;
}
int orc_behz_bases(void *h, int limbs, uint64_t *bsk_out, uint64_t *gamma) {
    Orc *o = (Orc *)h;
    if (!o->level_exists(limbs)) return -1;
    const RnsTool &rt = o->tool(limbs);
    for (size_t i = 0; i < rt.Bsk.size(); i++) bsk_out[i] = rt.Bsk[i].p;
    *gamma = rt.gamma.p;
    return (int)rt.Bsk.size();
}
int orc_bsk_ntt_tables(void *h, int limbs, int idx, uint64_t *rop, uint64_t *iop, uint64_t *inv_degree) {
    Orc *o = (Orc *)h;
    if (!o->level_exists(limbs)) return -1;
    const NttTable &t = *o->tool(limbs).bsk_tables[idx];
    for (size_t i = 0; i < o->N; i++) { rop[i] = t.root[i].op; iop[i] = t.iroot[i].op; }
    *inv_degree = t.inv_n.op;
    return 0;
}
static void run_ntt(u64 *data, const NttTable &t, int mode) {
    switch (mode) {
    case 0: ntt_fwd_lazy(data, t); break;
    case 1: ntt_fwd(data, t); break;
    case 2: ntt_inv_lazy(data, t); break;
    default: ntt_inv(data, t); break;
    }
}
int orc_ntt(void *h, int prime_idx, uint64_t *data, int mode) {
    Orc *o = (Orc *)h;
    run_ntt(data, *o->key_tables[prime_idx], mode);
    return 0;
}
int orc_ntt_standalone(uint64_t N, uint64_t p, uint64_t *data, int mode) {
    try {
        NttTable t(63 - __builtin_clzll(N), p);
        run_ntt(data, t, mode);
        return 0;
    } catch (...) { return -1; }
}
int orc_rns_stage(void *h, int limbs, int stage, const uint64_t *in, uint64_t *out) {
    Orc *o = (Orc *)h;
    return guarded(o, [&] {
        const RnsTool &rt = o->tool(limbs);
        size_t N = o->N;
        switch (stage) {
        case ORC_ST_FASTBCONV_MTILDE: rt.fastbconv_mtilde(in, out); break;
        case ORC_ST_SMMRQ: rt.sm_mrq(in, out); break;
        case ORC_ST_FASTFLOOR: rt.fast_floor(in, out); break;
        case ORC_ST_FASTBCONV_SK: rt.fastbconv_sk(in, out); break;
        case ORC_ST_DIVROUND_QLAST: std::copy(in, in + limbs * N, out); rt.divide_round_qlast(out); break;
        case ORC_ST_DIVROUND_QLAST_NTT: std::copy(in, in + limbs * N, out); rt.divide_round_qlast_ntt(out, o->key_tables); break;
        case ORC_ST_MODT_DIV_QLAST: std::copy(in, in + limbs * N, out); rt.modt_divide_qlast(out); break;
        default: throw std::invalid_argument("stage");
        }
    });
}
int orc_set_kswitch_key(void *h, uint32_t which, const uint64_t *data) {
    Orc *o = (Orc *)h;
    size_t n = (o->K - 1) * 2 * o->K * o->N;
    if (which == 0 || which == 0x80000000u) o->relin_key.assign(data, data + n);
    else if (which & 0x80000000u) o->relin_keys_hi[(int)(which & 0xFFFFu)].assign(data, data + n);
    else o->galois_keys[which].assign(data, data + n);
    return 0;
}
int orc_eval(void *h, int op, const orc_ct_desc *ad, const uint64_t *a, const orc_ct_desc *bd, const uint64_t *b,
             int64_t iarg, orc_ct_desc *od, uint64_t *out) {
    Orc *o = (Orc *)h;
    return guarded(o, [&] {
        Eval ev(o);
        Ct x = make_ct(o, ad, a);
        switch (op) {
        case ORC_OP_ADD: ev.add_sub(x, make_ct(o, bd, b), false); break;
        case ORC_OP_SUB: ev.add_sub(x, make_ct(o, bd, b), true); break;
        case ORC_OP_NEGATE: ev.negate(x); break;
        case ORC_OP_MULTIPLY: ev.multiply(x, make_ct(o, bd, b)); break;
        case ORC_OP_SQUARE: { Ct y = x; ev.multiply(x, y); break; } // evaluator.cpp:796-1111: same values as x*x
        case ORC_OP_RELIN: ev.relinearize(x); break;
        case ORC_OP_MODSWITCH_NEXT: ev.mod_switch_to_next(x); break;
        case ORC_OP_RESCALE_NEXT: ev.rescale_to_next(x); break;
        case ORC_OP_APPLY_GALOIS: ev.apply_galois_ct(x, (uint32_t)iarg); break;
        case ORC_OP_ROTATE_ROWS:
            if (o->scheme == ORC_CKKS) throw std::logic_error("unsupported scheme");
            ev.rotate(x, (int)iarg); break;
        case ORC_OP_ROTATE_COLUMNS:
            if (o->scheme == ORC_CKKS) throw std::logic_error("unsupported scheme");
            ev.apply_galois_ct(x, (uint32_t)(2 * o->N - 1)); break;
        case ORC_OP_ROTATE_VECTOR:
            if (o->scheme != ORC_CKKS) throw std::logic_error("unsupported scheme");
            ev.rotate(x, (int)iarg); break;
        case ORC_OP_CONJUGATE:
            if (o->scheme != ORC_CKKS) throw std::logic_error("unsupported scheme");
            ev.apply_galois_ct(x, (uint32_t)(2 * o->N - 1)); break;
        case ORC_OP_TO_NTT: ev.to_ntt(x); break;
        case ORC_OP_FROM_NTT: ev.from_ntt(x); break;
        case ORC_OP_MULTIPLY_PLAIN_NTT: ev.multiply_plain_ntt(x, b, bd ? bd->scale : 1.0); break;
        case ORC_OP_APPLY_KEYSWITCH: ev.apply_key_switching(x); break;
        case ORC_OP_NEGACYCLIC_SHIFT: ev.negacyclic_shift(x, (size_t)iarg); break;
        case ORC_OP_ADD_PLAIN: ev.add_plain(x, b, (size_t)iarg, bd ? bd->scale : 1.0, false); break;
        case ORC_OP_SUB_PLAIN: ev.add_plain(x, b, (size_t)iarg, bd ? bd->scale : 1.0, true); break;
        case ORC_OP_MULTIPLY_PLAIN: ev.multiply_plain_normal(x, b, (size_t)(iarg & 0xFFFFFFFF), (iarg >> 32) & 1); break; // bit 32: CUDA-evaluator semantics
        default: throw std::invalid_argument("op");
        }
        od->limbs = x.limbs; od->size = x.size; od->is_ntt = x.ntt; od->scale = x.scale; od->correction_factor = x.cf;
        std::copy(x.d.begin(), x.d.begin() + (size_t)x.size * x.limbs * o->N, out);
    });
}
int orc_plain_to_ntt(void *h, const uint64_t *plain, int n_coeffs, int limbs, uint64_t *out) {
    Orc *o = (Orc *)h;
    return guarded(o, [&] { Eval(o).plain_to_ntt(plain, (size_t)n_coeffs, limbs, out); });
}
uint32_t orc_galois_elt_from_step(void *h, int step) {
    Orc *o = (Orc *)h;
    try { return elt_from_step(o->N, step); } catch (...) { return 0; }
}
void orc_apply_galois(uint64_t N, uint32_t elt, uint64_t p, const uint64_t *in, uint64_t *out) {
    apply_galois(N, 63 - __builtin_clzll(N), elt, Mod(p), in, out);
}
void orc_apply_galois_ntt(uint64_t N, uint32_t elt, const uint64_t *in, uint64_t *out) {
    apply_galois_ntt(N, 63 - __builtin_clzll(N), elt, in, out);
}
int orc_decrypt(void *h, const uint64_t *sk, const orc_ct_desc *ad, const uint64_t *a, uint64_t *plain_out) {
    Orc *o = (Orc *)h;
    return guarded(o, [&] {
        Eval ev(o);
        Ct x = make_ct(o, ad, a);
        ev.decrypt(x, sk, plain_out);
    });
}

double orc_time_mul_relin(void *h, const uint64_t *a, const uint64_t *b, int count, int threads) {
    Orc *o = (Orc *)h;
    orc_ct_desc d{o->first_limbs, 2, 0, 1.0, 1};
    Ct x = make_ct(o, &d, a), y = make_ct(o, &d, b);
    auto work = [&](int n) {
        Eval ev(o);
        for (int i = 0; i < n; i++) { Ct z = x; ev.multiply(z, y); ev.relinearize(z); }
    };
    auto t0 = std::chrono::steady_clock::now();
    if (threads <= 1) work(count);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < threads; i++) th.emplace_back(work, count / threads + (i < count % threads ? 1 : 0));
        for (auto &t : th) t.join();
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
double orc_time_ntt(void *h, int prime_idx, const uint64_t *limb, int count) {
    Orc *o = (Orc *)h;
    std::vector<u64> buf(limb, limb + o->N);
    const NttTable &t = *o->key_tables[prime_idx];
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < count; i++) { ntt_fwd_lazy(buf.data(), t); for (auto &v : buf) { if (v >= 2 * t.m.p) v -= 2 * t.m.p; if (v >= t.m.p) v -= t.m.p; } ntt_inv_lazy(buf.data(), t); }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

} // extern "C"
