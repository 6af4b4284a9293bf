// hostmath.cpp -- see hostmath.h.  Plain 128-bit integer arithmetic on the host; none of this is on
// the hot path (it runs once per context).
#include "hostmath.h"
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <map>

namespace troyhip {
namespace host {

u64 mul_mod(u64 a, u64 b, u64 p) { return (u64)(((u128)a * b) % p); }

u64 pow_mod(u64 a, u64 e, u64 p) {
    u64 r = 1 % p;
    a %= p;
    for (; e; e >>= 1) {
        if (e & 1) r = mul_mod(r, a, p);
        a = mul_mod(a, a, p);
    }
    return r;
}

bool inv_mod(u64 a, u64 p, u64 &out) {
    a %= p;
    if (!a) return false;
    __int128 old_r = p, r = a, old_s = 0, s = 1;
    while (r) {
        __int128 q = old_r / r, tmp = old_r - q * r;
        old_r = r; r = tmp;
        tmp = old_s - q * s; old_s = s; s = tmp;
    }
    if (old_r != 1) return false;
    out = (u64)(old_s < 0 ? old_s + p : old_s);
    return true;
}
u64 inv_mod_checked(u64 a, u64 p) {
    u64 r;
    if (!inv_mod(a, p, r)) throw Error(ST_LOGIC_ERROR, "invalid rns bases");
    return r;
}

// Deterministic Miller-Rabin for 64-bit inputs (the reference uses base 2 + random bases,
// numth.cpp:162-247; both accept exactly the primes).
bool is_prime(u64 v) {
    static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (v < 2) return false;
    for (u64 s : small) {
        if (v == s) return true;
        if (v % s == 0) return false;
    }
    u64 d = v - 1;
    int r = 0;
    while (!(d & 1)) { d >>= 1; r++; }
    for (u64 a : small) {
        u64 x = pow_mod(a, d, v);
        if (x == 1 || x == v - 1) continue;
        bool composite = true;
        for (int i = 1; i < r && composite; i++) {
            x = mul_mod(x, x, v);
            if (x == v - 1) composite = false;
        }
        if (composite) return false;
    }
    return true;
}

std::vector<u64> get_primes(u64 factor, int bits, size_t count) { // numth.cpp:261-285
    std::vector<u64> out;
    u64 v = ((u64(1) << bits) - 1) / factor * factor + 1, lower = u64(1) << (bits - 1);
    for (; count && v > lower; v -= factor)
        if (is_prime(v)) { out.push_back(v); count--; }
    if (count) throw Error(ST_LOGIC_ERROR, "failed to find enough qualifying primes");
    return out;
}

std::vector<u64> coeff_modulus_create(u64 N, const std::vector<int> &bits) { // modulus.cpp:80-121
    if (N < 2 || N > 131072 || (N & (N - 1))) throw Error(ST_INVALID_ARGUMENT, "poly_modulus_degree is invalid");
    std::map<int, size_t> need;
    for (int b : bits) {
        if (b < 2 || b > 60) throw Error(ST_INVALID_ARGUMENT, "bit_sizes is invalid");
        need[b]++;
    }
    std::map<int, std::vector<u64>> pool;
    for (auto &kv : need) pool[kv.first] = get_primes(2 * N, kv.first, kv.second);
    std::vector<u64> out;
    for (int b : bits) { out.push_back(pool[b].back()); pool[b].pop_back(); }
    return out;
}

// Minimum over all primitive degree-th roots of unity mod p (numth.cpp:335-363).
bool minimal_primitive_root(u64 degree, u64 p, u64 &out) {
    if ((p - 1) % degree) return false;
    u64 cofactor = (p - 1) / degree, g = 0;
    for (u64 base = 2; base < 4096 && !g; base++) {
        u64 c = pow_mod(base, cofactor, p);
        if (pow_mod(c, degree / 2, p) == p - 1) g = c;
    }
    if (!g) return false;
    u64 g2 = mul_mod(g, g, p), cur = g, best = g;
    for (u64 i = 0; i < degree / 2; i++) { // odd powers g, g^3, ...
        best = std::min(best, cur);
        cur = mul_mod(cur, g2, p);
    }
    out = best;
    return true;
}

uint32_t reverse_bits(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

int bit_length_of_product(const std::vector<u64> &v) {
    std::vector<u64> acc{1};
    for (u64 m : v) {
        u64 carry = 0;
        for (auto &w : acc) {
            u128 t = (u128)w * m + carry;
            w = (u64)t;
            carry = (u64)(t >> 64);
        }
        if (carry) acc.push_back(carry);
    }
    return (int)(64 * (acc.size() - 1) + (64 - __builtin_clzll(acc.back())));
}

u64 product_mod(const std::vector<u64> &v, u64 p) {
    u64 r = 1 % p;
    for (u64 x : v) r = mul_mod(r, x % p, p);
    return r;
}

uint32_t galois_elt_from_step(u64 N, int step) { // galois.cpp:44-86
    uint32_t n = (uint32_t)N, m = 2 * n;
    if (step == 0) return m - 1;
    uint32_t pos = (uint32_t)std::abs(step);
    if (pos >= (n >> 1)) throw Error(ST_INVALID_ARGUMENT, "step count too large");
    uint32_t e = step < 0 ? (n >> 1) - pos : pos;
    u64 g = 1;
    while (e--) g = (g * 3) & (m - 1);
    return (uint32_t)g;
}

std::vector<int> naf(int value) { // numth.h:16-36
    std::vector<int> res;
    bool neg = value < 0;
    value = std::abs(value);
    for (int i = 0; value; i++) {
        int zi = (value & 1) ? 2 - (value & 3) : 0;
        value = (value - zi) >> 1;
        if (zi) res.push_back((neg ? -zi : zi) * (1 << i));
    }
    return res;
}

void NttTable::build(int logn_, u64 p_) {
    logn = logn_;
    p = p_;
    size_t n = size_t(1) << logn;
    if (!minimal_primitive_root(2 * n, p, psi)) throw Error(ST_INVALID_ARGUMENT, "invalid modulus");
    u64 ipsi = inv_mod_checked(psi, p);
    root.assign(n, Shoup{0, 0});
    iroot.assign(n, Shoup{0, 0});
    root[0] = make_shoup(1, p);
    iroot[0] = make_shoup(1, p);
    u64 f = psi, b = ipsi;
    for (size_t i = 1; i < n; i++) {
        root[reverse_bits((uint32_t)i, logn)] = make_shoup(f, p);
        iroot[reverse_bits((uint32_t)(i - 1), logn) + 1] = make_shoup(b, p);
        f = mul_mod(f, psi, p);
        b = mul_mod(b, ipsi, p);
    }
    u64 ninv = inv_mod_checked(n % p, p);
    inv_n = make_shoup(ninv, p);
    iroot_last_scaled = make_shoup(mul_mod(iroot[n - 1].op, ninv, p), p);
}

void BaseConv::build(const std::vector<u64> &in_, const std::vector<u64> &out_) {
    in = in_;
    out = out_;
    size_t ni = in.size(), no = out.size();
    inv_punct.resize(ni);
    mat.assign(no, std::vector<u64>(ni));
    for (size_t i = 0; i < ni; i++) {
        std::vector<u64> others;
        for (size_t k = 0; k < ni; k++) if (k != i) others.push_back(in[k]);
        inv_punct[i] = inv_mod_checked(product_mod(others, in[i]), in[i]);
        for (size_t o = 0; o < no; o++) mat[o][i] = product_mod(others, out[o]);
    }
}

// The auxiliary base B u {m_sk} of the BEHZ multiplication is INTERNAL: the product's limbs in base q do not depend on which primes it holds.
// With R' the integer the extension + small Montgomery step produce (a function of the q-residues alone: rns.cpp:805-913), D the integer tensor
// of two such R', the floor step leaves F = floor(t D / q) - alpha' (alpha' from the q-residues of t D alone, rns.cpp:915-960), and the
// Shenoy-Kumaresan step returns F mod q EXACTLY whenever |F| fits the base (rns.cpp:962-1037) -- every residue in the auxiliary base is the residue
// of one of these integers, so any base of pairwise coprime primes, coprime to q, with the reference's size margin gives the same limbs.
// The reference takes 61-bit primes (rns.cpp:610-635); they land in the slowest butterfly class (range guards: 20 instructions).  aux_auto picks,
// when it costs no extra limb under the reference's own size rule, primes below 2^50 (FP64 butterflies, fpmod.h) or below 2^58 (guard-free
// integer butterflies); `exclude` = every key prime (the auxiliary primes must not repeat one).  gamma (decryption only) stays the reference's.
void RnsLevel::build(u64 N, const std::vector<u64> &q_, u64 t_, bool aux_auto, const std::vector<u64> &exclude) {
    q = q_;
    t = t_;
    size_t nq = q.size();
    int t_bits = t ? 64 - __builtin_clzll(t) : 0;
    const int need = 32 + t_bits + bit_length_of_product(q);
    auto count_for = [&](int bits) { size_t n = nq; while (need >= bits * (int)n + bits) n++; return n; };
    size_t nB = nq;
    if (need >= 61 * (int)nq + 61) nB++;                                   // rns.cpp:610-615
    auto aux = get_primes(2 * N, 61, nB + 2);                              // rns.cpp:629-635
    gamma = aux[1];
    aux_bits = 61;
    if (aux_auto) {
        for (int bits : {50, 58}) {
            if (count_for(bits) != nB || (u64(1) << (bits - 1)) <= 2 * N) continue;
            // candidates: not a key prime, not a prime of q, and not the plain modulus either (a batching t of this size class would otherwise
            // become m_sk: harmless for the arithmetic -- no inverse of t modulo the base is ever taken -- but impossible in the reference, whose
            // 61-bit base cannot meet a <= 60-bit t; excluded so that the two bases stay comparable case by case)
            std::vector<u64> pool, pick;
            try { pool = get_primes(2 * N, bits, nB + 2 + exclude.size() + nq); } catch (const std::exception &) { continue; } // too few primes of this size: next class / the reference's base
            for (u64 p : pool)
                if (p != t && std::find(exclude.begin(), exclude.end(), p) == exclude.end() && std::find(q.begin(), q.end(), p) == q.end() && pick.size() < nB + 1) pick.push_back(p);
            if (pick.size() < nB + 1) continue;
            aux.assign(nB + 2, 0);
            aux[0] = pick[0];
            for (size_t i = 0; i < nB; i++) aux[2 + i] = pick[1 + i];
            aux_bits = bits;
            break;
        }
    }
    m_sk = aux[0];
    B.assign(aux.begin() + 2, aux.begin() + 2 + nB);
    Bsk = B;
    Bsk.push_back(m_sk);
    q_to_Bsk.build(q, Bsk);
    q_to_mtilde.build(q, {m_tilde});
    B_to_q.build(B, q);
    B_to_msk.build(B, {m_sk});
    prod_B_mod_q.resize(nq);
    for (size_t i = 0; i < nq; i++) prod_B_mod_q[i] = product_mod(B, q[i]);
    size_t nb = Bsk.size();
    prod_q_mod_Bsk.resize(nb);
    inv_prod_q_mod_Bsk.resize(nb);
    inv_mtilde_mod_Bsk.resize(nb);
    for (size_t i = 0; i < nb; i++) {
        prod_q_mod_Bsk[i] = product_mod(q, Bsk[i]);
        inv_prod_q_mod_Bsk[i] = inv_mod_checked(prod_q_mod_Bsk[i], Bsk[i]);
        inv_mtilde_mod_Bsk[i] = inv_mod_checked(m_tilde % Bsk[i], Bsk[i]);
    }
    inv_prod_B_mod_msk = inv_mod_checked(product_mod(B, m_sk), m_sk);
    neg_inv_prod_q_mod_mtilde = (m_tilde - inv_mod_checked(product_mod(q, m_tilde), m_tilde)) % m_tilde;
    inv_q_last_mod_q.resize(nq - 1);
    for (size_t i = 0; i + 1 < nq; i++) inv_q_last_mod_q[i] = inv_mod_checked(q[nq - 1] % q[i], q[i]);
    if (t) {
        inv_q_last_mod_t = inv_mod_checked(q[nq - 1] % t, t);
        q_last_mod_t = q[nq - 1] % t;
        q_to_tgamma.build(q, {t, gamma});
        inv_gamma_mod_t = inv_mod_checked(gamma % t, t);
        prod_tgamma_mod_q.resize(nq);
        for (size_t i = 0; i < nq; i++) prod_tgamma_mod_q[i] = mul_mod(t % q[i], gamma % q[i], q[i]);
        neg_inv_q_mod_t = (t - inv_mod_checked(product_mod(q, t), t)) % t;
        neg_inv_q_mod_gamma = (gamma - inv_mod_checked(product_mod(q, gamma), gamma)) % gamma;
    }
}


// ---- BLAKE2b, written from RFC 7693 (section 3.2 compression function F, 3.3 padding / finalisation) ----
static inline u64 rotr64(u64 x, int n) { return (x >> n) | (x << (64 - n)); }
void blake2b(void *out, size_t outlen, const void *in, size_t inlen) {
    static const u64 IV[8] = {0x6A09E667F3BCC908ULL, 0xBB67AE8584CAA73BULL, 0x3C6EF372FE94F82BULL, 0xA54FF53A5F1D36F1ULL,
                              0x510E527FADE682D1ULL, 0x9B05688C2B3E6C1FULL, 0x1F83D9ABFB41BD6BULL, 0x5BE0CD19137E2179ULL};
    static const uint8_t SIGMA[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    u64 h[8];
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010000ULL ^ (u64)outlen; // parameter block: digest length, key length 0, fanout 1, depth 1
    const uint8_t *p = (const uint8_t *)in;
    u128 t = 0;
    auto compress = [&](const uint8_t *block, bool last) {
        u64 m[16], v[16];
        for (int i = 0; i < 16; i++) {
            u64 w = 0;
            for (int b = 0; b < 8; b++) w |= (u64)block[8 * i + b] << (8 * b);
            m[i] = w;
        }
        for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
        v[12] ^= (u64)t;
        v[13] ^= (u64)(t >> 64);
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, u64 x, u64 y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr64(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];     v[b] = rotr64(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y; v[d] = rotr64(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];     v[b] = rotr64(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; r++) {
            const uint8_t *s = SIGMA[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    };
    while (inlen > 128) { // every block but the last
        t += 128;
        compress(p, false);
        p += 128;
        inlen -= 128;
    }
    uint8_t last[128] = {0};
    std::memcpy(last, p, inlen);
    t += inlen;
    compress(last, true);
    uint8_t digest[64];
    for (int i = 0; i < 8; i++)
        for (int b = 0; b < 8; b++) digest[8 * i + b] = (uint8_t)(h[i] >> (8 * b));
    std::memcpy(out, digest, outlen);
}
void parms_id(int scheme, u64 N, const std::vector<u64> &primes, int limbs, u64 plain_modulus, u64 out[4]) {
    std::vector<u64> w;
    w.push_back((u64)scheme);
    w.push_back(N);
    for (int i = 0; i < limbs; i++) w.push_back(primes[i]);
    w.push_back(plain_modulus); // Modulus always spans one word (0 for CKKS)
    blake2b(out, 32, w.data(), w.size() * 8);
}

} // namespace host
} // namespace troyhip
