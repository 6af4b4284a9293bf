// hostcrypto.cpp -- CPU-side key generation, encryption and decryption (BASELINE config A "plumbing" row, SURVEY.md
// section 8a-7).  The reference does exactly this work on the CPU too: KeyGeneratorCuda delegates to the CPU
// KeyGenerator (src/keygenerator_cuda.cuh) and config A is "encrypt -> add -> decrypt on troy CPU path".
//
// Semantics follow the reference (src/keygenerator.cpp:120-329, src/utils/rlwe.cpp:21-232, src/encryptor.cpp:88-260,
// src/utils/scalingvariant.cpp:53-93, src/decryptor.cpp:115-371, src/utils/rns.cpp:462-548,1039-1095); the randomness
// source is our own (a ChaCha20 stream keyed by the caller's seed instead of the reference's Blake2xb), so fresh
// ciphertexts are not bit-identical to the reference's but decrypt -- with this file's decryptor and with the
// reference's own Decryptor -- to the same plaintext.  Decryption is deterministic and bit-exact.
// Nothing here touches the GPU; a host-only context (troyhip_context_create_host) is enough.
#include "hostcrypto.h"
#include <map>
#include <memory>
#include <mutex>
#include <cstring>

namespace troyhip {
namespace hostcrypto {

// ------------------------------------------------------------------ ChaCha20 stream (RFC 8439 block function)
static inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
void Rng::refill() {
    uint32_t x[16];
    std::memcpy(x, state, sizeof(x));
#define QR(a, b, c, d) a += b; d ^= a; d = rotl(d, 16); c += d; b ^= c; b = rotl(b, 12); a += b; d ^= a; d = rotl(d, 8); c += d; b ^= c; b = rotl(b, 7);
    for (int i = 0; i < 10; i++) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
#undef QR
    for (int i = 0; i < 16; i++) block[i] = x[i] + state[i];
    if (++state[12] == 0) ++state[13];
    pos = 0;
}
Rng::Rng(u64 seed_lo, u64 seed_hi, u64 stream) {
    static const uint32_t sigma[4] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    std::memcpy(state, sigma, 16);
    const u64 k[4] = {seed_lo, seed_hi, seed_lo ^ 0x9E3779B97F4A7C15ULL, seed_hi ^ 0xD1B54A32D192ED03ULL};
    std::memcpy(state + 4, k, 32);
    state[12] = state[13] = 0;
    state[14] = 0x74726f79u ^ (uint32_t)stream;         // "troy" ^ nonce: independent keystreams for key generation,
    state[15] = 0x68697031u ^ (uint32_t)(stream >> 32); // "hip1"          relin / Galois keys and every encryption
    pos = 16;
}
uint32_t Rng::next32() {
    if (pos >= 16) refill();
    return block[pos++];
}
u64 Rng::next64() { return ((u64)next32() << 32) | next32(); }
u64 Rng::uniform_below(u64 bound) { // rejection sampling, exactly uniform
    const u64 limit = ~u64(0) - (~u64(0) % bound + 1) % bound;
    for (;;) {
        u64 r = next64();
        if (r <= limit) return r % bound;
    }
}

// ------------------------------------------------------------------ host transforms (canonical in / out)
// Harvey butterflies with lazy ranges, as the reference's CPU transform runs them (dwthandler.h:88-372): values stay in [0, 4p) forward / [0, 2p) inverse
// between stages and are reduced once at the end -- same canonical outputs as a fully reduced transform, a third of its time (no branch per butterfly)
static inline u64 lazy_mul(u64 y, const Shoup w, u64 p) { return w.op * y - mulhi64(y, w.quo) * p; } // in [0, 2p) for any 64-bit y
void ntt_forward(u64 *a, const host::NttTable &t) { // src/utils/dwthandler.h:88-204
    const u64 p = t.p, two_p = 2 * p;
    const size_t n = size_t(1) << t.logn;
    for (size_t m = 1, gap = n >> 1; m < n; m <<= 1, gap >>= 1)
        for (size_t i = 0; i < m; i++) {
            const Shoup w = t.root[m + i];
            u64 *x = a + 2 * i * gap, *y = x + gap;
            for (size_t j = 0; j < gap; j++) {
                u64 u = x[j];
                u -= u >= two_p ? two_p : 0;
                const u64 v = lazy_mul(y[j], w, p);
                x[j] = u + v;
                y[j] = u + two_p - v;
            }
        }
    for (size_t j = 0; j < n; j++) { // [0, 4p) -> [0, p)
        u64 v = a[j];
        v -= v >= two_p ? two_p : 0;
        a[j] = v >= p ? v - p : v;
    }
}
void ntt_inverse(u64 *a, const host::NttTable &t) { // dwthandler.h:215-372
    const u64 p = t.p, two_p = 2 * p;
    const size_t n = size_t(1) << t.logn;
    for (size_t m = n >> 1, gap = 1; m >= 1; m >>= 1, gap <<= 1) {
        for (size_t i = 0; i < m; i++) {
            const Shoup w = t.iroot[n - 2 * m + 1 + i];
            u64 *x = a + 2 * i * gap, *y = x + gap;
            for (size_t j = 0; j < gap; j++) {
                const u64 u = x[j], v = y[j]; // both below 2p
                u64 s = u + v;
                s -= s >= two_p ? two_p : 0;
                x[j] = s;
                y[j] = lazy_mul(u + two_p - v, w, p);
            }
        }
    }
    for (size_t j = 0; j < n; j++) {
        const u64 v = lazy_mul(a[j], t.inv_n, p);
        a[j] = v >= p ? v - p : v;
    }
}

namespace {
struct Polys {
    const Context &c;
    size_t N;
    explicit Polys(const Context &ctx) : c(ctx), N(ctx.N) {}
    u64 prime(int l) const { return c.primes[l]; }
    const host::NttTable &table(int l) const { return c.tables[l]; }
    Mod mod(int l) const { return make_mod(c.primes[l]); }
};

// ternary / CBD samplers in RNS form over the first `limbs` key primes (rlwe.cpp:21-110)
void sample_ternary(const Polys &P, Rng &rng, int limbs, u64 *out) {
    for (size_t i = 0; i < P.N; i++) {
        u64 r = rng.uniform_below(3); // 0,1,2 -> -1,0,1
        for (int l = 0; l < limbs; l++) out[l * P.N + i] = r == 0 ? P.prime(l) - 1 : r - 1;
    }
}
void sample_cbd(const Polys &P, Rng &rng, int limbs, u64 *out) { // 21 - 21 coin flips, sigma ~ 3.24
    for (size_t i = 0; i < P.N; i++) {
        u64 bits = rng.next64();
        int noise = __builtin_popcountll(bits & 0x1FFFFF) - __builtin_popcountll((bits >> 21) & 0x1FFFFF);
        for (int l = 0; l < limbs; l++) out[l * P.N + i] = noise < 0 ? P.prime(l) - (u64)(-noise) : (u64)noise;
    }
}
void sample_uniform(const Polys &P, Rng &rng, int limbs, u64 *out) {
    for (int l = 0; l < limbs; l++)
        for (size_t i = 0; i < P.N; i++) out[l * P.N + i] = rng.uniform_below(P.prime(l));
}
// (c0, c1) = (-(a*s + e), a) in NTT form over `limbs` primes: rlwe.cpp:234-331 encryptZeroSymmetric (NTT-form output)
// a_rng: where the uniform polynomial `a` comes from -- the encryption's own stream, or a stream keyed by a public 64-bit seed alone (seeded form)
void encrypt_zero_symmetric_ntt(const Polys &P, Rng &rng, const u64 *sk, int limbs, u64 *c0, u64 *c1, u64 e_scale, Rng *a_rng = nullptr) {
    const size_t N = P.N;
    sample_uniform(P, a_rng ? *a_rng : rng, limbs, c1);
    std::vector<u64> e((size_t)limbs * N);
    sample_cbd(P, rng, limbs, e.data());
    for (int l = 0; l < limbs; l++) {
        const Mod m = P.mod(l);
        ntt_forward(&e[l * N], P.table(l));
        const u64 es = e_scale % m.p;
        for (size_t i = 0; i < N; i++) {
            u64 ee = e_scale == 1 ? e[l * N + i] : mulmod(e[l * N + i], es, m);
            u64 as = mulmod(c1[l * N + i], sk[l * N + i], m);
            c0[l * N + i] = negmod(addmod(as, ee, m.p), m.p);
        }
    }
}
} // namespace

void keygen_secret(const Context &c, Rng &rng, u64 *sk) { // keygenerator.cpp:120-160
    Polys P(c);
    sample_ternary(P, rng, c.K, sk);
    for (int l = 0; l < c.K; l++) ntt_forward(sk + l * c.N, c.tables[l]);
}
void keygen_public(const Context &c, Rng &rng, const u64 *sk, u64 *pk) { // keygenerator.cpp:162-190
    Polys P(c);
    const u64 e_scale = c.scheme == SCHEME_BGV ? c.t : 1; // BGV errors are multiples of t (rlwe.cpp:300-320)
    encrypt_zero_symmetric_ntt(P, rng, sk, c.K, pk, pk + (size_t)c.K * c.N, e_scale);
}
// keygenerator.cpp:294-329 generateOneKswitchKey: key j encrypts q_special * new_key in limb j only
void keygen_kswitch(const Context &c, Rng &rng, const u64 *sk, const u64 *new_key, u64 *out) {
    if (c.K < 2) throw Error(ST_LOGIC_ERROR, "keyswitching is not supported by the context");
    Polys P(c);
    const size_t N = c.N, K = c.K;
    const u64 e_scale = c.scheme == SCHEME_BGV ? c.t : 1;
    for (size_t j = 0; j + 1 < K; j++) {
        u64 *c0 = out + (j * 2) * K * N, *c1 = c0 + K * N;
        encrypt_zero_symmetric_ntt(P, rng, sk, (int)K, c0, c1, e_scale);
        const Mod m = P.mod((int)j);
        const u64 factor = c.primes[K - 1] % m.p;
        for (size_t i = 0; i < N; i++) c0[j * N + i] = addmod(c0[j * N + i], mulmod(new_key[j * N + i], factor, m), m.p);
    }
}
void relin_source(const Context &c, const u64 *sk, u64 *out) { // s^2 (keygenerator.cpp:228-262)
    for (int l = 0; l < c.K; l++) {
        const Mod m = make_mod(c.primes[l]);
        for (size_t i = 0; i < c.N; i++) out[l * c.N + i] = mulmod(sk[l * c.N + i], sk[l * c.N + i], m);
    }
}
void galois_source(const Context &c, const u64 *sk, uint32_t elt, u64 *out) { // sigma_g(s) in NTT form (keygenerator.cpp:264-292, galois.cpp:18-35)
    const size_t N = c.N;
    if (!(elt & 1) || elt >= 2 * N) throw Error(ST_INVALID_ARGUMENT, "Galois element is not valid");
    for (size_t i = 0; i < N; i++) {
        uint32_t rev = host::reverse_bits((uint32_t)(i + N), c.logn + 1);
        u64 raw = (((u64)elt * rev) >> 1) & (N - 1);
        size_t src = host::reverse_bits((uint32_t)raw, c.logn);
        for (int l = 0; l < c.K; l++) out[l * N + i] = sk[l * N + src];
    }
}

// rns.cpp:805-830 / 832-877 / 1097-1140 on one polynomial, host version
static void host_mod_switch(const Context &c, int limbs, u64 *x, bool ntt_form) {
    const size_t N = c.N;
    const host::RnsLevel &r = c.level(limbs).rns;
    const int nl = limbs - 1;
    const u64 qk = c.primes[nl], half = qk >> 1;
    u64 *xl = x + (size_t)nl * N;
    if (c.scheme == SCHEME_BGV) {
        const Mod tm = make_mod(c.t);
        for (int l = 0; l < nl; l++) {
            const Mod m = make_mod(c.primes[l]);
            for (size_t i = 0; i < N; i++) {
                u64 negc = negmod(barrett64(xl[i], tm), tm.p);
                if (r.inv_q_last_mod_t != 1) negc = mulmod(negc, r.inv_q_last_mod_t, tm);
                u64 delta = mulmod(barrett64(negc, m), qk % m.p, m);
                u64 v = submod(submod(x[l * N + i], barrett64(xl[i], m), m.p), delta, m.p);
                x[l * N + i] = mulmod(v, r.inv_q_last_mod_q[l], m);
            }
        }
        return;
    }
    if (ntt_form) ntt_inverse(xl, c.tables[nl]);
    for (size_t i = 0; i < N; i++) xl[i] = addmod(xl[i], half, qk);
    std::vector<u64> tmp(N);
    for (int l = 0; l < nl; l++) {
        const Mod m = make_mod(c.primes[l]);
        const u64 half_mod = half % m.p;
        for (size_t i = 0; i < N; i++) tmp[i] = submod(barrett64(xl[i], m), half_mod, m.p);
        if (ntt_form) ntt_forward(tmp.data(), c.tables[l]);
        for (size_t i = 0; i < N; i++) x[l * N + i] = mulmod(submod(x[l * N + i], tmp[i], m.p), r.inv_q_last_mod_q[l], m);
    }
}

// the message lands in c0: BFV round(q*m/t) (scalingvariant.cpp:53-93), BGV m (scalingvariant.cpp:21-36), CKKS the RNS plaintext
// (encryptor.cpp:232-236)
static void add_message(const Context &c, const Polys &P, const u64 *plain, size_t n_coeffs, int limbs, u64 *ct) {
    const size_t N = c.N;
    if (c.scheme == SCHEME_BFV) { // scalingvariant.cpp:53-93: c0 += round(q * m / t)
        std::vector<u64> q(c.primes.begin(), c.primes.begin() + limbs);
        const u64 t = c.t, q_mod_t = host::product_mod(q, t), thr = (t + 1) >> 1;
        std::vector<u64> deltas(limbs); // floor(q/t) mod q_l = (q - q mod t) / t mod q_l = (-(q mod t)) * t^-1 mod q_l
        for (int l = 0; l < limbs; l++) {
            const Mod m = P.mod(l);
            deltas[l] = mulmod(negmod(q_mod_t % m.p, m.p), host::inv_mod_checked(t % m.p, m.p), m);
        }
        for (size_t i = 0; i < n_coeffs; i++) {
            const u128 num = (u128)plain[i] * q_mod_t + thr;
            const u64 fix = (u64)(num / t);
            for (int l = 0; l < limbs; l++) {
                const Mod m = P.mod(l);
                const u64 v = addmod(mulmod(plain[i] % m.p, deltas[l], m), fix % m.p, m.p);
                ct[l * N + i] = addmod(ct[l * N + i], v, m.p);
            }
        }
    } else if (c.scheme == SCHEME_BGV) { // scalingvariant.cpp:21-36
        for (int l = 0; l < limbs; l++)
            for (size_t i = 0; i < n_coeffs; i++) ct[l * N + i] = addmod(ct[l * N + i], plain[i] % c.primes[l], c.primes[l]);
    } else {
        for (int l = 0; l < limbs; l++)
            for (size_t i = 0; i < N; i++) ct[l * N + i] = addmod(ct[l * N + i], plain[l * N + i], c.primes[l]);
    }
}

// encryptor.cpp:88-260 (asymmetric).  BFV/BGV: plain = n_coeffs coefficients mod t, output [2][first_limbs][N]
// coefficient form.  CKKS: plain = [limbs][N] RNS polynomial in NTT form, output [2][limbs][N] NTT form.
void encrypt(const Context &c, Rng &rng, const u64 *pk, const u64 *plain, size_t n_coeffs, int limbs, u64 *ct) {
    Polys P(c);
    if (c.scheme != SCHEME_CKKS) limbs = c.first_limbs;
    if (!c.is_data_level(limbs)) throw Error(ST_INVALID_ARGUMENT, "plain is not valid for encryption parameters");
    if (c.scheme != SCHEME_CKKS && n_coeffs > c.N) throw Error(ST_INVALID_ARGUMENT, "plain is not valid for encryption parameters");
    encrypt_zero(c, rng, pk, limbs, ct);
    add_message(c, P, plain, n_coeffs, limbs, ct);
}
// Encryptor::encryptZero(parms_id) (encryptor.cpp:88-150 with is_asymmetric == true): any data level of any scheme
void encrypt_zero(const Context &c, Rng &rng, const u64 *pk, int limbs, u64 *ct) {
    Polys P(c);
    const size_t N = c.N, K = c.K;
    const bool ntt_out = c.scheme == SCHEME_CKKS;
    if (!c.is_data_level(limbs)) throw Error(ST_INVALID_ARGUMENT, "parms_id is not valid for encryption parameters");
    // encrypt zero at the level above (one more prime) when it exists, then divide by that prime (encryptor.cpp:118-150)
    const bool has_prev = limbs < (int)K;
    const int el = has_prev ? limbs + 1 : limbs;
    std::vector<u64> u((size_t)el * N), e((size_t)el * N), tmp((size_t)2 * el * N);
    sample_ternary(P, rng, el, u.data());
    for (int l = 0; l < el; l++) ntt_forward(&u[l * N], c.tables[l]);
    for (int j = 0; j < 2; j++) {
        sample_cbd(P, rng, el, e.data());
        for (int l = 0; l < el; l++) {
            const Mod m = P.mod(l);
            u64 *d = &tmp[((size_t)j * el + l) * N];
            const u64 *pkl = pk + ((size_t)j * K + l) * N;
            for (size_t i = 0; i < N; i++) d[i] = mulmod(u[l * N + i], pkl[i], m);
            if (!ntt_out) ntt_inverse(d, c.tables[l]);
            else ntt_forward(&e[l * N], c.tables[l]);
            const u64 ts = c.scheme == SCHEME_BGV ? c.t % m.p : 1;
            for (size_t i = 0; i < N; i++) d[i] = addmod(d[i], ts == 1 ? e[l * N + i] : mulmod(e[l * N + i], ts, m), m.p);
        }
    }
    for (int j = 0; j < 2; j++) {
        u64 *src = &tmp[(size_t)j * el * N];
        if (has_prev) host_mod_switch(c, el, src, ntt_out);
        std::memcpy(ct + (size_t)j * limbs * N, src, sizeof(u64) * limbs * N);
    }
}

// encryptor.cpp:88-148 with is_asymmetric == false + rlwe.cpp:234-345: (c0, c1) = (-(a*s + e) + message, a) sampled directly at
// the level of the plaintext (no switch from the level above); BFV/BGV leave coefficient form, CKKS stays in NTT form
void encrypt_symmetric(const Context &c, Rng &rng, const u64 *sk, const u64 *plain, size_t n_coeffs, int limbs, u64 *ct) {
    Polys P(c);
    if (c.scheme != SCHEME_CKKS) limbs = c.first_limbs;
    if (!c.is_data_level(limbs)) throw Error(ST_INVALID_ARGUMENT, "plain is not valid for encryption parameters");
    if (c.scheme != SCHEME_CKKS && n_coeffs > c.N) throw Error(ST_INVALID_ARGUMENT, "plain is not valid for encryption parameters");
    encrypt_zero_symmetric(c, rng, sk, limbs, ct);
    add_message(c, P, plain, n_coeffs, limbs, ct);
}
// Encryptor::encryptZeroSymmetric(parms_id) (encryptor.cpp:88-148 with is_asymmetric == false): any data level of any scheme
void encrypt_zero_symmetric(const Context &c, Rng &rng, const u64 *sk, int limbs, u64 *ct) {
    Polys P(c);
    const size_t N = c.N;
    const bool ntt_out = c.scheme == SCHEME_CKKS;
    if (!c.is_data_level(limbs)) throw Error(ST_INVALID_ARGUMENT, "parms_id is not valid for encryption parameters");
    u64 *c0 = ct, *c1 = ct + (size_t)limbs * N;
    encrypt_zero_symmetric_ntt(P, rng, sk, limbs, c0, c1, c.scheme == SCHEME_BGV ? c.t : 1);
    if (!ntt_out)
        for (int l = 0; l < limbs; l++) {
            ntt_inverse(c0 + (size_t)l * N, c.tables[l]);
            ntt_inverse(c1 + (size_t)l * N, c.tables[l]);
        }
}

// ---- seeded symmetric ciphertexts.  The reference expands the seed with curand (XORWOW states keyed by seed + index: src/utils/rlwe_cuda.cu:25-31,
// 292-303), a generator that exists only inside that library; here the expander is the ChaCha20 stream keyed by (seed, 0) on its own nonce.  The
// WIRE FORMAT (the seed field and a payload of c0 alone) is the reference's; the seed -> c1 map is each library's own, so seeded blobs are read
// by the library that wrote them -- unseeded blobs are interchangeable.
static Rng seed_stream(u64 a_seed) { return Rng(a_seed, 0, (u64)9 << 32); }
void expand_seed(const Context &c, u64 a_seed, int limbs, u64 *c1) {
    Polys P(c);
    if (!c.is_data_level(limbs)) throw Error(ST_INVALID_ARGUMENT, "parms_id is not valid for encryption parameters");
    if (!a_seed) throw Error(ST_INVALID_ARGUMENT, "the seed of a seeded ciphertext is not zero");
    Rng a = seed_stream(a_seed);
    sample_uniform(P, a, limbs, c1);
    if (c.scheme != SCHEME_CKKS)
        for (int l = 0; l < limbs; l++) ntt_inverse(c1 + (size_t)l * c.N, c.tables[l]);
}
void encrypt_symmetric_seeded(const Context &c, Rng &rng, u64 a_seed, const u64 *sk, const u64 *plain, size_t n_coeffs, int limbs, u64 *ct) {
    Polys P(c);
    const size_t N = c.N;
    if (!a_seed) throw Error(ST_INVALID_ARGUMENT, "the seed of a seeded ciphertext is not zero");
    if (plain && c.scheme != SCHEME_CKKS) limbs = c.first_limbs;
    if (!c.is_data_level(limbs)) throw Error(ST_INVALID_ARGUMENT, plain ? "plain is not valid for encryption parameters" : "parms_id is not valid for encryption parameters");
    if (plain && c.scheme != SCHEME_CKKS && n_coeffs > c.N) throw Error(ST_INVALID_ARGUMENT, "plain is not valid for encryption parameters");
    Rng a = seed_stream(a_seed);
    u64 *c0 = ct, *c1 = ct + (size_t)limbs * N;
    encrypt_zero_symmetric_ntt(P, rng, sk, limbs, c0, c1, c.scheme == SCHEME_BGV ? c.t : 1, &a);
    if (c.scheme != SCHEME_CKKS)
        for (int l = 0; l < limbs; l++) {
            ntt_inverse(c0 + (size_t)l * N, c.tables[l]);
            ntt_inverse(c1 + (size_t)l * N, c.tables[l]);
        }
    if (plain) add_message(c, P, plain, n_coeffs, limbs, ct);
}

// decryptor.cpp:115-371.  ct [size][limbs][N]; BFV/BGV: N plaintext coefficients; CKKS: [limbs][N] RNS plaintext (NTT form)
void decrypt(const Context &c, const u64 *sk, const u64 *ct, int size, int limbs, bool is_ntt, u64 correction_factor, u64 *out) {
    const size_t N = c.N;
    if (!c.is_data_level(limbs) || size < 2) throw Error(ST_INVALID_ARGUMENT, "encrypted is not valid for encryption parameters");
    std::vector<u64> acc((size_t)limbs * N, 0), tmp(N), spow(sk, sk + (size_t)limbs * N);
    for (int i = 1; i < size; i++) {
        for (int l = 0; l < limbs; l++) {
            const Mod m = make_mod(c.primes[l]);
            std::memcpy(tmp.data(), ct + ((size_t)i * limbs + l) * N, sizeof(u64) * N);
            if (!is_ntt) ntt_forward(tmp.data(), c.tables[l]);
            for (size_t k = 0; k < N; k++) acc[l * N + k] = addmod(acc[l * N + k], mulmod(tmp[k], spow[l * N + k], m), m.p);
            if (i + 1 < size)
                for (size_t k = 0; k < N; k++) spow[l * N + k] = mulmod(spow[l * N + k], sk[l * N + k], m);
        }
    }
    for (int l = 0; l < limbs; l++) {
        if (!is_ntt) ntt_inverse(&acc[l * N], c.tables[l]);
        for (size_t k = 0; k < N; k++) acc[l * N + k] = addmod(acc[l * N + k], ct[l * N + k], c.primes[l]);
    }
    const host::RnsLevel &r = c.level(limbs).rns;
    if (c.scheme == SCHEME_CKKS) {
        std::memcpy(out, acc.data(), sizeof(u64) * limbs * N);
    } else if (c.scheme == SCHEME_BFV) { // rns.cpp:1039-1095 decryptScaleAndRound
        const Mod tm = make_mod(c.t), gm = make_mod(r.gamma);
        const host::BaseConv &bc = r.q_to_tgamma;
        std::vector<u64> y(limbs);
        for (size_t k = 0; k < N; k++) {
            u128 st = 0, sg = 0;
            for (int l = 0; l < limbs; l++) {
                const Mod m = make_mod(c.primes[l]);
                u64 v = mulmod(acc[l * N + k], r.prod_tgamma_mod_q[l], m);
                v = mulmod(v, bc.inv_punct[l], m);
                st += (u128)v * bc.mat[0][l];
                sg += (u128)v * bc.mat[1][l];
                if ((l & 7) == 7) { st %= tm.p; sg %= gm.p; }
            }
            u64 vt = mulmod((u64)(st % tm.p), r.neg_inv_q_mod_t, tm);
            u64 vg = mulmod((u64)(sg % gm.p), r.neg_inv_q_mod_gamma, gm);
            u64 d;
            if (vg > (gm.p >> 1)) d = addmod(vt, barrett64(gm.p - vg, tm), tm.p);
            else d = submod(vt, barrett64(vg, tm), tm.p);
            out[k] = d ? mulmod(d, r.inv_gamma_mod_t, tm) : 0;
        }
    } else { // BGV: rns.cpp:462-548 exactConvertArray (double-precision rounding term, same summation order)
        const Mod tm = make_mod(c.t);
        host::BaseConv bc;
        std::vector<u64> q(c.primes.begin(), c.primes.begin() + limbs);
        bc.build(q, {c.t});
        const u64 q_mod_t = host::product_mod(q, c.t);
        u64 fix = 1;
        if (correction_factor != 1 && !host::inv_mod(correction_factor, c.t, fix)) throw Error(ST_LOGIC_ERROR, "invalid correction factor");
        for (size_t k = 0; k < N; k++) {
            double agg = 0.0;
            u128 sum = 0;
            for (int l = 0; l < limbs; l++) {
                const Mod m = make_mod(c.primes[l]);
                const u64 v = bc.inv_punct[l] == 1 ? barrett64(acc[l * N + k], m) : mulmod(acc[l * N + k], bc.inv_punct[l], m);
                agg += (double)v / (double)m.p;
                sum += (u128)v * bc.mat[0][l];
                if ((l & 7) == 7) sum %= tm.p;
            }
            agg += 0.5;
            const u64 rounded = (u64)agg;
            u64 d = submod((u64)(sum % tm.p), mulmod(rounded % tm.p, q_mod_t, tm), tm.p);
            if (correction_factor != 1) d = mulmod(d, fix, tm);
            out[k] = d;
        }
    }
}

// ------------------------------------------------------------------ BatchEncoder (src/batchencoder.cpp)
// slot i of the 2 x (N/2) matrix sits at the bit-reversed position of the exponent 3^i (row 0) / -3^i (row 1) of the primitive 2N-th root
// (populateMatrixRepsIndexMap, batchencoder.cpp:61-81); encode = scatter + inverse negacyclic NTT modulo t, decode = NTT + gather.  The table
// modulo t is the reference's plainNTTTables: minimal primitive root, SEAL order (host::NttTable).
namespace {
struct PlainTables {
    host::NttTable tb;
    std::vector<uint32_t> index_map;
};
const PlainTables &plain_tables(const Context &c) {
    static std::mutex mu;
    static std::map<std::pair<u64, int>, std::unique_ptr<PlainTables>> cache;
    if (c.scheme == SCHEME_CKKS) throw Error(ST_INVALID_ARGUMENT, "unsupported scheme");
    if (c.t < 2 || !host::is_prime(c.t) || (c.t - 1) % (2 * c.N)) throw Error(ST_LOGIC_ERROR, "batching is not enabled for the encryption parameters"); // batchencoder.cpp:93-96
    std::lock_guard<std::mutex> g(mu);
    auto &slot = cache[{c.t, c.logn}];
    if (!slot) {
        slot.reset(new PlainTables);
        slot->tb.build(c.logn, c.t);
        const size_t n = c.N, row = n >> 1, m = n << 1;
        slot->index_map.resize(n);
        u64 pos = 1;
        for (size_t i = 0; i < row; i++) {
            slot->index_map[i] = host::reverse_bits((uint32_t)((pos - 1) >> 1), c.logn);
            slot->index_map[row | i] = host::reverse_bits((uint32_t)((m - pos - 1) >> 1), c.logn);
            pos = (pos * 3) & (m - 1);
        }
    }
    return *slot;
}
} // namespace
void batch_encode(const Context &c, const u64 *values, size_t count, u64 *plain) {
    const PlainTables &pt = plain_tables(c);
    if (count > c.N) throw Error(ST_INVALID_ARGUMENT, "values_matrix size is too large");
    for (size_t i = 0; i < c.N; i++) plain[i] = 0;
    for (size_t i = 0; i < count; i++) {
        plain[pt.index_map[i]] = values[i] % c.t; // the reference stores the word as it is (release build); the residue is what the transform sees
    }
    ntt_inverse(plain, pt.tb);
}
void batch_decode(const Context &c, const u64 *plain, size_t n_coeffs, u64 *values) {
    const PlainTables &pt = plain_tables(c);
    std::vector<u64> tmp(c.N, 0);
    for (size_t i = 0; i < n_coeffs && i < c.N; i++) tmp[i] = plain[i];
    ntt_forward(tmp.data(), pt.tb);
    for (size_t i = 0; i < c.N; i++) values[i] = tmp[pt.index_map[i]];
}

} // namespace hostcrypto
} // namespace troyhip
