// context.cpp -- see context.h.
#include "context.h"
#include "fpmod.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace troyhip {

Arena::~Arena() {
    if (base_) (void)hipFree(base_);
    for (u64 *p : retired_) (void)hipFree(p);
}
void Arena::reserve(size_t count) {
    if (count <= cap_) return;
    // grow-only; the old block may still be referenced by work already queued on the stream, so it is
    // retired (freed with the context) instead of being freed here
    size_t want = std::max(count, cap_ * 2);
    u64 *nb = nullptr;
    HIP_CHECK(hipMalloc((void **)&nb, want * sizeof(u64)));
    if (base_) retired_.push_back(base_);
    base_ = nb;
    cap_ = want;
    used_ = 0;
}
u64 *Arena::take(size_t count) {
    count = (count + 31) & ~size_t(31); // keep 256-byte alignment
    if (used_ + count > cap_) {
        if (used_ != 0) throw Error(ST_LOGIC_ERROR, "scratch arena exhausted inside an op (reserve() must be called first)");
        reserve(count);
    }
    u64 *p = base_ + used_;
    used_ += count;
    return p;
}

// ---- matrix-core operand packing (behz.hip) ----
// A 61-bit entry M is written in eight BALANCED base-256 digits d_j in [-128, 127], M = sum d_j 2^(8j):
// d = bytes of (M + 0x8080..80) xor 0x80 each (adding 128 to every byte position propagates the carries, the xor
// recentres every byte).  Row (o, s) of the Toeplitz form holds, at k = (limb, i), the digit d_{s-i} of M[o][limb], so that
// (row . digits of y) = the coefficient of 2^(8s) in sum_limb y_limb * M[o][limb].
static inline int8_t balanced_digit(u64 m, int j) {
    if (j < 0 || j > 7) return 0;
    const u64 c80 = 0x8080808080808080ull;
    return (int8_t)((((m + c80) ^ c80) >> (8 * j)) & 0xFF);
}
// entries[o * 16 + limb]; fragment layout [row-block][k-block][lane][16 bytes]: lane = (tile row m = lane % 32, k half = lane / 32);
// tile row m -> output 2 rb + (m / 4) % 2, shift (m / 8) * 4 + m % 4  (so that lane l's accumulator register r is shift r of
// output 2 rb + l / 32, see the D layout of the instruction)
// bias (optional): limb `bias_slot` is the constant input 1 (digit 0 = 1); its column holds the 16 balanced digits of bias[o], one per shift
typedef unsigned __int128 u128;

// ---- second matrix-core form (behz2.hip).  W[o][limb] are the conversion-matrix entries of output o (already holding every folded
// row factor); A-matrix row (o, s) holds at k = (limb, i) the balanced digit s of W[o][limb] 2^(8 i) mod p[o].  Tile row m of
// row-block rb -> output 4 rb + 2 (m / 16) + (m / 4) % 2, shift m % 4 + 4 ((m / 8) % 2): lane l's accumulator registers 0-7 are the
// shifts of output 4 rb + l / 32, registers 8-15 those of output 4 rb + 2 + l / 32 (D layout of v_mfma_i32_32x32x32_i8).
// mod32: the modulus is 2^32 (m_tilde) and only shifts 0-3 exist.
static std::vector<uint8_t> pack_rows8(const std::vector<std::vector<u64>> &W, const std::vector<u64> &p, int KB, bool mod32 = false, bool half_major = false) {
    const int n_out = (int)W.size(), RB = (n_out + 3) / 4;
    std::vector<uint8_t> f((size_t)RB * KB * 64 * 16, 0);
    for (int rb = 0; rb < RB; rb++)
        for (int kb = 0; kb < KB; kb++)
            for (int lane = 0; lane < 64; lane++) {
                // half_major: output 4 rb + 2 (half) + (register group) -- the order in which a lane's results are the limbs of its next operand
                const int m = lane % 32, mm = m % 16, s = mm % 4 + 4 * (mm / 8);
                const int o = half_major ? 4 * rb + 2 * ((mm / 4) % 2) + m / 16 : 4 * rb + 2 * (m / 16) + (mm / 4) % 2;
                if (o >= n_out || (mod32 && s >= 4)) continue;
                for (int t = 0; t < 16; t++) {
                    const int k = 32 * kb + 16 * (lane / 32) + t, limb = k / 8, i = k % 8;
                    if (limb >= (int)W[o].size()) continue;
                    u64 w;
                    if (mod32) w = i < 4 ? (W[o][limb] << (8 * i)) & 0xFFFFFFFFull : 0;
                    else w = host::mul_mod(W[o][limb] % p[o], host::pow_mod(2, (u64)(8 * i), p[o]), p[o]);
                    f[(((size_t)rb * KB + kb) * 64 + lane) * 16 + t] = (uint8_t)balanced_digit(w, s);
                }
            }
    return f;
}
static BehzK2 make_k2(u64 p) {
    const int b = 64 - __builtin_clzll(p), nd = std::min(8, b / 8 + 1); // balanced digits 0 .. nd-1 of a residue can be non-zero
    BehzK2 k;
    k.p = p;
    k.negp = 0 - p;
    // the high half sum_{s>=4} C_s 2^(8(s-4)) is below 2^(21.4 + 8 (nd - 5)) in magnitude, the low half below 2^45.5
    k.bias1 = u64(1) << (nd >= 5 ? 8 * (nd - 5) + 22 : 22);
    const u128 hi = (u128)k.bias1 << 32, target = hi + ((u128)1 << 46);
    const u128 m = (target + p - 1) / p;
    k.biaslo = (u64)(m * p - hi); // in [2^46, 2^46 + p)
    const int sh = b - 2;         // 2^sh / p < 1/2: the estimate is at most one below the quotient
    k.sh32 = sh >= 32 ? (u32)(sh - 32) : 0;
    k.mu = (u32)std::min<u128>(((u128)1 << (sh + 32)) / p, 0xFFFFFFFFu);
    return k;
}
// probe builds: TROYHIP_BEHZ=valu keeps the matrix-core tables unbuilt (the VALU kernels then run at every size)
static bool behz_v2_enabled() {
    const char *e = probe_env("TROYHIP_BEHZ");
    return !(e && e[0] == 'v');
}

template <class T> T *Context::upload(const std::vector<T> &v, std::vector<void *> &owner) {
    if (v.empty()) return nullptr;
    T *d = nullptr;
    HIP_CHECK(hipMalloc((void **)&d, v.size() * sizeof(T)));
    owner.push_back(d);
    HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

int Context::register_prime(u64 p) {
    for (size_t i = 0; i < primes.size(); i++) if (primes[i] == p) return (int)i;
    if (primes.size() >= 255) throw Error(ST_LOGIC_ERROR, "too many primes");
    primes.push_back(p);
    tables.emplace_back();
    tables.back().build(logn, p);
    return (int)primes.size() - 1;
}
int Context::prime_id(u64 p) const {
    for (size_t i = 0; i < primes.size(); i++) if (primes[i] == p) return (int)i;
    throw Error(ST_INVALID_ARGUMENT, "prime is not part of the context");
}

Context::Context(int scheme_, u64 N_, const std::vector<u64> &key_primes, u64 t_, bool with_device)
    : has_device(with_device), scheme(scheme_), N(N_), t(scheme_ == SCHEME_CKKS ? 0 : t_) {
    if (scheme < SCHEME_BFV || scheme > SCHEME_BGV) throw Error(ST_INVALID_ARGUMENT, "unsupported scheme");
    if (N < 2 || N > 131072 || (N & (N - 1))) throw Error(ST_INVALID_ARGUMENT, "poly_modulus_degree is invalid");
    logn = 63 - __builtin_clzll(N);
    K = (int)key_primes.size();
    if (K < 1 || K > 64) throw Error(ST_INVALID_ARGUMENT, "coeff_modulus size is invalid");
    for (u64 p : key_primes) {
        if ((p >> 60) || p < 2) throw Error(ST_INVALID_ARGUMENT, "coeff_modulus primes must be at most 60 bits");
        if (std::count(key_primes.begin(), key_primes.end(), p) != 1) throw Error(ST_INVALID_ARGUMENT, "coeff_modulus primes must be distinct");
        if (!host::is_prime(p) || (p - 1) % (2 * N)) throw Error(ST_INVALID_ARGUMENT, "coeff_modulus primes must be NTT-friendly primes");
        register_prime(p);
    }
    if (scheme == SCHEME_CKKS && t_ != 0) throw Error(ST_INVALID_ARGUMENT, "plain_modulus must be zero"); // src/context.cpp:353-361 (invalid_plain_modulus_nonzero)
    if (scheme != SCHEME_CKKS) {
        if (t < 2 || (t >> 60)) throw Error(ST_INVALID_ARGUMENT, "plain_modulus is invalid");
        for (u64 p : key_primes) if (p % t == 0 || t % p == 0) throw Error(ST_INVALID_ARGUMENT, "plain_modulus must be coprime to coeff_modulus");
    }
    // modulus-switching chain (src/context.cpp:426-531): key level, then drop the last prime while valid
    first_limbs = K > 1 ? K - 1 : K;
    last_limbs = 0;
    for (int limbs = K; limbs >= 1; limbs--) {
        std::vector<u64> q(key_primes.begin(), key_primes.begin() + limbs);
        if (scheme != SCHEME_CKKS && host::bit_length_of_product(q) <= 64) {
            u128 prod = 1;
            for (u64 p : q) prod *= p;
            if (prod <= (u128)t) break; // plain modulus must stay below the coefficient modulus
        }
        build_level(limbs);
        last_limbs = limbs;
    }
    if (!has_level(first_limbs)) throw Error(ST_INVALID_ARGUMENT, "encryption parameters are not valid");
    if (has_device) {
        if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); device = 0; }
        upload_tables();
    }
}

// Small launches take the merged forms (one launch over the q-base and the B_sk-base rows of a product, one first pass for a mod-down: evaluator.cpp).
// Measured and NOT kept for them (profiles/r04_small_batch.txt): the two bases side by side on a companion stream -- a cross-stream event wait costs
// 8-12 us on this runtime, more than the 15 us kernels it would overlap (B = 1: 0.256 -> 0.279 ms); the digits of a key-switch group shared by two
// workgroups that add their partial sums atomically -- device-scope 64-bit atomics cost far more than the halved row chain saves (0.210 -> 0.275 ms).
bool Context::small_launch(u64 rows, unsigned per_cu) const { return has_device && ((rows * N) >> 11) < (u64)per_cu * device_cus(); }

Context::~Context() {
    for (void *p : dev_allocs_) (void)hipFree(p);
    for (auto &kv : levels) for (void *p : kv.second.dev_blocks) (void)hipFree(p);
}

const Level &Context::level(int limbs) const {
    auto it = levels.find(limbs);
    if (it == levels.end()) throw Error(ST_INVALID_ARGUMENT, "no such level in the modulus chain");
    return it->second;
}

// probe builds: TROYHIP_BFLY=guarded -- no prime is treated as "lean" (A/B runs of the guard-free butterflies)
static bool lean_allowed() {
    static const bool on = [] { const char *e = probe_env("TROYHIP_BFLY"); return !(e && std::strcmp(e, "guarded") == 0); }();
    return on;
}
// TROYHIP_FP64=off: no prime takes the FP64 instances (tests / A-B runs against the integer kernels)
static bool fp_allowed() {
    static const bool on = [] { const char *e = std::getenv("TROYHIP_FP64"); return !(e && std::strcmp(e, "off") == 0); }();
    return on;
}
static bool fp_prime(u64 p) { return fp_allowed() && p < (u64(1) << TROY_FP_MAX_BITS); }
// the FP64 twins of the two scalar constants of a descriptor; call again whenever inv_n / iroot_last_scaled are rescaled
static void set_fp_consts(PrimeDesc &d) {
    auto pair = [&](const Shoup &w) { return Shoup{fp_bits((double)w.op), fp_bits((double)w.op / (double)d.p)}; };
    d.inv_n_fp = d.root_fp ? pair(d.inv_n) : Shoup{0, 0};
    d.iroot_last_scaled_fp = d.root_fp ? pair(d.iroot_last_scaled) : Shoup{0, 0};
}
LimbMap Context::ct_map(int limbs) const {
    LimbMap m;
    std::memset(&m, 0, sizeof(m));
    for (int i = 0; i < limbs; i++) m.id[i] = (uint8_t)i;
    m.period = (uint32_t)limbs;
    m.inner = 1;
    for (int i = 0; lean_allowed() && i < limbs && i < 64; i++) m.lean |= (u64)(primes[i] >= (u64(1) << 33) && primes[i] < (u64(1) << 58)) << i;
    for (int i = 0; i < limbs && i < 64; i++) m.fp |= (u64)fp_prime(primes[i]) << i;
    m.host_primes = primes.data();
    return m;
}
LimbMap Context::ids_map(const std::vector<uint8_t> &ids, uint32_t inner) const {
    LimbMap m;
    std::memset(&m, 0, sizeof(m));
    if (ids.size() > 64) throw Error(ST_LOGIC_ERROR, "limb map too long");
    for (size_t i = 0; i < ids.size(); i++) m.id[i] = ids[i];
    m.period = (uint32_t)ids.size();
    m.inner = inner;
    for (size_t i = 0; lean_allowed() && i < ids.size(); i++) m.lean |= (u64)(primes[ids[i]] >= (u64(1) << 33) && primes[ids[i]] < (u64(1) << 58)) << i;
    for (size_t i = 0; i < ids.size(); i++) m.fp |= (u64)fp_prime(primes[ids[i]]) << i;
    m.host_primes = primes.data();
    return m;
}
LimbMap Context::single_map(int id) const { return ids_map({(uint8_t)id}); }

void Context::build_level(int limbs) {
    Level &lv = levels[limbs];
    lv.limbs = limbs;
    std::vector<u64> q(primes.begin(), primes.begin() + limbs);
    // TROYHIP_AUX_BASE=reference: the reference's 61-bit auxiliary primes (tests: tables against the reference, results against the default)
    static const bool aux_auto = [] { const char *e = std::getenv("TROYHIP_AUX_BASE"); return !(e && std::strcmp(e, "reference") == 0); }();
    std::vector<u64> key_primes(primes.begin(), primes.begin() + K);
    lv.rns.build(N, q, t, aux_auto && scheme == SCHEME_BFV, key_primes);
    for (u64 p : lv.rns.Bsk) lv.bsk_ids.push_back((uint8_t)register_prime(p));
}

void Context::upload_tables() {
    // NTT tables
    h_desc.resize(primes.size());
    for (size_t i = 0; i < primes.size(); i++) {
        const host::NttTable &tb = tables[i];
        Mod m = make_mod(tb.p);
        PrimeDesc &d = h_desc[i];
        d.p = m.p; d.cr0 = m.cr0; d.cr1 = m.cr1; d.two_p = 2 * m.p;
        d.inv_n = tb.inv_n;
        d.r64 = make_shoup((u64)((((u128)1) << 64) % m.p), m.p);
        d.iroot_last_scaled = tb.iroot_last_scaled;
        d.root = upload(tb.root, dev_allocs_);
        d.iroot = upload(tb.iroot, dev_allocs_);
        d.aux = Shoup{0, 0};
        d.root_fp = d.iroot_fp = nullptr;
        d.root_w = d.iroot_w = nullptr;
        if (fp_prime(tb.p)) { // the same twiddles as pairs of doubles (w, w / p): both exact / correctly rounded on the host (fpmod.h)
            auto to_fp = [&](const std::vector<Shoup> &v) {
                std::vector<Shoup> f(v.size());
                for (size_t j = 0; j < v.size(); j++) f[j] = Shoup{fp_bits((double)v[j].op), fp_bits((double)v[j].op / (double)tb.p)};
                return f;
            };
            d.root_fp = upload(to_fp(tb.root), dev_allocs_);
            d.iroot_fp = upload(to_fp(tb.iroot), dev_allocs_);
            std::vector<u64> rw(tb.root.size()), iw(tb.iroot.size(), 0); // compact w-only tables (device_types.h); the inverse one shifted by one entry
            for (size_t j = 0; j < tb.root.size(); j++) rw[j] = fp_bits((double)tb.root[j].op);
            for (size_t j = 0; j + 1 < tb.iroot.size(); j++) iw[j] = fp_bits((double)tb.iroot[j + 1].op);
            d.root_w = upload(rw, dev_allocs_);
            d.iroot_w = upload(iw, dev_allocs_);
        }
        set_fp_consts(d);
    }
    d_desc = upload(h_desc, dev_allocs_);
    if (scheme == SCHEME_CKKS) { // per-slot inverses of the prime a divide-and-round drops, for the fused correction transform (ntt1.hip)
        if (K >= 2) {
            std::vector<Shoup> v;
            for (int j = 0; j + 1 < K; j++) v.push_back(make_shoup(level(K).rns.inv_q_last_mod_q[j], primes[j]));
            d_inv_qk = upload(v, dev_allocs_);
        }
        for (auto &kv : levels) {
            Level &lv = kv.second;
            if (lv.limbs < 2) continue;
            std::vector<Shoup> v;
            for (int l = 0; l + 1 < lv.limbs; l++) v.push_back(make_shoup(lv.rns.inv_q_last_mod_q[l], primes[l]));
            lv.d_inv_qlast = upload(v, lv.dev_blocks);
        }
    }
    if ((scheme == SCHEME_BFV || scheme == SCHEME_BGV) && K >= 2) {
        std::vector<PrimeDesc> md = h_desc;
        const host::RnsLevel &kr = level(K).rns;
        for (int j = 0; j + 1 < K; j++) {
            PrimeDesc &d = md[j];
            const u64 f = kr.inv_q_last_mod_q[j] % d.p;
            d.inv_n = make_shoup(host::mul_mod(d.inv_n.op, f, d.p), d.p);
            d.iroot_last_scaled = make_shoup(host::mul_mod(d.iroot_last_scaled.op, f, d.p), d.p);
            d.aux = make_shoup(f, d.p);
            set_fp_consts(d);
        }
        d_desc_md = upload(md, dev_allocs_);
    }

    if (scheme != SCHEME_BFV) return;
    // BEHZ constants, per data level
    for (auto &kv : levels) {
        Level &lv = kv.second;
        if (!is_data_level(lv.limbs)) continue;
        const host::RnsLevel &r = lv.rns;
        const int L = lv.limbs, nB = (int)r.B.size(), nBsk = (int)r.Bsk.size();
        auto c = std::make_shared<BehzDev>();
        std::memset(c.get(), 0, sizeof(BehzDev));
        c->L = L; c->nB = nB; c->nBsk = nBsk;
        for (int l = 0; l < L; l++) c->q_id[l] = (uint8_t)l;
        for (int o = 0; o < nBsk; o++) c->bsk_id[o] = lv.bsk_ids[o];
        // Everything that multiplies a whole base-conversion row is folded into the matrix on the host, so the device does
        // ONE reduction per output (the results are the same canonical residues: same value modulo the same prime):
        //   extension  out_o = (sum_l y_l (q/q_l) + q r) m_tilde^-1        ->  E[o][l] = (q/q_l) m_tilde^-1,  ext_q[o] = q m_tilde^-1
        //   fastFloor  u_o   = (t db_o - sum_l y_l (q/q_l)) q^-1 [Bpre_o]  ->  F[o][l] = -(q/q_l) q^-1 [Bpre_o], T[o] = t q^-1 [Bpre_o]
        std::vector<Shoup> ext_pre(L), floor_pre(L);
        auto split3 = [](u64 m) { return Mat3{(u32)(m & 0x1FFFFF), (u32)((m >> 21) & 0x1FFFFF), (u32)(m >> 42), 0}; };
        std::vector<Mat3> ext_mat((size_t)nBsk * L), floor_mat((size_t)nBsk * L), floor_t(nBsk), B2q((size_t)L * nB), B2msk(nB);
        std::vector<u32> mt_row(L);
        std::vector<u64> ext_q(nBsk);
        for (int l = 0; l < L; l++) {
            u64 ql = r.q[l], ip = r.q_to_Bsk.inv_punct[l];
            ext_pre[l] = make_shoup(host::mul_mod(r.m_tilde % ql, ip, ql), ql);
            floor_pre[l] = make_shoup(host::mul_mod(t % ql, ip, ql), ql);
            mt_row[l] = (u32)r.q_to_mtilde.mat[0][l];
            for (int b = 0; b < nB; b++) B2q[(size_t)l * nB + b] = split3(r.B_to_q.mat[l][b]);
        }
        for (int o = 0; o < nBsk; o++) {
            const u64 p = r.Bsk[o];
            const u64 inv_mt = r.inv_mtilde_mod_Bsk[o];
            u64 fscale = r.inv_prod_q_mod_Bsk[o];                                      // q^-1
            if (o < nB) fscale = host::mul_mod(fscale, r.B_to_q.inv_punct[o], p);     // ... * (B/B_o)^-1 for the B part
            ext_q[o] = host::mul_mod(r.prod_q_mod_Bsk[o], inv_mt, p);
            floor_t[o] = split3(host::mul_mod(t % p, fscale, p));
            for (int l = 0; l < L; l++) {
                const u64 m = r.q_to_Bsk.mat[o][l];
                ext_mat[(size_t)o * L + l] = split3(host::mul_mod(m, inv_mt, p));
                const u64 f = host::mul_mod(m, fscale, p);
                floor_mat[(size_t)o * L + l] = split3(f ? p - f : 0);
            }
        }
        for (int b = 0; b < nB; b++) B2msk[b] = split3(r.B_to_msk.mat[0][b]);
        c->ext_pre = upload(ext_pre, lv.dev_blocks);
        c->ext_mat3 = upload(ext_mat, lv.dev_blocks);
        c->ext_mt_row = upload(mt_row, lv.dev_blocks);
        c->neg_inv_q_mod_mt = r.neg_inv_prod_q_mod_mtilde;
        c->ext_q = upload(ext_q, lv.dev_blocks);
        c->floor_pre = upload(floor_pre, lv.dev_blocks);
        c->floor_mat3 = upload(floor_mat, lv.dev_blocks);
        c->floor_t3 = upload(floor_t, lv.dev_blocks);
        // the one-step quotient estimate of the behz2 epilogues holds for every output prime >= 2^33 (behz2.hip, header); the auxiliary primes are
        // 61 bits in the reference's base and 58 or 50 bits in the library's own (RnsLevel::build)
        bool big_bsk = true;
        for (u64 p : r.Bsk) big_bsk = big_bsk && p >= (u64(1) << 33);
        if (L <= 15 && nBsk <= 16 && big_bsk && behz_v2_enabled()) { // second matrix-core form: behz2.hip
            const int KBx = (L + 1 + 3) / 4, KB1 = (L + 3) / 4, KB2 = (nB + 1 + 3) / 4;
            std::vector<std::vector<u64>> xw(nBsk), f1w(nBsk), f2w(L), mtw(2), mskw(2);
            std::vector<u64> bskp(r.Bsk.begin(), r.Bsk.end()), qp(r.q.begin(), r.q.begin() + L), two32(2, 0), mskp(2, r.m_sk);
            std::vector<BehzK2> xk(nBsk), f1k(nBsk), f2k(L);
            std::vector<Shoup> f1t(nBsk); // t q^-1 [(B/B_o)^-1 | B^-1] mod Bsk_o, the factor of the db_o term: rides on the inverse transform (floor_desc below)
            for (int o = 0; o < nBsk; o++) {
                const u64 p = r.Bsk[o];
                u64 fscale = r.inv_prod_q_mod_Bsk[o];                                               // q^-1
                if (o < nB) fscale = host::mul_mod(fscale, r.B_to_q.inv_punct[o], p);              // (B/B_o)^-1: u_o is pre-scaled for stage 2
                else fscale = host::mul_mod(fscale, r.inv_prod_B_mod_msk % p, p);                  // m_sk: z' = z_sk B^-1
                for (int l = 0; l < L; l++) {
                    xw[o].push_back(host::mul_mod(r.q_to_Bsk.mat[o][l], r.inv_mtilde_mod_Bsk[o], p));
                    const u64 f = host::mul_mod(r.q_to_Bsk.mat[o][l], fscale, p);
                    f1w[o].push_back(f ? p - f : 0);
                }
                xw[o].push_back(ext_q[o]);                                                          // the centred r against q m_tilde^-1
                xk[o] = f1k[o] = make_k2(p);
                f1t[o] = make_shoup(host::mul_mod(t % p, fscale, p), p);
            }
            for (int l = 0; l < L; l++) {
                for (int b = 0; b < nB; b++) f2w[l].push_back(r.B_to_q.mat[l][b]);
                const u64 pb = r.prod_B_mod_q[l] % r.q[l];
                f2w[l].push_back(pb ? r.q[l] - pb : 0);                                             // the centred alpha against -(B mod q_l)
                f2k[l] = make_k2(r.q[l]);
            }
            for (int h = 0; h < 2; h++) {
                for (int l = 0; l < L; l++) mtw[h].push_back(mt_row[l]);
                for (int b = 0; b < nB; b++) mskw[h].push_back(host::mul_mod(r.B_to_msk.mat[0][b], r.inv_prod_B_mod_msk, r.m_sk));
            }
            c->x_frag = upload(pack_rows8(xw, bskp, KBx), lv.dev_blocks);
            c->x_mt_frag = upload(pack_rows8(mtw, two32, KBx, true), lv.dev_blocks);
            c->x_k = upload(xk, lv.dev_blocks);
            c->f1_frag = upload(pack_rows8(f1w, bskp, KB1), lv.dev_blocks);
            if (KB2 <= 2) c->f1s_frag = upload(pack_rows8(f1w, bskp, KB1, false, true), lv.dev_blocks);
            c->f1_k = upload(f1k, lv.dev_blocks);
            c->f2_frag = upload(pack_rows8(f2w, qp, KB2), lv.dev_blocks);
            c->f2_msk_frag = upload(pack_rows8(mskw, mskp, KB2), lv.dev_blocks);
            c->f2_k = upload(f2k, lv.dev_blocks);
            c->msk_k = make_k2(r.m_sk);
            {   // prime table for the inverse transforms in front of the floor kernel: N^-1 (and the last twiddle, pre-scaled by N^-1) times the
                // factor the floor step multiplies every input residue by
                std::vector<PrimeDesc> fd = h_desc;
                auto scale = [&](PrimeDesc &d, u64 f) {
                    d.inv_n = make_shoup(host::mul_mod(d.inv_n.op, f % d.p, d.p), d.p);
                    d.iroot_last_scaled = make_shoup(host::mul_mod(d.iroot_last_scaled.op, f % d.p, d.p), d.p);
                    set_fp_consts(d);
                };
                for (int l = 0; l < L; l++) scale(fd[l], floor_pre[l].op);
                for (int o = 0; o < nBsk; o++) scale(fd[lv.bsk_ids[o]], f1t[o].op);
                c->floor_desc = upload(fd, lv.dev_blocks);
            }
            c->f2_fast = 1;
            for (int l = 0; l < L; l++) c->f2_fast = c->f2_fast && r.q[l] >= (u64(1) << 33);
            c->v2 = 1;
            // register-resident FP64 form for small bases of narrow primes (behz3.hip): the same folded rows as pairs of doubles (w, w / p)
            bool narrow = fp_allowed() && L <= 6;
            for (int l = 0; l < L; l++) narrow = narrow && r.q[l] >= (u64(1) << 33) && !(r.q[l] >> TROY_FP_MAX_BITS);
            for (u64 p : r.Bsk) narrow = narrow && !(p >> TROY_FP_MAX_BITS);
            if (narrow) {
                auto pair = [](std::vector<double> &v, u64 w, u64 p) { v.push_back((double)w); v.push_back((double)w / (double)p); };
                auto prime = [](std::vector<double> &v, u64 p) { v.push_back((double)p); v.push_back(1.0 / (double)p); };
                std::vector<double> ex, fl;
                for (int l = 0; l < L; l++) pair(ex, ext_pre[l].op, r.q[l]);
                for (int l = 0; l < L; l++) prime(ex, r.q[l]);
                for (int o = 0; o < nBsk; o++) prime(ex, r.Bsk[o]);
                for (int o = 0; o < nBsk; o++)
                    for (int l = 0; l <= L; l++) pair(ex, xw[o][l], r.Bsk[o]);          // E[o][0 .. L-1], C[o]
                for (int l = 0; l < L; l++) prime(fl, r.q[l]);
                for (int o = 0; o < nBsk; o++) prime(fl, r.Bsk[o]);
                for (int o = 0; o < nBsk; o++)
                    for (int l = 0; l < L; l++) pair(fl, f1w[o][l], r.Bsk[o]);
                for (int b = 0; b < nB; b++) pair(fl, mskw[0][b], r.m_sk);
                for (int l = 0; l < L; l++)
                    for (int b = 0; b <= nB; b++) pair(fl, f2w[l][b], r.q[l]);          // G[l][0 .. nB-1], H[l]
                c->fp_ext = upload(ex, lv.dev_blocks);
                c->fp_floor = upload(fl, lv.dev_blocks);
            }
        }
        c->B2q3 = upload(B2q, lv.dev_blocks);
        c->B2msk3 = upload(B2msk, lv.dev_blocks);
        c->inv_B_mod_msk = make_shoup(r.inv_prod_B_mod_msk, r.m_sk);
        c->prod_B_mod_q = upload(r.prod_B_mod_q, lv.dev_blocks);
        lv.behz = c;
    }
}

} // namespace troyhip
