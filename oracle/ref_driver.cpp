// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
//
// A thin C-ABI over the reference's own CPU half (namespace troy, umbrella header src/troy_cpu.h),
// compiled against the reference sources where they lie under /root/reference by oracle/Makefile
// into oracle/_ref/libtroyref_driver.so.  It exists to
//   (1) pin our CPU restatement (oracle/troy_oracle.cpp) against the real reference, and
//   (2) generate the golden vectors committed under tests/golden/ (tests/golden/gen_golden.py), and
//   (3) optionally serve as bench.py's cpu_baseline of kind "reference".
// This file contains no reference code: it only *calls* the reference's public API
// (troy::Evaluator, troy::KeyGenerator, troy::util::RNSTool, troy::util::NTTTables ...).
//
// Conventions: a "level" is identified by its number of RNS limbs (key level = K, first data level
// = K-1 when K > 1, ...).  All polynomial buffers are uint64 [size][limbs][N], the reference layout
// (src/ciphertext.h:300-360).

#include "troy_cpu.h"
#include <chrono>
#include <complex>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

using namespace troy;
using troy::util::ConstHostPointer;
using troy::util::HostPointer;
typedef uint64_t u64;

namespace {

struct Ref {
    EncryptionParameters parms{SchemeType::bfv};
    std::unique_ptr<SEALContext> ctx;
    std::unique_ptr<KeyGenerator> keygen;
    std::unique_ptr<Evaluator> ev;
    PublicKey pk;
    RelinKeys rlk;
    GaloisKeys gk;
    bool has_pk = false;
    std::unique_ptr<SecretKey> sk_override;
    size_t N = 0, K = 0;
    std::string err;
};

std::shared_ptr<const SEALContext::ContextData> level_data(Ref *r, size_t limbs) {
    auto cd = r->ctx->keyContextData();
    while (cd && cd->parms().coeffModulus().size() != limbs) cd = cd->nextContextData();
    return cd;
}

// Build a reference Ciphertext at the level with `limbs` primes from a raw buffer.
Ciphertext make_ct(Ref *r, size_t limbs, size_t size, const u64 *data, bool ntt, double scale, u64 cf) {
    auto cd = level_data(r, limbs);
    if (!cd) throw std::invalid_argument("no such level");
    Ciphertext ct;
    ct.resize(*r->ctx, cd->parmsID(), size);
    std::memcpy(ct.data(), data, sizeof(u64) * size * limbs * r->N);
    ct.isNttForm() = ntt;
    ct.scale() = scale;
    ct.correctionFactor() = cf;
    return ct;
}

void export_ct(const Ciphertext &ct, u64 *out) {
    std::memcpy(out, ct.data(), sizeof(u64) * ct.size() * ct.coeffModulusSize() * ct.polyModulusDegree());
}

bool default_ntt(Ref *r) { return r->parms.scheme() == SchemeType::ckks; }

template <class F> int guarded(Ref *r, F f) {
    try {
        f();
        return 0;
    } catch (const std::exception &e) {
        r->err = e.what();
        return -1;
    }
}

} // namespace

extern "C" {

// ---- parameter helpers (src/modulus.cpp:80-121, src/modulus.h:528) ----
int ref_coeff_modulus_create(u64 N, const int *bits, int n, u64 *out) {
    try {
        auto v = CoeffModulus::Create(N, std::vector<int>(bits, bits + n));
        for (int i = 0; i < n; i++) out[i] = v[i].value();
        return 0;
    } catch (...) {
        return -1;
    }
}
u64 ref_plain_batching(u64 N, int bits) { return PlainModulus::Batching(N, bits).value(); }

// modulus.h: Barrett constants of a modulus -> out[0..2] = const_ratio
void ref_modulus_const_ratio(u64 value, u64 *out) {
    Modulus m(value);
    out[0] = m.constRatio()[0];
    out[1] = m.constRatio()[1];
    out[2] = m.constRatio()[2];
}

// scheme: 1 = bfv, 2 = ckks, 3 = bgv (src/encryptionparams.h SchemeType)
void *ref_create(int scheme, u64 N, const u64 *primes, int K, u64 t, u64 seed) {
    auto *r = new Ref();
    try {
        r->parms = EncryptionParameters(scheme == 1 ? SchemeType::bfv : scheme == 2 ? SchemeType::ckks : SchemeType::bgv);
        r->parms.setPolyModulusDegree(N);
        std::vector<Modulus> q;
        for (int i = 0; i < K; i++) q.emplace_back(primes[i]);
        r->parms.setCoeffModulus(q);
        if (scheme != 2) r->parms.setPlainModulus(t);
        PRNGSeed s{};
        for (size_t i = 0; i < s.size(); i++) s[i] = seed * 0x9E3779B97F4A7C15ULL + i;
        r->parms.setRandomGenerator(std::make_shared<Blake2xbPRNGFactory>(s));
        r->ctx.reset(new SEALContext(r->parms, true, SecurityLevel::none));
        if (!r->ctx->parametersSet()) throw std::invalid_argument(r->ctx->parameterErrorMessage());
        r->ev.reset(new Evaluator(*r->ctx));
        r->N = N;
        r->K = K;
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_create: %s\n", e.what());
        delete r;
        return nullptr;
    }
    return r;
}
void ref_destroy(void *h) { delete (Ref *)h; }
const char *ref_last_error(void *h) { return ((Ref *)h)->err.c_str(); }

// number of levels in the chain (key level included) and the limb count of the first data level
int ref_chain(void *h, int *first_limbs, int *last_limbs) {
    Ref *r = (Ref *)h;
    int n = 0;
    for (auto cd = r->ctx->keyContextData(); cd; cd = cd->nextContextData()) n++;
    *first_limbs = (int)r->ctx->firstContextData()->parms().coeffModulus().size();
    *last_limbs = (int)r->ctx->lastContextData()->parms().coeffModulus().size();
    return n;
}

// ---- tables (src/utils/ntt.cpp:17-66) ----
// out arrays each N entries: root operand/quotient, inverse-root operand/quotient; inv_degree[2]
int ref_ntt_tables(void *h, int prime_idx, u64 *rop, u64 *rquo, u64 *iop, u64 *iquo, u64 *inv_degree, u64 *root) {
    Ref *r = (Ref *)h;
    const auto &tb = r->ctx->keyContextData()->smallNTTTables()[prime_idx];
    for (size_t i = 0; i < r->N; i++) {
        auto a = tb.getFromRootPowers(i), b = tb.getFromInvRootPowers(i);
        rop[i] = a.operand; rquo[i] = a.quotient; iop[i] = b.operand; iquo[i] = b.quotient;
    }
    inv_degree[0] = tb.invDegreeModulo().operand;
    inv_degree[1] = tb.invDegreeModulo().quotient;
    *root = tb.getRoot();
    return 0;
}

// BEHZ bases of the level with `limbs` primes (src/utils/rns.cpp:581-689).
// bsk_out: |Bsk| primes (B then m_sk); returns |Bsk|; *gamma = second aux prime.
int ref_behz_bases(void *h, int limbs, u64 *bsk_out, u64 *gamma) {
    Ref *r = (Ref *)h;
    auto cd = level_data(r, limbs);
    if (!cd) return -1;
    auto rt = cd->rnsTool();
    size_t n = rt->baseBsk()->size();
    for (size_t i = 0; i < n; i++) bsk_out[i] = (*rt->baseBsk())[i].value();
    *gamma = rt->gamma().value();
    return (int)n;
}

// Root tables of the Bsk primes of a level: same layout as ref_ntt_tables
int ref_bsk_ntt_tables(void *h, int limbs, int idx, u64 *rop, u64 *iop, u64 *inv_degree) {
    Ref *r = (Ref *)h;
    auto cd = level_data(r, limbs);
    if (!cd) return -1;
    const auto &tb = cd->rnsTool()->baseBskNttTables()[idx];
    for (size_t i = 0; i < r->N; i++) {
        rop[i] = tb.getFromRootPowers(i).operand;
        iop[i] = tb.getFromInvRootPowers(i).operand;
    }
    *inv_degree = tb.invDegreeModulo().operand;
    return 0;
}

// ---- limb transforms (src/utils/ntt.cpp:158-213) ----
// mode: 0 fwd lazy, 1 fwd full, 2 inv lazy, 3 inv full; operates on one limb in place
int ref_ntt(void *h, int prime_idx, u64 *data, int mode) {
    Ref *r = (Ref *)h;
    const auto &tb = r->ctx->keyContextData()->smallNTTTables()[prime_idx];
    switch (mode) {
    case 0: util::nttNegacyclicHarveyLazy(HostPointer<u64>(data), tb); break;
    case 1: util::nttNegacyclicHarvey(HostPointer<u64>(data), tb); break;
    case 2: util::inverseNttNegacyclicHarveyLazy(HostPointer<u64>(data), tb); break;
    default: util::inverseNttNegacyclicHarvey(HostPointer<u64>(data), tb); break;
    }
    return 0;
}

// ---- BEHZ stages on one polynomial (src/utils/rns.cpp:805-1146) ----
// stage: 0 fastbconvmTilde  in [limbs][N]            out [|Bsk|+1][N]
//        1 smMrq            in [|Bsk|+1][N]          out [|Bsk|][N]
//        2 fastFloor        in [limbs+|Bsk|][N]      out [|Bsk|][N]
//        3 fastbconvSk      in [|Bsk|][N]            out [limbs][N]
//        4 divideAndRoundqLastInplace      in/out [limbs][N] (out = in buffer copy)
//        5 divideAndRoundqLastNttInplace
//        6 modTAndDivideqLastInplace
int ref_rns_stage(void *h, int limbs, int stage, const u64 *in, u64 *out) {
    Ref *r = (Ref *)h;
    auto cd = level_data(r, limbs);
    if (!cd) return -1;
    auto rt = cd->rnsTool();
    return guarded(r, [&] {
        switch (stage) {
        case 0: rt->fastbconvmTilde(ConstHostPointer<u64>(in), HostPointer<u64>(out)); break;
        case 1: rt->smMrq(ConstHostPointer<u64>(in), HostPointer<u64>(out)); break;
        case 2: rt->fastFloor(ConstHostPointer<u64>(in), HostPointer<u64>(out)); break;
        case 3: rt->fastbconvSk(ConstHostPointer<u64>(in), HostPointer<u64>(out)); break;
        case 4:
            std::memcpy(out, in, sizeof(u64) * limbs * r->N);
            rt->divideAndRoundqLastInplace(HostPointer<u64>(out));
            break;
        case 5:
            std::memcpy(out, in, sizeof(u64) * limbs * r->N);
            rt->divideAndRoundqLastNttInplace(HostPointer<u64>(out), cd->smallNTTTables());
            break;
        case 6:
            std::memcpy(out, in, sizeof(u64) * limbs * r->N);
            rt->modTAndDivideqLastInplace(HostPointer<u64>(out));
            break;
        default: throw std::invalid_argument("stage");
        }
    });
}

// ---- keys ----
int ref_keygen(void *h, const uint32_t *galois_elts, int n_elts) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        r->keygen.reset(new KeyGenerator(*r->ctx));
        r->keygen->createPublicKey(r->pk);
        r->has_pk = true;
        if (r->K > 1) {
            r->keygen->createRelinKeys(r->rlk);
            if (n_elts > 0) r->keygen->createGaloisKeys(std::vector<uint32_t>(galois_elts, galois_elts + n_elts), r->gk);
        }
    });
}
// install an externally generated secret key [K][N] (NTT form) / public key [2][K][N] so the reference's own
// Decryptor / Encryptor can be run against material produced by the product's host-side key generator
int ref_set_secret_key(void *h, const u64 *data) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        r->sk_override.reset(new SecretKey());
        r->sk_override->data().resize(r->K * r->N);
        std::memcpy(r->sk_override->data().data(), data, sizeof(u64) * r->K * r->N);
        r->sk_override->parmsID() = r->ctx->keyParmsID();
    });
}
int ref_set_public_key(void *h, const u64 *data) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        Ciphertext &c = r->pk.data();
        c.resize(*r->ctx, r->ctx->keyParmsID(), 2);
        std::memcpy(c.data(), data, sizeof(u64) * 2 * r->K * r->N);
        c.isNttForm() = true;
        r->has_pk = true;
    });
}
// secret key: [K][N] NTT form at key level (src/keygenerator.cpp)
int ref_get_secret_key(void *h, u64 *out) {
    Ref *r = (Ref *)h;
    std::memcpy(out, r->keygen->secretKey().data().data(), sizeof(u64) * r->K * r->N);
    return 0;
}
int ref_get_public_key(void *h, u64 *out) {
    Ref *r = (Ref *)h;
    export_ct(r->pk.data(), out);
    return 0;
}
static void export_ksk(Ref *r, const std::vector<PublicKey> &kv, u64 *out) {
    size_t stride = 2 * r->K * r->N;
    for (size_t j = 0; j < kv.size(); j++) export_ct(kv[j].data(), out + j * stride);
}
// key-switch key layout: [decomp j < K-1][component 2][limb K][N]  (src/kswitchkeys.h)
int ref_get_relin_key(void *h, u64 *out) {
    Ref *r = (Ref *)h;
    export_ksk(r, r->rlk.key(2), out);
    return 0;
}
int ref_get_galois_key(void *h, uint32_t elt, u64 *out) {
    Ref *r = (Ref *)h;
    if (!r->gk.hasKey(elt)) return -1;
    export_ksk(r, r->gk.key(elt), out);
    return 0;
}
// Install synthetic key material (uniform limbs) so the key-switch arithmetic can be compared on
// arbitrary inputs.  which: 0 = relin key (power 2), 0x80000000 | i = relin key of index i (power i + 2), else Galois element.
int ref_set_kswitch_key(void *h, uint32_t which, const u64 *data) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        size_t L = r->K - 1, stride = 2 * r->K * r->N;
        std::vector<PublicKey> kv(L);
        for (size_t j = 0; j < L; j++) {
            Ciphertext &c = kv[j].data();
            c.resize(*r->ctx, r->ctx->keyParmsID(), 2);
            std::memcpy(c.data(), data + j * stride, sizeof(u64) * stride);
            c.isNttForm() = true;
        }
        const bool relin = which == 0 || (which & 0x80000000u);
        KSwitchKeys &ks = relin ? static_cast<KSwitchKeys &>(r->rlk) : static_cast<KSwitchKeys &>(r->gk);
        size_t index = relin ? RelinKeys::getIndex(2 + (which & 0xFFFFu)) : GaloisKeys::getIndex(which);
        if (ks.data().size() <= index) ks.data().resize(index + 1);
        ks.data()[index] = std::move(kv);
        ks.parmsID() = r->ctx->keyParmsID();
    });
}

// ---- evaluator ops on raw buffers (src/evaluator.cpp) ----
// op codes
enum { OP_ADD = 0, OP_SUB, OP_NEGATE, OP_MULTIPLY, OP_SQUARE, OP_RELIN, OP_MODSWITCH_NEXT, OP_RESCALE_NEXT,
       OP_APPLY_GALOIS, OP_ROTATE_ROWS, OP_ROTATE_COLUMNS, OP_ROTATE_VECTOR, OP_CONJUGATE, OP_TO_NTT, OP_FROM_NTT,
       OP_MULTIPLY_PLAIN_NTT, OP_ADD_PLAIN, OP_SUB_PLAIN, OP_MULTIPLY_PLAIN };

struct RefCtDesc { // mirrors python ctypes struct
    int limbs, size, is_ntt;
    double scale;
    u64 correction_factor;
};

// a: first operand, b: second operand (ct for add/sub/mul; plaintext [limbs][N] NTT form for
// OP_MULTIPLY_PLAIN_NTT; ignored otherwise); iarg: galois elt or step.
// out: caller buffer large enough; od receives the result descriptor.
int ref_eval(void *h, int op, const RefCtDesc *ad, const u64 *a, const RefCtDesc *bd, const u64 *b, int64_t iarg,
             RefCtDesc *od, u64 *out) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        Ciphertext x = make_ct(r, ad->limbs, ad->size, a, ad->is_ntt, ad->scale, ad->correction_factor);
        const Evaluator &ev = *r->ev;
        switch (op) {
        case OP_ADD: case OP_SUB: case OP_MULTIPLY: {
            Ciphertext y = make_ct(r, bd->limbs, bd->size, b, bd->is_ntt, bd->scale, bd->correction_factor);
            if (op == OP_ADD) ev.addInplace(x, y);
            else if (op == OP_SUB) ev.subInplace(x, y);
            else ev.multiplyInplace(x, y);
            break;
        }
        case OP_NEGATE: ev.negateInplace(x); break;
        case OP_SQUARE: ev.squareInplace(x); break;
        case OP_RELIN: ev.relinearizeInplace(x, r->rlk); break;
        case OP_MODSWITCH_NEXT: ev.modSwitchToNextInplace(x); break;
        case OP_RESCALE_NEXT: ev.rescaleToNextInplace(x); break;
        case OP_APPLY_GALOIS: ev.applyGaloisInplace(x, (uint32_t)iarg, r->gk); break;
        case OP_ROTATE_ROWS: ev.rotateRowsInplace(x, (int)iarg, r->gk); break;
        case OP_ROTATE_COLUMNS: ev.rotateColumnsInplace(x, r->gk); break;
        case OP_ROTATE_VECTOR: ev.rotateVectorInplace(x, (int)iarg, r->gk); break;
        case OP_CONJUGATE: ev.complexConjugateInplace(x, r->gk); break;
        case OP_TO_NTT: ev.transformToNttInplace(x); break;
        case OP_FROM_NTT: ev.transformFromNttInplace(x); break;
        case OP_MULTIPLY_PLAIN_NTT: {
            auto cd = level_data(r, ad->limbs);
            Plaintext p;
            p.resize(ad->limbs * r->N);
            std::memcpy(p.data(), b, sizeof(u64) * ad->limbs * r->N);
            p.parmsID() = cd->parmsID();
            p.scale() = bd ? bd->scale : 1.0;
            ev.multiplyPlainInplace(x, p);
            break;
        }
        case OP_ADD_PLAIN: case OP_SUB_PLAIN: case OP_MULTIPLY_PLAIN: {
            // b: BFV/BGV iarg coefficients mod t;  CKKS (add/sub only): [limbs][N] NTT form at the level of x
            Plaintext p;
            if (r->ctx->keyContextData()->parms().scheme() == SchemeType::ckks) {
                auto cd = level_data(r, ad->limbs);
                p.resize(ad->limbs * r->N);
                std::memcpy(p.data(), b, sizeof(u64) * ad->limbs * r->N);
                p.parmsID() = cd->parmsID();
                p.scale() = bd ? bd->scale : 1.0;
            } else {
                p.resize((size_t)iarg);
                std::memcpy(p.data(), b, sizeof(u64) * (size_t)iarg);
            }
            if (op == OP_ADD_PLAIN) ev.addPlainInplace(x, p);
            else if (op == OP_SUB_PLAIN) ev.subPlainInplace(x, p);
            else ev.multiplyPlainInplace(x, p);
            break;
        }
        default: throw std::invalid_argument("op");
        }
        od->limbs = (int)x.coeffModulusSize();
        od->size = (int)x.size();
        od->is_ntt = x.isNttForm();
        od->scale = x.scale();
        od->correction_factor = x.correctionFactor();
        export_ct(x, out);
    });
}

// Evaluator::transformToNttInplace(Plaintext&, parms_id) (src/evaluator.cpp:1972-2070): out [limbs][N]
int ref_plain_to_ntt(void *h, const u64 *plain, int n_coeffs, int limbs, u64 *out) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        auto cd = level_data(r, limbs);
        Plaintext p((size_t)n_coeffs);
        std::memcpy(p.data(), plain, sizeof(u64) * (size_t)n_coeffs);
        r->ev->transformToNttInplace(p, cd->parmsID());
        std::memcpy(out, p.data(), sizeof(u64) * (size_t)limbs * r->N);
    });
}
// parms_id of the level with `limbs` primes (src/encryptionparams.cpp:118-146: BLAKE2b-256 of the parameter words)
int ref_parms_id(void *h, int limbs, u64 *out4) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        auto cd = level_data(r, limbs);
        auto id = cd->parmsID();
        for (int i = 0; i < 4; i++) out4[i] = id[i];
    });
}
uint32_t ref_galois_elt_from_step(void *h, int step) {
    Ref *r = (Ref *)h;
    return r->ctx->keyContextData()->galoisTool()->getEltFromStep(step);
}

// ---- encrypt / decrypt plumbing (cfgA; src/encryptor.cpp, src/decryptor.cpp) ----
// BFV/BGV: plain = polynomial coefficients mod t (n_coeffs <= N); ct out at the first data level.
int ref_encrypt(void *h, const u64 *plain, int n_coeffs, RefCtDesc *od, u64 *out) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        Encryptor enc(*r->ctx, r->pk);
        Plaintext p(n_coeffs);
        std::memcpy(p.data(), plain, sizeof(u64) * n_coeffs);
        Ciphertext c;
        enc.encrypt(p, c);
        od->limbs = (int)c.coeffModulusSize(); od->size = (int)c.size(); od->is_ntt = c.isNttForm();
        od->scale = c.scale(); od->correction_factor = c.correctionFactor();
        export_ct(c, out);
    });
}
// decrypt to N plaintext coefficients (BFV/BGV); returns noise budget in *budget when not null
int ref_decrypt(void *h, const RefCtDesc *ad, const u64 *a, u64 *plain_out, int *budget) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        Decryptor dec(*r->ctx, r->sk_override ? *r->sk_override : r->keygen->secretKey());
        Ciphertext x = make_ct(r, ad->limbs, ad->size, a, ad->is_ntt, ad->scale, ad->correction_factor);
        Plaintext p;
        dec.decrypt(x, p);
        std::memset(plain_out, 0, sizeof(u64) * r->N);
        std::memcpy(plain_out, p.data(), sizeof(u64) * std::min<size_t>(p.coeffCount(), r->N));
        if (budget) *budget = (r->parms.scheme() == SchemeType::ckks) ? 0 : dec.invariantNoiseBudget(x);
    });
}
// BFV/BGV batching (src/batchencoder.cpp): values[N] <-> plaintext polynomial[N]
int ref_batch_encode(void *h, const u64 *values, u64 *plain_out) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        BatchEncoder be(*r->ctx);
        Plaintext p;
        be.encode(std::vector<u64>(values, values + r->N), p);
        std::memset(plain_out, 0, sizeof(u64) * r->N);
        std::memcpy(plain_out, p.data(), sizeof(u64) * std::min<size_t>(p.coeffCount(), r->N));
    });
}
int ref_batch_decode(void *h, const u64 *plain, u64 *values_out) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        BatchEncoder be(*r->ctx);
        Plaintext p(r->N);
        std::memcpy(p.data(), plain, sizeof(u64) * r->N);
        std::vector<u64> v;
        be.decode(p, v);
        std::memcpy(values_out, v.data(), sizeof(u64) * r->N);
    });
}
// CKKS: encode N/2 complex slots (interleaved re,im doubles) at the level with `limbs` primes ->
// plaintext [limbs][N] in NTT form; encrypt it; decrypt+decode.
int ref_ckks_encode(void *h, const double *slots, double scale, int limbs, u64 *plain_out) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        CKKSEncoder enc(*r->ctx);
        std::vector<std::complex<double>> v(r->N / 2);
        for (size_t i = 0; i < v.size(); i++) v[i] = {slots[2 * i], slots[2 * i + 1]};
        Plaintext p;
        enc.encode(v, level_data(r, limbs)->parmsID(), scale, p);
        std::memcpy(plain_out, p.data(), sizeof(u64) * limbs * r->N);
    });
}
int ref_ckks_encrypt(void *h, const u64 *plain, double scale, int limbs, RefCtDesc *od, u64 *out) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        Encryptor enc(*r->ctx, r->pk);
        Plaintext p;
        p.resize(limbs * r->N);
        std::memcpy(p.data(), plain, sizeof(u64) * limbs * r->N);
        p.parmsID() = level_data(r, limbs)->parmsID();
        p.scale() = scale;
        Ciphertext c;
        enc.encrypt(p, c);
        od->limbs = (int)c.coeffModulusSize(); od->size = (int)c.size(); od->is_ntt = c.isNttForm();
        od->scale = c.scale(); od->correction_factor = c.correctionFactor();
        export_ct(c, out);
    });
}
int ref_ckks_decrypt_decode(void *h, const RefCtDesc *ad, const u64 *a, double *slots_out) {
    Ref *r = (Ref *)h;
    return guarded(r, [&] {
        Decryptor dec(*r->ctx, r->keygen->secretKey());
        Ciphertext x = make_ct(r, ad->limbs, ad->size, a, ad->is_ntt, ad->scale, ad->correction_factor);
        Plaintext p;
        dec.decrypt(x, p);
        CKKSEncoder enc(*r->ctx);
        std::vector<std::complex<double>> v;
        enc.decode(p, v);
        for (size_t i = 0; i < v.size(); i++) { slots_out[2 * i] = v[i].real(); slots_out[2 * i + 1] = v[i].imag(); }
    });
}

// ---- CPU baseline (kind "reference"): time `reps` multiply+relinearize on given inputs ----
double ref_time_mul_relin(void *h, const RefCtDesc *ad, const u64 *a, const u64 *b, int reps) {
    Ref *r = (Ref *)h;
    try {
        Ciphertext x = make_ct(r, ad->limbs, ad->size, a, ad->is_ntt, ad->scale, ad->correction_factor);
        Ciphertext y = make_ct(r, ad->limbs, ad->size, b, ad->is_ntt, ad->scale, ad->correction_factor);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) {
            Ciphertext z;
            r->ev->multiply(x, y, z);
            r->ev->relinearizeInplace(z, r->rlk);
        }
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } catch (const std::exception &e) {
        r->err = e.what();
        return -1.0;
    }
}

} // extern "C"
