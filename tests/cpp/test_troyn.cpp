// Drop-in check of include/troyn.hpp: user code written against the reference's troyn:: interface
// (src/troy_cuda.cuh) -- KernelProvider::initialize, EncryptionParameters, SEALContext, KeyGenerator, Encryptor,
// Evaluator, Decryptor -- compiled with plain g++ and linked to libtroyhip.so.  The scenarios follow the reference's
// own evaluator tests (test/evaluator.cu style: encrypt, operate on the GPU, decrypt, compare with the plaintext
// computation), with polynomial plaintexts because encoders are out of scope.
#include "troyn.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <sstream>

using namespace troyn;

static int failures = 0;
#define EXPECT(cond, what)                                              \
    do {                                                                \
        if (!(cond)) { std::printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__); failures++; } \
        else std::printf("ok   %s\n", what);                            \
    } while (0)

using Poly = std::vector<uint64_t>;

static Poly negacyclic_mul(const Poly &a, const Poly &b, uint64_t t) {
    const size_t n = a.size();
    Poly r(n, 0);
    for (size_t i = 0; i < n; i++) {
        if (!a[i]) continue;
        for (size_t j = 0; j < n; j++) {
            if (!b[j]) continue;
            unsigned __int128 v = (unsigned __int128)a[i] * b[j] % t;
            size_t k = i + j;
            if (k >= n) { k -= n; v = (t - (uint64_t)v) % t; }
            r[k] = (uint64_t)((r[k] + (uint64_t)v) % t);
        }
    }
    return r;
}
static Poly automorphism(const Poly &a, uint32_t elt, uint64_t t) { // x -> x^elt in Z_t[x]/(x^n+1)
    const size_t n = a.size();
    Poly r(n, 0);
    for (size_t i = 0; i < n; i++) {
        size_t k = (size_t)((uint64_t)i * elt % (2 * n));
        if (k >= n) r[k - n] = (t - a[i]) % t; else r[k] = a[i];
    }
    return r;
}
static Poly sparse_poly(size_t n, uint64_t t, unsigned seed, size_t terms) {
    std::mt19937_64 g(seed);
    Poly p(n, 0);
    for (size_t i = 0; i < terms; i++) p[g() % n] = g() % t;
    return p;
}
static Poly dense_poly(size_t n, uint64_t t, unsigned seed) {
    std::mt19937_64 g(seed);
    Poly p(n);
    for (auto &x : p) x = g() % t;
    return p;
}

static void scenario(SchemeType scheme, size_t n, std::vector<int> bits, int tbits) {
    std::printf("-- scheme %d N=%zu K=%zu\n", (int)scheme, n, bits.size());
    EncryptionParameters parms(scheme);
    parms.setPolyModulusDegree(n);
    parms.setCoeffModulus(CoeffModulus::Create(n, bits));
    parms.setPlainModulus(PlainModulus::Batching(n, tbits));
    SEALContext context(parms, true, SecurityLevel::none);
    const uint64_t t = parms.plainModulus().value();

    KeyGenerator keygen(context);
    PublicKey pk = keygen.createPublicKey();
    RelinKeys rlk = keygen.createRelinKeys();
    GaloisKeys gk;
    keygen.createGaloisKeys(std::vector<int>{1, -1, 4}, gk);
    Encryptor encryptor(context, pk);
    Decryptor decryptor(context, keygen.secretKey());
    Evaluator evaluator(context);

    const Poly va = dense_poly(n, t, 1), vb = sparse_poly(n, t, 2, 24);
    Plaintext pa(va), pb(vb), out;
    Ciphertext a, b, c;
    encryptor.encrypt(pa, a);
    encryptor.encrypt(pb, b);
    decryptor.decrypt(a, out);
    EXPECT(out == pa, "encrypt -> decrypt");

    // encryptZero / encryptZeroSymmetric at the first and at the last level (src/encryptor_cuda.cuh:170-320), and the small members around
    // them: Ciphertext::isTransparent / reserve / sizeCapacity / release, Plaintext::setZero / release
    {
        Encryptor both(context, pk, keygen.secretKey());
        Ciphertext z = both.encryptZero(), zs = both.encryptZeroSymmetric(), zl, zsl;
        both.encryptZero(context.lastParmsID(), zl);
        both.encryptZeroSymmetric(context.lastParmsID(), zsl);
        bool zeros = true;
        for (const Ciphertext *ct : {&z, &zs, &zl, &zsl}) {
            decryptor.decrypt(*ct, out);
            zeros = zeros && out == Plaintext(Poly(n, 0)) && !ct->isTransparent() && ct->size() == 2 && !ct->isNttForm();
        }
        EXPECT(zeros, "encryptZero / encryptZeroSymmetric decrypt to zero at the first and at the last level");
        EXPECT(z.parmsID() == context.firstParmsID() && zl.parmsID() == context.lastParmsID() && zl.coeffModulusSize() < z.coeffModulusSize(),
               "encryptZero(parms_id) lands on that level");
        evaluator.addInplace(z, a);
        decryptor.decrypt(z, out);
        EXPECT(out == pa, "an encryption of zero is an additive identity");
        bool thrown = false;
        try { both.encryptZero(context.keyParmsID(), zl); } catch (const std::invalid_argument &) { thrown = true; }
        EXPECT(thrown, "encryptZero at the key level: invalid_argument");
        Ciphertext grown = a;
        grown.reserve(5);
        decryptor.decrypt(grown, out);
        EXPECT(grown.sizeCapacity() == 5 && grown.size() == 2 && out == pa, "Ciphertext::reserve keeps the polynomials");
        grown.release();
        EXPECT(grown.size() == 0 && grown.isTransparent(), "Ciphertext::release");
        Plaintext pz = pa;
        pz.setZero(n / 2);
        bool tail = true;
        for (size_t i = 0; i < n; i++) tail = tail && pz[i] == (i < n / 2 ? pa[i] : 0);
        pz.setZero(0, 1);
        EXPECT(tail && pz[0] == 0, "Plaintext::setZero(start) / setZero(start, length)");
        thrown = false;
        try { pz.setZero(n); } catch (const std::out_of_range &) { thrown = true; }
        pz.release();
        EXPECT(thrown && pz.coeffCount() == 0, "Plaintext::setZero out of range: out_of_range; release");
    }

    {   // Modulus::isPrime / constRatio (src/modulus.h:16-24) and ContextData::qualifiers() (src/context.cpp:286-305, 411-417)
        const Modulus q0 = parms.coeffModulus()[0];
        const std::array<uint64_t, 3> cr = q0.constRatio();
        const unsigned __int128 back = (((unsigned __int128)cr[1] << 64) | cr[0]) * q0.value() + cr[2];
        EXPECT(q0.isPrime() && !Modulus(q0.value() + 2 * n).isPrime() + !Modulus(q0.value() - 1).isPrime() >= 1 && !Modulus(1ull << 41).isPrime() &&
                   Modulus(0xffffffffffc0001ull).isPrime() && !Modulus(0xffffffffffc0001ull * 3).isPrime() && back == 0 && cr[2] < q0.value(),
               "Modulus::isPrime, constRatio (floor(2^128 / p) and the remainder)");
        const EncryptionParameterQualifiers ql = context.firstContextData()->qualifiers();
        bool descending = true;
        for (size_t i = 0; i + 1 < parms.coeffModulus().size(); i++) descending = descending && parms.coeffModulus()[i].value() > parms.coeffModulus()[i + 1].value();
        EXPECT(ql.parametersSet() && ql.using_batching && ql.using_fast_plain_lift && ql.using_ntt && ql.using_fft &&
                   context.keyContextData()->qualifiers().using_descending_modulus_chain == descending,
               "qualifiers(): batching prime, fast plain lift, descending chain");
    }

    // add / sub / negate
    evaluator.add(a, b, c);
    decryptor.decrypt(c, out);
    Poly want(n);
    for (size_t i = 0; i < n; i++) want[i] = (pa[i] + pb[i]) % t;
    EXPECT(out == Plaintext(want), "add");
    evaluator.subInplace(c, b);
    evaluator.negateInplace(c);
    decryptor.decrypt(c, out);
    for (size_t i = 0; i < n; i++) want[i] = (t - pa[i]) % t;
    EXPECT(out == Plaintext(want), "sub, negate");

    // multiply + relinearize (BASELINE config A: BFVRelinearize)
    evaluator.multiply(a, b, c);
    EXPECT(c.size() == 3, "multiply gives size 3");
    Poly prod = negacyclic_mul(vb, va, t);
    decryptor.decrypt(c, out);
    EXPECT(out == Plaintext(prod), "decrypt size-3 product");
    evaluator.relinearizeInplace(c, rlk);
    EXPECT(c.size() == 2, "relinearize gives size 2");
    decryptor.decrypt(c, out);
    EXPECT(out == Plaintext(prod), "multiply + relinearize");

    // plaintext operands: ct + pt, ct - pt, ct * pt (coefficient form), and ct_ntt * pt_ntt after transformToNtt
    Ciphertext e;
    evaluator.addPlain(a, pb, e);
    decryptor.decrypt(e, out);
    for (size_t i = 0; i < n; i++) want[i] = (va[i] + vb[i]) % t;
    EXPECT(out == Plaintext(want), "addPlain");
    evaluator.subPlainInplace(e, pb);
    decryptor.decrypt(e, out);
    EXPECT(out == pa, "subPlain");
    evaluator.multiplyPlain(a, pb, e);
    decryptor.decrypt(e, out);
    EXPECT(out == Plaintext(prod), "multiplyPlain (coefficient form)");
    {
        Ciphertext antt = a;
        Plaintext pbn = pb;
        evaluator.transformToNttInplace(antt);
        evaluator.transformToNttInplace(pbn, antt.parmsID());
        evaluator.multiplyPlainInplace(antt, pbn);
        evaluator.transformFromNttInplace(antt);
        EXPECT(antt.toHost() == e.toHost(), "multiplyPlain NTT path == coefficient path (same limbs)");
    }

    // exponentiate through multiplyMany (one level: the small test parameters have no noise budget for two)
    {
        Ciphertext c2 = b;
        evaluator.exponentiateInplace(c2, 2, rlk);
        decryptor.decrypt(c2, out);
        EXPECT(out == Plaintext(negacyclic_mul(vb, vb, t)), "exponentiate(2)");
        Ciphertext sh = a;
        evaluator.negacyclicShiftInplace(sh, 5);
        decryptor.decrypt(sh, out);
        Poly x5(n, 0);
        x5[5] = 1;
        EXPECT(out == Plaintext(negacyclic_mul(x5, va, t)), "negacyclicShift(5) == times x^5");
    }

    // save / load and saveTerms / loadTerms (src/ciphertext_cuda.cu:16-143)
    {
        std::stringstream ss;
        c.save(ss, context);
        Ciphertext back;
        back.load(ss, context);
        EXPECT(back.toHost() == c.toHost() && back.size() == c.size() && back.isNttForm() == c.isNttForm(), "save -> load round trip");
        std::stringstream st;
        const std::vector<size_t> terms = {0, 7, n - 1};
        c.saveTerms(st, context, evaluator, terms);
        EXPECT(st.str().size() < ss.str().size(), "saveTerms writes less than save");
        Ciphertext part;
        part.loadTerms(st, context, evaluator, terms);
        decryptor.decrypt(part, out);
        bool ok = true;
        for (size_t id : terms) ok = ok && out[id] == prod[id];
        EXPECT(ok, "loadTerms keeps the listed coefficients of the plaintext");
        bool threw2 = false;
        std::stringstream again(ss.str());
        try { part.loadTerms(again, context, evaluator, terms); } catch (const std::invalid_argument &) { threw2 = true; }
        EXPECT(threw2, "loadTerms on a full ciphertext -> invalid_argument");
    }

    // mod switch keeps the plaintext
    Ciphertext d;
    evaluator.modSwitchToNext(c, d);
    EXPECT(d.coeffModulusSize() + 1 == c.coeffModulusSize(), "modSwitchToNext drops a limb");
    decryptor.decrypt(d, out);
    EXPECT(out == Plaintext(prod), "modSwitchToNext");

    // Galois automorphism and NAF rotation (steps 3 = 4 - 1 with keys {1,-1,4})
    uint32_t e1 = 0;
    check(troyhip_galois_elt_from_step(context.handle(), 1, &e1));
    Ciphertext g = a;
    evaluator.applyGaloisInplace(g, e1, gk);
    decryptor.decrypt(g, out);
    EXPECT(out == Plaintext(automorphism(va, e1, t)), "applyGalois");
    Ciphertext r = a;
    evaluator.rotateRowsInplace(r, 3, gk);
    Poly w = va;
    uint32_t e4 = 0, em1 = 0;
    check(troyhip_galois_elt_from_step(context.handle(), 4, &e4));
    check(troyhip_galois_elt_from_step(context.handle(), -1, &em1));
    w = automorphism(automorphism(w, e4, t), em1, t);
    decryptor.decrypt(r, out);
    EXPECT(out == Plaintext(w), "rotateRows(3) by NAF");

    // error behaviour: same exception classes as the reference
    bool threw = false;
    try { GaloisKeys none; evaluator.applyGaloisInplace(g, e1, none); } catch (const std::invalid_argument &) { threw = true; }
    EXPECT(threw, "missing Galois key -> invalid_argument");
    threw = false;
    try { evaluator.rotateVectorInplace(g, 1, gk); } catch (const std::logic_error &) { threw = true; }
    EXPECT(threw, "rotateVector on BFV/BGV -> logic_error");
    threw = false;
    try { evaluator.addInplace(c, d); } catch (const std::invalid_argument &) { threw = true; }
    EXPECT(threw, "level mismatch -> invalid_argument");
}

// Advisor findings of round 5: (1) on a seeded Encryptor two consecutive symmetric encryptions must never share c1 (the public polynomial a): with
// one a under one key, c0 - c0' = delta m + e - e' gives the plaintext away; (2) keys of another context (a loaded key can have any shape) are
// refused where they are used, with the reference's messages, instead of being read past their end
static void key_hygiene() {
    auto make = [](std::vector<int> bits) {
        EncryptionParameters p(SchemeType::bfv);
        p.setPolyModulusDegree(4096);
        p.setCoeffModulus(CoeffModulus::Create(4096, bits));
        p.setPlainModulus(PlainModulus::Batching(4096, 20));
        return p;
    };
    SEALContext small(make({36, 36, 37}), true, SecurityLevel::none), big(make({36, 36, 37, 38}), true, SecurityLevel::none);
    KeyGenerator kg_small(small, 5, 6), kg_big(big, 7, 8);
    // (1) the seeds of consecutive symmetric encryptions, in every order of the two entry points
    Encryptor enc(big, kg_big.createPublicKey(), 11, 12);
    enc.setSecretKey(kg_big.secretKey());
    const Plaintext m(sparse_poly(4096, 1 << 19, 3, 9));
    std::vector<Ciphertext> cts;
    cts.push_back(enc.encryptZeroSymmetric());
    cts.push_back(enc.encryptSymmetric(m));
    cts.push_back(enc.encryptSymmetric(m));
    cts.push_back(enc.encryptZeroSymmetric());
    cts.push_back(enc.encryptZeroSymmetric());
    cts.push_back(enc.encryptSymmetric(m));
    bool distinct = true;
    for (size_t i = 0; i < cts.size(); i++)
        for (size_t j = i + 1; j < cts.size(); j++) {
            const std::vector<uint64_t> a = cts[i].toHost(), b = cts[j].toHost();
            const size_t poly = a.size() / 2;
            if (cts[i].seed() == 0 || cts[i].seed() == cts[j].seed() || std::equal(a.begin() + poly, a.end(), b.begin() + poly)) distinct = false;
        }
    EXPECT(distinct, "consecutive seeded symmetric encryptions never share seed() or c1");
    Decryptor dec(big, kg_big.secretKey());
    Plaintext out;
    dec.decrypt(cts[1], out);
    EXPECT(out == m, "seeded symmetric encryption still decrypts");
    // (2) keys saved under the 3-prime context, loaded, and offered to the 4-prime context
    auto refused = [](auto &&f, const char *needle) {
        try { f(); } catch (const std::invalid_argument &e) { return std::string(e.what()).find(needle) != std::string::npos; }
        return false;
    };
    std::stringstream ps, ss, rs, gs;
    kg_small.createPublicKey().save(ps);
    kg_small.secretKey().save(ss);
    kg_small.createRelinKeys().save(rs);
    GaloisKeys gsmall;
    kg_small.createGaloisKeys(std::vector<int>{1}, gsmall);
    gsmall.save(gs);
    PublicKey pk2; pk2.load(ps);
    SecretKey sk2; sk2.load(ss);
    RelinKeys rk2; rk2.load(rs);
    GaloisKeys gk2; gk2.load(gs);
    EXPECT(refused([&] { Encryptor e(big, pk2); }, "public key is not valid"), "Encryptor refuses a public key of another context");
    EXPECT(refused([&] { Encryptor e(big, sk2); }, "secret key is not valid"), "Encryptor refuses a secret key of another context");
    EXPECT(refused([&] { Encryptor e(big, kg_big.createPublicKey()); e.setPublicKey(pk2); }, "public key is not valid"), "setPublicKey refuses it too");
    EXPECT(refused([&] { Encryptor e(big, kg_big.createPublicKey()); e.setSecretKey(sk2); }, "secret key is not valid"), "setSecretKey refuses it too");
    EXPECT(refused([&] { Decryptor d(big, sk2); }, "secret key is not valid"), "Decryptor refuses a secret key of another context");
    Evaluator ev(big);
    Encryptor pub(big, kg_big.createPublicKey(), 21, 22);
    Ciphertext x = pub.encrypt(m), y;
    ev.multiply(x, x, y);
    EXPECT(refused([&] { Ciphertext z = y; ev.relinearizeInplace(z, rk2); }, "kswitch_keys is not valid"), "relinearize refuses keys of another context");
    EXPECT(refused([&] { Ciphertext z; ev.relinearize(y, rk2, z); }, "kswitch_keys is not valid"), "relinearize (destination form) refuses them");
    uint32_t e1 = 0;
    check(troyhip_galois_elt_from_step(big.handle(), 1, &e1));
    EXPECT(refused([&] { Ciphertext z = x; ev.applyGaloisInplace(z, e1, gk2); }, "kswitch_keys is not valid"), "applyGalois refuses keys of another context");
    EXPECT(refused([&] { Ciphertext z = x; ev.applyKeySwitchingInplace(z, rk2); }, "kswitch_keys is not valid"), "applyKeySwitching refuses keys of another context");
    // the same keys under their own context still work after the round trip
    Evaluator evs(small);
    Encryptor es(small, pk2, 31, 32);
    Decryptor ds(small, sk2);
    Ciphertext xs = es.encrypt(m), ys;
    evs.multiply(xs, xs, ys);
    evs.relinearizeInplace(ys, rk2);
    ds.decrypt(ys, out);
    EXPECT(out == Plaintext(negacyclic_mul(sparse_poly(4096, 1 << 19, 3, 9), sparse_poly(4096, 1 << 19, 3, 9), small.parms().plainModulus().value())),
           "loaded keys work under their own context");
    // (3) a forged header must not drive an allocation: 70 bytes that announce a maximal ciphertext
    std::stringstream forged;
    wire::CtFields f{{1, 2, 3, 4}, false, 16, size_t(1) << 17, 64, 1.0, 1, 0, false};
    wire::put_fields(forged, f);
    wire::put<size_t>(forged, f.size * f.n * f.limbs);
    EXPECT(refused([&] { Ciphertext c; c.load(forged); }, "stream ended"), "a header without its payload is refused before any large allocation");
    std::stringstream oversized;
    f.size = 17;
    wire::put_fields(oversized, f);
    EXPECT(refused([&] { Ciphertext c; c.load(oversized); }, "does not hold a ciphertext"), "more polynomials than the library handles are refused");
}

int main() {
    bool threw = false;
    try {
        EncryptionParameters p(SchemeType::bfv);
        p.setPolyModulusDegree(4096);
        p.setCoeffModulus(CoeffModulus::Create(4096, {40, 40, 40}));
        p.setPlainModulus(PlainModulus::Batching(4096, 20));
        SEALContext c(p, true, SecurityLevel::none);
    } catch (const std::invalid_argument &) { threw = true; }
    EXPECT(threw, "context before KernelProvider::initialize -> invalid_argument");
    KernelProvider::initialize();
    scenario(SchemeType::bfv, 4096, {40, 40, 40}, 20);   // BASELINE config A shape
    scenario(SchemeType::bgv, 8192, {50, 40, 40, 50}, 20);
    key_hygiene();
    std::printf(failures ? "FAILED %d\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
