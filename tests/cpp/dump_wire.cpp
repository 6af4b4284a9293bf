// dump_wire -- writes what the troyn:: serializers produce (Ciphertext::save / saveTerms, the seeded symmetric form, PublicKey / SecretKey /
// KSwitchKeys::save: src/ciphertext_cuda.cu:16-192, publickey_cuda.cuh:252, secretkey_cuda.cuh:292, kswitchkeys_cuda.cuh:330) next to the RAW
// words of the same objects, so that tests/test_wire_format.py can build the expected bytes INDEPENDENTLY (struct.pack of the reference's
// field order + the raw words + the reference's own parms_id from tests/golden/golden_wire.json) and compare byte for byte.  It also loads
// every blob back and checks the objects (and, for keys, their USE) against the originals.  TEST INFRASTRUCTURE.
//   usage: dump_wire <outdir>
#include "troy_cuda.cuh"
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

using namespace troyn;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)
template <class E, class F> static bool throws(F f) { try { f(); } catch (const E &) { return true; } catch (...) {} return false; }

static void write_file(const std::string &path, const std::string &bytes) { std::ofstream(path, std::ios::binary).write(bytes.data(), (std::streamsize)bytes.size()); }
static void write_words(const std::string &path, const std::vector<uint64_t> &w) {
    std::ofstream(path, std::ios::binary).write(reinterpret_cast<const char *>(w.data()), (std::streamsize)(w.size() * 8));
}
static std::vector<uint64_t> device_words(const uint64_t *p, size_t n) {
    std::vector<uint64_t> h(n);
    check(troyhip_copy_d2h(h.data(), p, n * 8, nullptr));
    return h;
}
template <class T> static std::string saved(const T &obj) { std::ostringstream s; obj.save(s); return s.str(); }

static void run(const std::string &dir, const std::string &tag, SchemeType scheme) {
    const size_t N = 64;
    EncryptionParameters parms(scheme);
    parms.setPolyModulusDegree(N);
    parms.setCoeffModulus(CoeffModulus::Create(N, {40, 40, 40}));
    if (scheme != SchemeType::ckks) parms.setPlainModulus(PlainModulus::Batching(N, 17));
    SEALContext context(parms, true, SecurityLevel::none);
    KeyGenerator keygen(context, 11, 22);
    PublicKey pk = keygen.createPublicKey();
    const SecretKey &sk = keygen.secretKey();
    RelinKeys rlk = keygen.createRelinKeys();
    GaloisKeys gk;
    keygen.createGaloisKeys(std::vector<uint32_t>{3, (uint32_t)(2 * N - 1)}, gk);
    Encryptor encryptor(context, pk, 5, 6), sym(context, pk, 7, 8);
    sym.setSecretKey(sk);
    Evaluator evaluator(context);
    Decryptor decryptor(context, sk);

    Plaintext plain;
    if (scheme == SchemeType::ckks) CKKSEncoder(context).encodePolynomial(std::vector<double>{1.5, -2.0, 3.25}, std::ldexp(1.0, 20), plain);
    else plain = "1x^3 + 2";

    { // the level ids, for the comparison with the reference's (golden_wire.json)
        std::ostringstream ids;
        for (size_t limbs = context.keyLimbs(); limbs >= context.lastLimbs(); limbs--) {
            const ParmsID &id = context.parmsIDOfLimbs(limbs);
            if (id == parmsIDZero) continue;
            ids << limbs;
            for (uint64_t w : id) ids << " " << w;
            ids << "\n";
        }
        write_file(dir + "/" + tag + "_ids.txt", ids.str());
    }

    // ---- an ordinary ciphertext, a size-3 product, a termed ciphertext
    Ciphertext ct = encryptor.encrypt(plain), ct3;
    evaluator.multiply(ct, ct, ct3);
    write_file(dir + "/" + tag + "_ct.bin", saved(ct));
    write_words(dir + "/" + tag + "_ct.raw", ct.toHost());
    write_file(dir + "/" + tag + "_ct3.bin", saved(ct3));
    write_words(dir + "/" + tag + "_ct3.raw", ct3.toHost());
    {
        std::ostringstream meta;
        meta.precision(17);
        meta << ct3.scale() << "\n";
        write_file(dir + "/" + tag + "_ct3.scale", meta.str());
    }
    const std::vector<size_t> terms{0, 3, 17, 63};
    {
        std::ostringstream s;
        ct.saveTerms(s, evaluator, terms);
        write_file(dir + "/" + tag + "_terms.bin", s.str());
        Ciphertext back;
        std::istringstream in(s.str());
        back.loadTerms(in, evaluator, terms);
        Plaintext p0, p1;
        // the unlisted coefficients of c0 are dropped, so the decryption agrees only on the listed ones -- checked on the limbs by the Python side;
        // here: form, level and c1 survive
        CHECK(back.isNttForm() == ct.isNttForm() && back.parmsID() == ct.parmsID() && back.size() == 2);
        Ciphertext a = ct, b = back;
        if (a.isNttForm()) { evaluator.transformFromNttInplace(a); evaluator.transformFromNttInplace(b); }
        const std::vector<uint64_t> ha = a.toHost(), hb = b.toHost();
        const size_t poly = a.coeffModulusSize() * N;
        CHECK(std::equal(ha.begin() + (long)poly, ha.end(), hb.begin() + (long)poly));
        for (size_t j = 0; j < a.coeffModulusSize(); j++)
            for (size_t i = 0; i < N; i++) {
                const bool listed = std::find(terms.begin(), terms.end(), i) != terms.end();
                CHECK(hb[j * N + i] == (listed ? ha[j * N + i] : 0));
            }
        CHECK(throws<std::invalid_argument>([&] { std::istringstream t(s.str()); Ciphertext c; c.load(t, context); })); // a termed stream needs its indices
    }
    { // round trips of the ordinary forms, with and without the context
        Ciphertext a, b;
        std::istringstream s1(saved(ct)), s2(saved(ct3));
        a.load(s1, context);
        b.load(s2);
        CHECK(a.toHost() == ct.toHost() && a.parmsID() == ct.parmsID() && a.isNttForm() == ct.isNttForm() && a.scale() == ct.scale());
        CHECK(b.toHost() == ct3.toHost() && b.size() == 3 && b.scale() == ct3.scale());
        Plaintext d0, d1;
        decryptor.decrypt(ct, d0);
        decryptor.decrypt(a, d1);
        CHECK(d0 == d1);
    }

    // ---- the seeded form: a fresh symmetric ciphertext travels as (seed, c0)
    Ciphertext s0;
    sym.encryptSymmetric(plain, s0);
    CHECK(s0.seed() != 0);
    const std::string sym_blob = saved(s0);
    write_file(dir + "/" + tag + "_sym.bin", sym_blob);
    write_words(dir + "/" + tag + "_sym.raw", s0.toHost());
    {
        std::ostringstream meta;
        meta << s0.seed() << "\n";
        write_file(dir + "/" + tag + "_sym.seed", meta.str());
        Ciphertext back;
        std::istringstream in(sym_blob);
        back.load(in, context);
        CHECK(back.seed() == 0 && back.size() == 2 && back.toHost() == s0.toHost()); // c1 regenerated from the seed, limb for limb
        Plaintext d0, d1;
        decryptor.decrypt(s0, d0);
        decryptor.decrypt(back, d1);
        CHECK(d0 == d1);
        CHECK(throws<std::invalid_argument>([&] { std::istringstream t(sym_blob); Ciphertext c; c.load(t); })); // "seed is not zero." without a context
        CHECK(throws<std::invalid_argument>([&] { std::ostringstream t; s0.saveTerms(t, evaluator, terms); }));  // "Seed is not zero."
        // any evaluator op makes it an ordinary ciphertext: the whole of it is saved again
        Ciphertext neg = s0;
        CHECK(neg.seed() == s0.seed());
        evaluator.negateInplace(neg);
        CHECK(neg.seed() == 0 && saved(neg).size() == saved(ct).size());
        // an encryption of zero at the next level down, seeded too
        Ciphertext z;
        sym.encryptZeroSymmetric(context.firstContextData()->nextContextData()->parmsID(), z);
        CHECK(z.seed() != 0 && z.seed() != s0.seed() && z.coeffModulusSize() == ct.coeffModulusSize() - 1);
        Ciphertext zb;
        std::istringstream zin(saved(z));
        zb.load(zin, context);
        CHECK(zb.toHost() == z.toHost());
        Plaintext dz;
        decryptor.decrypt(zb, dz);
        if (scheme != SchemeType::ckks) CHECK(dz.isZero());
    }

    // ---- keys
    write_file(dir + "/" + tag + "_pk.bin", saved(pk));
    write_words(dir + "/" + tag + "_pk.raw", pk.data);
    write_file(dir + "/" + tag + "_sk.bin", saved(sk));
    write_words(dir + "/" + tag + "_sk.raw", sk.data);
    write_file(dir + "/" + tag + "_rlk.bin", saved(rlk));
    const size_t K = context.keyLimbs(), key_words = (K - 1) * 2 * K * N;
    write_words(dir + "/" + tag + "_rlk.raw", device_words(rlk.device(0), key_words));
    write_file(dir + "/" + tag + "_gk.bin", saved(gk));
    write_words(dir + "/" + tag + "_gk1.raw", device_words(gk.device(GaloisKeys::getIndex(3)), key_words));
    write_words(dir + "/" + tag + "_gk63.raw", device_words(gk.device(GaloisKeys::getIndex((uint32_t)(2 * N - 1))), key_words));
    {
        PublicKey pk2;
        SecretKey sk2;
        RelinKeys rlk2;
        GaloisKeys gk2;
        std::istringstream i1(saved(pk)), i2(saved(sk)), i3(saved(rlk)), i4(saved(gk));
        pk2.load(i1);
        sk2.load(i2);
        rlk2.load(i3);
        gk2.load(i4);
        CHECK(pk2.data == pk.data && pk2.parmsID() == context.keyParmsID());
        CHECK(sk2.data == sk.data && sk2.parmsID() == context.keyParmsID());
        CHECK(rlk2.parmsID() == context.keyParmsID() && rlk2.hasKey(2) && rlk2.size() == 1);
        CHECK(gk2.hasKey(3) && gk2.hasKey((uint32_t)(2 * N - 1)) && !gk2.hasKey(5) && gk2.size() == 2);
        CHECK(saved(pk2) == saved(pk) && saved(sk2) == saved(sk) && saved(rlk2) == saved(rlk) && saved(gk2) == saved(gk));
        // the loaded keys WORK: same ciphertexts under the loaded public key decrypt under the loaded secret key; relinearization and rotation
        // with the loaded keys give the limbs the original keys give
        Encryptor e2(context, pk2, 5, 6);
        Decryptor d2(context, sk2);
        Ciphertext c2 = e2.encrypt(plain);
        CHECK(c2.toHost() == ct.toHost());
        Plaintext da, db;
        decryptor.decrypt(ct, da);
        d2.decrypt(c2, db);
        CHECK(da == db);
        Ciphertext r1 = ct3, r2 = ct3;
        evaluator.relinearizeInplace(r1, rlk);
        evaluator.relinearizeInplace(r2, rlk2);
        CHECK(r1.toHost() == r2.toHost());
        Ciphertext g1 = ct, g2 = ct;
        evaluator.applyGaloisInplace(g1, 3, gk);
        evaluator.applyGaloisInplace(g2, 3, gk2);
        CHECK(g1.toHost() == g2.toHost());
        // a truncated stream is refused, not read past its end
        const std::string cut = saved(rlk).substr(0, saved(rlk).size() / 2);
        CHECK(throws<std::invalid_argument>([&] { std::istringstream t(cut); RelinKeys k; k.load(t); }));
    }
    { // every loader, every truncation point of interest, and absurd size fields: an exception, never a fault or a giant allocation
        const std::string blobs[5] = {saved(ct), saved(pk), saved(sk), saved(rlk), sym_blob};
        for (int which = 0; which < 5; which++) {
            const std::string &b = blobs[which];
            auto load_it = [&](const std::string &bytes) {
                std::istringstream t(bytes);
                if (which == 0) { Ciphertext c; c.load(t, context); }
                else if (which == 1) { PublicKey k; k.load(t); }
                else if (which == 2) { SecretKey k; k.load(t); }
                else if (which == 3) { RelinKeys k; k.load(t); }
                else { Ciphertext c; c.load(t, context); }
            };
            for (size_t cutpos : {size_t(0), size_t(7), size_t(32), size_t(33), size_t(41), size_t(57), size_t(74), size_t(82), b.size() / 3, b.size() - 9, b.size() - 1})
                if (cutpos < b.size()) CHECK(throws<std::invalid_argument>([&] { load_it(b.substr(0, cutpos)); }));
            for (size_t field : {size_t(32), size_t(33), size_t(40), size_t(41), size_t(49), size_t(57)}) { // a size field blown up to 2^61
                if (field + 8 > b.size()) continue;
                std::string bad = b;
                const uint64_t huge = uint64_t(1) << 61;
                std::memcpy(&bad[field], &huge, 8);
                bool refused_or_loaded = true; // some offsets hit a double or a flag in one format and a size in another: it must not crash; a size must be refused
                try { load_it(bad); } catch (const std::invalid_argument &) {} catch (const std::bad_alloc &) { refused_or_loaded = false; } catch (const std::length_error &) { refused_or_loaded = false; }
                CHECK(refused_or_loaded);
            }
        }
    }
}

int main(int argc, char **argv) {
    if (argc < 2) { std::printf("usage: dump_wire <outdir>\n"); return 2; }
    try {
        KernelProvider::initialize();
        run(argv[1], "bfv", SchemeType::bfv);
        run(argv[1], "bgv", SchemeType::bgv);
        run(argv[1], "ckks", SchemeType::ckks);
    } catch (const std::exception &e) {
        std::printf("FAIL exception: %s\n", e.what());
        return 1;
    }
    if (failures) { std::printf("%d FAILED\n", failures); return 1; }
    std::printf("ALL OK\n");
    return 0;
}
