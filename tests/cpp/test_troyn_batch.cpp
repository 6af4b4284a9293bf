// The slab-batched forms of troyn::Evaluator (include/troyn.hpp, "the hot path, batched") against the per-ciphertext forms of the reference's
// interface (src/evaluator_cuda.cuh:85-115,193-198,292-344): every batched call must leave, limb for limb, what the loop over single
// ciphertexts leaves -- for operands that are slab members (used where they lie) and for operands that are scattered allocations (packed
// first), in BFV, BGV and CKKS.  argv[1] = polynomial degree (small on the emulator build, 8192 on the GPU), argv[2] = batch size.
#include "troyn.hpp"
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace troyn;
using std::vector;

static int failures = 0;
#define EXPECT(cond, what)                                                                        \
    do {                                                                                          \
        if (!(cond)) { std::printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__); failures++; } \
        else std::printf("ok   %s\n", what);                                                      \
    } while (0)

static bool same(const vector<Ciphertext> &a, const vector<Ciphertext> &b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++) {
        if (a[i].size() != b[i].size() || a[i].coeffModulusSize() != b[i].coeffModulusSize() || a[i].isNttForm() != b[i].isNttForm() || a[i].scale() != b[i].scale() ||
            a[i].correctionFactor() != b[i].correctionFactor() || a[i].parmsID() != b[i].parmsID())
            return false;
        if (a[i].toHost() != b[i].toHost()) return false;
    }
    return true;
}
static bool isRun(const vector<Ciphertext> &v) { return Ciphertext::isRun(Ciphertext::pointers(v)); }

struct Fixture {
    SchemeType scheme;
    SEALContext *context = nullptr;
    KeyGenerator *keygen = nullptr;
    Encryptor *encryptor = nullptr;
    Decryptor *decryptor = nullptr;
    Evaluator *evaluator = nullptr;
    BatchEncoder *batch = nullptr;
    CKKSEncoder *ckks = nullptr;
    RelinKeys rlk;
    GaloisKeys gk;
    PublicKey pk;
    size_t n;
    std::mt19937_64 rng{7};
    double delta = (double)(1ull << 30);

    Fixture(SchemeType s, size_t n_, vector<int> bits) : scheme(s), n(n_) {
        EncryptionParameters parms(s);
        parms.setPolyModulusDegree(n);
        parms.setCoeffModulus(CoeffModulus::Create(n, bits));
        if (s != SchemeType::ckks) parms.setPlainModulus(PlainModulus::Batching(n, 20));
        context = new SEALContext(parms, true, SecurityLevel::none);
        keygen = new KeyGenerator(*context);
        keygen->createPublicKey(pk);
        keygen->createRelinKeys(rlk);
        keygen->createGaloisKeys(std::vector<int>{1, -1, 4}, gk);
        encryptor = new Encryptor(*context, pk);
        decryptor = new Decryptor(*context, keygen->secretKey());
        evaluator = new Evaluator(*context);
        if (s == SchemeType::ckks) ckks = new CKKSEncoder(*context);
        else batch = new BatchEncoder(*context);
    }
    ~Fixture() { delete batch; delete ckks; delete evaluator; delete decryptor; delete encryptor; delete keygen; delete context; }
    Plaintext plain() {
        Plaintext p;
        if (ckks) {
            vector<std::complex<double>> v(n / 2);
            for (auto &x : v) x = (double)(rng() % 64);
            ckks->encode(v, delta, p);
        } else {
            vector<uint64_t> v(n);
            for (auto &x : v) x = rng() % 64;
            batch->encode(v, p);
        }
        return p;
    }
    vector<Ciphertext> fresh(size_t count) { // scattered: every ciphertext its own allocation
        vector<Ciphertext> v(count);
        for (auto &c : v) encryptor->encrypt(plain(), c);
        return v;
    }
};

static void scenario(SchemeType scheme, size_t n, size_t B) {
    const char *tag = scheme == SchemeType::bfv ? "bfv" : scheme == SchemeType::bgv ? "bgv" : "ckks";
    std::printf("-- %s N=%zu batch %zu\n", tag, n, B);
    Fixture f(scheme, n, {50, 40, 40, 50});
    const Evaluator &ev = *f.evaluator;
    const bool ckks = scheme == SchemeType::ckks;
    auto what = [&](const char *s) { static std::string keep; keep = std::string(tag) + " " + s; return keep.c_str(); };

    const vector<Ciphertext> a = f.fresh(B), b = f.fresh(B);
    const vector<Ciphertext> sa = Ciphertext::packBatch(a), sb = Ciphertext::packBatch(b);
    EXPECT(isRun(sa) && !isRun(a) && same(sa, a), what("packBatch keeps the ciphertexts; its members are a run, scattered ones are not"));

    // multiply / square: loop of the reference's calls against one batched call, slab operands and scattered operands
    vector<Ciphertext> prod(B), sq(B);
    for (size_t i = 0; i < B; i++) { ev.multiply(a[i], b[i], prod[i]); ev.square(a[i], sq[i]); }
    const vector<Ciphertext> prod_s = ev.multiplyBatch(sa, sb), prod_p = ev.multiplyBatch(a, b);
    EXPECT(same(prod_s, prod) && isRun(prod_s), what("multiplyBatch (slab operands) == loop of multiply; the result is a slab"));
    EXPECT(same(prod_p, prod), what("multiplyBatch (scattered operands, packed on the way)"));
    EXPECT(same(ev.squareBatch(sa), sq) && same(ev.squareBatch(a), sq), what("squareBatch == loop of square"));
    {
        vector<Ciphertext> inpl = a;
        ev.multiplyInplaceBatch(Ciphertext::pointers(inpl), Ciphertext::pointers(b));
        EXPECT(same(inpl, prod) && isRun(inpl), what("multiplyInplaceBatch: the items become members of the result slab"));
    }

    // relinearize: destination form over the batch, in place on a slab, in place on scattered items
    vector<Ciphertext> rel(B);
    for (size_t i = 0; i < B; i++) ev.relinearize(prod[i], f.rlk, rel[i]);
    EXPECT(same(ev.relinearizeBatch(prod_s, f.rlk), rel) && same(ev.relinearizeBatch(prod, f.rlk), rel), what("relinearizeBatch == loop of relinearize"));
    vector<Ciphertext> rel_s = ev.multiplyBatch(sa, sb), rel_p = prod;
    ev.relinearizeInplaceBatch(rel_s, f.rlk);
    ev.relinearizeInplaceBatch(rel_p, f.rlk);
    EXPECT(same(rel_s, rel) && isRun(rel_s), what("relinearizeInplaceBatch on a slab (stays a run: members use 2 of 3 polynomials)"));
    EXPECT(same(rel_p, rel) && isRun(rel_p), what("relinearizeInplaceBatch on scattered items (packed, items become slab members)"));

    // the next level: modSwitchToNext (all schemes), rescaleToNext (CKKS), from the strided run the in-place relinearize left
    vector<Ciphertext> nxt(B);
    for (size_t i = 0; i < B; i++) {
        if (ckks) ev.rescaleToNext(rel[i], nxt[i]); else ev.modSwitchToNext(rel[i], nxt[i]);
    }
    const vector<Ciphertext> nxt_s = ckks ? ev.rescaleToNextBatch(rel_s) : ev.modSwitchToNextBatch(rel_s);
    EXPECT(same(nxt_s, nxt) && isRun(nxt_s), what(ckks ? "rescaleToNextBatch == loop of rescaleToNext" : "modSwitchToNextBatch == loop of modSwitchToNext"));
    {
        vector<Ciphertext> inpl = rel;
        if (ckks) ev.rescaleToNextInplaceBatch(inpl); else ev.modSwitchToNextInplaceBatch(inpl);
        EXPECT(same(inpl, nxt), what("...InplaceBatch on scattered items"));
        if (ckks) {
            vector<Ciphertext> drop(B);
            for (size_t i = 0; i < B; i++) ev.modSwitchToNext(rel[i], drop[i]);
            EXPECT(same(ev.modSwitchToNextBatch(rel), drop), what("modSwitchToNextBatch (CKKS: drops the last prime)"));
        }
    }

    // rotations through the NAF path (step 3 = 4 - 1 with keys {1, -1, 4}), column swap / conjugation, one Galois element
    {
        vector<Ciphertext> want = nxt, got = nxt_s, got_p = nxt;
        for (auto &c : want) { if (ckks) ev.rotateVectorInplace(c, 3, f.gk); else ev.rotateRowsInplace(c, 3, f.gk); }
        if (ckks) { ev.rotateVectorInplaceBatch(got, 3, f.gk); ev.rotateVectorInplaceBatch(got_p, 3, f.gk); }
        else { ev.rotateRowsInplaceBatch(got, 3, f.gk); ev.rotateRowsInplaceBatch(got_p, 3, f.gk); }
        EXPECT(same(got, want) && same(got_p, want), what("rotate...InplaceBatch(3) == loop (slab and scattered)"));
        uint32_t e1 = 0;
        check(troyhip_galois_elt_from_step(f.context->handle(), 1, &e1));
        for (auto &c : want) ev.applyGaloisInplace(c, e1, f.gk);
        ev.applyGaloisInplaceBatch(got, e1, f.gk);
        EXPECT(same(got, want), what("applyGaloisInplaceBatch == loop of applyGaloisInplace"));
        bool threw = false;
        try { if (ckks) ev.rotateRowsInplaceBatch(got, 1, f.gk); else ev.rotateVectorInplaceBatch(got, 1, f.gk); } catch (const std::logic_error &) { threw = true; }
        EXPECT(threw, what("the other scheme's rotation -> logic_error, as the single form"));
    }

    // element-wise, transforms and plaintext operands over a batch
    {
        vector<Ciphertext> want = a, got = sa, got_p = a;
        for (size_t i = 0; i < B; i++) { ev.addInplace(want[i], b[i]); ev.negateInplace(want[i]); ev.subInplace(want[i], b[i]); }
        ev.addInplaceBatch(Ciphertext::pointers(got), Ciphertext::pointers(sb));
        ev.negateInplaceBatch(Ciphertext::pointers(got));
        ev.subInplaceBatch(Ciphertext::pointers(got), Ciphertext::pointers(b));
        ev.addInplaceBatch(Ciphertext::pointers(got_p), Ciphertext::pointers(b));
        ev.negateInplaceBatch(Ciphertext::pointers(got_p));
        ev.subInplaceBatch(Ciphertext::pointers(got_p), Ciphertext::pointers(sb));
        EXPECT(same(got, want) && same(got_p, want), what("add / negate / sub InplaceBatch == loop"));
        const Plaintext p = f.plain();
        Plaintext addend = p;
        if (ckks) { // the addend has to carry the product's scale, as a caller of the single form would arrange
            vector<std::complex<double>> ones(n / 2, 1.0);
            f.ckks->encode(ones, a[0].scale() * p.scale(), addend);
        }
        for (auto &c : want) { ev.multiplyPlainInplace(c, p); ev.addPlainInplace(c, addend); }
        ev.multiplyPlainInplaceBatch(Ciphertext::pointers(got), p);
        ev.addPlainInplaceBatch(Ciphertext::pointers(got), addend);
        EXPECT(same(got, want), what("multiplyPlainInplaceBatch / addPlainInplaceBatch == loop"));
        if (!ckks) {
            for (auto &c : want) ev.transformToNttInplace(c);
            ev.transformToNttInplaceBatch(Ciphertext::pointers(got));
            const bool fwd = same(got, want);
            for (auto &c : want) ev.transformFromNttInplace(c);
            ev.transformFromNttInplaceBatch(Ciphertext::pointers(got));
            EXPECT(fwd && same(got, want), what("transformToNtt / FromNtt InplaceBatch == loop"));
        }
    }

    // a list of mixed shapes: the in-place forms group consecutive items of one shape (sizes 3, 3, 2, 2, 3)
    if (B >= 3) {
        vector<Ciphertext> mixed = {prod[0], prod[1], rel[0], rel[1], prod[2]}, want = mixed;
        for (auto &c : want) ev.relinearizeInplace(c, f.rlk);
        ev.relinearizeInplaceBatch(mixed, f.rlk);
        EXPECT(same(mixed, want), what("relinearizeInplaceBatch over mixed sizes (grouped by shape)"));
        bool threw = false;
        try { ev.multiplyBatch(vector<const Ciphertext *>{&prod[0], &rel[0]}, vector<const Ciphertext *>{&a[0], &a[1]}); } catch (const std::invalid_argument &) { threw = true; }
        EXPECT(threw, what("multiplyBatch over different shapes -> invalid_argument"));
    }
    // the decrypted product is the product: the batch path end to end (BFV / BGV exact; CKKS within the encoding's noise)
    {
        Plaintext pa, pb, pr;
        f.decryptor->decrypt(a[B - 1], pa);
        f.decryptor->decrypt(b[B - 1], pb);
        f.decryptor->decrypt(rel_s[B - 1], pr);
        bool ok = true;
        if (ckks) {
            vector<std::complex<double>> va, vb, vr;
            f.ckks->decode(pa, va); f.ckks->decode(pb, vb); f.ckks->decode(pr, vr);
            for (size_t i = 0; i < va.size(); i++) ok = ok && std::abs(vr[i] - va[i] * vb[i]) < 1e-2 * (1.0 + std::abs(va[i] * vb[i]));
        } else {
            vector<uint64_t> va, vb, vr;
            f.batch->decode(pa, va); f.batch->decode(pb, vb); f.batch->decode(pr, vr);
            const uint64_t t = f.context->firstContextData()->parms().plainModulus().value();
            for (size_t i = 0; i < va.size(); i++) ok = ok && vr[i] == va[i] * vb[i] % t;
        }
        EXPECT(ok, what("multiplyBatch -> relinearizeInplaceBatch decrypts to the slot-wise product"));
    }
}

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)std::atol(argv[1]) : 8192, B = argc > 2 ? (size_t)std::atol(argv[2]) : 5;
    try {
        KernelProvider::initialize();
        scenario(SchemeType::bfv, n, B);
        scenario(SchemeType::bgv, n, B);
        scenario(SchemeType::ckks, n, B);
    } catch (const std::exception &e) {
        std::printf("FAIL exception: %s\n", e.what());
        failures++;
    }
    std::printf(failures ? "FAILED %d\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
