// The operation sequences of the reference's test/timetest.cu (TimeTest::testEncrypt/testAdd/testAddPlain/testMultiplyPlain/testSquare/
// testMemoryPool :93-205, TimeTestCKKS :209-327, TimeTestBFVBGV :331-466) written as the reference writes them -- the same classes, members,
// encoders and Evaluator calls, INCLUDING the out-of-place forms (rotateVector / rotateRows / modSwitchToNext / rescaleToNext with a
// destination) -- compiled against include/troyn.hpp instead of src/troy_cuda.cuh, with assertions on the decrypted slot values where the
// reference has timers.  argv[1] = polynomial degree (the CPU suite runs it small on the emulator build, the GPU suite at 8192).
#include "troyn.hpp"
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

using namespace troyn;
using std::complex;
using std::vector;

static int failures = 0;
#define EXPECT(cond, what)                                                                        \
    do {                                                                                          \
        if (!(cond)) { std::printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__); failures++; } \
        else std::printf("ok   %s\n", what);                                                      \
    } while (0)

namespace troytest {

class TimeTest {
protected:
    Encryptor *encryptor = nullptr;
    Decryptor *decryptor = nullptr;
    Evaluator *evaluator = nullptr;
    SEALContext *context = nullptr;
    RelinKeys rlk;
    PublicKey pk;
    GaloisKeys gk;
    KeyGenerator *keygen = nullptr;
    std::string tag;

public:
    virtual ~TimeTest() {
        delete encryptor; delete evaluator; delete decryptor; delete keygen; delete context;
    }
    virtual Plaintext randomPlaintext() = 0;
    virtual Ciphertext randomCiphertext() = 0;
    // what the slots of a plaintext / ciphertext hold, as doubles (exact integers for BFV / BGV)
    virtual vector<double> slotsOf(const Plaintext &p) = 0;
    vector<double> slotsOf(const Ciphertext &c) { Plaintext p; decryptor->decrypt(c, p); return slotsOf(p); }
    virtual bool close(const vector<double> &a, const vector<double> &b) = 0;
    virtual vector<double> mulSlots(const vector<double> &a, const vector<double> &b) = 0;
    virtual vector<double> addSlots(const vector<double> &a, const vector<double> &b) = 0;
    void expect(bool ok, const char *what) { EXPECT(ok, (tag + " " + what).c_str()); }

    void testEncrypt(int repeatCount = 2) {
        auto p1 = randomPlaintext();
        Ciphertext c2;
        Plaintext p2;
        for (int t = 0; t < repeatCount; t++) {
            encryptor->encrypt(p1, c2);
            decryptor->decrypt(c2, p2);
        }
        expect(close(slotsOf(p1), slotsOf(p2)), "encrypt -> decrypt");
    }
    void testAdd(int repeatCount = 2) {
        auto c1 = randomCiphertext();
        auto c2 = randomCiphertext();
        Ciphertext c3;
        const auto s1 = slotsOf(c1), s2 = slotsOf(c2);
        for (int t = 0; t < repeatCount; t++) {
            evaluator->add(c1, c2, c3);
            evaluator->addInplace(c3, c1);
        }
        expect(close(slotsOf(c3), addSlots(addSlots(s1, s2), s1)), "add / addInplace");
    }
    void testAddPlain(int repeatCount = 2) {
        auto c1 = randomCiphertext();
        auto p2 = randomPlaintext();
        Ciphertext c3;
        for (int t = 0; t < repeatCount; t++) {
            evaluator->addPlain(c1, p2, c3);
            evaluator->addPlainInplace(c3, p2);
        }
        expect(close(slotsOf(c3), addSlots(addSlots(slotsOf(c1), slotsOf(p2)), slotsOf(p2))), "addPlain / addPlainInplace");
    }
    void testMultiplyPlain(int repeatCount = 1) {
        auto c1 = randomCiphertext();
        auto p2 = randomPlaintext();
        Ciphertext c3;
        for (int t = 0; t < repeatCount; t++) {
            evaluator->multiplyPlain(c1, p2, c3);
            evaluator->multiplyPlainInplace(c3, p2);
        }
        const auto sp = slotsOf(p2);
        expect(close(slotsOf(c3), mulSlots(mulSlots(slotsOf(c1), sp), sp)), "multiplyPlain / multiplyPlainInplace");
    }
    void testSquare(int repeatCount = 2) {
        auto c1 = randomCiphertext();
        Ciphertext c2, c3;
        for (int t = 0; t < repeatCount; t++) {
            evaluator->square(c1, c2);
            c3 = c1;
            evaluator->squareInplace(c3);
        }
        const auto s1 = slotsOf(c1);
        expect(c2.size() == 3 && c3.size() == 3, "square gives size 3");
        expect(close(slotsOf(c2), mulSlots(s1, s1)), "square");
        expect(close(slotsOf(c3), mulSlots(s1, s1)), "squareInplace");
    }
    void testMemoryPool(int repeatCount = 8) {
        auto c1 = randomCiphertext();
        Ciphertext c2;
        for (int t = 0; t < repeatCount; t++) evaluator->square(c1, c2);
        const auto first = slotsOf(c2);
        for (int t = 0; t < repeatCount; t++) {
            Ciphertext c3;
            evaluator->square(c1, c3);
            if (t + 1 == repeatCount) expect(close(slotsOf(c3), first), "square into fresh ciphertexts (memory pool reuse)");
        }
    }
};

class TimeTestCKKS : public TimeTest {
    CKKSEncoder *encoder = nullptr;
    size_t slotCount;
    int dataBound;
    double delta;

public:
    TimeTestCKKS(size_t polyModulusDegree, vector<int> qs, int dataBound = 1 << 6, double delta = static_cast<double>(1 << 16)) {
        KernelProvider::initialize();
        tag = "ckks";
        slotCount = polyModulusDegree / 2;
        this->dataBound = dataBound;
        this->delta = delta;
        EncryptionParameters parms(SchemeType::ckks);
        parms.setPolyModulusDegree(polyModulusDegree);
        parms.setCoeffModulus(CoeffModulus::Create(polyModulusDegree, qs));
        context = new SEALContext(parms, true, SecurityLevel::none);
        keygen = new KeyGenerator(*context);
        keygen->createPublicKey(pk);
        keygen->createRelinKeys(rlk);
        keygen->createGaloisKeys(gk);
        encoder = new CKKSEncoder(*context);
        encryptor = new Encryptor(*context, pk);
        decryptor = new Decryptor(*context, keygen->secretKey());
        evaluator = new Evaluator(*context);
    }
    ~TimeTestCKKS() override { delete encoder; }

    static vector<complex<double>> randomVector(size_t count, int data_bound) {
        vector<complex<double>> input(count, 0.0);
        for (size_t i = 0; i < count; i++) input[i] = static_cast<double>(rand() % data_bound);
        return input;
    }
    Plaintext randomPlaintext() override {
        auto p = randomVector(slotCount, dataBound);
        Plaintext ret;
        encoder->encode(p, delta, ret);
        return ret;
    }
    Ciphertext randomCiphertext() override {
        auto r = randomPlaintext();
        Ciphertext ret;
        encryptor->encrypt(r, ret);
        return ret;
    }
    vector<double> slotsOf(const Plaintext &p) override {
        vector<complex<double>> v;
        encoder->decode(p, v);
        vector<double> r(v.size());
        for (size_t i = 0; i < v.size(); i++) r[i] = v[i].real();
        return r;
    }
    using TimeTest::slotsOf;
    bool close(const vector<double> &a, const vector<double> &b) override {
        if (a.size() != b.size()) return false;
        double worst = 0, scale = 1;
        for (size_t i = 0; i < a.size(); i++) { worst = std::max(worst, std::fabs(a[i] - b[i])); scale = std::max(scale, std::fabs(b[i])); }
        if (worst > 2e-2 * scale) std::printf("   (worst error %.4g against magnitude %.4g)\n", worst, scale);
        return worst <= 2e-2 * scale; // delta = 2^16: the encoding alone carries about 2^-10 of noise per slot at these sizes
    }
    vector<double> mulSlots(const vector<double> &a, const vector<double> &b) override { vector<double> r(a.size()); for (size_t i = 0; i < a.size(); i++) r[i] = a[i] * b[i]; return r; }
    vector<double> addSlots(const vector<double> &a, const vector<double> &b) override { vector<double> r(a.size()); for (size_t i = 0; i < a.size(); i++) r[i] = a[i] + b[i]; return r; }

    void testEncode() {
        auto m1 = randomVector(slotCount, dataBound);
        vector<complex<double>> m2;
        Plaintext p1;
        encoder->encode(m1, delta, p1);
        encoder->decode(p1, m2);
        double worst = 0;
        for (size_t i = 0; i < slotCount; i++) worst = std::max(worst, std::abs(m1[i] - m2[i]));
        expect(m2.size() == slotCount && worst < 1e-2, "encode -> decode");
    }
    void testMultiplyRescale() {
        auto c1 = randomCiphertext();
        auto c2 = randomCiphertext();
        Ciphertext c3, c4, c5;
        evaluator->multiply(c1, c2, c3);
        evaluator->rescaleToNext(c3, c4);
        c5 = c1;
        evaluator->multiplyInplace(c5, c2);
        evaluator->rescaleToNextInplace(c5);
        const auto want = mulSlots(slotsOf(c1), slotsOf(c2));
        expect(c4.coeffModulusSize() + 1 == c3.coeffModulusSize() && c4.parmsID() == c5.parmsID(), "rescaleToNext drops one prime");
        expect(close(slotsOf(c4), want), "multiply -> rescaleToNext (destination forms)");
        expect(close(slotsOf(c5), want), "multiplyInplace -> rescaleToNextInplace");
        // relinearized twin of the same product (TimeTest registers these timers as "Relinearize-*")
        Ciphertext c6;
        evaluator->relinearize(c3, rlk, c6);
        evaluator->rescaleToNextInplace(c6);
        expect(c6.size() == 2 && close(slotsOf(c6), want), "multiply -> relinearize -> rescaleToNextInplace");
    }
    void testRotateVector() {
        auto c1 = randomCiphertext();
        Ciphertext c2;
        const auto s1 = slotsOf(c1);
        evaluator->rotateVector(c1, 1, gk, c2);
        evaluator->rotateVectorInplace(c1, 1, gk);
        vector<double> want(s1.size());
        for (size_t i = 0; i < s1.size(); i++) want[i] = s1[(i + 1) % s1.size()];
        expect(close(slotsOf(c2), want), "rotateVector (destination form)");
        expect(close(slotsOf(c1), want), "rotateVectorInplace");
        Ciphertext c3;
        evaluator->complexConjugate(c2, gk, c3); // real slots: conjugation changes nothing
        expect(close(slotsOf(c3), want), "complexConjugate (destination form)");
    }
    void testAll() {
        testEncode();
        testEncrypt();
        testAdd();
        testAddPlain();
        testMultiplyRescale();
        testMultiplyPlain();
        testSquare();
        testRotateVector();
        testMemoryPool();
    }
};

class TimeTestBFVBGV : public TimeTest {
    BatchEncoder *encoder = nullptr;
    size_t slotCount;
    int dataBound;
    uint64_t t = 0;

public:
    TimeTestBFVBGV(bool bgv, size_t polyModulusDegree, uint64_t plainModulusBitSize, vector<int> qs, int dataBound = 1 << 6) {
        KernelProvider::initialize();
        tag = bgv ? "bgv" : "bfv";
        slotCount = polyModulusDegree; // the reference fills N / 2 of the N slots (timetest.cu:339)
        this->dataBound = dataBound;
        EncryptionParameters parms(bgv ? SchemeType::bgv : SchemeType::bfv);
        parms.setPolyModulusDegree(polyModulusDegree);
        parms.setPlainModulus(PlainModulus::Batching(polyModulusDegree, (int)plainModulusBitSize));
        parms.setCoeffModulus(CoeffModulus::Create(polyModulusDegree, qs));
        t = parms.plainModulus().value();
        context = new SEALContext(parms, true, SecurityLevel::none);
        keygen = new KeyGenerator(*context);
        keygen->createPublicKey(pk);
        keygen->createRelinKeys(rlk);
        keygen->createGaloisKeys(gk);
        encoder = new BatchEncoder(*context);
        encryptor = new Encryptor(*context, pk);
        decryptor = new Decryptor(*context, keygen->secretKey());
        evaluator = new Evaluator(*context);
    }
    ~TimeTestBFVBGV() override { delete encoder; }

    static vector<int64_t> randomVector(size_t count, int data_bound) {
        vector<int64_t> input(count, 0);
        for (size_t i = 0; i < count; i++) input[i] = rand() % data_bound;
        return input;
    }
    Plaintext randomPlaintext() override {
        auto p = randomVector(slotCount / 2, dataBound);
        Plaintext ret;
        encoder->encode(p, ret);
        return ret;
    }
    Ciphertext randomCiphertext() override {
        auto r = randomPlaintext();
        Ciphertext ret;
        encryptor->encrypt(r, ret);
        return ret;
    }
    vector<double> slotsOf(const Plaintext &p) override {
        vector<uint64_t> v;
        encoder->decode(p, v);
        return vector<double>(v.begin(), v.end());
    }
    using TimeTest::slotsOf;
    bool close(const vector<double> &a, const vector<double> &b) override { return a == b; } // integers: exact
    vector<double> mulSlots(const vector<double> &a, const vector<double> &b) override {
        vector<double> r(a.size());
        for (size_t i = 0; i < a.size(); i++) r[i] = (double)(uint64_t)((unsigned __int128)(uint64_t)a[i] * (uint64_t)b[i] % t);
        return r;
    }
    vector<double> addSlots(const vector<double> &a, const vector<double> &b) override {
        vector<double> r(a.size());
        for (size_t i = 0; i < a.size(); i++) r[i] = (double)(((uint64_t)a[i] + (uint64_t)b[i]) % t);
        return r;
    }

    void testEncode() {
        auto m1 = randomVector(slotCount / 2, dataBound);
        vector<int64_t> m2;
        Plaintext p1;
        encoder->encode(m1, p1);
        encoder->decode(p1, m2);
        bool same = m2.size() == slotCount;
        for (size_t i = 0; same && i < slotCount; i++) same = m2[i] == (i < m1.size() ? m1[i] : 0);
        expect(same, "encode -> decode");
        vector<int64_t> neg{-3, 5, -7};
        encoder->encode(neg, p1);
        encoder->decode(p1, m2);
        expect(m2[0] == -3 && m2[1] == 5 && m2[2] == -7 && m2[3] == 0, "signed encode -> decode");
    }
    void testMultiplyRescale() {
        auto c1 = randomCiphertext();
        auto c2 = randomCiphertext();
        Ciphertext c3, c4, c5;
        evaluator->multiply(c1, c2, c3);
        evaluator->modSwitchToNext(c3, c4);
        c5 = c1;
        evaluator->multiplyInplace(c5, c2);
        evaluator->modSwitchToNextInplace(c5);
        const auto want = mulSlots(slotsOf(c1), slotsOf(c2));
        expect(c4.coeffModulusSize() + 1 == c3.coeffModulusSize() && c4.parmsID() == c5.parmsID(), "modSwitchToNext drops one prime");
        expect(close(slotsOf(c4), want), "multiply -> modSwitchToNext (destination forms)");
        expect(close(slotsOf(c5), want), "multiplyInplace -> modSwitchToNextInplace");
        Ciphertext c6;
        evaluator->relinearize(c3, rlk, c6);
        expect(c6.size() == 2 && close(slotsOf(c6), want), "multiply -> relinearize (destination form)");
        Ciphertext c7;
        evaluator->modSwitchTo(c6, context->lastParmsID(), c7);
        expect(c7.parmsID() == context->lastParmsID() && close(slotsOf(c7), want), "modSwitchTo(last level) (destination form)");
    }
    void testRotateVector() {
        auto c1 = randomCiphertext();
        Ciphertext c2;
        const auto s1 = slotsOf(c1);
        evaluator->rotateRows(c1, 1, gk, c2);
        evaluator->rotateRowsInplace(c1, 1, gk);
        const size_t row = s1.size() / 2;
        vector<double> want(s1.size());
        for (size_t i = 0; i < row; i++) { want[i] = s1[(i + 1) % row]; want[row + i] = s1[row + (i + 1) % row]; }
        expect(close(slotsOf(c2), want), "rotateRows (destination form)");
        expect(close(slotsOf(c1), want), "rotateRowsInplace");
        Ciphertext c3;
        evaluator->rotateColumns(c2, gk, c3);
        vector<double> swapped(s1.size());
        for (size_t i = 0; i < row; i++) { swapped[i] = want[row + i]; swapped[row + i] = want[i]; }
        expect(close(slotsOf(c3), swapped), "rotateColumns (destination form)");
        Ciphertext c4, c5;
        evaluator->transformToNtt(c3, c4);
        evaluator->transformFromNtt(c4, c5);
        expect(c4.isNttForm() && !c5.isNttForm() && close(slotsOf(c5), swapped), "transformToNtt -> transformFromNtt (destination forms)");
    }
    void testAll() {
        testEncode();
        testEncrypt();
        testAdd();
        testAddPlain();
        testMultiplyRescale();
        testMultiplyPlain();
        testSquare();
        testRotateVector();
        testMemoryPool();
    }
};

} // namespace troytest

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)std::atol(argv[1]) : 8192;
    try {
        std::printf("----- CKKS -----\n");
        troytest::TimeTestCKKS test(n, {60, 40, 40, 40, 40, 60}, 1 << 6, static_cast<double>(1ull << 30));
        test.testAll();
        std::printf("----- BFV -----\n");
        troytest::TimeTestBFVBGV test2(false, n, 20, {60, 40, 40, 40, 40, 60});
        test2.testAll();
        std::printf("----- BGV -----\n");
        troytest::TimeTestBFVBGV test3(true, n, 20, {60, 40, 40, 40, 40, 60});
        test3.testAll();
    } catch (const std::exception &e) {
        std::printf("FAIL exception: %s\n", e.what());
        failures++;
    }
    std::printf(failures ? "FAILED %d\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
