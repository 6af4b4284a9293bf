// User code of the reference's BFV application layer, compiled with plain g++ against include/troyn.hpp + include/troyn_linear.hpp
// and linked to libtroyhip.so.  The flows are those of the reference's own test/app/linear.cu (testMatmulInts with and without LWE
// packing, testMatmulCipherInts, testConv2dInt: same call sequence on the same helper API, fixture of its LinearTest class with a
// power-of-two plain modulus), here with exact comparisons instead of a printed difference, plus the pieces of troyn:: those flows
// lean on: BatchEncoder::encodePolynomial / decodePolynomial (src/batchencoder_cuda.cu:124-170, 267-286), Plaintext save / load
// (src/plaintext_cuda.cu:7-27), KeyGenerator::createAutomorphismKeys (src/keygenerator.cpp:350-358).
// usage: test_troyn_linear [N = 4096]
#include "troyn_linear.hpp"
#include <cstdio>
#include <cstdlib>
#include <sstream>

using namespace troyn;
using std::vector;

static int failures = 0;
#define EXPECT(cond, what)                                              \
    do {                                                                \
        if (!(cond)) { std::printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__); failures++; } \
        else std::printf("ok   %s\n", what);                            \
    } while (0)
template <class E, class F> static bool throws(F f) {
    try { f(); } catch (const E &) { return true; } catch (...) { return false; }
    return false;
}

struct LinearTest { // test/app/linear.cu:158-187
    size_t slotCount;
    uint64_t modulus;
    std::unique_ptr<SEALContext> context;
    std::unique_ptr<KeyGenerator> keygen;
    std::unique_ptr<BatchEncoder> encoder;
    std::unique_ptr<Encryptor> encryptor;
    std::unique_ptr<Decryptor> decryptor;
    std::unique_ptr<Evaluator> evaluator;
    PublicKey pk;
    RelinKeys rlk;
    GaloisKeys autok;

    LinearTest(size_t polyModulusDegree, vector<int> qs, uint64_t plainModulus) : slotCount(polyModulusDegree), modulus(plainModulus) {
        EncryptionParameters parms(SchemeType::bfv);
        parms.setPolyModulusDegree(polyModulusDegree);
        parms.setPlainModulus(plainModulus);
        parms.setCoeffModulus(CoeffModulus::Create(polyModulusDegree, qs));
        context.reset(new SEALContext(parms, true, SecurityLevel::none));
        keygen.reset(new KeyGenerator(*context));
        keygen->createPublicKey(pk);
        keygen->createRelinKeys(rlk);
        autok = keygen->createAutomorphismKeys();
        encoder.reset(new BatchEncoder(*context));
        encryptor.reset(new Encryptor(*context, pk));
        encryptor->setSecretKey(keygen->secretKey());
        decryptor.reset(new Decryptor(*context, keygen->secretKey()));
        evaluator.reset(new Evaluator(*context));
    }

    vector<uint64_t> randomVector(size_t count) {
        vector<uint64_t> v(count);
        for (auto &x : v) x = (((uint64_t)std::rand() << 31) ^ (uint64_t)std::rand()) % modulus;
        return v;
    }
    uint64_t mulmod(uint64_t a, uint64_t b) const { return (uint64_t)((unsigned __int128)a * b % modulus); }

    // ---- troyn:: pieces under the helpers
    void testPolynomialEncoding() {
        Plaintext p;
        encoder->encodePolynomial(vector<uint64_t>{1, modulus + 5, 7}, p);
        EXPECT(p.coeffCount() == 3 && p[0] == 1 && p[1] == 5 && p[2] == 7 && !p.isNttForm(), "encodePolynomial: values mod t are the coefficients, as many as given");
        vector<uint64_t> back;
        encoder->decodePolynomial(p, back);
        EXPECT(back == (vector<uint64_t>{1, 5, 7}), "decodePolynomial returns them");
        encoder->encodePolynomial(vector<int64_t>{-1, 2, -(int64_t)3}, p);
        EXPECT(p.coeffCount() == slotCount && p[0] == modulus - 1 && p[1] == 2 && p[2] == modulus - 3 && p[3] == 0, "signed form: t - |v|, padded to N");
        vector<int64_t> sback;
        encoder->decodePolynomial(p, sback);
        EXPECT(sback.size() == slotCount && sback[0] == -1 && sback[1] == 2 && sback[2] == -3 && sback[5] == 0, "signed decode is centred");
        EXPECT(throws<std::invalid_argument>([&] { encoder->encodePolynomial(vector<uint64_t>(slotCount + 1, 1), p); }), "more than N values: invalid_argument");

        // a plaintext round trip on the wire, and through an encryption
        vector<uint64_t> v = randomVector(slotCount);
        encoder->encodePolynomial(v, p);
        std::ostringstream out;
        p.save(out);
        EXPECT(out.str().size() == 32 + 8 + 8 + 8 + 8 * slotCount, "Plaintext::save: parms_id, coeff_count, scale, word count, words");
        std::istringstream in(out.str());
        Plaintext q;
        q.load(in);
        EXPECT(q == p && q.coeffCount() == slotCount && !q.isNttForm(), "Plaintext::load restores it");
        Plaintext dec;
        decryptor->decrypt(encryptor->encryptSymmetric(p), dec);
        encoder->decodePolynomial(dec, back);
        EXPECT(back == v, "encryptSymmetric -> decrypt returns the polynomial");
        EXPECT(autok.hasKey((uint32_t)(slotCount + 1)) && autok.hasKey(5) && autok.hasKey(3) && !autok.hasKey(7), "createAutomorphismKeys: X -> X^(N / 2^k + 1), k = 0 .. log N - 1");
    }

    vector<uint64_t> plainMatmul(const vector<uint64_t> &x, const vector<uint64_t> &w, const vector<uint64_t> &s, size_t batchSize, size_t inputDims, size_t outputDims) {
        vector<uint64_t> y(batchSize * outputDims, 0);
        for (size_t i = 0; i < batchSize; i++)
            for (size_t k = 0; k < outputDims; k++) {
                uint64_t acc = s[i * outputDims + k];
                for (size_t j = 0; j < inputDims; j++) acc = (acc + mulmod(x[i * inputDims + j], w[j * outputDims + k])) % modulus;
                y[i * outputDims + k] = acc;
            }
        return y;
    }

    // test/app/linear.cu:213-290 (cipher = false) and :294-372 (cipher = true: the weights are encrypted too, relinearize after the mod switch)
    void testMatmulInts(size_t batchSize, size_t inputDims, size_t outputDims, bool packLwes, bool cipher, const char *what) {
        auto w = randomVector(inputDims * outputDims);
        auto x = randomVector(inputDims * batchSize);
        auto s = randomVector(batchSize * outputDims);
        LinearHelper::MatmulHelper helper(batchSize, inputDims, outputDims, slotCount, 0, packLwes);
        auto wEncoded = helper.encodeWeights(*encoder, w.data());
        if (!cipher) { // the weights travel once (serializeEncodedWeights), LinearHelper.cuh:625-660
            std::ostringstream sout;
            helper.serializeEncodedWeights(wEncoded, sout);
            std::istringstream sin(sout.str());
            auto back = helper.deserializeEncodedWeights(sin);
            bool same = back.data.size() == wEncoded.data.size();
            for (size_t i = 0; same && i < back.data.size(); i++)
                for (size_t j = 0; j < back[i].size(); j++) same = same && back[i][j] == wEncoded[i][j];
            EXPECT(same, "encoded weights survive serializeEncodedWeights / deserializeEncodedWeights");
            wEncoded = std::move(back);
        }
        auto xEncoded = helper.encodeInputs(*encoder, x.data());
        auto xEnc = xEncoded.encrypt(*encryptor);
        { // serialize
            std::ostringstream sout;
            xEnc.save(sout);
            std::istringstream sin(sout.str());
            xEnc = LinearHelper::Cipher2d();
            xEnc.load(sin, *context);
        }
        LinearHelper::Cipher2d yEnc;
        if (cipher) {
            auto wEnc = wEncoded.encrypt(*encryptor);
            yEnc = helper.matmulCipher(*evaluator, xEnc, wEnc);
            yEnc.modSwitchToNext(*evaluator);
            yEnc.relinearize(*evaluator, rlk);
        } else {
            yEnc = helper.matmul(*evaluator, xEnc, wEncoded);
            yEnc.modSwitchToNext(*evaluator);
        }
        const size_t before = yEnc.data.size() * yEnc[0].size();
        if (packLwes) yEnc = helper.packOutputs(*evaluator, autok, yEnc);
        const size_t after = yEnc.data.size() * yEnc[0].size();
        auto sEncoded = helper.encodeOutputs(*encoder, s.data());
        yEnc.addPlainInplace(*evaluator, sEncoded);
        size_t wire = 0;
        { // serialize
            std::ostringstream sout;
            helper.serializeOutputs(*evaluator, yEnc, sout);
            wire = sout.str().size();
            std::istringstream sin(sout.str());
            yEnc = helper.deserializeOutputs(*evaluator, sin);
        }
        auto yDec = helper.decryptOutputs(*encoder, *decryptor, yEnc);
        std::printf("     %s: %zu result ciphertexts -> %zu on the wire, %zu bytes\n", what, before, after, wire);
        EXPECT(yDec == plainMatmul(x, w, s, batchSize, inputDims, outputDims), what);
        if (packLwes) EXPECT(after < before || before == 1, "packOutputs folds the results");
    }

    // test/app/linear.cu:458-558
    void testConv2dInt(size_t batchSize, size_t inputChannels, size_t outputChannels, size_t imageHeight, size_t imageWidth, size_t kernelHeight, size_t kernelWidth,
                       const char *what) {
        auto weights = randomVector(inputChannels * outputChannels * kernelHeight * kernelWidth);
        auto x = randomVector(batchSize * inputChannels * imageHeight * imageWidth);
        const size_t yh = imageHeight - kernelHeight + 1, yw = imageWidth - kernelWidth + 1;
        auto s = randomVector(batchSize * outputChannels * yh * yw);
        LinearHelper::Conv2dHelper helper(batchSize, imageHeight, imageWidth, kernelHeight, kernelWidth, inputChannels, outputChannels, slotCount);
        auto encodedWeights = helper.encodeWeights(*encoder, weights);
        auto xEnc = helper.encryptInputs(*encryptor, *encoder, x);
        { // serialize
            std::ostringstream sout;
            xEnc.save(sout);
            std::istringstream sin(sout.str());
            xEnc = LinearHelper::Cipher2d();
            xEnc.load(sin, *context);
        }
        auto yEnc = helper.conv2d(*evaluator, xEnc, encodedWeights);
        auto sEncoded = helper.encodeOutputs(*encoder, s);
        yEnc.addPlainInplace(*evaluator, sEncoded);
        { // serialize
            std::ostringstream sout;
            helper.serializeOutputs(*evaluator, yEnc, sout);
            std::istringstream sin(sout.str());
            yEnc = helper.deserializeOutputs(*evaluator, sin);
        }
        auto yDec = helper.decryptOutputs(*encoder, *decryptor, yEnc);
        vector<uint64_t> y(batchSize * outputChannels * yh * yw, 0);
        for (size_t b = 0; b < batchSize; b++)
            for (size_t oc = 0; oc < outputChannels; oc++)
                for (size_t yi = 0; yi < yh; yi++)
                    for (size_t yj = 0; yj < yw; yj++) {
                        uint64_t element = s[((b * outputChannels + oc) * yh + yi) * yw + yj];
                        for (size_t ic = 0; ic < inputChannels; ic++)
                            for (size_t ki = 0; ki < kernelHeight; ki++)
                                for (size_t kj = 0; kj < kernelWidth; kj++)
                                    element = (element + mulmod(x[((b * inputChannels + ic) * imageHeight + yi + ki) * imageWidth + yj + kj],
                                                                weights[((oc * inputChannels + ic) * kernelHeight + ki) * kernelWidth + kj])) % modulus;
                        y[((b * outputChannels + oc) * yh + yi) * yw + yj] = element;
                    }
        std::printf("     %s: %zu x %zu result ciphertexts\n", what, yEnc.data.size(), helper.getTotalBatchSize() ? yEnc[0].size() : (size_t)0);
        EXPECT(yDec == y, what);
    }

    // the other operand placements (LinearHelper.cuh:470-492, 973-1033): plain inputs x encrypted weights, and every objective of the block search
    void testReverseAndObjectives() {
        const size_t batchSize = 3, inputDims = 20, outputDims = 12;
        auto w = randomVector(inputDims * outputDims), x = randomVector(inputDims * batchSize), zero = vector<uint64_t>(batchSize * outputDims, 0);
        const vector<uint64_t> expect = plainMatmul(x, w, zero, batchSize, inputDims, outputDims);
        for (int objective = 0; objective < 3; objective++)
            for (bool pack : {false, true}) {
                LinearHelper::MatmulHelper helper(batchSize, inputDims, outputDims, slotCount, objective, pack);
                auto wEnc = helper.encodeWeights(*encoder, w.data()).encrypt(*encryptor);
                auto yEnc = helper.matmulReverse(*evaluator, helper.encodeInputs(*encoder, x.data()), wEnc);
                if (pack) yEnc = helper.packOutputs(*evaluator, autok, yEnc);
                char what[96];
                std::snprintf(what, sizeof what, "matmulReverse (plain inputs, encrypted weights), objective %d%s", objective, pack ? ", packed" : "");
                EXPECT(helper.decryptOutputs(*encoder, *decryptor, yEnc) == expect, what);
            }
        EXPECT(throws<std::runtime_error>([&] { LinearHelper::MatmulHelper(2, 4, 4, slotCount, 7, false); }), "an unknown objective: runtime_error");
        // conv2d with encrypted weights, and with plain inputs x encrypted weights
        const size_t ic = 2, oc = 2, ih = 7, iw = 6, kh = 3, kw = 2, yh = ih - kh + 1, yw = iw - kw + 1;
        auto cw = randomVector(ic * oc * kh * kw), cx = randomVector(ic * ih * iw);
        vector<uint64_t> cy(oc * yh * yw, 0);
        for (size_t o = 0; o < oc; o++)
            for (size_t i = 0; i < yh; i++)
                for (size_t j = 0; j < yw; j++) {
                    uint64_t acc = 0;
                    for (size_t c = 0; c < ic; c++)
                        for (size_t a = 0; a < kh; a++)
                            for (size_t b = 0; b < kw; b++) acc = (acc + mulmod(cx[(c * ih + i + a) * iw + j + b], cw[((o * ic + c) * kh + a) * kw + b])) % modulus;
                    cy[(o * yh + i) * yw + j] = acc;
                }
        LinearHelper::Conv2dHelper conv(1, ih, iw, kh, kw, ic, oc, slotCount);
        auto cwEnc = conv.encodeWeights(*encoder, cw).encrypt(*encryptor);
        auto viaCipher = conv.conv2dCipher(*evaluator, conv.encryptInputs(*encryptor, *encoder, cx), cwEnc);
        viaCipher.relinearize(*evaluator, rlk);
        EXPECT(conv.decryptOutputs(*encoder, *decryptor, viaCipher) == cy, "conv2dCipher (both operands encrypted), relinearized");
        auto viaReverse = conv.conv2dReverse(*evaluator, conv.encodeInputs(*encoder, cx), cwEnc);
        EXPECT(conv.decryptOutputs(*encoder, *decryptor, viaReverse) == cy, "conv2dReverse (plain inputs, encrypted weights)");
    }

    void testCipher2d() { // the element-wise members of Cipher2d (LinearHelper.cuh:104-207)
        LinearHelper::MatmulHelper helper(2, 8, 4, slotCount, 0, false);
        auto x = randomVector(16), z = randomVector(16);
        auto a = helper.encodeInputs(*encoder, x.data()), b = helper.encodeInputs(*encoder, z.data());
        auto ca = a.encrypt(*encryptor), cb = b.encrypt(*encryptor);
        ca.addInplace(*evaluator, cb);
        auto cc = ca.addPlain(*evaluator, b);
        cc.multiplyScalarInplace(*encoder, *evaluator, 3);
        Plaintext p;
        decryptor->decrypt(cc[0][0], p);
        vector<uint64_t> coeffs;
        encoder->decodePolynomial(p, coeffs);
        bool good = true;
        for (size_t j = 0; j < 8; j++) good = good && coeffs[j] == (unsigned __int128)(x[j] + 2 * (unsigned __int128)z[j]) * 3 % modulus;
        EXPECT(good, "Cipher2d addInplace / addPlain / multiplyScalarInplace");
        // switch_key (LinearHelper.cuh:128-137): a grid encrypted under ANOTHER secret key, taken to ours by the key our generator makes for
        // it (KeyGenerator::createKeySwitchingKeys, src/keygenerator.cpp:360-366)
        KeyGenerator other(*context);
        Encryptor theirs(*context, other.createPublicKey());
        theirs.setSecretKey(other.secretKey());
        auto foreign = a.encrypt(theirs);
        decryptor->decrypt(foreign[0][0], p);
        encoder->decodePolynomial(p, coeffs);
        bool unreadable = false;
        for (size_t j = 0; j < 8; j++) unreadable = unreadable || coeffs[j] != x[j];
        foreign.switch_key(*evaluator, keygen->createKeySwitchingKeys(other.secretKey()));
        decryptor->decrypt(foreign[0][0], p);
        encoder->decodePolynomial(p, coeffs);
        good = unreadable;
        for (size_t j = 0; j < 8; j++) good = good && coeffs[j] == x[j];
        EXPECT(good, "Cipher2d switch_key with createKeySwitchingKeys: another key's ciphertexts decrypt under ours");
        LinearHelper::Cipher2d ragged = ca;
        ragged.data.pop_back();
        EXPECT(throws<std::invalid_argument>([&] { ragged.addInplace(*evaluator, cb); }), "shape mismatch: invalid_argument");
        EXPECT(throws<std::invalid_argument>([&] { LinearHelper::MatmulHelper(2, 8, 4, slotCount, 0, false).packOutputs(*evaluator, autok, ca); }), "packOutputs without packLwe: invalid_argument");
        EXPECT(throws<std::invalid_argument>([&] { helper.matmul(*evaluator, LinearHelper::Cipher2d(), a); }), "matmul with the wrong grid: invalid_argument");
    }
};

int main(int argc, char **argv) {
    const size_t N = argc > 1 ? (size_t)std::atol(argv[1]) : 4096;
    KernelProvider::initialize();
    std::srand(0);
    LinearTest test(N, {60, 60, 60}, 1ul << 41); // test/app/linear.cu:578 (there N = 16384)
    test.testPolynomialEncoding();
    test.testCipher2d();
    test.testReverseAndObjectives();
    test.testMatmulInts(4, 6, 8, false, false, "matmul 4 x 6 x 8, results as they come");
    test.testMatmulInts(4, 6, 8, true, false, "matmul 4 x 6 x 8, LWE-packed results");
    test.testMatmulInts(5, 37, 21, true, false, "matmul 5 x 37 x 21 (ragged blocks), LWE-packed results");
    test.testMatmulInts(3, 50, 101, false, false, "matmul 3 x 50 x 101, results as they come");
    test.testMatmulInts(16, 32, 600, true, false, "matmul 16 x 32 x 600, many results LWE-packed eight to one");
    test.testMatmulInts(4, 6, 8, true, true, "matmulCipher 4 x 6 x 8, relinearized and LWE-packed");
    test.testMatmulInts(8, 16, 32, false, true, "matmulCipher 8 x 16 x 32, relinearized");
    test.testConv2dInt(1, 2, 3, 9, 9, 3, 3, "conv2d 2 -> 3 channels, 9 x 9 image, 3 x 3 kernel");
    test.testConv2dInt(2, 3, 2, 12, 10, 3, 2, "conv2d batch 2, 3 -> 2 channels, 12 x 10 image, 3 x 2 kernel");
    test.testConv2dInt(1, 2, 2, 80, 70, 5, 5, "conv2d 80 x 70 image cut into overlapping blocks");
    std::printf(failures ? "%d FAILURES\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
