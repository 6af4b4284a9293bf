// User code of the reference's CKKS application layer, compiled with plain g++ against include/troyn.hpp + include/troyn_app.hpp
// and linked to libtroyhip.so.  The flows are those of the reference's own test/app/linear_ckks.cu (testMatmul, testFullMatmul,
// testConv2d, testFullConv2d: same call sequence on the same helper API), here with assertions instead of a printed difference,
// plus the pieces of troyn:: those flows lean on: CKKSEncoder::encodePolynomial / decodePolynomial (src/ckks_cuda.cu:455-575,
// 983-1049), Encryptor::encryptSymmetric, the ParmsID / ContextData chain and a CKKS multiply -> relinearize -> rescale.
#include "troyn_app.hpp"
#include <chrono>
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

static double max_abs_diff(const vector<double> &a, const vector<double> &b) {
    double d = a.size() == b.size() ? 0 : 1e300;
    for (size_t i = 0; i < a.size() && i < b.size(); i++) d = std::max(d, std::fabs(a[i] - b[i]));
    return d;
}

// the fixture of test/app/linear_ckks.cu:112-160
struct LinearTest {
    size_t slotCount;
    int dataBound;
    double delta;
    std::unique_ptr<SEALContext> context;
    std::unique_ptr<KeyGenerator> keygen;
    std::unique_ptr<CKKSEncoder> encoder;
    std::unique_ptr<Encryptor> encryptor;
    std::unique_ptr<Decryptor> decryptor;
    std::unique_ptr<Evaluator> evaluator;
    PublicKey pk;
    RelinKeys rlk;
    vector<ParmsID> parmIDs;

    LinearTest(size_t polyModulusDegree, vector<int> qs, int dataBound, double delta) : slotCount(polyModulusDegree / 2), dataBound(dataBound), delta(delta) {
        EncryptionParameters parms(SchemeType::ckks);
        parms.setPolyModulusDegree(polyModulusDegree);
        parms.setCoeffModulus(CoeffModulus::Create(polyModulusDegree, qs));
        context.reset(new SEALContext(parms, true, SecurityLevel::none));
        keygen.reset(new KeyGenerator(*context));
        keygen->createPublicKey(pk);
        keygen->createRelinKeys(rlk);
        encoder.reset(new CKKSEncoder(*context));
        encryptor.reset(new Encryptor(*context, pk));
        encryptor->setSecretKey(keygen->secretKey());
        decryptor.reset(new Decryptor(*context, keygen->secretKey()));
        evaluator.reset(new Evaluator(*context));
        std::shared_ptr<const SEALContext::ContextDataCuda> cd = context->firstContextData();
        while (cd) {
            parmIDs.push_back(cd->parmsID());
            cd = cd->nextContextData();
        }
    }

    vector<double> randomRealVector(size_t count, bool signed_values = false) {
        vector<double> v(count);
        for (auto &x : v) x = (double)(std::rand() % dataBound) * (signed_values && std::rand() % 2 ? -1.0 : 1.0);
        return v;
    }
    vector<double> decrypt(const Ciphertext &c) {
        Plaintext p;
        decryptor->decrypt(c, p);
        vector<double> r;
        encoder->decodePolynomial(p, r);
        return r;
    }

    // ---- troyn:: pieces under the helpers
    void testContextChain() {
        EXPECT(parmIDs.size() == context->firstLimbs() - context->lastLimbs() + 1, "firstContextData -> nextContextData walks every data level");
        EXPECT(parmIDs.front() == context->firstParmsID() && parmIDs.back() == context->lastParmsID(), "chain ends at first / last parms_id");
        EXPECT(context->keyContextData()->parmsID() == context->keyParmsID() && context->keyParmsID() != context->firstParmsID(), "key level has its own parms_id");
        EXPECT(context->keyContextData()->nextContextData()->parmsID() == context->firstParmsID(), "key level -> first level");
        EXPECT(context->lastContextData()->chainIndex() == 0 && context->firstContextData()->chainIndex() == parmIDs.size() - 1, "chainIndex counts down to 0");
        EXPECT(context->firstContextData()->parms().coeffModulus().size() == context->firstLimbs(), "ContextData::parms() holds the level's primes");
        ParmsID bogus = context->firstParmsID();
        bogus[0] ^= 1;
        EXPECT(!context->getContextData(bogus) && !context->getContextData(parmsIDZero), "unknown parms_id -> nullptr");
    }

    void testEncoder() {
        const size_t n = slotCount * 2;
        vector<double> v = randomRealVector(n, true);
        v[1] = -0.49 / delta; // rounds to zero
        v[2] = 2.5 / delta;   // half away from zero (C round()): 3
        v[3] = -2.5 / delta;  // -3
        for (const ParmsID &id : parmIDs) {
            Plaintext p;
            encoder->encodePolynomial(v, id, delta, p);
            EXPECT(p.isNttForm() && p.parmsID() == id && p.scale() == delta && p.coeffCount() == (size_t)id.limbs * n, "encodePolynomial: NTT form at the level asked for");
            vector<double> back;
            encoder->decodePolynomial(p, back);
            bool exact = back.size() == n;
            for (size_t i = 4; i < n && exact; i++) exact = back[i] == v[i];
            EXPECT(exact && back[1] == 0 && back[2] == 3 / delta && back[3] == -3 / delta, "decodePolynomial(encodePolynomial(v)) == round(v * scale) / scale, exactly");
        }
        Plaintext p;
        encoder->encodePolynomial(vector<double>{1, 2, 3}, delta, p);
        EXPECT(p.parmsID() == context->firstParmsID(), "encodePolynomial without parms_id -> first level");
        vector<double> back;
        encoder->decodePolynomial(p, back);
        EXPECT(back.size() == n && back[0] == 1 && back[2] == 3 && back[3] == 0 && back[n - 1] == 0, "short input is zero-padded");
        // magnitudes beyond 64 and beyond 128 bits (the reference's second and third decomposition paths)
        if (context->firstContextData()->totalCoeffModulusBitCount() > 140) {
            const double big = std::ldexp(1.0, 70) + std::ldexp(1.0, 30), huge = -(std::ldexp(1.0, 130) + std::ldexp(1.0, 90));
            encoder->encodePolynomial(vector<double>{big, huge, -big}, 1.0, p);
            encoder->decodePolynomial(p, back);
            EXPECT(back[0] == big && back[1] == huge && back[2] == -big, "coefficients of 71 and 131 bits survive the round trip");
        }
        EXPECT(throws<std::invalid_argument>([&] { Plaintext q; encoder->encodePolynomial(vector<double>(n + 1, 1.0), delta, q); }), "more than N values -> invalid_argument");
        EXPECT(throws<std::invalid_argument>([&] { Plaintext q; encoder->encodePolynomial(vector<double>{1.0}, parmIDs.back(), std::ldexp(1.0, 400), q); }),
               "encoded values are too large -> invalid_argument");
        EXPECT(throws<std::invalid_argument>([&] { Plaintext q; encoder->encodePolynomial(vector<double>{1.0}, parmsIDZero, delta, q); }), "unknown parms_id -> invalid_argument");
        EXPECT(throws<std::invalid_argument>([&] { Plaintext q(vector<uint64_t>(n, 0)); vector<double> o; encoder->decodePolynomial(q, o); }), "decodePolynomial of a coefficient-form plaintext -> invalid_argument");
        EXPECT(throws<std::invalid_argument>([&] { Plaintext q = p; q.scale() = -1; vector<double> o; encoder->decodePolynomial(q, o); }), "scale out of bounds -> invalid_argument");
    }

    // polynomial product through multiply -> relinearize -> rescale, public-key and secret-key encryption
    void testMultiplyRescale() {
        if (parmIDs.size() < 2) return;
        const size_t n = slotCount * 2;
        vector<double> a(n, 0.0), b(n, 0.0), want(n, 0.0);
        for (size_t i = 0; i < 24; i++) { a[std::rand() % n] = std::rand() % dataBound; b[std::rand() % n] = std::rand() % dataBound; }
        for (size_t i = 0; i < n; i++)
            for (size_t j = 0; j < n && a[i] != 0; j++) {
                if (b[j] == 0) continue;
                if (i + j < n) want[i + j] += a[i] * b[j]; else want[i + j - n] -= a[i] * b[j];
            }
        Plaintext pa, pb;
        encoder->encodePolynomial(a, delta, pa);
        encoder->encodePolynomial(b, delta, pb);
        Ciphertext ca = encryptor->encrypt(pa), cb = encryptor->encryptSymmetric(pb);
        EXPECT(ca.isNttForm() && ca.scale() == delta && ca.parmsID() == context->firstParmsID(), "CKKS encrypt: NTT form, scale and level of the plaintext");
        EXPECT(max_abs_diff(decrypt(ca), a) < 1e-2 && max_abs_diff(decrypt(cb), b) < 1e-2, "encrypt / encryptSymmetric -> decrypt -> decodePolynomial");
        evaluator->multiplyInplace(ca, cb);
        evaluator->relinearizeInplace(ca, rlk);
        EXPECT(ca.size() == 2 && ca.scale() == delta * delta, "multiply + relinearize: size 2, scale squared");
        const double before = max_abs_diff(decrypt(ca), want);
        evaluator->rescaleToNextInplace(ca);
        EXPECT(ca.parmsID() == parmIDs[1] && ca.scale() < delta * delta, "rescaleToNext: next level, scale divided by the dropped prime");
        const double after = max_abs_diff(decrypt(ca), want);
        std::printf("     product error %.3g before, %.3g after rescale\n", before, after);
        EXPECT(before < 0.05 && after < 0.05, "CKKS encrypt -> multiply -> relinearize -> rescale -> decrypt == polynomial product");
        EXPECT(throws<std::logic_error>([&] { Encryptor e(*context, pk); Plaintext q = pa; e.encryptSymmetric(q); }), "encryptSymmetric without a secret key -> logic_error");
        EXPECT(throws<std::invalid_argument>([&] { Plaintext q(vector<uint64_t>(n, 1)); encryptor->encrypt(q); }), "CKKS encrypt of a coefficient-form plaintext -> invalid_argument");
    }

    // ---- test/app/linear_ckks.cu:162-196
    double testMatmul(size_t batchSize, size_t inputDims, size_t outputDims, bool compare_paths = false) {
        auto weights = randomRealVector(inputDims * outputDims, true);
        auto x = randomRealVector(batchSize * inputDims, true);
        auto lastParmsID = context->lastParmsID();

        LinearHelperCKKS::MatmulHelper helper(batchSize, inputDims, outputDims, slotCount);
        helper.encodeWeights(*encoder, lastParmsID, weights, delta);
        auto xEnc = helper.encryptInputs(*encryptor, *encoder, lastParmsID, x, delta);
        auto yEnc = helper.matmul(*evaluator, xEnc);
        auto yDec = helper.decryptOutputs(*encoder, *decryptor, yEnc);

        auto timed = [&](const LinearHelperCKKS::Cipher2d &in) {
            check(troyhip_stream_synchronize(nullptr));
            const auto t0 = std::chrono::steady_clock::now();
            auto out = helper.matmul(*evaluator, in);
            check(troyhip_stream_synchronize(nullptr));
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        };
        if (compare_paths) { // the same inputs as individually allocated ciphertexts take the per-ciphertext loop: same bits
            LinearHelperCKKS::Cipher2d loose;
            for (auto &row : xEnc.data) {
                loose.data.emplace_back();
                for (auto &ct : row) loose.data.back().push_back(ct); // a copy owns its storage
            }
            auto yLoose = helper.matmul(*evaluator, loose);
            bool same = yLoose.data.size() == yEnc.data.size();
            for (size_t i = 0; i < yEnc.data.size() && same; i++)
                for (size_t j = 0; j < yEnc[i].size() && same; j++)
                    same = yLoose[i][j].toHost() == yEnc[i][j].toHost() && yLoose[i][j].scale() == yEnc[i][j].scale() && yLoose[i][j].parmsID() == yEnc[i][j].parmsID();
            EXPECT(same, "matmul: one batched launch per weight block == the per-ciphertext loop, bit for bit");
            std::printf("     matmul %zux%zu batch %zu: %.3f ms batched slab, %.3f ms per-ciphertext loop\n", inputDims, outputDims, batchSize, timed(xEnc), timed(loose));
        }

        vector<double> y(batchSize * outputDims, 0);
        for (size_t i = 0; i < batchSize; i++)
            for (size_t j = 0; j < inputDims; j++)
                for (size_t k = 0; k < outputDims; k++) y[i * outputDims + k] += x[i * inputDims + j] * weights[j * outputDims + k];
        return max_abs_diff(y, yDec);
    }

    // ---- test/app/linear_ckks.cu:198-276: two-party flow with input shares, an output mask and both serialisations
    double testFullMatmul(size_t batchSize, size_t inputDims, size_t outputDims) {
        auto weights = randomRealVector(inputDims * outputDims);
        auto xClient = randomRealVector(batchSize * inputDims);
        auto xServer = randomRealVector(batchSize * inputDims);
        auto lastParmsID = context->lastParmsID();

        LinearHelperCKKS::MatmulHelper helper(batchSize, inputDims, outputDims, slotCount);
        helper.encodeWeights(*encoder, lastParmsID, weights, delta);
        auto r = randomRealVector(batchSize * outputDims);
        auto xServerEncoded = helper.encodeInputs(*encoder, lastParmsID, xServer, delta);
        auto rEncoded = helper.encodeOutputs(*encoder, lastParmsID, r, delta * delta);

        auto xEnc = helper.encryptInputs(*encryptor, *encoder, lastParmsID, xClient, delta);
        size_t full_bytes = 0, term_bytes = 0;
        { // serialize
            std::ostringstream sout;
            xEnc.save(sout);
            auto p = sout.str();
            full_bytes = p.size();
            std::istringstream sin(p);
            xEnc = LinearHelperCKKS::Cipher2d();
            xEnc.load(sin, *context);
        }
        helper.addPlainInplace(*evaluator, xEnc, xServerEncoded);
        auto yEnc = helper.matmul(*evaluator, xEnc);
        helper.addPlainInplace(*evaluator, yEnc, rEncoded);
        { // serialize
            std::ostringstream sout;
            helper.serializeOutputs(*evaluator, yEnc, sout);
            auto p = sout.str();
            term_bytes = p.size();
            std::istringstream sin(p);
            yEnc = helper.deserializeOutputs(*evaluator, sin);
        }
        auto yDec = helper.decryptOutputs(*encoder, *decryptor, yEnc);
        for (size_t i = 0; i < batchSize * outputDims; i++) yDec[i] -= r[i];
        std::printf("     xEnc %zu bytes, yEnc (terms only) %zu bytes\n", full_bytes, term_bytes);

        vector<double> y(batchSize * outputDims, 0);
        for (size_t i = 0; i < batchSize; i++)
            for (size_t j = 0; j < inputDims; j++)
                for (size_t k = 0; k < outputDims; k++) y[i * outputDims + k] += (xClient[i * inputDims + j] + xServer[i * inputDims + j]) * weights[j * outputDims + k];
        return max_abs_diff(y, yDec);
    }

    static vector<double> conv_plain(const vector<double> &x, const vector<double> &weights, size_t batchSize, size_t inputChannels, size_t outputChannels,
                                     size_t imageHeight, size_t imageWidth, size_t kernelHeight, size_t kernelWidth) {
        const size_t yh = imageHeight - kernelHeight + 1, yw = imageWidth - kernelWidth + 1;
        vector<double> y(batchSize * outputChannels * yh * yw, 0);
        for (size_t b = 0; b < batchSize; b++)
            for (size_t oc = 0; oc < outputChannels; oc++)
                for (size_t yi = 0; yi < yh; yi++)
                    for (size_t yj = 0; yj < yw; yj++) {
                        double element = 0;
                        for (size_t ic = 0; ic < inputChannels; ic++)
                            for (size_t u = 0; u < kernelHeight; u++)
                                for (size_t v = 0; v < kernelWidth; v++)
                                    element += x[((b * inputChannels + ic) * imageHeight + yi + u) * imageWidth + yj + v] * weights[((oc * inputChannels + ic) * kernelHeight + u) * kernelWidth + v];
                        y[((b * outputChannels + oc) * yh + yi) * yw + yj] = element;
                    }
        return y;
    }

    // ---- test/app/linear_ckks.cu:278-324
    double testConv2d(size_t batchSize, size_t inputChannels, size_t outputChannels, size_t imageHeight, size_t imageWidth, size_t kernelHeight, size_t kernelWidth) {
        auto weights = randomRealVector(inputChannels * outputChannels * kernelHeight * kernelWidth, true);
        auto x = randomRealVector(batchSize * inputChannels * imageHeight * imageWidth, true);
        auto lastParmsID = context->lastParmsID();
        LinearHelperCKKS::Conv2dHelper helper(batchSize, imageHeight, imageWidth, kernelHeight, kernelWidth, inputChannels, outputChannels, slotCount);
        helper.encodeWeights(*encoder, lastParmsID, weights, delta);
        auto xEnc = helper.encryptInputs(*encryptor, *encoder, lastParmsID, x, delta);
        auto yEnc = helper.conv2d(*evaluator, xEnc);
        auto yDec = helper.decryptOutputs(*encoder, *decryptor, yEnc);
        return max_abs_diff(conv_plain(x, weights, batchSize, inputChannels, outputChannels, imageHeight, imageWidth, kernelHeight, kernelWidth), yDec);
    }

    // ---- test/app/linear_ckks.cu:327-425
    double testFullConv2d(size_t batchSize, size_t inputChannels, size_t outputChannels, size_t imageHeight, size_t imageWidth, size_t kernelHeight, size_t kernelWidth) {
        auto weights = randomRealVector(inputChannels * outputChannels * kernelHeight * kernelWidth);
        auto xClient = randomRealVector(batchSize * inputChannels * imageHeight * imageWidth);
        auto xServer = randomRealVector(batchSize * inputChannels * imageHeight * imageWidth);
        auto lastParmsID = context->lastParmsID();
        LinearHelperCKKS::Conv2dHelper helper(batchSize, imageHeight, imageWidth, kernelHeight, kernelWidth, inputChannels, outputChannels, slotCount);
        helper.encodeWeights(*encoder, lastParmsID, weights, delta);
        const size_t yh = imageHeight - kernelHeight + 1, yw = imageWidth - kernelWidth + 1;
        auto r = randomRealVector(batchSize * outputChannels * yh * yw);
        auto xServerEncoded = helper.encodeInputs(*encoder, lastParmsID, xServer, delta);
        auto rEncoded = helper.encodeOutputs(*encoder, lastParmsID, r, delta * delta);

        auto xEnc = helper.encryptInputs(*encryptor, *encoder, lastParmsID, xClient, delta);
        {
            std::ostringstream sout;
            xEnc.save(sout);
            std::istringstream sin(sout.str());
            xEnc = LinearHelperCKKS::Cipher2d();
            xEnc.load(sin, *context);
        }
        helper.addPlainInplace(*evaluator, xEnc, xServerEncoded);
        auto yEnc = helper.conv2d(*evaluator, xEnc);
        helper.addPlainInplace(*evaluator, yEnc, rEncoded);
        {
            std::ostringstream sout;
            helper.serializeOutputs(*evaluator, yEnc, sout);
            std::istringstream sin(sout.str());
            yEnc = helper.deserializeOutputs(*evaluator, sin);
        }
        auto yDec = helper.decryptOutputs(*encoder, *decryptor, yEnc);
        for (size_t i = 0; i < yDec.size(); i++) yDec[i] -= r[i];
        vector<double> xs(xClient.size());
        for (size_t i = 0; i < xs.size(); i++) xs[i] = xClient[i] + xServer[i];
        return max_abs_diff(conv_plain(xs, weights, batchSize, inputChannels, outputChannels, imageHeight, imageWidth, kernelHeight, kernelWidth), yDec);
    }
};

int main() {
    std::srand(0);
    KernelProvider::initialize();
    {
        // the parameters of the reference's own main (test/app/linear_ckks.cu:429-434): N = 4096, {50, 50}, values < 10, scale 2^15
        LinearTest test(4096, {50, 50}, 10, (double)(1 << 15));
        test.testContextChain();
        test.testEncoder();
        double d = test.testMatmul(4, 128, 128, true);
        std::printf("     difference %.3g\n", d);
        // noise: ~N weight coefficients of size ~bound * scale meet a fresh error (sigma 3.2) each: sqrt(4096) * 3.2 * 5.5 / 2^15 = 0.03
        // standard deviation per output, results of size up to 128 * 81
        EXPECT(d < 0.5, "MatmulHelper 128x128, batch 4 (N=4096)");
        d = test.testMatmul(2, 300, 17);
        EXPECT(d < 0.5, "MatmulHelper 300x17 (ragged blocks)");
        d = test.testFullMatmul(3, 200, 50);
        std::printf("     difference %.3g\n", d);
        EXPECT(d < 0.5, "testFullMatmul: shares + mask + save/load + serializeOutputs/deserializeOutputs");
        d = test.testConv2d(2, 3, 2, 10, 12, 3, 3);
        std::printf("     difference %.3g\n", d);
        EXPECT(d < 0.1, "Conv2dHelper 3->2 channels, 10x12 image, 3x3 kernel");
        d = test.testConv2d(1, 2, 1, 70, 66, 3, 5);
        EXPECT(d < 0.1, "Conv2dHelper blocked (image larger than sqrt(N) per side)");
        d = test.testFullConv2d(1, 8, 3, 16, 16, 1, 1);
        std::printf("     difference %.3g\n", d);
        EXPECT(d < 0.5, "testFullConv2d: 1x1 kernel, channel groups, serialisation");
        EXPECT(throws<std::invalid_argument>([&] { LinearHelperCKKS::MatmulHelper h(1, 4, 4, 2048); h.encodeWeights(*test.encoder, test.context->lastParmsID(), vector<double>(3), 1.0); }),
               "Weight size incorrect. -> invalid_argument");
    }
    {
        // BASELINE config E shape: N = 8192, 128x128 layer; a chain with a rescale level for the multiply test
        LinearTest test(8192, {60, 40, 40, 60}, 16, (double)(1ull << 40));
        test.testContextChain();
        test.testEncoder();
        test.testMultiplyRescale();
        LinearTest app(8192, {60, 60}, 64, (double)(1 << 20));
        double d = app.testMatmul(32, 128, 128, true);
        std::printf("     difference %.3g\n", d);
        EXPECT(d < 0.1, "MatmulHelper 128x128, batch 32 (N=8192)");
    }
    std::printf(failures ? "FAILED %d\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
