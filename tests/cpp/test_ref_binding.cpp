// The reference-side binding (tests/cpp/ref_binding.hpp) at work: keys, encoders, encryption and decryption are the REFERENCE's own CPU
// classes (namespace troy, src/troy_cpu.h); only the evaluator runs on libtroyhip -- reference KeyGenerator / Encryptor -> upload ->
// troyn::Evaluator (this repo's kernels) -> download -> reference Decryptor / encoder, compared with the plaintext computation AND, limb
// for limb, with the reference's own troy::Evaluator on the same ciphertexts (the evaluator is deterministic: same inputs, same keys,
// same stored residues).  Built by tests/test_ref_binding.py in the CPU suite against oracle/_ref/libtroyref.so + the emulator build.
#include "ref_binding.hpp"
#include <complex>
#include <cstdio>
#include <cstring>
#include <random>

static int failures = 0;
#define EXPECT(cond, what)                                                                        \
    do {                                                                                          \
        if (!(cond)) { std::printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__); failures++; } \
        else std::printf("ok   %s\n", what);                                                      \
    } while (0)

static bool same_limbs(const troy::Ciphertext &a, const troy::Ciphertext &b) {
    if (a.size() != b.size() || a.coeffModulusSize() != b.coeffModulusSize() || a.isNttForm() != b.isNttForm() || a.parmsID() != b.parmsID()) return false;
    return std::memcmp(a.data(), b.data(), a.size() * a.coeffModulusSize() * a.polyModulusDegree() * 8) == 0;
}

static void integer_scheme(troy::SchemeType scheme, size_t n) {
    const char *nm = scheme == troy::SchemeType::bfv ? "bfv" : "bgv";
    troy::EncryptionParameters parms(scheme);
    parms.setPolyModulusDegree(n);
    parms.setCoeffModulus(troy::CoeffModulus::Create(n, {50, 40, 40, 50}));
    parms.setPlainModulus(troy::PlainModulus::Batching(n, 20));
    troy::SEALContext host(parms, true, troy::SecurityLevel::none);
    const uint64_t t = parms.plainModulus().value();
    troy::KeyGenerator keygen(host);
    troy::PublicKey pk; keygen.createPublicKey(pk);
    troy::RelinKeys rlk; keygen.createRelinKeys(rlk);
    troy::GaloisKeys gk; keygen.createGaloisKeys(std::vector<int>{1, 2}, gk);
    troy::BatchEncoder encoder(host);
    troy::Encryptor encryptor(host, pk);
    troy::Decryptor decryptor(host, keygen.secretKey());
    troy::Evaluator cpu_eval(host);

    troyn::SEALContext context(host);
    troyn::Evaluator evaluator(context);
    troyn::RelinKeys d_rlk(rlk);
    troyn::GaloisKeys d_gk(gk);

    std::mt19937_64 rng(7);
    std::vector<uint64_t> v1(n), v2(n);
    for (size_t i = 0; i < n; i++) { v1[i] = rng() % t; v2[i] = rng() % t; }
    troy::Plaintext p1, p2;
    encoder.encode(v1, p1);
    encoder.encode(v2, p2);
    troy::Ciphertext c1, c2;
    encryptor.encrypt(p1, c1);
    encryptor.encrypt(p2, c2);

    troyn::Ciphertext d1(c1), d2(c2), d3;
    evaluator.multiply(d1, d2, d3);
    evaluator.relinearizeInplace(d3, d_rlk);
    evaluator.rotateRowsInplace(d3, 1, d_gk);
    evaluator.modSwitchToNextInplace(d3);
    troy::Ciphertext got = d3.cpu(context);

    troy::Ciphertext want;
    cpu_eval.multiply(c1, c2, want);
    cpu_eval.relinearizeInplace(want, rlk);
    cpu_eval.rotateRowsInplace(want, 1, gk);
    cpu_eval.modSwitchToNextInplace(want);
    EXPECT(same_limbs(got, want), (std::string(nm) + " multiply -> relinearize -> rotateRows -> modSwitchToNext: limbs equal troy::Evaluator's").c_str());

    troy::Plaintext out;
    decryptor.decrypt(got, out);
    std::vector<uint64_t> dec;
    encoder.decode(out, dec);
    const size_t row = n / 2;
    bool ok = dec.size() == n;
    for (size_t i = 0; ok && i < n; i++) {
        const size_t src = i < row ? (i + 1) % row : row + (i - row + 1) % row;
        ok = dec[i] == (uint64_t)((unsigned __int128)v1[src] * v2[src] % t);
    }
    EXPECT(ok, (std::string(nm) + " decrypts (reference Decryptor + BatchEncoder) to the rotated slot-wise product").c_str());
}

static void ckks(size_t n) {
    troy::EncryptionParameters parms(troy::SchemeType::ckks);
    parms.setPolyModulusDegree(n);
    parms.setCoeffModulus(troy::CoeffModulus::Create(n, {60, 40, 40, 60}));
    troy::SEALContext host(parms, true, troy::SecurityLevel::none);
    troy::KeyGenerator keygen(host);
    troy::PublicKey pk; keygen.createPublicKey(pk);
    troy::RelinKeys rlk; keygen.createRelinKeys(rlk);
    troy::GaloisKeys gk; keygen.createGaloisKeys(std::vector<int>{1}, gk);
    troy::CKKSEncoder encoder(host);
    troy::Encryptor encryptor(host, pk);
    troy::Decryptor decryptor(host, keygen.secretKey());
    troy::Evaluator cpu_eval(host);

    troyn::SEALContext context(host);
    troyn::Evaluator evaluator(context);
    troyn::RelinKeys d_rlk(rlk);
    troyn::GaloisKeys d_gk(gk);

    const size_t slots = n / 2;
    const double scale = (double)(1ull << 40);
    std::mt19937_64 rng(11);
    std::vector<std::complex<double>> z1(slots), z2(slots);
    for (size_t i = 0; i < slots; i++) { z1[i] = {(double)(rng() % 64) / 8.0, (double)(rng() % 64) / 8.0}; z2[i] = {(double)(rng() % 64) / 8.0, 0.0}; }
    troy::Plaintext p1, p2;
    encoder.encode(z1, scale, p1);
    encoder.encode(z2, scale, p2);
    troy::Ciphertext c1, c2;
    encryptor.encrypt(p1, c1);
    encryptor.encrypt(p2, c2);

    troyn::Ciphertext d1(c1), d2(c2), d3, d4;
    evaluator.multiply(d1, d2, d3);
    evaluator.relinearizeInplace(d3, d_rlk);
    evaluator.rescaleToNextInplace(d3);
    evaluator.rotateVector(d3, 1, d_gk, d4);
    troy::Ciphertext got = d4.cpu(context);

    troy::Ciphertext want;
    cpu_eval.multiply(c1, c2, want);
    cpu_eval.relinearizeInplace(want, rlk);
    cpu_eval.rescaleToNextInplace(want);
    cpu_eval.rotateVectorInplace(want, 1, gk);
    EXPECT(same_limbs(got, want), "ckks multiply -> relinearize -> rescaleToNext -> rotateVector: limbs equal troy::Evaluator's");
    EXPECT(got.scale() == want.scale(), "ckks scale bookkeeping equals troy::Evaluator's");

    troy::Plaintext out;
    decryptor.decrypt(got, out);
    std::vector<std::complex<double>> dec;
    encoder.decode(out, dec);
    double worst = 0;
    for (size_t i = 0; i < slots; i++) worst = std::max(worst, std::abs(dec[i] - z1[(i + 1) % slots] * z2[(i + 1) % slots]));
    EXPECT(worst < 1e-3, "ckks decrypts (reference Decryptor + CKKSEncoder) to the rotated slot-wise product");
}

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)std::atol(argv[1]) : 4096;
    try {
        troyn::KernelProvider::initialize();
        integer_scheme(troy::SchemeType::bfv, n);
        integer_scheme(troy::SchemeType::bgv, n);
        ckks(n);
    } catch (const std::exception &e) {
        std::printf("FAIL exception: %s\n", e.what());
        failures++;
    }
    std::printf(failures ? "FAILED %d\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
