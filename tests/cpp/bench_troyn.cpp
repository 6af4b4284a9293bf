// bench_troyn -- the metric of BASELINE.json (ct x ct multiply + relinearize, ops/s) measured THROUGH include/troyn.hpp, the C++ surface a caller of
// the reference writes against (src/troy_cuda.cuh; Evaluator::multiply / relinearizeInplace, src/evaluator_cuda.cuh:85-115), next to the C-ABI line of
// bench.py.  Three ways of calling, each at the batch sizes given:
//   single   the reference's own calls, one ciphertext at a time:        evaluator.multiply(a, b, c); evaluator.relinearizeInplace(c, rlk);
//   loop     B ciphertexts through those calls in a loop (what a caller gets who does not batch)
//   batch    B ciphertexts through the slab-batched forms:                c = evaluator.multiplyBatch(a, b); evaluator.relinearizeInplaceBatch(c, rlk);
// Every timed configuration is verified: items 0 and B - 1 of the batched result equal, limb for limb, what the single calls give for the same
// operands, and the single result decrypts to the slot-wise product (real keys, real encryptions: KeyGenerator / Encryptor / Decryptor of troyn.hpp).
// Wall clock around `steps` repetitions, device synchronised before each reading (the reference's test/timetest.cu reads its clock WITHOUT
// synchronising; these numbers include the whole operation).
//
//   bench_troyn <workload> <steps> <batch> [<batch> ..]      workload: bfv_n32768_l14 (the headline) | bfv_n8192_l4 (BASELINE configs[1]) | bfv_n4096_l2 (CPU emulator smoke)
// One JSON object per line on stdout.
#include "troyn.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

using namespace troyn;
using std::vector;

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void sync() { check(troyhip_stream_synchronize(nullptr)); }

int main(int argc, char **argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: bench_troyn <workload> <steps> <batch> [<batch> ..]\n"); return 2; }
    const std::string workload = argv[1];
    const int steps = std::atoi(argv[2]);
    size_t n = 0;
    vector<int> bits;
    if (workload == "bfv_n32768_l14") { n = 32768; bits.push_back(60); for (int i = 0; i < 13; i++) bits.push_back(58); bits.push_back(60); }
    else if (workload == "bfv_n8192_l4") { n = 8192; bits = {40, 36, 36, 36, 40}; }
    else if (workload == "bfv_n4096_l2") { n = 4096; bits = {36, 36, 37}; }
    else { std::fprintf(stderr, "unknown workload %s\n", workload.c_str()); return 2; }
    int failures = 0;
    try {
        KernelProvider::initialize();
        EncryptionParameters parms(SchemeType::bfv);
        parms.setPolyModulusDegree(n);
        parms.setCoeffModulus(CoeffModulus::Create(n, bits));
        parms.setPlainModulus(PlainModulus::Batching(n, 20));
        SEALContext context(parms, true, SecurityLevel::none);
        KeyGenerator keygen(context);
        PublicKey pk = keygen.createPublicKey();
        RelinKeys rlk = keygen.createRelinKeys();
        Encryptor encryptor(context, pk);
        Decryptor decryptor(context, keygen.secretKey());
        BatchEncoder encoder(context);
        Evaluator evaluator(context);
        const uint64_t t = parms.plainModulus().value();

        // a handful of distinct encryptions, repeated through the batch (the time of the operation does not depend on the residues)
        const size_t distinct = 4;
        std::mt19937_64 rng(11);
        vector<vector<uint64_t>> slots_a(distinct, vector<uint64_t>(n)), slots_b(distinct, vector<uint64_t>(n));
        vector<Ciphertext> ea(distinct), eb(distinct);
        for (size_t i = 0; i < distinct; i++) {
            for (auto &x : slots_a[i]) x = rng() % t;
            for (auto &x : slots_b[i]) x = rng() % t;
            Plaintext pa, pb;
            encoder.encode(slots_a[i], pa);
            encoder.encode(slots_b[i], pb);
            encryptor.encrypt(pa, ea[i]);
            encryptor.encrypt(pb, eb[i]);
        }
        // the single calls: reference result per distinct pair + the decrypt check
        vector<Ciphertext> single(distinct);
        for (size_t i = 0; i < distinct; i++) {
            evaluator.multiply(ea[i], eb[i], single[i]);
            evaluator.relinearizeInplace(single[i], rlk);
            Plaintext pr;
            decryptor.decrypt(single[i], pr);
            vector<uint64_t> got;
            encoder.decode(pr, got);
            bool ok = true;
            for (size_t k = 0; k < n; k++) ok = ok && got[k] == (uint64_t)((unsigned __int128)slots_a[i][k] * slots_b[i][k] % t);
            if (!ok) { std::printf("{\"error\": \"single multiply + relinearize of pair %zu does not decrypt to the product\"}\n", i); failures++; }
        }

        for (int arg = 3; arg < argc; arg++) {
            const size_t B = (size_t)std::atol(argv[arg]);
            if (!B) continue;
            vector<const Ciphertext *> pa(B), pb(B);
            for (size_t i = 0; i < B; i++) { pa[i] = &ea[i % distinct]; pb[i] = &eb[i % distinct]; }
            const vector<Ciphertext> a = Ciphertext::packBatch(pa), b = Ciphertext::packBatch(pb); // resident operand slabs, as a batching caller holds them
            auto report = [&](const char *mode, double ms, bool verified) {
                std::printf("{\"workload\": \"%s\", \"api\": \"troyn.hpp\", \"mode\": \"%s\", \"batch\": %zu, \"steps\": %d, \"ms_per_step\": %.4f, \"ops_per_s\": %.1f, \"verified\": %s}\n",
                            workload.c_str(), mode, B, steps, ms, 1e3 * (double)B / ms, verified ? "true" : "false");
                std::fflush(stdout);
                if (!verified) failures++;
            };
            {   // batch
                vector<Ciphertext> c;
                for (int w = 0; w < 2; w++) { c = evaluator.multiplyBatch(a, b); evaluator.relinearizeInplaceBatch(c, rlk); }
                sync();
                const double t0 = now_ms();
                for (int s = 0; s < steps; s++) { c = evaluator.multiplyBatch(a, b); evaluator.relinearizeInplaceBatch(c, rlk); }
                sync();
                const double ms = (now_ms() - t0) / steps;
                const bool ok = c.size() == B && c[0].size() == 2 && c[0].toHost() == single[0].toHost() && c[B - 1].toHost() == single[(B - 1) % distinct].toHost();
                report(B == 1 ? "batch(1)" : "batch", ms, ok);
            }
            if (B <= 16) { // the reference's calls, one ciphertext at a time
                vector<Ciphertext> c(B);
                auto once = [&]() { for (size_t i = 0; i < B; i++) { evaluator.multiply(a[i], b[i], c[i]); evaluator.relinearizeInplace(c[i], rlk); } };
                once(); once();
                sync();
                const double t0 = now_ms();
                for (int s = 0; s < steps; s++) once();
                sync();
                const double ms = (now_ms() - t0) / steps;
                const bool ok = c[0].toHost() == single[0].toHost() && c[B - 1].toHost() == single[(B - 1) % distinct].toHost();
                report(B == 1 ? "single" : "loop", ms, ok);
            }
        }
    } catch (const std::exception &e) {
        std::printf("{\"error\": \"%s\"}\n", e.what());
        failures++;
    }
    std::printf(failures ? "FAILED %d\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
