// include/troyn_devices.hpp + the multi-device half of include/troyhip.h (round 6): a batch sharded over several contexts, each on "its" device, a host
// thread per member -- against the same batch on ONE context, limb for limb.  usage: test_troyn_devices <N> <members> [distinct]
//   distinct = 1: member i takes device i (the emulator's virtual devices, HIP_EMUL_DEVICES >= members; a real multi-GPU node)
//   distinct = 0: every member shares device 0 (what a one-GPU box can run: per-context tables and scratch, threads, one device queue)
#include "troyn_devices.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>

using namespace troyn;

static int failures = 0;
#define EXPECT(cond, what)                                              \
    do {                                                                \
        if (!(cond)) { std::printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__); failures++; } \
        else std::printf("ok   %s\n", what);                            \
    } while (0)

static std::vector<uint64_t> poly(size_t n, uint64_t t, unsigned seed) {
    std::mt19937_64 g(seed);
    std::vector<uint64_t> p(n);
    for (auto &x : p) x = g() % t;
    return p;
}

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)std::atoll(argv[1]) : 4096;
    const size_t members = argc > 2 ? (size_t)std::atoll(argv[2]) : 3;
    const bool distinct = argc > 3 && std::atoi(argv[3]) != 0;
    KernelProvider::initialize(0);

    // ---- KernelProvider's public statics (src/kernelprovider.cuh:35-85)
    {
        std::vector<uint64_t> h(1000), back(1000, 7);
        for (size_t i = 0; i < h.size(); i++) h[i] = i * i + 3;
        uint64_t *a = KernelProvider::malloc<uint64_t>(h.size()), *b = KernelProvider::malloc<uint64_t>(h.size());
        EXPECT(a && b && KernelProvider::malloc<uint64_t>(0) == nullptr, "KernelProvider::malloc (zero length -> nullptr)");
        KernelProvider::copy(a, h.data(), h.size());
        KernelProvider::copyOnDevice(b, a, h.size());
        KernelProvider::retrieve(back.data(), b, back.size());
        EXPECT(back == h, "copy -> copyOnDevice -> retrieve");
        KernelProvider::memsetZero(b + 10, 20);
        KernelProvider::retrieve(back.data(), b, back.size());
        bool ok = true;
        for (size_t i = 0; i < h.size(); i++) ok = ok && back[i] == ((i >= 10 && i < 30) ? 0 : h[i]);
        EXPECT(ok, "memsetZero");
        KernelProvider::free(a);
        KernelProvider::free(b);
        KernelProvider::copy<uint64_t>(nullptr, nullptr, 0); // zero lengths are no-ops, as in the reference
    }

    EncryptionParameters parms(SchemeType::bfv);
    parms.setPolyModulusDegree(n);
    parms.setCoeffModulus(CoeffModulus::Create(n, {40, 40, 40, 40}));
    parms.setPlainModulus(PlainModulus::Batching(n, 20));
    const uint64_t t = parms.plainModulus().value();

    std::vector<int> devices(members, 0);
    if (distinct) {
        if ((size_t)KernelProvider::deviceCount() < members) { std::printf("needs %zu devices, %d visible\n", members, KernelProvider::deviceCount()); return 2; }
        for (size_t i = 0; i < members; i++) devices[i] = (int)i;
    }
    DeviceGroup group(parms, devices, true, SecurityLevel::none);
    EXPECT(group.size() == members && group.home() == devices[0] && KernelProvider::currentDevice() == devices[0], "DeviceGroup: one context per member, caller left on the home device");
    bool placed = true;
    for (size_t i = 0; i < members; i++) placed = placed && group.context(i).device() == devices[i];
    EXPECT(placed, "every context reports the device it was created on");

    // keys and inputs on the home device, through the home context
    const SEALContext &home = group.context(0);
    KeyGenerator keygen(home, 1, 2);
    Encryptor enc(home, keygen.createPublicKey(), 3, 4);
    Decryptor dec(home, keygen.secretKey());
    RelinKeys rlk = keygen.createRelinKeys();
    GaloisKeys gk;
    keygen.createGaloisKeys(std::vector<int>{1}, gk);
    const size_t batch = 3 * members + 2; // uneven shards
    std::vector<Ciphertext> a, b;
    for (size_t i = 0; i < batch; i++) {
        a.push_back(enc.encrypt(Plaintext(poly(n, t, 100 + (unsigned)i))));
        b.push_back(enc.encrypt(Plaintext(poly(n, t, 200 + (unsigned)i))));
    }
    // the reference result: the whole batch on the home context
    const Evaluator &ev = group.evaluator(0);
    std::vector<Ciphertext> want = ev.multiplyBatch(a, b);
    ev.relinearizeInplaceBatch(want, rlk);
    ev.rotateRowsInplaceBatch(want, 1, gk);

    const auto ranges = group.shards(batch);
    size_t covered = 0;
    bool contiguous = true;
    for (const auto &r : ranges) { contiguous = contiguous && r.first == covered; covered += r.second; }
    EXPECT(contiguous && covered == batch && ranges[0].second == batch / members + (batch % members ? 1 : 0) && ranges.back().second == batch / members,
           "shardBatch: contiguous ranges, the remainder in the first shards");

    std::vector<RelinKeys> rlks = group.replicate(rlk);
    std::vector<GaloisKeys> gks = group.replicate(gk);
    std::vector<std::vector<Ciphertext>> sa = group.scatter(a), sb = group.scatter(b);
    bool on_their_devices = true;
    for (size_t i = 0; i < members; i++) on_their_devices = on_their_devices && sa[i].size() == ranges[i].second && sa[i][0].parmsID() == group.context(i).firstParmsID();
    EXPECT(on_their_devices, "scatter: one slab per member, bound to the member's context");
    group.parallel([&](size_t i) {
        std::vector<Ciphertext> prod = group.evaluator(i).multiplyBatch(sa[i], sb[i]);
        group.evaluator(i).relinearizeInplaceBatch(prod, rlks[i]);
        group.evaluator(i).rotateRowsInplaceBatch(prod, 1, gks[i]);
        sa[i] = std::move(prod);
    });
    EXPECT(KernelProvider::currentDevice() == devices[0], "parallel leaves the caller on the home device");
    std::vector<Ciphertext> got = group.gather(sa);
    bool same = got.size() == batch;
    for (size_t i = 0; same && i < batch; i++) same = got[i].size() == 2 && got[i].toHost() == want[i].toHost();
    EXPECT(same, "sharded multiply + relinearize + rotate equals the single-context batch, limb for limb");
    Plaintext out, ref;
    dec.decrypt(got[batch - 1], out);
    dec.decrypt(want[batch - 1], ref);
    EXPECT(out == ref, "... and decrypts under the home context");

    // an exception in one member surfaces in the caller, the others finish
    bool threw = false;
    try { group.parallel([&](size_t i) { if (i == members - 1) throw std::invalid_argument("member failed"); }); } catch (const std::invalid_argument &) { threw = true; }
    EXPECT(threw, "parallel rethrows a member's exception");
    // a key of another device's member is still a key of the same PARAMETERS: what is refused is a context mismatch, not a device id
    std::printf(failures ? "FAILED %d\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
