// troyn.hpp -- header-only C++ mirror of the reference's troyn:: interface (src/troy_cuda.cuh:20-43) over the C ABI
// of libtroyhip.so (include/troyhip.h).  Same class and method names, argument meaning and exception classes as the
// reference for the in-scope surface: user code written as
//
//     #include "troy_cuda.cuh"          ->   #include "troyn.hpp"
//     using namespace troyn;
//     KernelProvider::initialize();
//     EncryptionParameters parms(SchemeType::bfv); ... SEALContext context(parms, true, SecurityLevel::none);
//     KeyGenerator keygen(context); Encryptor encryptor(context, pk); Evaluator evaluator(context); ...
//
// compiles against it.  Key generation, encryption and decryption run on the CPU (as KeyGeneratorCuda does in the
// reference, src/keygenerator_cuda.cuh); everything in EvaluatorCuda's hot path runs on the GPU.  Out of scope here
// exactly as in SURVEY.md section 2: encoders (BatchEncoder/CKKSEncoder), key serialization, symmetric encryption.
// No HIP headers are needed: the ABI is plain pointers, sizes and status codes.
#pragma once
#include "troyhip.h"
#include <algorithm>
#include <cstdint>
#include <istream>
#include <ostream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace troyn {

inline void check(int rc) { // status -> the reference's exception classes
    if (rc == TROYHIP_OK) return;
    const std::string m = troyhip_last_error();
    switch (rc) {
    case TROYHIP_INVALID_ARGUMENT:
    case TROYHIP_NOT_INITIALIZED: throw std::invalid_argument(m);
    case TROYHIP_LOGIC_ERROR: throw std::logic_error(m);
    case TROYHIP_OUT_OF_RANGE: throw std::out_of_range(m);
    default: throw std::runtime_error(m); // "CUDA error." in the reference (src/kernelprovider.cuh:6-14)
    }
}

enum class SchemeType : uint8_t { none = 0, bfv = 1, ckks = 2, bgv = 3 }; // src/encryptionparams.h
enum class SecurityLevel : int { none = 0, tc128 = 128, tc192 = 192, tc256 = 256 };

class KernelProvider { // src/kernelprovider.cuh:24-33
public:
    static void initialize(int device = 0) { check(troyhip_initialize(device)); }
};

class Modulus { // src/modulus.h:16-24 (value only; Barrett constants live inside the library)
public:
    Modulus(uint64_t v = 0) : value_(v) {}
    uint64_t value() const { return value_; }
    bool isZero() const { return value_ == 0; }
private:
    uint64_t value_;
};

struct CoeffModulus { // src/modulus.h:485
    static std::vector<Modulus> Create(size_t poly_modulus_degree, std::vector<int> bit_sizes) {
        std::vector<uint64_t> out(bit_sizes.size());
        check(troyhip_coeff_modulus_create(poly_modulus_degree, bit_sizes.data(), (int)bit_sizes.size(), out.data()));
        return std::vector<Modulus>(out.begin(), out.end());
    }
};
struct PlainModulus { // src/modulus.h:528
    static Modulus Batching(size_t poly_modulus_degree, int bit_size) {
        uint64_t v;
        check(troyhip_plain_modulus_batching(poly_modulus_degree, bit_size, &v));
        return Modulus(v);
    }
};

class EncryptionParameters { // src/encryptionparams_cuda.cuh:63-170
public:
    explicit EncryptionParameters(SchemeType s = SchemeType::none) : scheme_(s) {}
    void setPolyModulusDegree(size_t n) { n_ = n; }
    void setCoeffModulus(const std::vector<Modulus> &q) { q_ = q; }
    void setPlainModulus(const Modulus &t) { t_ = t; }
    void setPlainModulus(uint64_t t) { t_ = Modulus(t); }
    SchemeType scheme() const { return scheme_; }
    size_t polyModulusDegree() const { return n_; }
    const std::vector<Modulus> &coeffModulus() const { return q_; }
    const Modulus &plainModulus() const { return t_; }
private:
    SchemeType scheme_;
    size_t n_ = 0;
    std::vector<Modulus> q_;
    Modulus t_;
};

using ParmsID = int; // the level is identified by its limb count (the reference hashes the parameters, src/encryptionparams.cpp:118-146)

class SEALContext { // src/context_cuda.cuh:146-186
public:
    SEALContext(const EncryptionParameters &parms, bool expand_mod_chain = true, SecurityLevel sec = SecurityLevel::tc128) : parms_(parms) {
        (void)expand_mod_chain;
        (void)sec; // SecurityLevel::none semantics: parameter security is not policed
        std::vector<uint64_t> q;
        for (auto &m : parms.coeffModulus()) q.push_back(m.value());
        troyhip_context *c = nullptr;
        check(troyhip_context_create((int)parms.scheme(), parms.polyModulusDegree(), q.data(), (int)q.size(), parms.plainModulus().value(), &c));
        ctx_.reset(c, [](troyhip_context *p) { troyhip_context_destroy(p); });
        check(troyhip_context_info(c, &info_));
    }
    troyhip_context *handle() const { return ctx_.get(); }
    const EncryptionParameters &parms() const { return parms_; }
    ParmsID keyParmsID() const { return info_.key_limbs; }
    ParmsID firstParmsID() const { return info_.first_limbs; }
    ParmsID lastParmsID() const { return info_.last_limbs; }
    size_t polyModulusDegree() const { return info_.poly_modulus_degree; }
    size_t keyLimbs() const { return (size_t)info_.key_limbs; }
private:
    EncryptionParameters parms_;
    std::shared_ptr<troyhip_context> ctx_;
    troyhip_context_info_t info_{};
};

// DeviceArray<uint64_t> (src/utils/devicearray.cuh): deep copy on copy, steal on move
class DeviceArray {
public:
    DeviceArray() = default;
    explicit DeviceArray(size_t words) { resize(words); }
    DeviceArray(const DeviceArray &o) { *this = o; }
    DeviceArray(DeviceArray &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DeviceArray &operator=(const DeviceArray &o) {
        if (this == &o) return *this;
        resize(o.n_);
        if (n_) check(troyhip_copy_d2d(p_, o.p_, n_ * 8, nullptr));
        return *this;
    }
    DeviceArray &operator=(DeviceArray &&o) noexcept { std::swap(p_, o.p_); std::swap(n_, o.n_); return *this; }
    ~DeviceArray() { if (p_) troyhip_free(p_); }
    void resize(size_t words) {
        if (words == n_) return;
        uint64_t *np = nullptr;
        if (words) check(troyhip_malloc((void **)&np, words * 8));
        if (p_ && np) check(troyhip_copy_d2d(np, p_, std::min(words, n_) * 8, nullptr));
        if (p_) { check(troyhip_stream_synchronize(nullptr)); troyhip_free(p_); }
        p_ = np;
        n_ = words;
    }
    uint64_t *get() const { return p_; }
    size_t size() const { return n_; }
private:
    uint64_t *p_ = nullptr;
    size_t n_ = 0;
};

class Plaintext { // src/plaintext.h: host coefficients (BFV/BGV: mod t; CKKS / NTT-form multiplyPlain: [limbs][N])
public:
    Plaintext() = default;
    explicit Plaintext(const std::vector<uint64_t> &coeffs) : data_(coeffs) {}
    // "1x^10 + 2"-style constructor of the reference is not reproduced; use setCoeff
    void resize(size_t n) { data_.resize(n, 0); }
    size_t coeffCount() const { return data_.size(); }
    uint64_t *data() { return data_.data(); }
    const uint64_t *data() const { return data_.data(); }
    uint64_t &operator[](size_t i) { return data_[i]; }
    const uint64_t &operator[](size_t i) const { return data_[i]; }
    bool operator==(const Plaintext &o) const {
        size_t n = std::max(data_.size(), o.data_.size());
        for (size_t i = 0; i < n; i++)
            if ((i < data_.size() ? data_[i] : 0) != (i < o.data_.size() ? o.data_[i] : 0)) return false;
        return true;
    }
    double &scale() { return scale_; }
    double scale() const { return scale_; }
    bool isNttForm() const { return ntt_limbs_ != 0; }
    ParmsID parmsID() const { return ntt_limbs_; } // level of an NTT-form plaintext (0 = coefficient form, parms_id_zero)
    void setNttForm(ParmsID limbs) { ntt_limbs_ = limbs; }
private:
    std::vector<uint64_t> data_;
    double scale_ = 1.0;
    ParmsID ntt_limbs_ = 0;
};

class Ciphertext { // src/ciphertext_cuda.cuh:12-268
public:
    Ciphertext() = default;
    explicit Ciphertext(const SEALContext &c) : n_(c.polyModulusDegree()) {}
    size_t size() const { return (size_t)d_.size; }
    size_t coeffModulusSize() const { return (size_t)d_.limbs; }
    size_t polyModulusDegree() const { return n_; }
    ParmsID parmsID() const { return d_.limbs; }
    bool isNttForm() const { return d_.is_ntt_form != 0; }
    bool &isNttFormRef() { ntt_shadow_ = d_.is_ntt_form != 0; return ntt_shadow_; }
    double &scale() { return d_.scale; }
    double scale() const { return d_.scale; }
    uint64_t &correctionFactor() { return d_.correction_factor; }
    uint64_t correctionFactor() const { return d_.correction_factor; }
    // device storage is kept at capacity max(size, 3) polynomials so multiply / relinearize run in place
    void resize(size_t n, size_t limbs, size_t size) {
        n_ = n;
        const size_t cap = std::max<size_t>(size, 3);
        buf_.resize(cap * limbs * n);
        d_.data = buf_.get();
        d_.batch_stride = cap * limbs * n;
        d_.size = (int)size;
        d_.limbs = (int)limbs;
    }
    std::vector<uint64_t> toHost() const { // CiphertextCuda::cpu / toHost: [size][limbs][N]
        std::vector<uint64_t> h(size() * coeffModulusSize() * n_);
        if (!h.empty()) check(troyhip_copy_d2h(h.data(), d_.data, h.size() * 8, nullptr));
        return h;
    }
    void fromHost(const std::vector<uint64_t> &h, size_t n, size_t limbs, size_t size, bool ntt, double scale = 1.0, uint64_t cf = 1) {
        resize(n, limbs, size);
        if (h.size() != size * limbs * n) throw std::invalid_argument("encrypted is not valid for encryption parameters");
        check(troyhip_copy_h2d(d_.data, h.data(), h.size() * 8, nullptr));
        d_.is_ntt_form = ntt; d_.scale = scale; d_.correction_factor = cf;
    }
    troyhip_ct *raw() { d_.data = buf_.get(); return &d_; }
    const troyhip_ct *raw() const { return &d_; }
    // wire format of CiphertextCuda::save / load / saveTerms / loadTerms (src/ciphertext_cuda.cu:16-143); the context supplies
    // the 256-bit parms_id the reference object carries itself (defined after Evaluator below)
    inline void save(std::ostream &stream, const SEALContext &context) const;
    inline void load(std::istream &stream, const SEALContext &context);
    inline void saveTerms(std::ostream &stream, const SEALContext &context, const class Evaluator &evaluator, const std::vector<size_t> &termIds) const;
    inline void loadTerms(std::istream &stream, const SEALContext &context, const class Evaluator &evaluator, const std::vector<size_t> &termIds);
    // value semantics: deep copy
    Ciphertext(const Ciphertext &o) : buf_(o.buf_), d_(o.d_), n_(o.n_) { d_.data = buf_.get(); }
    Ciphertext(Ciphertext &&o) noexcept = default;
    Ciphertext &operator=(const Ciphertext &o) { buf_ = o.buf_; d_ = o.d_; n_ = o.n_; d_.data = buf_.get(); return *this; }
    Ciphertext &operator=(Ciphertext &&o) noexcept = default;
private:
    DeviceArray buf_;
    troyhip_ct d_{nullptr, 0, 0, 0, 0, 1.0, 1};
    size_t n_ = 0;
    bool ntt_shadow_ = false;
};

class SecretKey { public: std::vector<uint64_t> data; };  // [K][N] NTT form (host), src/secretkey.h
class PublicKey { public: std::vector<uint64_t> data; };  // [2][K][N] NTT form (host), src/publickey.h

class KSwitchKeys { // src/kswitchkeys_cuda.cuh:43-56: data()[index] on the device, uploaded from the host key
public:
    bool hasKeyIndex(size_t index) const { return keys_.count(index) != 0; }
    const uint64_t *device(size_t index) const { return keys_.at(index)->get(); }
    void upload(size_t index, const std::vector<uint64_t> &host) {
        auto a = std::make_shared<DeviceArray>(host.size());
        check(troyhip_copy_h2d(a->get(), host.data(), host.size() * 8, nullptr));
        keys_[index] = a;
    }
    const std::map<size_t, std::shared_ptr<DeviceArray>> &all() const { return keys_; }
protected:
    std::map<size_t, std::shared_ptr<DeviceArray>> keys_;
};
class RelinKeys : public KSwitchKeys { // src/relinkeys_cuda.cuh:56-59
public:
    static size_t getIndex(size_t key_power) {
        if (key_power < 2) throw std::invalid_argument("key_power cannot be less than 2");
        return key_power - 2;
    }
    bool hasKey(size_t key_power) const { return hasKeyIndex(getIndex(key_power)); }
};
class GaloisKeys : public KSwitchKeys { // src/galoiskeys_cuda.cuh:74-77
public:
    static size_t getIndex(uint32_t galois_elt) { return (galois_elt - 1) >> 1; } // src/utils/galois_cuda.cuh:45-48
    bool hasKey(uint32_t galois_elt) const { return hasKeyIndex(getIndex(galois_elt)); }
};

class KeyGenerator { // src/keygenerator_cuda.cuh: runs on the CPU
public:
    // KeyGenerator(context): the key stream is seeded from the operating system's entropy source, as the reference's default
    // PRNG factory is (std::random_device, src/randomgen.cpp:23,72)
    explicit KeyGenerator(const SEALContext &c) : c_(c) {
        uint64_t s[2];
        check(troyhip_random_bytes(s, sizeof(s)));
        lo_ = s[0]; hi_ = s[1];
        generate();
    }
    // deterministic keys for tests and fixtures ONLY (the reference's counterpart: parms.setRandomGenerator with a fixed PRNGSeed)
    KeyGenerator(const SEALContext &c, uint64_t seed_lo, uint64_t seed_hi) : c_(c), lo_(seed_lo), hi_(seed_hi) { generate(); }
    const SecretKey &secretKey() const { return sk_; }
    void createPublicKey(PublicKey &pk) const { pk = pk_; }
    PublicKey createPublicKey() const { return pk_; }
    void createRelinKeys(RelinKeys &rlk) const {
        std::vector<uint64_t> h(ksk_words());
        check(troyhip_host_relin_key(c_.handle(), lo_, hi_, sk_.data.data(), h.data()));
        rlk.upload(RelinKeys::getIndex(2), h);
    }
    RelinKeys createRelinKeys() const { RelinKeys r; createRelinKeys(r); return r; }
    void createGaloisKeys(const std::vector<uint32_t> &galois_elts, GaloisKeys &gk) const {
        for (uint32_t e : galois_elts) {
            std::vector<uint64_t> h(ksk_words());
            check(troyhip_host_galois_key(c_.handle(), lo_, hi_, sk_.data.data(), e, h.data()));
            gk.upload(GaloisKeys::getIndex(e), h);
        }
    }
    void createGaloisKeys(const std::vector<int> &steps, GaloisKeys &gk) const {
        std::vector<uint32_t> elts;
        for (int s : steps) { uint32_t e; check(troyhip_galois_elt_from_step(c_.handle(), s, &e)); elts.push_back(e); }
        createGaloisKeys(elts, gk);
    }
private:
    void generate() {
        const size_t K = c_.keyLimbs(), N = c_.polyModulusDegree();
        sk_.data.resize(K * N);
        pk_.data.resize(2 * K * N);
        check(troyhip_host_keygen(c_.handle(), lo_, hi_, sk_.data.data(), pk_.data.data()));
    }
    size_t ksk_words() const { const size_t K = c_.keyLimbs(); return (K - 1) * 2 * K * c_.polyModulusDegree(); }
    const SEALContext &c_;
    uint64_t lo_, hi_;
    SecretKey sk_;
    PublicKey pk_;
};

class Encryptor { // src/encryptor_cuda.cuh (public-key path), CPU sampling + upload
public:
    // every encryption draws a fresh 128-bit seed for (u, e0, e1) from the operating system (src/randomgen.cpp:23,72)
    Encryptor(const SEALContext &c, const PublicKey &pk) : c_(c), pk_(pk), seeded_(false), lo_(0), hi_(0) {}
    // deterministic stream (seed, call counter) for tests ONLY
    Encryptor(const SEALContext &c, const PublicKey &pk, uint64_t seed_lo, uint64_t seed_hi = 0) : c_(c), pk_(pk), seeded_(true), lo_(seed_lo), hi_(seed_hi) {}
    void encrypt(const Plaintext &plain, Ciphertext &dst) const {
        const size_t N = c_.polyModulusDegree();
        const bool ckks = c_.parms().scheme() == SchemeType::ckks;
        const int limbs = ckks ? (int)(plain.coeffCount() / N) : c_.firstParmsID();
        std::vector<uint64_t> h((size_t)2 * limbs * N);
        uint64_t s[2] = {lo_ + (++counter_), hi_};
        if (!seeded_) check(troyhip_random_bytes(s, sizeof(s)));
        check(troyhip_host_encrypt(c_.handle(), s[0], s[1], pk_.data.data(), plain.data(), ckks ? N : plain.coeffCount(), limbs, h.data()));
        dst.fromHost(h, N, limbs, 2, ckks, ckks ? plain.scale() : 1.0, 1); // destination.scale() = plain.scale(): src/encryptor.cpp:235
    }
private:
    const SEALContext &c_;
    PublicKey pk_;
    bool seeded_;
    uint64_t lo_, hi_;
    mutable uint64_t counter_ = 0;
};

class Decryptor { // src/decryptor_cuda.cuh:13-60: the secret key is uploaded once, decryption runs on the device
public:
    Decryptor(const SEALContext &c, const SecretKey &sk) : c_(c), sk_(sk.data.size()) {
        check(troyhip_copy_h2d(sk_.get(), sk.data.data(), sk.data.size() * 8, nullptr));
    }
    void decrypt(const Ciphertext &ct, Plaintext &dst) const {
        const size_t N = c_.polyModulusDegree();
        const bool ckks = c_.parms().scheme() == SchemeType::ckks;
        const size_t words = ckks ? ct.coeffModulusSize() * N : N;
        DeviceArray out(words);
        check(troyhip_decrypt(c_.handle(), ct.raw(), sk_.get(), out.get(), words, 1, nullptr));
        dst.resize(words);
        check(troyhip_copy_d2h(dst.data(), out.get(), words * 8, nullptr));
        if (ckks) { dst.setNttForm(ct.parmsID()); dst.scale() = ct.scale(); }
    }
private:
    const SEALContext &c_;
    DeviceArray sk_;
};

class Evaluator { // src/evaluator_cuda.cuh:13-361 -- every method const, non-copyable
public:
    explicit Evaluator(const SEALContext &c) : c_(c) {}
    Evaluator(const Evaluator &) = delete;
    Evaluator &operator=(const Evaluator &) = delete;

    void negateInplace(Ciphertext &a) const { check(troyhip_negate(h(), a.raw(), 1, nullptr)); }
    void negate(const Ciphertext &a, Ciphertext &d) const { d = a; negateInplace(d); }
    void addInplace(Ciphertext &a, const Ciphertext &b) const { check(troyhip_add(h(), a.raw(), b.raw(), 1, nullptr)); }
    void add(const Ciphertext &a, const Ciphertext &b, Ciphertext &d) const { d = a; addInplace(d, b); }
    void addMany(const std::vector<Ciphertext> &v, Ciphertext &d) const {
        if (v.empty()) throw std::invalid_argument("encrypteds cannot be empty");
        d = v[0];
        for (size_t i = 1; i < v.size(); i++) addInplace(d, v[i]);
    }
    void subInplace(Ciphertext &a, const Ciphertext &b) const { check(troyhip_sub(h(), a.raw(), b.raw(), 1, nullptr)); }
    void sub(const Ciphertext &a, const Ciphertext &b, Ciphertext &d) const { d = a; subInplace(d, b); }
    void multiplyInplace(Ciphertext &a, const Ciphertext &b) const {
        grow(a, a.size() + b.size() - 1);
        check(troyhip_multiply(h(), a.raw(), b.raw(), a.raw(), 1, nullptr));
    }
    void multiply(const Ciphertext &a, const Ciphertext &b, Ciphertext &d) const { d = a; multiplyInplace(d, b); }
    void squareInplace(Ciphertext &a) const { multiplyInplace(a, a); }
    void square(const Ciphertext &a, Ciphertext &d) const { d = a; squareInplace(d); }
    void relinearizeInplace(Ciphertext &a, const RelinKeys &k) const { // to size 2 from any size <= 16 (src/evaluator_cuda.cu:703-744)
        const size_t need = a.size() > 2 ? a.size() - 2 : 0;
        std::vector<const uint64_t *> keys(need ? need : 1, nullptr);
        for (size_t i = 0; i < need; i++) {
            if (!k.hasKey(i + 2)) throw std::invalid_argument("not enough relinearization keys");
            keys[i] = k.device(RelinKeys::getIndex(i + 2));
        }
        check(troyhip_relinearize_keys(h(), a.raw(), keys.data(), (int)need, 1, nullptr));
    }
    void relinearize(const Ciphertext &a, const RelinKeys &k, Ciphertext &d) const { d = a; relinearizeInplace(d, k); }
    // applyKeySwitchingInplace (evaluator_cuda.cu:1365-1378), negacyclicShiftInplace (:2342-2351)
    void applyKeySwitchingInplace(Ciphertext &a, const KSwitchKeys &k) const {
        if (k.all().size() != 1) throw std::invalid_argument("kswitch_keys.data().size() != 1");
        check(troyhip_apply_key_switching(h(), a.raw(), k.all().begin()->second->get(), 1, nullptr));
    }
    void negacyclicShiftInplace(Ciphertext &a, size_t shift) const { check(troyhip_negacyclic_shift(h(), a.raw(), shift, 1, nullptr)); }
    // multiplyMany / exponentiate (src/evaluator.cpp:1502-1601): pairwise products appended to the work list, each relinearized
    void multiplyMany(const std::vector<Ciphertext> &v, const RelinKeys &rk, Ciphertext &d) const {
        if (v.empty()) throw std::invalid_argument("encrypteds vector must not be empty");
        need(SchemeType::ckks, false);
        if (v.size() == 1) { d = v[0]; return; }
        std::vector<Ciphertext> work;
        for (size_t i = 0; i + 1 < v.size(); i += 2) { Ciphertext t; multiply(v[i], v[i + 1], t); relinearizeInplace(t, rk); work.push_back(std::move(t)); }
        if (v.size() & 1) work.push_back(v.back());
        for (size_t i = 0; i + 1 < work.size(); i += 2) { Ciphertext t; multiply(work[i], work[i + 1], t); relinearizeInplace(t, rk); work.push_back(std::move(t)); }
        d = work.back();
    }
    void exponentiateInplace(Ciphertext &a, uint64_t exponent, const RelinKeys &rk) const {
        if (exponent == 0) throw std::invalid_argument("exponent cannot be 0");
        if (exponent == 1) return;
        std::vector<Ciphertext> v((size_t)exponent, a);
        multiplyMany(v, rk, a);
    }
    void modSwitchToNextInplace(Ciphertext &a) const { next(a, troyhip_mod_switch_to_next); }
    void modSwitchToNext(const Ciphertext &a, Ciphertext &d) const { d = a; modSwitchToNextInplace(d); }
    void modSwitchToInplace(Ciphertext &a, ParmsID parms_id) const {
        if (parms_id > a.parmsID()) throw std::invalid_argument("cannot switch to higher level modulus");
        while (a.parmsID() != parms_id) modSwitchToNextInplace(a);
    }
    void rescaleToNextInplace(Ciphertext &a) const { next(a, troyhip_rescale_to_next); }
    void rescaleToNext(const Ciphertext &a, Ciphertext &d) const { d = a; rescaleToNextInplace(d); }
    void rescaleToInplace(Ciphertext &a, ParmsID parms_id) const {
        if (parms_id > a.parmsID()) throw std::invalid_argument("cannot switch to higher level modulus");
        while (a.parmsID() != parms_id) rescaleToNextInplace(a);
    }
    void applyGaloisInplace(Ciphertext &a, uint32_t galois_elt, const GaloisKeys &gk) const {
        if (!gk.hasKey(galois_elt)) throw std::invalid_argument("Galois key not present");
        check(troyhip_apply_galois(h(), a.raw(), galois_elt, gk.device(GaloisKeys::getIndex(galois_elt)), 1, nullptr));
    }
    void rotateRowsInplace(Ciphertext &a, int steps, const GaloisKeys &gk) const { need(SchemeType::ckks, false); rotate(a, steps, 0, gk); }
    void rotateColumnsInplace(Ciphertext &a, const GaloisKeys &gk) const { need(SchemeType::ckks, false); rotate(a, 0, 1, gk); }
    void rotateVectorInplace(Ciphertext &a, int steps, const GaloisKeys &gk) const { need(SchemeType::ckks, true); rotate(a, steps, 0, gk); }
    void complexConjugateInplace(Ciphertext &a, const GaloisKeys &gk) const { need(SchemeType::ckks, true); rotate(a, 0, 1, gk); }
    void transformToNttInplace(Ciphertext &a) const { check(troyhip_transform_to_ntt(h(), a.raw(), 1, nullptr)); }
    void transformFromNttInplace(Ciphertext &a) const { check(troyhip_transform_from_ntt(h(), a.raw(), 1, nullptr)); }
    // multiplyPlainInplace (evaluator_cuda.cu:1722-1755): NTT-form pair -> multiplyPlainNtt, coefficient-form pair -> multiplyPlainNormal
    void multiplyPlainInplace(Ciphertext &a, const Plaintext &plain) const {
        if (a.isNttForm() != plain.isNttForm() && c_.parms().scheme() != SchemeType::ckks) throw std::invalid_argument("NTT form mismatch");
        DeviceArray p(plain.coeffCount());
        check(troyhip_copy_h2d(p.get(), plain.data(), plain.coeffCount() * 8, nullptr));
        if (a.isNttForm()) check(troyhip_multiply_plain_ntt(h(), a.raw(), p.get(), plain.scale(), 1, nullptr));
        else check(troyhip_multiply_plain(h(), a.raw(), p.get(), plain.coeffCount(), 0, 1, nullptr));
        check(troyhip_stream_synchronize(nullptr));
    }
    void multiplyPlain(const Ciphertext &a, const Plaintext &plain, Ciphertext &d) const { d = a; multiplyPlainInplace(d, plain); }
    // addPlainInplace / subPlainInplace (evaluator_cuda.cu:1654-1720)
    void addPlainInplace(Ciphertext &a, const Plaintext &plain) const { plain_addsub(a, plain, 0); }
    void subPlainInplace(Ciphertext &a, const Plaintext &plain) const { plain_addsub(a, plain, 1); }
    void addPlain(const Ciphertext &a, const Plaintext &plain, Ciphertext &d) const { d = a; addPlainInplace(d, plain); }
    void subPlain(const Ciphertext &a, const Plaintext &plain, Ciphertext &d) const { d = a; subPlainInplace(d, plain); }
    // transformToNttInplace(Plaintext&, parms_id) (evaluator_cuda.cu:1866-1948)
    void transformToNttInplace(Plaintext &plain, ParmsID parms_id) const {
        if (plain.isNttForm()) throw std::invalid_argument("plain is already in NTT form");
        const size_t n = c_.polyModulusDegree();
        DeviceArray p(plain.coeffCount()), out((size_t)parms_id * n);
        check(troyhip_copy_h2d(p.get(), plain.data(), plain.coeffCount() * 8, nullptr));
        check(troyhip_plain_to_ntt(h(), p.get(), plain.coeffCount(), 0, parms_id, out.get(), 1, nullptr));
        plain.resize((size_t)parms_id * n);
        check(troyhip_copy_d2h(plain.data(), out.get(), (size_t)parms_id * n * 8, nullptr));
        plain.setNttForm(parms_id);
    }

private:
    troyhip_context *h() const { return c_.handle(); }
    void plain_addsub(Ciphertext &a, const Plaintext &plain, int sub) const {
        DeviceArray p(plain.coeffCount());
        check(troyhip_copy_h2d(p.get(), plain.data(), plain.coeffCount() * 8, nullptr));
        check(troyhip_add_plain(h(), a.raw(), p.get(), plain.coeffCount(), 0, plain.scale(), sub, 1, nullptr));
        check(troyhip_stream_synchronize(nullptr));
    }
    void need(SchemeType s, bool equal) const {
        if ((c_.parms().scheme() == s) != equal) throw std::logic_error("unsupported scheme");
    }
    static void grow(Ciphertext &a, size_t size) { // keep device capacity >= size (and >= 3)
        if (a.raw()->batch_stride < size * a.coeffModulusSize() * a.polyModulusDegree()) {
            Ciphertext b;
            b.resize(a.polyModulusDegree(), a.coeffModulusSize(), size);
            check(troyhip_copy_d2d(b.raw()->data, a.raw()->data, a.size() * a.coeffModulusSize() * a.polyModulusDegree() * 8, nullptr));
            troyhip_ct *rb = b.raw();
            rb->size = a.raw()->size; rb->is_ntt_form = a.raw()->is_ntt_form; rb->scale = a.raw()->scale; rb->correction_factor = a.raw()->correction_factor;
            a = std::move(b);
        }
    }
    template <class F> void next(Ciphertext &a, F fn) const {
        Ciphertext out;
        out.resize(a.polyModulusDegree(), a.coeffModulusSize() > 1 ? a.coeffModulusSize() - 1 : 1, a.size());
        check(fn(h(), a.raw(), out.raw(), 1, nullptr));
        a = std::move(out);
    }
    void rotate(Ciphertext &a, int steps, int conj, const GaloisKeys &gk) const {
        std::vector<uint32_t> elts;
        std::vector<const uint64_t *> ptrs;
        for (auto &kv : gk.all()) { elts.push_back((uint32_t)(2 * kv.first + 1)); ptrs.push_back(kv.second->get()); }
        check(troyhip_rotate(h(), a.raw(), steps, conj, elts.data(), ptrs.data(), (int)elts.size(), 1, nullptr));
    }
    const SEALContext &c_;
};

// ---- ciphertext serialization: a raw little-endian field dump (src/serialize.h savet/loadt)
namespace wire {
template <class T> inline void put(std::ostream &s, const T &v) { s.write(reinterpret_cast<const char *>(&v), sizeof(T)); }
template <class T> inline T get(std::istream &s) {
    T v{};
    s.read(reinterpret_cast<char *>(&v), sizeof(T));
    if (!s) throw std::invalid_argument("stream ended inside a ciphertext");
    return v;
}
struct Header { bool ntt; size_t size, n, limbs; double scale; uint64_t cf, seed; bool terms; };
inline void put_header(std::ostream &s, const SEALContext &c, const Ciphertext &ct, bool terms) {
    uint64_t id[4];
    check(troyhip_context_parms_id(c.handle(), (int)ct.coeffModulusSize(), id));
    s.write(reinterpret_cast<const char *>(id), 32);
    put<bool>(s, ct.isNttForm()); put<size_t>(s, ct.size()); put<size_t>(s, ct.polyModulusDegree()); put<size_t>(s, ct.coeffModulusSize());
    put<double>(s, ct.scale()); put<uint64_t>(s, ct.correctionFactor()); put<uint64_t>(s, 0); put<bool>(s, terms);
}
inline Header get_header(std::istream &s, const SEALContext &c) {
    uint64_t id[4], mine[4];
    s.read(reinterpret_cast<char *>(id), 32);
    Header h;
    h.ntt = get<bool>(s); h.size = get<size_t>(s); h.n = get<size_t>(s); h.limbs = get<size_t>(s);
    h.scale = get<double>(s); h.cf = get<uint64_t>(s); h.seed = get<uint64_t>(s); h.terms = get<bool>(s);
    if (h.n != c.polyModulusDegree() || h.limbs < 1 || h.limbs > c.keyLimbs() || h.size < 1) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    check(troyhip_context_parms_id(c.handle(), (int)h.limbs, mine));
    if (!std::equal(id, id + 4, mine)) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    if (h.seed) throw std::invalid_argument("seed is not zero.");
    return h;
}
} // namespace wire

inline void Ciphertext::save(std::ostream &stream, const SEALContext &context) const {
    wire::put_header(stream, context, *this, false);
    const std::vector<uint64_t> h = toHost();
    wire::put<size_t>(stream, h.size());
    stream.write(reinterpret_cast<const char *>(h.data()), (std::streamsize)(h.size() * 8));
}
inline void Ciphertext::load(std::istream &stream, const SEALContext &context) {
    const wire::Header h = wire::get_header(stream, context);
    if (h.terms) throw std::invalid_argument("Trying to load a termed ciphertext, but indices is not specified");
    const size_t words = wire::get<size_t>(stream);
    if (words != h.size * h.limbs * h.n) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    std::vector<uint64_t> host(words);
    stream.read(reinterpret_cast<char *>(host.data()), (std::streamsize)(words * 8));
    if (!stream) throw std::invalid_argument("stream ended inside a ciphertext");
    fromHost(host, h.n, h.limbs, h.size, h.ntt, h.scale, h.cf);
}
inline void Ciphertext::saveTerms(std::ostream &stream, const SEALContext &context, const Evaluator &evaluator, const std::vector<size_t> &termIds) const {
    std::vector<uint64_t> h;
    if (isNttForm()) {
        Ciphertext copy = *this;
        evaluator.transformFromNttInplace(copy);
        h = copy.toHost();
    } else h = toHost();
    wire::put_header(stream, context, *this, true);
    const size_t n = polyModulusDegree(), limbs = coeffModulusSize();
    for (size_t id : termIds) {
        if (id >= n) throw std::invalid_argument("term index out of range");
        for (size_t j = 0; j < limbs; j++) wire::put<uint64_t>(stream, h[j * n + id]);
    }
    const size_t offset = n * limbs;
    wire::put<size_t>(stream, h.size() - offset);
    stream.write(reinterpret_cast<const char *>(h.data() + offset), (std::streamsize)((h.size() - offset) * 8));
}
inline void Ciphertext::loadTerms(std::istream &stream, const SEALContext &context, const Evaluator &evaluator, const std::vector<size_t> &termIds) {
    const wire::Header h = wire::get_header(stream, context);
    if (!h.terms) throw std::invalid_argument("Trying to load a normal ciphertext, but term indices is specified");
    std::vector<uint64_t> host(h.size * h.limbs * h.n, 0); // unlisted coefficients of c0: zero
    for (size_t id : termIds) {
        if (id >= h.n) throw std::invalid_argument("term index out of range");
        for (size_t j = 0; j < h.limbs; j++) host[j * h.n + id] = wire::get<uint64_t>(stream);
    }
    const size_t offset = h.n * h.limbs, words = wire::get<size_t>(stream);
    if (words != host.size() - offset) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    stream.read(reinterpret_cast<char *>(host.data() + offset), (std::streamsize)(words * 8));
    if (!stream) throw std::invalid_argument("stream ended inside a ciphertext");
    fromHost(host, h.n, h.limbs, h.size, false, h.scale, h.cf);
    if (h.ntt) evaluator.transformToNttInplace(*this);
}

} // namespace troyn
