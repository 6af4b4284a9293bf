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
#include <array>
#include <cmath>
#include <complex>
#include <cstdint>
#include <functional>
#include <istream>
#include <limits>
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

class KernelProvider { // src/kernelprovider.cuh:24-85
public:
    static void initialize(int device = 0) { check(troyhip_initialize(device)); }
    static void checkInitialized() { if (!troyhip_is_initialized()) throw std::invalid_argument("KernelProvider not initialized."); } // :24-27
    // the reference's public statics (:35-85): lengths in ELEMENTS of T; zero lengths are no-ops (malloc returns nullptr).  malloc / free go through the
    // library's caching pool (the reference's cudaMalloc / cudaFree synchronise the device, a pooled pair does not); the copies are synchronous like
    // cudaMemcpy, on the calling thread's current device.
    template <typename T> static T *malloc(size_t length) {
        checkInitialized();
        if (length == 0) return nullptr;
        void *p = nullptr;
        check(troyhip_malloc(&p, length * sizeof(T)));
        return static_cast<T *>(p);
    }
    template <typename T> static void free(T *pointer) { checkInitialized(); if (pointer) { troyhip_stream_synchronize(nullptr); troyhip_free((void *)pointer); } }
    template <typename T> static void copy(T *deviceDestPtr, const T *hostFromPtr, size_t length) {
        checkInitialized();
        if (length) check(troyhip_copy_h2d(deviceDestPtr, hostFromPtr, length * sizeof(T), nullptr));
    }
    template <typename T> static void copyOnDevice(T *deviceDestPtr, const T *deviceFromPtr, size_t length) {
        checkInitialized();
        if (!length) return;
        check(troyhip_copy_d2d(deviceDestPtr, deviceFromPtr, length * sizeof(T), nullptr));
        check(troyhip_stream_synchronize(nullptr));
    }
    template <typename T> static void retrieve(T *hostDestPtr, const T *deviceFromPtr, size_t length) {
        checkInitialized();
        if (length) check(troyhip_copy_d2h(hostDestPtr, deviceFromPtr, length * sizeof(T), nullptr));
    }
    template <typename T> static void memsetZero(T *devicePtr, size_t length) {
        if (!length) return;
        check(troyhip_memset_zero(devicePtr, length * sizeof(T), nullptr));
        check(troyhip_stream_synchronize(nullptr));
    }
    // more than one GPU in one process (no reference counterpart: the reference is cudaSetDevice(0)): HIP's current device is per host thread
    static int deviceCount() { int n = 0; check(troyhip_device_count(&n)); return n; }
    static void setDevice(int device) { check(troyhip_set_device(device)); }
    static int currentDevice() { int d = 0; check(troyhip_get_device(&d)); return d; }
};

class Modulus { // src/modulus.h:16-24 (value only; Barrett constants live inside the library)
public:
    Modulus(uint64_t v = 0) : value_(v) {}
    uint64_t value() const { return value_; }
    bool isZero() const { return value_ == 0; }
    int bitCount() const { return value_ ? 64 - __builtin_clzll(value_) : 0; } // src/modulus.h: bit_count_
    size_t uint64Count() const { return value_ ? 1 : 0; }
    uint64_t reduce(uint64_t v) const { // src/modulus.h:362-370
        if (!value_) throw std::logic_error("cannot reduce modulo a zero modulus");
        return v % value_;
    }
    bool operator==(const Modulus &o) const { return value_ == o.value_; }
    bool operator!=(const Modulus &o) const { return value_ != o.value_; }
    bool operator==(uint64_t v) const { return value_ == v; }
    bool operator!=(uint64_t v) const { return value_ != v; }
    bool operator<(const Modulus &o) const { return value_ < o.value_; }
    // is_prime_ (src/modulus.cpp:80-121 via util::isPrime): Miller-Rabin with the twelve bases that decide every 64-bit integer
    bool isPrime() const {
        const uint64_t n = value_;
        if (n < 2) return false;
        for (uint64_t p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
            if (n == p) return true;
            if (n % p == 0) return false;
        }
        uint64_t d = n - 1;
        int r = 0;
        while (!(d & 1)) { d >>= 1; r++; }
        auto mul = [n](uint64_t a, uint64_t b) { return (uint64_t)((unsigned __int128)a * b % n); };
        for (uint64_t a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
            uint64_t x = 1, b = a % n, e = d;
            for (; e; e >>= 1, b = mul(b, b))
                if (e & 1) x = mul(x, b);
            if (x == 1 || x == n - 1) continue;
            bool witness = true;
            for (int i = 1; i < r && witness; i++) {
                x = mul(x, x);
                if (x == n - 1) witness = false;
            }
            if (witness) return false;
        }
        return true;
    }
    // const_ratio_ (src/modulus.h:16-24): floor(2^128 / value) as two words, then 2^128 mod value
    std::array<uint64_t, 3> constRatio() const {
        if (!value_) return {{0, 0, 0}};
        const unsigned __int128 top = ~(unsigned __int128)0; // 2^128 - 1
        unsigned __int128 q = top / value_;
        uint64_t r = (uint64_t)(top % value_) + 1;           // 2^128 = q value + r
        if (r == value_) { q += 1; r = 0; }
        return {{(uint64_t)q, (uint64_t)(q >> 64), r}};
    }
private:
    uint64_t value_;
};

struct CoeffModulus { // src/modulus.h:440-520
#include "troyn_hestd.inc"
    // MaxBitCount (src/modulus.cpp:14-51): the largest total coefficient-modulus bit count the security standard allows at this degree
    static int MaxBitCount(size_t poly_modulus_degree, SecurityLevel sec_level = SecurityLevel::tc128) {
        return sec_level == SecurityLevel::none ? std::numeric_limits<int>::max() : stdMaxBits((int)sec_level, poly_modulus_degree);
    }
    // BFVDefault (src/modulus.cpp:53-78): the primes SEAL ships for the standard degrees
    static std::vector<Modulus> BFVDefault(size_t poly_modulus_degree, SecurityLevel sec_level = SecurityLevel::tc128) {
        if (!MaxBitCount(poly_modulus_degree, sec_level)) throw std::invalid_argument("non-standard poly_modulus_degree");
        if (sec_level == SecurityLevel::none) throw std::invalid_argument("invalid security level");
        const std::vector<uint64_t> primes = stdDefaultPrimes((int)sec_level, poly_modulus_degree);
        if (primes.empty()) throw std::out_of_range("no default coefficient modulus for this poly_modulus_degree"); // map::at in the reference
        return std::vector<Modulus>(primes.begin(), primes.end());
    }
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

// ParmsID (src/encryptionparams.h: util::HashFunction::hash_block_type = std::array<uint64_t, 4>): the BLAKE2b-256 hash of
// the level's parameters (src/encryptionparams.cpp:118-146), computed by the library (troyhip_context_parms_id).  The shim
// also carries the level's limb count, which is what the C ABI identifies a level by; comparisons are on the hash.
struct ParmsID : std::array<uint64_t, 4> {
    int limbs = 0;
    ParmsID() : std::array<uint64_t, 4>{{0, 0, 0, 0}} {}
    bool operator==(const ParmsID &o) const { return static_cast<const std::array<uint64_t, 4> &>(*this) == static_cast<const std::array<uint64_t, 4> &>(o); }
    bool operator!=(const ParmsID &o) const { return !(*this == o); }
};
static const ParmsID parmsIDZero{}; // src/encryptionparams.h: parmsIDZero

// EncryptionParameterQualifiers (src/context.h:22-215): what the context found out about a level's parameters.  A context that was
// constructed has valid parameters (invalid ones throw from the constructor), so parameter_error is always success here.
struct EncryptionParameterQualifiers {
    enum class ErrorType : int { none = -1, success = 0 };
    ErrorType parameter_error = ErrorType::success;
    bool parametersSet() const noexcept { return parameter_error == ErrorType::success; }
    const char *parameterErrorName() const noexcept { return "success"; }
    const char *parameterErrorMessage() const noexcept { return "valid"; }
    bool using_fft = true, using_ntt = true, using_batching = false, using_fast_plain_lift = false, using_descending_modulus_chain = false;
    SecurityLevel sec_level = SecurityLevel::none;
};

class ContextData; // below

class SEALContext { // src/context_cuda.cuh:146-186
public:
    using ContextDataCuda = ContextData; // the nested name user code spells (src/context_cuda.cuh:20)
    SEALContext(const EncryptionParameters &parms, bool expand_mod_chain = true, SecurityLevel sec = SecurityLevel::tc128) : parms_(parms) {
        (void)expand_mod_chain;
        std::vector<uint64_t> q;
        for (auto &m : parms.coeffModulus()) q.push_back(m.value());
        // the bit count of the PRODUCT of the primes (total_coeff_modulus_bit_count_, src/context.cpp:180-183), which can be up to k - 1 bits
        // below the sum of the primes' own bit counts
        int total_bits = 0;
        {
            std::vector<uint64_t> prod{1};
            for (uint64_t v : q) {
                unsigned __int128 carry = 0;
                for (auto &w : prod) { const unsigned __int128 t = (unsigned __int128)w * v + carry; w = (uint64_t)t; carry = t >> 64; }
                if (carry) prod.push_back((uint64_t)carry);
            }
            while (prod.size() > 1 && !prod.back()) prod.pop_back();
            total_bits = (int)(64 * (prod.size() - 1));
            for (uint64_t t = prod.back(); t; t >>= 1) total_bits++;
        }
        // src/context.cpp:181-197: with a security level, the key-level modulus must fit the standard's bound (the reference records
        // ErrorType::invalid_parameters_insecure and every later use throws; here invalid parameters throw from the constructor)
        if (sec != SecurityLevel::none && total_bits > CoeffModulus::MaxBitCount(parms.polyModulusDegree(), sec))
            throw std::invalid_argument("encryption parameters are not set correctly: parameters are not compliant with HomomorphicEncryption.org security standard");
        sec_ = sec;
        troyhip_context *c = nullptr;
        check(troyhip_context_create((int)parms.scheme(), parms.polyModulusDegree(), q.data(), (int)q.size(), parms.plainModulus().value(), &c));
        ctx_.reset(c, [](troyhip_context *p) { troyhip_context_destroy(p); });
        check(troyhip_context_info(c, &info_));
        auto ids = std::make_shared<std::vector<ParmsID>>((size_t)info_.key_limbs + 1);
        for (int l = info_.last_limbs; l <= info_.key_limbs; l++) {
            if (l != info_.key_limbs && l > info_.first_limbs) continue;
            ParmsID id;
            check(troyhip_context_parms_id(c, l, id.data()));
            id.limbs = l;
            (*ids)[(size_t)l] = id;
        }
        ids_ = ids;
    }
    troyhip_context *handle() const { return ctx_.get(); }
    int device() const { int d = 0; check(troyhip_context_device(ctx_.get(), &d)); return d; } // the HIP device this context lives on (the creating thread's current one)
    const EncryptionParameters &parms() const { return parms_; }
    const ParmsID &keyParmsID() const { return (*ids_)[(size_t)info_.key_limbs]; }
    const ParmsID &firstParmsID() const { return (*ids_)[(size_t)info_.first_limbs]; }
    const ParmsID &lastParmsID() const { return (*ids_)[(size_t)info_.last_limbs]; }
    // the ParmsID of the level with `limbs` primes (parmsIDZero if there is no such level)
    const ParmsID &parmsIDOfLimbs(size_t limbs) const { return limbs < ids_->size() ? (*ids_)[limbs] : parmsIDZero; }
    std::shared_ptr<const std::vector<ParmsID>> levelIDs() const { return ids_; }
    // getContextData / firstContextData / keyContextData / lastContextData (src/context_cuda.cuh:146-186): nullptr for an unknown id
    inline std::shared_ptr<const ContextData> getContextData(const ParmsID &parms_id) const;
    std::shared_ptr<const ContextData> keyContextData() const { return getContextData(keyParmsID()); }
    std::shared_ptr<const ContextData> firstContextData() const { return getContextData(firstParmsID()); }
    std::shared_ptr<const ContextData> lastContextData() const { return getContextData(lastParmsID()); }
    bool using_keyswitching() const { return info_.key_limbs > 1; }
    size_t polyModulusDegree() const { return info_.poly_modulus_degree; }
    size_t keyLimbs() const { return (size_t)info_.key_limbs; }
    size_t firstLimbs() const { return (size_t)info_.first_limbs; }
    size_t lastLimbs() const { return (size_t)info_.last_limbs; }
    SecurityLevel securityLevel() const { return sec_; }
private:
    EncryptionParameters parms_;
    SecurityLevel sec_ = SecurityLevel::none;
    std::shared_ptr<troyhip_context> ctx_;
    troyhip_context_info_t info_{};
    std::shared_ptr<std::vector<ParmsID>> ids_;
};

// SEALContextCuda::ContextDataCuda (src/context_cuda.cuh:20-140), the members user code reads: the level's parameters, its id,
// its position in the chain (chain_index counts down to 0 at the last level, src/context.cpp:522-529) and its neighbours
class ContextData {
public:
    ContextData(const SEALContext &c, size_t limbs) : c_(&c), limbs_(limbs), parms_(c.parms().scheme()) {
        parms_.setPolyModulusDegree(c.parms().polyModulusDegree());
        parms_.setPlainModulus(c.parms().plainModulus());
        std::vector<Modulus> q(c.parms().coeffModulus().begin(), c.parms().coeffModulus().begin() + (long)limbs);
        parms_.setCoeffModulus(q);
    }
    const EncryptionParameters &parms() const { return parms_; }
    const ParmsID &parmsID() const { return c_->parmsIDOfLimbs(limbs_); }
    // the product of the level's primes as little-endian words, one per prime (ContextData::totalCoeffModulus, src/context.h)
    std::vector<uint64_t> totalCoeffModulus() const {
        std::vector<uint64_t> q(limbs_, 0);
        q[0] = 1;
        for (auto &m : parms_.coeffModulus()) {
            unsigned __int128 carry = 0;
            for (auto &w : q) { unsigned __int128 t = (unsigned __int128)w * m.value() + carry; w = (uint64_t)t; carry = t >> 64; }
        }
        return q;
    }
    int totalCoeffModulusBitCount() const {
        const std::vector<uint64_t> q = totalCoeffModulus();
        for (size_t i = q.size(); i-- > 0;)
            if (q[i]) return (int)(64 * i) + 64 - __builtin_clzll(q[i]);
        return 0;
    }
    // qualifiers() (src/context_cuda.cuh:92, src/context.cpp:286-305, 364-368, 411-417): batching = t is a prime congruent to 1 modulo 2N (an NTT
    // modulo t exists); fast plain lift = every prime of the level exceeds t; descending chain = the level's primes strictly decrease
    EncryptionParameterQualifiers qualifiers() const {
        EncryptionParameterQualifiers q;
        const auto &primes = parms_.coeffModulus();
        const uint64_t t = parms_.plainModulus().value(), N = parms_.polyModulusDegree();
        if (parms_.scheme() == SchemeType::ckks) q.using_batching = true;
        else {
            q.using_batching = t > 2 && t % (2 * N) == 1 && Modulus(t).isPrime();
            q.using_fast_plain_lift = true;
            for (auto &m : primes) q.using_fast_plain_lift = q.using_fast_plain_lift && m.value() > t;
        }
        q.sec_level = c_->securityLevel();
        q.using_descending_modulus_chain = true;
        for (size_t i = 0; i + 1 < primes.size(); i++) q.using_descending_modulus_chain = q.using_descending_modulus_chain && primes[i].value() > primes[i + 1].value();
        return q;
    }
    size_t chainIndex() const { return limbs_ == c_->keyLimbs() ? (c_->using_keyswitching() ? c_->firstLimbs() - c_->lastLimbs() + 1 : 0) : limbs_ - c_->lastLimbs(); }
    std::shared_ptr<const ContextData> nextContextData() const { // one prime fewer; none below the last level
        if (limbs_ == c_->keyLimbs() && c_->using_keyswitching()) return c_->firstContextData();
        return limbs_ > c_->lastLimbs() ? c_->getContextData(c_->parmsIDOfLimbs(limbs_ - 1)) : nullptr;
    }
    std::shared_ptr<const ContextData> prevContextData() const {
        if (limbs_ == c_->keyLimbs()) return nullptr;
        return limbs_ == c_->firstLimbs() ? c_->keyContextData() : c_->getContextData(c_->parmsIDOfLimbs(limbs_ + 1));
    }
private:
    const SEALContext *c_;
    size_t limbs_;
    EncryptionParameters parms_;
};
inline std::shared_ptr<const ContextData> SEALContext::getContextData(const ParmsID &parms_id) const {
    for (const ParmsID &id : *ids_)
        if (id.limbs && id == parms_id) return std::make_shared<const ContextData>(*this, (size_t)id.limbs);
    return nullptr;
}

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

// ---- serialization: a raw little-endian field dump (src/serialize.h savet / loadt)
namespace wire {
template <class T> inline void put(std::ostream &s, const T &v) { s.write(reinterpret_cast<const char *>(&v), sizeof(T)); }
template <class T> inline T get(std::istream &s) {
    T v{};
    s.read(reinterpret_cast<char *>(&v), sizeof(T));
    if (!s) throw std::invalid_argument("stream ended inside a serialized object");
    return v;
}
inline void put_words(std::ostream &s, const uint64_t *w, size_t count) { s.write(reinterpret_cast<const char *>(w), (std::streamsize)(count * 8)); }
inline void get_words(std::istream &s, uint64_t *w, size_t count) {
    s.read(reinterpret_cast<char *>(w), (std::streamsize)(count * 8));
    if (!s) throw std::invalid_argument("stream ended inside a serialized object");
}
// `words` payload words into a vector that grows with what the stream really delivers (16 MiB at a time): a forged header cannot make the
// loader allocate gigabytes for a stream of a few bytes
inline std::vector<uint64_t> get_vector(std::istream &s, size_t words, size_t room = 0) {
    std::vector<uint64_t> v;
    const size_t step = size_t(1) << 21;
    for (size_t at = 0; at < words;) {
        const size_t n = std::min(step, words - at);
        v.resize(at + n);
        get_words(s, v.data() + at, n);
        at += n;
    }
    v.resize(words + room, 0);
    return v;
}
constexpr size_t max_ct_size = 16; // polynomials per ciphertext the library handles (relinearize: src/evaluator_cuda.cu:703-744 takes "any size <= 16")
// the fields CiphertextCuda::save writes ahead of the data (src/ciphertext_cuda.cu:16-25): parms_id, is_ntt_form, size, poly_modulus_degree,
// coeff_modulus_size, scale, correction_factor, seed, terms
struct CtFields { uint64_t id[4]; bool ntt; size_t size, n, limbs; double scale; uint64_t cf, seed; bool terms; };
inline void put_fields(std::ostream &s, const CtFields &f) {
    s.write(reinterpret_cast<const char *>(f.id), 32);
    put<bool>(s, f.ntt); put<size_t>(s, f.size); put<size_t>(s, f.n); put<size_t>(s, f.limbs);
    put<double>(s, f.scale); put<uint64_t>(s, f.cf); put<uint64_t>(s, f.seed); put<bool>(s, f.terms);
}
inline CtFields get_fields(std::istream &s) {
    CtFields f;
    s.read(reinterpret_cast<char *>(f.id), 32);
    f.ntt = get<bool>(s); f.size = get<size_t>(s); f.n = get<size_t>(s); f.limbs = get<size_t>(s);
    f.scale = get<double>(s); f.cf = get<uint64_t>(s); f.seed = get<uint64_t>(s); f.terms = get<bool>(s);
    // sizes that cannot be a ciphertext (a corrupted or foreign stream) stop here, not in an allocation
    if (!f.n || (f.n & (f.n - 1)) || f.n > (size_t(1) << 17) || f.limbs < 1 || f.limbs > 64 || f.size > max_ct_size) throw std::invalid_argument("the stream does not hold a ciphertext");
    return f;
}
} // namespace wire

class Plaintext { // src/plaintext.h: host coefficients (BFV/BGV: mod t; CKKS / NTT-form multiplyPlain: [limbs][N])
public:
    Plaintext() = default;
    explicit Plaintext(const std::vector<uint64_t> &coeffs) : data_(coeffs) {}
    // the hexadecimal polynomial form "7FFx^3 + 1x^1 + 3" (src/plaintext.h:126,246; src/plaintext_cuda.cuh:47-51,95-99)
    Plaintext(const std::string &hex_poly) { *this = hex_poly; }
    Plaintext &operator=(const std::string &hex_poly) {
        if (isNttForm()) throw std::logic_error("cannot set an NTT transformed Plaintext");
        // terms "<hex>[x^<dec>]" in strictly decreasing degree, joined by " + "; the constant term, if any, ends the string
        struct Term { size_t begin, digits, power; };
        std::vector<Term> terms;
        const size_t len = hex_poly.size();
        size_t at = 0, widest = 0;
        long previous = -1; // degree of the term before (none yet)
        auto hex_value = [](char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'A' && c <= 'F' ? c - 'A' + 10 : c >= 'a' && c <= 'f' ? c - 'a' + 10 : -1; };
        while (at < len) {
            Term t{at, 0, 0};
            while (at < len && hex_value(hex_poly[at]) >= 0) at++;
            t.digits = at - t.begin;
            if (!t.digits) throw std::invalid_argument("unable to parse hex_poly");
            size_t lead = t.begin; // significant bits of the coefficient: leading zero digits do not count
            while (lead < at && hex_value(hex_poly[lead]) == 0) lead++;
            if (lead < at) {
                const int top = hex_value(hex_poly[lead]);
                widest = std::max(widest, (at - lead - 1) * 4 + (top >= 8 ? 4 : top >= 4 ? 3 : top >= 2 ? 2 : 1));
            }
            if (at < len) { // "x^" and a decimal degree
                if (hex_poly[at] != 'x' || at + 1 >= len || hex_poly[at + 1] != '^') throw std::invalid_argument("unable to parse hex_poly");
                at += 2;
                while (at < len && hex_poly[at] >= '0' && hex_poly[at] <= '9') {
                    t.power = t.power * 10 + (size_t)(hex_poly[at++] - '0');
                    if (t.power >= (size_t(1) << 20)) throw std::invalid_argument("unable to parse hex_poly"); // no ring of this library is that large: refuse before allocating
                }
            }
            if (previous >= 0 && (long)t.power >= previous) throw std::invalid_argument("unable to parse hex_poly");
            previous = (long)t.power;
            terms.push_back(t);
            if (at < len) {
                if (hex_poly.compare(at, 3, " + ") != 0) throw std::invalid_argument("unable to parse hex_poly");
                at += 3;
            }
        }
        if (terms.empty() || !widest) { setZero(); return *this; } // "" and "0": the present coefficients are cleared, the count stays
        if (widest > 64) throw std::invalid_argument("hex_poly has too large coefficients");
        dev_.reset();
        data_.assign(terms[0].power + 1, 0);
        for (const Term &t : terms) {
            uint64_t v = 0;
            for (size_t i = t.begin; i < t.begin + t.digits; i++) v = (v << 4) | (uint64_t)hex_value(hex_poly[i]);
            data_[t.power] = v;
        }
        return *this;
    }
    // the constant polynomial (src/plaintext.h:255-262): one coefficient, coefficient form
    Plaintext &operator=(uint64_t const_coeff) {
        dev_.reset();
        data_.assign(1, const_coeff);
        parms_id_ = ParmsID();
        return *this;
    }
    // "7FFx^3 + 1x^1 + 3": upper-case hexadecimal coefficients, decreasing degree, zero terms left out, "0" for the zero polynomial (src/plaintext.h:491-498)
    std::string to_string() const {
        if (isNttForm()) throw std::invalid_argument("cannot convert NTT transformed plaintext to string");
        static const char digits[] = "0123456789ABCDEF";
        std::string out;
        for (size_t i = data_.size(); i-- > 0;) {
            if (!data_[i]) continue;
            if (!out.empty()) out += " + ";
            int shift = 60;
            while (shift && !((data_[i] >> shift) & 15)) shift -= 4;
            for (; shift >= 0; shift -= 4) out += digits[(data_[i] >> shift) & 15];
            if (i) out += "x^" + std::to_string(i);
        }
        return out.empty() ? std::string("0") : out;
    }
    bool isZero() const { return std::all_of(data_.begin(), data_.end(), [](uint64_t w) { return w == 0; }); } // src/plaintext.h:426-429
    size_t significantCoeffCount() const { size_t n = data_.size(); while (n && !data_[n - 1]) n--; return n; } // src/plaintext.h:450-457
    size_t nonzeroCoeffCount() const { return (size_t)std::count_if(data_.begin(), data_.end(), [](uint64_t w) { return w != 0; }); }
    // resize (src/plaintext_cuda.cuh:79-87): not for an NTT-form plaintext -- the library's own writers go through assignWords
    void resize(size_t n) {
        if (isNttForm()) throw std::logic_error("cannot reserve for an NTT transformed Plaintext");
        dev_.reset();
        data_.resize(n, 0);
    }
    void keepWords(size_t n) { dev_.reset(); data_.resize(n, 0); } // the first n words stay (an NTT-form plaintext drops its last limb)
    void assignWords(size_t n) { dev_.reset(); parms_id_ = ParmsID(); data_.assign(n, 0); } // n zero words, coefficient form, whatever the object held
    size_t coeffCount() const { return data_.size(); }
    // setZero / capacity / reserve / shrinkToFit / release (src/plaintext_cuda.cuh:73-165)
    void setZero(size_t start_coeff, size_t length) {
        if (!length) return;
        if (start_coeff + length - 1 >= data_.size()) throw std::out_of_range("length must be non-negative and start_coeff + length - 1 must be within [0, coeff_count)");
        dev_.reset();
        std::fill(data_.begin() + (std::ptrdiff_t)start_coeff, data_.begin() + (std::ptrdiff_t)(start_coeff + length), 0);
    }
    void setZero(size_t start_coeff) {
        if (start_coeff >= data_.size()) throw std::out_of_range("start_coeff must be within [0, coeff_count)");
        setZero(start_coeff, data_.size() - start_coeff);
    }
    void setZero() { dev_.reset(); std::fill(data_.begin(), data_.end(), 0); }
    size_t capacity() const noexcept { return data_.capacity(); }
    void reserve(size_t capacity) { data_.reserve(capacity); }
    void shrinkToFit() { data_.shrink_to_fit(); }
    void release() noexcept { dev_.reset(); std::vector<uint64_t>().swap(data_); parms_id_ = ParmsID(); scale_ = 1.0; }
    uint64_t *data() { dev_.reset(); return data_.data(); }
    const uint64_t *data() const { return data_.data(); }
    uint64_t &operator[](size_t i) { dev_.reset(); return data_[i]; }
    const uint64_t &operator[](size_t i) const { return data_[i]; }
    // the same polynomial (leading zero coefficients ignored), the same form and level, scales that agree to rounding (src/plaintext.h:396-410)
    bool operator==(const Plaintext &o) const {
        if (isNttForm() != o.isNttForm() || (isNttForm() && parms_id_ != o.parms_id_)) return false;
        const size_t n = significantCoeffCount();
        if (n != o.significantCoeffCount() || !std::equal(data_.begin(), data_.begin() + (std::ptrdiff_t)n, o.data_.begin())) return false;
        const double scale_max = std::max(std::max(std::fabs(scale_), std::fabs(o.scale_)), 1.0); // util::areClose (src/utils/common.h)
        return std::fabs(scale_ - o.scale_) < std::numeric_limits<double>::epsilon() * scale_max;
    }
    bool operator!=(const Plaintext &o) const { return !(*this == o); }
    double &scale() { return scale_; }
    double scale() const { return scale_; }
    bool isNttForm() const { return parms_id_.limbs != 0; }
    const ParmsID &parmsID() const { return parms_id_; } // level of an NTT-form plaintext (parmsIDZero = coefficient form)
    void setNttForm(const ParmsID &id) { parms_id_ = id; }
    // PlaintextCuda keeps its coefficients in device memory; here the device copy is a mirror of the host vector, made on first use by
    // an Evaluator call and dropped by any mutable access, so a weight plaintext that is multiplied many times is uploaded once
    const uint64_t *device() const {
        if (!dev_) {
            dev_ = std::make_shared<DeviceArray>(data_.size());
            if (!data_.empty()) check(troyhip_copy_h2d(dev_->get(), data_.data(), data_.size() * 8, nullptr));
        }
        return dev_->get();
    }
    void adoptDevice(std::shared_ptr<DeviceArray> d) { dev_ = std::move(d); } // the encoder hands over the buffer it transformed
    // PlaintextCuda::save / load (src/plaintext_cuda.cu:7-27): parms_id (32 bytes), coeff_count, scale, word count, words
    void save(std::ostream &stream) const {
        stream.write(reinterpret_cast<const char *>(parms_id_.data()), 32);
        const size_t count = data_.size();
        stream.write(reinterpret_cast<const char *>(&count), sizeof(size_t));
        stream.write(reinterpret_cast<const char *>(&scale_), sizeof(double));
        stream.write(reinterpret_cast<const char *>(&count), sizeof(size_t));
        stream.write(reinterpret_cast<const char *>(data_.data()), (std::streamsize)(count * 8));
    }
    // a coefficient-form plaintext needs nothing else; an NTT-form one carries only the HASH of its level, which the overload
    // with the context resolves
    void load(std::istream &stream) { read(stream, nullptr); }
    void load(std::istream &stream, const SEALContext &context) {
        auto ids = context.levelIDs();
        read(stream, [&](const uint64_t *hash) {
            for (const ParmsID &id : *ids)
                if (id.limbs && std::equal(hash, hash + 4, id.data())) return id.limbs;
            throw std::invalid_argument("plain is not valid for encryption parameters");
            return 0;
        });
    }
private:
    void read(std::istream &stream, const std::function<int(const uint64_t *)> &limbs_of) {
        ParmsID id;
        size_t count = 0, words = 0;
        double scale = 1.0;
        stream.read(reinterpret_cast<char *>(id.data()), 32);
        stream.read(reinterpret_cast<char *>(&count), sizeof(size_t));
        stream.read(reinterpret_cast<char *>(&scale), sizeof(double));
        stream.read(reinterpret_cast<char *>(&words), sizeof(size_t));
        if (!stream || words > (size_t(1) << 23)) throw std::invalid_argument("stream ended inside a plaintext");
        std::vector<uint64_t> host(words);
        stream.read(reinterpret_cast<char *>(host.data()), (std::streamsize)(words * 8));
        if (!stream) throw std::invalid_argument("stream ended inside a plaintext");
        if (id != parmsIDZero) {
            if (!limbs_of) throw std::invalid_argument("an NTT-form plaintext is loaded with its context: load(stream, context)");
            id.limbs = limbs_of(id.data());
        }
        dev_.reset();
        data_ = std::move(host);
        scale_ = scale;
        parms_id_ = id;
    }
    std::vector<uint64_t> data_;
    double scale_ = 1.0;
    ParmsID parms_id_;
    mutable std::shared_ptr<DeviceArray> dev_;
};

class Ciphertext { // src/ciphertext_cuda.cuh:12-268
public:
    Ciphertext() = default;
    explicit Ciphertext(const SEALContext &c) : n_(c.polyModulusDegree()), ids_(c.levelIDs()) {}
    size_t size() const { return (size_t)d_.size; }
    size_t coeffModulusSize() const { return (size_t)d_.limbs; }
    size_t polyModulusDegree() const { return n_; }
    // the 256-bit id of the ciphertext's level (parmsIDZero before the object has met a context)
    const ParmsID &parmsID() const { return ids_ && (size_t)d_.limbs < ids_->size() ? (*ids_)[(size_t)d_.limbs] : parmsIDZero; }
    void bind(const SEALContext &c) { ids_ = c.levelIDs(); }
    void bind(const Ciphertext &o) { ids_ = o.ids_; }
    bool isNttForm() const { return d_.is_ntt_form != 0; }
    bool &isNttFormRef() { ntt_shadow_ = d_.is_ntt_form != 0; return ntt_shadow_; }
    double &scale() { return d_.scale; }
    double scale() const { return d_.scale; }
    uint64_t &correctionFactor() { return d_.correction_factor; }
    uint64_t correctionFactor() const { return d_.correction_factor; }
    // a ciphertext that owns its storage keeps capacity max(size, 3) polynomials so multiply / relinearize run in place
    void resize(size_t n, size_t limbs, size_t size) {
        n_ = n;
        seed_ = 0;
        const size_t words = std::max<size_t>(size, 3) * limbs * n;
        if (!store_ || !own_ || store_->size() != words) {
            auto fresh = std::make_shared<DeviceArray>(words);
            if (store_ && d_.data && d_.batch_stride) check(troyhip_copy_d2d(fresh->get(), d_.data, std::min<size_t>(words, d_.batch_stride) * 8, nullptr));
            store_ = std::move(fresh);
            own_ = true;
        }
        d_.data = store_->get();
        d_.batch_stride = words;
        d_.size = (int)size;
        d_.limbs = (int)limbs;
    }
    // reserve / sizeCapacity / release / isTransparent (src/ciphertext_cuda.cuh:53, 141-180; src/ciphertext.h:438-443).  A ciphertext that owns
    // its storage always has room for max(size, 3) polynomials (see resize); reserve grows that
    size_t sizeCapacity() const noexcept { return d_.limbs && n_ ? (size_t)d_.batch_stride / ((size_t)d_.limbs * n_) : 0; }
    void reserve(size_t size_capacity) {
        if (size_capacity < 2) throw std::invalid_argument("invalid size_capacity");
        if (!d_.limbs || !n_ || size_capacity <= sizeCapacity()) return;
        const size_t keep = size();
        auto fresh = std::make_shared<DeviceArray>(size_capacity * (size_t)d_.limbs * n_);
        if (d_.data) check(troyhip_copy_d2d(fresh->get(), d_.data, keep * (size_t)d_.limbs * n_ * 8, nullptr));
        store_ = std::move(fresh);
        own_ = true;
        d_.data = store_->get();
        d_.batch_stride = size_capacity * (size_t)d_.limbs * n_;
    }
    void release() noexcept { store_.reset(); own_ = true; seed_ = 0; d_ = troyhip_ct{nullptr, 0, 0, 0, 0, 1.0, 1}; }
    bool isTransparent() const {
        if (!d_.data || size() < 2) return true;
        const std::vector<uint64_t> h = toHost();
        return std::all_of(h.begin() + (std::ptrdiff_t)((size_t)d_.limbs * n_), h.end(), [](uint64_t w) { return w == 0; });
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
    // ---- batch slabs (no counterpart in the reference, whose every CiphertextCuda is its own allocation): `count` ciphertexts of
    // one shape carved out of ONE device allocation, dense [count][size][limbs][N] -- the layout the library's batched entry points
    // take -- each still an ordinary Ciphertext.  A member that has to grow, or is copied, moves to storage of its own.
    static std::vector<Ciphertext> allocateBatch(size_t count, const Ciphertext &like) {
        const size_t stride = like.size() * like.coeffModulusSize() * like.n_;
        auto slab = std::make_shared<DeviceArray>(count * stride);
        std::vector<Ciphertext> out(count);
        for (size_t b = 0; b < count; b++) {
            Ciphertext &c = out[b];
            c.store_ = slab;
            c.own_ = false;
            c.d_ = like.d_;
            c.d_.data = slab->get() + b * stride;
            c.d_.batch_stride = stride;
            c.n_ = like.n_;
            c.ids_ = like.ids_;
        }
        return out;
    }
    // ... of `size` polynomials over `limbs` primes each (a product, the next level), the rest of the metadata from `like`
    static std::vector<Ciphertext> allocateBatch(size_t count, const Ciphertext &like, size_t size, size_t limbs) {
        Ciphertext shape;
        shape.d_ = like.d_;
        shape.d_.size = (int)size;
        shape.d_.limbs = (int)limbs;
        shape.n_ = like.n_;
        shape.ids_ = like.ids_;
        return allocateBatch(count, shape);
    }
    // the same ciphertexts, copied into one slab (device-to-device); they must agree in shape and metadata
    static std::vector<Ciphertext> packBatch(const std::vector<const Ciphertext *> &items) {
        if (items.empty()) return {};
        std::vector<Ciphertext> out = allocateBatch(items.size(), *items[0]);
        for (size_t b = 0; b < items.size(); b++) {
            if (!items[b]->sameShape(*items[0])) throw std::invalid_argument("packBatch: ciphertexts of different shape");
            check(troyhip_copy_d2d(out[b].d_.data, items[b]->d_.data, out[b].d_.batch_stride * 8, nullptr));
        }
        return out;
    }
    static std::vector<Ciphertext> packBatch(const std::vector<Ciphertext> &items) { return packBatch(pointers(items)); }
    static std::vector<const Ciphertext *> pointers(const std::vector<Ciphertext> &items) {
        std::vector<const Ciphertext *> out;
        out.reserve(items.size());
        for (const Ciphertext &c : items) out.push_back(&c);
        return out;
    }
    static std::vector<Ciphertext *> pointers(std::vector<Ciphertext> &items) {
        std::vector<Ciphertext *> out;
        out.reserve(items.size());
        for (Ciphertext &c : items) out.push_back(&c);
        return out;
    }
    // do these ciphertexts, in this order, form a dense run of one slab?
    static bool isBatch(const std::vector<const Ciphertext *> &items) {
        if (items.empty() || !items[0]->store_ || items[0]->own_) return false;
        const Ciphertext &head = *items[0];
        if (head.d_.batch_stride != head.size() * head.coeffModulusSize() * head.n_) return false;
        for (size_t b = 1; b < items.size(); b++) {
            const Ciphertext &c = *items[b];
            if (c.store_ != head.store_ || c.own_ || !c.sameShape(head) || c.d_.data != head.d_.data + b * head.d_.batch_stride) return false;
        }
        return true;
    }
    // ... or at least consecutive members of one slab with one common stride?  (A member may use fewer polynomials than its stride holds: what
    // relinearizeInplaceBatch leaves of a batch of products.)  That is all a batched library call needs: (data, batch_stride) of the head.
    static bool isRun(const std::vector<const Ciphertext *> &items) {
        if (items.empty() || !items[0]->store_ || items[0]->own_) return false;
        const Ciphertext &head = *items[0];
        if (head.d_.batch_stride < head.size() * head.coeffModulusSize() * head.n_) return false;
        for (size_t b = 1; b < items.size(); b++) {
            const Ciphertext &c = *items[b];
            if (c.store_ != head.store_ || c.own_ || !c.sameShape(head) || c.d_.batch_stride != head.d_.batch_stride || c.d_.data != head.d_.data + b * head.d_.batch_stride) return false;
        }
        return true;
    }
    bool sameShape(const Ciphertext &o) const {
        return n_ == o.n_ && d_.size == o.d_.size && d_.limbs == o.d_.limbs && d_.is_ntt_form == o.d_.is_ntt_form && d_.scale == o.d_.scale &&
               d_.correction_factor == o.d_.correction_factor;
    }
    void copyMeta(const troyhip_ct &m) { d_.size = m.size; d_.limbs = m.limbs; d_.is_ntt_form = m.is_ntt_form; d_.scale = m.scale; d_.correction_factor = m.correction_factor; }
    troyhip_ct *raw() { seed_ = 0; return &d_; } // whoever takes the mutable view may rewrite the polynomials: c1 is no longer the seed's expansion
    const troyhip_ct *raw() const { return &d_; }
    // seed() (src/ciphertext_cuda.cuh:189-190): non-zero for a fresh symmetric encryption, whose c1 is the expansion of this 64-bit seed -- save() then
    // writes c0 alone and load(stream, context) regenerates c1.  The reference never clears it (an Evaluator op on such a ciphertext followed by
    // save() would drop the modified c1); here every mutable access clears it.
    uint64_t seed() const noexcept { return seed_; }
    uint64_t &seed() noexcept { return seed_; }
    // wire format of CiphertextCuda::save / load / saveTerms / loadTerms (src/ciphertext_cuda.cu:16-143); the context supplies
    // the 256-bit parms_id the reference object carries itself (defined after Evaluator below)
    inline void save(std::ostream &stream, const SEALContext &context) const;
    inline void load(std::istream &stream, const SEALContext &context);
    inline void saveTerms(std::ostream &stream, const SEALContext &context, const class Evaluator &evaluator, const std::vector<size_t> &termIds) const;
    inline void loadTerms(std::istream &stream, const SEALContext &context, const class Evaluator &evaluator, const std::vector<size_t> &termIds);
    // the reference's own signatures (src/ciphertext_cuda.cuh:183-187): the context is the one the object is bound to (save), none
    // (load(stream): fields taken as they come, as there), or the evaluator's (saveTerms / loadTerms)
    inline void save(std::ostream &stream) const;
    inline void load(std::istream &stream);
    inline void saveTerms(std::ostream &stream, const class Evaluator &evaluator, const std::vector<size_t> &termIds) const;
    inline void loadTerms(std::istream &stream, const class Evaluator &evaluator, const std::vector<size_t> &termIds);
    // value semantics: a copy gets storage of its own (also when the source is a slab member)
    Ciphertext(const Ciphertext &o) : d_(o.d_), n_(o.n_), ids_(o.ids_) { clone(o); seed_ = o.seed_; }
    Ciphertext(Ciphertext &&o) noexcept = default;
    Ciphertext &operator=(const Ciphertext &o) {
        if (this != &o) { d_ = o.d_; n_ = o.n_; ids_ = o.ids_; clone(o); seed_ = o.seed_; }
        return *this;
    }
    Ciphertext &operator=(Ciphertext &&o) noexcept = default;
private:
    void clone(const Ciphertext &o) {
        store_.reset();
        own_ = true;
        d_.data = nullptr;
        if (!o.store_) return;
        const size_t used = o.size() * o.coeffModulusSize() * o.n_, words = std::max<size_t>(o.size(), 3) * o.coeffModulusSize() * o.n_;
        store_ = std::make_shared<DeviceArray>(words);
        d_.data = store_->get();
        d_.batch_stride = words;
        if (used) check(troyhip_copy_d2d(d_.data, o.d_.data, used * 8, nullptr));
    }
    std::shared_ptr<DeviceArray> store_;
    bool own_ = true;
    troyhip_ct d_{nullptr, 0, 0, 0, 0, 1.0, 1};
    size_t n_ = 0;
    bool ntt_shadow_ = false;
    uint64_t seed_ = 0;
    std::shared_ptr<const std::vector<ParmsID>> ids_;
};

// SecretKeyCuda (src/secretkey_cuda.cuh): [K][N] in NTT form at the key level, kept on the host (key generation and encryption run there).
// save / load (:292-297) is the reference's plaintext format -- parms_id, coeff_count, scale, word count, words (src/plaintext_cuda.cu:7-27)
class SecretKey {
public:
    std::vector<uint64_t> data;
    ParmsID parms_id; // the key level (stamped by KeyGenerator; read back by load)
    const ParmsID &parmsID() const noexcept { return parms_id; }
    ParmsID &parmsID() noexcept { return parms_id; }
    void save(std::ostream &stream) const {
        stream.write(reinterpret_cast<const char *>(parms_id.data()), 32);
        wire::put<size_t>(stream, data.size());
        wire::put<double>(stream, 1.0);
        wire::put<size_t>(stream, data.size());
        wire::put_words(stream, data.data(), data.size());
    }
    void load(std::istream &stream) {
        ParmsID id;
        stream.read(reinterpret_cast<char *>(id.data()), 32);
        const size_t count = wire::get<size_t>(stream);
        (void)wire::get<double>(stream);
        const size_t words = wire::get<size_t>(stream);
        if (words != count || words > (size_t(1) << 23)) throw std::invalid_argument("the stream does not hold a secret key"); // at most 64 limbs of 2^17 coefficients
        data = wire::get_vector(stream, words);
        parms_id = id; // the limb count behind the hash comes back when the key meets its context (limbs stays 0 until then)
    }
};
// PublicKeyCuda (src/publickey_cuda.cuh): a size-2 ciphertext [2][K][N] in NTT form at the key level (host).  save / load (:252-257) is the
// reference's ciphertext format (src/ciphertext_cuda.cu:16-43)
class PublicKey {
public:
    std::vector<uint64_t> data;
    ParmsID parms_id;
    size_t poly_modulus_degree = 0, coeff_modulus_size = 0;
    const ParmsID &parmsID() const noexcept { return parms_id; }
    ParmsID &parmsID() noexcept { return parms_id; }
    void save(std::ostream &stream) const {
        if (data.empty() || !poly_modulus_degree) throw std::logic_error("the public key has not been generated");
        wire::CtFields f{{parms_id[0], parms_id[1], parms_id[2], parms_id[3]}, true, 2, poly_modulus_degree, coeff_modulus_size, 1.0, 1, 0, false};
        wire::put_fields(stream, f);
        wire::put<size_t>(stream, data.size());
        wire::put_words(stream, data.data(), data.size());
    }
    void load(std::istream &stream) {
        const wire::CtFields f = wire::get_fields(stream);
        if (f.terms) throw std::invalid_argument("Trying to load a termed ciphertext, but indices is not specified");
        if (f.seed) throw std::invalid_argument("seed is not zero.");
        const size_t words = wire::get<size_t>(stream);
        if (f.size != 2 || words != 2 * f.limbs * f.n) throw std::invalid_argument("the stream does not hold a public key");
        data = wire::get_vector(stream, words);
        std::copy(f.id, f.id + 4, parms_id.begin());
        parms_id.limbs = (int)f.limbs;
        poly_modulus_degree = f.n;
        coeff_modulus_size = f.limbs;
    }
};

class KSwitchKeys { // src/kswitchkeys_cuda.cuh:43-56: data()[index] on the device, uploaded from the host key
public:
    bool hasKeyIndex(size_t index) const { return keys_.count(index) != 0; }
    const uint64_t *device(size_t index) const { return keys_.at(index)->get(); }
    // the key an evaluator hands to the key-switch kernels: it must have been generated (or saved) under the SAME context -- same key-level
    // parms_id, [K - 1][2][K][N] words -- or the kernels would read past it (a loaded key can have any shape)
    const uint64_t *device(size_t index, const ParmsID &key_parms_id, size_t key_limbs, size_t poly_modulus_degree) const {
        const auto &a = keys_.at(index);
        if (parms_id_ != key_parms_id || key_limbs < 2 || a->size() != (key_limbs - 1) * 2 * key_limbs * poly_modulus_degree)
            throw std::invalid_argument("kswitch_keys is not valid for encryption parameters");
        return a->get();
    }
    void upload(size_t index, const std::vector<uint64_t> &host) {
        auto a = std::make_shared<DeviceArray>(host.size());
        check(troyhip_copy_h2d(a->get(), host.data(), host.size() * 8, nullptr));
        keys_[index] = a;
    }
    const std::map<size_t, std::shared_ptr<DeviceArray>> &all() const { return keys_; }
    void clear() { keys_.clear(); } // src/kswitchkeys_cuda.cuh
    // the same keys on another device (keys are replicated over the GPUs of a batch shard: BASELINE north_star).  Called with `to_device` current.
    template <class K> static K replicate(const K &k, int from_device, int to_device) {
        K out;
        out.describe(k.parms_id_, k.n_, k.limbs_);
        for (const auto &kv : k.keys_) {
            auto a = std::make_shared<DeviceArray>(kv.second->size());
            check(troyhip_copy_peer(a->get(), to_device, kv.second->get(), from_device, kv.second->size() * 8, nullptr));
            out.keys_[kv.first] = a;
        }
        check(troyhip_stream_synchronize(nullptr));
        return out;
    }
    // the key level and shape (stamped by KeyGenerator, read back by load): one key is [K - 1][2][K][N]
    void describe(const ParmsID &key_parms_id, size_t poly_modulus_degree, size_t key_limbs) { parms_id_ = key_parms_id; n_ = poly_modulus_degree; limbs_ = key_limbs; }
    const ParmsID &parmsID() const noexcept { return parms_id_; }
    ParmsID &parmsID() noexcept { return parms_id_; }
    size_t size() const noexcept { return keys_.size(); } // how many key indices hold a key (src/kswitchkeys.h: size())
    // save / load (src/kswitchkeys_cuda.cuh:330-356): parms_id, the length of the index vector, and per index the digit count followed by one
    // public-key-format ciphertext [2][K][N] per digit (an index without a key has digit count 0)
    void save(std::ostream &stream) const {
        stream.write(reinterpret_cast<const char *>(parms_id_.data()), 32);
        const size_t slots = keys_.empty() ? 0 : keys_.rbegin()->first + 1;
        wire::put<size_t>(stream, slots);
        if (slots && (!n_ || limbs_ < 2)) throw std::logic_error("the key-switching keys have no shape: generate them with KeyGenerator or load them");
        const size_t digits = limbs_ - 1, digit_words = 2 * limbs_ * n_;
        std::vector<uint64_t> host(digits * digit_words);
        for (size_t i = 0; i < slots; i++) {
            auto it = keys_.find(i);
            if (it == keys_.end()) { wire::put<size_t>(stream, 0); continue; }
            if (it->second->size() != host.size()) throw std::logic_error("a key-switching key of unexpected size");
            check(troyhip_copy_d2h(host.data(), it->second->get(), host.size() * 8, nullptr));
            wire::put<size_t>(stream, digits);
            for (size_t j = 0; j < digits; j++) {
                wire::CtFields f{{parms_id_[0], parms_id_[1], parms_id_[2], parms_id_[3]}, true, 2, n_, limbs_, 1.0, 1, 0, false};
                wire::put_fields(stream, f);
                wire::put<size_t>(stream, digit_words);
                wire::put_words(stream, host.data() + j * digit_words, digit_words);
            }
        }
    }
    void load(std::istream &stream) {
        ParmsID id;
        stream.read(reinterpret_cast<char *>(id.data()), 32);
        const size_t slots = wire::get<size_t>(stream);
        if (slots > (size_t(1) << 18)) throw std::invalid_argument("the stream does not hold key-switching keys");
        std::map<size_t, std::shared_ptr<DeviceArray>> fresh;
        size_t n = 0, limbs = 0;
        for (size_t i = 0; i < slots; i++) {
            const size_t digits = wire::get<size_t>(stream);
            if (!digits) continue;
            if (digits > 63) throw std::invalid_argument("the stream does not hold key-switching keys"); // one digit per data prime: at most 64 limbs
            std::vector<uint64_t> host;
            for (size_t j = 0; j < digits; j++) {
                const wire::CtFields f = wire::get_fields(stream);
                const size_t words = wire::get<size_t>(stream);
                if (f.terms || f.seed || f.size != 2 || f.limbs != digits + 1 || words != 2 * f.limbs * f.n || (n && (f.n != n || f.limbs != limbs)))
                    throw std::invalid_argument("the stream does not hold key-switching keys");
                n = f.n; limbs = f.limbs;
                const std::vector<uint64_t> digit = wire::get_vector(stream, words); // (grows with the stream, not with the header)
                host.insert(host.end(), digit.begin(), digit.end());
            }
            auto a = std::make_shared<DeviceArray>(host.size());
            check(troyhip_copy_h2d(a->get(), host.data(), host.size() * 8, nullptr));
            fresh[i] = a;
        }
        keys_ = std::move(fresh);
        parms_id_ = id;
        parms_id_.limbs = (int)limbs;
        n_ = n; limbs_ = limbs;
    }
protected:
    std::map<size_t, std::shared_ptr<DeviceArray>> keys_;
    ParmsID parms_id_;
    size_t n_ = 0, limbs_ = 0;
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

// LWECiphertextCuda (src/lwe_cuda.cuh): the LWE sample (c1: one polynomial [limbs][N] in coefficient form, c0: one word per limb)
// that extractLWE cuts out of an RLWE ciphertext
class LWECiphertext {
public:
    const ParmsID &parmsID() const { return parms_id_; }
    size_t polyModulusDegree() const { return n_; }
    size_t coeffModulusSize() const { return limbs_; }
    double scale() const { return scale_; }
    uint64_t correctionFactor() const { return cf_; }
    const DeviceArray &c1() const { return c1_; }
    const std::vector<uint64_t> &c0() const { return c0_; }
private:
    friend class Evaluator;
    ParmsID parms_id_;
    size_t n_ = 0, limbs_ = 0;
    double scale_ = 1.0;
    uint64_t cf_ = 1;
    DeviceArray c1_;
    std::vector<uint64_t> c0_;
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
        rlk.describe(c_.keyParmsID(), c_.polyModulusDegree(), c_.keyLimbs());
        rlk.upload(RelinKeys::getIndex(2), h);
    }
    RelinKeys createRelinKeys() const { RelinKeys r; createRelinKeys(r); return r; }
    void createGaloisKeys(const std::vector<uint32_t> &galois_elts, GaloisKeys &gk) const {
        gk.describe(c_.keyParmsID(), c_.polyModulusDegree(), c_.keyLimbs());
        for (uint32_t e : galois_elts) {
            std::vector<uint64_t> h(ksk_words());
            check(troyhip_host_galois_key(c_.handle(), lo_, hi_, sk_.data.data(), e, h.data()));
            gk.upload(GaloisKeys::getIndex(e), h);
        }
    }
    // every key rotate / conjugate can ask for: X -> X^(2N-1) and X -> X^(3^(2^i)), X^(3^-(2^i)) (GaloisTool::getEltsAll,
    // src/utils/galois.cpp:101-126)
    void createGaloisKeys(GaloisKeys &gk) const {
        const uint64_t m = 2 * (uint64_t)c_.polyModulusDegree();
        std::vector<uint32_t> elts{(uint32_t)(m - 1)};
        uint64_t pos = 3, neg = 1;
        while (neg * 3 % m != 1) neg += 2; // 3^-1 mod 2N
        for (uint64_t span = 2; span < m / 2; span <<= 1) {
            elts.push_back((uint32_t)pos);
            elts.push_back((uint32_t)neg);
            pos = pos * pos % m;
            neg = neg * neg % m;
        }
        createGaloisKeys(elts, gk);
    }
    GaloisKeys createGaloisKeys() const { GaloisKeys g; createGaloisKeys(g); return g; }
    // createKeySwitchingKeys (src/keygenerator.cpp:360-366): ONE key, which takes a ciphertext under `new_key` to one under this generator's
    // secret key (Evaluator::applyKeySwitchingInplace)
    KSwitchKeys createKeySwitchingKeys(const SecretKey &new_key) const {
        if (new_key.data.size() != sk_.data.size()) throw std::invalid_argument("new_key is not valid for encryption parameters");
        std::vector<uint64_t> h(ksk_words());
        check(troyhip_host_kswitch_key(c_.handle(), lo_, hi_, sk_.data.data(), new_key.data.data(), h.data()));
        KSwitchKeys k;
        k.describe(c_.keyParmsID(), c_.polyModulusDegree(), c_.keyLimbs());
        k.upload(0, h);
        return k;
    }
    // the keys of fieldTraceInplace / packLWECiphertexts: X -> X^(N/2^k + 1), k = 0 .. log2(N) - 1 (src/keygenerator.cpp:350-358)
    GaloisKeys createAutomorphismKeys() const {
        std::vector<uint32_t> elts;
        for (size_t n = c_.polyModulusDegree(); n >= 2; n >>= 1) elts.push_back((uint32_t)(n + 1));
        GaloisKeys g;
        createGaloisKeys(elts, g);
        return g;
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
        sk_.parms_id = pk_.parms_id = c_.keyParmsID();
        pk_.poly_modulus_degree = N;
        pk_.coeff_modulus_size = K;
    }
    size_t ksk_words() const { const size_t K = c_.keyLimbs(); return (K - 1) * 2 * K * c_.polyModulusDegree(); }
    const SEALContext &c_;
    uint64_t lo_, hi_;
    SecretKey sk_;
    PublicKey pk_;
};

class Encryptor { // src/encryptor_cuda.cuh:20-300, CPU sampling + upload
public:
    // every encryption draws a fresh 128-bit seed for its samples from the operating system (src/randomgen.cpp:23,72)
    // keys of another context (a loaded key can have any shape) are refused here, as the reference's CPU classes refuse them (isValidFor: src/decryptor.cpp:62-65
    // "secret key is not valid for encryption parameters", src/encryptor.h:92,107 setPublicKey / setSecretKey): the host encryption reads 2 K N / K N words of them
    Encryptor(const SEALContext &c, const PublicKey &pk) : c_(c), pk_(valid(c, pk)) {}
    Encryptor(const SEALContext &c, const SecretKey &sk) : c_(c), sk_(valid(c, sk)) {}
    Encryptor(const SEALContext &c, const PublicKey &pk, const SecretKey &sk) : c_(c), pk_(valid(c, pk)), sk_(valid(c, sk)) {}
    // deterministic stream (seed, call counter) for tests ONLY
    Encryptor(const SEALContext &c, const PublicKey &pk, uint64_t seed_lo, uint64_t seed_hi = 0) : c_(c), pk_(valid(c, pk)), seeded_(true), lo_(seed_lo), hi_(seed_hi) {}
    void setPublicKey(const PublicKey &pk) { pk_ = valid(c_, pk); }
    void setSecretKey(const SecretKey &sk) { sk_ = valid(c_, sk); }
    static const PublicKey &valid(const SEALContext &c, const PublicKey &pk) {
        if (pk.parms_id != c.keyParmsID() || pk.data.size() != 2 * c.keyLimbs() * c.polyModulusDegree()) throw std::invalid_argument("public key is not valid for encryption parameters");
        return pk;
    }
    static const SecretKey &valid(const SEALContext &c, const SecretKey &sk) {
        if (sk.parms_id != c.keyParmsID() || sk.data.size() != c.keyLimbs() * c.polyModulusDegree()) throw std::invalid_argument("secret key is not valid for encryption parameters");
        return sk;
    }
    // The CPU reference refuses without a public key (src/encryptor.cpp:157-160); EncryptorCuda has no such check and its own caller
    // test/evaluator_cuda.cu:2566-2569 (BFVKeySwitching) encrypts through an Encryptor that was given ONLY a secret key -- meaning "a
    // ciphertext under that key".  An encryptor that holds just a secret key therefore encrypts symmetrically; one that holds neither throws.
    void encrypt(const Plaintext &plain, Ciphertext &dst) const {
        if (pk_.data.empty()) {
            if (sk_.data.empty()) throw std::logic_error("public key is not set");
            run(troyhip_host_encrypt_symmetric, sk_.data, plain, dst);
            return;
        }
        run(troyhip_host_encrypt, pk_.data, plain, dst);
    }
    Ciphertext encrypt(const Plaintext &plain) const { Ciphertext d; encrypt(plain, d); return d; }
    // encryptSymmetric (src/encryptor_cuda.cuh:259-290): (-(a s + e) + m, a) at the plaintext's own level
    // The ciphertext carries the seed its c1 was expanded from (dst.seed() != 0: src/utils/rlwe_cuda.cu:292-303), so save() writes half of it.
    void encryptSymmetric(const Plaintext &plain, Ciphertext &dst) const {
        if (sk_.data.empty()) throw std::logic_error("secret key is not set"); // src/encryptor.cpp:164-167
        const uint64_t a_seed = fresh_a_seed(counter_ + 1); // run() below consumes call number counter_ + 1: the seed belongs to THAT call
        run([a_seed](const troyhip_context *c, uint64_t lo, uint64_t hi, const uint64_t *key, const uint64_t *pl, uint64_t count, int limbs, uint64_t *out) {
            return troyhip_host_encrypt_symmetric_seeded(c, lo, hi, a_seed, key, pl, count, limbs, out);
        }, sk_.data, plain, dst);
        dst.seed() = a_seed;
    }
    Ciphertext encryptSymmetric(const Plaintext &plain) const { Ciphertext d; encryptSymmetric(plain, d); return d; }
    // encryptZero / encryptZeroSymmetric (src/encryptor_cuda.cuh:170-237, 292-320; src/encryptor.cpp:88-150): zero at the first data level or at
    // the level `parms_id` names -- the asymmetric form encrypts one level up and divides by the extra prime, as the reference does
    void encryptZero(const ParmsID &parms_id, Ciphertext &dst) const {
        if (pk_.data.empty()) throw std::logic_error("public key is not set");
        zero(pk_.data, 0, parms_id, dst);
    }
    void encryptZero(Ciphertext &dst) const { encryptZero(c_.firstParmsID(), dst); }
    Ciphertext encryptZero(const ParmsID &parms_id) const { Ciphertext d; encryptZero(parms_id, d); return d; }
    Ciphertext encryptZero() const { return encryptZero(c_.firstParmsID()); }
    void encryptZeroSymmetric(const ParmsID &parms_id, Ciphertext &dst) const {
        if (sk_.data.empty()) throw std::logic_error("secret key is not set");
        zero(sk_.data, 1, parms_id, dst);
    }
    // a public 64-bit seed for c1, never zero (zero means "not seeded" on the wire): from the deterministic stream of a seeded encryptor, else fresh
    // `call` = the value of the call counter the calling encryption consumes (each call its own: two encryptions never share c1)
    uint64_t fresh_a_seed(uint64_t call) const {
        uint64_t a = 0;
        if (seeded_) a = (lo_ ^ 0x9E3779B97F4A7C15ull) + 0xD1B54A32D192ED03ull * call;
        else check(troyhip_random_bytes(&a, sizeof(a)));
        return a ? a : 1;
    }
    void encryptZeroSymmetric(Ciphertext &dst) const { encryptZeroSymmetric(c_.firstParmsID(), dst); }
    Ciphertext encryptZeroSymmetric(const ParmsID &parms_id) const { Ciphertext d; encryptZeroSymmetric(parms_id, d); return d; }
    Ciphertext encryptZeroSymmetric() const { return encryptZeroSymmetric(c_.firstParmsID()); }
private:
    void zero(const std::vector<uint64_t> &key, int symmetric, const ParmsID &id, Ciphertext &dst) const {
        if (!c_.getContextData(id) || id.limbs > (int)c_.firstLimbs()) throw std::invalid_argument("parms_id is not valid for encryption parameters");
        const size_t N = c_.polyModulusDegree();
        const bool ckks = c_.parms().scheme() == SchemeType::ckks;
        std::vector<uint64_t> h((size_t)2 * id.limbs * N);
        uint64_t s[2] = {lo_ + (++counter_), hi_};
        if (!seeded_) check(troyhip_random_bytes(s, sizeof(s)));
        const uint64_t a_seed = symmetric ? fresh_a_seed(counter_) : 0;
        if (symmetric) check(troyhip_host_encrypt_symmetric_seeded(c_.handle(), s[0], s[1], a_seed, key.data(), nullptr, 0, id.limbs, h.data()));
        else check(troyhip_host_encrypt_zero(c_.handle(), s[0], s[1], key.data(), 0, id.limbs, h.data()));
        dst.fromHost(h, N, (size_t)id.limbs, 2, ckks, 1.0, 1);
        dst.bind(c_);
        dst.seed() = a_seed;
    }
    template <class F> void run(F fn, const std::vector<uint64_t> &key, const Plaintext &plain, Ciphertext &dst) const {
        const size_t N = c_.polyModulusDegree();
        const bool ckks = c_.parms().scheme() == SchemeType::ckks;
        if (ckks && !plain.isNttForm()) throw std::invalid_argument("plain must be in NTT form");  // src/encryptor.cpp:216-219
        if (!ckks && plain.isNttForm()) throw std::invalid_argument("plain cannot be in NTT form"); // src/encryptor.cpp:202-205
        const int limbs = ckks ? (int)(plain.coeffCount() / N) : (int)c_.firstLimbs();
        std::vector<uint64_t> h((size_t)2 * limbs * N);
        uint64_t s[2] = {lo_ + (++counter_), hi_};
        if (!seeded_) check(troyhip_random_bytes(s, sizeof(s)));
        check(fn(c_.handle(), s[0], s[1], key.data(), plain.data(), ckks ? N : plain.coeffCount(), limbs, h.data()));
        dst.fromHost(h, N, limbs, 2, ckks, ckks ? plain.scale() : 1.0, 1); // destination.scale() = plain.scale(): src/encryptor.cpp:235
        dst.bind(c_);
    }
    const SEALContext &c_;
    PublicKey pk_;
    SecretKey sk_;
    bool seeded_ = false;
    uint64_t lo_ = 0, hi_ = 0;
    mutable uint64_t counter_ = 0;
};

class Decryptor { // src/decryptor_cuda.cuh:13-60: the secret key is uploaded once, decryption runs on the device
public:
    Decryptor(const SEALContext &c, const SecretKey &sk) : c_(c), sk_(Encryptor::valid(c, sk).data.size()) { // src/decryptor.cpp:62-65: a key of another context is refused
        check(troyhip_copy_h2d(sk_.get(), sk.data.data(), sk.data.size() * 8, nullptr));
    }
    void decrypt(const Ciphertext &ct, Plaintext &dst) const {
        const size_t N = c_.polyModulusDegree();
        const bool ckks = c_.parms().scheme() == SchemeType::ckks;
        const size_t words = ckks ? ct.coeffModulusSize() * N : N;
        DeviceArray out(words);
        check(troyhip_decrypt(c_.handle(), ct.raw(), sk_.get(), out.get(), words, 1, nullptr));
        dst.assignWords(words);
        check(troyhip_copy_d2h(dst.data(), out.get(), words * 8, nullptr));
        dst.setNttForm(ckks ? ct.parmsID() : parmsIDZero);
        if (ckks) dst.scale() = ct.scale();
    }
private:
    const SEALContext &c_;
    DeviceArray sk_;
};

// BatchEncoderCuda (src/batchencoder_cuda.cuh:12-86; CPU twin src/batchencoder.cpp:14-245): the 2 x (N/2) matrix of slot values modulo t <-> the plaintext
// polynomial.  Integer work on the host (troyhip_host_batch_encode / _decode: scatter by the 3^i index map + negacyclic NTT modulo t).
class BatchEncoder {
public:
    explicit BatchEncoder(const SEALContext &c) : c_(c), slots_(c.polyModulusDegree()) {
        const SchemeType s = c.parms().scheme();
        if (s != SchemeType::bfv && s != SchemeType::bgv) throw std::invalid_argument("unsupported scheme"); // batchencoder.cpp:23-26
    }
    size_t slotCount() const noexcept { return slots_; }
    void encode(const std::vector<uint64_t> &values, Plaintext &destination) const {
        if (values.size() > slots_) throw std::invalid_argument("values_matrix size is too large");
        destination = Plaintext();
        destination.resize(slots_);
        check(troyhip_host_batch_encode(c_.handle(), values.data(), values.size(), destination.data()));
    }
    void encode(const std::vector<int64_t> &values, Plaintext &destination) const { // negative values are stored as t + value (batchencoder.cpp:133-137)
        const uint64_t t = c_.parms().plainModulus().value();
        std::vector<uint64_t> u(values.size());
        for (size_t i = 0; i < values.size(); i++) u[i] = values[i] < 0 ? t + (uint64_t)values[i] : (uint64_t)values[i];
        encode(u, destination);
    }
    void decode(const Plaintext &plain, std::vector<uint64_t> &destination) const {
        if (plain.isNttForm()) throw std::invalid_argument("plain cannot be in NTT form");
        destination.assign(slots_, 0);
        check(troyhip_host_batch_decode(c_.handle(), plain.data(), plain.coeffCount(), destination.data()));
    }
    // encodePolynomial / decodePolynomial (src/batchencoder_cuda.cu:124-170, 267-286): the values ARE the coefficients (mod t), no slot
    // transform -- the packing app/LinearHelper.cuh is built on.  As there: the unsigned form keeps values.size() coefficients, the signed
    // form pads to N; decoding returns min(coeffCount, N) unsigned or N signed coefficients.
    void encodePolynomial(const std::vector<uint64_t> &values, Plaintext &destination) const {
        if (values.size() > slots_) throw std::invalid_argument("values_matrix size is too large");
        const uint64_t t = c_.parms().plainModulus().value();
        destination = Plaintext();
        destination.resize(values.size());
        uint64_t *d = destination.data();
        for (size_t i = 0; i < values.size(); i++) d[i] = values[i] % t;
    }
    void encodePolynomial(const std::vector<int64_t> &values, Plaintext &destination) const {
        if (values.size() > slots_) throw std::invalid_argument("values_matrix size is too large");
        const uint64_t t = c_.parms().plainModulus().value();
        destination = Plaintext();
        destination.resize(slots_);
        uint64_t *d = destination.data();
        for (size_t i = 0; i < values.size(); i++) d[i] = values[i] < 0 ? t - ((uint64_t)(-values[i]) % t) : (uint64_t)values[i] % t;
    }
    void decodePolynomial(const Plaintext &plain, std::vector<uint64_t> &destination) const {
        const size_t n = std::min(plain.coeffCount(), slots_);
        destination.assign(plain.data(), plain.data() + n);
    }
    void decodePolynomial(const Plaintext &plain, std::vector<int64_t> &destination) const {
        const uint64_t t = c_.parms().plainModulus().value(), half = t >> 1;
        destination.assign(slots_, 0);
        for (size_t i = 0; i < std::min(plain.coeffCount(), slots_); i++) destination[i] = plain[i] > half ? (int64_t)(plain[i] - t) : (int64_t)plain[i];
    }
    void decode(const Plaintext &plain, std::vector<int64_t> &destination) const { // values above t / 2 come back negative (batchencoder.cpp:215-243)
        std::vector<uint64_t> u;
        decode(plain, u);
        const uint64_t t = c_.parms().plainModulus().value(), half = (t + 1) >> 1;
        destination.resize(slots_);
        for (size_t i = 0; i < slots_; i++) destination[i] = u[i] >= half ? (int64_t)u[i] - (int64_t)t : (int64_t)u[i];
    }
private:
    const SEALContext &c_;
    size_t slots_;
};

// CKKSEncoderCuda (src/ckks_cuda.cuh:13-113), the coefficient packing app/LinearHelperCKKS.cuh is built on: encodePolynomial puts
// round(value * scale) into the polynomial's COEFFICIENTS (no canonical embedding) and leaves the plaintext in NTT form at the
// chosen level; decodePolynomial inverts it.  The per-coefficient integer work runs here on the host exactly as the reference's
// kernels define it (ckks_cuda.cu:211-330, 793-831); the transforms run on the GPU (troyhip_ntt).
class CKKSEncoder {
public:
    explicit CKKSEncoder(const SEALContext &c) : c_(c), slots_(c.polyModulusDegree() / 2) {
        if (c.parms().scheme() != SchemeType::ckks) throw std::invalid_argument("unsupported scheme"); // src/ckks.cpp:22-25
    }
    size_t slotCount() const noexcept { return slots_; }

    // encode / decode of N/2 complex slots (src/ckks_cuda.cuh:43-110; CPU twin src/ckks.cpp:98-260, 388-487): the canonical embedding -- slot i is
    // the value of the plaintext polynomial at zeta^(3^i), its conjugate at zeta^(-3^i), zeta = exp(i pi / N).  Double-precision FFT on the
    // host (the reference's own encoder is floating point too: results agree to rounding, not bit for bit); the coefficients then go
    // through encodePolynomial / decodePolynomial, i.e. the same exact integer packing and the GPU transforms.
    void encode(const std::vector<std::complex<double>> &values, const ParmsID &parms_id, double scale, Plaintext &destination) const {
        if (values.size() > slots_) throw std::invalid_argument("values_size is too large");
        const size_t n = slots_ * 2;
        const int logn = log2_of(n);
        std::vector<std::complex<double>> a(n, 0.0);
        for (size_t i = 0; i < values.size(); i++) { a[index_map(i, logn)] = values[i]; a[index_map(i + slots_, logn)] = std::conj(values[i]); }
        // inverse of the forward stage (j, j + t) -> (u + v W, u - v W): u = (x + y) / 2, v = (x - y) conj(W) / 2; the halvings are one 1 / n at the end
        for (size_t m = n >> 1, t = 1; m >= 1; m >>= 1, t <<= 1)
            for (size_t i = 0; i < m; i++) {
                const std::complex<double> w = std::conj(root(reverse_bits(m + i, logn), n));
                for (size_t j = 2 * i * t; j < 2 * i * t + t; j++) { const auto x = a[j], y = a[j + t]; a[j] = x + y; a[j + t] = (x - y) * w; }
            }
        std::vector<double> coeffs(n);
        for (size_t i = 0; i < n; i++) coeffs[i] = a[i].real() / (double)n;
        encodePolynomial(coeffs, parms_id, scale, destination);
    }
    void encode(const std::vector<std::complex<double>> &values, double scale, Plaintext &destination) const { encode(values, c_.firstParmsID(), scale, destination); }
    void encode(const std::vector<double> &values, const ParmsID &parms_id, double scale, Plaintext &destination) const {
        encode(std::vector<std::complex<double>>(values.begin(), values.end()), parms_id, scale, destination);
    }
    void encode(const std::vector<double> &values, double scale, Plaintext &destination) const { encode(values, c_.firstParmsID(), scale, destination); }
    void encode(double value, const ParmsID &parms_id, double scale, Plaintext &destination) const { // the constant polynomial (src/ckks.cpp:262-330)
        encodePolynomial(std::vector<double>{value}, parms_id, scale, destination);
    }
    void encode(double value, double scale, Plaintext &destination) const { encode(value, c_.firstParmsID(), scale, destination); }
    // one complex value in every slot (src/ckks_cuda.cuh:54-64; src/ckks.h: encodeInternal(std::complex<double>, ...) fills the slot vector with it)
    void encode(std::complex<double> value, const ParmsID &parms_id, double scale, Plaintext &destination) const {
        encode(std::vector<std::complex<double>>(slots_, value), parms_id, scale, destination);
    }
    void encode(std::complex<double> value, double scale, Plaintext &destination) const { encode(value, c_.firstParmsID(), scale, destination); }
    // an integer, exactly and without scaling (src/ckks_cuda.cuh:67-76, src/ckks_cuda.cu:733-790): the constant polynomial `value`, whose NTT form is
    // `value` modulo the prime in every position of every limb; scale 1
    void encode(std::int64_t value, const ParmsID &parms_id, Plaintext &destination) const {
        auto level = c_.getContextData(parms_id);
        if (!level) throw std::invalid_argument("parms_id is not valid for encryption parameters");
        const uint64_t magnitude = value < 0 ? (uint64_t)0 - (uint64_t)value : (uint64_t)value;
        const int bits = (magnitude ? 64 - __builtin_clzll(magnitude) : 0) + 2;
        if (bits >= level->totalCoeffModulusBitCount()) throw std::invalid_argument("encoded value is too large");
        const auto &q = level->parms().coeffModulus();
        const size_t n = slots_ * 2;
        destination.assignWords(q.size() * n);
        uint64_t *d = destination.data();
        for (size_t j = 0; j < q.size(); j++) {
            const uint64_t p = q[j].value(), r = magnitude % p;
            std::fill(d + j * n, d + (j + 1) * n, value < 0 && r ? p - r : r);
        }
        destination.setNttForm(parms_id);
        destination.scale() = 1.0;
    }
    void encode(std::int64_t value, Plaintext &destination) const { encode(value, c_.firstParmsID(), destination); }
    void decode(const Plaintext &plain, std::vector<std::complex<double>> &destination) const {
        std::vector<double> coeffs;
        decodePolynomial(plain, coeffs);
        const size_t n = slots_ * 2;
        const int logn = log2_of(n);
        std::vector<std::complex<double>> a(coeffs.begin(), coeffs.end());
        for (size_t m = 1, t = n >> 1; m < n; m <<= 1, t >>= 1)
            for (size_t i = 0; i < m; i++) {
                const std::complex<double> w = root(reverse_bits(m + i, logn), n);
                for (size_t j = 2 * i * t; j < 2 * i * t + t; j++) { const auto u = a[j], v = a[j + t] * w; a[j] = u + v; a[j + t] = u - v; }
            }
        destination.resize(slots_);
        for (size_t i = 0; i < slots_; i++) destination[i] = a[index_map(i, logn)];
    }
    void decode(const Plaintext &plain, std::vector<double> &destination) const {
        std::vector<std::complex<double>> c;
        decode(plain, c);
        destination.resize(c.size());
        for (size_t i = 0; i < c.size(); i++) destination[i] = c[i].real();
    }

    // ckks_cuda.cu:455-575 encodePolynomialInternal
    void encodePolynomial(const std::vector<double> &values, const ParmsID &parms_id, double scale, Plaintext &destination) const {
        auto level = c_.getContextData(parms_id);
        if (!level) throw std::invalid_argument("parms_id is not valid for encryption parameters");
        const size_t n = slots_ * 2;
        if (values.size() > n) throw std::invalid_argument("values_size is too large");
        const auto &q = level->parms().coeffModulus();
        const size_t limbs = q.size();
        std::vector<double> scaled(n, 0.0);
        double largest = 0;
        for (size_t i = 0; i < values.size(); i++) {
            scaled[i] = values[i] * scale;
            largest = std::max(largest, std::fabs(scaled[i]));
        }
        // one more bit for the sign; nothing below 1.0 goes through log2 (ckks_cuda.cu:516-521)
        const int bits = (int)std::ceil(std::log2(std::max(largest, 1.0))) + 1;
        if (bits >= level->totalCoeffModulusBitCount()) throw std::invalid_argument("encoded values are too large");
        std::vector<uint64_t> rns(limbs * n);
        for (size_t i = 0; i < n; i++) {
            const double r = std::round(scaled[i]);
            const bool negative = r < 0;
            // |r| = mant * 2^shift exactly (mant < 2^64): the three magnitude ranges of the reference (<= 64, <= 128, more bits)
            // all reduce this same integer, so one exact path serves them
            int e = 0;
            const double frac = std::frexp(std::fabs(r), &e);
            const int shift = e > 64 ? e - 64 : 0;
            const uint64_t mant = (uint64_t)std::ldexp(frac, e - shift);
            for (size_t j = 0; j < limbs; j++) {
                const uint64_t p = q[j].value();
                uint64_t v = mant % p;
                if (shift) v = (uint64_t)((unsigned __int128)v * pow2_mod((unsigned)shift, p) % p);
                rns[j * n + i] = negative && v ? p - v : v;
            }
        }
        auto dev = std::make_shared<DeviceArray>(limbs * n);
        std::vector<uint64_t> primes;
        for (auto &m : q) primes.push_back(m.value());
        check(troyhip_copy_h2d(dev->get(), rns.data(), rns.size() * 8, nullptr));
        check(troyhip_ntt(c_.handle(), dev->get(), limbs, primes.data(), (int)limbs, 1, 0, nullptr));
        destination.assignWords(limbs * n);
        check(troyhip_copy_d2h(destination.data(), dev->get(), limbs * n * 8, nullptr));
        destination.adoptDevice(dev); // the transformed buffer doubles as the plaintext's device copy
        destination.setNttForm(parms_id);
        destination.scale() = scale;
    }
    void encodePolynomial(const std::vector<double> &values, double scale, Plaintext &destination) const {
        encodePolynomial(values, c_.firstParmsID(), scale, destination);
    }

    // ckks_cuda.cu:983-1049 decodePolynomialInternal: inverse NTT, CRT composition, centred lift, times 1/scale -- the double
    // arithmetic word by word in the order of gDecodeInternal (ckks_cuda.cu:793-831) so the doubles come out the same
    void decodePolynomial(const Plaintext &plain, std::vector<double> &destination) const {
        if (!plain.isNttForm()) throw std::invalid_argument("plain is not in NTT form");
        auto level = c_.getContextData(plain.parmsID());
        if (!level) throw std::invalid_argument("plain is not valid for encryption parameters");
        const size_t n = slots_ * 2;
        const auto &q = level->parms().coeffModulus();
        const size_t limbs = q.size();
        if (plain.coeffCount() != limbs * n) throw std::invalid_argument("plain is not valid for encryption parameters");
        if (plain.scale() <= 0 || (int)std::log2(plain.scale()) >= level->totalCoeffModulusBitCount()) throw std::invalid_argument("scale out of bounds");
        std::vector<uint64_t> primes;
        for (auto &m : q) primes.push_back(m.value());
        DeviceArray dev(limbs * n);
        check(troyhip_copy_d2d(dev.get(), plain.device(), limbs * n * 8, nullptr));
        check(troyhip_ntt(c_.handle(), dev.get(), limbs, primes.data(), (int)limbs, 1, 1, nullptr));
        std::vector<uint64_t> rns(limbs * n);
        check(troyhip_copy_d2h(rns.data(), dev.get(), rns.size() * 8, nullptr));

        const std::vector<uint64_t> total = level->totalCoeffModulus();
        std::vector<uint64_t> half = total; // upper_half_threshold = (Q + 1) >> 1 (src/context.cpp:383-388)
        for (size_t w = 0, carry = 1; w < limbs && carry; w++) carry = ++half[w] == 0;
        for (size_t w = 0; w < limbs; w++) half[w] = (half[w] >> 1) | (w + 1 < limbs ? half[w + 1] << 63 : 0);
        // mixed-radix constants: inv[i][j] = (q_j)^-1 mod q_i for j < i
        std::vector<std::vector<uint64_t>> inv(limbs);
        for (size_t i = 0; i < limbs; i++)
            for (size_t j = 0; j < i; j++) inv[i].push_back(inv_mod(primes[j] % primes[i], primes[i]));

        destination.assign(n, 0.0);
        const double inv_scale = 1.0 / plain.scale(), two_pow_64 = std::pow(2.0, 64);
        std::vector<uint64_t> digit(limbs), word(limbs);
        for (size_t k = 0; k < n; k++) {
            // Garner: x = d0 + q0 (d1 + q1 (d2 + ...)), digits d_i in [0, q_i)
            for (size_t i = 0; i < limbs; i++) {
                const uint64_t p = primes[i];
                uint64_t v = rns[i * n + k];
                for (size_t j = 0; j < i; j++) {
                    const uint64_t dj = digit[j] % p;
                    v = (uint64_t)((unsigned __int128)(v >= dj ? v - dj : v + p - dj) * inv[i][j] % p);
                }
                digit[i] = v;
            }
            std::fill(word.begin(), word.end(), 0); // the composed integer in base 2^64 (what composeArray leaves, rns_cuda.cu)
            for (size_t i = limbs; i-- > 0;) {
                unsigned __int128 carry = digit[i];
                for (auto &w : word) { unsigned __int128 t = (unsigned __int128)w * primes[i] + carry; w = (uint64_t)t; carry = t >> 64; }
            }
            int cmp = 0;
            for (size_t w = limbs; w-- > 0 && !cmp;) cmp = word[w] < half[w] ? -1 : word[w] > half[w] ? 1 : 0;
            double acc = 0, unit = inv_scale;
            for (size_t w = 0; w < limbs; w++, unit *= two_pow_64) {
                if (cmp < 0) {
                    acc += word[w] ? (double)word[w] * unit : 0.0;
                } else if (word[w] > total[w]) {
                    acc += (double)(word[w] - total[w]) * unit;
                } else {
                    const uint64_t diff = total[w] - word[w];
                    acc -= diff ? (double)diff * unit : 0.0;
                }
            }
            destination[k] = acc;
        }
    }
private:
    static int log2_of(size_t n) { int l = 0; while ((size_t(1) << l) < n) l++; return l; }
    static size_t reverse_bits(size_t x, int bits) { size_t r = 0; for (int b = 0; b < bits; b++) r |= ((x >> b) & 1) << (bits - 1 - b); return r; }
    static std::complex<double> root(size_t k, size_t n) { // exp(2 pi i k / 2n)
        const double ang = 3.14159265358979323846264338327950288 * (double)k / (double)n;
        return {std::cos(ang), std::sin(ang)};
    }
    // matrix_reps_index_map_ (src/ckks.cpp:50-69): slot i -> bit-reversed (3^i - 1) / 2, slot i + N/2 -> bit-reversed (2N - 3^i - 1) / 2
    size_t index_map(size_t i, int logn) const {
        const uint64_t m = (uint64_t)slots_ * 4;
        uint64_t pos = 1, base = 3;
        for (size_t e = i % slots_; e; e >>= 1, base = base * base & (m - 1))
            if (e & 1) pos = pos * base & (m - 1);
        return reverse_bits((size_t)((i < slots_ ? pos - 1 : m - pos - 1) >> 1), logn);
    }
    static uint64_t pow2_mod(unsigned e, uint64_t p) {
        unsigned __int128 r = 1, b = 2 % p;
        for (; e; e >>= 1, b = b * b % p)
            if (e & 1) r = r * b % p;
        return (uint64_t)r;
    }
    static uint64_t inv_mod(uint64_t a, uint64_t p) { // p prime
        unsigned __int128 r = 1, b = a % p;
        for (uint64_t e = p - 2; e; e >>= 1, b = b * b % p)
            if (e & 1) r = r * b % p;
        return (uint64_t)r;
    }
    const SEALContext &c_;
    size_t slots_;
};

class Evaluator { // src/evaluator_cuda.cuh:13-361 -- every method const, non-copyable
public:
    explicit Evaluator(const SEALContext &c) : c_(c) {}
    const SEALContext &context() const { return c_; }
    // a key-switching key as the kernels take it, refused unless it belongs to this evaluator's context (KSwitchKeys::device)
    const uint64_t *key_of(const KSwitchKeys &k, size_t index) const { return k.device(index, c_.keyParmsID(), c_.keyLimbs(), c_.polyModulusDegree()); }
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
    // the destination forms write a fresh ciphertext: the operands are read where they lie (the reference copies, then multiplies in place)
    void multiply(const Ciphertext &a, const Ciphertext &b, Ciphertext &d) const {
        if (&d == &a) { multiplyInplace(d, b); return; }
        Ciphertext out;
        out.resize(a.polyModulusDegree(), a.coeffModulusSize(), a.size() + b.size() - 1);
        check(troyhip_multiply(h(), a.raw(), b.raw(), out.raw(), 1, nullptr));
        out.bind(a);
        d = std::move(out);
    }
    void squareInplace(Ciphertext &a) const { multiplyInplace(a, a); }
    void square(const Ciphertext &a, Ciphertext &d) const { multiply(a, a, d); }
    void relinearizeInplace(Ciphertext &a, const RelinKeys &k) const { // to size 2 from any size <= 16 (src/evaluator_cuda.cu:703-744)
        const size_t need = a.size() > 2 ? a.size() - 2 : 0;
        std::vector<const uint64_t *> keys(need ? need : 1, nullptr);
        for (size_t i = 0; i < need; i++) {
            if (!k.hasKey(i + 2)) throw std::invalid_argument("not enough relinearization keys");
            keys[i] = key_of(k, RelinKeys::getIndex(i + 2));
        }
        check(troyhip_relinearize_keys(h(), a.raw(), keys.data(), (int)need, 1, nullptr));
    }
    // the reference copies and relinearizes in place; from size 3 the library reads the operand where it lies (troyhip_relinearize_to): no copy
    void relinearize(const Ciphertext &a, const RelinKeys &k, Ciphertext &d) const {
        if (a.size() != 3 || &d == &a) { d = a; relinearizeInplace(d, k); return; }
        if (!k.hasKey(2)) throw std::invalid_argument("not enough relinearization keys");
        const uint64_t *key = key_of(k, RelinKeys::getIndex(2));
        Ciphertext out;
        out.resize(a.polyModulusDegree(), a.coeffModulusSize(), 2);
        check(troyhip_relinearize_to(h(), a.raw(), out.raw(), &key, 1, 1, nullptr));
        out.bind(a);
        d = std::move(out);
    }
    // applyKeySwitchingInplace (evaluator_cuda.cu:1365-1378), negacyclicShiftInplace (:2342-2351)
    void applyKeySwitchingInplace(Ciphertext &a, const KSwitchKeys &k) const {
        if (k.all().size() != 1) throw std::invalid_argument("kswitch_keys.data().size() != 1");
        check(troyhip_apply_key_switching(h(), a.raw(), key_of(k, k.all().begin()->first), 1, nullptr));
    }
    void applyKeySwitching(const Ciphertext &a, const KSwitchKeys &k, Ciphertext &d) const { d = a; applyKeySwitchingInplace(d, k); } // src/evaluator_cuda.cuh:104-110
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
    void exponentiate(const Ciphertext &a, uint64_t exponent, const RelinKeys &rk, Ciphertext &d) const { d = a; exponentiateInplace(d, exponent, rk); } // :206-211
    void modSwitchToNextInplace(Ciphertext &a) const { next(a, troyhip_mod_switch_to_next); }
    void modSwitchToNext(const Ciphertext &a, Ciphertext &d) const { d = a; modSwitchToNextInplace(d); }
    void modSwitchToInplace(Ciphertext &a, const ParmsID &parms_id) const {
        if (!c_.getContextData(parms_id)) throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (parms_id.limbs > a.parmsID().limbs) throw std::invalid_argument("cannot switch to higher level modulus");
        while (a.parmsID() != parms_id) modSwitchToNextInplace(a);
    }
    void modSwitchTo(const Ciphertext &a, const ParmsID &parms_id, Ciphertext &d) const { d = a; modSwitchToInplace(d, parms_id); } // :162-167
    // Plaintext forms (src/evaluator_cuda.cuh:140-151,170-177; modSwitchDropToNext(PlaintextCuda&) src/evaluator_cuda.cu:891-925): an NTT-form
    // plaintext [limbs][N] loses its last limb; same checks, same messages
    void modSwitchToNextInplace(Plaintext &plain) const {
        if (!plain.isNttForm()) throw std::invalid_argument("plain is not in NTT form");
        auto cd = c_.getContextData(plain.parmsID());
        if (!cd) throw std::invalid_argument("plain is not valid for encryption parameters");
        auto nx = cd->nextContextData();
        if (!nx) throw std::invalid_argument("end of modulus switching chain reached");
        if (!scaleWithinBounds(plain.scale(), *nx)) throw std::invalid_argument("scale out of bounds");
        plain.keepWords(nx->parms().coeffModulus().size() * c_.polyModulusDegree());
        plain.setNttForm(nx->parmsID());
    }
    void modSwitchToNext(const Plaintext &plain, Plaintext &d) const { d = plain; modSwitchToNextInplace(d); }
    void modSwitchToInplace(Plaintext &plain, const ParmsID &parms_id) const { // src/evaluator_cuda.cu:982-1008
        auto cd = c_.getContextData(plain.parmsID()), target = c_.getContextData(parms_id);
        if (!cd) throw std::invalid_argument("plain is not valid for encryption parameters");
        if (!target) throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (!plain.isNttForm()) throw std::invalid_argument("plain is not in NTT form");
        if (cd->chainIndex() < target->chainIndex()) throw std::invalid_argument("cannot switch to higher level modulus");
        while (plain.parmsID() != parms_id) modSwitchToNextInplace(plain);
    }
    void modSwitchTo(const Plaintext &plain, const ParmsID &parms_id, Plaintext &d) const { d = plain; modSwitchToInplace(d, parms_id); }
    void rescaleToNextInplace(Ciphertext &a) const { next(a, troyhip_rescale_to_next); }
    void rescaleToNext(const Ciphertext &a, Ciphertext &d) const { d = a; rescaleToNextInplace(d); }
    void rescaleToInplace(Ciphertext &a, const ParmsID &parms_id) const {
        if (!c_.getContextData(parms_id)) throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (parms_id.limbs > a.parmsID().limbs) throw std::invalid_argument("cannot switch to higher level modulus");
        while (a.parmsID() != parms_id) rescaleToNextInplace(a);
    }
    void rescaleTo(const Ciphertext &a, const ParmsID &parms_id, Ciphertext &d) const { d = a; rescaleToInplace(d, parms_id); } // :193-198
    void applyGaloisInplace(Ciphertext &a, uint32_t galois_elt, const GaloisKeys &gk) const {
        if (!gk.hasKey(galois_elt)) throw std::invalid_argument("Galois key not present");
        check(troyhip_apply_galois(h(), a.raw(), galois_elt, key_of(gk, GaloisKeys::getIndex(galois_elt)), 1, nullptr));
    }
    void rotateRowsInplace(Ciphertext &a, int steps, const GaloisKeys &gk) const { need(SchemeType::ckks, false); rotate(a, steps, 0, gk); }
    void rotateColumnsInplace(Ciphertext &a, const GaloisKeys &gk) const { need(SchemeType::ckks, false); rotate(a, 0, 1, gk); }
    void rotateVectorInplace(Ciphertext &a, int steps, const GaloisKeys &gk) const { need(SchemeType::ckks, true); rotate(a, steps, 0, gk); }
    void complexConjugateInplace(Ciphertext &a, const GaloisKeys &gk) const { need(SchemeType::ckks, true); rotate(a, 0, 1, gk); }
    // the out-of-place forms of src/evaluator_cuda.cuh:273-349: destination = encrypted; ...Inplace(destination)
    void applyGalois(const Ciphertext &a, uint32_t galois_elt, const GaloisKeys &gk, Ciphertext &d) const { d = a; applyGaloisInplace(d, galois_elt, gk); }
    void rotateRows(const Ciphertext &a, int steps, const GaloisKeys &gk, Ciphertext &d) const { d = a; rotateRowsInplace(d, steps, gk); }
    void rotateColumns(const Ciphertext &a, const GaloisKeys &gk, Ciphertext &d) const { d = a; rotateColumnsInplace(d, gk); }
    void rotateVector(const Ciphertext &a, int steps, const GaloisKeys &gk, Ciphertext &d) const { d = a; rotateVectorInplace(d, steps, gk); }
    void complexConjugate(const Ciphertext &a, const GaloisKeys &gk, Ciphertext &d) const { d = a; complexConjugateInplace(d, gk); }
    void transformToNttInplace(Ciphertext &a) const { check(troyhip_transform_to_ntt(h(), a.raw(), 1, nullptr)); }
    void transformFromNttInplace(Ciphertext &a) const { check(troyhip_transform_from_ntt(h(), a.raw(), 1, nullptr)); }
    void transformToNtt(const Ciphertext &a, Ciphertext &d) const { d = a; transformToNttInplace(d); }     // :246-250
    void transformFromNtt(const Ciphertext &a, Ciphertext &d) const { d = a; transformFromNttInplace(d); } // :254-258
    // multiplyPlainInplace (evaluator_cuda.cu:1722-1755): NTT-form pair -> multiplyPlainNtt, coefficient-form pair -> multiplyPlainNormal
    void multiplyPlainInplace(Ciphertext &a, const Plaintext &plain) const {
        if (a.isNttForm() != plain.isNttForm() && c_.parms().scheme() != SchemeType::ckks) throw std::invalid_argument("NTT form mismatch");
        if (a.isNttForm() && plain.isNttForm() && a.parmsID() != plain.parmsID()) throw std::invalid_argument("encrypted_ntt and plain_ntt parameter mismatch");
        if (a.isNttForm()) check(troyhip_multiply_plain_ntt(h(), a.raw(), plain.device(), plain.scale(), 1, nullptr));
        else check(troyhip_multiply_plain(h(), a.raw(), plain.device(), plain.coeffCount(), 0, 1, nullptr));
    }
    void multiplyPlain(const Ciphertext &a, const Plaintext &plain, Ciphertext &d) const { d = a; multiplyPlainInplace(d, plain); }
    // ---- batched forms over slab members (Ciphertext::allocateBatch / packBatch): ONE library launch for the whole run of
    // ciphertexts, the same plaintext against each (no counterpart in the reference, which loops; used by troyn_app.hpp)
    std::vector<Ciphertext> multiplyPlainBatch(const std::vector<const Ciphertext *> &column, const Plaintext &plain) const {
        if (column.empty()) return {};
        const Ciphertext &head = *column[0];
        if (head.isNttForm() != plain.isNttForm() && c_.parms().scheme() != SchemeType::ckks) throw std::invalid_argument("NTT form mismatch");
        if (head.isNttForm() && plain.isNttForm() && head.parmsID() != plain.parmsID()) throw std::invalid_argument("encrypted_ntt and plain_ntt parameter mismatch");
        const size_t count = column.size();
        std::vector<Ciphertext> out;
        if (Ciphertext::isBatch(column)) { // a dense run of one slab: one copy
            out = Ciphertext::allocateBatch(count, head);
            check(troyhip_copy_d2d(out[0].raw()->data, head.raw()->data, count * head.raw()->batch_stride * 8, nullptr));
        } else out = Ciphertext::packBatch(column); // anything else: one copy per ciphertext
        troyhip_ct t = *out[0].raw();
        if (head.isNttForm()) check(troyhip_multiply_plain_ntt(h(), &t, plain.device(), plain.scale(), count, nullptr));
        else check(troyhip_multiply_plain(h(), &t, plain.device(), plain.coeffCount(), 0, count, nullptr));
        for (auto &c : out) c.copyMeta(t);
        return out;
    }
    // sum_i column_i (x) plain_i for whole slab columns in ONE pass (troyhip_multiply_plain_accumulate): what a linear layer's
    // multiplyPlain + addInplace loop computes per output block, same residues, every operand read once.  Up to 16 products.
    std::vector<Ciphertext> multiplyPlainAccumulateBatch(const std::vector<std::vector<const Ciphertext *>> &columns, const std::vector<const Plaintext *> &plains) const {
        if (columns.empty() || columns.size() != plains.size() || columns.size() > 16) throw std::invalid_argument("multiplyPlainAccumulateBatch: 1 to 16 (column, plaintext) pairs");
        const size_t count = columns.size(), batch = columns[0].size();
        std::vector<troyhip_ct> views(count);
        std::vector<const troyhip_ct *> ct_ptrs(count);
        std::vector<const uint64_t *> pl_ptrs(count);
        for (size_t i = 0; i < count; i++) {
            if (columns[i].size() != batch || !Ciphertext::isBatch(columns[i])) throw std::invalid_argument("multiplyPlainAccumulateBatch: the ciphertexts are not dense runs of slabs");
            const Ciphertext &head = *columns[i][0];
            if (!head.isNttForm() || !plains[i]->isNttForm()) throw std::invalid_argument("NTT form mismatch");
            if (head.parmsID() != plains[i]->parmsID()) throw std::invalid_argument("encrypted_ntt and plain_ntt parameter mismatch");
            if (plains[i]->scale() != plains[0]->scale()) throw std::invalid_argument("scale mismatch");
            views[i] = *head.raw();
            ct_ptrs[i] = &views[i];
            pl_ptrs[i] = plains[i]->device();
        }
        std::vector<Ciphertext> out = Ciphertext::allocateBatch(batch, *columns[0][0]);
        troyhip_ct t = *out[0].raw();
        check(troyhip_multiply_plain_accumulate(h(), ct_ptrs.data(), pl_ptrs.data(), (int)count, plains[0]->scale(), &t, batch, nullptr));
        for (auto &c : out) c.copyMeta(t);
        return out;
    }
    void addInplaceBatch(std::vector<Ciphertext> &acc, const std::vector<Ciphertext> &x) const {
        if (acc.size() != x.size()) throw std::invalid_argument("Size incorrect.");
        std::vector<const Ciphertext *> pa, px;
        for (auto &c : acc) pa.push_back(&c);
        for (auto &c : x) px.push_back(&c);
        if (!Ciphertext::isBatch(pa) || !Ciphertext::isBatch(px)) {
            for (size_t b = 0; b < acc.size(); b++) addInplace(acc[b], x[b]);
            return;
        }
        troyhip_ct t = *acc[0].raw();
        check(troyhip_add(h(), &t, x[0].raw(), acc.size(), nullptr));
        for (auto &c : acc) c.copyMeta(t);
    }
    // ---- the hot path, batched (round 4).  One ciphertext cannot fill 256 compute units: a single multiply + relinearize at N = 2^15 runs at a third
    // of the rate a batch of 128 reaches.  Every method below is ONE library call over `count` independent ciphertexts of one shape -- the calls of
    // src/evaluator_cuda.cuh:85-115,193-198,292-344 with the reference's checks and messages, on std::vector operands.  Operands that already are
    // consecutive members of one slab (Ciphertext::allocateBatch, or the result of a ...Batch call) are used where they lie; anything else is
    // packed into a slab first (one device copy per ciphertext -- small next to the operation).  Results are slab members: chains of ...Batch
    // calls never copy.  The ...InplaceBatch forms take pointers, group consecutive items of equal shape, and leave every item a slab member.
    std::vector<Ciphertext> multiplyBatch(const std::vector<const Ciphertext *> &a, const std::vector<const Ciphertext *> &b) const {
        if (a.size() != b.size()) throw std::invalid_argument("multiplyBatch: operand counts differ");
        if (a.empty()) return {};
        std::vector<Ciphertext> pa, pb;
        const bool square = a == b;
        const troyhip_ct va = runOf(a, pa), vb = square ? va : runOf(b, pb);
        std::vector<Ciphertext> out = Ciphertext::allocateBatch(a.size(), *a[0], a[0]->size() + b[0]->size() - 1, a[0]->coeffModulusSize());
        troyhip_ct t = *out[0].raw();
        check(troyhip_multiply(h(), &va, &vb, &t, a.size(), nullptr));
        for (auto &c : out) c.copyMeta(t);
        return out;
    }
    std::vector<Ciphertext> multiplyBatch(const std::vector<Ciphertext> &a, const std::vector<Ciphertext> &b) const { return multiplyBatch(Ciphertext::pointers(a), Ciphertext::pointers(b)); }
    std::vector<Ciphertext> squareBatch(const std::vector<const Ciphertext *> &a) const { return multiplyBatch(a, a); }
    std::vector<Ciphertext> squareBatch(const std::vector<Ciphertext> &a) const { return squareBatch(Ciphertext::pointers(a)); }
    void multiplyInplaceBatch(const std::vector<Ciphertext *> &a, const std::vector<const Ciphertext *> &b) const {
        std::vector<Ciphertext> out = multiplyBatch(std::vector<const Ciphertext *>(a.begin(), a.end()), b);
        for (size_t i = 0; i < a.size(); i++) *a[i] = std::move(out[i]);
    }
    // relinearize(encrypted, relin_keys, destination) over a batch: from size 3 the operands are read where they lie (troyhip_relinearize_to)
    std::vector<Ciphertext> relinearizeBatch(const std::vector<const Ciphertext *> &a, const RelinKeys &k) const {
        if (a.empty()) return {};
        std::vector<Ciphertext> pa;
        const troyhip_ct va = runOf(a, pa);
        const std::vector<const uint64_t *> keys = relinKeys(*a[0], k);
        std::vector<Ciphertext> out = Ciphertext::allocateBatch(a.size(), *a[0], a[0]->size() == 3 ? 2 : a[0]->size(), a[0]->coeffModulusSize());
        troyhip_ct t = *out[0].raw();
        check(troyhip_relinearize_to(h(), &va, &t, keys.data(), (int)(a[0]->size() > 2 ? a[0]->size() - 2 : 0), a.size(), nullptr));
        for (auto &c : out) c.copyMeta(t);
        return out;
    }
    std::vector<Ciphertext> relinearizeBatch(const std::vector<Ciphertext> &a, const RelinKeys &k) const { return relinearizeBatch(Ciphertext::pointers(a), k); }
    void relinearizeInplaceBatch(const std::vector<Ciphertext *> &items, const RelinKeys &k) const {
        inplaceBatch(items, [&](troyhip_ct *v, size_t count, const Ciphertext &head) {
            const std::vector<const uint64_t *> keys = relinKeys(head, k);
            check(troyhip_relinearize_keys(h(), v, keys.data(), (int)(head.size() > 2 ? head.size() - 2 : 0), count, nullptr));
        });
    }
    void relinearizeInplaceBatch(std::vector<Ciphertext> &items, const RelinKeys &k) const { relinearizeInplaceBatch(Ciphertext::pointers(items), k); }
    void applyKeySwitchingInplaceBatch(const std::vector<Ciphertext *> &items, const KSwitchKeys &k) const {
        if (k.all().size() != 1) throw std::invalid_argument("kswitch_keys.data().size() != 1");
        const uint64_t *key = key_of(k, k.all().begin()->first);
        inplaceBatch(items, [&](troyhip_ct *v, size_t count, const Ciphertext &) { check(troyhip_apply_key_switching(h(), v, key, count, nullptr)); });
    }
    std::vector<Ciphertext> modSwitchToNextBatch(const std::vector<const Ciphertext *> &a) const { return nextBatch(a, troyhip_mod_switch_to_next); }
    std::vector<Ciphertext> modSwitchToNextBatch(const std::vector<Ciphertext> &a) const { return modSwitchToNextBatch(Ciphertext::pointers(a)); }
    std::vector<Ciphertext> rescaleToNextBatch(const std::vector<const Ciphertext *> &a) const { return nextBatch(a, troyhip_rescale_to_next); }
    std::vector<Ciphertext> rescaleToNextBatch(const std::vector<Ciphertext> &a) const { return rescaleToNextBatch(Ciphertext::pointers(a)); }
    void modSwitchToNextInplaceBatch(const std::vector<Ciphertext *> &items) const { nextInplaceBatch(items, troyhip_mod_switch_to_next); }
    void modSwitchToNextInplaceBatch(std::vector<Ciphertext> &items) const { modSwitchToNextInplaceBatch(Ciphertext::pointers(items)); }
    void rescaleToNextInplaceBatch(const std::vector<Ciphertext *> &items) const { nextInplaceBatch(items, troyhip_rescale_to_next); }
    void rescaleToNextInplaceBatch(std::vector<Ciphertext> &items) const { rescaleToNextInplaceBatch(Ciphertext::pointers(items)); }
    void applyGaloisInplaceBatch(const std::vector<Ciphertext *> &items, uint32_t galois_elt, const GaloisKeys &gk) const {
        if (!gk.hasKey(galois_elt)) throw std::invalid_argument("Galois key not present");
        const uint64_t *key = key_of(gk, GaloisKeys::getIndex(galois_elt));
        inplaceBatch(items, [&](troyhip_ct *v, size_t count, const Ciphertext &) { check(troyhip_apply_galois(h(), v, galois_elt, key, count, nullptr)); });
    }
    void applyGaloisInplaceBatch(std::vector<Ciphertext> &items, uint32_t galois_elt, const GaloisKeys &gk) const { applyGaloisInplaceBatch(Ciphertext::pointers(items), galois_elt, gk); }
    void rotateRowsInplaceBatch(const std::vector<Ciphertext *> &items, int steps, const GaloisKeys &gk) const { need(SchemeType::ckks, false); rotateBatch(items, steps, 0, gk); }
    void rotateRowsInplaceBatch(std::vector<Ciphertext> &items, int steps, const GaloisKeys &gk) const { rotateRowsInplaceBatch(Ciphertext::pointers(items), steps, gk); }
    void rotateColumnsInplaceBatch(const std::vector<Ciphertext *> &items, const GaloisKeys &gk) const { need(SchemeType::ckks, false); rotateBatch(items, 0, 1, gk); }
    void rotateColumnsInplaceBatch(std::vector<Ciphertext> &items, const GaloisKeys &gk) const { rotateColumnsInplaceBatch(Ciphertext::pointers(items), gk); }
    void rotateVectorInplaceBatch(const std::vector<Ciphertext *> &items, int steps, const GaloisKeys &gk) const { need(SchemeType::ckks, true); rotateBatch(items, steps, 0, gk); }
    void rotateVectorInplaceBatch(std::vector<Ciphertext> &items, int steps, const GaloisKeys &gk) const { rotateVectorInplaceBatch(Ciphertext::pointers(items), steps, gk); }
    void complexConjugateInplaceBatch(const std::vector<Ciphertext *> &items, const GaloisKeys &gk) const { need(SchemeType::ckks, true); rotateBatch(items, 0, 1, gk); }
    void complexConjugateInplaceBatch(std::vector<Ciphertext> &items, const GaloisKeys &gk) const { complexConjugateInplaceBatch(Ciphertext::pointers(items), gk); }
    // the element-wise and transform calls over a batch (cheap per ciphertext, but a launch each when looped: 8-15 us apiece)
    void negateInplaceBatch(const std::vector<Ciphertext *> &items) const {
        inplaceBatch(items, [&](troyhip_ct *v, size_t count, const Ciphertext &) { check(troyhip_negate(h(), v, count, nullptr)); });
    }
    void addInplaceBatch(const std::vector<Ciphertext *> &acc, const std::vector<const Ciphertext *> &x) const { addsubBatch(acc, x, false); }
    void subInplaceBatch(const std::vector<Ciphertext *> &acc, const std::vector<const Ciphertext *> &x) const { addsubBatch(acc, x, true); }
    void transformToNttInplaceBatch(const std::vector<Ciphertext *> &items) const {
        inplaceBatch(items, [&](troyhip_ct *v, size_t count, const Ciphertext &) { check(troyhip_transform_to_ntt(h(), v, count, nullptr)); });
    }
    void transformFromNttInplaceBatch(const std::vector<Ciphertext *> &items) const {
        inplaceBatch(items, [&](troyhip_ct *v, size_t count, const Ciphertext &) { check(troyhip_transform_from_ntt(h(), v, count, nullptr)); });
    }
    void negacyclicShiftInplaceBatch(const std::vector<Ciphertext *> &items, size_t shift) const {
        inplaceBatch(items, [&](troyhip_ct *v, size_t count, const Ciphertext &) { check(troyhip_negacyclic_shift(h(), v, shift, count, nullptr)); });
    }
    void divideByPolyModulusDegreeInplaceBatch(const std::vector<Ciphertext *> &items, uint64_t mul = 1) const {
        inplaceBatch(items, [&](troyhip_ct *v, size_t count, const Ciphertext &) { check(troyhip_divide_by_poly_modulus_degree(h(), v, mul, count, nullptr)); });
    }
    // fieldTraceInplace (src/evaluator_cuda.cu:2278-2294) over a batch: log2(N) - logn rounds of (copy, automorphism X -> X^(N / 2^k + 1), add),
    // each round one key switch over all the ciphertexts
    void fieldTraceInplaceBatch(const std::vector<Ciphertext *> &items, const GaloisKeys &automorphism_keys, size_t logn) const {
        if (items.empty()) return;
        const std::vector<const Ciphertext *> view(items.begin(), items.end());
        for (size_t poly_degree = c_.polyModulusDegree(); poly_degree > (size_t(1) << logn); poly_degree >>= 1) {
            std::vector<Ciphertext> temp = Ciphertext::packBatch(view);
            applyGaloisInplaceBatch(temp, (uint32_t)(poly_degree + 1), automorphism_keys);
            addInplaceBatch(items, Ciphertext::pointers(const_cast<const std::vector<Ciphertext> &>(temp)));
        }
    }
    void multiplyPlainInplaceBatch(const std::vector<Ciphertext *> &items, const Plaintext &plain) const { // one plaintext against every item
        inplaceBatch(items, [&](troyhip_ct *v, size_t count, const Ciphertext &head) {
            if (head.isNttForm() != plain.isNttForm() && c_.parms().scheme() != SchemeType::ckks) throw std::invalid_argument("NTT form mismatch");
            if (head.isNttForm() && plain.isNttForm() && head.parmsID() != plain.parmsID()) throw std::invalid_argument("encrypted_ntt and plain_ntt parameter mismatch");
            if (head.isNttForm()) check(troyhip_multiply_plain_ntt(h(), v, plain.device(), plain.scale(), count, nullptr));
            else check(troyhip_multiply_plain(h(), v, plain.device(), plain.coeffCount(), 0, count, nullptr));
        });
    }
    void addPlainInplaceBatch(const std::vector<Ciphertext *> &items, const Plaintext &plain) const {
        inplaceBatch(items, [&](troyhip_ct *v, size_t count, const Ciphertext &) { check(troyhip_add_plain(h(), v, plain.device(), plain.coeffCount(), 0, plain.scale(), 0, count, nullptr)); });
    }
    // addPlainInplace / subPlainInplace (evaluator_cuda.cu:1654-1720)
    void addPlainInplace(Ciphertext &a, const Plaintext &plain) const { plain_addsub(a, plain, 0); }
    void subPlainInplace(Ciphertext &a, const Plaintext &plain) const { plain_addsub(a, plain, 1); }
    void addPlain(const Ciphertext &a, const Plaintext &plain, Ciphertext &d) const { d = a; addPlainInplace(d, plain); }
    void subPlain(const Ciphertext &a, const Plaintext &plain, Ciphertext &d) const { d = a; subPlainInplace(d, plain); }
    // transformToNttInplace(Plaintext&, parms_id) (evaluator_cuda.cu:1866-1948)
    void transformToNttInplace(Plaintext &plain, const ParmsID &parms_id) const {
        if (plain.isNttForm()) throw std::invalid_argument("plain is already in NTT form");
        if (!c_.getContextData(parms_id)) throw std::invalid_argument("parms_id is not valid for the current context");
        const size_t n = c_.polyModulusDegree(), limbs = (size_t)parms_id.limbs;
        if (!plain.coeffCount()) { // Plaintext("0") holds no coefficient at all: the zero polynomial, zero in every limb
            plain.assignWords(limbs * n);
            plain.setNttForm(parms_id);
            return;
        }
        DeviceArray p(plain.coeffCount()), out(limbs * n);
        check(troyhip_copy_h2d(p.get(), plain.data(), plain.coeffCount() * 8, nullptr));
        check(troyhip_plain_to_ntt(h(), p.get(), plain.coeffCount(), 0, (int)limbs, out.get(), 1, nullptr));
        plain.assignWords(limbs * n);
        check(troyhip_copy_d2h(plain.data(), out.get(), limbs * n * 8, nullptr));
        plain.setNttForm(parms_id);
    }
    void transformToNtt(const Plaintext &plain, const ParmsID &parms_id, Plaintext &d) const { d = plain; transformToNttInplace(d, parms_id); } // :237-242

    void negacyclicShift(const Ciphertext &a, size_t shift, Ciphertext &d) const { d = a; negacyclicShiftInplace(d, shift); }
    // divideByPolyModulusDegreeInplace (src/evaluator_cuda.cu:2262-2276): every limb times N^-1 (times `mul`)
    void divideByPolyModulusDegreeInplace(Ciphertext &a, uint64_t mul = 1) const { check(troyhip_divide_by_poly_modulus_degree(h(), a.raw(), mul, 1, nullptr)); }
    // extractLWE / assembleLWE / fieldTraceInplace / packLWECiphertexts (src/evaluator_cuda.cu:2178-2340), composed exactly as there
    LWECiphertext extractLWE(const Ciphertext &encrypted, size_t term) const {
        if (encrypted.size() != 2) throw std::invalid_argument("Encrypted size must be 2 to be extracted.");
        if (encrypted.isNttForm()) { Ciphertext t = encrypted; transformFromNttInplace(t); return extractLWE(t, term); }
        const size_t N = c_.polyModulusDegree(), L = encrypted.coeffModulusSize();
        Ciphertext c1;
        c1.resize(N, L, 1);
        c1.bind(c_);
        check(troyhip_copy_d2d(c1.raw()->data, encrypted.raw()->data + L * N, L * N * 8, nullptr));
        troyhip_ct *r = c1.raw();
        r->is_ntt_form = 0; r->scale = encrypted.scale(); r->correction_factor = encrypted.correctionFactor();
        negacyclicShiftInplace(c1, term == 0 ? 0 : 2 * N - term);
        LWECiphertext out;
        out.parms_id_ = encrypted.parmsID(); out.n_ = N; out.limbs_ = L; out.scale_ = encrypted.scale(); out.cf_ = encrypted.correctionFactor();
        out.c1_.resize(L * N);
        check(troyhip_copy_d2d(out.c1_.get(), c1.raw()->data, L * N * 8, nullptr));
        const std::vector<uint64_t> host = encrypted.toHost();
        out.c0_.resize(L);
        for (size_t l = 0; l < L; l++) out.c0_[l] = host[l * N + term];
        return out;
    }
    Ciphertext assembleLWE(const LWECiphertext &lwe, size_t term) const {
        const size_t N = lwe.n_, L = lwe.limbs_;
        Ciphertext c1;
        c1.resize(N, L, 1);
        c1.bind(c_);
        check(troyhip_copy_d2d(c1.raw()->data, lwe.c1_.get(), L * N * 8, nullptr));
        troyhip_ct *r = c1.raw();
        r->is_ntt_form = 0; r->scale = lwe.scale_; r->correction_factor = lwe.cf_;
        negacyclicShiftInplace(c1, term);
        std::vector<uint64_t> host(2 * L * N, 0), sh = c1.toHost();
        std::copy(sh.begin(), sh.end(), host.begin() + (long)(L * N));
        for (size_t l = 0; l < L; l++) host[l * N + term] = lwe.c0_[l];
        Ciphertext out;
        out.fromHost(host, N, L, 2, false, lwe.scale_, lwe.cf_);
        out.bind(c_);
        return out;
    }
    void fieldTraceInplace(Ciphertext &encrypted, const GaloisKeys &automorphism_keys, size_t logn) const {
        size_t poly_degree = c_.polyModulusDegree();
        while (poly_degree > (size_t(1) << logn)) {
            Ciphertext temp = encrypted;
            applyGaloisInplace(temp, (uint32_t)(poly_degree + 1), automorphism_keys);
            addInplace(encrypted, temp);
            poly_degree >>= 1;
        }
    }
    Ciphertext packLWECiphertexts(const std::vector<LWECiphertext> &lwes, const GaloisKeys &automorphism_keys) const {
        if (lwes.empty()) throw std::invalid_argument("LWE ciphertexts must not be empty.");
        for (const LWECiphertext &w : lwes)
            if (w.parmsID() != lwes[0].parmsID()) throw std::invalid_argument("LWE ciphertexts must have same parmsID.");
        const size_t N = c_.polyModulusDegree();
        const bool ckks = c_.parms().scheme() == SchemeType::ckks;
        size_t l = 0;
        while ((size_t(1) << l) < lwes.size()) l++;
        Ciphertext zero = assembleLWE(lwes[0], 0);
        check(troyhip_memset_zero(zero.raw()->data, 2 * zero.coeffModulusSize() * N * 8, nullptr));
        std::vector<Ciphertext> rl(size_t(1) << l);
        for (size_t i = 0; i < rl.size(); i++) {
            size_t index = 0;
            for (size_t b = 0; b < l; b++) index |= ((i >> b) & 1) << (l - 1 - b);
            if (index < lwes.size()) { rl[i] = assembleLWE(lwes[index], 0); divideByPolyModulusDegreeInplace(rl[i]); }
            else rl[i] = zero;
        }
        for (size_t layer = 0; layer < l; layer++) {
            const size_t gap = size_t(1) << layer, shift = N >> (layer + 1);
            for (size_t offset = 0; offset < rl.size(); offset += 2 * gap) {
                Ciphertext &even = rl[offset], &odd = rl[offset + gap];
                Ciphertext temp;
                negacyclicShift(odd, shift, temp);
                sub(even, temp, odd);
                addInplace(even, temp);
                if (ckks) transformToNttInplace(odd);
                applyGaloisInplace(odd, (uint32_t)((size_t(1) << (layer + 1)) + 1), automorphism_keys);
                if (ckks) transformFromNttInplace(odd);
                addInplace(even, odd);
            }
        }
        Ciphertext ret = std::move(rl[0]);
        fieldTraceInplace(ret, automorphism_keys, l);
        if (ckks) transformToNttInplace(ret);
        return ret;
    }

private:
    troyhip_context *h() const { return c_.handle(); }
    // isScaleWithinBounds (src/evaluator_cuda.cu:32-51)
    static bool scaleWithinBounds(double scale, const ContextData &cd) {
        const int bound = cd.parms().scheme() == SchemeType::ckks ? cd.totalCoeffModulusBitCount() : cd.parms().plainModulus().bitCount();
        return !(scale <= 0 || (int)std::log2(scale) >= bound);
    }
    void plain_addsub(Ciphertext &a, const Plaintext &plain, int sub) const {
        check(troyhip_add_plain(h(), a.raw(), plain.device(), plain.coeffCount(), 0, plain.scale(), sub, 1, nullptr));
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
            b.bind(a);
            a = std::move(b);
        }
    }
    template <class F> void next(Ciphertext &a, F fn) const {
        Ciphertext out;
        out.resize(a.polyModulusDegree(), a.coeffModulusSize() > 1 ? a.coeffModulusSize() - 1 : 1, a.size());
        check(fn(h(), a.raw(), out.raw(), 1, nullptr));
        out.bind(a);
        a = std::move(out);
    }
    // ---- batch plumbing
    // the (data, stride) view of `items` as one batch: in place when they are a run of one slab, else packed into `packed` first
    troyhip_ct runOf(const std::vector<const Ciphertext *> &items, std::vector<Ciphertext> &packed) const {
        for (const Ciphertext *c : items)
            if (!c->sameShape(*items[0])) throw std::invalid_argument("batch: ciphertexts of different shape");
        if (items.size() == 1 || Ciphertext::isRun(items)) return *items[0]->raw();
        packed = Ciphertext::packBatch(items);
        return *packed[0].raw();
    }
    // fn(view, count, head) once per group of consecutive items of equal shape; afterwards every item of a group of two or more is a member of one slab
    template <class F> void inplaceBatch(const std::vector<Ciphertext *> &items, F fn) const {
        for (size_t i = 0; i < items.size();) {
            size_t j = i + 1;
            while (j < items.size() && items[j]->sameShape(*items[i])) j++;
            const size_t count = j - i;
            std::vector<const Ciphertext *> group(items.begin() + (long)i, items.begin() + (long)j);
            if (count == 1 || Ciphertext::isRun(group)) {
                troyhip_ct t = *items[i]->raw();
                fn(&t, count, *items[i]);
                for (size_t b = i; b < j; b++) items[b]->copyMeta(t);
            } else {
                std::vector<Ciphertext> packed = Ciphertext::packBatch(group);
                troyhip_ct t = *packed[0].raw();
                fn(&t, count, *items[i]);
                for (size_t b = 0; b < count; b++) { packed[b].copyMeta(t); *items[i + b] = std::move(packed[b]); }
            }
            i = j;
        }
    }
    std::vector<const uint64_t *> relinKeys(const Ciphertext &head, const RelinKeys &k) const {
        const size_t need = head.size() > 2 ? head.size() - 2 : 0;
        std::vector<const uint64_t *> keys(need ? need : 1, nullptr);
        for (size_t i = 0; i < need; i++) {
            if (!k.hasKey(i + 2)) throw std::invalid_argument("not enough relinearization keys");
            keys[i] = key_of(k, RelinKeys::getIndex(i + 2));
        }
        return keys;
    }
    template <class F> std::vector<Ciphertext> nextBatch(const std::vector<const Ciphertext *> &a, F fn) const {
        if (a.empty()) return {};
        std::vector<Ciphertext> pa;
        const troyhip_ct va = runOf(a, pa);
        std::vector<Ciphertext> out = Ciphertext::allocateBatch(a.size(), *a[0], a[0]->size(), a[0]->coeffModulusSize() > 1 ? a[0]->coeffModulusSize() - 1 : 1);
        troyhip_ct t = *out[0].raw();
        check(fn(h(), &va, &t, a.size(), nullptr));
        for (auto &c : out) c.copyMeta(t);
        return out;
    }
    template <class F> void nextInplaceBatch(const std::vector<Ciphertext *> &items, F fn) const {
        for (size_t i = 0; i < items.size();) {
            size_t j = i + 1;
            while (j < items.size() && items[j]->sameShape(*items[i])) j++;
            std::vector<Ciphertext> out = nextBatch(std::vector<const Ciphertext *>(items.begin() + (long)i, items.begin() + (long)j), fn);
            for (size_t b = i; b < j; b++) *items[b] = std::move(out[b - i]);
            i = j;
        }
    }
    void rotateBatch(const std::vector<Ciphertext *> &items, int steps, int conj, const GaloisKeys &gk) const {
        std::vector<uint32_t> elts;
        std::vector<const uint64_t *> ptrs;
        for (auto &kv : gk.all()) { elts.push_back((uint32_t)(2 * kv.first + 1)); ptrs.push_back(kv.second->get()); }
        inplaceBatch(items, [&](troyhip_ct *v, size_t count, const Ciphertext &) { check(troyhip_rotate(h(), v, steps, conj, elts.data(), ptrs.data(), (int)elts.size(), count, nullptr)); });
    }
    void addsubBatch(const std::vector<Ciphertext *> &acc, const std::vector<const Ciphertext *> &x, bool sub) const {
        if (acc.size() != x.size()) throw std::invalid_argument("Size incorrect.");
        for (size_t i = 0; i < acc.size();) { // groups in which both sides keep one shape
            size_t j = i + 1;
            while (j < acc.size() && acc[j]->sameShape(*acc[i]) && x[j]->sameShape(*x[i])) j++;
            std::vector<Ciphertext> px;
            const troyhip_ct vx = runOf(std::vector<const Ciphertext *>(x.begin() + (long)i, x.begin() + (long)j), px);
            inplaceBatch(std::vector<Ciphertext *>(acc.begin() + (long)i, acc.begin() + (long)j),
                         [&](troyhip_ct *v, size_t count, const Ciphertext &) { check((sub ? troyhip_sub : troyhip_add)(h(), v, &vx, count, nullptr)); });
            i = j;
        }
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
struct Header { bool ntt; size_t size, n, limbs; double scale; uint64_t cf, seed; bool terms; };
inline void put_header(std::ostream &s, const uint64_t *id, const Ciphertext &ct, bool terms) {
    s.write(reinterpret_cast<const char *>(id), 32);
    put<bool>(s, ct.isNttForm()); put<size_t>(s, ct.size()); put<size_t>(s, ct.polyModulusDegree()); put<size_t>(s, ct.coeffModulusSize());
    put<double>(s, ct.scale()); put<uint64_t>(s, ct.correctionFactor()); put<uint64_t>(s, ct.seed()); put<bool>(s, terms);
}
// the data behind the header (src/ciphertext_cuda.cu:26-42): every polynomial -- or, for a seeded ciphertext, c0 alone
inline void put_payload(std::ostream &s, const Ciphertext &ct) {
    if (ct.seed() && ct.size() > 2) throw std::invalid_argument("Seed exists but size is not 2.");
    const std::vector<uint64_t> h = ct.toHost();
    const size_t words = ct.seed() ? ct.coeffModulusSize() * ct.polyModulusDegree() : h.size();
    put<size_t>(s, words);
    put_words(s, h.data(), words);
}
inline Header get_header(std::istream &s, const SEALContext &c) {
    uint64_t id[4], mine[4];
    s.read(reinterpret_cast<char *>(id), 32);
    Header h;
    h.ntt = get<bool>(s); h.size = get<size_t>(s); h.n = get<size_t>(s); h.limbs = get<size_t>(s);
    h.scale = get<double>(s); h.cf = get<uint64_t>(s); h.seed = get<uint64_t>(s); h.terms = get<bool>(s);
    if (h.n != c.polyModulusDegree() || h.limbs < 1 || h.limbs > c.keyLimbs() || h.size < 1 || h.size > max_ct_size) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    check(troyhip_context_parms_id(c.handle(), (int)h.limbs, mine));
    if (!std::equal(id, id + 4, mine)) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    return h;
}
inline void put_header(std::ostream &s, const SEALContext &c, const Ciphertext &ct, bool terms) {
    uint64_t id[4];
    check(troyhip_context_parms_id(c.handle(), (int)ct.coeffModulusSize(), id));
    put_header(s, id, ct, terms);
}
} // namespace wire

inline void Ciphertext::save(std::ostream &stream, const SEALContext &context) const {
    wire::put_header(stream, context, *this, false);
    wire::put_payload(stream, *this);
}
// load(stream, context) (src/ciphertext_cuda.cu:145-190): a seeded blob carries c0 alone, c1 is expanded from the seed; the loaded object is unseeded
inline void Ciphertext::load(std::istream &stream, const SEALContext &context) {
    const wire::Header h = wire::get_header(stream, context);
    if (h.terms) throw std::invalid_argument("Trying to load a termed ciphertext, but indices is not specified");
    if (h.seed && h.size > 2) throw std::invalid_argument("Seed exists but size is not 2.");
    const size_t words = wire::get<size_t>(stream), poly = h.limbs * h.n;
    if (words != (h.seed ? poly : h.size * poly)) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    std::vector<uint64_t> host = wire::get_vector(stream, words, h.seed ? poly : 0);
    if (h.seed) check(troyhip_host_expand_seed(context.handle(), h.seed, (int)h.limbs, host.data() + poly));
    fromHost(host, h.n, h.limbs, h.seed ? 2 : h.size, h.ntt, h.scale, h.cf);
    bind(context);
}
inline void Ciphertext::saveTerms(std::ostream &stream, const SEALContext &context, const Evaluator &evaluator, const std::vector<size_t> &termIds) const {
    std::vector<uint64_t> h;
    if (isNttForm()) {
        Ciphertext copy = *this;
        evaluator.transformFromNttInplace(copy);
        h = copy.toHost();
    } else h = toHost();
    if (seed()) throw std::invalid_argument("Seed is not zero."); // src/ciphertext_cuda.cu:66-68
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
    if (h.seed) throw std::invalid_argument("seed is not zero.");
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
    bind(context);
    if (h.ntt) evaluator.transformToNttInplace(*this);
}
inline void Ciphertext::save(std::ostream &stream) const {
    if (parmsID() == parmsIDZero) throw std::logic_error("the ciphertext has not met a context"); // nothing to put in the parms_id field
    wire::put_header(stream, parmsID().data(), *this, false);
    wire::put_payload(stream, *this);
}
inline void Ciphertext::load(std::istream &stream) { // src/ciphertext_cuda.cu:65-88: no validation without a context
    const wire::CtFields f = wire::get_fields(stream); // sizes no ring of this library has stop here, not in an allocation
    const bool ntt = f.ntt;
    const size_t size = f.size, n = f.n, limbs = f.limbs;
    const double scale = f.scale;
    const uint64_t cf = f.cf;
    if (f.seed) throw std::invalid_argument("seed is not zero.");
    if (f.terms) throw std::invalid_argument("Trying to load a termed ciphertext, but indices is not specified");
    const size_t words = wire::get<size_t>(stream);
    if (words != size * limbs * n) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    const std::vector<uint64_t> host = wire::get_vector(stream, words);
    fromHost(host, n, limbs, size, ntt, scale, cf);
}
inline void Ciphertext::saveTerms(std::ostream &stream, const Evaluator &evaluator, const std::vector<size_t> &termIds) const { saveTerms(stream, evaluator.context(), evaluator, termIds); }
inline void Ciphertext::loadTerms(std::istream &stream, const Evaluator &evaluator, const std::vector<size_t> &termIds) { loadTerms(stream, evaluator.context(), evaluator, termIds); }

} // namespace troyn
