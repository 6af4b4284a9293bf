// ref_binding.hpp -- the reference-side binding of INTEGRATION.md as a header that COMPILES against the reference's own CPU half:
// what a maintainer of lightbulb128/troy would put behind src/*_cuda.cuh to run the evaluator on libtroyhip.so while keeping troy's
// CPU classes (KeyGenerator, Encryptor, Decryptor, encoders, SEALContext bookkeeping) exactly as they are.
//
//     #include "troy_cpu.h"      (the reference, /root/reference/src -- present in the build container only)
//     #include "troyhip.h"       (this repo's C ABI)
//
// TEST INFRASTRUCTURE: compiled and run by tests/test_ref_binding.py in the CPU suite (skipped where /root/reference is absent)
// against oracle/_ref/libtroyref.so (the reference CPU half, built by oracle/Makefile) and the emulator build of the library.  It
// contains no reference code -- it calls the reference's public API -- and the product does not use it (include/troyn.hpp is the
// self-contained mirror).  Each class cites the reference class it stands in for.
#pragma once
#include "troy_cpu.h"
#include "troyhip.h"
#include <algorithm>
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
    default: throw std::runtime_error(m); // "CUDA error." in the reference
    }
}

struct KernelProvider { // src/kernelprovider.cuh:24-33
    static void initialize(int device = 0) { check(troyhip_initialize(device)); }
};

class SEALContext { // SEALContextCuda, src/context_cuda.cuh:146-186: built from the CPU context
public:
    explicit SEALContext(const troy::SEALContext &host) : host_(host) {
        const auto &parms = host.keyContextData()->parms();
        std::vector<uint64_t> q;
        for (auto &m : parms.coeffModulus()) q.push_back(m.value());
        check(troyhip_context_create((int)parms.scheme(), parms.polyModulusDegree(), q.data(), (int)q.size(), parms.plainModulus().value(), &ctx_));
    }
    SEALContext(const SEALContext &) = delete;
    ~SEALContext() { troyhip_context_destroy(ctx_); }
    troyhip_context *handle() const { return ctx_; }
    const troy::SEALContext &host() const { return host_; } // parms_id bookkeeping stays with the CPU object
    // the level with one prime fewer (ContextData::nextContextData)
    troy::ParmsID next(const troy::ParmsID &id) const {
        auto cd = host_.getContextData(id);
        if (!cd || !cd->nextContextData()) throw std::invalid_argument("end of modulus switching chain reached");
        return cd->nextContextData()->parmsID();
    }
private:
    const troy::SEALContext &host_;
    troyhip_context *ctx_ = nullptr;
};

class Ciphertext { // CiphertextCuda, src/ciphertext_cuda.cuh:18-181
public:
    Ciphertext() = default;
    explicit Ciphertext(const troy::Ciphertext &h) { // upload, :20-28
        resize(h.size(), h.coeffModulusSize(), h.polyModulusDegree());
        check(troyhip_copy_h2d(d_.data, h.data(), words() * 8, nullptr));
        d_.is_ntt_form = h.isNttForm();
        d_.scale = h.scale();
        d_.correction_factor = h.correctionFactor();
        parms_id_ = h.parmsID();
    }
    troy::Ciphertext cpu(const SEALContext &c) const { // toHost / cpu()
        troy::Ciphertext h;
        h.resize(c.host(), parms_id_, (size_t)d_.size);
        check(troyhip_copy_d2h(h.data(), d_.data, words() * 8, nullptr));
        h.isNttForm() = d_.is_ntt_form != 0;
        h.scale() = d_.scale;
        h.correctionFactor() = d_.correction_factor;
        return h;
    }
    Ciphertext(const Ciphertext &o) { *this = o; } // deep copy, src/utils/devicearray.cuh:153-164
    Ciphertext &operator=(const Ciphertext &o) {
        if (this == &o) return *this;
        resize((size_t)o.d_.size, (size_t)o.d_.limbs, o.n_);
        check(troyhip_copy_d2d(d_.data, o.d_.data, words() * 8, nullptr));
        d_.is_ntt_form = o.d_.is_ntt_form;
        d_.scale = o.d_.scale;
        d_.correction_factor = o.d_.correction_factor;
        parms_id_ = o.parms_id_;
        return *this;
    }
    Ciphertext(Ciphertext &&o) noexcept { swap(o); }
    Ciphertext &operator=(Ciphertext &&o) noexcept { swap(o); return *this; }
    ~Ciphertext() { troyhip_free(d_.data); }
    void swap(Ciphertext &o) noexcept { std::swap(d_, o.d_); std::swap(n_, o.n_); std::swap(cap_words_, o.cap_words_); std::swap(parms_id_, o.parms_id_); }
    troyhip_ct *raw() { return &d_; }
    const troyhip_ct *raw() const { return &d_; }
    size_t size() const { return (size_t)d_.size; }
    size_t coeffModulusSize() const { return (size_t)d_.limbs; }
    size_t polyModulusDegree() const { return n_; }
    bool isNttForm() const { return d_.is_ntt_form != 0; }
    double scale() const { return d_.scale; }
    troy::ParmsID &parmsID() { return parms_id_; }
    const troy::ParmsID &parmsID() const { return parms_id_; }
    void resize(size_t size, size_t limbs, size_t n) { // capacity stays at max(size, 3) polynomials so that multiply can run in place
        n_ = n;
        const size_t cap = std::max<size_t>(size, 3);
        if (cap * limbs * n > cap_words_) {
            troyhip_free(d_.data);
            d_.data = nullptr;
            check(troyhip_malloc((void **)&d_.data, cap * limbs * n * 8));
            cap_words_ = cap * limbs * n;
        }
        d_.size = (int32_t)size;
        d_.limbs = (int32_t)limbs;
        d_.batch_stride = cap * limbs * n;
    }
private:
    size_t words() const { return (size_t)d_.size * (size_t)d_.limbs * n_; }
    troyhip_ct d_{};
    size_t n_ = 0, cap_words_ = 0;
    troy::ParmsID parms_id_{};
};

// KSwitchKeysCuda (src/kswitchkeys_cuda.cuh:43-56): data()[index] = one device array [K-1][2][K][N] per key, uploaded from the CPU
// object's vector<PublicKey> (one PublicKey per decomposition limb, each a size-2 NTT-form ciphertext at the key level)
class KSwitchKeys {
public:
    KSwitchKeys() = default;
    explicit KSwitchKeys(const troy::KSwitchKeys &h) {
        const auto &all = h.data();
        keys_.resize(all.size(), nullptr);
        for (size_t idx = 0; idx < all.size(); idx++) {
            const std::vector<troy::PublicKey> &kv = all[idx];
            if (kv.empty()) continue;
            const troy::Ciphertext &first = kv[0].data();
            const size_t per = first.size() * first.coeffModulusSize() * first.polyModulusDegree(); // 2 K N words
            check(troyhip_malloc((void **)&keys_[idx], kv.size() * per * 8));
            for (size_t j = 0; j < kv.size(); j++) check(troyhip_copy_h2d(keys_[idx] + j * per, kv[j].data().data(), per * 8, nullptr));
        }
    }
    KSwitchKeys(const KSwitchKeys &) = delete;
    ~KSwitchKeys() { for (uint64_t *p : keys_) troyhip_free(p); }
    bool has(size_t index) const { return index < keys_.size() && keys_[index]; }
    const uint64_t *key(size_t index) const { return keys_[index]; }
    size_t slots() const { return keys_.size(); }
private:
    std::vector<uint64_t *> keys_;
};
class RelinKeys : public KSwitchKeys { // src/relinkeys_cuda.cuh:56-59
public:
    explicit RelinKeys(const troy::RelinKeys &h) : KSwitchKeys(h) {}
    static size_t getIndex(size_t key_power) { return key_power - 2; }
};
class GaloisKeys : public KSwitchKeys { // src/galoiskeys_cuda.cuh:74-77, index (g - 1) / 2 (src/utils/galois_cuda.cuh:45-48)
public:
    explicit GaloisKeys(const troy::GaloisKeys &h) : KSwitchKeys(h) {
        for (size_t i = 0; i < slots(); i++)
            if (has(i)) { elts_.push_back((uint32_t)(2 * i + 1)); ptrs_.push_back(key(i)); }
    }
    const uint32_t *elts() const { return elts_.data(); }
    const uint64_t *const *ptrs() const { return ptrs_.data(); }
    int count() const { return (int)elts_.size(); }
private:
    std::vector<uint32_t> elts_;
    std::vector<const uint64_t *> ptrs_;
};

class Evaluator { // EvaluatorCuda, src/evaluator_cuda.cuh:13-361 -- every method const
public:
    explicit Evaluator(const SEALContext &c) : c_(c) {}
    void addInplace(Ciphertext &a, const Ciphertext &b) const { check(troyhip_add(c_.handle(), a.raw(), b.raw(), 1, nullptr)); }   // :37-41
    void multiplyInplace(Ciphertext &a, const Ciphertext &b) const {                                                              // :85-97
        check(troyhip_multiply(c_.handle(), a.raw(), b.raw(), a.raw(), 1, nullptr));
    }
    void multiply(const Ciphertext &a, const Ciphertext &b, Ciphertext &d) const { d = a; multiplyInplace(d, b); }
    void relinearizeInplace(Ciphertext &a, const RelinKeys &k) const {                                                             // :111-115
        check(troyhip_relinearize(c_.handle(), a.raw(), k.key(RelinKeys::getIndex(2)), 1, nullptr));
    }
    void rescaleToNextInplace(Ciphertext &a) const { next(a, troyhip_rescale_to_next); }                                           // :186-190
    void modSwitchToNextInplace(Ciphertext &a) const { next(a, troyhip_mod_switch_to_next); }                                      // :131-135
    void rotateRowsInplace(Ciphertext &a, int steps, const GaloisKeys &gk) const {                                                 // :276-285
        check(troyhip_rotate(c_.handle(), a.raw(), steps, 0, gk.elts(), gk.ptrs(), gk.count(), 1, nullptr));
    }
    void rotateVectorInplace(Ciphertext &a, int steps, const GaloisKeys &gk) const { rotateRowsInplace(a, steps, gk); }            // :313-321
    void rotateVector(const Ciphertext &a, int steps, const GaloisKeys &gk, Ciphertext &d) const { d = a; rotateVectorInplace(d, steps, gk); }
    void rotateRows(const Ciphertext &a, int steps, const GaloisKeys &gk, Ciphertext &d) const { d = a; rotateRowsInplace(d, steps, gk); }
    void rotateColumnsInplace(Ciphertext &a, const GaloisKeys &gk) const {                                                         // :295-304
        check(troyhip_rotate(c_.handle(), a.raw(), 0, 1, gk.elts(), gk.ptrs(), gk.count(), 1, nullptr));
    }
private:
    template <class F> void next(Ciphertext &a, F fn) const {
        Ciphertext out;
        out.resize(a.size(), a.coeffModulusSize() - 1, a.polyModulusDegree());
        check(fn(c_.handle(), a.raw(), out.raw(), 1, nullptr));
        out.parmsID() = c_.next(a.parmsID());
        a.swap(out);
    }
    const SEALContext &c_;
};

} // namespace troyn
