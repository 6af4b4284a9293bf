// evaluator.h -- batched homomorphic operations on raw device buffers: the MI355X counterpart of
// EvaluatorCuda (src/evaluator_cuda.cu).  Every op works on a batch of B independent ciphertexts that
// share one shape (size, level, form); B = 1 is the reference's per-ciphertext call.  All launches go
// to the caller's stream; nothing here synchronises.
#pragma once
#include "context.h"

namespace troyhip {

// A batch of ciphertexts: item b, polynomial i, limb l, coefficient n lives at
//   data[b * bstride + (i * limbs + l) * N + n]        (the reference's [size][limb][N] layout,
// src/ciphertext_cuda.cuh:62-67, with an explicit batch stride)
struct CtBatch {
    u64 *data = nullptr;
    u64 bstride = 0;
    int size = 0, limbs = 0;
    bool ntt = false;
    double scale = 1.0;
    u64 cf = 1; // BGV correction factor
};

// Key-switching key in HBM: [K-1 decomposition limbs][2 components][K limbs][N], NTT form
// (the reference keeps one allocation per decomposition limb, src/kswitchkeys_cuda.cuh:43-56)
struct KsKey {
    const u64 *data = nullptr;
};

class Evaluator {
public:
    explicit Evaluator(Context &ctx) : c(ctx) {}
    Context &c;

    void add_sub(CtBatch &a, const CtBatch &b, u64 batch, bool sub, hipStream_t s);
    // plaintext operands: BFV/BGV plain = n_coeffs coefficients mod t per item; CKKS add/sub: [limbs][N] NTT-form rows.
    // plain_bstride = 0 broadcasts one plaintext to the whole batch
    void add_plain(CtBatch &ct, const u64 *plain, u64 n_coeffs, u64 plain_bstride, double plain_scale, bool sub, u64 batch, hipStream_t s);
    void multiply_plain(CtBatch &ct, const u64 *plain, u64 n_coeffs, u64 plain_bstride, u64 batch, hipStream_t s);
    // DecryptorCuda::decrypt (decryptor_cuda.cu:61-330): sk [K][N] NTT form (device); out: BFV/BGV N coefficients mod t per item
    // (stride out_bstride), CKKS the RNS plaintext [limbs][N] (NTT form)
    void decrypt(const CtBatch &ct, const u64 *sk, u64 *out, u64 out_bstride, u64 batch, hipStream_t s);
    // applyKeySwitchingInplace (evaluator_cuda.cu:1365-1378) and negacyclicShift (evaluator_cuda.cu:2342-2351)
    void apply_key_switching(CtBatch &ct, const KsKey &key, u64 batch, hipStream_t s);
    void negacyclic_shift(CtBatch &ct, u64 shift, u64 batch, hipStream_t s);
    void divide_by_degree(CtBatch &ct, u64 mul, u64 batch, hipStream_t s);
    void plain_to_ntt(const u64 *plain, u64 n_coeffs, u64 plain_bstride, int limbs, u64 *out, u64 count, hipStream_t s);
    void negate(CtBatch &a, u64 batch, hipStream_t s);
    // out may alias a or b; out.size/limbs/... are set; out.data/out.bstride are the caller's
    void multiply(const CtBatch &a, const CtBatch &b, CtBatch &out, u64 batch, hipStream_t s);
    // base != nullptr: ct is to be taken as (base, 0) -- polynomial 0 from base[b * base_bstride ..], polynomial 1 zero -- whatever it holds;
    // base_polys == 2: as (base[b][0], base[b][1])
    void switch_key(CtBatch &ct, const u64 *target, u64 t_bstride, const KsKey &key, u64 batch, hipStream_t s, const u64 *base = nullptr, u64 base_bstride = 0, int base_polys = 1);
    // relinearize (not in place): size 3 -> 2 reads the operand where it lies and writes out; larger sizes copy and run in place
    void relinearize_to(const CtBatch &in, CtBatch &out, const KsKey *keys, int n_keys, u64 batch, hipStream_t s);
    void relinearize(CtBatch &ct, const KsKey &key, u64 batch, hipStream_t s);
    void relinearize(CtBatch &ct, const KsKey *keys, int n_keys, u64 batch, hipStream_t s); // keys[i]: relin key of index i (power i + 2)
    void mod_switch_to_next(const CtBatch &in, CtBatch &out, u64 batch, hipStream_t s);
    void rescale_to_next(const CtBatch &in, CtBatch &out, u64 batch, hipStream_t s);
    void apply_galois(CtBatch &ct, uint32_t elt, const KsKey &key, u64 batch, hipStream_t s);
    void transform_to_ntt(CtBatch &ct, u64 batch, hipStream_t s);
    void transform_from_ntt(CtBatch &ct, u64 batch, hipStream_t s);
    void multiply_plain_ntt(CtBatch &ct, const u64 *plain, double plain_scale, u64 batch, hipStream_t s);
    // out = sum_i cts[i] (x) plains[i]: the multiplyPlain + addInplace loop of a linear layer as one pass (NTT-form operands of one level)
    void multiply_plain_accumulate(const CtBatch *cts, const u64 *const *plains, int count, double plain_scale, CtBatch &out, u64 batch, hipStream_t s);

    // scratch words needed by the ops (so callers can pre-reserve outside timed regions)
    size_t scratch_multiply(int sa, int sb, int limbs, u64 batch) const;
    size_t scratch_switch_key(int limbs, u64 batch) const;

private:
    void check_ct(const CtBatch &a) const;
    bool scale_ok(double scale, int limbs) const;
    void mod_switch_scale(const CtBatch &in, CtBatch &out, u64 batch, hipStream_t s);
    void balance_correction(u64 f1, u64 f2, u64 &f, u64 &e1, u64 &e2) const;
    void scalar_mul(CtBatch &a, u64 scalar, u64 batch, hipStream_t s);
};

} // namespace troyhip
