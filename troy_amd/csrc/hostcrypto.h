// hostcrypto.h -- CPU-side key generation / encryption / decryption (see hostcrypto.cpp).
#pragma once
#include "context.h"

namespace troyhip {
namespace hostcrypto {

class Rng { // ChaCha20 keystream keyed by a 128-bit seed; `stream` (the 64-bit nonce) separates the uses of one seed
public:
    Rng(u64 seed_lo, u64 seed_hi, u64 stream = 0);
    uint32_t next32();
    u64 next64();
    u64 uniform_below(u64 bound);
private:
    void refill();
    uint32_t state[16], block[16];
    int pos;
};

void ntt_forward(u64 *a, const host::NttTable &t);
void ntt_inverse(u64 *a, const host::NttTable &t);
void keygen_secret(const Context &c, Rng &rng, u64 *sk);                                   // [K][N] NTT form
void keygen_public(const Context &c, Rng &rng, const u64 *sk, u64 *pk);                    // [2][K][N] NTT form
void keygen_kswitch(const Context &c, Rng &rng, const u64 *sk, const u64 *new_key, u64 *out); // [K-1][2][K][N]
void relin_source(const Context &c, const u64 *sk, u64 *out);
void galois_source(const Context &c, const u64 *sk, uint32_t elt, u64 *out);
void encrypt(const Context &c, Rng &rng, const u64 *pk, const u64 *plain, size_t n_coeffs, int limbs, u64 *ct);
void encrypt_symmetric(const Context &c, Rng &rng, const u64 *sk, const u64 *plain, size_t n_coeffs, int limbs, u64 *ct);
void encrypt_zero(const Context &c, Rng &rng, const u64 *pk, int limbs, u64 *ct);           // [2][limbs][N]: any data level
void encrypt_zero_symmetric(const Context &c, Rng &rng, const u64 *sk, int limbs, u64 *ct);
// the seeded form (src/utils/rlwe_cuda.cu:262-330, src/ciphertext_cuda.cu:145-190): c1 is a function of a 64-bit seed alone -- expand_seed writes it in the form
// the ciphertext stores it (NTT form for CKKS, coefficient form otherwise) -- so a fresh symmetric ciphertext travels as (seed, c0)
void expand_seed(const Context &c, u64 a_seed, int limbs, u64 *c1);
void encrypt_symmetric_seeded(const Context &c, Rng &rng, u64 a_seed, const u64 *sk, const u64 *plain, size_t n_coeffs, int limbs, u64 *ct); // plain == nullptr: zero
void decrypt(const Context &c, const u64 *sk, const u64 *ct, int size, int limbs, bool is_ntt, u64 correction_factor, u64 *out);
// BatchEncoder (src/batchencoder.cpp:61-190): `count` <= N slot values modulo t <-> the plaintext polynomial [N] (coefficient form)
void batch_encode(const Context &c, const u64 *values, size_t count, u64 *plain);
void batch_decode(const Context &c, const u64 *plain, size_t n_coeffs, u64 *values);

} // namespace hostcrypto
} // namespace troyhip
