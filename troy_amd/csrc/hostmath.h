// hostmath.h -- host-side precompute for libtroyhip (product code, independent of oracle/).
//
// The reference builds every device constant on the CPU and uploads it when SEALContextCuda is
// constructed (src/context_cuda.cu:5-62, src/utils/rns_cuda.cu:206-269, src/utils/ntt_cuda.cuh:24-29).
// This file reproduces those numbers from the encryption parameters alone: prime selection
// (src/utils/numth.cpp:261-285, src/modulus.cpp:80-121), minimal primitive roots (numth.cpp:335-363),
// root tables in SEAL order (src/utils/ntt.cpp:17-66) and the BEHZ constants (src/utils/rns.cpp:581-803).
#pragma once
#include "modarith.h"
#include <vector>

namespace troyhip {
namespace host {

u64 mul_mod(u64 a, u64 b, u64 p);
u64 pow_mod(u64 a, u64 e, u64 p);
bool inv_mod(u64 a, u64 p, u64 &out);
u64 inv_mod_checked(u64 a, u64 p);
bool is_prime(u64 v);
std::vector<u64> get_primes(u64 factor, int bits, size_t count);
std::vector<u64> coeff_modulus_create(u64 N, const std::vector<int> &bits);
bool minimal_primitive_root(u64 degree, u64 p, u64 &out);
uint32_t reverse_bits(uint32_t x, int bits);
int bit_length_of_product(const std::vector<u64> &v);
u64 product_mod(const std::vector<u64> &v, u64 p);
// BLAKE2b (RFC 7693), unkeyed, outlen bytes (<= 64): the reference's parms_id hash (src/utils/hash.h:25-32 calls blake2b with a
// 32-byte digest over the parameter words, src/encryptionparams.cpp:118-146)
void blake2b(void *out, size_t outlen, const void *in, size_t inlen);
// parms_id of the level holding the first `limbs` primes: hash of (scheme, N, those primes, plain modulus)
void parms_id(int scheme, u64 N, const std::vector<u64> &primes, int limbs, u64 plain_modulus, u64 out[4]);
uint32_t galois_elt_from_step(u64 N, int step);
std::vector<int> naf(int value);

struct NttTable {
    u64 p = 0, psi = 0;
    int logn = 0;
    std::vector<Shoup> root;  // root[bitrev(i)] = psi^i            (ntt.cpp:38-43)
    std::vector<Shoup> iroot; // iroot[bitrev(i-1)+1] = psi^-i      (ntt.cpp:49-54)
    Shoup inv_n{0, 0};
    Shoup iroot_last_scaled{0, 0}; // iroot[N-1] * N^-1 (dwthandler.h:289-330 folds N^-1 into the last stage)
    void build(int logn, u64 p);
};

// One fast-base-conversion (rns.cpp:556-572): mat[o][i] = prod_{k != i} in_k mod out_o
struct BaseConv {
    std::vector<u64> in, out;
    std::vector<u64> inv_punct;            // (prod_{k != i} in_k)^-1 mod in_i
    std::vector<std::vector<u64>> mat;
    void build(const std::vector<u64> &in, const std::vector<u64> &out);
};

// Everything RNSTool::initialize computes for one level (rns.cpp:581-803)
struct RnsLevel {
    std::vector<u64> q, B, Bsk;
    u64 m_sk = 0, gamma = 0, m_tilde = u64(1) << 32, t = 0;
    BaseConv q_to_Bsk, q_to_mtilde, B_to_q, B_to_msk, q_to_tgamma;
    std::vector<u64> prod_B_mod_q, prod_q_mod_Bsk, inv_prod_q_mod_Bsk, inv_mtilde_mod_Bsk;
    u64 inv_prod_B_mod_msk = 0, neg_inv_prod_q_mod_mtilde = 0;
    std::vector<u64> inv_q_last_mod_q; // q_last^-1 mod q_i, i < |q|-1
    u64 inv_q_last_mod_t = 1, q_last_mod_t = 0;
    // decryption (BFV): gamma-correction constants (rns.cpp:748-779)
    u64 inv_gamma_mod_t = 0;
    std::vector<u64> prod_tgamma_mod_q;
    u64 neg_inv_q_mod_t = 0, neg_inv_q_mod_gamma = 0;
    int aux_bits = 61; // size class of B u {m_sk}: 61 = the reference's primes, 58 / 50 = the guard-free / FP64 classes (see build())
    void build(u64 N, const std::vector<u64> &q, u64 t, bool aux_auto = false, const std::vector<u64> &exclude = {});
};

} // namespace host
} // namespace troyhip
