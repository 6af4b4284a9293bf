/* troy_oracle.h -- C ABI of the CPU parity oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar CPU restatement of the reference's CPU path
 * (lightbulb128/troy, src/troy_cpu.h) for the hot path named in SURVEY.md section 8.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product (troy_amd/,
 * include/troyhip.h) never does.
 *
 * Parity status: PINNED -- checked against (a) the known-answer vectors of the reference's own
 * unit tests (tests/golden/kat_*.json, transcribed values from test/utils/{ntt,uintarithsmallmod,
 * rns,galois}.cpp) and (b) outputs of the reference itself compiled here (oracle/_ref), committed
 * as tests/golden/*.npz by tests/golden/gen_golden.py.
 *
 * Buffer conventions are identical to oracle/ref_driver.cpp: a level is named by its limb count,
 * ciphertexts are uint64 [size][limbs][N].
 */
#ifndef TROY_ORACLE_H
#define TROY_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int limbs, size, is_ntt;
    double scale;
    uint64_t correction_factor;
} orc_ct_desc;

enum { ORC_BFV = 1, ORC_CKKS = 2, ORC_BGV = 3 };
enum { ORC_OP_ADD = 0, ORC_OP_SUB, ORC_OP_NEGATE, ORC_OP_MULTIPLY, ORC_OP_SQUARE, ORC_OP_RELIN,
       ORC_OP_MODSWITCH_NEXT, ORC_OP_RESCALE_NEXT, ORC_OP_APPLY_GALOIS, ORC_OP_ROTATE_ROWS,
       ORC_OP_ROTATE_COLUMNS, ORC_OP_ROTATE_VECTOR, ORC_OP_CONJUGATE, ORC_OP_TO_NTT, ORC_OP_FROM_NTT,
       ORC_OP_MULTIPLY_PLAIN_NTT, ORC_OP_ADD_PLAIN, ORC_OP_SUB_PLAIN, ORC_OP_MULTIPLY_PLAIN,
       ORC_OP_APPLY_KEYSWITCH /* CUDA-only API: oracle only */, ORC_OP_NEGACYCLIC_SHIFT /* CUDA-only API: oracle only */ };
enum { ORC_ST_FASTBCONV_MTILDE = 0, ORC_ST_SMMRQ, ORC_ST_FASTFLOOR, ORC_ST_FASTBCONV_SK,
       ORC_ST_DIVROUND_QLAST, ORC_ST_DIVROUND_QLAST_NTT, ORC_ST_MODT_DIV_QLAST };

/* scalar known-answer entry points (src/utils/uintarithsmallmod.h) */
uint64_t orc_barrett_reduce_64(uint64_t x, uint64_t p);
uint64_t orc_barrett_reduce_128(uint64_t lo, uint64_t hi, uint64_t p);
uint64_t orc_multiply_uint_mod(uint64_t a, uint64_t b, uint64_t p);
uint64_t orc_shoup_quotient(uint64_t w, uint64_t p);
uint64_t orc_multiply_uint_mod_lazy(uint64_t x, uint64_t w, uint64_t p);
uint64_t orc_exponentiate_uint_mod(uint64_t a, uint64_t e, uint64_t p);
int orc_try_invert_uint_mod(uint64_t a, uint64_t p, uint64_t *out);
uint64_t orc_dot_product_mod(const uint64_t *a, const uint64_t *b, int n, uint64_t p);
int orc_is_prime(uint64_t p);
int orc_try_minimal_primitive_root(uint64_t degree, uint64_t p, uint64_t *out);
void orc_modulus_const_ratio(uint64_t p, uint64_t *out3);
void orc_naf(int value, int *out, int *n_out);

int orc_get_primes(uint64_t factor, int bits, int count, uint64_t *out);
int orc_coeff_modulus_create(uint64_t N, const int *bits, int n, uint64_t *out);
uint64_t orc_plain_batching(uint64_t N, int bits);

/* transformToNttInplace(Plaintext, parms_id of `limbs`): plain = n_coeffs values mod t -> out [limbs][N] */
int orc_plain_to_ntt(void *h, const uint64_t *plain, int n_coeffs, int limbs, uint64_t *out);

void *orc_create(int scheme, uint64_t N, const uint64_t *primes, int K, uint64_t t);
void orc_destroy(void *h);
const char *orc_last_error(void *h);
int orc_chain(void *h, int *first_limbs, int *last_limbs);
int orc_ntt_tables(void *h, int prime_idx, uint64_t *rop, uint64_t *rquo, uint64_t *iop, uint64_t *iquo,
                   uint64_t *inv_degree2, uint64_t *root);
int orc_behz_bases(void *h, int limbs, uint64_t *bsk_out, uint64_t *gamma);
int orc_bsk_ntt_tables(void *h, int limbs, int idx, uint64_t *rop, uint64_t *iop, uint64_t *inv_degree);
int orc_ntt(void *h, int prime_idx, uint64_t *data, int mode);
/* stand-alone transform for arbitrary (N, p): used by the KAT tests */
int orc_ntt_standalone(uint64_t N, uint64_t p, uint64_t *data, int mode);
int orc_rns_stage(void *h, int limbs, int stage, const uint64_t *in, uint64_t *out);
int orc_set_kswitch_key(void *h, uint32_t which, const uint64_t *data);
int orc_eval(void *h, int op, const orc_ct_desc *ad, const uint64_t *a, const orc_ct_desc *bd, const uint64_t *b,
             int64_t iarg, orc_ct_desc *od, uint64_t *out);
uint32_t orc_galois_elt_from_step(void *h, int step);
void orc_apply_galois(uint64_t N, uint32_t elt, uint64_t p, const uint64_t *in, uint64_t *out);
void orc_apply_galois_ntt(uint64_t N, uint32_t elt, const uint64_t *in, uint64_t *out);

/* cfgA plumbing: deterministic decryption given the secret key [K][N] (NTT form, key level) */
int orc_decrypt(void *h, const uint64_t *sk, const orc_ct_desc *ad, const uint64_t *a, uint64_t *plain_out);

/* CPU baseline ("port"): `count` multiply+relinearize ops spread over `threads` std::threads.
 * returns seconds of wall time; inputs are two size-2 ciphertexts at the first data level. */
double orc_time_mul_relin(void *h, const uint64_t *a, const uint64_t *b, int count, int threads);
double orc_time_ntt(void *h, int prime_idx, const uint64_t *limb, int count);

#ifdef __cplusplus
}
#endif
#endif
