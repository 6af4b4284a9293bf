/* troyhip.h -- C ABI of libtroyhip.so, the MI355X (gfx950) evaluator that drops in under the
 * troyn:: API of lightbulb128/troy (src/troy_cuda.cuh).
 *
 * The reference has no FFI seam: troyn:: is a set of aliases over CUDA classes compiled into
 * libtroy.so (src/troy_cuda.cuh:20-43).  This header is the seam a maintainer binds instead: each
 * entry point names the reference member function(s) it replaces.  INTEGRATION.md shows the C++ side.
 *
 * Conventions
 *  - every function returns a status (0 = ok); troyhip_last_error() returns the message of the last
 *    failure on the calling thread.  Status values mirror the reference's exception classes
 *    (SURVEY.md 8b): 1 std::invalid_argument, 2 std::logic_error, 3 std::out_of_range,
 *    4 std::runtime_error ("CUDA error." in the reference), 5 "KernelProvider not initialized."
 *  - all ciphertext data are device pointers to uint64 in the reference's layout
 *    [size][coeff_modulus_size][poly_modulus_degree] (src/ciphertext_cuda.cuh:62-67); a batch of B
 *    independent ciphertexts of identical shape is addressed with an explicit batch stride.
 *  - `stream` is a hipStream_t (NULL = default stream).  No call synchronises unless it says so.
 *  - the caller owns every buffer; the library owns the context tables and one grow-only scratch
 *    arena per context (ops on one context must be issued in stream order).
 */
#ifndef TROYHIP_H
#define TROYHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { TROYHIP_OK = 0, TROYHIP_INVALID_ARGUMENT = 1, TROYHIP_LOGIC_ERROR = 2, TROYHIP_OUT_OF_RANGE = 3,
       TROYHIP_RUNTIME_ERROR = 4, TROYHIP_NOT_INITIALIZED = 5 };
enum { TROYHIP_BFV = 1, TROYHIP_CKKS = 2, TROYHIP_BGV = 3 }; /* SchemeType, src/encryptionparams.h */

typedef struct troyhip_context troyhip_context;

/* CiphertextCuda metadata + data pointer (src/ciphertext_cuda.cuh:251-267) */
typedef struct {
    uint64_t *data;            /* device */
    uint64_t batch_stride;     /* uint64 words between consecutive ciphertexts of the batch */
    int32_t size;              /* number of polynomials */
    int32_t limbs;             /* coeffModulusSize(): identifies the level (parms_id) */
    int32_t is_ntt_form;
    double scale;              /* CKKS */
    uint64_t correction_factor;/* BGV */
} troyhip_ct;

typedef struct {
    int32_t scheme;
    uint64_t poly_modulus_degree;
    int32_t key_limbs;         /* K: coeff_modulus size at the key level */
    int32_t first_limbs;       /* limbs of a fresh ciphertext (firstParmsID) */
    int32_t last_limbs;        /* limbs at lastParmsID */
    uint64_t plain_modulus;
} troyhip_context_info_t;

/* ---- KernelProvider (src/kernelprovider.cuh:24-85) ---- */
int troyhip_initialize(int device);                 /* KernelProvider::initialize (cudaSetDevice(0) there) */
int troyhip_is_initialized(void);
/* More than one GPU in one process (no reference counterpart: KernelProvider::initialize is cudaSetDevice(0), src/kernelprovider.cuh:29-33).  HIP's current
 * device is PER HOST THREAD.  A context belongs to the device that was current on the thread that created it; every entry point that takes a context first
 * makes that device the calling thread's current one and leaves it so (tables, scratch, launches, the caching pool and its streams all follow the current
 * device).  So: troyhip_initialize once; troyhip_set_device(d) + troyhip_context_create for each device; then either a host thread per device or one thread
 * walking over the contexts.  troyhip_malloc allocates on the calling thread's current device; troyhip_free returns a block to the device it came from.
 * A stream (troyhip_stream_create) belongs to the device that was current when it was made and must be used with contexts of that device.
 * troyhip_copy_peer moves bytes between two devices (hipMemcpyPeerAsync over xGMI; same device: an ordinary device copy) on `stream`, a stream of the
 * calling thread's current device. */
int troyhip_device_count(int *count);
int troyhip_set_device(int device);
int troyhip_get_device(int *device);
int troyhip_context_device(const troyhip_context *ctx, int *device);  /* -1 for a host-only context */
int troyhip_copy_peer(void *dst, int dst_device, const void *src, int src_device, size_t bytes, void *stream);
const char *troyhip_last_error(void);
const char *troyhip_build_info(void);               /* "gfx950" for the product build */
int troyhip_malloc(void **out, size_t bytes);       /* KernelProvider::malloc */
int troyhip_free(void *p);                          /* KernelProvider::free */
/* malloc/free go through a caching pool with the reference's MemoryPoolCuda policy (src/utils/memorypool_cuda.cuh:40-58): a freed
 * block serves a later request of size <= block <= 2 * size.  Reuse is ordered against the default stream, every stream made by
 * troyhip_stream_create and every stream announced with troyhip_stream_register: a block freed while work on one of THOSE streams still
 * uses it is not handed out before that work has passed.  A stream made elsewhere (a caller-created hipStream_t, a torch / RCCL stream) is
 * announced automatically the first time any entry point of this header is handed it; work the CALLER launches on such a stream before the
 * library has ever seen it is not covered -- troyhip_stream_register it first, or synchronise it before troyhip_free.  A stream the pool
 * knows -- registered explicitly OR announced automatically -- must be released with troyhip_stream_unregister BEFORE its owner destroys it:
 * every later troyhip_free records an event on each known stream, and a destroyed hipStream_t is a dangling handle (the pool drops a stream
 * whose record fails, but whether the runtime fails cleanly on a dead handle is the runtime's business, not a guarantee of this interface).
 * troyhip_pool_release synchronises the device and returns every cached block to the driver. */
int troyhip_pool_release(void);
int troyhip_copy_h2d(void *dst, const void *src, size_t bytes, void *stream);   /* KernelProvider::copy */
int troyhip_copy_d2h(void *dst, const void *src, size_t bytes, void *stream);   /* KernelProvider::retrieve */
int troyhip_copy_d2d(void *dst, const void *src, size_t bytes, void *stream);   /* KernelProvider::copyOnDevice */
int troyhip_memset_zero(void *dst, size_t bytes, void *stream);                 /* KernelProvider::memsetZero */
int troyhip_stream_synchronize(void *stream);
/* every operation takes a stream (NULL = the default stream); one context may only be driven from ONE stream at a time (its
 * scratch arena is not shared between concurrent operations) -- use one context per stream to overlap independent batches */
int troyhip_stream_create(void **stream);
int troyhip_stream_destroy(void *stream);
/* a stream created elsewhere that will run work on buffers of troyhip_malloc: the pool then orders reuse against it too;
 * unregister (synchronises the stream) before destroying it.  No reference counterpart (the reference has one stream). */
int troyhip_stream_register(void *stream);
int troyhip_stream_unregister(void *stream);
int troyhip_mem_info(size_t *free_bytes, size_t *total_bytes);
/* "0000:c1:00.0"-style PCI address of a device (hipDeviceGetPCIBusId): bench.py reports it per rank and binds the rank's host thread to the
 * device's NUMA node (one process per GPU) */
int troyhip_device_pci_bus_id(int device, char *out, size_t capacity);
/* HIP-event timers on the caller's stream (bench.py roofline measurement) */
/* TEST SUPPORT: runs one primitive of the device arithmetic (kernelutils.cuh:94-404 counterparts in modarith.h / bfly.h) on n device
 * operands.  op: 0 barrett64(a), 1 barrett128(a = lo, b = hi), 2 mulmod(a, b), 3 mul_shoup(a, w = b, quotient c), 4 mul_lazy (result
 * in [0, 2p)), 5 reduce_prod(a * b), 6 / 7 / 11 / 12 forward butterflies (guarded, guard-free, SGPR-twiddle forms; X = a, Y = b,
 * twiddle c; out = canonical X', Y' interleaved), 8 / 13 inverse butterflies, 9 last inverse stage (aux = N^-1), 10 128-bit MAC */
int troyhip_test_modarith(int op, const uint64_t *a, const uint64_t *b, const uint64_t *c, uint64_t p, uint64_t aux, uint64_t *out, uint64_t n, void *stream);
/* per-kernel timing: while enabled every kernel launch is bracketed by HIP events on its own stream; the report is JSON text
 * [{"name", "calls", "total_us"}, ...] in first-launch order and clears the log (bench.py: roofline.per_kernel) */
/* path counters ("ks_fp_launches", "ks_int_launches", "ntt1_fp_launches", "ntt1_int_launches", "ntt2_fp_launches", "ntt2_int_launches", "behz_fp_launches",
 * "behz_mfma_launches", "behz_valu_launches", "ntt2_wide_launches"): which kernel class the launchers chose so far in
 * this process -- the parity tests read them so that a test of the FP64 instances cannot pass on the integer kernels unnoticed.  No
 * reference counterpart (test / diagnostics support, like troyhip_ktime_*). */
int troyhip_stat(const char *name, uint64_t *value);
/* first 16 hex digits of the SHA-256 over the library's sources (troy_amd/csrc/Makefile): ties a measurement file (profiles/ *_traffic.json) to the
 * build it was taken on; bench.py reports `traffic: null` when the two differ */
const char *troyhip_build_id(void);
int troyhip_ktime_enable(int on);
int troyhip_ktime_report(char *out, size_t capacity);
int troyhip_timer_create(void **timer);
int troyhip_timer_destroy(void *timer);
int troyhip_timer_start(void *timer, void *stream);
int troyhip_timer_stop(void *timer, void *stream);
int troyhip_timer_elapsed_ms(void *timer, float *ms); /* synchronises on the stop event */

/* ---- parameters (src/modulus.h:485,528; src/modulus.cpp:80-121) ---- */
int troyhip_coeff_modulus_create(uint64_t poly_modulus_degree, const int *bit_sizes, int count, uint64_t *out);
int troyhip_plain_modulus_batching(uint64_t poly_modulus_degree, int bit_size, uint64_t *out);

/* ---- SEALContextCuda (src/context_cuda.cuh:146-186, src/context_cuda.cu:5-62) ----
 * Builds every table the reference computes on the CPU and uploads (NTT roots, BEHZ constants,
 * mod-switch factors) for the whole modulus-switching chain; SecurityLevel::none semantics. */
int troyhip_context_create(int scheme, uint64_t poly_modulus_degree, const uint64_t *coeff_modulus, int coeff_modulus_size,
                           uint64_t plain_modulus, troyhip_context **out);
/* the same tables on the host only (no GPU touched, no troyhip_initialize needed): enough for the troyhip_host_* calls */
int troyhip_context_create_host(int scheme, uint64_t poly_modulus_degree, const uint64_t *coeff_modulus, int coeff_modulus_size,
                                uint64_t plain_modulus, troyhip_context **out);
int troyhip_context_destroy(troyhip_context *ctx);
int troyhip_context_info(const troyhip_context *ctx, troyhip_context_info_t *out);
/* BEHZ auxiliary base of a level (RNSToolCuda, src/utils/rns_cuda.cu:206-269): bsk_out gets |Bsk| primes (B then m_sk) */
int troyhip_context_behz_bases(const troyhip_context *ctx, int limbs, uint64_t *bsk_out, int *bsk_size, uint64_t *gamma);
/* host copy of the NTT tables of one prime of the context (NTTTablesCuda, src/utils/ntt_cuda.cuh:81-100); each array N entries */
int troyhip_context_ntt_tables(const troyhip_context *ctx, uint64_t prime, uint64_t *root_operand, uint64_t *root_quotient,
                               uint64_t *inv_root_operand, uint64_t *inv_root_quotient, uint64_t *inv_degree2, uint64_t *root);
/* BLAKE2b (RFC 7693), unkeyed, 1..64 output bytes: the hash behind parms_id (src/utils/hash.h) */
int troyhip_blake2b(void *out, size_t outlen, const void *in, size_t inlen);
/* parms_id of the level with `limbs` primes: BLAKE2b-256 of (scheme, N, primes, plain modulus), src/encryptionparams.cpp:118-146 */
int troyhip_context_parms_id(const troyhip_context *ctx, int limbs, uint64_t out[4]);
/* A context (its scratch arena) is owned by the stream of the first operation that uses it; an operation on ANOTHER stream is refused
 * with TROYHIP_LOGIC_ERROR until the owner is released -- call this after synchronising the owning stream.  One context per stream is
 * the intended use (bench.py: one context per lane). */
int troyhip_context_release_stream(troyhip_context *ctx);
int troyhip_context_reserve_scratch(troyhip_context *ctx, size_t words);
int troyhip_context_scratch_words(const troyhip_context *ctx, int op, int limbs, uint64_t batch, size_t *words); /* op: 0 multiply(2x2), 1 switch_key */
int troyhip_galois_elt_from_step(const troyhip_context *ctx, int step, uint32_t *out);   /* GaloisToolCuda::getEltFromStep (galois_cuda.cu:44-86) */

/* ---- CPU-side KeyGenerator / Encryptor / Decryptor (BASELINE config A plumbing; the reference runs these on the CPU too:
 * src/keygenerator_cuda.cuh delegates to src/keygenerator.cpp).  ALL buffers are host memory.  Randomness: a ChaCha20
 * stream keyed by (seed_lo, seed_hi) -- fresh ciphertexts are not bit-identical to the reference's Blake2xb-driven ones
 * but decrypt to the same plaintext; decryption is deterministic and bit-exact. ----
 * secret_key [K][N] (NTT form, src/secretkey.h), public_key [2][K][N] (src/publickey.h), kswitch keys [K-1][2][K][N]. */
/* n bytes from the operating system's entropy source (getrandom(2)); the mirrors seed KeyGenerator / Encryptor from it when the caller
 * passes no seed, as the reference seeds its PRNG factory from std::random_device (src/randomgen.cpp:23,72) */
int troyhip_random_bytes(void *out, size_t n);
int troyhip_host_keygen(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, uint64_t *secret_key, uint64_t *public_key); /* KeyGenerator(ctx), createPublicKey */
int troyhip_host_relin_key(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, const uint64_t *secret_key, uint64_t *out);  /* createRelinKeys */
int troyhip_host_galois_key(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, const uint64_t *secret_key, uint32_t galois_elt, uint64_t *out); /* createGaloisKeys({elt}) */
/* Encryptor::encryptZero(parms_id) / encryptZeroSymmetric(parms_id) (src/encryptor.cpp:88-150, src/encryptor_cuda.cuh:170-320): an encryption of
 * zero at ANY data level (`limbs` primes) of any scheme; key = the public key (symmetric == 0) or the secret key; ct_out [2][limbs][N], NTT form
 * for CKKS and coefficient form otherwise, scale 1, correction factor 1 */
int troyhip_host_encrypt_zero(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, const uint64_t *key, int symmetric, int limbs, uint64_t *ct_out);
/* KeyGenerator::createKeySwitchingKeys(new_key) (src/keygenerator.cpp:294-329, 360-366): the key that takes a ciphertext under `new_key`
 * ([key_limbs][N], NTT form, as troyhip_host_keygen writes a secret key) to one under `secret_key`; layout of `out` as troyhip_host_relin_key */
int troyhip_host_kswitch_key(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, const uint64_t *secret_key, const uint64_t *new_key, uint64_t *out);
/* Encryptor::encrypt (public key).  BFV/BGV: plain = n_coeffs coefficients mod t -> ct [2][first_limbs][N];  CKKS: plain = [limbs][N] NTT-form RNS
 * polynomial -> ct [2][limbs][N] NTT form */
int troyhip_host_encrypt(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, const uint64_t *public_key, const uint64_t *plain,
                         uint64_t n_coeffs, int limbs, uint64_t *ct_out);
/* Encryptor::encryptSymmetric (secret key; src/encryptor.cpp:88-148 with is_asymmetric false, src/utils/rlwe.cpp:234-345): same
 * operand and result layout as troyhip_host_encrypt, (c0, c1) = (-(a s + e) + message, a) sampled at the plaintext's own level */
int troyhip_host_encrypt_symmetric(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, const uint64_t *secret_key, const uint64_t *plain,
                                   uint64_t n_coeffs, int limbs, uint64_t *ct_out);
/* The seeded form of a fresh symmetric ciphertext (src/utils/rlwe_cuda.cu:262-330 sets CiphertextCuda::seed(); src/ciphertext_cuda.cu:26-35 saves c0 alone;
 * :145-190 load(stream, context) regenerates c1): c1 is a function of the public 64-bit `a_seed` (non-zero) alone.  plain == NULL: an encryption of zero
 * at the level of `limbs` primes (encryptZeroSymmetric); otherwise operands and result as troyhip_host_encrypt_symmetric.  troyhip_host_expand_seed
 * writes c1 [limbs][N] in the form the ciphertext stores it (NTT form for CKKS, coefficient form otherwise).  The wire format is the reference's; the
 * seed -> c1 expansion is this library's (ChaCha20; the reference's is curand's XORWOW, which exists only inside that library). */
int troyhip_host_encrypt_symmetric_seeded(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, uint64_t a_seed, const uint64_t *secret_key,
                                          const uint64_t *plain, uint64_t n_coeffs, int limbs, uint64_t *ct_out);
int troyhip_host_expand_seed(const troyhip_context *ctx, uint64_t a_seed, int limbs, uint64_t *c1_out);
/* Decryptor::decrypt.  BFV/BGV: N plaintext coefficients;  CKKS: the [limbs][N] RNS plaintext (NTT form) */
int troyhip_host_decrypt(const troyhip_context *ctx, const uint64_t *secret_key, const uint64_t *ct, int size, int limbs, int is_ntt_form,
                         uint64_t correction_factor, uint64_t *plain_out);
/* BatchEncoder::encode / decode (src/batchencoder.cpp:84-190; CUDA twin src/batchencoder_cuda.cu): `count` <= N slot values modulo t in the reference's
 * 2 x (N/2) matrix order <-> the plaintext polynomial [N] in coefficient form.  BFV / BGV with a batching plain modulus (t prime, t = 1 mod 2N). */
int troyhip_host_batch_encode(const troyhip_context *ctx, const uint64_t *values, uint64_t count, uint64_t *plain_out);
int troyhip_host_batch_decode(const troyhip_context *ctx, const uint64_t *plain, uint64_t n_coeffs, uint64_t *values_out);

/* ---- kernel_util (src/kernelutils.cuh:562-672) ----
 * rows limb-polynomials of N coefficients at `data`; row r is reduced modulo row_primes[(r / inner) % period].
 * kNttNegacyclicHarvey / kInverseNttNegacyclicHarvey: inputs are residues in [0,p) (what every operation of this library stores; the
 * kernels tolerate lazy values below 2p), outputs canonical in [0,p). */
int troyhip_ntt(troyhip_context *ctx, uint64_t *data, uint64_t rows, const uint64_t *row_primes, int period, int inner, int inverse, void *stream);
/* synthetic uniform residues (bench / tests): value(row, n) = splitmix64(seed ^ (row0+row)*C, n) mod p_row */
int troyhip_fill_uniform(troyhip_context *ctx, uint64_t *data, uint64_t rows, const uint64_t *row_primes, int period, int inner,
                         uint64_t seed, uint64_t row0, void *stream);

/* ---- EvaluatorCuda (src/evaluator_cuda.cuh:23-349) ---- all ops take a batch of `batch` ciphertexts */
int troyhip_negate(troyhip_context *ctx, troyhip_ct *a, uint64_t batch, void *stream);                       /* negateInplace */
int troyhip_add(troyhip_context *ctx, troyhip_ct *a, const troyhip_ct *b, uint64_t batch, void *stream);     /* addInplace */
int troyhip_sub(troyhip_context *ctx, troyhip_ct *a, const troyhip_ct *b, uint64_t batch, void *stream);     /* subInplace */
/* multiply / multiplyInplace / square: out->data, out->batch_stride are inputs (may alias a or b), the rest of *out is set */
int troyhip_multiply(troyhip_context *ctx, const troyhip_ct *a, const troyhip_ct *b, troyhip_ct *out, uint64_t batch, void *stream);
/* key-switching keys: device array [K-1][2][K][N] (NTT form): KSwitchKeysCuda::data()[index] (src/kswitchkeys_cuda.cuh:43-56) */
int troyhip_relinearize(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *relin_key, uint64_t batch, void *stream);      /* relinearizeInplace, size 3 */
/* relinearizeInplace from any size <= 16 (relinearizeInternal, src/evaluator_cuda.cu:703-744): relin_keys[i] = device key of index i
 * (RelinKeys::getIndex(i + 2)), n_keys >= size - 2; the step sequence is the reference's, see evaluator.cpp */
int troyhip_relinearize_keys(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *const *relin_keys, int n_keys, uint64_t batch, void *stream);
/* relinearize(encrypted, relin_keys, destination) (src/evaluator_cuda.cuh: copy + relinearizeInplace): out->data / out->batch_stride are inputs
 * (a distinct buffer of at least 2 polynomials per item; at least in->size for in->size > 3), the rest of *out is set.  From size 3 the operand is
 * read where it lies (c2 as the key-switch target, c0 / c1 as what the result is accumulated onto): no copy of the operand is made. */
int troyhip_relinearize_to(troyhip_context *ctx, const troyhip_ct *in, troyhip_ct *out, const uint64_t *const *relin_keys, int n_keys, uint64_t batch, void *stream);
int troyhip_switch_key(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *target, uint64_t target_batch_stride,
                       const uint64_t *kswitch_key, uint64_t batch, void *stream);                            /* applyKeySwitchingInplace / switchKeyInplace */
int troyhip_mod_switch_to_next(troyhip_context *ctx, const troyhip_ct *in, troyhip_ct *out, uint64_t batch, void *stream);   /* modSwitchToNext */
int troyhip_rescale_to_next(troyhip_context *ctx, const troyhip_ct *in, troyhip_ct *out, uint64_t batch, void *stream);      /* rescaleToNext */
int troyhip_apply_galois(troyhip_context *ctx, troyhip_ct *ct, uint32_t galois_elt, const uint64_t *galois_key, uint64_t batch, void *stream); /* applyGaloisInplace */
/* rotateRows / rotateVector (steps != 0), rotateColumns / complexConjugate (steps == 0 and conjugate != 0).
 * key_elts/keys: the Galois keys the caller holds; a missing key is decomposed by NAF as the reference does
 * (evaluator_cuda.cu:2119-2176). */
int troyhip_rotate(troyhip_context *ctx, troyhip_ct *ct, int steps, int conjugate, const uint32_t *key_elts, const uint64_t *const *keys,
                   int n_keys, uint64_t batch, void *stream);
int troyhip_transform_to_ntt(troyhip_context *ctx, troyhip_ct *ct, uint64_t batch, void *stream);            /* transformToNttInplace(Ciphertext) */
int troyhip_transform_from_ntt(troyhip_context *ctx, troyhip_ct *ct, uint64_t batch, void *stream);          /* transformFromNttInplace */
int troyhip_multiply_plain_ntt(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *plain, double plain_scale, uint64_t batch, void *stream); /* multiplyPlainInplace, NTT-form operands */
/* out = sum_i cts[i] (x) plains[i], 1 <= count <= 16, NTT-form operands of one level and one scale; `out` is a caller-allocated batch
 * (data, batch_stride) that is none of the operands.  One pass instead of the multiplyPlain + addInplace loop of a linear layer
 * (app/LinearHelperCKKS.cuh:227-248 MatmulHelper::matmul, :536-556 Conv2dHelper::conv2d): same residues, every operand read once. */
int troyhip_multiply_plain_accumulate(troyhip_context *ctx, const troyhip_ct *const *cts, const uint64_t *const *plains, int count, double plain_scale, troyhip_ct *out,
                                      uint64_t batch, void *stream);
/* ---- plaintext operands in coefficient form.  BFV/BGV: plain = plain_coeff_count coefficients mod t per item
 * (device memory); plain_batch_stride = words between the plaintexts of consecutive batch items, 0 = one plaintext for all. ---- */
/* addPlainInplace / subPlainInplace (src/evaluator_cuda.cu:1654-1720, src/utils/scalingvariant_cuda.cu:21-176).
 * CKKS: plain is the RNS polynomial [limbs][N] (NTT form) at the level of ct, plain_scale must equal ct->scale. */
int troyhip_add_plain(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *plain, uint64_t plain_coeff_count, uint64_t plain_batch_stride, double plain_scale,
                      int subtract, uint64_t batch, void *stream);
/* multiplyPlainInplace for coefficient-form operands = multiplyPlainNormal (src/evaluator_cuda.cu:1757-1815), BFV/BGV */
int troyhip_multiply_plain(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *plain, uint64_t plain_coeff_count, uint64_t plain_batch_stride, uint64_t batch,
                           void *stream);
/* transformToNttInplace(Plaintext&, parms_id) (src/evaluator_cuda.cu:1866-1948): out [count][limbs][N], limbs identifies the level */
int troyhip_plain_to_ntt(troyhip_context *ctx, const uint64_t *plain, uint64_t plain_coeff_count, uint64_t plain_batch_stride, int limbs, uint64_t *out,
                         uint64_t count, void *stream);
/* applyKeySwitchingInplace (src/evaluator_cuda.cu:1365-1378): size-2 ciphertext, c1 is switched to the key `kswitch_key` */
int troyhip_apply_key_switching(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *kswitch_key, uint64_t batch, void *stream);
/* negacyclicShiftInplace (src/evaluator_cuda.cu:2342-2351): every polynomial times x^shift, 0 <= shift < 2N (x^N = -1) */
int troyhip_negacyclic_shift(troyhip_context *ctx, troyhip_ct *ct, uint64_t shift, uint64_t batch, void *stream);
/* divideByPolyModulusDegreeInplace (src/evaluator_cuda.cu:2259-2273): every limb times N^-1 * mul modulo its prime (LWE packing) */
int troyhip_divide_by_poly_modulus_degree(troyhip_context *ctx, troyhip_ct *ct, uint64_t mul, uint64_t batch, void *stream);
/* ---- DecryptorCuda::decrypt (src/decryptor_cuda.cu:61-330; dotProductCtSkArray + decryptScaleAndRound / decryptModt,
 * src/utils/rns_cuda.cu:510-621), SURVEY 8-f3.  secret_key: device [K][N] (NTT form, src/secretkey.h).  plain_out (device):
 * BFV/BGV N coefficients mod t per item, items plain_batch_stride words apart; CKKS the RNS plaintext [limbs][N] (NTT form). */
int troyhip_decrypt(troyhip_context *ctx, const troyhip_ct *ct, const uint64_t *secret_key, uint64_t *plain_out, uint64_t plain_batch_stride, uint64_t batch,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif
