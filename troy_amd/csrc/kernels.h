// kernels.h -- launch interface of the gfx950 kernels (ntt.hip, poly.hip, behz.hip).
#pragma once
#include "device_types.h"

namespace troyhip {

// compute units of the current device (256 on MI355X), read once.  The launchers size their per-workgroup loops by it: a loop over many rows / tiles
// per workgroup amortises twiddles and fragments when the launch is large, and starves the chip when it is small (a single ciphertext).
unsigned device_cus();
// the largest per-workgroup count <= cap (halving) that still leaves about four workgroups per compute unit; `groups` = workgroups per unit of count 1
inline unsigned plan_per_workgroup(size_t units, unsigned cap, size_t groups) {
    const size_t want = 4 * (size_t)device_cus();
    unsigned r = cap ? cap : 1;
    while (r > 1 && ((units + r - 1) / r) * groups < want) r = (r + 1) / 2;
    return r;
}

// ---- ntt.hip ----
void launch_ntt(u64 *data, const PrimeDesc *primes, const LimbMap &map, size_t rows, int logn, bool inverse, hipStream_t stream);
// key switching: transforms of all (digit, output prime) pairs + inner product with the key in one pair of launches (ntt2.hip)
void launch_ntt2_ks_mac(u64 *D, const u64 *src, u64 src_ostride, const PrimeDesc *primes, const LimbMap &map, size_t rows, int logn, const u64 *key, u64 *acc,
                        const uint8_t *key_limb, unsigned K, const u64 *ckks_target, u64 t_bstride, bool lazy, u64 src_bound, hipStream_t stream);
// BEHZ multiply: forward transforms of two size-2 operands + ciphertext tensor in one pass pair (ntt2.hip)
bool ntt2_tensor_supported(int logn);
bool ntt2_ks_mac_supported(int logn);
// src_slots != 0 (both bases of a product in ONE launch): map covers the q primes followed by the B_sk primes; the first src_slots (= q) slots of every
// polynomial are read from the operands src_a / src_b ([batch][2][src_slots][N]), the B_sk slots lie in xa / xb ([batch][2][period][N]) already
void launch_ntt2_tensor(u64 *xa, const u64 *src_a, u64 *xb, const u64 *src_b, u64 *out, const PrimeDesc *primes, const LimbMap &map, size_t batch, int logn,
                        hipStream_t stream, unsigned src_slots = 0);
// forward transform of `src` (same row layout, left untouched) into `data`
void launch_ntt_from(u64 *data, const u64 *src, const PrimeDesc *primes, const LimbMap &map, size_t rows, int logn, hipStream_t stream);

// ---- ntt2.hip (N >= 4096) ----
bool ntt2_supported(int logn);
// Mod-down of BFV (kind 0) / BGV (kind 2) key switching as the store epilogue of the two-pass inverse transform (every size the
// single-pass kernel does not take): `data` = acc[2 batch][dl + 1][N], slot dl already in coefficient form, `primes` = Context::d_desc_md
// (N^-1 constants carry qk^-1, aux = qk^-1); the data slots are transformed and  ct[b][k][j] += (acc_j - share of the special limb) qk^-1
// is applied instead of storing them (evaluator.cpp:2528-2648).
// BGV: `share` = the 128-bit integers al + k_t qk per (item, coefficient) from launch_ks_bgv_share (poly.hip).
// base != nullptr: the ciphertext being accumulated into is (base, 0), i.e. ct[b][0] = base[b] + ..., ct[b][1] = ... (rotations: base = sigma(c0) in a
// temporary; spares the copy into ct[b][0] and the zero fill of ct[b][1])
struct Ntt2ModDown { int kind; u64 *ct; u64 ct_bstride; unsigned dl; u64 qk, half; const u64 *share; const u64 *base = nullptr; u64 base_bstride = 0; int base_polys = 1; };
// passes: bit 0 = the first pass of the transform, bit 1 = the second (a caller that runs the first pass of several slot ranges as ONE launch and the second
// passes separately: the key-switch mod-down of a small launch, whose special limb and data limbs differ only in the last pass).
// plan_begin / plan_count (with passes == 2): the slot range the shared FIRST pass ran over -- the FP64 bound walk of this launch assumes the largest
// prime of that range, because that is the bound the lazy doubles it reads were left with
void launch_ntt2_slots(u64 *data, const u64 *src, u64 src_ostride, bool src_reduce, const PrimeDesc *primes, const LimbMap &map, size_t rows, int logn,
                       bool inverse, hipStream_t stream, bool src_same_layout, u64 src_bound, unsigned slot_begin, unsigned slot_count, const Ntt2ModDown *md,
                       unsigned passes = 3, unsigned plan_begin = 0, unsigned plan_count = 0);
void launch_ntt2(u64 *data, const u64 *src, u64 src_ostride, bool src_reduce, const PrimeDesc *primes, const LimbMap &map, size_t rows, int logn,
                 bool inverse, hipStream_t stream, bool src_same_layout = false, u64 src_bound = 0);

// ---- ntt1.hip (N = 2^12 .. 2^15: one HBM round trip per limb-transform) ----
// feature: 0 plain transform, 1 the BFV mod-down epilogue (Ntt1ModDown), 2 the CKKS correction form (Ntt1Corr), 3 the inverse from a strided source
bool ntt1_supported(int logn, const LimbMap &map, size_t rows, int feature = 0);
// BFV mod-down by the special prime (ks_moddown_kernel<0>, evaluator.cpp:2528-2648) fused into the inverse transform of the key-switch
// accumulators acc[o = 2 b + cpt][slot][N]: the slots below `dl` leave the kernel as  ct[b][cpt][slot] += (acc - [t']_q + [half]_q) qk^-1
// instead of being stored; slot `dl` (the special limb) must already be in coefficient form.  `primes` is then Context::d_desc_md,
// whose N^-1 constants carry qk^-1 and whose `aux` is qk^-1 itself.
// base != nullptr: accumulate onto (base[b], 0) instead of onto what ct holds (rotations)
struct Ntt1ModDown { u64 *ct; u64 ct_bstride; u64 dl, qk, half; const u64 *base = nullptr; u64 base_bstride = 0; int base_polys = 1; }; // base_polys == 2: onto (base[b][0], base[b][1]) -- relinearize out of place
// CKKS divide-and-round by a prime qx in NTT form (divideAndRoundqLastNttInplace, rns.cpp:832-877; the mod-down of the CKKS key switch,
// evaluator.cpp:2600-2648): the correction polynomial corr_slot = [(last + half) mod qx]_p + (p - [half]_p) is BUILT on load from the
// coefficient-form residues `last` of qx (one row per outer index), transformed, and COMBINED on store:
//   out = (in + p - NTT(corr)) * qx^-1 mod p,   stored, or added to what out holds (accumulate)
// so the correction never exists in memory and the two element-wise kernels around the transform disappear.  Row (o, slot):
// in[(o / group) * in_gstride + (o % group) * in_ostride + slot * N ..] (in_gstride == 0: o * in_ostride),
// out[(o / group) * out_gstride + (o % group) * out_ostride + slot * N ..], inv[slot] = qx^-1 mod p_slot.
struct Ntt1Corr {
    const u64 *last, *in;
    u64 in_ostride;
    u64 *out;
    u64 out_gstride, out_ostride;
    unsigned group;
    const Shoup *inv;
    u64 qx, half;
    bool accumulate;
    u64 in_gstride = 0;
    // accumulate with base != nullptr: what is added to is (base, 0, 0, ..) per group instead of what out holds -- member 0 of group g reads
    // base[g * base_gstride + slot * N ..], the other members start from zero (rotations: base = sigma(c0) in a temporary)
    const u64 *base = nullptr;
    u64 base_gstride = 0;
    int base_polys = 1; // 2: members 0 and 1 of a group read base[g * base_gstride + {0, 1} * out_ostride ..] (relinearize out of place)
};
// slot_mask: only these prime slots of the row pattern are transformed
// logn is mandatory (round-4 advisor): a caller at N = 2^12 .. 2^14 that forgot it would run the 2^15 kernel over rows 2 .. 8 times shorter
void launch_ntt1(u64 *data, const u64 *src, const PrimeDesc *primes, const LimbMap &map, size_t rows, int logn, bool inverse, hipStream_t stream, u64 slot_mask = ~0ull,
                 const Ntt1ModDown *md = nullptr, const Ntt1Corr *cr = nullptr, u64 src_ostride = 0);

// ---- poly.hip ----
void launch_ew(int op, const u64 *a, const u64 *b, u64 *out, const PrimeDesc *primes, const LimbMap &map, int logn, u64 rows, hipStream_t s);
void launch_mul_scalar(u64 *x, const PrimeDesc *primes, const LimbMap &map, const u64 *scalars, int logn, u64 rows, hipStream_t s);
void launch_mul_plain(u64 *a, const u64 *plain, const PrimeDesc *primes, const LimbMap &map, int logn, u64 limbs, u64 rows, hipStream_t s,
                      u64 rows_per_item = 0, u64 plain_bstride = 0); // rows_per_item != 0: item b = row / rows_per_item uses plain + b * plain_bstride

// plaintext operands (evaluator_cuda.cu:1654-1948, utils/scalingvariant_cuda.cu:21-176)
struct PlainArgs {
    const PrimeDesc *primes;
    LimbMap map;          // prime ids of the level's limbs
    u64 delta[64];        // BFV: floor(q/t) mod q_l
    u64 t_p, t_cr0, t_cr1;
    u64 q_mod_t, thr;     // q mod t, (t+1)/2
    u64 cf;               // BGV correction factor of the ciphertext
    int logn;
    u64 limbs, n_coeffs;  // plaintext: n_coeffs coefficients mod t per item
    u64 items;            // batch
    u64 plain_bstride;    // words between the plaintexts of consecutive items (0: one plaintext for all)
};
// kind: 0 BFV scaling variant, 2 BGV (times correction factor), 1 rows (CKKS: plain is [limbs][N] like the ciphertext)
void launch_add_plain(int kind, bool sub, u64 *ct0, u64 ct_bstride, const u64 *plain, const PlainArgs &a, hipStream_t s);
// lifted[item][l][n] = plain coefficient as a residue of q_l (upper half shifted by q - t), zero beyond n_coeffs
void launch_plain_lift(const u64 *plain, u64 *lifted, const PlainArgs &a, hipStream_t s);
void launch_tensor(int s1, int s2, const u64 *a, const u64 *b, u64 *out, u64 a_bstride, u64 b_bstride, const PrimeDesc *primes, const LimbMap &map,
                   int logn, u64 limbs, u64 batch, hipStream_t s);
void launch_galois(bool ntt_form, const u64 *in, u64 in_bstride, u64 *out, u64 out_bstride, const PrimeDesc *primes, const LimbMap &map, int logn, uint32_t elt, u64 limbs,
                   u64 batch, hipStream_t s);

struct ModSwitchArgs {
    const PrimeDesc *primes;
    LimbMap map;         // prime ids of the level's limbs (period = limbs)
    Shoup inv_qlast[64]; // q_last^-1 mod q_l
    u64 half, t_p, t_cr0, t_cr1, inv_qlast_mod_t;
    int logn;
    u64 limbs;           // limbs of the INPUT level
    u64 polys;           // batch * size
};
void launch_modswitch(int kind, const u64 *in, u64 *out, const ModSwitchArgs &a, hipStream_t s);
void launch_rescale_stepA(const u64 *last, u64 last_pstride, u64 *corr, const ModSwitchArgs &a, hipStream_t s);
void launch_rescale_stepB(const u64 *in, const u64 *corr, u64 *out, const ModSwitchArgs &a, hipStream_t s);
void launch_drop_last(const u64 *in, u64 *out, int logn, u64 limbs, u64 polys, hipStream_t s);
// group != 0: poly i sits at (i / group) * gstride + (i % group) * pstride (a strided batch of ciphertexts)
void launch_gather_limb(const u64 *in, u64 *out, int logn, u64 pstride, u64 limb, u64 polys, hipStream_t s, u64 group = 0, u64 gstride = 0);

struct KsArgs {
    const PrimeDesc *primes;
    uint8_t key_id[65];   // prime id of output index i (i < dl: i ; i == dl: special prime)
    uint8_t key_limb[65]; // limb index inside the key of output index i
    Shoup inv_qk[64];     // q_special^-1 mod q_j
    u64 half;             // floor(q_special / 2)
    u64 t_p, t_cr0, t_cr1, inv_qk_mod_t;
    int logn;
    u64 dl;               // decomposition limbs = ciphertext limbs
    u64 K;                // key-level limbs
    u64 batch;
};
void launch_ks_expand(const u64 *target, u64 t_bstride, u64 *D, const KsArgs &a, hipStream_t s);
void launch_ks_mac(const u64 *D, const u64 *key, const u64 *ckks_target, u64 t_bstride, u64 *acc, const KsArgs &a, hipStream_t s);
void launch_ks_moddown(int kind, const u64 *acc, u64 *ct, u64 ct_bstride, const KsArgs &a, hipStream_t s);
void launch_ks_bgv_share(const u64 *acc, u64 *share /* [2 batch][N][2] */, const KsArgs &a, hipStream_t s);
void launch_ks_ckks_corr(const u64 *last, u64 *corr, const KsArgs &a, hipStream_t s);
void launch_ks_ckks_combine(const u64 *acc, const u64 *corr, u64 *ct, u64 ct_bstride, const KsArgs &a, hipStream_t s);
// ---- decryption (decryptor_cuda.cu:61-330, rns_cuda.cu:510-621) ----
struct DecryptArgs {
    const PrimeDesc *primes;
    LimbMap map;
    int logn;
    u64 limbs, size, batch;
    u64 ct_bstride, out_bstride;
    // BFV decryptScaleAndRound: y_l = x_l * (t gamma (q/q_l)^-1) mod q_l ; sums against (q/q_l) mod t and mod gamma
    Shoup pre[64];
    u64 mat_t[64], mat_g[64];
    u64 t_p, t_cr0, t_cr1, g_p, g_cr0, g_cr1;
    u64 neg_inv_q_mod_t, neg_inv_q_mod_gamma, inv_gamma_mod_t;
    // BGV decryptModt: exact conversion to t with the double-precision rounding term, times correction_factor^-1
    u64 q_mod_t, inv_cf;
};
// acc[b][l][n] = sum_{i>=1} x[b][i-1][l][n] * spow[i-1][l][n]  (x: the NTT-form polynomials c_1.., spow: s, s^2, ..)
void launch_dot_sk(const u64 *x, const u64 *spow, u64 *acc, const DecryptArgs &a, hipStream_t s);
// acc[b][l][n] += c0 of ciphertext b
void launch_add_c0(const u64 *ct, u64 *acc, const DecryptArgs &a, hipStream_t s);
void launch_decrypt_final(int scheme, const u64 *acc, u64 *out, const DecryptArgs &a, hipStream_t s);
void launch_negacyclic_shift(const u64 *in, u64 in_bstride, u64 *out, u64 out_bstride, const PrimeDesc *primes, const LimbMap &map, int logn, u64 shift, u64 rows_per_item,
                             u64 limbs, u64 batch, hipStream_t s);
void launch_copy_strided(const u64 *src, u64 src_bstride, u64 *dst, u64 dst_bstride, u64 count, u64 batch, hipStream_t s);
void launch_zero_strided(u64 *dst, u64 dst_bstride, u64 count, u64 batch, hipStream_t s);
void launch_fill_uniform(u64 *out, const PrimeDesc *primes, const LimbMap &map, int logn, u64 seed, u64 row0, u64 rows, hipStream_t s);

// ---- selftest.hip (test support) ----
void launch_modarith_probe(int op, const u64 *a, const u64 *b, const u64 *c, u64 p, u64 aux_value, u64 *out, u64 n, hipStream_t s);

// sum of up to 16 ciphertext (x) plaintext products in NTT form (poly.hip)
struct MulPlainAccArgs { const u64 *ct[16]; u64 ct_bstride[16]; const u64 *plain[16]; int count; };
void launch_mul_plain_acc(const MulPlainAccArgs &x, u64 *out, u64 out_bstride, const PrimeDesc *primes, const LimbMap &map, int logn, u64 limbs, u64 size, u64 batch,
                          hipStream_t s);

// ---- behz.hip ----
// base-change matrix entry split into 21-bit limbs (m = m0 + m1 2^21 + m2 2^42): see behz.hip
struct Mat3 { u32 m0, m1, m2, pad; };

// epilogue constants of one output prime in the second matrix-core form (behz2.hip): the bias biaslo + bias1 2^32 is a multiple of
// p that keeps both halves of the recombined sum positive (bias1 = 2^(8 nd - 18), nd = digit rows of a residue of this prime), the
// quotient estimate reads 32 bits of the sum from bit sh = bitlen(p) - 2 (sh32 = sh - 32) against mu = floor(2^(sh+32) / p); negp = 2^64 - p
struct BehzK2 { u64 p, negp, biaslo, bias1; u32 mu, sh32; };

// device-resident constants of one level (built by Context, see context.cpp)
struct BehzDev {
    int L, nB, nBsk;
    uint8_t q_id[64], bsk_id[66];     // prime ids
    // --- extension q -> Bsk with the m_tilde Montgomery correction (all row factors folded, see context.cpp) ---
    const Shoup *ext_pre;             // [L]   (m_tilde * (q/q_l)^-1) mod q_l
    const Mat3 *ext_mat3;             // [nBsk][L]  (q/q_l) * m_tilde^-1 mod Bsk_o
    const u32 *ext_mt_row;            // [L]   (q/q_l) mod m_tilde = 2^32
    u64 neg_inv_q_mod_mt;             // -q^-1 mod 2^32
    const u64 *ext_q;                 // [nBsk]  q * m_tilde^-1 mod Bsk_o
    // --- floor + Shenoy-Kumaresan ---
    const Shoup *floor_pre;           // [L]   (t * (q/q_l)^-1) mod q_l
    const Mat3 *floor_mat3;           // [nBsk][L]  -(q/q_l) * q^-1 [* (B/B_o)^-1 for o < nB] mod Bsk_o
    const Mat3 *floor_t3;             // [nBsk]     t * q^-1 [* (B/B_o)^-1] mod Bsk_o
    const Mat3 *B2q3;                 // [L][nB]   (B/B_b) mod q_l
    const Mat3 *B2msk3;               // [nB]      (B/B_b) mod m_sk
    Shoup inv_B_mod_msk;
    const u64 *prod_B_mod_q;          // [L]
    // --- second matrix-core form (behz2.hip): rows reduced modulo the output prime, 8 byte-shifts per output, a row-block = 4 outputs;
    // fragments [row-block][k-block][lane] x 16 bytes with k-blocks of 4 limbs x 8 digits.  v2 != 0 when built (L <= 15, |Bsk| <= 16)
    int v2, f2_fast;                  // f2_fast: every q prime >= 2^33 (one-step quotient estimate), else the two-word reduction
    const void *x_frag;               // extension: [ceil(nBsk/4)][KBx][64], KBx = ceil((L+1)/4): limbs 0..L-1 and the r column at limb L
    const void *x_mt_frag;            // [KBx][64]  the m_tilde row (modulo 2^32, 4 shifts) in both halves of the tile
    const BehzK2 *x_k;                // [nBsk]
    const void *f1_frag;              // floor stage 1: [ceil(nBsk/4)][KB1][64], KB1 = ceil(L/4); the m_sk row carries B^-1 too
    const void *f1s_frag;             // the same rows in the order (row-block, half, j) -> output 4 rb + 2 half + j, for the small-base kernels (nullptr: not built)
    const BehzK2 *f1_k;               // [nBsk]
    const void *f2_frag;              // stage 2: [ceil(L/4)][KB2][64], KB2 = ceil((nB+1)/4): the B limbs and the alpha column at limb nB
    const void *f2_msk_frag;          // [KB2][64]  (B/B_b) B^-1 mod m_sk in both halves of the tile
    const BehzK2 *f2_k;               // [L]
    BehzK2 msk_k;
    // the floor kernel of this form takes its inputs PRE-SCALED: the inverse transforms that produce dq / db run with this copy of the
    // prime table, whose N^-1 constants carry t (q/q_l)^-1 mod q_l (q limbs) resp. t q^-1 [(B/B_o)^-1 | B^-1] mod Bsk_o (Bsk limbs),
    // so the per-coefficient multiplications by those factors cost nothing (behz_floor_prescaled() tells the evaluator)
    const PrimeDesc *floor_desc;
    // --- register-resident FP64 form (behz3.hip): small bases whose primes all lie below 2^50 -- the rows above as pairs of doubles (w, w / p); nullptr: not built
    const double *fp_ext, *fp_floor;
};
bool behz3_supported(const BehzDev &c);
void launch_behz3_extend(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const BehzDev &c, u64 N, u64 polys, hipStream_t s, const u64 *in2 = nullptr, u64 split = 0);
void launch_behz3_floor_sk(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride, u64 *out, u64 out_pstride, const BehzDev &c, u64 N, u64 polys, hipStream_t s);
bool behz_floor_prescaled(const BehzDev &c);
void launch_behz2_extend(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, const BehzDev &c, u64 N, u64 polys, hipStream_t s,
                         const u64 *in2 = nullptr, u64 split = 0);
void launch_behz2_floor_sk(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, const BehzDev &c, u64 N,
                           u64 polys, hipStream_t s);
// in2 != nullptr: polynomials [split, polys) are read from in2 (both operands of a small product through one launch; the outputs are one run)
void launch_behz_extend(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, const BehzDev &c, u64 N, u64 polys, hipStream_t s,
                        const u64 *in2 = nullptr, u64 split = 0);
void launch_behz_floor_sk(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, const BehzDev &c, u64 N,
                          u64 polys, hipStream_t s);

} // namespace troyhip
