// behz.hip -- BEHZ RNS base conversion for BFV multiply on gfx950.
//
// Replaces BaseConverterCuda::fastConvertArray (gFastConvertArrayStepA/B, src/utils/rns_cuda.cu:96-145),
// RNSToolCuda::fastbconvmTilde + smMrq (rns_cuda.cu:423-464, 500-508) and fastFloor + fastbconvSk
// (rns_cuda.cu:365-421, 466-498); CPU twins src/utils/rns.cpp:415-459, 879-1037.
//
// The reference runs 6 launches per polynomial for the extension and 5 for floor+SK, writes a transposed
// (uncoalesced) temporary, reduces every dot-product term and divides 128/64 bits per coefficient per limb on the
// device.  Here each direction is ONE launch over the whole batch.  This work is a per-coefficient integer
// mat-vec (|in| x |out| = 14 x 16 at the headline size) and is VALU-bound, so the kernel is built around the
// cheapest exact multiply-accumulate gfx950 offers:
//   * a 256-thread workgroup owns 64 consecutive coefficients; the pre-scaled input residues go to LDS once
//     ([limb][64], lane = coefficient, conflict-free);
//   * every wave computes FOUR output limbs at a time from one LDS read per input limb (register blocking);
//   * the base-change matrix entries are wave-uniform and are read with scalar loads, pre-split on the host into
//     three 21-bit limbs: (32-bit half of the residue) x (21-bit limb) < 2^53, so up to 2^11 products accumulate in a
//     plain 64-bit register -- one v_mad_u64_u32 per partial product, NO carry chains (6 per term instead of the
//     ~14 instructions of a 128-bit multiply-add); the six column sums are recombined and reduced once per output;
//   * every per-row factor of the reference's step sequence (m_tilde^-1, q^-1, t, (B/B_b)^-1) is folded into the matrix on
//     the host, so an output is ONE mat-vec row and ONE 128-bit reduction (the reference reduces, multiplies and
//     reduces again per step; the residues written to HBM are the same canonical values);
//   * all Shoup quotients are precomputed on the host.
#include "kernels.h"
#include <cstdlib>

namespace troyhip {

#define BEHZ_THREADS 256
#define BEHZ_COEFFS 64
#define BEHZ_OPW 4 // output limbs per wave per sweep

#ifdef TROYHIP_CPU_EMUL
#define BEHZ_UNIFORM(x) (x)
typedef const u32 *cu32_ptr;
typedef const u64 *cu64_ptr;
#else
#define BEHZ_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// constant address space => scalar (s_load) reads for wave-uniform addresses
typedef const __attribute__((address_space(4))) u32 *cu32_ptr;
typedef const __attribute__((address_space(4))) u64 *cu64_ptr;
#endif
typedef const Mat3 *mat3_ptr;
typedef const Shoup *cshoup_ptr;
__device__ __forceinline__ Mat3 ld_mat3(const Mat3 *p) {
    const cu32_ptr q = (cu32_ptr)p;
    return Mat3{q[0], q[1], q[2], 0};
}
__device__ __forceinline__ Shoup ld_shoup(const Shoup *p) {
    const cu64_ptr q = (cu64_ptr)p;
    return Shoup{q[0], q[1]};
}

// six carry-free column accumulators of sum_l x_l * m_l  (x = x0 + x1 2^32, m = m0 + m1 2^21 + m2 2^42)
struct Acc6 {
    u64 a0, a1, a2, b0, b1, b2;
};
__device__ __forceinline__ void acc_zero(Acc6 &a) { a.a0 = a.a1 = a.a2 = a.b0 = a.b1 = a.b2 = 0; }
__device__ __forceinline__ void acc_mac(Acc6 &a, u32 x0, u32 x1, const Mat3 m) {
    a.a0 += (u64)x0 * m.m0; a.a1 += (u64)x0 * m.m1; a.a2 += (u64)x0 * m.m2;
    a.b0 += (u64)x1 * m.m0; a.b1 += (u64)x1 * m.m1; a.b2 += (u64)x1 * m.m2;
}
// exact 128-bit value of the sum (it is < 2^128 by construction: <= 64 terms of 60 x 61 bits)
__device__ __forceinline__ U128 acc_value(const Acc6 &a) {
    u128 v = (u128)a.a0 + ((u128)a.a1 << 21) + ((u128)a.a2 << 42) + ((u128)a.b0 << 32) + ((u128)a.b1 << 53) + ((u128)a.b2 << 74);
    return U128{(u64)v, (u64)(v >> 64)};
}
__device__ __forceinline__ u32 acc_low32(const Acc6 &a) { return (u32)(a.a0 + (a.a1 << 21)); } // value mod 2^32
__device__ __forceinline__ void add128(U128 &v, u64 x, u64 y) { // v += x*y
    u64 lo = x * y, hi = mulhi64(x, y);
    v.lo += lo;
    v.hi += hi + (v.lo < lo);
}

// (lo, hi) mod p, canonical: the high word is folded with 2^64 mod p (one Shoup multiply, valid for ANY 64-bit hi), the low
// word takes a single-word Barrett step; 17 32-bit multiplies instead of the 24 of the two-word Barrett reduction
__device__ __forceinline__ u64 reduce128(const U128 v, const PrimeDesc &pd) {
    const u64 a = mul_lazy(v.hi, pd.r64.op, pd.r64.quo, pd.p);   // [0, 2p)
    const u64 b = v.lo - mulhi64(v.lo, pd.cr1) * pd.p;            // [0, 2p)
    u64 s = a + b;                                                // [0, 4p), p < 2^61
    s = s >= pd.two_p ? s - pd.two_p : s;
    return s >= pd.p ? s - pd.p : s;
}

// in [polys][L][N] (canonical, coefficient form) -> out [polys][nBsk][N]
// fastbconvmTilde (rns.cpp:1012-1037) fused with smMrq (rns.cpp:943-983):
//   out_o = (sum_l y_l (q/q_l) + q r~) m_tilde^-1 mod Bsk_o,  y_l = x_l m_tilde (q/q_l)^-1 mod q_l,
//   r~ = the centred m_tilde residue  -(sum_l y_l (q/q_l)) q^-1 mod 2^32
// m_tilde^-1 is folded into the matrix and into q (context.cpp), so each output costs one mat-vec row and ONE reduction.
__global__ __launch_bounds__(BEHZ_THREADS) void behz_extend_kernel(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, BehzDev c, u64 N) {
    TROY_DYN_LDS(u64, lds);
    u64 *y = lds;                               // [L][64]
    const unsigned lane = threadIdx.x & 63;
    const int w = BEHZ_UNIFORM((int)(threadIdx.x >> 6));
    const u64 n = (u64)blockIdx.x * BEHZ_COEFFS + lane;
    const u64 poly = blockIdx.y;
    const bool live = n < N;
    const u64 *x = in + poly * in_pstride;
    const mat3_ptr mat = (mat3_ptr)c.ext_mat3;
    const cshoup_ptr ext_pre = (cshoup_ptr)c.ext_pre;
    for (int l = w; l < c.L; l += 4) {
        const u64 p = primes[c.q_id[l]].p;
        const Shoup pre = ld_shoup(ext_pre + l);
        y[l * BEHZ_COEFFS + lane] = live ? mul_shoup(x[(u64)l * N + n], pre.op, pre.quo, p) : 0;
    }
    __syncthreads();
    // the m_tilde residue only needs the low 32 bits of every term; each wave computes it for its own lanes
    u32 rsum = 0;
    {
        const cu32_ptr mt = (cu32_ptr)c.ext_mt_row;
        for (int l = 0; l < c.L; l++) rsum += (u32)y[l * BEHZ_COEFFS + lane] * mt[l];
    }
    const u64 r_mt = ((u64)rsum * c.neg_inv_q_mod_mt) & 0xFFFFFFFFull;
    // sweep: wave w owns outputs o = ob + w + 4 i
    for (int ob = 0; ob < c.nBsk; ob += 4 * BEHZ_OPW) {
        Acc6 acc[BEHZ_OPW];
        int o[BEHZ_OPW];
#pragma unroll
        for (int i = 0; i < BEHZ_OPW; i++) { acc_zero(acc[i]); o[i] = ob + w + 4 * i; }
        for (int l = 0; l < c.L; l++) {
            const u64 v = y[l * BEHZ_COEFFS + lane];
            const u32 x0 = (u32)v, x1 = (u32)(v >> 32);
#pragma unroll
            for (int i = 0; i < BEHZ_OPW; i++) {
                const int oo = o[i] < c.nBsk ? o[i] : c.nBsk - 1; // clamp (result discarded)
                acc_mac(acc[i], x0, x1, ld_mat3(mat + oo * c.L + l));
            }
        }
#pragma unroll
        for (int i = 0; i < BEHZ_OPW; i++) {
            if (o[i] < c.nBsk) {
                const PrimeDesc &pd = primes[c.bsk_id[o[i]]];
                u64 temp = r_mt;                                   // centred representative of r (rns.cpp:966-975)
                if (temp >= (u64(1) << 31)) temp += pd.p - (u64(1) << 32);
                U128 v = acc_value(acc[i]);
                add128(v, temp, ((cu64_ptr)c.ext_q)[o[i]]);       // < 2^126 + 2^122
                const u64 r = reduce128(v, pd);
                if (live) out[poly * out_pstride + (u64)o[i] * N + n] = r;
            }
        }
    }
}

// dq [polys][L][N], db [polys][nBsk][N] (after the inverse NTT; any representative < 2^64) -> out [polys][L][N]
// steps (6)-(8) of bfvMultiply (evaluator.cpp:575-623): multiply by t, fastFloor (rns.cpp:985-1010),
// fastbconvSk (rns.cpp:879-941).  fastFloor:  z_o = (t db_o - conv_{q->Bsk}(t dq)_o) q^-1 mod Bsk_o  and the B part is
// pre-scaled for the next conversion, u_b = z_b (B/B_b)^-1 mod B_b: with q^-1 (and (B/B_b)^-1) folded into the negated
// matrix and into t, z / u is one mat-vec row over the L+1 inputs (y_0..y_{L-1}, db_o) and ONE reduction.
__global__ __launch_bounds__(BEHZ_THREADS) void behz_floor_sk_kernel(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride, u64 *out, u64 out_pstride,
                                                                      const PrimeDesc *primes, BehzDev c, u64 N) {
    TROY_DYN_LDS(u64, lds);
    u64 *y = lds;                                  // [L][64]
    u64 *u = lds + (u64)c.L * BEHZ_COEFFS;         // [nB][64]   pre-scaled B residues of the floor result
    u64 *zsk = u + (u64)c.nB * BEHZ_COEFFS;        // [64]       m_sk residue of the floor result
    const unsigned lane = threadIdx.x & 63;
    const int w = BEHZ_UNIFORM((int)(threadIdx.x >> 6));
    const u64 n = (u64)blockIdx.x * BEHZ_COEFFS + lane;
    const u64 poly = blockIdx.y;
    const bool live = n < N;
    const cshoup_ptr floor_pre = (cshoup_ptr)c.floor_pre;
    for (int l = w; l < c.L; l += 4) {
        const u64 p = primes[c.q_id[l]].p;
        const Shoup pre = ld_shoup(floor_pre + l);
        y[l * BEHZ_COEFFS + lane] = live ? mul_shoup(dq[poly * dq_pstride + (u64)l * N + n], pre.op, pre.quo, p) : 0;
    }
    __syncthreads();
    {
        const mat3_ptr mat = (mat3_ptr)c.floor_mat3;
        const mat3_ptr tmat = (mat3_ptr)c.floor_t3;
        for (int ob = 0; ob < c.nBsk; ob += 4 * BEHZ_OPW) {
            Acc6 acc[BEHZ_OPW];
            int o[BEHZ_OPW];
#pragma unroll
            for (int i = 0; i < BEHZ_OPW; i++) {
                acc_zero(acc[i]);
                o[i] = ob + w + 4 * i;
                const int oo = o[i] < c.nBsk ? o[i] : c.nBsk - 1;
                const u64 xb = live ? db[poly * db_pstride + (u64)oo * N + n] : 0;
                acc_mac(acc[i], (u32)xb, (u32)(xb >> 32), ld_mat3(tmat + oo));
            }
            for (int l = 0; l < c.L; l++) {
                const u64 v = y[l * BEHZ_COEFFS + lane];
                const u32 x0 = (u32)v, x1 = (u32)(v >> 32);
#pragma unroll
                for (int i = 0; i < BEHZ_OPW; i++) {
                    const int oo = o[i] < c.nBsk ? o[i] : c.nBsk - 1;
                    acc_mac(acc[i], x0, x1, ld_mat3(mat + oo * c.L + l));
                }
            }
#pragma unroll
            for (int i = 0; i < BEHZ_OPW; i++) {
                const int oo = o[i];
                if (oo < c.nBsk) {
                    const u64 r = reduce128(acc_value(acc[i]), primes[c.bsk_id[oo]]);
                    if (oo < c.nB) u[oo * BEHZ_COEFFS + lane] = r;
                    else zsk[lane] = r;
                }
            }
        }
    }
    __syncthreads();
    // Shenoy-Kumaresan: alpha = (conv_{B->m_sk}(z) - z_sk) * B^-1 mod m_sk (every wave recomputes it: nB terms)
    const PrimeDesc &psk = primes[c.bsk_id[c.nB]];
    u64 alpha;
    {
        Acc6 a;
        acc_zero(a);
        const mat3_ptr mv = (mat3_ptr)c.B2msk3;
        for (int b = 0; b < c.nB; b++) {
            const u64 v = u[b * BEHZ_COEFFS + lane];
            acc_mac(a, (u32)v, (u32)(v >> 32), ld_mat3(mv + b));
        }
        const u64 conv_sk = reduce128(acc_value(a), psk);
        alpha = mul_shoup(conv_sk + (psk.p - zsk[lane]), c.inv_B_mod_msk.op, c.inv_B_mod_msk.quo, psk.p);
    }
    const bool neg = alpha > (psk.p >> 1);
    {
        const mat3_ptr mat = (mat3_ptr)c.B2q3;
        for (int ob = 0; ob < c.L; ob += 4 * BEHZ_OPW) {
            Acc6 acc[BEHZ_OPW];
            int o[BEHZ_OPW];
#pragma unroll
            for (int i = 0; i < BEHZ_OPW; i++) { acc_zero(acc[i]); o[i] = ob + w + 4 * i; }
            for (int b = 0; b < c.nB; b++) {
                const u64 v = u[b * BEHZ_COEFFS + lane];
                const u32 x0 = (u32)v, x1 = (u32)(v >> 32);
#pragma unroll
                for (int i = 0; i < BEHZ_OPW; i++) {
                    const int oo = o[i] < c.L ? o[i] : c.L - 1;
                    acc_mac(acc[i], x0, x1, ld_mat3(mat + oo * c.nB + b));
                }
            }
#pragma unroll
            for (int i = 0; i < BEHZ_OPW; i++) {
                const int l = o[i];
                if (l < c.L) {
                    const PrimeDesc &pd = primes[c.q_id[l]];
                    U128 v = acc_value(acc[i]);
                    const u64 pb = ((cu64_ptr)c.prod_B_mod_q)[l];
                    if (neg) add128(v, psk.p - alpha, pb);  // alpha represents a negative value
                    else add128(v, alpha, pd.p - pb);
                    if (live) out[poly * out_pstride + (u64)l * N + n] = reduce128(v, pd);
                }
            }
        }
    }
}


// ================================================================ matrix-core path (MFMA int8)
// The base conversion is a matrix product  S[c][o] = sum_l y[c][l] * M[o][l]  over the integers (then one reduction per
// entry).  Written in balanced base-256 digits it becomes an int8 GEMM that v_mfma_i32_32x32x32_i8 computes exactly:
//   * y_l (< 2^61) -> eight signed digits = bytes of (y + 0x80..80) ^ 0x80..80: TWO instructions per residue, and the
//     64-bit register pair IS the 8 consecutive K-bytes of the B operand (K index = limb * 8 + digit);
//   * M[o][l] is expanded on the host into 16 Toeplitz rows (o, s): row (o, s) . digits(y) = coefficient of 2^(8 s);
//   * the product is taken transposed (rows = (output, shift), columns = coefficients), so after the 4 k-blocks a lane
//     holds ALL 16 shift coefficients of ONE (coefficient, output) pair in its 16 accumulator registers: no cross-lane
//     traffic; sum_s C_s 2^(8s) is recombined with 64/128-bit adds and reduced once.
// K = 16 limbs x 8 digits = 128 (4 k-blocks), 16 output slots x 16 shifts = 256 rows (8 row-blocks): L <= 16, |Bsk| <= 16.
// Per (coefficient, output) the VALU does ~70 instructions instead of 14 x 6 v_mad_u64_u32 + ~65; the 14 x 15 x 64
// byte products run on the matrix cores (32 MFMA per 32 coefficients, 2 k cycles of the MFMA pipe per 32 coefficients).
#define BEHZ_TILE 64
// probe hooks (tools/behz_probe.sh builds throw-away variants): bit0 no HBM loads, bit1 no MFMA, bit2 no recombine/reduce
// epilogue, bit3 no workgroup barriers, bit4 no HBM stores.  0 in the product.
#ifndef BEHZ_TPW
#define BEHZ_TPW 8 // tiles per workgroup
#endif
#ifndef BEHZ_EXP
#define BEHZ_EXP 0
#endif
#define BEHZ_MFMA(fa, fb, fc)                                                                                             \
    do {                                                                                                                 \
        if (BEHZ_EXP & 2) (fc).v[0] += (fb).bytes[0] + (fa).bytes[0];                                                    \
        else TROY_MFMA_I8(fa, fb, fc);                                                                                   \
    } while (0)
#define BEHZ_SYNC()                                                                                                       \
    do {                                                                                                                 \
        if (!(BEHZ_EXP & 8)) __syncthreads();                                                                            \
    } while (0)
#define BEHZ_LOAD(expr, fake) ((BEHZ_EXP & 1) ? (u64)(fake) : (expr))
#define BEHZ_LIVE(cond, r) ((BEHZ_EXP & 16) ? (r) == ~0ull : (cond))
__device__ __forceinline__ void mfma_zero(MfmaAcc &a) {
#pragma unroll
    for (int r = 0; r < 16; r++) a.v[r] = 0;
}
// sum_s C_s 2^(8 s), C_s signed 32-bit (|C_s| < 2^22); the total is known to be in [0, 2^126).
// Four coefficients at a time fit 64 bits: W_j = (C_4j + C_4j+1 2^8) + (C_4j+2 + C_4j+3 2^8) 2^16 -- two 32-bit shift-adds
// and one v_mad_i64_i32; then V = W_0 + W_1 2^32 + W_2 2^64 + W_3 2^96 word by word with sign words.
__device__ __forceinline__ long long mad_i64_i32(int a, int b, long long c) {
#ifdef TROYHIP_CPU_EMUL
    return (long long)a * b + c;
#else
    long long d;
    u64 sink;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(sink) : "v"(a), "s"(b), "v"(c));
    return d;
#endif
}
__device__ __forceinline__ U128 mfma_recombine(const MfmaAcc &a) {
    long long w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p01 = (int)((u32)a.v[4 * j] + ((u32)a.v[4 * j + 1] << 8)), p23 = (int)((u32)a.v[4 * j + 2] + ((u32)a.v[4 * j + 3] << 8));
        w[j] = mad_i64_i32(p23, 1 << 16, (long long)p01);
    }
    // lo = W0 + (W1 << 32) ; hi = sext(W0) + (W1 >> 32) + W2 + (W3 << 32) + carry   (all modulo 2^64: the result fits 128 bits)
    const u64 w0 = (u64)w[0], t1 = (u64)w[1] << 32;
    const u64 lo = w0 + t1;
    u64 hi = (u64)(w[0] >> 63) + (u64)(w[1] >> 32) + (u64)w[2] + ((u64)w[3] << 32) + (lo < w0);
    return U128{lo, hi};
}
__device__ __forceinline__ MfmaFrag ld_frag(const void *base, size_t idx) { return reinterpret_cast<const MfmaFrag *>(base)[idx]; }

// per-output constants staged in LDS: with the transposed product the output index differs between the two halves of a
// wave, so these are per-lane reads (LDS, not dependent global loads)
struct BehzOutConst {
    u64 p, cr1, two_p, r64_op, r64_quo, extra; // extra: ext_q[o] (extension)
};
__device__ __forceinline__ u64 reduce128c(const U128 v, const BehzOutConst &k) {
    const u64 a = mul_lazy(v.hi, k.r64_op, k.r64_quo, k.p);
    const u64 b = v.lo - mulhi64(v.lo, k.cr1) * k.p;
    u64 s = a + b;
    s = s >= k.two_p ? s - k.two_p : s;
    return s >= k.p ? s - k.p : s;
}

// (sum of the 16 shift coefficients) + x * y, reduced: one output residue
__device__ __forceinline__ u64 behz_finish(const MfmaAcc &acc, u64 x, u64 y, const BehzOutConst &k) {
    if (BEHZ_EXP & 4) return (u64)(u32)acc.v[0] ^ x ^ ((u64)(u32)acc.v[15] << 32);
    U128 v = mfma_recombine(acc);
    add128(v, x, y);
    return reduce128c(v, k);
}

// recombine + reduce only: the correction term and a non-negativity bias went through the matrix product (BehzDev::ext_fold / floor_fold)
__device__ __forceinline__ u64 behz_finish_folded(const MfmaAcc &acc, const BehzOutConst &k) {
    if (BEHZ_EXP & 4) return (u64)(u32)acc.v[0] ^ ((u64)(u32)acc.v[15] << 32);
    return reduce128c(mfma_recombine(acc), k);
}
// The two limbs after the last real one (zero padding in LDS) become this coefficient's extra inputs: the eight balanced digits of a
// signed value, and the constant 1.  Word `pos` of a lane's fragment of k-block kb is limb 4 kb + 2 half + pos, and the extra limbs
// sit in the last k-block: FOLD = 2 (limb count = 2 mod 4) -> both words of the half-1 lanes; FOLD = 1 (1 mod 4) -> word 1 of the
// half-0 lanes and word 0 of the half-1 lanes.
template <int KB, int FOLD> __device__ __forceinline__ void behz_patch_fold(MfmaFrag (&bf)[KB], unsigned half, long long value) {
    const u64 c80 = 0x8080808080808080ull;
    const u64 digits = ((u64)value + c80) ^ c80;
    MfmaFrag f = bf[KB - 1];
    if (FOLD == 2) {
        frag_set_word(f, 0, digits);
        frag_set_word(f, 1, 1);
        if (half) bf[KB - 1] = f;
    } else {
        if (half) frag_set_word(f, 0, 1);
        else frag_set_word(f, 1, digits);
        bf[KB - 1] = f;
    }
}

// same contract as behz_extend_kernel; a 256-thread workgroup walks `tiles_per_wg` tiles of 64 coefficients of one
// polynomial; wave w owns row-blocks w and w + 4 (outputs 2w, 2w+1, 2w+8, 2w+9) and keeps their A-fragments in registers
template <int KB, int FOLD> __global__ __launch_bounds__(BEHZ_THREADS) void behz_extend_mfma_kernel(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes,
                                                                                          BehzDev c, u64 N, unsigned tiles_per_wg) {
    __shared__ __attribute__((aligned(16))) u64 ydig[8 * BEHZ_TILE * 2]; // [limb pair][coefficient] 16-byte units
    __shared__ BehzOutConst oc[16];
    const unsigned lane = threadIdx.x & 63, half = lane >> 5, cl = lane & 31;
    const int w = BEHZ_UNIFORM((int)(threadIdx.x >> 6));
    const u64 poly = blockIdx.y;
    // ragged edges (padding limbs, absent outputs, a short last tile) are handled by the buffer range check, not by branches
    const BufRsrc rin = make_rsrc(in + poly * in_pstride, (u32)((u64)c.L * N * 8));
    const BufRsrc rout = make_rsrc(out + poly * out_pstride, (u32)((u64)c.nBsk * N * 8));
    const u32 n32 = (u32)N;
    const int RB = (c.nBsk + 1) >> 1;
    const cshoup_ptr ext_pre = (cshoup_ptr)c.ext_pre;
    if ((int)threadIdx.x < c.nBsk) {
        const PrimeDesc &pd = primes[c.bsk_id[threadIdx.x]];
        oc[threadIdx.x] = BehzOutConst{pd.p, pd.cr1, pd.two_p, pd.r64.op, pd.r64.quo, c.ext_q[threadIdx.x]};
    }
    MfmaFrag af[2][KB], am[KB];
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int rb = w + 4 * j < RB ? w + 4 * j : 0;
            af[j][kb] = ld_frag(c.ext_frag, ((size_t)rb * 4 + kb) * 64 + lane);
        }
        am[kb] = ld_frag(c.ext_mt_frag, (size_t)kb * 64 + lane);
    }
    const u64 c80 = 0x8080808080808080ull;
    // this wave converts limbs w, w+4, ...: their constants are wave-uniform and loaded once; the residues of the NEXT tile are
    // fetched while the current one is multiplied (one exposed memory latency per workgroup instead of several per tile)
    u64 qp[KB];
    Shoup qpre[KB];
#pragma unroll
    for (int i = 0; i < KB; i++) {
        const int l = w + 4 * i;
        const int lc = l < c.L ? l : 0;
        const unsigned id = BEHZ_UNIFORM((unsigned)c.q_id[lc]);
        qp[i] = ((cu64_ptr)&primes[id])[0];
        qpre[i] = ld_shoup(ext_pre + lc);
    }
    u64 xr[KB];
    auto fetch = [&](unsigned t) { // a tile beyond N, a padding limb: offset out of range, the load returns 0
        const u32 n = (blockIdx.x * tiles_per_wg + t) * BEHZ_TILE + lane;
#pragma unroll
        for (int i = 0; i < KB; i++) {
            const u32 l = (u32)(w + 4 * i);
            xr[i] = BEHZ_LOAD(buf_load_u64(rin, (l < (u32)c.L && n < n32) ? (l * n32 + n) * 8u : TROY_BUF_OOB), (lane + n) * 0x9E3779B97F4Aull + l);
        }
    };
    fetch(0);
    // the loop below waits for the prefetched residues with "all but the stores issued after them"; on entry no store is in
    // flight yet, and the compiler merges both cases into a full drain (vmcnt(0)) unless the entry looks the same: four
    // out-of-range stores (dropped by the range check) stand in for the previous tile's
#pragma unroll
    for (int i = 0; i < 4; i++) buf_store_u64(rout, TROY_BUF_OOB + 8 * i, 0);
    for (unsigned t = 0; t < tiles_per_wg; t++) {
        const u32 n0 = (blockIdx.x * tiles_per_wg + t) * BEHZ_TILE;
        if (n0 >= n32) break;
        // y_l = x_l * m_tilde * (q/q_l)^-1 mod q_l, stored as balanced digits; limbs L..4KB-1 are zero padding (loaded as 0)
#pragma unroll
        for (int i = 0; i < KB; i++) {
            const int l = w + 4 * i;
            const u64 v = (mul_shoup(xr[i], qpre[i].op, qpre[i].quo, qp[i]) + c80) ^ c80;
            ydig[(((l >> 1) * BEHZ_TILE) + lane) * 2 + (l & 1)] = v;
        }
        fetch(t + 1 < tiles_per_wg ? t + 1 : 0x7FFFFFu); // past the last tile: out of range
        BEHZ_SYNC();
#pragma unroll
        for (int sub = 0; sub < BEHZ_TILE / 32; sub++) {
            const unsigned cc = sub * 32 + cl;
            MfmaFrag bf[KB];
#pragma unroll
            for (int kb = 0; kb < KB; kb++) bf[kb] = ld_frag(ydig, (size_t)(2 * kb + half) * BEHZ_TILE + cc);
            // r = -(sum_l y_l (q/q_l)) q^-1 mod 2^32, from shifts 0..3 of the m_tilde row
            MfmaAcc acc;
            mfma_zero(acc);
#pragma unroll
            for (int kb = 0; kb < KB; kb++) BEHZ_MFMA(am[kb], bf[kb], acc);
            const u32 rsum = (u32)acc.v[0] + ((u32)acc.v[1] << 8) + ((u32)acc.v[2] << 16) + ((u32)acc.v[3] << 24);
            const u64 r_mt = ((u64)rsum * c.neg_inv_q_mod_mt) & 0xFFFFFFFFull;
            // centred representative of r (rns.cpp:966-975) as one more input of the product: out = (sum + r q) m_tilde^-1
            if (FOLD) behz_patch_fold<KB, FOLD>(bf, half, (long long)r_mt - (long long)((r_mt >> 31) << 32));
#pragma unroll
            for (int j = 0; j < 2; j++) {
                // the store is issued on both sides of the (wave-uniform) branch, so the number of stores in flight is a constant
                u64 r = 0;
                u32 off = TROY_BUF_OOB;
                if (w + 4 * j < RB) {
                    mfma_zero(acc);
#pragma unroll
                    for (int kb = 0; kb < KB; kb++) BEHZ_MFMA(af[j][kb], bf[kb], acc);
                    const u32 o = 2 * (u32)(w + 4 * j) + half;
                    const BehzOutConst k = oc[o < (u32)c.nBsk ? o : 0];
                    if (FOLD) r = behz_finish_folded(acc, k);
                    else {
                        u64 temp = r_mt;                               // centred: a negative r is represented by r + p
                        if (temp >= (u64(1) << 31)) temp += k.p - (u64(1) << 32);
                        r = behz_finish(acc, temp, k.extra, k);
                    }
                    if (BEHZ_LIVE(o < (u32)c.nBsk && n0 + cc < n32, r)) off = (o * n32 + n0 + cc) * 8u;
                }
                buf_store_u64(rout, off, r);
            }
        }
        BEHZ_SYNC();
    }
}


// same contract as behz_floor_sk_kernel.  Two chained int8 GEMMs per tile of 64 coefficients:
//   (1) rows of floor_frag1 x digits(y)  -> + db_o T_o -> u_b (b < |B|, as digits for the next product) and z_sk
//   (2) rows of floor_frag2 x digits(u)  -> + alpha-term -> out_l ;  alpha comes from the B -> m_sk row, which every wave
//       evaluates for its own lanes (one extra row-block instead of a broadcast and a barrier).
// Stage-1 fragments live in registers, stage-2 fragments are shared through LDS (all four waves need different row-blocks of
// the same 32 KiB, and 3 workgroups per CU must fit).
template <int KB, int FOLD> __global__ __launch_bounds__(BEHZ_THREADS) void behz_floor_sk_mfma_kernel(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride, u64 *out,
                                                                                           u64 out_pstride, const PrimeDesc *primes, BehzDev c, u64 N,
                                                                                           unsigned tiles_per_wg) {
    TROY_DYN_LDS(u64, lds);
    u64 *ydig = lds;                                          // [8 limb pairs][64] 16-byte units (8 KiB)
    u64 *udig = ydig + 8 * BEHZ_TILE * 2;                     // same, for the B residues of the floor result
    u64 *zsk = udig + 8 * BEHZ_TILE * 2;                      // [64]
    BehzOutConst *oc1 = reinterpret_cast<BehzOutConst *>(zsk + BEHZ_TILE); // [16] Bsk outputs (extra = T_o)
    BehzOutConst *oc2 = oc1 + 16;                             // [16] q outputs (extra = prod_B mod q_l)
    u64 *frag2 = reinterpret_cast<u64 *>(oc2 + 16);           // [RB2 + 1][4][64] 16-byte fragments: floor_frag2 then floor_msk_frag
    const unsigned lane = threadIdx.x & 63, half = lane >> 5, cl = lane & 31;
    const int w = BEHZ_UNIFORM((int)(threadIdx.x >> 6));
    const u64 poly = blockIdx.y;
    const BufRsrc rq = make_rsrc(dq + poly * dq_pstride, (u32)((u64)c.L * N * 8));
    const BufRsrc rb = make_rsrc(db + poly * db_pstride, (u32)((u64)c.nBsk * N * 8));
    const BufRsrc rout = make_rsrc(out + poly * out_pstride, (u32)((u64)c.L * N * 8));
    const u32 n32 = (u32)N;
    const int RB1 = (c.nBsk + 1) >> 1, RB2 = (c.L + 1) >> 1;
    // m_sk's constants: wave-uniform, read once with scalar loads (a PrimeDesc reference would be re-read with vector loads in
    // the tile loop, and each of those drains the outstanding stores)
    BehzOutConst psk;
    {
        const cu64_ptr pk = (cu64_ptr)&primes[BEHZ_UNIFORM((unsigned)c.bsk_id[c.nB])];
        psk = BehzOutConst{pk[0], pk[2], pk[3], pk[8], pk[9], 0}; // PrimeDesc: p, cr0, cr1, two_p, inv_n, iroot_last_scaled, r64
    }
    // ---- per-workgroup setup
    if ((int)threadIdx.x < c.nBsk) {
        const PrimeDesc &pd = primes[c.bsk_id[threadIdx.x]];
        oc1[threadIdx.x] = BehzOutConst{pd.p, pd.cr1, pd.two_p, pd.r64.op, pd.r64.quo, c.floor_t[threadIdx.x]};
    }
    if ((int)threadIdx.x < c.L) {
        const PrimeDesc &pd = primes[c.q_id[threadIdx.x]];
        oc2[threadIdx.x] = BehzOutConst{pd.p, pd.cr1, pd.two_p, pd.r64.op, pd.r64.quo, c.prod_B_mod_q[threadIdx.x]};
    }
    for (unsigned i = threadIdx.x; i < 8 * BEHZ_TILE * 2; i += BEHZ_THREADS) udig[i] = 0; // padding limbs stay zero
    {
        const ulonglong2 *g2 = reinterpret_cast<const ulonglong2 *>(c.floor_frag2), *gm = reinterpret_cast<const ulonglong2 *>(c.floor_msk_frag);
        ulonglong2 *f = reinterpret_cast<ulonglong2 *>(frag2);
        for (unsigned i = threadIdx.x; i < (unsigned)RB2 * 256; i += BEHZ_THREADS) f[i] = g2[i];
        for (unsigned i = threadIdx.x; i < 256; i += BEHZ_THREADS) f[RB2 * 256 + i] = gm[i];
    }
    MfmaFrag af[2][KB];
#pragma unroll
    for (int kb = 0; kb < KB; kb++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int rb = w + 4 * j < RB1 ? w + 4 * j : 0;
            af[j][kb] = ld_frag(c.floor_frag1, ((size_t)rb * 4 + kb) * 64 + lane);
        }
    const cshoup_ptr floor_pre = (cshoup_ptr)c.floor_pre;
    u64 qp[KB];
    Shoup qpre[KB];
#pragma unroll
    for (int i = 0; i < KB; i++) {
        const int l = w + 4 * i;
        const int lc = l < c.L ? l : 0;
        const unsigned id = BEHZ_UNIFORM((unsigned)c.q_id[lc]);
        qp[i] = ((cu64_ptr)&primes[id])[0];
        qpre[i] = ld_shoup(floor_pre + lc);
    }
    const u64 c80 = 0x8080808080808080ull;
    // both operands of a tile are fetched while the previous tile is computed: the q residues of this wave's limbs (xr) and the
    // Bsk residues this lane needs in the stage-1 epilogue, (sub, j) -> db[o = 2 (w + 4 j) + half][n0 + 32 sub + cl] (dbn)
    u64 xr[KB], dbn[2][2];
    auto fetch = [&](unsigned t) { // out-of-range offsets (padding limb, absent output, tile beyond N) load 0
        const u32 t0 = (blockIdx.x * tiles_per_wg + t) * BEHZ_TILE;
        const u32 n = t0 + lane;
#pragma unroll
        for (int i = 0; i < KB; i++) {
            const u32 l = (u32)(w + 4 * i);
            xr[i] = BEHZ_LOAD(buf_load_u64(rq, (l < (u32)c.L && n < n32) ? (l * n32 + n) * 8u : TROY_BUF_OOB), (lane + n) * 0x9E3779B97F4Aull + l);
        }
#pragma unroll
        for (int sub = 0; sub < 2; sub++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const u32 o = 2 * (u32)(w + 4 * j) + half;
                const u32 n = t0 + sub * 32 + cl;
                dbn[sub][j] = BEHZ_LOAD(buf_load_u64(rb, (o < (u32)c.nBsk && n < n32) ? (o * n32 + n) * 8u : TROY_BUF_OOB), (lane + n) * 0x9E3779B97F4Aull + o);
            }
    };
    fetch(0);
#pragma unroll
    for (int i = 0; i < 4; i++) buf_store_u64(rout, TROY_BUF_OOB + 8 * i, 0); // see behz_extend_mfma_kernel: keeps the loop's waits counted
    for (unsigned t = 0; t < tiles_per_wg; t++) {
        const u32 n0 = (blockIdx.x * tiles_per_wg + t) * BEHZ_TILE;
        if (n0 >= n32) break;
#pragma unroll
        for (int i = 0; i < KB; i++) {
            const int l = w + 4 * i;
            const u64 v = (mul_shoup(xr[i], qpre[i].op, qpre[i].quo, qp[i]) + c80) ^ c80;
            ydig[(((l >> 1) * BEHZ_TILE) + lane) * 2 + (l & 1)] = v;
        }
        const u64 dbv[2][2] = {{dbn[0][0], dbn[0][1]}, {dbn[1][0], dbn[1][1]}};
        fetch(t + 1 < tiles_per_wg ? t + 1 : 0x7FFFFFu); // past the last tile: out of range
        BEHZ_SYNC();
        // ---- stage 1
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            const unsigned cc = sub * 32 + cl;
            MfmaFrag bf[KB];
#pragma unroll
            for (int kb = 0; kb < KB; kb++) bf[kb] = ld_frag(ydig, (size_t)(2 * kb + half) * BEHZ_TILE + cc);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if (w + 4 * j >= RB1) break;
                MfmaAcc acc;
                mfma_zero(acc);
#pragma unroll
                for (int kb = 0; kb < KB; kb++) BEHZ_MFMA(af[j][kb], bf[kb], acc);
                const int o = 2 * (w + 4 * j) + (int)half;
                if (o < c.nBsk) {
                    const BehzOutConst k = oc1[o];
                    const u64 r = behz_finish(acc, dbv[sub][j], k.extra, k);
                    if (o < c.nB) udig[(((o >> 1) * BEHZ_TILE) + cc) * 2 + (o & 1)] = (r + c80) ^ c80;
                    else zsk[cc] = r;
                }
            }
        }
        BEHZ_SYNC();
        // ---- stage 2: Shenoy-Kumaresan
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            const unsigned cc = sub * 32 + cl;
            MfmaFrag bf[KB];
#pragma unroll
            for (int kb = 0; kb < KB; kb++) bf[kb] = ld_frag(udig, (size_t)(2 * kb + half) * BEHZ_TILE + cc);
            MfmaAcc acc;
            mfma_zero(acc);
#pragma unroll
            for (int kb = 0; kb < KB; kb++) BEHZ_MFMA(ld_frag(frag2, ((size_t)RB2 * 4 + kb) * 64 + lane), bf[kb], acc);
            const u64 conv_sk = reduce128c(mfma_recombine(acc), psk);
            const u64 alpha = mul_shoup(conv_sk + (psk.p - zsk[cc]), c.inv_B_mod_msk.op, c.inv_B_mod_msk.quo, psk.p);
            const bool neg = alpha > (psk.p >> 1); // alpha > m_sk / 2 represents a negative value
            if (FOLD) behz_patch_fold<KB, FOLD>(bf, half, (long long)(neg ? alpha - psk.p : alpha)); // out_l = sum - alpha (B mod q_l)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                u64 r = 0;
                u32 off = TROY_BUF_OOB;
                if (w + 4 * j < RB2) {
                    mfma_zero(acc);
#pragma unroll
                    for (int kb = 0; kb < KB; kb++) BEHZ_MFMA(ld_frag(frag2, ((size_t)(w + 4 * j) * 4 + kb) * 64 + lane), bf[kb], acc);
                    const u32 l = 2 * (u32)(w + 4 * j) + half;
                    const BehzOutConst k = oc2[l < (u32)c.L ? l : 0];
                    if (FOLD) r = behz_finish_folded(acc, k);
                    else r = behz_finish(acc, neg ? psk.p - alpha : alpha, neg ? k.extra : k.p - k.extra, k);
                    if (BEHZ_LIVE(l < (u32)c.L && n0 + cc < n32, r)) off = (l * n32 + n0 + cc) * 8u;
                }
                buf_store_u64(rout, off, r);
            }
        }
        BEHZ_SYNC();
    }
}

// TROYHIP_BEHZ=valu forces the VALU kernels (they remain the path for L > 16 or |Bsk| > 16); read once
static bool behz_use_mfma() {
    static const bool v = [] { const char *e = getenv("TROYHIP_BEHZ"); return !(e && e[0] == 'v'); }();
    return v;
}
bool behz_floor_prescaled(const BehzDev &c) { return c.v2 && c.floor_desc && behz_use_mfma(); }
void launch_behz_extend(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, const BehzDev &c, u64 N, u64 polys, hipStream_t s, const u64 *in2,
                        u64 split) {
    if (!polys) return;
    if (c.v2 && behz_use_mfma()) return launch_behz2_extend(in, in_pstride, out, out_pstride, primes, c, N, polys, s, in2, split);
    if (in2) { // the other forms take one operand per launch
        launch_behz_extend(in, in_pstride, out, out_pstride, primes, c, N, split, s);
        launch_behz_extend(in2, in_pstride, out + split * out_pstride, out_pstride, primes, c, N, polys - split, s);
        return;
    }
    const bool mfma = c.ext_frag && behz_use_mfma();
    const u64 tiles = ceil_div(N, (u64)BEHZ_TILE);
    const unsigned tpw = tiles >= 64 ? BEHZ_TPW : 1; // amortise the A-fragment loads over several tiles when there are enough workgroups
    size_t lds = (size_t)c.L * BEHZ_COEFFS * sizeof(u64);
    for (u64 p0 = 0; p0 < polys; p0 += 65535) { // gridDim.y limit
        const u64 np = polys - p0 < 65535 ? polys - p0 : 65535;
        if (mfma) {
            const dim3 grid((unsigned)ceil_div(tiles, (u64)tpw), (unsigned)np);
            const u64 *pi = in + p0 * in_pstride;
            u64 *po = out + p0 * out_pstride;
#define BEHZ_EXT_LAUNCH(KB_, F_) TROY_LAUNCH(HIP_KERNEL_NAME(behz_extend_mfma_kernel<KB_, F_>), grid, dim3(BEHZ_THREADS), 0, s, pi, in_pstride, po, out_pstride, primes, c, N, tpw)
            switch (((c.L + 3) / 4) * 4 + c.ext_fold) { // k-blocks, fold form
            case 4: BEHZ_EXT_LAUNCH(1, 0); break;
            case 5: BEHZ_EXT_LAUNCH(1, 1); break;
            case 6: BEHZ_EXT_LAUNCH(1, 2); break;
            case 8: BEHZ_EXT_LAUNCH(2, 0); break;
            case 9: BEHZ_EXT_LAUNCH(2, 1); break;
            case 10: BEHZ_EXT_LAUNCH(2, 2); break;
            case 12: BEHZ_EXT_LAUNCH(3, 0); break;
            case 13: BEHZ_EXT_LAUNCH(3, 1); break;
            case 14: BEHZ_EXT_LAUNCH(3, 2); break;
            case 17: BEHZ_EXT_LAUNCH(4, 1); break;
            case 18: BEHZ_EXT_LAUNCH(4, 2); break;
            default: BEHZ_EXT_LAUNCH(4, 0); break;
            }
#undef BEHZ_EXT_LAUNCH
        } else
            TROY_LAUNCH(behz_extend_kernel, dim3(ceil_div(N, BEHZ_COEFFS), (unsigned)np), dim3(BEHZ_THREADS), lds, s, in + p0 * in_pstride, in_pstride, out + p0 * out_pstride,
                        out_pstride, primes, c, N);
    }
    launch_check("behz_extend_kernel");
}
void launch_behz_floor_sk(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, const BehzDev &c, u64 N,
                          u64 polys, hipStream_t s) {
    if (!polys) return;
    if (c.v2 && behz_use_mfma()) return launch_behz2_floor_sk(dq, dq_pstride, db, db_pstride, out, out_pstride, primes, c, N, polys, s);
    const bool mfma = c.floor_frag1 && behz_use_mfma();
    const u64 tiles = ceil_div(N, (u64)BEHZ_TILE);
    const unsigned tpw = tiles >= 64 ? BEHZ_TPW : 1;
    const int kb = ((c.L > c.nB ? c.L : c.nB) + 3) / 4;
    const size_t lds_mfma = (size_t)(2 * 8 * BEHZ_TILE * 2 + BEHZ_TILE) * sizeof(u64) + 32 * sizeof(BehzOutConst) + (size_t)((c.L + 1) / 2 + 1) * 4 * 64 * 16;
    size_t lds = (size_t)(c.L + c.nB + 1) * BEHZ_COEFFS * sizeof(u64);
    for (u64 p0 = 0; p0 < polys; p0 += 65535) {
        const u64 np = polys - p0 < 65535 ? polys - p0 : 65535;
        const u64 *pq = dq + p0 * dq_pstride, *pb = db + p0 * db_pstride;
        u64 *po = out + p0 * out_pstride;
        if (mfma) {
            const dim3 grid((unsigned)ceil_div(tiles, (u64)tpw), (unsigned)np);
#define BEHZ_FLOOR_LAUNCH(KB_, F_)                                                                                         \
    TROY_LAUNCH(HIP_KERNEL_NAME(behz_floor_sk_mfma_kernel<KB_, F_>), grid, dim3(BEHZ_THREADS), lds_mfma, s, pq, dq_pstride, pb, db_pstride, po, out_pstride, primes, c, N, tpw)
            switch (kb * 4 + c.floor_fold) {
            case 4: BEHZ_FLOOR_LAUNCH(1, 0); break;
            case 5: BEHZ_FLOOR_LAUNCH(1, 1); break;
            case 6: BEHZ_FLOOR_LAUNCH(1, 2); break;
            case 8: BEHZ_FLOOR_LAUNCH(2, 0); break;
            case 9: BEHZ_FLOOR_LAUNCH(2, 1); break;
            case 10: BEHZ_FLOOR_LAUNCH(2, 2); break;
            case 12: BEHZ_FLOOR_LAUNCH(3, 0); break;
            case 13: BEHZ_FLOOR_LAUNCH(3, 1); break;
            case 14: BEHZ_FLOOR_LAUNCH(3, 2); break;
            case 17: BEHZ_FLOOR_LAUNCH(4, 1); break;
            case 18: BEHZ_FLOOR_LAUNCH(4, 2); break;
            default: BEHZ_FLOOR_LAUNCH(4, 0); break;
            }
#undef BEHZ_FLOOR_LAUNCH
        } else {
            TROY_LAUNCH(behz_floor_sk_kernel, dim3(ceil_div(N, BEHZ_COEFFS), (unsigned)np), dim3(BEHZ_THREADS), lds, s, pq, dq_pstride, pb, db_pstride, po, out_pstride, primes, c,
                        N);
        }
    }
    launch_check("behz_floor_sk_kernel");
}

} // namespace troyhip
