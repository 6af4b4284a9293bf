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
//     ~14 instructions of a 128-bit multiply-add); the six column sums are recombined and Barrett-reduced once per
//     output;
//   * all Shoup quotients are precomputed on the host.
#include "kernels.h"

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

// in [polys][L][N] (canonical, coefficient form) -> out [polys][nBsk][N]
// fastbconvmTilde (rns.cpp:1012-1037) fused with smMrq (rns.cpp:943-983)
__global__ __launch_bounds__(BEHZ_THREADS) void behz_extend_kernel(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, BehzDev c, u64 N) {
    TROY_DYN_LDS(u64, lds);
    u64 *y = lds;                               // [L][64]
    u64 *rmt = lds + (u64)c.L * BEHZ_COEFFS;    // [64]  the m_tilde residue
    u64 *sres = rmt + BEHZ_COEFFS;              // [nBsk][64]  q -> Bsk sums awaiting the Montgomery correction
    const unsigned lane = threadIdx.x & 63;
    const int w = BEHZ_UNIFORM((int)(threadIdx.x >> 6));
    const u64 n = (u64)blockIdx.x * BEHZ_COEFFS + lane;
    const u64 poly = blockIdx.y;
    const bool live = n < N;
    const u64 *x = in + poly * in_pstride;
    const mat3_ptr mat = (mat3_ptr)c.q2bsk3;
    const cshoup_ptr ext_pre = (cshoup_ptr)c.ext_pre;
    for (int l = w; l < c.L; l += 4) {
        const u64 p = primes[c.q_id[l]].p;
        const Shoup pre = ld_shoup(ext_pre + l);
        y[l * BEHZ_COEFFS + lane] = live ? mul_shoup(x[(u64)l * N + n], pre.op, pre.quo, p) : 0;
    }
    __syncthreads();
    const int n_out = c.nBsk + 1; // Bsk limbs then m_tilde
    // sweep: wave w owns outputs o = w + 4*i; the m_tilde row (o = nBsk) is produced in the FIRST sweep that contains it
    for (int ob = 0; ob < n_out; ob += 4 * BEHZ_OPW) {
        Acc6 acc[BEHZ_OPW];
        int o[BEHZ_OPW];
#pragma unroll
        for (int i = 0; i < BEHZ_OPW; i++) { acc_zero(acc[i]); o[i] = ob + w + 4 * i; }
        for (int l = 0; l < c.L; l++) {
            const u64 v = y[l * BEHZ_COEFFS + lane];
            const u32 x0 = (u32)v, x1 = (u32)(v >> 32);
#pragma unroll
            for (int i = 0; i < BEHZ_OPW; i++) {
                const int oo = o[i] < n_out ? o[i] : n_out - 1; // clamp (result discarded)
                acc_mac(acc[i], x0, x1, ld_mat3(mat + oo * c.L + l));
            }
        }
#pragma unroll
        for (int i = 0; i < BEHZ_OPW; i++) {
            if (o[i] == c.nBsk) rmt[lane] = acc_low32(acc[i]);
        }
#pragma unroll
        for (int i = 0; i < BEHZ_OPW; i++) {
            if (o[i] < c.nBsk) {
                const U128 v = acc_value(acc[i]);
                sres[o[i] * BEHZ_COEFFS + lane] = barrett128(v.lo, v.hi, mod_of(primes[c.bsk_id[o[i]]]));
            }
        }
    }
    __syncthreads();
    // Montgomery correction (smMrq): out_o = (sum_o + q * r) * m_tilde^-1 mod Bsk_o, r centred
    const u64 r_mt = ((u64)(u32)rmt[lane] * c.neg_inv_q_mod_mt) & 0xFFFFFFFFull;
    for (int oo = w; oo < c.nBsk; oo += 4) {
        const PrimeDesc &pd = primes[c.bsk_id[oo]];
        const Mod m = mod_of(pd);
        u64 temp = r_mt;
        if (temp >= (u64(1) << 31)) temp += m.p - (u64(1) << 32);
        U128 a{sres[oo * BEHZ_COEFFS + lane], 0};
        add128(a, temp, ((cu64_ptr)c.prod_q_mod_bsk)[oo]);
        const u64 v = barrett128(a.lo, a.hi, m);
        const Shoup im = ld_shoup(c.inv_mt_mod_bsk + oo);
        if (live) out[poly * out_pstride + (u64)oo * N + n] = mul_shoup(v, im.op, im.quo, m.p);
    }
}

// dq [polys][L][N], db [polys][nBsk][N] (after the inverse NTT; any representative < 2^64) -> out [polys][L][N]
// steps (6)-(8) of bfvMultiply (evaluator.cpp:575-623): multiply by t, fastFloor (rns.cpp:985-1010),
// fastbconvSk (rns.cpp:879-941)
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
    // fastFloor: z_o = (t*db_o - conv_{q->Bsk}(t*dq)_o) * q^-1 mod Bsk_o ; u_b = z_b * (B/B_b)^-1 mod B_b
    {
        const mat3_ptr mat = (mat3_ptr)c.q2bsk3;
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
                    const int oo = o[i] < c.nBsk ? o[i] : c.nBsk - 1;
                    acc_mac(acc[i], x0, x1, ld_mat3(mat + oo * c.L + l));
                }
            }
#pragma unroll
            for (int i = 0; i < BEHZ_OPW; i++) {
                const int oo = o[i];
                if (oo < c.nBsk) {
                    const Mod m = mod_of(primes[c.bsk_id[oo]]);
                    const U128 v = acc_value(acc[i]);
                    const u64 conv = barrett128(v.lo, v.hi, m);
                    const Shoup tb = ld_shoup(c.t_mod_bsk + oo), iq = ld_shoup(c.inv_q_mod_bsk + oo);
                    const u64 xb = live ? mul_shoup(db[poly * db_pstride + (u64)oo * N + n], tb.op, tb.quo, m.p) : 0;
                    const u64 z = mul_shoup(xb + (m.p - conv), iq.op, iq.quo, m.p);
                    if (oo < c.nB) {
                        const Shoup bp = ld_shoup(c.B_pre + oo);
                        u[oo * BEHZ_COEFFS + lane] = mul_shoup(z, bp.op, bp.quo, m.p);
                    } else {
                        zsk[lane] = z;
                    }
                }
            }
        }
    }
    __syncthreads();
    // Shenoy-Kumaresan: alpha = (conv_{B->m_sk}(z) - z_sk) * B^-1 mod m_sk (every wave recomputes it: nB terms)
    const PrimeDesc &psk = primes[c.bsk_id[c.nB]];
    const Mod msk = mod_of(psk);
    u64 alpha;
    {
        Acc6 a;
        acc_zero(a);
        const mat3_ptr mv = (mat3_ptr)c.B2msk3;
        for (int b = 0; b < c.nB; b++) {
            const u64 v = u[b * BEHZ_COEFFS + lane];
            acc_mac(a, (u32)v, (u32)(v >> 32), ld_mat3(mv + b));
        }
        const U128 v = acc_value(a);
        const u64 conv_sk = barrett128(v.lo, v.hi, msk);
        alpha = mul_shoup(conv_sk + (msk.p - zsk[lane]), c.inv_B_mod_msk.op, c.inv_B_mod_msk.quo, msk.p);
    }
    const bool neg = alpha > (msk.p >> 1);
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
                    const Mod m = mod_of(primes[c.q_id[l]]);
                    U128 v = acc_value(acc[i]);
                    const u64 pb = ((cu64_ptr)c.prod_B_mod_q)[l];
                    if (neg) add128(v, msk.p - alpha, pb);  // alpha represents a negative value
                    else add128(v, alpha, m.p - pb);
                    if (live) out[poly * out_pstride + (u64)l * N + n] = barrett128(v.lo, v.hi, m);
                }
            }
        }
    }
}

void launch_behz_extend(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, const BehzDev &c, u64 N, u64 polys, hipStream_t s) {
    if (!polys) return;
    size_t lds = (size_t)(c.L + 1 + c.nBsk) * BEHZ_COEFFS * sizeof(u64);
    for (u64 p0 = 0; p0 < polys; p0 += 65535) { // gridDim.y limit
        const u64 np = polys - p0 < 65535 ? polys - p0 : 65535;
        TROY_LAUNCH(behz_extend_kernel, dim3(ceil_div(N, BEHZ_COEFFS), (unsigned)np), dim3(BEHZ_THREADS), lds, s, in + p0 * in_pstride, in_pstride, out + p0 * out_pstride,
                    out_pstride, primes, c, N);
    }
    launch_check("behz_extend_kernel");
}
void launch_behz_floor_sk(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, const BehzDev &c, u64 N,
                          u64 polys, hipStream_t s) {
    if (!polys) return;
    size_t lds = (size_t)(c.L + c.nB + 1) * BEHZ_COEFFS * sizeof(u64);
    for (u64 p0 = 0; p0 < polys; p0 += 65535) {
        const u64 np = polys - p0 < 65535 ? polys - p0 : 65535;
        TROY_LAUNCH(behz_floor_sk_kernel, dim3(ceil_div(N, BEHZ_COEFFS), (unsigned)np), dim3(BEHZ_THREADS), lds, s, dq + p0 * dq_pstride, dq_pstride, db + p0 * db_pstride,
                    db_pstride, out + p0 * out_pstride, out_pstride, primes, c, N);
    }
    launch_check("behz_floor_sk_kernel");
}

} // namespace troyhip
