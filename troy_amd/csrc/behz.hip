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



// The matrix-core kernels (behz2.hip) are the path wherever their tables exist (L <= 15, |Bsk| <= 16, primes of at least 33 bits: BehzDev::v2); the VALU
// kernels above take every other base.  Probe builds (-DTROYHIP_PROBES) read TROYHIP_BEHZ=valu to force the VALU kernels (A/B runs).
static bool behz_use_mfma() {
    static const bool v = [] { const char *e = probe_env("TROYHIP_BEHZ"); return !(e && e[0] == 'v'); }();
    return v;
}
bool behz_floor_prescaled(const BehzDev &c) { return c.v2 && c.floor_desc && behz_use_mfma(); }
void launch_behz_extend(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, const BehzDev &c, u64 N, u64 polys, hipStream_t s, const u64 *in2,
                        u64 split) {
    if (!polys) return;
    if (c.v2 && behz_use_mfma()) return launch_behz2_extend(in, in_pstride, out, out_pstride, primes, c, N, polys, s, in2, split);
    if (in2) { // the VALU form takes one operand per launch
        launch_behz_extend(in, in_pstride, out, out_pstride, primes, c, N, split, s);
        launch_behz_extend(in2, in_pstride, out + split * out_pstride, out_pstride, primes, c, N, polys - split, s);
        return;
    }
    stats::counter(stats::BEHZ_VALU_LAUNCHES).fetch_add(1, std::memory_order_relaxed);
    const size_t lds = (size_t)c.L * BEHZ_COEFFS * sizeof(u64);
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
    if (c.v2 && behz_use_mfma()) return launch_behz2_floor_sk(dq, dq_pstride, db, db_pstride, out, out_pstride, primes, c, N, polys, s);
    stats::counter(stats::BEHZ_VALU_LAUNCHES).fetch_add(1, std::memory_order_relaxed);
    const size_t lds = (size_t)(c.L + c.nB + 1) * BEHZ_COEFFS * sizeof(u64);
    for (u64 p0 = 0; p0 < polys; p0 += 65535) {
        const u64 np = polys - p0 < 65535 ? polys - p0 : 65535;
        TROY_LAUNCH(behz_floor_sk_kernel, dim3(ceil_div(N, BEHZ_COEFFS), (unsigned)np), dim3(BEHZ_THREADS), lds, s, dq + p0 * dq_pstride, dq_pstride, db + p0 * db_pstride, db_pstride,
                    out + p0 * out_pstride, out_pstride, primes, c, N);
    }
    launch_check("behz_floor_sk_kernel");
}

} // namespace troyhip
