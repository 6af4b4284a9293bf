// behz.hip -- BEHZ RNS base conversion for BFV multiply on gfx950.
//
// Replaces BaseConverterCuda::fastConvertArray (gFastConvertArrayStepA/B, src/utils/rns_cuda.cu:96-145),
// RNSToolCuda::fastbconvmTilde + smMrq (rns_cuda.cu:423-464, 500-508) and fastFloor + fastbconvSk
// (rns_cuda.cu:365-421, 466-498); CPU twins src/utils/rns.cpp:415-459, 879-1037.
//
// The reference runs 6 launches per polynomial for the extension and 5 for floor+SK, writes a
// transposed (uncoalesced) temporary, reduces every dot-product term and performs a 128/64-bit
// division per coefficient per limb on the device.  Here each direction is ONE launch over the whole
// batch: a 256-thread workgroup owns 64 consecutive coefficients; the per-limb pre-scaled residues
// are staged in LDS ([limb][64], conflict-free: lane = coefficient), every output limb is a 128-bit
// lazy dot product reduced once (Barrett-128), and all Shoup quotients come precomputed from the host.
// Wave w of the workgroup produces output limbs o == w (mod 4), so the matrix row M[o][*] is
// wave-uniform (scalar loads).
#include "kernels.h"

namespace troyhip {

#define BEHZ_THREADS 256
#define BEHZ_COEFFS 64


// in [polys][L][N] (canonical, coefficient form) -> out [polys][nBsk][N]
// fastbconvmTilde (rns.cpp:1012-1037) fused with smMrq (rns.cpp:943-983)
__global__ __launch_bounds__(BEHZ_THREADS) void behz_extend_kernel(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, BehzDev c, u64 N) {
    TROY_DYN_LDS(u64, lds);
    u64 *y = lds;                           // [L][64]
    u64 *sres = lds + (u64)c.L * BEHZ_COEFFS; // [nBsk+1][64]
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const u64 n = (u64)blockIdx.x * BEHZ_COEFFS + lane;
    const u64 poly = blockIdx.y;
    const bool live = n < N;
    const u64 *x = in + poly * in_pstride;
    for (int l = w; l < c.L; l += 4) {
        const u64 p = primes[c.q_id[l]].p;
        const Shoup pre = c.ext_pre[l];
        y[l * BEHZ_COEFFS + lane] = live ? mul_shoup(x[(u64)l * N + n], pre.op, pre.quo, p) : 0;
    }
    __syncthreads();
    for (int o = w; o <= c.nBsk; o += 4) {
        U128 acc{0, 0};
        const u64 *row = c.q2bsk + (u64)o * c.L;
        for (int l = 0; l < c.L; l++) mac128(acc, y[l * BEHZ_COEFFS + lane], row[l]);
        u64 r;
        if (o == c.nBsk) r = acc.lo & 0xFFFFFFFFull; // mod m_tilde = 2^32
        else r = barrett128(acc.lo, acc.hi, mod_of(primes[c.bsk_id[o]]));
        sres[o * BEHZ_COEFFS + lane] = r;
    }
    __syncthreads();
    const u64 r_mt = (sres[c.nBsk * BEHZ_COEFFS + lane] * c.neg_inv_q_mod_mt) & 0xFFFFFFFFull;
    for (int o = w; o < c.nBsk; o += 4) {
        const PrimeDesc &pd = primes[c.bsk_id[o]];
        const Mod m = mod_of(pd);
        u64 temp = r_mt;
        if (temp >= (u64(1) << 31)) temp += m.p - (u64(1) << 32); // centred remainder
        // (input + q * r) * m_tilde^-1 mod Bsk_o
        U128 acc{sres[o * BEHZ_COEFFS + lane], 0};
        mac128(acc, temp, c.prod_q_mod_bsk[o]);
        u64 v = barrett128(acc.lo, acc.hi, m);
        const Shoup im = c.inv_mt_mod_bsk[o];
        if (live) out[poly * out_pstride + (u64)o * N + n] = mul_shoup(v, im.op, im.quo, m.p);
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
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const u64 n = (u64)blockIdx.x * BEHZ_COEFFS + lane;
    const u64 poly = blockIdx.y;
    const bool live = n < N;
    for (int l = w; l < c.L; l += 4) {
        const u64 p = primes[c.q_id[l]].p;
        const Shoup pre = c.floor_pre[l];
        y[l * BEHZ_COEFFS + lane] = live ? mul_shoup(dq[poly * dq_pstride + (u64)l * N + n], pre.op, pre.quo, p) : 0;
    }
    __syncthreads();
    for (int o = w; o < c.nBsk; o += 4) {
        const PrimeDesc &pd = primes[c.bsk_id[o]];
        const Mod m = mod_of(pd);
        U128 acc{0, 0};
        const u64 *row = c.q2bsk + (u64)o * c.L;
        for (int l = 0; l < c.L; l++) mac128(acc, y[l * BEHZ_COEFFS + lane], row[l]);
        const u64 conv = barrett128(acc.lo, acc.hi, m);
        const Shoup tb = c.t_mod_bsk[o], iq = c.inv_q_mod_bsk[o];
        const u64 xb = live ? mul_shoup(db[poly * db_pstride + (u64)o * N + n], tb.op, tb.quo, m.p) : 0;
        const u64 z = mul_shoup(xb + (m.p - conv), iq.op, iq.quo, m.p);
        if (o < c.nB) {
            const Shoup bp = c.B_pre[o];
            u[o * BEHZ_COEFFS + lane] = mul_shoup(z, bp.op, bp.quo, m.p);
        } else {
            zsk[lane] = z;
        }
    }
    __syncthreads();
    // alpha_sk (every wave recomputes it for its own lane: nB MACs, saves a barrier)
    const PrimeDesc &psk = primes[c.bsk_id[c.nB]];
    const Mod msk = mod_of(psk);
    U128 a{0, 0};
    for (int b = 0; b < c.nB; b++) mac128(a, u[b * BEHZ_COEFFS + lane], c.B2msk[b]);
    const u64 conv_sk = barrett128(a.lo, a.hi, msk);
    const u64 alpha = mul_shoup(conv_sk + (msk.p - zsk[lane]), c.inv_B_mod_msk.op, c.inv_B_mod_msk.quo, msk.p);
    const bool neg = alpha > (msk.p >> 1);
    for (int l = w; l < c.L; l += 4) {
        const Mod m = mod_of(primes[c.q_id[l]]);
        U128 acc{0, 0};
        const u64 *row = c.B2q + (u64)l * c.nB;
        for (int b = 0; b < c.nB; b++) mac128(acc, u[b * BEHZ_COEFFS + lane], row[b]);
        const u64 pb = c.prod_B_mod_q[l];
        if (neg) mac128(acc, msk.p - alpha, pb);      // alpha represents a negative value
        else mac128(acc, alpha, m.p - pb);
        if (live) out[poly * out_pstride + (u64)l * N + n] = barrett128(acc.lo, acc.hi, m);
    }
}

void launch_behz_extend(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, const BehzDev &c, u64 N, u64 polys, hipStream_t s) {
    if (!polys) return;
    size_t lds = (size_t)(c.L + c.nBsk + 1) * BEHZ_COEFFS * sizeof(u64);
    TROY_LAUNCH(behz_extend_kernel, dim3(ceil_div(N, BEHZ_COEFFS), (unsigned)polys), dim3(BEHZ_THREADS), lds, s, in, in_pstride, out, out_pstride, primes, c, N);
    launch_check("behz_extend_kernel");
}
void launch_behz_floor_sk(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, const BehzDev &c, u64 N,
                          u64 polys, hipStream_t s) {
    if (!polys) return;
    size_t lds = (size_t)(c.L + c.nB + 1) * BEHZ_COEFFS * sizeof(u64);
    TROY_LAUNCH(behz_floor_sk_kernel, dim3(ceil_div(N, BEHZ_COEFFS), (unsigned)polys), dim3(BEHZ_THREADS), lds, s, dq, dq_pstride, db, db_pstride, out, out_pstride,
                primes, c, N);
    launch_check("behz_floor_sk_kernel");
}

} // namespace troyhip
