// poly.hip -- streaming polynomial kernels for gfx950: element-wise modular ops, ciphertext tensor,
// Galois permutations, mod-switch / rescale, and the key-switch MAC / mod-down kernels.
//
// Replaces: gAdd/gSub/gNegatePolyCoeffmod, gDyadicProductCoeffmod,
// gDyadicConvolutionCoeffmod, gMultiplyPolyScalarCoeffmod (src/kernelutils.cu:30-328, 495-535),
// gApplyGalois / gApplyGaloisNtt (src/utils/galois_cuda.cu:139-196), gDivideAndRoundqLastInplace,
// gDivideAndRoundqLastNttInplaceStepA/B, gModTAndDivideqLastInplace (src/utils/rns_cuda.cu:271-345,
// 579-604) and gSwitchKeyInplaceUtilA..G (src/evaluator_cuda.cu:1012-1161).
//
// All of these are HBM-bound streams.  Unlike the reference (one thread per coefficient index with the
// limb/poly loops inside the thread, 128 blocks in flight) every kernel here exposes the full
// batch x limb x coefficient space to the grid, and consecutive lanes touch consecutive coefficients
// (8-16 B per lane) so each wave issues full 512 B - 1 KiB transactions.
#include "kernels.h"

namespace troyhip {

#define EW_THREADS 256

__device__ __forceinline__ const PrimeDesc &prime_of(const PrimeDesc *primes, const LimbMap &m, u64 row) {
    return primes[m.id[(row / m.inner) % m.period]];
}
// the same for a 32-bit row index that comes from the block id (workgroup-uniform: two short divisions per workgroup instead of two 64-bit ones per thread)
__device__ __forceinline__ const PrimeDesc &prime_of_row(const PrimeDesc *primes, const LimbMap &m, u32 row) { return primes[m.id[(row / m.inner) % m.period]]; }
// Row-structured element-wise kernels (round 6): grid x = pairs of coefficients of a row, y = rows (a stride loop beyond 65535 rows).  The flat forms derived the row
// and its prime from 64-bit divisions per thread, about 200 instructions next to an addition or one product -- they were issue-bound streaming kernels.
static dim3 row_grid(int logn, u64 rows) { return dim3((unsigned)ceil_div((u64(1) << logn) / 2 + ((u64(1) << logn) & 1), EW_THREADS), (unsigned)std::min<u64>(rows, 65535)); }

// ---------------------------------------------------------------- element-wise
// op: 0 add, 1 sub, 2 negate(a), 3 dyadic product a*b (Barrett-128)
template <int OP> __global__ __launch_bounds__(EW_THREADS) void ew_kernel(const u64 *a, const u64 *b, u64 *out, const PrimeDesc *primes, LimbMap map, int logn, u32 rows) {
    const u32 n = (blockIdx.x * EW_THREADS + threadIdx.x) * 2;
    if (n >= (1u << logn)) return;
    for (u32 row = blockIdx.y; row < rows; row += gridDim.y) {
        const PrimeDesc &pd = prime_of_row(primes, map, row);
        const u64 p = pd.p, i = ((u64)row << logn) + n;
        ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(a + i), y{0, 0}, r;
        if (OP != 2) y = *reinterpret_cast<const ulonglong2 *>(b + i);
        if (OP == 0) { r.x = addmod(x.x, y.x, p); r.y = addmod(x.y, y.y, p); }
        else if (OP == 1) { r.x = submod(x.x, y.x, p); r.y = submod(x.y, y.y, p); }
        else if (OP == 2) { r.x = negmod(x.x, p); r.y = negmod(x.y, p); }
        else { const Mod m = mod_of(pd); r.x = mulmod(x.x, y.x, m); r.y = mulmod(x.y, y.y, m); }
        *reinterpret_cast<ulonglong2 *>(out + i) = r;
    }
}
void launch_ew(int op, const u64 *a, const u64 *b, u64 *out, const PrimeDesc *primes, const LimbMap &map, int logn, u64 rows, hipStream_t s) {
    if (!rows) return;
    if (logn < 1 || rows >> 32) throw Error(ST_LOGIC_ERROR, "element-wise: row shape");
    const dim3 grid = row_grid(logn, rows);
    switch (op) {
    case 0: TROY_LAUNCH(HIP_KERNEL_NAME(ew_kernel<0>), grid, dim3(EW_THREADS), 0, s, a, b, out, primes, map, logn, (u32)rows); break;
    case 1: TROY_LAUNCH(HIP_KERNEL_NAME(ew_kernel<1>), grid, dim3(EW_THREADS), 0, s, a, b, out, primes, map, logn, (u32)rows); break;
    case 2: TROY_LAUNCH(HIP_KERNEL_NAME(ew_kernel<2>), grid, dim3(EW_THREADS), 0, s, a, b, out, primes, map, logn, (u32)rows); break;
    default: TROY_LAUNCH(HIP_KERNEL_NAME(ew_kernel<3>), grid, dim3(EW_THREADS), 0, s, a, b, out, primes, map, logn, (u32)rows); break;
    }
    launch_check("ew_kernel");
}

// x[row][n] *= scalar[row % limbs]  (scalars already reduced mod the limb's prime)
struct ScalarArgs { u64 s[64]; };
__global__ __launch_bounds__(EW_THREADS) void mul_scalar_kernel(u64 *x, const PrimeDesc *primes, LimbMap map, ScalarArgs sc, int logn, u32 rows) {
    const u32 n = (blockIdx.x * EW_THREADS + threadIdx.x) * 2;
    if (n >= (1u << logn)) return;
    for (u32 row = blockIdx.y; row < rows; row += gridDim.y) {
        const PrimeDesc &pd = prime_of_row(primes, map, row);
        const Mod m = mod_of(pd);
        const u64 s = sc.s[(row / map.inner) % map.period], i = ((u64)row << logn) + n;
        ulonglong2 v = *reinterpret_cast<ulonglong2 *>(x + i);
        v.x = mulmod(v.x, s, m);
        v.y = mulmod(v.y, s, m);
        *reinterpret_cast<ulonglong2 *>(x + i) = v;
    }
}
void launch_mul_scalar(u64 *x, const PrimeDesc *primes, const LimbMap &map, const u64 *scalars, int logn, u64 rows, hipStream_t s) {
    if (!rows) return;
    if (logn < 1 || rows >> 32) throw Error(ST_LOGIC_ERROR, "element-wise: row shape");
    ScalarArgs sc;
    for (unsigned i = 0; i < 64; i++) sc.s[i] = i < map.period ? scalars[i] : 0;
    TROY_LAUNCH(mul_scalar_kernel, row_grid(logn, rows), dim3(EW_THREADS), 0, s, x, primes, map, sc, logn, (u32)rows);
    launch_check("mul_scalar_kernel");
}

// ct (x) plaintext in NTT form: out[b][i][l][n] = a[b][i][l][n] * plain[(b)][l][n]   (multiplyPlainNtt)
__global__ __launch_bounds__(EW_THREADS) void mul_plain_kernel(u64 *a, const u64 *plain, const PrimeDesc *primes, LimbMap map, int logn, u32 limbs, u32 rows,
                                                               u32 rows_per_item, u64 plain_bstride) {
    const u32 n = (blockIdx.x * EW_THREADS + threadIdx.x) * 2;
    if (n >= (1u << logn)) return;
    for (u32 row = blockIdx.y; row < rows; row += gridDim.y) {
        const u32 l = row % limbs;
        const Mod m = mod_of(prime_of_row(primes, map, row));
        const u64 *pl = rows_per_item ? plain + (u64)(row / rows_per_item) * plain_bstride : plain;
        const u64 i = ((u64)row << logn) + n;
        ulonglong2 v = *reinterpret_cast<ulonglong2 *>(a + i);
        const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(pl + ((u64)l << logn) + n);
        v.x = mulmod(v.x, w.x, m);
        v.y = mulmod(v.y, w.y, m);
        *reinterpret_cast<ulonglong2 *>(a + i) = v;
    }
}
void launch_mul_plain(u64 *a, const u64 *plain, const PrimeDesc *primes, const LimbMap &map, int logn, u64 limbs, u64 rows, hipStream_t s, u64 rows_per_item,
                      u64 plain_bstride) {
    if (!rows) return;
    if (logn < 1 || rows >> 32 || rows_per_item >> 32) throw Error(ST_LOGIC_ERROR, "element-wise: row shape");
    TROY_LAUNCH(mul_plain_kernel, row_grid(logn, rows), dim3(EW_THREADS), 0, s, a, plain, primes, map, logn, (u32)limbs, (u32)rows, (u32)rows_per_item, plain_bstride);
    launch_check("mul_plain_kernel");
}

// sum_i ct_i (x) plain_i in NTT form, one pass: out[b][k][l][n] = sum_i ct_i[b][k][l][n] * plain_i[l][n] mod p.  The loop of multiplyPlain +
// addInplace a linear layer runs per output block (app/LinearHelperCKKS.cuh:227-248, 536-556) costs 2 count + (count - 1) kernels and moves
// every partial product through HBM; here every operand is read once, the products accumulate in 128 bits (count * p^2 < 2^128) and
// one Barrett step gives the same canonical residue (sums and products of residues are exact whatever the order of reductions).
// Grid (x: pairs of coefficients of a row, y: row = polynomial x limb, z: batch item): every index comes from the block id -- no 64-bit division per thread, the
// prime and its Barrett constants are workgroup-uniform (scalar registers).  The flat form spent ~150 of its ~300 VALU instructions per thread on i / item_words
// and row % limbs, which made a streaming kernel issue-bound (0.595 of the HBM peak at the 128 x 128 matmul; round 6).
#ifndef MPA_PAIRS
#define MPA_PAIRS 2 // pairs of coefficients per thread (the second pair half a row away: both sets of loads are in flight before the first product)
#endif
__global__ __launch_bounds__(EW_THREADS) void mul_plain_acc_kernel(MulPlainAccArgs x, u64 *out, u64 out_bstride, const PrimeDesc *primes, LimbMap map, int logn, u32 limbs, u32 batch) {
    const u32 n = (blockIdx.x * EW_THREADS + threadIdx.x) * 2; // two coefficients per pair
    const u32 N = 1u << logn, span = N / MPA_PAIRS < 2 ? 2 : N / MPA_PAIRS; // pair j of this thread: coefficients n + j span, n + j span + 1 (tiny rings: fewer pairs)
    if (n >= span) return;
    const u32 row = blockIdx.y, l = row % limbs;
    const u64 r = ((u64)row << logn) + n;
    const Mod m = mod_of(prime_of(primes, map, l));
    for (u32 b = blockIdx.z; b < batch; b += gridDim.z) { // (batches beyond the grid's z extent: a stride loop)
        U128 s[MPA_PAIRS][2];
#pragma unroll
        for (int j = 0; j < MPA_PAIRS; j++) s[j][0] = s[j][1] = U128{0, 0};
        for (int t = 0; t < x.count; t++) {
            ulonglong2 v[MPA_PAIRS], w[MPA_PAIRS];
#pragma unroll
            for (int j = 0; j < MPA_PAIRS; j++) {
                if (j && n + j * span >= N) { v[j] = w[j] = ulonglong2{0, 0}; continue; }
                v[j] = *reinterpret_cast<const ulonglong2 *>(x.ct[t] + (u64)b * x.ct_bstride[t] + r + j * span);
                w[j] = *reinterpret_cast<const ulonglong2 *>(x.plain[t] + ((u64)l << logn) + n + j * span);
            }
#pragma unroll
            for (int j = 0; j < MPA_PAIRS; j++) {
                mac128(s[j][0], v[j].x, w[j].x);
                mac128(s[j][1], v[j].y, w[j].y);
            }
        }
#pragma unroll
        for (int j = 0; j < MPA_PAIRS; j++) {
            if (j && n + j * span >= N) continue;
            ulonglong2 o;
            o.x = barrett128(s[j][0].lo, s[j][0].hi, m);
            o.y = barrett128(s[j][1].lo, s[j][1].hi, m);
            *reinterpret_cast<ulonglong2 *>(out + (u64)b * out_bstride + r + j * span) = o;
        }
    }
}
void launch_mul_plain_acc(const MulPlainAccArgs &x, u64 *out, u64 out_bstride, const PrimeDesc *primes, const LimbMap &map, int logn, u64 limbs, u64 size, u64 batch,
                          hipStream_t s) {
    if (!batch || !size || !limbs) return;
    if (logn < 1) throw Error(ST_LOGIC_ERROR, "mul_plain_acc: two coefficients per access");
    const u64 span = std::max<u64>((u64(1) << logn) / MPA_PAIRS, 2);
    const unsigned gx = (unsigned)ceil_div(span / 2, EW_THREADS), gz = (unsigned)std::min<u64>(batch, 65535);
    TROY_LAUNCH(mul_plain_acc_kernel, dim3(gx, (unsigned)(size * limbs), gz), dim3(EW_THREADS), 0, s, x, out, out_bstride, primes, map, logn, (u32)limbs, (u32)batch);
    launch_check("mul_plain_acc_kernel");
}

// ---------------------------------------------------------------- plaintext operands
// addPlain / subPlain on c0 (evaluator_cuda.cu:1654-1720):
//   BFV  (scalingvariant_cuda.cu:21-176 multiplyAdd/SubPlainWithScalingVariant): c0_l[j] +-= m_j * Delta_l + floor((m_j (q mod t) + (t+1)/2) / t)
//   BGV  (addPlainWithoutScalingVariant): c0_l[j] +-= (m_j * correction_factor mod t) mod q_l
// one thread = one plaintext coefficient of one item, looping over the limbs (the 128/64 division is done once)
template <int KIND, bool SUB> __global__ __launch_bounds__(EW_THREADS) void add_plain_kernel(u64 *ct0, u64 ct_bstride, const u64 *plain, PlainArgs a) {
    const u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    if (idx >= a.items * a.n_coeffs) return;
    const u64 b = idx / a.n_coeffs, j = idx % a.n_coeffs;
    const u64 mval = plain[b * a.plain_bstride + j];
    const Mod tm{a.t_p, a.t_cr0, a.t_cr1};
    u64 fix = 0, pc = 0;
    if (KIND == 0) {
        u64 lo = mval * a.q_mod_t, hi = mulhi64(mval, a.q_mod_t);
        lo += a.thr;
        hi += lo < a.thr;
        fix = div128(lo, hi, tm);
    } else {
        pc = a.cf == 1 ? mval : mulmod(mval, a.cf, tm);
    }
    u64 *c = ct0 + b * ct_bstride + j;
    for (u64 l = 0; l < a.limbs; l++, c += u64(1) << a.logn) {
        const Mod m = mod_of(a.primes[a.map.id[l]]);
        u64 v;
        if (KIND == 0) {
            u64 lo = mval * a.delta[l], hi = mulhi64(mval, a.delta[l]);
            lo += fix;
            hi += lo < fix;
            v = barrett128(lo, hi, m);
        } else {
            v = barrett64(pc, m);
        }
        *c = SUB ? submod(*c, v, m.p) : addmod(*c, v, m.p);
    }
}
// CKKS: the plaintext is an RNS polynomial [limbs][N] in NTT form at the level of the ciphertext
template <bool SUB> __global__ __launch_bounds__(EW_THREADS) void add_plain_rows_kernel(u64 *ct0, u64 ct_bstride, const u64 *plain, PlainArgs a) {
    const u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    const u64 per = a.limbs << a.logn;
    if (idx >= a.items * per) return;
    const u64 b = idx / per, r = idx % per;
    const u64 p = a.primes[a.map.id[r >> a.logn]].p;
    u64 *c = ct0 + b * ct_bstride + r;
    const u64 v = plain[b * a.plain_bstride + r];
    *c = SUB ? submod(*c, v, p) : addmod(*c, v, p);
}
void launch_add_plain(int kind, bool sub, u64 *ct0, u64 ct_bstride, const u64 *plain, const PlainArgs &a, hipStream_t s) {
    if (kind == 1) {
        const u64 total = a.items * a.limbs << a.logn;
        if (!total) return;
        dim3 grid(ceil_div(total, EW_THREADS)), blk(EW_THREADS);
        if (sub) TROY_LAUNCH(HIP_KERNEL_NAME(add_plain_rows_kernel<true>), grid, blk, 0, s, ct0, ct_bstride, plain, a);
        else TROY_LAUNCH(HIP_KERNEL_NAME(add_plain_rows_kernel<false>), grid, blk, 0, s, ct0, ct_bstride, plain, a);
    } else {
        const u64 total = a.items * a.n_coeffs;
        if (!total) return;
        dim3 grid(ceil_div(total, EW_THREADS)), blk(EW_THREADS);
        if (kind == 0 && !sub) TROY_LAUNCH(HIP_KERNEL_NAME(add_plain_kernel<0, false>), grid, blk, 0, s, ct0, ct_bstride, plain, a);
        else if (kind == 0) TROY_LAUNCH(HIP_KERNEL_NAME(add_plain_kernel<0, true>), grid, blk, 0, s, ct0, ct_bstride, plain, a);
        else if (!sub) TROY_LAUNCH(HIP_KERNEL_NAME(add_plain_kernel<2, false>), grid, blk, 0, s, ct0, ct_bstride, plain, a);
        else TROY_LAUNCH(HIP_KERNEL_NAME(add_plain_kernel<2, true>), grid, blk, 0, s, ct0, ct_bstride, plain, a);
    }
    launch_check("add_plain_kernel");
}
// Lifting of plaintext coefficients to the ciphertext base (gMultiplyPlainNormalUtilA/B, evaluator_cuda.cu:1722-1755, and the
// same step of transformToNttInplace(Plaintext)): m -> m for m < (t+1)/2, else m + (q - t).  Modulo q_l that is
// m - t (q = 0 mod q_l); stored canonical -- the reference leaves m + (q_l - t) unreduced in fast-lift mode and feeds the
// lazy NTT, whose output is canonical either way.
__global__ __launch_bounds__(EW_THREADS) void plain_lift_kernel(const u64 *plain, u64 *lifted, PlainArgs a) {
    const u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    const u64 per = a.limbs << a.logn;
    if (idx >= a.items * per) return;
    const u64 b = idx / per, r = idx % per, l = r >> a.logn, j = r & ((u64(1) << a.logn) - 1);
    u64 v = 0;
    if (j < a.n_coeffs) {
        const Mod m = mod_of(a.primes[a.map.id[l]]);
        const u64 mval = plain[b * a.plain_bstride + j];
        v = barrett64(mval, m);
        if (mval >= a.thr) v = submod(v, barrett64(a.t_p, m), m.p);
    }
    lifted[idx] = v;
}
void launch_plain_lift(const u64 *plain, u64 *lifted, const PlainArgs &a, hipStream_t s) {
    const u64 total = a.items * a.limbs << a.logn;
    if (!total) return;
    TROY_LAUNCH(plain_lift_kernel, dim3(ceil_div(total, EW_THREADS)), dim3(EW_THREADS), 0, s, plain, lifted, a);
    launch_check("plain_lift_kernel");
}

// ---------------------------------------------------------------- ciphertext tensor (a-3 / A.7)
// out[b][i][l][n] = sum_{j+k=i} a[b][j][l][n] * b[b][k][l][n] mod p_l.  Inputs may be lazy (< 4p).
// One thread = one (b, l, n); all S1+S2 operands live in registers; 3..5 outputs.
// Grid (x: pairs of coefficients of a row, y: limb, z: batch item): the indices come from the block id (the flat form's i / limbs and i % limbs were 64-bit
// divisions per thread -- a third of the instructions of a kernel that is issue-bound, 0.94 of the VALU peak); the terms of one output accumulate in 128 bits
// (at most three products of operands below 2^63) and are reduced ONCE: the same canonical residue as the reference's sum of reduced products.
template <int S1, int S2> __global__ __launch_bounds__(EW_THREADS) void tensor_kernel(const u64 *a, const u64 *b, u64 *out, u64 a_bstride, u64 b_bstride,
                                                                                       const PrimeDesc *primes, LimbMap map, int logn, u32 limbs, u32 batch) {
    static_assert(S1 <= 3 && S2 <= 3, "three lazy products fit 128 bits");
    const u32 n = (blockIdx.x * EW_THREADS + threadIdx.x) * 2; // two coefficients (16 bytes) per thread
    const u64 N = u64(1) << logn;
    if (n >= N) return;
    const u32 l = blockIdx.y;
    const Mod m = mod_of(primes[map.id[l]]);
    for (u32 bb = blockIdx.z; bb < batch; bb += gridDim.z) {
        ulonglong2 x[S1], y[S2];
#pragma unroll
        for (int j = 0; j < S1; j++) x[j] = *reinterpret_cast<const ulonglong2 *>(a + (u64)bb * a_bstride + ((u64)j * limbs + l) * N + n);
#pragma unroll
        for (int k = 0; k < S2; k++) y[k] = *reinterpret_cast<const ulonglong2 *>(b + (u64)bb * b_bstride + ((u64)k * limbs + l) * N + n);
#pragma unroll
        for (int d = 0; d < S1 + S2 - 1; d++) {
            U128 s0{0, 0}, s1{0, 0};
#pragma unroll
            for (int j = 0; j < S1; j++) {
                const int k = d - j;
                if (k >= 0 && k < S2) {
                    mac128(s0, x[j].x, y[k].x);
                    mac128(s1, x[j].y, y[k].y);
                }
            }
            ulonglong2 r;
            r.x = barrett128(s0.lo, s0.hi, m);
            r.y = barrett128(s1.lo, s1.hi, m);
            *reinterpret_cast<ulonglong2 *>(out + ((u64)bb * (S1 + S2 - 1) + d) * limbs * N + (u64)l * N + n) = r;
        }
    }
}
// any sizes (destination up to SEAL_CIPHERTEXT_SIZE_MAX = 16 polynomials, evaluator_cuda.cu:342-364 / :407-423): operands are re-read per
// output polynomial (they sit in L1/L2); the register-resident forms above cover the sizes real circuits use
__global__ __launch_bounds__(EW_THREADS) void tensor_any_kernel(const u64 *a, const u64 *b, u64 *out, u64 a_bstride, u64 b_bstride, int s1, int s2, const PrimeDesc *primes,
                                                                LimbMap map, int logn, u64 limbs, u64 total) {
    u64 i = ((u64)blockIdx.x * EW_THREADS + threadIdx.x) * 2;
    if (i >= total) return;
    const u64 N = u64(1) << logn;
    u64 n = i & (N - 1), bl = i >> logn, l = bl % limbs, bb = bl / limbs;
    const Mod m = mod_of(primes[map.id[l]]);
    const int ds = s1 + s2 - 1;
    for (int d = 0; d < ds; d++) {
        ulonglong2 r{0, 0};
        const int j0 = d - (s2 - 1) > 0 ? d - (s2 - 1) : 0, j1 = d < s1 - 1 ? d : s1 - 1;
        for (int j = j0; j <= j1; j++) {
            const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(a + bb * a_bstride + (j * limbs + l) * N + n);
            const ulonglong2 y = *reinterpret_cast<const ulonglong2 *>(b + bb * b_bstride + ((d - j) * limbs + l) * N + n);
            r.x = addmod(r.x, mulmod(x.x, y.x, m), m.p);
            r.y = addmod(r.y, mulmod(x.y, y.y, m), m.p);
        }
        *reinterpret_cast<ulonglong2 *>(out + (bb * ds + d) * limbs * N + l * N + n) = r;
    }
}
void launch_tensor(int s1, int s2, const u64 *a, const u64 *b, u64 *out, u64 a_bstride, u64 b_bstride, const PrimeDesc *primes, const LimbMap &map,
                   int logn, u64 limbs, u64 batch, hipStream_t s) {
    u64 total = (batch * limbs) << logn;
    if (!total) return;
    if (logn < 1) throw Error(ST_LOGIC_ERROR, "tensor: N < 2");
    dim3 grid(ceil_div(total / 2, EW_THREADS)), blk(EW_THREADS);
    const dim3 grid3((unsigned)ceil_div((u64(1) << logn) / 2, EW_THREADS), (unsigned)limbs, (unsigned)std::min<u64>(batch, 65535));
#define TENSOR_CASE(A, B) if (s1 == A && s2 == B) { TROY_LAUNCH(HIP_KERNEL_NAME(tensor_kernel<A, B>), grid3, blk, 0, s, a, b, out, a_bstride, b_bstride, primes, map, logn, (u32)limbs, (u32)batch); launch_check("tensor_kernel"); return; }
    TENSOR_CASE(2, 2) TENSOR_CASE(2, 3) TENSOR_CASE(3, 2) TENSOR_CASE(3, 3) TENSOR_CASE(1, 1) TENSOR_CASE(1, 2) TENSOR_CASE(2, 1) TENSOR_CASE(1, 3) TENSOR_CASE(3, 1)
#undef TENSOR_CASE
    if (s1 < 1 || s2 < 1 || s1 + s2 - 1 > 16) throw Error(ST_INVALID_ARGUMENT, "invalid size");
    TROY_LAUNCH(tensor_any_kernel, grid, blk, 0, s, a, b, out, a_bstride, b_bstride, s1, s2, primes, map, logn, limbs, total);
    launch_check("tensor_any_kernel");
}

// ---------------------------------------------------------------- Galois (a-6 / A.11)
// coefficient form: out[(i*g) mod N] = +-in[i]   (src/utils/galois.cpp:143-162).  One launch covers `batch` items of `limbs` rows
// each; item b reads in + b * in_bstride and writes out + b * out_bstride.
// Written as a GATHER: coefficient n goes to position n g mod N with the sign (-1)^floor(n g / N), i.e. output j takes input n0 = j g^-1 mod 2N when
// n0 < N and the negated input n0 - N otherwise (exactly one of j, j + N is n g mod 2N for an n below N, g being odd).  The scatter form wrote 64
// different cache lines per wave-store (0.28-0.30 of the HBM roofline at N = 2^16); scattered READS of a row that fits L2 are served from there and the
// stores are whole lines.  elt_inv = g^-1 mod 2N comes from the host.
// (the grid-indexed form -- x: coefficient, y: limb, z: batch item, as galois_ntt_kernel below -- measured 2 % SLOWER here: the gather is bound by its scattered L2 reads,
// not by the index arithmetic; profiles/r06_elementwise_ab.txt)
__global__ __launch_bounds__(EW_THREADS) void galois_coeff_kernel(const u64 *in, u64 in_bstride, u64 *out, u64 out_bstride, const PrimeDesc *primes, LimbMap map, int logn,
                                                                  uint32_t elt_inv, u64 limbs, u64 total) {
    u64 i = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i >= total) return;
    const u64 N = u64(1) << logn;
    u64 row = i >> logn, j = i & (N - 1), b = row / limbs, l = row % limbs;
    const u64 p = primes[map.id[l]].p;
    const u64 n0 = (j * elt_inv) & (2 * N - 1);
    u64 v = in[b * in_bstride + (l << logn) + (n0 & (N - 1))];
    if (n0 >> logn) v = negmod(v, p);
    out[b * out_bstride + (l << logn) + j] = v;
}
// kNegacyclicShiftPolyCoeffmod (kernelutils.cu; CPU polyarithsmallmod.cpp:128-152): out[(i + shift) mod N] = +-in[i]
__global__ __launch_bounds__(EW_THREADS) void negacyclic_shift_kernel(const u64 *in, u64 in_bstride, u64 *out, u64 out_bstride, const PrimeDesc *primes, LimbMap map, int logn,
                                                                      u64 shift, u64 rows_per_item, u64 limbs, u64 total) {
    u64 i = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i >= total) return;
    const u64 N = u64(1) << logn;
    u64 row = i >> logn, n = i & (N - 1), b = row / rows_per_item, r = row % rows_per_item;
    const u64 p = primes[map.id[r % limbs]].p;
    const u64 raw = n + shift, idx = raw & (N - 1);
    u64 v = in[b * in_bstride + (r << logn) + n];
    if ((raw & N) && v) v = p - v;
    out[b * out_bstride + (r << logn) + idx] = v;
}
void launch_negacyclic_shift(const u64 *in, u64 in_bstride, u64 *out, u64 out_bstride, const PrimeDesc *primes, const LimbMap &map, int logn, u64 shift, u64 rows_per_item,
                             u64 limbs, u64 batch, hipStream_t s) {
    u64 total = (batch * rows_per_item) << logn;
    if (!total) return;
    TROY_LAUNCH(negacyclic_shift_kernel, dim3(ceil_div(total, EW_THREADS)), dim3(EW_THREADS), 0, s, in, in_bstride, out, out_bstride, primes, map, logn, shift, rows_per_item,
                limbs, total);
    launch_check("negacyclic_shift_kernel");
}
// NTT form: out[i] = in[bitrev(((g * bitrev(i + N, logN+1)) >> 1) mod N, logN)]   (galois.cpp:18-35).  Grid x: coefficient, y: limb, z: batch item -- no 64-bit
// division per thread (round 6: 957 -> 860 us in the CKKS chain)
__global__ __launch_bounds__(EW_THREADS) void galois_ntt_kernel(const u64 *in, u64 in_bstride, u64 *out, u64 out_bstride, int logn, uint32_t elt, u32 batch) {
    const u32 n = blockIdx.x * EW_THREADS + threadIdx.x;
    const u32 N = 1u << logn;
    if (n >= N) return;
    const u32 l = blockIdx.y;
    uint32_t rev = __brev((uint32_t)(n + N)) >> (32 - (logn + 1));
    u64 raw = (((u64)elt * rev) >> 1) & (N - 1);
    uint32_t src = logn ? (__brev((uint32_t)raw) >> (32 - logn)) : 0;
    for (u32 b = blockIdx.z; b < batch; b += gridDim.z) out[(u64)b * out_bstride + ((u64)l << logn) + n] = in[(u64)b * in_bstride + ((u64)l << logn) + src];
}
void launch_galois(bool ntt_form, const u64 *in, u64 in_bstride, u64 *out, u64 out_bstride, const PrimeDesc *primes, const LimbMap &map, int logn, uint32_t elt, u64 limbs,
                   u64 batch, hipStream_t s) {
    u64 total = (batch * limbs) << logn;
    if (!total) return;
    const dim3 grid((unsigned)ceil_div(u64(1) << logn, EW_THREADS), (unsigned)limbs, (unsigned)std::min<u64>(batch, 65535)), blk(EW_THREADS);
    if (ntt_form) TROY_LAUNCH(galois_ntt_kernel, grid, blk, 0, s, in, in_bstride, out, out_bstride, logn, elt, (u32)batch);
    else {
        uint32_t inv = 1; // g^-1 mod 2N by Newton steps (g odd): x <- x (2 - g x), five steps reach 32 bits
        for (int it = 0; it < 5; it++) inv *= 2u - elt * inv;
        inv &= (uint32_t)((2u << logn) - 1);
        TROY_LAUNCH(galois_coeff_kernel, dim3(ceil_div(total, EW_THREADS)), blk, 0, s, in, in_bstride, out, out_bstride, primes, map, logn, inv, limbs, total);
    }
    launch_check("galois_kernel");
}

// ---------------------------------------------------------------- mod-switch / rescale (a-4 / A.10)
// kind 0: BFV divideAndRoundqLastInplace (rns.cpp:805-830); kind 2: BGV modTAndDivideqLastInplace (rns.cpp:1097-1140).
// in [polys][limbs][N] -> out [polys][limbs-1][N]
template <int KIND> __global__ __launch_bounds__(EW_THREADS) void modswitch_kernel(const u64 *in, u64 *out, ModSwitchArgs a) {
    u64 i = (u64)blockIdx.x * EW_THREADS + threadIdx.x; // over polys * (limbs-1) * N
    const u64 N = u64(1) << a.logn, nl = a.limbs - 1;
    if (i >= a.polys * nl * N) return;
    u64 n = i & (N - 1), pl = i >> a.logn, l = pl % nl, poly = pl / nl;
    const PrimeDesc &pd = a.primes[a.map.id[l]];
    const PrimeDesc &pq = a.primes[a.map.id[nl]];
    const Mod m = mod_of(pd);
    const u64 x = in[(poly * a.limbs + l) * N + n];
    const u64 xl = in[(poly * a.limbs + nl) * N + n];
    u64 r;
    if (KIND == 0) {
        u64 tl = addmod(xl, a.half, pq.p);                           // x_last + floor(q_last/2) mod q_last
        u64 tmp = submod(barrett64(tl, m), barrett64(a.half, m), m.p);
        r = mul_shoup(submod(x, tmp, m.p), a.inv_qlast[l], m.p);
    } else {
        const Mod tm{a.t_p, a.t_cr0, a.t_cr1};
        u64 negc = negmod(barrett64(xl, tm), tm.p);
        if (a.inv_qlast_mod_t != 1) negc = mulmod(negc, a.inv_qlast_mod_t, tm);
        u64 delta = mulmod(barrett64(negc, m), barrett64(pq.p, m), m);
        u64 v = x + 2 * m.p - barrett64(xl, m) - delta;
        r = mul_shoup(v, a.inv_qlast[l], m.p);
    }
    out[(poly * nl + l) * N + n] = r;
}
// CKKS divideAndRoundqLastNttInplace (rns.cpp:832-877), around two NTT launches:
//  step A: last[poly][n] (coefficient form, canonical) -> corr[poly][l][n] = ((last + half) mod q_last) mod q_l + (q_l - half mod q_l)
//  (NTT of corr over the L-1 limbs)
//  step B: out[poly][l][n] = (x[poly][l][n] + q_l - corr) * q_last^-1 mod q_l
__global__ __launch_bounds__(EW_THREADS) void rescale_stepA_kernel(const u64 *last, u64 last_pstride, u64 *corr, ModSwitchArgs a) {
    u64 i = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    const u64 N = u64(1) << a.logn, nl = a.limbs - 1;
    if (i >= a.polys * nl * N) return;
    u64 n = i & (N - 1), pl = i >> a.logn, l = pl % nl, poly = pl / nl;
    const PrimeDesc &pd = a.primes[a.map.id[l]];
    const PrimeDesc &pq = a.primes[a.map.id[nl]];
    const Mod m = mod_of(pd);
    u64 tl = addmod(last[poly * last_pstride + n], a.half, pq.p);
    corr[i] = barrett64(tl, m) + (m.p - barrett64(a.half, m));
}
__global__ __launch_bounds__(EW_THREADS) void rescale_stepB_kernel(const u64 *in, const u64 *corr, u64 *out, ModSwitchArgs a) {
    u64 i = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    const u64 N = u64(1) << a.logn, nl = a.limbs - 1;
    if (i >= a.polys * nl * N) return;
    u64 n = i & (N - 1), pl = i >> a.logn, l = pl % nl, poly = pl / nl;
    const u64 p = a.primes[a.map.id[l]].p;
    u64 x = in[(poly * a.limbs + l) * N + n];
    out[i] = mul_shoup(x + p - corr[i], a.inv_qlast[l], p);
}
// drop the last limb: out[poly][l][n] = in[poly][l][n], l < limbs-1  (modSwitchDropToNext)
__global__ __launch_bounds__(EW_THREADS) void drop_last_kernel(const u64 *in, u64 *out, int logn, u64 limbs, u64 total) {
    u64 i = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i >= total) return;
    const u64 N = u64(1) << logn, nl = limbs - 1;
    u64 n = i & (N - 1), pl = i >> logn, l = pl % nl, poly = pl / nl;
    out[i] = in[(poly * limbs + l) * N + n];
}
void launch_modswitch(int kind, const u64 *in, u64 *out, const ModSwitchArgs &a, hipStream_t s) {
    u64 total = a.polys * (a.limbs - 1) << a.logn;
    if (!total) return;
    dim3 grid(ceil_div(total, EW_THREADS)), blk(EW_THREADS);
    if (kind == 0) TROY_LAUNCH(HIP_KERNEL_NAME(modswitch_kernel<0>), grid, blk, 0, s, in, out, a);
    else TROY_LAUNCH(HIP_KERNEL_NAME(modswitch_kernel<2>), grid, blk, 0, s, in, out, a);
    launch_check("modswitch_kernel");
}
void launch_rescale_stepA(const u64 *last, u64 last_pstride, u64 *corr, const ModSwitchArgs &a, hipStream_t s) {
    u64 total = a.polys * (a.limbs - 1) << a.logn;
    if (!total) return;
    TROY_LAUNCH(rescale_stepA_kernel, dim3(ceil_div(total, EW_THREADS)), dim3(EW_THREADS), 0, s, last, last_pstride, corr, a);
    launch_check("rescale_stepA_kernel");
}
void launch_rescale_stepB(const u64 *in, const u64 *corr, u64 *out, const ModSwitchArgs &a, hipStream_t s) {
    u64 total = a.polys * (a.limbs - 1) << a.logn;
    if (!total) return;
    TROY_LAUNCH(rescale_stepB_kernel, dim3(ceil_div(total, EW_THREADS)), dim3(EW_THREADS), 0, s, in, corr, out, a);
    launch_check("rescale_stepB_kernel");
}
void launch_drop_last(const u64 *in, u64 *out, int logn, u64 limbs, u64 polys, hipStream_t s) {
    u64 total = polys * (limbs - 1) << logn;
    if (!total) return;
    TROY_LAUNCH(drop_last_kernel, dim3(ceil_div(total, EW_THREADS)), dim3(EW_THREADS), 0, s, in, out, logn, limbs, total);
    launch_check("drop_last_kernel");
}
// gather one limb out of every poly: out[poly][n] = in[poly*pstride + limb*N + n]
__global__ __launch_bounds__(EW_THREADS) void gather_limb_kernel(const u64 *in, u64 *out, int logn, u64 pstride, u64 limb, u64 total, u64 group, u64 gstride) {
    u64 i = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i >= total) return;
    u64 n = i & ((u64(1) << logn) - 1), poly = i >> logn;
    const u64 base = group ? (poly / group) * gstride + (poly % group) * pstride : poly * pstride;
    out[i] = in[base + (limb << logn) + n];
}
void launch_gather_limb(const u64 *in, u64 *out, int logn, u64 pstride, u64 limb, u64 polys, hipStream_t s, u64 group, u64 gstride) {
    u64 total = polys << logn;
    if (!total) return;
    TROY_LAUNCH(gather_limb_kernel, dim3(ceil_div(total, EW_THREADS)), dim3(EW_THREADS), 0, s, in, out, logn, pstride, limb, total, group, gstride);
    launch_check("gather_limb_kernel");
}

// ---------------------------------------------------------------- key switching (a-5 / A.9)
// D[b][i][j][n] = target[b][j][n] mod p_i   for i <= dl, j < dl   (evaluator.cpp:2432-2442; a copy when q_j <= p_i)
__global__ __launch_bounds__(EW_THREADS) void ks_expand_kernel(const u64 *target, u64 t_bstride, u64 *D, KsArgs a) {
    u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x; // over batch * dl * N (one input coefficient)
    const u64 N = u64(1) << a.logn;
    if (idx >= a.batch * a.dl * N) return;
    u64 n = idx & (N - 1), bj = idx >> a.logn, j = bj % a.dl, b = bj / a.dl;
    const u64 x = target[b * t_bstride + j * N + n];
    for (u64 i = 0; i <= a.dl; i++) {
        const Mod m = mod_of(a.primes[a.key_id[i]]);
        D[((b * (a.dl + 1) + i) * a.dl + j) * N + n] = barrett64(x, m);
    }
}
// ---- key-switch inner product ----
// One 61 x 61-bit product accumulated without a carry chain across words: the aligned partial products (lo*lo at bit 0,
// hi*hi at bit 64) go to accumulator A, the cross products (bit 32) to accumulator B; each v_mad_u64_u32 adds into a
// 64-bit word and its carry-out is counted in a separate 32-bit register.  7 instructions per multiply-accumulate
// (4 v_mad_u64_u32 + 3 v_addc) against ~27 for the compiler's 128-bit code.  Valid for < 64 terms of operands < 2^61.
struct MacAcc {
    u64 alo, ahi, blo;
    u32 acnt, bhi;
};
__device__ __forceinline__ void mac_zero(MacAcc &a) { a.alo = a.ahi = a.blo = 0; a.acnt = a.bhi = 0; }
__device__ __forceinline__ U128 mac_value(const MacAcc &a) {
    U128 v{a.alo, a.ahi + a.acnt};
    const u64 tlo = a.blo << 32, thi = (a.blo >> 32) | ((u64)a.bhi << 32);
    v.lo += tlo;
    v.hi += thi + (v.lo < tlo);
    return v;
}
#ifdef TROYHIP_CPU_EMUL
__device__ __forceinline__ void mac4(MacAcc (&a)[4], const u64 (&x)[4], const u64 (&k)[4]) {
    for (int i = 0; i < 4; i++) {
        const u64 xl = (u32)x[i], xh = x[i] >> 32, kl = (u32)k[i], kh = k[i] >> 32;
        u64 t = a[i].alo + xl * kl;
        a[i].acnt += t < a[i].alo;
        a[i].alo = t;
        t = a[i].blo + xl * kh;
        a[i].bhi += t < a[i].blo;
        a[i].blo = t;
        t = a[i].blo + xh * kl;
        a[i].bhi += t < a[i].blo;
        a[i].blo = t;
        a[i].ahi += xh * kh;
    }
}
#else
// four independent accumulators, instruction-interleaved: a carry writer and its reader are >= 3 instructions apart
// (VALU carry -> VALU hazard, see bfly.h), carries in VCC + three SGPR pairs
#define TROY_MAC4_COL(ACC, CNT, XW, KW)                                                                                   \
    asm("v_mad_u64_u32 %0, vcc, %11, %15, %0\n\t"                                                                        \
        "v_mad_u64_u32 %1, %8, %12, %16, %1\n\t"                                                                         \
        "v_mad_u64_u32 %2, %9, %13, %17, %2\n\t"                                                                         \
        "v_mad_u64_u32 %3, %10, %14, %18, %3\n\t"                                                                        \
        "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"                                                                          \
        "v_addc_co_u32 %5, %8, 0, %5, %8\n\t"                                                                            \
        "v_addc_co_u32 %6, %9, 0, %6, %9\n\t"                                                                            \
        "v_addc_co_u32 %7, %10, 0, %7, %10"                                                                               \
        : "+v"(a[0].ACC), "+v"(a[1].ACC), "+v"(a[2].ACC), "+v"(a[3].ACC), "+v"(a[0].CNT), "+v"(a[1].CNT), "+v"(a[2].CNT), "+v"(a[3].CNT),  \
          "=&s"(sb), "=&s"(sc), "=&s"(sd)                                                                                 \
        : "v"(XW(x[0])), "v"(XW(x[1])), "v"(XW(x[2])), "v"(XW(x[3])), "v"(KW(k[0])), "v"(KW(k[1])), "v"(KW(k[2])), "v"(KW(k[3]))          \
        : "vcc")
__device__ __forceinline__ u32 mac_lo32(u64 v) { return (u32)v; }
__device__ __forceinline__ u32 mac_hi32(u64 v) { return (u32)(v >> 32); }
__device__ __forceinline__ void mac4(MacAcc (&a)[4], const u64 (&x)[4], const u64 (&k)[4]) {
    u64 sb, sc, sd;
    TROY_MAC4_COL(alo, acnt, mac_lo32, mac_lo32);
    TROY_MAC4_COL(blo, bhi, mac_lo32, mac_hi32);
    TROY_MAC4_COL(blo, bhi, mac_hi32, mac_lo32);
    asm("v_mad_u64_u32 %0, vcc, %7, %11, %0\n\t"
        "v_mad_u64_u32 %1, %4, %8, %12, %1\n\t"
        "v_mad_u64_u32 %2, %5, %9, %13, %2\n\t"
        "v_mad_u64_u32 %3, %6, %10, %14, %3"
        : "+v"(a[0].ahi), "+v"(a[1].ahi), "+v"(a[2].ahi), "+v"(a[3].ahi), "=&s"(sb), "=&s"(sc), "=&s"(sd)
        : "v"(mac_hi32(x[0])), "v"(mac_hi32(x[1])), "v"(mac_hi32(x[2])), "v"(mac_hi32(x[3])), "v"(mac_hi32(k[0])), "v"(mac_hi32(k[1])), "v"(mac_hi32(k[2])),
          "v"(mac_hi32(k[3]))
        : "vcc");
}
#endif

// acc[b][k][i][n] = sum_j opnd(b,i,j)[n] * key[j][k][limb(i)][n] mod p_i ; operands canonical (< 2^61), dl < 64
// ckks_target != nullptr: operand (i == j) comes from the NTT-form input itself (evaluator.cpp:2424-2427)
// A thread owns V consecutive coefficients of NB consecutive batch items (NB * V = 4 accumulators per key component):
// every key word it loads is used NB times.
#ifndef KS_MAC_NB
#define KS_MAC_NB 4
#endif
template <int NB, int V> __global__ __launch_bounds__(EW_THREADS) void ks_mac_kernel(const u64 *D, const u64 *key, const u64 *ckks_target, u64 t_bstride, u64 *acc, KsArgs a) {
    static_assert(NB * V == 4, "mac4 works on four accumulators");
    // block order: (output prime i, coefficient window, batch group) with the batch group FASTEST, so that the workgroups
    // in flight at any moment share a narrow window of the key (it stays in L2 while every group passes over it)
    const u64 N = u64(1) << a.logn, rl = a.dl + 1;
    const int logv = V == 2 ? 1 : 0;
    const u64 groups = (a.batch + NB - 1) / NB;
    const u64 g = blockIdx.x % groups, wi = blockIdx.x / groups;        // wi = i * windows + window
    const u64 idx = wi * EW_THREADS + threadIdx.x;                       // over (dl+1) * (N / V)
    if (idx >= rl << (a.logn - logv)) return;
    const u64 n = (idx & ((N >> logv) - 1)) << logv, i = idx >> (a.logn - logv), b0 = g * NB;
    const Mod m = mod_of(a.primes[a.key_id[i]]);
    const u64 kl = a.key_limb[i];
    MacAcc s0[4], s1[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { mac_zero(s0[t]); mac_zero(s1[t]); }
    for (u64 j = 0; j < a.dl; j++) {
        const u64 *kp = key + ((j * 2) * a.K + kl) * N + n;
        u64 k0[4], k1[4], x[4];
        if (V == 2) {
            const ulonglong2 t0 = *reinterpret_cast<const ulonglong2 *>(kp), t1 = *reinterpret_cast<const ulonglong2 *>(kp + a.K * N);
#pragma unroll
            for (int b = 0; b < NB; b++) { k0[b * V] = t0.x; k0[b * V + V - 1] = t0.y; k1[b * V] = t1.x; k1[b * V + V - 1] = t1.y; }
        } else {
            const u64 t0 = kp[0], t1 = kp[a.K * N];
#pragma unroll
            for (int b = 0; b < NB; b++) { k0[b] = t0; k1[b] = t1; }
        }
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const u64 bb = b0 + b < a.batch ? b0 + b : a.batch - 1; // ragged last group: recompute the last item, store nothing
            const u64 *xp = (ckks_target && i == j) ? ckks_target + bb * t_bstride + j * N + n : D + ((bb * rl + i) * a.dl + j) * N + n;
            if (V == 2) {
                const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(xp);
                x[b * V] = t.x; x[b * V + V - 1] = t.y;
            } else {
                x[b] = xp[0];
            }
        }
        mac4(s0, x, k0);
        mac4(s1, x, k1);
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
        if (b0 + b >= a.batch) break;
        u64 r0[V], r1[V];
#pragma unroll
        for (int v = 0; v < V; v++) {
            const U128 v0 = mac_value(s0[b * V + v]), v1 = mac_value(s1[b * V + v]);
            r0[v] = barrett128(v0.lo, v0.hi, m);
            r1[v] = barrett128(v1.lo, v1.hi, m);
        }
        u64 *o0 = acc + (((b0 + b) * 2 + 0) * rl + i) * N + n, *o1 = acc + (((b0 + b) * 2 + 1) * rl + i) * N + n;
        if (V == 2) {
            ulonglong2 t0, t1;
            t0.x = r0[0]; t0.y = r0[V - 1]; t1.x = r1[0]; t1.y = r1[V - 1];
            *reinterpret_cast<ulonglong2 *>(o0) = t0;
            *reinterpret_cast<ulonglong2 *>(o1) = t1;
        } else {
            o0[0] = r0[0]; o1[0] = r1[0];
        }
    }
}
// BFV (kind 0) / BGV (kind 2) mod-down, everything in coefficient form (evaluator.cpp:2528-2648):
//   ct[b][k][j][n] += (acc_j - [t']_{q_j} + [half]_{q_j}) * qk^-1 mod q_j, t' = (acc_last + half) mod qk       (BFV)
//   ct[b][k][j][n] += (acc_j - [acc_last]_{q_j} - [k_t]_{q_j} * qk) * qk^-1,  k_t = -acc_last * qk^-1 mod t     (BGV)
template <int KIND> __global__ __launch_bounds__(EW_THREADS) void ks_moddown_kernel(const u64 *acc, u64 *ct, u64 ct_bstride, KsArgs a) {
    u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x; // over batch * 2 * dl * N
    const u64 N = u64(1) << a.logn, rl = a.dl + 1;
    if (idx >= a.batch * 2 * a.dl * N) return;
    u64 n = idx & (N - 1), r = idx >> a.logn, j = r % a.dl, bk = r / a.dl, k = bk & 1, b = bk >> 1;
    const PrimeDesc &pj = a.primes[a.key_id[j]];
    const PrimeDesc &pk = a.primes[a.key_id[a.dl]];
    const Mod m = mod_of(pj);
    const u64 aj = acc[((b * 2 + k) * rl + j) * N + n];
    const u64 al = acc[((b * 2 + k) * rl + a.dl) * N + n];
    u64 v;
    if (KIND == 0) {
        u64 tl = addmod(al, a.half, pk.p);
        v = aj + (m.p - barrett64(tl, m)) + barrett64(a.half, m);
    } else {
        const Mod tm{a.t_p, a.t_cr0, a.t_cr1};
        u64 kt = negmod(barrett64(al, tm), tm.p);
        if (a.inv_qk_mod_t != 1) kt = mulmod(kt, a.inv_qk_mod_t, tm);
        u64 delta = mulmod(barrett64(kt, m), barrett64(pk.p, m), m);
        v = aj + 2 * m.p - (delta + barrett64(al, m));
    }
    v = mul_shoup(v, a.inv_qk[j], m.p);
    u64 *c = ct + b * ct_bstride + (k * a.dl + j) * N + n;
    *c = addmod(*c, v, m.p);
}
// BGV mod-down inside the inverse transform (Ntt2ModDown, ntt2.hip): what the special limb contributes to EVERY data limb, once per
// coefficient instead of once per (limb, coefficient):  S = al + k_t qk  as a 128-bit integer, k_t = -al qk^-1 mod t; the epilogue of
// limb j subtracts [S]_{q_j} = [al]_{q_j} + [k_t]_{q_j} qk  (mod q_j).  share[o][n] = (lo, hi).
__global__ __launch_bounds__(EW_THREADS) void ks_bgv_share_kernel(const u64 *acc, u64 *share, KsArgs a) {
    const u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x; // over batch * 2 * N
    const u64 N = u64(1) << a.logn, rl = a.dl + 1;
    if (idx >= a.batch * 2 * N) return;
    const u64 n = idx & (N - 1), o = idx >> a.logn;
    const u64 qk = a.primes[a.key_id[a.dl]].p;
    const u64 al = acc[(o * rl + a.dl) * N + n];
    const Mod tm{a.t_p, a.t_cr0, a.t_cr1};
    u64 kt = negmod(barrett64(al, tm), tm.p);
    if (a.inv_qk_mod_t != 1) kt = mulmod(kt, a.inv_qk_mod_t, tm);
    const u64 lo = kt * qk, hi = mulhi64(kt, qk);
    ulonglong2 v;
    v.x = lo + al;
    v.y = hi + (v.x < lo ? 1 : 0);
    reinterpret_cast<ulonglong2 *>(share)[idx] = v;
}
// CKKS mod-down, NTT form: step F builds the correction polynomial from the coefficient-form special limb
//   corr[b][k][j][n] = [t']_{q_j} + (q_j - [half]_{q_j}),  t' = (last + half) mod qk
// (NTT over corr), step G: ct += (acc_j + q_j - corr_j) * qk^-1 mod q_j
__global__ __launch_bounds__(EW_THREADS) void ks_ckks_corr_kernel(const u64 *last /*[b*2][N]*/, u64 *corr, KsArgs a) {
    u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x; // batch * 2 * dl * N
    const u64 N = u64(1) << a.logn;
    if (idx >= a.batch * 2 * a.dl * N) return;
    u64 n = idx & (N - 1), r = idx >> a.logn, j = r % a.dl, bk = r / a.dl;
    const Mod m = mod_of(a.primes[a.key_id[j]]);
    const u64 qk = a.primes[a.key_id[a.dl]].p;
    u64 tl = addmod(last[bk * N + n], a.half, qk);
    corr[idx] = barrett64(tl, m) + (m.p - barrett64(a.half, m));
}
__global__ __launch_bounds__(EW_THREADS) void ks_ckks_combine_kernel(const u64 *acc, const u64 *corr, u64 *ct, u64 ct_bstride, KsArgs a) {
    u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    const u64 N = u64(1) << a.logn, rl = a.dl + 1;
    if (idx >= a.batch * 2 * a.dl * N) return;
    u64 n = idx & (N - 1), r = idx >> a.logn, j = r % a.dl, bk = r / a.dl, k = bk & 1, b = bk >> 1;
    const u64 p = a.primes[a.key_id[j]].p;
    const u64 aj = acc[((b * 2 + k) * rl + j) * N + n];
    u64 v = mul_shoup(aj + p - corr[idx], a.inv_qk[j], p);
    u64 *c = ct + b * ct_bstride + (k * a.dl + j) * N + n;
    *c = addmod(*c, v, p);
}

void launch_ks_expand(const u64 *target, u64 t_bstride, u64 *D, const KsArgs &a, hipStream_t s) {
    u64 total = a.batch * a.dl << a.logn;
    TROY_LAUNCH(ks_expand_kernel, dim3(ceil_div(total, EW_THREADS)), dim3(EW_THREADS), 0, s, target, t_bstride, D, a);
    launch_check("ks_expand_kernel");
}
void launch_ks_mac(const u64 *D, const u64 *key, const u64 *ckks_target, u64 t_bstride, u64 *acc, const KsArgs &a, hipStream_t s) {
    if (a.dl >= 64) throw Error(ST_LOGIC_ERROR, "ks_mac: more than 63 digits");
    constexpr int NB = KS_MAC_NB, V = 4 / KS_MAC_NB;
    const u64 groups = (a.batch + NB - 1) / NB, per_group = (u64)(a.dl + 1) << (a.logn - (V == 2 ? 1 : 0));
    TROY_LAUNCH(HIP_KERNEL_NAME(ks_mac_kernel<NB, V>), dim3(ceil_div(per_group, EW_THREADS) * groups), dim3(EW_THREADS), 0, s, D, key, ckks_target, t_bstride, acc, a);
    launch_check("ks_mac_kernel");
}
void launch_ks_moddown(int kind, const u64 *acc, u64 *ct, u64 ct_bstride, const KsArgs &a, hipStream_t s) {
    u64 total = a.batch * 2 * a.dl << a.logn;
    dim3 grid(ceil_div(total, EW_THREADS)), blk(EW_THREADS);
    if (kind == 0) TROY_LAUNCH(HIP_KERNEL_NAME(ks_moddown_kernel<0>), grid, blk, 0, s, acc, ct, ct_bstride, a);
    else TROY_LAUNCH(HIP_KERNEL_NAME(ks_moddown_kernel<2>), grid, blk, 0, s, acc, ct, ct_bstride, a);
    launch_check("ks_moddown_kernel");
}
void launch_ks_bgv_share(const u64 *acc, u64 *share, const KsArgs &a, hipStream_t s) {
    const u64 total = a.batch * 2 << a.logn;
    TROY_LAUNCH(ks_bgv_share_kernel, dim3(ceil_div(total, EW_THREADS)), dim3(EW_THREADS), 0, s, acc, share, a);
    launch_check("ks_bgv_share_kernel");
}
void launch_ks_ckks_corr(const u64 *last, u64 *corr, const KsArgs &a, hipStream_t s) {
    u64 total = a.batch * 2 * a.dl << a.logn;
    TROY_LAUNCH(ks_ckks_corr_kernel, dim3(ceil_div(total, EW_THREADS)), dim3(EW_THREADS), 0, s, last, corr, a);
    launch_check("ks_ckks_corr_kernel");
}
void launch_ks_ckks_combine(const u64 *acc, const u64 *corr, u64 *ct, u64 ct_bstride, const KsArgs &a, hipStream_t s) {
    u64 total = a.batch * 2 * a.dl << a.logn;
    TROY_LAUNCH(ks_ckks_combine_kernel, dim3(ceil_div(total, EW_THREADS)), dim3(EW_THREADS), 0, s, acc, corr, ct, ct_bstride, a);
    launch_check("ks_ckks_combine_kernel");
}

// ---------------------------------------------------------------- decryption
// dotProductCtSkArray (decryptor_cuda.cu:284-330): every product reduced, then added modulo p
__global__ __launch_bounds__(EW_THREADS) void dot_sk_kernel(const u64 *x, const u64 *spow, u64 *acc, DecryptArgs a) {
    const u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    const u64 per = a.limbs << a.logn;
    if (idx >= a.batch * per) return;
    const u64 b = idx / per, r = idx % per;
    const Mod m = mod_of(a.primes[a.map.id[r >> a.logn]]);
    u64 s = 0;
    for (u64 i = 0; i + 1 < a.size; i++) s = addmod(s, mulmod(x[(b * (a.size - 1) + i) * per + r], spow[i * per + r], m), m.p);
    acc[idx] = s;
}
__global__ __launch_bounds__(EW_THREADS) void add_c0_kernel(const u64 *ct, u64 *acc, DecryptArgs a) {
    const u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    const u64 per = a.limbs << a.logn;
    if (idx >= a.batch * per) return;
    const u64 b = idx / per, r = idx % per;
    acc[idx] = addmod(acc[idx], ct[b * a.ct_bstride + r], a.primes[a.map.id[r >> a.logn]].p);
}
// BFV: decryptScaleAndRound (rns_cuda.cu:510-577, CPU rns.cpp:1039-1095): round(t/q * x) through the base {t, gamma}
__global__ __launch_bounds__(EW_THREADS) void decrypt_bfv_kernel(const u64 *acc, u64 *out, DecryptArgs a) {
    const u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    const u64 N = u64(1) << a.logn;
    if (idx >= a.batch * N) return;
    const u64 b = idx >> a.logn, n = idx & (N - 1);
    const Mod tm{a.t_p, a.t_cr0, a.t_cr1}, gm{a.g_p, a.g_cr0, a.g_cr1};
    U128 st{0, 0}, sg{0, 0};
    for (u64 l = 0; l < a.limbs; l++) {
        const u64 p = a.primes[a.map.id[l]].p;
        const u64 v = mul_shoup(acc[(b * a.limbs + l) * N + n], a.pre[l], p);
        mac128(st, v, a.mat_t[l]);
        mac128(sg, v, a.mat_g[l]);
        if ((l & 7) == 7) { st.lo = barrett128(st.lo, st.hi, tm); st.hi = 0; sg.lo = barrett128(sg.lo, sg.hi, gm); sg.hi = 0; }
    }
    const u64 vt = mulmod(barrett128(st.lo, st.hi, tm), a.neg_inv_q_mod_t, tm);
    const u64 vg = mulmod(barrett128(sg.lo, sg.hi, gm), a.neg_inv_q_mod_gamma, gm);
    u64 d;
    if (vg > (gm.p >> 1)) d = addmod(vt, barrett64(gm.p - vg, tm), tm.p);
    else d = submod(vt, barrett64(vg, tm), tm.p);
    out[b * a.out_bstride + n] = d ? mulmod(d, a.inv_gamma_mod_t, tm) : 0;
}
// BGV: decryptModt = exactConvertArray to t (rns_cuda.cu:147-180: the rounding term is a double-precision sum taken in limb
// order, exactly as the CPU path rns.cpp:462-548 does) times correction_factor^-1
__global__ __launch_bounds__(EW_THREADS) void decrypt_bgv_kernel(const u64 *acc, u64 *out, DecryptArgs a) {
    const u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    const u64 N = u64(1) << a.logn;
    if (idx >= a.batch * N) return;
    const u64 b = idx >> a.logn, n = idx & (N - 1);
    const Mod tm{a.t_p, a.t_cr0, a.t_cr1};
    double agg = 0.0;
    U128 sum{0, 0};
    for (u64 l = 0; l < a.limbs; l++) {
        const u64 p = a.primes[a.map.id[l]].p;
        const u64 v = mul_shoup(acc[(b * a.limbs + l) * N + n], a.pre[l], p);
        agg += (double)v / (double)p;
        mac128(sum, v, a.mat_t[l]);
        if ((l & 7) == 7) { sum.lo = barrett128(sum.lo, sum.hi, tm); sum.hi = 0; }
    }
    agg += 0.5;
    const u64 rounded = (u64)agg;
    u64 d = submod(barrett128(sum.lo, sum.hi, tm), mulmod(barrett64(rounded, tm), a.q_mod_t, tm), tm.p);
    if (a.inv_cf != 1) d = mulmod(d, a.inv_cf, tm);
    out[b * a.out_bstride + n] = d;
}
void launch_dot_sk(const u64 *x, const u64 *spow, u64 *acc, const DecryptArgs &a, hipStream_t s) {
    const u64 total = a.batch * a.limbs << a.logn;
    if (!total) return;
    TROY_LAUNCH(dot_sk_kernel, dim3(ceil_div(total, EW_THREADS)), dim3(EW_THREADS), 0, s, x, spow, acc, a);
    launch_check("dot_sk_kernel");
}
void launch_add_c0(const u64 *ct, u64 *acc, const DecryptArgs &a, hipStream_t s) {
    const u64 total = a.batch * a.limbs << a.logn;
    if (!total) return;
    TROY_LAUNCH(add_c0_kernel, dim3(ceil_div(total, EW_THREADS)), dim3(EW_THREADS), 0, s, ct, acc, a);
    launch_check("add_c0_kernel");
}
void launch_decrypt_final(int scheme, const u64 *acc, u64 *out, const DecryptArgs &a, hipStream_t s) {
    const u64 total = a.batch << a.logn;
    if (!total) return;
    dim3 grid(ceil_div(total, EW_THREADS)), blk(EW_THREADS);
    if (scheme == 1 /* BFV */) TROY_LAUNCH(decrypt_bfv_kernel, grid, blk, 0, s, acc, out, a);
    else TROY_LAUNCH(decrypt_bgv_kernel, grid, blk, 0, s, acc, out, a);
    launch_check("decrypt_kernel");
}

// ---------------------------------------------------------------- strided copy / zero helpers
// dst[b*dst_bstride + i] = src[b*src_bstride + i], i < count
__global__ __launch_bounds__(EW_THREADS) void copy_strided_kernel(const u64 *src, u64 src_bstride, u64 *dst, u64 dst_bstride, u64 count, u64 batch) {
    u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    if (idx >= batch * count) return;
    u64 b = idx / count, i = idx % count;
    dst[b * dst_bstride + i] = src[b * src_bstride + i];
}
void launch_copy_strided(const u64 *src, u64 src_bstride, u64 *dst, u64 dst_bstride, u64 count, u64 batch, hipStream_t s) {
    if (!batch || !count) return;
    TROY_LAUNCH(copy_strided_kernel, dim3(ceil_div(batch * count, EW_THREADS)), dim3(EW_THREADS), 0, s, src, src_bstride, dst, dst_bstride, count, batch);
    launch_check("copy_strided_kernel");
}
__global__ __launch_bounds__(EW_THREADS) void zero_strided_kernel(u64 *dst, u64 dst_bstride, u64 count, u64 batch) {
    u64 idx = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    if (idx >= batch * count) return;
    dst[(idx / count) * dst_bstride + idx % count] = 0;
}
void launch_zero_strided(u64 *dst, u64 dst_bstride, u64 count, u64 batch, hipStream_t s) {
    if (!batch || !count) return;
    TROY_LAUNCH(zero_strided_kernel, dim3(ceil_div(batch * count, EW_THREADS)), dim3(EW_THREADS), 0, s, dst, dst_bstride, count, batch);
    launch_check("zero_strided_kernel");
}

// ---------------------------------------------------------------- synthetic data (bench / tests)
// value(row, n) = splitmix64 stream seeded by (seed ^ row * 0xD1B54A32D192ED03), output n, reduced mod p_row
__device__ __forceinline__ u64 splitmix64_at(u64 seed, u64 n) {
    u64 z = seed + (n + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(EW_THREADS) void fill_uniform_kernel(u64 *out, const PrimeDesc *primes, LimbMap map, int logn, u64 seed, u64 row0, u64 total) {
    u64 i = (u64)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i >= total) return;
    u64 row = i >> logn, n = i & ((u64(1) << logn) - 1);
    const Mod m = mod_of(prime_of(primes, map, row));
    u64 z = splitmix64_at(seed ^ ((row0 + row) * 0xD1B54A32D192ED03ULL), n);
    out[i] = barrett64(z, m);
}
void launch_fill_uniform(u64 *out, const PrimeDesc *primes, const LimbMap &map, int logn, u64 seed, u64 row0, u64 rows, hipStream_t s) {
    u64 total = rows << logn;
    if (!total) return;
    TROY_LAUNCH(fill_uniform_kernel, dim3(ceil_div(total, EW_THREADS)), dim3(EW_THREADS), 0, s, out, primes, map, logn, seed, row0, total);
    launch_check("fill_uniform_kernel");
}

} // namespace troyhip
