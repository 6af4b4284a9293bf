// behz2.hip -- BEHZ base conversion on the matrix cores, second form: rows reduced modulo the OUTPUT prime on the host.
//
// Same contracts as behz_extend_kernel / behz_floor_sk_kernel (behz.hip; reference: src/utils/rns_cuda.cu:96-145, 365-508,
// CPU twins src/utils/rns.cpp:415-459, 879-1037) and the same canonical residues.
//
// A conversion output is  out_o = (sum_l y_l M[o][l]) mod p_o  with y_l < 2^61 written in eight balanced base-256 digits
// d_{l,k}.  Instead of multiplying digit strings (16 byte-shifts per output, a 126-bit sum, a 128-bit reduction: behz.hip's
// first MFMA form), the host reduces every column of the digit expansion:
//     out_o = ( sum_{l,k} d_{l,k} * W[o][l][k] ) mod p_o,      W[o][l][k] = M[o][l] 2^(8k) mod p_o  (< 2^61, eight digits)
// so row (o, s) of the int8 matrix is digit s of W -- EIGHT shifts per output.  That halves the v_mfma_i32_32x32x32_i8 count
// (a 32-row block is 4 outputs, not 2) and bounds the recombined sum by 2^(8 nd + 13.4), nd <= 8 the digit count of a residue of
// p_o: with a bias (a multiple of p_o) it is a non-negative V < 2^(8 nd + 15.1) and ONE 32-bit quotient estimate reduces it:
//     q^ = mulhi32(V >> sh, floor(2^(sh+32) / p)),  sh = bitlen(p) - 2,   r = (V - q^ p) mod 2^64 in [0, 2p),   one conditional subtraction
// (V >> sh < 2^25.1; error of q^ below 2^sh/p + 2^-6.9 < 0.51, for every prime >= 2^33).  Recombine + reduce is ~20 VALU instructions
// per output instead of ~85; the kernels are VALU-issue bound (profiles/r01_behz_probe.txt), so that is where the time goes.
// The per-coefficient correction terms are extra INPUT columns of the product: the centred m_tilde residue r against
// q m_tilde^-1 (extension), the centred alpha against -(B mod q_l) (Shenoy-Kumaresan), at limb position L (nB) of the digit
// matrix -- patched into the B fragment in registers, whatever L mod 4 is.  B^-1 mod m_sk is folded into the m_sk rows of both
// stages, so alpha is one modular subtraction.
#include "kernels.h"
#include <cstdlib>

namespace troyhip {

#define B2_THREADS 256
#define B2_TILE 64
#ifndef B2_TPW
#define B2_TPW 8 // tiles of 64 coefficients per workgroup
#endif
// probe hooks for throw-away builds (tools/behz_probe.sh): bit0 no HBM loads, bit1 no MFMA, bit2 no recombine/reduce, bit4 no stores
#ifndef B2_EXP
#define B2_EXP 0
#endif

#ifdef TROYHIP_CPU_EMUL
#define B2_UNIFORM(x) (x)
typedef const u64 *b2_cu64;
#else
#define B2_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
typedef const __attribute__((address_space(4))) u64 *b2_cu64; // wave-uniform addresses: scalar loads
#endif

__device__ __forceinline__ long long b2_mad(int a, int b, long long c) { // c + sext(a) * b in one instruction
#ifdef TROYHIP_CPU_EMUL
    return (long long)a * b + c;
#else
    long long d;
    u64 sink;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(sink) : "v"(a), "s"(b), "v"(c));
    return d;
#endif
}
__device__ __forceinline__ MfmaFrag b2_frag(const void *base, size_t idx) { return reinterpret_cast<const MfmaFrag *>(base)[idx]; }
__device__ __forceinline__ void b2_zero(MfmaAcc &a) {
#pragma unroll
    for (int r = 0; r < 16; r++) a.v[r] = 0;
}
#define B2_MFMA(fa, fb, fc)                                                                                               \
    do {                                                                                                                 \
        if (B2_EXP & 2) (fc).v[0] += (fb).bytes[0] + (fa).bytes[0];                                                      \
        else TROY_MFMA_I8(fa, fb, fc);                                                                                   \
    } while (0)
#define B2_LOAD(expr, fake) ((B2_EXP & 1) ? (u64)(fake) : (expr))

// V = lo + top 2^64 (top < 2^16): the biased sum of one output
struct B2Sum { u64 lo; u32 top; };
// sum_s C_s 2^(8s) over accumulator registers BASE .. BASE + 7 (|C_s| < 2^21.4), plus `addend` (>= 2^46, < 2^63) into the low
// half and `bias1` (at least the magnitude of the high half) into the high half: both halves stay positive, so the words combine
// without sign handling
template <int BASE> __device__ __forceinline__ B2Sum b2_recombine(const MfmaAcc &a, u64 addend, u64 bias1) {
    // two's-complement arithmetic on the 32-bit words (a left shift of a negative int is undefined in C++17: shift the unsigned image)
    auto sh8 = [](int lo, int hi) { return (int)((u32)lo + ((u32)hi << 8)); };
    const int p01 = sh8(a.v[BASE], a.v[BASE + 1]), p23 = sh8(a.v[BASE + 2], a.v[BASE + 3]);
    const int p45 = sh8(a.v[BASE + 4], a.v[BASE + 5]), p67 = sh8(a.v[BASE + 6], a.v[BASE + 7]);
    const u64 w0 = (u64)b2_mad(p01, 1, b2_mad(p23, 1 << 16, (long long)addend));
    const u64 w1 = (u64)b2_mad(p45, 1, b2_mad(p67, 1 << 16, (long long)bias1));
    const u32 w0h = (u32)(w0 >> 32), v1 = w0h + (u32)w1;
    const u32 v2 = (u32)(w1 >> 32) + (v1 < w0h);
    return B2Sum{(u64)(u32)w0 | ((u64)v1 << 32), v2};
}
__device__ __forceinline__ u64 b2_reduce(const B2Sum v, const BehzK2 &k) {
    const u32 vt = (u32)((((u64)v.top << 32) | (v.lo >> 32)) >> k.sh32); // bits sh .. sh + 31 of V (sh >= 32; everything above is zero)
    const u32 qh = (u32)(((u64)vt * k.mu) >> 32);
    const u64 r = v.lo + (u64)qh * k.negp; // V - qh p modulo 2^64: the true value is in [0, 2p)
    return r >= k.p ? r - k.p : r;
}
// output primes below 2^33 (q side only): the same sum through the two-word reduction of behz.hip
struct B2Slow { u64 p, cr1, two_p, r64_op, r64_quo; };
__device__ __forceinline__ u64 b2_reduce_slow(const B2Sum v, const B2Slow &k) {
    const u64 a = mul_lazy((u64)v.top, k.r64_op, k.r64_quo, k.p);
    const u64 b = v.lo - mulhi64(v.lo, k.cr1) * k.p;
    u64 s = a + b;
    s = s >= k.two_p ? s - k.two_p : s;
    return s >= k.p ? s - k.p : s;
}
template <int BASE> __device__ __forceinline__ u64 b2_finish(const MfmaAcc &acc, u64 addend, const BehzK2 &k) {
    if (B2_EXP & 4) return (u64)(u32)acc.v[BASE] ^ addend ^ ((u64)(u32)acc.v[BASE + 7] << 32);
    return b2_reduce(b2_recombine<BASE>(acc, addend, k.bias1), k);
}

// words of a fragment: limb (4 kb + 2 half + pos) of this lane's coefficient
__device__ __forceinline__ u64 b2_digits(u64 v) {
    const u64 c80 = 0x8080808080808080ull;
    return (v + c80) ^ c80;
}
// the extra input (limb index `xl`, one past the real limbs) sits in the last k-block, in the lanes of half (xl % 4) / 2, word xl % 2
struct B2Patch { bool w0, w1; };
__device__ __forceinline__ B2Patch b2_patch_where(int xl, unsigned half) {
    const bool mine = half == (unsigned)((xl & 3) >> 1);
    return B2Patch{mine && !(xl & 1), mine && (xl & 1)};
}
__device__ __forceinline__ void b2_patch(MfmaFrag &f, const B2Patch w, u64 digits) {
    MfmaFrag g = f;
    frag_set_word(g, 0, digits);
    if (w.w0) f = g;
    g = f;
    frag_set_word(g, 1, digits);
    if (w.w1) f = g;
}

// ---------------------------------------------------------------- extension q -> Bsk (fastbconvmTilde + smMrq)
// in [polys][L][N] -> out [polys][nBsk][N].  A 256-thread workgroup walks `tiles_per_wg` tiles of 64 coefficients of one polynomial;
// wave w converts limbs w, w+4, .. to digits (LDS, double-buffered: one barrier per tile), evaluates the m_tilde row for its own
// lanes and owns row-block w = outputs 4w .. 4w+3: a lane finishes outputs 4w + half (accumulators 0-7) and 4w + 2 + half (8-15).
// polynomials from `split` on are read from in2 (the second operand of a product: both go through one launch when it is small)
template <int KB> __global__ __launch_bounds__(B2_THREADS) void behz2_extend_kernel(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, BehzDev c,
                                                                                   u64 N, unsigned tiles_per_wg, const u64 *in2, unsigned split) {
    __shared__ __attribute__((aligned(16))) u64 ydig[2][2 * KB * B2_TILE * 2]; // [buffer][limb pair][coefficient] 16-byte units
    const unsigned lane = threadIdx.x & 63, half = lane >> 5, cl = lane & 31;
    const int w = B2_UNIFORM((int)(threadIdx.x >> 6));
    const u64 poly = blockIdx.y;
    const BufRsrc rin = make_rsrc(poly < split ? in + poly * in_pstride : in2 + (poly - split) * in_pstride, (u32)((u64)c.L * N * 8));
    const BufRsrc rout = make_rsrc(out + poly * out_pstride, (u32)((u64)c.nBsk * N * 8));
    const u32 n32 = (u32)N;
    const bool owner = w < ((c.nBsk + 3) >> 2);
    MfmaFrag af[KB], am[KB];
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
        af[kb] = b2_frag(c.x_frag, ((size_t)(owner ? w : 0) * KB + kb) * 64 + lane);
        am[kb] = b2_frag(c.x_mt_frag, (size_t)kb * 64 + lane);
    }
    BehzK2 k2[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int o = 4 * w + 2 * j + (int)half;
        k2[j] = c.x_k[o < c.nBsk ? o : c.nBsk - 1];
    }
    const B2Patch where = b2_patch_where(c.L, half);
    u64 qp[KB];
    Shoup qpre[KB];
#pragma unroll
    for (int i = 0; i < KB; i++) {
        const int l = w + 4 * i;
        const int lc = l < c.L ? l : 0;
        const unsigned id = B2_UNIFORM((unsigned)c.q_id[lc]);
        qp[i] = ((b2_cu64)&primes[id])[0];
        qpre[i] = Shoup{((b2_cu64)(c.ext_pre + lc))[0], ((b2_cu64)(c.ext_pre + lc))[1]};
    }
    u64 xr[KB];
    auto fetch = [&](unsigned t) { // a tile beyond N, a padding limb: offset out of range, the load returns 0
        const u32 n = (blockIdx.x * tiles_per_wg + t) * B2_TILE + lane;
#pragma unroll
        for (int i = 0; i < KB; i++) {
            const u32 l = (u32)(w + 4 * i);
            xr[i] = B2_LOAD(buf_load_u64(rin, (l < (u32)c.L && n < n32) ? (l * n32 + n) * 8u : TROY_BUF_OOB), (lane + n) * 0x9E3779B97F4Aull + l);
        }
    };
    fetch(0);
    // stand-ins for "the previous tile's stores" so that the loop's waits are counted, not full drains (see behz.hip)
#pragma unroll
    for (int i = 0; i < 4; i++) buf_store_u64(rout, TROY_BUF_OOB + 8 * i, 0);
    for (unsigned t = 0; t < tiles_per_wg; t++) {
        const u32 n0 = (blockIdx.x * tiles_per_wg + t) * B2_TILE;
        if (n0 >= n32) break;
        u64 *yd = ydig[t & 1];
        // y_l = x_l m_tilde (q/q_l)^-1 mod q_l as balanced digits; limbs L .. 4 KB - 1 (the extra input's slot and padding) are zero
#pragma unroll
        for (int i = 0; i < KB; i++) {
            const int l = w + 4 * i;
            yd[(((l >> 1) * B2_TILE) + lane) * 2 + (l & 1)] = b2_digits(mul_shoup(xr[i], qpre[i].op, qpre[i].quo, qp[i]));
        }
        fetch(t + 1 < tiles_per_wg ? t + 1 : 0x7FFFFFu); // past the last tile: out of range
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < B2_TILE / 32; sub++) {
            const unsigned cc = sub * 32 + cl;
            MfmaFrag bf[KB];
#pragma unroll
            for (int kb = 0; kb < KB; kb++) bf[kb] = b2_frag(yd, (size_t)(2 * kb + half) * B2_TILE + cc);
            // r = -(sum_l y_l (q/q_l)) q^-1 mod 2^32 from the four shift rows of the m_tilde block (both halves hold them)
            MfmaAcc acc;
            b2_zero(acc);
#pragma unroll
            for (int kb = 0; kb < KB; kb++) B2_MFMA(am[kb], bf[kb], acc);
            const u32 rsum = (u32)acc.v[0] + ((u32)acc.v[1] << 8) + ((u32)acc.v[2] << 16) + ((u32)acc.v[3] << 24);
            const u64 r_mt = ((u64)rsum * c.neg_inv_q_mod_mt) & 0xFFFFFFFFull;
            // centred representative of r (rns.cpp:966-975) as one more input: out = (sum + r q) m_tilde^-1
            b2_patch(bf[KB - 1], where, b2_digits((u64)((long long)r_mt - (long long)((r_mt >> 31) << 32))));
            u64 r[2] = {0, 0};
            if (owner) {
                b2_zero(acc);
#pragma unroll
                for (int kb = 0; kb < KB; kb++) B2_MFMA(af[kb], bf[kb], acc);
                r[0] = b2_finish<0>(acc, k2[0].biaslo, k2[0]);
                r[1] = b2_finish<8>(acc, k2[1].biaslo, k2[1]);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) { // the same number of stores on every path keeps the outstanding-store count a constant
                const u32 o = 4 * (u32)w + 2 * j + half;
                const bool live = (B2_EXP & 16) ? r[j] == ~0ull : (owner && o < (u32)c.nBsk && n0 + cc < n32);
                buf_store_u64(rout, live ? (o * n32 + n0 + cc) * 8u : TROY_BUF_OOB, r[j]);
            }
        }
    }
}

// ---------------------------------------------------------------- floor + Shenoy-Kumaresan (fastFloor + fastbconvSk)
// y [polys][L][N], dbt [polys][nBsk][N] -> out [polys][L][N], inputs PRE-SCALED and canonical: y_l = t dq_l (q/q_l)^-1 mod q_l and
// dbt_o = db_o T_o mod Bsk_o, which the inverse transforms in front deliver for free (BehzDev::floor_desc).  Two chained products per tile:
//   (1) rows of f1_frag x digits(y), + dbt_o  ->  u_b (digits, LDS) for o < nB, and z' = z_sk B^-1 mod m_sk
//   (2) the m_sk row x digits(u) -> alpha = conv' - z' (every wave, for its own lanes), then rows of f2_frag x [digits(u), alpha] -> out_l
// All A fragments stay in registers (one row-block per wave and stage); two barriers per tile.
template <int KB1, int KB2, bool FAST2> __global__ __launch_bounds__(B2_THREADS) void behz2_floor_sk_kernel(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride,
                                                                                                           u64 *out, u64 out_pstride, const PrimeDesc *primes, BehzDev c,
                                                                                                           u64 N, unsigned tiles_per_wg) {
    __shared__ __attribute__((aligned(16))) u64 ydig[2 * KB1 * B2_TILE * 2];
    __shared__ __attribute__((aligned(16))) u64 udig[2 * KB2 * B2_TILE * 2];
    __shared__ u64 zsk[B2_TILE];
    const unsigned lane = threadIdx.x & 63, half = lane >> 5, cl = lane & 31;
    const int w = B2_UNIFORM((int)(threadIdx.x >> 6));
    const u64 poly = blockIdx.y;
    const BufRsrc rq = make_rsrc(dq + poly * dq_pstride, (u32)((u64)c.L * N * 8));
    const BufRsrc rb = make_rsrc(db + poly * db_pstride, (u32)((u64)c.nBsk * N * 8));
    const BufRsrc rout = make_rsrc(out + poly * out_pstride, (u32)((u64)c.L * N * 8));
    const u32 n32 = (u32)N;
    const bool owner1 = w < ((c.nBsk + 3) >> 2), owner2 = w < ((c.L + 3) >> 2);
    for (unsigned i = threadIdx.x; i < 2 * KB2 * B2_TILE * 2; i += B2_THREADS) udig[i] = 0; // limbs nB .. stay zero
    MfmaFrag af1[KB1], af2[KB2], amsk[KB2];
#pragma unroll
    for (int kb = 0; kb < KB1; kb++) af1[kb] = b2_frag(c.f1_frag, ((size_t)(owner1 ? w : 0) * KB1 + kb) * 64 + lane);
#pragma unroll
    for (int kb = 0; kb < KB2; kb++) {
        af2[kb] = b2_frag(c.f2_frag, ((size_t)(owner2 ? w : 0) * KB2 + kb) * 64 + lane);
        amsk[kb] = b2_frag(c.f2_msk_frag, (size_t)kb * 64 + lane);
    }
    BehzK2 k1[2], k2[2];
    B2Slow s2[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int o = 4 * w + 2 * j + (int)half;
        const int oc = o < c.nBsk ? o : c.nBsk - 1, lc = o < c.L ? o : c.L - 1;
        k1[j] = c.f1_k[oc];
        k2[j] = c.f2_k[lc];
        if (!FAST2) {
            const PrimeDesc &pd = primes[c.q_id[lc]];
            s2[j] = B2Slow{pd.p, pd.cr1, pd.two_p, pd.r64.op, pd.r64.quo};
        }
    }
    const BehzK2 msk = c.msk_k; // kernel argument: scalar registers
    const B2Patch where = b2_patch_where(c.nB, half);
    // both operands of a tile are fetched while the previous tile is computed: the q residues of this wave's limbs (xr) and the Bsk
    // residues this lane adds in the stage-1 epilogue, (sub, j) -> db[o = 4w + 2j + half][n0 + 32 sub + cl] (dbn)
    u64 xr[KB1], dbn[2][2];
    auto fetch = [&](unsigned t) {
        const u32 t0 = (blockIdx.x * tiles_per_wg + t) * B2_TILE;
        const u32 n = t0 + lane;
#pragma unroll
        for (int i = 0; i < KB1; i++) {
            const u32 l = (u32)(w + 4 * i);
            xr[i] = B2_LOAD(buf_load_u64(rq, (l < (u32)c.L && n < n32) ? (l * n32 + n) * 8u : TROY_BUF_OOB), (lane + n) * 0x9E3779B97F4Aull + l);
        }
#pragma unroll
        for (int sub = 0; sub < 2; sub++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const u32 o = 4 * (u32)w + 2 * j + half;
                const u32 m = t0 + sub * 32 + cl;
                dbn[sub][j] = B2_LOAD(buf_load_u64(rb, (o < (u32)c.nBsk && m < n32) ? (o * n32 + m) * 8u : TROY_BUF_OOB), (lane + m) * 0x9E3779B97F4Aull + o);
            }
    };
    fetch(0);
#pragma unroll
    for (int i = 0; i < 4; i++) buf_store_u64(rout, TROY_BUF_OOB + 8 * i, 0);
    for (unsigned t = 0; t < tiles_per_wg; t++) {
        const u32 n0 = (blockIdx.x * tiles_per_wg + t) * B2_TILE;
        if (n0 >= n32) break;
        // the digits of y_l (canonical: the conversion sums the integers); a padding limb was loaded as 0
#pragma unroll
        for (int i = 0; i < KB1; i++) {
            const int l = w + 4 * i;
            ydig[(((l >> 1) * B2_TILE) + lane) * 2 + (l & 1)] = b2_digits(xr[i]);
        }
        const u64 dbv[2][2] = {{dbn[0][0], dbn[0][1]}, {dbn[1][0], dbn[1][1]}};
        fetch(t + 1 < tiles_per_wg ? t + 1 : 0x7FFFFFu);
        __syncthreads(); // also orders this tile's udig / zsk writes after the previous tile's stage-2 reads
        // ---- stage 1: u_o = (t db_o - conv(t dq)_o) q^-1 [(B/B_o)^-1 | B^-1 for m_sk] mod Bsk_o
        if (owner1) {
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                const unsigned cc = sub * 32 + cl;
                MfmaAcc acc;
                b2_zero(acc);
#pragma unroll
                for (int kb = 0; kb < KB1; kb++) B2_MFMA(af1[kb], b2_frag(ydig, (size_t)(2 * kb + half) * B2_TILE + cc), acc);
                const u64 r0 = b2_finish<0>(acc, k1[0].biaslo + dbv[sub][0], k1[0]);
                const u64 r1 = b2_finish<8>(acc, k1[1].biaslo + dbv[sub][1], k1[1]);
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int o = 4 * w + 2 * j + (int)half;
                    const u64 r = j ? r1 : r0;
                    if (o < c.nB) udig[(((o >> 1) * B2_TILE) + cc) * 2 + (o & 1)] = b2_digits(r);
                    else if (o == c.nB) zsk[cc] = r;
                }
            }
        }
        __syncthreads();
        // ---- stage 2: Shenoy-Kumaresan
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            const unsigned cc = sub * 32 + cl;
            MfmaFrag bf[KB2];
#pragma unroll
            for (int kb = 0; kb < KB2; kb++) bf[kb] = b2_frag(udig, (size_t)(2 * kb + half) * B2_TILE + cc);
            MfmaAcc acc;
            b2_zero(acc);
#pragma unroll
            for (int kb = 0; kb < KB2; kb++) B2_MFMA(amsk[kb], bf[kb], acc);
            // alpha = (conv_{B->m_sk}(z) - z_sk) B^-1 mod m_sk; above m_sk / 2 it stands for a negative value (rns.cpp:905-930)
            const u64 conv = b2_finish<0>(acc, msk.biaslo, msk);
            const u64 z = zsk[cc];
            const u64 alpha = conv >= z ? conv - z : conv + msk.p - z;
            const bool neg = alpha > (msk.p >> 1);
            b2_patch(bf[KB2 - 1], where, b2_digits(neg ? alpha - msk.p : alpha)); // out_l = sum - alpha (B mod q_l)
            u64 r[2] = {0, 0};
            if (owner2) {
                b2_zero(acc);
#pragma unroll
                for (int kb = 0; kb < KB2; kb++) B2_MFMA(af2[kb], bf[kb], acc);
                if (FAST2) {
                    r[0] = b2_finish<0>(acc, k2[0].biaslo, k2[0]);
                    r[1] = b2_finish<8>(acc, k2[1].biaslo, k2[1]);
                } else {
                    r[0] = b2_reduce_slow(b2_recombine<0>(acc, k2[0].biaslo, k2[0].bias1), s2[0]);
                    r[1] = b2_reduce_slow(b2_recombine<8>(acc, k2[1].biaslo, k2[1].bias1), s2[1]);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const u32 l = 4 * (u32)w + 2 * j + half;
                const bool live = (B2_EXP & 16) ? r[j] == ~0ull : (owner2 && l < (u32)c.L && n0 + cc < n32);
                buf_store_u64(rout, live ? (l * n32 + n0 + cc) * 8u : TROY_BUF_OOB, r[j]);
            }
        }
    }
}

// ================================================================ small bases: everything in registers
// With at most 8 output primes (L <= 7: the default parameter sets up to N = 8192) the row-block partition above leaves waves idle --
// two row-blocks, four waves.  Here a wave owns 32 COLUMNS of the tile and all row-blocks: each lane loads the residues of its own
// four limbs (k-block kb, half h: limbs 4kb + 2h, 4kb + 2h + 1) of its own coefficient and builds the B fragments in registers, so
// there is no LDS, no barrier, and every wave does the same work.  In floor + Shenoy-Kumaresan the stage-1 rows are packed in the
// order (row-block, half, j) -> output 4 rb + 2 h + j (BehzDev::f1s_frag), so a lane's two finished outputs of row-block rb ARE the
// two limbs of k-block rb it must feed to stage 2: the second product needs no exchange either (only z' = z_sk B^-1 crosses the
// halves of the wave, one shuffle).
#define B2S_TILE 128 // coefficients per workgroup and step: 32 per wave

__device__ __forceinline__ u64 b2_other_half(u64 v) { return __shfl_xor(v, 32); }

// per-lane conversion constants of this lane's limbs (l = 4 kb + 2 half + pos), or a zero factor for padding limbs
template <int KB> struct B2Limbs { u64 p[KB][2]; Shoup pre[KB][2]; u32 row[KB][2]; };

template <int KB, int RB> __global__ __launch_bounds__(B2_THREADS) void behz2s_extend_kernel(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes,
                                                                                            BehzDev c, u64 N, unsigned tiles_per_wg, const u64 *in2, unsigned split) {
    const unsigned lane = threadIdx.x & 63, half = lane >> 5, cl = lane & 31;
    const unsigned w = threadIdx.x >> 6;
    const u64 poly = blockIdx.y;
    const BufRsrc rin = make_rsrc(poly < split ? in + poly * in_pstride : in2 + (poly - split) * in_pstride, (u32)((u64)c.L * N * 8));
    const BufRsrc rout = make_rsrc(out + poly * out_pstride, (u32)((u64)c.nBsk * N * 8));
    const u32 n32 = (u32)N;
    MfmaFrag af[RB][KB], am[KB];
    BehzK2 k2[RB][2];
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
        am[kb] = b2_frag(c.x_mt_frag, (size_t)kb * 64 + lane);
#pragma unroll
        for (int rb = 0; rb < RB; rb++) af[rb][kb] = b2_frag(c.x_frag, ((size_t)rb * KB + kb) * 64 + lane);
    }
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int o = 4 * rb + 2 * j + (int)half;
            k2[rb][j] = c.x_k[o < c.nBsk ? o : c.nBsk - 1];
        }
    B2Limbs<KB> lm;
#pragma unroll
    for (int kb = 0; kb < KB; kb++)
#pragma unroll
        for (int pos = 0; pos < 2; pos++) {
            const int l = 4 * kb + 2 * (int)half + pos;
            const bool real = l < c.L;
            lm.p[kb][pos] = primes[c.q_id[real ? l : 0]].p;
            lm.pre[kb][pos] = real ? c.ext_pre[l] : Shoup{0, 0};
            lm.row[kb][pos] = real ? (u32)l : 0xFFFFFFFFu;
        }
    const B2Patch where = b2_patch_where(c.L, half);
    u64 xr[KB][2];
    auto fetch = [&](unsigned t) {
        const u32 n = (blockIdx.x * tiles_per_wg + t) * B2S_TILE + 32 * w + cl;
#pragma unroll
        for (int kb = 0; kb < KB; kb++)
#pragma unroll
            for (int pos = 0; pos < 2; pos++)
                xr[kb][pos] = buf_load_u64(rin, (lm.row[kb][pos] != 0xFFFFFFFFu && n < n32) ? (lm.row[kb][pos] * n32 + n) * 8u : TROY_BUF_OOB);
    };
    fetch(0);
    for (unsigned t = 0; t < tiles_per_wg; t++) {
        const u32 n = (blockIdx.x * tiles_per_wg + t) * B2S_TILE + 32 * w + cl;
        if ((blockIdx.x * tiles_per_wg + t) * B2S_TILE >= n32) break;
        MfmaFrag bf[KB];
#pragma unroll
        for (int kb = 0; kb < KB; kb++)
#pragma unroll
            for (int pos = 0; pos < 2; pos++)
                frag_set_word(bf[kb], pos, b2_digits(mul_shoup(xr[kb][pos], lm.pre[kb][pos].op, lm.pre[kb][pos].quo, lm.p[kb][pos]))); // padding limb: 0 * 0
        fetch(t + 1 < tiles_per_wg ? t + 1 : 0x3FFFFFu);
        MfmaAcc acc;
        b2_zero(acc);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) B2_MFMA(am[kb], bf[kb], acc);
        const u32 rsum = (u32)acc.v[0] + ((u32)acc.v[1] << 8) + ((u32)acc.v[2] << 16) + ((u32)acc.v[3] << 24);
        const u64 r_mt = ((u64)rsum * c.neg_inv_q_mod_mt) & 0xFFFFFFFFull;
        b2_patch(bf[KB - 1], where, b2_digits((u64)((long long)r_mt - (long long)((r_mt >> 31) << 32))));
#pragma unroll
        for (int rb = 0; rb < RB; rb++) {
            b2_zero(acc);
#pragma unroll
            for (int kb = 0; kb < KB; kb++) B2_MFMA(af[rb][kb], bf[kb], acc);
            const u64 r0 = b2_finish<0>(acc, k2[rb][0].biaslo, k2[rb][0]), r1 = b2_finish<8>(acc, k2[rb][1].biaslo, k2[rb][1]);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const u32 o = 4 * (u32)rb + 2 * j + half;
                buf_store_u64(rout, (o < (u32)c.nBsk && n < n32) ? (o * n32 + n) * 8u : TROY_BUF_OOB, j ? r1 : r0);
            }
        }
    }
}

// KB1 = k-blocks of the q side (= row-blocks of stage 2), KB2 = k-blocks of the B side + alpha (= row-blocks of stage 1)
template <int KB1, int KB2, bool FAST2> __global__ __launch_bounds__(B2_THREADS) void behz2s_floor_sk_kernel(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride,
                                                                                                            u64 *out, u64 out_pstride, const PrimeDesc *primes, BehzDev c,
                                                                                                            u64 N, unsigned tiles_per_wg) {
    const unsigned lane = threadIdx.x & 63, half = lane >> 5, cl = lane & 31;
    const unsigned w = threadIdx.x >> 6;
    const u64 poly = blockIdx.y;
    const BufRsrc rq = make_rsrc(dq + poly * dq_pstride, (u32)((u64)c.L * N * 8));
    const BufRsrc rb_ = make_rsrc(db + poly * db_pstride, (u32)((u64)c.nBsk * N * 8));
    const BufRsrc rout = make_rsrc(out + poly * out_pstride, (u32)((u64)c.L * N * 8));
    const u32 n32 = (u32)N;
    MfmaFrag af1[KB2][KB1], af2[KB1][KB2], amsk[KB2];
    BehzK2 k1[KB2][2], k2[KB1][2];
    B2Slow s2[KB1][2];
#pragma unroll
    for (int kb = 0; kb < KB2; kb++) amsk[kb] = b2_frag(c.f2_msk_frag, (size_t)kb * 64 + lane);
#pragma unroll
    for (int rb = 0; rb < KB2; rb++) {
#pragma unroll
        for (int kb = 0; kb < KB1; kb++) af1[rb][kb] = b2_frag(c.f1s_frag, ((size_t)rb * KB1 + kb) * 64 + lane);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int o = 4 * rb + 2 * (int)half + j; // the permuted stage-1 order
            k1[rb][j] = c.f1_k[o < c.nBsk ? o : c.nBsk - 1];
        }
    }
#pragma unroll
    for (int rb = 0; rb < KB1; rb++) {
#pragma unroll
        for (int kb = 0; kb < KB2; kb++) af2[rb][kb] = b2_frag(c.f2_frag, ((size_t)rb * KB2 + kb) * 64 + lane);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int l = 4 * rb + 2 * j + (int)half, lc = l < c.L ? l : c.L - 1;
            k2[rb][j] = c.f2_k[lc];
            if (!FAST2) {
                const PrimeDesc &pd = primes[c.q_id[lc]];
                s2[rb][j] = B2Slow{pd.p, pd.cr1, pd.two_p, pd.r64.op, pd.r64.quo};
            }
        }
    }
    const BehzK2 msk = c.msk_k;
    const B2Patch where = b2_patch_where(c.nB, half);
    u64 xr[KB1][2], dbn[KB2][2];
    auto fetch = [&](unsigned t) {
        const u32 n = (blockIdx.x * tiles_per_wg + t) * B2S_TILE + 32 * w + cl;
#pragma unroll
        for (int kb = 0; kb < KB1; kb++)
#pragma unroll
            for (int pos = 0; pos < 2; pos++) {
                const u32 l = 4 * (u32)kb + 2 * half + pos;
                xr[kb][pos] = buf_load_u64(rq, (l < (u32)c.L && n < n32) ? (l * n32 + n) * 8u : TROY_BUF_OOB);
            }
#pragma unroll
        for (int rb = 0; rb < KB2; rb++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const u32 o = 4 * (u32)rb + 2 * half + j;
                dbn[rb][j] = buf_load_u64(rb_, (o < (u32)c.nBsk && n < n32) ? (o * n32 + n) * 8u : TROY_BUF_OOB);
            }
    };
    fetch(0);
    for (unsigned t = 0; t < tiles_per_wg; t++) {
        const u32 n = (blockIdx.x * tiles_per_wg + t) * B2S_TILE + 32 * w + cl;
        if ((blockIdx.x * tiles_per_wg + t) * B2S_TILE >= n32) break;
        MfmaFrag bf1[KB1], bf2[KB2];
#pragma unroll
        for (int kb = 0; kb < KB1; kb++)
#pragma unroll
            for (int pos = 0; pos < 2; pos++) frag_set_word(bf1[kb], pos, b2_digits(xr[kb][pos])); // pre-scaled, canonical; padding limbs loaded as 0
        u64 dbv[KB2][2];
#pragma unroll
        for (int rb = 0; rb < KB2; rb++) { dbv[rb][0] = dbn[rb][0]; dbv[rb][1] = dbn[rb][1]; }
        fetch(t + 1 < tiles_per_wg ? t + 1 : 0x3FFFFFu);
        // ---- stage 1: this lane's outputs 4 rb + 2 half + j become limbs 4 rb + 2 half + j of its stage-2 operand
        u64 z = 0;
        MfmaAcc acc;
#pragma unroll
        for (int rb = 0; rb < KB2; rb++) {
            b2_zero(acc);
#pragma unroll
            for (int kb = 0; kb < KB1; kb++) B2_MFMA(af1[rb][kb], bf1[kb], acc);
            const u64 r0 = b2_finish<0>(acc, k1[rb][0].biaslo + dbv[rb][0], k1[rb][0]), r1 = b2_finish<8>(acc, k1[rb][1].biaslo + dbv[rb][1], k1[rb][1]);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int o = 4 * rb + 2 * (int)half + j;
                const u64 r = j ? r1 : r0;
                frag_set_word(bf2[rb], j, o < c.nB ? b2_digits(r) : 0);
                if (o == c.nB) z = r;
            }
        }
        // z' sits in the lanes of one half (the one that owns output nB): the other half of the wave fetches it
        {
            const bool mine = half == (unsigned)((c.nB & 3) >> 1);
            const u64 other = b2_other_half(z);
            z = mine ? z : other;
        }
        // ---- stage 2
        b2_zero(acc);
#pragma unroll
        for (int kb = 0; kb < KB2; kb++) B2_MFMA(amsk[kb], bf2[kb], acc);
        const u64 conv = b2_finish<0>(acc, msk.biaslo, msk);
        const u64 alpha = conv >= z ? conv - z : conv + msk.p - z;
        const bool neg = alpha > (msk.p >> 1);
        b2_patch(bf2[KB2 - 1], where, b2_digits(neg ? alpha - msk.p : alpha));
#pragma unroll
        for (int rb = 0; rb < KB1; rb++) {
            b2_zero(acc);
#pragma unroll
            for (int kb = 0; kb < KB2; kb++) B2_MFMA(af2[rb][kb], bf2[kb], acc);
            u64 r0, r1;
            if (FAST2) {
                r0 = b2_finish<0>(acc, k2[rb][0].biaslo, k2[rb][0]);
                r1 = b2_finish<8>(acc, k2[rb][1].biaslo, k2[rb][1]);
            } else {
                r0 = b2_reduce_slow(b2_recombine<0>(acc, k2[rb][0].biaslo, k2[rb][0].bias1), s2[rb][0]);
                r1 = b2_reduce_slow(b2_recombine<8>(acc, k2[rb][1].biaslo, k2[rb][1].bias1), s2[rb][1]);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const u32 l = 4 * (u32)rb + 2 * j + half;
                buf_store_u64(rout, (l < (u32)c.L && n < n32) ? (l * n32 + n) * 8u : TROY_BUF_OOB, j ? r1 : r0);
            }
        }
    }
}

// tiles a workgroup walks (fragments and per-lane constants are set up once per workgroup); small launches -- a few polynomials -- take fewer, so that
// the grid still covers the chip (plan_per_workgroup, kernels.h)
// probe builds: TROYHIP_BEHZ=mfma keeps the matrix-core kernels at the base sizes the register-resident FP64 form (behz3.hip) takes (A/B runs)
static bool behz3_enabled() {
    static const bool v = [] { const char *e = probe_env("TROYHIP_BEHZ"); return !(e && e[0] == 'm'); }();
    return v;
}
static unsigned b2_tiles_per_wg(u64 tiles, u64 polys) { return plan_per_workgroup(tiles, tiles >= 64 ? B2_TPW : 1, polys); }
// small-base kernels: a wave takes 32 columns per step, so the per-workgroup setup wants more steps
static unsigned b2s_tiles_per_wg(u64 tiles, u64 polys) { // 8 / 16 / 32 / 64 at N = 8192 and a large batch: 389 / 352 / 347 / 390 us
    return plan_per_workgroup(tiles, tiles >= 64 ? 16u : (tiles >= 16 ? (unsigned)(tiles / 4) : 1u), polys);
}

void launch_behz2_extend(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, const BehzDev &c, u64 N, u64 polys, hipStream_t s, const u64 *in2,
                         u64 split) {
    if (behz3_supported(c) && behz3_enabled()) return launch_behz3_extend(in, in_pstride, out, out_pstride, c, N, polys, s, in2, split); // small bases of narrow primes
    if (!in2) split = polys;
    stats::counter(stats::BEHZ_MFMA_LAUNCHES).fetch_add(1, std::memory_order_relaxed);
    if (in2 && polys > 65535) throw Error(ST_LOGIC_ERROR, "behz2 extend: the two-operand form is for small launches");
    const int kb = (c.L + 1 + 3) / 4;
    const bool small = c.f1s_frag && kb <= 2 && c.nBsk <= 8; // everything-in-registers form
    const u64 tiles = ceil_div(N, (u64)(small ? B2S_TILE : B2_TILE));
    const unsigned tpw = small ? b2s_tiles_per_wg(tiles, polys) : b2_tiles_per_wg(tiles, polys);
    for (u64 p0 = 0; p0 < polys; p0 += 65535) { // gridDim.y limit
        const u64 np = polys - p0 < 65535 ? polys - p0 : 65535;
        const dim3 grid((unsigned)ceil_div(tiles, (u64)tpw), (unsigned)np);
        const u64 *pi = in + p0 * in_pstride;
        u64 *po = out + p0 * out_pstride;
        const unsigned sp = (unsigned)(split > p0 ? (split - p0 < 65535 ? split - p0 : 65535) : 0); // polynomials of this slice that come from `in`
#define B2S_EXT(KB_, RB_) TROY_LAUNCH(HIP_KERNEL_NAME(behz2s_extend_kernel<KB_, RB_>), grid, dim3(B2_THREADS), 0, s, pi, in_pstride, po, out_pstride, primes, c, N, tpw, in2, sp)
        if (small) {
            const int rb = (c.nBsk + 3) / 4;
            if (kb == 1 && rb == 1) B2S_EXT(1, 1);
            else if (kb == 1) B2S_EXT(1, 2);
            else B2S_EXT(2, 2);
            continue;
        }
#undef B2S_EXT
#define B2_EXT(KB_) TROY_LAUNCH(HIP_KERNEL_NAME(behz2_extend_kernel<KB_>), grid, dim3(B2_THREADS), 0, s, pi, in_pstride, po, out_pstride, primes, c, N, tpw, in2, sp)
        switch (kb) {
        case 1: B2_EXT(1); break;
        case 2: B2_EXT(2); break;
        case 3: B2_EXT(3); break;
        case 4: B2_EXT(4); break;
        default: throw Error(ST_LOGIC_ERROR, "BEHZ base sizes outside the matrix-core form");
        }
#undef B2_EXT
    }
    launch_check("behz2_extend_kernel");
}

void launch_behz2_floor_sk(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride, u64 *out, u64 out_pstride, const PrimeDesc *primes, const BehzDev &c, u64 N,
                           u64 polys, hipStream_t s) {
    if (behz3_supported(c) && behz3_enabled()) return launch_behz3_floor_sk(dq, dq_pstride, db, db_pstride, out, out_pstride, c, N, polys, s);
    stats::counter(stats::BEHZ_MFMA_LAUNCHES).fetch_add(1, std::memory_order_relaxed);
    const int kb1 = (c.L + 3) / 4, kb2 = (c.nB + 1 + 3) / 4; // kb2 is kb1 or kb1 + 1 (nB is L or L + 1)
    const bool small = c.f1s_frag && kb2 <= 2;
    const u64 tiles = ceil_div(N, (u64)(small ? B2S_TILE : B2_TILE));
    const unsigned tpw = small ? b2s_tiles_per_wg(tiles, polys) : b2_tiles_per_wg(tiles, polys);
    for (u64 p0 = 0; p0 < polys; p0 += 65535) {
        const u64 np = polys - p0 < 65535 ? polys - p0 : 65535;
        const dim3 grid((unsigned)ceil_div(tiles, (u64)tpw), (unsigned)np);
        const u64 *pq = dq + p0 * dq_pstride, *pb = db + p0 * db_pstride;
        u64 *po = out + p0 * out_pstride;
#define B2S_FLOOR(K1_, K2_)                                                                                               \
    do {                                                                                                                 \
        if (c.f2_fast) TROY_LAUNCH(HIP_KERNEL_NAME(behz2s_floor_sk_kernel<K1_, K2_, true>), grid, dim3(B2_THREADS), 0, s, pq, dq_pstride, pb, db_pstride, po, out_pstride, primes, c, N, tpw); \
        else TROY_LAUNCH(HIP_KERNEL_NAME(behz2s_floor_sk_kernel<K1_, K2_, false>), grid, dim3(B2_THREADS), 0, s, pq, dq_pstride, pb, db_pstride, po, out_pstride, primes, c, N, tpw); \
    } while (0)
        if (small) {
            if (kb1 == 1 && kb2 == 1) B2S_FLOOR(1, 1);
            else if (kb1 == 1) B2S_FLOOR(1, 2);
            else B2S_FLOOR(2, 2);
            continue;
        }
#undef B2S_FLOOR
#define B2_FLOOR(K1_, K2_, F_)                                                                                            \
    TROY_LAUNCH(HIP_KERNEL_NAME(behz2_floor_sk_kernel<K1_, K2_, F_>), grid, dim3(B2_THREADS), 0, s, pq, dq_pstride, pb, db_pstride, po, out_pstride, primes, c, N, tpw)
#define B2_FLOOR_F(K1_, K2_)                                                                                              \
    do {                                                                                                                 \
        if (c.f2_fast) B2_FLOOR(K1_, K2_, true);                                                                         \
        else B2_FLOOR(K1_, K2_, false);                                                                                  \
    } while (0)
        switch (kb1 * 8 + kb2) {
        case 1 * 8 + 1: B2_FLOOR_F(1, 1); break;
        case 1 * 8 + 2: B2_FLOOR_F(1, 2); break;
        case 2 * 8 + 2: B2_FLOOR_F(2, 2); break;
        case 2 * 8 + 3: B2_FLOOR_F(2, 3); break;
        case 3 * 8 + 3: B2_FLOOR_F(3, 3); break;
        case 3 * 8 + 4: B2_FLOOR_F(3, 4); break;
        case 4 * 8 + 4: B2_FLOOR_F(4, 4); break;
        default: throw Error(ST_LOGIC_ERROR, "BEHZ base sizes outside the matrix-core form");
        }
#undef B2_FLOOR_F
#undef B2_FLOOR
    }
    launch_check("behz2_floor_sk_kernel");
}

} // namespace troyhip
