// behz3.hip -- BEHZ base conversion for SMALL bases of narrow primes: one coefficient per thread, everything in registers, FP64.
//
// Same contracts as behz2_extend / behz2_floor_sk (behz2.hip; reference: fastbconvmTilde + smMrq, fastFloor + fastbconvSk,
// src/utils/rns_cuda.cu:365-508, CPU twins src/utils/rns.cpp:879-1037) and the same canonical residues.
//
// The matrix-core kernels of behz2.hip are shaped for L >= 8: at L = 4 (BASELINE configs[1]: BFV N = 8192) the 5 x 5 / 5 x 4 matrices fill a
// fraction of a 32 x 32 x 32 int8 tile, and extension + floor run at 0.39 / 0.53 of the HBM roofline, 23 % of that step.  When every prime of the
// level -- the q primes and the library's own auxiliary base (hostmath.cpp: RnsLevel::build picks primes below 2^50 when that costs no extra limb) --
// lies in [2^33, 2^50), the whole conversion is a handful of exact FP64 modular products per coefficient (fpmod.h: error-free product, rounded
// quotient; every intermediate an exact integer below 2^53):
//   extension   y_l = x_l (m~ (q/q_l)^-1) mod q_l (canonical);  r = centred((sum_l y_l (q/q_l)) (-q^-1) mod 2^32)  [32-bit integer arithmetic];
//               out_o = (sum_l y_l E[o][l] + r C[o]) mod p_o,    E, C = the folded rows of context.cpp (m~^-1 inside)
//   floor + SK  u_o = (sum_l dq'_l F[o][l] + db'_o) mod p_o  (inputs PRE-SCALED by the inverse transforms, BehzDev::floor_desc), canonical;
//               alpha = centred((sum_b u_b S[b] - z') mod m_sk);   out_l = (sum_b u_b G[l][b] + alpha H[l]) mod q_l
// -- term for term the expressions the other two forms evaluate, on canonical operands with canonical intermediates, so the stored residues
// are the same numbers.  A thread keeps y (u) in L (nB) doubles and streams the outputs; the constants are wave-uniform scalar loads.
// No LDS, no barrier, ~300 instructions and 72-104 bytes per coefficient: memory-bound.
#include "kernels.h"
#include "fpmod.h"

namespace troyhip {

#define B3_THREADS 256
#ifndef TROYHIP_CPU_EMUL
#include <cstdio>
#define B3_KTAG(...) do { if (ktime::enabled) { static thread_local char tagbuf[64]; std::snprintf(tagbuf, sizeof(tagbuf), __VA_ARGS__); ktime::tag = tagbuf; } } while (0)
#else
#define B3_KTAG(...)
#endif

#ifdef TROYHIP_CPU_EMUL
typedef const double *b3_cd;
typedef const u32 *b3_cu32;
#else
typedef const __attribute__((address_space(4))) double *b3_cd; // wave-uniform addresses: scalar loads
typedef const __attribute__((address_space(4))) u32 *b3_cu32;
#endif

// x mod p as an exact integer double in [0, p) (|x| < 2^53)
__device__ __forceinline__ double b3_canon(double x, const FpPrime &c) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
    const double r = fp_reduce(x, c); // |r| < p / 2 + 2
    return r < 0.0 ? r + c.p : r;
}

// table layouts (doubles), built by context.cpp (Context::build_level):
//   extension:  [L] (pre, pre / q_l) | [L] (q_l, 1 / q_l) | [NB] (p_o, 1 / p_o) | [NB][L + 1] (w, w / p_o): E[o][0 .. L-1], C[o]
//   floor + SK: [L] (q_l, 1 / q_l) | [NB] (p_o, 1 / p_o), the last one m_sk | [NB][L] (F, F / p_o) | [NB - 1] (S, S / m_sk) | [L][NB] (w, w / q_l): G[l][0 .. nB-1], H[l]
template <int L, int NB> __global__ __launch_bounds__(B3_THREADS) void behz3_extend_kernel(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, BehzDev c, u64 N, const u64 *in2,
                                                                                          unsigned split) {
    const u64 n = (u64)blockIdx.x * B3_THREADS + threadIdx.x;
    if (n >= N) return;
    const u64 poly = blockIdx.y;
    const u64 *x = (poly < split ? in + poly * in_pstride : in2 + (poly - split) * in_pstride) + n;
    u64 *o_ = out + poly * out_pstride + n;
    const b3_cd T = (b3_cd)c.fp_ext;
    double y[L];
    u32 rsum = 0;
#pragma unroll
    for (int l = 0; l < L; l++) {
        const FpPrime fq{T[2 * L + 2 * l], T[2 * L + 2 * l + 1]};
        const double v = b3_canon(fp_mulmod_wp(fp_from_u64(x[(u64)l * N]), T[2 * l], T[2 * l + 1], fq), fq); // canonical y_l, an exact integer
        y[l] = v;
        rsum += (u32)fp_bits(v + 4503599627370496.0) * ((b3_cu32)c.ext_mt_row)[l]; // low word of 2^52 + y_l = y_l mod 2^32
    }
    const u32 r_mt = rsum * (u32)c.neg_inv_q_mod_mt;
    const double r = (double)(int)r_mt; // the centred representative of r (rns.cpp:966-975): r >= 2^31 stands for r - 2^32
    const b3_cd P = T + 4 * L, M = T + 4 * L + 2 * NB;
#pragma unroll
    for (int o = 0; o < NB; o++) {
        const FpPrime fp{P[2 * o], P[2 * o + 1]};
        const b3_cd row = M + 2 * (L + 1) * o;
        double s = fp_mulmod_wp(r, row[2 * L], row[2 * L + 1], fp);
#pragma unroll
        for (int l = 0; l < L; l++) s += fp_mulmod_wp(y[l], row[2 * l], row[2 * l + 1], fp); // each term within 0.75 p: the sum stays far below 2^53
        o_[(u64)o * N] = fp_canonical(s, fp, 0);
    }
}

template <int L, int NB> __global__ __launch_bounds__(B3_THREADS) void behz3_floor_sk_kernel(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride, u64 *out, u64 out_pstride,
                                                                                            BehzDev c, u64 N) {
    constexpr int nB = NB - 1;
    const u64 n = (u64)blockIdx.x * B3_THREADS + threadIdx.x;
    if (n >= N) return;
    const u64 poly = blockIdx.y;
    const u64 *xq = dq + poly * dq_pstride + n, *xb = db + poly * db_pstride + n;
    u64 *o_ = out + poly * out_pstride + n;
    const b3_cd T = (b3_cd)c.fp_floor;
    const b3_cd Q = T, P = T + 2 * L, F = T + 2 * L + 2 * NB, S = F + 2 * NB * L, G = S + 2 * nB;
    double yq[L];
#pragma unroll
    for (int l = 0; l < L; l++) yq[l] = fp_from_u64(xq[(u64)l * N]); // pre-scaled, canonical
    // ---- stage 1: u_o for the B primes (kept), z' for m_sk
    double u[NB];
#pragma unroll
    for (int o = 0; o < NB; o++) {
        const FpPrime fp{P[2 * o], P[2 * o + 1]};
        double s = fp_from_u64(xb[(u64)o * N]);
#pragma unroll
        for (int l = 0; l < L; l++) s += fp_mulmod_wp(yq[l], F[2 * (L * o + l)], F[2 * (L * o + l) + 1], fp);
        u[o] = b3_canon(s, fp);
    }
    // ---- Shenoy-Kumaresan: alpha = (conv_{B -> m_sk}(u) - z') mod m_sk (B^-1 is folded into S and into z'), centred around m_sk / 2
    const FpPrime fs{P[2 * nB], P[2 * nB + 1]};
    double a = -u[nB];
#pragma unroll
    for (int b = 0; b < nB; b++) a += fp_mulmod_wp(u[b], S[2 * b], S[2 * b + 1], fs);
    a = b3_canon(a, fs);
    const double half = __builtin_floor(fs.p * 0.5); // (m_sk - 1) / 2: exact
    const double alpha = a > half ? a - fs.p : a;
    // ---- back to base q
#pragma unroll
    for (int l = 0; l < L; l++) {
        const FpPrime fq{Q[2 * l], Q[2 * l + 1]};
        const b3_cd row = G + 2 * NB * l;
        double s = fp_mulmod_wp(alpha, row[2 * nB], row[2 * nB + 1], fq);
#pragma unroll
        for (int b = 0; b < nB; b++) s += fp_mulmod_wp(u[b], row[2 * b], row[2 * b + 1], fq);
        o_[(u64)l * N] = fp_canonical(s, fq, 0);
    }
}

bool behz3_supported(const BehzDev &c) { return c.fp_ext && c.fp_floor && c.L >= 1 && c.L <= 6 && (c.nBsk == c.L + 1 || c.nBsk == c.L + 2); }

template <int L> static void launch_ext(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const BehzDev &c, u64 N, u64 polys, hipStream_t s, const u64 *in2, u64 split) {
    for (u64 p0 = 0; p0 < polys; p0 += 65535) { // gridDim.y limit
        const u64 np = polys - p0 < 65535 ? polys - p0 : 65535;
        const dim3 grid(ceil_div(N, B3_THREADS), (unsigned)np);
        const u64 *pi = in + p0 * in_pstride;
        u64 *po = out + p0 * out_pstride;
        const unsigned sp = (unsigned)(split > p0 ? (split - p0 < 65535 ? split - p0 : 65535) : 0);
        const u64 *pi2 = in2 ? in2 + (p0 > split ? (p0 - split) * in_pstride : 0) : nullptr;
        B3_KTAG("behz3_extend_kernel<%d, %d>", L, c.nBsk);
        if (c.nBsk == L + 1) TROY_LAUNCH(HIP_KERNEL_NAME(behz3_extend_kernel<L, L + 1>), grid, dim3(B3_THREADS), 0, s, pi, in_pstride, po, out_pstride, c, N, pi2, sp);
        else TROY_LAUNCH(HIP_KERNEL_NAME(behz3_extend_kernel<L, L + 2>), grid, dim3(B3_THREADS), 0, s, pi, in_pstride, po, out_pstride, c, N, pi2, sp);
    }
}
template <int L> static void launch_floor(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride, u64 *out, u64 out_pstride, const BehzDev &c, u64 N, u64 polys, hipStream_t s) {
    for (u64 p0 = 0; p0 < polys; p0 += 65535) {
        const u64 np = polys - p0 < 65535 ? polys - p0 : 65535;
        const dim3 grid(ceil_div(N, B3_THREADS), (unsigned)np);
        const u64 *pq = dq + p0 * dq_pstride, *pb = db + p0 * db_pstride;
        u64 *po = out + p0 * out_pstride;
        B3_KTAG("behz3_floor_sk_kernel<%d, %d>", L, c.nBsk);
        if (c.nBsk == L + 1) TROY_LAUNCH(HIP_KERNEL_NAME(behz3_floor_sk_kernel<L, L + 1>), grid, dim3(B3_THREADS), 0, s, pq, dq_pstride, pb, db_pstride, po, out_pstride, c, N);
        else TROY_LAUNCH(HIP_KERNEL_NAME(behz3_floor_sk_kernel<L, L + 2>), grid, dim3(B3_THREADS), 0, s, pq, dq_pstride, pb, db_pstride, po, out_pstride, c, N);
    }
}
void launch_behz3_extend(const u64 *in, u64 in_pstride, u64 *out, u64 out_pstride, const BehzDev &c, u64 N, u64 polys, hipStream_t s, const u64 *in2, u64 split) {
    if (!in2) split = polys;
    stats::counter(stats::BEHZ_FP_LAUNCHES).fetch_add(1, std::memory_order_relaxed);
    switch (c.L) {
    case 1: launch_ext<1>(in, in_pstride, out, out_pstride, c, N, polys, s, in2, split); break;
    case 2: launch_ext<2>(in, in_pstride, out, out_pstride, c, N, polys, s, in2, split); break;
    case 3: launch_ext<3>(in, in_pstride, out, out_pstride, c, N, polys, s, in2, split); break;
    case 4: launch_ext<4>(in, in_pstride, out, out_pstride, c, N, polys, s, in2, split); break;
    case 5: launch_ext<5>(in, in_pstride, out, out_pstride, c, N, polys, s, in2, split); break;
    case 6: launch_ext<6>(in, in_pstride, out, out_pstride, c, N, polys, s, in2, split); break;
    default: throw Error(ST_LOGIC_ERROR, "behz3: base sizes outside the register-resident form");
    }
    launch_check("behz3_extend_kernel");
}
void launch_behz3_floor_sk(const u64 *dq, u64 dq_pstride, const u64 *db, u64 db_pstride, u64 *out, u64 out_pstride, const BehzDev &c, u64 N, u64 polys, hipStream_t s) {
    stats::counter(stats::BEHZ_FP_LAUNCHES).fetch_add(1, std::memory_order_relaxed);
    switch (c.L) {
    case 1: launch_floor<1>(dq, dq_pstride, db, db_pstride, out, out_pstride, c, N, polys, s); break;
    case 2: launch_floor<2>(dq, dq_pstride, db, db_pstride, out, out_pstride, c, N, polys, s); break;
    case 3: launch_floor<3>(dq, dq_pstride, db, db_pstride, out, out_pstride, c, N, polys, s); break;
    case 4: launch_floor<4>(dq, dq_pstride, db, db_pstride, out, out_pstride, c, N, polys, s); break;
    case 5: launch_floor<5>(dq, dq_pstride, db, db_pstride, out, out_pstride, c, N, polys, s); break;
    case 6: launch_floor<6>(dq, dq_pstride, db, db_pstride, out, out_pstride, c, N, polys, s); break;
    default: throw Error(ST_LOGIC_ERROR, "behz3: base sizes outside the register-resident form");
    }
    launch_check("behz3_floor_sk_kernel");
}

} // namespace troyhip
