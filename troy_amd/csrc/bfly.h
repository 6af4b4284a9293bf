// bfly.h -- hand-scheduled Harvey butterflies for gfx950, four at a time.
//
// gfx950 has no 64-bit integer multiplier and (measured, profiles/r01_microbench_valu.txt) every carry-propagating or
// 64-bit VALU op costs ~1.5-1.8x a plain v_add_u32, so the NTT is bound by the NUMBER of VALU instructions per
// butterfly.  hipcc's lowering of the textbook butterfly (ntt.hip) issues ~31; the formulation below issues ~20:
//   * wider lazy ranges: forward values live in [0,8p), inverse values in [0,4p) (p < 2^61 so 8p < 2^64).  This
//     admits a quotient estimate that drops the low x low partial product: q~ = floor(y*w'/2^64) - {0,1}, so the
//     lazy product w*y - q~*p lies in [0,3p) and costs 3 v_mad_u64_u32 instead of 1 v_mul_hi + 3 v_mad + fix-ups;
//   * the conditional subtraction uses the borrow of v_sub_co/v_subb directly (no v_cmp_u64);
//   * w*y - q~*p is evaluated as w*y + q~*(2^64 - p): both products accumulate through v_mad_u64_u32's free 64-bit
//     addend, which also absorbs the "+ u" of X' = u + v;  Y' = 2u + 3p - X';
//   * every instruction is pinned with inline asm so LLVM cannot re-associate the limbs back into generic 64-bit
//     multiplies (it does, and then emits multiplications by zero and ~8 v_mov per butterfly).
// Every stage of every round has exactly FOUR independent butterflies per thread; the carry-chained pieces are
// written for four butterflies at once and interleaved, because gfx940-class hardware needs 2 wait states between a
// VALU instruction that writes a carry/mask SGPR (vcc) and a VALU instruction that reads it, and the compiler's hazard
// recogniser does not look inside inline asm: interleaving the four chains satisfies the hazard with useful work
// instead of s_nop.
// All values stay congruent to the reference's butterflies (src/utils/dwthandler.h:88-372) modulo p; the caller
// reduces stored results to [0,p), so outputs are bit-identical.
#pragma once
#include "modarith.h"

namespace troyhip {

struct PrimeConst { u64 p, two_p, three_p, four_p, negp; }; // negp = 2^64 - p
__device__ __forceinline__ PrimeConst make_prime_const(u64 p) { return PrimeConst{p, 2 * p, 3 * p, 4 * p, 0 - p}; }

__device__ __forceinline__ u32 lo32(u64 x) { return (u32)x; }
__device__ __forceinline__ u32 hi32(u64 x) { return (u32)(x >> 32); }
__device__ __forceinline__ u64 mk64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

#ifdef TROYHIP_CPU_EMUL
// x[i] = x[i] >= m ? x[i] - m : x[i]
__device__ __forceinline__ void csub4(u64 (&x)[4], u64 m) { for (int i = 0; i < 4; i++) x[i] = x[i] >= m ? x[i] - m : x[i]; }
__device__ __forceinline__ void sub4(u64 (&r)[4], const u64 (&a)[4], const u64 (&b)[4]) { for (int i = 0; i < 4; i++) r[i] = a[i] - b[i]; }
__device__ __forceinline__ u64 mulhi_approx1(u64 y, u64 wq) { // floor(y*wq/2^64) - {0,1}: low x low product dropped
    const u64 y0 = lo32(y), y1 = hi32(y), q0 = lo32(wq), q1 = hi32(wq);
    const u128 s = (u128)(y1 * q0) + (u128)(y0 * q1);
    return y1 * q1 + (u64)(s >> 32);
}
__device__ __forceinline__ void mulhi_approx4(u64 (&q)[4], const u64 (&y)[4], const Shoup (&w)[4]) { for (int i = 0; i < 4; i++) q[i] = mulhi_approx1(y[i], w[i].quo); }
__device__ __forceinline__ u64 mul_acc(u64 acc, u64 y, u64 w, u64 q, u64 negp) { return acc + w * y + q * negp; }
__device__ __forceinline__ void mulhi_approx4_u(u64 (&q)[4], const u64 (&y)[4], const Shoup (&w)[4]) { mulhi_approx4(q, y, w); }
__device__ __forceinline__ u64 mul_acc_u(u64 acc, u64 y, u64 w, u64 q, u64 negp) { return mul_acc(acc, y, w, q, negp); }
__device__ __forceinline__ u64 mul_acc0(u64 y, u64 w, u64 q, u64 negp) { return mul_acc(0, y, w, q, negp); }
__device__ __forceinline__ u64 mul_acc0_u(u64 y, u64 w, u64 q, u64 negp) { return mul_acc(0, y, w, q, negp); }
__device__ __forceinline__ void mulhi_exact4(u64 (&q)[4], const u64 (&y)[4], const Shoup (&w)[4]) { for (int i = 0; i < 4; i++) q[i] = (u64)(((u128)y[i] * w[i].quo) >> 64); }
__device__ __forceinline__ void mulhi_exact4_u(u64 (&q)[4], const u64 (&y)[4], const Shoup (&w)[4]) { mulhi_exact4(q, y, w); }
#else
// x[i] = x[i] >= m ? x[i] - m : x[i]   (x < 2m).  m is wave-uniform; its high word must sit in a VGPR because
// vcc + an SGPR would exceed the single constant-bus read a gfx9 VALU instruction may make.
__device__ __forceinline__ void csub4(u64 (&x)[4], u64 m) {
    u32 a0, a1, b0, b1, c0, c1, d0, d1;
    u64 sb, sc, sd;
    asm("v_subrev_co_u32 %0, vcc, %19, %11\n\t"
        "v_subrev_co_u32 %2, %8, %19, %13\n\t"
        "v_subrev_co_u32 %4, %9, %19, %15\n\t"
        "v_subrev_co_u32 %6, %10, %19, %17\n\t"
        "v_subbrev_co_u32 %1, vcc, %20, %12, vcc\n\t"
        "v_subbrev_co_u32 %3, %8, %20, %14, %8\n\t"
        "v_subbrev_co_u32 %5, %9, %20, %16, %9\n\t"
        "v_subbrev_co_u32 %7, %10, %20, %18, %10\n\t"
        "v_cndmask_b32 %0, %0, %11, vcc\n\t"
        "v_cndmask_b32 %1, %1, %12, vcc\n\t"
        "v_cndmask_b32 %2, %2, %13, %8\n\t"
        "v_cndmask_b32 %3, %3, %14, %8\n\t"
        "v_cndmask_b32 %4, %4, %15, %9\n\t"
        "v_cndmask_b32 %5, %5, %16, %9\n\t"
        "v_cndmask_b32 %6, %6, %17, %10\n\t"
        "v_cndmask_b32 %7, %7, %18, %10"
        : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1), "=&v"(c0), "=&v"(c1), "=&v"(d0), "=&v"(d1), "=&s"(sb), "=&s"(sc), "=&s"(sd)
        : "v"(lo32(x[0])), "v"(hi32(x[0])), "v"(lo32(x[1])), "v"(hi32(x[1])), "v"(lo32(x[2])), "v"(hi32(x[2])), "v"(lo32(x[3])), "v"(hi32(x[3])),
          "s"(lo32(m)), "v"(hi32(m))
        : "vcc");
    x[0] = mk64(a0, a1); x[1] = mk64(b0, b1); x[2] = mk64(c0, c1); x[3] = mk64(d0, d1);
}
// r[i] = a[i] - b[i]
__device__ __forceinline__ void sub4(u64 (&r)[4], const u64 (&a)[4], const u64 (&b)[4]) {
    u32 r0, r1, r2, r3, r4, r5, r6, r7;
    u64 sb, sc, sd;
    asm("v_sub_co_u32 %0, vcc, %11, %19\n\t"
        "v_sub_co_u32 %2, %8, %13, %21\n\t"
        "v_sub_co_u32 %4, %9, %15, %23\n\t"
        "v_sub_co_u32 %6, %10, %17, %25\n\t"
        "v_subb_co_u32 %1, vcc, %12, %20, vcc\n\t"
        "v_subb_co_u32 %3, %8, %14, %22, %8\n\t"
        "v_subb_co_u32 %5, %9, %16, %24, %9\n\t"
        "v_subb_co_u32 %7, %10, %18, %26, %10"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&s"(sb), "=&s"(sc), "=&s"(sd)
        : "v"(lo32(a[0])), "v"(hi32(a[0])), "v"(lo32(a[1])), "v"(hi32(a[1])), "v"(lo32(a[2])), "v"(hi32(a[2])), "v"(lo32(a[3])), "v"(hi32(a[3])),
          "v"(lo32(b[0])), "v"(hi32(b[0])), "v"(lo32(b[1])), "v"(hi32(b[1])), "v"(lo32(b[2])), "v"(hi32(b[2])), "v"(lo32(b[3])), "v"(hi32(b[3]))
        : "vcc");
    r[0] = mk64(r0, r1); r[1] = mk64(r2, r3); r[2] = mk64(r4, r5); r[3] = mk64(r6, r7);
}
// q~[i] = y1*q1 + floor((y1*q0 + y0*q1) / 2^32)  =  floor(y*wq/2^64) - {0,1}
// TWC = "v": per-lane twiddles in VGPRs; TWC = "s": wave-uniform twiddles read straight from SGPRs (one scalar operand per
// instruction, as the gfx9 constant bus allows) -- no v_mov copies and no VGPRs for them
// (Round 5, measured and dropped: the pair (hi32(s), carry) the last multiply-add takes costs a v_mov per butterfly -- the compiler cannot place the carry
// next to hi32(s) because gfx90a-class VGPR tuples must be even-aligned and hi32(s) is the ODD half of its own pair; the same rule rejects a hand-written
// block with fixed scratch registers ("vgpr tuples must be 64 bit aligned").  Every alternative that avoids the pair costs at least one instruction more.)
#define TROY_DEF_MULHI_APPROX4(NAME, TWC)                                                                                                     \
    __device__ __forceinline__ void NAME(u64 (&q)[4], const u64 (&y)[4], const Shoup (&w)[4]) {                                               \
        u64 s0, s1, s2, s3, sb, sc, sd;                                                                                                       \
        u32 c0, c1, c2, c3;                                                                                                                   \
        asm("v_mad_u64_u32 %0, vcc, %12, %13, 0\n\t"                                                                                          \
            "v_mad_u64_u32 %1, %8, %16, %17, 0\n\t"                                                                                           \
            "v_mad_u64_u32 %2, %9, %20, %21, 0\n\t"                                                                                           \
            "v_mad_u64_u32 %3, %10, %24, %25, 0\n\t"                                                                                          \
            "v_mad_u64_u32 %0, vcc, %11, %14, %0\n\t"                                                                                         \
            "v_mad_u64_u32 %1, %8, %15, %18, %1\n\t"                                                                                          \
            "v_mad_u64_u32 %2, %9, %19, %22, %2\n\t"                                                                                          \
            "v_mad_u64_u32 %3, %10, %23, %26, %3\n\t"                                                                                         \
            "v_cndmask_b32 %4, 0, 1, vcc\n\t"                                                                                                 \
            "v_cndmask_b32 %5, 0, 1, %8\n\t"                                                                                                  \
            "v_cndmask_b32 %6, 0, 1, %9\n\t"                                                                                                  \
            "v_cndmask_b32 %7, 0, 1, %10"                                                                                                     \
            : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&s"(sb), "=&s"(sc), "=&s"(sd)         \
            : "v"(lo32(y[0])), "v"(hi32(y[0])), TWC(lo32(w[0].quo)), TWC(hi32(w[0].quo)), "v"(lo32(y[1])), "v"(hi32(y[1])), TWC(lo32(w[1].quo)), \
              TWC(hi32(w[1].quo)), "v"(lo32(y[2])), "v"(hi32(y[2])), TWC(lo32(w[2].quo)), TWC(hi32(w[2].quo)), "v"(lo32(y[3])), "v"(hi32(y[3])), \
              TWC(lo32(w[3].quo)), TWC(hi32(w[3].quo))                                                                                        \
            : "vcc");                                                                                                                         \
        const u64 s[4] = {s0, s1, s2, s3};                                                                                                    \
        const u32 c[4] = {c0, c1, c2, c3};                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                                                       \
            const u64 hs = mk64(hi32(s[i]), c[i]);                                                                                            \
            u64 d, sink;                                                                                                                      \
            asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(sink) : "v"(hi32(y[i])), TWC(hi32(w[i].quo)), "v"(hs));                    \
            q[i] = d;                                                                                                                         \
        }                                                                                                                                     \
    }
TROY_DEF_MULHI_APPROX4(mulhi_approx4, "v")
TROY_DEF_MULHI_APPROX4(mulhi_approx4_u, "s")
// q[i] = floor(y*wq/2^64) EXACTLY: the high word of the low x low product rides in on the first multiply-add's 64-bit addend -- one v_mul_hi_u32 more
// per value than the approximate form.  With the exact Shoup quotient the lazy product w*y - q*p lies in [0,2p) instead of [0,3p): worth it where the
// result is reduced to the canonical residue right away (the last stage of an inverse transform: one conditional subtraction instead of two).
#define TROY_DEF_MULHI_EXACT4(NAME, TWC)                                                                                                      \
    __device__ __forceinline__ void NAME(u64 (&q)[4], const u64 (&y)[4], const Shoup (&w)[4]) {                                               \
        u64 s0, s1, s2, s3, sb, sc, sd;                                                                                                       \
        u32 c0, c1, c2, c3;                                                                                                                   \
        u64 l[4];                                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                                                       \
            u32 h;                                                                                                                            \
            asm("v_mul_hi_u32 %0, %1, %2" : "=v"(h) : "v"(lo32(y[i])), TWC(lo32(w[i].quo)));                                                  \
            l[i] = mk64(h, 0);                                                                                                                \
        }                                                                                                                                     \
        asm("v_mad_u64_u32 %0, vcc, %12, %13, %27\n\t"                                                                                        \
            "v_mad_u64_u32 %1, %8, %16, %17, %28\n\t"                                                                                         \
            "v_mad_u64_u32 %2, %9, %20, %21, %29\n\t"                                                                                         \
            "v_mad_u64_u32 %3, %10, %24, %25, %30\n\t"                                                                                        \
            "v_mad_u64_u32 %0, vcc, %11, %14, %0\n\t"                                                                                         \
            "v_mad_u64_u32 %1, %8, %15, %18, %1\n\t"                                                                                          \
            "v_mad_u64_u32 %2, %9, %19, %22, %2\n\t"                                                                                          \
            "v_mad_u64_u32 %3, %10, %23, %26, %3\n\t"                                                                                         \
            "v_cndmask_b32 %4, 0, 1, vcc\n\t"                                                                                                 \
            "v_cndmask_b32 %5, 0, 1, %8\n\t"                                                                                                  \
            "v_cndmask_b32 %6, 0, 1, %9\n\t"                                                                                                  \
            "v_cndmask_b32 %7, 0, 1, %10"                                                                                                     \
            : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&s"(sb), "=&s"(sc), "=&s"(sd)         \
            : "v"(lo32(y[0])), "v"(hi32(y[0])), TWC(lo32(w[0].quo)), TWC(hi32(w[0].quo)), "v"(lo32(y[1])), "v"(hi32(y[1])), TWC(lo32(w[1].quo)), \
              TWC(hi32(w[1].quo)), "v"(lo32(y[2])), "v"(hi32(y[2])), TWC(lo32(w[2].quo)), TWC(hi32(w[2].quo)), "v"(lo32(y[3])), "v"(hi32(y[3])), \
              TWC(lo32(w[3].quo)), TWC(hi32(w[3].quo)), "v"(l[0]), "v"(l[1]), "v"(l[2]), "v"(l[3])                                            \
            : "vcc");                                                                                                                         \
        const u64 s[4] = {s0, s1, s2, s3};                                                                                                    \
        const u32 c[4] = {c0, c1, c2, c3};                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                                                       \
            const u64 hs = mk64(hi32(s[i]), c[i]);                                                                                            \
            u64 d, sink;                                                                                                                      \
            asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(sink) : "v"(hi32(y[i])), TWC(hi32(w[i].quo)), "v"(hs));                    \
            q[i] = d;                                                                                                                         \
        }                                                                                                                                     \
    }
TROY_DEF_MULHI_EXACT4(mulhi_exact4, "v")
TROY_DEF_MULHI_EXACT4(mulhi_exact4_u, "s")
// acc + w*y + q*negp (mod 2^64); no carry chains -> no hazards, one butterfly per block
#define TROY_DEF_MUL_ACC(NAME, TWC)                                                                                                           \
    __device__ __forceinline__ u64 NAME(u64 acc, u64 y, u64 w, u64 q, u64 negp) {                                                             \
        const u32 y0 = lo32(y), y1 = hi32(y), w0 = lo32(w), w1 = hi32(w), d0 = lo32(q), d1 = hi32(q);                                         \
        u64 a, r, sc0, sc1;                                                                                                                   \
        u32 m0, m1, m2, m3, c0, h;                                                                                                            \
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(a), "=s"(sc0) : TWC(w0), "v"(y0), "v"(acc));                                            \
        asm("v_mul_lo_u32 %0, %1, %2" : "=v"(m0) : TWC(w0), "v"(y1));                                                                         \
        asm("v_mul_lo_u32 %0, %1, %2" : "=v"(m1) : TWC(w1), "v"(y0));                                                                         \
        asm("v_mul_lo_u32 %0, %1, %2" : "=v"(m2) : "v"(d0), "s"(hi32(negp)));                                                                 \
        asm("v_mul_lo_u32 %0, %1, %2" : "=v"(m3) : "v"(d1), "s"(lo32(negp)));                                                                 \
        asm("v_add3_u32 %0, %1, %2, %3" : "=v"(c0) : "v"(m0), "v"(m1), "v"(m2));                                                              \
        asm("v_add3_u32 %0, %1, %2, %3" : "=v"(h) : "v"(hi32(a)), "v"(c0), "v"(m3));                                                          \
        const u64 a2 = mk64(lo32(a), h);                                                                                                      \
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(sc1) : "v"(d0), "s"(lo32(negp)), "v"(a2));                                     \
        return r;                                                                                                                             \
    }
TROY_DEF_MUL_ACC(mul_acc, "v")
TROY_DEF_MUL_ACC(mul_acc_u, "s")
// the same with acc = 0 (every inverse butterfly): the zero is the instruction's inline constant, not a register pair the compiler has to keep or re-make
#define TROY_DEF_MUL_ACC0(NAME, TWC)                                                                                                          \
    __device__ __forceinline__ u64 NAME(u64 y, u64 w, u64 q, u64 negp) {                                                                      \
        const u32 y0 = lo32(y), y1 = hi32(y), w0 = lo32(w), w1 = hi32(w), d0 = lo32(q), d1 = hi32(q);                                         \
        u64 a, r, sc0, sc1;                                                                                                                   \
        u32 m0, m1, m2, m3, c0, h;                                                                                                            \
        asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(a), "=s"(sc0) : TWC(w0), "v"(y0));                                                       \
        asm("v_mul_lo_u32 %0, %1, %2" : "=v"(m0) : TWC(w0), "v"(y1));                                                                         \
        asm("v_mul_lo_u32 %0, %1, %2" : "=v"(m1) : TWC(w1), "v"(y0));                                                                         \
        asm("v_mul_lo_u32 %0, %1, %2" : "=v"(m2) : "v"(d0), "s"(hi32(negp)));                                                                 \
        asm("v_mul_lo_u32 %0, %1, %2" : "=v"(m3) : "v"(d1), "s"(lo32(negp)));                                                                 \
        asm("v_add3_u32 %0, %1, %2, %3" : "=v"(c0) : "v"(m0), "v"(m1), "v"(m2));                                                              \
        asm("v_add3_u32 %0, %1, %2, %3" : "=v"(h) : "v"(hi32(a)), "v"(c0), "v"(m3));                                                          \
        const u64 a2 = mk64(lo32(a), h);                                                                                                      \
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(sc1) : "v"(d0), "s"(lo32(negp)), "v"(a2));                                     \
        return r;                                                                                                                             \
    }
TROY_DEF_MUL_ACC0(mul_acc0, "v")
TROY_DEF_MUL_ACC0(mul_acc0_u, "s")
#endif

// Four forward butterflies (Cooley-Tukey, src/utils/dwthandler.h:88-204): X,Y in [0,8p) -> [0,8p)
// UNI (all four forms below): the twiddles are wave-uniform and sit in SGPRs
template <bool UNI = false> __device__ __forceinline__ void ct_bfly4(u64 (&X)[4], u64 (&Y)[4], const Shoup (&w)[4], const PrimeConst &c) {
    csub4(X, c.four_p);                                       // u in [0,4p)
    u64 q[4], xn[4], t[4];
    if (UNI) mulhi_approx4_u(q, Y, w); else mulhi_approx4(q, Y, w);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        xn[i] = UNI ? mul_acc_u(X[i], Y[i], w[i].op, q[i], c.negp) : mul_acc(X[i], Y[i], w[i].op, q[i], c.negp);   // u + v, v in [0,3p) -> [0,7p)
        t[i] = (X[i] << 1) + c.three_p;
    }
    sub4(Y, t, xn);                                           // u + 3p - v -> (0,7p)
#pragma unroll
    for (int i = 0; i < 4; i++) X[i] = xn[i];
}
// Four inverse butterflies (Gentleman-Sande, dwthandler.h:215-372): X,Y in [0,4p) -> [0,4p)
template <bool UNI = false> __device__ __forceinline__ void gs_bfly4(u64 (&X)[4], u64 (&Y)[4], const Shoup (&w)[4], const PrimeConst &c) {
    u64 s[4], t[4], d[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { s[i] = X[i] + Y[i]; t[i] = X[i] + c.four_p; }
    sub4(d, t, Y);                                            // u + 4p - v in (0,8p)
    csub4(s, c.four_p);
    if (UNI) mulhi_approx4_u(q, d, w); else mulhi_approx4(q, d, w);
#pragma unroll
    for (int i = 0; i < 4; i++) { X[i] = s[i]; Y[i] = UNI ? mul_acc0_u(d[i], w[i].op, q[i], c.negp) : mul_acc0(d[i], w[i].op, q[i], c.negp); }
}
// last inverse stage with N^-1 folded in (dwthandler.h:289-330): -> [0,3p)
template <bool UNI = false> __device__ __forceinline__ void gs_bfly4_last(u64 (&X)[4], u64 (&Y)[4], const Shoup (&w_scaled)[4], const Shoup inv_n, const PrimeConst &c) {
    u64 s[4], t[4], d[4], q[4];
    const Shoup wn[4] = {inv_n, inv_n, inv_n, inv_n};
#pragma unroll
    for (int i = 0; i < 4; i++) { s[i] = X[i] + Y[i]; t[i] = X[i] + c.four_p; }
    sub4(d, t, Y);
    csub4(s, c.four_p);
    if (UNI) mulhi_approx4_u(q, s, wn); else mulhi_approx4(q, s, wn);
#pragma unroll
    for (int i = 0; i < 4; i++) X[i] = UNI ? mul_acc0_u(s[i], inv_n.op, q[i], c.negp) : mul_acc0(s[i], inv_n.op, q[i], c.negp);
    if (UNI) mulhi_approx4_u(q, d, w_scaled); else mulhi_approx4(q, d, w_scaled);
#pragma unroll
    for (int i = 0; i < 4; i++) Y[i] = UNI ? mul_acc0_u(d[i], w_scaled[i].op, q[i], c.negp) : mul_acc0(d[i], w_scaled[i].op, q[i], c.negp);
}
// ---- guard-free ("lean") butterflies: no conditional subtraction at all; the CALLER tracks the value bound (in units of p) per
// stage and inserts a reduction (barrett_lite4) only where the next stage could leave 64 bits.  For the primes the
// reference's parameter generator produces below 2^58 (CoeffModulus::Create with <= 58-bit sizes) a whole 15-stage forward
// transform needs none: every stage adds at most 3p to the bound (v = w*y mod p lazily in [0,3p)).
// forward: X' = X + v, Y' = X + 3p - v; both outputs < bound(X) + 3p.  15 VALU instructions.
template <bool UNI = false> __device__ __forceinline__ void ct_bfly4_ng(u64 (&X)[4], u64 (&Y)[4], const Shoup (&w)[4], const PrimeConst &c) {
    u64 q[4], xn[4], t[4];
    if (UNI) mulhi_approx4_u(q, Y, w); else mulhi_approx4(q, Y, w);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        xn[i] = UNI ? mul_acc_u(X[i], Y[i], w[i].op, q[i], c.negp) : mul_acc(X[i], Y[i], w[i].op, q[i], c.negp);
        t[i] = (X[i] << 1) + c.three_p;
    }
    sub4(Y, t, xn);
#pragma unroll
    for (int i = 0; i < 4; i++) X[i] = xn[i];
}
// inverse: X' = X + Y, Y' = (X + kp - Y) * w lazily in [0,3p); kp = a multiple of p that is >= bound(Y); the caller keeps
// bound(X) + kp < 2^64.  16 VALU instructions.
template <bool UNI = false> __device__ __forceinline__ void gs_bfly4_ng(u64 (&X)[4], u64 (&Y)[4], const Shoup (&w)[4], u64 kp, const PrimeConst &c) {
    u64 t[4], d[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = X[i] + kp;
    sub4(d, t, Y);
#pragma unroll
    for (int i = 0; i < 4; i++) X[i] = X[i] + Y[i];
    if (UNI) mulhi_approx4_u(q, d, w); else mulhi_approx4(q, d, w);
#pragma unroll
    for (int i = 0; i < 4; i++) Y[i] = UNI ? mul_acc0_u(d[i], w[i].op, q[i], c.negp) : mul_acc0(d[i], w[i].op, q[i], c.negp);
}
// last inverse stage, guard-free: bound(X), bound(Y) <= kp / p and bound(X) + bound(Y) < 2^64 / p; both outputs come out of a
// multiplication (by N^-1, by the pre-scaled twiddle), i.e. in [0,3p) whatever the inputs were
template <bool UNI = false> __device__ __forceinline__ void gs_bfly4_last_ng(u64 (&X)[4], u64 (&Y)[4], const Shoup (&w_scaled)[4], const Shoup inv_n, u64 kp, const PrimeConst &c) {
    u64 s[4], t[4], d[4], q[4];
    const Shoup wn[4] = {inv_n, inv_n, inv_n, inv_n};
#pragma unroll
    for (int i = 0; i < 4; i++) { s[i] = X[i] + Y[i]; t[i] = X[i] + kp; }
    sub4(d, t, Y);
    if (UNI) mulhi_approx4_u(q, s, wn); else mulhi_approx4(q, s, wn);
#pragma unroll
    for (int i = 0; i < 4; i++) X[i] = UNI ? mul_acc0_u(s[i], inv_n.op, q[i], c.negp) : mul_acc0(s[i], inv_n.op, q[i], c.negp);
    if (UNI) mulhi_approx4_u(q, d, w_scaled); else mulhi_approx4(q, d, w_scaled);
#pragma unroll
    for (int i = 0; i < 4; i++) Y[i] = UNI ? mul_acc0_u(d[i], w_scaled[i].op, q[i], c.negp) : mul_acc0(d[i], w_scaled[i].op, q[i], c.negp);
}
// the same two with one kp PER butterfly (the caller tracks a bound per register: inv_stages_lean, ntt1.hip).  EXACT: the two multiplications of the
// last stage use the exact quotient, outputs in [0,2p)
template <bool UNI = false> __device__ __forceinline__ void gs_bfly4_ng_k(u64 (&X)[4], u64 (&Y)[4], const Shoup (&w)[4], const u64 (&kp)[4], const PrimeConst &c) {
    u64 t[4], d[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = X[i] + kp[i];
    sub4(d, t, Y);
#pragma unroll
    for (int i = 0; i < 4; i++) X[i] = X[i] + Y[i];
    if (UNI) mulhi_approx4_u(q, d, w); else mulhi_approx4(q, d, w);
#pragma unroll
    for (int i = 0; i < 4; i++) Y[i] = UNI ? mul_acc0_u(d[i], w[i].op, q[i], c.negp) : mul_acc0(d[i], w[i].op, q[i], c.negp);
}
template <bool UNI = false, bool EXACT = false> __device__ __forceinline__ void gs_bfly4_last_ng_k(u64 (&X)[4], u64 (&Y)[4], const Shoup (&w_scaled)[4], const Shoup inv_n, const u64 (&kp)[4], const PrimeConst &c) {
    u64 s[4], t[4], d[4], q[4];
    const Shoup wn[4] = {inv_n, inv_n, inv_n, inv_n};
#pragma unroll
    for (int i = 0; i < 4; i++) { s[i] = X[i] + Y[i]; t[i] = X[i] + kp[i]; }
    sub4(d, t, Y);
    if (EXACT) { if (UNI) mulhi_exact4_u(q, s, wn); else mulhi_exact4(q, s, wn); }
    else { if (UNI) mulhi_approx4_u(q, s, wn); else mulhi_approx4(q, s, wn); }
#pragma unroll
    for (int i = 0; i < 4; i++) X[i] = UNI ? mul_acc0_u(s[i], inv_n.op, q[i], c.negp) : mul_acc0(s[i], inv_n.op, q[i], c.negp);
    if (EXACT) { if (UNI) mulhi_exact4_u(q, d, w_scaled); else mulhi_exact4(q, d, w_scaled); }
    else { if (UNI) mulhi_approx4_u(q, d, w_scaled); else mulhi_approx4(q, d, w_scaled); }
#pragma unroll
    for (int i = 0; i < 4; i++) Y[i] = UNI ? mul_acc0_u(d[i], w_scaled[i].op, q[i], c.negp) : mul_acc0(d[i], w_scaled[i].op, q[i], c.negp);
}
// any x < 2^64 -> the same residue in [0, 4p), for p >= 2^33: q = floor(hi32(x) * floor(2^64/p) / 2^32) is at most 2.5 below x / p
// (never above).  4 instructions per value -- the price of dropping the range guard from several inverse stages in a row.
__device__ __forceinline__ void lite_reduce4(u64 (&x)[4], u32 mu, const PrimeConst &c) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u32 q = (u32)(((u64)hi32(x[i]) * mu) >> 32);
        x[i] += (u64)q * c.negp;
    }
}
__device__ __forceinline__ void lite_reduce1(u64 &x, u32 mu, const PrimeConst &c) {
    const u32 q = (u32)(((u64)hi32(x) * mu) >> 32);
    x += (u64)q * c.negp;
}
// end of a guard-free forward transform: x < 64p (46p after 15 stages, 59p in the two-pass form), p in [2^33, 2^58) -> canonical.
// q = floor((x >> sh) * mu / 2^32) with sh = bitlen(p) - 26, mu = floor(2^(sh+32) / p) = floor(2^64 / p) >> (58 - bitlen(p)) is at most
// one below floor(x / p) (x >> sh < 2^32, so the truncation of mu costs less than 1), never above: one multiply-high, one 64-bit
// multiply-subtract and one conditional subtraction -- half the instructions of the two-word Barrett step (barrett64)
struct LeanFinal { unsigned sh; u32 mu; };
__device__ __forceinline__ LeanFinal make_lean_final(u64 p, u64 cr1) {
    const int b = 64 - __builtin_clzll(p);
    return LeanFinal{(unsigned)(b - 26), (u32)(cr1 >> (58 - b))};
}
__device__ __forceinline__ void lean_final4(u64 (&x)[4], const LeanFinal f, const PrimeConst &c) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u32 q = (u32)(((u64)(u32)(x[i] >> f.sh) * f.mu) >> 32);
        x[i] += (u64)q * c.negp;
    }
    csub4(x, c.p);
}
template <int NV> __device__ __forceinline__ void lean_final(u64 (&x)[NV], const LeanFinal f, const PrimeConst &c) {
#pragma unroll
    for (int h = 0; h < NV / 4; h++) {
        u64 v[4] = {x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3]};
        lean_final4(v, f, c);
#pragma unroll
        for (int i = 0; i < 4; i++) x[4 * h + i] = v[i];
    }
}
// final normalisations to the canonical residue, four values at a time
__device__ __forceinline__ void reduce4_from_8p(u64 (&x)[4], const PrimeConst &c) { csub4(x, c.four_p); csub4(x, c.two_p); csub4(x, c.p); }
__device__ __forceinline__ void reduce4_from_4p(u64 (&x)[4], const PrimeConst &c) { csub4(x, c.two_p); csub4(x, c.p); }


// ---- 128-bit multiply-accumulate, four independent accumulators: acc[i] += x[i] * k[i]  (acc < 2^128 by the caller's bound)
// 10 instructions per term (4 v_mad_u64_u32 + 6 carry adds); every carry is consumed by the next step of the same lane
// group, and the four groups are interleaved so that a carry writer and its reader are 4 instructions apart.
struct Acc128 { u32 a0, a1, a2, a3; };
#ifdef TROYHIP_CPU_EMUL
__device__ __forceinline__ void mac128x4(Acc128 (&acc)[4], const u64 (&x)[4], const u64 (&k)[4]) {
    for (int i = 0; i < 4; i++) {
        unsigned __int128 v = ((unsigned __int128)acc[i].a3 << 96) | ((unsigned __int128)acc[i].a2 << 64) | ((unsigned __int128)acc[i].a1 << 32) | acc[i].a0;
        v += (unsigned __int128)x[i] * k[i];
        acc[i].a0 = (u32)v; acc[i].a1 = (u32)(v >> 32); acc[i].a2 = (u32)(v >> 64); acc[i].a3 = (u32)(v >> 96);
    }
}
#else
__device__ __forceinline__ void mac128x4(Acc128 (&acc)[4], const u64 (&x)[4], const u64 (&k)[4]) {
    u64 lo[4], hi[4], t[4];
    u64 sb, sc, sd;
#pragma unroll
    for (int i = 0; i < 4; i++) { lo[i] = mk64(acc[i].a0, acc[i].a1); hi[i] = mk64(acc[i].a2, acc[i].a3); }
    u32 a2[4], a3[4], a1[4], a0[4];
    // S1: [a0:a1] += xl*kl -> carry ; S2: a2 += carry -> carry ; S3: a3 += carry
    asm("v_mad_u64_u32 %0, vcc, %15, %19, %0\n\t"
        "v_mad_u64_u32 %1, %12, %16, %20, %1\n\t"
        "v_mad_u64_u32 %2, %13, %17, %21, %2\n\t"
        "v_mad_u64_u32 %3, %14, %18, %22, %3\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32 %5, %12, 0, %5, %12\n\t"
        "v_addc_co_u32 %6, %13, 0, %6, %13\n\t"
        "v_addc_co_u32 %7, %14, 0, %7, %14\n\t"
        "v_addc_co_u32 %8, vcc, 0, %8, vcc\n\t"
        "v_addc_co_u32 %9, %12, 0, %9, %12\n\t"
        "v_addc_co_u32 %10, %13, 0, %10, %13\n\t"
        "v_addc_co_u32 %11, %14, 0, %11, %14"
        : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(acc[0].a2), "+v"(acc[1].a2), "+v"(acc[2].a2), "+v"(acc[3].a2), "+v"(acc[0].a3), "+v"(acc[1].a3),
          "+v"(acc[2].a3), "+v"(acc[3].a3), "=&s"(sb), "=&s"(sc), "=&s"(sd)
        : "v"(lo32(x[0])), "v"(lo32(x[1])), "v"(lo32(x[2])), "v"(lo32(x[3])), "v"(lo32(k[0])), "v"(lo32(k[1])), "v"(lo32(k[2])), "v"(lo32(k[3]))
        : "vcc");
    // S4: t = xl*kh ; S5: t += xh*kl -> carry (weight 2^96) ; S6: a3 += carry
    asm("v_mad_u64_u32 %0, vcc, %11, %19, 0\n\t"
        "v_mad_u64_u32 %1, %8, %12, %20, 0\n\t"
        "v_mad_u64_u32 %2, %9, %13, %21, 0\n\t"
        "v_mad_u64_u32 %3, %10, %14, %22, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %15, %23, %0\n\t"
        "v_mad_u64_u32 %1, %8, %16, %24, %1\n\t"
        "v_mad_u64_u32 %2, %9, %17, %25, %2\n\t"
        "v_mad_u64_u32 %3, %10, %18, %26, %3\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32 %5, %8, 0, %5, %8\n\t"
        "v_addc_co_u32 %6, %9, 0, %6, %9\n\t"
        "v_addc_co_u32 %7, %10, 0, %7, %10"
        : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "+v"(acc[0].a3), "+v"(acc[1].a3), "+v"(acc[2].a3), "+v"(acc[3].a3), "=&s"(sb), "=&s"(sc), "=&s"(sd)
        : "v"(lo32(x[0])), "v"(lo32(x[1])), "v"(lo32(x[2])), "v"(lo32(x[3])), "v"(hi32(x[0])), "v"(hi32(x[1])), "v"(hi32(x[2])), "v"(hi32(x[3])), "v"(hi32(k[0])),
          "v"(hi32(k[1])), "v"(hi32(k[2])), "v"(hi32(k[3])), "v"(lo32(k[0])), "v"(lo32(k[1])), "v"(lo32(k[2])), "v"(lo32(k[3]))
        : "vcc");
#pragma unroll
    for (int i = 0; i < 4; i++) { a0[i] = lo32(lo[i]); a1[i] = hi32(lo[i]); }
    // S7: a1 += t.lo -> carry ; S8: a2 += t.hi + carry -> carry ; S9: a3 += carry
    asm("v_add_co_u32 %0, vcc, %0, %15\n\t"
        "v_add_co_u32 %1, %12, %1, %16\n\t"
        "v_add_co_u32 %2, %13, %2, %17\n\t"
        "v_add_co_u32 %3, %14, %3, %18\n\t"
        "v_addc_co_u32 %4, vcc, %4, %19, vcc\n\t"
        "v_addc_co_u32 %5, %12, %5, %20, %12\n\t"
        "v_addc_co_u32 %6, %13, %6, %21, %13\n\t"
        "v_addc_co_u32 %7, %14, %7, %22, %14\n\t"
        "v_addc_co_u32 %8, vcc, 0, %8, vcc\n\t"
        "v_addc_co_u32 %9, %12, 0, %9, %12\n\t"
        "v_addc_co_u32 %10, %13, 0, %10, %13\n\t"
        "v_addc_co_u32 %11, %14, 0, %11, %14"
        : "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]), "+v"(acc[0].a2), "+v"(acc[1].a2), "+v"(acc[2].a2), "+v"(acc[3].a2), "+v"(acc[0].a3), "+v"(acc[1].a3),
          "+v"(acc[2].a3), "+v"(acc[3].a3), "=&s"(sb), "=&s"(sc), "=&s"(sd)
        : "v"(lo32(t[0])), "v"(lo32(t[1])), "v"(lo32(t[2])), "v"(lo32(t[3])), "v"(hi32(t[0])), "v"(hi32(t[1])), "v"(hi32(t[2])), "v"(hi32(t[3]))
        : "vcc");
    // S10: [a2:a3] += xh*kh
#pragma unroll
    for (int i = 0; i < 4; i++) hi[i] = mk64(acc[i].a2, acc[i].a3);
    asm("v_mad_u64_u32 %0, vcc, %7, %11, %0\n\t"
        "v_mad_u64_u32 %1, %4, %8, %12, %1\n\t"
        "v_mad_u64_u32 %2, %5, %9, %13, %2\n\t"
        "v_mad_u64_u32 %3, %6, %10, %14, %3"
        : "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "=&s"(sb), "=&s"(sc), "=&s"(sd)
        : "v"(hi32(x[0])), "v"(hi32(x[1])), "v"(hi32(x[2])), "v"(hi32(x[3])), "v"(hi32(k[0])), "v"(hi32(k[1])), "v"(hi32(k[2])), "v"(hi32(k[3]))
        : "vcc");
#pragma unroll
    for (int i = 0; i < 4; i++) { acc[i].a0 = a0[i]; acc[i].a1 = a1[i]; acc[i].a2 = lo32(hi[i]); acc[i].a3 = hi32(hi[i]); }
}
#endif

} // namespace troyhip
