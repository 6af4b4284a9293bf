// fpmod.h -- exact modular arithmetic in DOUBLE PRECISION for primes below 2^50 ("narrow" primes: every 40- / 50-bit prime of a CKKS or
// BGV chain).  gfx950 has no 64-bit integer multiplier -- the integer guard-free butterfly of bfly.h costs 15 VALU instructions, 9 of
// them 32-bit multiplies -- but it has full-rate FP64 FMA.  With values kept as EXACT INTEGERS in doubles (signed, lazy range) a modular
// product is six instructions and a butterfly eight (profiles/r03_microbench_fp64.txt: 4.1 T butterflies/s against 2.35 T):
//     h = y * w                 rounded product
//     l = fma(y, w, -h)         its rounding error: h + l = y w exactly (error-free product)
//     q = rint(y * (w / p))     quotient estimate, |q - y w / p| <= 1/2 + |y| 2^-52
//     r = fma(-q, p, h) + l     = y w - q p exactly: both steps are exact because the true values are integers below 2^53
// so r = y w (mod p) with |r| <= (1/2 + |y| 2^-52) p -- whatever the roundings inside were, r is an exact integer congruent to y w, and
// every stored limb is reduced to the canonical residue in [0, p): results are bit-identical to the integer path by construction
// (the reference computes the same residues with Harvey butterflies, src/utils/dwthandler.h:88-372).
// Requirements the callers keep (ntt2.hip tracks the bound per round on the host, fp_plan): |y| < 2^52, every sum below 2^53.
#pragma once
#include "modarith.h"
#include <cmath>

namespace troyhip {

#define TROY_FP_MAX_BITS 50 // primes strictly below 2^50 take the FP64 path

struct FpPrime { double p, pinv; };
TROY_HD FpPrime make_fp_prime(u64 p) { return FpPrime{(double)p, 1.0 / (double)p}; }

TROY_HD double fp_of_bits(u64 b) { return __builtin_bit_cast(double, b); }
TROY_HD u64 fp_bits(double d) { return __builtin_bit_cast(u64, d); }
// exact for x < 2^52: the integer is the mantissa of 2^52 + x (one OR on the high word, one subtraction)
TROY_HD double fp_from_u64(u64 x) { return fp_of_bits(x | 0x4330000000000000ull) - 4503599627370496.0; }
// d = an integer in [0, 2^52)
TROY_HD u64 fp_to_u64(double d) { return fp_bits(d + 4503599627370496.0) & 0x000fffffffffffffull; }

#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__ && !defined(TROYHIP_CPU_EMUL)
// the prime is wave-uniform: keep (p, 1 / p) in scalar registers (the conversion and the division run on the vector unit, whose results the
// compiler would otherwise hold in four VGPRs for the whole kernel)
__device__ __forceinline__ FpPrime make_fp_prime_uniform(u64 p) {
    const FpPrime v = make_fp_prime(p);
    auto uni = [](double d) {
        const u64 b = __builtin_bit_cast(u64, d);
        const u64 r = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)b);
        return __builtin_bit_cast(double, r);
    };
    return FpPrime{uni(v.p), uni(v.pinv)};
}
#else
TROY_HD FpPrime make_fp_prime_uniform(u64 p) { return make_fp_prime(p); }
#endif

// y * w mod p for a table constant w with wp = w / p (correctly rounded): |result| <= (1/2 + |y| 2^-52) p
TROY_HD double fp_mulmod_wp(double y, double w, double wp, const FpPrime &c) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
    const double h = y * w;
    const double l = __builtin_fma(y, w, -h);
    const double q = __builtin_rint(y * wp);
    const double r = __builtin_fma(-q, c.p, h);
    return r + l;
}
// y * k mod p for a run-time operand k in [0, p) (the quotient comes from the rounded product): |result| <= (1/2 + 3 |y| 2^-53) p
TROY_HD double fp_mulmod_pinv(double y, double k, const FpPrime &c) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
    const double h = y * k;
    const double l = __builtin_fma(y, k, -h);
    const double q = __builtin_rint(h * c.pinv);
    const double r = __builtin_fma(-q, c.p, h);
    return r + l;
}
// any integer |x| < 2^53 -> the congruent integer with |result| <= p / 2 + 1
TROY_HD double fp_reduce(double x, const FpPrime &c) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
    const double q = __builtin_rint(x * c.pinv);
    return __builtin_fma(-q, c.p, x);
}
// ... -> the canonical residue in [0, p) as an integer word.  r = fp_reduce(x) is an exact integer with |r| <= p / 2 + |x| 2^-52 (the
// quotient estimate is off by at most 1/2 + (|x| / p) 2^-52), i.e. |r| < p / 2 + 2 for |x| < 2^53: a non-negative r is already below p
// and a negative one lands in [p / 2 - 2, p) after ONE addition of p -- no second range check.  The addition is an fma with a 0 / 1
// multiplier (one select on its high word) onto r + 2^52, whose mantissa is the answer: 8 instructions where the select-add-convert-
// compare-subtract form took 15
TROY_HD u64 fp_canonical(double x, const FpPrime &c, u64 p) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
    (void)p;
    const double r = fp_reduce(x, c);
    const double neg = fp_of_bits((u64)(r < 0.0 ? 0x3FF00000u : 0u) << 32); // 1.0 or 0.0
    const double t = __builtin_fma(neg, c.p, r + 4503599627370496.0);        // 2^52 + r (+ p): both steps exact, result in [2^52, 2^52 + p)
    return fp_bits(t) & 0x000fffffffffffffull;
}

// ---- host side: when to reduce.  Bounds are in units of p; `lim` = 2^53 / p.  A butterfly stage maps a bound b to
// b + 1/2 + b p 2^-52 (X + v; v from fp_mulmod_wp with |y| <= b p).  fp_plan walks the rounds of a pass (stages per round given) and
// sets bit r of the mask when the values must be reduced (to 1/2 + 2^-40) before round r so that no value reaches `lim`.
struct FpPlan { unsigned mask; double out_bound; };
// c = 1: twiddle pairs (w, w / p), fp_mulmod_wp;  c = 1.5: single-double twiddles, fp_mulmod_pinv (three roundings in the quotient estimate)
inline double fp_stage_bound(double b, double p, double c = 1.0) { return b + 0.5 + c * b * p * 0x1p-52; }
inline FpPlan fp_plan(u64 pmax, double b_in, const int *rounds, int n_rounds, double c = 1.0) {
    const double p = (double)pmax, lim = 0x1p53 / p * 0.98; // 2 % slack for the "+1" terms
    FpPlan out{0, b_in};
    double b = b_in;
    for (int r = 0; r < n_rounds; r++) {
        double t = b;
        for (int s = 0; s < rounds[r]; s++) t = fp_stage_bound(t, p, c);
        if (t >= lim) {
            out.mask |= 1u << r;
            t = 0.5 + 0x1p-40;
            for (int s = 0; s < rounds[r]; s++) t = fp_stage_bound(t, p, c);
            if (t >= lim) { out.mask = ~0u; out.out_bound = -1.0; return out; } // even a freshly reduced round passes the limit: no FP64 schedule (as fp_plan_inv)
        }
        b = t;
    }
    out.out_bound = b;
    return out;
}

// inverse (Gentleman-Sande) stages: X' = X + Y doubles the bound per stage (Y' comes out of a multiplication, below 1/2 + 2 b p 2^-52 <= 5/2);
// a stage's outputs must stay within 2^53, i.e. a round of R stages needs 2^R b <= lim.  Reduction sites: before each round.
inline FpPlan fp_plan_inv(u64 pmax, double b_in, const int *rounds, int n_rounds) {
    const double p = (double)pmax, lim = 0x1p53 / p;
    FpPlan out{0, b_in};
    double b = b_in;
    for (int r = 0; r < n_rounds; r++) {
        double t = std::ldexp(b, rounds[r]);
        if (t > lim) {
            out.mask |= 1u << r;
            t = std::ldexp(0.5 + 0x1p-40, rounds[r]);
            if (t > lim) { out.mask = ~0u; out.out_bound = -1.0; return out; } // cannot be scheduled at round granularity: the caller keeps the integer kernels
        }
        b = t < 2.5 ? 2.5 : t;
    }
    out.out_bound = b;
    return out;
}

} // namespace troyhip
