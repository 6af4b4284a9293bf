// modarith.h -- 64-bit modular arithmetic shared by host precompute and gfx950 kernels.
//
// Replaces the scalar device inlines of the reference (src/kernelutils.cuh:77-421: dBarrettReduce64,
// dBarrettReduce128, dMultiplyUintMod, dMultiplyUintModLazy, dAddUintMod, dSubUintMod, ...).  All
// stored results are canonical residues, so any exact modular arithmetic is bit-identical to the
// reference; we are free to choose the reduction schedule.  gfx950 has no 64x64 multiplier: every
// 64-bit product below lowers to v_mad_u64_u32 / v_mul_hi_u32 chains, so the number of 64-bit
// multiplies per butterfly is the quantity to minimise.
#pragma once
#include "rt.h"

namespace troyhip {

#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
#define TROY_HD __host__ __device__ __forceinline__
#else
#define TROY_HD __host__ __device__ inline
#endif

// Modulus with its Barrett constant floor(2^128/p) (reference: src/modulus.h:16-24, modulus.cpp:27-37)
struct Mod {
    u64 p;
    u64 cr0, cr1; // const_ratio lo, hi
};

struct Shoup { // MultiplyUIntModOperand (src/utils/uintarithsmallmod.h:166-186)
    u64 op, quo;
};

TROY_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
    return __umul64hi(a, b);
#else
    return (u64)(((u128)a * b) >> 64);
#endif
}

TROY_HD u64 barrett64(u64 x, const Mod &m) { // kernelutils.cuh:99-110
    u64 q = mulhi64(x, m.cr1);
    u64 r = x - q * m.p;
    return r >= m.p ? r - m.p : r;
}

// (lo,hi) mod p, result canonical (kernelutils.cuh:161-189)
TROY_HD u64 barrett128(u64 lo, u64 hi, const Mod &m) {
    u64 carry = mulhi64(lo, m.cr0);
    u64 t2lo = lo * m.cr1, t2hi = mulhi64(lo, m.cr1);
    u64 tmp1 = t2lo + carry;
    u64 tmp3 = t2hi + (tmp1 < carry);
    u64 t4lo = hi * m.cr0, t4hi = mulhi64(hi, m.cr0);
    u64 tmp1b = tmp1 + t4lo;
    u64 c2 = t4hi + (tmp1b < tmp1);
    u64 q = hi * m.cr1 + tmp3 + c2;
    u64 r = lo - q * m.p;
    return r >= m.p ? r - m.p : r;
}
// floor((hi:lo) / p) for a value < p * 2^64 (the quotient fits one word): the Barrett quotient estimate, corrected once
TROY_HD u64 div128(u64 lo, u64 hi, const Mod &m) {
    u64 carry = mulhi64(lo, m.cr0);
    u64 t2lo = lo * m.cr1, t2hi = mulhi64(lo, m.cr1);
    u64 tmp1 = t2lo + carry;
    u64 tmp3 = t2hi + (tmp1 < carry);
    u64 t4lo = hi * m.cr0, t4hi = mulhi64(hi, m.cr0);
    u64 tmp1b = tmp1 + t4lo;
    u64 c2 = t4hi + (tmp1b < tmp1);
    u64 q = hi * m.cr1 + tmp3 + c2;
    u64 r = lo - q * m.p;
    return r >= m.p ? q + 1 : q;
}
TROY_HD u64 mulmod(u64 a, u64 b, const Mod &m) { return barrett128(a * b, mulhi64(a, b), m); }
// Barrett reduction specialised for x < 2 p^2 (a product of two residues, or the sum of two): with k = bit length of p,
// q = floor(floor(x / 2^(k-1)) * floor(2^(63+k) / p) / 2^64) satisfies floor(x/p) - 2 <= q <= floor(x/p) for k <= 61, so one
// 64 x 64 high product, one low product and two conditional subtractions give the canonical residue (the generic barrett128
// above takes three high and three low products).  mu = floor(2^128 / p) >> (65 - k) is exactly floor(2^(63+k) / p).
struct ProdMod { u64 p, two_p, mu; int shift; };
TROY_HD ProdMod make_prod_mod(const Mod &m) {
    const int k = 64 - __builtin_clzll(m.p);
    const u128 cr = ((u128)m.cr1 << 64) | m.cr0;
    return ProdMod{m.p, 2 * m.p, (u64)(cr >> (65 - k)), k - 1};
}
TROY_HD u64 reduce_prod(u128 x, const ProdMod &t) {
    const u64 q = mulhi64((u64)(x >> t.shift), t.mu);
    u64 r = (u64)x - q * t.p;
    r = r >= t.two_p ? r - t.two_p : r;
    return r >= t.p ? r - t.p : r;
}
TROY_HD u64 addmod(u64 a, u64 b, u64 p) { u64 s = a + b; return s >= p ? s - p : s; }
TROY_HD u64 submod(u64 a, u64 b, u64 p) { return a >= b ? a - b : a + p - b; }
TROY_HD u64 negmod(u64 a, u64 p) { return a ? p - a : 0; }

// x * w mod p in [0, 2p) for ANY 64-bit x (kernelutils.cuh:129-136)
TROY_HD u64 mul_lazy(u64 x, u64 w, u64 wq, u64 p) { return w * x - mulhi64(x, wq) * p; }
TROY_HD u64 mul_shoup(u64 x, u64 w, u64 wq, u64 p) { u64 r = mul_lazy(x, w, wq, p); return r >= p ? r - p : r; }
TROY_HD u64 mul_lazy(u64 x, const Shoup &w, u64 p) { return mul_lazy(x, w.op, w.quo, p); }
TROY_HD u64 mul_shoup(u64 x, const Shoup &w, u64 p) { return mul_shoup(x, w.op, w.quo, p); }

// 128-bit accumulator helpers
struct U128 { u64 lo, hi; };
TROY_HD void mac128(U128 &acc, u64 a, u64 b) {
    u64 lo = a * b, hi = mulhi64(a, b);
    acc.lo += lo;
    acc.hi += hi + (acc.lo < lo);
}

// ---- host-only helpers ----
inline Mod make_mod(u64 p) {
    Mod m{p, 0, 0};
    if (p) {
        u128 hi = ((u128)1 << 64) / p;
        u128 rem = ((u128)1 << 64) - hi * p;
        m.cr1 = (u64)hi;
        m.cr0 = (u64)((rem << 64) / p);
    }
    return m;
}
inline Shoup make_shoup(u64 w, u64 p) { return Shoup{w, (u64)((((u128)w) << 64) / p)}; }

} // namespace troyhip
