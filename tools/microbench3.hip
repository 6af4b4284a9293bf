// microbench3.hip -- round-3 probes:
//   * issue cost of the FP64 instructions an exact double-precision modular butterfly needs;
//   * FP64 butterflies for primes below 2^50 (values kept as exact integers in doubles, signed lazy range) against the 15-instruction
//     integer guard-free butterfly of bfly.h, with an on-device exactness check against 128-bit integer arithmetic;
//   * key-switch inner-product forms: mac128x4 (10 instructions per term, 4 VGPRs per accumulator), the carry-counter form
//     (7 instructions, 8 VGPRs per accumulator), the FP64 form for narrow primes (7 instructions, 2 VGPRs per accumulator).
// Build: hipcc --offload-arch=gfx950 -O3 -I troy_amd/csrc tools/microbench3.hip -o tools/microbench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include "../troy_amd/csrc/bfly.h"

using namespace troyhip;
#define ITERS 2048

struct FpPrime { double p, pinv; };
struct FpTw { double w, wp; }; // w and w / p

// y * w mod p, result in about (-0.5p - e, 0.5p + e); y, w exact integers, |y| < 2^51, 0 <= w < p < 2^50
// quotient from the rounded product and 1/p (twiddle = one double)
__device__ __forceinline__ double fp_mulmod_pinv(double y, double w, const FpPrime &c) {
#pragma clang fp contract(off)
    const double h = y * w;
    const double l = __builtin_fma(y, w, -h);
    const double q = __builtin_rint(h * c.pinv);
    const double r = __builtin_fma(-q, c.p, h);
    return r + l;
}
// quotient from y and the precomputed w / p (twiddle = two doubles, the quotient does not wait for h)
__device__ __forceinline__ double fp_mulmod_wp(double y, const FpTw &t, const FpPrime &c) {
#pragma clang fp contract(off)
    const double h = y * t.w;
    const double l = __builtin_fma(y, t.w, -h);
    const double q = __builtin_rint(y * t.wp);
    const double r = __builtin_fma(-q, c.p, h);
    return r + l;
}
__device__ __forceinline__ double fp_reduce(double x, const FpPrime &c) {
#pragma clang fp contract(off)
    const double q = __builtin_rint(x * c.pinv);
    return __builtin_fma(-q, c.p, x);
}

// VAR 0: integer guard-free CT; 1: fp CT (pinv); 2: fp CT (w/p); 3: fp CT (w/p) + a reduction of both outputs every 4th stage (50-bit primes);
// 4: fp GS (w/p); 5: integer guard-free GS
template <int VAR> __global__ __launch_bounds__(256) void bfly_probe(u64 *out, u64 p, Shoup w, FpPrime fc, FpTw ft) {
    extern __shared__ u64 dyn[];
    const u32 t = threadIdx.x + blockIdx.x * 256;
    u64 r = 0;
    if (VAR == 0 || VAR == 5) {
        u64 x[8];
        const PrimeConst pc = make_prime_const(p);
        for (int i = 0; i < 8; i++) x[i] = (t * 0x9E3779B97F4A7C15ull + i * 0xBF58476D1CE4E5B9ull) % p;
        Shoup ww[4] = {w, w, w, w};
        const u64 kp = 8 * p;
        for (int it = 0; it < ITERS / 16; it++) {
#pragma unroll
            for (int st = 0; st < 3; st++) {
                const int half = 4 >> st;
                u64 X[4], Y[4]; int n = 0, ix[4], iy[4];
#pragma unroll
                for (int blk = 0; blk < (1 << st); blk++)
#pragma unroll
                    for (int k = 0; k < half; k++) { ix[n] = blk * 2 * half + k; iy[n] = ix[n] + half; n++; }
#pragma unroll
                for (int i = 0; i < 4; i++) { X[i] = x[ix[i]]; Y[i] = x[iy[i]]; }
                if (VAR == 0) ct_bfly4_ng(X, Y, ww, pc); else gs_bfly4_ng(X, Y, ww, kp, pc);
#pragma unroll
                for (int i = 0; i < 4; i++) { x[ix[i]] = X[i]; x[iy[i]] = Y[i]; }
            }
        }
        for (int i = 0; i < 8; i++) r ^= x[i];
    } else {
        double x[8];
        for (int i = 0; i < 8; i++) x[i] = (double)((t * 0x9E3779B97F4A7C15ull + i * 0xBF58476D1CE4E5B9ull) % p);
        for (int it = 0; it < ITERS / 16; it++) {
#pragma unroll
            for (int st = 0; st < 3; st++) {
                const int half = 4 >> st;
                int n = 0, ix[4], iy[4];
#pragma unroll
                for (int blk = 0; blk < (1 << st); blk++)
#pragma unroll
                    for (int k = 0; k < half; k++) { ix[n] = blk * 2 * half + k; iy[n] = ix[n] + half; n++; }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    double X = x[ix[i]], Y = x[iy[i]];
                    if (VAR == 4) {
                        const double s = X + Y, d = X - Y;
                        X = s;
                        Y = fp_mulmod_wp(d, ft, fc);
                    } else {
                        const double v = VAR == 1 ? fp_mulmod_pinv(Y, ft.w, fc) : fp_mulmod_wp(Y, ft, fc);
                        Y = X - v;
                        X = X + v;
                    }
                    x[ix[i]] = X; x[iy[i]] = Y;
                }
            }
            if (VAR == 3 || VAR == 4 || (it & 15) == 15) { // VAR 3: one reduction per value per 3 stages (the 50-bit schedule needs one per 4-6); others: keep the probe's values in range
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = fp_reduce(x[i], fc);
            }
        }
        for (int i = 0; i < 8; i++) r ^= (u64)__double_as_longlong(x[i]);
    }
    if (r == 0x1234567) dyn[threadIdx.x] = r;
    out[t] = r;
}

// exactness: random |y| < 2^ybits, w < p: fp result congruent to y*w mod p and bounded; out[0] = failures, out[1] = max |r| / p * 1000
template <int FORM> __global__ void fp_check(unsigned long long *res, u64 p, int ybits, u32 seed) {
    const u32 t = threadIdx.x + blockIdx.x * blockDim.x;
    u64 s = (u64)t * 0x9E3779B97F4A7C15ull + seed;
    auto next = [&]() { s += 0x9E3779B97F4A7C15ull; u64 z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    const FpPrime fc{(double)p, 1.0 / (double)p};
    unsigned long long fails = 0, worst = 0;
    for (int it = 0; it < 256; it++) {
        const u64 wi = next() % p;
        long long yi = (long long)(next() >> (64 - ybits));
        if (next() & 1) yi = -yi;
        if (it < 8) yi = (it & 1) ? ((1ll << ybits) - 1 - it) : -((1ll << ybits) - 1 - it); // the extremes
        const FpTw ft{(double)wi, (double)wi / (double)p};
        const double r = FORM ? fp_mulmod_wp((double)yi, ft, fc) : fp_mulmod_pinv((double)yi, ft.w, fc);
        const long long ri = (long long)r;
        if ((double)ri != r) { fails++; continue; }
        __int128 d = (__int128)yi * (__int128)wi - (__int128)ri;
        if (d % (__int128)p != 0) fails++;
        const unsigned long long mag = (unsigned long long)(fabs(r) / (double)p * 1000.0);
        if (mag > worst) worst = mag;
    }
    atomicAdd(&res[0], fails);
    atomicMax(&res[1], worst);
}

// ---- inner-product forms: 16 terms per "row" (8 coefficients x 2 key components)
struct AccC { u64 P, Q, R; u32 cp, cq; };
__device__ __forceinline__ void mac_counter4(AccC (&a)[4], const u64 (&x)[4], const u64 (&k)[4]) {
    u64 sb, sc, sd;
    // P += xl*kl -> carry -> cp ; Q += xl*kh -> cq ; Q += xh*kl -> cq ; R += xh*kh
    asm("v_mad_u64_u32 %0, vcc, %23, %31, %0\n\t"
        "v_mad_u64_u32 %1, %20, %24, %32, %1\n\t"
        "v_mad_u64_u32 %2, %21, %25, %33, %2\n\t"
        "v_mad_u64_u32 %3, %22, %26, %34, %3\n\t"
        "v_addc_co_u32 %12, vcc, 0, %12, vcc\n\t"
        "v_addc_co_u32 %13, %20, 0, %13, %20\n\t"
        "v_addc_co_u32 %14, %21, 0, %14, %21\n\t"
        "v_addc_co_u32 %15, %22, 0, %15, %22\n\t"
        "v_mad_u64_u32 %4, vcc, %23, %35, %4\n\t"
        "v_mad_u64_u32 %5, %20, %24, %36, %5\n\t"
        "v_mad_u64_u32 %6, %21, %25, %37, %6\n\t"
        "v_mad_u64_u32 %7, %22, %26, %38, %7\n\t"
        "v_addc_co_u32 %16, vcc, 0, %16, vcc\n\t"
        "v_addc_co_u32 %17, %20, 0, %17, %20\n\t"
        "v_addc_co_u32 %18, %21, 0, %18, %21\n\t"
        "v_addc_co_u32 %19, %22, 0, %19, %22\n\t"
        "v_mad_u64_u32 %4, vcc, %27, %31, %4\n\t"
        "v_mad_u64_u32 %5, %20, %28, %32, %5\n\t"
        "v_mad_u64_u32 %6, %21, %29, %33, %6\n\t"
        "v_mad_u64_u32 %7, %22, %30, %34, %7\n\t"
        "v_addc_co_u32 %16, vcc, 0, %16, vcc\n\t"
        "v_addc_co_u32 %17, %20, 0, %17, %20\n\t"
        "v_addc_co_u32 %18, %21, 0, %18, %21\n\t"
        "v_addc_co_u32 %19, %22, 0, %19, %22\n\t"
        "v_mad_u64_u32 %8, vcc, %27, %35, %8\n\t"
        "v_mad_u64_u32 %9, %20, %28, %36, %9\n\t"
        "v_mad_u64_u32 %10, %21, %29, %37, %10\n\t"
        "v_mad_u64_u32 %11, %22, %30, %38, %11"
        : "+v"(a[0].P), "+v"(a[1].P), "+v"(a[2].P), "+v"(a[3].P), "+v"(a[0].Q), "+v"(a[1].Q), "+v"(a[2].Q), "+v"(a[3].Q), "+v"(a[0].R), "+v"(a[1].R), "+v"(a[2].R),
          "+v"(a[3].R), "+v"(a[0].cp), "+v"(a[1].cp), "+v"(a[2].cp), "+v"(a[3].cp), "+v"(a[0].cq), "+v"(a[1].cq), "+v"(a[2].cq), "+v"(a[3].cq), "=&s"(sb), "=&s"(sc),
          "=&s"(sd)
        : "v"(lo32(x[0])), "v"(lo32(x[1])), "v"(lo32(x[2])), "v"(lo32(x[3])), "v"(hi32(x[0])), "v"(hi32(x[1])), "v"(hi32(x[2])), "v"(hi32(x[3])), "v"(lo32(k[0])),
          "v"(lo32(k[1])), "v"(lo32(k[2])), "v"(lo32(k[3])), "v"(hi32(k[0])), "v"(hi32(k[1])), "v"(hi32(k[2])), "v"(hi32(k[3]))
        : "vcc");
}

template <int VAR> __global__ __launch_bounds__(256) void mac_probe(u64 *out, u64 p, FpPrime fc) {
    const u32 t = threadIdx.x + blockIdx.x * 256;
    u64 r = 0;
    u64 x[8], k[8];
    for (int i = 0; i < 8; i++) { x[i] = (t * 0x9E3779B97F4A7C15ull + i * 0xBF58476D1CE4E5B9ull) % p; k[i] = (t * 0xD1B54A32D192ED03ull + i * 0x94D049BB133111EBull) % p; }
    if (VAR == 0) {
        Acc128 acc[2][2][4];
        for (int c = 0; c < 2; c++) for (int g = 0; g < 2; g++) for (int e = 0; e < 4; e++) acc[c][g][e] = Acc128{0, 0, 0, 0};
        for (int it = 0; it < ITERS / 4; it++) {
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    const u64 xx[4] = {x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
                    const u64 kk[4] = {k[4 * g] ^ c, k[4 * g + 1] ^ c, k[4 * g + 2] ^ c, k[4 * g + 3] ^ c};
                    mac128x4(acc[c][g], xx, kk);
                }
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("" : "+v"(x[i]), "+v"(k[i]));
        }
        for (int c = 0; c < 2; c++) for (int g = 0; g < 2; g++) for (int e = 0; e < 4; e++) r ^= acc[c][g][e].a0 ^ acc[c][g][e].a1 ^ acc[c][g][e].a2 ^ acc[c][g][e].a3;
    } else if (VAR == 1) {
        AccC acc[2][2][4];
        for (int c = 0; c < 2; c++) for (int g = 0; g < 2; g++) for (int e = 0; e < 4; e++) acc[c][g][e] = AccC{0, 0, 0, 0, 0};
        for (int it = 0; it < ITERS / 4; it++) {
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    const u64 xx[4] = {x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
                    const u64 kk[4] = {k[4 * g] ^ c, k[4 * g + 1] ^ c, k[4 * g + 2] ^ c, k[4 * g + 3] ^ c};
                    mac_counter4(acc[c][g], xx, kk);
                }
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("" : "+v"(x[i]), "+v"(k[i]));
        }
        for (int c = 0; c < 2; c++) for (int g = 0; g < 2; g++) for (int e = 0; e < 4; e++) r ^= acc[c][g][e].P ^ acc[c][g][e].Q ^ acc[c][g][e].R ^ acc[c][g][e].cp ^ acc[c][g][e].cq;
    } else {
        double acc[2][8], xd[8], kd[8];
        for (int i = 0; i < 8; i++) { xd[i] = (double)x[i]; kd[i] = (double)k[i]; acc[0][i] = acc[1][i] = 0; }
        for (int it = 0; it < ITERS / 4; it++) {
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int i = 0; i < 8; i++) acc[c][i] += fp_mulmod_pinv(xd[i], kd[i] + c, fc);
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("" : "+v"(xd[i]), "+v"(kd[i]));
        }
        for (int c = 0; c < 2; c++) for (int i = 0; i < 8; i++) r ^= (u64)__double_as_longlong(acc[c][i]);
    }
    out[t] = r;
}

#define CH 8
template <int OP> __global__ __launch_bounds__(256) void probe(u32 *out, u32 sb) {
    u32 t = threadIdx.x + blockIdx.x * 256;
    double w[CH];
    u32 a[CH];
    const double b = 1.0 + t * 1e-9, c = 0.999999 - t * 1e-10;
    for (int i = 0; i < CH; i++) { w[i] = 1.0 + i + t * 1e-6; a[i] = t + i; }
    for (int it = 0; it < ITERS * 2; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(sb));
            if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(w[i]) : "v"(b), "v"(c));
            if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(w[i]) : "v"(c));
            if (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(w[i]) : "v"(c));
            if (OP == 4) asm volatile("v_rndne_f64 %0, %0" : "+v"(w[i]));
            if (OP == 5) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(w[i]) : "v"(a[i]));
            if (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(w[i]) : "s"(b), "v"(c));
            if (OP == 7) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(sb) : "vcc");
            if (OP == 8) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(a[i]) : "v"(w[i]));
            if (OP == 9) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(w[i]) : "v"(a[i] & 1));
        }
    }
    u32 r = 0;
    for (int i = 0; i < CH; i++) r ^= a[i] ^ (u32)__double_as_longlong(w[i]) ^ (u32)(__double_as_longlong(w[i]) >> 32);
    out[t] = r;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
    u32 *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto timeit = [&](auto &&launch) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return best;
    };
    const char *names[] = {"v_add_u32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_rndne_f64", "v_cvt_f64_u32", "v_fma_f64(s)", "v_mad_u64_u32", "v_cvt_u32_f64", "v_ldexp_f64"};
    const int blocks = cus * 8;
    double base = 0;
    for (int op = 0; op < 10; op++) {
        float ms = timeit([&] {
            switch (op) {
#define C(n) case n: probe<n><<<blocks, 256>>>(out, 12345u); break;
                C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9)
#undef C
            }
        });
        double wave_instr = (double)blocks * 4 * ITERS * 2 * CH;
        double cyc = clk / (wave_instr / (cus * 4.0) / (ms * 1e-3));
        if (op == 0) base = cyc;
        printf("%-22s %8.3f ms -> %6.2f nominal cycles/wave-instr/SIMD (x%.2f of v_add_u32)\n", names[op], ms, cyc, cyc / base);
    }
    // exactness of the FP64 modular product
    unsigned long long *res;
    hipMalloc(&res, 16);
    const u64 primes[] = {1099510054913ull /* 40-bit, 1 mod 2^16 */, 1125899906826241ull /* 50-bit, 1 mod 2^17 */, 562949953216513ull /* 49-bit */};
    for (u64 pp : primes)
        for (int ybits : {44, 48, 50, 51, 52})
            for (int form = 0; form < 2; form++) {
                hipMemset(res, 0, 16);
                if (form) fp_check<1><<<1024, 256>>>(res, pp, ybits, 7u); else fp_check<0><<<1024, 256>>>(res, pp, ybits, 7u);
                unsigned long long h[2];
                hipMemcpy(h, res, 16, hipMemcpyDeviceToHost);
                printf("fp mulmod check p=%llu (%d bits) |y|<2^%d form %s: %llu failures of %d, max |r|/p = %.3f\n", (unsigned long long)pp, 64 - __builtin_clzll(pp), ybits,
                       form ? "w/p" : "1/p", h[0], 1024 * 256 * 256, h[1] / 1000.0);
            }
    const u64 p = 288230376150630401ull; // a 58-bit prime = 1 mod 2^16
    const u64 wv = 0x123456789abcdefULL % p;
    const Shoup sw{wv, (u64)((((unsigned __int128)wv) << 64) / p)};
    const u64 pn = 1099510054913ull;
    const FpPrime fc{(double)pn, 1.0 / (double)pn};
    const u64 wn = 0x123456789abcdefULL % pn;
    const FpTw ft{(double)wn, (double)wn / (double)pn};
    const char *vn[] = {"int ct_bfly4_ng (15)", "fp64 ct 1/p form (8)", "fp64 ct w/p form (8)", "fp64 ct w/p + reduce/3 st", "fp64 gs w/p + reduce/3 st", "int gs_bfly4_ng (16)"};
    for (int occ = 0; occ < 2; occ++) {
        const size_t lds = occ ? 40 * 1024 : 0;
        for (int v = 0; v < 6; v++) {
            float ms = timeit([&] {
                switch (v) {
#define C(n) case n: bfly_probe<n><<<blocks, 256, lds>>>((u64 *)out, n == 0 || n == 5 ? p : pn, sw, fc, ft); break;
                    C(0) C(1) C(2) C(3) C(4) C(5)
#undef C
                }
            });
            double bf = (double)blocks * 256 * (ITERS / 16) * 12;
            double per_s = bf / (ms * 1e-3);
            printf("%-28s %s waves/SIMD %8.3f ms -> %7.1f G butterflies/s chip-wide, %.2f nominal cycles per wave-butterfly per SIMD; N=2^15 limb floor %.4f us\n", vn[v],
                   occ ? "4" : "8", ms, per_s / 1e9, clk / (per_s / 64 / (cus * 4.0)), 245760.0 / per_s * 1e6);
        }
    }
    const char *mn[] = {"mac128x4 (10 instr, 4 VGPR/acc)", "carry-counter (7 instr, 8 VGPR/acc)", "fp64 mulmod+add (7 instr, 2 VGPR/acc)"};
    for (int v = 0; v < 3; v++) {
        float ms = timeit([&] {
            switch (v) {
            case 0: mac_probe<0><<<blocks, 256>>>((u64 *)out, p, fc); break;
            case 1: mac_probe<1><<<blocks, 256>>>((u64 *)out, p, fc); break;
            case 2: mac_probe<2><<<blocks, 256>>>((u64 *)out, pn, fc); break;
            }
        });
        double terms = (double)blocks * 256 * (ITERS / 4) * 16;
        double per_s = terms / (ms * 1e-3);
        printf("%-40s %8.3f ms -> %7.1f G terms/s chip-wide, %.2f nominal cycles per wave-term per SIMD\n", mn[v], ms, per_s / 1e9, clk / (per_s / 64 / (cus * 4.0)));
    }
    return 0;
}
