// microbench.hip -- VALU issue-rate probes for the 64-bit modular multiply on gfx950 (SURVEY.md section 7 "hard parts":
// the guide does not give the 32-bit integer multiply rate).  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
// Prints wave-instructions per cycle per SIMD relative to v_add_u32 (a full-rate op: 1 wave-instruction / 2 cycles on SIMD-32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../troy_amd/csrc/bfly.h"

#define ITERS 4096
#define CHAINS 8

template <int OP> __global__ __launch_bounds__(256) void probe(uint32_t *out) {
    uint32_t t = threadIdx.x + blockIdx.x * 256;
    uint32_t a[CHAINS];
    uint64_t w[CHAINS];
    uint32_t b = t * 2654435761u + 12345u, c = t ^ 0x9E3779B9u;
    for (int i = 0; i < CHAINS; i++) { a[i] = t + i * 77; w[i] = ((uint64_t)t << 32) | (i * 1315423911u); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "vcc");
            if (OP == 4) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 6) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i + 1) % CHAINS]));
            if (OP == 7) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 8) asm volatile("v_mad_u32_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            if (OP == 10) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 11) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 12) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < CHAINS; i++) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[t] = r;
}

// the real thing: Harvey butterflies on 64-bit data, 8 independent butterflies per thread
__device__ __forceinline__ uint64_t mul_lazy(uint64_t x, uint64_t w, uint64_t wq, uint64_t p) { return w * x - __umul64hi(x, wq) * p; }
__global__ __launch_bounds__(256) void bfly_probe(uint64_t *out, uint64_t p, uint64_t w, uint64_t wq) {
    uint32_t t = threadIdx.x + blockIdx.x * 256;
    uint64_t x[8], y[8];
    const uint64_t two_p = 2 * p;
    for (int i = 0; i < 8; i++) { x[i] = (t * 0x9E3779B97F4A7C15ull + i) % p; y[i] = (t * 0xBF58476D1CE4E5B9ull + 3 * i) % p; }
    for (int it = 0; it < ITERS / 4; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t u = x[i] >= two_p ? x[i] - two_p : x[i];
            uint64_t v = mul_lazy(y[i], w, wq, p);
            x[i] = u + v;
            y[i] = u + two_p - v;
        }
    }
    uint64_t r = 0;
    for (int i = 0; i < 8; i++) r ^= x[i] ^ y[i];
    out[t] = r;
}

// production butterflies (troy_amd/csrc/bfly.h): 8 values per thread, one radix-8 round per iteration (12 butterflies)
__global__ __launch_bounds__(256) void bfly4_probe(uint64_t *out, uint64_t p, troyhip::Shoup w, long long *cyc) {
    using namespace troyhip;
    uint32_t t = threadIdx.x + blockIdx.x * 256;
    u64 x[8];
    const PrimeConst pc = make_prime_const(p);
    for (int i = 0; i < 8; i++) x[i] = (t * 0x9E3779B97F4A7C15ull + i * 0xBF58476D1CE4E5B9ull) % p;
    Shoup ww[4] = {w, w, w, w};
    long long c0 = clock64();
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
            ct_bfly4(X, Y, ww, pc);
#pragma unroll
            for (int i = 0; i < 4; i++) { x[ix[i]] = X[i]; x[iy[i]] = Y[i]; }
        }
    }
    long long c1 = clock64();
    uint64_t r = 0;
    for (int i = 0; i < 8; i++) r ^= x[i];
    out[t] = r;
    if (t == 0) *cyc = c1 - c0;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3; // Hz
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    const int blocks = cus * 8; // 8 blocks x 4 waves = 32 waves per CU = 8 per SIMD
    uint32_t *out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_lshl_add_u64",
                           "v_add_co_u32", "v_mad_u32_u16", "v_cndmask_b32", "v_add3_u32", "v_mul_hi_u32_u24", "v_mov_b32_dpp"};
    double base = 0;
    for (int op = 0; op < 13; op++) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            switch (op) {
#define C(n) case n: probe<n><<<blocks, 256>>>(out); break;
                C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12)
#undef C
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double wave_instr = (double)blocks * 4 * ITERS * CHAINS; // wave-instructions issued
        double per_simd_per_s = wave_instr / (cus * 4.0) / (best * 1e-3);
        double cyc = clk / per_simd_per_s;
        if (op == 0) base = cyc;
        printf("%-18s %8.3f ms  -> %6.2f cycles/wave-instr/SIMD at nominal clock (x%.2f of v_add_u32)\n", names[op], best, cyc, cyc / base);
    }
    {
        uint64_t p = 0xffffffffffc0001ULL >> 3 | 1, w = 0x123456789abcdefULL % p;
        uint64_t wq = (uint64_t)((((unsigned __int128)w) << 64) / p);
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            bfly_probe<<<blocks, 256>>>((uint64_t *)out, p, w, wq);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double bf = (double)blocks * 256 * (ITERS / 4) * 8;
        double per_s = bf / (best * 1e-3);
        printf("harvey butterfly   %8.3f ms  -> %.1f G butterflies/s chip-wide = %.2f cycles per wave-butterfly per SIMD; "
               "N=2^15 limb NTT compute floor %.3f us\n", best, per_s / 1e9, clk / (per_s / 64 / (cus * 4.0)), 245760.0 / per_s * 1e6);
    }
    {
        uint64_t p = 0xffffffffffc0001ULL >> 3 | 1, w = 0x123456789abcdefULL % p;
        troyhip::Shoup sw{w, (uint64_t)((((unsigned __int128)w) << 64) / p)};
        long long *dcyc; hipMalloc(&dcyc, 8);
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            bfly4_probe<<<blocks, 256>>>((uint64_t *)out, p, sw, dcyc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        long long hc; hipMemcpy(&hc, dcyc, 8, hipMemcpyDeviceToHost);
        double bf = (double)blocks * 256 * (ITERS / 16) * 12;
        double per_s = bf / (best * 1e-3);
        printf("bfly.h ct_bfly4    %8.3f ms  -> %.1f G butterflies/s chip-wide; wave0 ran %lld shader cycles in %.3f ms => %.0f MHz effective; "
               "N=2^15 limb NTT compute floor %.3f us\n", best, per_s / 1e9, hc, best, hc / (best * 1e-3) / 1e6, 245760.0 / per_s * 1e6);
    }
    return 0;
}
