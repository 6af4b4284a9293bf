// selftest.hip -- device-side unit probes of modarith.h / bfly.h (the a-1 row of SURVEY.md section 8: dBarrettReduce64/128,
// dMultiplyUintMod(Lazy), the butterfly forms, the 128-bit multiply-accumulate), driven by tests/test_gpu_parity.py through
// troyhip_test_modarith with edge values (p - 1, lazy 2p-1 / 4p-1 / 8p-1 inputs, 36..61-bit primes) and the reference's own
// known-answer vectors (test/utils/uintarithsmallmod.cpp).  Test support: not on any product path.
#include "kernels.h"
#include "bfly.h"

namespace troyhip {

// out layout: ops 0..5: n words; butterfly ops: 2n words (X', Y' canonical); op 10: one word (sum of a[i] * b[i] mod p)
__global__ __launch_bounds__(64) void modarith_probe_kernel(int op, const u64 *a, const u64 *b, const u64 *c, Mod m, Shoup aux, u64 *out, u64 n) {
    const u64 i = (u64)blockIdx.x * 64 + threadIdx.x;
    const u64 p = m.p;
    if (op <= 5) {
        if (i >= n) return;
        const u64 x = a[i], y = b ? b[i] : 0;
        u64 r = 0;
        if (op == 0) r = barrett64(x, m);
        if (op == 1) r = barrett128(x, y, m);
        if (op == 2) r = mulmod(x, y, m);
        if (op == 3) r = mul_shoup(x, y, c[i], p);        // c = floor(y * 2^64 / p)
        if (op == 4) r = mul_lazy(x, y, c[i], p);         // in [0, 2p), returned as is
        if (op == 5) r = reduce_prod((u128)x * y, make_prod_mod(m));
        out[i] = r;
        return;
    }
    if (op == 10) { // lazy 128-bit accumulation of n products in four interleaved accumulators, one reduction
        if (i != 0) return;
        Acc128 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (u64 k = 0; k + 4 <= n; k += 4) {
            const u64 xx[4] = {a[k], a[k + 1], a[k + 2], a[k + 3]}, kk[4] = {b[k], b[k + 1], b[k + 2], b[k + 3]};
            mac128x4(acc, xx, kk);
        }
        u64 r = 0;
        for (int j = 0; j < 4; j++) r = addmod(r, barrett128(mk64(acc[j].a0, acc[j].a1), mk64(acc[j].a2, acc[j].a3), m), p);
        out[0] = r;
        return;
    }
    // butterflies, four at a time: X = a, Y = b, twiddle operand c (quotient formed here)
    const u64 base = i * 4;
    if (base >= n) return;
    u64 X[4], Y[4];
    Shoup w[4];
    for (int j = 0; j < 4; j++) {
        const u64 k = base + j < n ? base + j : n - 1;
        X[j] = a[k];
        Y[j] = b[k];
        const u64 wv = (op == 11 || op == 12) ? c[0] : c[k];
        w[j].op = wv;
        w[j].quo = (u64)((((u128)wv) << 64) / p);
    }
    const PrimeConst pc = make_prime_const(p);
    if (op == 6) ct_bfly4(X, Y, w, pc);
    if (op == 7) ct_bfly4_ng(X, Y, w, pc);
    if (op == 8) gs_bfly4(X, Y, w, pc);
    if (op == 9) gs_bfly4_last(X, Y, w, aux, pc);       // aux = N^-1 as a Shoup operand; c = the pre-scaled twiddle
    if (op == 11) ct_bfly4<true>(X, Y, w, pc);           // wave-uniform twiddle c[0] read from SGPRs
    if (op == 12) ct_bfly4_ng<true>(X, Y, w, pc);
    if (op == 13) gs_bfly4_ng(X, Y, w, 8 * p, pc);      // kp = 8p: inputs below 8p
    for (int j = 0; j < 4; j++) {
        if (base + j >= n) break;
        out[2 * (base + j)] = barrett64(X[j], m);
        out[2 * (base + j) + 1] = barrett64(Y[j], m);
    }
}

void launch_modarith_probe(int op, const u64 *a, const u64 *b, const u64 *c, u64 p, u64 aux_value, u64 *out, u64 n, hipStream_t s) {
    const Mod m = make_mod(p);
    const Shoup aux = make_shoup(aux_value % p, p);
    const u64 threads = (op >= 6 && op != 10) ? (n + 3) / 4 : n;
    TROY_LAUNCH(modarith_probe_kernel, dim3(ceil_div(threads ? threads : 1, 64)), dim3(64), 0, s, op, a, b, c, m, aux, out, n);
    launch_check("modarith_probe_kernel");
}

} // namespace troyhip
