// test_fp_plan.cpp -- CPU unit test of the exact-FP64 modular arithmetic and its host-side bound walk (troy_amd/csrc/fpmod.h), compiled for the host
// with g++ -DTROYHIP_CPU_EMUL.  The kernels trust fp_plan / fp_plan_inv to say before which rounds the lazy values must be reduced; a wrong mask would
// otherwise only show as a mismatch on the GPU.  Here a plain-loop model of the transforms runs on doubles under the masks the walk returns, for primes
// of 33 .. 50 bits and the round shapes the kernels use, and checks
//   (1) every product: congruent to the integer product, within the magnitude bound the header states;
//   (2) every intermediate value: an exact integer below 2^53 and below the bound the walk carries for that round;
//   (3) the final canonical residues: equal to the integer transform's (unsigned __int128 arithmetic);
//   (4) the walk refuses (out_bound < 0) schedules in which even a freshly reduced round would pass 2^53.
#include "fpmod.h"
#include <cstdio>
#include <random>
#include <vector>

using namespace troyhip;
typedef unsigned __int128 u128;
static int failures = 0;
#define CHECK(c, ...) do { if (!(c)) { if (failures++ < 20) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } } while (0)

static u64 mulmod(u64 a, u64 b, u64 p) { return (u64)((u128)a * b % p); }
static u64 powmod(u64 a, u64 e, u64 p) { u64 r = 1; for (; e; e >>= 1, a = mulmod(a, a, p)) if (e & 1) r = mulmod(r, a, p); return r; }
static bool is_prime(u64 n) {
    if (n < 2 || n % 2 == 0) return n == 2;
    u64 d = n - 1; int s = 0;
    while (d % 2 == 0) d /= 2, s++;
    for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        if (a % n == 0) continue;
        u64 x = powmod(a % n, d, n);
        if (x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int i = 1; i < s && comp; i++) { x = mulmod(x, x, n); if (x == n - 1) comp = false; }
        if (comp) return false;
    }
    return true;
}
// largest prime below 2^bits with p = 1 (mod 2n)
static u64 ntt_prime(int bits, u64 two_n) { u64 p = ((1ull << bits) - 1) / two_n * two_n + 1; while (!is_prime(p)) p -= two_n; return p; }
static u64 primitive_root_2n(u64 p, u64 two_n, std::mt19937_64 &rng) {
    for (;;) { const u64 g = powmod(rng() % (p - 2) + 2, (p - 1) / two_n, p); if (powmod(g, two_n / 2, p) == p - 1) return g; }
}
static unsigned bitrev(unsigned x, int bits) { unsigned r = 0; for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i); return r; }
// signed value of an integer-valued double modulo p
static u64 residue(double v, u64 p) { const long long i = (long long)v; const long long r = i % (long long)p; return (u64)(r < 0 ? r + (long long)p : r); }
static bool is_integer(double v) { return v == std::nearbyint(v) && std::fabs(v) < 0x1p53; }

static void check_products(u64 p, std::mt19937_64 &rng) {
    const FpPrime c = make_fp_prime(p);
    for (int ybits : {20, 44, 48, 50, 51, 52}) {
        for (int it = 0; it < 20000; it++) {
            u64 ymag = rng() & ((1ull << ybits) - 1);
            if (it < 4) ymag = (1ull << ybits) - 1 - it; // the extremes
            const double y = (it & 1) ? -(double)ymag : (double)ymag;
            u64 w = rng() % p;
            if (it == 5) w = p - 1;
            if (it == 6) w = (p - 1) / 2;
            const double wd = (double)w, wp = wd / (double)p;
            const u64 want = mulmod(residue(y, p), w, p);
            const double r1 = fp_mulmod_wp(y, wd, wp, c), r2 = fp_mulmod_pinv(y, wd, c);
            CHECK(is_integer(r1) && residue(r1, p) == want, "mulmod_wp p=%llu y=%.0f w=%llu", (unsigned long long)p, y, (unsigned long long)w);
            CHECK(is_integer(r2) && residue(r2, p) == want, "mulmod_pinv p=%llu y=%.0f w=%llu", (unsigned long long)p, y, (unsigned long long)w);
            CHECK(std::fabs(r1) <= (0.5 + (double)ymag * 0x1p-52) * (double)p + 1, "mulmod_wp bound p=%llu y=%.0f: %.0f", (unsigned long long)p, y, r1);
            CHECK(std::fabs(r2) <= (0.5 + 3 * (double)ymag * 0x1p-53) * (double)p + 1, "mulmod_pinv bound p=%llu y=%.0f: %.0f", (unsigned long long)p, y, r2);
        }
    }
    for (int it = 0; it < 20000; it++) { // reduction and canonical form of anything below 2^53
        double x = (double)(rng() >> 11);
        if (it == 0) x = 0x1p53 - 1;
        if (it == 1) x = (double)(p / 2);
        if (it == 2) x = (double)(p / 2 + 1);
        if (it == 3) x = (double)p;
        if (it == 4) x = (double)(p - 1);
        if (it & 1) x = -x;
        const double r = fp_reduce(x, c);
        CHECK(is_integer(r) && residue(r, p) == residue(x, p) && std::fabs(r) <= (double)p / 2 + 2, "fp_reduce p=%llu x=%.0f", (unsigned long long)p, x);
        CHECK(fp_canonical(x, c, p) == residue(x, p), "fp_canonical p=%llu x=%.0f", (unsigned long long)p, x);
    }
    for (u64 x : std::initializer_list<u64>{0, 1, p - 1, (1ull << 52) - 1}) CHECK(fp_to_u64(fp_from_u64(x)) == x && fp_from_u64(x) == (double)x, "conversion of %llu", (unsigned long long)x);
}

struct Tables { std::vector<u64> w, winv; }; // powers of the 2n-th root in bit-reversed order, as a merged negacyclic transform indexes them
static Tables make_tables(u64 p, int logn, std::mt19937_64 &rng) {
    const unsigned n = 1u << logn;
    const u64 g = primitive_root_2n(p, 2ull * n, rng), gi = powmod(g, p - 2, p);
    Tables t{std::vector<u64>(n), std::vector<u64>(n)};
    u64 a = 1, b = 1;
    for (unsigned i = 0; i < n; i++, a = mulmod(a, g, p), b = mulmod(b, gi, p)) { t.w[bitrev(i, logn)] = a; t.winv[bitrev(i, logn)] = b; }
    return t;
}

// forward (Cooley-Tukey, stage s has m = 2^s groups with twiddle w[m + i]); `form`: 0 = (w, w / p) pairs, 1 = single doubles (1 / p quotient)
static void forward_model(u64 p, int logn, const int *rounds, int n_rounds, double b_in, int form, std::mt19937_64 &rng, int input_kind) {
    const unsigned n = 1u << logn;
    const FpPrime c = make_fp_prime(p);
    const Tables t = make_tables(p, logn, rng);
    const FpPlan plan = fp_plan(p, b_in, rounds, n_rounds, form ? 1.5 : 1.0);
    if (plan.out_bound < 0) return; // the kernels keep the integer path
    const double pd = (double)p, lim = 0x1p53;
    std::vector<u64> xi(n);
    std::vector<double> x(n);
    for (unsigned i = 0; i < n; i++) { // inputs: integers of magnitude up to b_in p, congruent to xi
        const u64 r = input_kind == 0 ? p - 1 : (input_kind == 1 ? rng() % p : ((i & 1) ? p - 1 : 0));
        const double lift = std::floor(b_in - 1.0) * pd * ((input_kind == 2 && (i & 2)) ? -1.0 : 1.0);
        x[i] = (double)r + (std::fabs((double)r + lift) < b_in * pd ? lift : 0.0);
        xi[i] = residue(x[i], p);
    }
    double bound = b_in;
    int s = 0;
    for (int r = 0; r < n_rounds; r++) {
        if (plan.mask >> r & 1u) { for (auto &v : x) v = fp_reduce(v, c); bound = 0.5 + 0x1p-40; }
        for (int k = 0; k < rounds[r]; k++, s++) {
            const unsigned m = 1u << s, half = n >> (s + 1);
            for (unsigned i = 0; i < m; i++) {
                const u64 w = t.w[m + i];
                const double wd = (double)w, wp = wd / pd;
                for (unsigned j = 2 * i * half; j < (2 * i + 1) * half; j++) {
                    const double v = form ? fp_mulmod_pinv(x[j + half], wd, c) : fp_mulmod_wp(x[j + half], wd, wp, c);
                    const u64 vi = mulmod(xi[j + half], w, p);
                    const u64 a = xi[j];
                    xi[j] = (a + vi) % p, xi[j + half] = (a + p - vi) % p;
                    const double X = x[j];
                    x[j] = X + v, x[j + half] = X - v;
                }
            }
            bound = fp_stage_bound(bound, pd, form ? 1.5 : 1.0);
            double mx = 0;
            for (unsigned i = 0; i < n; i++) { mx = std::fmax(mx, std::fabs(x[i])); CHECK(is_integer(x[i]) && residue(x[i], p) == xi[i], "forward value p=%llu stage %d", (unsigned long long)p, s); }
            CHECK(mx < lim && mx <= bound * pd + 2 * (s + 1), "forward bound p=%llu stage %d: %.3f p > %.3f p", (unsigned long long)p, s, mx / pd, bound);
        }
    }
    CHECK(std::fabs(bound - plan.out_bound) <= 1e-9 * bound, "out_bound %.6f vs %.6f", plan.out_bound, bound);
    for (unsigned i = 0; i < n; i++) CHECK(fp_canonical(x[i], c, p) == xi[i], "forward result p=%llu", (unsigned long long)p);
}

// inverse (Gentleman-Sande): X' = X + Y, Y' = (X - Y) w
static void inverse_model(u64 p, int logn, const int *rounds, int n_rounds, std::mt19937_64 &rng, int input_kind) {
    const unsigned n = 1u << logn;
    const FpPrime c = make_fp_prime(p);
    const Tables t = make_tables(p, logn, rng);
    const FpPlan plan = fp_plan_inv(p, 1.0, rounds, n_rounds);
    if (plan.out_bound < 0) return;
    const double pd = (double)p, lim = 0x1p53;
    std::vector<u64> xi(n);
    std::vector<double> x(n);
    for (unsigned i = 0; i < n; i++) { xi[i] = input_kind == 0 ? p - 1 : (input_kind == 1 ? rng() % p : ((i & 1) ? p - 1 : 0)); x[i] = (double)xi[i]; }
    double bound = 1.0;
    int s = logn - 1;
    for (int r = 0; r < n_rounds; r++) {
        if (plan.mask >> r & 1u) { for (auto &v : x) v = fp_reduce(v, c); bound = 0.5 + 0x1p-40; }
        for (int k = 0; k < rounds[r]; k++, s--) {
            const unsigned m = 1u << s, half = n >> (s + 1);
            for (unsigned i = 0; i < m; i++) {
                const u64 w = t.winv[m + i];
                const double wd = (double)w, wp = wd / pd;
                for (unsigned j = 2 * i * half; j < (2 * i + 1) * half; j++) {
                    const double X = x[j], Y = x[j + half];
                    CHECK(std::fabs(X - Y) < lim, "inverse difference p=%llu stage %d", (unsigned long long)p, s);
                    x[j] = X + Y, x[j + half] = fp_mulmod_wp(X - Y, wd, wp, c);
                    const u64 a = xi[j], b = xi[j + half];
                    xi[j] = (a + b) % p, xi[j + half] = mulmod((a + p - b) % p, w, p);
                }
            }
            double mx = 0;
            for (unsigned i = 0; i < n; i++) { mx = std::fmax(mx, std::fabs(x[i])); CHECK(is_integer(x[i]) && residue(x[i], p) == xi[i], "inverse value p=%llu stage %d", (unsigned long long)p, s); }
            bound *= 2;
            CHECK(mx < lim && mx <= std::fmax(bound, 2.5) * pd + 2, "inverse bound p=%llu stage %d: %.3f p > %.3f p", (unsigned long long)p, s, mx / pd, bound);
        }
        bound = bound < 2.5 ? 2.5 : bound;
    }
    CHECK(std::fabs(bound - plan.out_bound) <= 1e-9 * bound, "inverse out_bound %.6f vs %.6f", plan.out_bound, bound);
    for (unsigned i = 0; i < n; i++) CHECK(fp_canonical(x[i], c, p) == xi[i], "inverse result p=%llu", (unsigned long long)p);
}

int main() {
    std::mt19937_64 rng(20260402);
    // round shapes of the kernels: single pass N = 2^15 (register round + sub-block rounds), N = 2^12 .. 2^14, the two passes at N = 2^13 .. 2^17
    struct Shape { int n, r[6]; } fwd[] = {{4, {5, 4, 4, 2}}, {4, {2, 4, 4, 2}}, {4, {3, 4, 4, 2}}, {4, {4, 4, 4, 2}}, {5, {4, 3, 3, 3, 3}}, {5, {3, 3, 3, 3, 3}}, {6, {4, 4, 3, 3, 3, 0}},
                                  {2, {8, 7}}, {1, {15}}, {3, {1, 1, 13}}};
    struct ShapeI { int n, r[6]; } inv[] = {{5, {2, 4, 4, 4, 1}}, {5, {2, 4, 4, 2, 0}}, {5, {3, 3, 3, 3, 3}}, {4, {3, 3, 3, 4}}, {2, {8, 7}}, {1, {12}}};
    int planned = 0, refused = 0;
    for (int bits : {33, 34, 36, 40, 45, 48, 49, 50}) {
        {
            const u64 p = ntt_prime(bits, 2ull << 12);
            check_products(p, rng);
        }
        for (const auto &sh : fwd) {
            int logn = 0, n_rounds = 0;
            for (int i = 0; i < sh.n; i++) if (sh.r[i]) logn += sh.r[i], n_rounds++;
            const int cap = logn > 12 ? 12 : logn; // the model is O(N log N) per case: transforms up to N = 4096, the walk on the full shape
            int rr[6], nr = 0, left = cap;
            for (int i = 0; i < n_rounds && left > 0; i++) { rr[nr] = sh.r[i] < left ? sh.r[i] : left; left -= rr[nr++]; }
            const u64 p = ntt_prime(bits, 2ull << cap);
            for (double b_in : {1.0, 5.0}) for (int form = 0; form < 2; form++) {
                const FpPlan full = fp_plan(p, b_in, sh.r, n_rounds, form ? 1.5 : 1.0);
                (full.out_bound < 0 ? refused : planned)++;
                if (full.out_bound >= 0) { // what the walk promises on the full shape: replay it independently
                    double b = b_in; bool ok = true;
                    for (int r = 0; r < n_rounds; r++) {
                        if (full.mask >> r & 1u) b = 0.5 + 0x1p-40;
                        for (int k = 0; k < sh.r[r]; k++) { b = fp_stage_bound(b, (double)p, form ? 1.5 : 1.0); ok = ok && b * (double)p < 0x1p53; }
                    }
                    CHECK(ok, "fp_plan lets a value pass 2^53: bits %d b_in %.0f form %d", bits, b_in, form);
                }
                for (int kind = 0; kind < 3; kind++) forward_model(p, cap, rr, nr, b_in, form, rng, kind);
            }
        }
        for (const auto &sh : inv) {
            int logn = 0, n_rounds = 0;
            for (int i = 0; i < sh.n; i++) if (sh.r[i]) logn += sh.r[i], n_rounds++;
            const int cap = logn > 12 ? 12 : logn;
            int rr[6], nr = 0, left = cap;
            for (int i = 0; i < n_rounds && left > 0; i++) { rr[nr] = sh.r[i] < left ? sh.r[i] : left; left -= rr[nr++]; }
            const u64 p = ntt_prime(bits, 2ull << cap);
            const FpPlan full = fp_plan_inv(p, 1.0, sh.r, n_rounds);
            (full.out_bound < 0 ? refused : planned)++;
            if (full.out_bound >= 0) {
                double b = 1.0; bool ok = true;
                for (int r = 0; r < n_rounds; r++) {
                    if (full.mask >> r & 1u) b = 0.5 + 0x1p-40;
                    b = std::ldexp(b, sh.r[r]); ok = ok && b * (double)p <= 0x1p53;
                    if (b < 2.5) b = 2.5;
                }
                CHECK(ok, "fp_plan_inv lets a sum pass 2^53: bits %d", bits);
            }
            for (int kind = 0; kind < 3; kind++) inverse_model(p, cap, rr, nr, rng, kind);
        }
    }
    { // unschedulable: one round of 15 stages at 50 bits (8.5 p > 2^53 / p = 8), twelve inverse stages in a round at 45 bits (2^11 p > 2^53)
        const int one[1] = {15}, twelve[1] = {12};
        CHECK(fp_plan((1ull << 50) - 27, 1.0, one, 1).out_bound < 0, "fp_plan accepted 15 stages at 50 bits");
        CHECK(fp_plan_inv((1ull << 45) - 55, 1.0, twelve, 1).out_bound < 0, "fp_plan_inv accepted 12 stages at 45 bits");
        CHECK(fp_plan((1ull << 40) - 87, 1.0, one, 1).out_bound > 0, "fp_plan refused 15 stages at 40 bits");
    }
    std::printf("fp_plan: %d schedules walked and modelled, %d refused, %d failures\n", planned, refused, failures);
    return failures ? 1 : 0;
}
