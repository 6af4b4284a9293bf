// evaluator.cpp -- see evaluator.h.  Host-side orchestration only; all arithmetic is in the kernels.
#include "evaluator.h"
#include "kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace troyhip {

// The fused forms below are what runs wherever the kernels support the shape; the unfused kernels are the fallback for the shapes they do not (N < 4096,
// N = 2^17, ciphertext sizes other than 2 x 2, ...).  Probe builds (-DTROYHIP_PROBES: the CPU emulator build of the test suite, `make probes`) can force a
// fallback at a fused shape: TROYHIP_TENSOR / TROYHIP_KS / TROYHIP_MODDOWN / TROYHIP_CORR = split.  The shipped library ignores them.
// TROYHIP_TENSOR=split keeps the ciphertext tensor of the BEHZ multiply in its own kernel; read once
static bool tensor_fused() {
    static const bool v = [] { const char *e = probe_env("TROYHIP_TENSOR"); return !(e && e[0] == 's'); }();
    return v;
}
// TROYHIP_KS=split keeps the transforms and the inner product in separate kernels (tests, measurements); read once
static bool ks_fused() {
    static const bool v = [] { const char *e = probe_env("TROYHIP_KS"); return !(e && e[0] == 's'); }();
    return v;
}

// TROYHIP_MODDOWN=split keeps the BFV mod-down in its own kernel behind the inverse transform (tests, measurements); read once
static bool ks_moddown_fused() {
    static const bool v = [] { const char *e = probe_env("TROYHIP_MODDOWN"); return !(e && e[0] == 's'); }();
    return v;
}

// TROYHIP_CORR=split keeps the CKKS divide-and-round correction as element-wise kernels around a plain transform (tests, measurements)
static bool corr_fused() {
    static const bool v = [] { const char *e = probe_env("TROYHIP_CORR"); return !(e && e[0] == 's'); }();
    return v;
}
// Small launches (Context::small_launch: one ciphertext, a few small ones) take merged forms: one launch over the q-base and the B_sk-base rows of a
// product, one first pass over the special limb and the data limbs of a mod-down.  TROYHIP_SMALL=split keeps them on the kernels of the large batch,
// TROYHIP_SMALL=merged uses the merged forms at every size (tests, measurements); read once
static bool take_merged(const Context &c, u64 rows) {
    static const int mode = [] { const char *e = getenv("TROYHIP_SMALL"); return !e ? 0 : (e[0] == 's' ? 1 : (e[0] == 'm' ? 2 : 0)); }();
    return mode == 2 || (mode == 0 && c.small_launch(rows));
}
static bool primes_at_least_33_bits(const Context &c, int limbs) { // what the lazy reductions of the fused epilogues (lite_reduce4) take
    for (int l = 0; l < limbs; l++)
        if (c.primes[l] < (u64(1) << 33)) return false;
    return true;
}

static inline u64 poly_words(const Context &c, int limbs) { return (u64)limbs * c.N; }

void Evaluator::check_ct(const CtBatch &a) const {
    if (!c.has_device) throw Error(ST_LOGIC_ERROR, "this context was created host-only (troyhip_context_create_host)");
    if (!a.data) throw Error(ST_INVALID_ARGUMENT, "encrypted is not valid for encryption parameters");
    if (!c.is_data_level(a.limbs)) throw Error(ST_INVALID_ARGUMENT, "encrypted is not valid for encryption parameters");
    if (a.size < 1 || a.size > 16) throw Error(ST_INVALID_ARGUMENT, "encrypted is not valid for encryption parameters");
    if (a.bstride < (u64)a.size * poly_words(c, a.limbs)) throw Error(ST_INVALID_ARGUMENT, "batch stride is smaller than one ciphertext");
}

// evaluator_cuda.cu:32-51 isScaleWithinBounds
bool Evaluator::scale_ok(double scale, int limbs) const {
    std::vector<u64> q(c.primes.begin(), c.primes.begin() + limbs);
    int bound = host::bit_length_of_product(q);
    return !(scale <= 0 || ((int)std::log2(scale) >= bound));
}

// evaluator_cuda.cu:53-115 balanceCorrectionFactors
void Evaluator::balance_correction(u64 f1, u64 f2, u64 &f, u64 &e1_out, u64 &e2_out) const {
    const u64 t = c.t, half_t = t / 2;
    auto sum_abs = [&](u64 x, u64 y) {
        int64_t xb = (int64_t)(x > half_t ? x - t : x), yb = (int64_t)(y > half_t ? y - t : y);
        return std::llabs(xb) + std::llabs(yb);
    };
    u64 ratio;
    if (!host::inv_mod(f1, t, ratio)) throw Error(ST_LOGIC_ERROR, "invalid correction factor1");
    ratio = host::mul_mod(ratio, f2 % t, t);
    u64 e1 = ratio, e2 = 1;
    int64_t sum = sum_abs(e1, e2);
    int64_t prev_a = (int64_t)t, prev_b = 0, a = (int64_t)ratio, b = 1;
    auto gcd = [](u64 x, u64 y) { while (y) { u64 r = x % y; x = y; y = r; } return x; };
    while (a != 0) {
        int64_t q = prev_a / a, tmp = prev_a % a;
        prev_a = a; a = tmp;
        tmp = prev_b - b * q; prev_b = b; b = tmp;
        u64 a_mod = (u64)std::llabs(a) % t;
        if (a < 0 && a_mod) a_mod = t - a_mod;
        u64 b_mod = (u64)std::llabs(b) % t;
        if (b < 0 && b_mod) b_mod = t - b_mod;
        if (a_mod != 0 && gcd(a_mod, t) == 1) {
            int64_t ns = sum_abs(a_mod, b_mod);
            if (ns < sum) { sum = ns; e1 = a_mod; e2 = b_mod; }
        }
    }
    f = host::mul_mod(e1, f1 % t, t);
    e1_out = e1;
    e2_out = e2;
}

void Evaluator::scalar_mul(CtBatch &a, u64 scalar, u64 batch, hipStream_t s) {
    u64 sc[64];
    for (int l = 0; l < a.limbs; l++) sc[l] = scalar % c.primes[l];
    LimbMap map = c.ct_map(a.limbs);
    const u64 words = (u64)a.size * poly_words(c, a.limbs);
    if (a.bstride == words) launch_mul_scalar(a.data, c.d_desc, map, sc, c.logn, batch * a.size * a.limbs, s);
    else for (u64 b = 0; b < batch; b++) launch_mul_scalar(a.data + b * a.bstride, c.d_desc, map, sc, c.logn, (u64)a.size * a.limbs, s);
}

// addInplace / subInplace (evaluator_cuda.cu:145-260)
void Evaluator::add_sub(CtBatch &a, const CtBatch &b_in, u64 batch, bool sub, hipStream_t s) {
    check_ct(a);
    check_ct(b_in);
    if (a.limbs != b_in.limbs) throw Error(ST_INVALID_ARGUMENT, "encrypted1 and encrypted2 parameter mismatch");
    if (a.ntt != b_in.ntt) throw Error(ST_INVALID_ARGUMENT, "NTT form mismatch");
    if (!(a.scale == b_in.scale || std::fabs(a.scale - b_in.scale) < std::ldexp(std::max(std::fabs(a.scale), 1.0), -40))) throw Error(ST_INVALID_ARGUMENT, "scale mismatch");
    const u64 pw = poly_words(c, a.limbs);
    CtBatch b = b_in;
    c.arena.begin(s);
    if (a.cf != b.cf) {
        // BGV: balance the correction factors first (evaluator_cuda.cu:170-190).  Without a plain modulus (CKKS) there is nothing to balance against:
        // the reference's ciphertexts all carry factor 1 there; a descriptor that says otherwise is refused (it used to divide by zero)
        if (!c.t) throw Error(ST_INVALID_ARGUMENT, "correction factor mismatch");
        u64 f, e1, e2;
        balance_correction(a.cf, b.cf, f, e1, e2);
        scalar_mul(a, e1, batch, s);
        u64 *copy = c.arena.take(batch * b.size * pw);
        launch_copy_strided(b.data, b.bstride, copy, b.size * pw, b.size * pw, batch, s);
        b.data = copy;
        b.bstride = b.size * pw;
        scalar_mul(b, e2, batch, s);
        a.cf = f;
        b.cf = f;
    }
    const int mn = std::min(a.size, b.size), mx = std::max(a.size, b.size);
    if (a.bstride < (u64)mx * pw) throw Error(ST_INVALID_ARGUMENT, "destination batch stride too small for the result size");
    LimbMap map = c.ct_map(a.limbs);
    if (a.size == b.size && a.bstride == a.size * pw && b.bstride == a.bstride) {
        launch_ew(sub ? 1 : 0, a.data, b.data, a.data, c.d_desc, map, c.logn, batch * a.size * a.limbs, s);
    } else {
        for (u64 i = 0; i < batch; i++) {
            u64 *x = a.data + i * a.bstride;
            const u64 *y = b.data + i * b.bstride;
            launch_ew(sub ? 1 : 0, x, y, x, c.d_desc, map, c.logn, (u64)mn * a.limbs, s);
            if (a.size < b.size) { // tail polys: copy (add) or negate (sub), evaluator_cuda.cu:199-204,253-258
                if (sub) launch_ew(2, y + mn * pw, nullptr, x + mn * pw, c.d_desc, map, c.logn, (u64)(b.size - mn) * a.limbs, s);
                else launch_copy_strided(y + mn * pw, 0, x + mn * pw, 0, (b.size - mn) * pw, 1, s);
            }
        }
    }
    a.size = mx;
}

void Evaluator::negate(CtBatch &a, u64 batch, hipStream_t s) { // evaluator_cuda.cu:117-143
    check_ct(a);
    LimbMap map = c.ct_map(a.limbs);
    const u64 words = (u64)a.size * poly_words(c, a.limbs);
    if (a.bstride == words) launch_ew(2, a.data, nullptr, a.data, c.d_desc, map, c.logn, batch * a.size * a.limbs, s);
    else for (u64 b = 0; b < batch; b++) launch_ew(2, a.data + b * a.bstride, nullptr, a.data + b * a.bstride, c.d_desc, map, c.logn, (u64)a.size * a.limbs, s);
}

size_t Evaluator::scratch_multiply(int sa, int sb, int limbs, u64 batch) const {
    const u64 N = c.N;
    const int ds = sa + sb - 1;
    size_t pad = 64;
    if (c.scheme == SCHEME_BFV) {
        const int nb = (int)c.level(limbs).rns.Bsk.size();
        return (size_t)batch * N * ((size_t)(sa + sb) * (limbs + nb) + (size_t)ds * (limbs + nb) * 2) + 8 * pad;
    }
    return (size_t)batch * N * limbs * (sa + sb + ds) + 4 * pad;
}

// multiplyInplace -> bfvMultiply / ckksMultiply / bgvMultiply (evaluator_cuda.cu:262-501)
void Evaluator::multiply(const CtBatch &a, const CtBatch &b, CtBatch &out, u64 batch, hipStream_t s) {
    check_ct(a);
    check_ct(b);
    if (a.limbs != b.limbs) throw Error(ST_INVALID_ARGUMENT, "encrypted1 and encrypted2 parameter mismatch");
    const int L = a.limbs, sa = a.size, sb = b.size, ds = sa + sb - 1;
    const u64 N = c.N, pw = poly_words(c, L);
    if (!out.data || out.bstride < (u64)ds * pw) throw Error(ST_INVALID_ARGUMENT, "destination batch stride too small for the result size");
    if (ds > 16) throw Error(ST_INVALID_ARGUMENT, "invalid size"); // Ciphertext::resize beyond SEAL_CIPHERTEXT_SIZE_MAX (src/ciphertext.h:441, defines.h:34)
    LimbMap qmap = c.ct_map(L);
    c.arena.begin(s);
    c.arena.reserve(scratch_multiply(sa, sb, L, batch));
    double new_scale = a.scale;
    u64 new_cf = a.cf;
    const bool same = (a.data == b.data && a.bstride == b.bstride && sa == sb);
    // a batch of ONE has no stride: whatever capacity the ciphertext's allocation has (the host mirrors keep room for three polynomials), its
    // polynomials are dense -- the single-ciphertext calls of the reference's interface take the fused, copy-free paths too
    auto dense = [&](u64 bstride, u64 words) { return batch == 1 || bstride == words; };

    if (c.scheme == SCHEME_BFV) {
        if (a.ntt || b.ntt) throw Error(ST_INVALID_ARGUMENT, "encrypted1 or encrypted2 cannot be in NTT form");
        const Level &lv = c.level(L);
        const int nb = (int)lv.rns.Bsk.size();
        const u64 bw = (u64)nb * N;
        const int sp = sa + sb;
        u64 *xq = c.arena.take(batch * sp * pw), *xb = c.arena.take(batch * sp * bw);
        u64 *dq = c.arena.take(batch * ds * pw), *db = c.arena.take(batch * ds * bw);
        // BEHZ steps (1)-(3): extend q -> Bsk, forward NTT in both bases
        const LimbMap bmap = c.ids_map(lv.bsk_ids);
        // A SMALL product (one ciphertext, a few small ones) runs both bases through the same launches: the scratch holds q and B_sk limbs of a polynomial
        // behind each other ([poly][L + |Bsk|][N]), the forward pass takes the q limbs from the operands and the B_sk limbs from the extension, and
        // transform + tensor, the inverse transform and the floor step each see ONE row set -- 7 launches instead of 13, each twice as wide.
        // The large batch keeps the per-base launches.
        const bool merged = dense(a.bstride, (u64)sa * pw) && dense(b.bstride, (u64)sb * pw) && sa == 2 && sb == 2 && ntt2_tensor_supported(c.logn) && tensor_fused() &&
                            L + nb <= 64 && take_merged(c, batch * sp * (u64)(L + nb));
        bool all_done = false;
        if (merged) {
            const u64 mw = (u64)(L + nb) * N; // words of one polynomial in both bases
            std::vector<uint8_t> ids;
            for (int l = 0; l < L; l++) ids.push_back((uint8_t)l);
            ids.insert(ids.end(), lv.bsk_ids.begin(), lv.bsk_ids.end());
            const LimbMap mmap = c.ids_map(ids);
            u64 *X = xq, *D = dq; // the arena carved xq | xb | dq | db back to back: X spans the first two, D the last two
            if (xb != xq + batch * sp * pw || db != dq + batch * ds * pw) throw Error(ST_LOGIC_ERROR, "multiply: the merged scratch is not contiguous (arena alignment changed?)");
            u64 *X_b = same ? X : X + batch * sa * mw;
            if (same || batch * sp > 65535) {
                launch_behz_extend(a.data, pw, X + (u64)L * N, mw, c.d_desc, *lv.behz, N, batch * sa, s);
                if (!same) launch_behz_extend(b.data, pw, X_b + (u64)L * N, mw, c.d_desc, *lv.behz, N, batch * sb, s);
            } else launch_behz_extend(a.data, pw, X + (u64)L * N, mw, c.d_desc, *lv.behz, N, batch * sp, s, b.data, batch * sa); // both operands: X_b follows X
            launch_ntt2_tensor(X, a.data, X_b, b.data, D, c.d_desc, mmap, batch, c.logn, s, (unsigned)L);
            const PrimeDesc *idesc = behz_floor_prescaled(*lv.behz) ? lv.behz->floor_desc : c.d_desc;
            launch_ntt(D, idesc, mmap, batch * ds * (u64)(L + nb), c.logn, true, s);
            u64 *res = dense(out.bstride, (u64)ds * pw) ? out.data : X; // X is dead by now and large enough
            launch_behz_floor_sk(D, mw, D + (u64)L * N, mw, res, pw, c.d_desc, *lv.behz, N, batch * ds, s);
            if (res != out.data) launch_copy_strided(res, ds * pw, out.data, out.bstride, ds * pw, batch, s);
            all_done = true;
        } else if (dense(a.bstride, (u64)sa * pw) && dense(b.bstride, (u64)sb * pw)) {
            // dense operands are consumed in place: the extension reads them directly and the first NTT pass reads them
            // as its out-of-place source, so no staging copy is made.  Scratch layout: [a-part | b-part] in each base.
            // squaring (same operand twice): extended and transformed once, the tensor reads it as both factors
            const bool fused = sa == 2 && sb == 2 && ntt2_tensor_supported(c.logn) && tensor_fused();
            const bool once = same && fused;
            u64 *xq_a = xq, *xq_b = once ? xq : xq + batch * sa * pw, *xb_a = xb, *xb_b = once ? xb : xb + batch * sa * bw;
            launch_behz_extend(a.data, pw, xb_a, bw, c.d_desc, *lv.behz, N, batch * sa, s);
            if (!once) launch_behz_extend(b.data, pw, xb_b, bw, c.d_desc, *lv.behz, N, batch * sb, s);
            if (fused) {
                // (3)+(4) in one pass pair per base: the second NTT pass keeps a0, a1, b0, b1 in registers and stores d0, d1, d2
                launch_ntt2_tensor(xq_a, a.data, xq_b, b.data, dq, c.d_desc, qmap, batch, c.logn, s);
                launch_ntt2_tensor(xb_a, nullptr, xb_b, nullptr, db, c.d_desc, bmap, batch, c.logn, s);
            } else {
                launch_ntt_from(xq_a, a.data, c.d_desc, qmap, batch * sa * L, c.logn, s);
                launch_ntt_from(xq_b, b.data, c.d_desc, qmap, batch * sb * L, c.logn, s);
                launch_ntt(xb, c.d_desc, bmap, batch * sp * nb, c.logn, false, s);
                // (4) tensor in both bases
                launch_tensor(sa, sb, xq_a, xq_b, dq, sa * pw, sb * pw, c.d_desc, qmap, c.logn, L, batch, s);
                launch_tensor(sa, sb, xb_a, xb_b, db, sa * bw, sb * bw, c.d_desc, bmap, c.logn, nb, batch, s);
            }
        } else {
            launch_copy_strided(a.data, a.bstride, xq, sp * pw, sa * pw, batch, s);
            launch_copy_strided(b.data, b.bstride, xq + sa * pw, sp * pw, sb * pw, batch, s);
            launch_behz_extend(xq, pw, xb, bw, c.d_desc, *lv.behz, N, batch * sp, s);
            launch_ntt(xq, c.d_desc, qmap, batch * sp * L, c.logn, false, s);
            launch_ntt(xb, c.d_desc, bmap, batch * sp * nb, c.logn, false, s);
            // (4) tensor in both bases
            launch_tensor(sa, sb, xq, xq + sa * pw, dq, sp * pw, sp * pw, c.d_desc, qmap, c.logn, L, batch, s);
            launch_tensor(sa, sb, xb, xb + sa * bw, db, sp * bw, sp * bw, c.d_desc, bmap, c.logn, nb, batch, s);
        }
        // (5) inverse NTT; when the floor kernel takes pre-scaled inputs, the factors of step (6) ride on the N^-1 constants
        if (!all_done) {
            const PrimeDesc *idesc = behz_floor_prescaled(*lv.behz) ? lv.behz->floor_desc : c.d_desc;
            launch_ntt(dq, idesc, qmap, batch * ds * L, c.logn, true, s);
            launch_ntt(db, idesc, c.ids_map(lv.bsk_ids), batch * ds * nb, c.logn, true, s);
        }
        // (6)-(8) multiply by t, floor, Shenoy-Kumaresan back to base q
        if (all_done) {
        } else if (dense(out.bstride, (u64)ds * pw)) {
            launch_behz_floor_sk(dq, pw, db, bw, out.data, pw, c.d_desc, *lv.behz, N, batch * ds, s);
        } else {
            u64 *res = xq; // xq is dead by now and large enough (sp >= ds)
            launch_behz_floor_sk(dq, pw, db, bw, res, pw, c.d_desc, *lv.behz, N, batch * ds, s);
            launch_copy_strided(res, ds * pw, out.data, out.bstride, ds * pw, batch, s);
        }
    } else if (c.scheme == SCHEME_CKKS) {
        if (!(a.ntt && b.ntt)) throw Error(ST_INVALID_ARGUMENT, "encrypted1 or encrypted2 must be in NTT form");
        new_scale = a.scale * b.scale;
        if (!scale_ok(new_scale, L)) throw Error(ST_INVALID_ARGUMENT, "scale out of bounds");
        if (out.data != a.data && out.data != b.data && dense(out.bstride, (u64)ds * pw)) {
            // a fresh dense destination: the tensor writes it directly
            launch_tensor(sa, sb, a.data, b.data, out.data, a.bstride, b.bstride, c.d_desc, qmap, c.logn, L, batch, s);
        } else {
            u64 *d = c.arena.take(batch * ds * pw);
            launch_tensor(sa, sb, a.data, b.data, d, a.bstride, b.bstride, c.d_desc, qmap, c.logn, L, batch, s);
            launch_copy_strided(d, ds * pw, out.data, out.bstride, ds * pw, batch, s);
        }
    } else {
        if (a.ntt || b.ntt) throw Error(ST_INVALID_ARGUMENT, "encryped1 or encrypted2 must be not in NTT form");
        // BGV (evaluator_cuda.cu:463-500): NTT -> tensor -> INTT.  Dense size-2 operands go through the fused transform + tensor
        // pass pair (consumed in place, like the q base of the BFV path); a fresh dense destination is written directly.
        const int sp = sa + sb;
        const bool direct_out = out.data != a.data && out.data != b.data && dense(out.bstride, (u64)ds * pw);
        u64 *d = direct_out ? out.data : c.arena.take(batch * ds * pw);
        if (dense(a.bstride, (u64)sa * pw) && dense(b.bstride, (u64)sb * pw) && sa == 2 && sb == 2 && ntt2_tensor_supported(c.logn) && tensor_fused()) {
            u64 *xa = c.arena.take(batch * sa * pw), *xb = same ? xa : c.arena.take(batch * sb * pw);
            launch_ntt2_tensor(xa, a.data, xb, b.data, d, c.d_desc, qmap, batch, c.logn, s);
        } else {
            u64 *x = c.arena.take(batch * sp * pw);
            launch_copy_strided(a.data, a.bstride, x, sp * pw, sa * pw, batch, s);
            launch_copy_strided(b.data, b.bstride, x + sa * pw, sp * pw, sb * pw, batch, s);
            launch_ntt(x, c.d_desc, qmap, batch * sp * L, c.logn, false, s);
            launch_tensor(sa, sb, x, x + sa * pw, d, sp * pw, sp * pw, c.d_desc, qmap, c.logn, L, batch, s);
        }
        launch_ntt(d, c.d_desc, qmap, batch * ds * L, c.logn, true, s);
        if (!direct_out) launch_copy_strided(d, ds * pw, out.data, out.bstride, ds * pw, batch, s);
        new_cf = host::mul_mod(a.cf % c.t, b.cf % c.t, c.t);
    }
    out.size = ds;
    out.limbs = L;
    out.ntt = a.ntt;
    out.scale = new_scale;
    out.cf = new_cf;
}

size_t Evaluator::scratch_switch_key(int limbs, u64 batch) const {
    const size_t N = c.N, dl = limbs, rl = limbs + 1;
    return batch * N * (dl /*t_target*/ + rl * dl /*D*/ + 2 * rl /*acc*/ + 2 * dl /*corr*/ + 2 /*last*/) + 512;
}

// switchKeyInplace (evaluator_cuda.cu:1163-1362; CPU src/evaluator.cpp:2310-2653)
void Evaluator::switch_key(CtBatch &ct, const u64 *target, u64 t_bstride, const KsKey &key, u64 batch, hipStream_t s, const u64 *base, u64 base_bstride, int base_polys) {
    check_ct(ct);
    if (!target) throw Error(ST_INVALID_ARGUMENT, "target_iter");
    if (c.K < 2) throw Error(ST_LOGIC_ERROR, "keyswitching is not supported by the context");
    if (!key.data) throw Error(ST_INVALID_ARGUMENT, "kswitch_keys is not valid for encryption parameters");
    if (c.scheme == SCHEME_BFV && ct.ntt) throw Error(ST_INVALID_ARGUMENT, "BFV encrypted cannot be in NTT form");
    if (c.scheme == SCHEME_CKKS && !ct.ntt) throw Error(ST_INVALID_ARGUMENT, "CKKS encrypted must be in NTT form");
    if (c.scheme == SCHEME_BGV && ct.ntt) throw Error(ST_INVALID_ARGUMENT, "BGV encrypted cannot be in NTT form");
    if (ct.size < 2) throw Error(ST_INVALID_ARGUMENT, "encrypted size must be at least 2");
    const u64 N = c.N, dl = ct.limbs, rl = dl + 1, K = c.K;
    const host::RnsLevel &key_rns = c.level((int)K).rns;
    KsArgs a;
    std::memset(&a, 0, sizeof(a));
    a.primes = c.d_desc;
    for (u64 i = 0; i < dl; i++) { a.key_id[i] = (uint8_t)i; a.key_limb[i] = (uint8_t)i; }
    a.key_id[dl] = (uint8_t)(K - 1);
    a.key_limb[dl] = (uint8_t)(K - 1);
    const u64 qk = c.primes[K - 1];
    for (u64 j = 0; j < dl; j++) a.inv_qk[j] = make_shoup(key_rns.inv_q_last_mod_q[j], c.primes[j]);
    a.half = qk >> 1;
    if (c.t) {
        Mod tm = make_mod(c.t);
        a.t_p = tm.p; a.t_cr0 = tm.cr0; a.t_cr1 = tm.cr1;
        a.inv_qk_mod_t = key_rns.inv_q_last_mod_t;
    }
    a.logn = c.logn; a.dl = dl; a.K = K; a.batch = batch;

    c.arena.begin(s);
    c.arena.reserve(scratch_switch_key((int)dl, batch));
    u64 *D = c.arena.take(batch * rl * dl * N);
    u64 *acc = c.arena.take(batch * 2 * rl * N);
    std::vector<uint8_t> out_ids(a.key_id, a.key_id + rl);

    const u64 *coeff_target = target;
    u64 ct_tb = t_bstride;
    if (c.scheme == SCHEME_CKKS) { // bring the target to coefficient form (evaluator_cuda.cu:1215-1216)
        u64 *tt = c.arena.take(batch * dl * N);
        const LimbMap tmap = c.ct_map((int)dl);
        if (ntt1_supported(c.logn, tmap, batch * dl, 3)) { // out of place: the single-pass inverse reads the strided target itself
            launch_ntt1(tt, target, c.d_desc, tmap, batch * dl, c.logn, true, s, ~0ull, nullptr, nullptr, t_bstride);
        } else {
            launch_copy_strided(target, t_bstride, tt, dl * N, dl * N, batch, s);
            launch_ntt(tt, c.d_desc, tmap, batch * dl, c.logn, true, s);
        }
        coeff_target = tt;
        ct_tb = dl * N;
    }
    // the digits are canonical residues of the ciphertext primes: an output prime p with 8p above the largest of them needs no reduction
    u64 src_bound = 0;
    for (u64 j = 0; j < dl; j++) src_bound = std::max(src_bound, c.primes[j]);
    // decompose + extend every limb to every output prime, ONE batched NTT over all (L+1)*L rows, inner product with the key
    const u64 *mac_target = c.scheme == SCHEME_CKKS ? target : nullptr;
    if (ntt2_ks_mac_supported(c.logn) && ks_fused()) {
        // fused: the first NTT pass reads the target and reduces it modulo each output prime on the fly; the second pass keeps
        // the transforms in registers and accumulates them against the key -- the expanded digits are never written back
        // lazy accumulation is exact while dl * (bound of the lazy transform) * p < 2^128 for every output prime (CKKS replaces one
        // operand by a canonical value): the bound is 8p with guarded butterflies, 59p with the guard-free ones (primes below 2^58)
        const LimbMap ks_map = c.ids_map(out_ids, (uint32_t)dl);
        bool lazy = true;
        for (u64 i = 0; i < rl; i++) {
            const long double p = (long double)c.primes[out_ids[i]];
            const long double bound = ((ks_map.lean >> i) & 1) ? 59.0L : 8.0L;
            lazy = lazy && (long double)dl * bound * p * p < 3.0e38L; // 2^128 = 3.4e38
        }
        launch_ntt2_ks_mac(D, coeff_target, ct_tb, c.d_desc, ks_map, batch * rl * dl, c.logn, key.data, acc, a.key_limb, (unsigned)K,
                           mac_target, t_bstride, lazy, src_bound, s);
    } else {
        if (ntt2_supported(c.logn)) {
            launch_ntt2(D, coeff_target, ct_tb, true, c.d_desc, c.ids_map(out_ids, (uint32_t)dl), batch * rl * dl, c.logn, false, s, false, src_bound);
        } else {
            launch_ks_expand(coeff_target, ct_tb, D, a, s);
            launch_ntt(D, c.d_desc, c.ids_map(out_ids, (uint32_t)dl), batch * rl * dl, c.logn, false, s);
        }
        // inner products with the key (128-bit lazy accumulation, one reduction per output)
        launch_ks_mac(D, key.data, mac_target, t_bstride, acc, a, s);
    }

    const LimbMap amap_md = c.ids_map(out_ids);
    bool md_primes33 = true; // the single-pass epilogue's lazy reduction (lite_reduce4) wants primes of at least 33 bits
    for (u64 j = 0; j < dl; j++) md_primes33 = md_primes33 && c.primes[j] >= (u64(1) << 33);
    const bool md_single = c.scheme == SCHEME_BFV && c.d_desc_md && md_primes33 && ntt1_supported(c.logn, amap_md, batch * 2 * rl, 1) && ks_moddown_fused();
    const bool md_two_pass = c.scheme != SCHEME_CKKS && !md_single && c.d_desc_md && ntt2_supported(c.logn) && ks_moddown_fused();
    const bool ckks_single = c.scheme == SCHEME_CKKS && c.d_inv_qk && corr_fused() && primes_at_least_33_bits(c, (int)dl) &&
                             ntt1_supported(c.logn, c.ct_map((int)dl), batch * 2 * dl, 2);
    if (base && !md_two_pass && !md_single && !ckks_single) { // the fused epilogues take (base, 0) directly; the element-wise forms accumulate onto what ct holds
        if (base != ct.data) launch_copy_strided(base, base_bstride, ct.data, ct.bstride, (u64)base_polys * dl * N, batch, s);
        if (base_polys < 2) launch_zero_strided(ct.data + dl * N, ct.bstride, dl * N, batch, s);
        base = nullptr;
    }
    if (c.scheme == SCHEME_CKKS) {
        // special-prime limb -> coefficient form, correction polynomial -> NTT form, combine
        u64 *last = c.arena.take(batch * 2 * N), *corr = c.arena.take(batch * 2 * dl * N);
        launch_gather_limb(acc, last, c.logn, rl * N, dl, batch * 2, s);
        launch_ntt(last, c.d_desc, c.single_map((int)K - 1), batch * 2, c.logn, true, s);
        const LimbMap cmap = c.ct_map((int)dl);
        if (ckks_single) {
            // the correction is built, transformed and combined by ONE single-pass transform (Ntt1Corr): no corr buffer, no element-wise kernels
            Ntt1Corr cr{last, acc, rl * N, ct.data, ct.bstride, dl * N, 2, c.d_inv_qk, qk, a.half, true};
            cr.base = base;
            cr.base_gstride = base_bstride;
            cr.base_polys = base_polys;
            launch_ntt1(nullptr, nullptr, c.d_desc, cmap, batch * 2 * dl, c.logn, false, s, ~0ull, nullptr, &cr);
        } else {
            launch_ks_ckks_corr(last, corr, a, s);
            launch_ntt(corr, c.d_desc, cmap, batch * 2 * dl, c.logn, false, s);
            launch_ks_ckks_combine(acc, corr, ct.data, ct.bstride, a, s);
        }
    } else {
        const LimbMap &amap = amap_md;
        if (md_single) {
            // single-pass inverse: the special limb first, then the data limbs with the mod-down as their store epilogue (no acc round trip,
            // no separate memory-bound kernel)
            launch_ntt1(acc, nullptr, c.d_desc, amap, batch * 2 * rl, c.logn, true, s, u64(1) << dl);
            Ntt1ModDown md{ct.data, ct.bstride, dl, qk, a.half};
            md.base = base;
            md.base_bstride = base_bstride;
            md.base_polys = base_polys;
            launch_ntt1(acc, nullptr, c.d_desc_md, amap, batch * 2 * rl, c.logn, true, s, (u64(1) << dl) - 1, &md);
        } else if (md_two_pass) {
            // two-pass inverse, same shape: the special limb first, then the data limbs with the mod-down (BFV or BGV) as the last pass's epilogue
            // the first pass is the same for every slot (N^-1 and qk^-1 ride on the last inverse stage): a small launch runs it over the special limb
            // and the data limbs at once, then the two last passes -- three kernels instead of four, the tiny special-limb launch (2 batch rows) half gone
            const bool one_first_pass = take_merged(c, batch * 2 * rl);
            if (one_first_pass) {
                launch_ntt2_slots(acc, nullptr, 0, false, c.d_desc, amap, batch * 2 * rl, c.logn, true, s, false, 0, 0, (unsigned)rl, nullptr, 1);
                // the second passes plan their FP64 reductions from the bound the SHARED first pass left: the largest prime of slots 0 .. rl, not of their own range
                launch_ntt2_slots(acc, nullptr, 0, false, c.d_desc, amap, batch * 2 * rl, c.logn, true, s, false, 0, (unsigned)dl, 1, nullptr, 2, 0, (unsigned)rl);
            } else {
                launch_ntt2_slots(acc, nullptr, 0, false, c.d_desc, amap, batch * 2 * rl, c.logn, true, s, false, 0, (unsigned)dl, 1, nullptr);
            }
            u64 *share = nullptr;
            if (c.scheme == SCHEME_BGV) { // what the special limb takes out of every data limb, once per coefficient (the CKKS part of the reservation is free)
                share = c.arena.take(batch * 4 * N);
                launch_ks_bgv_share(acc, share, a, s);
            }
            Ntt2ModDown md{c.scheme == SCHEME_BFV ? 0 : 2, ct.data, ct.bstride, (unsigned)dl, qk, a.half, share};
            md.base = base;
            md.base_bstride = base_bstride;
            md.base_polys = base_polys;
            launch_ntt2_slots(acc, nullptr, 0, false, c.d_desc_md, amap, batch * 2 * rl, c.logn, true, s, false, 0, 0, (unsigned)dl, &md, one_first_pass ? 2u : 3u, 0,
                              one_first_pass ? (unsigned)rl : 0u);
        } else {
            launch_ntt(acc, c.d_desc, amap, batch * 2 * rl, c.logn, true, s);
            launch_ks_moddown(c.scheme == SCHEME_BFV ? 0 : 2, acc, ct.data, ct.bstride, a, s);
        }
    }
}

// relinearizeInternal (evaluator_cuda.cu:703-744), destination size 2, from any size up to SEAL_CIPHERTEXT_SIZE_MAX.  Exactly as
// the reference does it: `encrypted_iter` is set to the LAST polynomial once (:729) and not advanced, so each of the size - 2
// steps switches that same polynomial, step i with the key of index getIndex(size - 1 - i) = size - 3 - i (:731-735); the
// polynomials 2 .. size-2 are dropped by the final resize (:742).  Bit-exact parity means reproducing this, not "fixing" it.
void Evaluator::relinearize(CtBatch &ct, const KsKey *keys, int n_keys, u64 batch, hipStream_t s) {
    check_ct(ct);
    if (ct.size == 2) return;
    if (ct.size < 2 || ct.size > 16) throw Error(ST_INVALID_ARGUMENT, "destination_size must be at least 2 and less than or equal to current count");
    const int size = ct.size;
    if (n_keys < size - 2) throw Error(ST_INVALID_ARGUMENT, "not enough relinearization keys");
    for (int i = 0; i < size - 2; i++)
        if (!keys[i].data) throw Error(ST_INVALID_ARGUMENT, "not enough relinearization keys");
    const u64 pw = poly_words(c, ct.limbs);
    for (int i = 0; i < size - 2; i++) switch_key(ct, ct.data + (u64)(size - 1) * pw, ct.bstride, keys[size - 3 - i], batch, s);
    ct.size = 2;
}

// relinearize(encrypted, relin_keys, destination) (evaluator_cuda.cuh: copy + relinearizeInplace).  From size 3 the key switch reads c2 of
// the operand as its target and accumulates onto (c0, c1) of the operand -- written to `out` by the mod-down epilogue -- so the copy the
// reference makes does not happen; other sizes take the reference's route.
void Evaluator::relinearize_to(const CtBatch &in, CtBatch &out, const KsKey *keys, int n_keys, u64 batch, hipStream_t s) {
    check_ct(in);
    const u64 pw = poly_words(c, in.limbs);
    if (!out.data || out.data == in.data) throw Error(ST_INVALID_ARGUMENT, "relinearize: destination must be a distinct buffer");
    if (in.size != 3 || out.bstride < 2 * pw) { // general sizes: the reference's copy, then in place (needs room for every polynomial)
        if (out.bstride < (u64)in.size * pw) throw Error(ST_INVALID_ARGUMENT, "relinearize: destination too small");
        launch_copy_strided(in.data, in.bstride, out.data, out.bstride, (u64)in.size * pw, batch, s);
        const u64 *d = out.data;
        const u64 bs = out.bstride;
        out = in;
        out.data = const_cast<u64 *>(d);
        out.bstride = bs;
        relinearize(out, keys, n_keys, batch, s);
        return;
    }
    if (n_keys < 1 || !keys[0].data) throw Error(ST_INVALID_ARGUMENT, "not enough relinearization keys");
    u64 *d = out.data;
    const u64 bs = out.bstride;
    out = in;
    out.data = d;
    out.bstride = bs;
    out.size = 2;
    switch_key(out, in.data + 2 * pw, in.bstride, keys[0], batch, s, in.data, in.bstride, 2);
}
void Evaluator::relinearize(CtBatch &ct, const KsKey &key, u64 batch, hipStream_t s) { relinearize(ct, &key, 1, batch, s); }

// modSwitchScaleToNext (evaluator_cuda.cu:749-824)
void Evaluator::mod_switch_scale(const CtBatch &in, CtBatch &out, u64 batch, hipStream_t s) {
    check_ct(in);
    const int L = in.limbs, nl = L - 1;
    if (L < 2 || !c.is_data_level(nl)) throw Error(ST_INVALID_ARGUMENT, "end of modulus switching chain reached");
    if (c.scheme == SCHEME_BFV && in.ntt) throw Error(ST_INVALID_ARGUMENT, "BFV encrypted cannot be in NTT form");
    if (c.scheme == SCHEME_CKKS && !in.ntt) throw Error(ST_INVALID_ARGUMENT, "CKKS encrypted must be in NTT form");
    if (c.scheme == SCHEME_BGV && in.ntt) throw Error(ST_INVALID_ARGUMENT, "BGV encrypted cannot be in NTT form");
    const u64 N = c.N, pw = poly_words(c, L), npw = poly_words(c, nl);
    if (!out.data || out.bstride < (u64)in.size * npw) throw Error(ST_INVALID_ARGUMENT, "destination batch stride too small for the result size");
    const host::RnsLevel &r = c.level(L).rns;
    ModSwitchArgs a;
    std::memset(&a, 0, sizeof(a));
    a.primes = c.d_desc;
    a.map = c.ct_map(L);
    for (int l = 0; l < nl; l++) a.inv_qlast[l] = make_shoup(r.inv_q_last_mod_q[l], c.primes[l]);
    a.half = c.primes[L - 1] >> 1;
    if (c.t) {
        Mod tm = make_mod(c.t);
        a.t_p = tm.p; a.t_cr0 = tm.cr0; a.t_cr1 = tm.cr1;
        a.inv_qlast_mod_t = r.inv_q_last_mod_t;
    }
    a.logn = c.logn; a.limbs = L; a.polys = batch * in.size;

    c.arena.begin(s);
    const bool dense_in = batch == 1 || in.bstride == (u64)in.size * pw, dense_out = batch == 1 || out.bstride == (u64)in.size * npw; // one item has no stride
    {   // everything this op carves, reserved up front: take() can only grow an EMPTY arena (a rescale as the first op on a fresh
        // context, or at a larger batch than the ops before it, used to fail with "scratch arena exhausted inside an op")
        size_t need = 4 * 32;
        need += batch * in.size * pw; // staging copy of a strided or overlapping input
        if (!dense_out) need += batch * in.size * npw;
        if (c.scheme == SCHEME_CKKS) need += batch * in.size * N + batch * in.size * npw;
        c.arena.reserve(need);
    }
    const u64 *src = in.data;
    // CKKS through the fused correction transform reads a strided batch (a relinearized ciphertext keeps its three-polynomial stride) as it lies
    const bool ckks_fused = c.scheme == SCHEME_CKKS && c.level(L).d_inv_qlast && corr_fused() && primes_at_least_33_bits(c, nl) &&
                            ntt1_supported(c.logn, c.ct_map(nl), batch * in.size * nl, 2);
    // the output ranges [out.data + b out.bstride, + size npw) must not overlap what a later row still reads: when the two batches share memory
    // (a direct C-ABI caller rescaling a strided batch onto itself) the input is staged first, as every strided input was before the fused path
    bool overlaps = false;
    if (batch) { // an empty batch touches nothing (and (batch - 1) * stride would wrap)
        const u64 *in_end = in.data + (batch - 1) * in.bstride + (u64)in.size * pw, *out_end = out.data + (batch - 1) * out.bstride + (u64)in.size * npw;
        overlaps = out.data < in_end && in.data < out_end;
    }
    bool staged = false;
    if ((!dense_in && !ckks_fused) || overlaps) {
        staged = true;
        u64 *tmp = c.arena.take(batch * in.size * pw);
        launch_copy_strided(in.data, in.bstride, tmp, in.size * pw, in.size * pw, batch, s);
        src = tmp;
    }
    u64 *dst = dense_out ? out.data : c.arena.take(batch * in.size * npw);
    if (dst == src) throw Error(ST_INVALID_ARGUMENT, "mod switch cannot run in place");
    if (c.scheme == SCHEME_CKKS) {
        u64 *last = c.arena.take(batch * in.size * N), *corr = c.arena.take(batch * in.size * npw);
        if (ckks_fused && !dense_in && !staged) launch_gather_limb(src, last, c.logn, pw, (u64)nl, batch * in.size, s, (u64)in.size, in.bstride);
        else launch_gather_limb(src, last, c.logn, pw, (u64)nl, batch * in.size, s);
        launch_ntt(last, c.d_desc, c.single_map(L - 1), batch * in.size, c.logn, true, s);
        const LimbMap cmap = c.ct_map(nl);
        const Level &lvl = c.level(L);
        if (ckks_fused) {
            Ntt1Corr cr{last, src, (u64)pw, dst, (u64)in.size * npw, (u64)npw, (unsigned)in.size, lvl.d_inv_qlast, c.primes[L - 1], a.half, false};
            if (!dense_in && !staged) cr.in_gstride = in.bstride;
            launch_ntt1(nullptr, nullptr, c.d_desc, cmap, batch * in.size * nl, c.logn, false, s, ~0ull, nullptr, &cr);
        } else {
            launch_rescale_stepA(last, N, corr, a, s);
            launch_ntt(corr, c.d_desc, cmap, batch * in.size * nl, c.logn, false, s);
            launch_rescale_stepB(src, corr, dst, a, s);
        }
    } else {
        launch_modswitch(c.scheme == SCHEME_BFV ? 0 : 2, src, dst, a, s);
    }
    if (!dense_out) launch_copy_strided(dst, in.size * npw, out.data, out.bstride, in.size * npw, batch, s);
    out.size = in.size;
    out.limbs = nl;
    out.ntt = in.ntt;
    out.scale = in.scale;
    out.cf = in.cf;
    if (c.scheme == SCHEME_CKKS) out.scale = in.scale / (double)c.primes[L - 1];
    else if (c.scheme == SCHEME_BGV) out.cf = host::mul_mod(in.cf % c.t, r.inv_q_last_mod_t, c.t);
}

// modSwitchToNext (evaluator_cuda.cu:926-980): CKKS drops the last limb, BFV/BGV scale
void Evaluator::mod_switch_to_next(const CtBatch &in, CtBatch &out, u64 batch, hipStream_t s) {
    if (c.scheme != SCHEME_CKKS) { mod_switch_scale(in, out, batch, s); return; }
    check_ct(in);
    const int L = in.limbs, nl = L - 1;
    if (L < 2 || !c.is_data_level(nl)) throw Error(ST_INVALID_ARGUMENT, "end of modulus switching chain reached");
    if (!in.ntt) throw Error(ST_INVALID_ARGUMENT, "CKKS encrypted must be in NTT form");
    if (!scale_ok(in.scale, nl)) throw Error(ST_INVALID_ARGUMENT, "scale out of bounds");
    const u64 pw = poly_words(c, L), npw = poly_words(c, nl);
    if (!out.data || out.bstride < (u64)in.size * npw) throw Error(ST_INVALID_ARGUMENT, "destination batch stride too small for the result size");
    if (in.bstride == (u64)in.size * pw && out.bstride == (u64)in.size * npw) {
        launch_drop_last(in.data, out.data, c.logn, L, batch * in.size, s);
    } else {
        for (u64 b = 0; b < batch; b++) launch_drop_last(in.data + b * in.bstride, out.data + b * out.bstride, c.logn, L, in.size, s);
    }
    out.size = in.size; out.limbs = nl; out.ntt = in.ntt; out.scale = in.scale; out.cf = in.cf;
}

void Evaluator::rescale_to_next(const CtBatch &in, CtBatch &out, u64 batch, hipStream_t s) { // evaluator_cuda.cu:1380-1404
    if (c.scheme != SCHEME_CKKS) throw Error(ST_INVALID_ARGUMENT, "unsupported operation for scheme type");
    mod_switch_scale(in, out, batch, s);
}

// applyGaloisInplace (evaluator_cuda.cu:2024-2117)
void Evaluator::apply_galois(CtBatch &ct, uint32_t elt, const KsKey &key, u64 batch, hipStream_t s) {
    check_ct(ct);
    if (!key.data) throw Error(ST_INVALID_ARGUMENT, "Galois key not present");
    if (!(elt & 1) || elt >= 2 * c.N) throw Error(ST_INVALID_ARGUMENT, "Galois element is not valid");
    if (ct.size > 2) throw Error(ST_INVALID_ARGUMENT, "encrypted size must be 2");
    const u64 pw = poly_words(c, ct.limbs);
    const int L = ct.limbs;
    // sigma(c0), sigma(c1) into dense temporaries; they must outlive switch_key's own scratch, so they are
    // carved from the arena tail after reserving the key-switch working set
    c.arena.begin(s);
    const size_t ks = scratch_switch_key(L, batch);
    c.arena.reserve(ks + 2 * batch * pw + 128);
    (void)c.arena.take(ks);
    u64 *t0 = c.arena.take(batch * pw), *t1 = c.arena.take(batch * pw);
    LimbMap map = c.ct_map(L);
    const bool ntt_form = c.scheme == SCHEME_CKKS;
    launch_galois(ntt_form, ct.data, ct.bstride, t0, pw, c.d_desc, map, c.logn, elt, L, batch, s);
    launch_galois(ntt_form, ct.data + pw, ct.bstride, t1, pw, c.d_desc, map, c.logn, elt, L, batch, s);
    // switch_key resets the arena but never grows it now, so t0 and t1 (beyond its working set) stay intact; it takes ct as (t0, 0)
    switch_key(ct, t1, pw, key, batch, s, t0, pw);
}

void Evaluator::transform_to_ntt(CtBatch &ct, u64 batch, hipStream_t s) { // evaluator_cuda.cu:1950-1985
    check_ct(ct);
    if (ct.ntt) throw Error(ST_INVALID_ARGUMENT, "encrypted is already in NTT form");
    const u64 words = (u64)ct.size * poly_words(c, ct.limbs);
    if (ct.bstride == words) launch_ntt(ct.data, c.d_desc, c.ct_map(ct.limbs), batch * ct.size * ct.limbs, c.logn, false, s);
    else for (u64 b = 0; b < batch; b++) launch_ntt(ct.data + b * ct.bstride, c.d_desc, c.ct_map(ct.limbs), (u64)ct.size * ct.limbs, c.logn, false, s);
    ct.ntt = true;
}
void Evaluator::transform_from_ntt(CtBatch &ct, u64 batch, hipStream_t s) { // evaluator_cuda.cu:1987-2021
    check_ct(ct);
    if (!ct.ntt) throw Error(ST_INVALID_ARGUMENT, "encrypted_ntt is not in NTT form");
    const u64 words = (u64)ct.size * poly_words(c, ct.limbs);
    if (ct.bstride == words) launch_ntt(ct.data, c.d_desc, c.ct_map(ct.limbs), batch * ct.size * ct.limbs, c.logn, true, s);
    else for (u64 b = 0; b < batch; b++) launch_ntt(ct.data + b * ct.bstride, c.d_desc, c.ct_map(ct.limbs), (u64)ct.size * ct.limbs, c.logn, true, s);
    ct.ntt = false;
}
// multiplyPlainNtt (evaluator_cuda.cu:1824-1863): one plaintext [limbs][N] (NTT form) for the whole batch
void Evaluator::multiply_plain_ntt(CtBatch &ct, const u64 *plain, double plain_scale, u64 batch, hipStream_t s) {
    check_ct(ct);
    if (!ct.ntt) throw Error(ST_INVALID_ARGUMENT, "encrypted_ntt is not in NTT form");
    if (!plain) throw Error(ST_INVALID_ARGUMENT, "plain_ntt is not valid for encryption parameters");
    double new_scale = ct.scale * plain_scale;
    if (c.scheme == SCHEME_CKKS && !scale_ok(new_scale, ct.limbs)) throw Error(ST_INVALID_ARGUMENT, "scale out of bounds");
    const u64 words = (u64)ct.size * poly_words(c, ct.limbs);
    LimbMap map = c.ct_map(ct.limbs);
    if (ct.bstride == words) launch_mul_plain(ct.data, plain, c.d_desc, map, c.logn, ct.limbs, batch * ct.size * ct.limbs, s);
    else for (u64 b = 0; b < batch; b++) launch_mul_plain(ct.data + b * ct.bstride, plain, c.d_desc, map, c.logn, ct.limbs, (u64)ct.size * ct.limbs, s);
    ct.scale = new_scale;
}

void Evaluator::multiply_plain_accumulate(const CtBatch *cts, const u64 *const *plains, int count, double plain_scale, CtBatch &out, u64 batch, hipStream_t s) {
    if (count < 1 || count > 16) throw Error(ST_INVALID_ARGUMENT, "multiply_plain_accumulate takes 1 to 16 products");
    const CtBatch &a0 = cts[0];
    check_ct(a0);
    const u64 words = (u64)a0.size * poly_words(c, a0.limbs);
    MulPlainAccArgs x;
    std::memset(&x, 0, sizeof(x));
    x.count = count;
    for (int i = 0; i < count; i++) {
        const CtBatch &a = cts[i];
        check_ct(a);
        if (!a.ntt) throw Error(ST_INVALID_ARGUMENT, "encrypted_ntt is not in NTT form");
        if (!plains[i]) throw Error(ST_INVALID_ARGUMENT, "plain_ntt is not valid for encryption parameters");
        if (a.size != a0.size || a.limbs != a0.limbs) throw Error(ST_INVALID_ARGUMENT, "encrypted1 and encrypted2 parameter mismatch"); // what addInplace would say
        if (a.scale != a0.scale) throw Error(ST_INVALID_ARGUMENT, "scale mismatch");
        if (a.data == out.data) throw Error(ST_INVALID_ARGUMENT, "the destination cannot be one of the operands");
        x.ct[i] = a.data; x.ct_bstride[i] = a.bstride; x.plain[i] = plains[i];
    }
    const double new_scale = a0.scale * plain_scale;
    if (c.scheme == SCHEME_CKKS && !scale_ok(new_scale, a0.limbs)) throw Error(ST_INVALID_ARGUMENT, "scale out of bounds");
    if (!out.data || out.bstride < words) throw Error(ST_INVALID_ARGUMENT, "destination batch stride too small for the result size");
    launch_mul_plain_acc(x, out.data, out.bstride, c.d_desc, c.ct_map(a0.limbs), c.logn, a0.limbs, a0.size, batch, s);
    out.size = a0.size; out.limbs = a0.limbs; out.ntt = true; out.scale = new_scale; out.cf = a0.cf;
}

// ---- plaintext operands ----
static PlainArgs plain_args(Context &c, int limbs, u64 n_coeffs, u64 plain_bstride, u64 items, u64 cf) {
    if (limbs < 1 || limbs > c.K) throw Error(ST_INVALID_ARGUMENT, "parms_id is not valid for the current context");
    PlainArgs a;
    std::memset(&a, 0, sizeof(a));
    a.primes = c.d_desc;
    a.map = c.ct_map(limbs);
    a.logn = c.logn;
    a.limbs = (u64)limbs;
    a.n_coeffs = n_coeffs;
    a.items = items;
    a.plain_bstride = plain_bstride;
    a.cf = cf;
    if (c.t) {
        const Mod tm = make_mod(c.t);
        a.t_p = tm.p; a.t_cr0 = tm.cr0; a.t_cr1 = tm.cr1;
        a.thr = (c.t + 1) >> 1;
        std::vector<u64> q(c.primes.begin(), c.primes.begin() + limbs);
        a.q_mod_t = host::product_mod(q, c.t);
        // Delta_l = floor(q/t) mod q_l = -(q mod t) * t^-1 mod q_l  (context.cpp:307-331 divides the multi-word q)
        for (int l = 0; l < limbs; l++) {
            const Mod m = make_mod(q[l]);
            a.delta[l] = mulmod(negmod(a.q_mod_t % m.p, m.p), host::inv_mod_checked(c.t % m.p, m.p), m);
        }
    }
    return a;
}
// addPlainInplace / subPlainInplace (evaluator_cuda.cu:1654-1720)
void Evaluator::add_plain(CtBatch &ct, const u64 *plain, u64 n_coeffs, u64 plain_bstride, double plain_scale, bool sub, u64 batch, hipStream_t s) {
    check_ct(ct);
    if (!plain) throw Error(ST_INVALID_ARGUMENT, "plain is not valid for encryption parameters");
    if (c.scheme == SCHEME_BFV && ct.ntt) throw Error(ST_INVALID_ARGUMENT, "BFV encrypted cannot be in NTT form");
    if (c.scheme == SCHEME_CKKS && !ct.ntt) throw Error(ST_INVALID_ARGUMENT, "CKKS encrypted must be in NTT form");
    if (c.scheme == SCHEME_BGV && ct.ntt) throw Error(ST_INVALID_ARGUMENT, "BGV encrypted cannot be in NTT form");
    if (c.scheme == SCHEME_CKKS) {
        if (std::fabs(ct.scale - plain_scale) >= std::ldexp(1.0, -23)) throw Error(ST_INVALID_ARGUMENT, "scale mismatch");
        const PlainArgs a = plain_args(c, ct.limbs, c.N, plain_bstride, batch, 1);
        launch_add_plain(1, sub, ct.data, ct.bstride, plain, a, s);
        return;
    }
    if (n_coeffs > c.N) throw Error(ST_INVALID_ARGUMENT, "plain is not valid for encryption parameters");
    const PlainArgs a = plain_args(c, ct.limbs, n_coeffs, plain_bstride, batch, ct.cf);
    launch_add_plain(c.scheme == SCHEME_BFV ? 0 : 2, sub, ct.data, ct.bstride, plain, a, s);
}
// transformToNttInplace(Plaintext, parms_id) (evaluator_cuda.cu:1866-1948): out [count][limbs][N]
void Evaluator::plain_to_ntt(const u64 *plain, u64 n_coeffs, u64 plain_bstride, int limbs, u64 *out, u64 count, hipStream_t s) {
    if (!plain || !out || n_coeffs > c.N || c.scheme == SCHEME_CKKS) throw Error(ST_INVALID_ARGUMENT, "plain is not valid for encryption parameters");
    const PlainArgs a = plain_args(c, limbs, n_coeffs, plain_bstride, count, 1);
    launch_plain_lift(plain, out, a, s);
    launch_ntt(out, c.d_desc, c.ct_map(limbs), count * limbs, c.logn, false, s);
}
// multiplyPlainNormal (evaluator_cuda.cu:1757-1815): generic branch for every plaintext, exactly like the reference's CUDA
// evaluator (its CPU twin short-cuts one-coefficient plaintexts, evaluator.cpp:1816-1867)
void Evaluator::multiply_plain(CtBatch &ct, const u64 *plain, u64 n_coeffs, u64 plain_bstride, u64 batch, hipStream_t s) {
    check_ct(ct);
    if (ct.ntt) throw Error(ST_INVALID_ARGUMENT, "NTT form mismatch");
    if (c.scheme == SCHEME_CKKS) throw Error(ST_INVALID_ARGUMENT, "CKKS encrypted must be in NTT form");
    const u64 items = plain_bstride ? batch : 1, pw = poly_words(c, ct.limbs);
    c.arena.begin(s);
    c.arena.reserve(items * pw);
    u64 *temp = c.arena.take(items * pw);
    plain_to_ntt(plain, n_coeffs, plain_bstride, ct.limbs, temp, items, s);
    transform_to_ntt(ct, batch, s);
    const u64 words = (u64)ct.size * pw;
    const LimbMap map = c.ct_map(ct.limbs);
    const u64 rpi = plain_bstride ? (u64)ct.size * ct.limbs : 0;
    if (ct.bstride == words) launch_mul_plain(ct.data, temp, c.d_desc, map, c.logn, ct.limbs, batch * ct.size * ct.limbs, s, rpi, pw);
    else for (u64 b = 0; b < batch; b++) launch_mul_plain(ct.data + b * ct.bstride, temp + (plain_bstride ? b * pw : 0), c.d_desc, map, c.logn, ct.limbs, (u64)ct.size * ct.limbs, s);
    transform_from_ntt(ct, batch, s);
}

// applyKeySwitchingInplace (evaluator_cuda.cu:1365-1378): c1 is the target, c1 := 0, switch with the given key
void Evaluator::apply_key_switching(CtBatch &ct, const KsKey &key, u64 batch, hipStream_t s) {
    check_ct(ct);
    if (!key.data) throw Error(ST_INVALID_ARGUMENT, "kswitch_keys.data().size() != 1");
    if (ct.size != 2) throw Error(ST_INVALID_ARGUMENT, "encrypted.size() != 2");
    const u64 pw = poly_words(c, ct.limbs);
    // c1 is read as the target by the first stages of the key switch and written only by its last one (the mod-down), which takes the
    // ciphertext as (c0, 0): no copy of c1, no zero fill
    switch_key(ct, ct.data + pw, ct.bstride, key, batch, s, ct.data, ct.bstride, 1);
}
// negacyclicShift (evaluator_cuda.cu:2342-2351): every limb of every polynomial is multiplied by x^shift
void Evaluator::negacyclic_shift(CtBatch &ct, u64 shift, u64 batch, hipStream_t s) {
    check_ct(ct);
    if (shift >= 2 * c.N) throw Error(ST_INVALID_ARGUMENT, "shift");  // x^(N+k) = -x^k: shifts up to 2N-1 (extractLWE uses 2N - term)
    if (shift == 0) return;
    const u64 words = (u64)ct.size * poly_words(c, ct.limbs);
    c.arena.begin(s);
    c.arena.reserve(batch * words);
    u64 *tmp = c.arena.take(batch * words);
    launch_copy_strided(ct.data, ct.bstride, tmp, words, words, batch, s);
    launch_negacyclic_shift(tmp, words, ct.data, ct.bstride, c.d_desc, c.ct_map(ct.limbs), c.logn, shift, (u64)ct.size * ct.limbs, ct.limbs, batch, s);
}

// divideByPolyModulusDegreeInplace (evaluator_cuda.cu:2259-2273): every limb times N^-1 (times `mul`) modulo its prime
void Evaluator::divide_by_degree(CtBatch &ct, u64 mul, u64 batch, hipStream_t s) {
    check_ct(ct);
    u64 sc[64];
    for (int l = 0; l < ct.limbs; l++) {
        const u64 p = c.primes[l];
        sc[l] = host::mul_mod(host::inv_mod_checked(c.N % p, p), mul % p, p);
    }
    const LimbMap map = c.ct_map(ct.limbs);
    const u64 words = (u64)ct.size * poly_words(c, ct.limbs);
    if (ct.bstride == words) launch_mul_scalar(ct.data, c.d_desc, map, sc, c.logn, batch * ct.size * ct.limbs, s);
    else for (u64 b = 0; b < batch; b++) launch_mul_scalar(ct.data + b * ct.bstride, c.d_desc, map, sc, c.logn, (u64)ct.size * ct.limbs, s);
}

// ---- decryption ----
void Evaluator::decrypt(const CtBatch &ct, const u64 *sk, u64 *out, u64 out_bstride, u64 batch, hipStream_t s) {
    check_ct(ct);
    if (!sk || !out) throw Error(ST_INVALID_ARGUMENT, "secret key / destination");
    if (ct.size < 2) throw Error(ST_INVALID_ARGUMENT, "encrypted is not valid for encryption parameters");
    if (c.scheme == SCHEME_CKKS && !ct.ntt) throw Error(ST_INVALID_ARGUMENT, "CKKS encrypted must be in NTT form");
    const u64 N = c.N, limbs = ct.limbs, pw = poly_words(c, ct.limbs), np = ct.size - 1;
    const host::RnsLevel &r = c.level(ct.limbs).rns;
    DecryptArgs a;
    std::memset(&a, 0, sizeof(a));
    a.primes = c.d_desc; a.map = c.ct_map(ct.limbs); a.logn = c.logn;
    a.limbs = limbs; a.size = ct.size; a.batch = batch; a.ct_bstride = ct.bstride; a.out_bstride = out_bstride;
    c.arena.begin(s);
    c.arena.reserve(batch * np * pw + np * pw + batch * pw + 512);
    u64 *x = c.arena.take(batch * np * pw), *spow = c.arena.take(np * pw), *acc = c.arena.take(batch * pw);
    // c_1 .. c_{size-1} in NTT form
    launch_copy_strided(ct.data + pw, ct.bstride, x, np * pw, np * pw, batch, s);
    if (!ct.ntt) launch_ntt(x, c.d_desc, a.map, batch * np * limbs, c.logn, false, s);
    // s, s^2, .. at this level (the key is stored at the key level: limb l of the key is limb l here)
    HIP_CHECK(hipMemcpyAsync(spow, sk, pw * sizeof(u64), hipMemcpyDeviceToDevice, s));
    for (u64 i = 1; i < np; i++) launch_ew(3, spow + (i - 1) * pw, sk, spow + i * pw, c.d_desc, a.map, c.logn, limbs, s);
    launch_dot_sk(x, spow, acc, a, s);
    if (!ct.ntt) launch_ntt(acc, c.d_desc, a.map, batch * limbs, c.logn, true, s);
    launch_add_c0(ct.data, acc, a, s);
    if (c.scheme == SCHEME_CKKS) {
        launch_copy_strided(acc, pw, out, out_bstride, pw, batch, s);
        return;
    }
    const Mod tm = make_mod(c.t);
    a.t_p = tm.p; a.t_cr0 = tm.cr0; a.t_cr1 = tm.cr1;
    if (c.scheme == SCHEME_BFV) {
        const Mod gm = make_mod(r.gamma);
        a.g_p = gm.p; a.g_cr0 = gm.cr0; a.g_cr1 = gm.cr1;
        const host::BaseConv &bc = r.q_to_tgamma;
        for (u64 l = 0; l < limbs; l++) {
            const u64 ql = c.primes[l];
            a.pre[l] = make_shoup(host::mul_mod(r.prod_tgamma_mod_q[l], bc.inv_punct[l], ql), ql);
            a.mat_t[l] = bc.mat[0][l];
            a.mat_g[l] = bc.mat[1][l];
        }
        a.neg_inv_q_mod_t = r.neg_inv_q_mod_t; a.neg_inv_q_mod_gamma = r.neg_inv_q_mod_gamma; a.inv_gamma_mod_t = r.inv_gamma_mod_t;
    } else {
        host::BaseConv bc;
        std::vector<u64> q(c.primes.begin(), c.primes.begin() + limbs);
        bc.build(q, {c.t});
        for (u64 l = 0; l < limbs; l++) { a.pre[l] = make_shoup(bc.inv_punct[l], q[l]); a.mat_t[l] = bc.mat[0][l]; }
        a.q_mod_t = host::product_mod(q, c.t);
        a.inv_cf = 1;
        if (ct.cf != 1 && !host::inv_mod(ct.cf, c.t, a.inv_cf)) throw Error(ST_LOGIC_ERROR, "invalid correction factor");
    }
    launch_decrypt_final(c.scheme, acc, out, a, s);
}

} // namespace troyhip
