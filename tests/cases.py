"""Shared definition of the parity scenarios.

One scenario = one encryption-parameter set + the op sequence of SURVEY.md section 8(a) run on inputs that are
regenerated from (seed, shape) by troy_amd.synth.  The same scenario runs on three backends:
  * RefBackend    -- the real reference CPU path (oracle/_ref), only in the build container: generates golden files
  * OracleBackend -- our CPU restatement (oracle/troy_oracle.cpp)
  * GpuBackend    -- the product: libtroyhip.so through the C ABI (troy_amd.api)
and the tests compare their outputs limb-for-limb (bit-exact).
"""
import hashlib

import numpy as np

from troy_amd import synth

BFV, CKKS, BGV = 1, 2, 3

# name -> parameters.  Small ones are stored in full in tests/golden/, large ones as SHA-256 of the limbs.
CONFIGS = {
    # reference test-suite sized parameters (test/evaluator_cuda.cu uses N=64..128, 30-60 bit primes)
    "bfv_n64_k3": dict(scheme=BFV, N=64, bits=[40, 40, 40], tbits=10),
    "bfv_n128_k4": dict(scheme=BFV, N=128, bits=[40, 40, 40, 40], tbits=10),       # BFVRelinearize parameters
    "bfv_n128_k5_60": dict(scheme=BFV, N=128, bits=[60, 60, 60, 60, 60], tbits=40),  # forces |B| = L+1 (rns.cpp:610-615)
    "ckks_n128_k6": dict(scheme=CKKS, N=128, bits=[30] * 6, tbits=0),              # CKKSEncryptMultiplyRelinRescaleDecrypt
    "bgv_n128_k4": dict(scheme=BGV, N=128, bits=[40, 36, 36, 40], tbits=10),
    # BASELINE.json configs
    "cfgA_bfv_n4096_k3": dict(scheme=BFV, N=4096, bits=[36, 36, 37], tbits=20),
    "cfgB_bfv_n8192_k5": dict(scheme=BFV, N=8192, bits=[40, 36, 36, 36, 40], tbits=20),
    "ckks_n4096_k4": dict(scheme=CKKS, N=4096, bits=[40, 30, 30, 40], tbits=0),
    "bgv_n4096_k3": dict(scheme=BGV, N=4096, bits=[36, 36, 37], tbits=20),
    "cfgNS_bfv_n32768_k15": dict(scheme=BFV, N=32768, bits=[60] + [58] * 13 + [60], tbits=20),
    "cfgC_ckks_n32768_k15": dict(scheme=CKKS, N=32768, bits=[60] + [40] * 13 + [60], tbits=0),
    "bfv_n2048_k3": dict(scheme=BFV, N=2048, bits=[40, 40, 40], tbits=16),           # largest N of the generic (single LDS tile) NTT
    "bfv_n16384_k4": dict(scheme=BFV, N=16384, bits=[50, 45, 45, 50], tbits=20),      # k1 = 5 strided pass
    "ckks_n16384_k4": dict(scheme=CKKS, N=16384, bits=[50, 40, 40, 50], tbits=0),
    "bfv_n131072_k3": dict(scheme=BFV, N=131072, bits=[50, 50, 50], tbits=20),       # largest N: 10-stage contiguous pass, unfused paths
    "cfgD_bgv_n65536_k15": dict(scheme=BGV, N=65536, bits=[60] + [50] * 13 + [60], tbits=20),  # relinearize + rotateRows
}
SMALL = ["bfv_n64_k3", "bfv_n128_k4", "bfv_n128_k5_60", "ckks_n128_k6", "bgv_n128_k4"]
MEDIUM = ["cfgA_bfv_n4096_k3", "cfgB_bfv_n8192_k5", "ckks_n4096_k4", "bgv_n4096_k3"]
LARGE = ["cfgNS_bfv_n32768_k15", "cfgC_ckks_n32768_k15", "cfgD_bgv_n65536_k15"]
CHAIN = ["ckks_n128_k6", "ckks_n4096_k4", "cfgC_ckks_n32768_k15"]  # scenario_chain (depth 3; ckks_n4096_k4 has 3 data levels: depth 2 there)
SIZES = ["bfv_n64_k3", "bfv_n128_k5_60", "ckks_n128_k6", "bgv_n128_k4", "cfgA_bfv_n4096_k3", "ckks_n4096_k4", "bgv_n4096_k3"]  # scenario_sizes

KEY_STEPS = (1, -1, 4)  # Galois keys present; rotations by 5 = naf [1, 4] and 3 = naf [-1, 4] exercise the NAF path
SEED = 0x5EED


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr, dtype=np.uint64).tobytes()).hexdigest()


class Meta:
    def __init__(self, data, is_ntt, scale, cf):
        self.data, self.is_ntt, self.scale, self.cf = np.asarray(data, dtype=np.uint64), bool(is_ntt), float(scale), int(cf)


def scenario(backend, cfg, light=False):
    """Runs the op list on `backend`; returns {name: Meta}.  `light` skips the per-level sweep (large N)."""
    scheme, N = cfg["scheme"], cfg["N"]
    primes = backend.primes
    K = len(primes)
    L = K - 1
    ntt = scheme == CKKS
    out = {}
    backend.set_relin_key(synth.uniform_kswitch_key(SEED + 1, primes, N))
    # steps need |step| < N / 2 (galois.cpp: "step count too large"): tiny rings keep the keys and rotations that exist there
    key_steps = [s for s in KEY_STEPS if abs(s) < N // 2]
    elts = [backend.elt_from_step(s) for s in key_steps] + [2 * N - 1]
    for i, e in enumerate(elts):
        backend.set_galois_key(e, synth.uniform_kswitch_key(SEED + 10 + i, primes, N))

    levels = [L] if light else list(range(L, backend.last_limbs - 1, -1))
    for limbs in levels:
        q = primes[:limbs]
        tag = f"l{limbs}"
        xa = synth.uniform_ct(SEED + 100 + limbs, q, 2, N)[0]
        xb = synth.uniform_ct(SEED + 200 + limbs, q, 2, N)[0]
        x3 = synth.uniform_ct(SEED + 300 + limbs, q, 3, N)[0]
        cfb = 3 if scheme == BGV else 1
        a = lambda: backend.ct(xa, ntt)  # noqa: E731
        b = lambda: backend.ct(xb, ntt, cf=cfb)  # noqa: E731
        c3 = lambda: backend.ct(x3, ntt)  # noqa: E731
        m = backend.multiply(a(), b())
        out[f"{tag}/multiply"] = backend.export(m)
        out[f"{tag}/relinearize"] = backend.export(backend.relinearize(m))
        if light:
            r = backend.relinearize(backend.multiply(a(), b()))
            if scheme == CKKS and limbs > backend.last_limbs:
                r = backend.rescale(r)
                out[f"{tag}/mul_relin_rescale"] = backend.export(r)
                if 1 in key_steps:
                    out[f"{tag}/mul_relin_rescale_rotate1"] = backend.export(backend.rotate(r, 1))
            elif 1 in key_steps:
                out[f"{tag}/rotate1"] = backend.export(backend.rotate(a(), 1))
            continue
        out[f"{tag}/add"] = backend.export(backend.add(a(), b()))
        out[f"{tag}/sub"] = backend.export(backend.sub(a(), b()))
        out[f"{tag}/negate"] = backend.export(backend.negate(a()))
        out[f"{tag}/square"] = backend.export(backend.square(a()))
        out[f"{tag}/add_size3"] = backend.export(backend.add(a(), c3()))
        out[f"{tag}/sub_size3"] = backend.export(backend.sub(a(), c3()))
        if limbs > backend.last_limbs:
            out[f"{tag}/mod_switch"] = backend.export(backend.mod_switch(c3()))
            if scheme == CKKS:
                out[f"{tag}/rescale"] = backend.export(backend.rescale(a()))
        out[f"{tag}/apply_galois"] = backend.export(backend.apply_galois(a(), elts[0]))
        for s in (1, 5, 3):
            if abs(s) < N // 2 and len(key_steps) == len(KEY_STEPS):
                out[f"{tag}/rotate{s}"] = backend.export(backend.rotate(a(), s))
        out[f"{tag}/conjugate"] = backend.export(backend.conjugate(a()))
        if not ntt:
            out[f"{tag}/to_ntt"] = backend.export(backend.to_ntt(a()))
        an = backend.ct(xa, True)
        out[f"{tag}/from_ntt"] = backend.export(backend.from_ntt(an))
        pl = synth.uniform_rows(SEED + 400 + limbs, q, limbs, N)
        out[f"{tag}/multiply_plain_ntt"] = backend.export(backend.multiply_plain(backend.ct(xa, True), pl))
        # plaintext operands in coefficient form: full, ragged (N-5 coefficients) and -- for add/sub -- a monomial
        if scheme == CKKS:
            out[f"{tag}/add_plain"] = backend.export(backend.add_plain(a(), pl, N))
            out[f"{tag}/sub_plain"] = backend.export(backend.sub_plain(a(), pl, N))
        else:
            t = backend.t
            for n in ((N, N - 5) if N > 5 else (N,)):
                pt = synth.uniform_rows(SEED + 500 + limbs + n, [t], 1, n)[0]
                out[f"{tag}/add_plain{n}"] = backend.export(backend.add_plain(b(), pt, n))
                out[f"{tag}/sub_plain{n}"] = backend.export(backend.sub_plain(b(), pt, n))
                out[f"{tag}/multiply_plain{n}"] = backend.export(backend.multiply_plain_normal(b(), pt, n))
                out[f"{tag}/plain_to_ntt{n}"] = Meta(backend.plain_to_ntt(pt, limbs)[None], True, 1.0, 1)
            nm = min(7, N)
            mono = np.zeros(nm, dtype=np.uint64)
            mono[nm - 1] = t - 1
            out[f"{tag}/add_plain_mono"] = backend.export(backend.add_plain(b(), mono, nm))
    return out


RELIN_KEY = 0x80000000  # set_kswitch_key(RELIN_KEY | i, ..): relinearization key of index i (power i + 2)


def scenario_sizes(backend, cfg):
    """General ciphertext sizes: 3x2, 3x3 multiply, square of a size-3 ciphertext, relinearize 4 -> 2 and 5 -> 2
    with the keys of index 0..2 (src/evaluator_cuda.cu:283-501, 703-744), at the first data level.  Returns {name: Meta}."""
    scheme, N = cfg["scheme"], cfg["N"]
    primes = backend.primes
    L = len(primes) - 1
    ntt = scheme == CKKS
    for i in range(3):
        backend.set_relin_key(synth.uniform_kswitch_key(SEED + 1 + 7 * i, primes, N), index=i)  # index 0 = the scenario() key
    q = primes[:L]
    xa = synth.uniform_ct(SEED + 100 + L, q, 2, N)[0]
    x3 = synth.uniform_ct(SEED + 300 + L, q, 3, N)[0]
    y3 = synth.uniform_ct(SEED + 600 + L, q, 3, N)[0]
    cfb = 3 if scheme == BGV else 1
    out = {}
    m32 = backend.multiply(backend.ct(x3, ntt), backend.ct(xa, ntt, cf=cfb))
    out["sizes/multiply_3x2"] = backend.export(m32)
    out["sizes/relin_4to2"] = backend.export(backend.relinearize(m32))
    m33 = backend.multiply(backend.ct(x3, ntt), backend.ct(y3, ntt, cf=cfb))
    out["sizes/multiply_3x3"] = backend.export(m33)
    out["sizes/relin_5to2"] = backend.export(backend.relinearize(m33))
    out["sizes/square_3"] = backend.export(backend.square(backend.ct(x3, ntt)))
    out["sizes/multiply_2x3"] = backend.export(backend.multiply(backend.ct(xa, ntt), backend.ct(y3, ntt, cf=cfb)))
    return out


# ------------------------------------------------------------------ backends
class _EvalBackend:
    """RefBackend / OracleBackend: both expose the `eval(op, a, b, iarg)` surface of oracle/ref.py."""

    def __init__(self, impl_module, cfg, primes, t):
        from oracle import ref as R
        self.R = R
        cls = impl_module.Ref if hasattr(impl_module, "Ref") else impl_module.Oracle
        self.impl = cls(cfg["scheme"], cfg["N"], primes, t)
        self.cfg, self.primes = cfg, primes
        self.last_limbs = self.impl.chain()[2]

    def ct(self, data, ntt, cf=1):
        return self.R.Ct(data, ntt, 1.0, cf)

    def export(self, c):
        return Meta(c.data, c.is_ntt, c.scale, c.correction_factor)

    def set_relin_key(self, k, index=0):
        self.impl.set_kswitch_key(0 if index == 0 else (RELIN_KEY | index), k)

    def set_galois_key(self, elt, k):
        self.impl.set_kswitch_key(elt, k)

    def elt_from_step(self, s):
        return self.impl.elt_from_step(s)

    def _e(self, op, a, b=None, iarg=0):
        return self.impl.eval(op, a, b, iarg)

    def add(self, a, b): return self._e(self.R.OP_ADD, a, b)
    def sub(self, a, b): return self._e(self.R.OP_SUB, a, b)
    def negate(self, a): return self._e(self.R.OP_NEGATE, a)
    def multiply(self, a, b): return self._e(self.R.OP_MULTIPLY, a, b)
    def square(self, a): return self._e(self.R.OP_SQUARE, a)
    def relinearize(self, a): return self._e(self.R.OP_RELIN, a)
    def mod_switch(self, a): return self._e(self.R.OP_MODSWITCH_NEXT, a)
    def rescale(self, a): return self._e(self.R.OP_RESCALE_NEXT, a)
    def apply_galois(self, a, elt): return self._e(self.R.OP_APPLY_GALOIS, a, iarg=elt)
    def rotate(self, a, s): return self._e(self.R.OP_ROTATE_VECTOR if self.cfg["scheme"] == CKKS else self.R.OP_ROTATE_ROWS, a, iarg=s)
    def conjugate(self, a): return self._e(self.R.OP_CONJUGATE if self.cfg["scheme"] == CKKS else self.R.OP_ROTATE_COLUMNS, a)
    def to_ntt(self, a): return self._e(self.R.OP_TO_NTT, a)
    def from_ntt(self, a): return self._e(self.R.OP_FROM_NTT, a)
    def multiply_plain(self, a, pl): return self._e(self.R.OP_MULTIPLY_PLAIN_NTT, a, pl)
    def add_plain(self, a, pl, n): return self._e(self.R.OP_ADD_PLAIN, a, pl, iarg=n)
    def sub_plain(self, a, pl, n): return self._e(self.R.OP_SUB_PLAIN, a, pl, iarg=n)
    def multiply_plain_normal(self, a, pl, n): return self._e(self.R.OP_MULTIPLY_PLAIN, a, pl, iarg=n)
    def plain_to_ntt(self, pl, limbs): return self.impl.plain_to_ntt(pl, limbs)

    @property
    def t(self):
        return self.impl.t


def ref_backend(cfg):
    from oracle import ref
    primes = ref.coeff_modulus_create(cfg["N"], cfg["bits"])
    t = ref.plain_batching(cfg["N"], cfg["tbits"]) if cfg["scheme"] != CKKS else 0
    return _EvalBackend(ref, cfg, primes, t)


def oracle_backend(cfg):
    from oracle import oracle
    primes = oracle.coeff_modulus_create(cfg["N"], cfg["bits"])
    t = oracle.plain_batching(cfg["N"], cfg["tbits"]) if cfg["scheme"] != CKKS else 0
    return _EvalBackend(oracle, cfg, primes, t)


class GpuBackend:
    """The product path: every op goes through libtroyhip.so's C ABI (troy_amd.api)."""

    def __init__(self, cfg, batch=1):
        from troy_amd import api
        self.api, self.cfg, self.batch = api, cfg, batch
        self.primes = api.CoeffModulus.Create(cfg["N"], cfg["bits"])
        self.t = api.PlainModulus.Batching(cfg["N"], cfg["tbits"]) if cfg["scheme"] != CKKS else 0
        self.ctx = api.SEALContext(cfg["scheme"], cfg["N"], self.primes, self.t)
        self.ev = api.Evaluator(self.ctx)
        self.last_limbs = self.ctx.last_limbs
        self.rlk, self.gk = api.RelinKeys(self.ctx), api.GaloisKeys(self.ctx)

    def ct(self, data, ntt, cf=1):
        d = np.broadcast_to(np.asarray(data, dtype=np.uint64), (self.batch,) + data.shape)  # same ct in every batch slot
        return self.api.Ciphertext.from_numpy(self.ctx, d, ntt, 1.0, cf, capacity=3)

    def export(self, c):
        d = c.cpu()
        for b in range(1, d.shape[0]):
            assert np.array_equal(d[b], d[0]), "batch items diverged"
        return Meta(d[0], c.is_ntt_form, c.scale, c.correction_factor)

    def set_relin_key(self, k, index=0): self.rlk.set(index, k)
    def set_galois_key(self, elt, k): self.gk.set_elt(elt, k)
    def elt_from_step(self, s): return self.ctx.galois_elt_from_step(s)
    def add(self, a, b): self.ev.addInplace(a, b); return a
    def sub(self, a, b): self.ev.subInplace(a, b); return a
    def negate(self, a): self.ev.negateInplace(a); return a
    def multiply(self, a, b): return self.ev.multiply(a, b)
    def square(self, a): return self.ev.square(a)
    def relinearize(self, a): return self.ev.relinearize(a, self.rlk)
    def mod_switch(self, a): return self.ev.modSwitchToNext(a)
    def rescale(self, a): return self.ev.rescaleToNext(a)
    def apply_galois(self, a, elt): self.ev.applyGaloisInplace(a, elt, self.gk); return a

    def rotate(self, a, s):
        (self.ev.rotateVectorInplace if self.cfg["scheme"] == CKKS else self.ev.rotateRowsInplace)(a, s, self.gk)
        return a

    def conjugate(self, a):
        (self.ev.complexConjugateInplace if self.cfg["scheme"] == CKKS else self.ev.rotateColumnsInplace)(a, self.gk)
        return a

    def to_ntt(self, a): self.ev.transformToNttInplace(a); return a
    def from_ntt(self, a): self.ev.transformFromNttInplace(a); return a

    def multiply_plain(self, a, pl):
        self.ev.multiplyPlainInplace(a, self.api.DeviceBuffer.from_numpy(pl))
        return a

    def add_plain(self, a, pl, n): self.ev.addPlainInplace(a, self.api.DeviceBuffer.from_numpy(pl), n); return a
    def sub_plain(self, a, pl, n): self.ev.subPlainInplace(a, self.api.DeviceBuffer.from_numpy(pl), n); return a
    def multiply_plain_normal(self, a, pl, n): self.ev.multiplyPlainNormalInplace(a, self.api.DeviceBuffer.from_numpy(pl), n); return a

    def plain_to_ntt(self, pl, limbs):
        out = self.ev.transformPlainToNtt(self.api.DeviceBuffer.from_numpy(pl), limbs, len(pl))
        return out.to_numpy().reshape(limbs, self.cfg["N"])


def compare(got, exp, names=None):
    """bit-exact comparison of two scenario outputs; returns list of mismatching names"""
    bad = []
    for k in (names or exp.keys()):
        g, e = got[k], exp[k]
        ok = g.data.shape == e.data.shape and np.array_equal(g.data, e.data) and g.is_ntt == e.is_ntt and g.cf == e.cf \
            and abs(g.scale - e.scale) <= 1e-12 * abs(e.scale)
        if not ok:
            bad.append(k)
    return bad


def check_plain_monomial_and_batch(gpu_cfg_name):
    """multiplyPlain by a one-coefficient plaintext follows the reference's CUDA evaluator (generic branch,
    evaluator_cuda.cu:1757-1815; the CPU evaluator short-cuts it differently, oracle flag bit 32 selects the CUDA semantics),
    and per-item plaintexts (plain_batch_stride != 0) give each batch row its own result."""
    cfg = CONFIGS[gpu_cfg_name]
    B = 3
    be = GpuBackend(cfg, batch=B)
    orc = oracle_backend(cfg)
    N, L, t = cfg["N"], len(be.primes) - 1, be.t
    q = be.primes[:L]
    from oracle import ref as R
    xs = synth.uniform_ct(991, q, 2, N, B)
    mono = np.zeros(9, dtype=np.uint64)
    mono[8] = t - 2
    c = be.api.Ciphertext.from_numpy(be.ctx, xs, False, 1.0, 1, capacity=3)
    be.ev.multiplyPlainNormalInplace(c, be.api.DeviceBuffer.from_numpy(mono), 9)
    got = c.cpu()
    for b in range(B):
        exp = orc.impl.eval(R.OP_MULTIPLY_PLAIN, R.Ct(xs[b], False), mono, iarg=9 | (1 << 32))
        assert np.array_equal(got[b], exp.data), b
    # one plaintext per batch item
    pts = synth.uniform_rows(992, [t], B, N - 3)
    for name, op in (("add", R.OP_ADD_PLAIN), ("sub", R.OP_SUB_PLAIN), ("mul", R.OP_MULTIPLY_PLAIN)):
        c = be.api.Ciphertext.from_numpy(be.ctx, xs, False, 1.0, 1, capacity=3)
        buf = be.api.DeviceBuffer.from_numpy(pts)
        if name == "add":
            be.ev.addPlainInplace(c, buf, N - 3, per_item=True)
        elif name == "sub":
            be.ev.subPlainInplace(c, buf, N - 3, per_item=True)
        else:
            be.ev.multiplyPlainNormalInplace(c, buf, N - 3, per_item=True)
        got = c.cpu()
        for b in range(B):
            exp = orc.impl.eval(op, R.Ct(xs[b], False), pts[b], iarg=N - 3)
            assert np.array_equal(got[b], exp.data), (name, b)
    ntt = be.ev.transformPlainToNtt(be.api.DeviceBuffer.from_numpy(pts), L, N - 3, count=B).to_numpy().reshape(B, L, N)
    for b in range(B):
        assert np.array_equal(ntt[b], orc.impl.plain_to_ntt(pts[b], L))


def check_dense_multiply(cfg_name, batch=3):
    """BFV multiply consumes DENSE operands in place (extension and first NTT pass read them directly); strided operands
    (capacity > size) go through a staging copy.  Both must give the oracle's product, and must leave the inputs untouched."""
    cfg = CONFIGS[cfg_name]
    be = GpuBackend(cfg, batch=batch)
    orc = oracle_backend(cfg)
    from oracle import ref as R
    N, L = cfg["N"], len(be.primes) - 1
    q = be.primes[:L]
    ntt = cfg["scheme"] == CKKS
    xa, xb = synth.uniform_ct(881, q, 2, N, batch), synth.uniform_ct(882, q, 2, N, batch)
    exp = [orc.impl.eval(R.OP_MULTIPLY, R.Ct(xa[i], ntt), R.Ct(xb[i], ntt)).data for i in range(batch)]
    for cap in (None, 3):
        a = be.api.Ciphertext.from_numpy(be.ctx, xa, ntt, 1.0, 1, capacity=cap)
        b = be.api.Ciphertext.from_numpy(be.ctx, xb, ntt, 1.0, 1, capacity=cap)
        m = be.ev.multiply(a, b).cpu()
        for i in range(batch):
            assert np.array_equal(m[i], exp[i]), (cap, i)
        assert np.array_equal(a.cpu(), xa) and np.array_equal(b.cpu(), xb)
        sq = be.ev.square(a).cpu()
        for i in range(batch):
            assert np.array_equal(sq[i], orc.impl.eval(R.OP_SQUARE, R.Ct(xa[i], ntt)).data), (cap, i)


def check_gpu_decrypt(cfg_name, batch=3):
    """DecryptorCuda::decrypt on the device vs the CPU oracle (pinned on the reference's Decryptor), bit-exact:
    uniform synthetic ciphertexts of size 2 and 3 at every level (decryption is plain arithmetic, it does not need a valid
    encryption), BGV with a non-trivial correction factor."""
    cfg = CONFIGS[cfg_name]
    be = GpuBackend(cfg, batch=batch)
    orc = oracle_backend(cfg)
    from oracle import ref as R
    N, K = cfg["N"], len(be.primes)
    sk = synth.uniform_rows(771, be.primes, K, N)                      # any [K][N] residues serve as an NTT-form key
    skd = be.api.DeviceBuffer.from_numpy(sk)
    ntt = cfg["scheme"] == CKKS
    for limbs in range(K - 1, be.last_limbs - 1, -1):
        q = be.primes[:limbs]
        for size in (2, 3):
            for cf in ((1, 5) if cfg["scheme"] == BGV else (1,)):
                xs = synth.uniform_ct(772 + limbs + size, q, size, N, batch)
                c = be.api.Ciphertext.from_numpy(be.ctx, xs, ntt, 1.0, cf, capacity=3)
                got = be.ev.decrypt(c, skd)
                for b in range(batch):
                    exp = orc.impl.decrypt(R.Ct(xs[b], ntt, 1.0, cf), sk)
                    assert np.array_equal(got[b].reshape(-1), exp), (limbs, size, cf, b)


def check_api_compositions(cfg_name):
    """applyKeySwitching, negacyclicShift (CUDA-only API of the reference, restated in the oracle from evaluator_cuda.cu) and
    the host-level compositions multiplyMany / exponentiate / modSwitchTo in the reference's order of operations."""
    cfg = CONFIGS[cfg_name]
    B = 2
    be = GpuBackend(cfg, batch=B)
    orc = oracle_backend(cfg)
    from oracle import ref as R
    N, K = cfg["N"], len(be.primes)
    L = K - 1
    q = be.primes[:L]
    key = synth.uniform_kswitch_key(55, be.primes, N)
    be.rlk.set(0, key)
    orc.impl.set_kswitch_key(0, key)
    ntt = cfg["scheme"] == CKKS
    xs = [synth.uniform_ct(60 + i, q, 2, N, B) for i in range(3)]
    mk = lambda i: be.api.Ciphertext.from_numpy(be.ctx, xs[i], ntt, 1.0, 1, capacity=3)  # noqa: E731
    ock = lambda i, b: R.Ct(xs[i][b], ntt)  # noqa: E731
    # applyKeySwitching with a single-key KSwitchKeys object
    ksk = be.api.KSwitchKeys(be.ctx)
    ksk.set(0, key)
    got = be.ev.applyKeySwitching(mk(0), ksk).cpu()
    for b in range(B):
        assert np.array_equal(got[b], orc.impl.eval(R.OP_APPLY_KEYSWITCH, ock(0, b)).data), b
    # negacyclic shift
    for shift in (0, 1, N // 2 + 3, N - 1):
        got = be.ev.negacyclicShift(mk(1), shift).cpu()
        for b in range(B):
            assert np.array_equal(got[b], orc.impl.eval(R.OP_NEGACYCLIC_SHIFT, ock(1, b), iarg=shift).data), (shift, b)
    if cfg["scheme"] == CKKS:
        return
    # multiplyMany over three ciphertexts = relin(relin(x0 x1) x2); exponentiate 3 = relin(relin(x x) x)
    got = be.ev.multiplyMany([mk(0), mk(1), mk(2)], be.rlk).cpu()
    for b in range(B):
        e = orc.impl.eval(R.OP_RELIN, orc.impl.eval(R.OP_MULTIPLY, ock(0, b), ock(1, b)))
        e = orc.impl.eval(R.OP_RELIN, orc.impl.eval(R.OP_MULTIPLY, e, ock(2, b)))
        assert np.array_equal(got[b], e.data), b
    got = be.ev.exponentiate(mk(0), 3, be.rlk).cpu()
    for b in range(B):
        e = orc.impl.eval(R.OP_RELIN, orc.impl.eval(R.OP_SQUARE, ock(0, b)))
        e = orc.impl.eval(R.OP_RELIN, orc.impl.eval(R.OP_MULTIPLY, e, ock(0, b)))
        assert np.array_equal(got[b], e.data), b
    if L - 1 >= be.last_limbs:
        got = be.ev.modSwitchTo(mk(0), be.last_limbs)
        for b in range(B):
            e = ock(0, b)
            while e.limbs > be.last_limbs:
                e = orc.impl.eval(R.OP_MODSWITCH_NEXT, e)
            assert np.array_equal(got.cpu()[b], e.data), b


def check_ckks_matmul_helper(N=4096, bits=(40, 30, 30, 40), batch=3, dims=(128, 128), seed=7):
    """BASELINE config E through the reference's app API (app/LinearHelperCKKS.cuh MatmulHelper, troy_amd/app.py): encrypted
    [batch x in] times plaintext [in x out], decrypted on the device; floating point: |error| < 1e-3 at scale 2^22 x 2^22."""
    from troy_amd import api, app, capi
    primes = api.CoeffModulus.Create(N, list(bits))
    ctx = api.SEALContext(capi.CKKS, N, primes, 0)
    kg = api.KeyGenerator(ctx, seed=(31, 32))
    enc = api.Encryptor(ctx, kg.createPublicKey())
    enc.setSecretKey(kg.secretKey())  # the helpers encrypt symmetrically, as the reference's (test/app/linear_ckks.cu:138)
    ev = api.Evaluator(ctx)
    encoder = app.CKKSPolyEncoder(ctx)
    L = len(primes) - 1
    rng = np.random.default_rng(seed)
    # encode / decode round trip, including negative and half-way values
    v = rng.uniform(-4, 4, N)
    v[:4] = [0.5, -0.5, 1.5, -2.5]
    back = encoder.decodePolynomial(encoder.encodePolynomial(v, L, 2.0 ** 22), 2.0 ** 22)
    assert np.max(np.abs(back - v)) < 2.0 ** -22
    I, J = dims
    X = rng.uniform(-1, 1, (batch, I))
    W = rng.uniform(-1, 1, (I, J))
    h = app.MatmulHelper(batch, I, J, N // 2)
    assert h.blockHeight * h.blockWidth <= N and h.blockHeight >= 1
    h.encodeWeights(encoder, L, W, 2.0 ** 22)
    a = h.encryptInputs(enc, encoder, L, X, 2.0 ** 22)
    out = h.matmul(ev, a)
    sk = api.DeviceBuffer.from_numpy(kg.secretKey())
    got = h.decryptOutputs(ev, encoder, sk, out)
    assert np.max(np.abs(got - X @ W)) < 1e-3, np.max(np.abs(got - X @ W))
    # serializeOutputs / deserializeOutputs (saveTerms / loadTerms): only the read coefficients of c0 travel, same answer
    import io
    stream = io.BytesIO()
    h.serializeOutputs(ev, out, stream)
    full = sum(2 * o.limbs * N * 8 * o.batch for o in out)
    assert len(stream.getvalue()) < 0.6 * full + 200 * batch * len(out)
    again = h.deserializeOutputs(ev, ctx, io.BytesIO(stream.getvalue()))
    assert np.array_equal(h.decryptOutputs(ev, encoder, sk, again), got)


def check_lwe_pack(scheme=BFV, N=256, bits=(40, 40, 40, 40), tbits=14, n_lwe=5, batch=2):
    """extractLWE / assembleLWE / packLWECiphertexts (CUDA-only API of the reference; verified by decryption): coefficient
    `term` of a fresh encryption is extracted as an LWE sample and n of them are packed back into one ciphertext whose
    coefficients j * N / n' hold the n messages (BFV/BGV: exactly; the factor N^-1 is applied by the packing)."""
    from troy_amd import api
    primes = api.CoeffModulus.Create(N, list(bits))
    t = api.PlainModulus.Batching(N, tbits)
    ctx = api.SEALContext(scheme, N, primes, t)
    kg = api.KeyGenerator(ctx, seed=(41, 42))
    enc = api.Encryptor(ctx, kg.createPublicKey())
    dec = api.Decryptor(ctx, kg.secretKey())
    ev = api.Evaluator(ctx)
    gk = api.GaloisKeys(ctx)
    l = 0
    while (1 << l) < n_lwe:
        l += 1
    elts = sorted({(1 << k) + 1 for k in range(1, l + 1)} | {d + 1 for d in [N >> k for k in range(0, 20)] if d > (1 << l)})
    for e, key in kg.createGaloisKeys(elts).items():
        gk.set_elt(e, key)
    rng = np.random.default_rng(5)
    msgs = rng.integers(0, t, (n_lwe, batch, N), dtype=np.uint64)
    terms = [int(x) for x in rng.integers(0, N, n_lwe)]
    lwes = []
    for i in range(n_lwe):
        ct = api.Ciphertext.from_numpy(ctx, np.stack([enc.encrypt(msgs[i][b]) for b in range(batch)]), capacity=3)
        lwe = ev.extractLWE(ct, terms[i])
        # assemble back at term 0: coefficient 0 must decrypt to the extracted message
        back = ev.assembleLWE(lwe, 0).cpu()
        for b in range(batch):
            assert dec.decrypt(back[b])[0] == msgs[i][b][terms[i]], (i, b)
        lwes.append(lwe)
    packed = ev.packLWECiphertexts(lwes, gk).cpu()
    step = N >> l
    for b in range(batch):
        pt = dec.decrypt(packed[b])
        for i in range(n_lwe):
            assert pt[i * step] == msgs[i][b][terms[i]], (i, b, int(pt[i * step]), int(msgs[i][b][terms[i]]))


def check_ckks_conv2d_helper(N=4096, bits=(40, 30, 30, 40), batch=2, image=(12, 12), kernel=(3, 3), channels=(5, 3), seed=11):
    """app/LinearHelperCKKS.cuh Conv2dHelper through troy_amd/app.py: encrypted images x plaintext kernels (valid convolution,
    i.e. cross-correlation with the flipped kernel as the reference packs it) vs numpy; |error| < 1e-3.  A second case forces the
    blocked path (image larger than sqrt(N) per side)."""
    from troy_amd import api, app, capi
    primes = api.CoeffModulus.Create(N, list(bits))
    ctx = api.SEALContext(capi.CKKS, N, primes, 0)
    kg = api.KeyGenerator(ctx, seed=(51, 52))
    enc = api.Encryptor(ctx, kg.createPublicKey())
    ev = api.Evaluator(ctx)
    encoder = app.CKKSPolyEncoder(ctx)
    skd = api.DeviceBuffer.from_numpy(kg.secretKey())
    L = len(primes) - 1
    rng = np.random.default_rng(seed)
    H, Wd = image
    kh, kw = kernel
    ic, oc = channels
    X = rng.uniform(-1, 1, (batch, ic, H, Wd))
    Wt = rng.uniform(-1, 1, (oc, ic, kh, kw))
    h = app.Conv2dHelper(batch, H, Wd, kh, kw, ic, oc, N // 2)
    h.encodeWeights(encoder, L, Wt, 2.0 ** 22)
    a = h.encryptInputs(enc, encoder, L, X, 2.0 ** 22)
    out = h.conv2d(ev, a)
    got = h.decryptOutputs(ev, encoder, skd, out)
    import io
    stream = io.BytesIO()
    h.serializeOutputs(ev, out, stream)  # saveTerms: the last channel slot of c0 only
    again = h.deserializeOutputs(ev, ctx, io.BytesIO(stream.getvalue()))
    assert np.array_equal(h.decryptOutputs(ev, encoder, skd, again), got)
    exp = np.zeros((batch, oc, H - kh + 1, Wd - kw + 1))
    for b in range(batch):
        for o in range(oc):
            for i in range(H - kh + 1):
                for j in range(Wd - kw + 1):
                    exp[b, o, i, j] = np.sum(X[b, :, i:i + kh, j:j + kw] * Wt[o])
    assert np.max(np.abs(got - exp)) < 1e-3, np.max(np.abs(got - exp))
    return h


def check_bfv_multiply_limb_count(K, N=256, batch=2, seed=900, big=False):
    """BFV multiply (both BEHZ kernels) at L = K - 1 limbs against the oracle: L runs over every k-block count of the matrix-core
    kernels and, past 15 limbs, over the VALU kernels.  big=False: 40/45-bit primes (6 digit rows per q-side output), big=True:
    55/60-bit primes (the headline's case), big="small": 30/32-bit primes (below 2^33 the q side of Shenoy-Kumaresan takes the
    two-word reduction).  The last batch items are extreme inputs: every residue p - 1, and a 0 / 1 pattern."""
    from troy_amd import api, synth
    bits = {True: [60] + [55] * (K - 2) + [60], False: [45] + [40] * (K - 2) + [45], "small": [32] + [30] * (K - 2) + [32]}[big]
    cfg = dict(scheme=BFV, N=N, bits=bits, tbits=14)
    be, ob = GpuBackend(cfg), oracle_backend(cfg)
    L = K - 1
    xa, xb = synth.uniform_ct(seed + K, be.primes[:L], 2, N, batch + 2), synth.uniform_ct(seed + 50 + K, be.primes[:L], 2, N, batch + 2)
    top = np.array(be.primes[:L], dtype=np.uint64)[None, :, None] - np.uint64(1)
    xa[batch], xb[batch] = top, top
    xa[batch + 1], xb[batch + 1] = (np.arange(N, dtype=np.uint64) & np.uint64(1))[None, None, :], top
    r = be.ev.multiply(api.Ciphertext.from_numpy(be.ctx, xa, False), api.Ciphertext.from_numpy(be.ctx, xb, False))
    got = r.cpu()
    for i in range(batch + 2):
        e = ob.multiply(ob.ct(xa[i], False), ob.ct(xb[i], False))
        assert np.array_equal(got[i], ob.export(e).data), (K, i)
    return sha(got)


def check_multiply_plain_accumulate(N=256, batch=3):
    """troyhip_multiply_plain_accumulate (one pass) == the multiplyPlain + addInplace loop of the reference's linear helpers, limb for limb:
    CKKS and BFV (NTT-form operands), sizes 2 and 3, 1 / 3 / 16 products, a strided operand; the argument errors"""
    from troy_amd import api, capi, synth
    for scheme, bits, tbits in ((CKKS, [50, 40, 40, 50], 0), (BFV, [58, 57, 60], 20)):
        primes = api.CoeffModulus.Create(N, bits)
        ctx = api.SEALContext(scheme, N, primes, api.PlainModulus.Batching(N, tbits) if tbits else 0)
        ev = api.Evaluator(ctx)
        L = len(primes) - 1
        q = primes[:L]
        for size, count in ((2, 1), (2, 3), (3, 16)):
            cts = [api.Ciphertext.from_numpy(ctx, synth.uniform_ct(70 + i, q, size, N, batch), True, scale=2.0 ** 20, capacity=3 if i == 1 else None) for i in range(count)]
            pls = [api.DeviceBuffer.from_numpy(synth.uniform_rows(90 + i, q, L, N)) for i in range(count)]
            got = ev.multiplyPlainAccumulate(cts, pls, 2.0 ** 10)
            acc = None
            for ct, pl in zip(cts, pls):
                prod = ct.copy()
                ev.multiplyPlainInplace(prod, pl, 2.0 ** 10)
                if acc is None:
                    acc = prod
                else:
                    ev.addInplace(acc, prod)
            assert np.array_equal(got.cpu(), acc.cpu()) and got.scale == acc.scale and got.is_ntt_form and got.size() == size, (scheme, size, count)
        a = api.Ciphertext.from_numpy(ctx, synth.uniform_ct(1, q, 2, N, batch), True)
        pl = api.DeviceBuffer.from_numpy(synth.uniform_rows(2, q, L, N))
        for bad in (lambda: ev.multiplyPlainAccumulate([], [], 1.0), lambda: ev.multiplyPlainAccumulate([a] * 17, [pl] * 17, 1.0),
                    lambda: ev.multiplyPlainAccumulate([a, api.Ciphertext.from_numpy(ctx, synth.uniform_ct(3, q[:L - 1], 2, N, batch), True)], [pl, pl], 1.0) if L > 1 else (_ for _ in ()).throw(capi.InvalidArgument(0, "")),
                    lambda: ev.multiplyPlainAccumulate([api.Ciphertext.from_numpy(ctx, synth.uniform_ct(4, q, 2, N, batch), False)], [pl], 1.0)):
            try:
                bad()
            except capi.InvalidArgument:
                continue
            raise AssertionError("multiplyPlainAccumulate accepted a bad argument")


def random_config(seed, sizes=(256, 1024, 4096), pool=None):
    """a seeded random parameter set: scheme, N, 2..6 primes whose sizes sit on the thresholds the kernels branch on (2^33: BEHZ one-step
    reduction and guard-free butterflies start; 2^50: Bsk-sized; 2^58: guard-free butterflies end; 60 bits: largest allowed)"""
    rng = np.random.default_rng(seed)
    scheme = (BFV, CKKS, BGV)[seed % 3]  # balanced over consecutive seeds
    N = int(rng.choice(sizes))
    K = int(rng.integers(2, 7))
    pool = pool or [33, 34, 36, 40, 45, 49, 50, 51, 55, 57, 58, 59, 60]
    bits = [int(rng.choice(pool)) for _ in range(K)]
    if scheme == CKKS:  # rescaling divides by the last data prime: keep the primes at least as large as a sensible scale
        bits = [max(b, 36) for b in bits]
    return dict(scheme=scheme, N=N, bits=bits, tbits=int(rng.integers(14, 21)))


def check_random_config(seed, sizes=(256, 1024, 4096), batch=2, light=False, pool=None):
    """the whole op list of `scenario` (every level, every op) on a random parameter set: product vs CPU oracle, limb for limb"""
    from oracle import oracle
    from troy_amd import api
    cfg = random_config(seed, sizes, pool)
    if cfg["scheme"] != CKKS:  # not every size has a batching prime (N = 4096: none of 14 or 15 bits): both sides must say so, then move up
        while True:
            ours = theirs = None
            try:
                ours = api.PlainModulus.Batching(cfg["N"], cfg["tbits"])
            except Exception:
                pass
            try:
                theirs = oracle.plain_batching(cfg["N"], cfg["tbits"])
            except Exception:
                pass
            assert ours == theirs, (cfg, ours, theirs)
            if ours is not None:
                break
            cfg["tbits"] += 1
    try:
        be = GpuBackend(cfg, batch=batch)
    except Exception as e:  # a parameter set the context rejects (plain modulus not below the coefficient modulus, ...): both sides must agree
        try:
            oracle_backend(cfg)
        except Exception:
            return cfg, None
        raise e
    got, exp = scenario(be, cfg, light=light), scenario(oracle_backend(cfg), cfg, light=light)
    bad = compare(got, exp)
    assert not bad, (cfg, bad[:8])
    return cfg, len(got)


def mul_relin_hash(name, batch=2, seed=4242):
    """SHA-256 of multiply + relinearize on seeded inputs of config `name` (GPU backend): used to compare kernel variants that are
    selected by environment switches in child processes"""
    from troy_amd import api, synth
    cfg = CONFIGS[name]
    be = GpuBackend(cfg)
    L, N, ntt = len(be.primes) - 1, cfg["N"], cfg["scheme"] == CKKS
    be.set_relin_key(synth.uniform_kswitch_key(seed, be.primes, N))
    xa, xb = synth.uniform_ct(seed + 1, be.primes[:L], 2, N, batch), synth.uniform_ct(seed + 2, be.primes[:L], 2, N, batch)
    r = be.ev.multiply(api.Ciphertext.from_numpy(be.ctx, xa, ntt), api.Ciphertext.from_numpy(be.ctx, xb, ntt))
    h1 = sha(r.cpu())
    be.ev.relinearizeInplace(r, be.rlk)
    h = h1 + ":" + sha(r.cpu())
    if ntt and L - 1 >= be.last_limbs:  # CKKS: the rescale too (its correction transform has a fused and an element-wise form)
        h += ":" + sha(be.ev.rescaleToNext(r).cpu())
    # a rotation: its key switch accumulates onto (sigma(c0), 0) -- directly in the fused epilogues, through a copy + zero fill in the others
    elt = be.elt_from_step(1)
    be.set_galois_key(elt, synth.uniform_kswitch_key(seed + 3, be.primes, N))
    h += ":" + sha(be.rotate(r, 1).cpu())
    return h


def check_relinearize_out_of_place(name, batch=3, seed=99):
    """relinearize(a, keys) -- from size 3 the operand is read where it lies (troyhip_relinearize_to), larger sizes copy -- against the copy +
    relinearizeInplace it replaces, limb for limb; the operand must come out untouched"""
    from troy_amd import api, synth
    cfg = CONFIGS[name]
    be = GpuBackend(cfg, batch=batch)
    L, N, ntt = len(be.primes) - 1, cfg["N"], cfg["scheme"] == CKKS
    be.set_relin_key(synth.uniform_kswitch_key(seed, be.primes, N), 0)
    be.set_relin_key(synth.uniform_kswitch_key(seed + 1, be.primes, N), 1)
    for size in (3, 4):
        x = synth.uniform_ct(seed + size, be.primes[:L], size, N, batch)
        a = api.Ciphertext.from_numpy(be.ctx, x, ntt, 1.0, 1, capacity=size + (size == 3))  # a strided operand too
        want = a.copy()
        be.ev.relinearizeInplace(want, be.rlk)
        got = be.ev.relinearize(a, be.rlk)
        assert got.size() == 2 and np.array_equal(got.cpu(), want.cpu()), (name, size)
        assert np.array_equal(a.cpu(), x), (name, size, "operand modified")


def check_modswitch_as_first_op(cfg_name, batch=2):
    """rescaleToNext / modSwitchToNext as the FIRST operation on a fresh context (nothing has sized the scratch arena
    yet), with strided (capacity 3) and dense operands, then again at a larger batch.  Results against the oracle."""
    cfg = CONFIGS[cfg_name]
    orc = oracle_backend(cfg)
    from oracle import ref as R
    scheme, N = cfg["scheme"], cfg["N"]
    ntt = scheme == CKKS
    for cap, B in ((3, batch), (None, batch), (3, 3 * batch)):
        be = GpuBackend(cfg, batch=B)      # fresh context: empty arena
        L = len(be.primes) - 1
        xs = synth.uniform_ct(771 + B, be.primes[:L], 2, N, B)
        c = be.api.Ciphertext.from_numpy(be.ctx, xs, ntt, 1.0, 1, capacity=cap)
        if scheme == CKKS:
            got = be.ev.rescaleToNext(c).cpu()
            op = R.OP_RESCALE_NEXT
        else:
            got = be.ev.modSwitchToNext(c).cpu()
            op = R.OP_MODSWITCH_NEXT
        for b in range(B):
            assert np.array_equal(got[b], orc.impl.eval(op, R.Ct(xs[b], ntt)).data), (cap, B, b)
        # and a second, larger op on the same context right after a small one
        if cap is None:
            xs2 = synth.uniform_ct(779, be.primes[:L], 2, N, 4 * B)
            c2 = be.api.Ciphertext.from_numpy(be.ctx, xs2, ntt, 1.0, 1, capacity=3)
            got2 = (be.ev.rescaleToNext(c2) if scheme == CKKS else be.ev.modSwitchToNext(c2)).cpu()
            for b in range(4 * B):
                assert np.array_equal(got2[b], orc.impl.eval(op, R.Ct(xs2[b], ntt)).data), ("grow", b)


def _is_prime(n):
    if n < 2:
        return False
    for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % q == 0:
            return n == q
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):  # deterministic below 3.3e24
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def check_aux_base(bsk, gamma, golden_level, key_primes, N):
    """The auxiliary base B u {m_sk} of the BEHZ multiplication is internal (troy_amd/csrc/hostmath.cpp, RnsLevel::build: the product's limbs do not
    depend on it).  Either it IS the reference's (TROYHIP_AUX_BASE=reference: equality with the golden file), or it is the library's own choice: as
    many primes as the reference takes, NTT-friendly, pairwise distinct, none of them a key prime, all of one size class (below 2^50 for the FP64
    butterflies or below 2^58 for the guard-free integer ones).  gamma (decryption) is always the reference's."""
    assert str(gamma) == golden_level["gamma"]
    if [str(x) for x in bsk] == golden_level["bsk"]:
        return "reference"
    bsk = [int(x) for x in bsk]
    assert len(bsk) == len(golden_level["bsk"]), "the auxiliary base may not cost an extra limb"
    assert len(set(bsk)) == len(bsk) and not set(bsk) & {int(p) for p in key_primes}
    bits = {p.bit_length() for p in bsk}
    assert bits in ({50}, {58}), bits
    for p in bsk:
        assert (p - 1) % (2 * N) == 0 and _is_prime(p), p
    return "own-%d" % bits.pop()


def check_rescale_onto_itself(cfg, batch=3):
    """troyhip_rescale_to_next / troyhip_mod_switch_to_next called straight through the C ABI with the OUTPUT batch laid over the input
    batch (a strided input -- capacity 3 -- rescaled onto its own buffer with a dense output stride): the rows a later item still reads are
    overwritten by an earlier item's result unless the input is staged.  Result against the same op into a fresh buffer."""
    import ctypes as C
    be = GpuBackend(cfg, batch=batch)
    scheme, N = cfg["scheme"], cfg["N"]
    ntt = scheme == CKKS
    L = len(be.primes) - 1
    xs = synth.uniform_ct(881, be.primes[:L], 2, N, batch)
    fresh = be.api.Ciphertext.from_numpy(be.ctx, xs, ntt, 1.0, 1, capacity=3)
    want = (be.ev.rescaleToNext(fresh) if scheme == CKKS else be.ev.modSwitchToNext(fresh)).cpu()
    c = be.api.Ciphertext.from_numpy(be.ctx, xs, ntt, 1.0, 1, capacity=3)
    si, so = c.struct(), c.struct()
    so.batch_stride = 2 * (L - 1) * N          # dense output items over the strided input items
    fn = be.api.KernelProvider.lib().troyhip_rescale_to_next if scheme == CKKS else be.api.KernelProvider.lib().troyhip_mod_switch_to_next
    capi_mod = __import__("troy_amd.capi", fromlist=["check"])
    capi_mod.check(be.api.KernelProvider.lib(), fn(be.ctx.h, C.byref(si), C.byref(so), C.c_uint64(batch), None))
    be.api.synchronize()
    got = c.buf.to_numpy(batch * 2 * (L - 1) * N).reshape(batch, 2, L - 1, N)
    assert np.array_equal(got, want)


def check_general_sizes(cfg_name, batch=2):
    """GPU product against golden_sizes.json (generated from the reference itself)"""
    import json
    import os
    exp = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_sizes.json")))[cfg_name]
    cfg = CONFIGS[cfg_name]
    out = scenario_sizes(GpuBackend(cfg, batch=batch), cfg)
    assert set(out) == set(exp)
    for k, m in out.items():
        assert sha(m.data) == exp[k]["sha256"] and list(m.data.shape) == exp[k]["shape"], k
        assert m.is_ntt == exp[k]["is_ntt"] and m.cf == exp[k]["cf"] and abs(m.scale - exp[k]["scale"]) <= 1e-12 * abs(exp[k]["scale"]), k


def check_size_limits(cfg_name):
    from troy_amd import capi
    from oracle import ref as R
    cfg = CONFIGS[cfg_name]
    be = GpuBackend(cfg, batch=1)
    orc = oracle_backend(cfg)
    N, L = cfg["N"], len(be.primes) - 1
    ntt = cfg["scheme"] == CKKS
    x9 = synth.uniform_ct(1234, be.primes[:L], 9, N)[0]
    y8 = synth.uniform_ct(1235, be.primes[:L], 8, N)[0]
    a = be.api.Ciphertext.from_numpy(be.ctx, x9[None], ntt, 1.0, 1, capacity=9)
    b = be.api.Ciphertext.from_numpy(be.ctx, y8[None], ntt, 1.0, 1, capacity=8)
    got = be.ev.multiply(a, b)
    assert got.size() == 16
    assert np.array_equal(got.cpu()[0], orc.impl.eval(R.OP_MULTIPLY, R.Ct(x9, ntt), R.Ct(y8, ntt)).data)
    try:
        be.ev.multiply(a, be.api.Ciphertext.from_numpy(be.ctx, x9[None], ntt, 1.0, 1, capacity=9), destination=be.api.Ciphertext(be.ctx, 1, 17, L, capacity=17))
        raise AssertionError("9 x 9 must be rejected")
    except capi.InvalidArgument as e:
        assert "invalid size" in str(e)
    rl = be.api.RelinKeys(be.ctx)
    rl.set(0, synth.uniform_kswitch_key(5, be.primes, N))
    try:
        be.ev.relinearizeInplace(got, rl)
        raise AssertionError("relinearize from size 16 with one key must be rejected")
    except capi.InvalidArgument as e:
        assert "not enough relinearization keys" in str(e)


# ------------------------------------------------------------------ LWE extraction / packing: oracle restatement (limb for limb)
def _orc_shift(orc, ct, shift):
    from oracle import ref as R
    return orc.impl.eval(R.OP_NEGACYCLIC_SHIFT, ct, iarg=shift)


def oracle_extract_lwe(orc, ct, term):
    """EvaluatorCuda::extractLWE (src/evaluator_cuda.cu:2213-2246) on the CPU oracle: (c1 shifted by 2N - term, coefficient `term` of c0)"""
    from oracle import ref as R
    N = ct.data.shape[2]
    if ct.is_ntt:
        ct = orc.impl.eval(R.OP_FROM_NTT, ct)
    c1 = _orc_shift(orc, R.Ct(ct.data[1:2], False, ct.scale, ct.correction_factor), 0 if term == 0 else 2 * N - term)
    return c1.data[0].copy(), ct.data[0, :, term].copy(), ct.scale, ct.correction_factor


def oracle_assemble_lwe(orc, lwe, term):
    """EvaluatorCuda::assembleLWE (:2185-2206)"""
    from oracle import ref as R
    c1, c0, scale, cf = lwe
    L, N = c1.shape
    sh = _orc_shift(orc, R.Ct(c1[None], False, scale, cf), term).data[0]
    data = np.zeros((2, L, N), dtype=np.uint64)
    data[1] = sh
    data[0, :, term] = c0
    return R.Ct(data, False, scale, cf)


def oracle_pack_lwes(orc, lwes, scheme, primes):
    """EvaluatorCuda::packLWECiphertexts + fieldTraceInplace + divideByPolyModulusDegreeInplace (:2248-2340), composed from the
    oracle's negacyclic shift / add / sub / applyGalois / transform ops exactly in the reference's order"""
    from oracle import ref as R
    E = orc.impl
    L, N = lwes[0][0].shape
    ell = 0
    while (1 << ell) < len(lwes):
        ell += 1
    zero = oracle_assemble_lwe(orc, lwes[0], 0)
    zero = R.Ct(np.zeros_like(zero.data), False, zero.scale, zero.correction_factor)

    def div_n(ct):  # kMultiplyInvPolyDegreeCoeffmod: every limb times N^-1 mod q_l
        d = ct.data.copy()
        for l in range(L):
            p = int(primes[l])
            inv = pow(N, -1, p)
            d[:, l] = np.array([(int(v) * inv) % p for v in d[:, l].reshape(-1)], dtype=np.uint64).reshape(d[:, l].shape)
        return R.Ct(d, ct.is_ntt, ct.scale, ct.correction_factor)

    rl = []
    for i in range(1 << ell):
        idx = int(format(i, "0%db" % ell)[::-1], 2) if ell else 0
        rl.append(div_n(oracle_assemble_lwe(orc, lwes[idx], 0)) if idx < len(lwes) else zero)
    for layer in range(ell):
        gap, shift = 1 << layer, N >> (layer + 1)
        for off in range(0, 1 << ell, 2 * gap):
            even, odd = rl[off], rl[off + gap]
            temp = _orc_shift(orc, odd, shift)
            new_odd = E.eval(R.OP_SUB, even, temp)
            even = E.eval(R.OP_ADD, even, temp)
            if scheme == CKKS:
                new_odd = E.eval(R.OP_TO_NTT, new_odd)
            new_odd = E.eval(R.OP_APPLY_GALOIS, new_odd, iarg=(1 << (layer + 1)) + 1)
            if scheme == CKKS:
                new_odd = E.eval(R.OP_FROM_NTT, new_odd)
            rl[off] = E.eval(R.OP_ADD, even, new_odd)
            rl[off + gap] = new_odd
    ret = rl[0]
    degree = N
    while degree > (1 << ell):  # fieldTraceInplace (:2248-2257): applyGalois on `ret` as it is -- for CKKS `ret` is in coefficient form
        t = E.eval(R.OP_APPLY_GALOIS, ret, iarg=degree + 1)  # here, so the reference (and the oracle, and the product) reject it
        ret = E.eval(R.OP_ADD, ret, t)
        degree >>= 1
    if scheme == CKKS:
        ret = E.eval(R.OP_TO_NTT, ret)
    return ret


def check_lwe_limbs(cfg_name, n_lwe=3, batch=2):
    """extractLWE / assembleLWE / packLWECiphertexts / fieldTrace of the product, LIMB FOR LIMB against the oracle
    restatement above (synthetic uniform ciphertexts and Galois keys: the arithmetic is oblivious to key validity)."""
    from oracle import ref as R
    cfg = CONFIGS[cfg_name]
    scheme, N = cfg["scheme"], cfg["N"]
    be = GpuBackend(cfg, batch=batch)
    orc = oracle_backend(cfg)
    primes = be.primes
    L = len(primes) - 1
    ell = 0
    while (1 << ell) < n_lwe:
        ell += 1
    elts = sorted({(1 << k) + 1 for k in range(1, ell + 1)} | {d + 1 for d in [N >> k for k in range(0, 20)] if d > (1 << ell)})
    for i, e in enumerate(elts):
        key = synth.uniform_kswitch_key(SEED + 900 + i, primes, N)
        be.set_galois_key(e, key)
        orc.set_galois_key(e, key)
    ntt = scheme == CKKS
    xs = [synth.uniform_ct(SEED + 800 + i, primes[:L], 2, N, batch) for i in range(n_lwe)]
    terms = [(7 * i + 3) % N for i in range(n_lwe)]
    g_lwes, o_lwes = [], [[] for _ in range(batch)]
    for i in range(n_lwe):
        ct = be.api.Ciphertext.from_numpy(be.ctx, xs[i], ntt, 1.0, 1, capacity=3)
        lwe = be.ev.extractLWE(ct, terms[i])
        g_lwes.append(lwe)
        c1 = lwe.c1.cpu()
        for b in range(batch):
            o = oracle_extract_lwe(orc, R.Ct(xs[i][b], ntt), terms[i])
            assert np.array_equal(c1[b, 0], o[0]) and np.array_equal(np.asarray(lwe.c0)[b], o[1]), ("extract", i, b)
            o_lwes[b].append(o)
        back = be.ev.assembleLWE(lwe, 5).cpu()
        for b in range(batch):
            assert np.array_equal(back[b][:2], oracle_assemble_lwe(orc, o_lwes[b][i], 5).data), ("assemble", i, b)
    if scheme == CKKS:
        # the reference's fieldTraceInplace hands a coefficient-form CKKS ciphertext to applyGalois -> switchKeyInplace, which throws
        # invalid_argument("CKKS encrypted must be in NTT form") (src/evaluator_cuda.cu:1185): same behaviour on all three
        from troy_amd import capi
        for fn, exc in ((lambda: be.ev.packLWECiphertexts(g_lwes, be.gk), capi.InvalidArgument), (lambda: oracle_pack_lwes(orc, o_lwes[0], scheme, primes), Exception)):
            try:
                fn()
                raise AssertionError("packLWECiphertexts must reject the CKKS field trace like the reference")
            except exc as e:
                assert "NTT form" in str(e), str(e)
        return
    packed = be.ev.packLWECiphertexts(g_lwes, be.gk)
    got = packed.cpu()
    for b in range(batch):
        exp = oracle_pack_lwes(orc, o_lwes[b], scheme, primes)
        assert np.array_equal(got[b][:2], exp.data), ("pack", b)
        assert packed.is_ntt_form == exp.is_ntt


def scenario_chain(backend, cfg, depth=3):
    """BASELINE configs[2] as SURVEY.md 8d states it: multiply -> relinearize -> rescale -> rotate(1) chained to depth >= 3, limbs
    exported after EVERY op.  CKKS only."""
    N = cfg["N"]
    primes = backend.primes
    L = len(primes) - 1
    backend.set_relin_key(synth.uniform_kswitch_key(SEED + 1, primes, N))
    backend.set_galois_key(backend.elt_from_step(1), synth.uniform_kswitch_key(SEED + 10, primes, N))
    x = backend.ct(synth.uniform_ct(SEED + 100 + L, primes[:L], 2, N)[0], True)
    out = {}
    depth = min(depth, L - backend.last_limbs)
    for d in range(depth):
        limbs = L - d
        b = backend.ct(synth.uniform_ct(SEED + 200 + limbs, primes[:limbs], 2, N)[0], True)
        x = backend.multiply(x, b)
        out[f"chain{d}/multiply"] = backend.export(x)
        x = backend.relinearize(x)
        out[f"chain{d}/relinearize"] = backend.export(x)
        x = backend.rescale(x)
        out[f"chain{d}/rescale"] = backend.export(x)
        x = backend.rotate(x, 1)
        out[f"chain{d}/rotate1"] = backend.export(x)
    return out


def check_chain(cfg_name, batch=1, depth=3):
    import json
    import os
    exp = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_chain.json")))[cfg_name]
    cfg = CONFIGS[cfg_name]
    out = scenario_chain(GpuBackend(cfg, batch=batch), cfg, depth)
    assert set(out) == set(exp)
    for k, m in out.items():
        assert sha(m.data) == exp[k]["sha256"] and list(m.data.shape) == exp[k]["shape"], k
        assert m.is_ntt == exp[k]["is_ntt"] and abs(m.scale - exp[k]["scale"]) <= 1e-9 * abs(exp[k]["scale"]), k


def check_distinct_batch_relin_rotate(cfg_name, batch=8):
    """BASELINE configs[3] shape: `batch` DISTINCT size-3 ciphertexts, relinearize + rotateRows(1), every item against the oracle"""
    from oracle import ref as R
    cfg = CONFIGS[cfg_name]
    be = GpuBackend(cfg, batch=batch)
    orc = oracle_backend(cfg)
    N, primes = cfg["N"], be.primes
    L = len(primes) - 1
    rk = synth.uniform_kswitch_key(SEED + 1, primes, N)
    gk = synth.uniform_kswitch_key(SEED + 10, primes, N)
    e1 = be.elt_from_step(1)
    be.set_relin_key(rk)
    be.set_galois_key(e1, gk)
    orc.set_relin_key(rk)
    orc.set_galois_key(e1, gk)
    ntt = cfg["scheme"] == CKKS
    xs = synth.uniform_ct(SEED + 1234, primes[:L], 3, N, batch)
    c = be.api.Ciphertext.from_numpy(be.ctx, xs, ntt, 1.0, 1, capacity=3)
    be.ev.relinearizeInplace(c, be.rlk)
    be.rotate(c, 1)
    got = c.cpu()
    op = R.OP_ROTATE_VECTOR if ntt else R.OP_ROTATE_ROWS
    for b in range(batch):
        exp = orc.impl.eval(op, orc.impl.eval(R.OP_RELIN, R.Ct(xs[b], ntt)), iarg=1)
        assert np.array_equal(got[b][:2], exp.data), b


# ------------------------------------------------------------------ device-side scalar arithmetic
def check_device_modarith(api, kat):
    """modarith.h / bfly.h on the device through troyhip_test_modarith: the reference's own known-answer vectors
    (tests/golden/kat_reference_tests.json, test/utils/uintarithsmallmod.cpp) and edge values -- operands 0, 1, p-1, lazy inputs
    2p-1 / 4p-1 / 8p-1, primes of 36, 58, 60 and 61 bits -- against exact integer arithmetic."""
    import ctypes as C
    from troy_amd import capi
    lib = api.KernelProvider._lib if getattr(api.KernelProvider, "_lib", None) is not None else capi.load()

    def run(op, a, b, c, p, n_out, aux=1):
        bufs = [api.DeviceBuffer.from_numpy(np.array(v, dtype=np.uint64)) if v is not None else None for v in (a, b, c)]
        out = api.DeviceBuffer(n_out)
        capi.check(lib, lib.troyhip_test_modarith(op, *[C.c_void_p(x.ptr) if x is not None else None for x in bufs], C.c_uint64(p), C.c_uint64(aux),
                                                  C.c_void_p(out.ptr), C.c_uint64(len(a)), None))
        return [int(v) for v in out.to_numpy()]

    # --- reference KATs
    for p, lo, hi, exp in kat["barrett_reduce_128"]["cases"]:
        if int(p) >> 61:
            continue
        assert run(1, [int(lo)], [int(hi)], None, int(p), 1) == [int(exp)]
    for p, x, y, exp in kat["multiply_uint_mod"]["cases"]:
        assert run(2, [int(x)], [int(y)], None, int(p), 1) == [int(exp)]
    for p, x, w, exp in kat["multiply_uint_mod_lazy"]["cases"]:
        p, x, w = int(p), int(x), int(w)
        r = run(4, [x], [w], [(w << 64) // p], p, 1)[0]
        assert r == int(exp)  # the exact lazy value of the reference's multiplyUIntModLazy (in [0, 2p))
    # --- edge values
    rng = np.random.default_rng(11)
    M64 = (1 << 64) - 1
    for p in (2305843009213554689, 1152921504606830593, 288230376150630401, 68718428161, 1152921504606584833):
        assert p < 1 << 61
        lazy = [0, 1, p - 1, p, p + 1, 2 * p - 1, 2 * p, 4 * p - 1, 4 * p, min(8 * p - 1, M64), M64, M64 - 1] + [int(v) for v in rng.integers(0, 1 << 63, 20, dtype=np.uint64) * 2 + 1]
        lazy = [v & M64 for v in lazy]
        res = [v % p for v in lazy]
        assert run(0, lazy, None, None, p, len(lazy)) == res
        pairs = [(x % p, y % p) for x in lazy[:12] for y in (0, 1, p - 1, (p - 1) // 2, lazy[14] % p)]
        xs, ys = [a for a, _ in pairs], [b for _, b in pairs]
        assert run(2, xs, ys, None, p, len(xs)) == [(a * b) % p for a, b in pairs]
        assert run(5, xs, ys, None, p, len(xs)) == [(a * b) % p for a, b in pairs]
        prods = [a * b + (p - 1) * (p - 1) for a, b in pairs]  # up to 2 p^2: the lazy-sum range of barrett128 callers
        assert run(1, [v & M64 for v in prods], [v >> 64 for v in prods], None, p, len(prods)) == [v % p for v in prods]
        ws = [w for w in (0, 1, p - 1, p // 3, lazy[15] % p)]
        for w in ws:
            q = (w << 64) // p
            got = run(3, lazy, [w] * len(lazy), [q] * len(lazy), p, len(lazy))
            assert got == [(x * w) % p for x in lazy], (p, w)
            gl = run(4, lazy, [w] * len(lazy), [q] * len(lazy), p, len(lazy))
            assert all(g % p == (x * w) % p and g < 2 * p for g, x in zip(gl, lazy))
        # butterflies: forward inputs below 8p (4p for the 61-bit class is 2^63 > ...: 8p < 2^64 always), inverse below 4p
        top = min(8 * p - 1, M64)
        fx = [0, p - 1, 2 * p - 1, 4 * p - 1, top, 4 * p, p, 1] + [int(v) % (8 * p) & M64 for v in rng.integers(0, 1 << 63, 24, dtype=np.uint64) * 2]
        fy = fx[::-1]
        tw = [(7 * i * i + 3) % p for i in range(len(fx))]
        tw[0], tw[1], tw[2] = p - 1, 1, 0
        expf = []
        for x, y, w in zip(fx, fy, tw):
            expf += [(x + w * y) % p, (x - w * y) % p]
        assert run(6, fx, fy, tw, p, 2 * len(fx)) == expf, ("ct_bfly4", p)
        uni = [(x + tw[3] * y) % p if j == 0 else (x - tw[3] * y) % p for x, y in zip(fx, fy) for j in range(2)]
        assert run(11, fx, fy, [tw[3]] + tw[1:], p, 2 * len(fx)) == uni, ("ct_bfly4<uniform>", p)
        if p < 1 << 58:
            big = [v % (56 * p) for v in rng.integers(0, 1 << 63, len(fx), dtype=np.uint64).astype(object) * 2]  # guard-free: inputs up to 56p, outputs + 3p < 2^64
            big = [int(v) for v in big]
            expn = []
            for x, y, w in zip(big, big[::-1], tw):
                expn += [(x + w * y) % p, (x - w * y) % p]
            assert run(7, big, big[::-1], tw, p, 2 * len(fx)) == expn, ("ct_bfly4_ng", p)
            unn = [(x + tw[3] * y) % p if j == 0 else (x - tw[3] * y) % p for x, y in zip(big, big[::-1]) for j in range(2)]
            assert run(12, big, big[::-1], [tw[3]] + tw[1:], p, 2 * len(fx)) == unn, ("ct_bfly4_ng<uniform>", p)
            g8 = [v % (8 * p) for v in big]
            expg = []
            for x, y, w in zip(g8, g8[::-1], tw):
                expg += [(x + y) % p, ((x - y) * w) % p]
            assert run(13, g8, g8[::-1], tw, p, 2 * len(fx)) == expg, ("gs_bfly4_ng", p)
        ix = [v % (4 * p) for v in fx]
        iy = ix[::-1]
        expi = []
        for x, y, w in zip(ix, iy, tw):
            expi += [(x + y) % p, ((x - y) * w) % p]
        assert run(8, ix, iy, tw, p, 2 * len(ix)) == expi, ("gs_bfly4", p)
        ninv = pow(32768, -1, p)
        expl = []
        for x, y, w in zip(ix, iy, tw):
            expl += [((x + y) * ninv) % p, ((x - y) * w) % p]
        assert run(9, ix, iy, tw, p, 2 * len(ix), aux=ninv) == expl, ("gs_bfly4_last", p)
        # 128-bit multiply-accumulate: 7 terms per accumulator, lazy operands below 8p where 7 * 8p * p < 2^128 (the evaluator's
        # `lazy` condition, evaluator.cpp switch_key), canonical ones for the 61-bit class
        n = 28
        bound = 8 * p if p < (1 << 60) else p
        ma = [int(v) % bound for v in rng.integers(0, 1 << 63, n, dtype=np.uint64).astype(object) * 2]
        mb = [int(v) % p for v in rng.integers(0, 1 << 62, n, dtype=np.uint64)]
        ma[:4] = [bound - 1] * 4
        mb[:4] = [p - 1] * 4
        assert run(10, ma, mb, None, p, 1) == [sum(x * y for x, y in zip(ma, mb)) % p], ("mac128x4", p)


def check_save_load(api):
    """CiphertextCuda::save / load wire format (src/ciphertext_cuda.cu:16-104): field layout, parms_id check, round trip"""
    import io
    import struct
    N = 64
    primes = api.CoeffModulus.Create(N, [40, 40, 40])
    ctx = api.SEALContext(api.CKKS, N, primes, 0)
    x = synth.uniform_ct(3, primes[:2], 2, N, 2)
    c = api.Ciphertext.from_numpy(ctx, x, True, 2.0 ** 20, 1)
    s = io.BytesIO()
    c.save(s, index=1)
    blob = s.getvalue()
    assert len(blob) == 32 + 1 + 8 * 3 + 8 + 8 + 8 + 1 + 8 + 2 * 2 * N * 8
    assert struct.unpack_from("<?QQQd", blob, 32) == (True, 2, N, 2, 2.0 ** 20)
    # the whole blob against an independent construction: the REFERENCE's parms_id of this level (tests/golden/golden_wire.json, from oracle/_ref), the
    # fields in the order of src/ciphertext_cuda.cu:16-25, the word count, the polynomials as uploaded ([poly][limb][N]); a permuted payload differs
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_wire.json")))["ckks"]
    assert [int(p) for p in primes] == gold["primes"]
    head = np.array(gold["parms_id"]["2"], dtype=np.uint64).tobytes() + struct.pack("<?QQQdQQ?Q", True, 2, N, 2, 2.0 ** 20, 1, 0, False, 2 * 2 * N)
    assert blob == head + np.ascontiguousarray(x[1]).tobytes()
    assert blob != head + np.ascontiguousarray(x[1][:, ::-1]).tobytes()
    back = api.Ciphertext.load(ctx, io.BytesIO(blob))
    assert np.array_equal(back.cpu()[0], x[1]) and back.is_ntt_form and back.scale == 2.0 ** 20
    other = api.SEALContext(api.CKKS, N, api.CoeffModulus.Create(N, [40, 30, 40]), 0)
    try:
        api.Ciphertext.load(other, io.BytesIO(blob))
        raise AssertionError("a blob of other parameters must be rejected")
    except ValueError:
        pass


def check_save_terms(api):
    """CiphertextCuda::saveTerms / loadTerms (src/ciphertext_cuda.cu:44-80, 106-143): c0 keeps only the listed coefficients (in
    coefficient form, [term][limb]), c1 travels whole; NTT-form ciphertexts are transformed on both sides"""
    import io
    import struct
    N = 64
    primes = api.CoeffModulus.Create(N, [40, 40, 40])
    ctx = api.SEALContext(api.CKKS, N, primes, 0)
    ev = api.Evaluator(ctx)
    x = synth.uniform_ct(5, primes[:2], 2, N, 1)
    c = api.Ciphertext.from_numpy(ctx, x, True, 2.0 ** 20, 1)
    terms = [0, 3, 17, 63]
    s = io.BytesIO()
    c.saveTerms(s, ev, terms)
    blob = s.getvalue()
    head = 32 + 1 + 8 * 3 + 8 + 8 + 8 + 1
    assert len(blob) == head + len(terms) * 2 * 8 + 8 + 2 * N * 8
    assert blob[head - 1] == 1  # terms flag
    coeff = ev.transformFromNtt(c).cpu()[0]
    assert np.array_equal(np.frombuffer(blob[head:head + 16], dtype=np.uint64), coeff[0][:, 0])
    assert struct.unpack_from("<Q", blob, head + len(terms) * 16) == (2 * N,)
    back = api.Ciphertext.loadTerms(ctx, io.BytesIO(blob), ev, terms)
    assert back.is_ntt_form and back.scale == 2.0 ** 20 and back.size() == 2
    got = ev.transformFromNtt(back).cpu()[0]
    assert np.array_equal(got[1], coeff[1])
    assert np.array_equal(got[0][:, terms], coeff[0][:, terms])
    mask = np.ones(N, dtype=bool)
    mask[terms] = False
    assert not got[0][:, mask].any()
    for bad in (lambda: api.Ciphertext.load(ctx, io.BytesIO(blob)),):  # a termed stream needs the indices
        try:
            bad()
            raise AssertionError("expected ValueError")
        except ValueError:
            pass
    full = io.BytesIO()
    c.save(full)
    try:
        api.Ciphertext.loadTerms(ctx, io.BytesIO(full.getvalue()), ev, terms)
        raise AssertionError("expected ValueError")
    except ValueError:
        pass


def check_moddown_shared_first_pass(scheme, N=8192, bits=(46, 46, 46, 48), tbits=17, batch=1, seed=31, expect_fp_two_pass=True):
    """the key-switch mod-down of a SMALL launch (one first inverse pass over the special limb and the data limbs, the last passes launched on their
    own) with an FP64-class special prime LARGER than the FP64-class data primes: the separately launched last passes must plan their reductions
    from the bound the shared first pass left, i.e. from the largest prime of all the slots (round-4 advisor finding).  Uniform and extreme rows."""
    from oracle import ref
    from troy_amd import capi, synth
    cfg = {"scheme": scheme, "N": N, "bits": list(bits), "tbits": tbits}
    be, ob = GpuBackend(cfg, batch=batch), oracle_backend(cfg)
    L = len(be.primes) - 1
    assert be.primes[-1] > max(be.primes[:L]) and be.primes[-1] < (1 << 50)
    ntt = scheme == CKKS
    top_key = np.stack([np.stack([np.stack([np.full(N, p - 1, dtype=np.uint64) for p in be.primes])] * 2)] * L)
    for kind, rk in (("uniform", synth.uniform_kswitch_key(seed, be.primes, N)), ("extreme", top_key)):
        be.set_relin_key(rk)
        ob.set_relin_key(rk)
        q = be.primes[:L]
        rows = [synth.uniform_ct(seed + 1, q, 3, N, 1)[0],
                np.stack([np.stack([np.full(N, p - 1, dtype=np.uint64) for p in q])] * 3),
                np.stack([np.stack([(np.arange(N, dtype=np.uint64) & np.uint64(1)) * np.uint64(p - 1) for p in q])] * 3)]
        for x in rows:
            before = capi.stat("ntt2_fp_launches", be.api.KernelProvider._lib)
            got = be.ev.relinearize(be.ct(x, ntt), be.rlk).cpu()[0]
            exp = ob.relinearize(ref.Ct(x, ntt))
            assert np.array_equal(got, exp.data), (kind, cfg)
            if expect_fp_two_pass:
                assert capi.stat("ntt2_fp_launches", be.api.KernelProvider._lib) > before, "the FP64 two-pass kernels did not run"
