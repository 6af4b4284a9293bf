"""ctypes view of oracle/_ref/libtroyref_driver.so -- the REAL reference CPU path (troy_cpu.h).

TEST INFRASTRUCTURE ONLY.  Used by tests/golden/gen_golden.py (in the build container, where
/root/reference exists) to produce golden vectors, by tests to pin oracle/troy_oracle.cpp, and by
bench.py's cpu_baseline leg (kind "reference") when the prebuilt .so travelled to the GPU box.
The product path (troy_amd/) never imports this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libtroyref_driver.so")

BFV, CKKS, BGV = 1, 2, 3
(OP_ADD, OP_SUB, OP_NEGATE, OP_MULTIPLY, OP_SQUARE, OP_RELIN, OP_MODSWITCH_NEXT, OP_RESCALE_NEXT,
 OP_APPLY_GALOIS, OP_ROTATE_ROWS, OP_ROTATE_COLUMNS, OP_ROTATE_VECTOR, OP_CONJUGATE, OP_TO_NTT,
 OP_FROM_NTT, OP_MULTIPLY_PLAIN_NTT, OP_ADD_PLAIN, OP_SUB_PLAIN, OP_MULTIPLY_PLAIN, OP_APPLY_KEYSWITCH, OP_NEGACYCLIC_SHIFT) = range(21)
(ST_FASTBCONV_MTILDE, ST_SMMRQ, ST_FASTFLOOR, ST_FASTBCONV_SK, ST_DIVROUND_QLAST, ST_DIVROUND_QLAST_NTT,
 ST_MODT_DIV_QLAST) = range(7)


class CtDesc(C.Structure):
    _fields_ = [("limbs", C.c_int), ("size", C.c_int), ("is_ntt", C.c_int), ("scale", C.c_double),
                ("correction_factor", C.c_uint64)]


def available():
    return os.path.exists(_SO)


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(_SO)
        l.ref_create.restype = C.c_void_p
        l.ref_last_error.restype = C.c_char_p
        l.ref_plain_batching.restype = C.c_uint64
        l.ref_galois_elt_from_step.restype = C.c_uint32
        l.ref_time_mul_relin.restype = C.c_double
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def coeff_modulus_create(N, bits):
    out = np.zeros(len(bits), dtype=np.uint64)
    b = (C.c_int * len(bits))(*bits)
    if lib().ref_coeff_modulus_create(C.c_uint64(N), b, len(bits), _p(out)) != 0:
        raise RuntimeError("CoeffModulus::Create failed")
    return [int(x) for x in out]


def plain_batching(N, bits):
    return int(lib().ref_plain_batching(C.c_uint64(N), bits))


def modulus_const_ratio(p):
    out = np.zeros(3, dtype=np.uint64)
    lib().ref_modulus_const_ratio(C.c_uint64(p), _p(out))
    return [int(x) for x in out]


class Ct:
    """A raw ciphertext: data uint64 [size][limbs][N] + the reference's metadata."""

    def __init__(self, data, is_ntt=False, scale=1.0, correction_factor=1):
        self.data = np.ascontiguousarray(data, dtype=np.uint64)
        assert self.data.ndim == 3
        self.is_ntt, self.scale, self.correction_factor = bool(is_ntt), float(scale), int(correction_factor)

    @property
    def size(self):
        return self.data.shape[0]

    @property
    def limbs(self):
        return self.data.shape[1]

    def desc(self):
        return CtDesc(self.limbs, self.size, int(self.is_ntt), self.scale, self.correction_factor)


class Ref:
    def __init__(self, scheme, N, primes, t=0, seed=1):
        self.scheme, self.N, self.primes, self.t = scheme, N, [int(p) for p in primes], int(t)
        self.K = len(primes)
        arr = np.array(self.primes, dtype=np.uint64)
        self.h = lib().ref_create(scheme, C.c_uint64(N), _p(arr), self.K, C.c_uint64(self.t), C.c_uint64(seed))
        if not self.h:
            raise RuntimeError("ref_create failed")
        self.h = C.c_void_p(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("reference threw: " + lib().ref_last_error(self.h).decode())

    def chain(self):
        a, b = C.c_int(), C.c_int()
        n = lib().ref_chain(self.h, C.byref(a), C.byref(b))
        return n, a.value, b.value

    def ntt_tables(self, prime_idx):
        N = self.N
        rop, rquo, iop, iquo = (np.zeros(N, dtype=np.uint64) for _ in range(4))
        invd = np.zeros(2, dtype=np.uint64)
        root = C.c_uint64()
        lib().ref_ntt_tables(self.h, prime_idx, _p(rop), _p(rquo), _p(iop), _p(iquo), _p(invd), C.byref(root))
        return dict(root=root.value, root_op=rop, root_quo=rquo, inv_op=iop, inv_quo=iquo, inv_degree=invd)

    def behz_bases(self, limbs):
        out = np.zeros(limbs + 3, dtype=np.uint64)
        g = C.c_uint64()
        n = lib().ref_behz_bases(self.h, limbs, _p(out), C.byref(g))
        if n < 0:
            raise RuntimeError("no such level")
        return [int(x) for x in out[:n]], g.value

    def bsk_ntt_tables(self, limbs, idx):
        rop, iop = np.zeros(self.N, dtype=np.uint64), np.zeros(self.N, dtype=np.uint64)
        invd = C.c_uint64()
        lib().ref_bsk_ntt_tables(self.h, limbs, idx, _p(rop), _p(iop), C.byref(invd))
        return rop, iop, invd.value

    def ntt(self, prime_idx, limb, mode):
        d = np.ascontiguousarray(limb, dtype=np.uint64).copy()
        lib().ref_ntt(self.h, prime_idx, _p(d), mode)
        return d

    def rns_stage(self, limbs, stage, inp, out_limbs):
        inp = np.ascontiguousarray(inp, dtype=np.uint64)
        out = np.zeros((out_limbs, self.N), dtype=np.uint64)
        self._chk(lib().ref_rns_stage(self.h, limbs, stage, _p(inp), _p(out)))
        return out

    def keygen(self, galois_elts=()):
        e = np.array(list(galois_elts), dtype=np.uint32)
        self._chk(lib().ref_keygen(self.h, _p(e), len(e)))

    def set_secret_key(self, sk):
        sk = np.ascontiguousarray(sk, dtype=np.uint64)
        self._chk(lib().ref_set_secret_key(self.h, _p(sk)))

    def set_public_key(self, pk):
        pk = np.ascontiguousarray(pk, dtype=np.uint64)
        self._chk(lib().ref_set_public_key(self.h, _p(pk)))

    def secret_key(self):
        out = np.zeros((self.K, self.N), dtype=np.uint64)
        lib().ref_get_secret_key(self.h, _p(out))
        return out

    def public_key(self):
        out = np.zeros((2, self.K, self.N), dtype=np.uint64)
        lib().ref_get_public_key(self.h, _p(out))
        return out

    def relin_key(self):
        out = np.zeros((self.K - 1, 2, self.K, self.N), dtype=np.uint64)
        lib().ref_get_relin_key(self.h, _p(out))
        return out

    def galois_key(self, elt):
        out = np.zeros((self.K - 1, 2, self.K, self.N), dtype=np.uint64)
        if lib().ref_get_galois_key(self.h, C.c_uint32(elt), _p(out)) != 0:
            raise KeyError(elt)
        return out

    def set_kswitch_key(self, which, data):
        data = np.ascontiguousarray(data, dtype=np.uint64)
        assert data.shape == (self.K - 1, 2, self.K, self.N)
        self._chk(lib().ref_set_kswitch_key(self.h, C.c_uint32(which), _p(data)))

    def elt_from_step(self, step):
        return int(lib().ref_galois_elt_from_step(self.h, step))

    def eval(self, op, a, b=None, iarg=0, out_size=None):
        ad = a.desc()
        if isinstance(b, Ct):
            bd, bptr = b.desc(), _p(b.data)
            bdp = C.byref(bd)
        elif b is not None:  # plaintext array
            barr = np.ascontiguousarray(b, dtype=np.uint64)
            bd = CtDesc(a.limbs, 1, 1, 1.0, 1)
            bdp, bptr = C.byref(bd), _p(barr)
        else:
            bdp, bptr = None, None
        od = CtDesc()
        osz = a.size + (b.size if isinstance(b, Ct) else 1)  # a product has a.size + b.size - 1 polynomials
        out = np.zeros((osz + 1, a.limbs, self.N), dtype=np.uint64).reshape(-1)
        self._chk(lib().ref_eval(self.h, op, C.byref(ad), _p(a.data), bdp, bptr, C.c_int64(iarg), C.byref(od), _p(out)))
        n = od.size * od.limbs * self.N
        return Ct(out[:n].reshape(od.size, od.limbs, self.N).copy(), od.is_ntt, od.scale, od.correction_factor)

    def parms_id(self, limbs):
        out = np.zeros(4, dtype=np.uint64)
        self._chk(lib().ref_parms_id(self.h, limbs, _p(out)))
        return [int(x) for x in out]

    def plain_to_ntt(self, plain, limbs):
        plain = np.ascontiguousarray(plain, dtype=np.uint64)
        out = np.zeros((limbs, self.N), dtype=np.uint64)
        self._chk(lib().ref_plain_to_ntt(self.h, _p(plain), len(plain), limbs, _p(out)))
        return out

    def encrypt(self, plain):
        plain = np.ascontiguousarray(plain, dtype=np.uint64)
        od = CtDesc()
        out = np.zeros(2 * self.K * self.N, dtype=np.uint64)
        self._chk(lib().ref_encrypt(self.h, _p(plain), len(plain), C.byref(od), _p(out)))
        n = od.size * od.limbs * self.N
        return Ct(out[:n].reshape(od.size, od.limbs, self.N).copy(), od.is_ntt, od.scale, od.correction_factor)

    def decrypt(self, ct):
        out = np.zeros(self.N, dtype=np.uint64)
        budget = C.c_int()
        d = ct.desc()
        self._chk(lib().ref_decrypt(self.h, C.byref(d), _p(ct.data), _p(out), C.byref(budget)))
        return out, budget.value

    def batch_encode(self, values):
        v = np.ascontiguousarray(values, dtype=np.uint64)
        out = np.zeros(self.N, dtype=np.uint64)
        self._chk(lib().ref_batch_encode(self.h, _p(v), _p(out)))
        return out

    def batch_decode(self, plain):
        p = np.ascontiguousarray(plain, dtype=np.uint64)
        out = np.zeros(self.N, dtype=np.uint64)
        self._chk(lib().ref_batch_decode(self.h, _p(p), _p(out)))
        return out

    def ckks_encode(self, slots, scale, limbs):
        s = np.ascontiguousarray(np.asarray(slots, dtype=np.complex128)).view(np.float64)
        out = np.zeros((limbs, self.N), dtype=np.uint64)
        self._chk(lib().ref_ckks_encode(self.h, _p(s), C.c_double(scale), limbs, _p(out)))
        return out

    def ckks_encrypt(self, plain, scale):
        plain = np.ascontiguousarray(plain, dtype=np.uint64)
        limbs = plain.shape[0]
        od = CtDesc()
        out = np.zeros(2 * self.K * self.N, dtype=np.uint64)
        self._chk(lib().ref_ckks_encrypt(self.h, _p(plain), C.c_double(scale), limbs, C.byref(od), _p(out)))
        n = od.size * od.limbs * self.N
        return Ct(out[:n].reshape(od.size, od.limbs, self.N).copy(), od.is_ntt, od.scale, od.correction_factor)

    def ckks_decrypt_decode(self, ct):
        out = np.zeros(self.N, dtype=np.float64)
        d = ct.desc()
        self._chk(lib().ref_ckks_decrypt_decode(self.h, C.byref(d), _p(ct.data), _p(out)))
        return out.view(np.complex128).copy()

    def time_mul_relin(self, a, b, reps):
        d = a.desc()
        t = lib().ref_time_mul_relin(self.h, C.byref(d), _p(a.data), _p(b.data), reps)
        if t < 0:
            self._chk(-1)
        return t
