"""ctypes view of oracle/libtroy_oracle.so -- our CPU restatement of the reference's CPU path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product path (troy_amd/) never imports this module.  Same surface as oracle/ref.py so the
tests can run one against the other.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .ref import (BFV, BGV, CKKS, Ct, CtDesc, OP_ADD, OP_APPLY_GALOIS, OP_CONJUGATE, OP_FROM_NTT,  # noqa: F401
                  OP_MODSWITCH_NEXT, OP_MULTIPLY, OP_MULTIPLY_PLAIN_NTT, OP_ADD_PLAIN, OP_SUB_PLAIN, OP_MULTIPLY_PLAIN, OP_APPLY_KEYSWITCH, OP_NEGACYCLIC_SHIFT, OP_NEGATE, OP_RELIN, OP_RESCALE_NEXT,
                  OP_ROTATE_COLUMNS, OP_ROTATE_ROWS, OP_ROTATE_VECTOR, OP_SQUARE, OP_SUB, OP_TO_NTT,
                  ST_DIVROUND_QLAST, ST_DIVROUND_QLAST_NTT, ST_FASTBCONV_MTILDE, ST_FASTBCONV_SK, ST_FASTFLOOR,
                  ST_MODT_DIV_QLAST, ST_SMMRQ)

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtroy_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        l = C.CDLL(_SO)
        l.orc_create.restype = C.c_void_p
        l.orc_last_error.restype = C.c_char_p
        for f in ("orc_barrett_reduce_64", "orc_barrett_reduce_128", "orc_multiply_uint_mod", "orc_shoup_quotient",
                  "orc_multiply_uint_mod_lazy", "orc_exponentiate_uint_mod", "orc_dot_product_mod", "orc_plain_batching"):
            getattr(l, f).restype = C.c_uint64
        l.orc_galois_elt_from_step.restype = C.c_uint32
        l.orc_time_mul_relin.restype = C.c_double
        l.orc_time_ntt.restype = C.c_double
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


U = C.c_uint64


# ---- scalar entry points ----
def barrett_reduce_64(x, p):
    return int(lib().orc_barrett_reduce_64(U(x), U(p)))


def barrett_reduce_128(lo, hi, p):
    return int(lib().orc_barrett_reduce_128(U(lo), U(hi), U(p)))


def multiply_uint_mod(a, b, p):
    return int(lib().orc_multiply_uint_mod(U(a), U(b), U(p)))


def shoup_quotient(w, p):
    return int(lib().orc_shoup_quotient(U(w), U(p)))


def multiply_uint_mod_lazy(x, w, p):
    return int(lib().orc_multiply_uint_mod_lazy(U(x), U(w), U(p)))


def exponentiate_uint_mod(a, e, p):
    return int(lib().orc_exponentiate_uint_mod(U(a), U(e), U(p)))


def try_invert_uint_mod(a, p):
    out = U()
    ok = lib().orc_try_invert_uint_mod(U(a), U(p), C.byref(out))
    return (bool(ok), out.value)


def dot_product_mod(a, b, p):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    return int(lib().orc_dot_product_mod(_p(a), _p(b), len(a), U(p)))


def is_prime(p):
    return bool(lib().orc_is_prime(U(p)))


def try_minimal_primitive_root(degree, p):
    out = U()
    ok = lib().orc_try_minimal_primitive_root(U(degree), U(p), C.byref(out))
    return (bool(ok), out.value)


def modulus_const_ratio(p):
    out = np.zeros(3, dtype=np.uint64)
    lib().orc_modulus_const_ratio(U(p), _p(out))
    return [int(x) for x in out]


def naf(value):
    out = (C.c_int * 64)()
    n = C.c_int()
    lib().orc_naf(value, out, C.byref(n))
    return [out[i] for i in range(n.value)]


def get_primes(factor, bits, count):
    out = np.zeros(count, dtype=np.uint64)
    if lib().orc_get_primes(U(factor), bits, count, _p(out)) != 0:
        raise RuntimeError("failed to find enough qualifying primes")
    return [int(x) for x in out]


def coeff_modulus_create(N, bits):
    out = np.zeros(len(bits), dtype=np.uint64)
    b = (C.c_int * len(bits))(*bits)
    if lib().orc_coeff_modulus_create(U(N), b, len(bits), _p(out)) != 0:
        raise RuntimeError("CoeffModulus::Create failed")
    return [int(x) for x in out]


def plain_batching(N, bits):
    v = int(lib().orc_plain_batching(U(N), bits))
    if v == 0:
        raise ValueError("failed to find enough qualifying primes")
    return v


def ntt_standalone(N, p, data, mode):
    d = np.ascontiguousarray(data, dtype=np.uint64).copy()
    if lib().orc_ntt_standalone(U(N), U(p), _p(d), mode) != 0:
        raise RuntimeError("invalid modulus")
    return d


def apply_galois(N, elt, p, data):
    d = np.ascontiguousarray(data, dtype=np.uint64)
    out = np.zeros_like(d)
    lib().orc_apply_galois(U(N), C.c_uint32(elt), U(p), _p(d), _p(out))
    return out


def apply_galois_ntt(N, elt, data):
    d = np.ascontiguousarray(data, dtype=np.uint64)
    out = np.zeros_like(d)
    lib().orc_apply_galois_ntt(U(N), C.c_uint32(elt), _p(d), _p(out))
    return out


class Oracle:
    def __init__(self, scheme, N, primes, t=0, seed=None):
        self.scheme, self.N, self.primes, self.t = scheme, N, [int(p) for p in primes], int(t)
        self.K = len(primes)
        arr = np.array(self.primes, dtype=np.uint64)
        h = lib().orc_create(scheme, U(N), _p(arr), self.K, U(self.t))
        if not h:
            raise RuntimeError("orc_create failed")
        self.h = C.c_void_p(h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("oracle threw: " + lib().orc_last_error(self.h).decode())

    def chain(self):
        a, b = C.c_int(), C.c_int()
        n = lib().orc_chain(self.h, C.byref(a), C.byref(b))
        return n, a.value, b.value

    def ntt_tables(self, prime_idx):
        N = self.N
        rop, rquo, iop, iquo = (np.zeros(N, dtype=np.uint64) for _ in range(4))
        invd = np.zeros(2, dtype=np.uint64)
        root = U()
        lib().orc_ntt_tables(self.h, prime_idx, _p(rop), _p(rquo), _p(iop), _p(iquo), _p(invd), C.byref(root))
        return dict(root=root.value, root_op=rop, root_quo=rquo, inv_op=iop, inv_quo=iquo, inv_degree=invd)

    def behz_bases(self, limbs):
        out = np.zeros(limbs + 3, dtype=np.uint64)
        g = U()
        n = lib().orc_behz_bases(self.h, limbs, _p(out), C.byref(g))
        if n < 0:
            raise RuntimeError("no such level")
        return [int(x) for x in out[:n]], g.value

    def bsk_ntt_tables(self, limbs, idx):
        rop, iop = np.zeros(self.N, dtype=np.uint64), np.zeros(self.N, dtype=np.uint64)
        invd = U()
        lib().orc_bsk_ntt_tables(self.h, limbs, idx, _p(rop), _p(iop), C.byref(invd))
        return rop, iop, invd.value

    def ntt(self, prime_idx, limb, mode):
        d = np.ascontiguousarray(limb, dtype=np.uint64).copy()
        lib().orc_ntt(self.h, prime_idx, _p(d), mode)
        return d

    def rns_stage(self, limbs, stage, inp, out_limbs):
        inp = np.ascontiguousarray(inp, dtype=np.uint64)
        out = np.zeros((out_limbs, self.N), dtype=np.uint64)
        self._chk(lib().orc_rns_stage(self.h, limbs, stage, _p(inp), _p(out)))
        return out

    def set_kswitch_key(self, which, data):
        data = np.ascontiguousarray(data, dtype=np.uint64)
        assert data.shape == (self.K - 1, 2, self.K, self.N)
        self._chk(lib().orc_set_kswitch_key(self.h, C.c_uint32(which), _p(data)))

    def elt_from_step(self, step):
        return int(lib().orc_galois_elt_from_step(self.h, step))

    def eval(self, op, a, b=None, iarg=0):
        ad = a.desc()
        if isinstance(b, Ct):
            bd = b.desc()
            bdp, bptr = C.byref(bd), _p(b.data)
        elif b is not None:
            barr = np.ascontiguousarray(b, dtype=np.uint64)
            bd = CtDesc(a.limbs, 1, 1, 1.0, 1)
            bdp, bptr = C.byref(bd), _p(barr)
        else:
            bdp, bptr = None, None
        od = CtDesc()
        osz = a.size + (b.size if isinstance(b, Ct) else 1)  # a product has a.size + b.size - 1 polynomials
        out = np.zeros((osz + 1) * a.limbs * self.N, dtype=np.uint64)
        self._chk(lib().orc_eval(self.h, op, C.byref(ad), _p(a.data), bdp, bptr, C.c_int64(iarg), C.byref(od), _p(out)))
        n = od.size * od.limbs * self.N
        return Ct(out[:n].reshape(od.size, od.limbs, self.N).copy(), od.is_ntt, od.scale, od.correction_factor)

    def plain_to_ntt(self, plain, limbs):
        plain = np.ascontiguousarray(plain, dtype=np.uint64)
        out = np.zeros((limbs, self.N), dtype=np.uint64)
        self._chk(lib().orc_plain_to_ntt(self.h, _p(plain), len(plain), limbs, _p(out)))
        return out

    def decrypt(self, ct, sk):
        sk = np.ascontiguousarray(sk, dtype=np.uint64)
        n = ct.limbs * self.N if self.scheme == CKKS else self.N
        out = np.zeros(n, dtype=np.uint64)
        d = ct.desc()
        self._chk(lib().orc_decrypt(self.h, _p(sk), C.byref(d), _p(ct.data), _p(out)))
        return out

    def time_mul_relin(self, a, b, count, threads=1):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        return float(lib().orc_time_mul_relin(self.h, _p(a), _p(b), count, threads))

    def time_ntt(self, prime_idx, limb, count):
        limb = np.ascontiguousarray(limb, dtype=np.uint64)
        return float(lib().orc_time_ntt(self.h, prime_idx, _p(limb), count))
