"""Host-side mirror of the reference's troyn:: interface over the C ABI (include/troyhip.h).

Names and argument meaning follow src/troy_cuda.cuh / src/evaluator_cuda.cuh (KernelProvider,
SEALContext, Ciphertext, RelinKeys/GaloisKeys, Evaluator::multiply / relinearizeInplace /
rotateRowsInplace / rescaleToNextInplace ...), with one extension: a `Ciphertext` here is a *batch* of B
independent ciphertexts of identical shape (B = 1 is the reference's object).  The C++ form of the same
mirror is include/troyn.hpp (+ include/troyn_app.hpp for the app helpers).  Everything runs through libtroyhip.so; there is no CPU path.
"""
import ctypes as C
import os
import struct

import numpy as np

from . import capi
from .capi import BFV, BGV, CKKS, CtStruct  # noqa: F401


def _u64p(a):
    return a.ctypes.data_as(C.c_void_p)


class KernelProvider:
    """src/kernelprovider.cuh:24-85"""
    _lib = None

    @classmethod
    def initialize(cls, device=0, _lib=None):
        cls._lib = _lib or capi.load()
        capi.check(cls._lib, cls._lib.troyhip_initialize(int(device)))

    @classmethod
    def lib(cls):
        if cls._lib is None:
            cls._lib = capi.load()  # calls below then fail with "KernelProvider not initialized."
        return cls._lib


class DeviceBuffer:
    """DeviceArray<uint64_t> (src/utils/devicearray.cuh): owning device allocation, deep copy on copy()."""

    def __init__(self, words):
        self.lib = KernelProvider.lib()
        self.words = int(words)
        p = C.c_void_p()
        capi.check(self.lib, self.lib.troyhip_malloc(C.byref(p), C.c_size_t(self.words * 8)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        b = cls(arr.size)
        capi.check(b.lib, b.lib.troyhip_copy_h2d(C.c_void_p(b.ptr), _u64p(arr), C.c_size_t(arr.size * 8), None))
        return b

    def to_numpy(self, words=None, offset=0):
        n = self.words - offset if words is None else int(words)
        out = np.empty(n, dtype=np.uint64)
        capi.check(self.lib, self.lib.troyhip_copy_d2h(_u64p(out), C.c_void_p(self.ptr + 8 * offset), C.c_size_t(n * 8), None))
        return out

    def copy(self):
        b = DeviceBuffer(self.words)
        capi.check(self.lib, self.lib.troyhip_copy_d2d(C.c_void_p(b.ptr), C.c_void_p(self.ptr), C.c_size_t(self.words * 8), None))
        return b

    def copy_from(self, src, words, src_offset_words=0, dst_offset_words=0):
        """device-to-device copy of `words` u64 out of another buffer"""
        capi.check(self.lib, self.lib.troyhip_copy_d2d(C.c_void_p(self.ptr + 8 * dst_offset_words), C.c_void_p(src.ptr + 8 * src_offset_words), C.c_size_t(int(words) * 8), None))

    def zero(self):
        capi.check(self.lib, self.lib.troyhip_memset_zero(C.c_void_p(self.ptr), C.c_size_t(self.words * 8), None))

    def __del__(self):
        if getattr(self, "ptr", None):
            try:
                self.lib.troyhip_free(C.c_void_p(self.ptr))
            except Exception:
                pass
            self.ptr = None


def synchronize(stream=None):
    lib = KernelProvider.lib()
    capi.check(lib, lib.troyhip_stream_synchronize(stream))


class CoeffModulus:
    @staticmethod
    def Create(poly_modulus_degree, bit_sizes):  # src/modulus.h:485
        lib = KernelProvider.lib()
        out = np.zeros(len(bit_sizes), dtype=np.uint64)
        bits = (C.c_int * len(bit_sizes))(*bit_sizes)
        capi.check(lib, lib.troyhip_coeff_modulus_create(C.c_uint64(poly_modulus_degree), bits, len(bit_sizes), _u64p(out)))
        return [int(x) for x in out]


class PlainModulus:
    @staticmethod
    def Batching(poly_modulus_degree, bit_size):  # src/modulus.h:528
        lib = KernelProvider.lib()
        out = C.c_uint64()
        capi.check(lib, lib.troyhip_plain_modulus_batching(C.c_uint64(poly_modulus_degree), bit_size, C.byref(out)))
        return out.value


class SEALContext:
    """SEALContextCuda (src/context_cuda.cuh:146-186); SecurityLevel::none semantics."""

    def __init__(self, scheme, poly_modulus_degree, coeff_modulus, plain_modulus=0, host_only=False):
        """host_only=True builds the tables on the host only (no GPU needed): enough for KeyGenerator/Encryptor/Decryptor"""
        self.lib = KernelProvider.lib()
        self.host_only = bool(host_only)
        self.scheme, self.N = scheme, int(poly_modulus_degree)
        self.coeff_modulus = [int(p) for p in coeff_modulus]
        self.plain_modulus = int(plain_modulus)
        arr = np.array(self.coeff_modulus, dtype=np.uint64)
        h = C.c_void_p()
        create = self.lib.troyhip_context_create_host if self.host_only else self.lib.troyhip_context_create
        capi.check(self.lib, create(scheme, C.c_uint64(self.N), _u64p(arr), len(arr), C.c_uint64(self.plain_modulus), C.byref(h)))
        self.h = h
        info = capi.ContextInfo()
        capi.check(self.lib, self.lib.troyhip_context_info(self.h, C.byref(info)))
        self.key_limbs, self.first_limbs, self.last_limbs = info.key_limbs, info.first_limbs, info.last_limbs

    def __del__(self):
        if getattr(self, "h", None):
            try:
                self.lib.troyhip_context_destroy(self.h)
            except Exception:
                pass
            self.h = None

    def behz_bases(self, limbs):
        out = np.zeros(limbs + 3, dtype=np.uint64)
        n, g = C.c_int(), C.c_uint64()
        capi.check(self.lib, self.lib.troyhip_context_behz_bases(self.h, limbs, _u64p(out), C.byref(n), C.byref(g)))
        return [int(x) for x in out[:n.value]], g.value

    def ntt_tables(self, prime):
        N = self.N
        rop, rquo, iop, iquo = (np.zeros(N, dtype=np.uint64) for _ in range(4))
        invd = np.zeros(2, dtype=np.uint64)
        root = C.c_uint64()
        capi.check(self.lib, self.lib.troyhip_context_ntt_tables(self.h, C.c_uint64(prime), _u64p(rop), _u64p(rquo), _u64p(iop), _u64p(iquo), _u64p(invd), C.byref(root)))
        return dict(root=root.value, root_op=rop, root_quo=rquo, inv_op=iop, inv_quo=iquo, inv_degree=invd)

    def galois_elt_from_step(self, step):
        out = C.c_uint32()
        capi.check(self.lib, self.lib.troyhip_galois_elt_from_step(self.h, step, C.byref(out)))
        return out.value

    def reserve_scratch(self, words):
        capi.check(self.lib, self.lib.troyhip_context_reserve_scratch(self.h, C.c_size_t(int(words))))

    def scratch_words(self, op, limbs, batch):
        out = C.c_size_t()
        capi.check(self.lib, self.lib.troyhip_context_scratch_words(self.h, op, limbs, C.c_uint64(batch), C.byref(out)))
        return out.value

    # kernel_util level entry points
    def ntt(self, buf, rows, row_primes, inverse=False, inner=1, offset_words=0, stream=None):
        pr = np.array([int(p) for p in row_primes], dtype=np.uint64)
        capi.check(self.lib, self.lib.troyhip_ntt(self.h, C.c_void_p(buf.ptr + 8 * offset_words), C.c_uint64(rows), _u64p(pr), len(pr), inner, int(inverse), stream))

    def fill_uniform(self, buf, rows, row_primes, seed, row0=0, inner=1, offset_words=0, stream=None):
        pr = np.array([int(p) for p in row_primes], dtype=np.uint64)
        capi.check(self.lib, self.lib.troyhip_fill_uniform(self.h, C.c_void_p(buf.ptr + 8 * offset_words), C.c_uint64(rows), _u64p(pr), len(pr), inner, C.c_uint64(seed), C.c_uint64(row0), stream))


class Ciphertext:
    """CiphertextCuda (src/ciphertext_cuda.cuh:12-268) x batch.  Device data [batch][capacity][limbs][N]."""

    def __init__(self, context, batch, size, limbs, is_ntt_form=False, scale=1.0, correction_factor=1, capacity=None, buf=None):
        self.context, self.batch, self._size, self.limbs = context, int(batch), int(size), int(limbs)
        self.is_ntt_form, self.scale, self.correction_factor = bool(is_ntt_form), float(scale), int(correction_factor)
        self.capacity = int(capacity or size)
        self.buf = buf if buf is not None else DeviceBuffer(self.batch * self.capacity * self.limbs * context.N)

    @classmethod
    def from_numpy(cls, context, data, is_ntt_form=False, scale=1.0, correction_factor=1, capacity=None):
        """data: uint64 [batch][size][limbs][N] (or [size][limbs][N] for a single ciphertext)."""
        data = np.asarray(data, dtype=np.uint64)
        if data.ndim == 3:
            data = data[None]
        B, size, limbs, N = data.shape
        assert N == context.N
        cap = int(capacity or size)
        if cap != size:
            full = np.zeros((B, cap, limbs, N), dtype=np.uint64)
            full[:, :size] = data
            data = full
        return cls(context, B, size, limbs, is_ntt_form, scale, correction_factor, cap, DeviceBuffer.from_numpy(data))

    # ---- wire format of CiphertextCuda::save / load (src/ciphertext_cuda.cu:16-43, 82-104): a raw little-endian field dump
    #   parms_id (4 x u64, BLAKE2b-256 of the level's parameters) | is_ntt_form (1 byte) | size, poly_modulus_degree,
    #   coeff_modulus_size (u64 each) | scale (f64) | correction_factor (u64) | seed (u64, 0) | terms (1 byte, 0) |
    #   data word count (u64) | data [size][limbs][N] u64
    def save(self, stream, index=0):
        """writes batch item `index` in the reference's format"""
        import struct
        ctx = self.context
        pid = np.zeros(4, dtype=np.uint64)
        capi.check(ctx.lib, ctx.lib.troyhip_context_parms_id(ctx.h, int(self.limbs), _u64p(pid)))
        data = np.ascontiguousarray(self.cpu()[index])
        stream.write(pid.tobytes())
        stream.write(struct.pack("<?QQQdQQ?Q", bool(self.is_ntt_form), self._size, ctx.N, self.limbs, float(self.scale), int(self.correction_factor), 0, False, data.size))
        stream.write(data.tobytes())

    @classmethod
    def load(cls, context, stream):
        """reads one ciphertext; the parms_id must name a level of `context`"""
        import struct
        pid = np.frombuffer(stream.read(32), dtype=np.uint64)
        ntt, size, n, limbs, scale, cf, seed, terms, words = struct.unpack("<?QQQdQQ?Q", stream.read(struct.calcsize("<?QQQdQQ?Q")))
        if terms:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "Trying to load a termed ciphertext, but indices is not specified")
        if seed:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "seed is not zero.")
        mine = np.zeros(4, dtype=np.uint64)
        if n != context.N or limbs < 1 or limbs > context.key_limbs or words != size * limbs * n:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "encrypted is not valid for encryption parameters")
        capi.check(context.lib, context.lib.troyhip_context_parms_id(context.h, int(limbs), _u64p(mine)))
        if not np.array_equal(mine, pid):
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "encrypted is not valid for encryption parameters")
        data = np.frombuffer(stream.read(8 * words), dtype=np.uint64).reshape(1, size, limbs, n)
        return cls.from_numpy(context, data, ntt, scale, cf)

    # CiphertextCuda::saveTerms / loadTerms (src/ciphertext_cuda.cu:44-80, 106-143): the sender of a matmul/conv result keeps
    # only the coefficients of c0 the receiver will read.  Same header with terms = 1; then, in coefficient form, c0 as
    # [term][limb] words for the listed terms, the word count of the remaining polynomials, and c1.. in full.
    def coeff_host(self, evaluator):
        """host copy [batch][size][limbs][N] in coefficient form (what saveTerms writes from)"""
        return (evaluator.transformFromNtt(self) if self.is_ntt_form else self).cpu()

    def saveTerms(self, stream, evaluator, term_ids, index=0, coeff_host=None):
        import struct
        ctx = self.context
        data = (self.coeff_host(evaluator) if coeff_host is None else coeff_host)[index]
        pid = np.zeros(4, dtype=np.uint64)
        capi.check(ctx.lib, ctx.lib.troyhip_context_parms_id(ctx.h, int(self.limbs), _u64p(pid)))
        stream.write(pid.tobytes())
        stream.write(struct.pack("<?QQQdQQ?", bool(self.is_ntt_form), self._size, ctx.N, self.limbs, float(self.scale), int(self.correction_factor), 0, True))
        ids = np.asarray(list(term_ids), dtype=np.int64)
        if ids.size and (ids.min() < 0 or ids.max() >= ctx.N):
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "term index out of range")
        stream.write(np.ascontiguousarray(data[0][:, ids].T).tobytes())  # [term][limb]
        rest = np.ascontiguousarray(data[1:])
        stream.write(struct.pack("<Q", rest.size))
        stream.write(rest.tobytes())

    @classmethod
    def loadTerms(cls, context, stream, evaluator, term_ids):
        """the unlisted coefficients of c0 are zero (the reference leaves them unspecified)"""
        data, ntt, scale, cf = cls.load_terms_host(context, stream, term_ids)
        ct = cls.from_numpy(context, data, False, scale, cf)
        if ntt:
            evaluator.transformToNttInplace(ct)
        return ct

    @staticmethod
    def load_terms_host(context, stream, term_ids):
        """-> (coefficient-form data [1][size][limbs][N], is_ntt_form, scale, correction_factor)"""
        import struct
        pid = np.frombuffer(stream.read(32), dtype=np.uint64)
        ntt, size, n, limbs, scale, cf, seed, terms = struct.unpack("<?QQQdQQ?", stream.read(struct.calcsize("<?QQQdQQ?")))
        if not terms:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "Trying to load a normal ciphertext, but term indices is specified")
        if seed:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "seed is not zero.")
        if n != context.N or limbs < 1 or limbs > context.key_limbs or size < 1:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "encrypted is not valid for encryption parameters")
        mine = np.zeros(4, dtype=np.uint64)
        capi.check(context.lib, context.lib.troyhip_context_parms_id(context.h, int(limbs), _u64p(mine)))
        if not np.array_equal(mine, pid):
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "encrypted is not valid for encryption parameters")
        ids = np.asarray(list(term_ids), dtype=np.int64)
        if ids.size and (ids.min() < 0 or ids.max() >= n):
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "term index out of range")
        data = np.zeros((1, size, limbs, n), dtype=np.uint64)
        c0 = np.frombuffer(stream.read(8 * ids.size * limbs), dtype=np.uint64).reshape(ids.size, limbs)
        data[0, 0][:, ids] = c0.T
        (words,) = struct.unpack("<Q", stream.read(8))
        if words != (size - 1) * limbs * n:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "encrypted is not valid for encryption parameters")
        data[0, 1:] = np.frombuffer(stream.read(8 * words), dtype=np.uint64).reshape(size - 1, limbs, n)
        return data, ntt, scale, cf

    def cpu_poly_view(self):
        """host copy [batch][size][limbs][N] (alias of cpu(), named for the LWE helpers)"""
        return self.cpu()

    def cpu(self):  # CiphertextCuda::cpu / toHost
        n = self.batch * self.capacity * self.limbs * self.context.N
        a = self.buf.to_numpy(n).reshape(self.batch, self.capacity, self.limbs, self.context.N)
        return a[:, :self._size].copy()

    def size(self):
        return self._size

    def coeffModulusSize(self):
        return self.limbs

    def polyModulusDegree(self):
        return self.context.N

    def isNttForm(self):
        return self.is_ntt_form

    @property
    def bstride(self):
        return self.capacity * self.limbs * self.context.N

    def struct(self):
        return CtStruct(self.buf.ptr, self.bstride, self._size, self.limbs, int(self.is_ntt_form), self.scale, self.correction_factor)

    def _absorb(self, st):
        self._size, self.limbs, self.is_ntt_form = st.size, st.limbs, bool(st.is_ntt_form)
        self.scale, self.correction_factor = st.scale, st.correction_factor

    def copy(self):  # deep device copy (src/utils/devicearray.cuh:153-164)
        return Ciphertext(self.context, self.batch, self._size, self.limbs, self.is_ntt_form, self.scale, self.correction_factor, self.capacity, self.buf.copy())


class LWECiphertext:
    """LWECiphertextCuda (src/ciphertext_cuda.cuh): c1 = one polynomial per item (a size-1 batched Ciphertext, coefficient
    form), c0 = one residue per limb per item (numpy [batch][limbs])."""

    def __init__(self, c1, c0):
        self.c1, self.c0 = c1, np.ascontiguousarray(c0, dtype=np.uint64)


class KSwitchKeys:
    """KSwitchKeysCuda (src/kswitchkeys_cuda.cuh:43-56): data()[index] = one key-switching key, uploaded from
    the host array [K-1][2][K][N] (NTT form)."""

    def __init__(self, context):
        self.context, self.keys = context, {}

    def set(self, index, host_array):
        a = np.ascontiguousarray(host_array, dtype=np.uint64)
        K, N = self.context.key_limbs, self.context.N
        if a.shape != (K - 1, 2, K, N):
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "kswitch_keys is not valid for encryption parameters")
        self.keys[index] = DeviceBuffer.from_numpy(a)

    def hasKey(self, index):
        return index in self.keys


class RelinKeys(KSwitchKeys):  # src/relinkeys_cuda.cuh:56-59
    @staticmethod
    def getIndex(key_power):
        if key_power < 2:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "key_power cannot be less than 2")
        return key_power - 2


class GaloisKeys(KSwitchKeys):  # src/galoiskeys_cuda.cuh:74-77
    @staticmethod
    def getIndex(galois_elt):  # src/utils/galois_cuda.cuh:45-48
        if not galois_elt & 1:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "galois_elt is not valid")
        return (galois_elt - 1) >> 1

    def set_elt(self, galois_elt, host_array):
        self.set(self.getIndex(galois_elt), host_array)


class Evaluator:
    """EvaluatorCuda (src/evaluator_cuda.cuh:13-361) over batches."""

    def __init__(self, context, stream=None):
        self.context, self.lib, self.stream = context, context.lib, stream

    def _chk(self, rc):
        capi.check(self.lib, rc)

    # -- negate / add / sub
    def negateInplace(self, a):
        st = a.struct()
        self._chk(self.lib.troyhip_negate(self.context.h, C.byref(st), C.c_uint64(a.batch), self.stream))
        a._absorb(st)

    def addInplace(self, a, b):
        if b.size() > a.capacity:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "destination capacity too small")
        sa, sb = a.struct(), b.struct()
        self._chk(self.lib.troyhip_add(self.context.h, C.byref(sa), C.byref(sb), C.c_uint64(a.batch), self.stream))
        a._absorb(sa)

    def subInplace(self, a, b):
        if b.size() > a.capacity:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "destination capacity too small")
        sa, sb = a.struct(), b.struct()
        self._chk(self.lib.troyhip_sub(self.context.h, C.byref(sa), C.byref(sb), C.c_uint64(a.batch), self.stream))
        a._absorb(sa)

    def add(self, a, b):
        r = self._grown_copy(a, max(a.size(), b.size()))
        self.addInplace(r, b)
        return r

    def sub(self, a, b):
        r = self._grown_copy(a, max(a.size(), b.size()))
        self.subInplace(r, b)
        return r

    def negate(self, a):
        r = a.copy()
        self.negateInplace(r)
        return r

    def _grown_copy(self, a, cap):
        if cap <= a.capacity:
            return a.copy()
        return Ciphertext.from_numpy(a.context, a.cpu(), a.is_ntt_form, a.scale, a.correction_factor, capacity=cap)

    # -- multiply / square
    def multiply(self, a, b, destination=None):
        ds = a.size() + b.size() - 1
        out = destination or Ciphertext(a.context, a.batch, ds, a.limbs, capacity=ds)
        if out.capacity < ds:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "destination capacity too small")
        sa, sb, so = a.struct(), b.struct(), out.struct()
        self._chk(self.lib.troyhip_multiply(self.context.h, C.byref(sa), C.byref(sb), C.byref(so), C.c_uint64(a.batch), self.stream))
        out._absorb(so)
        return out

    def multiplyInplace(self, a, b):
        ds = a.size() + b.size() - 1
        if a.capacity >= ds:
            return self.multiply(a, b, a)
        out = self.multiply(a, b)
        a.__dict__.update(out.__dict__)
        return a

    def square(self, a):
        return self.multiply(a, a)

    def squareInplace(self, a):
        return self.multiplyInplace(a, a)

    # -- key switching
    def relinearizeInplace(self, a, relin_keys):
        """relinearizeInternal to size 2 from any size <= 16 (src/evaluator_cuda.cu:703-744): needs the keys of index 0 .. size-3"""
        need = max(a.size() - 2, 0)
        for idx in range(need):
            if not relin_keys.hasKey(idx):
                raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "not enough relinearization keys")
        st = a.struct()
        ptrs = (C.c_void_p * max(need, 1))(*[relin_keys.keys[i].ptr for i in range(need)])
        self._chk(self.lib.troyhip_relinearize_keys(self.context.h, C.byref(st), ptrs, need, C.c_uint64(a.batch), self.stream))
        a._absorb(st)

    def relinearize(self, a, relin_keys, destination=None):
        """relinearize(encrypted, relin_keys, destination): the reference copies and relinearizes in place; from size 3 the library reads the
        operand where it lies and writes the size-2 result to the destination (troyhip_relinearize_to): no copy of the operand"""
        need = max(a.size() - 2, 0)
        for idx in range(need):
            if not relin_keys.hasKey(idx):
                raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "not enough relinearization keys")
        cap = 2 if a.size() == 3 else a.size()
        out = destination or Ciphertext(a.context, a.batch, cap, a.limbs, capacity=cap)
        if out is a or out.capacity < cap:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "destination must be a distinct ciphertext of sufficient capacity")
        si, so = a.struct(), out.struct()
        ptrs = (C.c_void_p * max(need, 1))(*[relin_keys.keys[i].ptr for i in range(need)])
        self._chk(self.lib.troyhip_relinearize_to(self.context.h, C.byref(si), C.byref(so), ptrs, need, C.c_uint64(a.batch), self.stream))
        out._absorb(so)
        return out

    def applyKeySwitchingInplace(self, a, kswitch_keys):
        """applyKeySwitchingInplace (evaluator_cuda.cu:1365-1378): kswitch_keys must hold exactly one key; c1 of a size-2
        ciphertext is switched to it."""
        if len(kswitch_keys.keys) != 1:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "kswitch_keys.data().size() != 1")
        key = next(iter(kswitch_keys.keys.values()))
        st = a.struct()
        self._chk(self.lib.troyhip_apply_key_switching(self.context.h, C.byref(st), C.c_void_p(key.ptr), C.c_uint64(a.batch), self.stream))
        a._absorb(st)

    def applyKeySwitching(self, a, kswitch_keys):
        r = a.copy()
        self.applyKeySwitchingInplace(r, kswitch_keys)
        return r

    def negacyclicShiftInplace(self, a, shift):  # evaluator_cuda.cu:2342-2351
        st = a.struct()
        self._chk(self.lib.troyhip_negacyclic_shift(self.context.h, C.byref(st), C.c_uint64(shift), C.c_uint64(a.batch), self.stream))
        a._absorb(st)

    def negacyclicShift(self, a, shift):
        r = a.copy()
        self.negacyclicShiftInplace(r, shift)
        return r

    # -- compositions, in the reference's order of operations (src/evaluator.cpp)
    def addMany(self, cts):  # evaluator.cpp addMany: left fold
        if not cts:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "encrypteds cannot be empty")
        r = cts[0].copy()
        for c in cts[1:]:
            r = self.add(r, c) if c.size() > r.capacity else (self.addInplace(r, c) or r)
        return r

    def multiplyMany(self, cts, relin_keys):
        """evaluator.cpp:1502-1572: pairwise products are appended to the work list until one ciphertext is left; every
        product is relinearized (BFV / BGV only)."""
        if not cts:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "encrypteds vector must not be empty")
        if self.context.scheme not in (BFV, BGV):
            raise capi.LogicError(capi.LOGIC_ERROR, "unsupported scheme")
        if len(cts) == 1:
            return cts[0].copy()
        work = []
        for i in range(0, len(cts) - 1, 2):
            t = self.multiply(cts[i], cts[i + 1])
            self.relinearizeInplace(t, relin_keys)
            work.append(t)
        if len(cts) & 1:
            work.append(cts[-1])
        i = 0
        while i < len(work) - 1:
            t = self.multiply(work[i], work[i + 1])
            self.relinearizeInplace(t, relin_keys)
            work.append(t)
            i += 2
        return work[-1]

    def exponentiate(self, a, exponent, relin_keys):  # evaluator.cpp:1574-1601
        if exponent == 0:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "exponent cannot be 0")
        if exponent == 1:
            return a.copy()
        return self.multiplyMany([a] * int(exponent), relin_keys)

    def exponentiateInplace(self, a, exponent, relin_keys):
        a.__dict__.update(self.exponentiate(a, exponent, relin_keys).__dict__)

    def modSwitchTo(self, a, limbs):
        """modSwitchTo(encrypted, parms_id): parms_id is named by its limb count here."""
        if limbs > a.limbs:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "cannot switch to higher level modulus")
        r = a
        while r.limbs > limbs:
            r = self.modSwitchToNext(r)
        return r.copy() if r is a else r

    def modSwitchToInplace(self, a, limbs):
        a.__dict__.update(self.modSwitchTo(a, limbs).__dict__)

    def rescaleTo(self, a, limbs):
        if limbs > a.limbs:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "cannot switch to higher level modulus")
        r = a
        while r.limbs > limbs:
            r = self.rescaleToNext(r)
        return r.copy() if r is a else r

    def rescaleToInplace(self, a, limbs):
        a.__dict__.update(self.rescaleTo(a, limbs).__dict__)

    # ---- LWE extraction / packing (evaluator_cuda.cu:2178-2340; CUDA-only API of the reference).  Host-level compositions of
    # negacyclicShift, add / sub, applyGalois and a per-limb scalar multiply; the batch dimension is carried through.
    def divideByPolyModulusDegreeInplace(self, a, mul=1):
        st = a.struct()
        self._chk(self.lib.troyhip_divide_by_poly_modulus_degree(self.context.h, C.byref(st), C.c_uint64(mul), C.c_uint64(a.batch), self.stream))
        a._absorb(st)

    def extractLWE(self, a, term):
        """LWE sample of coefficient `term`: c1 = x^(2N - term) * c1(x) (so that its constant-term inner product with the key is
        coefficient `term` of c1 * s), c0 = coefficient `term` of c0(x)."""
        if a.size() != 2:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "Encrypted size must be 2 to be extracted.")
        if a.is_ntt_form:
            a = self.transformFromNtt(a)
        N, L, B = self.context.N, a.limbs, a.batch
        x = a.cpu_poly_view()
        c1 = Ciphertext.from_numpy(a.context, x[:, 1:2], False, a.scale, a.correction_factor)
        self.negacyclicShiftInplace(c1, 0 if term == 0 else 2 * N - term)
        c0 = np.ascontiguousarray(x[:, 0, :, term])
        return LWECiphertext(c1, c0)

    def assembleLWE(self, lwe, term):
        """RLWE ciphertext whose coefficient `term` decrypts to the LWE message (the other coefficients are noise-like)"""
        c1 = lwe.c1.copy()
        self.negacyclicShiftInplace(c1, term)
        B, L, N = c1.batch, c1.limbs, self.context.N
        data = np.zeros((B, 2, L, N), dtype=np.uint64)
        data[:, 1] = c1.cpu()[:, 0]
        data[:, 0, :, term] = lwe.c0
        return Ciphertext.from_numpy(c1.context, data, False, c1.scale, c1.correction_factor, capacity=3)

    def fieldTraceInplace(self, a, galois_keys, logn):
        degree = self.context.N
        while degree > (1 << logn):
            t = self.applyGalois(a, degree + 1, galois_keys)
            self.addInplace(a, t)
            degree >>= 1

    def packLWECiphertexts(self, lwes, galois_keys):
        """evaluator_cuda.cu:2275-2340: n LWE samples -> one RLWE ciphertext whose coefficients 0, N/n', 2N/n', .. carry them
        (n' = n rounded up to a power of two); needs the Galois keys of the elements 2^k + 1."""
        if not lwes:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "LWE ciphertexts must not be empty.")
        N = self.context.N
        ckks = self.context.scheme == CKKS
        l = 0
        while (1 << l) < len(lwes):
            l += 1
        zero = self.assembleLWE(lwes[0], 0)
        zero.buf.zero()
        rl = []
        for i in range(1 << l):
            idx = int(format(i, "0%db" % l)[::-1], 2) if l else 0
            if idx < len(lwes):
                c = self.assembleLWE(lwes[idx], 0)
                self.divideByPolyModulusDegreeInplace(c)
                rl.append(c)
            else:
                rl.append(zero.copy())
        for layer in range(l):
            gap, shift = 1 << layer, N >> (layer + 1)
            for off in range(0, 1 << l, 2 * gap):
                even, odd = rl[off], rl[off + gap]
                temp = self.negacyclicShift(odd, shift)
                new_odd = self.sub(even, temp)
                self.addInplace(even, temp)
                if ckks:
                    self.transformToNttInplace(new_odd)
                self.applyGaloisInplace(new_odd, (1 << (layer + 1)) + 1, galois_keys)
                if ckks:
                    self.transformFromNttInplace(new_odd)
                self.addInplace(even, new_odd)
                rl[off + gap] = new_odd
        ret = rl[0]
        self.fieldTraceInplace(ret, galois_keys, l)
        if ckks:
            self.transformToNttInplace(ret)
        return ret

    def _copy_then(self, fn, a, *args):
        r = a.copy()
        fn(r, *args)
        return r

    def applyGalois(self, a, galois_elt, galois_keys): return self._copy_then(self.applyGaloisInplace, a, galois_elt, galois_keys)
    def rotateRows(self, a, steps, galois_keys): return self._copy_then(self.rotateRowsInplace, a, steps, galois_keys)
    def rotateColumns(self, a, galois_keys): return self._copy_then(self.rotateColumnsInplace, a, galois_keys)
    def rotateVector(self, a, steps, galois_keys): return self._copy_then(self.rotateVectorInplace, a, steps, galois_keys)
    def complexConjugate(self, a, galois_keys): return self._copy_then(self.complexConjugateInplace, a, galois_keys)
    def transformToNtt(self, a): return self._copy_then(self.transformToNttInplace, a)
    def transformFromNtt(self, a): return self._copy_then(self.transformFromNttInplace, a)

    def applyGaloisInplace(self, a, galois_elt, galois_keys):
        idx = GaloisKeys.getIndex(galois_elt)
        if not galois_keys.hasKey(idx):
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "Galois key not present")
        st = a.struct()
        self._chk(self.lib.troyhip_apply_galois(self.context.h, C.byref(st), C.c_uint32(galois_elt), C.c_void_p(galois_keys.keys[idx].ptr), C.c_uint64(a.batch), self.stream))
        a._absorb(st)

    def _rotate(self, a, steps, conjugate, galois_keys):
        elts = [2 * i + 1 for i in galois_keys.keys]
        n = len(elts)
        e = (C.c_uint32 * max(n, 1))(*elts)
        k = (C.c_void_p * max(n, 1))(*[galois_keys.keys[(x - 1) >> 1].ptr for x in elts])
        st = a.struct()
        self._chk(self.lib.troyhip_rotate(self.context.h, C.byref(st), int(steps), int(conjugate), e, k, n, C.c_uint64(a.batch), self.stream))
        a._absorb(st)

    def rotateRowsInplace(self, a, steps, galois_keys):
        if self.context.scheme not in (BFV, BGV):
            raise capi.LogicError(capi.LOGIC_ERROR, "unsupported scheme")
        self._rotate(a, steps, 0, galois_keys)

    def rotateColumnsInplace(self, a, galois_keys):
        if self.context.scheme not in (BFV, BGV):
            raise capi.LogicError(capi.LOGIC_ERROR, "unsupported scheme")
        self._rotate(a, 0, 1, galois_keys)

    def rotateVectorInplace(self, a, steps, galois_keys):
        if self.context.scheme != CKKS:
            raise capi.LogicError(capi.LOGIC_ERROR, "unsupported scheme")
        self._rotate(a, steps, 0, galois_keys)

    def complexConjugateInplace(self, a, galois_keys):
        if self.context.scheme != CKKS:
            raise capi.LogicError(capi.LOGIC_ERROR, "unsupported scheme")
        self._rotate(a, 0, 1, galois_keys)

    # -- modulus switching
    def _next(self, a, fn):
        out = Ciphertext(a.context, a.batch, a.size(), max(a.limbs - 1, 1), capacity=a.size())
        si, so = a.struct(), out.struct()
        self._chk(fn(self.context.h, C.byref(si), C.byref(so), C.c_uint64(a.batch), self.stream))
        out._absorb(so)
        return out

    def modSwitchToNext(self, a):
        return self._next(a, self.lib.troyhip_mod_switch_to_next)

    def modSwitchToNextInplace(self, a):
        a.__dict__.update(self.modSwitchToNext(a).__dict__)

    def rescaleToNext(self, a):
        return self._next(a, self.lib.troyhip_rescale_to_next)

    def rescaleToNextInplace(self, a):
        a.__dict__.update(self.rescaleToNext(a).__dict__)

    # -- NTT form
    def transformToNttInplace(self, a):
        st = a.struct()
        self._chk(self.lib.troyhip_transform_to_ntt(self.context.h, C.byref(st), C.c_uint64(a.batch), self.stream))
        a._absorb(st)

    def transformFromNttInplace(self, a):
        st = a.struct()
        self._chk(self.lib.troyhip_transform_from_ntt(self.context.h, C.byref(st), C.c_uint64(a.batch), self.stream))
        a._absorb(st)

    def multiplyPlainInplace(self, a, plain_ntt, plain_scale=1.0):
        """NTT-form operands only (multiplyPlainNtt, evaluator_cuda.cu:1824-1863); plain_ntt: DeviceBuffer [limbs][N]."""
        st = a.struct()
        self._chk(self.lib.troyhip_multiply_plain_ntt(self.context.h, C.byref(st), C.c_void_p(plain_ntt.ptr), C.c_double(plain_scale), C.c_uint64(a.batch), self.stream))
        a._absorb(st)

    def multiplyPlainAccumulate(self, cts, plains, plain_scale=1.0):
        """sum_i cts[i] (x) plains[i] as ONE pass (troyhip_multiply_plain_accumulate): what the multiplyPlain + addInplace loop of
        MatmulHelper::matmul / Conv2dHelper::conv2d computes per output block (app/LinearHelperCKKS.cuh:227-248, 536-556), same residues.
        cts: batched NTT-form ciphertexts of one shape and scale; plains: DeviceBuffer [limbs][N] each; 1..16 products."""
        if not 1 <= len(cts) <= 16 or len(cts) != len(plains):
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "multiplyPlainAccumulate takes 1 to 16 (ciphertext, plaintext) pairs")
        a0 = cts[0]
        out = Ciphertext(self.context, a0.batch, a0.size(), a0.limbs, True, a0.scale, a0.correction_factor, capacity=a0.size())
        structs = [c.struct() for c in cts]
        ct_ptrs = (C.POINTER(CtStruct) * len(cts))(*[C.pointer(st) for st in structs])
        pl_ptrs = (C.c_void_p * len(plains))(*[p.ptr for p in plains])
        st = out.struct()
        self._chk(self.lib.troyhip_multiply_plain_accumulate(self.context.h, ct_ptrs, pl_ptrs, len(cts), C.c_double(plain_scale), C.byref(st), C.c_uint64(a0.batch), self.stream))
        out._absorb(st)
        return out

    # ---- plaintext operands in coefficient form (evaluator_cuda.cu:1654-1948).  plain: DeviceBuffer holding either ONE
    # plaintext (n_coeffs coefficients mod t; CKKS: [limbs][N] NTT rows) or one per batch item (per_item=True, back to back)
    def _plain_stride(self, a, n_coeffs, per_item):
        if not per_item:
            return 0
        return a.limbs * self.context.N if self.context.scheme == capi.CKKS else n_coeffs

    def addPlainInplace(self, a, plain, n_coeffs=None, plain_scale=1.0, per_item=False, _sub=0):
        n = self.context.N if n_coeffs is None else int(n_coeffs)
        st = a.struct()
        self._chk(self.lib.troyhip_add_plain(self.context.h, C.byref(st), C.c_void_p(plain.ptr), C.c_uint64(n), C.c_uint64(self._plain_stride(a, n, per_item)),
                                             C.c_double(plain_scale), _sub, C.c_uint64(a.batch), self.stream))
        a._absorb(st)

    def subPlainInplace(self, a, plain, n_coeffs=None, plain_scale=1.0, per_item=False):
        self.addPlainInplace(a, plain, n_coeffs, plain_scale, per_item, _sub=1)

    def multiplyPlainNormalInplace(self, a, plain, n_coeffs=None, per_item=False):
        """multiplyPlainInplace with coefficient-form operands (multiplyPlainNormal)."""
        n = self.context.N if n_coeffs is None else int(n_coeffs)
        st = a.struct()
        self._chk(self.lib.troyhip_multiply_plain(self.context.h, C.byref(st), C.c_void_p(plain.ptr), C.c_uint64(n), C.c_uint64(self._plain_stride(a, n, per_item)),
                                                  C.c_uint64(a.batch), self.stream))
        a._absorb(st)

    def decrypt(self, a, secret_key):
        """DecryptorCuda::decrypt on the device: secret_key = DeviceBuffer [K][N] (NTT form).  Returns a numpy array
        [batch][N] (BFV/BGV coefficients mod t) or [batch][limbs][N] (CKKS RNS plaintext, NTT form)."""
        N = self.context.N
        per = a.limbs * N if self.context.scheme == capi.CKKS else N
        out = DeviceBuffer(a.batch * per)
        st = a.struct()
        self._chk(self.lib.troyhip_decrypt(self.context.h, C.byref(st), C.c_void_p(secret_key.ptr), C.c_void_p(out.ptr), C.c_uint64(per), C.c_uint64(a.batch), self.stream))
        r = out.to_numpy()
        return r.reshape(a.batch, a.limbs, N) if self.context.scheme == capi.CKKS else r.reshape(a.batch, N)

    def transformPlainToNtt(self, plain, limbs, n_coeffs=None, count=1):
        """transformToNttInplace(Plaintext, parms_id): returns a DeviceBuffer [count][limbs][N]."""
        n = self.context.N if n_coeffs is None else int(n_coeffs)
        out = DeviceBuffer(count * limbs * self.context.N)
        self._chk(self.lib.troyhip_plain_to_ntt(self.context.h, C.c_void_p(plain.ptr), C.c_uint64(n), C.c_uint64(n if count > 1 else 0), int(limbs),
                                                C.c_void_p(out.ptr), C.c_uint64(count), self.stream))
        return out


# ---------------------------------------------------------------- CPU-side keys / encryption / decryption (host buffers)
class KeyGenerator:
    """KeyGeneratorCuda delegates to the CPU KeyGenerator in the reference (src/keygenerator_cuda.cuh); so does this one
    (troy_amd/csrc/hostcrypto.cpp).  Keys are numpy arrays in the reference's layouts."""

    def __init__(self, context, seed=None):
        """seed=None (the default): 128 bits from os.urandom, as the reference seeds its PRNG from std::random_device
        (src/randomgen.cpp:23,72); an explicit (lo, hi) pair gives deterministic keys for tests ONLY."""
        if seed is None:
            seed = struct.unpack("<QQ", os.urandom(16))
        self.context, self.lib, self.seed = context, context.lib, (int(seed[0]), int(seed[1]))
        K, N = context.key_limbs, context.N
        self._sk = np.zeros((K, N), dtype=np.uint64)
        self._pk = np.zeros((2, K, N), dtype=np.uint64)
        capi.check(self.lib, self.lib.troyhip_host_keygen(context.h, C.c_uint64(self.seed[0]), C.c_uint64(self.seed[1]), _u64p(self._sk), _u64p(self._pk)))

    def secretKey(self):
        return self._sk

    def createPublicKey(self):
        return self._pk

    def _ksk(self):
        K, N = self.context.key_limbs, self.context.N
        return np.zeros((K - 1, 2, K, N), dtype=np.uint64)

    def createRelinKeys(self):
        out = self._ksk()
        capi.check(self.lib, self.lib.troyhip_host_relin_key(self.context.h, C.c_uint64(self.seed[0]), C.c_uint64(self.seed[1]), _u64p(self._sk), _u64p(out)))
        return out

    def createKeySwitchingKeys(self, new_key):
        """KeyGenerator::createKeySwitchingKeys (src/keygenerator.cpp:360-366): the host key array that takes a ciphertext under `new_key`
        (another generator's secretKey()) to one under this generator's secret key; KSwitchKeys.set(0, .) + applyKeySwitchingInplace use it"""
        new_key = np.ascontiguousarray(new_key, dtype=np.uint64)
        if new_key.shape != self._sk.shape:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "new_key is not valid for encryption parameters")
        out = self._ksk()
        capi.check(self.lib, self.lib.troyhip_host_kswitch_key(self.context.h, C.c_uint64(self.seed[0]), C.c_uint64(self.seed[1]), _u64p(self._sk), _u64p(new_key), _u64p(out)))
        return out

    def createAutomorphismKeys(self):
        """KeyGenerator::createAutomorphismKeys (src/keygenerator.cpp:350-358): the keys of fieldTraceInplace / packLWECiphertexts,
        X -> X^(N / 2^k + 1) for k = 0 .. log2(N) - 1; returns {elt: host key array}"""
        elts, n = [], self.context.N
        while n >= 2:
            elts.append(n + 1)
            n >>= 1
        return self.createGaloisKeys(elts)

    def createGaloisKeys(self, galois_elts):
        """returns {elt: host key array}"""
        keys = {}
        for e in galois_elts:
            out = self._ksk()
            capi.check(self.lib, self.lib.troyhip_host_galois_key(self.context.h, C.c_uint64(self.seed[0]), C.c_uint64(self.seed[1]), _u64p(self._sk), C.c_uint32(e), _u64p(out)))
            keys[int(e)] = out
        return keys


class Encryptor:
    """Encryptor::encrypt with a public key (src/encryptor.cpp:88-260), on the CPU."""

    def __init__(self, context, public_key, seed=None):
        """seed=None: every encrypt() draws a fresh 128-bit seed for (u, e0, e1) from os.urandom; an explicit (lo, hi) pair
        gives the deterministic stream (seed, call counter) for tests ONLY."""
        self.context, self.lib = context, context.lib
        self.pk = None if public_key is None else np.ascontiguousarray(public_key, dtype=np.uint64)
        self.sk = None
        self.seed, self.counter = (None if seed is None else (int(seed[0]), int(seed[1]))), 0

    def setSecretKey(self, secret_key):  # src/encryptor_cuda.cuh:98
        self.sk = np.ascontiguousarray(secret_key, dtype=np.uint64)

    def _run(self, fn, key, plain):
        ctx = self.context
        plain = np.ascontiguousarray(plain, dtype=np.uint64)
        if ctx.scheme == CKKS:
            limbs = plain.shape[0]
            n = ctx.N
        else:
            limbs = ctx.first_limbs
            n = plain.size
        out = np.zeros((2, limbs, ctx.N), dtype=np.uint64)
        self.counter += 1
        lo, hi = struct.unpack("<QQ", os.urandom(16)) if self.seed is None else ((self.seed[0] + self.counter) & (2**64 - 1), self.seed[1])
        capi.check(self.lib, fn(ctx.h, C.c_uint64(lo), C.c_uint64(hi), _u64p(key), _u64p(plain), C.c_uint64(n), limbs, _u64p(out)))
        return out

    def encrypt(self, plain, limbs=None):
        """BFV/BGV: plain = coefficients mod t (<= N of them) -> uint64 [2][first_limbs][N];
        CKKS: plain = [limbs][N] NTT-form RNS polynomial -> [2][limbs][N]"""
        if self.pk is None:
            raise RuntimeError("public key is not set")  # encryptor.cpp:157-160 (std::logic_error)
        return self._run(self.lib.troyhip_host_encrypt, self.pk, plain)

    def encryptSymmetric(self, plain):
        """Encryptor::encryptSymmetric (src/encryptor_cuda.cuh:259-290): (-(a s + e) + m, a) at the plaintext's level; same layouts as encrypt"""
        if getattr(self, "sk", None) is None:
            raise RuntimeError("secret key is not set")  # encryptor.cpp:164-167
        return self._run(self.lib.troyhip_host_encrypt_symmetric, self.sk, plain)


    def _zero(self, key, symmetric, limbs):
        ctx = self.context
        limbs = ctx.first_limbs if limbs is None else int(limbs)
        out = np.zeros((2, limbs, ctx.N), dtype=np.uint64)
        self.counter += 1
        lo, hi = struct.unpack("<QQ", os.urandom(16)) if self.seed is None else ((self.seed[0] + self.counter) & (2**64 - 1), self.seed[1])
        capi.check(self.lib, self.lib.troyhip_host_encrypt_zero(ctx.h, C.c_uint64(lo), C.c_uint64(hi), _u64p(key), int(symmetric), limbs, _u64p(out)))
        return out

    def encryptZero(self, limbs=None):
        """Encryptor::encryptZero(parms_id) (src/encryptor_cuda.cuh:170-237): zero under the public key at the level with `limbs` primes (default: the
        first data level) -> uint64 [2][limbs][N], NTT form for CKKS, scale 1"""
        if self.pk is None:
            raise RuntimeError("public key is not set")
        return self._zero(self.pk, False, limbs)

    def encryptZeroSymmetric(self, limbs=None):
        """Encryptor::encryptZeroSymmetric(parms_id) (src/encryptor_cuda.cuh:292-320)"""
        if getattr(self, "sk", None) is None:
            raise RuntimeError("secret key is not set")
        return self._zero(self.sk, True, limbs)


class Decryptor:
    """Decryptor::decrypt (src/decryptor.cpp:115-371), on the CPU; deterministic."""

    def __init__(self, context, secret_key):
        self.context, self.lib = context, context.lib
        self.sk = np.ascontiguousarray(secret_key, dtype=np.uint64)

    def decrypt(self, ct, is_ntt_form=None, correction_factor=1):
        ctx = self.context
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        size, limbs, N = ct.shape
        if is_ntt_form is None:
            is_ntt_form = ctx.scheme == CKKS
        out = np.zeros(limbs * N if ctx.scheme == CKKS else N, dtype=np.uint64)
        capi.check(self.lib, self.lib.troyhip_host_decrypt(ctx.h, _u64p(self.sk), _u64p(ct), size, limbs, int(is_ntt_form), C.c_uint64(correction_factor), _u64p(out)))
        return out.reshape(limbs, N) if ctx.scheme == CKKS else out


class BatchEncoder:
    """BatchEncoder::encode / decode (src/batchencoder.cpp:84-190): the 2 x (N/2) slot matrix modulo t <-> the plaintext polynomial, on the
    host (troyhip_host_batch_encode / _decode)."""

    def __init__(self, context):
        if context.scheme == CKKS:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "unsupported scheme")  # batchencoder.cpp:23-26
        self.context, self.lib = context, context.lib

    def slotCount(self):
        return self.context.N

    def encode(self, values):
        v = np.ascontiguousarray(np.asarray(values, dtype=np.int64) % np.int64(self.context.plain_modulus), dtype=np.uint64)
        if v.size > self.context.N:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "values_matrix size is too large")
        out = np.zeros(self.context.N, dtype=np.uint64)
        capi.check(self.lib, self.lib.troyhip_host_batch_encode(self.context.h, _u64p(v), C.c_uint64(v.size), _u64p(out)))
        return out

    def decode(self, plain):
        p = np.ascontiguousarray(plain, dtype=np.uint64)
        out = np.zeros(self.context.N, dtype=np.uint64)
        capi.check(self.lib, self.lib.troyhip_host_batch_decode(self.context.h, _u64p(p), C.c_uint64(p.size), _u64p(out)))
        return out

    def encodePolynomial(self, values):
        """BatchEncoderCuda::encodePolynomial (src/batchencoder_cuda.cu:124-170): the values ARE the coefficients, modulo t.  Unsigned input keeps
        len(values) coefficients; signed input (any negative value, or a signed dtype) is padded to N, negative v stored as t - |v|"""
        v = np.asarray(values)
        if v.size > self.context.N:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "values_matrix size is too large")
        t = int(self.context.plain_modulus)
        if v.dtype.kind == "i":
            out = np.zeros(self.context.N, dtype=np.uint64)
            out[:v.size] = [(t - (-int(x)) % t) if x < 0 else int(x) % t for x in v.ravel()]
            return out
        return (v.astype(np.uint64).ravel() % np.uint64(t)).astype(np.uint64)

    def decodePolynomial(self, plain, signed=False):
        """decodePolynomial (src/batchencoder_cuda.cu:267-286): min(len(plain), N) coefficients; signed=True: N centred values"""
        p = np.ascontiguousarray(plain, dtype=np.uint64).ravel()[:self.context.N]
        if not signed:
            return p.copy()
        t = int(self.context.plain_modulus)
        out = np.zeros(self.context.N, dtype=np.int64)
        out[:p.size] = [int(x) - t if int(x) > t >> 1 else int(x) for x in p]
        return out
