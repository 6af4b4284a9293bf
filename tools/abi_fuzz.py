"""C-ABI metadata fuzz (development tool, runs on the CPU emulator build or on the GPU): random -- mostly INVALID -- ciphertext descriptors (size 0 .. 17,
limbs 0 .. K + 1, either form, strides from too small to generous, odd scales and correction factors) through the evaluator entry points.  The buffers
are large enough for anything a descriptor can claim, so every call must come back with TROYHIP_OK or an error code and a message; a fault, a hang or
an unknown code is a bug.  usage: python tools/abi_fuzz.py [calls = 3000] [seed = 1]     (TROYHIP_LIB selects the library; tests run it on the emulator)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import troy_amd as ta  # noqa: E402
from troy_amd import api, capi, synth  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
path = os.environ.get("TROYHIP_LIB")
lib = capi.load(path) if path else capi.load()
api.KernelProvider.initialize(0, _lib=lib)
rng = np.random.default_rng(seed)
N, B = 64, 3
KNOWN = {capi.INVALID_ARGUMENT, capi.LOGIC_ERROR, capi.OUT_OF_RANGE, capi.RUNTIME_ERROR, capi.NOT_INITIALIZED, 0}
stats = {}
for scheme, bits in ((capi.BFV, [40, 36, 36, 40]), (capi.CKKS, [50, 40, 40, 50]), (capi.BGV, [36, 36, 34, 40])):
    primes = ta.CoeffModulus.Create(N, bits)
    K = len(primes)
    ctx = ta.SEALContext(scheme, N, primes, ta.PlainModulus.Batching(N, 14) if scheme != capi.CKKS else 0)
    words = 17 * (K + 1) * N * 2  # per batch item: more than any descriptor below can address
    bufs = [api.DeviceBuffer.from_numpy(synth.uniform_rows(7 + i, [primes[0]], 1, words * B)[0] % np.uint64(primes[-2])) for i in range(3)]
    key = api.DeviceBuffer.from_numpy(synth.uniform_kswitch_key(3, primes, N))
    plain = api.DeviceBuffer.from_numpy(synth.uniform_rows(9, [primes[0]], 1, (K + 1) * N)[0])

    pr = (C.c_uint64 * 3)(*[int(p) for p in primes[:3]])
    bad_pr = (C.c_uint64 * 2)(int(primes[0]), 97)
    elts = (C.c_uint32 * 2)(3, 2 * N - 1)
    keys = (C.c_void_p * 2)(key.ptr, key.ptr)

    def desc(buf):
        size = int(rng.choice([0, 1, 2, 2, 2, 3, 3, 4, 16, 17]))
        limbs = int(rng.choice([0, 1, K - 2, K - 1, K - 1, K - 1, K, K + 1]))
        dense = size * limbs * N
        stride = int(rng.choice([dense, dense, max(dense, 3 * max(limbs, 1) * N), words, 0, max(dense - 1, 0)]))
        if rng.integers(0, 12) == 0:
            return api.CtStruct(None, stride, size, limbs, int(rng.integers(0, 2)), 1.0, 1)  # no storage at all
        return api.CtStruct(buf.ptr, stride, size, limbs, int(rng.integers(0, 2)), float(rng.choice([1.0, 2.0 ** 20, 2.0 ** 40, 0.0, -1.0, float("inf")])),
                            int(rng.choice([1, 1, 3, 0])))

    ops = {
        "add": lambda a, b, o: lib.troyhip_add(ctx.h, C.byref(a), C.byref(b), C.c_uint64(B), None),
        "sub": lambda a, b, o: lib.troyhip_sub(ctx.h, C.byref(a), C.byref(b), C.c_uint64(B), None),
        "negate": lambda a, b, o: lib.troyhip_negate(ctx.h, C.byref(a), C.c_uint64(B), None),
        "multiply": lambda a, b, o: lib.troyhip_multiply(ctx.h, C.byref(a), C.byref(b), C.byref(o), C.c_uint64(B), None),
        "relinearize": lambda a, b, o: lib.troyhip_relinearize(ctx.h, C.byref(a), C.c_void_p(key.ptr), C.c_uint64(B), None),
        "apply_key_switching": lambda a, b, o: lib.troyhip_apply_key_switching(ctx.h, C.byref(a), C.c_void_p(key.ptr), C.c_uint64(B), None),
        "negacyclic_shift": lambda a, b, o: lib.troyhip_negacyclic_shift(ctx.h, C.byref(a), C.c_uint64(int(rng.integers(0, 3 * N))), C.c_uint64(B), None),
        "divide_by_degree": lambda a, b, o: lib.troyhip_divide_by_poly_modulus_degree(ctx.h, C.byref(a), C.c_uint64(int(rng.integers(0, 5))), C.c_uint64(B), None),
        "mod_switch_to_next": lambda a, b, o: lib.troyhip_mod_switch_to_next(ctx.h, C.byref(a), C.byref(o), C.c_uint64(B), None),
        "rescale_to_next": lambda a, b, o: lib.troyhip_rescale_to_next(ctx.h, C.byref(a), C.byref(o), C.c_uint64(B), None),
        "apply_galois": lambda a, b, o: lib.troyhip_apply_galois(ctx.h, C.byref(a), C.c_uint32(int(rng.choice([3, 5, 2 * N - 1, 4, 2 * N + 1, 0]))), C.c_void_p(key.ptr), C.c_uint64(B), None),
        "rotate": lambda a, b, o: lib.troyhip_rotate(ctx.h, C.byref(a), int(rng.integers(-N, N)), int(rng.integers(0, 2)), elts, keys, int(rng.integers(0, 3)), C.c_uint64(B), None),
        "transform_to_ntt": lambda a, b, o: lib.troyhip_transform_to_ntt(ctx.h, C.byref(a), C.c_uint64(B), None),
        "transform_from_ntt": lambda a, b, o: lib.troyhip_transform_from_ntt(ctx.h, C.byref(a), C.c_uint64(B), None),
        "add_plain": lambda a, b, o: lib.troyhip_add_plain(ctx.h, C.byref(a), C.c_void_p(plain.ptr), C.c_uint64(int(rng.choice([0, 1, N - 5, N, N + 1]))), C.c_uint64(int(rng.choice([0, N]))),
                                                           C.c_double(float(rng.choice([1.0, 2.0 ** 20]))), int(rng.integers(0, 2)), C.c_uint64(B), None),
        "multiply_plain": lambda a, b, o: lib.troyhip_multiply_plain(ctx.h, C.byref(a), C.c_void_p(plain.ptr), C.c_uint64(int(rng.choice([0, 1, N - 5, N, N + 1]))), C.c_uint64(int(rng.choice([0, N]))),
                                                                     C.c_uint64(B), None),
        "decrypt": lambda a, b, o: lib.troyhip_decrypt(ctx.h, C.byref(a), C.c_void_p(key.ptr), C.c_void_p(bufs[2].ptr), C.c_uint64(int(rng.choice([0, N, (K + 1) * N]))), C.c_uint64(B), None),
        "switch_key": lambda a, b, o: lib.troyhip_switch_key(ctx.h, C.byref(a), C.c_void_p(bufs[1].ptr), C.c_uint64(int(rng.choice([0, N, K * N, words]))), C.c_void_p(key.ptr), C.c_uint64(B), None),
        "ntt": lambda a, b, o: lib.troyhip_ntt(ctx.h, C.c_void_p(bufs[2].ptr), C.c_uint64(int(rng.choice([0, 1, 2, 3, 6, 7]))), pr, int(rng.choice([0, 1, 2, 3, 65])), int(rng.choice([0, 1, 2, 3])),
                                               int(rng.integers(0, 2)), None),
        "ntt_foreign_prime": lambda a, b, o: lib.troyhip_ntt(ctx.h, C.c_void_p(bufs[2].ptr), C.c_uint64(2), bad_pr, 2, 1, 0, None),
        "relinearize_keys": lambda a, b, o: lib.troyhip_relinearize_keys(ctx.h, C.byref(a), keys, int(rng.choice([-1, 0, 1, 2, 15])), C.c_uint64(B), None),
        "plain_to_ntt": lambda a, b, o: lib.troyhip_plain_to_ntt(ctx.h, C.c_void_p(plain.ptr), C.c_uint64(int(rng.choice([0, 1, N, N + 1]))), C.c_uint64(int(rng.choice([0, N]))),
                                                                 int(rng.choice([-1, 0, 1, K - 1, K, K + 1])), C.c_void_p(bufs[2].ptr), C.c_uint64(int(rng.integers(0, 3))), None),
        "null_operands": lambda a, b, o: lib.troyhip_multiply(ctx.h, None, C.byref(b), C.byref(o), C.c_uint64(B), None) | lib.troyhip_add(ctx.h, C.byref(a), None, C.c_uint64(B), None),
        "multiply_plain_ntt": lambda a, b, o: lib.troyhip_multiply_plain_ntt(ctx.h, C.byref(a), C.c_void_p(plain.ptr), C.c_double(float(rng.choice([1.0, 2.0 ** 30, 0.0]))), C.c_uint64(B), None),
    }
    for _ in range(calls // 3):
        name = list(ops)[int(rng.integers(0, len(ops)))]
        a, b, o = desc(bufs[0]), desc(bufs[1]), desc(bufs[2])
        if os.environ.get("ABI_FUZZ_TRACE"):
            print("call", scheme, name, [(d.size, d.limbs, d.batch_stride, d.is_ntt_form, d.scale, d.correction_factor) for d in (a, b, o)], flush=True)
        rc = ops[name](a, b, o)
        if rc not in KNOWN or (rc != 0 and not lib.troyhip_last_error()):
            print("BAD RETURN", scheme, name, rc, lib.troyhip_last_error())
            sys.exit(2)
        if rc == 0:  # what was accepted must have been a plausible operand: a real level, 1 .. 16 polynomials, items that do not overlap
            used = (a, b) if name in ("add", "sub", "multiply") else (a,)
            if name in ("decrypt", "switch_key", "rotate", "ntt", "ntt_foreign_prime", "plain_to_ntt", "null_operands"):
                used = ()  # their operand rules differ (sizes up to the key powers at hand; a target instead of c1; rotation by 0 steps returns before any check, as the reference does)
            for d in used:
                if not (1 <= d.limbs <= K - 1 and 1 <= d.size <= 16 and d.batch_stride >= d.size * d.limbs * N and d.data):
                    print("ACCEPTED AN IMPLAUSIBLE OPERAND", scheme, name, d.size, d.limbs, d.batch_stride, d.is_ntt_form)
                    sys.exit(3)
        stats[(name, rc)] = stats.get((name, rc), 0) + 1
    api.synchronize()
ok = sum(v for (n, rc), v in stats.items() if rc == 0)
print(f"{sum(stats.values())} calls, {ok} accepted, {sum(stats.values()) - ok} refused with an error code; no fault")
