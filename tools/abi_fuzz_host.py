"""The host half of the C ABI against nonsense (no GPU touched: troyhip_context_create_host and the troyhip_host_* / parameter helpers): random schemes,
ring degrees that are not powers of two, moduli that are even, composite, repeated, too large or not NTT-friendly, plain moduli above the coefficient modulus,
levels that do not exist, coefficient counts above N -- every call returns TROYHIP_OK or an error code with a message; contexts that ARE created work.
usage: python tools/abi_fuzz_host.py [calls = 4000] [seed = 1]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from troy_amd import capi  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = C.CDLL(os.environ.get("TROYHIP_LIB", capi.LIB_PATH))
lib.troyhip_last_error.restype = C.c_char_p
rng = np.random.default_rng(seed)
KNOWN = {capi.INVALID_ARGUMENT, capi.LOGIC_ERROR, capi.OUT_OF_RANGE, capi.RUNTIME_ERROR, capi.NOT_INITIALIZED, 0}
U = C.c_uint64
stats = {"create ok": 0, "create refused": 0, "host ok": 0, "host refused": 0, "helper ok": 0, "helper refused": 0}


def check(what, rc):
    if rc not in KNOWN or (rc != 0 and not lib.troyhip_last_error()):
        print("BAD RETURN", what, rc, lib.troyhip_last_error())
        sys.exit(2)
    return rc


def good_primes(N, bits):
    out = np.zeros(len(bits), dtype=np.uint64)
    rc = lib.troyhip_coeff_modulus_create(U(N), (C.c_int * len(bits))(*bits), len(bits), out.ctypes.data_as(C.c_void_p))
    return [int(x) for x in out] if rc == 0 else None


for it in range(calls):
    kind = int(rng.integers(0, 3))
    if kind == 0:  # the parameter helpers
        N = int(rng.choice([0, 1, 2, 3, 8, 64, 100, 4096, 1 << 17, 1 << 18, 1 << 40]))
        bits = [int(b) for b in rng.choice([0, 1, 2, 14, 20, 36, 60, 61, 64, -3], size=int(rng.integers(0, 6)))]
        out = np.zeros(8, dtype=np.uint64)
        rc = check("coeff_modulus_create", lib.troyhip_coeff_modulus_create(U(N), (C.c_int * max(len(bits), 1))(*bits), len(bits), out.ctypes.data_as(C.c_void_p)))
        rc2 = check("plain_modulus_batching", lib.troyhip_plain_modulus_batching(U(N), int(rng.choice([0, 1, 13, 20, 61, 64, -1])), out.ctypes.data_as(C.c_void_p)))
        stats["helper ok" if rc == 0 or rc2 == 0 else "helper refused"] += 1
        continue
    # a context from (mostly) broken parameters
    N = int(rng.choice([0, 1, 2, 3, 8, 64, 64, 64, 100, 128, 1 << 17, 1 << 18]))
    base = good_primes(64 if N not in (8, 64, 128) else N, [36, 36, 40]) or [68718428161, 68718952449, 1099511480321]
    primes = list(base)
    flaw = int(rng.integers(0, 8))
    if flaw == 0: primes[1] = primes[0]                      # repeated
    elif flaw == 1: primes[0] += 1                           # even
    elif flaw == 2: primes[0] = 0
    elif flaw == 3: primes[2] = (1 << 62) + 1                # too large
    elif flaw == 4: primes[1] = 68718428161 * 3 % (1 << 61)  # not a prime / not NTT-friendly
    elif flaw == 5: primes = primes[: int(rng.integers(0, 2))]  # none or one
    t = int(rng.choice([0, 1, 2, 193, 12289, 65537, primes[0] if primes else 5, (1 << 61) - 1, 1 << 63]))
    scheme = int(rng.choice([0, 1, 1, 2, 2, 3, 3, 4, 9, -1]))
    arr = np.array(primes if primes else [0], dtype=np.uint64)
    h = C.c_void_p()
    rc = check("context_create_host", lib.troyhip_context_create_host(scheme, U(N), arr.ctypes.data_as(C.c_void_p), len(primes), U(t), C.byref(h)))
    stats["create ok" if rc == 0 else "create refused"] += 1
    if rc != 0:
        continue
    if os.environ.get("ABI_FUZZ_TRACE"):
        print("created", scheme, N, "flaw", flaw, "t", t, "primes", primes)
    K = len(primes)
    sk, pk = np.zeros(K * N, dtype=np.uint64), np.zeros(2 * K * N, dtype=np.uint64)
    check("host_keygen", lib.troyhip_host_keygen(h, U(1), U(2), sk.ctypes.data_as(C.c_void_p), pk.ctypes.data_as(C.c_void_p)))
    big = np.zeros((2 * (K + 2) * max(N, 1)) * 2 + 64, dtype=np.uint64)
    plain = (rng.integers(0, max(t, 2), size=(K + 2) * N + 8, dtype=np.uint64)).astype(np.uint64)
    P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    for _ in range(6):
        limbs = int(rng.choice([0, 1, K - 1, K - 1, K, K + 1, -1]))
        n = int(rng.choice([0, 1, N - 1, N, N + 1]))
        which = int(rng.integers(0, 7))
        if which == 0: rc = lib.troyhip_host_encrypt(h, U(3), U(4), P(pk), P(plain), U(n), limbs, P(big))
        elif which == 1: rc = lib.troyhip_host_encrypt_symmetric(h, U(3), U(4), P(sk), P(plain), U(n), limbs, P(big))
        elif which == 2: rc = lib.troyhip_host_encrypt_zero(h, U(3), U(4), P(sk), int(rng.integers(0, 2)), limbs, P(big))
        elif which == 3: rc = lib.troyhip_host_decrypt(h, P(sk), P(big), int(rng.choice([0, 1, 2, 3, 17])), limbs, int(rng.integers(0, 2)), U(int(rng.choice([0, 1, 3]))), P(plain))
        elif which == 4: rc = lib.troyhip_host_batch_encode(h, P(plain), U(n), P(big))
        elif which == 5: rc = lib.troyhip_host_galois_key(h, U(5), U(6), P(sk), C.c_uint32(int(rng.choice([0, 1, 3, 4, 2 * N - 1, 2 * N + 1]))), P(np.zeros(max(K - 1, 1) * 2 * K * N + 8, dtype=np.uint64)))
        else: rc = lib.troyhip_context_parms_id(h, limbs, P(big))
        check("host call %d" % which, rc)
        stats["host ok" if rc == 0 else "host refused"] += 1
    lib.troyhip_context_destroy(h)
print(", ".join(f"{k}: {v}" for k, v in stats.items()) + "; no fault")
