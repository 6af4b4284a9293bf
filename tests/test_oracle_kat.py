"""The CPU oracle against the known-answer vectors of the reference's OWN unit tests
(tests/golden/kat_reference_tests.json, values transcribed from test/utils/*.cpp)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

KAT = json.load(open(os.path.join(GOLDEN, "kat_reference_tests.json")))


def test_ntt_root_powers(oracle_lib):
    k = KAT["ntt_root_powers"]
    p = int(k["modulus"])
    for logn, key in ((1, "logn1"), (2, "logn2")):
        ok, root = oracle_lib.try_minimal_primitive_root(2 << logn, p)
        assert ok
        N = 1 << logn
        for idx, val in k[key].items():
            # root_powers[bitrev(i)] = psi^i
            i = int(format(int(idx), f"0{logn}b")[::-1], 2)
            assert oracle_lib.exponentiate_uint_mod(root, i, p) == int(val)
        assert N


def test_ntt_forward(oracle_lib):
    k = KAT["ntt_forward"]
    p = int(k["modulus"])
    for c in k["cases"]:
        out = oracle_lib.ntt_standalone(k["N"], p, np.array([int(x) for x in c["in"]], dtype=np.uint64), 1)
        assert [int(x) for x in out] == [int(x) for x in c["out"]]


def test_ntt_roundtrip(oracle_lib):
    p = int(KAT["ntt_forward"]["modulus"])
    rng = np.random.default_rng(0)
    x = rng.integers(0, p, 8, dtype=np.uint64)  # test/utils/ntt.cpp:101-132
    y = oracle_lib.ntt_standalone(8, p, x, 1)
    assert np.array_equal(oracle_lib.ntt_standalone(8, p, y, 3), x)
    assert np.array_equal(oracle_lib.ntt_standalone(8, p, np.zeros(8, dtype=np.uint64), 3), np.zeros(8, dtype=np.uint64))


def test_barrett_reduce_128(oracle_lib):
    for p, lo, hi, exp in KAT["barrett_reduce_128"]["cases"]:
        assert oracle_lib.barrett_reduce_128(int(lo), int(hi), int(p)) == int(exp)


def test_multiply_uint_mod(oracle_lib):
    for p, a, b, exp in KAT["multiply_uint_mod"]["cases"]:
        assert oracle_lib.multiply_uint_mod(int(a), int(b), int(p)) == int(exp)


def test_shoup_operand_and_lazy(oracle_lib):
    for p, w, quo in KAT["multiply_uint_mod_operand"]["cases"]:
        assert oracle_lib.shoup_quotient(int(w), int(p)) == int(quo)
    for p, x, w, exp in KAT["multiply_uint_mod_lazy"]["cases"]:
        assert oracle_lib.multiply_uint_mod_lazy(int(x), int(w), int(p)) == int(exp)


def test_dot_product_mod(oracle_lib):
    k = KAT["dot_product_mod"]
    for n, exp in k["cases"]:
        a = np.full(max(n, 1), int(k["a"]), dtype=np.uint64)
        b = np.full(max(n, 1), int(k["b"]), dtype=np.uint64)
        assert oracle_lib.dot_product_mod(a[:n], b[:n], int(k["modulus"])) == int(exp)
    p = oracle_lib.get_primes(2048, 61, 1)[0]  # second half of the reference test: (p-1)^2 summed n times == n
    a = np.full(64, p - 1, dtype=np.uint64)
    for n in (0, 1, 2, 15, 16, 17, 32, 64):
        assert oracle_lib.dot_product_mod(a[:n], a[:n], p) == n


def test_galois(oracle_lib):
    k = KAT["apply_galois"]
    out = oracle_lib.apply_galois(k["N"], k["elt"], int(k["modulus"]), np.array(k["in"], dtype=np.uint64))
    assert [int(x) for x in out] == k["out"]
    k = KAT["apply_galois_ntt"]
    out = oracle_lib.apply_galois_ntt(k["N"], k["elt"], np.array(k["in"], dtype=np.uint64))
    assert [int(x) for x in out] == k["out"]


def test_coeff_modulus_create(oracle_lib):
    k = KAT["coeff_modulus_create"]
    assert oracle_lib.coeff_modulus_create(k["N"], k["bits"]) == [int(x) for x in k["out"]]


def test_naf(oracle_lib):  # src/utils/numth.h:16-36
    assert oracle_lib.naf(5) == [1, 4]
    assert oracle_lib.naf(3) == [-1, 4]
    assert oracle_lib.naf(-5) == [-1, -4]
    assert oracle_lib.naf(7) == [-1, 8]
    assert oracle_lib.naf(0) == []


def test_blake2b_rfc7693_and_parms_id_vs_reference():
    """the product's BLAKE2b (written from RFC 7693) on the RFC's "abc" vector, and parms_id of every level against the
    reference's EncryptionParameters::computeParmsID (src/encryptionparams.cpp:118-146) through oracle/_ref"""
    import ctypes as C
    import os
    import subprocess
    from conftest import ROOT
    from troy_amd import capi
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    lib = capi.load(os.path.join(ROOT, "tests", "emul", "libtroyhip_emul.so"))
    out = (C.c_uint8 * 64)()
    assert lib.troyhip_blake2b(out, C.c_size_t(64), b"abc", C.c_size_t(3)) == 0
    assert bytes(out).hex() == ("ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d1"
                                "7d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923")
    from oracle import ref as R
    if not R.available():
        pytest.skip("oracle/_ref not built")
    for scheme, N, bits, tb in ((1, 4096, [36, 36, 37], 20), (2, 256, [40, 40, 40], 0), (3, 128, [40, 36, 36, 40], 10)):
        primes = R.coeff_modulus_create(N, bits)
        t = R.plain_batching(N, tb) if tb else 0
        r = R.Ref(scheme, N, primes, t)
        arr = np.array(primes, dtype=np.uint64)
        h = C.c_void_p()
        assert lib.troyhip_context_create_host(scheme, C.c_uint64(N), arr.ctypes.data_as(C.c_void_p), len(primes), C.c_uint64(t), C.byref(h)) == 0
        for limbs in range(len(primes), r.chain()[2] - 1, -1):
            o = np.zeros(4, dtype=np.uint64)
            assert lib.troyhip_context_parms_id(h, limbs, o.ctypes.data_as(C.c_void_p)) == 0
            assert [int(x) for x in o] == r.parms_id(limbs), (scheme, limbs)
        lib.troyhip_context_destroy(h)
