"""C-ABI metadata fuzz (tools/abi_fuzz.py) on the CPU emulator build of the kernel sources: thousands of random, mostly invalid ciphertext descriptors
through the evaluator entry points -- every call returns TROYHIP_OK or an error code with a message, what is accepted was a plausible operand, nothing
faults.  (Round 3: this is how the division by zero of a CKKS add / sub with unequal correction factors was found.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [1, 3, 12])
def test_abi_metadata_fuzz_on_emulator(seed):
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    env = dict(os.environ, TROYHIP_LIB=os.path.join(ROOT, "tests", "emul", "libtroyhip_emul.so"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "abi_fuzz.py"), "3000", str(seed)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "no fault" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("seed", [1, 2])
def test_abi_host_side_fuzz(seed):
    """the host half (troyhip_context_create_host, troyhip_host_*, the parameter helpers: no GPU) against broken parameters -- ring degrees that are
    not powers of two, moduli that are even, composite, repeated or too large, a plain modulus for CKKS, levels that do not exist"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "abi_fuzz_host.py"), "2000", str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "no fault" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_ckks_context_refuses_a_plain_modulus():
    """src/context.cpp:353-361: CKKS parameters with a non-zero plain modulus are invalid (the library used to ignore the value)"""
    import ctypes as C
    import numpy as np
    from troy_amd import capi
    lib = C.CDLL(capi.LIB_PATH)
    lib.troyhip_last_error.restype = C.c_char_p
    out = np.zeros(3, dtype=np.uint64)
    assert lib.troyhip_coeff_modulus_create(C.c_uint64(64), (C.c_int * 3)(40, 36, 40), 3, out.ctypes.data_as(C.c_void_p)) == 0
    h = C.c_void_p()
    assert lib.troyhip_context_create_host(capi.CKKS, C.c_uint64(64), out.ctypes.data_as(C.c_void_p), 3, C.c_uint64(12289), C.byref(h)) == capi.INVALID_ARGUMENT
    assert lib.troyhip_last_error() == b"plain_modulus must be zero"
    assert lib.troyhip_context_create_host(capi.CKKS, C.c_uint64(64), out.ctypes.data_as(C.c_void_p), 3, C.c_uint64(0), C.byref(h)) == 0
    lib.troyhip_context_destroy(h)
