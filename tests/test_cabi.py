"""The C-ABI library loads and exports every symbol include/troyhip.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from troy_amd import capi

HEADER = os.path.join(ROOT, "include", "troyhip.h")


def header_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(troyhip_[a-z0-9_]+)\s*\(", txt)))


def test_header_matches_binding_list():
    assert header_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_symbol():
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(capi.LIB_PATH)
    for s in header_symbols():
        assert hasattr(lib, s), s
    lib.troyhip_build_info.restype = ctypes.c_char_p
    assert lib.troyhip_build_info() == b"gfx950"


def test_code_object_is_gfx950():
    """the shared object carries a gfx950 code object (and nothing else)"""
    data = open(capi.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in data
    for other in (b"gfx90a", b"gfx942", b"gfx1100"):
        assert b"amdgcn-amd-amdhsa--" + other not in data


def test_product_loader_refuses_emulated_library():
    emul = os.path.join(ROOT, "tests", "emul", "libtroyhip_emul.so")
    if not os.path.exists(emul):
        import subprocess
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    lib = ctypes.CDLL(emul)
    lib.troyhip_build_info.restype = ctypes.c_char_p
    assert lib.troyhip_build_info() != b"gfx950"  # capi.load() only accepts "gfx950" for the product path


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libtroyhip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        capi.load()


def test_not_initialized_error():
    """every call except initialize fails with the reference's message before KernelProvider::initialize
    (src/kernelprovider.cuh:24-27)"""
    lib = ctypes.CDLL(capi.LIB_PATH)
    lib.troyhip_last_error.restype = ctypes.c_char_p
    if lib.troyhip_is_initialized():
        pytest.skip("already initialised in this process")
    p = ctypes.c_void_p()
    assert lib.troyhip_malloc(ctypes.byref(p), ctypes.c_size_t(64)) == capi.NOT_INITIALIZED
    assert lib.troyhip_last_error() == b"KernelProvider not initialized."


def test_host_only_parameter_helpers():
    """CoeffModulus::Create / PlainModulus::Batching need no device"""
    import numpy as np
    lib = ctypes.CDLL(capi.LIB_PATH)
    out = np.zeros(5, dtype=np.uint64)
    bits = (ctypes.c_int * 5)(40, 36, 36, 36, 40)
    assert lib.troyhip_coeff_modulus_create(ctypes.c_uint64(8192), bits, 5, out.ctypes.data_as(ctypes.c_void_p)) == 0
    assert [int(x) for x in out] == [1099510890497, 68718346241, 68718428161, 68719230977, 1099511480321]
    bad = (ctypes.c_int * 1)(61)
    assert lib.troyhip_coeff_modulus_create(ctypes.c_uint64(8192), bad, 1, out.ctypes.data_as(ctypes.c_void_p)) == capi.INVALID_ARGUMENT


def test_null_context_is_an_argument_error():
    """a null context handle is refused like any other bad argument -- INVALID_ARGUMENT and a message, not a fault (host-side entry points: no
    device needed; the device-side ones say NOT_INITIALIZED first, as every call does before KernelProvider::initialize)"""
    import numpy as np
    lib = ctypes.CDLL(capi.LIB_PATH)
    lib.troyhip_last_error.restype = ctypes.c_char_p
    buf = np.zeros(16, dtype=np.uint64)
    p = buf.ctypes.data_as(ctypes.c_void_p)
    z = ctypes.c_uint64(0)
    for name, call in {
        "host_keygen": lambda: lib.troyhip_host_keygen(None, z, z, p, p),
        "host_relin_key": lambda: lib.troyhip_host_relin_key(None, z, z, p, p),
        "host_kswitch_key": lambda: lib.troyhip_host_kswitch_key(None, z, z, p, p, p),
        "host_encrypt_zero": lambda: lib.troyhip_host_encrypt_zero(None, z, z, p, 0, 1, p),
        "host_batch_encode": lambda: lib.troyhip_host_batch_encode(None, p, z, p),
        "context_parms_id": lambda: lib.troyhip_context_parms_id(None, 1, p),
    }.items():
        assert call() == capi.INVALID_ARGUMENT, name
        assert lib.troyhip_last_error() == b"null context", name
