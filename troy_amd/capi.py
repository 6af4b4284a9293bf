"""ctypes binding of libtroyhip.so (include/troyhip.h) -- the C ABI of the MI355X evaluator.

The library is the gfx950 build produced by troy_amd/csrc/Makefile.  There is no CPU fallback: if the
shared object is missing, or it is not the gfx950 build, importing/using this module fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtroyhip.so")

OK, INVALID_ARGUMENT, LOGIC_ERROR, OUT_OF_RANGE, RUNTIME_ERROR, NOT_INITIALIZED = range(6)
BFV, CKKS, BGV = 1, 2, 3


class TroyHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class InvalidArgument(TroyHipError, ValueError):  # std::invalid_argument
    pass


class LogicError(TroyHipError):  # std::logic_error
    pass


class OutOfRange(TroyHipError, IndexError):  # std::out_of_range
    pass


class NotInitialized(InvalidArgument):  # invalid_argument("KernelProvider not initialized.")
    pass


_EXC = {INVALID_ARGUMENT: InvalidArgument, LOGIC_ERROR: LogicError, OUT_OF_RANGE: OutOfRange,
        RUNTIME_ERROR: TroyHipError, NOT_INITIALIZED: NotInitialized}


class CtStruct(C.Structure):
    _fields_ = [("data", C.c_void_p), ("batch_stride", C.c_uint64), ("size", C.c_int32), ("limbs", C.c_int32),
                ("is_ntt_form", C.c_int32), ("scale", C.c_double), ("correction_factor", C.c_uint64)]


class ContextInfo(C.Structure):
    _fields_ = [("scheme", C.c_int32), ("poly_modulus_degree", C.c_uint64), ("key_limbs", C.c_int32),
                ("first_limbs", C.c_int32), ("last_limbs", C.c_int32), ("plain_modulus", C.c_uint64)]


# every symbol include/troyhip.h declares (tests/test_cabi.py checks the header against this list)
SYMBOLS = [
    "troyhip_initialize", "troyhip_is_initialized", "troyhip_device_count", "troyhip_set_device", "troyhip_get_device", "troyhip_context_device", "troyhip_copy_peer", "troyhip_last_error", "troyhip_build_info", "troyhip_malloc",
    "troyhip_free", "troyhip_pool_release", "troyhip_copy_h2d", "troyhip_copy_d2h", "troyhip_copy_d2d", "troyhip_memset_zero",
    "troyhip_stream_synchronize", "troyhip_stream_create", "troyhip_stream_destroy", "troyhip_stream_register", "troyhip_stream_unregister", "troyhip_mem_info", "troyhip_device_pci_bus_id", "troyhip_timer_create", "troyhip_timer_destroy",
    "troyhip_timer_start", "troyhip_timer_stop", "troyhip_timer_elapsed_ms", "troyhip_coeff_modulus_create",
    "troyhip_plain_modulus_batching", "troyhip_context_create", "troyhip_context_create_host", "troyhip_context_destroy",
    "troyhip_host_keygen", "troyhip_host_relin_key", "troyhip_host_galois_key", "troyhip_host_kswitch_key", "troyhip_host_encrypt_zero", "troyhip_host_encrypt", "troyhip_host_encrypt_symmetric", "troyhip_host_encrypt_symmetric_seeded", "troyhip_host_expand_seed", "troyhip_multiply_plain_accumulate", "troyhip_host_decrypt", "troyhip_context_info",
    "troyhip_context_behz_bases", "troyhip_context_ntt_tables", "troyhip_test_modarith", "troyhip_ktime_enable", "troyhip_ktime_report", "troyhip_blake2b", "troyhip_random_bytes", "troyhip_context_parms_id", "troyhip_context_release_stream", "troyhip_context_reserve_scratch",
    "troyhip_context_scratch_words", "troyhip_galois_elt_from_step", "troyhip_ntt", "troyhip_fill_uniform",
    "troyhip_negate", "troyhip_add", "troyhip_sub", "troyhip_multiply", "troyhip_relinearize", "troyhip_relinearize_keys", "troyhip_relinearize_to", "troyhip_switch_key",
    "troyhip_mod_switch_to_next", "troyhip_rescale_to_next", "troyhip_apply_galois", "troyhip_rotate",
    "troyhip_transform_to_ntt", "troyhip_transform_from_ntt", "troyhip_multiply_plain_ntt", "troyhip_add_plain", "troyhip_multiply_plain",
    "troyhip_stat", "troyhip_build_id", "troyhip_host_batch_encode", "troyhip_host_batch_decode", "troyhip_plain_to_ntt", "troyhip_decrypt", "troyhip_apply_key_switching", "troyhip_negacyclic_shift", "troyhip_divide_by_poly_modulus_degree",
]

_lib = None


def load(path=None):
    """Load libtroyhip.so.  `path` is for the test-suite only (tests/emul build); the package itself always
    loads the in-tree gfx950 library."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # TROYHIP_LIB: development switch for same-box A/B runs of two builds of the library (tools/ntt_probe.sh variants); it must still be
    # a gfx950 build of this library -- the check below applies to it as to the in-tree file
    p = path or os.environ.get("TROYHIP_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(
            f"{p} is missing: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). troy_amd has no CPU fallback.")
    lib = C.CDLL(p)
    lib.troyhip_last_error.restype = C.c_char_p
    lib.troyhip_build_info.restype = C.c_char_p
    lib.troyhip_build_id.restype = C.c_char_p
    if path is None:
        info = lib.troyhip_build_info().decode()
        if info != "gfx950":
            raise ImportError(f"{p} is not the gfx950 build ({info}); refusing to use it as the product library")
        _lib = lib
    return lib


def check(lib, rc):
    if rc != OK:
        msg = lib.troyhip_last_error().decode()
        raise _EXC.get(rc, TroyHipError)(rc, msg)


def stat(name, lib=None):
    """path counter of the library (troyhip_stat): 'ks_fp_launches', 'ks_int_launches', 'ntt1_fp_launches', 'ntt1_int_launches'"""
    lib = lib or load()
    v = C.c_uint64(0)
    check(lib, lib.troyhip_stat(name.encode(), C.byref(v)))
    return int(v.value)


def build_id(lib=None):
    """hash of the sources the loaded library was built from (troyhip_build_id) -- profiles/*_traffic.json are stamped with it"""
    return (lib or load()).troyhip_build_id().decode()
