"""troy_amd -- MI355X-native (gfx950) homomorphic-encryption evaluator behind the troyn:: API of
lightbulb128/troy.  Product = troy_amd/libtroyhip.so (hand-written HIP kernels + host precompute, C ABI in
include/troyhip.h); this package is the thin host-side mirror used by tests and bench.py.
"""
from . import capi  # noqa: F401
from .api import (BFV, BGV, CKKS, BatchEncoder, Ciphertext, CoeffModulus, Decryptor, DeviceBuffer, Encryptor, Evaluator, GaloisKeys,  # noqa: F401
                  KernelProvider, KeyGenerator, KSwitchKeys, PlainModulus, RelinKeys, SEALContext, synchronize)
